#!/bin/bash
# round 4: how the time after _exit of a HIP process depends on the threads that are alive at that moment (tools/exit_cost.hip)
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r04_exit4; mkdir -p $out build
hipcc -O2 tools/exit_cost.hip -o build/exit_cost 2> $out/build.log || exit 1
stat() { awk '/usage_usec/ {printf "%s", $2}' /sys/fs/cgroup/cpu.stat; }
row() {   # args of exit_cost; 8 runs, prints the after-exit times
  local res=""
  for rep in 1 2 3 4 5 6 7 8; do
    local u0=$(stat); local line=$(build/exit_cost "$@"); local t1=$(date +%s.%N); local u1=$(stat)
    res="$res $(python3 -c "l='$line'.split(); print('%.3f/%.2f' % ($t1-float(l[0]), ($u1-$u0)/1e6))")"
  done
  echo "[$*] after-exit s / cgroup cpu s:$res"
}
{
for n in 0 1 2 4 8 16 32 48 96; do row 8 96 64 $n 0 0; done
for n in 4 16 48; do row 8 96 64 $n 0 1; done
for n in 4 16; do row 8 96 64 $n 0 2; done
for n in 0 4 16; do row 8 96 64 $n 0 3; done
for n in 0 4 48; do row 8 96 64 $n 0 4; done
} 2>&1 | tee $out/threads.txt
