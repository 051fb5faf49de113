#!/bin/bash
# round 4: kernel-side teardown (between the last line a process prints and the moment its parent sees it gone) of HIP
# processes by what they allocated (tools/exit_cost.hip), and of the CLI itself, back to back and with pauses in between
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r04_exit2; mkdir -p $out build
hipcc -O2 tools/exit_cost.hip -o build/exit_cost 2> $out/build.log || exit 1
one() {
  local t0=$(date +%s.%N); local line=$("$@"); local t1=$(date +%s.%N)
  python3 -c "l='$line'.split(); print('start->exit-call %.3f s (init %s, work %s)  after _exit %.3f s   [$*]' % (float(l[0])-$t0, l[2], l[4], $t1-float(l[0])))"
}
{
for rep in 1 2 3 4; do one build/exit_cost; done
for rep in 1 2 3 4; do one build/exit_cost 0 0 0 0 0; sleep 0.5; done
for rep in 1 2 3; do one build/exit_cost 40; done
for rep in 1 2 3; do one build/exit_cost 0 96; done
for rep in 1 2 3; do one build/exit_cost 0 1024; done
for rep in 1 2 3; do one build/exit_cost 0 0 1024; done
for rep in 1 2 3; do one build/exit_cost 0 0 0 48; done
for rep in 1 2 3; do one build/exit_cost 0 0 0 0 2048; done
for rep in 1 2 3; do one build/exit_cost 40 96 256 48 1024; done
} 2>&1 | tee $out/exit_cost.txt
