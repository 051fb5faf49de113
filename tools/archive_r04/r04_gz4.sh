#!/bin/bash
# round 4: user / system CPU of the CLI on one .fq.gz (cgroup cpu.stat), by mode
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r04_gz4; mkdir -p $out
python - <<'PY' > $out/prep.log 2>&1
import sys
sys.path.insert(0, '.')
import ntsm_amd, bench
s = ntsm_amd.SynthShort(20241218, 96287, read_seed=7, sites_path='/tmp/r04_sites.fa')
s.write_fastq('/tmp/r04.fq', 0, int(4e7), threads=32)
bench.pigz_like('/tmp/r04.fq', '/tmp/r04.fq.gz', threads=48)
PY
stat() { awk '/^user_usec|^system_usec|nr_throttled/ {printf "%s ", $2}' /sys/fs/cgroup/cpu.stat; }
one() {
  local u0=($(stat)); local t0=$(date +%s.%N)
  local line=$(env "$@" build/ntsmCount -s /tmp/r04_sites.fa -t 16 $F 2>&1 >/dev/null | grep -o "Time: [0-9.]* s Memory: [0-9]* kbytes")
  local t1=$(date +%s.%N); local u1=($(stat))
  python3 -c "w=$t1-$t0; print('wall %.3f s  user %.2f s  system %.2f s  throttled periods %d  | $line [$* $F]' % (w, (${u1[0]}-${u0[0]})/1e6, (${u1[1]}-${u0[1]})/1e6, ${u1[2]}-${u0[2]}))"
}
{
F=/tmp/r04.fq.gz
for rep in 1 2 3; do one NTSM_X=1; done
for rep in 1 2 3; do one NTSM_NO_EARLY=1; done
for rep in 1 2; do one NTSM_KEEP_MAPPED=1; done
F=/tmp/r04.fq
for rep in 1 2 3; do one NTSM_X=1; done
} 2>&1 | tee $out/cpu.txt
