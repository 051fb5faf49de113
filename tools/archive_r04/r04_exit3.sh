#!/bin/bash
# round 4: the CLI's time after `Time:` with the teardown inside exit(2) (NTSM_SYNC_EXIT=1) and handed to the CLONE_VM child
# (default), six runs back to back each (the next run's start-up meets the previous run's teardown), plain and .gz input
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r04_exit3; mkdir -p $out
python - <<'PY' > $out/prep.log 2>&1
import sys
sys.path.insert(0, '.')
import ntsm_amd, bench
s = ntsm_amd.SynthShort(20241218, 96287, read_seed=7, sites_path='/tmp/r04_sites.fa')
s.write_fastq('/tmp/r04.fq', 0, int(4e7), threads=32)
bench.pigz_like('/tmp/r04.fq', '/tmp/r04.fq.gz', threads=48)
PY
one() {
  local t0=$(date +%s.%N)
  local line=$(env "$@" build/ntsmCount -s /tmp/r04_sites.fa -t 16 $F 2>&1 >$out/counts.$$ | grep -o "Time: [0-9.]* s")
  local t1=$(date +%s.%N)
  python3 -c "w=$t1-$t0; t=float('$line'.split()[1]); print('wall %.3f s  Time: %.3f s  rest %.3f s  %s [$* $F]' % (w, t, w-t, '$(sha256sum < $out/counts.$$ | cut -c1-12)'))"
}
{
for F in /tmp/r04.fq /tmp/r04.fq.gz; do
for mode in NTSM_SYNC_EXIT=1 NTSM_X=1 NTSM_SYNC_EXIT=1 NTSM_X=1; do
  s0=$(date +%s.%N)
  for rep in 1 2 3 4 5 6; do one $mode; done
  s1=$(date +%s.%N)
  python3 -c "print('   6 runs back to back, $mode: %.3f s' % ($s1-$s0))"
done
done
sleep 1; echo "processes left:"; ps -eo pid,ppid,stat,comm | grep -i ntsm || echo none
} 2>&1 | tee $out/runs.txt
rm -f $out/counts.*
