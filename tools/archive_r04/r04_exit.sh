#!/bin/bash
# round 4: what the process costs AFTER `Time:` is printed (address-space teardown) on one .fq.gz of the bench's CLI size,
# with the compressed input's pages released by the decoders as they go (default) and left mapped (NTSM_KEEP_MAPPED=1)
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r04_exit; mkdir -p $out
python - <<'PY' > $out/prep.log 2>&1
import sys, os, time
sys.path.insert(0, '.')
import ntsm_amd, bench
s = ntsm_amd.SynthShort(20241218, 96287, read_seed=7, sites_path='/tmp/r04_sites.fa')
s.write_fastq('/tmp/r04.fq', 0, int(4e7), threads=32)
bench.pigz_like('/tmp/r04.fq', '/tmp/r04.fq.gz', threads=48)
PY
F=/tmp/r04.fq.gz
one() {
  local t0=$(date +%s.%N)
  local line=$(env "$@" build/ntsmCount -s /tmp/r04_sites.fa -t 16 $F 2>&1 >/dev/null | grep -o "Time: [0-9.]* s Memory: [0-9]* kbytes")
  local t1=$(date +%s.%N)
  python3 -c "import sys; w=$t1-$t0; t=float('$line'.split()[1]); print('wall %.3f s  Time: %.3f s  after+before %.3f s  rss %s kB  -> %.2f Gbases/s  [$*]' % (w, t, w-t, '$line'.split()[4], 6.0/w))"
}
for rep in 1 2 3 4 5; do
  one NTSM_KEEP_MAPPED=1
  one NTSM_X=1
  one NTSM_X=1 NTSM_GZ_DECODERS=10
  one NTSM_X=1 NTSM_GZ_DECODERS=14
done 2>&1 | tee $out/gz.txt
F=/tmp/r04.fq
for rep in 1 2 3; do one NTSM_X=1; done 2>&1 | tee $out/plain.txt
