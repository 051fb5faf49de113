#!/bin/bash
# round 4: decoder-thread sweep and early-ingest A/B of the CLI on one .fq.gz and one .fq (2e7 reads)
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r04_gz2; mkdir -p $out
python - <<'PY' > $out/prep.log 2>&1
import sys, os, time
sys.path.insert(0, '.')
import ntsm_amd, bench
s = ntsm_amd.SynthShort(20241218, 96287, read_seed=7, sites_path='/tmp/r04_sites.fa')
s.write_fastq('/tmp/r04.fq', 0, int(2e7), threads=32)
bench.pigz_like('/tmp/r04.fq', '/tmp/r04.fq.gz', threads=48)
PY
run() { for rep in 1 2 3; do env "$@" NTSM_PHASE_TIMES=1 build/ntsmCount -s /tmp/r04_sites.fa -t 16 $F 2>&1 >/dev/null | grep -E "early|inflate|parse\+count|Time" | sed 's/.*: lanes/lanes/; s/Memory.*//' | tr '\n' ' '; echo "[$* $F]"; done; }
F=/tmp/r04.fq.gz
for d in 8 12 16 20 24 32; do run NTSM_NO_EARLY=1 NTSM_GZ_DECODERS=$d; done 2>&1 | tee $out/decoders.txt
for d in 12 16 20 24; do run NTSM_GZ_DECODERS=$d; done 2>&1 | tee $out/early_gz.txt
F=/tmp/r04.fq
run NTSM_NO_EARLY=1 2>&1 | tee $out/plain.txt
run NTSM_X=1 2>&1 | tee -a $out/plain.txt
