#!/bin/bash
# round 4: early ingest with a pre-populated chunk pool, A/B (4e7 reads like the bench's CLI legs)
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r04_early2; mkdir -p $out
python - <<'PY' > $out/prep.log 2>&1
import sys, os, time
sys.path.insert(0, '.')
import ntsm_amd, bench
s = ntsm_amd.SynthShort(20241218, 96287, read_seed=7, sites_path='/tmp/r04_sites.fa')
s.write_fastq('/tmp/r04.fq', 0, int(4e7), threads=32)
bench.pigz_like('/tmp/r04.fq', '/tmp/r04.fq.gz', threads=48)
PY
run() { for rep in 1 2 3; do env "$@" NTSM_PHASE_TIMES=1 build/ntsmCount -s /tmp/r04_sites.fa -t 16 $F 2>&1 >/dev/null | grep -E "early|inflate|parse\+count|Time|first GPU" | sed 's/.*: lanes/lanes/; s/.*early ingest/early ingest/; s/Memory.*//; s/.*sites loaded + first GPU context/ctx/' | tr '\n' ' '; echo "[$* $F]"; done; }
F=/tmp/r04.fq
run NTSM_NO_EARLY=1 2>&1 | tee $out/plain.txt
run NTSM_EARLY=all 2>&1 | tee -a $out/plain.txt
run NTSM_EARLY=all NTSM_EARLY_NO_POPULATE=1 2>&1 | tee -a $out/plain.txt
F=/tmp/r04.fq.gz
run NTSM_NO_EARLY=1 2>&1 | tee $out/gz.txt
run NTSM_X=1 2>&1 | tee -a $out/gz.txt
run NTSM_EARLY_NO_POPULATE=1 2>&1 | tee -a $out/gz.txt
run NTSM_GZ_DECODERS=16 2>&1 | tee -a $out/gz.txt
