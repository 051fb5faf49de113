#!/bin/bash
# round 4, step 1: where does the time go at 2.5 M site k-mers (n10_full: 96287 sites x 26 k-mers)?
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r04_mid; mkdir -p $out
export NTSM_STRESS_SITES=96287 NTSM_STRESS_SEED=20241218 NTSM_STRESS_MIN_KEEP=13 NTSM_STRESS_READS=1e8
python tools/stress_sweep.py 0:0 0:2002560 0:2002816 0:2003328 0:2003584 0:2004096 0:3000022 0:3000024 0:3000025 4:0 4:1001024 4:1001536 4:1003072 > $out/full.jsonl 2> $out/full.err
NTSM_STRESS_P_EMBED=0.06 python tools/stress_sweep.py 0:0 4:0 > $out/full_p06.jsonl 2>> $out/full.err
NTSM_STRESS_P_EMBED=0.0 python tools/stress_sweep.py 0:0 4:0 > $out/full_p00.jsonl 2>> $out/full.err
# the bench set for comparison on the same box
NTSM_STRESS_MIN_KEEP=0 python tools/stress_sweep.py 0:0 4:0 > $out/n10.jsonl 2>> $out/full.err
NTSM_STRESS_MIN_KEEP=0 NTSM_STRESS_P_EMBED=0.167 python tools/stress_sweep.py 0:0 > $out/n10_p167.jsonl 2>> $out/full.err
cat $out/*.jsonl
