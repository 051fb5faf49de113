#!/bin/bash
# tools/mid_study2.sh -- round 5, 2.5 M-key set: 2-D sweep first-level filter size x drain Bloom size of the shipped one-level
# form (the 1-D sweeps of round 4 were flat; tools/sim_mid_set.cpp's cache model prefers a smaller first level with the 1 MiB
# drain Bloom -- measured here).
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r05_mid; mkdir -p $out
specs="0:0"
for blocks in 1536 1792 2048 2304 2560 2816 3072; do for d in 21 22 23 24; do specs="$specs 0:$((2000000+blocks)),$((3000000+d))"; done; done
NTSM_STRESS_READS=${READS:-1e8} NTSM_STRESS_SITES=96287 NTSM_STRESS_SEED=20241218 NTSM_STRESS_MIN_KEEP=13 python3 tools/stress_sweep.py $specs > $out/grid_keep13.jsonl 2> $out/grid_keep13.err
python3 - <<'PY'
import json
rows = [json.loads(l) for l in open("gpurun_out/r05_mid/grid_keep13.jsonl") if l.strip()]
for r in rows: print("   %-22s %7.1f" % (r["spec"], r["gbases_per_s"]))
PY
