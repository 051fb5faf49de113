#!/bin/bash
# tools/memkind_study.sh -- round 5: what a miss moves.  profiles/r05_memory_side.txt: every L2 miss of the count kernel is a
# 128-byte fabric read, of which a filter block uses 16 bytes and a key bucket 32; on configs[4] that is 17.8 B/base = 7.2 TB/s of
# fabric traffic at 404 Gbases/s.  Does memory allocated fine-grained / uncached (hipExtMallocWithFlags) make the L2 ask for
# less per miss, and keep those lines out of the L2 that holds the Bloom / the filter?  ntsm_set_tuning(4000000 + v):
# v & 3 = kind of the block filter, v >> 2 = kind of the key table (0 ordinary, 1 fine-grained, 3 uncached).
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r05_memkind; mkdir -p $out
specs="0:0 0:4000001 0:4000003 0:4000004 0:4000012 0:4000005 0:4000015"
NTSM_STRESS_READS=${READS:-1e8} NTSM_STRESS_SITES=1e6 python3 tools/stress_sweep.py $specs > $out/stress.jsonl 2> $out/stress.err
NTSM_STRESS_READS=${READS:-1e8} NTSM_STRESS_SITES=96287 NTSM_STRESS_SEED=20241218 NTSM_STRESS_MIN_KEEP=13 python3 tools/stress_sweep.py $specs > $out/n10_full.jsonl 2> $out/n10_full.err
NTSM_STRESS_READS=${READS:-1e8} NTSM_STRESS_SITES=96287 NTSM_STRESS_SEED=20241218 NTSM_STRESS_MIN_KEEP=0 python3 tools/stress_sweep.py $specs > $out/bench.jsonl 2> $out/bench.err
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_memkind/*.jsonl")):
    print(f)
    for l in open(f):
        if l.strip():
            r = json.loads(l); print("   %-12s %7.1f Gbases/s  (%d keys)" % (r["spec"], r["gbases_per_s"], r["site_kmers"]))
for f in sorted(glob.glob("gpurun_out/r05_memkind/*.err")):
    t = open(f).read().strip()
    if t: print(f, t[-400:])
PY
