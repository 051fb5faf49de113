#!/bin/bash
# tools/mid_study.sh -- round 5 design study for site sets of 2-3 M k-mers.
# Candidate: "two-level-12" = the two-level kernel form with 12-mer minimizers (make m12: -DNTSM_TWO_M=12): a one-word Bloom
# over the DISTINCT SITE MINIMIZERS in the L2 in front of the one-level kernel's own 128-bit blocks.  A read run whose
# minimizer is no site minimizer stops at the Bloom word; the blocks are then asked by a third of the runs only and may be
# smaller at the same overall pass rate, so that Bloom + blocks + drain Bloom fit the 4 MiB L2 again.
# Sweeps Bloom size x block-filter size x drain Bloom on/off on the 2.5 M-key set (n10_full) and the 1.54 M-key bench set,
# against the shipped one-level form; every configuration must give the same hit count (tools/stress_sweep.py asserts it).
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r05_mid; mkdir -p $out
export NTSM_STRESS_SITES=96287 NTSM_STRESS_SEED=20241218 NTSM_STRESS_READS=${READS:-1e8}
for keep in 13 0; do
  export NTSM_STRESS_MIN_KEEP=$keep
  python3 tools/stress_sweep.py 0:0 2:0 > $out/base_keep$keep.jsonl 2> $out/base_keep$keep.err
  specs="0:0"
  for bloom in ${BLOOMS:-384 512 768 1024}; do
    for blocks in ${BLOCKS:-1024 1536 2048 2560 3072}; do
      for drain in 2 3; do specs="$specs 4:$((1000000+bloom)),$((2000000+blocks)),$drain"; done
    done
  done
  for blocks in 4096 6144 8192; do specs="$specs 4:1000512,$((2000000+blocks)),3 4:1000768,$((2000000+blocks)),3"; done
  NTSM_HIP_LIB=libntsm_hip_m12.so python3 tools/stress_sweep.py $specs > $out/m12_keep$keep.jsonl 2> $out/m12_keep$keep.err
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r05_mid/*.jsonl")):
    rows = [json.loads(l) for l in open(f) if l.strip()]
    print(f)
    for r in sorted(rows, key=lambda r: -r["gbases_per_s"])[:12]:
        print("   %-34s %7.1f Gbases/s  two_level=%s bloom %.2f MiB  keys %d" % (r["spec"], r["gbases_per_s"], r["two_level"], r["bloom_MiB"], r["site_kmers"]))
PY
