#!/bin/bash
# tools/profile_round.sh <rNN> -- the round's rocprofv3 evidence in one parametrised run (replaces profile_r03.sh / profile_r04.sh):
#   profiles/<rNN>_full/      kernel trace + PMC passes of bench.py on configs[1] (tools/profile.sh)   -> profiles/<rNN>_traffic.json
#   profiles/<rNN>_stress/    kernel trace + PMC passes of tools/stress_sweep.py on configs[4]          -> profiles/<rNN>_stress_traffic.json
#   profiles/<rNN>_n10_full/  the same on the 2.5 M-key set                                             -> profiles/<rNN>_n10_full_traffic.json
# Every set also gets the memory-side passes of round 5: fabric read requests by size (32 / 64 / 128 B),
# the share "destined for DRAM (MC)" and the average fabric read latency (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ).  gfx950's
# rocprofv3 exposes no Infinity-Cache (MALL) hit / miss and no HBM (UMC / DF) counter -- profiles/r05_counters/ holds the full
# list -- so the split between Infinity Cache and HBM is argued from sizes, not counted.
# One counter group per pass, never combined with trace domains; the profiled program stands directly behind `--`.
set -u
R=${1:?usage: profile_round.sh rNN}
READS=${NTSM_PROFILE_SET_READS:-1e9}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
EA1="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"
EA2="TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"
EA3="TCC_EA0_RD_UNCACHED_32B_sum TCC_EA0_WRREQ_ATOMIC_DRAM_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum"
TCC="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY"

# ---- configs[1] through bench.py
bash tools/profile.sh ${R}_full --steps 3 --warmup 1 --no-cpu-baseline --no-check --other-configs none > gpurun_out/prof_${R}_full.txt 2>&1
P=gpurun_out/prof_${R}_full
export TMPDIR=/tmp
i=6
for grp in "$EA1" "$EA2" "$EA3"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $grp --output-format csv -d "$ROOT/$P/pmc$i" -- python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-check --other-configs none > "$ROOT/$P/pmc$i.log" 2>&1)
done
mkdir -p profiles/${R}_full
cp $(find $P/trace -name "*kernel_stats.csv" | head -1) profiles/${R}_full/kernel_stats.csv
i=0; for d in $(ls -d $P/pmc*/ | sort -V); do i=$((i+1)); f=$(find $d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep -E "Counter_Name|ntsm_count" $f > profiles/${R}_full/pmc$i.csv; done
cp gpurun_out/prof_${R}_full.txt profiles/${R}_full/summary.txt
python3 tools/make_traffic.py $P profiles/${R}_traffic.json 1.5e11 > /dev/null

# ---- configs[4] and the 2.5 M-key set through tools/stress_sweep.py
one_set() {   # <name> <env assignments...>
  local name=$1; shift
  local S=$ROOT/gpurun_out/prof_${R}_$name; mkdir -p $S
  env "$@" NTSM_STRESS_READS=$READS python3 tools/stress_sweep.py 0:0 > $S/rate.jsonl 2> $S/rate.err
  for pass in "trace:--kernel-trace --stats" "pmc_tcc:--pmc $TCC" "pmc_sq:--pmc $SQ" "pmc_ea1:--pmc $EA1" "pmc_ea2:--pmc $EA2" "pmc_ea3:--pmc $EA3"; do
    local tag=${pass%%:*} opts=${pass#*:}
    (cd /tmp && export "$@" NTSM_STRESS_READS=$READS && rocprofv3 $opts --output-format csv -d $S/$tag -- python3 $ROOT/tools/stress_sweep.py 0:0 > $S/$tag.log 2>&1)
  done
  mkdir -p profiles/${R}_$name
  cp $(find $S/trace -name "*kernel_stats.csv" | head -1) profiles/${R}_$name/kernel_stats.csv
  for g in pmc_tcc pmc_sq pmc_ea1 pmc_ea2 pmc_ea3; do f=$(find $S/$g -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep -E "Counter_Name|ntsm_count" $f > profiles/${R}_$name/$g.csv; done
  cp $S/rate.jsonl profiles/${R}_$name/rate.jsonl
  python3 tools/stress_traffic.py $S profiles/${R}_${name}_traffic.json > profiles/${R}_$name/summary.txt
}
one_set stress NTSM_STRESS_SITES=1e6
one_set n10_full NTSM_STRESS_SITES=96287 NTSM_STRESS_SEED=20241218 NTSM_STRESS_MIN_KEEP=13
python3 tools/memory_side.py $R > profiles/${R}_memory_side.txt
mkdir -p gpurun_out/${R}_profiles; cp -r profiles/${R}_full profiles/${R}_stress profiles/${R}_n10_full profiles/${R}_*traffic.json profiles/${R}_memory_side.* gpurun_out/${R}_profiles/ 2>/dev/null
tail -20 profiles/${R}_full/summary.txt; cat profiles/${R}_stress/summary.txt; cat profiles/${R}_n10_full/summary.txt; cat profiles/${R}_memory_side.txt
