#!/bin/bash
# tools/profile_r03.sh -- the round's rocprofv3 evidence, written under gpurun_out/prof_r03/ and summarised into profiles/:
#   profiles/r03_full/      kernel trace + 6 PMC passes of bench.py on configs[1] (tools/profile.sh) -> profiles/r03_traffic.json
#   profiles/r03_stress/    kernel trace + 2 PMC passes of tools/stress_sweep.py on configs[4]        -> profiles/r03_stress_traffic.json
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
bash tools/profile.sh r03_full --steps 3 --warmup 1 --no-cpu-baseline --no-check --other-configs none > gpurun_out/prof_r03_full.txt 2>&1
mkdir -p profiles/r03_full
P=gpurun_out/prof_r03_full
cp $(find $P/trace -name "*kernel_stats.csv" | head -1) profiles/r03_full/kernel_stats.csv
i=0; for d in $P/pmc*/; do i=$((i+1)); f=$(find $d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep -E "Counter_Name|ntsm_count" $f > profiles/r03_full/pmc$i.csv; done
cp gpurun_out/prof_r03_full.txt profiles/r03_full/summary.txt
python3 tools/make_traffic.py $P profiles/r03_traffic.json 1.5e11 > /dev/null
S=$ROOT/gpurun_out/prof_r03_stress; mkdir -p $S
python3 tools/stress_sweep.py 0:0 > $S/rate.jsonl 2> $S/rate.err
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $S/trace -- python3 $ROOT/tools/stress_sweep.py 0:0 > $S/trace.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $S/pmc_tcc -- python3 $ROOT/tools/stress_sweep.py 0:0 > $S/pmc_tcc.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $S/pmc_sq -- python3 $ROOT/tools/stress_sweep.py 0:0 > $S/pmc_sq.log 2>&1
cd $ROOT
mkdir -p profiles/r03_stress
cp $(find $S/trace -name "*kernel_stats.csv" | head -1) profiles/r03_stress/kernel_stats.csv
for g in pmc_tcc pmc_sq; do f=$(find $S/$g -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep -E "Counter_Name|ntsm_count" $f > profiles/r03_stress/$g.csv; done
cp $S/rate.jsonl profiles/r03_stress/rate.jsonl
python3 tools/stress_traffic.py $S profiles/r03_stress_traffic.json > profiles/r03_stress/summary.txt
mkdir -p gpurun_out/r03_profiles; cp -r profiles/r03_full profiles/r03_stress profiles/r03_traffic.json profiles/r03_stress_traffic.json gpurun_out/r03_profiles/
tail -30 profiles/r03_full/summary.txt; cat profiles/r03_stress/summary.txt
