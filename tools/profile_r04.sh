#!/bin/bash
# tools/profile_r04.sh -- the round's rocprofv3 evidence, written under gpurun_out/prof_r04/ and summarised into profiles/:
#   profiles/r04_full/      kernel trace + 6 PMC passes of bench.py on configs[1] (tools/profile.sh) -> profiles/r04_traffic.json
#   profiles/r04_stress/    kernel trace + 2 PMC passes of tools/stress_sweep.py on configs[4]        -> profiles/r04_stress_traffic.json
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
bash tools/profile.sh r04_full --steps 3 --warmup 1 --no-cpu-baseline --no-check --other-configs none > gpurun_out/prof_r04_full.txt 2>&1
mkdir -p profiles/r04_full
P=gpurun_out/prof_r04_full
cp $(find $P/trace -name "*kernel_stats.csv" | head -1) profiles/r04_full/kernel_stats.csv
i=0; for d in $P/pmc*/; do i=$((i+1)); f=$(find $d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep -E "Counter_Name|ntsm_count" $f > profiles/r04_full/pmc$i.csv; done
cp gpurun_out/prof_r04_full.txt profiles/r04_full/summary.txt
python3 tools/make_traffic.py $P profiles/r04_traffic.json 1.5e11 > /dev/null
S=$ROOT/gpurun_out/prof_r04_stress; mkdir -p $S
python3 tools/stress_sweep.py 0:0 > $S/rate.jsonl 2> $S/rate.err
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $S/trace -- python3 $ROOT/tools/stress_sweep.py 0:0 > $S/trace.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $S/pmc_tcc -- python3 $ROOT/tools/stress_sweep.py 0:0 > $S/pmc_tcc.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $S/pmc_sq -- python3 $ROOT/tools/stress_sweep.py 0:0 > $S/pmc_sq.log 2>&1
cd $ROOT
mkdir -p profiles/r04_stress
cp $(find $S/trace -name "*kernel_stats.csv" | head -1) profiles/r04_stress/kernel_stats.csv
for g in pmc_tcc pmc_sq; do f=$(find $S/$g -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep -E "Counter_Name|ntsm_count" $f > profiles/r04_stress/$g.csv; done
cp $S/rate.jsonl profiles/r04_stress/rate.jsonl
python3 tools/stress_traffic.py $S profiles/r04_stress_traffic.json > profiles/r04_stress/summary.txt
mkdir -p gpurun_out/r04_profiles; cp -r profiles/r04_full profiles/r04_stress profiles/r04_traffic.json profiles/r04_stress_traffic.json gpurun_out/r04_profiles/
# n10_full (2.5 M site k-mers, bench.py's other_configs.n10_full): the same passes over tools/stress_sweep.py on that set
N=$ROOT/gpurun_out/prof_r04_n10_full; mkdir -p $N
export NTSM_STRESS_SITES=96287 NTSM_STRESS_SEED=20241218 NTSM_STRESS_MIN_KEEP=13
python3 tools/stress_sweep.py 0:0 > $N/rate.jsonl 2> $N/rate.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $N/trace -- python3 $ROOT/tools/stress_sweep.py 0:0 > $N/trace.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $N/pmc_tcc -- python3 $ROOT/tools/stress_sweep.py 0:0 > $N/pmc_tcc.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $N/pmc_sq -- python3 $ROOT/tools/stress_sweep.py 0:0 > $N/pmc_sq.log 2>&1
cd $ROOT
unset NTSM_STRESS_SITES NTSM_STRESS_SEED NTSM_STRESS_MIN_KEEP
mkdir -p profiles/r04_n10_full
cp $(find $N/trace -name "*kernel_stats.csv" | head -1) profiles/r04_n10_full/kernel_stats.csv
for g in pmc_tcc pmc_sq; do f=$(find $N/$g -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep -E "Counter_Name|ntsm_count" $f > profiles/r04_n10_full/$g.csv; done
cp $N/rate.jsonl profiles/r04_n10_full/rate.jsonl
python3 tools/stress_traffic.py $N profiles/r04_n10_full_traffic.json > profiles/r04_n10_full/summary.txt
mkdir -p gpurun_out/r04_profiles; cp -r profiles/r04_n10_full profiles/r04_n10_full_traffic.json gpurun_out/r04_profiles/
tail -30 profiles/r04_full/summary.txt; cat profiles/r04_stress/summary.txt; cat profiles/r04_n10_full/summary.txt
