#!/bin/bash
# tools/run_kernel_study.sh [sizes|pmc|ab LIB...] -- the measurements behind DESIGN.md section 4.2d (run-anchored kernel, round 5).
#   sizes  automatic minimizer-blocked form against the run-anchored kernel (5:0, and with 2 / 4 MiB filters) for site sets of
#          1.56 M ... 8.3 M k-mers (96287-site geometry, all 13 k-mers of a window kept, scaled by the number of sites)
#   pmc    rocprofv3 --pmc passes (SQ / TCC / GRBM, one group per pass) of the run-anchored kernel on the 2.5 M-key set
#   icache instruction-cache counters (SQC_ICACHE_*, SQ_IFETCH*) of both kernels on the 2.5 M-key set
#   ab     the 2.5 M-key set through variant 5 of several builds of the library (make xlib XNAME=.. XFLAGS=..), twice, same box
cd "$(dirname "$0")/.." || exit 1
ROOT=$PWD
mode=${1:-sizes}; shift || true
case $mode in
sizes)
  for spec in "60000 13" "77000 13" "96287 13" "130000 13" "160000 13" "220000 13" "320000 13"; do
    set -- $spec
    NTSM_STRESS_READS=1e8 NTSM_STRESS_SITES=$1 NTSM_STRESS_SEED=20241218 NTSM_STRESS_MIN_KEEP=$2 python3 tools/stress_sweep.py 2:0 4:0 5:0 5:2002048 5:2004096 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   keys %8d  %-10s %7.1f Gbases/s two_level=%s' % (d['site_kmers'], d['spec'], d['gbases_per_s'], d['two_level']))"
  done;;
pmc)
  out=$ROOT/gpurun_out/r05_run_pmc; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
  export NTSM_STRESS_READS=1e8 NTSM_STRESS_SITES=96287 NTSM_STRESS_SEED=20241218 NTSM_STRESS_MIN_KEEP=13
  i=0
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
             "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    (cd /tmp && rocprofv3 --pmc $grp --output-format csv -d $out/pmc$i -- python3 $ROOT/tools/stress_sweep.py 5:0 > $out/pmc$i.log 2>&1)
  done
  python3 - $out <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]; acc = {}
for f in sorted(glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "ntsm_count_run" in r.get("Kernel_Name", ""): per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for c, d in per.items():
        vals = sorted(d.values()); acc[c] = vals[len(vals) // 2]
bases = 1.5e10
for c, v in sorted(acc.items()): print("   %-28s %.4g   per base %.4f" % (c, v, v / bases * (64 if c.startswith("SQ_INSTS") else 1)))
if "GRBM_GUI_ACTIVE" in acc and "SQ_INSTS_VALU" in acc:
    print("   VALU busy (x 4.2 cycles / 1024 SIMDs): %.3f" % (acc["SQ_INSTS_VALU"] * 4.2 / (1024 * acc["GRBM_GUI_ACTIVE"] / 8)))
PY
  ;;
icache)
  # instruction-cache counters of the run-anchored kernel (5:0) and of the minimizer-blocked one (2:0) on the 2.5 M-key set
  out=$ROOT/gpurun_out/r05_run_icache; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
  export NTSM_STRESS_READS=1e8 NTSM_STRESS_SITES=96287 NTSM_STRESS_SEED=20241218 NTSM_STRESS_MIN_KEEP=13
  for spec in 5:0 2:0; do
    i=0
    for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
      i=$((i+1))
      (cd /tmp && rocprofv3 --pmc $grp --output-format csv -d $out/${spec/:/_}_$i -- python3 $ROOT/tools/stress_sweep.py $spec > $out/${spec/:/_}_$i.log 2>&1)
    done
    echo "   spec $spec"
    python3 - $out ${spec/:/_} <<'PY'
import csv, glob, os, sys, collections
out, tag = sys.argv[1], sys.argv[2]; acc = {}
for f in sorted(glob.glob(os.path.join(out, tag + "_*", "**", "*counter_collection.csv"), recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "ntsm_count_" in r.get("Kernel_Name", ""): per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for c, d in per.items():
        vals = sorted(d.values()); acc[c] = vals[len(vals) // 2]
for c, v in sorted(acc.items()): print("      %-28s %.4g   per wave-position %.4f" % (c, v, v / (1.5e10 / 64)))
PY
  done;;
ab)
  export NTSM_STRESS_READS=1e8 NTSM_STRESS_SITES=96287 NTSM_STRESS_SEED=20241218 NTSM_STRESS_MIN_KEEP=13
  for rep in 1 2; do for lib in libntsm_hip.so "$@"; do
    NTSM_HIP_LIB=$lib python3 tools/stress_sweep.py 5:0 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   %-28s %7.1f Gbases/s hits %d' % ('$lib', d['gbases_per_s'], d['hits_per_pass']))"
  done; done;;
*) echo "usage: run_kernel_study.sh sizes|pmc|icache|ab LIB..."; exit 2;;
esac
