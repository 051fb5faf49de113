#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
ROOT=$PWD; out=$ROOT/gpurun_out/r05_run_pmc2; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
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
out = sys.argv[1]
acc = {}
for f in sorted(glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        if "ntsm_count_run" in r.get("Kernel_Name", ""):
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for c, d in per.items():
        vals = sorted(d.values()); acc[c] = vals[len(vals) // 2]
bases = 1.5e10
for c, v in sorted(acc.items()): print("   %-28s %.4g   per base %.4f" % (c, v, v / bases * (64 if c.startswith("SQ_INSTS") else 1)))
if "GRBM_GUI_ACTIVE" in acc and "SQ_INSTS_VALU" in acc:
    cyc = acc["GRBM_GUI_ACTIVE"] / 8
    print("   VALU busy (x4.2 cycles / 1024 SIMDs): %.3f" % (acc["SQ_INSTS_VALU"] * 4.2 / (1024 * cyc)))
PY
