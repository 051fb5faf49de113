#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
ROOT=$PWD; out=$ROOT/gpurun_out/r05_run_pmc; mkdir -p $out; export TMPDIR=/tmp
V=${V:-0,5}
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
           "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_WAVES" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && rocprofv3 --pmc $grp --output-format csv -d $out/pmc$i -- python3 $ROOT/tools/run_kernel_check.py ${READS:-1e8} $V > $out/pmc$i.log 2>&1)
done
python3 - $out <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(out, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "ntsm_count" in k:
            name = "run" if "run_kernel" in k else "mz"
            per[(name, r["Counter_Name"])][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for (name, c), d in per.items():
        vals = sorted(d.values()); acc[name][c] = vals[len(vals) // 2]
bases = 1.5e10
for name in acc:
    print("==", name)
    for c, v in sorted(acc[name].items()): print("   %-28s %.4g   per base %.4f" % (c, v, v / bases * (64 if c.startswith("SQ_INSTS") else 1)))
PY
