#!/bin/bash
# round 4: counters of the count kernel at 1.54 M (hs_n10_like) and 2.50 M (n10_full) site k-mers, same reads, same box
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r04_mid_pmc; mkdir -p $OUT
export TMPDIR=/tmp NTSM_STRESS_SITES=96287 NTSM_STRESS_SEED=20241218 NTSM_STRESS_READS=1e8
cd /tmp
for set in 0 13; do
  export NTSM_STRESS_MIN_KEEP=$set
  python3 $ROOT/tools/stress_sweep.py ${SPEC:-0:0} > $OUT/rate_$set.jsonl 2> $OUT/rate_$set.err
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/tcc_$set -- python3 $ROOT/tools/stress_sweep.py ${SPEC:-0:0} > $OUT/tcc_$set.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq_$set -- python3 $ROOT/tools/stress_sweep.py ${SPEC:-0:0} > $OUT/sq_$set.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_FLAT --output-format csv -d $OUT/sq2_$set -- python3 $ROOT/tools/stress_sweep.py ${SPEC:-0:0} > $OUT/sq2_$set.log 2>&1
done
python3 - $OUT <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
for s in (0, 13):
    r = json.loads([l for l in open(os.path.join(out, "rate_%d.jsonl" % s)) if l.startswith("{")][-1])
    bases = r["reads"] * 150.0
    c = {}
    for g in ("tcc", "sq", "sq2"):
        for p in glob.glob(os.path.join(out, "%s_%d" % (g, s), "**", "*counter_collection.csv"), recursive=True):
            per = collections.defaultdict(lambda: collections.defaultdict(float))
            for row in csv.DictReader(open(p)):
                if "ntsm_count" in row["Kernel_Name"]:
                    per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
            for k, v in per.items():
                vals = sorted(v.values()); c[k] = vals[len(vals) // 2]
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8.0
    doc = {"site_kmers": r["site_kmers"], "gbases_per_s": r["gbases_per_s"], "kernel_ms": r["kernel_ms"], "hits_per_read": r["hits_per_pass"] / r["reads"],
           "l2_req_per_base": c.get("TCC_REQ_sum", 0) / bases, "l2_miss_per_base": c.get("TCC_MISS_sum", 0) / bases, "fabric_rd_per_base": c.get("TCC_EA0_RDREQ_sum", 0) / bases,
           "valu_per_position": c.get("SQ_INSTS_VALU", 0) * 64 / (bases * 151 / 150), "salu_per_position": c.get("SQ_INSTS_SALU", 0) * 64 / (bases * 151 / 150),
           "lds_per_position": c.get("SQ_INSTS_LDS", 0) * 64 / (bases * 151 / 150), "vmem_rd_per_position": c.get("SQ_INSTS_VMEM_RD", 0) * 64 / (bases * 151 / 150),
           "valu_busy": c.get("SQ_INSTS_VALU", 0) * 4.2 / (1024.0 * cyc) if cyc else None,
           "l2_req_rate_frac_of_cap": c.get("TCC_REQ_sum", 0) / (cyc / 2.4e9) / 266e9 if cyc else None, "raw": c}
    print(json.dumps(doc))
PY
