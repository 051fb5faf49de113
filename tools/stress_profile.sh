#!/bin/bash
# tools/stress_profile.sh -- BASELINE.json configs[4] (1,000,000 sites = 16 M k-mers): rate and L2 counters of the count kernel
# for first-level filters from L2-resident (3 MiB) to Infinity-Cache-resident (24 / 32 MiB) -- the "LDS-vs-HBM crossover"
# the north star asks for is, on this chip, L2 versus Infinity Cache.  Output: gpurun_out/r02_stress/
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r02_stress
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for f in 123 25 124 26 125 27 126 28; do      # log2(bits): 123 = 3 MiB, 25 = 4, 124 = 6, 26 = 8, 125 = 12, 27 = 16, 126 = 24 (auto), 28 = 32 MiB
  NTSM_STRESS_FLOG=$f NTSM_STRESS_READS=${NTSM_STRESS_READS:-1e8} python3 "$ROOT/tools/config_runs.py" stress1 > "$OUT/rate_f$f.json" 2> "$OUT/rate_f$f.err"
  NTSM_STRESS_FLOG=$f NTSM_STRESS_READS=${NTSM_STRESS_READS:-1e8} rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d "$OUT/pmc_f$f" -- python3 "$ROOT/tools/config_runs.py" stress1 > "$OUT/pmc_f$f.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
print("filter_log2  MiB   kernel_ms  Gbases/s  TCC_REQ/base  TCC_MISS/base  L2_hit  EA_RDREQ/base  queued%")
for f in (123, 25, 124, 26, 125, 27, 126, 28):
    try:
        r = json.loads([l for l in open(os.path.join(out, "rate_f%d.json" % f)) if l.startswith("{")][-1])
    except Exception as e:
        print(f, "no rate", e); continue
    acc = collections.defaultdict(float); n = collections.defaultdict(int)
    for p in glob.glob(os.path.join(out, "pmc_f%d" % f, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(p)):
            if "ntsm_count" in row["Kernel_Name"]:
                acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
    bases = r["reads"] * 150.0
    per = {k: acc[k] / max(n[k], 1) for k in acc}          # per dispatch = per pass
    req, miss, ea = per.get("TCC_REQ_sum", 0), per.get("TCC_MISS_sum", 0), per.get("TCC_EA0_RDREQ_sum", 0)
    mib = (3 << (f - 100)) / 8 / 2 ** 20 if f >= 100 else (1 << f) / 8 / 2 ** 20
    print("%-11s %5.1f %9.2f %9.1f %13.4f %14.4f %7.3f %14.4f" % (f, mib, r["kernel_ms"], r["gbases_per_s"], req / bases, miss / bases,
          1 - miss / max(req, 1), ea / bases))
PY
