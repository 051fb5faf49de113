#!/bin/bash
# tools/window_family.sh [reads] -- the automatic form choice (tables.cpp: 1.8 M <= keys < 9 M -> run-anchored kernel) on a SECOND
# synthetic family (the window constants were fitted on sites with all 13 k-mers kept).  Here: 3 .. 13 k-mers
# kept per allele (the bench set's structure), site counts from 1.5 M to 12 M keys; per size the automatic choice (0) beside the
# forced one-level (2), two-level (4) and run-anchored (5) forms.  One JSON line per configuration -> gpurun_out/window_family.jsonl
cd "$(dirname "$0")/.." || exit 1
R=${1:-1e8}
out=gpurun_out/window_family.jsonl; mkdir -p gpurun_out; : > $out
for sites in 96287 115000 135000 160000 210000 270000 380000 560000 750000; do
  NTSM_STRESS_SITES=$sites NTSM_STRESS_SEED=20241218 NTSM_STRESS_MIN_KEEP=0 NTSM_STRESS_READS=$R python3 tools/stress_sweep.py 0:0 2:0 4:0 5:0 >> $out 2>> gpurun_out/window_family.err
done
python3 - $out <<'PY'
import json, sys, collections
rows = collections.OrderedDict()
for l in open(sys.argv[1]):
    d = json.loads(l)
    rows.setdefault(d["site_kmers"], {})[d["spec"]] = d
print("%10s %10s %8s %8s %8s %8s  %s" % ("keys", "auto form", "auto", "1-level", "2-level", "run", "best"))
for k, r in rows.items():
    a = r["0:0"]
    form = "run" if a.get("run_form") else "two-level" if a["two_level"] else "one-level"
    g = {s: r[s]["gbases_per_s"] for s in r}
    best = max(("2:0", "4:0", "5:0"), key=lambda s: g.get(s, 0))
    print("%10d %10s %8.0f %8.0f %8.0f %8.0f  %s%s" % (k, form, g["0:0"], g.get("2:0", 0), g.get("4:0", 0), g.get("5:0", 0), {"2:0": "one-level", "4:0": "two-level", "5:0": "run"}[best],
          "" if g["0:0"] >= 0.97 * g[best] else "   <-- auto is %.0f %% below" % (100 * (1 - g["0:0"] / g[best]))))
PY
