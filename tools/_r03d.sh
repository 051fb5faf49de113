set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/r03d; mkdir -p $OUT; cd $ROOT
python3 tools/stress_sweep.py 0:0 0:2 0:1001792 0:1002048 0:1002304 0:1002560 0:1002048,2 4:28,1002048 4:127,1002048 4:125,1002048 > $OUT/sweep.jsonl 2> $OUT/sweep.err
tail -3 $OUT/sweep.err
cat $OUT/sweep.jsonl | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-22s %-16s two=%d bloom %.2f MiB  %7.2f ms  %6.1f Gb/s' % (d['lib'], d['spec'], d['two_level'], d['bloom_MiB'], d['kernel_ms'], d['gbases_per_s']))"
export TMPDIR=/tmp; cd /tmp
for spec in 0:1002048 0:1002560; do
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/pmc_tcc_$spec -- python3 $ROOT/tools/stress_sweep.py $spec > $OUT/pmc_tcc_$spec.log 2>&1
python3 - $OUT/pmc_tcc_$spec <<'PY'
import csv, glob, os, sys, collections
acc = {}
for p in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    d = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(p)):
        if "ntsm_count" in row["Kernel_Name"]:
            d[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for k, v in d.items():
        vals = sorted(v.values()); acc[k] = vals[len(vals) // 2]
print(sys.argv[1], {k: round(v / 1.5e10, 5) for k, v in acc.items()})
PY
done
