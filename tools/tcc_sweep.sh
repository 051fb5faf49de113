cd /tmp; export TMPDIR=/tmp
for f in 0 123 25 122; do
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d /root/repo/gpurun_out/tcc_f$f -- python3 /root/repo/bench.py --reads 1e8 --steps 2 --warmup 1 --no-cpu-baseline --filter-log2 $f > /root/repo/gpurun_out/tcc_f$f.log 2>&1
  python3 - $f <<'PY'
import csv,glob,sys,collections,json
f=sys.argv[1]
acc=collections.defaultdict(float); n=collections.defaultdict(int)
for p in glob.glob('/root/repo/gpurun_out/tcc_f%s/**/*counter_collection.csv'%f, recursive=True):
    for r in csv.DictReader(open(p)):
        if 'ntsm_count' in r['Kernel_Name']:
            acc[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
ms=None
for l in open('/root/repo/gpurun_out/tcc_f%s.log'%f):
    if l.startswith('{'): ms=json.loads(l)['roofline']['avg_launch_ms']
print('f=%s'%f, 'ms=%.2f'%ms, {k: '%.3g'%(acc[k]/max(n[k],1)) for k in sorted(acc)})
PY
done
