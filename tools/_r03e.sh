set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/r03e; mkdir -p $OUT; cd $ROOT
for ns in 1.2e5 1.6e5 2.5e5 5e5; do
NTSM_STRESS_SITES=$ns python3 tools/stress_sweep.py 2:0 4:0 > $OUT/sweep_$ns.jsonl 2> $OUT/sweep_$ns.err
done
python3 tools/stress_sweep.py 0:0 > $OUT/sweep_1e6.jsonl 2> $OUT/sweep_1e6.err
cat $OUT/sweep_*.jsonl | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%9d kmers %-10s two=%d bloom %.2f MiB  %7.2f ms  %6.1f Gb/s' % (d['site_kmers'], d['spec'], d['two_level'], d['bloom_MiB'], d['kernel_ms'], d['gbases_per_s']))"
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "random_reads_vs_oracle_n10 or large_site_set or early_stop_resident or fuzz" 2>&1 | tail -3
