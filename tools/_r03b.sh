mkdir -p gpurun_out/r03b
(time python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "random_reads_vs_oracle_n10 or large_site_set or early_stop_resident or fuzz or golden") > gpurun_out/r03b/pytest.log 2>&1; tail -5 gpurun_out/r03b/pytest.log
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --other-configs stress > gpurun_out/r03b/bench_stress.json 2> gpurun_out/r03b/bench.err; tail -c 600 gpurun_out/r03b/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r03b/bench_stress.json') if l.startswith('{')][-1])
print(d['value']/1e9, d['other_configs']['stress'])
PY
for f in 0 266 265 264 220; do NTSM_STRESS_FLOG=$f NTSM_STRESS_READS=1e8 python tools/config_runs.py stress1 2>&1 | tail -1; done
NTSM_STRESS_KERNEL=2 NTSM_STRESS_READS=1e8 python tools/config_runs.py stress1 2>&1 | tail -1
