#!/bin/bash
# tools/long_profile.sh -- BASELINE.json configs[2]: 5e6 ONT-like reads (N50 ~ 21 kb, 5 % errors) resident, one full pass and the
# exact -m 10 early stop; rocprofv3 kernel statistics of the same run.  Output: gpurun_out/r02_long/
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/r02_long
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
python3 "$ROOT/tools/config_runs.py" long > "$OUT/run.json" 2> "$OUT/run.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/tools/config_runs.py" long > "$OUT/trace.log" 2>&1
cat "$OUT/run.json"
cat "$OUT"/trace/*/*kernel_stats.csv | cut -c1-220
