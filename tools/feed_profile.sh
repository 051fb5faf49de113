#!/bin/bash
# tools/feed_profile.sh <tag> -- copy / kernel overlap trace of the host-fed path (VERDICT r5 next #1b).
# One rocprofv3 trace per feed mode (the program directly after `--`), summarised by tools/feed_trace.py.
# Output: gpurun_out/feed_<tag>/<leg>/{summary.txt, trace CSVs}; copy summary.txt (+ *_stats.csv) into profiles/r06_feed/.
set -u
TAG=${1:-r06}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/feed_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for leg in submit submit_pinned submit_1thread lanes_packed_16; do
  lanes=16
  rocprofv3 --kernel-trace --memory-copy-trace --hip-trace --stats --output-format csv -d "$OUT/$leg" -- \
      "$ROOT/build/ntsm_feed_bench" --reads 1e7 --reps 1 --legs $leg --lanes $lanes > "$OUT/$leg.json" 2> "$OUT/$leg.err"
  link=$(python3 -c "import json,sys; d=json.load(open(sys.argv[1])); print(list(d['legs'].values())[0]['link_bytes'])" "$OUT/$leg.json")
  python3 "$ROOT/tools/feed_trace.py" "$OUT/$leg" --link-bytes "$link" $( [ $leg = lanes_packed_16 ] && echo --min-copy-us 30 ) > "$OUT/$leg.summary.txt" 2>&1
  tail -3 "$OUT/$leg.err"; head -12 "$OUT/$leg.summary.txt"
done
