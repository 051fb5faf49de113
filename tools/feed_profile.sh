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

# The CLI itself on a plain FASTQ (16 feeder threads, packed lanes): the same trace.  NTSM_CLEAN_EXIT=1: rocprofv3 flushes at exit.
READS=${NTSM_FEED_CLI_READS:-10000000}
D=$(mktemp -d /tmp/ntsm_feedcli_XXXXXX)
python3 - "$ROOT" "$D" "$READS" <<'PY'
import sys
sys.path.insert(0, sys.argv[1])
import ntsm_amd
s = ntsm_amd.SynthShort(20241218, 96287, read_seed=7, sites_path=sys.argv[2] + "/sites.fa")
s.write_fastq(sys.argv[2] + "/reads.fq", 0, int(sys.argv[3]), threads=16, qual_model=1)
PY
NTSM_CLEAN_EXIT=1 NTSM_PHASE_TIMES=1 rocprofv3 --kernel-trace --memory-copy-trace --hip-trace --stats --output-format csv -d "$OUT/cli_plain" -- \
    "$ROOT/build/ntsmCount" -s "$D/sites.fa" -t 16 "$D/reads.fq" > "$OUT/cli_plain.counts.txt" 2> "$OUT/cli_plain.err"
# packed lanes: 152 positions per 150 bp read, 3/8 byte per position
python3 "$ROOT/tools/feed_trace.py" "$OUT/cli_plain" --min-copy-us 15 --link-bytes $((READS * 152 * 3 / 8)) > "$OUT/cli_plain.summary.txt" 2>&1
grep -E "^\[phase\]|Time:" "$OUT/cli_plain.err" | head -8 >> "$OUT/cli_plain.summary.txt"
rm -rf "$D" "$OUT/cli_plain.counts.txt"
head -12 "$OUT/cli_plain.summary.txt"
