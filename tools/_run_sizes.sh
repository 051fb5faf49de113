#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
for spec in "60000 13" "77000 13" "96287 13" "130000 13" "160000 13" "220000 13" "320000 13"; do
  set -- $spec
  NTSM_STRESS_READS=1e8 NTSM_STRESS_SITES=$1 NTSM_STRESS_SEED=20241218 NTSM_STRESS_MIN_KEEP=$2 python3 tools/stress_sweep.py 0:0 5:0 5:2002048 5:2004096 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   keys %8d  %-10s %7.1f Gbases/s two_level=%s' % (d['site_kmers'], d['spec'], d['gbases_per_s'], d['two_level']))"
done
