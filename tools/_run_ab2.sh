#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
export NTSM_STRESS_READS=1e8 NTSM_STRESS_SITES=96287 NTSM_STRESS_SEED=20241218 NTSM_STRESS_MIN_KEEP=13
for rep in 1 2; do
for lib in libntsm_hip.so libntsm_hip_rG.so libntsm_hip_rH.so libntsm_hip_rI.so; do
  NTSM_HIP_LIB=$lib python3 tools/stress_sweep.py 5:0 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('   %-24s %7.1f Gbases/s hits %d' % ('$lib', d['gbases_per_s'], d['hits_per_pass']))"
done; done
