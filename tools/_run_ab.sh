#!/bin/bash
# scratch: variant 5 under several builds (bench set only: fast)
cd "$(dirname "$0")/.." || exit 1
for lib in ${LIBS:-libntsm_hip.so libntsm_hip_run96w4.so libntsm_hip_run128abl1.so libntsm_hip_run128abl2.so libntsm_hip_run96w4abl1.so}; do
  echo "== $lib"
  NTSM_HIP_LIB=$lib python3 tools/run_kernel_check.py ${READS:-1e8} 5 2>&1 | grep -v amdgpu.ids | python3 -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: print(l.strip()[:200]); continue
    print('   %-14s v%d  %7.1f Gbases/s  hits %d' % (d['set'], d['variant'], d['gbases_per_s'], d['hits']))"
done
