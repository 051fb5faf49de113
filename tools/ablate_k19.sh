#!/bin/bash
# Ablations of the default k = 19 kernel (build: hipcc ... -DNTSM_ABLATION -o ntsm_amd/libntsm_hip_abl.so; counts are wrong
# by construction, so --no-check).  NTSM_DEBUG_KERNEL bits: 1 no lookups, 2 no bucket reads, 4 no atomics,
# 8 no filter-block requests, 16 every lane reads block 0.
for dbg in ${DBG:-0 1 2 4 8}; do
  NTSM_DEBUG_KERNEL=$dbg NTSM_HIP_LIB=$PWD/ntsm_amd/libntsm_hip_abl.so timeout 300 python bench.py --no-cpu-baseline --no-check --reads ${1:-3e8} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('debug=$dbg', round(d['value']/1e9,1), 'Gbases/s', round(d['ms_per_step'],2),'ms')"
done
