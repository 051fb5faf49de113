#!/bin/bash
# A/B of kernel variants: bench.py at 3e8 reads for every ntsm_amd/libntsm_hip*.so (NTSM_HIP_LIB selects the library)
for lib in ntsm_amd/libntsm_hip*.so; do
  NTSM_HIP_LIB=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --reads ${1:-3e8} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']/1e9,1), 'Gbases/s', round(d['ms_per_step'],2),'ms', d['check'])"
done
