#!/bin/bash
# tools/ablate.sh [reads] -- builds the ablation libraries and runs the three studies of tools/ablate.py, then bench.py on every
# ntsm_amd/libntsm_hip.so and build/lib/libntsm_hip*.so as an A/B of the kernel builds present (default, tab, m12, abl ...).  Ablation libraries count WRONGLY
# by construction (ntsm_amd/csrc/ntsm_ablation.inc): nothing here is ever shipped.
cd "$(dirname "$0")/.." || exit 1
R=${1:-1e8}
make ablation > /dev/null || exit 1
python3 tools/ablate.py workload $R
python3 tools/ablate.py grid $R
NTSM_HIP_LIB=libntsm_hip_abl.so python3 tools/ablate.py switches $R
for lib in ntsm_amd/libntsm_hip.so build/lib/libntsm_hip*.so; do
  case $lib in *abl*) chk=--no-check;; *) chk=;; esac
  NTSM_HIP_LIB=$(basename $lib) timeout 600 python3 bench.py --no-cpu-baseline $chk --other-configs none --reads 3e8 2>/dev/null |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']/1e9,1), 'Gbases/s', round(d['ms_per_step'],2), 'ms', d['check'].get('equals_generic_kernel_sum_of_pieces_below_2GiB'))"
done
