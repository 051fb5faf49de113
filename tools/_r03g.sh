set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/r03g; mkdir -p $OUT; cd $ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "packed_lane or threads_over_files or block_parallel or producer_lanes or pipes or soak" 2>&1 | tail -3
python3 tools/e2e_pack.py 1.6e8 16,32,64 2>&1 | tee $OUT/e2e_prefault.txt | grep -v amdgpu.ids
NTSM_NO_PREFAULT=1 python3 tools/e2e_pack.py 1.6e8 16,32 2>&1 | tee $OUT/e2e_noprefault.txt | grep -v amdgpu.ids
