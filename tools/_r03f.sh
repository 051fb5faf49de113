set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/r03f; mkdir -p $OUT; cd $ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "packed_lane or threads_over_files or block_parallel or producer_lanes or pipes or soak" 2>&1 | tail -3
python3 tools/e2e_pack.py 4e7 8,16,32,64 2>&1 | tee $OUT/e2e_4e7.txt | grep -v amdgpu.ids
python3 tools/e2e_pack.py 1.6e8 16,32,64 2>&1 | tee $OUT/e2e_1.6e8.txt | grep -v amdgpu.ids
