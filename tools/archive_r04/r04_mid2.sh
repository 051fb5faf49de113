#!/bin/bash
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r04_mid2; mkdir -p $out
export NTSM_STRESS_SITES=96287 NTSM_STRESS_SEED=20241218 NTSM_STRESS_READS=1e8
for lib in libntsm_hip.so libntsm_hip_w5.so; do
  NTSM_HIP_LIB=$lib NTSM_STRESS_MIN_KEEP=13 python tools/stress_sweep.py 0:0 0:2002560 0:2003584 4:0 >> $out/full.jsonl 2>> $out/err
  NTSM_HIP_LIB=$lib NTSM_STRESS_MIN_KEEP=0 python tools/stress_sweep.py 0:0 >> $out/n10.jsonl 2>> $out/err
done
cat $out/*.jsonl
