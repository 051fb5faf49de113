#!/bin/bash
# round 4: is the pod's CPU quota (cpu.max = 16 CPUs of 256) what the CLI's last tenths of a second go to?  cpu.stat's
# throttled time around each run, by thread counts; and what a process that only initialises HIP costs to start and end.
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r04_throttle; mkdir -p $out
python - <<'PY' > $out/prep.log 2>&1
import sys, os, time
sys.path.insert(0, '.')
import ntsm_amd, bench
s = ntsm_amd.SynthShort(20241218, 96287, read_seed=7, sites_path='/tmp/r04_sites.fa')
s.write_fastq('/tmp/r04.fq', 0, int(4e7), threads=32)
bench.pigz_like('/tmp/r04.fq', '/tmp/r04.fq.gz', threads=48)
PY
cat > /tmp/hipnull.cpp <<'CPP'
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <cstdio>
int main(int argc, char **) { void *p = nullptr; hipMalloc(&p, 1 << 20); hipStream_t s; hipStreamCreate(&s); hipDeviceSynchronize(); if (argc > 1) return 0; _exit(0); }
CPP
hipcc -O2 /tmp/hipnull.cpp -o /tmp/hipnull 2>>$out/prep.log
stat() { awk '/nr_throttled|throttled_usec|usage_usec/ {printf "%s=%s ", $1, $2}' /sys/fs/cgroup/cpu.stat; }
one() {
  local a=$(stat); local t0=$(date +%s.%N)
  local line=$(env "$@" 2>&1 >/dev/null | grep -o "Time: [0-9.]* s" | head -1)
  local t1=$(date +%s.%N); local b=$(stat)
  python3 - "$a" "$b" "$t0" "$t1" "$line" "$*" <<'PY'
import sys
a = dict(x.split('=') for x in sys.argv[1].split()); b = dict(x.split('=') for x in sys.argv[2].split())
w = float(sys.argv[4]) - float(sys.argv[3]); t = float(sys.argv[5].split()[1]) if sys.argv[5] else 0.0
print('wall %.3f s  Time: %.3f s  rest %.3f s | cpu used %.2f s, throttled %d periods / %.3f s | %s' % (w, t, w - t, (int(b['usage_usec']) - int(a['usage_usec'])) / 1e6, int(b['nr_throttled']) - int(a['nr_throttled']), (int(b['throttled_usec']) - int(a['throttled_usec'])) / 1e6, sys.argv[6]))
PY
}
S="-s /tmp/r04_sites.fa"
{
for rep in 1 2 3; do one /tmp/hipnull; one /tmp/hipnull clean; done
for rep in 1 2 3; do
  one build/ntsmCount $S -t 16 /tmp/r04.fq
  one build/ntsmCount $S -t 12 /tmp/r04.fq
  one build/ntsmCount $S -t 8 /tmp/r04.fq
  one NTSM_GZ_DECODERS=12 build/ntsmCount $S -t 16 /tmp/r04.fq.gz
  one NTSM_GZ_DECODERS=12 build/ntsmCount $S -t 4 /tmp/r04.fq.gz
  one NTSM_GZ_DECODERS=10 build/ntsmCount $S -t 6 /tmp/r04.fq.gz
  one NTSM_GZ_DECODERS=8 build/ntsmCount $S -t 8 /tmp/r04.fq.gz
  one NTSM_GZ_DECODERS=14 build/ntsmCount $S -t 3 /tmp/r04.fq.gz
done
} 2>&1 | tee $out/runs.txt
