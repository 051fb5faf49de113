#!/bin/bash
# round 4: the FASTQ record scanner on the GPU box's CPU: memchr-based (NTSM_SCALAR_PARSE=1) against predict-and-verify, one thread, 31 MB and 1.9 GB of text
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r04_parse; mkdir -p $out
H=ntsm_amd/csrc/host
g++ -O3 -std=c++17 -I $H tools/parse_bench.cpp $H/parallel_fastq.cpp $H/pack2.cpp -pthread -o build/parse_bench || exit 1
python - <<'PY' > $out/prep.log 2>&1
import sys
sys.path.insert(0, '.')
import ntsm_amd
s = ntsm_amd.SynthShort(20241218, 96287, read_seed=7, sites_path='/tmp/r04_sites.fa')
s.write_fastq('/tmp/small.fq', 0, 100000, threads=4)
s.write_fastq('/tmp/big.fq', 0, 6000000, threads=16)
PY
{
for f in /tmp/small.fq /tmp/big.fq; do
  echo "== $f"; build/parse_bench $f 3
  echo "== $f NTSM_SCALAR_PARSE=1"; NTSM_SCALAR_PARSE=1 build/parse_bench $f 3
done
} 2>&1 | tee $out/parse_bench.txt
