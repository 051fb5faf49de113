#!/bin/bash
# round 4: the parallel gzip ingest on the GPU box: decoder alone (tools/gunzip_bench.cpp), then the CLI on one .fq.gz
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r04_gz; mkdir -p $out
H=ntsm_amd/csrc/host
g++ -O3 -std=c++17 -I $H tools/gunzip_bench.cpp $H/gz_stream.cpp $H/gz_parallel.cpp $H/inflate.cpp $H/inflate_spec.cpp $H/crc32_fast.cpp -o build/gunzip_bench -lz -pthread
python - <<'PY' > $out/prep.log 2>&1
import sys, os, time
sys.path.insert(0, '.')
import ntsm_amd, bench
s = ntsm_amd.SynthShort(20241218, 96287, read_seed=7, sites_path='/tmp/r04_sites.fa')
t = time.time(); s.write_fastq('/tmp/r04.fq', 0, int(float(os.environ.get('GZ_READS', 2e7))), threads=32); print('fastq', time.time() - t, os.path.getsize('/tmp/r04.fq'))
t = time.time(); n = bench.pigz_like('/tmp/r04.fq', '/tmp/r04.fq.gz', threads=48); print('pigz_like', time.time() - t, n)
PY
cat $out/prep.log
nproc; lscpu | grep -E "Model name|Socket|Thread|L3" 
for c in 262144 524288 1048576 2097152; do echo chunk $c; CHUNK=$c NTSM_PGZ_PROF=1 build/gunzip_bench /tmp/r04.fq.gz 1 8 16 32 64 2>&1; done | tee $out/gunzip_bench.txt
for t in 1 8 16 32; do for rep in 1 2; do NTSM_PHASE_TIMES=1 build/ntsmCount -s /tmp/r04_sites.fa -t $t /tmp/r04.fq.gz 2>&1 >/tmp/r04_counts_$t.txt | grep -E "phase|Time" | tr '\n' ' '; echo "[-t $t]"; done; done | tee $out/cli.txt
for c in 524288 1048576 4194304; do NTSM_GZ_CHUNK=$c NTSM_PHASE_TIMES=1 build/ntsmCount -s /tmp/r04_sites.fa -t 16 /tmp/r04.fq.gz 2>&1 >/dev/null | grep -E "inflate|Time" | tr '\n' ' '; echo "[chunk $c]"; done | tee -a $out/cli.txt
NTSM_PHASE_TIMES=1 build/ntsmCount -s /tmp/r04_sites.fa -t 16 /tmp/r04.fq 2>&1 >/tmp/r04_counts_plain.txt | grep -E "phase|Time" | tr '\n' ' '; echo "[plain -t 16]"
sha256sum /tmp/r04_counts_*.txt | awk '{print $1}' | sort | uniq -c
