#!/bin/bash
# round 4: the gzip ingest after the inner-loop work: loops alone (tools/inflate_bench.cpp), decoder alone (gunzip_bench), the CLI
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/r04_gz3; mkdir -p $out
H=ntsm_amd/csrc/host
g++ -O3 -std=c++17 -I $H tools/inflate_bench.cpp $H/inflate.cpp $H/inflate_spec.cpp $H/crc32_fast.cpp -o build/inflate_bench -lz
g++ -O3 -std=c++17 -I $H tools/gunzip_bench.cpp $H/gz_stream.cpp $H/gz_parallel.cpp $H/inflate.cpp $H/inflate_spec.cpp $H/crc32_fast.cpp -o build/gunzip_bench -lz -pthread
python - <<'PY' > $out/prep.log 2>&1
import sys
sys.path.insert(0, '.')
import ntsm_amd, bench
s = ntsm_amd.SynthShort(20241218, 96287, read_seed=7, sites_path='/tmp/r04_sites.fa')
s.write_fastq('/tmp/r04s.fq', 0, int(2e6), threads=16)
bench.pigz_like('/tmp/r04s.fq', '/tmp/r04s.fq.gz', threads=16)
s.write_fastq('/tmp/r04.fq', 0, int(4e7), threads=32)
bench.pigz_like('/tmp/r04.fq', '/tmp/r04.fq.gz', threads=48)
PY
build/inflate_bench /tmp/r04s.fq.gz 5 2>&1 | tee $out/inflate_bench.txt
NTSM_PGZ_PROF=1 build/gunzip_bench /tmp/r04.fq.gz 1 8 12 16 2>&1 | tee $out/gunzip_bench.txt
stat() { awk '/usage_usec/ {printf "%s", $2}' /sys/fs/cgroup/cpu.stat; }
one() {
  local u0=$(stat); local t0=$(date +%s.%N)
  local line=$(env "$@" NTSM_PHASE_TIMES=1 build/ntsmCount -s /tmp/r04_sites.fa -t 16 /tmp/r04.fq.gz 2>&1 >/dev/null | grep -E "early ingest|inflate\+parse|Time:" | sed 's/.*early ingest/early ingest/; s/.*: lanes/lanes/; s/ Memory.*//' | tr '\n' ' ')
  local t1=$(date +%s.%N); local u1=$(stat)
  python3 -c "w=$t1-$t0; print('wall %.3f s -> %.2f Gbases/s, cgroup cpu %.2f s | $line [$*]' % (w, 6.0/w, ($u1-$u0)/1e6))"
}
{
for rep in 1 2 3 4; do one NTSM_X=1; done
for rep in 1 2 3; do one NTSM_GZ_DECODERS=10; done
for rep in 1 2 3; do one NTSM_GZ_DECODERS=14; done
for rep in 1 2 3; do one NTSM_NO_EARLY=1; done
} 2>&1 | tee $out/cli.txt
