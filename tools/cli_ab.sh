#!/bin/bash
# tools/cli_ab.sh <kind> [reads] [reps] -- interleaved A/B of build/ntsmCount -t 16 under sets of environment settings, on a generated
# input of the bench's CLI shape (realistic qualities): wall clock as the caller sees it, the CLI's own `Time:`, and the gap between
# the two (exit cost).  One runner for what rounds 3-4 did with a script per question (tools/archive_r04/r04_exit*.sh,
# r04_gz*.sh, r04_early*.sh).  kind:
#   exit    default (synchronous) | NTSM_FAST_EXIT=1 | NTSM_CLEAN_EXIT=1                      on plain and .gz
#   gz      NTSM_GZ_DECODERS = 8 / 12 / 16 / 20 ; NTSM_GZ_CHUNK = 512 KiB / 2 MiB            on the .gz
#   early   default | NTSM_NO_EARLY=1 | NTSM_EARLY=all                                        on plain and .gz
#   pack    packer of the producer lanes: best the CPU has (AVX-512 VBMI) | NTSM_PACK_IMPL=2 (AVX2) | NTSM_NO_PACK=1 (raw bytes); with the phase times   on plain and .gz
cd "$(dirname "$0")/.." || exit 1
kind=${1:?usage: cli_ab.sh exit|gz|early|pack [reads] [reps]}; reads=${2:-4e7}; reps=${3:-3}
out=gpurun_out/cli_ab_$kind; mkdir -p $out
python3 - "$reads" <<'PY' > $out/prep.log 2>&1
import sys
sys.path.insert(0, '.')
import ntsm_amd, bench
s = ntsm_amd.SynthShort(20241218, 96287, read_seed=7, sites_path='/tmp/cli_ab_sites.fa')
s.write_fastq('/tmp/cli_ab.fq', 0, int(float(sys.argv[1])), threads=16, qual_model=1)
bench.pigz_like('/tmp/cli_ab.fq', '/tmp/cli_ab.fq.gz', threads=16)
PY
one() {   # <file> <env assignments...>
  local f=$1; shift
  local t0=$(date +%s.%N)
  local line=$(env "$@" build/ntsmCount -s /tmp/cli_ab_sites.fa -t 16 $f 2>&1 >/dev/null | grep -o "Time: [0-9.]* s Memory: [0-9]* kbytes")
  local t1=$(date +%s.%N)
  python3 -c "w=$t1-$t0; t=float('$line'.split()[1]); print('%-10s wall %.3f s  Time: %.3f s  outside %.3f s  rss %s kB  [$*]' % ('$(basename $f)', w, t, w-t, '$line'.split()[4]))"
}
case $kind in
  exit)  sets=("NTSM_X=1" "NTSM_FAST_EXIT=1" "NTSM_CLEAN_EXIT=1"); files="/tmp/cli_ab.fq /tmp/cli_ab.fq.gz";;
  gz)    sets=("NTSM_GZ_DECODERS=8" "NTSM_GZ_DECODERS=12" "NTSM_GZ_DECODERS=16" "NTSM_GZ_DECODERS=20" "NTSM_GZ_CHUNK=524288" "NTSM_GZ_CHUNK=2097152"); files="/tmp/cli_ab.fq.gz";;
  early) sets=("NTSM_X=1" "NTSM_NO_EARLY=1" "NTSM_EARLY=all"); files="/tmp/cli_ab.fq /tmp/cli_ab.fq.gz";;
  pack)  sets=("NTSM_PACK_IMPL=0" "NTSM_PACK_IMPL=2" "NTSM_NO_PACK=1"); files="/tmp/cli_ab.fq /tmp/cli_ab.fq.gz";;
  *) echo "unknown kind $kind"; exit 2;;
esac
for rep in $(seq $reps); do for f in $files; do for s in "${sets[@]}"; do one $f $s; done; done; done 2>&1 | tee $out/result.txt
if [ $kind = pack ]; then   # where the time goes: the CLI's own phase lines, once per setting
  for s in "${sets[@]}" "${sets[@]}"; do echo "== $s"; env $s NTSM_PHASE_TIMES=1 build/ntsmCount -s /tmp/cli_ab_sites.fa -t 16 /tmp/cli_ab.fq 2>&1 >/dev/null | grep -i "phase\|Time:"; done > $out/phases.txt 2>&1
fi
