set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/r03c; mkdir -p $OUT; cd $ROOT
HF="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-parameter -shared"
( /opt/rocm/bin/hipcc $HF -DNTSM_TWO_M=13 -o ntsm_amd/libntsm_hip_m13.so ntsm_amd/csrc/ntsm_hip.hip -ldl &
  /opt/rocm/bin/hipcc $HF -DNTSM_TWO_M=15 -o ntsm_amd/libntsm_hip_m15.so ntsm_amd/csrc/ntsm_hip.hip -ldl &
  /opt/rocm/bin/hipcc $HF -DNTSM_TWO_STEP_POS=8 -o ntsm_amd/libntsm_hip_hb8.so ntsm_amd/csrc/ntsm_hip.hip -ldl &
  /opt/rocm/bin/hipcc $HF -DNTSM_TWO_STEP_POS=2 -o ntsm_amd/libntsm_hip_hb2.so ntsm_amd/csrc/ntsm_hip.hip -ldl & wait ) 2>&1 | grep -i error
python3 tools/stress_sweep.py 0:0 0:272 0:224 0:225 4:125 4:27 4:126 4:28 2:0 > $OUT/sweep_m14.jsonl 2> $OUT/sweep_m14.err
for v in m13 m15 hb8 hb2; do NTSM_HIP_LIB=libntsm_hip_$v.so python3 tools/stress_sweep.py 0:0 0:224 > $OUT/sweep_$v.jsonl 2> $OUT/sweep_$v.err; done
cat $OUT/sweep_*.jsonl | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('%-22s %-10s two=%d bloom %.2f MiB  %7.2f ms  %6.1f Gb/s' % (d['lib'], d['spec'], d['two_level'], d['bloom_MiB'], d['kernel_ms'], d['gbases_per_s']))"
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/pmc_tcc -- python3 $ROOT/tools/stress_sweep.py 0:0 > $OUT/pmc_tcc.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $OUT/pmc_sq -- python3 $ROOT/tools/stress_sweep.py 0:0 > $OUT/pmc_sq.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
for g in ("pmc_tcc", "pmc_sq"):
    acc = collections.defaultdict(list)
    for p in glob.glob(os.path.join(out, g, "**", "*counter_collection.csv"), recursive=True):
        d = collections.defaultdict(lambda: collections.defaultdict(float))
        for row in csv.DictReader(open(p)):
            if "ntsm_count" in row["Kernel_Name"]:
                d[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
        for k, v in d.items():
            vals = sorted(v.values()); acc[k] = vals[len(vals) // 2]
    for k, v in acc.items():
        print("%-24s per launch %.5g  per base %.5f" % (k, v, v / 1.5e10))
PY
