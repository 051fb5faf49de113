#!/usr/bin/env python3
"""tools/stress_traffic.py <dir with pmc_tcc/ pmc_sq/ rate.jsonl> <profiles/rNN_stress_traffic.json>

Per-base L2 / fabric request counts of the count kernel on BASELINE.json configs[4] (1 M sites) from the rocprofv3 --pmc
passes of tools/profile_r03.sh over tools/stress_sweep.py, stamped with the hash of the kernel sources (bench.py reports
them as null once the sources have changed)."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_traffic import ALGORITHMIC_BYTES_PER_BASE, kernel_source_sha16, memory_side_bytes


def median_per_dispatch(d):
    acc = {}
    for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for row in csv.DictReader(open(p)):
            if "ntsm_count" in row["Kernel_Name"]:
                per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
        for k, v in per.items():
            vals = sorted(v.values())
            acc[k] = vals[len(vals) // 2]
    return acc


def main():
    src, out = sys.argv[1], sys.argv[2]
    rate = json.loads([l for l in open(os.path.join(src, "rate.jsonl")) if l.startswith("{")][-1])
    bases = rate["reads"] * 150.0
    c = median_per_dispatch(os.path.join(src, "pmc_tcc"))
    c.update(median_per_dispatch(os.path.join(src, "pmc_sq")))
    for g in ("pmc_ea1", "pmc_ea2", "pmc_ea3"):
        if os.path.isdir(os.path.join(src, g)):
            c.update(median_per_dispatch(os.path.join(src, g)))
    ms = memory_side_bytes(c)
    doc = {"source": "%s: rocprofv3 --pmc passes (one counter group each, no trace domains) over tools/stress_sweep.py 0:0" % os.path.basename(src.rstrip("/")),
           "kernel_source_sha16": kernel_source_sha16(), "workload": "%d site 19-mers, %d reads of 150 bp" % (rate["site_kmers"], rate["reads"]),
           "two_level": rate["two_level"], "bloom_MiB": rate["bloom_MiB"], "site_minimizers": rate["site_minimizers"],
           "kernel_ms_unprofiled": rate["kernel_ms"], "gbases_per_s_unprofiled": rate["gbases_per_s"],
           "l2_requests_per_base": c.get("TCC_REQ_sum", 0) / bases, "l2_hits_per_base": c.get("TCC_HIT_sum", 0) / bases,
           "l2_misses_per_base": c.get("TCC_MISS_sum", 0) / bases, "fabric_read_requests_per_base": c.get("TCC_EA0_RDREQ_sum", 0) / bases,
           "valu_insts_per_position": c["SQ_INSTS_VALU"] * 64 / (bases * 151 / 150) if c.get("SQ_INSTS_VALU") else None,
           "fabric_request_rate_G_per_s": c.get("TCC_EA0_RDREQ_sum", 0) / (rate["kernel_ms"] / 1e3) / 1e9,
           "traffic_bytes_per_base": (sum(ms) / bases) if ms else None, "traffic_read_bytes_per_base": (ms[0] / bases) if ms else None,
           "traffic_write_bytes_per_base": (ms[1] / bases) if ms else None,
           "traffic_over_algorithmic": (sum(ms) / bases / ALGORITHMIC_BYTES_PER_BASE) if ms else None,
           "traffic_definition": "memory side of the L2: 128 x TCC_EA0_RDREQ_128B + 64 x TCC_EA0_RDREQ_64B + 32 x other reads + 64 x TCC_EA0_WRREQ_64B + 32 x other writes, per launch / bases",
           "note": "fabric (Infinity Cache / HBM) read requests: ~55-65 G/s is the measured cap for random requests that miss the L2"}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
