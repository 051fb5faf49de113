#!/usr/bin/env python3
"""tools/memory_side.py <rNN> -- what the L2's memory-side (fabric, "EA") counters say about the count kernel on configs[1],
configs[4] and the 2.5 M-key set (tools/profile_round.sh passes pmc_ea1..3 / pmc7..9): read requests by size, bytes per base,
the share destined for DRAM (MC) as opposed to GMI / IO, and the average read latency in L2 clocks
(TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ).  gfx950's rocprofv3 has NO Infinity-Cache (MALL) hit / miss counter and no HBM-side (UMC /
data-fabric) counter (profiles/r05_counters/counter_names.txt: 688 names, none of them), so "how many of these requests reach
HBM" cannot be counted on this stack; what can be said is derived below from the sizes of the structures against the 256 MiB
Infinity Cache, and written next to the counters.  Output: profiles/<rNN>_memory_side.json + a text table on stdout."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def median_per_dispatch(paths):
    acc = {}
    for p in paths:
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        for row in csv.DictReader(open(p)):
            if "ntsm_count" in row.get("Kernel_Name", ""):
                per[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
        for k, v in per.items():
            vals = sorted(v.values())
            acc[k] = vals[len(vals) // 2]
    return acc


def main():
    r = sys.argv[1]
    sets = {"configs[1] (1.54 M keys)": (sorted(glob.glob(os.path.join(ROOT, "profiles", r + "_full", "pmc[789].csv"))), 1.5e11, None),
            "configs[4] (16 M keys)": (sorted(glob.glob(os.path.join(ROOT, "profiles", r + "_stress", "pmc_ea*.csv"))), None, os.path.join(ROOT, "profiles", r + "_stress", "rate.jsonl")),
            "n10_full (2.5 M keys)": (sorted(glob.glob(os.path.join(ROOT, "profiles", r + "_n10_full", "pmc_ea*.csv"))), None, os.path.join(ROOT, "profiles", r + "_n10_full", "rate.jsonl"))}
    doc = {"note": "rocprofv3 on gfx950 exposes no MALL hit/miss and no HBM-side counter (profiles/r05_counters/): the HBM share is an argument from sizes, not a count"}
    print("%-26s %9s %9s %9s %9s %10s %9s %9s %9s" % ("set", "rd/base", "32B", "64B", "128B", "B/base", "DRAM(MC)", "latency", "wr/base"))
    for name, (paths, bases, rate) in sets.items():
        if not paths:
            continue
        if rate and os.path.exists(rate):
            rj = json.loads([l for l in open(rate) if l.startswith("{")][-1])
            bases = rj["reads"] * 150.0
        c = median_per_dispatch(paths)
        rd, n32, n64, n128 = c.get("TCC_EA0_RDREQ_sum", 0), c.get("TCC_EA0_RDREQ_32B_sum", 0), c.get("TCC_EA0_RDREQ_64B_sum", 0), c.get("TCC_EA0_RDREQ_128B_sum", 0)
        # a request is 32, 64 or 128 bytes; the 32B counter tallies a 64-byte request as 2 and a 128-byte one as 4 (its description)
        by = 32.0 * n32 if n32 else 64.0 * n64 + 128.0 * n128
        row = {"fabric_read_requests_per_base": rd / bases, "requests_64B_per_base": n64 / bases, "requests_128B_per_base": n128 / bases,
               "read_32B_units_per_base": n32 / bases, "fabric_read_bytes_per_base": by / bases,
               "share_destined_for_dram_mc": c.get("TCC_EA0_RDREQ_DRAM_sum", 0) / rd if rd else None,
               "avg_fabric_read_latency_l2_clocks": c.get("TCC_EA0_RDREQ_LEVEL_sum", 0) / rd if rd else None,
               "fabric_write_requests_per_base": c.get("TCC_EA0_WRREQ_sum", 0) / bases,
               "atomic_requests_to_dram_per_base": c.get("TCC_EA0_WRREQ_ATOMIC_DRAM_sum", 0) / bases,
               "uncached_read_32B_units_per_base": c.get("TCC_EA0_RD_UNCACHED_32B_sum", 0) / bases,
               "dram_credit_stall_cycles_per_base": c.get("TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum", 0) / bases, "bases_per_launch": bases}
        doc[name] = row
        # the same fields next to the per-base request counts bench.py quotes (profiles/<rNN>_*traffic.json)
        tj = os.path.join(ROOT, "profiles", r + {"configs[1] (1.54 M keys)": "_traffic.json", "configs[4] (16 M keys)": "_stress_traffic.json", "n10_full (2.5 M keys)": "_n10_full_traffic.json"}[name])
        if os.path.exists(tj):
            t = json.load(open(tj))
            t.update({"fabric_read_bytes_per_base": row["fabric_read_bytes_per_base"], "share_of_fabric_reads_destined_for_dram_mc": row["share_destined_for_dram_mc"],
                      "avg_fabric_read_latency_l2_clocks": row["avg_fabric_read_latency_l2_clocks"], "hbm_read_bytes_per_base": None,
                      "hbm_note": "not countable: rocprofv3 on gfx950 has no Infinity-Cache (MALL) hit/miss and no HBM-side counter (profiles/r05_counters/counter_names.txt); "
                                  "TCC_EA0_RDREQ_DRAM counts requests DESTINED for local memory, Infinity-Cache hits included.  From sizes: the stream (1.007 B/base) must come "
                                  "from HBM; filters and key table of <= 256 MiB stay in the Infinity Cache (DESIGN.md section 7)"})
            json.dump(t, open(tj, "w"), indent=1)
        print("%-26s %9.4f %9.4f %9.4f %9.4f %10.3f %9.3f %9.0f %9.4f" % (name, row["fabric_read_requests_per_base"], row["read_32B_units_per_base"], row["requests_64B_per_base"],
              row["requests_128B_per_base"], row["fabric_read_bytes_per_base"], row["share_destined_for_dram_mc"] or 0, row["avg_fabric_read_latency_l2_clocks"] or 0,
              row["fabric_write_requests_per_base"]))
    json.dump(doc, open(os.path.join(ROOT, "profiles", r + "_memory_side.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
