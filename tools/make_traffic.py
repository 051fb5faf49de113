#!/usr/bin/env python3
"""tools/make_traffic.py <gpurun_out/prof_TAG> <profiles/rNN_traffic.json> <bases_per_launch>

Turns the PMC passes of tools/profile.sh into the per-base constants bench.py quotes in `roofline.traffic`
and `valu_busy_frac_from_pmc`.  The file carries the SHA-256 of the kernel sources it was measured on;
bench.py reports traffic = null when the sources have changed since (no silently stale constants).
HBM/fabric bytes follow /opt/skills/guides/MI355X_MICROARCH.md's rocprofv3 recipe: FETCH_SIZE and WRITE_SIZE
from their own --pmc passes, in KiB; FETCH_SIZE tallies 16-byte-per-lane coalesced reads at half on gfx950,
so half of the stream bytes are added back; the random 4/8/16-byte filter and table reads are left as counted."""
import collections, csv, glob, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ["ntsm_amd/csrc/kernels_mz.hip", "ntsm_amd/csrc/kernels_run.hip", "ntsm_amd/csrc/kernels_generic.hip", "ntsm_amd/csrc/kernels_common.h", "ntsm_amd/csrc/ntsm_hooks.h",
                  "ntsm_amd/csrc/ntsm_device.h", "ntsm_amd/csrc/runtime.cpp", "ntsm_amd/csrc/tables.cpp"]


def kernel_source_sha16():
    """SHA-256 over everything that decides what runs on the device and how it is launched: the kernel translation units and
    their shared headers, runtime.cpp (the launch code: grid heuristic, tile sizes), tables.cpp (filter and table geometry),
    ntsm_device.h and the build flags of the library (Makefile's HIPFLAGS line, the hiplib recipe and the libntsm_hip.so rule,
    where -D overrides of the build parameters would sit)."""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, f), "rb").read())
    mk = open(os.path.join(ROOT, "Makefile")).read().split("\n")
    for i, line in enumerate(mk):
        if line.startswith("HIPFLAGS") or line.startswith("ntsm_amd/libntsm_hip.so:") or line.startswith("\tpids=\"\"; for f in $(HIPLIB_DEV)") or line.startswith("\t$(HIPCC) $(HIPFLAGS) -shared -o $(1)"):
            h.update(line.encode())
            if line.startswith("ntsm_amd/libntsm_hip.so:") and i + 1 < len(mk):
                h.update(mk[i + 1].encode())
    return h.hexdigest()[:16]


def memory_side_bytes(c):
    """Bytes that crossed the L2's memory-side port per launch, from the fabric ("EA") request counters -- the ONE definition of
    `roofline.traffic` since round 6: reads = 128 x RDREQ_128B + 64 x RDREQ_64B + 32 x the rest, writes =
    64 x WRREQ_64B + 32 x the rest (atomics included: they are write requests on this port).  Returns (read, write) or None when
    the EA passes are missing.  FETCH_SIZE / WRITE_SIZE (the guide's generic recipe) stay in the file for comparison: on this
    kernel FETCH_SIZE undercounts (it tallies the 16-byte-per-lane stream loads at half), which is why rounds 1-5 added half the
    stream back by hand and ended up with two figures for one launch."""
    rd = c.get("TCC_EA0_RDREQ_sum")
    if not rd:
        return None
    n64, n128 = c.get("TCC_EA0_RDREQ_64B_sum", 0.0), c.get("TCC_EA0_RDREQ_128B_sum", 0.0)
    read = 128.0 * n128 + 64.0 * n64 + 32.0 * max(0.0, rd - n64 - n128)
    wr, w64 = c.get("TCC_EA0_WRREQ_sum", 0.0), c.get("TCC_EA0_WRREQ_64B_sum", 0.0)
    return read, 64.0 * w64 + 32.0 * max(0.0, wr - w64)


ALGORITHMIC_BYTES_PER_BASE = 158.0 / 150.0                       # 1 byte per base + 8 bytes of offset per 150 bp read


def main():
    prof, out, bases = sys.argv[1], sys.argv[2], float(sys.argv[3])
    per = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(prof, "pmc*", "**", "*counter_collection.csv"), recursive=True)):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if "ntsm_count" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
        for name, d in acc.items():
            vals = sorted(d.values())
            per[name] = vals[len(vals) // 2]                     # median dispatch (all launches do the same work)
    stream = bases * 151.0 / 150.0 if len(sys.argv) < 5 else float(sys.argv[4])   # optional 4th argument: stream bytes per launch
    fetch_raw, write = per["FETCH_SIZE"] * 1024.0, per["WRITE_SIZE"] * 1024.0
    fetch = fetch_raw + 0.5 * stream
    cycles = per["GRBM_GUI_ACTIVE"] / 8.0                        # summed over the 8 XCDs
    ms = memory_side_bytes(per)
    doc = {
        "source": "%s: separate rocprofv3 --pmc passes of tools/profile.sh (one counter group per pass, no trace domains)" % os.path.relpath(prof, ROOT),
        "kernel_source_sha16": kernel_source_sha16(),
        "bases_per_launch": bases,
        "fetch_bytes_per_launch_raw": fetch_raw, "write_bytes_per_launch": write, "fetch_bytes_per_launch_corrected": fetch,
        "correction": "+0.5 x stream bytes: FETCH_SIZE tallies 16-B/lane coalesced reads at half on gfx950; random 4/8/16-byte filter and table reads left as counted",
        "fetch_size_plus_half_stream_bytes_per_base": (fetch + write) / bases,     # rounds 1-5's figure, kept for comparison only
        "traffic_bytes_per_base": (sum(ms) / bases) if ms else None,
        "traffic_read_bytes_per_base": (ms[0] / bases) if ms else None, "traffic_write_bytes_per_base": (ms[1] / bases) if ms else None,
        "traffic_over_algorithmic": (sum(ms) / bases / ALGORITHMIC_BYTES_PER_BASE) if ms else None,
        "traffic_definition": "memory side of the L2: 128 x TCC_EA0_RDREQ_128B + 64 x TCC_EA0_RDREQ_64B + 32 x other reads + 64 x TCC_EA0_WRREQ_64B + 32 x other writes, per launch / bases",
        "note": "memory-side traffic (L2 misses): includes Infinity-Cache hits of the filters / key table; the read stream itself crosses HBM once",
        "valu_insts_per_position": per["SQ_INSTS_VALU"] / (stream / 64.0) if per.get("SQ_INSTS_VALU") else None,
        "valu_busy_frac": per["SQ_INSTS_VALU"] * 4.2 / (1024.0 * cycles) if per.get("SQ_INSTS_VALU") and cycles else None,
        "valu_busy_note": "SQ_INSTS_VALU x 4.2 cycles per wave64 instruction (profiles/r02_microbench/valu_rate.txt) / (1024 SIMDs x GRBM_GUI_ACTIVE/8)",
        "l2_requests_per_base": per["TCC_REQ_sum"] / bases if per.get("TCC_REQ_sum") else None,
        "l2_misses_per_base": per["TCC_MISS_sum"] / bases if per.get("TCC_MISS_sum") else None,
        "fabric_read_requests_per_base": per["TCC_EA0_RDREQ_sum"] / bases if per.get("TCC_EA0_RDREQ_sum") else None,
        "l2_requests_per_launch": per.get("TCC_REQ_sum"), "l2_misses_per_launch": per.get("TCC_MISS_sum"),
        "l2_request_rate_frac_of_cap": (per["TCC_REQ_sum"] / (cycles / 2.4e9) / 266e9) if per.get("TCC_REQ_sum") and cycles else None,
        "l2_cap_note": "cap = 266 G requests/s whatever the request width (profiles/r02_microbench/l2_policy.txt); time from GRBM_GUI_ACTIVE at 2.4 GHz",
    }
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
