#!/usr/bin/env python3
"""tools/feed_trace.py DIR [--min-copy-us T] [--link-bytes N] -- copy / kernel overlap of the host-fed path from a rocprofv3 trace
(`rocprofv3 --kernel-trace --memory-copy-trace --hip-trace --output-format csv -d DIR -- build/ntsm_feed_bench --legs submit ...`).

Takes the host-to-device copies that last at least --min-copy-us (the batches; rocprofv3's copy trace carries no size; default
300 us = a quarter of a 64 MiB batch at link speed; --link-bytes = what the leg handed to the link, for the GB/s lines) and the count kernels
(ntsm_count_*), and reports for the window [first batch copy start, last count kernel end]: time the link is busy (union of
the copies), time a count kernel runs, their overlap, time neither runs (idle), the gaps between consecutive copies and the
HIP API calls that took longest.  Prints a text summary and one JSON object (last line)."""
import csv
import glob
import json
import os
import sys


def union(iv):
    iv = sorted(iv)
    out = []
    for a, b in iv:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def total(iv):
    return sum(b - a for a, b in iv)


def intersect(x, y):
    i = j = 0
    t = 0
    while i < len(x) and j < len(y):
        a, b = max(x[i][0], y[j][0]), min(x[i][1], y[j][1])
        if b > a:
            t += b - a
        if x[i][1] < y[j][1]:
            i += 1
        else:
            j += 1
    return t


def main():
    d = sys.argv[1]
    min_us, link_bytes = 300.0, 0
    if "--min-copy-us" in sys.argv:
        min_us = float(sys.argv[sys.argv.index("--min-copy-us") + 1])
    if "--link-bytes" in sys.argv:
        link_bytes = int(float(sys.argv[sys.argv.index("--link-bytes") + 1]))
    copies, kernels, api = [], [], {}
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            if "HOST_TO_DEVICE" in r.get("Direction", "").upper() and (b - a) >= min_us * 1e3:
                copies.append((a, b, 0))
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "ntsm_count" in r["Kernel_Name"]:
                kernels.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    for f in glob.glob(os.path.join(d, "**", "*hip_api_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            a = api.setdefault(r["Function"], [0, 0, 0])
            dt = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            a[0] += 1
            a[1] += dt
            a[2] = max(a[2], dt)
    if not copies or not kernels:
        print("no batch copies / count kernels found under", d)
        return 1
    copies.sort()
    kernels.sort()
    # the leg: the LONGEST run of batch copies whose neighbours are less than 20 ms apart (the ceiling measurement and other
    # legs of the same process form runs of their own)
    runs, cur = [], [copies[0]]
    for c in copies[1:]:
        if c[0] - cur[-1][1] > 20_000_000:
            runs.append(cur)
            cur = []
        cur.append(c)
    runs.append(cur)
    want = os.environ.get("FEED_TRACE_RUN")
    runs_with_kernels = [r for r in runs if any(r[0][0] <= k[0] <= r[-1][1] + 20_000_000 for k in kernels)]
    # feed_bench's order: the resident reference (one pageable hipMemcpy, split by the runtime, + one kernel), the ceiling
    # measurement (copies only), then the leg -- the LAST run of batch copies that has count kernels in it
    run = runs_with_kernels[int(want)] if want is not None else runs_with_kernels[-1]
    t0 = run[0][0]
    ks = [k for k in kernels if t0 <= k[0] <= run[-1][1] + 20_000_000]
    t1 = max(run[-1][1], ks[-1][1])
    cu, ku = union([(a, b) for a, b, _ in run]), union([(a, b) for a, b, _ in ks])
    both = intersect(cu, ku)
    either = total(union([(a, b) for a, b, _ in run] + [(a, b) for a, b, _ in ks]))
    window = t1 - t0
    gaps = sorted((cu[i + 1][0] - cu[i][1]) / 1e3 for i in range(len(cu) - 1))
    nbytes = link_bytes
    out = {
        "window_ms": window / 1e6, "batch_copies": len(run), "count_kernels": len(ks), "bytes_copied": nbytes,
        "link_busy_ms": total(cu) / 1e6, "link_busy_frac": total(cu) / window,
        "kernel_busy_ms": total(ku) / 1e6, "kernel_busy_frac": total(ku) / window,
        "copy_and_kernel_overlap_ms": both / 1e6, "kernel_time_hidden_under_copies_frac": both / max(total(ku), 1),
        "idle_ms": (window - either) / 1e6, "idle_frac": (window - either) / window,
        "GBps_while_copying": nbytes / max(total(cu), 1), "GBps_over_window": nbytes / window,
        "avg_copy_ms": sum(b - a for a, b, _ in run) / len(run) / 1e6, "avg_kernel_ms": sum(b - a for a, b, _ in ks) / len(ks) / 1e6,
        "gaps_between_copies_us": {"n": len(gaps), "median": gaps[len(gaps) // 2] if gaps else None, "p90": gaps[int(len(gaps) * 0.9)] if gaps else None,
                                   "max": gaps[-1] if gaps else None, "sum_ms": sum(gaps) / 1e3},
        "runs_of_batch_copies_in_trace": [len(r) for r in runs],
    }
    print("window %.2f ms: %d batch copies (%.2f GB), %d count kernels" % (out["window_ms"], len(run), nbytes / 1e9, len(ks)))
    print("  link busy      %8.2f ms  %5.1f %%   (%.1f GB/s while copying, %.1f GB/s over the window)" % (out["link_busy_ms"], 100 * out["link_busy_frac"], out["GBps_while_copying"], out["GBps_over_window"]))
    print("  kernels busy   %8.2f ms  %5.1f %%   (%.1f %% of kernel time under a copy)" % (out["kernel_busy_ms"], 100 * out["kernel_busy_frac"], 100 * out["kernel_time_hidden_under_copies_frac"]))
    print("  neither (idle) %8.2f ms  %5.1f %%" % (out["idle_ms"], 100 * out["idle_frac"]))
    print("  gaps between consecutive copies: n %d, median %.1f us, p90 %.1f us, max %.1f us, sum %.2f ms" %
          (len(gaps), out["gaps_between_copies_us"]["median"] or 0, out["gaps_between_copies_us"]["p90"] or 0, out["gaps_between_copies_us"]["max"] or 0, out["gaps_between_copies_us"]["sum_ms"]))
    if api:
        print("  HIP API (whole process), by total time:")
        for name, (n, tot, mx) in sorted(api.items(), key=lambda kv: -kv[1][1])[:8]:
            print("    %-34s calls %6d  total %9.2f ms  max %8.3f ms" % (name, n, tot / 1e6, mx / 1e6))
        out["hip_api_top"] = {name: {"calls": n, "total_ms": tot / 1e6, "max_ms": mx / 1e6} for name, (n, tot, mx) in sorted(api.items(), key=lambda kv: -kv[1][1])[:8]}
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    sys.exit(main())
