#!/usr/bin/env python3
"""tools/isa_budget.py [out_dir] -- instruction budget of ntsm_count_mz_kernel<0, false, 128, false>'s main loop from the compiler's ISA.

Compiles ntsm_amd/csrc/kernels_mz.hip to gfx950 assembly, takes the basic blocks of the 8-position loop body (the
block with the eight filter-block loads and the blocks it falls through to up to the loop's back edge) and prints
the instruction counts by unit and by mnemonic, per 8 positions and per position.  With out_dir: also writes
main_loop.s (the loop body as compiled) and isa_budget.txt."""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_ZN12_GLOBAL__N_120ntsm_count_mz_kernelILi0ELb0ELi128ELb0EEEv15NtsmCountParams"   # <KMODE 0, PER_READ false, C 128, TWO false>


def main():
    out_dir = sys.argv[1] if len(sys.argv) > 1 else None
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fvisibility=hidden", "-S", "--cuda-device-only",
                               "-o", asm, os.path.join(ROOT, "ntsm_amd/csrc/kernels_mz.hip")], stderr=subprocess.DEVNULL, cwd=td)
        text = open(asm).read()
    body = text[text.index(KERNEL + ":"):]
    body = body[:body.index(".Lfunc_end")]
    blocks, order, cur = {}, [], None
    for line in body.split("\n"):
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            cur = m.group(1); blocks[cur] = []; order.append(cur); continue
        t = line.strip()
        if cur and t and not t.startswith((".", ";")):
            blocks[cur].append(t.split(";")[0].strip())
    main_blk = max(blocks, key=lambda b: sum(1 for i in blocks[b] if i.startswith("buffer_load_dwordx4") and "idxen" in i))   # the filter-block loads (the stream loads are buffer loads too, without idxen)
    ins = blocks[main_blk]
    unit = collections.Counter(); mnem = collections.Counter()
    for i in ins:
        op = i.split()[0]
        u = ("VALU" if op.startswith("v_") else "SALU" if op.startswith("s_") else "LDS" if op.startswith("ds_") else
             "VMEM" if op.startswith(("buffer_", "global_", "flat_")) else "other")
        unit[u] += 1
        if u == "VALU":
            mnem[re.sub(r"_e32$|_e64$|_sdwa$", "", op)] += 1
    # every block of the loop body, in layout order, up to the back edge
    i0 = order.index(main_blk)
    i1 = max(i for i, b in enumerate(order) if any(x.startswith(("s_cbranch", "s_branch")) and x.endswith(main_blk) for x in blocks[b]))
    rows = []
    for b in order[i0:i1 + 1]:
        v = sum(1 for x in blocks[b] if x.startswith("v_"))
        kind = ("drain (lookup pipeline, inlined at each of the 8 positions)" if any(x.startswith(("global_load", "global_atomic")) for x in blocks[b]) else
                "queue push" if any(x.startswith("ds_write_b64") for x in blocks[b]) else
                "phase A + C" if b == main_blk else "phase C of one position" if any("sdwa" in x for x in blocks[b]) else "control")
        rows.append((b, len(blocks[b]), v, kind))
    lines = ["main loop block %s of ntsm_count_mz_kernel<0, false, 128, false>: straight-line part of one 8-position step" % main_blk,
             "(phase A of 8 positions + phase C up to the first position with a positive; the queue push and the drain are",
             " in the blocks behind it and run for the ~54 % of positions where some lane passes the filter)", "",
             "%-8s %8s %12s" % ("unit", "per 8", "per position")]
    for u in ("VALU", "SALU", "LDS", "VMEM", "other"):
        lines.append("%-8s %8d %12.2f" % (u, unit[u], unit[u] / 8.0))
    lines += ["", "VALU by mnemonic (per 8 positions):"]
    for op, n in mnem.most_common():
        lines.append("  %-28s %4d" % (op, n))
    lines += ["", "all blocks of the loop body in layout order (instructions, VALU, role):"]
    by_kind = collections.Counter()
    for b, n, v, kind in rows:
        by_kind[kind] += v
        if kind != "control" or v:
            lines.append("  %-10s %4d %4d  %s" % (b, n, v, kind))
    lines += ["", "static VALU per 8-position step by role (drain blocks run once per >= 64 queued positives, not per step):"]
    for kind, v in by_kind.most_common():
        lines.append("  %-70s %5d" % (kind, v))
    txt = "\n".join(lines) + "\n"
    print(txt)
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        open(os.path.join(out_dir, "isa_budget.txt"), "w").write(txt)
        open(os.path.join(out_dir, "main_loop.s"), "w").write(main_blk + ":\n" + "\n".join("\t" + i for i in ins) + "\n")


if __name__ == "__main__":
    main()
