#!/usr/bin/env python3
"""tools/isa_hash.py [--check FILE | --write FILE] [-D...] -- per-kernel hash of the gfx950 ISA the compiler emits for the
kernel translation units of libntsm_hip.so (ntsm_amd/csrc/kernels_generic.hip, kernels_mz.hip).

A kernel's hash is the SHA-256 (first 16 hex digits) of its instructions with comments, directives and local label numbers
stripped, so it changes when -- and only when -- the code the GPU runs changes.  Used to show that a refactoring of the
sources around the kernels (round 5: the split of ntsm_hip.hip into six translation units, the ablation hooks moved behind
ntsm_hooks.h) left the product kernels bit-identical, and by tools/make_traffic.py-style profile gating.
  --write FILE   record the hashes (profiles/r05_isa_hashes.json)
  --check FILE   compare with a recorded file; exit 1 and list the kernels that differ
Extra -D flags go to the compiler (e.g. -DNTSM_WITH_TAB)."""
import hashlib, json, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ["ntsm_amd/csrc/kernels_generic.hip", "ntsm_amd/csrc/kernels_mz.hip", "ntsm_amd/csrc/kernels_run.hip"]


def kernel_hashes(defs=(), sources=SOURCES):
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for src in sources:
            asm = os.path.join(td, "k.s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-fvisibility=hidden", "-S", "--cuda-device-only"] + list(defs) +
                                  ["-o", asm, os.path.join(ROOT, src)], stderr=subprocess.DEVNULL, cwd=td)
            text = open(asm).read()
            for m in re.finditer(r"^(_Z\w+):\s*(?:;.*)?$", text, re.M):
                name = m.group(1)
                body = text[m.start():text.index(".Lfunc_end", m.start())]
                lines = [l.split(";")[0].rstrip() for l in body.split("\n")]
                lines = [re.sub(r"\.LBB\d+_", ".LBBx_", l) for l in lines if l.strip() and not l.strip().startswith(".")]
                demangled = subprocess.run(["c++filt", name], stdout=subprocess.PIPE).stdout.decode().strip()
                demangled = demangled.replace("(anonymous namespace)::", "")
                out[demangled] = {"instructions": len(lines) - 1, "sha16": hashlib.sha256("\n".join(lines[1:]).encode()).hexdigest()[:16]}
    return out


def main():
    args = sys.argv[1:]
    defs = [a for a in args if a.startswith("-D")]
    h = kernel_hashes(defs)
    if "--write" in args:
        json.dump({"flags": defs, "kernels": h}, open(args[args.index("--write") + 1], "w"), indent=1, sort_keys=True)
    for k in sorted(h):
        print("%s %6d  %s" % (h[k]["sha16"], h[k]["instructions"], k))
    if "--check" in args:
        ref = json.load(open(args[args.index("--check") + 1]))["kernels"]
        bad = [k for k in ref if k in h and h[k]["sha16"] != ref[k]["sha16"]] + [k for k in ref if k not in h]
        if bad:
            print("DIFFERENT from the recorded ISA: " + ", ".join(sorted(bad)))
            return 1
        print("all %d recorded kernels unchanged" % len(ref))
    return 0


if __name__ == "__main__":
    sys.exit(main())
