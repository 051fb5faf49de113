#!/usr/bin/env python3
"""All-pairs scoring rate of include/ntsm_eval_hip.h (ntsmEval's computeScore on the GPU): S samples x 96287 sites.
Prints pairs/s and pair-sites/s from the library's HIP events, and the oracle's single-thread rate on a sample of pairs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ntsm_amd.eval as ev

m = 96287
rng = np.random.default_rng(7)
for n in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "256,1024,2048").split(",")]:
    g = rng.integers(0, 3, size=(n, m))
    d = 8.0
    c = np.zeros((n, m, 2), dtype=np.uint32)
    c[:, :, 0] = rng.poisson(np.where(g == 0, d, np.where(g == 1, d / 2, 0.02)))
    c[:, :, 1] = rng.poisson(np.where(g == 2, d, np.where(g == 1, d / 2, 0.02)))
    ev.pairs(c[:8], 1)                                   # warm-up
    t0 = time.perf_counter(); rec, ms = ev.pairs(c, 1); wall = time.perf_counter() - t0
    pairs = n * (n - 1) // 2
    print("S=%d: %d pairs x %d sites: pair kernel %.1f ms = %.3g pairs/s = %.3g pair-sites/s (whole call incl. transfers %.2f s)"
          % (n, pairs, m, ms, pairs / (ms / 1e3), pairs * m / (ms / 1e3), wall), flush=True)
