#!/usr/bin/env python3
"""tools/sim_tab_filter.py -- host simulation of tabulation-hashed minimizer filters (design study, numpy only).

For the hs_n10_like site set and the bench's synthetic reads it measures, per candidate design:
  * minimizer change density (first-level L2 requests per valid window),
  * false-positive rate of the first-level filter on read k-mers that are NOT site k-mers,
  * how many site k-mers share a block (skew).
Designs: minimizer length m = 12 (three 4-mer table lookups) or m = 10 (two 5-mer lookups), block width 16 / 32 /
128 bits, pattern bits per piece.  Everything is a strand-symmetric function of forward-strand pieces, so no
reverse complement is needed per position.  Not on the product path.
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ntsm_amd  # noqa: E402

K = 19
rng = np.random.default_rng(12345)


def rc_index(y, h):
    """index of the reverse complement of an h-mer given little-endian 2-bit packing (first base lowest)"""
    out = np.zeros_like(y)
    for t in range(h):
        b = (y >> (2 * t)) & 3
        out |= (3 - b) << (2 * (h - 1 - t))
    return out


def tables(h, nbits_pat, width):
    n = 4 ** h
    y = np.arange(n, dtype=np.int64)
    r = rc_index(y, h)
    A = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    Cc = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    U = A ^ Cc[r]
    V = Cc ^ A[r]
    P = np.zeros(n, dtype=np.uint64)
    for _ in range(nbits_pat):
        P |= np.uint64(1) << rng.integers(0, width, n, dtype=np.uint64)
    return U, V, P, P[r]


def kmers_from_codes(codes, valid):
    """per position p (window = codes[p-18..p]): fw / rc big-endian codes, window validity"""
    n = len(codes)
    fw = np.zeros(n, dtype=np.uint64)
    rv = np.zeros(n, dtype=np.uint64)
    ok = np.ones(n, dtype=bool)
    c = codes.astype(np.uint64)
    for j in range(K):                       # base j of the window ending at p is codes[p - 18 + j]
        sh = np.zeros(n, dtype=np.uint64)
        sh[K - 1 - j:] = c[:n - (K - 1 - j)] if K - 1 - j else c
        v = np.zeros(n, dtype=bool)
        v[K - 1 - j:] = valid[:n - (K - 1 - j)] if K - 1 - j else valid
        fw |= sh << np.uint64(2 * (K - 1 - j))
        rv |= (np.uint64(3) - sh) << np.uint64(2 * j)
        ok &= v
    return fw, rv, ok


def piece_index(codes, h):
    """e_h(p): little-endian index of the h-mer ending at p"""
    n = len(codes)
    e = np.zeros(n, dtype=np.int64)
    for t in range(h):
        sh = np.zeros(n, dtype=np.int64)
        d = h - 1 - t
        sh[d:] = codes[:n - d] if d else codes
        e |= sh << (2 * t)
    return e


def shift(a, d, fill=0):
    out = np.full_like(a, fill)
    if d == 0:
        return a.copy()
    out[d:] = a[:-d]
    return out


def design_keys(codes, h, m, U, V):
    e = piece_index(codes, h)
    if m == 2 * h:
        key = U[shift(e, h)] ^ V[e]
    elif m == 3 * h:
        M = (U.astype(np.uint64) + V.astype(np.uint64)).astype(np.uint32)
        key = U[shift(e, 2 * h)] ^ M[shift(e, h)] ^ V[e]
    else:
        raise ValueError
    w = K - m + 1
    mz = key.copy()
    for d in range(1, w):
        mz = np.minimum(mz, shift(key, d, fill=0xFFFFFFFF))
    return e, key, mz


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
    tmp = tempfile.mkdtemp()
    sp = os.path.join(tmp, "s.fa")
    synth = ntsm_amd.SynthShort(20241218, 96287, k=K, read_seed=7, read_len=150, sites_path=sp)
    sites = ntsm_amd.Sites(sp, k=K)
    keys = np.sort(sites.keys)
    print("site k-mers: %d" % len(keys))
    # site k-mers as base sequences (forward = the canonical code's own bases), padded so that position 18 is the window end
    site_codes = np.zeros((len(keys), K), dtype=np.int64)
    for j in range(K):
        site_codes[:, j] = (keys >> np.uint64(2 * (K - 1 - j))) & np.uint64(3)
    flat_sites = site_codes.reshape(-1)                  # windows at positions 18, 37, ... of the concatenation
    ends = np.arange(len(keys)) * K + (K - 1)

    raw = synth.host_bytes(0, n_reads)
    lut = np.full(256, 4, dtype=np.int64)
    for ch, v in zip(b"ACGTUacgtu", [0, 1, 2, 3, 3, 0, 1, 2, 3, 3]):
        lut[ch] = v
    for v in range(4):
        lut[v] = v
    cd = lut[raw]
    valid = cd < 4
    codes = np.where(valid, cd, 0)
    fw, rv, ok = kmers_from_codes(codes, valid)
    canon = np.minimum(fw, rv)
    is_site = np.isin(canon, keys) & ok
    print("read positions %d, valid windows %d, true hits %d (%.3f%%)" % (len(codes), ok.sum(), is_site.sum(), 100.0 * is_site.sum() / ok.sum()))

    for (h, m, width, nb, nblk_log2, mult) in [
            (4, 12, 32, 2, 18, 3), (4, 12, 16, 2, 19, 3), (4, 12, 32, 2, 19, 1), (4, 12, 32, 3, 18, 3), (4, 12, 32, 1, 18, 3),
            (5, 10, 32, 2, 18, 3), (5, 10, 32, 2, 19, 1), (5, 10, 16, 2, 19, 3), (5, 10, 32, 3, 18, 3),
            (6, 12, 32, 2, 18, 3), (4, 12, 64, 2, 17, 3), (5, 10, 64, 2, 17, 3), (5, 10, 64, 3, 17, 3)]:
        U, V, PA, PB = tables(h, nb, width)
        n_blocks = mult << nblk_log2
        # site side
        e_s, key_s, mz_s = design_keys(flat_sites, h, m, U, V)
        mz_site = mz_s[ends]
        blk_site = ((mz_site.astype(np.uint64) >> np.uint64(2)) % np.uint64(n_blocks)).astype(np.int64)
        pat_site = PA[e_s[ends]] | PB[e_s[ends - (K - h)]]
        blocks = np.zeros(n_blocks, dtype=np.uint64)
        np.bitwise_or.at(blocks, blk_site, pat_site)
        occ = np.bincount(blk_site, minlength=n_blocks)
        # read side
        e_r, key_r, mz_r = design_keys(codes, h, m, U, V)
        blk_r = ((mz_r.astype(np.uint64) >> np.uint64(2)) % np.uint64(n_blocks)).astype(np.int64)
        pat_r = PA[e_r] | PB[shift(e_r, K - h)]
        passed = (blocks[blk_r] & pat_r) == pat_r
        neg = ok & ~is_site
        fp = (passed & neg).sum() / neg.sum()
        miss = (is_site & ~passed).sum()
        prev_ok = shift(ok, 1, fill=False)
        change = ok & ((mz_r != shift(mz_r, 1)) | ~prev_ok)
        dens = change.sum() / ok.sum()
        dens_all = ((mz_r != shift(mz_r, 1))).sum() / len(mz_r)
        bytes_total = n_blocks * width // 8
        # occupancy seen by queries
        qocc = occ[blk_r[neg]]
        print("h=%d m=%2d width=%3d bits/piece=%d blocks=%8d (%4.1f MiB): density %.4f (all positions %.4f)  FP %.3f%%  false negatives %d  "
              "keys/block mean %.2f max %d, seen by queries mean %.2f  distinct site minimizers %d"
              % (h, m, width, nb, n_blocks, bytes_total / 2 ** 20, dens, dens_all, 100 * fp, miss, occ.mean(), occ.max(), qocc.mean(),
                 len(np.unique(mz_site))))


if __name__ == "__main__":
    main()
