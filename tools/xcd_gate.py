#!/usr/bin/env python3
"""tools/xcd_gate.py -- the arithmetic behind DESIGN.md section 4.2b's verdict on the XCD-owned first level (round 3 verdict,
item 4), from the measured counters of the two-level form (profiles/r04_stress_traffic.json) and the measured caps of
DESIGN.md section 4.0.  Prints the gate's two numbers (fabric reads per base, HBM-side bound) and the third one it did not
list (L2 requests per base of both passes against the 266 G/s cap)."""
import json, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
t = json.load(open(os.path.join(ROOT, "profiles", "r04_stress_traffic.json")))
L2_CAP, FABRIC_CAP, HBM = 266e9, 56e9, 8e12
runs_per_kmer, bloom_pass = 0.294, 0.30            # tools/sim_two_level.cpp on the 16 M-key set, 2.25 MiB Bloom
valid = 132.0 / 150.0                              # windows per base of a 150 bp read
records = runs_per_kmer * bloom_pass * valid       # run records per base = block requests per base of today's form
rec_bytes = (12, 16)
print("today (measured): %.3f L2 requests, %.3f fabric reads per base, %.0f Gbases/s; fabric %.1f G/s of %.0f, L2 %.0f G/s of %.0f"
      % (t["l2_requests_per_base"], t["fabric_read_requests_per_base"], t["gbases_per_s_unprofiled"],
         t["fabric_request_rate_G_per_s"], FABRIC_CAP / 1e9, t["l2_requests_per_base"] * t["gbases_per_s_unprofiled"], L2_CAP / 1e9))
print("run records per base: %.3f" % records)
for b in rec_bytes:
    print("  %d-byte records: %.2f B/base written + read again -> HBM-side bound %.2f Tbases/s (gate: > 0.7)" % (b, 2 * b * records, HBM / (1.053 + 2 * b * records) / 1e12))
fabric = 0.008 + 0.02 + 0.015
print("fabric reads per base, two passes: stream 0.008 + Bloom misses <= 0.02 + buckets 0.015 = %.3f (gate: <= 0.07) -> fabric bound %.0f Gbases/s" % (fabric, FABRIC_CAP / fabric / 1e9))
p1 = t["l2_requests_per_base"]                      # the record store replaces the block load
p2 = 2 * records + 0.01                             # record load + block load per record, flush headers
tot = p1 + p2
print("L2 requests per base: pass 1 %.3f + pass 2 %.3f = %.3f -> ceiling %.0f Gbases/s at the cap, %.0f at the 83 %% the best kernel reaches (today: %.0f)"
      % (p1, p2, tot, L2_CAP / tot / 1e9, 0.83 * L2_CAP / tot / 1e9, t["gbases_per_s_unprofiled"]))

# ---------------------------------------------------------------------------------------------------------------------------------
# Round 6: the same two-pass idea with LINE-COMBINED stores.  The rejection above charges one L2 request per
# record; a workgroup that keeps P partition buffers of 128 bytes in LDS and flushes whole lines issues one 64-byte store request per
# four 16-byte records instead.  Redone with that, from round 5's counters (profiles/r05_stress_traffic.json) and what the requests
# of the shipped two-level form are made of (per base 0.259 Bloom words + 0.10 blocks + 0.0156 stream + ~0.015 buckets
# = 0.39; measured 0.378).  Go only if the model reaches 550 Gbases/s at 16 M keys.
t5 = json.load(open(os.path.join(ROOT, "profiles", "r05_stress_traffic.json")))
ACHIEVED_OF_L2_CAP = 0.835                       # the best kernel here (configs[1]) runs at 83.5 % of the 266 G/s cap
VALU_PER_POS_AT_912 = 34.9                       # configs[1]: 34.9 wave-instructions per position at 912 Gbases/s, 86.7 % VALU-busy
bloom_req = 0.294 * valid                        # one Bloom word per 14-mer minimizer run
rec = t5["fabric_read_requests_per_base"] - 0.008 - 0.014 - 0.02   # runs that pass the Bloom = block requests of today's form (0.097)
stream_req, bucket_req = 1.0 / 64.0, 0.03        # 64-byte stream requests; bucket load + counter atomic per look-up (true hits + 1.7 % false positives)
print("\nround 6: two passes with line-combined record stores")
print("records per base (measured): %.3f" % rec)
for name, rec_bytes, P, lds_slices, extra_pass in (("A: Bloom in pass 1, P = 64 partitions, slices of 512 KiB stay in the L2", 16, 64, False, False),
                                                    ("A': as A with a re-partition pass to 512 slices of 64 KiB, tested out of LDS", 16, 64, True, True),
                                                    ("B: no Bloom, EVERY run becomes a record, P = 320 slices of 100 KiB tested out of LDS", 12, 320, True, False)):
    n_rec = rec if not name.startswith("B") else bloom_req
    store_req = n_rec * rec_bytes / 64.0                                   # line-combined: 64-byte requests
    p1 = stream_req + (bloom_req if not name.startswith("B") else 0.0) + store_req + n_rec / 8.0 / 16.0   # + one reservation atomic per 16 lines
    p1b = 2 * store_req if extra_pass else 0.0                             # read + write every record once more
    p2 = store_req + (0.0 if lds_slices else n_rec) + bucket_req           # record lines in, one block request per record unless the slice is in LDS
    l2 = p1 + p1b + p2
    lines = 0.008 + (0.02 if not name.startswith("B") else 0.0) + (2 + (2 if extra_pass else 0)) * n_rec * rec_bytes / 128.0 + 0.015 + 32.0 * 2 ** 20 / 128.0 / 1.5e11
    hbm_bytes = 1.053 + (2 + (2 if extra_pass else 0)) * n_rec * rec_bytes
    lds_kib = P * 128 / 1024.0
    wg_per_cu = int(160 // (38.1 + lds_kib))
    # issue: pass 1 = today's phases A (+ B) without the block test (-4) + the partition push (+8 per record-carrying position share);
    # pass 2 = ~55 instructions per record (pop, rebuild <= 6 k-mers' bits, test, queue positives) -- the run kernel's per-run cost
    valu = (39.7 - 4.0 - (6.0 if name.startswith("B") else 0.0)) + 8.0 * n_rec / valid * valid + 55.0 * n_rec
    occ_penalty = {4: 1.0, 3: 0.85, 2: 0.7, 1: 0.5}.get(wg_per_cu, 0.5)     # round 5: three workgroups per CU instead of four cost 15 %
    ceil_l2 = ACHIEVED_OF_L2_CAP * L2_CAP / l2 / 1e9
    ceil_fab = FABRIC_CAP / lines / 1e9
    ceil_hbm = 6.29e12 / hbm_bytes / 1e9                                    # measured copy ceiling, MI355X_MICROARCH.md
    ceil_valu = 912.0 * VALU_PER_POS_AT_912 / valu * occ_penalty
    best = min(ceil_l2, ceil_fab, ceil_hbm, ceil_valu)
    print("  %s" % name)
    print("    L2 requests/base: pass 1 %.3f%s + pass 2 %.3f = %.3f -> %.0f Gbases/s at %.1f %% of the cap (%.0f at 100 %%)"
          % (p1, (" + re-partition %.3f" % p1b) if extra_pass else "", p2, l2, ceil_l2, 100 * ACHIEVED_OF_L2_CAP, L2_CAP / l2 / 1e9))
    print("    fabric lines/base %.3f -> %.0f; HBM bytes/base %.2f -> %.0f; LDS for P x 128 B: %.0f KiB beside 38.1 -> %d workgroups per CU; ~%.0f vector instructions/position -> %.0f"
          % (lines, ceil_fab, hbm_bytes, ceil_hbm, lds_kib, wg_per_cu, valu, ceil_valu))
    print("    model: %.0f Gbases/s (today %.0f; gate 550) -> %s" % (best, t5["gbases_per_s_unprofiled"], "GO" if best >= 550 else "no go"))
