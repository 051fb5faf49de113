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
