#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace database of tools/slab_overlap.py and reports, for the boundary-first schedule, how much
of the exchange kernels' time (k_slab_pack, k_slab_headers, k_slab_unpack_dev, k_slab_commit) ran WHILE an SPH-pass kernel
was executing, i.e. hidden behind it.  usage: slab_overlap_trace.py <results.db> [skip first N launches of k_slab_pack=12]"""
import json
import sqlite3
import sys

db = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 12
con = sqlite3.connect(db)
rows = con.execute("select name, start, end from kernels order by start").fetchall()
sph = [(s, e) for n, s, e in rows if "k_sph_walk" in n]
exch = [(n.split("(")[0].split("::")[-1], s, e) for n, s, e in rows if "k_slab_pack" in n or "k_slab_headers" in n or "k_slab_unpack_dev" in n or "k_slab_commit" in n]
packs = [x for x in exch if x[0].startswith("k_slab_pack")]
t0 = packs[skip][1] if len(packs) > skip else 0          # steady state: past the priming exchange and the sequential-schedule run
sph = [(s, e) for s, e in sph if e >= t0]
exch = [x for x in exch if x[1] >= t0]


def covered(s, e):
    tot = 0
    for a, b in sph:
        lo, hi = max(s, a), min(e, b)
        if hi > lo:
            tot += hi - lo
    return min(tot, e - s)


tot = sum(e - s for _, s, e in exch)
hid = sum(covered(s, e) for _, s, e in exch)
by = {}
for n, s, e in exch:
    d = by.setdefault(n, [0, 0, 0])
    d[0] += 1; d[1] += e - s; d[2] += covered(s, e)
print(json.dumps({"exchange_kernel_launches": len(exch), "exchange_kernel_time_us": round(tot / 1e3, 1), "of_it_while_an_sph_kernel_runs_us": round(hid / 1e3, 1),
                  "hidden_fraction": round(hid / max(tot, 1), 3),
                  "per_kernel": {k: {"launches": v[0], "avg_us": round(v[1] / v[0] / 1e3, 2), "hidden_fraction": round(v[2] / max(v[1], 1), 3)} for k, v in by.items()}}))
