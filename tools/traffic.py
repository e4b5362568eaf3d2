#!/usr/bin/env python3
"""HBM bytes per launch of the dominant kernel from two separate rocprofv3 PMC passes
(FETCH_SIZE, WRITE_SIZE), as /opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes:
counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) reads, which
is this kernel's read shape (float4 gathers, b128), so the read side is doubled; WRITE_SIZE is
taken as is.  usage: traffic.py <fetch_dir> <write_dir> <kernel substring> <workload> <neighbor> <out.json>"""
import csv
import glob
import json
import sys


def mean_counter(d, counter, kernel):
    vals = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {counter} rows for {kernel} under {d}")
    return sum(vals) / len(vals), len(vals)


fetch_dir, write_dir, kernel, workload, neighbor, out = sys.argv[1:7]
fetch_kib, nf = mean_counter(fetch_dir, "FETCH_SIZE", kernel)
write_kib, nw = mean_counter(write_dir, "WRITE_SIZE", kernel)
res = {
    "workload": workload, "neighbor": int(neighbor), "kernel": kernel,
    "fetch_size_kib_raw": fetch_kib, "write_size_kib_raw": write_kib, "launches": [nf, nw],
    "correction": "read side doubled (gfx950 FETCH_SIZE tallies 128-B requests at 64 B); write side as reported",
    "hbm_bytes_per_launch": (2.0 * fetch_kib + write_kib) * 1024.0,
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
