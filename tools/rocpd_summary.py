#!/usr/bin/env python3
"""Turns rocprofv3's rocpd (.db, sqlite) outputs into the small text summaries kept under profiles/.

    rocpd_summary.py stats  <results.db> <out.csv>            # --kernel-trace --stats run -> per-kernel table
    rocpd_summary.py pmc    <results.db> <out.json> [substr]  # --pmc run -> mean counter values per kernel
    rocpd_summary.py traffic <fetch.db> <write.db> <kernel substr> <workload> <neighbor> <out.json>
    rocpd_summary.py window <stats.db> <kernel substr> <first launch> <count> <out.json>     # exactly bench.py's timed launches
    rocpd_summary.py bench-counters <fetch.db> <write.db> <valu.db> <kernel> <first> <count> <workload> <source> <out.json> [<tcp.db> <csrc hash>]

traffic: HBM bytes per launch of the dominant kernel as /opt/skills/guides/MI355X_MICROARCH.md prescribes
(FETCH_SIZE / WRITE_SIZE from separate passes, KiB units; on gfx950 FETCH_SIZE tallies 128-byte requests at
64 B, and this kernel's reads are 16 B/lane gathers, so the read side is doubled; writes as reported)."""
import csv
import json
import sqlite3
import sys


def stats(db, out):
    con = sqlite3.connect(db)
    rows = con.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        for r in rows:
            w.writerow([r[0], r[1], round(r[2], 3), round(r[3], 3), round(r[4], 3)])
    return rows


def pmc_means(db, substr=""):
    con = sqlite3.connect(db)
    res = {}
    q = "select kernel_name, counter_name, avg(value), count(*), avg(duration) from counters_collection group by kernel_name, counter_name"
    for name, counter, mean, cnt, dur in con.execute(q):
        if substr and substr not in name:
            continue
        short = name.split("(")[0]
        res.setdefault(short, {})[counter] = {"mean": mean, "launches": cnt, "mean_duration_ns": dur}
    return res


def window_stats(db, substr, first, count):
    """Durations (us) of launches [first, first + count) of the kernel whose name contains substr, in start order."""
    con = sqlite3.connect(db)
    rows = con.execute("select name, start, end from kernels order by start").fetchall()
    d = [(e - s) / 1e3 for n, s, e in rows if substr in n]
    w = d[first:first + count]
    return {"kernel": substr, "launches_in_run": len(d), "first_launch": first, "launches": len(w),
            "average_us": sum(w) / max(len(w), 1), "min_us": min(w) if w else None, "max_us": max(w) if w else None,
            "average_us_all_launches": sum(d) / max(len(d), 1)}


def window_counter(db, substr, counter, first, count):
    """Mean of a PMC counter over launches [first, first + count) of the kernel (dispatch order)."""
    con = sqlite3.connect(db)
    q = "select kernel_name, dispatch_id, value from counters_collection where counter_name = ? order by dispatch_id"
    v = [val for name, did, val in con.execute(q, (counter,)) if substr in name]
    w = v[first:first + count]
    return {"mean": sum(w) / max(len(w), 1), "launches": len(w), "launches_in_run": len(v)}


def main():
    mode = sys.argv[1]
    if mode == "window":          # window <stats.db> <kernel substr> <first> <count> <out.json>
        res = window_stats(sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]))
        json.dump(res, open(sys.argv[6], "w"), indent=1)
        print(json.dumps(res))
        return
    if mode == "bench-counters":  # bench-counters <fetch.db> <write.db> <valu.db> <kernel> <first> <count> <workload> <source text> <out.json>
        fdb, wdb, vdb, kernel, first, count, workload, source, out = sys.argv[2:11]
        tdb = sys.argv[11] if len(sys.argv) > 11 else None          # optional: TCP_TOTAL_CACHE_ACCESSES_sum pass
        csrc = sys.argv[12] if len(sys.argv) > 12 else None         # optional: hash of the engine sources the passes ran on
        first, count = int(first), int(count)
        f = window_counter(fdb, kernel, "FETCH_SIZE", first, count)
        w = window_counter(wdb, kernel, "WRITE_SIZE", first, count)
        v = window_counter(vdb, kernel, "SQ_INSTS_VALU", first, count)
        res = {"workload": workload, "kernel": kernel, "first_launch": first, "launches": count,
               "fetch_size_kib_raw": f["mean"], "write_size_kib_raw": w["mean"],
               "correction": "read side doubled (gfx950 FETCH_SIZE tallies 128-B requests at 64 B); write side as reported",
               "hbm_bytes_per_launch": (2.0 * f["mean"] + w["mean"]) * 1024.0, "valu_wave_insts_per_launch": v["mean"], "source": source}
        if tdb:
            res["tcp_line_accesses_per_launch"] = window_counter(tdb, kernel, "TCP_TOTAL_CACHE_ACCESSES_sum", first, count)["mean"]
        if csrc:
            res["csrc_hash"] = csrc
        json.dump(res, open(out, "w"), indent=1)
        print(json.dumps(res))
        return
    if mode == "stats":
        for r in stats(sys.argv[2], sys.argv[3])[:6]:
            print(r)
    elif mode == "pmc":
        res = pmc_means(sys.argv[2], sys.argv[4] if len(sys.argv) > 4 else "")
        json.dump(res, open(sys.argv[3], "w"), indent=1)
        print(json.dumps(res)[:600])
    elif mode == "traffic":
        fdb, wdb, kernel, workload, neighbor, out = sys.argv[2:8]
        f = [v for k, v in pmc_means(fdb, kernel).items()][0]["FETCH_SIZE"]
        w = [v for k, v in pmc_means(wdb, kernel).items()][0]["WRITE_SIZE"]
        res = {"workload": workload, "neighbor": int(neighbor), "kernel": kernel,
               "fetch_size_kib_raw": f["mean"], "write_size_kib_raw": w["mean"], "launches": [f["launches"], w["launches"]],
               "correction": "read side doubled (gfx950 FETCH_SIZE tallies 128-B requests at 64 B); write side as reported",
               "hbm_bytes_per_launch": (2.0 * f["mean"] + w["mean"]) * 1024.0}
        json.dump(res, open(out, "w"), indent=1)
        print(json.dumps(res))


if __name__ == "__main__":
    main()
