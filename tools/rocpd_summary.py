#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (ROCm 7.2 default output, *_results.db): per-kernel launch count and duration
statistics, and -- for --pmc runs -- the counter values of each kernel's LAST `keep` dispatches (sum over dimensions), as
JSON / CSV text for profiles/.   python tools/rocpd_summary.py DB [--kernel SUBSTR] [--last N] [--csv]"""
import json
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    sub = sys.argv[sys.argv.index("--kernel") + 1] if "--kernel" in sys.argv else None
    last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 4
    c = db.cursor()
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    rows = c.execute("select * from kernels").fetchall()
    ix = {n: i for i, n in enumerate(cols)}
    by = defaultdict(list)
    for r in rows:
        by[r[ix["name"]]].append(r)
    out = {"kernels": [], "counters": []}
    for name, rs in sorted(by.items(), key=lambda kv: -sum(r[ix["duration"]] for r in kv[1])):
        if sub and sub not in name:
            continue
        d = sorted(r[ix["duration"]] for r in rs)
        out["kernels"].append({"name": name[:120], "launches": len(rs), "total_ms": round(sum(d) / 1e6, 4), "avg_us": round(sum(d) / len(d) / 1e3, 2),
                               "median_us": round(d[len(d) // 2] / 1e3, 2), "min_us": round(d[0] / 1e3, 2), "max_us": round(d[-1] / 1e3, 2),
                               **{k: rs[-1][ix[k]] for k in ("grid_size_x", "workgroup_size_x", "lds_size", "scratch_size", "vgpr_count", "accum_vgpr_count", "sgpr_count") if k in ix}})
    if "--clusters" in sys.argv:                  # launches of one kernel NAME span many shapes (a 256-ray shard ... a whole frame): group each
        for k in out["kernels"]:                  # kernel's durations into clusters of +-2 % and report the populous ones (count, mean, spread)
            d = sorted(r[ix["duration"]] for r in by[[n for n in by if n[:120] == k["name"]][0]])
            cl, cur = [], [d[0]]
            for v in d[1:]:
                if v <= cur[0] * 1.04:
                    cur.append(v)
                else:
                    cl.append(cur)
                    cur = [v]
            cl.append(cur)
            k["clusters"] = [{"launches": len(c), "mean_us": round(sum(c) / len(c) / 1e3, 2), "min_us": round(c[0] / 1e3, 2), "max_us": round(c[-1] / 1e3, 2)}
                             for c in sorted(cl, key=lambda c: -len(c))[:8] if len(c) >= 3]
    if "--timeline" in sys.argv:                  # the last N dispatches in time order: start relative to the first of them, duration, gap to the previous end
        n = int(sys.argv[sys.argv.index("--timeline") + 1])
        tl = sorted(rows, key=lambda r: r[ix["start"]])[-n:]
        t0, prev = tl[0][ix["start"]], None
        out["timeline"] = []
        for r in tl:
            out["timeline"].append({"name": r[ix["name"]][:60], "start_us": round((r[ix["start"]] - t0) / 1e3, 2), "dur_us": round(r[ix["duration"]] / 1e3, 2),
                                    "gap_us": None if prev is None else round((r[ix["start"]] - prev) / 1e3, 2)})
            prev = r[ix["end"]]
    try:
        ccols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
        cix = {n: i for i, n in enumerate(ccols)}
        crow = c.execute("select * from counters_collection").fetchall()
    except sqlite3.Error:
        crow = []
    if crow:
        per = defaultdict(lambda: defaultdict(float))          # (kernel, dispatch_id) -> counter -> value
        meta = {}
        for r in crow:
            key = (r[cix["kernel_name"]], r[cix["dispatch_id"]])
            per[key][r[cix["counter_name"]]] += r[cix["value"]]
            meta[key] = (r[cix["start"]], r[cix["end"]]) if "start" in cix else (0, 0)
        names = defaultdict(list)
        for (k, d) in per:
            names[k].append(d)
        for k, ds in names.items():
            if sub and sub not in k:
                continue
            for d in sorted(ds)[-last:]:
                s, e = meta[(k, d)]
                out["counters"].append({"kernel": k[:120], "dispatch": d, "duration_us": round((e - s) / 1e3, 2), **{n: v for n, v in sorted(per[(k, d)].items())}})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
