#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs: per-kernel stats and PMC counter sums per dispatch.

usage: rocpd_summary.py <results.db> [...]   (prints markdown; used to produce profiles/*.md)
"""
import sqlite3
import sys
from collections import defaultdict


def kernel_stats(db):
    rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                      "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("| kernel | calls | total ms | avg ms | min ms | max ms | % |")
    print("|---|---|---|---|---|---|---|")
    for name, n, tot, avg, mn, mx in rows:
        print(f"| `{name[:90]}` | {n} | {tot/1e6:.3f} | {avg/1e6:.3f} | {mn/1e6:.3f} | {mx/1e6:.3f} | {100*tot/total:.1f} |")


def counters(db):
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    try:
        rows = db.execute("select kernel_name, dispatch_id, counter_name, sum(value) from counters_collection "
                          "group by kernel_name, dispatch_id, counter_name").fetchall()
    except sqlite3.OperationalError:
        print("columns:", cols)
        return
    if not rows:
        return
    per = defaultdict(lambda: defaultdict(list))
    for k, d, c, v in rows:
        per[k][c].append(v)
    print("\n| kernel | counter | dispatches | mean per dispatch |")
    print("|---|---|---|---|")
    for k, cs in per.items():
        for c, vs in sorted(cs.items()):
            print(f"| `{k[:70]}` | {c} | {len(vs)} | {sum(vs)/len(vs):.6g} |")


for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    print(f"\n### {path}\n")
    kernel_stats(db)
    counters(db)
