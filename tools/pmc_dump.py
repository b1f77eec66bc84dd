#!/usr/bin/env python3
"""Per-kernel averages of whatever counters a rocprofv3 --pmc run collected: tools/pmc_dump.py <results.db> [kernel substring]"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
ncol = next(c for c in ("counter_name", "name") if c in cols)
sub = sys.argv[2] if len(sys.argv) > 2 else ""
agg = {}
for k, c, n, v, d in db.execute("select kernel_name, %s, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, %s" % (ncol, ncol)):
    if sub and sub not in k: continue
    k = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", k.replace("void ", "").replace("cunvsm::", ""))[:60]
    agg.setdefault(k, {"n": n, "us": d / 1e3})[c] = v
for k, e in sorted(agg.items(), key=lambda kv: -kv[1]["us"]):
    print(k, " ".join("%s=%.4g" % (a, b) for a, b in e.items()))
