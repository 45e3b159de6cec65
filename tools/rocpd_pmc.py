"""Per-kernel averages of PMC counters from rocprofv3 rocpd databases: python tools/rocpd_pmc.py <db> [kernel-substring]"""
import sqlite3
import sys

db = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else None
cols = [r[1] for r in c.execute(f"pragma table_info({view})")]
namec = "kernel_name" if "kernel_name" in cols else "name"
rows = c.execute(f"select {namec}, counter_name, count(*), avg(value), min(value), max(value) from {view} group by {namec}, counter_name").fetchall()
for r in rows:
    if sub in r[0]:
        print(f"{r[0][:70]:70s} {r[1]:28s} n={r[2]:4d} avg={r[3]:.6g} min={r[4]:.6g} max={r[5]:.6g}")
