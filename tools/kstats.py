#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 --kernel-trace run (rocpd .db or *_kernel_trace.csv): python tools/kstats.py PATH [nsteps]"""
import sys, sqlite3, csv, collections, glob, os
path = sys.argv[1]
nsteps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = []
dbs = glob.glob(os.path.join(path, "**", "*.db"), recursive=True) if os.path.isdir(path) else [path]
if dbs and dbs[0].endswith(".db"):
    cur = sqlite3.connect(dbs[0]).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = list(cur.execute(f"select s.kernel_name, d.end - d.start from {kd} d join {ks} s on d.kernel_id = s.id"))
else:
    for fn in glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True):
        with open(fn) as f:
            for r in csv.DictReader(f):
                rows.append((r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
agg = collections.defaultdict(lambda: [0, 0.0])
for k, d in rows:
    agg[k][0] += 1
    agg[k][1] += d
print("%-64s %8s %12s %12s %12s" % ("kernel", "calls", "total_ms", "avg_us", "ms/step"))
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-64s %8d %12.3f %12.1f %12.3f" % (k[:64], n, t / 1e6, t / n / 1e3, t / 1e6 / nsteps))
