"""Per-kernel / per-grid durations out of a rocprofv3 rocpd database (t_results.db).  usage: rocpd_summary.py DB [filter]"""
import sqlite3, collections, sys
db = sqlite3.connect(sys.argv[1]); flt = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
names = {r[0]: r[1] for r in db.execute(f"select id, kernel_name from {ks}")}
rows = list(db.execute(f"select kernel_id, start, end, grid_size_x, workgroup_size_x from {kd} order by start"))
agg = collections.defaultdict(lambda: [0, 0.0])
for kid, s, e, g, w in rows:
    n = names[kid]
    if flt not in n: continue
    key = (n[:64], g // max(w, 1))
    agg[key][0] += 1; agg[key][1] += (e - s) / 1e3
print(f"{len(rows)} dispatches, kernel time {sum(v[1] for v in agg.values()):.0f} us, span {(rows[-1][2] - rows[0][1]) / 1e3:.0f} us")
for k, v in sorted(agg.items(), key=lambda x: -x[1][1])[:40]:
    print(f"  {v[0]:6d} x {v[1] / v[0]:8.2f} us = {v[1]:10.1f}  grid {k[1]:6d}  {k[0]}")
