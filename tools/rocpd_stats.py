"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share.
   python tools/rocpd_stats.py <results.db> [top_n] > profiles/<name>.txt"""
import sqlite3, sys
db = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = con.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {name_col} order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
print(f"# rocprofv3 --kernel-trace summary of {db.split('/')[-2]}/{db.split('/')[-1]}: {len(rows)} kernels, total GPU kernel time {total/1e6:.3f} ms")
print(f"{'share%':>7} {'calls':>7} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>9}  kernel")
for n, c, t, a, mn, mx in rows[:top]:
    print(f"{100*t/total:7.2f} {c:7d} {t/1e6:10.3f} {a/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f}  {n[:150]}")
