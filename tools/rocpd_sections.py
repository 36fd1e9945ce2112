"""Attribute the kernels of a rocprofv3 (rocpd sqlite) kernel trace to sections delimited by xq_marker_kernel launches
(XQ_MARKERS=1; the marker's workgroup count is the id of the section that STARTS there... more precisely every kernel is
attributed to the last marker seen before it).  Prints per-section GPU time per step and the top kernels of each section.
   python tools/rocpd_sections.py <results.db> [top_per_section [sections,to,detail]] > profiles/<name>.txt"""
import collections
import sqlite3
import sys

NAMES = {20: "encoder fwd", 21: "quantizer fwd (+perturbation)", 22: "decoder fwd", 23: "semantic teacher + sem loss", 24: "(between model and loss)",
         30: "rec + LPIPS fwd", 31: "DiffAug + DinoDisc fwd (generator)", 32: "LPIPS bwd", 33: "DinoDisc bwd (generator)",
         34: "last-layer grads + surrogate", 35: "(loss tail)", 50: "backward: decoder", 60: "backward: quantizer", 61: "backward: encoder",
         62: "grad all-reduce launch", 40: "disc step: fake fwd", 41: "disc step: real fwd", 42: "disc step: loss", 43: "disc step: bwd",
         44: "disc step: optimizer", 45: "(wait)", 70: "AdamW + EMA", 71: "(after step)"}

db = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 6
detail = set(int(v) for v in sys.argv[3].split(",")) if len(sys.argv) > 3 else None   # sections to list with `top`; others get 3
con = sqlite3.connect(db)
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
gcol = [c for c in cols if c.lower() in ("grid_x", "grid_size_x", "grid_size")]
wcol = [c for c in cols if c.lower() in ("workgroup_x", "workgroup_size_x", "workgroup_size")]
if not gcol:
    raise SystemExit(f"no grid column in kernels view: {cols}")
q = f"select name, start, end, {gcol[0]}" + (f", {wcol[0]}" if wcol else ", 64") + " from kernels order by start"
rows = con.execute(q).fetchall()
sec = None
per = collections.defaultdict(float)
cnt = collections.defaultdict(int)
kern = collections.defaultdict(lambda: collections.defaultdict(float))
steps = 0
for name, st, en, gx, wx in rows:
    if "xq_marker_kernel" in name:
        sid = int(gx) // int(wx) if int(gx) % int(wx) == 0 and int(gx) >= int(wx) else int(gx)
        if sid == 20:
            steps += 1
        sec = sid
        continue
    if sec is None:
        continue
    per[sec] += en - st
    cnt[sec] += 1
    kern[sec][name] += en - st
steps = max(steps, 1)
tot = sum(per.values())
print(f"# {db.split('/')[-1]}: {steps} steps, {tot/steps/1e6:.2f} ms GPU kernel time per step (sum over sections)")
print(f"{'ms/step':>9} {'share%':>7} {'kernels':>8}  section")
order = [20, 21, 22, 23, 24, 30, 31, 32, 33, 34, 35, 50, 60, 61, 62, 40, 41, 42, 43, 44, 45, 70, 71]
for sid in order + sorted(set(per) - set(order)):
    if sid not in per:
        continue
    print(f"{per[sid]/steps/1e6:9.2f} {100*per[sid]/tot:7.2f} {cnt[sid]//steps:8d}  [{sid}] {NAMES.get(sid, '?')}")
    ncalls = collections.Counter()
    for n, t in sorted(kern[sid].items(), key=lambda kv: -kv[1])[:(top if detail is None or sid in detail else 3)]:
        print(f"{'':>27}{t/steps/1e6:8.2f} ms  {n[:110]}")
