"""Per-(kernel, grid size) averages from a rocprofv3 rocpd database: python prof_by_grid.py p_results.db [name filter] [min calls]."""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
objs = c.execute("select name, type from sqlite_master").fetchall()
cand = [n for n, t in objs if n.startswith("kernels")] or [n for n, t in objs if "kernel_dispatch" in n]
if not cand:
    print("no kernel view; objects:", objs); sys.exit(1)
view = cand[0]
cols = [r[1] for r in c.execute(f"pragma table_info({view})")]
def pick(*keys):
    for k in cols:
        if all(s in k.lower() for s in keys): return k
    return None
name, gx, wx, st, en = pick("name"), pick("grid", "x"), pick("workgroup", "x"), pick("start"), pick("end")
if not (name and gx and st and en):
    print("columns:", cols); sys.exit(1)
agg = collections.defaultdict(list)
for n, g, w, s, e in c.execute(f"select {name}, {gx}, {wx or gx}, {st}, {en} from {view}"):
    if flt in n: agg[(n[:70], g, w)].append((e - s) / 1e3)
for (n, g, w), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{sum(v)/1e3:8.2f} ms {len(v):6d} calls  avg {sum(v)/len(v):8.1f} us  med {v2[len(v2)//2]:8.1f}  grid {g:>8} wg {w:>5}  {n}")
