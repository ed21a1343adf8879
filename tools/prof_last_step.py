"""The kernel sequence of the LAST train step in a rocprofv3 rocpd database (a replayed hipGraph step of bench.py): one line per launch with
start offset, duration and the gap to the previous kernel's end.  python tools/prof_last_step.py p_results.db [anchor-substring]
The step is cut at the last-but-one launch of the anchor kernel (default: adam_kernel, once per step)."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
anchor = sys.argv[2] if len(sys.argv) > 2 else "adam_kernel"
objs = c.execute("select name, type from sqlite_master").fetchall()
view = ([n for n, t in objs if n.startswith("kernels")] or [n for n, t in objs if "kernel_dispatch" in n])[0]
cols = [r[1] for r in c.execute(f"pragma table_info({view})")]


def pick(*keys):
    for k in cols:
        if all(s in k.lower() for s in keys):
            return k
    return None


name, gx, st, en = pick("name"), pick("grid", "x"), pick("start"), pick("end")
rows = sorted(c.execute(f"select {name}, {gx}, {st}, {en} from {view}").fetchall(), key=lambda r: r[2])
marks = [i for i, r in enumerate(rows) if anchor in r[0]]
lo, hi = marks[-2] + 1, marks[-1] + 1
t0, prev_end, gaps, busy = rows[lo][2], rows[lo][2], 0.0, 0.0
print(f"# {hi - lo} launches between the last two '{anchor}' launches")
for n, g, s, e in rows[lo:hi]:
    gap = (s - prev_end) / 1e3
    gaps += max(gap, 0.0)
    busy += (e - s) / 1e3
    short = n.replace("void ", "").replace("ofa::", "")[:86]
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f}  gap {gap:6.1f}  grid {g:>9}  {short}")
    prev_end = max(prev_end, e)
print(f"# span {(rows[hi - 1][3] - t0) / 1e3:.1f} us: kernels {busy:.1f} us, gaps {gaps:.1f} us")
