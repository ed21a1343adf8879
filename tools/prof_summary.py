"""Per-kernel summary of a rocprofv3 --kernel-trace --stats run (the results .db): ms per step, calls per step, average duration.
usage: prof_summary.py <p_results.db> [steps] [rows] [--json out.json]
The JSON form ({kernel name: {ms_per_step, calls_per_step, avg_us}} + "__meta__") is what bench.py reads back from profiles/ to price
the GEMM family on the trace of the REPLAYED graph (roofline.rocprof)."""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from build_id import source_id  # noqa: E402

argv = [a for a in sys.argv[1:] if not a.startswith("--json")]
jpath = None
if "--json" in sys.argv:
    jpath = sys.argv[sys.argv.index("--json") + 1]
    argv = [a for a in argv if a != jpath]
c = sqlite3.connect(argv[0])
steps = float(argv[1]) if len(argv) > 1 else 1
rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
tot = sum(r[2] for r in rows)
print(f"total {tot/1e3/steps:.2f} ms/step over {steps} steps")
for r in rows[:int(argv[2]) if len(argv) > 2 else 30]:
    print(f"{r[2]/1e3/steps:7.3f} ms/step {r[1]/steps:7.1f} calls {r[3]:8.1f} us  {r[4]:5.1f}%  {r[0][:100]}")
if jpath:
    out = {"__meta__": {"steps": steps, "total_ms_per_step": tot / 1e3 / steps, "source": "rocprofv3 --kernel-trace --stats, top_kernels",
                        "source_id": source_id()}}      # the sources the traced command ran on (tools/build_id.py)
    for r in rows:
        out[r[0]] = {"ms_per_step": r[2] / 1e3 / steps, "calls_per_step": r[1] / steps, "avg_us": r[3]}
    json.dump(out, open(jpath, "w"), indent=0)
