"""Per-(kernel, grid) averages of every PMC counter of a rocprofv3 --pmc run: python pmc_dump.py p_results.db [name filter]."""
import sqlite3, sys, collections
c = sqlite3.connect(sys.argv[1]); flt = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
pm = [t for t in tabs if t.startswith('rocpd_pmc_event')][0]
sym = [t for t in tabs if 'kernel_symbol' in t][0]
info = [t for t in tabs if t.startswith('rocpd_info_pmc')][0]
names = {r[0]: r[1] for r in c.execute(f"select id, name from {info}")}
cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
gx = [k for k in cols if 'grid' in k.lower() and 'x' in k.lower()][0]
st = [k for k in cols if 'start' in k.lower()][0]; en = [k for k in cols if k.lower() == 'end' or 'end' in k.lower()][0]
per = collections.defaultdict(lambda: collections.defaultdict(float)); meta = {}
q = f"select s.kernel_name, p.pmc_id, p.value, d.id, d.{gx}, d.{st}, d.{en} from {pm} p join {kd} d on p.event_id = d.event_id join {sym} s on d.kernel_id = s.id"
for name, pid, val, did, g, s, e in c.execute(q):
    if flt in name:
        per[did][names.get(pid, str(pid))] += val
        meta[did] = (name[:60], g, (e - s) / 1e3)
agg = collections.defaultdict(lambda: [0, 0.0, collections.defaultdict(float)])
for did, cs in per.items():
    k = meta[did][:2]
    a = agg[k]; a[0] += 1; a[1] += meta[did][2]
    for n, v in cs.items(): a[2][n] += v
for (name, g), (n, us, cs) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    extra = ""
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and cs.get("GRBM_GUI_ACTIVE", 0) > 0:      # (1024 SIMDs, 8 XCDs: MI355X)
        extra += f"  [mfma {100.0 * (cs['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024) / (cs['GRBM_GUI_ACTIVE'] / 8):4.1f}%"
        if cs.get("SQ_LDS_IDX_ACTIVE", 0) > 0:
            extra += f" | lds-conflict {100.0 * cs.get('SQ_LDS_BANK_CONFLICT', 0.0) / cs['SQ_LDS_IDX_ACTIVE']:4.1f}%"
        extra += "]"
    print(f"{name} grid {g} calls {n} avg {us/n:.1f} us  " + "  ".join(f"{k}={v/n:.3g}" for k, v in sorted(cs.items())) + extra)
