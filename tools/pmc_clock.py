import sqlite3, sys, glob
for db in sorted(glob.glob(sys.argv[1] + '/*.db')):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if 'kernel_dispatch' in t][0]
    pm = [t for t in tabs if 'pmc_event' in t]
    sym = [t for t in tabs if 'kernel_symbol' in t][0]
    cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
    rows = c.execute(f"select d.id, s.kernel_name, d.end - d.start from {kd} d join {sym} s on d.kernel_id = s.id").fetchall()
    pmrows = {}
    if pm:
        pc = [r[1] for r in c.execute(f"pragma table_info({pm[0]})")]
        for r in c.execute(f"select event_id, value from {pm[0]}"):
            pmrows.setdefault(r[0], []).append(r[1])
    for (i, name, dur) in rows:
        if 'gemm_big' in name:
            v = pmrows.get(i, [])
            print(db.split('/')[-1], name[:40], 'dur_us', dur / 1e3, 'pmc', v, 'clk_GHz', [x / dur for x in v])
