import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tables if t.startswith('rocpd_kernel_dispatch')][0]
sym = [t for t in tables if t.startswith('rocpd_info_kernel_symbol')][0]
rows = db.execute('select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start' % (disp, sym)).fetchall()
print(len(rows), 'dispatches, span %.1f ms' % ((rows[-1][2] - rows[0][1]) / 1e6))
gaps = []
for (n0, s0, e0), (n1, s1, e1) in zip(rows[:-1], rows[1:]):
    gaps.append((s1 - e0, n0[:50], n1[:50], (s1 - rows[0][1]) / 1e6))
gaps.sort(reverse=True)
for g in gaps[:15]:
    print('gap %.3f ms at t=%.1f ms  %s -> %s' % (g[0] / 1e6, g[3], g[1], g[2]))
durs = sorted(((e - s) / 1e3, n[:50], (s - rows[0][1]) / 1e6) for n, s, e in rows)[-8:]
print('longest kernels (us):', durs)
