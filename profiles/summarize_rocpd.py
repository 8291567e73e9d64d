"""Per-kernel summary (count / total / average duration) of a rocprofv3 --kernel-trace run
(rocpd sqlite output).  Usage: python profiles/summarize_rocpd.py <results.db> [out.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tables if t.startswith('rocpd_kernel_dispatch')][0]
    sym = [t for t in tables if t.startswith('rocpd_info_kernel_symbol')][0]
    rows = db.execute("select s.kernel_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
                      "from %s d join %s s on d.kernel_id = s.id group by s.kernel_name order by 3 desc" % (disp, sym)).fetchall()
    total = sum(r[2] for r in rows)
    lines = ['rocprofv3 --kernel-trace summary of %s' % sys.argv[1],
             'total kernel time %.3f ms over %d dispatches' % (total / 1e6, sum(r[1] for r in rows)),
             '%-100s %7s %12s %12s %12s %12s %6s' % ('kernel', 'calls', 'total_ms', 'avg_us', 'min_us', 'max_us', 'pct')]
    for name, n, tot, mn, mx in rows:
        lines.append('%-100s %7d %12.3f %12.2f %12.2f %12.2f %6.2f' % (name[:100], n, tot / 1e6, tot / n / 1e3, mn / 1e3,
                                                                       mx / 1e3, 100.0 * tot / total))
    text = '\n'.join(lines) + '\n'
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(text)
    else:
        sys.stdout.write(text)


if __name__ == '__main__':
    main()
