"""Timeline view of a rocprofv3 --kernel-trace run (rocpd sqlite output): per optimizer step (delimited by the Adam launch) the
wall time, the time at least one kernel was running, the idle remainder, the summed kernel time (concurrency = sum / busy) and the
largest idle gaps with the kernels on either side.  Usage: python tools/timeline_rocpd.py <results.db> [out.txt] [skip_steps]"""
import sqlite3
import sys
from collections import defaultdict


def short(name):
    n = name.split('(')[0]
    for p in ('_ZN5upamd', 'void ', 'upamd::'):
        n = n.replace(p, '')
    return n[:44]


def main():
    db = sqlite3.connect(sys.argv[1])
    out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    tables = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    disp = [t for t in tables if t.startswith('rocpd_kernel_dispatch')][0]
    sym = [t for t in tables if t.startswith('rocpd_info_kernel_symbol')][0]
    cols = [r[1] for r in db.execute('pragma table_info(%s)' % disp)]
    lane = 'stream_id' if 'stream_id' in cols else ('queue_id' if 'queue_id' in cols else None)
    q = 'select s.kernel_name, d.start, d.end%s from %s d join %s s on d.kernel_id = s.id order by d.start' % (
        (', d.' + lane) if lane else '', disp, sym)
    rows = db.execute(q).fetchall()
    print('columns of %s: %s' % (disp, ', '.join(cols)), file=out)
    print('%d dispatches; lanes by %s' % (len(rows), lane), file=out)
    # steps: from the end of one adam launch to the end of the next
    ends = [r[2] for r in rows if 'adam_groups' in r[0]]
    steps = list(zip(ends[:-1], ends[1:]))[skip:]
    if not steps:
        print('no optimizer steps found', file=out)
        return
    tot = defaultdict(float)
    gaps = []
    lanes = defaultdict(float)
    per_step = []
    by_lane = defaultdict(lambda: [0.0, 0])
    for t0, t1 in steps:
        ks = [r for r in rows if r[1] >= t0 and r[2] <= t1]
        busy, cur_s, cur_e = 0, None, None
        prev_name = None
        last_end_name = None
        for r in sorted(ks, key=lambda r: r[1]):
            if cur_e is None:
                cur_s, cur_e, last_end_name = r[1], r[2], r[0]
                gaps.append((r[1] - t0, 'step start', short(r[0])))
            elif r[1] > cur_e:
                busy += cur_e - cur_s
                gaps.append((r[1] - cur_e, short(last_end_name), short(r[0])))
                cur_s, cur_e, last_end_name = r[1], r[2], r[0]
            elif r[2] > cur_e:
                cur_e, last_end_name = r[2], r[0]
            if lane:
                lanes[r[3]] += r[2] - r[1]
                by_lane[(r[3], short(r[0]))][0] += r[2] - r[1]
                by_lane[(r[3], short(r[0]))][1] += 1
        if cur_e is not None:
            busy += cur_e - cur_s
        summed = sum(r[2] - r[1] for r in ks)
        per_step.append((t1 - t0, busy, summed, len(ks)))
        for r in ks:
            tot[short(r[0])] += r[2] - r[1]
    n = len(per_step)
    wall = sum(p[0] for p in per_step) / n
    busy = sum(p[1] for p in per_step) / n
    summed = sum(p[2] for p in per_step) / n
    print('%d steps: wall %.1f us/step, >= 1 kernel running %.1f us (%.1f %%), idle %.1f us, summed kernel time %.1f us '
          '(concurrency %.2f), %.0f launches/step' % (n, wall / 1e3, busy / 1e3, 100 * busy / wall, (wall - busy) / 1e3,
                                                       summed / 1e3, summed / busy, sum(p[3] for p in per_step) / n), file=out)
    if lane:
        print('summed kernel time per %s (us/step): %s' % (lane, ', '.join('%s: %.0f' % (k, v / n / 1e3) for k, v in
                                                                             sorted(lanes.items(), key=lambda kv: -kv[1]))), file=out)
    if lane:
        for ln in sorted(lanes, key=lambda k: -lanes[k]):
            print('%s %s, us/step (launches/step x average us):' % (lane, ln), file=out)
            for (l2, k), (v, c) in sorted(by_lane.items(), key=lambda kv: -kv[1][0]):
                if l2 == ln and v / n / 1e3 >= 5.0:
                    print('  %8.1f  (%4.1f x %7.1f)  %s' % (v / n / 1e3, c / n, v / c / 1e3, k), file=out)
    agg = defaultdict(lambda: [0.0, 0])
    for g, a, b in gaps:
        agg[(a, b)][0] += g
        agg[(a, b)][1] += 1
    print('idle gaps by (kernel that ended last -> kernel that started next), us/step:', file=out)
    for (a, b), (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
        print('  %8.1f  (%5.1f x %4.1f us)  %-44s -> %s' % (g / n / 1e3, c / n, g / c / 1e3, a, b), file=out)
    print('kernel time by kernel, us/step:', file=out)
    for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:16]:
        print('  %8.1f  %s' % (v / n / 1e3, k), file=out)


if __name__ == '__main__':
    main()
