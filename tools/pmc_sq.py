"""Per-kernel SQ counter table from rocprofv3 PMC passes (any counters): mean per launch of every counter found under the
given directories, one row per kernel.

    rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d out/sqA -- python bench.py --steps 3 ...
    python tools/pmc_sq.py out/sqA out/sqB --md profiles/rNN_pmc_sq_edge.md --match edge_

Units (MI355X guide): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_INST_CYCLES_* count quad-cycles summed over all
waves; SQ_BUSY_CYCLES per SE; GRBM_GUI_ACTIVE summed over the 8 XCDs."""
import argparse
import csv
import glob
import os
import sys
from collections import defaultdict

sys.path.insert(0, __file__.rsplit('/', 1)[0])
from pmc_traffic import short_name  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('directories', nargs='+')
    ap.add_argument('--md')
    ap.add_argument('--match', default='')
    args = ap.parse_args()
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for d in args.directories:
        for path in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            per_dispatch = defaultdict(float)
            names = {}
            with open(path) as fh:
                for row in csv.DictReader(fh):
                    key = (row['Dispatch_Id'], row['Counter_Name'])
                    per_dispatch[key] += float(row['Counter_Value'])
                    names[row['Dispatch_Id']] = row['Kernel_Name']
            for (disp, ctr), v in per_dispatch.items():
                a = agg[short_name(names[disp])][ctr]
                a[0] += v
                a[1] += 1
    counters = sorted({c for k in agg.values() for c in k})
    lines = ['| kernel | launches | ' + ' | '.join(counters) + ' |', '|---|---|' + '---|' * len(counters)]
    for name in sorted(agg):
        if args.match and args.match not in name:
            continue
        a = agg[name]
        n = max(v[1] for v in a.values())
        lines.append('| %s | %d | ' % (name, n) + ' | '.join('%.4g' % (a[c][0] / max(a[c][1], 1)) if c in a else '-' for c in counters) + ' |')
    text = '\n'.join(lines)
    if args.md:
        with open(args.md, 'w') as fh:
            fh.write('# SQ counters per kernel launch (rocprofv3 PMC; mean per launch, summed over the chip)\n\n%s\n' % text)
    print(text)


if __name__ == '__main__':
    main()
