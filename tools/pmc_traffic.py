"""Summarise rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE) into per-kernel HBM bytes per launch.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out/fetch -- python bench.py --steps 4 --warmup 1 ...
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out/write -- python bench.py --steps 4 --warmup 1 ...
    python tools/pmc_traffic.py out/fetch out/write --json profiles/pmc_traffic.json --md profiles/rNN_pmc_step_traffic.md

Corrections (MI355X_MICROARCH.md, HBM / rocprofv3 section): both counters are in KB; on gfx950 FETCH_SIZE counts a
128-byte request as 64 bytes for wide coalesced reads, so the fetch figure is DOUBLED; WRITE_SIZE is used as is.
Counter passes are separate runs with --kernel-trace only (never combined with the HIP/HSA trace domains).
"""
import argparse
import csv
import glob
import json
import os
import re
from collections import defaultdict

from csrc_hash import csrc_hash

SHORT = [      # demangled (csv output) and mangled (rocpd) spellings
    (r'gemm_nt_dma2_kernel(<2, 2|ILi2ELi2)', 'gemm_nt_128'),          # round 2: the LDS-DMA kernel is the 128 x 128 node GEMM
    (r'gemm_nt_mfma_kernel(<128, 64, 64, false, false, 1, 32|ILi128ELi64ELi64ELb0ELb0ELi1ELi32)', 'gemm_nt_128_k32'),
    (r'gemm_nt_mfma_kernel(<128, 64, 64, false, false|ILi128ELi64ELi64ELb0ELb0)', 'gemm_nt_128'),
    (r'gemm_nt_mfma_kernel(<128, 64, 64, true|ILi128ELi64ELi64ELb1)', 'gemm_nt_128_rm'),
    (r'gemm_nt_mfma_kernel(<64|ILi64)', 'gemm_nt_64'),
    (r'gemm_nt_mfma_kernel(<32, 32, 32, false, false|ILi32ELi32ELi32ELb0ELb0)', 'gemm_nt_32'),
    (r'gemm_nt_mfma_kernel(<32|ILi32)', 'gemm_nt_32_rm'),
    (r'gemm_tn_mfma_kernel(<128, 64, 64, false|ILi128ELi64ELi64ELb0)', 'gemm_tn_128'),
    (r'gemm_tn_mfma_kernel(<128|ILi128)', 'gemm_tn_128_rm'),
    (r'gemm_tn_mfma_kernel(<32, 32, 32, false|ILi32ELi32ELi32ELb0)', 'gemm_tn_32'),
    (r'gemm_tn_mfma_kernel(<32|ILi32)', 'gemm_tn_32_rm'),
    (r'edge_fwd_kernel(<false, true, true|ILb0ELb1ELb1)', 'edge_fwd_layer1_folded'),
    (r'edge_bwd_kernel(<false, true, true|ILb0ELb1ELb1)', 'edge_bwd_layer1_folded'),
    (r'edge_fwd_kernel(<false|ILb0)', 'edge_fwd'),
    (r'edge_fwd_kernel(<true|ILb1)', 'edge_fwd_last'),
    (r'edge_bwd_kernel(<false|ILb0)', 'edge_bwd'),
    (r'edge_bwd_kernel(<true|ILb1)', 'edge_bwd_last'),
]


def short_name(k):
    for pat, name in SHORT:
        if re.search(pat, k):
            return name
    m = re.search(r'upamd(?:::|\d+)(\w+?)_kernel', k)
    return m.group(1) if m else re.sub(r'^void ', '', k)[:40]


def read_counter(directory, counter):
    """{kernel: [values in KB per dispatch]} from every *counter_collection.csv under `directory`."""
    out = defaultdict(list)
    files = glob.glob(os.path.join(directory, '**', '*counter_collection.csv'), recursive=True)
    if not files:
        raise SystemExit('no *counter_collection.csv under %s' % directory)
    for f in files:
        with open(f, newline='') as fh:
            per_dispatch = defaultdict(float)
            names = {}
            for row in csv.DictReader(fh):
                if row.get('Counter_Name') != counter:
                    continue
                key = (row.get('Process_Id'), row.get('Dispatch_Id'))
                per_dispatch[key] += float(row['Counter_Value'])          # summed over XCDs / instances
                names[key] = row['Kernel_Name']
            for key, v in per_dispatch.items():
                out[names[key]].append(v)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('fetch_dir')
    ap.add_argument('write_dir')
    ap.add_argument('--json')
    ap.add_argument('--md')
    ap.add_argument('--command', default='')
    args = ap.parse_args()
    fetch = read_counter(args.fetch_dir, 'FETCH_SIZE')
    write = read_counter(args.write_dir, 'WRITE_SIZE')
    rows = {}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, []), write.get(k, [])
        name = short_name(k)
        r = rows.setdefault(name, dict(launches=0, fetch_bytes=0.0, write_bytes=0.0, wl=0))
        r['launches'] += len(f)
        r['wl'] += len(w)
        r['fetch_bytes'] += 2.0 * 1024.0 * sum(f)      # KB -> bytes, gfx950 doubling
        r['write_bytes'] += 1024.0 * sum(w)
    table = {}
    for name, r in rows.items():
        n = max(r['launches'], 1)
        nw = max(r['wl'], 1)
        table[name] = dict(launches=r['launches'], fetch_bytes_per_launch=r['fetch_bytes'] / n,
                           write_bytes_per_launch=r['write_bytes'] / nw,
                           hbm_bytes_per_launch=r['fetch_bytes'] / n + r['write_bytes'] / nw)
    doc = dict(source='rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only); FETCH_SIZE doubled '
                      '(gfx950), KB -> bytes', command=args.command, kernels=table, csrc_hash=csrc_hash())
    if args.json:
        with open(args.json, 'w') as fh:
            json.dump(doc, fh, indent=1, sort_keys=True)
    lines = ['| kernel | launches | fetch MB / launch (corrected) | write MB / launch | HBM MB / launch |', '|---|---|---|---|---|']
    for name, t in sorted(table.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch'] * kv[1]['launches']):
        lines.append('| %s | %d | %.1f | %.1f | %.1f |' % (name, t['launches'], t['fetch_bytes_per_launch'] / 1e6,
                                                         t['write_bytes_per_launch'] / 1e6, t['hbm_bytes_per_launch'] / 1e6))
    text = '\n'.join(lines)
    if args.md:
        with open(args.md, 'w') as fh:
            fh.write('# HBM traffic per launch of one PPO step (rocprofv3 PMC)\n\n%s\n\n%s\n\n%s\n' % (doc['source'], args.command, text))
    print(text)


if __name__ == '__main__':
    main()
