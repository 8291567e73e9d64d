"""Per-kernel pipe utilisation from ONE rocprofv3 PMC pass (MI355X):

    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace \
        --output-format csv -d out/util -- python bench.py --steps 4 --warmup 1 --cpu-baseline off --no-kernel-events
    python tools/pmc_util.py out/util --md profiles/rNN_pmc_utilisation.md

Normalisation (profiles/archive/r01_pmc_utilisation.md): GRBM_GUI_ACTIVE is summed over the 8 XCDs (/ 8 = kernel duration in
cycles); SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs (/ 1024 = cycles a SIMD's matrix pipe is busy);
SQ_ACTIVE_INST_VALU counts 4-cycle units per SIMD (/ 1024 * 4)."""
import argparse
import sys
from collections import defaultdict

sys.path.insert(0, __file__.rsplit('/', 1)[0])
from pmc_traffic import read_counter, short_name  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('directory')
    ap.add_argument('--md')
    ap.add_argument('--json')
    args = ap.parse_args()
    names = ('GRBM_GUI_ACTIVE', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_ACTIVE_INST_VALU')
    data = {c: read_counter(args.directory, c) for c in names}
    agg = defaultdict(lambda: defaultdict(float))
    for c in names:
        for k, vals in data[c].items():
            agg[short_name(k)][c] += sum(vals)
            agg[short_name(k)]['n_' + c] += len(vals)
    lines = ['| kernel | launches | duration (k cycles) | MFMA pipe busy | VALU busy |', '|---|---|---|---|---|']
    rows = []
    js = {}
    for name, a in agg.items():
        n = max(a['n_GRBM_GUI_ACTIVE'], 1)
        dur = a['GRBM_GUI_ACTIVE'] / 8.0 / n
        if dur <= 0:
            continue
        mfma = a['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / n / dur
        valu = a['SQ_ACTIVE_INST_VALU'] * 4.0 / 1024.0 / n / dur
        rows.append((dur * n, '| %s | %d | %.0f | %.0f %% | %.0f %% |' % (name, n, dur / 1e3, 100 * mfma, 100 * valu)))
        js[name] = {'launches': int(n), 'duration_cycles': dur, 'mfma_busy': mfma, 'valu_busy': valu}
    lines += [r for _, r in sorted(rows, reverse=True)]
    text = '\n'.join(lines)
    if args.md:
        with open(args.md, 'w') as fh:
            fh.write('# Pipe utilisation per kernel of one PPO step (rocprofv3 PMC, one pass)\n\n'
                     'rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace '
                     '(hlg_d256, bench.py --steps 4 --warmup 1)\n\n%s\n' % text)
    if args.json:
        import json
        from csrc_hash import csrc_hash
        with open(args.json, 'w') as fh:
            json.dump({'kernels': js, 'source': 'rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES '
                                                'SQ_ACTIVE_INST_VALU --kernel-trace over bench.py (hlg_d256)',
                       'csrc_hash': csrc_hash()}, fh, indent=1)
    print(text)


if __name__ == '__main__':
    main()
