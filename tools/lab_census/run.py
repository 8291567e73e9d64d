"""Co-residency lab (round 6, review item 1): does the D = 256 step ever hold an MFMA-bound GEMM workgroup and a VALU-bound
message-passing workgroup ON THE SAME CU at the same time, and what does it do to each?  GPU box, after tools/lab_census/build.py:

    python tools/lab_census/run.py --mode serial|wgrad2|two_streams [--tune knob=value,...] [--rows 2048] [--out profiles/x.json]

  serial       the product's default step (GEMMs and walks one after the other on the caller's stream)
  wgrad2       tune knob side_wgrad = 2: layer l's weight-gradient GEMM on a side stream NEXT TO layer l-1's message-passing backward
  lanes        the product's opt-in UPAMD_LANES=2: one updater, every minibatch as two halves on two streams, gradients added
  two_streams  two half-minibatch steps (rows / 2 each, two engines, two streams) enqueued alternately: one half's GEMMs meet the
               other half's walks wherever the hardware dispatcher lets them -- the stream-level form of "software-pipeline two
               half-minibatches"

Every mode is timed twice: on the PRODUCT library (clean step time, no census) and on the census build (one step recorded: per
workgroup the kernel, the CU, start / end on the 100 MHz clock and the shader cycles in between)."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
torch.set_num_threads(1)

KNAMES = {1: 'gemm_nt_dma2', 2: 'gemm_tn_mfma', 3: 'edge_fwd', 4: 'edge_bwd'}
MFMA, WALK = (1, 2), (3, 4)


def covered(starts, ends):
    """piecewise-linear cumulative coverage of the union of [start, end) intervals: returns (t, c) with c(t) = time covered up to t"""
    if len(starts) == 0:
        return np.array([0.0, 1.0]), np.array([0.0, 0.0])
    ev = np.concatenate([starts, ends])
    dl = np.concatenate([np.ones(len(starts)), -np.ones(len(ends))])
    o = np.argsort(ev, kind='stable')
    ev, dl = ev[o], dl[o]
    depth = np.cumsum(dl)
    dt = np.diff(ev)
    c = np.concatenate([[0.0], np.cumsum(dt * (depth[:-1] > 0))])
    return ev, c


def depth_integral(starts, ends):
    """(t, I) with I(t) = integral of the number of intervals open up to t"""
    if len(starts) == 0:
        return np.array([0.0, 1.0]), np.array([0.0, 0.0])
    ev = np.concatenate([starts, ends])
    dl = np.concatenate([np.ones(len(starts)), -np.ones(len(ends))])
    o = np.argsort(ev, kind='stable')
    ev, dl = ev[o], dl[o]
    depth = np.cumsum(dl)
    return ev, np.concatenate([[0.0], np.cumsum(np.diff(ev) * depth[:-1])])


def analyse(rec):
    kid, hw = rec[:, 0], rec[:, 1]
    t0, t1 = rec[:, 2].astype(np.float64) / 100.0, rec[:, 4].astype(np.float64) / 100.0      # us
    cyc = (rec[:, 5] - rec[:, 3]).astype(np.float64)
    ok = (t1 > t0) & (rec[:, 4] != 0)
    kid, hw, t0, t1, cyc = kid[ok], hw[ok], t0[ok], t1[ok], cyc[ok]
    base = t0.min()
    t0, t1 = t0 - base, t1 - base
    cu = ((hw >> 32) << 8) | ((hw >> 8) & 0xFF)              # XCC_ID | SE_ID, SH_ID, CU_ID
    out = {'records': int(ok.sum()), 'dropped_unfinished': int((~ok).sum()), 'distinct_cus': int(len(np.unique(cu))),
           'span_us': float(t1.max())}
    out['per_kernel'] = {}
    for k, name in KNAMES.items():
        m = kid == k
        if m.any():
            d = t1[m] - t0[m]
            out['per_kernel'][name] = {'workgroups': int(m.sum()), 'wg_us_median': float(np.median(d)), 'wg_us_mean': float(d.mean()),
                                       'busy_span_us': float(covered(t0[m], t1[m])[1][-1]),
                                       'clock_ghz_median': float(np.median(cyc[m] / np.maximum(d, 1e-3)) / 1e3)}
    is_m, is_w = np.isin(kid, MFMA), np.isin(kid, WALK)
    # chip level: time with >= 1 MFMA workgroup resident anywhere, >= 1 walk workgroup anywhere, and both
    em, cm = covered(t0[is_m], t1[is_m])
    ew, cw = covered(t0[is_w], t1[is_w])
    grid = np.unique(np.concatenate([em, ew]))
    mid = 0.5 * (grid[1:] + grid[:-1])
    on_m = np.interp(mid + 1e-6, em, cm) - np.interp(mid - 1e-6, em, cm) > 1e-6
    on_w = np.interp(mid + 1e-6, ew, cw) - np.interp(mid - 1e-6, ew, cw) > 1e-6
    dt = np.diff(grid)
    out['chip_us'] = {'mfma_kernels_resident': float((dt * on_m).sum()), 'walk_kernels_resident': float((dt * on_w).sum()),
                      'both_in_flight': float((dt * (on_m & on_w)).sum())}
    # CU level: per workgroup, the share of its lifetime during which ITS CU also held >= 1 workgroup of the other class
    frac = np.zeros(len(kid))
    n_same, n_other = np.zeros(len(kid)), np.zeros(len(kid))      # time-averaged workgroups of the own / the other class on its CU
    cu_both = cu_either = 0.0
    for c in np.unique(cu):
        sel = cu == c
        im, iw = sel & is_m, sel & is_w
        e1, c1 = covered(t0[im], t1[im])
        e2, c2 = covered(t0[iw], t1[iw])
        if im.any() and iw.any():
            frac[im] = (np.interp(t1[im], e2, c2) - np.interp(t0[im], e2, c2)) / (t1[im] - t0[im])
            frac[iw] = (np.interp(t1[iw], e1, c1) - np.interp(t0[iw], e1, c1)) / (t1[iw] - t0[iw])
        d1, i1 = depth_integral(t0[im], t1[im])
        d2, i2 = depth_integral(t0[iw], t1[iw])
        for sel_, (ds, is_), (do, io) in ((im, (d1, i1), (d2, i2)), (iw, (d2, i2), (d1, i1))):
            if sel_.any():
                life = t1[sel_] - t0[sel_]
                n_same[sel_] = (np.interp(t1[sel_], ds, is_) - np.interp(t0[sel_], ds, is_)) / life
                n_other[sel_] = (np.interp(t1[sel_], do, io) - np.interp(t0[sel_], do, io)) / life
        g = np.unique(np.concatenate([e1, e2]))
        md = 0.5 * (g[1:] + g[:-1])
        a = np.interp(md + 1e-6, e1, c1) - np.interp(md - 1e-6, e1, c1) > 1e-6
        b = np.interp(md + 1e-6, e2, c2) - np.interp(md - 1e-6, e2, c2) > 1e-6
        d = np.diff(g)
        cu_both += float((d * (a & b)).sum())
        cu_either += float((d * (a | b)).sum())
    out['cu_level'] = {'cu_us_with_both_classes_resident': cu_both, 'cu_us_with_either': cu_either,
                       'share': cu_both / max(cu_either, 1e-9)}
    out['by_coresidency'] = {}
    for k, name in KNAMES.items():
        m = kid == k
        if not m.any():
            continue
        d, ck = t1[m] - t0[m], cyc[m] / np.maximum(t1[m] - t0[m], 1e-3) / 1e3
        row = {}
        for tag, lo, hi in (('alone (<10 % of its life next to the other class)', -1, 0.1), ('mixed', 0.1, 0.9),
                            ('co-resident (>90 %)', 0.9, 2)):
            s = (frac[m] > lo) & (frac[m] <= hi) if lo >= 0 else frac[m] <= hi
            if s.any():
                row[tag] = {'workgroups': int(s.sum()), 'wg_us_median': float(np.median(d[s])), 'wg_us_mean': float(d[s].mean()),
                            'clock_ghz_median': float(np.median(ck[s])),
                            'own_class_wgs_on_cu_mean': float(n_same[m][s].mean()), 'other_class_wgs_on_cu_mean': float(n_other[m][s].mean()),
                            # workgroups of this kernel a CU in this state retires per microsecond
                            'cu_rate_wg_per_us': float(n_same[m][s].mean() / d[s].mean())}
        out['by_coresidency'][name] = row
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', default='serial', choices=['serial', 'wgrad2', 'two_streams', 'lanes'])
    ap.add_argument('--tune', default='')
    ap.add_argument('--rows', type=int, default=2048)
    ap.add_argument('--steps', type=int, default=12)
    ap.add_argument('--census', type=int, default=1, help='0: time only on the product library; 2: time only on the census build '
                                                           '(its lab knob UPAMD_LAB_EDGE_LDS included); 1: census build, one step recorded')
    ap.add_argument('--out', default='')
    ap.add_argument('--lib-dir', default='csrc', help='csrc | csrc_nt128 (tools/lab_census/build.py nt128)')
    args = ap.parse_args()
    census_lib = os.path.join(ROOT, 'tools', 'lab_census', args.lib_dir, 'libupamd.so')
    if args.census:
        os.environ['UPAMD_LIB_PATH'] = census_lib
    import bench
    from drl_urban_planning_amd import native, PPOUpdater, synth
    assert (native.LIB_PATH == census_lib) == bool(args.census)
    tune = dict(kv.split('=') for kv in args.tune.split(',') if kv)
    if args.mode == 'wgrad2':
        tune.setdefault('side_wgrad', '2')
    for k, v in tune.items():
        native.tune(k, int(v))
    w = dict(bench.WORKLOADS['hlg_d256'])
    dev = torch.device('cuda', 0)
    if args.mode == 'lanes':
        os.environ['UPAMD_LANES'] = '2'         # the PRODUCT's opt-in two-lane step (agent.PPOUpdater._step_lanes)
    lanes = 2 if args.mode == 'two_streams' else 1
    rows = args.rows // lanes
    ups, its, streams = [], [], []
    for lane in range(lanes):
        cfg = bench.model_cfg(w)
        policy_net, value_net, ac = bench.build_networks(cfg, seed=0)
        ac.to(dev)
        up = PPOUpdater(policy_net, value_net, num_optim_epoch=4, mini_batch_size=rows)
        replay = synth.make_replay(max(8 * rows, 4096), w['community'], max_nodes=w['max_nodes'], max_edges=w['max_edges'], seed=100 + lane,
                                   unique=1024)
        np.random.seed(7 + lane)
        st = torch.cuda.Stream(device=dev) if lanes > 1 else torch.cuda.current_stream(dev)
        with torch.cuda.stream(st):
            up.attach()
            it = up.prepare(replay)
        ups.append(up); its.append(it); streams.append(st)
    torch.cuda.synchronize()
    state = [None] * lanes

    def one_round():            # one optimizer step per lane, enqueued alternately
        for lane in range(lanes):
            up, it = ups[lane], its[lane]
            with torch.cuda.stream(streams[lane]):
                if state[lane] is None or state[lane][1] >= state[lane][0].nb:
                    state[lane] = [up.make_epoch(it), 0]
                up.step(it, state[lane][0], state[lane][1])
                state[lane][1] += 1

    for _ in range(6):
        one_round()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_round()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / args.steps
    res = {'mode': args.mode, 'tune': tune, 'rows_per_round': rows * lanes, 'lanes': lanes, 'ms_per_round': ms,
           'samples_per_s': rows * lanes / ms * 1e3, 'library': ('census build' + ('' if args.lib_dir == 'csrc' else ' ' + args.lib_dir)) if args.census else 'product',
           'edge_lds_floor': os.environ.get('UPAMD_LAB_EDGE_LDS')}
    if args.census == 1:
        cap = 700000
        buf = torch.zeros(8 + 8 * cap, dtype=torch.int64, device=dev)
        buf[1] = cap
        torch.cuda.synchronize()
        native.check(native.lib().upamd_tiny_profile(C.c_void_p(buf.data_ptr())), 'census on')
        one_round()
        torch.cuda.synchronize()
        native.check(native.lib().upamd_tiny_profile(C.c_void_p(0)), 'census off')
        n = int(buf[0].item())
        rec = buf[8:8 + 8 * min(n, cap)].view(-1, 8).cpu().numpy()
        res['census'] = analyse(rec)
        res['census']['requested_records'] = n
    line = json.dumps(res)
    print(line)
    if args.out:
        with open(args.out, 'w') as fh:
            fh.write(json.dumps(res, indent=1) + '\n')


if __name__ == '__main__':
    main()
