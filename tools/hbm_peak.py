"""Measured HBM stream rates of the box (needs a GPU): read-only (sum), copy (read + write) and fill (write-only) over buffers far
larger than the 256 MB infinity cache, HIP-event timed.  The nominal 8 TB/s of MI355X_MICROARCH.md is what `roofline.peak` uses;
this prints what simple streaming kernels actually reach, which is the yardstick for the attention / head kernels.

    python tools/hbm_peak.py [--gb 4] [--iters 20]"""
import argparse

import torch


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gb', type=float, default=4.0)
    ap.add_argument('--iters', type=int, default=20)
    args = ap.parse_args()
    n = int(args.gb * 2 ** 30 / 4)
    x = torch.randn(n, device='cuda:0')
    y = torch.empty_like(x)
    by = n * 4
    t = timed(lambda: y.copy_(x), args.iters)
    print('copy  (read + write) %.2f GB each way: %.3f ms  %.0f GB/s' % (by / 1e9, t * 1e3, 2 * by / t / 1e9))
    t = timed(lambda: y.fill_(1.0), args.iters)
    print('fill  (write)        %.2f GB: %.3f ms  %.0f GB/s' % (by / 1e9, t * 1e3, by / t / 1e9))
    t = timed(lambda: x.sum(), args.iters)
    print('sum   (read)         %.2f GB: %.3f ms  %.0f GB/s' % (by / 1e9, t * 1e3, by / t / 1e9))
    a = x[: n // 2]
    t = timed(lambda: torch.add(a, a, out=y[: n // 2]), args.iters)
    print('add   (read + write) %.2f GB each way: %.3f ms  %.0f GB/s' % (by / 2e9, t * 1e3, by / t / 1e9))
    # at the size of one attention launch (0.6 GB read): does a 0.25 ms kernel get the same rate?
    m = int(0.6e9 / 4)
    t = timed(lambda: x[:m].sum(), args.iters)
    print('sum   (read)         0.60 GB: %.3f ms  %.0f GB/s' % (t * 1e3, m * 4 / t / 1e9))
    t = timed(lambda: y[:m].copy_(x[:m]), args.iters)
    print('copy  (read + write) 0.60 GB each way: %.3f ms  %.0f GB/s' % (t * 1e3, 2 * m * 4 / t / 1e9))


if __name__ == '__main__':
    main()
