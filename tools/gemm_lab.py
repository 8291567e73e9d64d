"""Kernel lab for the fp32-MFMA gemm_nt: every LDS-DMA configuration against the register-staged kernel, on the
shapes of the training step (needs a GPU).

    python tools/gemm_lab.py [--rows 565000] [--iters 10]

Prints TFLOP/s (algorithmic 2*M*K*N, HIP events) and the max abs difference to the register-staged result."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drl_urban_planning_amd import native  # noqa: E402
from kernel_bench import P, time_ms  # noqa: E402

VARIANTS = {0: 'register-staged (round 1)', 1: 'dma BK16 minw4', 2: 'dma BK32 minw2', 3: 'dma BK16 minw3',
            4: 'dma BK16 minw2', 5: 'dma BK64 minw1', 6: 'dma BK32 minw1'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=565000)
    ap.add_argument('--iters', type=int, default=10)
    ap.add_argument('--variants', default=','.join(str(v) for v in VARIANTS))
    args = ap.parse_args()
    lib = native.lib()
    dev = 'cuda:0'
    M = args.rows
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(0)
    for name, K, N, resid, bias, act in (('fwd P/Q K=256 N=512', 256, 512, False, False, 0),
                                         ('dgrad   K=512 N=256 +R', 512, 256, True, False, 0),
                                         ('K=256 N=256 +bias tanh', 256, 256, False, True, 1),
                                         ('K=64  N=128', 64, 128, False, False, 0)):
        A = torch.randn(K // 16, M, 16, device=dev)
        W = torch.randn(N, K, device=dev) * 0.05
        R = torch.randn(N // 16, M, 16, device=dev) if resid else None
        b = torch.randn(N, device=dev) if bias else None
        ref = None
        for v in [int(x) for x in args.variants.split(',')]:
            native.check(lib.upamd_tune(b'gemm_nt_dma', v))
            Cc = torch.zeros(N // 16, M, 16, device=dev)
            fn = lambda: native.check(lib.upamd_gemm_nt(P(A), M, K, 0, 0, P(W), N, K, P(b), P(R), P(Cc), 0, 0, act, 1.0, st))
            ms = time_ms(fn, args.iters)
            if ref is None:
                ref = Cc.clone()
                err = 0.0
            else:
                err = float((Cc - ref).abs().max())
            print('%-24s v%d %-26s %.3f ms  %6.1f TFLOP/s  max|d| vs v0 %.2e' % (name, v, VARIANTS.get(v, '?'), ms,
                                                                                 2.0 * M * K * N / ms / 1e9, err), flush=True)
        # ragged tail: M not a multiple of 128
        Mt = 128 * 37 + 5
        At, Ct0, Ct1 = A[:, :Mt].contiguous(), torch.zeros(N // 16, Mt, 16, device=dev), torch.zeros(N // 16, Mt, 16, device=dev)
        Rt = R[:, :Mt].contiguous() if resid else None
        native.check(lib.upamd_tune(b'gemm_nt_dma', 0))
        native.check(lib.upamd_gemm_nt(P(At), Mt, K, 0, 0, P(W), N, K, P(b), P(Rt), P(Ct0), 0, 0, act, 1.0, st))
        for v in [int(x) for x in args.variants.split(',') if int(x)]:
            native.check(lib.upamd_tune(b'gemm_nt_dma', v))
            Ct1.zero_()
            native.check(lib.upamd_gemm_nt(P(At), Mt, K, 0, 0, P(W), N, K, P(b), P(Rt), P(Ct1), 0, 0, act, 1.0, st))
            print('   tail M=%d v%d max|d| %.2e' % (Mt, v, float((Ct1 - Ct0).abs().max())))
    native.check(lib.upamd_tune(b'gemm_nt_dma', 0))


if __name__ == '__main__':
    main()
