"""Micro-benchmark of the fp32-MFMA GEMM building blocks through the C ABI (needs a GPU).

    python tools/kernel_bench.py [--rows 565000] [--iters 20]

Prints achieved TFLOP/s (algorithmic 2*M*K*N) per shape; random data, HIP-event timing."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drl_urban_planning_amd import native  # noqa: E402


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def time_ms(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rows', type=int, default=565000)
    ap.add_argument('--iters', type=int, default=20)
    args = ap.parse_args()
    lib = native.lib()
    dev = 'cuda:0'
    M = args.rows
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for name, K, N, resid in (('fwd  P/Q  K=256 N=512', 256, 512, False), ('dgrad    K=512 N=256 +R', 512, 256, True),
                              ('node enc K=32  N=256', 32, 256, False), ('head     K=1024 N=32', 1024, 32, False)):
        A = torch.randn(K // 16, M, 16, device=dev)
        W = torch.randn(N, K, device=dev) * 0.05
        R = torch.randn(N // 16, M, 16, device=dev) if resid else None
        Cc = torch.empty(N // 16, M, 16, device=dev)
        fn = lambda: native.check(lib.upamd_gemm_nt(P(A), M, K, 0, 0, P(W), N, K, None, P(R), P(Cc), 0, 0, 0, 1.0, st))
        ms = time_ms(fn, args.iters)
        print('gemm_nt %-26s M=%d  %.3f ms  %.1f TFLOP/s' % (name, M, ms, 2.0 * M * K * N / ms / 1e9))
    for name, I, J in (('wgrad I=512 J=256', 512, 256), ('node enc I=256 J=32', 256, 32)):
        A = torch.randn(I // 16, M, 16, device=dev)
        B = torch.randn(J // 16, M, 16, device=dev)
        scratch = torch.empty(int(lib.upamd_gemm_tn_scratch_floats(I, J, M)), device=dev)
        out = torch.empty(I, J, device=dev)
        fn = lambda: native.check(lib.upamd_gemm_tn(P(A), I, 0, P(B), J, 0, M, 0, P(scratch), P(out), st))
        ms = time_ms(fn, args.iters)
        print('gemm_tn %-26s M=%d  %.3f ms  %.1f TFLOP/s' % (name, M, ms, 2.0 * M * I * J / ms / 1e9))


if __name__ == '__main__':
    main()
