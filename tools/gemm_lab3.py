"""Kernel lab 3: where the time of the LDS-DMA gemm_nt goes -- K sweep (slope = main loop, intercept = per-tile fixed cost)
with and without the C write (needs a GPU)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drl_urban_planning_amd import native  # noqa: E402
from kernel_bench import P, time_ms  # noqa: E402


def main():
    lib = native.lib()
    dev = 'cuda:0'
    M = 565000
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(0)
    native.check(lib.upamd_tune(b'gemm_nt_dma', 1))
    native.check(lib.upamd_tune(b'gemm_stagger_cycles', 37000))
    for N in (512, 256, 128):
        for K in (64, 128, 256, 512, 1024):
            A = torch.randn(K // 16, M, 16, device=dev)
            W = torch.randn(N, K, device=dev) * 0.05
            Cc = torch.zeros(N // 16, M, 16, device=dev)
            fn = lambda: native.check(lib.upamd_gemm_nt(P(A), M, K, 0, 0, P(W), N, K, None, None, P(Cc), 0, 0, 0, 1.0, st))
            out = []
            for mode in (0, 1, 8, 9):
                native.check(lib.upamd_tune(b'gemm_stagger_mode', mode))
                ms = time_ms(fn, 8)
                out.append('mode %d: %.3f ms %6.1f TF' % (mode, ms, 2.0 * M * K * N / ms / 1e9))
            print('N=%4d K=%4d  ' % (N, K) + ' | '.join(out), flush=True)
            del A, W, Cc
    native.check(lib.upamd_tune(b'gemm_stagger_mode', 0))
    native.check(lib.upamd_tune(b'gemm_nt_dma', 0))


if __name__ == '__main__':
    main()
