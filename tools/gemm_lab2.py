"""Kernel lab 2: first-residency-round stagger of the LDS-DMA gemm_nt (needs a GPU)."""
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
    for name, K, N, resid in (('fwd P/Q K=256 N=512', 256, 512, False), ('dgrad K=512 N=256 +R', 512, 256, True),
                              ('K=256 N=256', 256, 256, False)):
        A = torch.randn(K // 16, M, 16, device=dev)
        W = torch.randn(N, K, device=dev) * 0.05
        R = torch.randn(N // 16, M, 16, device=dev) if resid else None
        Cc = torch.zeros(N // 16, M, 16, device=dev)
        fn = lambda: native.check(lib.upamd_gemm_nt(P(A), M, K, 0, 0, P(W), N, K, None, P(R), P(Cc), 0, 0, 0, 1.0, st))
        for v in (4, 1):
            native.check(lib.upamd_tune(b'gemm_nt_dma', v))
            for mode in (0, 1, 2, 3):
                for cyc in ((0,) if mode == 0 else (15000, 37000, 60000)):
                    native.check(lib.upamd_tune(b'gemm_stagger_mode', mode))
                    native.check(lib.upamd_tune(b'gemm_stagger_cycles', cyc))
                    ms = time_ms(fn, 10)
                    print('%-22s v%d stagger mode %d cycles %6d  %.3f ms  %6.1f TFLOP/s' % (name, v, mode, cyc, ms,
                                                                                          2.0 * M * K * N / ms / 1e9), flush=True)
    native.check(lib.upamd_tune(b'gemm_stagger_mode', 0))
    native.check(lib.upamd_tune(b'gemm_nt_dma', 0))


if __name__ == '__main__':
    main()
