"""Kernel lab 5: software-pipelined LDS-DMA gemm_nt variants (needs a GPU)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drl_urban_planning_amd import native  # noqa: E402
from kernel_bench import P, time_ms  # noqa: E402

NAMES = {0: 'register-staged', 1: 'dma simple', 2: 'dma pipelined', 3: 'dma pipelined+setprio', 4: 'dma pipelined asm reads minw4',
         5: 'asm reads minw3', 6: 'asm reads minw2'}


def main():
    lib = native.lib()
    dev = 'cuda:0'
    M = 565000
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(0)
    native.check(lib.upamd_tune(b'gemm_stagger_cycles', 37000))
    for name, K, N, resid, zero in (('fwd P/Q K=256 N=512', 256, 512, False, False), ('dgrad K=512 N=256 +R', 512, 256, True, False),
                                    ('K=256 N=256', 256, 256, False, False), ('K=1024 N=512 zeros', 1024, 512, False, True)):
        A = torch.zeros(K // 16, M, 16, device=dev) if zero else torch.randn(K // 16, M, 16, device=dev)
        W = torch.randn(N, K, device=dev) * 0.05
        R = torch.randn(N // 16, M, 16, device=dev) if resid else None
        ref = None
        for v in range(7):
            native.check(lib.upamd_tune(b'gemm_nt_dma', v))
            Cc = torch.zeros(N // 16, M, 16, device=dev)
            fn = lambda: native.check(lib.upamd_gemm_nt(P(A), M, K, 0, 0, P(W), N, K, None, P(R), P(Cc), 0, 0, 0, 1.0, st))
            res = []
            for mode in (0, 1):
                native.check(lib.upamd_tune(b'gemm_stagger_mode', mode))
                ms = time_ms(fn, 10)
                res.append('%.3f ms %6.1f TF' % (ms, 2.0 * M * K * N / ms / 1e9))
            if ref is None:
                ref = Cc.clone()
            err = float((Cc - ref).abs().max())
            print('%-22s v%d %-32s plain %s | staggered %s | max|d| %.1e' % (name, v, NAMES[v], res[0], res[1], err), flush=True)
        # ragged tail + repeated-run race screen on the asm variant
        Mt = 128 * 37 + 5
        At = A[:, :Mt].contiguous()
        Rt = R[:, :Mt].contiguous() if resid else None
        Ct0, Ct1 = torch.zeros(N // 16, Mt, 16, device=dev), torch.zeros(N // 16, Mt, 16, device=dev)
        native.check(lib.upamd_tune(b'gemm_nt_dma', 0))
        native.check(lib.upamd_gemm_nt(P(At), Mt, K, 0, 0, P(W), N, K, None, P(Rt), P(Ct0), 0, 0, 0, 1.0, st))
        for v in (2, 4):
            native.check(lib.upamd_tune(b'gemm_nt_dma', v))
            worst = 0.0
            for rep in range(20):
                Ct1.zero_()
                native.check(lib.upamd_gemm_nt(P(At), Mt, K, 0, 0, P(W), N, K, None, P(Rt), P(Ct1), 0, 0, 0, 1.0, st))
                worst = max(worst, float((Ct1 - Ct0).abs().max()))
            print('   tail M=%d v%d worst max|d| over 20 runs %.2e' % (Mt, v, worst))
    native.check(lib.upamd_tune(b'gemm_stagger_mode', 0))
    native.check(lib.upamd_tune(b'gemm_nt_dma', 0))


if __name__ == '__main__':
    main()
