"""Kernel lab 6: occupancy (LDS pad) x priority for the pipelined LDS-DMA gemm_nt (needs a GPU)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drl_urban_planning_amd import native  # noqa: E402
from kernel_bench import P, time_ms  # noqa: E402

NAMES = {0: 'register-staged', 1: 'dma simple', 2: 'pipelined', 3: 'pipelined+prio2', 4: 'asm reads', 5: 'asm reads+prio2',
         6: 'asm reads+prio2 minw2'}


def main():
    lib = native.lib()
    dev = 'cuda:0'
    M = 565000
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(0)
    native.check(lib.upamd_tune(b'gemm_stagger_cycles', 37000))
    native.check(lib.upamd_tune(b'gemm_stagger_mode', 1))
    for name, K, N, resid, zero in (('fwd P/Q K=256 N=512', 256, 512, False, False), ('dgrad K=512 N=256 +R', 512, 256, True, False),
                                    ('K=1024 N=512 zeros', 1024, 512, False, True)):
        A = torch.zeros(K // 16, M, 16, device=dev) if zero else torch.randn(K // 16, M, 16, device=dev)
        W = torch.randn(N, K, device=dev) * 0.05
        R = torch.randn(N // 16, M, 16, device=dev) if resid else None
        Cc = torch.zeros(N // 16, M, 16, device=dev)
        fn = lambda: native.check(lib.upamd_gemm_nt(P(A), M, K, 0, 0, P(W), N, K, None, P(R), P(Cc), 0, 0, 0, 1.0, st))
        for _ in range(30):
            fn()                      # clocks up
        for v in range(7):
            native.check(lib.upamd_tune(b'gemm_nt_dma', v))
            res = []
            for pad, wgs in ((0, 4), (12 * 1024, 3), (40 * 1024, 2), (70 * 1024, 1)):
                native.check(lib.upamd_tune(b'gemm_lds_pad', pad))
                ms = time_ms(fn, 10)
                res.append('%d WG/CU %6.1f TF' % (wgs, 2.0 * M * K * N / ms / 1e9))
                if v == 0:
                    break
            print('%-22s v%d %-24s %s' % (name, v, NAMES[v], ' | '.join(res)), flush=True)
    native.check(lib.upamd_tune(b'gemm_lds_pad', 0))
    native.check(lib.upamd_tune(b'gemm_stagger_mode', 0))
    native.check(lib.upamd_tune(b'gemm_nt_dma', 0))


if __name__ == '__main__':
    main()
