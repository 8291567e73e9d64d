"""Kernel lab: the register-staged gemm_nt against the LDS-DMA kernel at every workgroup tile, on the shapes of the
training step (needs a GPU).  Earlier lab rounds (staging variants, stagger, K sweep, occupancy x priority) are
summarised with their raw logs in profiles/archive/r02_gemm_lab.md."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drl_urban_planning_amd import native  # noqa: E402
from kernel_bench import P, time_ms  # noqa: E402

NAMES = {0: 'register-staged 128x128', 1: 'LDS-DMA 128x128 (default)', 2: 'LDS-DMA 256x128', 3: 'LDS-DMA 128x256', 4: 'LDS-DMA 256x256',
         5: '4 waves x (64x128)', 6: '4 waves x (64x128) + setprio', 7: '4 waves x (128x64)', 8: '4 waves x (128x64) + setprio',
         9: 'LDS-DMA 128x128 + setprio'}
VARIANTS = tuple(int(v) for v in os.environ.get('GEMM_LAB_VARIANTS', '0,1,2,3,4').split(','))


def main():
    lib = native.lib()
    dev = 'cuda:0'
    M = 565000
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(0)
    for name, K, N, resid, zero in (('fwd P/Q K=256 N=512', 256, 512, False, False), ('dgrad K=512 N=256 +R', 512, 256, True, False),
                                    ('K=1024 N=512 zeros', 1024, 512, False, True)):
        A = torch.zeros(K // 16, M, 16, device=dev) if zero else torch.randn(K // 16, M, 16, device=dev)
        W = torch.randn(N, K, device=dev) * 0.05
        R = torch.randn(N // 16, M, 16, device=dev) if resid else None
        Cc = torch.zeros(N // 16, M, 16, device=dev)
        fn = lambda: native.check(lib.upamd_gemm_nt(P(A), M, K, 0, 0, P(W), N, K, None, P(R), P(Cc), 0, 0, 0, 1.0, st))
        native.check(lib.upamd_tune(b'gemm_nt_dma', 0))
        for _ in range(30):
            fn()
        ref = Cc.clone()
        for v in VARIANTS:
            native.check(lib.upamd_tune(b'gemm_nt_dma', v))
            res = []
            for pad in (0, 12 * 1024):
                native.check(lib.upamd_tune(b'gemm_lds_pad', pad))
                Cc.zero_()
                ms = time_ms(fn, 10)
                res.append('pad %2dK %6.1f TF' % (pad // 1024, 2.0 * M * K * N / ms / 1e9))
            print('%-22s v%-2d %-30s %s  max|d| %.1e' % (name, v, NAMES[v], ' | '.join(res), float((Cc - ref).abs().max())), flush=True)
        Mt = 256 * 19 + 37
        At = A[:, :Mt].contiguous()
        Rt = R[:, :Mt].contiguous() if resid else None
        Ct0, Ct1 = torch.zeros(N // 16, Mt, 16, device=dev), torch.zeros(N // 16, Mt, 16, device=dev)
        native.check(lib.upamd_tune(b'gemm_nt_dma', 0))
        native.check(lib.upamd_gemm_nt(P(At), Mt, K, 0, 0, P(W), N, K, None, P(Rt), P(Ct0), 0, 0, 0, 1.0, st))
        for v in [x for x in VARIANTS if x]:
            native.check(lib.upamd_tune(b'gemm_nt_dma', v))
            Ct1.zero_()
            native.check(lib.upamd_gemm_nt(P(At), Mt, K, 0, 0, P(W), N, K, None, P(Rt), P(Ct1), 0, 0, 0, 1.0, st))
            print('   tail M=%d v%d max|d| %.2e' % (Mt, v, float((Ct1 - Ct0).abs().max())))
    native.check(lib.upamd_tune(b'gemm_lds_pad', 12 * 1024))
    native.check(lib.upamd_tune(b'gemm_nt_dma', 1))


if __name__ == '__main__':
    main()
