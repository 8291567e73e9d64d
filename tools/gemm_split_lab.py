"""Kernel lab: fp32 GEMM on the bf16 matrix pipe by exact 3-way operand splitting (csrc/gemm_split.hip, opt-in) against
the fp32-MFMA LDS-DMA kernel of the training step, on the step's shapes: accuracy of both against a float64 product,
and throughput (needs a GPU)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drl_urban_planning_amd import native  # noqa: E402
from kernel_bench import P, time_ms  # noqa: E402


def errs(Cc, ref64, rows):
    """max and rms error of panel-major C rows against the float64 reference, relative to the rms of the reference"""
    got = Cc[:, rows].permute(1, 0, 2).reshape(len(rows), -1).double()
    d = got - ref64
    scale = float(ref64.pow(2).mean().sqrt())
    return float(d.abs().max()) / scale, float(d.pow(2).mean().sqrt()) / scale


def main():
    lib = native.lib()
    dev = 'cuda:0'
    M = 565000
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.manual_seed(0)
    for name, K, N, resid, wide in (('fwd P/Q K=256 N=512', 256, 512, False, False), ('dgrad K=512 N=256 +R', 512, 256, True, False),
                                    ('fwd K=256 N=512, 6 decades of dynamic range', 256, 512, False, True)):
        A = torch.randn(K // 16, M, 16, device=dev)
        W = torch.randn(N, K, device=dev) * 0.05
        if wide:
            A = A * torch.pow(10.0, torch.rand_like(A) * 6 - 3)
            W = W * torch.pow(10.0, torch.rand_like(W) * 6 - 3)
        R = torch.randn(N // 16, M, 16, device=dev) if resid else None
        bias = torch.randn(N, device=dev)
        Cc = torch.zeros(N // 16, M, 16, device=dev)
        scratch = torch.zeros(int(lib.upamd_gemm_nt_split_scratch_bytes(N, K)), dtype=torch.uint8, device=dev)
        rows = torch.cat([torch.arange(0, 3000), torch.arange(M - 1500, M)]).to(dev)
        A64 = A[:, rows].permute(1, 0, 2).reshape(len(rows), K).double()
        ref = A64 @ W.double().t() + bias.double()
        if resid:
            ref = ref + R[:, rows].permute(1, 0, 2).reshape(len(rows), N).double()
        kern = {
            'fp32 MFMA (LDS-DMA, default)': lambda: native.check(lib.upamd_gemm_nt(P(A), M, K, 0, 0, P(W), N, K, P(bias), P(R), P(Cc), 0, 0, 0, 1.0, st)),
            'bf16 split, 6 products': lambda: native.check(lib.upamd_gemm_nt_split(P(A), M, K, P(W), N, K, P(bias), P(R), P(Cc), 0, 1.0, 6, P(scratch), st)),
            'bf16 split, 9 products': lambda: native.check(lib.upamd_gemm_nt_split(P(A), M, K, P(W), N, K, P(bias), P(R), P(Cc), 0, 1.0, 9, P(scratch), st)),
        }
        for kn, fn in kern.items():
            Cc.zero_()
            for _ in range(30):
                fn()
            ms = time_ms(fn, 20)
            emax, erms = errs(Cc, ref, rows)
            print('%-46s %-30s %7.1f TFLOP/s (fp32-equivalent)  %.3f ms   err/rms(ref): max %.2e rms %.2e' %
                  (name, kn, 2.0 * M * K * N / ms / 1e9, ms, emax, erms), flush=True)
        # ragged tail
        Mt = 256 * 19 + 37
        At = A[:, :Mt].contiguous()
        Rt = R[:, :Mt].contiguous() if resid else None
        Ct = torch.zeros(N // 16, Mt, 16, device=dev)
        native.check(lib.upamd_gemm_nt_split(P(At), Mt, K, P(W), N, K, P(bias), P(Rt), P(Ct), 0, 1.0, 6, P(scratch), st))
        r2 = torch.arange(Mt, device=dev)
        ref2 = At.permute(1, 0, 2).reshape(Mt, K).double() @ W.double().t() + bias.double()
        if resid:
            ref2 = ref2 + Rt.permute(1, 0, 2).reshape(Mt, N).double()
        print('   tail M=%d split6 err max %.2e rms %.2e' % ((Mt,) + errs(Ct, ref2, r2)), flush=True)


if __name__ == '__main__':
    main()
