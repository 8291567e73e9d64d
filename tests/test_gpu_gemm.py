"""Kernel-level parity of the fp32-MFMA GEMM building blocks against torch fp32 matmul (needs a GPU).
MFMA fp32 is an exact fma chain, so the only difference is summation order: rel tolerance 2e-5."""
import ctypes as C

import numpy as np
import pytest
import torch

from drl_urban_planning_amd import native

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def to_pm(x):       # [M, C] -> panel-major [C/16][M][16]
    M, Cc = x.shape
    return x.view(M, Cc // 16, 16).permute(1, 0, 2).contiguous()


def from_pm(x, M, Cc):
    return x.view(Cc // 16, M, 16).permute(1, 0, 2).reshape(M, Cc)


@pytest.mark.parametrize('M,K,N', [(1000, 256, 512), (777, 512, 256), (130, 32, 256), (4099, 1024, 32), (300, 16, 48),
                                     (515, 64, 64)])
@pytest.mark.parametrize('a_rm,c_rm', [(0, 0), (1, 1), (0, 1)])
def test_gemm_nt(M, K, N, a_rm, c_rm):
    if (a_rm or c_rm) and N % 32 != 0:
        pytest.skip('row-major operands need the MFMA path')
    lib = native.lib()
    g = torch.Generator(device='cpu').manual_seed(M + K + N)
    A = torch.randn(M, K, generator=g).to(DEV)
    # asymmetric, non-square weights catch operand / output transposes
    W = (torch.randn(N, K, generator=g) * 0.1).to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    R = torch.randn(M, N, generator=g).to(DEV)
    ref = 0.5 * torch.tanh(A.double() @ W.double().t() + bias.double() + R.double())
    A_in = A.contiguous() if a_rm else to_pm(A)
    R_in = R.contiguous() if c_rm else to_pm(R)
    Cout = torch.zeros(M * N, device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    native.check(lib.upamd_gemm_nt(P(A_in), M, K, K, a_rm, P(W), N, K, P(bias), P(R_in), P(Cout), N, c_rm, 1, 0.5, st))
    torch.cuda.synchronize()
    got = Cout.view(M, N) if c_rm else from_pm(Cout, M, N)
    np.testing.assert_allclose(got.cpu().numpy(), ref.float().cpu().numpy(), rtol=2e-5, atol=1e-5)


@pytest.mark.parametrize('M,I,J', [(5000, 512, 256), (999, 256, 32), (70000, 128, 128), (333, 32, 16), (2048, 256, 256)])
@pytest.mark.parametrize('rm', [0, 1])
def test_gemm_tn(M, I, J, rm):
    if rm and not (I % 128 == 0 and J % 32 == 0):
        pytest.skip('row-major operands need the MFMA path')
    lib = native.lib()
    g = torch.Generator(device='cpu').manual_seed(M + I + J)
    A = torch.randn(M, I, generator=g).to(DEV)
    B = torch.randn(M, J, generator=g).to(DEV)
    ref = (A.double().t() @ B.double()).float()
    A_in = A.contiguous() if rm else to_pm(A)
    B_in = B.contiguous() if rm else to_pm(B)
    scratch = torch.empty(int(lib.upamd_gemm_tn_scratch_floats(I, J, M)), device=DEV)
    out = torch.empty(I, J, device=DEV)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    native.check(lib.upamd_gemm_tn(P(A_in), I, I, P(B_in), J, J, M, rm, P(scratch), P(out), st))
    torch.cuda.synchronize()
    scale = float(ref.abs().max())
    np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=2e-5 * scale)


@pytest.mark.parametrize('M,K,N', [(4099, 256, 512), (777, 512, 256), (130, 16, 128), (1500, 48, 128)])
@pytest.mark.parametrize('nprod', [6, 9])
def test_gemm_nt_split_has_fp32_accuracy(M, K, N, nprod):
    """The opt-in split-bf16 product (csrc/gemm_split.hip: fp32 operands as three bf16 terms on the bf16 matrix pipe,
    fp32 accumulation) against a float64 product, next to the exact-fp32 MFMA kernel on the same inputs: its error must
    not exceed the fp32 kernel's on benign data (the dropped terms are below one fp32 rounding) and stays within a small
    factor of it on wide-dynamic-range data, with the fused bias / residual / tanh epilogue."""
    lib = native.lib()
    g = torch.Generator(device='cpu').manual_seed(M + K + N + nprod)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    scratch = torch.zeros(int(lib.upamd_gemm_nt_split_scratch_bytes(N, K)), dtype=torch.uint8, device=DEV)
    for wide in (False, True):
        A = torch.randn(M, K, generator=g)
        W = torch.randn(N, K, generator=g) * 0.1
        if wide:
            A = A * torch.pow(10.0, torch.rand(M, K, generator=g) * 6 - 3)
            W = W * torch.pow(10.0, torch.rand(N, K, generator=g) * 4 - 2)
        A, W = A.to(DEV), W.to(DEV)
        bias = torch.randn(N, generator=g).to(DEV)
        R = torch.randn(M, N, generator=g).to(DEV)
        for act in (0, 1):
            pre = A.double() @ W.double().t() + bias.double() + R.double()
            ref = 0.5 * (torch.tanh(pre) if act else pre)
            A_in, R_in = to_pm(A), to_pm(R)
            c_exact, c_split = torch.zeros(M * N, device=DEV), torch.zeros(M * N, device=DEV)
            native.check(lib.upamd_gemm_nt(P(A_in), M, K, K, 0, P(W), N, K, P(bias), P(R_in), P(c_exact), N, 0, act, 0.5, st))
            native.check(lib.upamd_gemm_nt_split(P(A_in), M, K, P(W), N, K, P(bias), P(R_in), P(c_split), act, 0.5, nprod,
                                                 P(scratch), st))
            torch.cuda.synchronize()
            e_exact = (from_pm(c_exact, M, N).double() - ref).abs()
            e_split = (from_pm(c_split, M, N).double() - ref).abs()
            scale = float(ref.abs().max())
            rms_exact, rms_split = float(e_exact.pow(2).mean().sqrt()), float(e_split.pow(2).mean().sqrt())
            # benign data: not worse than the exact-fp32 kernel.  Six decades of dynamic range and a short K (little
            # accumulation noise to hide behind): the dropped 2^-26 terms show, the error stays within 2.5x / 3x
            slack = 2.5 if wide else 1.1
            assert rms_split <= slack * rms_exact + 1e-9 * scale, (wide, act, rms_split, rms_exact)
            if not wide:      # (the maximum over heavy cancellations is dominated by single outliers of either kernel)
                assert float(e_split.max()) <= 2.0 * float(e_exact.max()) + 1e-7 * scale, (act, float(e_split.max()), float(e_exact.max()))
            if not wide:      # (with six decades of dynamic range the cancellation noise of ANY fp32 product exceeds this)
                np.testing.assert_allclose(from_pm(c_split, M, N).cpu().numpy(), ref.float().cpu().numpy(), rtol=2e-5,
                                           atol=1e-5 * max(1.0, scale))


def test_gemm_nt_split_rejects_unsupported_shapes():
    lib = native.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    A, W, Cc = torch.zeros(1, 64, 16, device=DEV), torch.zeros(96, 16, device=DEV), torch.zeros(6, 64, 16, device=DEV)
    scratch = torch.zeros(1 << 16, dtype=torch.uint8, device=DEV)
    assert lib.upamd_gemm_nt_split(P(A), 64, 16, P(W), 96, 16, None, None, P(Cc), 0, 1.0, 6, P(scratch), st) != 0     # N % 128
    W2 = torch.zeros(128, 16, device=DEV)
    assert lib.upamd_gemm_nt_split(P(A), 64, 16, P(W2), 128, 16, None, None, P(Cc), 0, 1.0, 7, P(scratch), st) != 0    # products
    assert lib.upamd_gemm_nt_split(P(A), 64, 16, P(W2), 128, 16, None, None, P(Cc), 0, 1.0, 6, None, st) != 0          # scratch
