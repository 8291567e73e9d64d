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
