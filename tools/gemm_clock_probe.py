"""Kernel lab 4: effective shader clock while the GEMM runs (side-stream clock probe) vs idle (needs a GPU)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drl_urban_planning_amd import native  # noqa: E402
from kernel_bench import P  # noqa: E402


def probe(lib, side, fn, reps, label):
    S = 400
    buf = torch.zeros(2 * S, dtype=torch.int64, device='cuda:0')
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        native.check(lib.upamd_clock_probe(P(buf), S, 2000, C.c_void_p(side.cuda_stream)))       # 20 us apart, 8 ms total
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    a = buf.cpu().numpy().reshape(S, 2)
    d = np.diff(a, axis=0)
    mhz = d[:, 0] / d[:, 1] * 100.0
    print('%-34s shader clock/wall tick ratio -> %.0f MHz median (p10 %.0f, p90 %.0f)' % (label, np.median(mhz),
          np.percentile(mhz, 10), np.percentile(mhz, 90)), flush=True)


def main():
    lib = native.lib()
    dev = 'cuda:0'
    M, K, N = 565000, 256, 512
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    side = torch.cuda.Stream()
    A = torch.randn(K // 16, M, 16, device=dev)
    A0 = torch.zeros(K // 16, M, 16, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    Cc = torch.zeros(N // 16, M, 16, device=dev)
    probe(lib, side, lambda: None, 0, 'idle')
    for v in (0, 1):
        native.check(lib.upamd_tune(b'gemm_nt_dma', v))
        for data, Ain in (('random', A), ('zeros', A0)):
            fn = lambda: native.check(lib.upamd_gemm_nt(P(Ain), M, K, 0, 0, P(W), N, K, None, None, P(Cc), 0, 0, 0, 1.0, st))
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(8):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 8
            probe(lib, side, fn, 8, 'gemm v%d %s data (%.1f TF)' % (v, data, 2.0 * M * K * N / ms / 1e9))
    native.check(lib.upamd_tune(b'gemm_nt_dma', 0))


if __name__ == '__main__':
    main()
