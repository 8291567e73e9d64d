// Micro-benchmark: issue rate of v_exp_f32 / v_rcp_f32 / v_fma_f32 on gfx950 (wave64), used to decide what
// bounds the message-passing kernels.  hipcc --offload-arch=gfx950 -O3 tools/valu_rate_bench.hip -o /tmp/valu_rate_bench
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    float a = threadIdx.x * 1e-3f, b = a + 0.5f, c = a + 0.25f, d = a + 0.75f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) { a = __builtin_amdgcn_exp2f(a); b = __builtin_amdgcn_exp2f(b); c = __builtin_amdgcn_exp2f(c); d = __builtin_amdgcn_exp2f(d); }
            if (MODE == 1) { a = __builtin_amdgcn_rcpf(a); b = __builtin_amdgcn_rcpf(b); c = __builtin_amdgcn_rcpf(c); d = __builtin_amdgcn_rcpf(d); }
            if (MODE == 2) { a = fmaf(a, 1.0001f, 0.5f); b = fmaf(b, 1.0001f, 0.5f); c = fmaf(c, 1.0001f, 0.5f); d = fmaf(d, 1.0001f, 0.5f); }
            if (MODE == 3) { a = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(a) + 1.f); b = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(b) + 1.f);
                             c = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(c) + 1.f); d = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(d) + 1.f); }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;
}

template <int MODE>
void run(const char *name, int ops_per_inner) {
    float *out;
    hipMalloc(&out, 256 * 2048 * 4);
    const int iters = 2000, blocks = 2048;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 10);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double lane_ops = (double)blocks * 256 * iters * 8 * ops_per_inner;
    // per SIMD: lane_ops / 1024 SIMDs / time -> lanes per ns; at ~2.1-2.4 GHz
    printf("%-10s %.3f ms  %.2f T lane-ops/s  = %.2f lanes/clk/SIMD @2.4GHz\n", name, ms, lane_ops / ms / 1e9,
           lane_ops / (ms * 1e-3) / 1024 / 2.4e9);
    hipFree(out);
}

int main() {
    run<2>("fma", 4);
    run<0>("exp2", 4);
    run<1>("rcp", 4);
    run<3>("exp+add+rcp", 4);
    return 0;
}
