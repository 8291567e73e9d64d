// What costs the fp32 MFMA pipe its last 15 %?  v_mfma_f32_32x32x2_f32 streams with GEMM ingredients added one at a time.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak.bin && tools/mfma_peak.bin
// LEVEL 0: MFMAs only (constant operands)          1: + operands re-read from LDS every 16 MFMAs (4 ds_read_b128 x 2)
//       2: + s_barrier every 32 MFMAs              3: + 4 LDS-DMA loads (1 KiB each, L2-resident source) every 32 MFMAs
// Every workgroup asks for 36 KB of LDS, so exactly `wgs` (<= 4) of them fit on a CU and a grid of 256 * wgs is balanced.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

template <int LEVEL>
__global__ __launch_bounds__(256, 4) void mfma_loop(float *out, const float *src, int iters) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 8192; i += 256) smem[i] = 1e-3f * (i & 7);
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float4 a[2][2], b[2][2];
    for (int g = 0; g < 2; ++g)
        for (int i = 0; i < 2; ++i) {
            a[g][i] = make_float4(1.f, 2.f, 3.f, 4.f);
            b[g][i] = make_float4(1e-3f, 2e-3f, 3e-3f, 4e-3f);
        }
    const int sw = ((lane >> 5) ^ ((lane >> 2) & 3)) * 4;
    const float *gsrc = src + (size_t)(blockIdx.x & 63) * 4096 + (size_t)tid * 4;
    for (int it = 0; it < iters; ++it) {
        if (LEVEL >= 2) {
            if (LEVEL >= 3 && LEVEL != 4 && LEVEL != 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        const float *st = smem + (it & 1) * 4096;
        if (LEVEL >= 1) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    a[g][i] = *reinterpret_cast<const float4 *>(st + ((w >> 1) * 64 + i * 32 + (lane & 31)) * 16 + (sw ^ (8 * g)));
                    b[g][i] = *reinterpret_cast<const float4 *>(st + 2048 + ((w & 1) * 64 + i * 32 + (lane & 31)) * 16 + (sw ^ (8 * g)));
                }
        }
        if (LEVEL == 3 || LEVEL == 6) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_global_load_lds((gptr_t)(gsrc + j * 1024), (lptr_t)(smem + ((it + 1) & 1) * 4096 + (w * 4 + j) * 256), 16, 0, 0);
        }
        if (LEVEL == 5) {          // half as many DMA instructions
#pragma unroll
            for (int j = 0; j < 2; ++j)
                __builtin_amdgcn_global_load_lds((gptr_t)(gsrc + j * 1024), (lptr_t)(smem + ((it + 1) & 1) * 4096 + (w * 4 + j) * 256), 16, 0, 0);
        }
        if (LEVEL == 4) {          // plain register loads instead (kept alive through an empty asm)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float4 v = *reinterpret_cast<const float4 *>(gsrc + j * 1024);
                asm volatile("" ::"v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
            }
        }
        if (LEVEL == 7) {          // one scalar-ish tiny load per 32 MFMAs (a dword per lane)
            float v = gsrc[0];
            asm volatile("" ::"v"(v));
        }
        if (LEVEL == 6) __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float av[2] = {t == 0 ? a[g][0].x : t == 1 ? a[g][0].y : t == 2 ? a[g][0].z : a[g][0].w,
                                     t == 0 ? a[g][1].x : t == 1 ? a[g][1].y : t == 2 ? a[g][1].z : a[g][1].w};
                const float bv[2] = {t == 0 ? b[g][0].x : t == 1 ? b[g][0].y : t == 2 ? b[g][0].z : b[g][0].w,
                                     t == 0 ? b[g][1].x : t == 1 ? b[g][1].y : t == 2 ? b[g][1].z : b[g][1].w};
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[j], av[i], acc[i * 2 + j], 0, 0, 0);
            }
        if (LEVEL == 6) __builtin_amdgcn_s_setprio(0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int LEVEL>
void run(float *out, const float *src, const char *what) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 1500;
    hipFuncSetAttribute(reinterpret_cast<const void *>(mfma_loop<LEVEL>), hipFuncAttributeMaxDynamicSharedMemorySize, 36 * 1024);
    for (int wgs = 1; wgs <= 4; ++wgs) {
        const int grid = 256 * wgs;
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_loop<LEVEL>, dim3(grid), dim3(256), 36 * 1024, 0, out, src, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        const double flop = (double)grid * 4 * iters * 32 * 4096.0;
        printf("level %d (%s), %d waves/SIMD: %.3f ms  %.1f TFLOP/s\n", LEVEL, what, wgs, best, flop / best / 1e9);
    }
}

int main() {
    float *out, *src;
    hipMalloc(&out, sizeof(float) * 256 * 1024);
    hipMalloc(&src, sizeof(float) * 64 * 4096 + 65536);
    hipMemset(src, 0, sizeof(float) * 64 * 4096 + 65536);
    // warm the clocks
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(mfma_loop<0>, dim3(768), dim3(256), 36 * 1024, 0, out, src, 500);
    hipDeviceSynchronize();
    run<0>(out, src, "MFMA only");
    run<1>(out, src, "+ LDS fragment reads");
    run<2>(out, src, "+ barrier per 32 MFMAs");
    run<3>(out, src, "+ 4 LDS-DMA per 32 MFMAs");
    run<5>(out, src, "+ 2 LDS-DMA per 32 MFMAs");
    run<4>(out, src, "+ 4 global_load_dwordx4 per 32 MFMAs");
    run<7>(out, src, "+ 1 global_load_dword per 32 MFMAs");
    run<6>(out, src, "+ 4 LDS-DMA, MFMAs at s_setprio 2");
    return 0;
}
