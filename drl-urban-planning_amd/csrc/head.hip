// Land-use pointer head, first Linear, on the candidate messages m ALONE (gfx950, wave64).
//
// The head's first Linear acts on [m ; c ; m*c ; m-c] (urban_planning/models/state_encoder.py:207-210, policy.py:19-43);
// factorised it is  hid = tanh(Wa' m + Wc (m*c) + constb)  with Wa' = Wa + Wd, and c = the encoded
// current node of the candidate's GRAPH.  The m*c half is therefore not a second input but a per-graph change of the
// weight:   Wa' m + Wc (m*c) = (Wa' + Wc diag(c_b)) m = W_b m.   With W_b built once per graph (32 x D: one LDS tile) the
// candidate tensor FE keeps only its m half: the last-layer forward writes half as much (0.42 GB less at B = 2048, D = 256),
// and both consumers -- this product and its weight gradient -- read half as much.  The weight gradient splits the same way:
//   dWa'[k][d] = sum_rows dpre[row][k] m[row][d],      dWc[k][d] = sum_b c_b[d] (sum_{rows of b} dpre[row][k] m[row][d]).
// Both kernels use v_mfma_f32_32x32x2_f32 with the hidden units (forward) / the output columns (gradient) as the lane
// dimension of the result; shapes: h0 = 32 hidden units, D a multiple of 32 (<= 256 for the gradient kernel).
#include "kernels.h"

namespace upamd {

#define META(t) (pk.meta + (int64_t)(t) * UPAMD_META_STRIDE)

namespace {

typedef float f32x16h __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float tanh_h(float x) {      // the GEMM epilogue's form (abs error ~1e-7)
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// hid(pm)[row][u] = tanh(sum_d W_b[u][d] m[row][d] + constb[b][u]).  One 256-thread workgroup per graph: W_b in LDS
// (row stride D + 4 floats: the 32 rows a 16-lane ds_read_b128 group touches fall on distinct banks), one 32-candidate
// tile per wave and trip.  MFMA: A = W_b (register dimension = hidden unit), B = m rows (lane dimension = candidate), so lane
// (r, kh) ends up with hidden units 8g + 4kh + t (g, t < 4) of candidate r: whole 16-byte groups of the panel-major hid rows.
// The K order is free: lane half kh covers columns 8 kh .. 8 kh + 7 of every 16-column panel.
template <int NP>
__global__ __launch_bounds__(256) void head_hidden_fwd_kernel(PackedView pk, MbView mb, const float *__restrict__ FE,
                                                              const float *__restrict__ C, const float *__restrict__ W1f,
                                                              const float *__restrict__ constb, float *__restrict__ hid) {
    constexpr int D = NP * 16, LDW = D + 4;
    extern __shared__ __attribute__((aligned(16))) float wb[];      // [32][LDW]
    const int b = blockIdx.x, t = mb.idx[b];
    const int nh = META(t)[2];
    if (nh == 0) return;
    const int64_t q0 = mb.he_off[b], NH = mb.Nhe;
    // W_b: 32 D elements over 256 threads, eight elements' loads in flight per thread (one at a time made this prologue the
    // longest part of the kernel: ~30 dependent L2 round trips per workgroup)
    constexpr int PER = 32 * D / 256;
#pragma unroll
    for (int i0 = 0; i0 < PER; i0 += 8) {
        float wa[8], wc[8], cv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (i0 + u < PER) {
                const int i = (i0 + u) * 256 + threadIdx.x, k = i / D, d = i % D;
                wa[u] = W1f[(int64_t)k * 2 * D + d];
                wc[u] = W1f[(int64_t)k * 2 * D + D + d];
                cv[u] = C[(int64_t)b * D + d];
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (i0 + u < PER) {
                const int i = (i0 + u) * 256 + threadIdx.x, k = i / D, d = i % D;
                wb[k * LDW + d] = fmaf(wc[u], cv[u], wa[u]);
            }
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 31, kh = lane >> 5;
    float4 cb[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) cb[g] = *reinterpret_cast<const float4 *>(constb + (int64_t)b * 32 + 8 * g + 4 * kh);
    const float *wrow = wb + r * LDW + 8 * kh;
    for (int q = 32 * w; q < nh; q += 128) {
        const int row = q + r;
        const bool in = row < nh;
        const int64_t grow = q0 + (in ? row : nh - 1);
        f32x16h acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int pc = 0; pc < NP; pc += 4) {                    // four panels (= 8 loads per lane) in flight at a time
            float4 x[4][2];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (pc + p < NP) {
                    const float4 *src = reinterpret_cast<const float4 *>(FE + ((int64_t)(pc + p) * NH + grow) * 16 + 8 * kh);
                    x[p][0] = src[0];
                    x[p][1] = src[1];
                }
            }
            // keep the eight loads together: left alone, the scheduler sinks each one next to its four MFMAs (shorter live
            // ranges next to the 128 hoisted weight registers) and a tile becomes 32 serial memory round trips
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (pc + p < NP) {
                    const float4 a0 = *reinterpret_cast<const float4 *>(wrow + (pc + p) * 16);
                    const float4 a1 = *reinterpret_cast<const float4 *>(wrow + (pc + p) * 16 + 4);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, x[p][0].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, x[p][0].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, x[p][0].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, x[p][0].w, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, x[p][1].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, x[p][1].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, x[p][1].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, x[p][1].w, acc, 0, 0, 0);
                }
            }
        }
        if (in) {
#pragma unroll
            for (int g = 0; g < 4; ++g)      // accumulators 4g .. 4g+3 = hidden units 8g + 4kh + (0..3): panel g >> 1, columns 8 (g & 1) + 4 kh ..
                *reinterpret_cast<float4 *>(hid + ((int64_t)(g >> 1) * NH + grow) * 16 + 8 * (g & 1) + 4 * kh) =
                    make_float4(tanh_h(acc[4 * g + 0] + cb[g].x), tanh_h(acc[4 * g + 1] + cb[g].y), tanh_h(acc[4 * g + 2] + cb[g].z),
                                tanh_h(acc[4 * g + 3] + cb[g].w));
        }
    }
}

// Weight gradient of the same Linear: slab[wg][d][k] (a part, rows 0 .. D-1) and slab[wg][D + d][k] (c part) = this workgroup's
// share of  sum_rows dpre[row][k] m[row][d]  and of its per-graph c_b[d]-weighted version.  Workgroup wg takes the graphs
// wg, wg + G, ... in that order (fixed assignment: bit-reproducible); wave w owns the 32-column tiles w and w + 4 of d.
// MFMA over the graph's candidates (two per step): A = dpre^T (register dimension = hidden unit k), B = m (lane dimension = the
// tile's column), both read as 4-byte elements of 64-byte row segments, eight steps (16 candidates) of loads in flight.
template <int TPW>      // column tiles per wave: D / 32 / 4 rounded up (1 or 2)
__global__ __launch_bounds__(256) void head_wgrad_kernel(PackedView pk, MbView mb, int NP, int G, const float *__restrict__ FE,
                                                         const float *__restrict__ C, const float *__restrict__ dprel,
                                                         float *__restrict__ slab) {
    const int D = NP * 16, tiles = D / 32;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 31, kh = lane >> 5;
    const int64_t NH = mb.Nhe;
    f32x16h ta[TPW], tc[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) { ta[j][i] = 0.f; tc[j][i] = 0.f; }
    const int64_t aoff = ((int64_t)(r >> 4) * NH) * 16 + (r & 15);      // dpre element (candidate 0, hidden unit r)
    for (int b = blockIdx.x; b < mb.B; b += G) {
        const int t = mb.idx[b];
        const int nh = META(t)[2];
        if (nh == 0) continue;
        const int64_t q0 = mb.he_off[b];
        f32x16h tg[TPW];
#pragma unroll
        for (int j = 0; j < TPW; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) tg[j][i] = 0.f;
        constexpr int ST = 8;                                   // MFMA steps (= 2 ST candidates) per trip
        // operands of one trip, requested a whole trip ahead and unconditionally (clamped rows; the last trip re-fetches
        // itself): with a branch around the prefetch the compiler waits for every outstanding load at the join
        auto fetch = [&](int s0, float (&av)[ST], float (&bv)[TPW][ST]) {
#pragma unroll
            for (int u = 0; u < ST; ++u) {
                const int64_t row = q0 + min(s0 + 2 * u + kh, nh - 1);
                av[u] = dprel[aoff + row * 16];
#pragma unroll
                for (int j = 0; j < TPW; ++j) {
                    const int tile = min(w + 4 * j, tiles - 1);      // (a wave without a j-th tile redoes its last one; never written out)
                    bv[j][u] = FE[((int64_t)(2 * tile + (r >> 4)) * NH + row) * 16 + (r & 15)];
                }
            }
        };
        float av[ST], bv[TPW][ST];
        fetch(0, av, bv);
        for (int s0 = 0; s0 < nh; s0 += 2 * ST) {
            float an[ST], bn[TPW][ST];
            fetch(s0 + 2 * ST < nh ? s0 + 2 * ST : s0, an, bn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < ST; ++u) {
                const float a = (s0 + 2 * u + kh < nh) ? av[u] : 0.f;      // candidates past the end contribute nothing
#pragma unroll
                for (int j = 0; j < TPW; ++j) tg[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[j][u], tg[j], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < ST; ++u) {
                av[u] = an[u];
#pragma unroll
                for (int j = 0; j < TPW; ++j) bv[j][u] = bn[j][u];
            }
        }
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            const int tile = w + 4 * j;
            const float cc = tile < tiles ? C[(int64_t)b * D + 32 * tile + r] : 0.f;      // the lane's column d = 32 tile + r
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                ta[j][i] += tg[j][i];
                tc[j][i] = fmaf(tg[j][i], cc, tc[j][i]);
            }
        }
    }
    // accumulator 4g + t = hidden unit k = 8g + 4kh + t, lane r = column d = 32 tile + r
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int tile = w + 4 * j;
        if (tile >= tiles) continue;
        float *sa = slab + ((int64_t)blockIdx.x * 2 * D + 32 * tile + r) * 32;
        float *sc = sa + (int64_t)D * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            *reinterpret_cast<float4 *>(sa + 8 * g + 4 * kh) = make_float4(ta[j][4 * g + 0], ta[j][4 * g + 1], ta[j][4 * g + 2], ta[j][4 * g + 3]);
            *reinterpret_cast<float4 *>(sc + 8 * g + 4 * kh) = make_float4(tc[j][4 * g + 0], tc[j][4 * g + 1], tc[j][4 * g + 2], tc[j][4 * g + 3]);
        }
    }
}

}  // namespace

static int g_fe_half = 1;
void set_fe_half(int on) { g_fe_half = on ? 1 : 0; }
bool head_fe_half_ok(int D, int h0) { return g_fe_half && h0 == 32 && D % 32 == 0 && D <= 256; }
int head_wgrad_groups(int B) { return B < 768 ? B : 768; }      // three resident workgroups per CU

int launch_head_hidden_fwd(const PackedView &pk, const MbView &mb, int D, const float *FE, const float *C, const float *W1f,
                           const float *constb, float *hid, hipStream_t st) {
    if (mb.Nhe == 0) return 0;
    const size_t lds = sizeof(float) * 32 * (size_t)(D + 4);
#define UPAMD_HH(NP_)                                                                                                       \
    case NP_:                                                                                                               \
        hipLaunchKernelGGL((head_hidden_fwd_kernel<NP_>), dim3(mb.B), dim3(256), lds, st, pk, mb, FE, C, W1f, constb, hid); \
        break
    switch (D / 16) {
        UPAMD_HH(2); UPAMD_HH(4); UPAMD_HH(6); UPAMD_HH(8); UPAMD_HH(10); UPAMD_HH(12); UPAMD_HH(14); UPAMD_HH(16);
        default: return fail(UPAMD_E_INVALID, "head_hidden_fwd: D = %d not covered", D);
    }
#undef UPAMD_HH
    UPAMD_HIP(hipGetLastError());
    return 0;
}

int launch_head_wgrad(const PackedView &pk, const MbView &mb, int D, const float *FE, const float *C, const float *dprel,
                      float *slab, int *S_out, hipStream_t st) {
    const int G = head_wgrad_groups(mb.B);
    *S_out = G;
    if (D / 32 <= 4)
        hipLaunchKernelGGL((head_wgrad_kernel<1>), dim3(G), dim3(256), 0, st, pk, mb, D / 16, G, FE, C, dprel, slab);
    else
        hipLaunchKernelGGL((head_wgrad_kernel<2>), dim3(G), dim3(256), 0, st, pk, mb, D / 16, G, FE, C, dprel, slab);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

}  // namespace upamd
