// Ragged-graph kernels (gfx950, wave64): minibatch input gather, single-query attention, pointer
// heads (candidate features, masked softmax).  The message-passing kernels live in edge.hip.
#include "kernels.h"

namespace upamd {

__device__ __forceinline__ float fast_tanh(float x) {
    float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

#define META(t) (pk.meta + (int64_t)(t) * UPAMD_META_STRIDE)

constexpr int64_t LDS_LIMIT = 160 * 1024;




// ------------------------------------------------------------------------------------------
// Single-query attention over a graph's node_mask nodes (state_encoder.py:150-161 with the
// key/value projections collapsed: score_j = r . h_j, out = Wvv (sum_j alpha_j h_j) + bvv).
// One workgroup per graph.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_reduce_sum(float v, float *red) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_reduce_max(float v, float *red) {
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// acc[row j] = sum_d vec[d] * H[j][d] for the graph's rows (vec in LDS)
__device__ __forceinline__ float row_dot(const float *__restrict__ Hg, int64_t M, int NP, int j, const float *vec) {
    float acc = 0.f;
    for (int p = 0; p < NP; ++p) {
        const float4 *h4 = reinterpret_cast<const float4 *>(Hg + ((int64_t)p * M + j) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 h = h4[q];
            const float *vv = vec + p * 16 + q * 4;
            acc = fmaf(h.x, vv[0], acc); acc = fmaf(h.y, vv[1], acc); acc = fmaf(h.z, vv[2], acc); acc = fmaf(h.w, vv[3], acc);
        }
    }
    return acc;
}

// out[d] = sum_j wgt[j] * H[j][d]  (wgt in LDS, len n); 256 threads = 16 cols x 16 row groups
__device__ __forceinline__ void weighted_colsum(const float *__restrict__ Hg, int64_t M, int NP, int n,
                                                const float *wgt, float *part, float *__restrict__ out) {
    const int c = threadIdx.x & 15, jg = threadIdx.x >> 4;
    for (int p = 0; p < NP; ++p) {
        float acc = 0.f;
        for (int j = jg; j < n; j += 16) acc = fmaf(wgt[j], Hg[((int64_t)p * M + j) * 16 + c], acc);
        __syncthreads();
        part[jg * 16 + c] = acc;
        __syncthreads();
        if (threadIdx.x < 16) {
            float tot = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) tot += part[q * 16 + threadIdx.x];
            out[p * 16 + threadIdx.x] = tot;
        }
    }
}

__global__ __launch_bounds__(256) void attn_fwd_kernel(PackedView pk, MbView mb, int NP, int heads,
                                                       const float *__restrict__ HL, const float *__restrict__ r,
                                                       float *__restrict__ alpha, float *__restrict__ s) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int D = NP * 16;
    float *vec = reinterpret_cast<float *>(smem);          // [D]
    float *sc = vec + D;                                   // [max_n]
    float *part = sc + mb.max_n;                           // [256]
    float *red = part + 256;                               // [4]
    const int b = blockIdx.x, t = mb.idx[b];
    const int32_t *m = META(t);
    const int n = m[0];
    const int64_t o = mb.node_off[b], M = mb.M;
    const float *Hg = HL + o * 16;
    const uint8_t *nmask = pk.nmask + m[9];
    for (int h = 0; h < heads; ++h) {
        __syncthreads();
        for (int i = threadIdx.x; i < D; i += 256) vec[i] = r[((int64_t)b * heads + h) * D + i];
        __syncthreads();
        float mx = -INFINITY;
        for (int j = threadIdx.x; j < n; j += 256) {
            const float v = nmask[j] ? row_dot(Hg, M, NP, j, vec) : -INFINITY;
            sc[j] = v;
            mx = fmaxf(mx, v);
        }
        mx = block_reduce_max(mx, red);
        float sum = 0.f;
        for (int j = threadIdx.x; j < n; j += 256) {
            const float ex = expf(sc[j] - mx);
            sc[j] = ex;
            sum += ex;
        }
        sum = block_reduce_sum(sum, red);
        const float inv = 1.f / sum;
        for (int j = threadIdx.x; j < n; j += 256) {
            const float a = sc[j] * inv;
            sc[j] = a;
            alpha[(int64_t)h * M + o + j] = a;
        }
        __syncthreads();
        weighted_colsum(Hg, M, NP, n, sc, part, s + ((int64_t)b * heads + h) * D);
    }
}

// ------------------------------------------------------------------------------------------
// Single-pass variants (D = 16 * NP with NP a power of two <= 16; the kernels above stay as the generic path).
// 16 lanes per node: lane c owns column c of every panel, i.e. NP values of the node's row in registers, so the
// score (a 16-lane reduction) and the softmax-weighted sum use the row while it is in registers -- H^L is read
// ONCE (online softmax: every one of the 16 node slots keeps a running max / normaliser / weighted sum, the slots
// are merged in a fixed order at the end).  The two-pass kernels read it twice from HBM (2048 graphs x 277 KB do
// not stay in L2).
// ------------------------------------------------------------------------------------------
constexpr float LOG2E = 1.4426950408889634f;
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * LOG2E); }    // exp(-inf) = 0

__device__ __forceinline__ float reduce16(float v) {      // sum over the 16 lanes of a node slot, result on all of them
    v += __shfl_xor(v, 8, 16);
    v += __shfl_xor(v, 4, 16);
    v += __shfl_xor(v, 2, 16);
    v += __shfl_xor(v, 1, 16);
    return v;
}

template <int NP>
__global__ __launch_bounds__(256) void attn_fwd16_kernel(PackedView pk, MbView mb, int heads, const float *__restrict__ HL,
                                                         const float *__restrict__ r, float *__restrict__ alpha,
                                                         float *__restrict__ s) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int D = NP * 16;
    float *sacc = reinterpret_cast<float *>(smem);         // [16 slots][D]
    float *sm = sacc + 16 * D;                             // [16] running max
    float *sl = sm + 16;                                   // [16] running normaliser
    float *sc = sl + 16;                                   // [max_n] raw scores
    const int b = blockIdx.x;
    const int32_t *m = mb.rows + (int64_t)b * UPAMD_META_STRIDE;
    const int n = m[0];
    const int64_t o = m[14], M = mb.M;
    const float *Hg = HL + o * 16;
    const uint8_t *nmask = pk.nmask + m[9];
    const int c = threadIdx.x & 15, slot = threadIdx.x >> 4;
    for (int h = 0; h < heads; ++h) {
        float vec[NP], acc[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            vec[p] = r[((int64_t)b * heads + h) * D + p * 16 + c];
            acc[p] = 0.f;
        }
        float mrun = -INFINITY, lrun = 0.f;
        for (int j = slot; j < n; j += 16) {
            // the mask byte and the node's row are fetched TOGETHER (the row unconditionally, its dot product in front of the
            // branch): with the branch first every trip was two dependent memory round trips, and the kernel is latency-bound
            const bool live = nmask[j] != 0;
            float hv[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) hv[p] = Hg[((int64_t)p * M + j) * 16 + c];
            float dot = 0.f;
#pragma unroll
            for (int p = 0; p < NP; ++p) dot = fmaf(hv[p], vec[p], dot);
            dot = reduce16(dot);
            if (!live) continue;
            if (c == 0) sc[j] = dot;
            const float mnew = fmaxf(mrun, dot);
            const float keep = fast_exp(mrun - mnew), w = fast_exp(dot - mnew);
            lrun = fmaf(lrun, keep, w);
#pragma unroll
            for (int p = 0; p < NP; ++p) acc[p] = fmaf(acc[p], keep, w * hv[p]);
            mrun = mnew;
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) sacc[slot * D + p * 16 + c] = acc[p];
        if (c == 0) {
            sm[slot] = mrun;
            sl[slot] = lrun;
        }
        __syncthreads();
        float mx = -INFINITY;
#pragma unroll
        for (int q = 0; q < 16; ++q) mx = fmaxf(mx, sm[q]);
        float L = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) L = fmaf(sl[q], fast_exp(sm[q] - mx), L);
        const float invL = 1.f / L;
        for (int d = threadIdx.x; d < D; d += 256) {
            float tot = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) tot = fmaf(sacc[q * D + d], fast_exp(sm[q] - mx), tot);
            s[((int64_t)b * heads + h) * D + d] = tot * invL;
        }
        for (int j = threadIdx.x; j < n; j += 256) alpha[(int64_t)h * M + o + j] = nmask[j] ? fast_exp(sc[j] - mx) * invL : 0.f;
        __syncthreads();
    }
}

// backward, single pass over H^L:  with t_j = ds . h_j and T = sum_k alpha_k t_k,
//   dscore_j = alpha_j (t_j - T),   dr = sum_j dscore_j h_j = (sum_j alpha_j t_j h_j) - T * s     (s = the forward output)
// so the weighted sum is accumulated while t_j is computed; GL is then written without touching H^L again.
template <int NP>
__global__ __launch_bounds__(256) void attn_bwd16_kernel(PackedView pk, MbView mb, int heads, const float *__restrict__ HL,
                                                         const float *__restrict__ r, const float *__restrict__ alpha,
                                                         const float *__restrict__ s, const float *__restrict__ ds,
                                                         const float *__restrict__ dhbarV, int ld_dhbarV,
                                                         float *__restrict__ GL, float *__restrict__ dr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int D = NP * 16;
    float *sacc = reinterpret_cast<float *>(smem);         // [16 slots][D]
    float *sT = sacc + 16 * D;                             // [16]
    float *dsl = sT + 16;                                  // [heads][D]
    float *rl = dsl + heads * D;                           // [heads][D]
    float *al = rl + heads * D;                            // [heads][max_n] alpha
    float *dscl = al + heads * mb.max_n;                   // [heads][max_n] t_j, then dscore_j
    const int b = blockIdx.x;
    const int32_t *m = mb.rows + (int64_t)b * UPAMD_META_STRIDE;
    const int n = m[0];
    const int64_t o = m[14], M = mb.M;
    const float *Hg = HL + o * 16;
    const uint8_t *nmask = pk.nmask + m[9];
    const int c = threadIdx.x & 15, slot = threadIdx.x >> 4;
    for (int i = threadIdx.x; i < heads * D; i += 256) {
        dsl[i] = ds[(int64_t)b * heads * D + i];
        rl[i] = r[(int64_t)b * heads * D + i];
    }
    for (int h = 0; h < heads; ++h) {
        float *a_h = al + h * mb.max_n, *d_h = dscl + h * mb.max_n;
        float vec[NP], acc[NP];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            vec[p] = ds[((int64_t)b * heads + h) * D + p * 16 + c];
            acc[p] = 0.f;
        }
        float Tpart = 0.f;
        for (int j = slot; j < n; j += 16) {
            const float a = alpha[(int64_t)h * M + o + j];
            const bool live = nmask[j] != 0;
            float t = 0.f;
            float hv[NP];                          // fetched with the mask byte, not behind it (see attn_fwd16_kernel)
#pragma unroll
            for (int p = 0; p < NP; ++p) hv[p] = Hg[((int64_t)p * M + j) * 16 + c];
#pragma unroll
            for (int p = 0; p < NP; ++p) t = fmaf(hv[p], vec[p], t);
            t = reduce16(t);
            if (!live) t = 0.f;
            if (live) {
                const float wgt = a * t;
                Tpart += wgt;
#pragma unroll
                for (int p = 0; p < NP; ++p) acc[p] = fmaf(wgt, hv[p], acc[p]);
            }
            if (c == 0) {
                a_h[j] = a;
                d_h[j] = t;
            }
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) sacc[slot * D + p * 16 + c] = acc[p];
        if (c == 0) sT[slot] = Tpart;
        __syncthreads();
        float T = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) T += sT[q];
        for (int d = threadIdx.x; d < D; d += 256) {
            float tot = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) tot += sacc[q * D + d];
            dr[((int64_t)b * heads + h) * D + d] = tot - T * s[((int64_t)b * heads + h) * D + d];
        }
        for (int j = threadIdx.x; j < n; j += 256) d_h[j] = a_h[j] * (d_h[j] - T);
        __syncthreads();
    }
    const float inv_nm = 1.f / (float)m[6];
    if constexpr (NP % 4 == 0) {
        // 16-byte stores: lane (pg = c / 4, q = c % 4) of a node slot owns columns 4q .. 4q+3 of the panels pg, pg + 4, ... -- a
        // quarter of the store instructions of the column-per-lane form below for the same bytes and the same per-element
        // expression (element-wise: bit-identical)
        const int pg = c >> 2, q4 = 4 * (c & 3);
        float4 dhv4[NP >= 4 ? NP / 4 : 1];
#pragma unroll
        for (int i = 0; i < NP / 4; ++i) {
            const float *x = dhbarV + (int64_t)b * ld_dhbarV + (pg + 4 * i) * 16 + q4;      // (a column slice: no alignment promise)
            dhv4[i] = make_float4(x[0] * inv_nm, x[1] * inv_nm, x[2] * inv_nm, x[3] * inv_nm);
        }
        for (int j = slot; j < n; j += 16) {
            const bool live = nmask[j] != 0;
#pragma unroll
            for (int i = 0; i < NP / 4; ++i) {
                const int p = pg + 4 * i, d = p * 16 + q4;
                float4 v = live ? dhv4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                for (int h = 0; h < heads; ++h) {
                    const float a = al[h * mb.max_n + j], dsc = dscl[h * mb.max_n + j];
                    const float4 dv = *reinterpret_cast<const float4 *>(dsl + h * D + d), rv = *reinterpret_cast<const float4 *>(rl + h * D + d);
                    v.x += a * dv.x + dsc * rv.x;
                    v.y += a * dv.y + dsc * rv.y;
                    v.z += a * dv.z + dsc * rv.z;
                    v.w += a * dv.w + dsc * rv.w;
                }
                *reinterpret_cast<float4 *>(GL + ((int64_t)p * M + o + j) * 16 + q4) = v;
            }
        }
        return;
    }
    float dhv[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) dhv[p] = dhbarV[(int64_t)b * ld_dhbarV + p * 16 + c] * inv_nm;
    for (int j = slot; j < n; j += 16) {
        const bool live = nmask[j] != 0;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int d = p * 16 + c;
            float v = live ? dhv[p] : 0.f;
            for (int h = 0; h < heads; ++h)
                v += al[h * mb.max_n + j] * dsl[h * D + d] + dscl[h * mb.max_n + j] * rl[h * D + d];
            GL[((int64_t)p * M + o + j) * 16 + c] = v;
        }
    }
}

template <int NP>
static int launch_attn_fwd16(const PackedView &pk, const MbView &mb, int heads, const float *HL, const float *r, float *alpha,
                             float *s, hipStream_t st) {
    const size_t lds = sizeof(float) * (size_t)(16 * NP * 16 + 32 + mb.max_n);
    hipLaunchKernelGGL((attn_fwd16_kernel<NP>), dim3(mb.B), dim3(256), lds, st, pk, mb, heads, HL, r, alpha, s);
    UPAMD_HIP(hipGetLastError());
    return 0;
}
template <int NP>
static int launch_attn_bwd16(const PackedView &pk, const MbView &mb, int heads, const float *HL, const float *r,
                             const float *alpha, const float *s, const float *ds, const float *dhbarV, int ld_dhbarV, float *GL,
                             float *dr, hipStream_t st) {
    const size_t lds = sizeof(float) * (size_t)(16 * NP * 16 + 16 + 2 * heads * NP * 16 + 2 * heads * mb.max_n);
    if (lds > 64 * 1024) return -1;
    hipLaunchKernelGGL((attn_bwd16_kernel<NP>), dim3(mb.B), dim3(256), lds, st, pk, mb, heads, HL, r, alpha, s, ds, dhbarV,
                       ld_dhbarV, GL, dr);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

int launch_attn_fwd(const PackedView &pk, const MbView &mb, int D, int heads, const float *HL, const float *r,
                    float *alpha, float *s, hipStream_t st) {
    if (mb.rows && (size_t)(16 * D + 32 + mb.max_n) * sizeof(float) <= 64 * 1024) {
        switch (D / 16) {
            case 1: return launch_attn_fwd16<1>(pk, mb, heads, HL, r, alpha, s, st);
            case 2: return launch_attn_fwd16<2>(pk, mb, heads, HL, r, alpha, s, st);
            case 4: return launch_attn_fwd16<4>(pk, mb, heads, HL, r, alpha, s, st);
            case 8: return launch_attn_fwd16<8>(pk, mb, heads, HL, r, alpha, s, st);
            case 16: return launch_attn_fwd16<16>(pk, mb, heads, HL, r, alpha, s, st);
            default: break;
        }
    }
    const size_t lds = sizeof(float) * (size_t)(D + mb.max_n + 256 + 8);
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(mb.B), dim3(256), lds, st, pk, mb, D / 16, heads, HL, r, alpha, s);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

// backward: GL (written) = masked-mean term + attention terms;  dr[b,h,:] = sum_j dscore_j h_j
__global__ __launch_bounds__(256) void attn_bwd_kernel(PackedView pk, MbView mb, int NP, int heads,
                                                       const float *__restrict__ HL, const float *__restrict__ r,
                                                       const float *__restrict__ alpha, const float *__restrict__ ds,
                                                       const float *__restrict__ dhbarV, int ld_dhbarV,
                                                       float *__restrict__ GL, float *__restrict__ dr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int D = NP * 16;
    float *dsl = reinterpret_cast<float *>(smem);          // [heads][D]
    float *rl = dsl + heads * D;                           // [heads][D]
    float *al = rl + heads * D;                            // [heads][max_n]
    float *dscl = al + heads * mb.max_n;                   // [heads][max_n]
    float *part = dscl + heads * mb.max_n;                 // [256]
    float *red = part + 256;                               // [4]
    const int b = blockIdx.x, t = mb.idx[b];
    const int32_t *m = META(t);
    const int n = m[0];
    const int64_t o = mb.node_off[b], M = mb.M;
    const float *Hg = HL + o * 16;
    const uint8_t *nmask = pk.nmask + m[9];
    for (int i = threadIdx.x; i < heads * D; i += 256) {
        dsl[i] = ds[(int64_t)b * heads * D + i];
        rl[i] = r[(int64_t)b * heads * D + i];
    }
    __syncthreads();
    for (int h = 0; h < heads; ++h) {
        float *a_h = al + h * mb.max_n, *d_h = dscl + h * mb.max_n;
        float dot = 0.f;
        for (int j = threadIdx.x; j < n; j += 256) {
            const float a = alpha[(int64_t)h * M + o + j];
            const float da = nmask[j] ? row_dot(Hg, M, NP, j, dsl + h * D) : 0.f;
            a_h[j] = a;
            d_h[j] = da;
            dot = fmaf(a, da, dot);
        }
        dot = block_reduce_sum(dot, red);
        for (int j = threadIdx.x; j < n; j += 256) d_h[j] = a_h[j] * (d_h[j] - dot);
        __syncthreads();
        weighted_colsum(Hg, M, NP, n, d_h, part, dr + ((int64_t)b * heads + h) * D);
    }
    __syncthreads();
    const float inv_nm = 1.f / (float)m[6];
    for (int i = threadIdx.x; i < n * 16; i += 256) {
        const int j = i >> 4, cc = i & 15;
        const bool live = nmask[j] != 0;
        for (int p = 0; p < NP; ++p) {
            const int d = p * 16 + cc;
            float v = live ? dhbarV[(int64_t)b * ld_dhbarV + d] * inv_nm : 0.f;
            for (int h = 0; h < heads; ++h)
                v += al[h * mb.max_n + j] * dsl[h * D + d] + dscl[h * mb.max_n + j] * rl[h * D + d];
            GL[((int64_t)p * M + o + j) * 16 + cc] = v;
        }
    }
}

int launch_attn_bwd(const PackedView &pk, const MbView &mb, int D, int heads, const float *HL, const float *r,
                    const float *alpha, const float *s, const float *ds, const float *dhbarV, int ld_dhbarV, float *GL,
                    float *dr, hipStream_t st) {
    if (mb.rows && s) {
        int rc = -1;
        switch (D / 16) {
            case 1: rc = launch_attn_bwd16<1>(pk, mb, heads, HL, r, alpha, s, ds, dhbarV, ld_dhbarV, GL, dr, st); break;
            case 2: rc = launch_attn_bwd16<2>(pk, mb, heads, HL, r, alpha, s, ds, dhbarV, ld_dhbarV, GL, dr, st); break;
            case 4: rc = launch_attn_bwd16<4>(pk, mb, heads, HL, r, alpha, s, ds, dhbarV, ld_dhbarV, GL, dr, st); break;
            case 8: rc = launch_attn_bwd16<8>(pk, mb, heads, HL, r, alpha, s, ds, dhbarV, ld_dhbarV, GL, dr, st); break;
            case 16: rc = launch_attn_bwd16<16>(pk, mb, heads, HL, r, alpha, s, ds, dhbarV, ld_dhbarV, GL, dr, st); break;
            default: break;
        }
        if (rc >= 0) return rc;      // -1: shape not covered, fall through to the two-pass kernel
    }
    const size_t lds = sizeof(float) * (size_t)(2 * heads * D + 2 * heads * mb.max_n + 256 + 8);
    if (lds > (size_t)LDS_LIMIT) return fail(UPAMD_E_LIMIT, "attn_bwd: LDS need %zu too large", lds);
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&attn_bwd_kernel), (int64_t)lds)) return rc;
    hipLaunchKernelGGL(attn_bwd_kernel, dim3(mb.B), dim3(256), lds, st, pk, mb, D / 16, heads, HL, r, alpha, ds, dhbarV, ld_dhbarV, GL, dr);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// Land-use pointer head (state_encoder.py:207-210, policy.py:19-43) in factorised form: the candidate
// features FE = [m ; m*c] come from the last layer's edge_fwd kernel (edge.hip); here are the per-row bias,
// the per-graph segment sums and the feature backward.
// ------------------------------------------------------------------------------------------
// hid(pm)[row][k] = constb[b(row)][k]: the per-row bias (Wb - Wd) c_b + b1 of the factorised first Linear,
// written into the hidden buffer that the GEMM then accumulates onto (R == C in place)
__global__ __launch_bounds__(256) void he_bias_rows_kernel(PackedView pk, MbView mb, int h0,
                                                           const float *__restrict__ constb, float *__restrict__ hid) {
    const int b = blockIdx.x, t = mb.idx[b];
    const int nh = META(t)[2];
    const int64_t q0 = mb.he_off[b], NH = mb.Nhe;
    for (int i = threadIdx.x; i < nh * h0; i += 256) {
        const int q = i / h0, k = i % h0;
        hid[((int64_t)(k >> 4) * NH + q0 + q) * 16 + (k & 15)] = constb[(int64_t)b * h0 + k];
    }
}
int launch_he_bias_rows(const PackedView &pk, const MbView &mb, int h0, const float *constb, float *hid, hipStream_t st) {
    if (mb.Nhe == 0) return 0;
    hipLaunchKernelGGL(he_bias_rows_kernel, dim3(mb.B), dim3(256), 0, st, pk, mb, h0, constb, hid);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

// backward of FE = [m ; m*c]:  dMhe(pm)[row][d] = live * (g1 + g2 * c),  dC_head[b][d] = sum_rows g2 * m
__global__ __launch_bounds__(256) void he_feat_bwd_kernel(PackedView pk, MbView mb, int NP,
                                                          const float *__restrict__ FE, const float *__restrict__ C,
                                                          const float *__restrict__ dFE, float *__restrict__ dMhe,
                                                          float *__restrict__ dC_head, int keep_dead) {
    __shared__ float part[256];
    const int b = blockIdx.x / NP, p = blockIdx.x % NP, t = mb.idx[b];
    const int32_t *m = META(t);
    const int nh = m[2];
    const int D = NP * 16;
    const int c = threadIdx.x & 15, qg = threadIdx.x >> 4;
    const int64_t q0 = mb.he_off[b], NH = mb.Nhe;
    float acc = 0.f;
    const float cc = C[(int64_t)b * D + p * 16 + c];
    for (int q = qg; q < nh; q += 16) {
        const int64_t row = q0 + q;
        const float live = (keep_dead || pk.he_live[m[11] + q]) ? 1.f : 0.f;      // sgnn: a non-live candidate's message is the constant 0
        const float mm = FE[((int64_t)p * NH + row) * 16 + c];
        const float g1 = dFE[((int64_t)p * NH + row) * 16 + c], g2 = dFE[((int64_t)(NP + p) * NH + row) * 16 + c];
        dMhe[((int64_t)p * NH + row) * 16 + c] = live * (g1 + g2 * cc);
        acc += g2 * mm;
    }
    part[qg * 16 + c] = acc;
    __syncthreads();
    if (threadIdx.x < 16) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) tot += part[q * 16 + threadIdx.x];
        dC_head[(int64_t)b * D + p * 16 + threadIdx.x] = tot;
    }
}

int launch_he_feat_bwd(const PackedView &pk, const MbView &mb, int D, const float *FE, const float *C,
                       const float *dFE, float *dMhe, float *dC_head, hipStream_t st, int keep_dead) {
    hipLaunchKernelGGL(he_feat_bwd_kernel, dim3(mb.B * (D / 16)), dim3(256), 0, st, pk, mb, D / 16, FE, C, dFE, dMhe, dC_head, keep_dead);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

// The same backward with dFE = dpre W1f (K = h0 = 32) computed on the matrix cores inside the kernel: dFE ([candidates, 2D],
// 0.8 GB at B = 2048, D = 256) is neither written by a GEMM nor read back.  One workgroup per (graph, group of four 32-feature
// tiles), one tile per wave: v_mfma_f32_32x32x2_f32 with the FEATURES as the register dimension (A = W1f^T rows) and the
// candidates as the lane dimension (B = dpre rows), so lane (r, kh) ends up with features 8g + 4kh + t (g, t < 4) of candidate
// r -- whole 16-byte groups of the panel-major FE / dMhe rows.  The K order is free: lane half kh covers k = 16 kh .. 16 kh + 15
// (one 64-byte dpre panel row per lane).  dC_head = sum over the graph's candidates of g2 * m in a fixed order (per lane over
// its candidates, then a butterfly over the 32 lanes).
typedef float f32x16g __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void he_feat_bwd_fused_kernel(PackedView pk, MbView mb, int NP, const float *__restrict__ FE,
                                                                const float *__restrict__ C, const float *__restrict__ dprel,
                                                                const float *__restrict__ W1fT, float *__restrict__ dMhe,
                                                                float *__restrict__ dC_head, int keep_dead) {
    const int b = blockIdx.x, t = mb.idx[b];
    const int32_t *m = META(t);
    const int nh = m[2];
    const int D = NP * 16;
    const int lane = threadIdx.x & 63, r = lane & 31, kh = lane >> 5;
    const int ft = blockIdx.y * 4 + (threadIdx.x >> 6);            // this wave's 32-feature tile
    if (ft >= D / 32) return;
    const int64_t q0 = mb.he_off[b], NH = mb.Nhe;
    float4 wa[4], wc[4], cc[4];
    {
        const float4 *pa = reinterpret_cast<const float4 *>(W1fT + (int64_t)(32 * ft + r) * 32 + 16 * kh);
        const float4 *pc = reinterpret_cast<const float4 *>(W1fT + (int64_t)(D + 32 * ft + r) * 32 + 16 * kh);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            wa[q] = pa[q];
            wc[q] = pc[q];
            cc[q] = *reinterpret_cast<const float4 *>(C + (int64_t)b * D + 32 * ft + 8 * q + 4 * kh);
        }
    }
    f32x16g sum2;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum2[i] = 0.f;
    // one tile of 32 candidates: its operands are requested a whole tile ahead (the kernel streams ~1 GB with 0.85 us of
    // MFMA work per tile: without the prefetch every tile paid its own memory round trip)
    struct Tile {
        float4 x0, x1, x2, x3, mm[4];
        uint8_t live;
        int64_t grow;
        bool in;
    };
    const uint8_t *livep = pk.he_live + m[11];                     // (loaded once: inside fetch it would put a full wait in front of every prefetch)
    auto fetch = [&](int q) -> Tile {
        Tile T;
        const int row = q + r;
        T.in = row < nh;
        const int rc = T.in ? row : nh - 1;                        // clamped: loads stay unconditional
        T.grow = q0 + rc;
        T.live = livep[rc];                                        // (raw byte; first: the oldest load completes first)
        const float4 *pd = reinterpret_cast<const float4 *>(dprel + ((int64_t)kh * NH + T.grow) * 16);
        T.x0 = pd[0]; T.x1 = pd[1]; T.x2 = pd[2]; T.x3 = pd[3];
#pragma unroll
        for (int g = 0; g < 4; ++g)
            T.mm[g] = *reinterpret_cast<const float4 *>(FE + ((int64_t)(2 * ft + (g >> 1)) * NH + T.grow) * 16 + 8 * (g & 1) + 4 * kh);
        return T;
    };
    Tile cur;
    if (nh > 0) cur = fetch(0);
    for (int q = 0; q < nh; q += 32) {
        // unconditional (the last trip re-fetches its own tile): a branch around the prefetch would make the compiler wait for
        // ALL outstanding loads at the join, i.e. for the prefetch itself
        const Tile nxt = fetch(q + 32 < nh ? q + 32 : q);
        __builtin_amdgcn_sched_barrier(0);
        const bool in = cur.in;
        const int64_t grow = cur.grow;
        const float4 x0 = cur.x0, x1 = cur.x1, x2 = cur.x2, x3 = cur.x3;
        const float live = (keep_dead || cur.live) ? 1.f : 0.f;
        float4 mm[4] = {cur.mm[0], cur.mm[1], cur.mm[2], cur.mm[3]};
        f32x16g a1, a2;
#pragma unroll
        for (int i = 0; i < 16; ++i) { a1[i] = 0.f; a2[i] = 0.f; }
#define UPAMD_HF_STEP(W_, X_)                                                      \
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[W_].x, X_.x, a1, 0, 0, 0);    \
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[W_].x, X_.x, a2, 0, 0, 0);    \
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[W_].y, X_.y, a1, 0, 0, 0);    \
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[W_].y, X_.y, a2, 0, 0, 0);    \
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[W_].z, X_.z, a1, 0, 0, 0);    \
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[W_].z, X_.z, a2, 0, 0, 0);    \
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[W_].w, X_.w, a1, 0, 0, 0);    \
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(wc[W_].w, X_.w, a2, 0, 0, 0);
        UPAMD_HF_STEP(0, x0)
        UPAMD_HF_STEP(1, x1)
        UPAMD_HF_STEP(2, x2)
        UPAMD_HF_STEP(3, x3)
#undef UPAMD_HF_STEP
        if (in) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                // accumulators 4g .. 4g+3 = features 32 ft + 8g + 4kh + (0..3) of candidate `row`
                const float4 out = make_float4(live * fmaf(a2[4 * g + 0], cc[g].x, a1[4 * g + 0]), live * fmaf(a2[4 * g + 1], cc[g].y, a1[4 * g + 1]),
                                               live * fmaf(a2[4 * g + 2], cc[g].z, a1[4 * g + 2]), live * fmaf(a2[4 * g + 3], cc[g].w, a1[4 * g + 3]));
                *reinterpret_cast<float4 *>(dMhe + ((int64_t)(2 * ft + (g >> 1)) * NH + grow) * 16 + 8 * (g & 1) + 4 * kh) = out;
                sum2[4 * g + 0] = fmaf(a2[4 * g + 0], mm[g].x, sum2[4 * g + 0]);
                sum2[4 * g + 1] = fmaf(a2[4 * g + 1], mm[g].y, sum2[4 * g + 1]);
                sum2[4 * g + 2] = fmaf(a2[4 * g + 2], mm[g].z, sum2[4 * g + 2]);
                sum2[4 * g + 3] = fmaf(a2[4 * g + 3], mm[g].w, sum2[4 * g + 3]);
            }
        }
        cur = nxt;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float v = sum2[i];
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);
        sum2[i] = v;
    }
    if (r == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4 *>(dC_head + (int64_t)b * D + 32 * ft + 8 * g + 4 * kh) =
                make_float4(sum2[4 * g + 0], sum2[4 * g + 1], sum2[4 * g + 2], sum2[4 * g + 3]);
    }
}

static int g_he_fused = 1;
void set_he_feat_fused(int on) { g_he_fused = on ? 1 : 0; }
bool he_feat_bwd_fused_ok(int D, int h0) { return g_he_fused && h0 == 32 && D % 32 == 0; }
int launch_he_feat_bwd_fused(const PackedView &pk, const MbView &mb, int D, const float *FE, const float *C, const float *dprel,
                             const float *W1fT, float *dMhe, float *dC_head, hipStream_t st, int keep_dead) {
    const int tiles = D / 32;
    hipLaunchKernelGGL(he_feat_bwd_fused_kernel, dim3(mb.B, (tiles + 3) / 4), dim3(256), 0, st, pk, mb, D / 16, FE, C, dprel, W1fT, dMhe,
                       dC_head, keep_dead);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

// road head input: rows of H^L of the road_mask candidates (stage-1 rows), policy.py:58
__global__ __launch_bounds__(256) void road_gather_kernel(PackedView pk, MbView mb, int NP, const float *__restrict__ HL,
                                                          float *__restrict__ XR) {
    const int b = blockIdx.x, t = mb.idx[b];
    const int32_t *m = META(t);
    const int nr = m[3];
    const int64_t o = mb.node_off[b], M = mb.M, q0 = mb.rn_off[b], NR = mb.Nrn;
    const int c = threadIdx.x & 15;
    for (int q = threadIdx.x >> 4; q < nr; q += 16) {
        const int v = pk.rn_node[m[12] + q];
        for (int p = 0; p < NP; ++p) XR[((int64_t)p * NR + q0 + q) * 16 + c] = HL[((int64_t)p * M + o + v) * 16 + c];
    }
}
int launch_road_gather(const PackedView &pk, const MbView &mb, int D, const float *HL, float *XR, hipStream_t st) {
    if (mb.Nrn == 0) return 0;
    hipLaunchKernelGGL(road_gather_kernel, dim3(mb.B), dim3(256), 0, st, pk, mb, D / 16, HL, XR);
    UPAMD_HIP(hipGetLastError());
    return 0;
}
__global__ __launch_bounds__(256) void road_scatter_add_kernel(PackedView pk, MbView mb, int NP,
                                                               const float *__restrict__ dXR, float *__restrict__ GL) {
    const int b = blockIdx.x, t = mb.idx[b];
    const int32_t *m = META(t);
    const int nr = m[3];
    const int64_t o = mb.node_off[b], M = mb.M, q0 = mb.rn_off[b], NR = mb.Nrn;
    const int c = threadIdx.x & 15;
    for (int q = threadIdx.x >> 4; q < nr; q += 16) {      // candidate nodes of a row are distinct: no conflicts
        const int v = pk.rn_node[m[12] + q];
        for (int p = 0; p < NP; ++p) GL[((int64_t)p * M + o + v) * 16 + c] += dXR[((int64_t)p * NR + q0 + q) * 16 + c];
    }
}
int launch_road_scatter_add(const PackedView &pk, const MbView &mb, int D, const float *dXR, float *GL, hipStream_t st) {
    if (mb.Nrn == 0) return 0;
    hipLaunchKernelGGL(road_scatter_add_kernel, dim3(mb.B), dim3(256), 0, st, pk, mb, D / 16, dXR, GL);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// Masked-softmax pointer head over each row's candidate list (policy.py:49-52,58-61,87-104 with
// torch.distributions.Categorical): log_prob of the taken action, entropy.  One wave per row.
// Rows whose stage is neither 0 nor 1 get logp = entropy = 0 (policy.py:90-91).  A row with no
// valid candidate reproduces the reference's fp32 behaviour on an all-pad row (log_prob = entropy = 0).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}





// ------------------------------------------------------------------------------------------
// rl-mlp encoder (MLPStateEncoder, state_encoder.py:217-308): no message passing.  h_nodes = node_encoder(x) (H^0);
// an edge's embedding is node_encoder(features of ONE endpoint) = the H^0 row of that endpoint (a candidate that is not
// a live edge has zero features: its embedding is the bias).  One workgroup per (graph, 16-column panel).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mlp_pool_fwd_kernel(PackedView pk, MbView mb, int NP, const float *__restrict__ H0,
                                                           const float *__restrict__ be, const float *__restrict__ C,
                                                           float *__restrict__ hbarV, float *__restrict__ FE, int fe_full) {
    __shared__ float part[256];
    const int b = blockIdx.x / NP, p = blockIdx.x % NP;
    const int32_t *m = mb.rows + (int64_t)b * UPAMD_META_STRIDE;
    const int n = m[0], nh = m[2], D = NP * 16;
    const int64_t o = m[14], M = mb.M, NH = mb.Nhe, q0 = m[15];
    const int c = threadIdx.x & 15, slot = threadIdx.x >> 4;
    const uint8_t *nmask = pk.nmask + m[9];
    const float *Hg = H0 + ((int64_t)p * M + o) * 16;
    float acc = 0.f;
    for (int j = slot; j < n; j += 16)
        if (nmask[j]) acc += Hg[(int64_t)j * 16 + c];
    part[slot * 16 + c] = acc;
    __syncthreads();
    if (threadIdx.x < 16) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) tot += part[q * 16 + threadIdx.x];
        hbarV[(int64_t)b * D + p * 16 + threadIdx.x] = tot / (float)m[6];
    }
    if (FE && nh > 0) {
        const float cc = C[(int64_t)b * D + p * 16 + c], bias = be[p * 16 + c];
        for (int q = slot; q < nh; q += 16) {
            const float mm = pk.he_live[m[11] + q] ? Hg[(int64_t)pk.he_sel[m[11] + q] * 16 + c] : bias;
            FE[((int64_t)p * NH + q0 + q) * 16 + c] = mm;
            if (fe_full) FE[((int64_t)(NP + p) * NH + q0 + q) * 16 + c] = mm * cc;
        }
    }
}

int launch_mlp_pool_fwd(const PackedView &pk, const MbView &mb, int D, const float *H0, const float *be, const float *C,
                        float *hbarV, float *FE, hipStream_t st, int fe_full) {
    hipLaunchKernelGGL(mlp_pool_fwd_kernel, dim3(mb.B * (D / 16)), dim3(256), 0, st, pk, mb, D / 16, H0, be, C, hbarV, FE, fe_full);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void mlp_pool_bwd_kernel(PackedView pk, MbView mb, int NP, const float *__restrict__ dhbarV,
                                                           int ld, const float *__restrict__ dMhe, float *__restrict__ G0,
                                                           float *__restrict__ dbe_extra) {
    __shared__ float part[256];
    const int b = blockIdx.x / NP, p = blockIdx.x % NP;
    const int32_t *m = mb.rows + (int64_t)b * UPAMD_META_STRIDE;
    const int n = m[0], nh = dMhe ? m[2] : 0, D = NP * 16;
    const int64_t o = m[14], M = mb.M, NH = mb.Nhe, q0 = m[15];
    const int c = threadIdx.x & 15, slot = threadIdx.x >> 4;
    const uint8_t *nmask = pk.nmask + m[9];
    const int32_t *hp = pk.hinc_ptr + m[13];
    const uint16_t *hhe = pk.hinc_he + 2 * (int64_t)m[11];
    const float gmean = dhbarV[(int64_t)b * ld + p * 16 + c] / (float)m[6];
    // node-centric, fixed order: a node sums the candidates that list it AND selected it (a self-loop candidate is listed
    // twice in a row: counted once)
    for (int j = slot; j < n; j += 16) {
        float v = nmask[j] ? gmean : 0.f;
        if (nh > 0) {
            const int k0 = hp[j], k1 = hp[j + 1];
            for (int k = k0; k < k1; ++k) {
                const int h = hhe[k];
                if (pk.he_sel[m[11] + h] != j || (k > k0 && hhe[k - 1] == h)) continue;
                v += dMhe[((int64_t)p * NH + q0 + h) * 16 + c];
            }
        }
        G0[((int64_t)p * M + o + j) * 16 + c] = v;
    }
    // candidates that are not live edges: their embedding is the bias
    float extra = 0.f;
    for (int q = slot; q < nh; q += 16)
        if (!pk.he_live[m[11] + q]) extra += dMhe[((int64_t)p * NH + q0 + q) * 16 + c];
    part[slot * 16 + c] = extra;
    __syncthreads();
    if (threadIdx.x < 16) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) tot += part[q * 16 + threadIdx.x];
        dbe_extra[(int64_t)b * D + p * 16 + threadIdx.x] = tot;
    }
}

int launch_mlp_pool_bwd(const PackedView &pk, const MbView &mb, int D, const float *dhbarV, int ld, const float *dMhe, float *G0,
                        float *dbe_extra, hipStream_t st) {
    hipLaunchKernelGGL(mlp_pool_bwd_kernel, dim3(mb.B * (D / 16)), dim3(256), 0, st, pk, mb, D / 16, dhbarV, ld, dMhe, G0, dbe_extra);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

// ---- fused forms: the second (bias-free) Linear of the pointer heads is evaluated inside the softmax kernels -----------
//   z = w2 . hid[candidate]  (policy.py:19-43: Linear(h0, 1, bias=False) + Flatten), hid panel-major [h0/16][N][16]
__device__ __forceinline__ float cand_logit(const float *__restrict__ hid, int64_t N, int64_t row, int h0,
                                            const float *__restrict__ w2) {
    float z = 0.f;
    for (int p = 0; p < h0 / 16; ++p) {
        const float4 *h4 = reinterpret_cast<const float4 *>(hid + ((int64_t)p * N + row) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 h = h4[q];
            const float *w = w2 + p * 16 + q * 4;
            z = fmaf(h.x, w[0], z); z = fmaf(h.y, w[1], z); z = fmaf(h.z, w[2], z); z = fmaf(h.w, w[3], z);
        }
    }
    return z;
}

__global__ __launch_bounds__(256) void pointer_fwd2_kernel(PackedView pk, MbView mb, const float *__restrict__ hidl,
                                                           const float *__restrict__ w2l, int h0l,
                                                           const float *__restrict__ hidr, const float *__restrict__ w2r,
                                                           int h0r, float *__restrict__ z_he, float *__restrict__ z_rn,
                                                           float *__restrict__ p_he, float *__restrict__ p_rn,
                                                           float *__restrict__ logp, float *__restrict__ ent,
                                                           float *__restrict__ lse_out, float *__restrict__ ent_keep) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= mb.B) return;
    const int lane = threadIdx.x & 63;
    const int32_t *m = META(mb.idx[b]);
    const int stage = m[4];
    const int cnt = stage == 0 ? m[2] : (stage == 1 ? m[3] : 0);
    if (stage > 1 || cnt == 0) {
        // stage 2 rows: log_prob = entropy = 0 (policy.py:90-91); a row without any valid candidate: every logit is the
        // pad constant, whose logsumexp is absorbed in fp32, so the reference's normalised logits are all 0
        if (lane == 0) { logp[b] = 0.f; ent[b] = 0.f; lse_out[b] = 0.f; ent_keep[b] = 0.f; }
        return;
    }
    const bool land = stage == 0;
    const int64_t off = land ? mb.he_off[b] : mb.rn_off[b];
    const int64_t NC = land ? mb.Nhe : mb.Nrn;
    const float *hid = land ? hidl : hidr;
    const float *w2 = land ? w2l : w2r;
    const int h0 = land ? h0l : h0r;
    float *z = (land ? z_he : z_rn) + off;
    float *pp = (land ? p_he : p_rn) + off;
    float mx = -INFINITY;
    for (int i = lane; i < cnt; i += 64) {
        const float zi = cand_logit(hid, NC, off + i, h0, w2);
        z[i] = zi;                                 // kept for the backward (re-read below by the SAME lane only)
        mx = fmaxf(mx, zi);
    }
    mx = wave_max(mx);
    float sum = 0.f;
    for (int i = lane; i < cnt; i += 64) sum += expf(z[i] - mx);
    sum = wave_sum(sum);
    const float lse = mx + logf(sum);
    float pz = 0.f;
    for (int i = lane; i < cnt; i += 64) {
        const float lp = z[i] - lse;
        const float p = expf(lp);
        pp[i] = p;
        pz += p * lp;
    }
    pz = wave_sum(pz);
    if (lane == 0) {
        const int a = m[5];
        const float za = a >= 0 ? cand_logit(hid, NC, off + a, h0, w2) : -4294967296.0f;    // recomputed: another lane wrote z[a]
        logp[b] = za - lse;
        ent[b] = -pz;
        ent_keep[b] = -pz;
        lse_out[b] = lse;
    }
}

int launch_pointer_fwd2(const PackedView &pk, const MbView &mb, const float *hidl, const float *w2l, int h0l, const float *hidr,
                        const float *w2r, int h0r, float *z_he, float *z_rn, float *p_he, float *p_rn, float *logp, float *ent,
                        float *lse, float *ent_keep, hipStream_t st) {
    hipLaunchKernelGGL(pointer_fwd2_kernel, dim3((mb.B + 3) / 4), dim3(256), 0, st, pk, mb, hidl, w2l, h0l, hidr, w2r, h0r, z_he,
                       z_rn, p_he, p_rn, logp, ent, lse, ent_keep);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

// Backward of the pointer heads' tail (second Linear + masked softmax + log-prob / entropy), one workgroup per row:
//   dz_k = dlogp (delta_ka - p_k) - dent p_k (log p_k + H);   dpre[k][j] = dz_k * w2[j] * (1 - hid[k][j]^2)   (panel-major)
// and, in the same launch, the per-row sums every weight / bias gradient of the head is reduced from afterwards:
//   rs_dzh[b][j] = sum_k dz_k hid[k][j]   (-> dw2),     rs_dpre[b][j] = sum_k dpre[k][j]   (-> db1; for the land-use head
//   it is also the row's dconst, the gradient of the head's per-row bias term).
// A row writes the sums of its own head and zeros into the other head's (and a stage-2 row zeros into both).
// (Replaces four column-sum launches and the candidate segment sum of round 1.)
__global__ __launch_bounds__(256) void pointer_bwd2_kernel(PackedView pk, MbView mb, const float *__restrict__ z_he,
                                                           const float *__restrict__ z_rn, const float *__restrict__ p_he,
                                                           const float *__restrict__ p_rn, const float *__restrict__ ent,
                                                           const float *__restrict__ lse, const float *__restrict__ dlogp,
                                                           const float *__restrict__ dent, const float *__restrict__ hidl,
                                                           const float *__restrict__ w2l, int h0l,
                                                           const float *__restrict__ hidr, const float *__restrict__ w2r,
                                                           int h0r, float *__restrict__ dz_he, float *__restrict__ dz_rn,
                                                           float *__restrict__ dprel, float *__restrict__ dprer,
                                                           float *__restrict__ rs_dzh_l, float *__restrict__ rs_dpre_l,
                                                           float *__restrict__ rs_dzh_r, float *__restrict__ rs_dpre_r) {
    __shared__ float part1[256], part2[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int32_t *m = META(mb.idx[b]);
    const int stage = m[4];
    const bool land = stage == 0, active = stage <= 1;
    // the other head's row sums (and both, for a row without a head) are zero
    if (rs_dzh_l && !(active && land) && tid < h0l) { rs_dzh_l[(int64_t)b * h0l + tid] = 0.f; rs_dpre_l[(int64_t)b * h0l + tid] = 0.f; }
    if (rs_dzh_r && !(active && !land) && tid < h0r) { rs_dzh_r[(int64_t)b * h0r + tid] = 0.f; rs_dpre_r[(int64_t)b * h0r + tid] = 0.f; }
    if (!active) return;
    const int cnt = land ? m[2] : m[3];
    const int64_t off = land ? mb.he_off[b] : mb.rn_off[b];
    const int64_t NC = land ? mb.Nhe : mb.Nrn;
    const float *z = (land ? z_he : z_rn) + off;
    const float *pp = (land ? p_he : p_rn) + off;
    float *dz = (land ? dz_he : dz_rn) + off;
    const float *hid = land ? hidl : hidr;
    const float *w2 = land ? w2l : w2r;
    float *dpre = land ? dprel : dprer;
    const int h0 = land ? h0l : h0r;
    const float gl = dlogp[b], ge = dent[b], H = ent[b], ls = lse[b];
    const int a = m[5];
    auto dz_of = [&](int i) -> float {
        const float p = pp[i];
        float v = -gl * p - ge * p * ((z[i] - ls) + H);
        if (i == a) v += gl;
        return v;
    };
    for (int i = tid; i < cnt; i += 256) {
        const float v = dz_of(i);
        dz[i] = v;
        for (int pnl = 0; pnl < h0 / 16; ++pnl) {
            const int64_t o = ((int64_t)pnl * NC + off + i) * 16;
            const float4 *h4 = reinterpret_cast<const float4 *>(hid + o);
            float4 *d4 = reinterpret_cast<float4 *>(dpre + o);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 h = h4[q];
                const float *w = w2 + pnl * 16 + q * 4;
                d4[q] = make_float4(v * w[0] * (1.f - h.x * h.x), v * w[1] * (1.f - h.y * h.y), v * w[2] * (1.f - h.z * h.z),
                                    v * w[3] * (1.f - h.w * h.w));
            }
        }
    }
    // row sums: thread (group g, hidden unit j) walks the candidates g, g + G, ... in order; the G partial sums are
    // combined in a fixed order (bit-reproducible)
    float *rs_dzh = land ? rs_dzh_l : rs_dzh_r, *rs_dpre = land ? rs_dpre_l : rs_dpre_r;
    if (!rs_dzh) return;
    int hp = 16;
    while (hp < h0) hp <<= 1;
    const int G = 256 / hp;
    const int j = tid % hp, g = tid / hp;
    float a1 = 0.f, a2 = 0.f;
    if (j < h0) {
        const float wj = w2[j];
        const float *hj = hid + ((int64_t)(j >> 4) * NC + off) * 16 + (j & 15);
#pragma unroll 4
        for (int i = g; i < cnt; i += G) {
            const float v = dz_of(i), h = hj[(int64_t)i * 16];
            a1 = fmaf(v * wj, 1.f - h * h, a1);
            a2 = fmaf(v, h, a2);
        }
    }
    part1[g * hp + j] = a1;
    part2[g * hp + j] = a2;
    __syncthreads();
    if (tid < h0) {
        float t1 = 0.f, t2 = 0.f;
        for (int q = 0; q < G; ++q) { t1 += part1[q * hp + tid]; t2 += part2[q * hp + tid]; }
        rs_dpre[(int64_t)b * h0 + tid] = t1;
        rs_dzh[(int64_t)b * h0 + tid] = t2;
    }
}

int launch_pointer_bwd2(const PackedView &pk, const MbView &mb, const float *z_he, const float *z_rn, const float *p_he,
                        const float *p_rn, const float *ent, const float *lse, const float *dlogp, const float *dent,
                        const float *hidl, const float *w2l, int h0l, const float *hidr, const float *w2r, int h0r, float *dz_he,
                        float *dz_rn, float *dprel, float *dprer, float *rs_dzh_l, float *rs_dpre_l, float *rs_dzh_r,
                        float *rs_dpre_r, hipStream_t st) {
    if (h0l > 256 || h0r > 256) return fail(UPAMD_E_LIMIT, "pointer heads: hidden width > 256 is not supported");
    hipLaunchKernelGGL(pointer_bwd2_kernel, dim3(mb.B), dim3(256), 0, st, pk, mb, z_he, z_rn, p_he, p_rn, ent, lse, dlogp, dent, hidl,
                       w2l, h0l, hidr, w2r, h0r, dz_he, dz_rn, dprel, dprer, rs_dzh_l, rs_dpre_l, rs_dzh_r, rs_dpre_r);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

}  // namespace upamd
