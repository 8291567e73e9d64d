// Fused "small" kernels of the training step (gfx950, wave64).
//
// Round 1 ran the per-sample part of the network -- numerical encoder, current-node encoder, the attention's dense
// projections, the value head -- the parameter preparation (collapsed / transposed / padded weights) and every small
// weight-gradient reduction as ~125 micro-launches of 4-15 us per optimizer step; at the reference's own model size
// (D = 16) the whole step was nothing but those.  Here they become a handful of launches:
//
//   permute_kernel    every gather-style parameter preparation (pads, transposes, P/Q row order, head factorisation)
//   gsmm_kernel       grouped strided small matmuls (collapsed weight products forward, their gradients backward)
//   chain_*_kernel    the per-sample [B, .] chains: R = 8 rows per workgroup held in LDS, one thread per output
//                     column, weights streamed from L2 in [K][N] order (coalesced); forward before / after the graph
//                     part, backward after / before it
//   gtn_kernel        grouped weight-gradient products dY^T X over the B rows on the fp32 MFMA (operands straight
//                     from global memory: 128-byte coalesced rows), split over row ranges into slabs
//   greduce_kernel    ONE fixed-order reduction of every slab of the step into the flat gradient buffer
//
// Every reduction keeps a fixed order: results are bit-reproducible run to run.
//
// Replaces the autograd of the reference's nn.Sequential / nn.Linear / nn.MultiheadAttention projections on [B, .]
// tensors (urban_planning/models/state_encoder.py:35-57,150-161,187-191,204-205; value.py:15-39).
#include <type_traits>

#include "kernels.h"

namespace upamd {

typedef float f32x16c __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float tanh_c(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

template <typename J>
__device__ __forceinline__ int find_job(const J *jobs, int n, int unit, int J::*begin) {
    int k = 0;
    while (k + 1 < n && unit >= jobs[k + 1].*begin) ++k;
    return k;
}

// ------------------------------------------------------------------------------------------ permute
__global__ __launch_bounds__(256) void permute_kernel(PermJobs P) {
    const int k = find_job(P.j, P.n, (int)blockIdx.x, &PermJob::blk_begin);
    const PermJob &J = P.j[k];
    const int g = ((int)blockIdx.x - J.blk_begin) * 256 + threadIdx.x;
    switch (J.kind) {
        case PERM_PAD_COLS: {                   // dst[r][aux] = src[r][cols] zero-padded
            if (g >= J.rows * J.aux) return;
            const int r = g / J.aux, c = g % J.aux;
            J.dst[g] = c < J.cols ? J.src[(int64_t)r * J.cols + c] : 0.f;
            break;
        }
        case PERM_TRANSPOSE: {                  // dst[c][rows] = src[r][cols]  (+ optional plain copy into dst2)
            if (g >= J.rows * J.cols) return;
            const int r = g / J.cols, c = g % J.cols;
            const float v = J.src[g];
            J.dst[(int64_t)c * J.rows + r] = v;
            if (J.dst2) J.dst2[g] = v;
            break;
        }
        case PERM_WCAT: {                       // W [D][2D] = [Wa | Wb] -> Wcat [2D][D] (rows in P/Q pair order, kernels.h) + transpose
            const int D = J.rows;
            if (g >= 2 * D * D) return;
            const int jp = g / D, kk = g % D;
            const int row = pq_col(jp), half = pq_side(jp);
            const float v = J.src[(int64_t)row * 2 * D + half * D + kk];
            J.dst[g] = v;
            if (J.dst2) J.dst2[(int64_t)kk * 2 * D + jp] = v;
            break;
        }
        case PERM_LAND_HEAD: {                  // W1 [h0][4D] -> W1f [h0][2D], Wbd [h0][D], W1fT [2D][h0], WbdT [D][h0]
            const int h0 = J.rows, D = J.cols;
            if (g >= h0 * D) return;
            const int kk = g / D, d = g % D;
            const float *w = J.src + (int64_t)kk * 4 * D;
            const float a = w[d] + w[3 * D + d], c = w[2 * D + d], bd = w[D + d] - w[3 * D + d];
            J.dst[(int64_t)kk * 2 * D + d] = a;
            J.dst[(int64_t)kk * 2 * D + D + d] = c;
            J.dst2[g] = bd;
            if (J.dst3) {
                J.dst3[(int64_t)d * h0 + kk] = a;
                J.dst3[(int64_t)(D + d) * h0 + kk] = c;
            }
            if (J.dst4) J.dst4[(int64_t)d * h0 + kk] = bd;
            break;
        }
        case PERM_LAND_SCATTER: {               // gW1 [h0][4D] += blocks of dW1f (src) and dWbd (src2)
            const int h0 = J.rows, D = J.cols;
            if (g >= h0 * D) return;
            const int kk = g / D, d = g % D;
            const float a = J.src[(int64_t)kk * 2 * D + d], cg = J.src[(int64_t)kk * 2 * D + D + d], bd = J.src2[g];
            float *w = J.dst + (int64_t)kk * 4 * D;
            w[d] += a;
            w[D + d] += bd;
            w[2 * D + d] += cg;
            w[3 * D + d] += a - bd;
            break;
        }
        default: break;
    }
}

int launch_permute(const PermJobs &P, int blocks, hipStream_t st) {
    if (P.n == 0 || blocks == 0) return 0;
    hipLaunchKernelGGL(permute_kernel, dim3(blocks), dim3(256), 0, st, P);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

int perm_add(PermJobs *P, int *blocks, int kind, const float *src, const float *src2, float *dst, float *dst2, float *dst3,
             float *dst4, int rows, int cols, int aux) {
    if (P->n >= PERM_MAX_JOBS) return fail(UPAMD_E_LIMIT, "too many permute jobs");
    PermJob &J = P->j[P->n++];
    J.kind = kind; J.src = src; J.src2 = src2; J.dst = dst; J.dst2 = dst2; J.dst3 = dst3; J.dst4 = dst4;
    J.rows = rows; J.cols = cols; J.aux = aux; J.blk_begin = *blocks;
    int64_t elems = (int64_t)rows * cols;
    if (kind == PERM_PAD_COLS) elems = (int64_t)rows * aux;
    if (kind == PERM_WCAT) elems = 2LL * rows * rows;
    *blocks += (int)((elems + 255) / 256);
    return 0;
}

// ------------------------------------------------------------------------------------------ grouped small matmul
// C[i*ldc + j] (=|+=) scale * ( sum_k A[i*sa0 + k*sa1] * B[k*sb0 + j*sb1] + bias[j] ) + u[i] * v[j]
__global__ __launch_bounds__(256) void gsmm_kernel(SmmJobs P) {
    // 32 x 32 output tile per workgroup (2 x 2 per thread): these products are a few MFLOP each, so the tile is sized for
    // workgroup count (a D x D product at D = 256 fills 64 CUs), not for reuse
    __shared__ __attribute__((aligned(16))) float As[32][34];
    __shared__ __attribute__((aligned(16))) float Bs[32][34];
    const int k = find_job(P.j, P.n, (int)blockIdx.x, &SmmJob::blk_begin);
    const SmmJob &J = P.j[k];
    const int t = (int)blockIdx.x - J.blk_begin;
    const int i0 = (t / J.tiles_j) * 32, j0 = (t % J.tiles_j) * 32;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int I = J.I, Jn = J.J, K = J.K;
    const int sa0 = (int)J.sa0, sa1 = (int)J.sa1, sb0 = (int)J.sb0, sb1 = (int)J.sb1;
    const float *__restrict__ A = J.A, *__restrict__ B = J.B;
    float acc[2][2] = {};
    const bool a_k_fast = (sa1 == 1), b_j_fast = (sb1 == 1);
    for (int k0 = 0; k0 < K; k0 += 32) {
        float ra[4], rb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {           // all eight loads of the step in flight before the LDS commit
            const int e = tid + 256 * q;
            int ai, ak, bj, bk;
            if (a_k_fast) { ak = e & 31; ai = e >> 5; } else { ai = e & 31; ak = e >> 5; }
            if (b_j_fast) { bj = e & 31; bk = e >> 5; } else { bk = e & 31; bj = e >> 5; }
            const int gi = i0 + ai, gk = k0 + ak, gj = j0 + bj, gk2 = k0 + bk;
            ra[q] = (gi < I && gk < K) ? A[gi * sa0 + gk * sa1] : 0.f;
            rb[q] = (gj < Jn && gk2 < K) ? B[gk2 * sb0 + gj * sb1] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q;
            int ai, ak, bj, bk;
            if (a_k_fast) { ak = e & 31; ai = e >> 5; } else { ai = e & 31; ak = e >> 5; }
            if (b_j_fast) { bj = e & 31; bk = e >> 5; } else { bk = e & 31; bj = e >> 5; }
            As[ak][ai] = ra[q];
            Bs[bk][bj] = rb[q];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            const float2 a = *reinterpret_cast<const float2 *>(&As[kk][ty * 2]);
            const float2 b = *reinterpret_cast<const float2 *>(&Bs[kk][tx * 2]);
            acc[0][0] = fmaf(a.x, b.x, acc[0][0]); acc[0][1] = fmaf(a.x, b.y, acc[0][1]);
            acc[1][0] = fmaf(a.y, b.x, acc[1][0]); acc[1][1] = fmaf(a.y, b.y, acc[1][1]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const int gi = i0 + ty * 2 + x;
        if (gi >= I) continue;
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int gj = j0 + tx * 2 + y;
            if (gj >= Jn) continue;
            float v = acc[x][y];
            if (J.bias) v += J.bias[gj];
            v *= J.scale;
            if (J.u) v = fmaf(J.u[gi], J.v[gj], v);
            float *dst = J.C + gi * J.ldc + gj;
            const float out = J.accumulate ? (*dst + v) : v;
            *dst = out;
            if (J.CT) J.CT[gj * J.ldct + gi] = out;
        }
    }
}

int smm_add(SmmJobs *P, int *blocks, int I, int Jn, int K, const float *A, int64_t sa0, int64_t sa1, const float *B, int64_t sb0,
            int64_t sb1, const float *bias, float *C, int64_t ldc, int accumulate, float scale, float *CT, int64_t ldct,
            const float *u, const float *v) {
    if (I <= 0 || Jn <= 0) return 0;
    if (P->n >= SMM_MAX_JOBS) return fail(UPAMD_E_LIMIT, "too many grouped-matmul jobs");
    SmmJob &J = P->j[P->n++];
    J.A = A; J.B = B; J.bias = bias; J.C = C; J.CT = CT; J.u = u; J.v = v;
    J.sa0 = sa0; J.sa1 = sa1; J.sb0 = sb0; J.sb1 = sb1; J.ldc = ldc; J.ldct = ldct;
    J.I = I; J.J = Jn; J.K = K; J.accumulate = accumulate; J.scale = scale;
    J.blk_begin = *blocks;
    J.tiles_j = (Jn + 31) / 32;
    *blocks += ((I + 31) / 32) * J.tiles_j;
    return 0;
}

int launch_gsmm(const SmmJobs &P, int blocks, hipStream_t st) {
    if (P.n == 0 || blocks == 0) return 0;
    hipLaunchKernelGGL(gsmm_kernel, dim3(blocks), dim3(256), 0, st, P);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------ per-sample chains
// Activations of the workgroup's R rows live in LDS as act[k][R] (the R rows of one feature are contiguous: a
// broadcast ds_read_b128 pair per feature).  out[n][r] = act( bias[n] + sum_k in[k][r] * Mat[k * ld + n] ) * scale:
// one thread per output column n, Mat read coalesced over n.  Narrow layers (N < 128) split the K range over
// 256 / npad thread groups and combine the partial sums in a fixed order.
constexpr int CH_R = 8;
constexpr int CH_PART = 1024;      // floats per row of the partial-sum scratch: `part` is [CH_PART][R] (32 KB)

// Scalar form (any N, ld): one thread per output column, full K range per thread (narrow layers split K over thread groups).
__device__ __forceinline__ void lin_rows_scalar(const float *__restrict__ Mat, int ld, int K, int N, const float *in,
                                                const float *__restrict__ bias, float *out, int act, float scale, float *part) {
    constexpr int R = CH_R;
    const int tid = threadIdx.x;
    if (N >= 128) {
        for (int n = tid; n < N; n += 256) {
            float acc[R];
            const float b = bias ? bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = b;
            const float *mp = Mat + n;
            int k = 0;
            // 16 independent (coalesced) weight loads in flight, then the FMAs.  (Deeper batches were measured: 32 and 64
            // loads in flight make two of the four chain kernels slower -- 64 exceeds what the 6-bit vmcnt counter tracks.)
            for (; k + 16 <= K; k += 16) {
                float w[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) w[u] = mp[(int64_t)(k + u) * ld];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const float4 i0 = *reinterpret_cast<const float4 *>(in + (k + u) * R), i1 = *reinterpret_cast<const float4 *>(in + (k + u) * R + 4);
                    acc[0] = fmaf(w[u], i0.x, acc[0]); acc[1] = fmaf(w[u], i0.y, acc[1]); acc[2] = fmaf(w[u], i0.z, acc[2]); acc[3] = fmaf(w[u], i0.w, acc[3]);
                    acc[4] = fmaf(w[u], i1.x, acc[4]); acc[5] = fmaf(w[u], i1.y, acc[5]); acc[6] = fmaf(w[u], i1.z, acc[6]); acc[7] = fmaf(w[u], i1.w, acc[7]);
                }
            }
            for (; k < K; ++k) {
                const float w = mp[(int64_t)k * ld];
                const float4 i0 = *reinterpret_cast<const float4 *>(in + k * R), i1 = *reinterpret_cast<const float4 *>(in + k * R + 4);
                acc[0] = fmaf(w, i0.x, acc[0]); acc[1] = fmaf(w, i0.y, acc[1]); acc[2] = fmaf(w, i0.z, acc[2]); acc[3] = fmaf(w, i0.w, acc[3]);
                acc[4] = fmaf(w, i1.x, acc[4]); acc[5] = fmaf(w, i1.y, acc[5]); acc[6] = fmaf(w, i1.z, acc[6]); acc[7] = fmaf(w, i1.w, acc[7]);
            }
#pragma unroll
            for (int r = 0; r < R; ++r) out[n * R + r] = (act ? tanh_c(acc[r]) : acc[r]) * scale;
        }
    } else {
        int npad = 16;
        while (npad < N) npad <<= 1;
        const int G = 256 / npad, n = tid % npad, g = tid / npad;
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.f;
        if (n < N) {
            int k = g;
            for (; k + 7 * G < K; k += 8 * G) {
                float w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) w[u] = Mat[(int64_t)(k + u * G) * ld + n];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float *ip = in + (k + u * G) * R;
                    const float4 i0 = *reinterpret_cast<const float4 *>(ip), i1 = *reinterpret_cast<const float4 *>(ip + 4);
                    acc[0] = fmaf(w[u], i0.x, acc[0]); acc[1] = fmaf(w[u], i0.y, acc[1]); acc[2] = fmaf(w[u], i0.z, acc[2]); acc[3] = fmaf(w[u], i0.w, acc[3]);
                    acc[4] = fmaf(w[u], i1.x, acc[4]); acc[5] = fmaf(w[u], i1.y, acc[5]); acc[6] = fmaf(w[u], i1.z, acc[6]); acc[7] = fmaf(w[u], i1.w, acc[7]);
                }
            }
            for (; k < K; k += G) {
                const float w = Mat[(int64_t)k * ld + n];
                const float4 i0 = *reinterpret_cast<const float4 *>(in + k * R), i1 = *reinterpret_cast<const float4 *>(in + k * R + 4);
                acc[0] = fmaf(w, i0.x, acc[0]); acc[1] = fmaf(w, i0.y, acc[1]); acc[2] = fmaf(w, i0.z, acc[2]); acc[3] = fmaf(w, i0.w, acc[3]);
                acc[4] = fmaf(w, i1.x, acc[4]); acc[5] = fmaf(w, i1.y, acc[5]); acc[6] = fmaf(w, i1.z, acc[6]); acc[7] = fmaf(w, i1.w, acc[7]);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) part[(g * npad + n) * R + r] = acc[r];
        __syncthreads();
        if (tid < N) {
            const float b = bias ? bias[tid] : 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float tot = b;
                for (int q = 0; q < G; ++q) tot += part[(q * npad + tid) * R + r];
                out[tid * R + r] = (act ? tanh_c(tot) : tot) * scale;
            }
        }
    }
    __syncthreads();
}

// The chains are latency-bound: with a column per thread a D x D layer is a serial run of K / 16 weight-load round trips
// (16 x ~1 us at D = 256), and a step runs ~12 of them back to back in four launches -- 0.27 ms of a 2.7 ms step at 256
// rows per GPU (the strong-scaling point), unchanged at 2048.  Vector form (N, ld multiples of 4, N <= 1024): a thread owns
// FOUR adjacent columns (one 16-byte weight load per k, coalesced over the group) and the K range is split over the
// G = 256 / tpg thread groups, so a thread's serial run is K / (16 G) round trips (4 instead of 16 at D = 256); the group
// partials go through `part` as [g][r][n] (conflict-free both ways) and are combined in the fixed order g = 0 .. G-1.
__device__ __forceinline__ void lin_rows(const float *__restrict__ Mat, int ld, int K, int N, const float *in,
                                         const float *__restrict__ bias, float *out, int act, float scale, float *part) {
    constexpr int R = CH_R;
    // narrow layers (N < 128) already split K in the scalar form, and their many small groups would only lengthen the combine
    if (N < 128 || (N & 3) || (ld & 3) || N > CH_PART || (reinterpret_cast<uintptr_t>(Mat) & 15)) {
        lin_rows_scalar(Mat, ld, K, N, in, bias, out, act, scale, part);
        return;
    }
    const int tid = threadIdx.x;
    int tpg = 1;                                   // threads per K group: the next power of two >= N / 4
    while (tpg * 4 < N) tpg <<= 1;
    const int G = 256 / tpg, g = tid / tpg, c0 = 4 * (tid % tpg);
    const int kc = (K + G - 1) / G;
    const int k0 = min(K, g * kc), k1 = min(K, k0 + kc);
    float acc[4][R];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < R; ++r) acc[c][r] = 0.f;
    if (c0 < N) {
        const float *mp = Mat + c0;
        int k = k0;
        for (; k + 16 <= k1; k += 16) {
            float4 w[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) w[u] = *reinterpret_cast<const float4 *>(mp + (int64_t)(k + u) * ld);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const float4 i0 = *reinterpret_cast<const float4 *>(in + (k + u) * R), i1 = *reinterpret_cast<const float4 *>(in + (k + u) * R + 4);
                const float iv[R] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    acc[0][r] = fmaf(w[u].x, iv[r], acc[0][r]);
                    acc[1][r] = fmaf(w[u].y, iv[r], acc[1][r]);
                    acc[2][r] = fmaf(w[u].z, iv[r], acc[2][r]);
                    acc[3][r] = fmaf(w[u].w, iv[r], acc[3][r]);
                }
                // keep the scheduler from hoisting all 16 x 8 activation reads above the FMAs (128 extra live registers)
                if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (; k < k1; ++k) {
            const float4 w = *reinterpret_cast<const float4 *>(mp + (int64_t)k * ld);
            const float4 i0 = *reinterpret_cast<const float4 *>(in + k * R), i1 = *reinterpret_cast<const float4 *>(in + k * R + 4);
            const float iv[R] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w};
#pragma unroll
            for (int r = 0; r < R; ++r) {
                acc[0][r] = fmaf(w.x, iv[r], acc[0][r]);
                acc[1][r] = fmaf(w.y, iv[r], acc[1][r]);
                acc[2][r] = fmaf(w.z, iv[r], acc[2][r]);
                acc[3][r] = fmaf(w.w, iv[r], acc[3][r]);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
            *reinterpret_cast<float4 *>(part + ((int64_t)g * R + r) * N + c0) = make_float4(acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
    }
    __syncthreads();
    for (int n = tid; n < N; n += 256) {
        const float b = bias ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float tot = b;
            for (int q = 0; q < G; ++q) tot += part[((int64_t)q * R + r) * N + n];
            out[n * R + r] = (act ? tanh_c(tot) : tot) * scale;
        }
    }
    __syncthreads();
}

// global row-major [B][ld] <-> LDS [N][R]
__device__ __forceinline__ void rows_load(const float *__restrict__ g, int64_t ld, int N, int b0, int nr, float *lds) {
    for (int i = threadIdx.x; i < N * CH_R; i += 256) {
        const int r = i / N, n = i % N;
        lds[n * CH_R + r] = r < nr ? g[(int64_t)(b0 + r) * ld + n] : 0.f;
    }
}
__device__ __forceinline__ void rows_store(float *__restrict__ g, int64_t ld, int N, int b0, int nr, const float *lds) {
    for (int i = threadIdx.x; i < N * nr; i += 256) {
        const int r = i / N, n = i % N;
        g[(int64_t)(b0 + r) * ld + n] = lds[n * CH_R + r];
    }
}

#define CHMETA(t) (a.pk.meta + (int64_t)(t) * UPAMD_META_STRIDE)

// ---- forward, before the graph part: row descriptors, inputs, numerical encoder, current-node encoder, query
// projections q0 / q1 / r, the land-use head's per-row bias; the extra workgroups gather the node features (Xp).
__global__ __launch_bounds__(256) void chain_fwd_pre_kernel(ChainFwdPre a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const ChainDims &d = a.d;
    const int nchain = (d.B + CH_R - 1) / CH_R;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= nchain) {
        if (a.part == CHAIN_LAYERS) return;
        // node features of graph b -> panel-major, K padded to 32; column 31 = 1 (a weight-gradient GEMM against Xp then
        // also yields the column sums; the forward weights of that column are zero padding)
        const int b = (int)blockIdx.x - nchain;
        const int t = a.mb.idx[b];
        const int32_t *m = CHMETA(t);
        const int n = m[0];
        const int64_t src0 = m[9], o = a.mb.node_off[b], M = a.mb.M;
        for (int i = tid; i < n * 8; i += 256) {
            const int v = i >> 3, q = i & 7;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < 6) val = *reinterpret_cast<const float4 *>(a.pk.X + (src0 + v) * UPAMD_NODE_PAD + q * 4);
            if (q == 7) val.w = 1.0f;
            const int panel = q >> 2, c4 = (q & 3) * 4;
            *reinterpret_cast<float4 *>(a.Xp + ((int64_t)panel * M + o + v) * 16 + c4) = val;
        }
        return;
    }
    const int b0 = (int)blockIdx.x * CH_R;
    const int nr = min(CH_R, d.B - b0);
    constexpr int R = CH_R;
    // LDS carve
    float *bufA = lds;                              // [maxnum][R]
    float *bufB = bufA + d.maxnum * R;
    float *cur = bufB + d.maxnum * R;               // [24][R]
    float *Cc = cur + UPAMD_NODE_PAD * R;           // [D][R]
    float *q0 = Cc + d.D * R;
    float *q1 = q0 + d.D * R;
    float *rr = q1 + d.D * R;                       // [heads * D][R]
    float *cb = rr + d.heads * d.D * R;             // [h0l][R]
    float *part = cb + d.h0l * R;                   // [CH_PART][R]
    // row descriptors (MbView::rows): meta row of the state + minibatch offsets
    for (int i = tid; a.part != CHAIN_LAYERS && i < nr * UPAMD_META_STRIDE; i += 256) {
        const int b = b0 + i / UPAMD_META_STRIDE, c = i % UPAMD_META_STRIDE;
        int32_t v;
        if (c == 14) v = a.mb.node_off[b];
        else if (c == 15) v = a.mb.he_off[b];
        else v = a.pk.meta[(int64_t)a.mb.idx[b] * UPAMD_META_STRIDE + c];
        a.rows[(int64_t)b * UPAMD_META_STRIDE + c] = v;
    }
    if (a.part == CHAIN_GATHER) return;
    for (int i = tid; i < d.Fn * R; i += 256) {
        const int r = i / d.Fn, k = i % d.Fn;
        const float v = r < nr ? a.pk.numerical[(int64_t)a.mb.idx[b0 + r] * d.Fn + k] : 0.f;
        bufA[k * R + r] = v;
        if (r < nr) a.U[0][(int64_t)(b0 + r) * d.Fn + k] = v;
    }
    for (int i = tid; i < UPAMD_NODE_PAD * R; i += 256) {
        const int r = i / UPAMD_NODE_PAD, k = i % UPAMD_NODE_PAD;
        const float v = r < nr ? a.pk.cur[(int64_t)a.mb.idx[b0 + r] * UPAMD_NODE_PAD + k] : 0.f;
        cur[k * R + r] = v;
        if (r < nr) a.curg[(int64_t)(b0 + r) * UPAMD_NODE_PAD + k] = v;
    }
    __syncthreads();
    // numerical encoder (state_encoder.py:35-57,187)
    {
        float *in = bufA, *out = bufB;
        int K = d.Fn;
        for (int i = 0; i < d.n_num; ++i) {
            const int N = d.num_hidden[i];
            lin_rows(a.WnT[i], N, K, N, in, a.bn[i], out, 1, 1.f, part);
            rows_store(a.U[i + 1], N, N, b0, nr, out);
            float *tmp = in; in = out; out = tmp;
            K = N;
        }
    }
    // current node through the node encoder (state_encoder.py:191)
    lin_rows(a.WeT, d.D, d.F, d.D, cur, a.be, Cc, 0, 1.f, part);
    rows_store(a.C, d.D, d.D, b0, nr, Cc);
    if (d.mlp) {
        // rl-mlp: mean edge embedding = node_encoder(mean of the selected endpoints' raw features) (state_encoder.py:290,296);
        // the mean features also become rows [B, 2B) of the node-encoder weight-gradient job, rows [2B, 3B) stay zero
        for (int i = tid; i < UPAMD_NODE_PAD * R; i += 256) {
            const int r = i / UPAMD_NODE_PAD, k = i % UPAMD_NODE_PAD;
            const float v = r < nr ? a.pk.xbar[(int64_t)a.mb.idx[b0 + r] * UPAMD_NODE_PAD + k] : 0.f;
            cur[k * R + r] = v;
            if (r < nr) {
                a.curg[(int64_t)(d.B + b0 + r) * UPAMD_NODE_PAD + k] = v;
                a.curg[(int64_t)(2 * d.B + b0 + r) * UPAMD_NODE_PAD + k] = 0.f;
            }
        }
        __syncthreads();
        lin_rows(a.WeT, d.D, d.F, d.D, cur, a.be, q0, 0, 1.f, part);
        rows_store(a.hbarE, d.D, d.D, b0, nr, q0);
        if (a.constb) {
            lin_rows(a.WbdT, d.h0l, d.D, d.h0l, Cc, a.b1l, cb, 0, 1.f, part);
            rows_store(a.constb, d.h0l, d.h0l, b0, nr, cb);
        }
        return;
    }
    // attention query path (state_encoder.py:150-156 + nn.MultiheadAttention's q in-projection and scaling)
    lin_rows(a.WqT, d.D, d.D, d.D, Cc, a.bq, q0, 0, 1.f, part);
    rows_store(a.q0, d.D, d.D, b0, nr, q0);
    lin_rows(a.WiqT, d.D, d.D, d.D, q0, a.biq, q1, 0, d.scale, part);
    rows_store(a.q1, d.D, d.D, b0, nr, q1);
    // r[h][j] = sum_{i in head h} q1[i] Wkk[i][j]: per head a [dh] x [D] product on the head's rows of Wkk
    for (int h = 0; h < d.heads; ++h)
        lin_rows(a.Wkk + (int64_t)h * d.dh * d.D, d.D, d.dh, d.D, q1 + h * d.dh * R, nullptr, rr + h * d.D * R, 0, 1.f, part);
    rows_store(a.r, (int64_t)d.heads * d.D, d.heads * d.D, b0, nr, rr);
    // land-use head: the c-only part of the first Linear becomes a per-row bias (policy.py:19-43, factorised)
    if (a.constb) {
        lin_rows(a.WbdT, d.h0l, d.D, d.h0l, Cc, a.b1l, cb, 0, 1.f, part);
        rows_store(a.constb, d.h0l, d.h0l, b0, nr, cb);
    }
}

int64_t chain_fwd_pre_lds(const ChainDims &d) {
    return sizeof(float) * (int64_t)CH_R * (2 * d.maxnum + UPAMD_NODE_PAD + 3 * d.D + d.heads * d.D + d.h0l + CH_PART);
}

// ---- forward, after the attention: o = Wvv s + bvv, out-projection, state_value, value head
__global__ __launch_bounds__(256) void chain_fwd_post_kernel(ChainFwdPost a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const ChainDims &d = a.d;
    constexpr int R = CH_R;
    const int tid = threadIdx.x;
    const int b0 = (int)blockIdx.x * CH_R;
    const int nr = min(CH_R, d.B - b0);
    float *ss = lds;                                // [heads * D][R]
    float *oo = ss + d.heads * d.D * R;             // [D][R]
    float *att = oo + d.D * R;                      // [D][R]
    float *sv = att + d.D * R;                      // [Wp][R]
    float *v1 = sv + d.Wp * R;                      // [maxval][R]
    float *v2 = v1 + d.maxval * R;
    float *part = v2 + d.maxval * R;                // [CH_PART][R]
    if (!d.mlp) rows_load(a.s, (int64_t)d.heads * d.D, d.heads * d.D, b0, nr, ss);
    __syncthreads();
    // o[i] = bvv[i] + sum_j s[h(i)][j] Wvv[i][j]      (WvvT[j][i]): per head the head's dh output columns
    for (int h = 0; !d.mlp && h < d.heads; ++h)
        lin_rows(a.WvvT + h * d.dh, d.D, d.D, d.dh, ss + h * d.D * R, a.bvv + h * d.dh, oo + h * d.dh * R, 0, 1.f, part);
    if (!d.mlp) {
        rows_store(a.o, d.D, d.D, b0, nr, oo);
        lin_rows(a.WoT, d.D, d.D, d.D, oo, a.bo, att, 0, 1.f, part);
        rows_store(a.att, d.D, d.D, b0, nr, att);
    }
    const int natt = d.mlp ? 0 : d.D;       // rl-mlp: no attended current node in state_value (state_encoder.py:299-300)
    // state_value = [h_num ; mean nodes ; mean edges ; attended current node ; stage]  (state_encoder.py:204-205)
    for (int i = tid; i < d.Wp * R; i += 256) {
        const int r = i / d.Wp, j = i % d.Wp;
        float v = 0.f;
        if (r < nr) {
            const int64_t b = b0 + r;
            if (j < d.S_last) v = a.Ulast[b * d.S_last + j];
            else if (j < d.S_last + d.D) v = a.hbarV[b * d.D + j - d.S_last];
            else if (j < d.S_last + 2 * d.D) v = a.hbarE[b * d.D + j - d.S_last - d.D];
            else if (j < d.S_last + 2 * d.D + natt) v = att[(j - d.S_last - 2 * d.D) * R + r];
            else if (j < d.W) v = (a.rows[b * UPAMD_META_STRIDE + 4] == (j - d.S_last - 2 * d.D - natt)) ? 1.f : 0.f;
            a.SV[b * d.Wp + j] = v;
        }
        sv[j * R + r] = v;
    }
    __syncthreads();
    // value head (value.py:15-39)
    {
        const float *in = sv;
        int K = d.W;
        float *out = v1;
        for (int i = 0; i < d.n_value; ++i) {
            const int N = d.value_hidden[i];
            const bool last = i == d.n_value - 1;
            lin_rows(a.WvT[i], N, K, N, in, a.bv[i], out, last ? 0 : 1, 1.f, part);
            if (last) {
                if (tid < nr) a.value[b0 + tid] = out[tid];           // N == 1: out[0 * R + r]
            } else {
                rows_store(a.V[i + 1], N, N, b0, nr, out);
            }
            in = out;
            out = (out == v1) ? v2 : v1;
            K = N;
        }
    }
}

int64_t chain_fwd_post_lds(const ChainDims &d) {
    return sizeof(float) * (int64_t)CH_R * (d.heads * d.D + 2 * d.D + d.Wp + 2 * d.maxval + CH_PART);
}

// ---- backward, the part after the attention in forward order: value head, numerical encoder, out-projection, Wvv.
// dX[k] = sum_n dY[n] W[n][k]: the [N][K] weights are read as they are (coalesced over k).
__global__ __launch_bounds__(256) void chain_bwd_post_kernel(ChainBwdPost a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const ChainDims &d = a.d;
    constexpr int R = CH_R;
    const int tid = threadIdx.x;
    const int b0 = (int)blockIdx.x * CH_R;
    const int nr = min(CH_R, d.B - b0);
    float *zA = lds;                                // [Wp][R]  (ping)
    float *zB = zA + d.Wp * R;                      // [Wp][R]  (pong)
    float *yy = zB + d.Wp * R;                      // [max(maxval, maxnum, D)][R] saved activations
    float *dd = yy + d.maxdim * R;                  // [D][R] do
    float *gzA = dd + d.D * R;                      // [maxnum][R] numerical-encoder gradients (ping / pong)
    float *gzB = gzA + d.maxnum * R;
    float *part = gzB + d.maxnum * R;
    // ---- value head
    float *dz = zA, *dn = zB;
    for (int i = tid; i < R; i += 256) dz[i] = i < nr ? a.dvalue[b0 + i] : 0.f;      // [1][R]
    __syncthreads();
    for (int i = d.n_value - 1; i >= 0; --i) {
        const int N = d.value_hidden[i];
        const int K = (i == 0) ? d.W : d.value_hidden[i - 1];
        if (i < d.n_value - 1) {                   // through the tanh of layer i: dz *= 1 - y^2
            rows_load(a.V[i + 1], N, N, b0, nr, yy);
            __syncthreads();
            for (int e = tid; e < N * R; e += 256) dz[e] *= 1.f - yy[e] * yy[e];
            __syncthreads();
        }
        rows_store(a.dAv[i], N, N, b0, nr, dz);
        lin_rows(a.Wv[i], K, N, K, dz, nullptr, dn, 0, 1.f, part);
        float *tmp = dz; dz = dn; dn = tmp;
    }
    // dz = dSV [W][R]; kept in global with row stride Wp (the graph kernels read the dhbarV / dhbarE column slices)
    for (int i = tid; i < d.Wp * nr; i += 256) {
        const int r = i / d.Wp, j = i % d.Wp;
        a.dSV[(int64_t)(b0 + r) * d.Wp + j] = j < d.W ? dz[j * R + r] : 0.f;
    }
    float *dsv = dz;                               // stays valid: the loops below write into `dn` / dd
    // ---- numerical encoder
    {
        float *gz = gzA, *gn = gzB;
        for (int e = tid; e < d.S_last * R; e += 256) gz[e] = dsv[e];
        __syncthreads();
        for (int i = d.n_num - 1; i >= 0; --i) {
            const int N = d.num_hidden[i];
            const int K = (i == 0) ? d.Fn : d.num_hidden[i - 1];
            rows_load(a.U[i + 1], N, N, b0, nr, yy);
            __syncthreads();
            for (int e = tid; e < N * R; e += 256) gz[e] *= 1.f - yy[e] * yy[e];
            __syncthreads();
            rows_store(a.dAn[i], N, N, b0, nr, gz);
            if (i > 0) {
                lin_rows(a.Wn[i], K, N, K, gz, nullptr, gn, 0, 1.f, part);
                float *tmp = gz; gz = gn; gn = tmp;
            }
        }
    }
    __syncthreads();
    if (d.mlp) return;
    // ---- attention output path: datt -> do = Wo^T datt -> ds[h] = Wvv[h-slice]^T do[h-slice]
    const float *datt = dsv + (d.S_last + 2 * d.D) * R;
    rows_store(a.datt, d.D, d.D, b0, nr, datt);
    lin_rows(a.Wo, d.D, d.D, d.D, datt, nullptr, dd, 0, 1.f, part);
    rows_store(a.dov, d.D, d.D, b0, nr, dd);
    for (int h = 0; h < d.heads; ++h) {            // ds[h][j] = sum_{i in head h} do[i] Wvv[i][j]   (yy is free here)
        lin_rows(a.Wvv + (int64_t)h * d.dh * d.D, d.D, d.dh, d.D, dd + h * d.dh * R, nullptr, yy, 0, 1.f, part);
        rows_store(a.ds + h * d.D, (int64_t)d.heads * d.D, d.D, b0, nr, yy);
        __syncthreads();
    }
}

int64_t chain_bwd_post_lds(const ChainDims &d) {
    return sizeof(float) * (int64_t)CH_R * (2 * d.Wp + d.maxdim + d.D + 2 * d.maxnum + CH_PART);
}

// ---- backward, the part before the graph in forward order: dr -> dq1 -> dq0 -> dC (+ the pointer head's terms)
__global__ __launch_bounds__(256) void chain_bwd_pre_kernel(ChainBwdPre a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const ChainDims &d = a.d;
    constexpr int R = CH_R;
    const int tid = threadIdx.x;
    const int b0 = (int)blockIdx.x * CH_R;
    const int nr = min(CH_R, d.B - b0);
    float *drr = lds;                               // [heads * D][R]
    float *g1 = drr + d.heads * d.D * R;            // [D][R]
    float *g0 = g1 + d.D * R;
    float *gc = g0 + d.D * R;
    float *dcb = gc + d.D * R;                      // [h0l][R]
    float *part = dcb + d.h0l * R;
    if (!d.mlp) rows_load(a.dr, (int64_t)d.heads * d.D, d.heads * d.D, b0, nr, drr);
    if (a.dconst) rows_load(a.dconst, d.h0l, d.h0l, b0, nr, dcb);
    __syncthreads();
    if (d.mlp) {
        // no query path: dC = dconst Wbd + the land-use head's m*c term; dhbarE goes behind dC (rows [B, 2B))
        for (int e = tid; e < d.D * R; e += 256) gc[e] = 0.f;
        __syncthreads();
        if (a.dconst) {
            lin_rows(a.Wbd, d.D, d.h0l, d.D, dcb, nullptr, g1, 0, 1.f, part);
            for (int e = tid; e < d.D * R; e += 256) gc[e] += g1[e];
            __syncthreads();
        }
        for (int i = tid; i < d.D * nr; i += 256) {
            const int r = i / d.D, n = i % d.D;
            float v = gc[n * R + r];
            if (a.dC_head) v += a.dC_head[(int64_t)(b0 + r) * d.D + n];
            a.dC[(int64_t)(b0 + r) * d.D + n] = v;
            a.dC[(int64_t)(d.B + b0 + r) * d.D + n] = a.dSV[(int64_t)(b0 + r) * d.Wp + d.S_last + d.D + n];
        }
        return;
    }
    // dq1[i] = scale * sum_j dr[h(i)][j] Wkk[i][j]      (WkkT[j][i])
    for (int h = 0; h < d.heads; ++h)
        lin_rows(a.WkkT + h * d.dh, d.D, d.D, d.dh, drr + h * d.D * R, nullptr, g1 + h * d.dh * R, 0, d.scale, part);
    rows_store(a.dq1, d.D, d.D, b0, nr, g1);
    lin_rows(a.Wiq, d.D, d.D, d.D, g1, nullptr, g0, 0, 1.f, part);
    rows_store(a.dq0, d.D, d.D, b0, nr, g0);
    lin_rows(a.Wq, d.D, d.D, d.D, g0, nullptr, gc, 0, 1.f, part);
    if (a.dconst) {                                 // + dconst Wbd  (the land-use head's per-row bias term)
        lin_rows(a.Wbd, d.D, d.h0l, d.D, dcb, nullptr, g1, 0, 1.f, part);
        for (int e = tid; e < d.D * R; e += 256) gc[e] += g1[e];
        __syncthreads();
    }
    for (int i = tid; i < d.D * nr; i += 256) {
        const int r = i / d.D, n = i % d.D;
        float v = gc[n * R + r];
        if (a.dC_head) v += a.dC_head[(int64_t)(b0 + r) * d.D + n];
        a.dC[(int64_t)(b0 + r) * d.D + n] = v;
    }
}

int64_t chain_bwd_pre_lds(const ChainDims &d) {
    return sizeof(float) * (int64_t)CH_R * (d.heads * d.D + 3 * d.D + d.h0l + CH_PART);
}

template <typename A>
static int launch_chain(void (*kern)(A), const A &a, int blocks, int64_t lds, hipStream_t st) {
    if (lds > 160 * 1024 - 512) return fail(UPAMD_E_LIMIT, "per-sample chain needs %lld bytes of LDS (model too wide)", (long long)lds);
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds)) return rc;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), (size_t)lds, st, a);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

int launch_chain_fwd_pre(const ChainFwdPre &a, hipStream_t st) {
    const int nchain = (a.d.B + CH_R - 1) / CH_R;
    // CHAIN_GATHER keeps the full grid (its chain workgroups write the row descriptors) but needs no LDS
    return launch_chain(chain_fwd_pre_kernel, a, nchain + (a.part == CHAIN_LAYERS ? 0 : a.d.B),
                        a.part == CHAIN_GATHER ? 0 : chain_fwd_pre_lds(a.d), st);
}
int launch_chain_fwd_post(const ChainFwdPost &a, hipStream_t st) {
    return launch_chain(chain_fwd_post_kernel, a, (a.d.B + CH_R - 1) / CH_R, chain_fwd_post_lds(a.d), st);
}
int launch_chain_bwd_post(const ChainBwdPost &a, hipStream_t st) {
    return launch_chain(chain_bwd_post_kernel, a, (a.d.B + CH_R - 1) / CH_R, chain_bwd_post_lds(a.d), st);
}
int launch_chain_bwd_pre(const ChainBwdPre &a, hipStream_t st) {
    return launch_chain(chain_bwd_pre_kernel, a, (a.d.B + CH_R - 1) / CH_R, chain_bwd_pre_lds(a.d), st);
}

// ------------------------------------------------------------------------------------------ grouped dY^T X on the MFMA
// slab[split][n][k] = sum_{rows of the split} A[row][n] * X[row][k]   (X == nullptr: a column of ones, K = 1).
// One wave per (job, 32 x 32 output tile, row split); v_mfma_f32_32x32x2_f32 with the two rows of a step as its k pair;
// both operands are 128-byte coalesced row segments read straight from global memory (the tensors are L2-resident).
__global__ __launch_bounds__(256) void gtn_kernel(TnJobs P) {
    const int gw = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (gw >= P.total_waves) return;
    const int k = find_job(P.j, P.n, gw, &TnJob::wave_begin);
    const TnJob &J = P.j[k];
    const int local = gw - J.wave_begin;
    const int split = local % J.splits, tile = local / J.splits;
    const int tn = tile % J.tiles_n, tk = tile / J.tiles_n;
    const int lane = threadIdx.x & 63, l31 = lane & 31, lhi = lane >> 5;
    const int r0 = split * J.chunk, r1 = min(J.rows, r0 + J.chunk);
    const int n = tn * 32 + l31, kk = tk * 32 + l31;
    const bool nok = n < J.N, kok = kk < J.K;
    // per-lane column offsets and the stride between consecutive rows.  Row-major: [rows][ld];  panel-major:
    // [cols/16][rows][16].  Out-of-range lanes / rows read a valid (clamped) address and are zeroed by `am` / `xm`: the
    // loads stay unconditional, so all 16 of an iteration are in flight together.
    const int nc = nok ? n : 0, kc = kok ? kk : 0;
    const int64_t acol = J.a_pm ? ((int64_t)(nc >> 4) * J.rows_total * 16 + (nc & 15)) : nc;
    const int64_t xcol = J.x_pm ? ((int64_t)(kc >> 4) * J.rows_total * 16 + (kc & 15)) : kc;
    const int64_t astep = J.a_pm ? 16 : J.lda, xstep = J.x_pm ? 16 : J.ldx;
    const float *ap = J.A + acol, *xp = J.X ? J.X + xcol : nullptr;
    const int last = r1 - 1;
    f32x16c acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    for (int r = r0; r < r1; r += 16) {
        float av[8], xv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int row = r + 2 * u + lhi;
            const int rc = row < r1 ? row : last;
            av[u] = ap[(int64_t)rc * astep];
            xv[u] = xp ? xp[(int64_t)rc * xstep] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const bool in = r + 2 * u + lhi < r1;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32((in && nok) ? av[u] : 0.f, (in && kok) ? xv[u] : 0.f, acc, 0, 0, 0);
        }
    }
    float *slab = J.slab + (int64_t)split * J.N * J.K;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int nn = tn * 32 + (q & 3) + 8 * (q >> 2) + 4 * lhi;
        if (nn < J.N && kok) slab[(int64_t)nn * J.K + kk] = acc[q];
    }
}

int tn_job_splits(int64_t rows) {
    int64_t s = (rows + 255) / 256;      // >= 256 rows per split
    if (s > 512) s = 512;
    if (s < 1) s = 1;
    return (int)s;
}

int tn_add(TnJobs *P, const float *A, int64_t lda, int N, const float *X, int64_t ldx, int K, int64_t rows, float *slab, int *S_out,
           int a_pm, int x_pm) {
    if (P->n >= TN_MAX_JOBS) return fail(UPAMD_E_LIMIT, "too many grouped weight-gradient jobs");
    TnJob &J = P->j[P->n++];
    J.A = A; J.X = X; J.slab = slab; J.lda = lda; J.ldx = ldx; J.N = N; J.K = K; J.rows = (int)rows; J.rows_total = rows;
    J.a_pm = a_pm; J.x_pm = x_pm;
    J.tiles_n = (N + 31) / 32; J.tiles_k = (K + 31) / 32;
    J.splits = tn_job_splits(rows);
    J.chunk = (int)(((rows + J.splits - 1) / J.splits + 15) / 16 * 16);
    J.wave_begin = P->total_waves;
    P->total_waves += J.tiles_n * J.tiles_k * J.splits;
    *S_out = J.splits;
    return 0;
}

int launch_gtn(const TnJobs &P, hipStream_t st) {
    if (P.n == 0 || P.total_waves == 0) return 0;
    hipLaunchKernelGGL(gtn_kernel, dim3((P.total_waves + 3) / 4), dim3(256), 0, st, P);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------ grouped slab reduction
// dst (+)= sum_s slab[s][i][j] in a FIXED order (16 slab groups summed by 16 threads per element, then combined 0..15).
// modes: 0 dst[i*ldd + j] for j < jkeep (+ dst2[i] += the j == J-1 column);  1 dst[j*ldd + i];
//        2 GCN un-permute: slab row i = P/Q position (pair order) -> dst[pq_col(i)*ldd + pq_side(i)*J + j];
//        3 I == 1, J == 2D in P/Q pair order: dst[column] += the P entries, dst2[j] = the whole vector (may be null)
__global__ __launch_bounds__(1024) void greduce_kernel(RedJobs P) {
    __shared__ float part[16][65];
    const int k = find_job(P.j, P.n, (int)blockIdx.x, &RedJob::blk_begin);
    const RedJob &J = P.j[k];
    const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int ij = ((int)blockIdx.x - J.blk_begin) * 64 + e;
    const bool in = ij < J.I * J.J;
    float acc = 0.f;
    if (in) {
        // eight slab loads in flight per thread (clamped, unconditional); one at a time left the reduction at ~1 TB/s
        for (int s0 = g; s0 < J.S; s0 += 16 * 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = J.slab[(int64_t)min(s0 + 16 * u, J.S - 1) * J.sstride + ij];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (s0 + 16 * u < J.S) acc += v[u];
        }
    }
    part[g][e] = acc;
    __syncthreads();
    if (g != 0 || !in) return;
    acc = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc += part[q][e];
    const int i = ij / J.J, j = ij % J.J;
    if (J.mode == 0) {
        if (j < J.jkeep) {
            float *p = J.dst + (int64_t)i * J.ldd + j;
            *p = J.overwrite ? acc : *p + acc;
        }
        if (J.dst2 && j == J.J - 1) J.dst2[i] += acc;
    } else if (J.mode == 1) {
        float *p = J.dst + (int64_t)j * J.ldd + i;
        *p = J.overwrite ? acc : *p + acc;
    } else if (J.mode == 2) {
        const int row = pq_col(i), half = pq_side(i);
        J.dst[(int64_t)row * J.ldd + half * J.J + j] += acc;
    } else {
        if (pq_side(j) == 0) J.dst[pq_col(j)] += acc;
        if (J.dst2) J.dst2[j] = acc;
    }
}

int red_add(RedJobs *P, int *blocks, const float *slab, int S, int64_t sstride, int I, int Jn, int mode, int jkeep, float *dst,
            int ldd, float *dst2, int overwrite) {
    if (I <= 0 || Jn <= 0) return 0;
    if (P->n >= RED_MAX_JOBS) return fail(UPAMD_E_LIMIT, "too many slab-reduction jobs");
    RedJob &J = P->j[P->n++];
    J.slab = slab; J.dst = dst; J.dst2 = dst2; J.S = S; J.sstride = sstride; J.I = I; J.J = Jn; J.mode = mode; J.jkeep = jkeep;
    J.ldd = ldd; J.overwrite = overwrite; J.blk_begin = *blocks;
    *blocks += (I * Jn + 63) / 64;
    return 0;
}

int launch_greduce(const RedJobs &P, int blocks, hipStream_t st) {
    if (P.n == 0 || blocks == 0) return 0;
    hipLaunchKernelGGL(greduce_kernel, dim3(blocks), dim3(1024), 0, st, P);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

}  // namespace upamd
