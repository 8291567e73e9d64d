// Small dense / per-sample kernels (gfx950): strided matmul for the [B, .] MLPs and weight products,
// reductions, PPO loss, GAE, Adam, first-step gradient clipping.
#include "kernels.h"

namespace upamd {

__device__ __forceinline__ float fast_tanh_d(float x) {
    float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// ------------------------------------------------------------------------------------------
// C[i*ldc + j] (=|+=) out_scale * act( sum_k A[i*sa0 + k*sa1] * B[k*sb0 + j*sb1] + bias[j] )
// 64x64 output tile per workgroup, 4x4 per thread, K staged through LDS in steps of 16.  Loads walk
// whichever stride is 1 so that any of X W^T, dY W, dY^T X is coalesced.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void smm_kernel(int I, int J, int K, const float *__restrict__ A, int64_t sa0,
                                                  int64_t sa1, const float *__restrict__ B, int64_t sb0, int64_t sb1,
                                                  const float *__restrict__ bias, float *__restrict__ C, int64_t ldc,
                                                  int accumulate, int act_tanh, float out_scale, int kchunk) {
    __shared__ __attribute__((aligned(16))) float As[16][68];
    __shared__ __attribute__((aligned(16))) float Bs[16][68];
    const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    float acc[4][4] = {};
    const bool a_k_fast = (sa1 == 1);
    const bool b_j_fast = (sb1 == 1);
    // split-K: slice blockIdx.z of the reduction range, partial result into slab z of C
    const int kbeg = blockIdx.z * kchunk;
    if (gridDim.z > 1) {
        C += (int64_t)blockIdx.z * I * ldc;
        K = (kbeg + kchunk < K) ? kbeg + kchunk : K;
    }
    for (int k0 = kbeg; k0 < K; k0 += 16) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = tid + 256 * q;
            int ai, ak, bj, bk;
            if (a_k_fast) { ak = e & 15; ai = e >> 4; } else { ai = e & 63; ak = e >> 6; }
            if (b_j_fast) { bj = e & 63; bk = e >> 6; } else { bk = e & 15; bj = e >> 4; }
            const int gi = i0 + ai, gk = k0 + ak;
            As[ak][ai] = (gi < I && gk < K) ? A[gi * sa0 + gk * sa1] : 0.f;
            const int gj = j0 + bj, gk2 = k0 + bk;
            Bs[bk][bj] = (gj < J && gk2 < K) ? B[gk2 * sb0 + gj * sb1] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float4 a = *reinterpret_cast<const float4 *>(&As[k][ty * 4]);
            const float4 b = *reinterpret_cast<const float4 *>(&Bs[k][tx * 4]);
            const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = fmaf(av[x], bv[y], acc[x][y]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        const int gi = i0 + ty * 4 + x;
        if (gi >= I) continue;
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const int gj = j0 + tx * 4 + y;
            if (gj >= J) continue;
            float v = acc[x][y];
            if (bias) v += bias[gj];
            if (act_tanh) v = fast_tanh_d(v);
            v *= out_scale;
            float *dst = C + gi * ldc + gj;
            *dst = accumulate ? (*dst + v) : v;
        }
    }
}

int launch_smm(int I, int J, int K, const float *A, int64_t sa0, int64_t sa1, const float *B, int64_t sb0,
               int64_t sb1, const float *bias, float *C, int64_t ldc, int accumulate, int act_tanh, float out_scale,
               hipStream_t st) {
    if (I <= 0 || J <= 0) return 0;
    dim3 grid((J + 63) / 64, (I + 63) / 64);
    hipLaunchKernelGGL(smm_kernel, grid, dim3(256), 0, st, I, J, K, A, sa0, sa1, B, sb0, sb1, bias, C, ldc, accumulate, act_tanh, out_scale, K);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

// split-K form: slabs[s][I][J] = partial products over K chunks (sum them with launch_reduce_slabs)
int smm_splits(int K) {
    int S = K / 16;       // down to one 16-wide K step per workgroup: these products are latency-, not FLOP-bound
    if (S > 32) S = 32;
    if (S < 1) S = 1;
    return S;
}
int launch_smm_splitk(int I, int J, int K, const float *A, int64_t sa0, int64_t sa1, const float *B, int64_t sb0,
                      int64_t sb1, float *slabs, int *S_out, hipStream_t st) {
    const int S = smm_splits(K);
    *S_out = S;
    if (I <= 0 || J <= 0) return 0;
    int kchunk = (K + S - 1) / S;
    kchunk = (kchunk + 15) / 16 * 16;
    dim3 grid((J + 63) / 64, (I + 63) / 64, S);
    hipLaunchKernelGGL(smm_kernel, grid, dim3(256), 0, st, I, J, K, A, sa0, sa1, B, sb0, sb1, nullptr, slabs, (int64_t)J, 0, 0, 1.f, kchunk);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

// dst[j] += sum_i X[i*ld + j].  One 1024-thread workgroup per 16 columns: 64 row groups x 16 columns,
// every thread sums a strided subset of the rows, then the 64 partials are added in a fixed order.
__global__ __launch_bounds__(1024) void colsum_rm_kernel(const float *__restrict__ X, int rows, int cols, int64_t ld,
                                                         float *__restrict__ dst) {
    __shared__ float part[64][17];
    const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int j = blockIdx.x * 16 + c;
    float acc = 0.f;
    if (j < cols)
        for (int i = rg; i < rows; i += 64) acc += X[i * ld + j];
    part[rg][c] = acc;
    __syncthreads();
    if (threadIdx.x < 16 && j < cols) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 64; ++q) tot += part[q][threadIdx.x];
        dst[j] += tot;
    }
}
int launch_colsum_rm(const float *X, int rows, int cols, int64_t ld, float *dst, hipStream_t st) {
    if (rows <= 0 || cols <= 0) return 0;
    hipLaunchKernelGGL(colsum_rm_kernel, dim3((cols + 15) / 16), dim3(1024), 0, st, X, rows, cols, ld, dst);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

// panel-major weighted column sums, two deterministic stages
constexpr int CS_ROWS = 512;       // rows per workgroup: enough workgroups for a 50 k-candidate minibatch to fill the chip
int colsum_pm_blocks(int64_t rows) { return (int)((rows + CS_ROWS - 1) / CS_ROWS); }

__global__ __launch_bounds__(256) void colsum_pm_kernel(const float *__restrict__ X, int64_t rows, int cols,
                                                        const float *__restrict__ w, float *__restrict__ part) {
    __shared__ float red[16][16];
    const int blk = blockIdx.x, p = blockIdx.y;
    const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int64_t r0 = (int64_t)blk * CS_ROWS;
    const int64_t r1 = (r0 + CS_ROWS < rows) ? r0 + CS_ROWS : rows;
    float acc = 0.f;
    for (int64_t r = r0 + rg; r < r1; r += 16) {
        const float x = X[((int64_t)p * rows + r) * 16 + c];
        acc = w ? fmaf(w[r], x, acc) : acc + x;
    }
    red[rg][c] = acc;
    __syncthreads();
    if (threadIdx.x < 16) {
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) tot += red[q][threadIdx.x];
        part[(int64_t)blk * cols + p * 16 + threadIdx.x] = tot;
    }
}
int launch_reduce_rows_add(const float *part, int nrows, int cols, float *dst, hipStream_t st) {
    return launch_colsum_rm(part, nrows, cols, cols, dst, st);
}
// partial sums only: part[blk][cols]; the caller reduces the *nblk rows (a RedJob of the step's final reduction)
int launch_colsum_pm_part(const float *X, int64_t rows, int cols, const float *w, float *part, int *nblk_out, hipStream_t st) {
    *nblk_out = 0;
    if (rows <= 0 || cols <= 0) return 0;
    const int nblk = colsum_pm_blocks(rows);
    *nblk_out = nblk;
    hipLaunchKernelGGL(colsum_pm_kernel, dim3(nblk, cols / 16), dim3(256), 0, st, X, rows, cols, w, part);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

int launch_colsum_pm(const float *X, int64_t rows, int cols, const float *w, float *part, float *dst, hipStream_t st) {
    if (rows <= 0 || cols <= 0) return 0;
    const int nblk = colsum_pm_blocks(rows);
    hipLaunchKernelGGL(colsum_pm_kernel, dim3(nblk, cols / 16), dim3(256), 0, st, X, rows, cols, w, part);
    UPAMD_HIP(hipGetLastError());
    return launch_reduce_rows_add(part, nblk, cols, dst, st);
}

__global__ void rowdot_pm_kernel(const float *__restrict__ X, int64_t rows, int cols, const float *__restrict__ w,
                                 float *__restrict__ z) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    float acc = 0.f;
    for (int p = 0; p < cols / 16; ++p) {
        const float4 *x4 = reinterpret_cast<const float4 *>(X + ((int64_t)p * rows + r) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 x = x4[q];
            const float *ww = w + p * 16 + q * 4;
            acc = fmaf(x.x, ww[0], acc); acc = fmaf(x.y, ww[1], acc); acc = fmaf(x.z, ww[2], acc); acc = fmaf(x.w, ww[3], acc);
        }
    }
    z[r] = acc;
}
int launch_rowdot_pm(const float *X, int64_t rows, int cols, const float *w, float *z, hipStream_t st) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(rowdot_pm_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, X, rows, cols, w, z);
    UPAMD_HIP(hipGetLastError());
    return 0;
}
__global__ void rowdot_bwd_pm_kernel(const float *__restrict__ X, int64_t rows, int cols, const float *__restrict__ w,
                                     const float *__restrict__ dz, float *__restrict__ dpre) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= rows * cols) return;
    const int64_t pr = g >> 4;            // (panel, row) pair index
    const int c = (int)(g & 15);
    const int p = (int)(pr / rows);
    const int64_t r = pr % rows;
    const float x = X[g];
    dpre[g] = dz[r] * w[p * 16 + c] * (1.f - x * x);
}
int launch_rowdot_bwd_pm(const float *X, int64_t rows, int cols, const float *w, const float *dz, float *dpre,
                         hipStream_t st) {
    if (rows <= 0) return 0;
    const int64_t tot = rows * cols;
    hipLaunchKernelGGL(rowdot_bwd_pm_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, X, rows, cols, w, dz, dpre);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

__global__ void tanh_bwd_kernel(float *__restrict__ dz, const float *__restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dz[i] *= 1.f - y[i] * y[i];
}
int launch_tanh_bwd(float *dz, const float *y, int64_t n, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(tanh_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dz, y, n);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

// state_value = [h_num ; mean nodes ; mean edges ; attended current node ; stage]  (state_encoder.py:204-205)
__global__ void assemble_sv_kernel(PackedView pk, MbView mb, int D, int S_last, const float *__restrict__ Ulast,
                                   const float *__restrict__ hbarV, const float *__restrict__ hbarE,
                                   const float *__restrict__ att, float *__restrict__ SV, int ld) {
    const int W = S_last + 3 * D + 3;
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= (int64_t)mb.B * ld) return;
    const int b = (int)(g / ld), j = (int)(g % ld);
    float v;
    if (j < S_last) v = Ulast[(int64_t)b * S_last + j];
    else if (j < S_last + D) v = hbarV[(int64_t)b * D + j - S_last];
    else if (j < S_last + 2 * D) v = hbarE[(int64_t)b * D + j - S_last - D];
    else if (j < S_last + 3 * D) v = att[(int64_t)b * D + j - S_last - 2 * D];
    else if (j < W) v = (pk.meta[(int64_t)mb.idx[b] * UPAMD_META_STRIDE + 4] == (j - S_last - 3 * D)) ? 1.f : 0.f;
    else v = 0.f;                                   // row padding up to ld
    SV[g] = v;
}
int launch_assemble_sv(const PackedView &pk, const MbView &mb, int D, int S_last, const float *Ulast,
                       const float *hbarV, const float *hbarE, const float *att, float *SV, int ld, hipStream_t st) {
    const int64_t tot = (int64_t)mb.B * ld;
    hipLaunchKernelGGL(assemble_sv_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, pk, mb, D, S_last, Ulast, hbarV, hbarE, att, SV, ld);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

// W [D][2D] = [Wa | Wb]  ->  Wcat [2D][D] with rows in the P/Q panel order
// (row j': panel j'/16 even -> Wa row (j'/32)*16 + j'%16, odd -> Wb row ...), and its transpose.
__global__ void prep_wcat_kernel(const float *__restrict__ W, int D, float *__restrict__ Wcat, float *__restrict__ WcatT) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= 2 * D * D) return;
    const int jp = g / D, k = g % D;
    const int row = (jp >> 5) * 16 + (jp & 15), half = (jp >> 4) & 1;
    const float v = W[(int64_t)row * 2 * D + half * D + k];
    Wcat[g] = v;
    if (WcatT) WcatT[(int64_t)k * 2 * D + jp] = v;
}
int launch_prep_wcat(const float *W, int D, float *Wcat, float *WcatT, hipStream_t st) {
    hipLaunchKernelGGL(prep_wcat_kernel, dim3((2 * D * D + 255) / 256), dim3(256), 0, st, W, D, Wcat, WcatT);
    UPAMD_HIP(hipGetLastError());
    return 0;
}
__global__ void pad_cols_kernel(const float *__restrict__ W, int rows, int cols, int cols_pad, float *__restrict__ out) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= rows * cols_pad) return;
    const int r = g / cols_pad, c = g % cols_pad;
    out[g] = c < cols ? W[(int64_t)r * cols + c] : 0.f;
}
int launch_pad_cols(const float *W, int rows, int cols, int cols_pad, float *out, hipStream_t st) {
    hipLaunchKernelGGL(pad_cols_kernel, dim3((rows * cols_pad + 255) / 256), dim3(256), 0, st, W, rows, cols, cols_pad, out);
    UPAMD_HIP(hipGetLastError());
    return 0;
}
__global__ void transpose_kernel(const float *__restrict__ W, int rows, int cols, float *__restrict__ out) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= rows * cols) return;
    const int r = g / cols, c = g % cols;
    out[(int64_t)c * rows + r] = W[g];
}
int launch_transpose(const float *W, int rows, int cols, float *out, hipStream_t st) {
    hipLaunchKernelGGL(transpose_kernel, dim3((rows * cols + 255) / 256), dim3(256), 0, st, W, rows, cols, out);
    UPAMD_HIP(hipGetLastError());
    return 0;
}
__global__ void axpy_kernel(float *__restrict__ dst, const float *__restrict__ src, int64_t n, float alpha) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = fmaf(alpha, src[i], dst[i]);
}
__global__ void scale_kernel(float *dst, int64_t n, float alpha) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] *= alpha;
}
int launch_scale(float *dst, int64_t n, float alpha, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dst, n, alpha);
    UPAMD_HIP(hipGetLastError());
    return 0;
}
// Land-use head, first Linear W1 = [Wa | Wb | Wc | Wd] ([h0][4D]) in factorised form:
//   W1f = [Wa + Wd | Wc] ([h0][2D]),  Wbd = Wb - Wd ([h0][D])
__global__ void prep_land_head_kernel(const float *__restrict__ W1, int D, int h0, float *__restrict__ W1f,
                                      float *__restrict__ Wbd) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= h0 * D) return;
    const int k = g / D, d = g % D;
    const float *w = W1 + (int64_t)k * 4 * D;
    W1f[(int64_t)k * 2 * D + d] = w[d] + w[3 * D + d];
    W1f[(int64_t)k * 2 * D + D + d] = w[2 * D + d];
    Wbd[g] = w[D + d] - w[3 * D + d];
}
int launch_prep_land_head(const float *W1, int D, int h0, float *W1f, float *Wbd, hipStream_t st) {
    hipLaunchKernelGGL(prep_land_head_kernel, dim3((h0 * D + 255) / 256), dim3(256), 0, st, W1, D, h0, W1f, Wbd);
    UPAMD_HIP(hipGetLastError());
    return 0;
}
// ... and its gradient: gW1 += [dW1f_a + ... ] mapped back onto the four blocks
__global__ void land_head_w_scatter_kernel(const float *__restrict__ dW1f, const float *__restrict__ dWbd, int D, int h0,
                                           float *__restrict__ gW1) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= h0 * D) return;
    const int k = g / D, d = g % D;
    const float a = dW1f[(int64_t)k * 2 * D + d], cgrad = dW1f[(int64_t)k * 2 * D + D + d], bd = dWbd[g];
    float *w = gW1 + (int64_t)k * 4 * D;
    w[d] += a;                 // Wa
    w[D + d] += bd;            // Wb
    w[2 * D + d] += cgrad;     // Wc
    w[3 * D + d] += a - bd;    // Wd
}
int launch_land_head_w_scatter(const float *dW1f, const float *dWbd, int D, int h0, float *gW1, hipStream_t st) {
    hipLaunchKernelGGL(land_head_w_scatter_kernel, dim3((h0 * D + 255) / 256), dim3(256), 0, st, dW1f, dWbd, D, h0, gW1);
    UPAMD_HIP(hipGetLastError());
    return 0;
}
// dst[p*16 + c] += src[(2p)*16 + c]: the P half of a [2D] vector in P/Q panel order
__global__ void add_p_panels_kernel(float *__restrict__ dst, const float *__restrict__ src, int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < D) dst[i] += src[(2 * (i >> 4)) * 16 + (i & 15)];
}
int launch_add_p_panels(float *dst, const float *src, int D, hipStream_t st) {
    hipLaunchKernelGGL(add_p_panels_kernel, dim3((D + 255) / 256), dim3(256), 0, st, dst, src, D);
    UPAMD_HIP(hipGetLastError());
    return 0;
}
int launch_axpy(float *dst, const float *src, int64_t n, float alpha, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dst, src, n, alpha);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// PPO loss + seeds (urban_planning_agent.py:326-333,363-371; agent_pg.py:19-23).  Single
// workgroup, fixed-order tree reductions.  torch.min ties / clamp edges follow autograd:
// inside the clip range both branches are equal and the gradient is A*ratio; outside it the
// gradient flows only if the unclipped branch is the smaller one.
// ------------------------------------------------------------------------------------------
// `rows` (optional): adv / ret / old_logp / exps are whole-replay arrays and minibatch row b is replay row rows[b]
// (saves the caller four gather launches per step).  Blocks 1.. zero `zero[0..nzero)` (the gradient buffer of the step).
__global__ __launch_bounds__(1024) void ppo_loss_kernel(int B, const float *__restrict__ value,
                                                        const float *__restrict__ logp, const float *__restrict__ ent,
                                                        const int64_t *__restrict__ rows,
                                                        const float *__restrict__ adv, const float *__restrict__ ret,
                                                        const float *__restrict__ old_logp, const float *__restrict__ exps,
                                                        float clip_eps, float cv, float ce, float inv_rows, float inv_ind,
                                                        float *__restrict__ dvalue, float *__restrict__ dlogp,
                                                        float *__restrict__ dent, float *__restrict__ losses,
                                                        float *__restrict__ zero, int64_t nzero) {
    if (blockIdx.x > 0) {
        const int64_t stride = (int64_t)(gridDim.x - 1) * 1024;
        for (int64_t i = (int64_t)(blockIdx.x - 1) * 1024 + threadIdx.x; i < nzero; i += stride) zero[i] = 0.f;
        return;
    }
    __shared__ float red[3][16];
    float sv = 0.f, ss = 0.f, se = 0.f;
    const float lo = 1.f - clip_eps, hi = 1.f + clip_eps;
    for (int b = threadIdx.x; b < B; b += 1024) {
        const int64_t t = rows ? rows[b] : b;
        const float diff = value[b] - ret[t];
        sv = fmaf(diff, diff, sv);
        dvalue[b] = cv * 2.f * diff * inv_rows;
        float gl = 0.f, ge = 0.f;
        if (exps[t] != 0.f) {
            const float ratio = expf(logp[b] - old_logp[t]);
            const float A = adv[t];
            const float s1 = ratio * A;
            const float s2 = fminf(fmaxf(ratio, lo), hi) * A;
            ss += fminf(s1, s2);
            se += ent[b];
            const bool inside = ratio >= lo && ratio <= hi;
            const float dsdr = inside ? A : (s1 < s2 ? A : 0.f);
            gl = -dsdr * ratio * inv_ind;
            ge = -ce * inv_ind;
        }
        dlogp[b] = gl;
        dent[b] = ge;
    }
    for (int off = 32; off > 0; off >>= 1) {
        sv += __shfl_xor(sv, off);
        ss += __shfl_xor(ss, off);
        se += __shfl_xor(se, off);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = sv; red[1][w] = ss; red[2][w] = se; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float tv = 0.f, ts = 0.f, te = 0.f;
        for (int q = 0; q < 16; ++q) { tv += red[0][q]; ts += red[1][q]; te += red[2][q]; }
        const float vl = tv * inv_rows, sl = -ts * inv_ind, el = -te * inv_ind;
        losses[0] = sl + cv * vl + ce * el;
        losses[1] = vl;
        losses[2] = sl;
        losses[3] = el;
    }
}
int launch_ppo_loss(int B, const float *value, const float *logp, const float *ent, const int64_t *rows, const float *adv,
                    const float *ret, const float *old_logp, const float *exps, float clip_eps, float cv, float ce,
                    float inv_rows, float inv_ind, float *dvalue, float *dlogp, float *dent, float *losses,
                    float *zero, int64_t nzero, hipStream_t st) {
    const unsigned zb = (zero && nzero > 0) ? (unsigned)std::min<int64_t>((nzero + 4095) / 4096, 512) : 0u;
    hipLaunchKernelGGL(ppo_loss_kernel, dim3(1 + zb), dim3(1024), 0, st, B, value, logp, ent, rows, adv, ret, old_logp, exps, clip_eps, cv, ce, inv_rows, inv_ind, dvalue, dlogp, dent, losses, zero, nzero);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// GAE (khrylib/rl/core/common.py:12-21).  Episodes are independent once masks[t] == 0 (all
// carried terms are multiplied by the mask), so every thread that sits on an episode end walks
// its episode backwards with the reference's exact operation order (no FMA contraction).
// ------------------------------------------------------------------------------------------
__global__ void gae_kernel(int64_t T, const float *__restrict__ rewards, const float *__restrict__ masks,
                           const float *__restrict__ values, float gamma, float gt, float *__restrict__ adv,
                           float *__restrict__ ret) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    // segment ends: the last row, or a row whose successor-coupling is cut (masks[t] == 0)
    if (!(t == T - 1 || masks[t] == 0.f)) return;
    float prev_value = 0.f, prev_adv = 0.f;
    for (int64_t i = t; i >= 0; --i) {
        if (i != t && masks[i] == 0.f) break;
        const float m = masks[i], v = values[i];
        const float delta = __fsub_rn(__fadd_rn(rewards[i], __fmul_rn(__fmul_rn(gamma, prev_value), m)), v);
        const float a = __fadd_rn(delta, __fmul_rn(__fmul_rn(gt, prev_adv), m));
        adv[i] = a;
        ret[i] = __fadd_rn(v, a);
        prev_value = v;
        prev_adv = a;
    }
}
int launch_gae(int64_t T, const float *rewards, const float *masks, const float *values, double gamma, double tau,
               float *adv, float *ret, hipStream_t st) {
    if (T <= 0) return 0;
    // python evaluates gamma * tau in double before it meets the float32 tensor (common.py:16)
    hipLaunchKernelGGL(gae_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, st, T, rewards, masks, values, (float)gamma, (float)(gamma * tau), adv, ret);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// torch.optim.Adam single-tensor semantics (coupled weight decay, bias corrections, eps outside sqrt)
// ------------------------------------------------------------------------------------------
__global__ void adam_kernel(int64_t n, float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                            float *__restrict__ v, float step_size, float b1, float b2, float one_m_b1, float one_m_b2,
                            float eps, float wd, float bc2_sqrt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float grad = g[i];
    const float param = p[i];
    if (wd != 0.f) grad = fmaf(wd, param, grad);
    const float mi = m[i] + (grad - m[i]) * one_m_b1;             // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = v[i] * b2 + one_m_b2 * grad * grad;          // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = param - step_size * (mi / denom);                      // param.addcdiv_(exp_avg, denom, -lr / bc1)
}
int launch_adam(int64_t n, float *p, const float *g, float *m, float *v, int step, double lr, double b1, double b2,
                double eps, double wd, hipStream_t st) {
    if (n <= 0) return 0;
    const double bc1 = 1.0 - pow(b1, step), bc2 = 1.0 - pow(b2, step);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, p, g, m, v, (float)(lr / bc1), (float)b1,
                       (float)b2, (float)(1.0 - b1), (float)(1.0 - b2), (float)eps, (float)wd, (float)sqrt(bc2));
    UPAMD_HIP(hipGetLastError());
    return 0;
}

// All optimizer groups of a step in ONE launch: group q covers [begin, end) with its own bias corrections (its own
// step count); a group with step == 0 is skipped (a head without rows: grad None in the reference).  Optionally the
// four loss scalars behind the gradients are copied out (the caller's per-step log row).
struct AdamGroups {
    int64_t begin[4], end[4];
    float step_size[4], bc2_sqrt[4];
    int n;
};
__global__ void adam_groups_kernel(AdamGroups G, float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                   float *__restrict__ v, float b1, float b2, float one_m_b1, float one_m_b2, float eps,
                                   float wd, const float *__restrict__ loss_src, float *__restrict__ loss_dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (loss_dst && i < 4) loss_dst[i] = loss_src[i];
    int q = -1;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < G.n && i >= G.begin[k] && i < G.end[k]) q = k;
    if (q < 0) return;
    float grad = g[i];
    const float param = p[i];
    if (wd != 0.f) grad = fmaf(wd, param, grad);
    const float mi = m[i] + (grad - m[i]) * one_m_b1;
    const float vi = v[i] * b2 + one_m_b2 * grad * grad;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / G.bc2_sqrt[q] + eps;
    p[i] = param - G.step_size[q] * (mi / denom);
}
int launch_adam_groups(int n_groups, const int64_t *begin, const int64_t *end, const int32_t *step, float *p, const float *g,
                       float *m, float *v, double lr, double b1, double b2, double eps, double wd, const float *loss_src,
                       float *loss_dst, hipStream_t st) {
    AdamGroups G;
    G.n = 0;
    int64_t hi = loss_dst ? 4 : 0;
    for (int k = 0; k < n_groups; ++k) {
        if (step[k] <= 0 || end[k] <= begin[k]) continue;
        const double bc1 = 1.0 - pow(b1, step[k]), bc2 = 1.0 - pow(b2, step[k]);
        G.begin[G.n] = begin[k];
        G.end[G.n] = end[k];
        G.step_size[G.n] = (float)(lr / bc1);
        G.bc2_sqrt[G.n] = (float)sqrt(bc2);
        hi = std::max(hi, end[k]);
        ++G.n;
    }
    if (hi <= 0) return 0;
    hipLaunchKernelGGL(adam_groups_kernel, dim3((unsigned)((hi + 255) / 256)), dim3(256), 0, st, G, p, g, m, v, (float)b1,
                       (float)b2, (float)(1.0 - b1), (float)(1.0 - b2), (float)eps, (float)wd, loss_src, loss_dst);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

// sum of squares: per-block partials into scratch, then one block adds them (fixed order) into *out_accum
__global__ __launch_bounds__(256) void sumsq_part_kernel(const float *__restrict__ x, int64_t n, float *__restrict__ scratch) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc = fmaf(x[i], x[i], acc);
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) scratch[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void sumsq_final_kernel(const float *__restrict__ scratch, int nblk, float *__restrict__ out_accum) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float acc = 0.f;
        for (int i = 0; i < nblk; ++i) acc += scratch[i];
        *out_accum += acc;
    }
}
int launch_sumsq(const float *x, int64_t n, float *scratch, float *out_accum, hipStream_t st) {
    if (n <= 0) return 0;
    int nblk = (int)((n + 255) / 256);
    if (nblk > 1024) nblk = 1024;
    hipLaunchKernelGGL(sumsq_part_kernel, dim3(nblk), dim3(256), 0, st, x, n, scratch);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(64), 0, st, scratch, nblk, out_accum);
    UPAMD_HIP(hipGetLastError());
    return 0;
}
// torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to <= 1
__global__ void clip_scale_kernel(float *__restrict__ g, int64_t n, const float *__restrict__ sumsq, float max_norm) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float coef = fminf(max_norm / (sqrtf(*sumsq) + 1e-6f), 1.f);
    g[i] *= coef;
}
int launch_clip_scale(float *g, int64_t n, const float *sumsq, float max_norm, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(clip_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, n, sumsq, max_norm);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

}  // namespace upamd
