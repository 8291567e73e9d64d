// fp32 GEMM on the bf16 matrix pipe by exact operand splitting (gfx950, wave64) -- OPT-IN, not on the default path.
//
//   C[M,N] = alpha * act(A[M,K] * W[N,K]^T + bias + R)       same contract as gemm_nt (panel-major A / C / R)
//
// Every fp32 operand is written as the sum of three bf16 numbers, x = x1 + x2 + x3 (round-to-nearest at each step, the
// remainders are exact in fp32: |x - x1| <= 2^-9 |x|, |x - x1 - x2| <= 2^-18 |x|, what is left after x3 <= 2^-26 |x|),
// and the product a*w is accumulated in fp32 from the six largest of the nine partial products
//   a1 w1 + (a1 w2 + a2 w1) + (a1 w3 + a2 w2 + a3 w1)        [+ a2 w3 + a3 w2 + a3 w3 with NPROD = 9]
// each of which is EXACT inside the matrix core (8-bit x 8-bit significands).  The dropped terms are <= 2^-26 |a w|
// relative, i.e. below the 2^-24 rounding of a single fp32 product: the result has fp32 GEMM accuracy (tests compare
// both kernels with a float64 product) while v_mfma_f32_32x32x16_bf16 runs at 16x the rate of v_mfma_f32_32x32x2_f32
// -- 6 instructions replace 16, a 2.67x higher ceiling for the same arithmetic.
//
// The weights are split once per call by split_w_kernel into three bf16 planes [3][N][K]; the activations are split on
// the way from HBM to LDS (each element once per workgroup: 4.5 vector-ALU ops), so the main loop reads ready-made
// bf16 fragments with ds_read_b128 from a swizzled, conflict-free LDS image (MI355X_MICROARCH.md, ds_read_b128 lane groups).
//
// Stands in for the same nn.Linear products as gemm.hip (urban_planning/models/state_encoder.py:59-82,110-130).
#include <type_traits>

#include "kernels.h"

namespace upamd {

typedef float f32x16s __attribute__((ext_vector_type(16)));
typedef float f32x2s __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2s __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float fast_tanh_s(float x) {
    float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// (x, y) -> three packed bf16 pairs with x = h + m + l (to 2^-26 relative)
__device__ __forceinline__ void split2(float x, float y, uint32_t &h, uint32_t &m, uint32_t &l) {
    f32x2s v = {x, y};
    const bf16x2s hb = __builtin_convertvector(v, bf16x2s);
    v -= __builtin_convertvector(hb, f32x2s);
    const bf16x2s mb = __builtin_convertvector(v, bf16x2s);
    v -= __builtin_convertvector(mb, f32x2s);
    const bf16x2s lb = __builtin_convertvector(v, bf16x2s);
    h = __builtin_bit_cast(uint32_t, hb);
    m = __builtin_bit_cast(uint32_t, mb);
    l = __builtin_bit_cast(uint32_t, lb);
}

// W fp32 [N][ldw] -> planes bf16 [3][N][K]
__global__ void split_w_kernel(const float *__restrict__ W, int N, int K, int64_t ldw, uint32_t *__restrict__ planes) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;          // one pair of k per thread
    const int64_t pairs = (int64_t)N * (K >> 1);
    if (g >= pairs) return;
    const int n = (int)(g / (K >> 1)), k = (int)(g % (K >> 1)) * 2;
    const float2 w = *reinterpret_cast<const float2 *>(W + (int64_t)n * ldw + k);
    uint32_t h, m, l;
    split2(w.x, w.y, h, m, l);
    planes[g] = h;
    planes[pairs + g] = m;
    planes[2 * pairs + g] = l;
}

template <int NPROD>
__global__ __launch_bounds__(256, 3) void gemm_nt_split_kernel(const float *__restrict__ A, int64_t M, int K,
                                                               const uint16_t *__restrict__ Wp, int N,
                                                               const float *__restrict__ bias,
                                                               const float *__restrict__ R, float *__restrict__ C,
                                                               int act_tanh, float alpha, int MT, int NT) {
    // 128 x 128 tile, 4 waves (64 x 64 each), K in chunks of 16 through two LDS stages: while the matrix cores work on
    // chunk k the same wave splits chunk k+1 into the other stage (vector ALU and LDS writes between the MFMAs)
    constexpr int BM = 128, BN = 128;
    constexpr int PLANE = BM * 32;                       // bytes of one plane of one operand per stage: 128 rows x 16 bf16
    constexpr int STAGE = 6 * PLANE;                     // A planes 0..2, W planes 0..2
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

    const int id = blockIdx.x;                           // XCD-aware tile order, see gemm_nt_dma2_kernel
    const int xcd = id & 7, slot = id >> 3;
    const int mt = (slot / NT) * 8 + xcd, nt = slot % NT;
    if (mt >= MT) return;
    const int64_t m0 = (int64_t)mt * BM;
    const int n0 = nt * BN;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int l31 = lane & 31, lhi = lane >> 5;

    f32x16s acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // staging: thread (row, half) moves 8 consecutive k of one row of each operand per chunk
    const int srow = tid >> 1, half = tid & 1;
    int64_t gm = m0 + srow;
    if (gm >= M) gm = M - 1;                             // rows past the end: valid address, result never stored
    const float4 *asrc = reinterpret_cast<const float4 *>(A + gm * 16 + half * 8);
    const int64_t astep = M * 4;                         // float4s per chunk (one 16-column panel)
    const int64_t wplane = (int64_t)N * K;               // bf16 elements per plane
    const uint16_t *wsrc = Wp + (int64_t)(n0 + srow) * K + half * 8;
    // LDS image of a plane: [row][2 slots of 8 bf16], slot' = slot ^ ((row >> 3) & 1): the 16 lanes of a ds_read_b128
    // group (rows {0-3,12-15,20-27} / {4-11,16-19,28-31} of a 32-row tile, one slot) then hit 16 distinct 16-byte columns
    const int sd = srow * 32 + ((half ^ ((srow >> 3) & 1)) << 4);

    // (named registers on purpose: small arrays end up in scratch memory)
    float4 ra0, ra1;
    uint4 rw0, rw1, rw2;
    auto gload = [&](int kc) {
        const float4 *a = asrc + kc * astep;
        ra0 = a[0]; ra1 = a[1];
        rw0 = *reinterpret_cast<const uint4 *>(wsrc + kc * 16);
        rw1 = *reinterpret_cast<const uint4 *>(wsrc + wplane + kc * 16);
        rw2 = *reinterpret_cast<const uint4 *>(wsrc + 2 * wplane + kc * 16);
    };
    auto commit = [&](int st) {
        unsigned char *d = smem + st * STAGE + sd;
        uint32_t h0, h1, h2, h3, m0_, m1, m2, m3, l0, l1, l2, l3;
        split2(ra0.x, ra0.y, h0, m0_, l0);
        split2(ra0.z, ra0.w, h1, m1, l1);
        split2(ra1.x, ra1.y, h2, m2, l2);
        split2(ra1.z, ra1.w, h3, m3, l3);
        *reinterpret_cast<uint4 *>(d + 0 * PLANE) = make_uint4(h0, h1, h2, h3);
        *reinterpret_cast<uint4 *>(d + 1 * PLANE) = make_uint4(m0_, m1, m2, m3);
        *reinterpret_cast<uint4 *>(d + 2 * PLANE) = make_uint4(l0, l1, l2, l3);
        *reinterpret_cast<uint4 *>(d + 3 * PLANE) = rw0;
        *reinterpret_cast<uint4 *>(d + 4 * PLANE) = rw1;
        *reinterpret_cast<uint4 *>(d + 5 * PLANE) = rw2;
    };

    // fragment byte offsets inside a plane: row-tile i; this lane half takes slot lhi (k = 8 lhi .. 8 lhi + 7)
    int aoff[2], boff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ra_ = wr * 64 + i * 32 + l31, rb_ = wc * 64 + i * 32 + l31;
        aoff[i] = ra_ * 32 + ((lhi ^ ((ra_ >> 3) & 1)) << 4);
        boff[i] = 3 * PLANE + rb_ * 32 + ((lhi ^ ((rb_ >> 3) & 1)) << 4);
    }

    const int KC = K >> 4;
    // one trip: fragments of chunk kc, its matrix chain, and -- in the same basic block, spread between the MFMAs by the
    // scheduling hints below -- the split of chunk kc+1 into the other stage and the fetch of chunk kc+2
    auto trip = [&](int kc, auto more) {
        constexpr bool MORE = decltype(more)::value;
        const unsigned char *st = smem + (kc & 1) * STAGE;
        bf16x8s a[2][3], b[2][3];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i][p] = *reinterpret_cast<const bf16x8s *>(st + p * PLANE + aoff[i]);
                b[i][p] = *reinterpret_cast<const bf16x8s *>(st + p * PLANE + boff[i]);
            }
        // smallest partial products first; consecutive MFMAs go to different accumulators (no back-to-back dependence);
        // operands swapped as in gemm_nt (a lane owns 4 consecutive columns).  The staging of the next chunk is placed by
        // hand behind the later product groups (sched_barrier fences keep it there): each group of 4 MFMAs is 128 cycles
        // of matrix pipe, enough issue slots for ~18 vector instructions, and the data fetched one trip ago has had the
        // first half of this trip on top to arrive.
        constexpr int PA[9] = {2, 1, 2, 0, 1, 2, 0, 1, 0}, PB[9] = {2, 2, 1, 2, 1, 0, 1, 0, 0};      // (a plane, w plane)
        constexpr int T0 = NPROD == 9 ? 0 : 3;
        unsigned char *d = smem + ((kc + 1) & 1) * STAGE + sd;
        uint32_t h0 = 0, h1 = 0, h2 = 0, h3 = 0, m0_ = 0, m1 = 0, m2 = 0, m3 = 0, l0 = 0, l1 = 0, l2 = 0, l3 = 0;
#pragma unroll
        for (int t = T0; t < 9; ++t) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j][PB[t]], a[i][PA[t]], acc[i][j], 0, 0, 0);
            if (MORE && t >= 6) {
                __builtin_amdgcn_sched_barrier(0);
                if (t == 6) {
                    split2(ra0.x, ra0.y, h0, m0_, l0);
                    split2(ra0.z, ra0.w, h1, m1, l1);
                } else if (t == 7) {
                    split2(ra1.x, ra1.y, h2, m2, l2);
                    split2(ra1.z, ra1.w, h3, m3, l3);
                } else {
                    *reinterpret_cast<uint4 *>(d + 0 * PLANE) = make_uint4(h0, h1, h2, h3);
                    *reinterpret_cast<uint4 *>(d + 1 * PLANE) = make_uint4(m0_, m1, m2, m3);
                    *reinterpret_cast<uint4 *>(d + 2 * PLANE) = make_uint4(l0, l1, l2, l3);
                    *reinterpret_cast<uint4 *>(d + 3 * PLANE) = rw0;
                    *reinterpret_cast<uint4 *>(d + 4 * PLANE) = rw1;
                    *reinterpret_cast<uint4 *>(d + 5 * PLANE) = rw2;
                    gload(kc + 2 < KC ? kc + 2 : KC - 1);        // (the last fetch is a harmless repeat: no branch in the trip)
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();                                 // stage (kc+1)&1 complete; every read of stage kc&1 done
    };
    gload(0);
    commit(0);
    gload(KC > 1 ? 1 : 0);
    __syncthreads();
    for (int kc = 0; kc + 1 < KC; ++kc) trip(kc, std::true_type{});
    trip(KC - 1, std::false_type{});

#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t gmo = m0 + wr * 64 + i * 32 + l31;
        if (gmo >= M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gn = n0 + wc * 64 + j * 32 + 8 * q + 4 * lhi;      // 4 consecutive columns gn..gn+3
                const int64_t o = ((int64_t)(gn >> 4) * M + gmo) * 16 + (gn & 15);
                float v[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = acc[i][j][4 * q + t] + (bias ? bias[gn + t] : 0.f);
                if (R) {
                    const float4 r4 = *reinterpret_cast<const float4 *>(R + o);
                    v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
                }
                if (act_tanh) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = fast_tanh_s(v[t]);
                }
                *reinterpret_cast<float4 *>(C + o) = make_float4(v[0] * alpha, v[1] * alpha, v[2] * alpha, v[3] * alpha);
            }
    }
}

bool gemm_nt_split_ok(const GemmNT &g) {
    return !g.a_rm && !g.c_rm && !g.w_kn && g.N % 128 == 0 && g.K % 16 == 0 && g.K >= 16 && g.ldw % 2 == 0 &&
           reinterpret_cast<uintptr_t>(g.A) % 16 == 0 && reinterpret_cast<uintptr_t>(g.W) % 8 == 0;
}

int64_t gemm_nt_split_scratch_bytes(int N, int K) { return 3LL * N * K * 2; }

// planes: device scratch of gemm_nt_split_scratch_bytes(N, K) bytes (16-byte aligned)
int launch_gemm_nt_split(const GemmNT &g, void *planes, int nprod, hipStream_t st, Profiler *prof) {
    if (!gemm_nt_split_ok(g)) return fail(UPAMD_E_INVALID, "gemm_nt_split: panel-major operands, N %% 128 == 0, K %% 16 == 0");
    if (nprod != 6 && nprod != 9) return fail(UPAMD_E_INVALID, "gemm_nt_split: 6 or 9 partial products");
    if (!planes || reinterpret_cast<uintptr_t>(planes) % 16) return fail(UPAMD_E_INVALID, "gemm_nt_split: scratch missing or misaligned");
    const int64_t pairs = (int64_t)g.N * (g.K >> 1);
    const int began = prof_begin(prof, "gemm_nt_split", st, 2.0 * g.M * g.K * g.N, 0.0);
    hipLaunchKernelGGL(split_w_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, g.W, g.N, g.K, g.ldw,
                       static_cast<uint32_t *>(planes));
    const int MT = (int)((g.M + 127) / 128), MT8 = (MT + 7) / 8 * 8, NT = g.N / 128;
    if (nprod == 6)
        hipLaunchKernelGGL(gemm_nt_split_kernel<6>, dim3(MT8 * NT), dim3(256), 0, st, g.A, g.M, g.K, static_cast<const uint16_t *>(planes),
                           g.N, g.bias, g.R, g.C, g.act_tanh, g.alpha, MT, NT);
    else
        hipLaunchKernelGGL(gemm_nt_split_kernel<9>, dim3(MT8 * NT), dim3(256), 0, st, g.A, g.M, g.K, static_cast<const uint16_t *>(planes),
                           g.N, g.bias, g.R, g.C, g.act_tanh, g.alpha, MT, NT);
    UPAMD_HIP(hipGetLastError());
    prof_end(prof, "gemm_nt_split", st, began);
    return 0;
}

}  // namespace upamd
