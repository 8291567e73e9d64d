// fp32 MFMA GEMMs for the dense per-node / per-candidate / per-sample MLP layers (gfx950, wave64).
//
// Both kernels use v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak):
// operand A: lane l holds A[i = l & 31][k = l >> 5]; operand B: lane l holds B[k = l >> 5][j = l & 31];
// accumulator reg r of lane l is D[row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)][col = l & 31].
//
//   gemm_nt : C[M,N] = alpha * act(A[M,K] * W[N,K]^T + bias + R)        M = nodes (~10^5..10^6) or rows of a
//             minibatch, K,N <= ~1024.  Forward P/Q projection, node encoder, pointer-head hidden layers, every
//             "dgrad", and the per-sample [B, D] linear layers (row-major operands).
//   gemm_tn : slabs[s][I][J] = A[rows_s, I]^T * B[rows_s, J]             reduction over rows, split-K.
//             Every weight gradient; the slabs are summed by reduce_slabs in a fixed order (deterministic,
//             no float atomics).
// A / C (and R) are panel-major [cols/16][rows][16] for per-node tensors, or row-major for per-sample tensors.
//
// Replaces the reference's nn.Linear calls on padded [B,E,2D] / [B,N,D] tensors
// (urban_planning/models/state_encoder.py:19,59-82,110-130; policy.py:19-43) and their autograd.
#include <map>
#include <mutex>
#include <utility>

#include "kernels.h"

namespace upamd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float fast_tanh(float x) {
    // tanh(x) = 1 - 2 / (exp(2x) + 1); v_exp_f32 + v_rcp_f32, abs error ~1e-7, saturates cleanly
    float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// ------------------------------------------------------------------------------------------ NT
template <int BN, int WM, int WN, bool A_RM, bool C_RM, int PK, int KC = 0>      // KC != 0: K known at compile time
__global__ __launch_bounds__(256, 4) void gemm_nt_mfma_kernel(const float *__restrict__ A, int64_t M, int K, int64_t lda,
                                                           const float *__restrict__ W, int N, int64_t ldw,
                                                           const float *__restrict__ bias,
                                                           const float *__restrict__ R, float *__restrict__ C,
                                                           int64_t ldc, int act_tanh, float alpha, int MT, int NT,
                                                           int w_kn) {
    constexpr int BM = 128, LD = 17;
    constexpr int WAVES_N = BN / WN;
    constexpr int TI = WM / 32, TJ = WN / 32;
    constexpr int NB = (BN * 4 + 255) / 256;
    static_assert((BM / WM) * WAVES_N == 4, "4 waves per workgroup");
    // PK = K panels (of 16) per pipeline stage / barrier (2 measured slower than 1: 106 vs 113 TFLOP/s)
    __shared__ float As[2][PK][BM * LD];
    __shared__ float Bs[2][PK][BN * LD];

    // XCD-aware mapping: workgroup id -> XCD id % 8 (observed dispatch); all N-tiles of an M-tile
    // land on one XCD so the A panel is fetched from HBM once and re-read from that XCD's L2.
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int mt = (slot / NT) * 8 + xcd, nt = slot % NT;
    if (mt >= MT) return;
    const int64_t m0 = (int64_t)mt * BM;
    const int n0 = nt * BN;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w / WAVES_N, wc = w % WAVES_N;
    const int l31 = lane & 31, lhi = lane >> 5;

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int KS = KC ? (KC >> 4) / PK : (K >> 4) / PK;      // pipeline stages
    float4 ra[PK][2], rb[PK][NB];
    auto load_tiles = [&](int ks) {
#pragma unroll
        for (int pp = 0; pp < PK; ++pp) {
            const int kp = ks * PK + pp;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = tid + 256 * q, row = e >> 2, c4 = (e & 3) * 4;
                const int64_t gm = m0 + row;
                const float *src = A_RM ? A + gm * lda + kp * 16 + c4 : A + ((int64_t)kp * M + gm) * 16 + c4;
                ra[pp][q] = gm < M ? *reinterpret_cast<const float4 *>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int e = tid + 256 * q, row = e >> 2, c4 = (e & 3) * 4;
                if (e < BN * 4) {
                    // [N][K] weights: 4 consecutive k of row n;  [K][N] weights: 4 consecutive n of row k
                    const float *src = w_kn ? W + (int64_t)(kp * 16 + e / (BN / 4)) * ldw + n0 + (e % (BN / 4)) * 4
                                            : W + (int64_t)(n0 + row) * ldw + kp * 16 + c4;
                    rb[pp][q] = *reinterpret_cast<const float4 *>(src);
                }
            }
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int pp = 0; pp < PK; ++pp) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = tid + 256 * q, row = e >> 2, c4 = (e & 3) * 4;
                float *d = &As[buf][pp][row * LD + c4];
                d[0] = ra[pp][q].x; d[1] = ra[pp][q].y; d[2] = ra[pp][q].z; d[3] = ra[pp][q].w;
            }
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                const int e = tid + 256 * q, row = e >> 2, c4 = (e & 3) * 4;
                if (e < BN * 4) {
                    if (w_kn) {
                        float *d = &Bs[buf][pp][(e % (BN / 4)) * 4 * LD + e / (BN / 4)];
                        d[0] = rb[pp][q].x; d[LD] = rb[pp][q].y; d[2 * LD] = rb[pp][q].z; d[3 * LD] = rb[pp][q].w;
                    } else {
                        float *d = &Bs[buf][pp][row * LD + c4];
                        d[0] = rb[pp][q].x; d[1] = rb[pp][q].y; d[2] = rb[pp][q].z; d[3] = rb[pp][q].w;
                    }
                }
            }
        }
    };

    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int ks = 0; ks < KS; ++ks) {
        const int buf = ks & 1;
        if (ks + 1 < KS) load_tiles(ks + 1);
#pragma unroll
        for (int pp = 0; pp < PK; ++pp) {
            const float *as = &As[buf][pp][(wr * WM + l31) * LD + lhi];
            const float *bs = &Bs[buf][pp][(wc * WN + l31) * LD + lhi];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                float a[TI], b[TJ];
#pragma unroll
                for (int i = 0; i < TI; ++i) a[i] = as[i * 32 * LD + 2 * s];
#pragma unroll
                for (int j = 0; j < TJ; ++j) b[j] = bs[j * 32 * LD + 2 * s];
#pragma unroll
                for (int i = 0; i < TI; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        // operands swapped on purpose: D[row = n][col = m], so a lane's 4 consecutive accumulator
                        // registers are 4 consecutive columns of one output row -> 16-byte stores in the epilogue
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[j], a[i], acc[i][j], 0, 0, 0);
            }
        }
        if (ks + 1 < KS) store_tiles(buf ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int64_t gm = m0 + wr * WM + i * 32 + l31;
        if (gm >= M) continue;
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gn = n0 + wc * WN + j * 32 + 8 * q + 4 * lhi;      // 4 consecutive columns gn..gn+3
                const int64_t o = C_RM ? gm * ldc + gn : ((int64_t)(gn >> 4) * M + gm) * 16 + (gn & 15);
                float v[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = acc[i][j][4 * q + t] + (bias ? bias[gn + t] : 0.f);
                if (R) {
                    if (C_RM) {
#pragma unroll
                        for (int t = 0; t < 4; ++t) v[t] += R[o + t];
                    } else {
                        const float4 r4 = *reinterpret_cast<const float4 *>(R + o);
                        v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
                    }
                }
                if (act_tanh) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = fast_tanh(v[t]);
                }
                if (C_RM) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) C[o + t] = v[t] * alpha;
                } else {
                    *reinterpret_cast<float4 *>(C + o) = make_float4(v[0] * alpha, v[1] * alpha, v[2] * alpha, v[3] * alpha);
                }
            }
    }
}

// ------------------------------------------------------------------------------------------ NT, LDS-DMA staging
// Same tile decomposition (128 x 128 per workgroup, 64 x 64 per wave, 32x32x2 fp32 MFMA) with the operands staged by
// the LDS-DMA path (global_load_lds_dwordx4: 1 KiB per wave instruction, no staging VGPRs, no ds_write pass) and read
// back with ds_read_b128:
//   * LDS image of one 16-wide K panel of an operand = 128 rows x 4 chunks of 16 B, slot (row, c') holds the row's
//     logical chunk c = c' ^ ((row >> 2) & 3).  The DMA destination is lane-linear, so the XOR swizzle is applied to the
//     per-lane SOURCE address (still whole 64-byte row segments per 4 lanes: coalescing is unchanged); a 16-lane
//     ds_read_b128 group then covers all 64 banks exactly once (rows r..r+3 x 4 row-quads) -- conflict-free.
//   * a lane reads chunk c = 2g + (lane >> 5) of its row: k = 8g + 4 (lane >> 5) + t, t = 0..3.  MFMA step (g, t)
//     therefore contracts k = 8g + t (lanes 0-31) and k = 8g + 4 + t (lanes 32-63): any assignment of k to steps is
//     valid as long as A and B use the same one.  One 16-byte read feeds four MFMA steps of a fragment.
//   * two LDS stages of one 16-wide K chunk each, one raw s_barrier per chunk; chunk ks + 1 is issued behind the fragment
//     reads of chunk ks (hipcc drains all LDS-DMA before a ds_read that follows one, so deeper rings buy nothing from
//     HIP source) and lands while chunk ks's 32 MFMAs run.  Software-pipelined variants (fragment reads between the MFMA
//     halves, inline-asm reads with counted lgkmcnt, s_setprio, 32- and 64-wide chunks) measured no better and were removed.
// Only the panel-major A / C, row-major [N][K] weight form (the dominant launches); everything else keeps the
// register-staged kernel above.

// Every workgroup of a launch starts at the same instant and the tiles cost the same, so without help the co-resident
// workgroups of a CU (and of the whole chip) run in lockstep: all of them stream their prologue, then all compute, then
// all write their C tiles at once -- the HBM-bound epilogue (N = 512: 1.2 GB per launch) is then NOT overlapped with
// anyone's MFMAs (measured: main loop ~150 TFLOP/s, whole launch ~120).  Delaying the workgroups of the FIRST
// residency round by (phase / 4) of a tile time de-synchronises them for the rest of the launch: later workgroups start
// whenever a slot frees up, i.e. already staggered.
//   mode 1: phase = (block >> 8) & 3 (round-robin placement over 256 CUs), 2: (block >> 3) & 3 (consecutive blocks of an XCD
//   fill a CU), 3: the wave's hardware slot id (co-resident waves of a SIMD have distinct slots).
__device__ __forceinline__ void first_wave_stagger(int mode, int cycles) {
    if (mode == 0) return;
    int phase;
    if (mode == 3) {
        phase = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 3;      // HW_REG_HW_ID.WAVE_ID[1:0]
    } else {
        if (blockIdx.x >= 1024) return;
        phase = (mode == 1 ? (blockIdx.x >> 8) : (blockIdx.x >> 3)) & 3;
    }
    if (phase == 0) return;
    const long long t0 = __builtin_readcyclecounter();
    const long long wait = (long long)phase * cycles;
    while (__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(32);
}

// Workgroup tile (32 TM WM) x (32 TN WN), one (32 TM) x (32 TN) output block per wave.  128 x 128 with 64 x 64 per wave
// (4 waves) is the measured optimum of the one-block-per-wave forms: 256 x 128 / 128 x 256 with EIGHT waves are 3-5 % slower,
// 256 x 256 (16 waves on one barrier) 14 % (profiles/archive/r02_gemm_lab.md).  TM / TN > 2 give a wave a bigger block instead (fewer
// DMA and fragment-read instructions per MFMA at two waves per SIMD); PRIO raises the wave's priority over its MFMA burst.
template <int WM, int WN, int TM = 2, int TN = 2, bool PRIO = false>
__global__ __launch_bounds__(64 * WM * WN, (TM * TN > 4 ? 2 : 1)) void gemm_nt_dma2_kernel(const float *__restrict__ A, int64_t M, int K,
                                                                    const float *__restrict__ W, int N, int64_t ldw,
                                                                    const float *__restrict__ bias,
                                                                    const float *__restrict__ R, float *__restrict__ C,
                                                                    int act_tanh, float alpha, int MT, int NT, int stagger_mode,
                                                                    int stagger_cycles, uint8_t *__restrict__ exp_flags) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, NW = WM * WN;
    constexpr int APANEL = BM * 16, STAGE = (BM + BN) * 16;      // floats: A panel then B panel
    constexpr int NI = (BM + BN) / 16;                           // 1 KiB LDS-DMA instructions per chunk
    constexpr int PER = NI / NW;                                 // ... per wave
    static_assert(NI % NW == 0, "the DMA instructions of a chunk must divide evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) float smem[];      // the ONLY LDS object of this kernel

    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int mt = (slot / NT) * 8 + xcd, nt = slot % NT;
    if (mt >= MT) return;
    first_wave_stagger(stagger_mode, stagger_cycles);
    const int64_t m0 = (int64_t)mt * BM;
    const int n0 = nt * BN;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w / WN, wc = w % WN;
    const int l31 = lane & 31, lhi = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // DMA instruction t of a chunk fills 16 rows (64 slots of 16 B, slot = 4 row + c') of the A (t < BM/16) or B panel
    const float *src[PER];
    int64_t step[PER];
    int dst[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int t = w * PER + j;
        const bool isA = t < BM / 16;
        const int row = (isA ? t : t - BM / 16) * 16 + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
        if (isA) {
            int64_t gm = m0 + row;
            if (gm >= M) gm = M - 1;                // rows past the end: valid address, result never stored
            src[j] = A + gm * 16 + c * 4;
            step[j] = M * 16;
            dst[j] = t * 256;
        } else {
            src[j] = W + (int64_t)(n0 + row) * ldw + c * 4;
            step[j] = 16;
            dst[j] = APANEL + (t - BM / 16) * 256;
        }
    }
    auto issue = [&](int kc, int s) {
#pragma unroll
        for (int j = 0; j < PER; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(src[j] + kc * step[j]), (lptr_t)(smem + s * STAGE + dst[j]), 16, 0, 0);
    };
    int aoff[TM][2], boff[TN][2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const int sw = ((2 * g + lhi) ^ ((l31 >> 2) & 3)) * 4;
#pragma unroll
        for (int i = 0; i < TM; ++i) aoff[i][g] = (wr * 32 * TM + i * 32 + l31) * 16 + sw;
#pragma unroll
        for (int j = 0; j < TN; ++j) boff[j][g] = APANEL + (wc * 32 * TN + j * 32 + l31) * 16 + sw;
    }
    const int KS = K >> 4;
    issue(0, 0);
    for (int ks = 0; ks < KS; ++ks) {
        wait_vmcnt<0>();                            // this wave's part of chunk ks has landed
        __builtin_amdgcn_s_barrier();               // ... and everyone's; all reads of chunk ks-1 are done
        const float *st = smem + (ks & 1) * STAGE;
        float4 a[2][TM], b[2][TN];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
#pragma unroll
            for (int i = 0; i < TM; ++i) a[g][i] = *reinterpret_cast<const float4 *>(st + aoff[i][g]);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[g][j] = *reinterpret_cast<const float4 *>(st + boff[j][g]);
        }
        if (ks + 1 < KS) issue(ks + 1, (ks + 1) & 1);      // behind the fragment reads (hipcc drains LDS-DMA before a ds_read)
        if (PRIO) __builtin_amdgcn_s_setprio(2);
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float av[TM], bv[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) av[i] = t == 0 ? a[g][i].x : t == 1 ? a[g][i].y : t == 2 ? a[g][i].z : a[g][i].w;
#pragma unroll
                for (int j = 0; j < TN; ++j) bv[j] = t == 0 ? b[g][j].x : t == 1 ? b[g][j].y : t == 2 ? b[g][j].z : b[g][j].w;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv[j], av[i], acc[i][j], 0, 0, 0);
            }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    }
    if (TM == 2 && TN == 2 && exp_flags) {
        // exp-form store (kernels.h: PQ_EXP_LIMIT): the wave's 64 x 64 block goes out as 2^(PQ_C2 x) when every |PQ_C2 x| of
        // the block is within the limit, else unchanged with its flag raised.  Wave-local decision: no barrier; the VALU
        // is ~13 % busy in this kernel, the 64 v_exp_f32 per lane are free.
        float mx = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, fabsf(acc[i][j][r]));
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) mx = fmaxf(mx, __shfl_xor(mx, sft));
        const bool lin = !(mx * PQ_C2 <= PQ_EXP_LIMIT);       // (fmaxf drops NaNs, so a NaN does not force the linear form; exp2(NaN) is NaN either way)
        if (lane == 0) exp_flags[(int64_t)(mt * (BM / 64) + wr) * (N >> 6) + (n0 >> 6) + wc] = lin ? 1 : 0;
        if (!lin) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = __builtin_amdgcn_exp2f(PQ_C2 * acc[i][j][r]);
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int64_t gm = m0 + wr * 32 * TM + i * 32 + l31;
        if (gm >= M) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gn = n0 + wc * 32 * TN + j * 32 + 8 * q + 4 * lhi;      // 4 consecutive columns gn..gn+3
                const int64_t o = ((int64_t)(gn >> 4) * M + gm) * 16 + (gn & 15);
                float v[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = acc[i][j][4 * q + t] + (bias ? bias[gn + t] : 0.f);
                if (R) {
                    const float4 r4 = *reinterpret_cast<const float4 *>(R + o);
                    v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
                }
                if (act_tanh) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = fast_tanh(v[t]);
                }
                *reinterpret_cast<float4 *>(C + o) = make_float4(v[0] * alpha, v[1] * alpha, v[2] * alpha, v[3] * alpha);
            }
    }
}

// which LDS-DMA configuration a plain panel-major launch uses (0 = the register-staged kernel); set through
// upamd_tune("gemm_nt_dma", v) by the kernel lab / tests
// defaults = the best configuration of the kernel lab (tools/gemm_lab*.py, profiles/archive/r02_gemm_lab.md): LDS-DMA staging,
// three workgroups per CU (12 KB of LDS padding), first-residency-round stagger
static int g_nt_dma_variant = 1, g_stagger_mode = 1, g_stagger_cycles = 37000, g_lds_pad = 12 * 1024;
static int g_nt_split = 0;                      // 0 | 6 | 9, see launch_gemm_nt_ex
static void *g_split_scratch = nullptr;
static int64_t g_split_scratch_bytes = 0;
void set_gemm_nt_split(int nprod) { g_nt_split = (nprod == 6 || nprod == 9) ? nprod : 0; }
void set_gemm_lds_pad(int bytes) { g_lds_pad = bytes; }
void set_gemm_nt_dma_variant(int v) { g_nt_dma_variant = v; }
void set_gemm_stagger(int mode, int cycles) {
    if (mode >= 0) g_stagger_mode = mode;
    if (cycles >= 0) g_stagger_cycles = cycles;
}

template <int WM, int WN, int TM = 2, int TN = 2, bool PRIO = false>
static int launch_nt_dma2(const GemmNT &g, hipStream_t st) {
    constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
    const int MT = (int)((g.M + BM - 1) / BM), MT8 = (MT + 7) / 8 * 8, NT = g.N / BN;
    const size_t lds = sizeof(float) * 2 * (size_t)(BM + BN) * 16 + (size_t)g_lds_pad;
    auto kern = gemm_nt_dma2_kernel<WM, WN, TM, TN, PRIO>;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), (int64_t)lds)) return rc;
    // exp-form store: only the 64 x 64-per-wave forms, and only for a bare product
    uint8_t *flags = (TM == 2 && TN == 2 && !g.bias && !g.R && !g.act_tanh && g.alpha == 1.f) ? g.exp_flags : nullptr;
    hipLaunchKernelGGL(kern, dim3(MT8 * NT), dim3(64 * WM * WN), lds, st, g.A, g.M, g.K, g.W, g.N, g.ldw, g.bias, g.R, g.C,
                       g.act_tanh, g.alpha, MT, NT, g_stagger_mode, g_stagger_cycles, flags);
    if (g.exp_used) *g.exp_used = flags != nullptr;
    return 0;
}

// mirrors launch_gemm_nt_ex's choice of kernel: true iff a launch of g with exp_flags set will write them
static bool nt_dma_ok(const GemmNT &g);
static int nt_tile(const GemmNT &g);
bool gemm_nt_exp_store_ok(const GemmNT &g) {
    if (g.M <= 0 || g.bias || g.R || g.act_tanh || g.alpha != 1.f || g.a_rm || g.c_rm) return false;
    if (!gemm_nt_mfma_ok(g) || g.K % 16 != 0 || g.N % 16 != 0) return false;
    const int bn = nt_tile(g);
    if (g.K == 32 || bn != 128) return false;
    if (g_nt_split && g_nt_dma_variant != 0 && gemm_nt_split_ok(g)) return false;
    switch (g_nt_dma_variant) {
        case 1: case 2: case 9: return nt_dma_ok(g);
        case 3: case 4: return nt_dma_ok(g) && g.N % 256 == 0;
        default: return false;
    }
}

int64_t gemm_exp_flag_bytes(int64_t M, int N) { return 2 * ((M + 127) / 128) * (int64_t)(N / 64 > 0 ? N / 64 : 1); }

static bool nt_dma_ok(const GemmNT &g) {
    return !g.a_rm && !g.c_rm && !g.w_kn && g.N % 128 == 0 && g.K % 16 == 0 && g.K >= 64 && g.ldw % 4 == 0 &&
           reinterpret_cast<uintptr_t>(g.A) % 16 == 0 && reinterpret_cast<uintptr_t>(g.W) % 16 == 0;
}

// generic fallback (panel-major, any N % 16 == 0): one thread per output element
__global__ void gemm_nt_generic_kernel(const float *__restrict__ A, int64_t M, int K, const float *__restrict__ W,
                                       int N, int64_t ldw, const float *__restrict__ bias, const float *__restrict__ R,
                                       float *__restrict__ C, int act_tanh, float alpha) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= M * N) return;
    const int64_t m = g / N;
    const int n = (int)(g % N);
    float acc = bias ? bias[n] : 0.f;
    const float *w = W + (int64_t)n * ldw;
    for (int kp = 0; kp < (K >> 4); ++kp) {
        const float4 *a4 = reinterpret_cast<const float4 *>(A + ((int64_t)kp * M + m) * 16);
        const float4 *w4 = reinterpret_cast<const float4 *>(w + kp * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 a = a4[q], b = w4[q];
            acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
        }
    }
    const int64_t o = ((int64_t)(n >> 4) * M + m) * 16 + (n & 15);
    if (R) acc += R[o];
    if (act_tanh) acc = fast_tanh(acc);
    C[o] = acc * alpha;
}

int ensure_dynamic_lds(const void *kernel, int64_t lds_bytes) {
    if (lds_bytes <= 64 * 1024) return 0;
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, int64_t> raised;
    int dev = 0;
    UPAMD_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    int64_t &have = raised[std::make_pair(dev, kernel)];
    if (lds_bytes > have) {
        UPAMD_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
        have = lds_bytes;
    }
    return 0;
}

int prof_begin(Profiler *prof, const char *name, hipStream_t st, double flops, double bytes) {
    if (!prof || !prof->on) return 0;
    KernelStat &k = prof->stats[name];
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1;
    (void)hipEventRecord(e0, st);
    k.ev.push_back(e0);
    k.ev.push_back(e1);
    k.launches++;
    k.flops += flops;
    k.bytes += bytes;
    return 1;
}
void prof_end(Profiler *prof, const char *name, hipStream_t st, int began) {
    if (began == 1) (void)hipEventRecord(prof->stats[name].ev.back(), st);
}

bool gemm_nt_mfma_ok(const GemmNT &g) {
    if (g.K % 16 != 0 || g.N % 32 != 0 || g.ldw % 4 != 0) return false;
    if (reinterpret_cast<uintptr_t>(g.A) % 16 || reinterpret_cast<uintptr_t>(g.W) % 16) return false;
    if (g.a_rm && g.lda % 4 != 0) return false;
    return true;
}

// N-tile width of a launch.  Small-M problems (the per-sample [B, D] layers: 16 M-tiles) take a narrower tile so that
// at least ~half the CUs get a workgroup.
static int64_t g_nt_min_wgs = 128;              // tune knob "nt_min_wgs" (tests force the 128-wide tile on small problems with 1)
void set_gemm_nt_min_wgs(int v) { g_nt_min_wgs = v > 0 ? v : 128; }
static int nt_tile(const GemmNT &g) {
    const int64_t MIN_WGS = g_nt_min_wgs;
    const int64_t MT = (g.M + 127) / 128;
    if (g.N % 128 == 0 && MT * (g.N / 128) >= MIN_WGS) return 128;
    if (g.N % 64 == 0 && (g.N % 128 != 0 || MT * (g.N / 64) >= MIN_WGS)) return 64;
    return 32;
}

template <bool A_RM, bool C_RM, int PK>
static void launch_nt_layout(const GemmNT &g, hipStream_t st, bool k32 = false) {
    const int BNsel = nt_tile(g);
    const int MT = (int)((g.M + 127) / 128);
    const int MT8 = (MT + 7) / 8 * 8;
    if (k32) {
        const int NT = g.N / 128;
        hipLaunchKernelGGL((gemm_nt_mfma_kernel<128, 64, 64, A_RM, C_RM, PK, 32>), dim3(MT8 * NT), dim3(256), 0, st, g.A, g.M, g.K,
                           g.lda, g.W, g.N, g.ldw, g.bias, g.R, g.C, g.ldc, g.act_tanh, g.alpha, MT, NT, g.w_kn ? 1 : 0);
    } else if (BNsel == 128) {
        const int NT = g.N / 128;
        hipLaunchKernelGGL((gemm_nt_mfma_kernel<128, 64, 64, A_RM, C_RM, PK>), dim3(MT8 * NT), dim3(256), 0, st, g.A, g.M, g.K, g.lda,
                           g.W, g.N, g.ldw, g.bias, g.R, g.C, g.ldc, g.act_tanh, g.alpha, MT, NT, g.w_kn ? 1 : 0);
    } else if (BNsel == 64) {
        const int NT = g.N / 64;
        hipLaunchKernelGGL((gemm_nt_mfma_kernel<64, 64, 32, A_RM, C_RM, PK>), dim3(MT8 * NT), dim3(256), 0, st, g.A, g.M, g.K, g.lda,
                           g.W, g.N, g.ldw, g.bias, g.R, g.C, g.ldc, g.act_tanh, g.alpha, MT, NT, g.w_kn ? 1 : 0);
    } else {
        const int NT = g.N / 32;
        hipLaunchKernelGGL((gemm_nt_mfma_kernel<32, 32, 32, A_RM, C_RM, PK>), dim3(MT8 * NT), dim3(256), 0, st, g.A, g.M, g.K, g.lda,
                           g.W, g.N, g.ldw, g.bias, g.R, g.C, g.ldc, g.act_tanh, g.alpha, MT, NT, g.w_kn ? 1 : 0);
    }
}

int launch_gemm_nt_ex(const GemmNT &g, hipStream_t st, Profiler *prof) {
    if (g.exp_used) *g.exp_used = false;
    if (g.M <= 0) return 0;
    const bool mfma = gemm_nt_mfma_ok(g);
    if (!mfma && (g.a_rm || g.c_rm)) return fail(UPAMD_E_INVALID, "gemm_nt: row-major operands need the MFMA path (K=%d N=%d)", g.K, g.N);
    if (g.K % 16 != 0 || g.N % 16 != 0) return fail(UPAMD_E_INVALID, "gemm_nt: K and N must be multiples of 16 (K=%d N=%d)", g.K, g.N);
    const double flops = 2.0 * (double)g.M * g.K * g.N;
    const double bytes = 4.0 * ((double)g.M * g.K + (double)g.M * g.N * (g.R ? 2 : 1) + (double)g.N * g.K);
    const bool rm = g.a_rm || g.c_rm;
    // K = 32 (the raw-feature GEMMs): its own instantiation with the two-panel K loop unrolled at compile time; it is
    // bound by writing C (HBM), not by the MFMA pipe, and is accounted separately from the K >= 64 launches
    const int bn = mfma ? nt_tile(g) : 0;
    const bool k32 = mfma && !rm && g.K == 32 && bn == 128;
    const char *pname = !mfma ? "gemm_nt_generic"
                        : k32 ? "gemm_nt_128_k32"
                              : (bn == 128 ? (rm ? "gemm_nt_128_rm" : "gemm_nt_128")
                                           : (bn == 64 ? (rm ? "gemm_nt_64_rm" : "gemm_nt_64") : (rm ? "gemm_nt_32_rm" : "gemm_nt_32")));
    int began = prof_begin(prof, pname, st, flops, bytes);
    if (began < 0) return fail(UPAMD_E_HIP, "hipEventCreate failed");
    const int dv = (mfma && !k32 && bn == 128) ? g_nt_dma_variant : 0;
    int dma_rc = 1;                             // 1 = not taken
    // opt-in (tune knob "gemm_split" = 6 | 9, default 0 = exact fp32 MFMA): the K >= 64 panel-major products on the bf16
    // matrix pipe from a three-way bf16 split of both operands (gemm_split.hip); everything else is unchanged
    if (g_nt_split && dv != 0 && gemm_nt_split_ok(g)) {
        const int64_t need = gemm_nt_split_scratch_bytes(g.N, g.K);
        if (need > g_split_scratch_bytes) {      // lab knob: one lazily grown device buffer (launches are stream-ordered)
            if (g_split_scratch) (void)hipFree(g_split_scratch);
            g_split_scratch = nullptr;
            g_split_scratch_bytes = 0;
            UPAMD_HIP(hipMalloc(&g_split_scratch, (size_t)need));
            g_split_scratch_bytes = need;
        }
        dma_rc = launch_gemm_nt_split(g, g_split_scratch, g_nt_split, st, nullptr);
    } else {
        switch (dv) {
            case 1: if (nt_dma_ok(g)) dma_rc = launch_nt_dma2<2, 2>(g, st); break;
            case 2: if (nt_dma_ok(g)) dma_rc = launch_nt_dma2<4, 2>(g, st); break;
            case 3: if (nt_dma_ok(g) && g.N % 256 == 0) dma_rc = launch_nt_dma2<2, 4>(g, st); break;
            case 4: if (nt_dma_ok(g) && g.N % 256 == 0) dma_rc = launch_nt_dma2<4, 4>(g, st); break;
            // lab: a bigger block per wave (4 waves): 64 x 128 (workgroup 128 x 256) and 128 x 64 (256 x 128), +1 = with s_setprio
            case 5: if (nt_dma_ok(g) && g.N % 256 == 0) dma_rc = launch_nt_dma2<2, 2, 2, 4, false>(g, st); break;
            case 6: if (nt_dma_ok(g) && g.N % 256 == 0) dma_rc = launch_nt_dma2<2, 2, 2, 4, true>(g, st); break;
            case 7: if (nt_dma_ok(g)) dma_rc = launch_nt_dma2<2, 2, 4, 2, false>(g, st); break;
            case 8: if (nt_dma_ok(g)) dma_rc = launch_nt_dma2<2, 2, 4, 2, true>(g, st); break;
            case 9: if (nt_dma_ok(g)) dma_rc = launch_nt_dma2<2, 2, 2, 2, true>(g, st); break;
            default: break;
        }
    }
    if (dma_rc < 0) return dma_rc;
    if (dma_rc == 0) {
    }
    else if (mfma) {
        if (g.a_rm && g.c_rm) launch_nt_layout<true, true, 1>(g, st);
        else if (g.a_rm) launch_nt_layout<true, false, 1>(g, st);
        else if (g.c_rm) launch_nt_layout<false, true, 1>(g, st);
        else if (k32) launch_nt_layout<false, false, 1>(g, st, true);
        else launch_nt_layout<false, false, 1>(g, st);
    } else {
        const int64_t total = g.M * g.N;
        hipLaunchKernelGGL(gemm_nt_generic_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, g.A, g.M, g.K, g.W, g.N,
                           g.ldw, g.bias, g.R, g.C, g.act_tanh, g.alpha);
    }
    prof_end(prof, pname, st, began);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

int launch_gemm_nt(const float *A, int64_t M, int K, const float *W, int N, const float *bias, const float *R,
                   float *C, int act_tanh, hipStream_t st, Profiler *prof) {
    GemmNT g;
    g.A = A; g.M = M; g.K = K; g.lda = 0; g.a_rm = false; g.W = W; g.N = N; g.ldw = K; g.bias = bias; g.R = R;
    g.C = C; g.ldc = 0; g.c_rm = false; g.act_tanh = act_tanh; g.alpha = 1.f;
    return launch_gemm_nt_ex(g, st, prof);
}

// ------------------------------------------------------------------------------------------ TN
template <int BJ, int WI, int WJ, bool IN_RM>
__global__ __launch_bounds__(256) void gemm_tn_mfma_kernel(const float *__restrict__ A, int I, int64_t lda,
                                                           const float *__restrict__ Bm, int J, int64_t ldb, int64_t M,
                                                           int64_t chunk, float *__restrict__ slabs, int IT, int JT, int S) {
    constexpr int BI = 128, PS = 272;   // panel stride in LDS: 16 rows * 16 + 16 pad (bank shift 16)
    constexpr int WAVES_J = BJ / WJ;
    constexpr int TI = WI / 32, TJ = WJ / 32;
    constexpr int PA = BI / 16, PB = BJ / 16;
    constexpr int NBL = (PB * 64 + 255) / 256;
    static_assert((BI / WI) * WAVES_J == 4, "4 waves per workgroup");
    __shared__ __attribute__((aligned(16))) float As[2][PA * PS];
    __shared__ __attribute__((aligned(16))) float Bs[2][PB * PS];

    // XCD-aware order: workgroup id -> XCD id % 8; all tiles of one row-split run on the same XCD so the
    // split's A / B row chunks are fetched from HBM once and shared through that XCD's L2
    const int tiles = IT * JT;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int split = (slot / tiles) * 8 + xcd, tile = slot % tiles;
    if (split >= S) return;
    const int it = tile / JT, jt = tile % JT;
    const int i0 = it * BI, j0 = jt * BJ;
    const int64_t r0 = (int64_t)split * chunk;
    const int64_t r1 = (r0 + chunk < M) ? r0 + chunk : M;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w / WAVES_J, wc = w % WAVES_J;
    const int l31 = lane & 31, lhi = lane >> 5;

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[2], rb[NBL];
    auto load_tiles = [&](int64_t rr) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = tid + 256 * q, panel = e >> 6, rem = e & 63, row = rem >> 2, c4 = (rem & 3) * 4;
            const int64_t gr = rr + row;
            const float *src = IN_RM ? A + gr * lda + i0 + panel * 16 + c4
                                     : A + ((int64_t)(i0 / 16 + panel) * M + gr) * 16 + c4;
            ra[q] = gr < r1 ? *reinterpret_cast<const float4 *>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < NBL; ++q) {
            const int e = tid + 256 * q, panel = e >> 6, rem = e & 63, row = rem >> 2, c4 = (rem & 3) * 4;
            const int64_t gr = rr + row;
            if (e < PB * 64) {
                const float *src = IN_RM ? Bm + gr * ldb + j0 + panel * 16 + c4
                                         : Bm + ((int64_t)(j0 / 16 + panel) * M + gr) * 16 + c4;
                rb[q] = gr < r1 ? *reinterpret_cast<const float4 *>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = tid + 256 * q, panel = e >> 6, rem = e & 63, row = rem >> 2, c4 = (rem & 3) * 4;
            *reinterpret_cast<float4 *>(&As[buf][panel * PS + row * 16 + c4]) = ra[q];
        }
#pragma unroll
        for (int q = 0; q < NBL; ++q) {
            const int e = tid + 256 * q, panel = e >> 6, rem = e & 63, row = rem >> 2, c4 = (rem & 3) * 4;
            if (e < PB * 64) *reinterpret_cast<float4 *>(&Bs[buf][panel * PS + row * 16 + c4]) = rb[q];
        }
    };

    int aoff[TI], boff[TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        const int ci = wr * WI + i * 32 + l31;
        aoff[i] = (ci >> 4) * PS + (ci & 15) + lhi * 16;
    }
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int cj = wc * WJ + j * 32 + l31;
        boff[j] = (cj >> 4) * PS + (cj & 15) + lhi * 16;
    }

    if (r0 < r1) {
        load_tiles(r0);
        store_tiles(0);
    }
    __syncthreads();
    int it_k = 0;
    for (int64_t rr = r0; rr < r1; rr += 16, ++it_k) {
        const int buf = it_k & 1;
        const bool more = rr + 16 < r1;
        if (more) load_tiles(rr + 16);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            float a[TI], b[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) a[i] = As[buf][aoff[i] + 32 * s];
#pragma unroll
            for (int j = 0; j < TJ; ++j) b[j] = Bs[buf][boff[j] + 32 * s];
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) store_tiles(buf ^ 1);
        __syncthreads();
    }

    float *slab = slabs + (int64_t)split * I * J;
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int gj = j0 + wc * WJ + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const int gi = i0 + wr * WI + i * 32 + row;
                slab[(int64_t)gi * J + gj] = acc[i][j][r];
            }
        }
}


__global__ void gemm_tn_generic_kernel(const float *__restrict__ A, int I, const float *__restrict__ Bm, int J,
                                       int64_t M, int64_t chunk, float *__restrict__ slabs) {
    const int ij = blockIdx.x * blockDim.x + threadIdx.x;
    const int split = blockIdx.y;
    if (ij >= I * J) return;
    const int i = ij / J, j = ij % J;
    const int64_t r0 = (int64_t)split * chunk;
    const int64_t r1 = (r0 + chunk < M) ? r0 + chunk : M;
    const float *a = A + ((int64_t)(i >> 4) * M) * 16 + (i & 15);
    const float *b = Bm + ((int64_t)(j >> 4) * M) * 16 + (j & 15);
    float acc = 0.f;
    for (int64_t r = r0; r < r1; ++r) acc = fmaf(a[r * 16], b[r * 16], acc);
    slabs[(int64_t)split * I * J + ij] = acc;
}

static bool tn_use_mfma(int I, int J) { return I % 128 == 0 && J % 32 == 0; }
bool tn_shape_mfma_ok(int I, int J) { return tn_use_mfma(I, J); }

int tn_splits(int I, int J, int64_t M) {
    if (M <= 0) return 1;
    int64_t tiles;
    if (tn_use_mfma(I, J)) {
        const int BJ = (J % 128 == 0) ? 128 : (J % 64 == 0 ? 64 : 32);
        tiles = (int64_t)(I / 128) * (J / BJ);
    } else {
        tiles = ((int64_t)I * J + 255) / 256;
    }
    int64_t S = (768 + tiles - 1) / tiles;       // ~3 workgroups per CU: measured best (1024: -2 %, 512: -1 %)
    const int64_t max_s = (M + 127) / 128;   // at least 128 rows per split
    if (S > max_s) S = max_s;
    if (S < 1) S = 1;
    if (S > 1024) S = 1024;
    return (int)S;
}

bool gemm_tn_mfma_ok(const GemmTN &g) {
    if (!tn_use_mfma(g.I, g.J)) return false;
    if (g.in_rm && (g.lda % 4 != 0 || g.ldb % 4 != 0)) return false;
    if (reinterpret_cast<uintptr_t>(g.A) % 16 || reinterpret_cast<uintptr_t>(g.Bm) % 16) return false;
    return true;
}

template <bool IN_RM>
static void launch_tn_layout(const GemmTN &g, int S, int64_t chunk, hipStream_t st) {
    const int IT = g.I / 128;
    if (g.J % 128 == 0) {
        const int JT = g.J / 128;
        hipLaunchKernelGGL((gemm_tn_mfma_kernel<128, 64, 64, IN_RM>), dim3((S + 7) / 8 * 8 * IT * JT), dim3(256), 0, st, g.A, g.I, g.lda, g.Bm, g.J,
                           g.ldb, g.M, chunk, g.slabs, IT, JT, S);
    } else if (g.J % 64 == 0) {
        const int JT = g.J / 64;
        hipLaunchKernelGGL((gemm_tn_mfma_kernel<64, 64, 32, IN_RM>), dim3((S + 7) / 8 * 8 * IT * JT), dim3(256), 0, st, g.A, g.I, g.lda, g.Bm, g.J,
                           g.ldb, g.M, chunk, g.slabs, IT, JT, S);
    } else {
        const int JT = g.J / 32;
        hipLaunchKernelGGL((gemm_tn_mfma_kernel<32, 32, 32, IN_RM>), dim3((S + 7) / 8 * 8 * IT * JT), dim3(256), 0, st, g.A, g.I, g.lda, g.Bm, g.J,
                           g.ldb, g.M, chunk, g.slabs, IT, JT, S);
    }
}

int launch_gemm_tn_ex(const GemmTN &g, int *S_out, hipStream_t st, Profiler *prof) {
    if (g.I % 16 != 0 || g.J % 16 != 0) return fail(UPAMD_E_INVALID, "gemm_tn: I and J must be multiples of 16 (I=%d J=%d)", g.I, g.J);
    const bool mfma = gemm_tn_mfma_ok(g);
    if (!mfma && g.in_rm) return fail(UPAMD_E_INVALID, "gemm_tn: row-major operands need the MFMA path (I=%d J=%d)", g.I, g.J);
    const int S = tn_splits(g.I, g.J, g.M);
    *S_out = S;
    if (g.M <= 0) {
        UPAMD_HIP(hipMemsetAsync(g.slabs, 0, sizeof(float) * (size_t)g.I * g.J, st));
        return 0;
    }
    int64_t chunk = (g.M + S - 1) / S;
    chunk = (chunk + 15) / 16 * 16;
    const double flops = 2.0 * (double)g.M * g.I * g.J;
    const double bytes = 4.0 * ((double)g.M * (g.I + g.J) + (double)S * g.I * g.J);
    const char *pname = !mfma ? "gemm_tn_generic" : (g.J % 128 == 0 ? "gemm_tn_128" : (g.J % 64 == 0 ? "gemm_tn_64" : "gemm_tn_32"));
    int began = prof_begin(prof, pname, st, flops, bytes);
    if (began < 0) return fail(UPAMD_E_HIP, "hipEventCreate failed");
    if (mfma) {
        if (g.in_rm) launch_tn_layout<true>(g, S, chunk, st);
        else launch_tn_layout<false>(g, S, chunk, st);
    } else {
        hipLaunchKernelGGL(gemm_tn_generic_kernel, dim3((g.I * g.J + 255) / 256, S), dim3(256), 0, st, g.A, g.I, g.Bm, g.J, g.M, chunk,
                           g.slabs);
    }
    prof_end(prof, pname, st, began);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

int launch_gemm_tn(const float *A, int I, const float *Bm, int J, int64_t M, float *slabs, int *S_out,
                   hipStream_t st, Profiler *prof) {
    GemmTN g;
    g.A = A; g.I = I; g.lda = 0; g.Bm = Bm; g.J = J; g.ldb = 0; g.M = M; g.in_rm = false; g.slabs = slabs;
    return launch_gemm_tn_ex(g, S_out, st, prof);
}

// dst (+)= sum_s slabs[s][i][j] in a FIXED order: 16 slab groups (s = g, g+16, ...) are summed by 16 threads per
// output element, then combined g = 0..15 -- deterministic, and 16x more parallel than one thread per element
// (the small-output reductions have up to 384 slabs for only 8192 elements).
__global__ __launch_bounds__(1024) void reduce_slabs_kernel(const float *__restrict__ slabs, int S, int I, int J, int mode,
                                                            int jkeep, float *__restrict__ dst, int ldd,
                                                            float *__restrict__ last_col_dst) {
    __shared__ float part[16][65];
    const int e = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int ij = blockIdx.x * 64 + e;
    const bool in = ij < I * J;
    float acc = 0.f;
    if (in)
        for (int s = g; s < S; s += 16) acc += slabs[(int64_t)s * I * J + ij];
    part[g][e] = acc;
    __syncthreads();
    if (g != 0 || !in) return;
    acc = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc += part[q][e];
    const int i = ij / J, j = ij % J;
    if (mode == 0) {
        if (j < jkeep) dst[(int64_t)i * ldd + j] += acc;
        if (last_col_dst && j == J - 1) last_col_dst[i] += acc;      // the B operand's last column was all ones: a column sum
    } else if (mode == 1) {
        dst[(int64_t)j * ldd + i] += acc;
    } else {
        // slab row i = P/Q position (pair order, kernels.h) of the layer; slab col j = input feature k
        const int row = pq_col(i);
        const int half = pq_side(i);
        dst[(int64_t)row * ldd + half * J + j] += acc;
    }
}

int launch_reduce_slabs(const float *slabs, int S, int I, int J, int mode, int jkeep, float *dst, int ldd,
                        hipStream_t st, float *last_col_dst) {
    hipLaunchKernelGGL(reduce_slabs_kernel, dim3((I * J + 63) / 64), dim3(1024), 0, st, slabs, S, I, J, mode, jkeep, dst, ldd,
                       last_col_dst);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

}  // namespace upamd
