// GCN message passing on the ragged batch (gfx950, wave64): forward and backward of one layer.
//
// Reference math (urban_planning/models/state_encoder.py:110-148,194-197): for a live edge (i,j)
//   m_ij = 1/2 [ tanh(W [h_i;h_j] + b) + tanh(W [h_j;h_i] + b) ],   h_v += sum_{e touches v} m_e / (deg_v + 1e-6)
// with W [h_i;h_j] = P_i + Q_j, where P = H Wa^T and Q = H Wb^T (W = [Wa | Wb]) come from the node GEMM.
//
// Work decomposition: one 1024-thread workgroup per (graph, 16-column panel).  In the panel-major
// layout the graph's P, Q and H slices for those 16 columns are contiguous runs of n*64 B; they are
// staged into LDS once with coalesced 16-byte loads (P/Q interleaved so a neighbour costs one
// ds_read_b128 for two columns), then every gather hits LDS.  A wave handles EIGHT nodes at a time --
// one per 8-lane group, each lane owning two adjacent columns -- walking that node's incidence list
// sequentially: a CSR-by-destination segment sum in a fixed order, no atomics, no cross-lane reduction,
// bit-reproducible.  Nodes are visited in the packer's degree-sorted order so the eight lists a wave walks
// together have (nearly) equal length; chunks of eight are dealt to the 16 waves in serpentine order.
// 16 waves per workgroup x 2 workgroups per CU = the full 8 waves per SIMD.
//
// The kernels are VALU-issue bound (PMC: SQ_ACTIVE_INST_VALU ~80 % of the SIMD cycles), so the walk is trimmed to
// the minimum: P and Q are staged as 2^(2 log2e P), 2^(2 log2e Q) ("exp form", stage_pq_exp below), so with
//   E = 2^(2 log2e (P_v + Q_u + b)) = eP_v * eb * eQ_u,   r = 1 / (1 + E) = v_rcp_f32(fma(eP_v eb, eQ_u, 1))
//   tanh(x) = 1 - 2 r                                      (abs error ~1e-7, clean saturation)
// the forward only accumulates r (sum of tanh = 2 deg - 2 sum r) and the backward uses 1 - tanh^2 = 4 (r - r^2):
// 2 FMAs + 2 v_rcp_f32 per incidence and column, no exponential in the loop.  Workgroups whose slice leaves the
// safe exponent range fall back to the linear form (P, Q pre-scaled by 2 log2e, v_exp_f32 in the loop).
#include <type_traits>

#include "kernels.h"

namespace upamd {

#define META(t) (pk.meta + (int64_t)(t) * UPAMD_META_STRIDE)

// Every kernel here is capped at 72 SGPRs (amdgpu_num_sgpr): the runtime's trap handler adds 16 SGPRs to each wave's
// allocation, so a kernel above 80 SGPRs allocates 112 -> 7 waves per SIMD -> with 16-wave workgroups only ONE
// workgroup per CU instead of two (measured: the last-layer forward went from 1.62 ms back to 1.05 ms).
constexpr int EDGE_THREADS = 1024;
constexpr int EDGE_WAVES = EDGE_THREADS / 64;
constexpr int64_t LDS_LIMIT = 160 * 1024 - 1024;      // dynamic LDS a launch may ask for (the kernels also own 256 static bytes)
constexpr int64_t LDS_HALF = 80 * 1024 - 2048;       // two workgroups per CU (with slack for the allocation granularity)
constexpr int64_t LDS_HALF_HARD = 80 * 1024 - 512;   // ... with next to no slack (256 static bytes + rounding): the DHM size class
constexpr float C2 = PQ_C2;                    // 2 * log2(e)

__device__ __forceinline__ float rcp1p_exp2(float x) {      // 1 / (1 + 2^x)
    return __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x) + 1.0f);
}

__host__ __device__ static inline int64_t a16(int64_t x) { return (x + 15) / 16 * 16; }

__host__ __device__ int64_t edge_lds_bytes(int max_n, int max_inc, bool bwd, bool last, bool stage, bool hlds, bool nblds) {
    (void)last;
    int64_t b = 0;
    if (stage) b += (int64_t)max_n * (hlds ? 192 : 128);      // P/Q interleaved (128 B/node) + H or dS (64 B/node)
    b = a16(b);
    b += a16(((int64_t)max_n + 1) * 4);                       // row_ptr
    if (nblds) b += a16((int64_t)max_inc * 2);                // neighbour ids (u16)
    b += a16((int64_t)max_n * 2);                             // processing order (u16)
    if (!bwd) b += a16(max_n);                                // node_mask bytes
    b += EDGE_WAVES * 2 * 16 * 4;                             // cross-wave reduction scratch
    return b;
}

struct EdgeLds {
    float2 *PQ;
    float *X;          // H (forward) or dS (backward)
    int *rp;
    uint16_t *nb, *ord;
    uint8_t *nm;
    float *red;
    unsigned char *aux;   // last layer: the row's candidate-edge lists (sized by the launcher from the spare LDS)
};

__device__ __forceinline__ EdgeLds carve(unsigned char *smem, int n, int e, bool stage, bool bwd, bool hlds = true, bool nblds = true) {
    EdgeLds L;
    int64_t o = 0;
    L.PQ = reinterpret_cast<float2 *>(smem);
    L.X = reinterpret_cast<float *>(smem + (int64_t)n * 128);      // (not there when !hlds: the forward then keeps H in HBM)
    if (stage) o = (int64_t)n * (hlds ? 192 : 128);
    o = (o + 15) / 16 * 16;
    L.rp = reinterpret_cast<int *>(smem + o); o += (((int64_t)n + 1) * 4 + 15) / 16 * 16;
    L.nb = reinterpret_cast<uint16_t *>(smem + o); if (nblds) o += ((int64_t)e * 4 + 15) / 16 * 16;
    L.ord = reinterpret_cast<uint16_t *>(smem + o); o += ((int64_t)n * 2 + 15) / 16 * 16;
    L.nm = reinterpret_cast<uint8_t *>(smem + o);
    if (!bwd) o += ((int64_t)n + 15) / 16 * 16;
    L.red = reinterpret_cast<float *>(smem + o);
    L.aux = smem + o + EDGE_WAVES * 2 * 16 * 4;
    return L;
}

// The P/Q tensor is in pair order (kernels.h): the 16 floats of node v in panel 2p are the column pairs 0..3 of the
// slice -- chunk j = (P_2j, P_2j+1, Q_2j, Q_2j+1), the order the packed two-column math of the walks consumes -- and
// panel 2p + 1 holds pairs 4..7.  The LDS row of a node (32 floats) is [panel 2p's 16 | panel 2p + 1's 16]: staging is a
// 16-byte-chunk copy.  Pg / Qg below are those two panel slices ("A" / "B" halves of the rows), not P and Q.
// float4 index i of a half (node i >> 2, chunk i & 3) -> its float offset in the LDS slice
__device__ __forceinline__ int pq_slot(int i) { return (i >> 2) * 32 + (i & 3) * 4; }

// linear form, pre-scaled by C2
__device__ __forceinline__ void stage_pq(float2 *PQl, const float *Pg, const float *Qg, int n) {
    const float4 *p4 = reinterpret_cast<const float4 *>(Pg);
    const float4 *q4 = reinterpret_cast<const float4 *>(Qg);
    float *dl = reinterpret_cast<float *>(PQl);
    for (int i = threadIdx.x; i < n * 4; i += EDGE_THREADS) {
        const float4 pp = p4[i], qq = q4[i];
        float *d = dl + pq_slot(i);
        *reinterpret_cast<float4 *>(d) = make_float4(C2 * pp.x, C2 * pp.y, C2 * pp.z, C2 * pp.w);
        *reinterpret_cast<float4 *>(d + 16) = make_float4(C2 * qq.x, C2 * qq.y, C2 * qq.z, C2 * qq.w);
    }
}

// Exp-form staging: PQ[v][c] = (2^(C2 P), 2^(C2 Q)).  2^(P'_v + Q'_u + b') = 2^(P'_v + b') * 2^(Q'_u) moves both
// exponentials out of the incidence loop (n*32 v_exp_f32 per item instead of 2 per incidence and column):
//   r = 1 / (1 + E) = rcp(fma(eP_v, eQ_u, 1)).
// Only valid while no factor over/underflows: every |C2 P|, |C2 Q| <= EF_LIMIT and |C2 b| <= EF_BIAS_LIMIT keep
// all factors and products inside [2^-118, 2^118].  The function returns false otherwise and the workgroup
// re-stages in linear form and walks with the exponentials inside the loop (same math, any magnitude).
constexpr float EF_LIMIT = 56.f, EF_BIAS_LIMIT = 6.f;
// The forward needs only r1 + r2 of an edge, and with t = 1 + E1, u = 1 + E2:  1/t + 1/u = (t + u) / (t u) -- ONE
// quarter-rate v_rcp_f32 instead of two (2 FMA + mul + add + rcp + FMA per column and incidence).  t u must stay
// finite: 2 (2 EF_LIMIT_FWD + EF_BIAS_LIMIT) < 127.
constexpr float EF_LIMIT_FWD = 28.f;

// writes the exp form of float4 i of both row halves of the slice; returns the largest |C2 x| seen
__device__ __forceinline__ float put_pq_exp(float2 *PQl, int i, const float4 &pp, const float4 &qq) {
    const float ax = C2 * pp.x, ay = C2 * pp.y, az = C2 * pp.z, aw = C2 * pp.w;
    const float bx = C2 * qq.x, by = C2 * qq.y, bz = C2 * qq.z, bw = C2 * qq.w;
    float *d = reinterpret_cast<float *>(PQl) + pq_slot(i);
    *reinterpret_cast<float4 *>(d) = make_float4(__builtin_amdgcn_exp2f(ax), __builtin_amdgcn_exp2f(ay), __builtin_amdgcn_exp2f(az),
                                                 __builtin_amdgcn_exp2f(aw));
    *reinterpret_cast<float4 *>(d + 16) = make_float4(__builtin_amdgcn_exp2f(bx), __builtin_amdgcn_exp2f(by), __builtin_amdgcn_exp2f(bz),
                                                      __builtin_amdgcn_exp2f(bw));
    return fmaxf(fmaxf(fmaxf(fabsf(ax), fabsf(ay)), fmaxf(fabsf(az), fabsf(aw))),
                 fmaxf(fmaxf(fabsf(bx), fabsf(by)), fmaxf(fabsf(bz), fabsf(bw))));
}

__device__ __forceinline__ bool stage_pq_exp(float2 *PQl, const float *Pg, const float *Qg, int n, float limit = EF_LIMIT) {
    const float4 *p4 = reinterpret_cast<const float4 *>(Pg);
    const float4 *q4 = reinterpret_cast<const float4 *>(Qg);
    float mx = 0.f;
    for (int i = threadIdx.x; i < n * 4; i += EDGE_THREADS) mx = fmaxf(mx, put_pq_exp(PQl, i, p4[i], q4[i]));
    return mx <= limit;
}

// Stage-in with ONE memory round trip: the workgroup-lifetime profile (s_memtime) showed the stage-in taking longer
// than the walk itself (15.5 k vs 12.9 k cycles), because the staging loops were seven dependent global round
// trips (P/Q twice, H twice, row pointers, neighbour ids, order/mask).  When the slice fits two trips per thread
// (n <= 512 nodes, e <= 2048 edges) every load is issued into registers first and only then committed to LDS.
__device__ __forceinline__ bool fits_batched(int n, int e) { return n * 4 <= 2 * EDGE_THREADS && e <= 2 * EDGE_THREADS; }

// ------------------------------------------------------------------------------------------
// LDS-DMA stage-in (round 3).  With the P/Q tensor in pair order and already in exp form (the node GEMM's epilogue writes
// 2^(C2 x) block by block, kernels.h: PQ_EXP_LIMIT) the slice a workgroup needs is a pure 16-byte-chunk copy: it goes
// HBM -> LDS by global_load_lds_dwordx4 without touching a VGPR, the exponentials and the ds_write pass of the register
// path are gone, and so is the 64-VGPR squeeze around them.  `pqflag` (one byte per 64 x 64 block of the GEMM output,
// 1 = that block is in linear form) is read alongside; a workgroup that meets a linear block -- or a bias outside the
// exp-form range -- converts its slice IN PLACE afterwards (dma_fixup: rare, slow, exact to ~3e-7).
// ------------------------------------------------------------------------------------------
// P/Q slice: wave-instruction t fills LDS bytes [1024 t, 1024 t + 1024) = nodes 8t .. 8t+7 (lane: node 8t + lane/8, chunk lane%8);
// chunks 0..3 come from panel 2p (A), 4..7 from panel 2p + 1 (B).  Lanes of nodes >= n are masked off.
__device__ __forceinline__ void dma_pq_slice(float2 *PQl, const float *Ag, const float *Bg, int n) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned char *l = reinterpret_cast<unsigned char *>(PQl);
    for (int t = w; t < ((n + 7) >> 3); t += EDGE_WAVES) {
        const int node = 8 * t + (lane >> 3);
        if (node < n)
            __builtin_amdgcn_global_load_lds((gptr_t)(((lane & 4) ? Bg : Ag) + node * 16 + 4 * (lane & 3)), (lptr_t)(l + t * 1024), 16, 0, 0);
    }
}
// a 16-column slice (H or G: 64 B per node): wave-instruction t = nodes 16t .. 16t+15, lane: node 16t + lane/4, chunk lane%4
__device__ __forceinline__ void dma_x_slice(float *Xl, const float *Xg, int n) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned char *l = reinterpret_cast<unsigned char *>(Xl);
    for (int t = w; t < ((n + 15) >> 4); t += EDGE_WAVES) {
        const int node = 16 * t + (lane >> 2);
        if (node < n) __builtin_amdgcn_global_load_lds((gptr_t)(Xg + node * 16 + 4 * (lane & 3)), (lptr_t)(l + t * 1024), 16, 0, 0);
    }
}
// 1 if any 64-row block of the GEMM output that this graph's rows [o, o + n) touch (column block of panel pair p) is linear
__device__ __forceinline__ int dma_flags_bad(const uint8_t *pqflag, int nfb, int64_t o, int n, int p) {
    const int64_t r0 = o >> 6;
    const int nr = (int)(((o + n - 1) >> 6) - r0) + 1;
    int bad = 0;
    for (int i = threadIdx.x; i < nr; i += EDGE_THREADS) bad |= pqflag[(r0 + i) * nfb + (p >> 1)];
    return bad;
}
// The slice in LDS is a raw copy with at least one linear block in it (or the bias is out of range): bring every float to
// the linear form C2 x (exp-form blocks through log2), then back to exp form if the workgroup's own slice allows it.
// Returns true when LDS holds the exp form.  Contains barriers: call from uniform control flow.
__device__ __forceinline__ bool dma_fixup(float2 *PQl, const uint8_t *pqflag, int nfb, int64_t o, int n, int p, float limit, bool bias_ok) {
    float4 *l4 = reinterpret_cast<float4 *>(PQl);
    float mx = 0.f;
    for (int i = threadIdx.x; i < n * 8; i += EDGE_THREADS) {            // 16-byte chunk i: node i / 8
        float4 v = l4[i];
        if (pqflag[((o + (i >> 3)) >> 6) * nfb + (p >> 1)])
            v = make_float4(C2 * v.x, C2 * v.y, C2 * v.z, C2 * v.w);
        else
            v = make_float4(__builtin_amdgcn_logf(v.x), __builtin_amdgcn_logf(v.y), __builtin_amdgcn_logf(v.z), __builtin_amdgcn_logf(v.w));
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        l4[i] = v;
    }
    const bool ef = !__syncthreads_or((mx <= limit && bias_ok) ? 0 : 1);
    if (ef) {
        for (int i = threadIdx.x; i < n * 8; i += EDGE_THREADS) {
            const float4 v = l4[i];
            l4[i] = make_float4(__builtin_amdgcn_exp2f(v.x), __builtin_amdgcn_exp2f(v.y), __builtin_amdgcn_exp2f(v.z), __builtin_amdgcn_exp2f(v.w));
        }
        __syncthreads();
    }
    return ef;
}
// the graph's lists (row pointers, neighbour ids, processing order, node-mask bytes) through registers: up to three neighbour
// words, one entry of the others per thread, all requested before anything is committed (one round trip)
struct ListRegs {
    uint32_t nb0, nb1, nb2, ord, nm;
    int rp;
};
__device__ __forceinline__ void lists_load(ListRegs &r, const int32_t *rpg, const uint32_t *nbg, const uint16_t *og, const uint8_t *nmg,
                                           int n, int e) {
    const int tid = threadIdx.x;
    r.nb0 = nbg[tid < e ? tid : 0];
    r.nb1 = nbg[tid + EDGE_THREADS < e ? tid + EDGE_THREADS : 0];
    r.nb2 = nbg[tid + 2 * EDGE_THREADS < e ? tid + 2 * EDGE_THREADS : 0];
    r.rp = rpg[tid <= n ? tid : 0];
    r.ord = og[tid < n ? tid : 0];
    r.nm = nmg ? nmg[tid < n ? tid : 0] : 0;
}
__device__ __forceinline__ void lists_commit(const ListRegs &r, const EdgeLds &L, const int32_t *rpg, const uint32_t *nbg, const uint16_t *og,
                                             const uint8_t *nmg, int n, int e) {
    const int tid = threadIdx.x;
    uint32_t *nb32 = reinterpret_cast<uint32_t *>(L.nb);
    if (tid < e) nb32[tid] = r.nb0;
    if (tid + EDGE_THREADS < e) nb32[tid + EDGE_THREADS] = r.nb1;
    if (tid + 2 * EDGE_THREADS < e) nb32[tid + 2 * EDGE_THREADS] = r.nb2;
    for (int i = tid + 3 * EDGE_THREADS; i < e; i += EDGE_THREADS) nb32[i] = nbg[i];
    if (tid <= n) L.rp[tid] = r.rp;
    if (tid < n) {
        L.ord[tid] = (uint16_t)r.ord;
        if (nmg) L.nm[tid] = (uint8_t)r.nm;
    }
    for (int i = tid + EDGE_THREADS; i <= n; i += EDGE_THREADS) L.rp[i] = rpg[i];
    for (int i = tid + EDGE_THREADS; i < n; i += EDGE_THREADS) {
        L.ord[i] = og[i];
        if (nmg) L.nm[i] = nmg[i];
    }
}

__device__ __forceinline__ float rcp1p_mul(float a, float b) {      // 1 / (1 + a*b)
    return __builtin_amdgcn_rcpf(fmaf(a, b, 1.0f));
}

// ------------------------------------------------------------------------------------------
// First GCN layer folded into the stage-in (FOLD): the layer's P/Q slice is an affine function of the <= 32 raw node
// features, PQ_1 = Xp (Wcat_1 We)^T + Wcat_1 be, and so is its input H_0 = Xp We^T + be.  Instead of reading them from
// HBM (and a K = 32 GEMM writing them there: 1.7 GB written + 2.9 GB re-read per step at B = 2048, D = 256) the workgroup
// computes its (graph, panel) slice with the matrix cores straight into LDS: a wave takes a 32-node tile,
// v_mfma_f32_32x32x2_f32 with the 32 LDS floats of a node (8 column pairs x [P,P,Q,Q], or the 16 H columns) as the
// register dimension and the nodes as the lane dimension, so a lane ends up with whole 16-byte LDS groups of ONE node.
// Only the first 24 features can be non-zero (node_dim <= 24; column 31 of Xp is the ones column of the bias-gradient
// trick and meets a zero weight), and the K order is free: lane half kh covers features 12 kh .. 12 kh + 11, three
// 16-byte loads per operand and 12 MFMA steps.  The backward recomputes the same slice with the same code
// (bit-identical to the forward's).  Everything stays within the 64 VGPRs of a two-workgroups-per-CU kernel.
// ------------------------------------------------------------------------------------------
typedef float f32x16e __attribute__((ext_vector_type(16)));
typedef float f32x2e __attribute__((ext_vector_type(2)));

// Operands of one 32-node tile, all loaded up front (ONE memory round trip per tile): the node features and the weight
// rows of either the tile's 32 P/Q floats or its 16 H_0 columns.  The bias rides along as a 13th k step (bias x 1, the
// upper lane half contributes 0), so the epilogue has no memory access of its own.
struct FoldOps {
    float4 xa[3], wa[3];
    float bias;
};

__device__ __forceinline__ void fold_load_x(const FoldArgs &fa, int64_t M, int64_t o, int n, int tile, FoldOps &ops) {
    const int lane = threadIdx.x & 63, r = lane & 31, kh = lane >> 5;
    const int vc = min(tile * 32 + r, n - 1);
    // features 12 kh + 4 q .. + 3:  kh = 0: panel 0 floats 0, 4, 8;  kh = 1: panel 0 float 12, panel 1 floats 0, 4
    const float *x0 = fa.Xp + (o + vc) * 16, *x1 = x0 + M * 16;
    ops.xa[0] = *reinterpret_cast<const float4 *>(kh ? x0 + 12 : x0);
    ops.xa[1] = *reinterpret_cast<const float4 *>(kh ? x1 : x0 + 4);
    ops.xa[2] = *reinterpret_cast<const float4 *>(kh ? x1 + 4 : x0 + 8);
}

__device__ __forceinline__ void fold_load_w(const FoldArgs &fa, int p, bool h, FoldOps &ops) {
    const int lane = threadIdx.x & 63, r = lane & 31, kh = lane >> 5;
    // weight row behind LDS float r of a node.  P/Q tile: pair q = r / 4 holds [P_2q, P_2q+1, Q_2q, Q_2q+1] and the rows of
    // W1c are in the same pair order (kernels.h): row 32 p + r; H tile: column r (< 16; the upper half of the tile is discarded)
    const int wrow = h ? 16 * p + (r & 15) : 32 * p + r;
    const float4 *w4 = reinterpret_cast<const float4 *>((h ? fa.We : fa.W1c) + (int64_t)wrow * 32 + 12 * kh);
#pragma unroll
    for (int q = 0; q < 3; ++q) ops.wa[q] = w4[q];
    ops.bias = (h ? fa.be : fa.b1c)[wrow];
}

// the tile's matrix chain and its write to LDS; returns the largest |C2 * value| of the P/Q floats written
__device__ __forceinline__ float fold_mma(const FoldOps &ops, int n, int tile, bool h, float2 *PQl, float *Xl, bool expform) {
    const int lane = threadIdx.x & 63, r = lane & 31, kh = lane >> 5;
    f32x16e acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ops.wa[q].x, ops.xa[q].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ops.wa[q].y, ops.xa[q].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ops.wa[q].z, ops.xa[q].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ops.wa[q].w, ops.xa[q].w, acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kh ? 0.f : ops.bias, 1.0f, acc, 0, 0, 0);
    // accumulator 4 g + t of this lane = LDS float 8 g + 4 kh + t of node tile * 32 + r
    const int v = tile * 32 + r;
    float mx = 0.f;
    if (v >= n) return mx;
    if (!h) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int q = 2 * g + kh;                  // column pair
            const float y0 = C2 * acc[4 * g + 0], y1 = C2 * acc[4 * g + 1];
            const float y2 = C2 * acc[4 * g + 2], y3 = C2 * acc[4 * g + 3];
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(y0), fabsf(y1))), fmaxf(fabsf(y2), fabsf(y3)));
            float4 *d = reinterpret_cast<float4 *>(reinterpret_cast<float *>(PQl) + v * 32 + 4 * q);
            if (expform)
                *d = make_float4(__builtin_amdgcn_exp2f(y0), __builtin_amdgcn_exp2f(y1), __builtin_amdgcn_exp2f(y2),
                                 __builtin_amdgcn_exp2f(y3));
            else
                *d = make_float4(y0, y1, y2, y3);
        }
    } else {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int c = 8 * g + 4 * kh;
            *reinterpret_cast<float4 *>(Xl + v * 16 + c) = make_float4(acc[4 * g + 0], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
        }
    }
    return mx;
}

// The (graph, panel) slice of the folded first layer: P/Q (always) and H_0 (WITH_H: the forward).  Up to 512 nodes
// (T <= 16 tiles) every wave has one trip: wave w < T takes P/Q tile w; the H tiles go to the idle waves T .. 15 first and
// the rest to the wave that holds the same nodes' P/Q tile (its node-feature operand is reused).  Split in two so the
// caller can commit its own in-flight loads to LDS between the operand fetch and the matrix chain.
struct FoldRole {
    int tile;          // < 0: this wave has no tile
    bool h, both;      // h: the first chain is the H tile;  both: a P/Q tile followed by the same nodes' H tile
};
template <bool WITH_H>
__device__ __forceinline__ FoldRole fold_begin(const FoldArgs &fa, int64_t M, int64_t o, int n, int p, FoldOps &ops) {
    const int w = threadIdx.x >> 6;
    const int T = (n + 31) >> 5;
    FoldRole R{-1, false, false};
    if (T > EDGE_WAVES) return R;                     // > 512 nodes: fold_end walks the tiles one after the other
    if (w < T) {
        R.tile = w;
        R.both = WITH_H && w >= EDGE_WAVES - T;
    } else if (WITH_H && w - T < T) {
        R.tile = w - T;
        R.h = true;
    }
    if (R.tile >= 0) {
        fold_load_x(fa, M, o, n, R.tile, ops);
        fold_load_w(fa, p, R.h, ops);
    }
    return R;
}
template <bool WITH_H>
__device__ __forceinline__ float fold_end(const FoldArgs &fa, int64_t M, int64_t o, int n, int p, const FoldRole &R, FoldOps &ops,
                                          float2 *PQl, float *Xl, bool expform) {
    const int T = (n + 31) >> 5;
    float mx = 0.f;
    if (T > EDGE_WAVES) {
        for (int it = threadIdx.x >> 6; it < (WITH_H ? 2 : 1) * T; it += EDGE_WAVES) {
            const bool h = it >= T;
            fold_load_x(fa, M, o, n, h ? it - T : it, ops);
            fold_load_w(fa, p, h, ops);
            mx = fmaxf(mx, fold_mma(ops, n, h ? it - T : it, h, PQl, Xl, expform));
        }
        return mx;
    }
    if (R.tile < 0) return mx;
    mx = fold_mma(ops, n, R.tile, R.h, PQl, Xl, expform);
    if (WITH_H && R.both) {
        fold_load_w(fa, p, true, ops);
        fold_mma(ops, n, R.tile, true, PQl, Xl, expform);
    }
    return mx;
}
// both halves back to back (re-staging in linear form, un-batched stage-in)
template <bool WITH_H>
__device__ __forceinline__ float fold_fill(const FoldArgs &fa, int64_t M, int64_t o, int n, int p, float2 *PQl, float *Xl,
                                           bool expform) {
    FoldOps ops;
    const FoldRole R = fold_begin<WITH_H>(fa, M, o, n, p, ops);
    return fold_end<WITH_H>(fa, M, o, n, p, R, ops, PQl, Xl, expform);
}

// ------------------------------------------------------------------------------------------
// forward: H_out = H_in + S / (deg + 1e-6).  The last layer also emits the masked node mean, the edge
// mean (= 1/2 sum_v S_v / e: every message is counted at both of its endpoints) and -- fused, while the
// P/Q slices are still in LDS -- the land-use pointer-head inputs of the row's candidate edges.  The head's
// first Linear acts on [m ; c ; m*c ; m-c] (state_encoder.py:207-210, m = the candidate's last-layer message);
// with W1 = [Wa|Wb|Wc|Wd] that is (Wa+Wd) m + Wc (m*c) + (Wb-Wd) c, so only FE = [m ; m*c] is materialised and
// the c-only term becomes a per-row bias.
// ------------------------------------------------------------------------------------------
// HLDS = false (staged P/Q, H left in HBM): the forward needs H only for the residual add at the end of a node's walk, so a
// graph whose slice is too big for two workgroups per CU WITH H (> ~350 nodes: the DHM-sized graphs) still fits without it.
// DMA = the LDS-DMA stage-in (the P/Q tensor carries exp-form flags); a template parameter so that the register-staged
// path and its 24 staging VGPRs do not exist in those instantiations.
template <bool LAST, bool STAGE, bool FOLD, bool HLDS = true, bool DMA = false>
__global__ __launch_bounds__(EDGE_THREADS) __attribute__((amdgpu_num_sgpr(72), amdgpu_waves_per_eu(8, 8))) void edge_fwd_kernel(PackedView pk, MbView mb, int NP,
                                                                const float *__restrict__ PQ,
                                                                const float *__restrict__ bias,
                                                                const float *__restrict__ Hin, float *__restrict__ Hout,
                                                                float *__restrict__ hbarV, float *__restrict__ hbarE,
                                                                const float *__restrict__ Ccur, float *__restrict__ FE,
                                                                int aux_cap, int fit, FoldArgs fa, int fe_full,
                                                                const uint8_t *__restrict__ pqflag, int nfb) {
    static_assert(!FOLD || STAGE, "the folded first layer computes its slice into LDS");
    static_assert(HLDS || STAGE, "H stays in HBM only next to a staged or folded P/Q slice");
    static_assert(!DMA || (STAGE && !FOLD), "the LDS-DMA stage-in belongs to the staged, not folded kernels");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x / NP, p = blockIdx.x % NP;
    const int32_t *m = mb.rows + (int64_t)b * UPAMD_META_STRIDE;      // one scalar load: meta row + minibatch offsets
    const int n = m[0], e = m[1];
    // size classes (see launch_edge_fwd): fit > 0 -> only graphs whose staged slice (with H) needs <= fit bytes of LDS,
    // fit < 0 -> only the larger ones
    if (fit != 0) {
        const int64_t need = edge_lds_bytes(n, 2 * e, false, LAST, true, true);
        if (fit > 0 ? need > fit : need <= -fit) return;
    }
    const int64_t o = m[14], M = mb.M;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ca = 2 * (lane & 7), g = lane >> 3;      // this lane's two columns (ca, ca+1); node slot within the wave
    const EdgeLds L = carve(smem, n, e, STAGE, false, HLDS);

    const float *Pg = PQ + ((int64_t)(2 * p) * M + o) * 16;
    const float *Qg = PQ + ((int64_t)(2 * p + 1) * M + o) * 16;
    const float *Hg = Hin + ((int64_t)p * M + o) * 16;
    float *Ho = Hout + ((int64_t)p * M + o) * 16;
    const float2 bc = make_float2(C2 * bias[p * 16 + ca], C2 * bias[p * 16 + ca + 1]);
    bool ok = false;
    const int32_t *rpg = pk.rowptr + m[13];
    const uint32_t *nbg = reinterpret_cast<const uint32_t *>(pk.inc_nbr + 2 * (int64_t)m[10]);
    const uint16_t *og = pk.order + m[9];
    const uint8_t *nmg = pk.nmask + m[9];
    if (FOLD) {
        // lists first (their loads are in flight while the matrix cores build the slice), then the slice
        const int i0 = tid, i1 = tid + EDGE_THREADS;
        const bool lb = fits_batched(n, e);
        uint32_t nb0 = 0, nb1 = 0, rOrd = 0, rNm = 0;
        int rRp = 0;
        if (lb) {
            nb0 = nbg[i0 < e ? i0 : 0]; nb1 = nbg[i1 < e ? i1 : 0];
            rRp = rpg[tid <= n ? tid : 0];
            rOrd = og[tid < n ? tid : 0]; rNm = nmg[tid < n ? tid : 0];
        }
        FoldOps ops;
        const FoldRole role = fold_begin<true>(fa, M, o, n, p, ops);
        if (lb) {
            if (i0 < e) reinterpret_cast<uint32_t *>(L.nb)[i0] = nb0;
            if (i1 < e) reinterpret_cast<uint32_t *>(L.nb)[i1] = nb1;
            if (tid <= n) L.rp[tid] = rRp;
            if (tid < n) {
                L.ord[tid] = (uint16_t)rOrd;
                L.nm[tid] = (uint8_t)rNm;
            }
        } else {
            for (int i = tid; i <= n; i += EDGE_THREADS) L.rp[i] = rpg[i];
            for (int i = tid; i < e; i += EDGE_THREADS) reinterpret_cast<uint32_t *>(L.nb)[i] = nbg[i];
            for (int i = tid; i < n; i += EDGE_THREADS) {
                L.ord[i] = og[i];
                L.nm[i] = nmg[i];
            }
        }
        // HLDS = false (large size class, round 3): the H_0 tiles go straight to the layer's OUTPUT slice in HBM and a node's row is
        // updated in place at the end of its walk -- 128 instead of 192 bytes of LDS per node, two workgroups per CU for DHM-sized
        // graphs.  (Rows written here are read by other waves of this workgroup only behind the barrier below: workgroup-scope
        // release / acquire, one CU, one L1.)
        const float mx = fold_end<true>(fa, M, o, n, p, role, ops, L.PQ, HLDS ? L.X : Ho, true);
        ok = mx <= EF_LIMIT_FWD && fmaxf(fabsf(bc.x), fabsf(bc.y)) <= EF_BIAS_LIMIT;
    } else if (DMA) {
        // LDS-DMA stage-in: the slices as raw 16-byte chunks, the lists through registers, one round trip for everything
        dma_pq_slice(L.PQ, Pg, Qg, n);
        if (HLDS) dma_x_slice(L.X, Hg, n);
        ListRegs lr;
        lists_load(lr, rpg, nbg, og, nmg, n, e);
        const int bad = dma_flags_bad(pqflag, nfb, o, n, p);
        lists_commit(lr, L, rpg, nbg, og, nmg, n, e);
        wait_vmcnt<0>();                                   // this wave's DMA has landed (the barrier below covers the others')
        ok = !bad && fmaxf(fabsf(bc.x), fabsf(bc.y)) <= EF_BIAS_LIMIT;
    } else if (STAGE && fits_batched(n, e)) {
        const float4 *p4 = reinterpret_cast<const float4 *>(Pg), *q4 = reinterpret_cast<const float4 *>(Qg);
        const float4 *h4 = reinterpret_cast<const float4 *>(Hg);
        // (plain named registers on purpose: with small arrays the compiler parked part of them in scratch memory)
        const int i0 = tid, i1 = tid + EDGE_THREADS;
        const bool in0 = i0 < n * 4, in1 = i1 < n * 4;
        // unconditional loads from clamped (always valid) indices: no divergent control flow between the loads, so
        // they all issue back to back; out-of-range lanes simply do not commit
        const int c0 = in0 ? i0 : 0, c1 = in1 ? i1 : 0;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 p0 = p4[c0], q0 = q4[c0], h0 = HLDS ? h4[c0] : z4, p1 = p4[c1], q1 = q4[c1], h1 = HLDS ? h4[c1] : z4;
        const uint32_t nb0 = nbg[i0 < e ? i0 : 0], nb1 = nbg[i1 < e ? i1 : 0];
        const int rRp = rpg[tid <= n ? tid : 0];
        const uint32_t rOrd = og[tid < n ? tid : 0], rNm = nmg[tid < n ? tid : 0];
        float mx = 0.f;
        if (in0) {
            mx = put_pq_exp(L.PQ, i0, p0, q0);
            if (HLDS) reinterpret_cast<float4 *>(L.X)[i0] = h0;
        }
        if (in1) {
            mx = fmaxf(mx, put_pq_exp(L.PQ, i1, p1, q1));
            if (HLDS) reinterpret_cast<float4 *>(L.X)[i1] = h1;
        }
        if (i0 < e) reinterpret_cast<uint32_t *>(L.nb)[i0] = nb0;
        if (i1 < e) reinterpret_cast<uint32_t *>(L.nb)[i1] = nb1;
        if (tid <= n) L.rp[tid] = rRp;
        if (tid < n) {
            L.ord[tid] = (uint16_t)rOrd;
            L.nm[tid] = (uint8_t)rNm;
        }
        ok = mx <= EF_LIMIT_FWD && fmaxf(fabsf(bc.x), fabsf(bc.y)) <= EF_BIAS_LIMIT;
    } else {
        if (STAGE) {
            ok = stage_pq_exp(L.PQ, Pg, Qg, n, EF_LIMIT_FWD) && fmaxf(fabsf(bc.x), fabsf(bc.y)) <= EF_BIAS_LIMIT;
            const float4 *h4 = reinterpret_cast<const float4 *>(Hg);
            for (int i = tid; HLDS && i < n * 4; i += EDGE_THREADS) reinterpret_cast<float4 *>(L.X)[i] = h4[i];
        }
        for (int i = tid; i <= n; i += EDGE_THREADS) L.rp[i] = rpg[i];
        for (int i = tid; i < e; i += EDGE_THREADS) reinterpret_cast<uint32_t *>(L.nb)[i] = nbg[i];
        for (int i = tid; i < n; i += EDGE_THREADS) {
            L.ord[i] = og[i];
            L.nm[i] = nmg[i];
        }
    }
    // last layer: the row's candidate edges (endpoints, live flag) next to the slice, so the pointer-head pass
    // below does not chase them through global memory
    const int nh = LAST ? m[2] : 0;
    // (decided per WORKGROUP from what its own graph leaves of the launch's dynamic LDS `aux_cap` bytes: a launch is sized for
    // the largest graph of the minibatch, which used to switch the staging off for every graph of it)
    const int64_t aux_avail = (int64_t)aux_cap - (L.aux - smem);
    const int nh8 = (nh + 7) & ~7;
    const bool cand_lds = LAST && FE && nh > 0 && (int64_t)nh8 * 5 <= aux_avail;
    uint16_t *a_src = reinterpret_cast<uint16_t *>(L.aux), *a_dst = a_src + nh8;
    uint8_t *a_live = reinterpret_cast<uint8_t *>(a_dst + nh8);
    if (cand_lds) {
        for (int i = tid; i < nh; i += EDGE_THREADS) {
            a_src[i] = pk.he_src[m[11] + i];
            a_dst[i] = pk.he_dst[m[11] + i];
            a_live[i] = pk.he_live[m[11] + i];
        }
    }
    bool ef = false;                       // LDS holds the exp form (workgroup-uniform)
    if (STAGE) {
        ef = !__syncthreads_or(ok ? 0 : 1);
        if (!ef) {                         // magnitudes outside the exp-form range: linear form, exponentials in the loop
            if (DMA) {
                ef = dma_fixup(L.PQ, pqflag, nfb, o, n, p, EF_LIMIT_FWD, fmaxf(fabsf(bc.x), fabsf(bc.y)) <= EF_BIAS_LIMIT);
            } else {
                if (FOLD) fold_fill<false>(fa, M, o, n, p, L.PQ, L.X, false);
                else stage_pq(L.PQ, Pg, Qg, n);
                __syncthreads();
            }
        }
    } else {
        __syncthreads();
    }

    // where a node's H row is read when it is not staged: the layer input -- or, for the folded layer, the output slice that
    // already holds H_0
    const float *Hsrc = (FOLD && !HLDS) ? Ho : Hg;
    // (P_ca, P_ca+1, Q_ca, Q_ca+1) of node u: exp form or scaled linear form
    auto pq4 = [&](int u) -> float4 {
        if (STAGE) return *reinterpret_cast<const float4 *>(L.PQ + u * 16 + ca);
        const float4 x = *reinterpret_cast<const float4 *>(((lane & 4) ? Qg : Pg) + u * 16 + 4 * (lane & 3));      // pair order
        if (pqflag && !pqflag[((o + u) >> 6) * nfb + (p >> 1)])     // the GEMM stored this block in exp form
            return make_float4(__builtin_amdgcn_logf(x.x), __builtin_amdgcn_logf(x.y), __builtin_amdgcn_logf(x.z), __builtin_amdgcn_logf(x.w));
        return make_float4(C2 * x.x, C2 * x.y, C2 * x.z, C2 * x.w);
    };
    float2 sumS = make_float2(0.f, 0.f), sumH = make_float2(0.f, 0.f);
    auto walk = [&](auto efc) {
        constexpr bool EF = decltype(efc)::value;
        // r1 + r2 of one edge and column: own-side values a1 (P side), a2 (Q side) with the bias folded in, neighbour-side
        // values n1 (its Q), n2 (its P).  Exp form: one reciprocal for both (see EF_LIMIT_FWD).
        auto rsum = [](float a1, float n1, float a2, float n2) -> float {
            if (EF) {
                const float t = fmaf(a1, n1, 1.0f), u = fmaf(a2, n2, 1.0f);
                return (t + u) * __builtin_amdgcn_rcpf(t * u);
            }
            return rcp1p_exp2(a1 + n1) + rcp1p_exp2(a2 + n2);
        };
        const float2 eb = make_float2(EF ? __builtin_amdgcn_exp2f(bc.x) : bc.x, EF ? __builtin_amdgcn_exp2f(bc.y) : bc.y);
        auto fold = [&](float x, float bb) -> float { return EF ? x * bb : x + bb; };
        const int nchunks = (n + 7) >> 3;
        // chunks are dealt to the waves in serpentine order (0..15, 31..16, 32..47, ...): the degree-sorted chunks get
        // shorter and shorter, so plain round-robin would give wave 0 the longest chunk of every round
        for (int rnd = 0, j = w; rnd * EDGE_WAVES < nchunks; ++rnd, j = rnd * EDGE_WAVES + ((rnd & 1) ? EDGE_WAVES - 1 - w : w)) {
            if (j >= nchunks) continue;
            const int vi = 8 * j + g;
            const bool valid = vi < n;
            const int v = L.ord[valid ? vi : n - 1];
            const float4 own = pq4(v);
            const float pv0 = fold(own.x, eb.x), qv0 = fold(own.z, eb.x), pv1 = fold(own.y, eb.y), qv1 = fold(own.w, eb.y);
            int k = L.rp[v];
            const int k1 = valid ? L.rp[v + 1] : k;
            const float degf = (float)(k1 - k);
            // H in HBM: the node's two columns are requested now and consumed after the incidence loop
            float2 hpre = make_float2(0.f, 0.f);
            if (STAGE && !HLDS) hpre = *reinterpret_cast<const float2 *>(Hsrc + v * 16 + ca);
            float acc0 = 0.f, acc1 = 0.f;          // sums over incidences of r1 + r2, per column
            // (one induction variable: the list pointer is also the loop's counter)
            for (const uint16_t *q = L.nb + k, *qe = L.nb + k1; q < qe; ++q) {
                const float4 nb = pq4(*q);
                acc0 += rsum(pv0, nb.z, qv0, nb.x);
                acc1 += rsum(pv1, nb.w, qv1, nb.y);
            }
            if (valid) {
                const float S0 = degf - acc0, S1 = degf - acc1;     // 1/2 sum (tanh1 + tanh2) = 1/2 (2 deg - 2 acc)
                const float inv = __builtin_amdgcn_rcpf(degf + 1e-6f);      // 1-ulp reciprocal instead of two IEEE divisions
                float2 h;
                if (STAGE && HLDS) {
                    float2 *hx = reinterpret_cast<float2 *>(L.X + v * 16 + ca);
                    h = *hx;
                    h.x = fmaf(S0, inv, h.x);
                    h.y = fmaf(S1, inv, h.y);
                    *hx = h;
                } else {
                    h = STAGE ? hpre : *reinterpret_cast<const float2 *>(Hg + v * 16 + ca);
                    h.x = fmaf(S0, inv, h.x);
                    h.y = fmaf(S1, inv, h.y);
                    *reinterpret_cast<float2 *>(Ho + v * 16 + ca) = h;
                }
                if (LAST) {
                    sumS.x += S0;
                    sumS.y += S1;
                    if (L.nm[v]) {
                        sumH.x += h.x;
                        sumH.y += h.y;
                    }
                }
            }
        }
        if (LAST && FE && nh > 0) {
            // pointer-head inputs of this row's candidate edges (8 candidates per wave pass)
            const int64_t NH = mb.Nhe, q0 = m[15];
            const float2 cc = *reinterpret_cast<const float2 *>(Ccur + (int64_t)b * (NP * 16) + p * 16 + ca);
            for (int q = 8 * w + g; q < nh; q += 8 * EDGE_WAVES) {
                float2 mm = make_float2(0.f, 0.f);
                if (cand_lds ? a_live[q] : pk.he_live[m[11] + q]) {
                    const float4 vi4 = pq4(cand_lds ? a_src[q] : pk.he_src[m[11] + q]);
                    const float4 vj4 = pq4(cand_lds ? a_dst[q] : pk.he_dst[m[11] + q]);
                    mm.x = 1.f - rsum(fold(vi4.x, eb.x), vj4.z, fold(vj4.x, eb.x), vi4.z);
                    mm.y = 1.f - rsum(fold(vi4.y, eb.y), vj4.w, fold(vj4.y, eb.y), vi4.w);
                }
                const int64_t row = q0 + q;
                *reinterpret_cast<float2 *>(FE + ((int64_t)p * NH + row) * 16 + ca) = mm;
                if (fe_full)       // (the m*c half is only materialised for head shapes head.hip does not cover)
                    *reinterpret_cast<float2 *>(FE + ((int64_t)(NP + p) * NH + row) * 16 + ca) = make_float2(mm.x * cc.x, mm.y * cc.y);
            }
        }
    };
    if (ef) walk(std::true_type{});
    else walk(std::false_type{});
    if (STAGE && HLDS) {
        __syncthreads();
        float4 *o4 = reinterpret_cast<float4 *>(Ho);
        for (int i = tid; i < n * 4; i += EDGE_THREADS) o4[i] = reinterpret_cast<const float4 *>(L.X)[i];
    }
    if (LAST) {
#pragma unroll
        for (int sft = 8; sft <= 32; sft <<= 1) {
            sumS.x += __shfl_xor(sumS.x, sft); sumS.y += __shfl_xor(sumS.y, sft);
            sumH.x += __shfl_xor(sumH.x, sft); sumH.y += __shfl_xor(sumH.y, sft);
        }
        if (g == 0) {
            *reinterpret_cast<float2 *>(L.red + (w * 2 + 0) * 16 + ca) = sumS;
            *reinterpret_cast<float2 *>(L.red + (w * 2 + 1) * 16 + ca) = sumH;
        }
        __syncthreads();
        if (tid < 32) {
            const int which = tid >> 4, cc = tid & 15;
            float tot = 0.f;
#pragma unroll
            for (int q = 0; q < EDGE_WAVES; ++q) tot += L.red[(q * 2 + which) * 16 + cc];
            const int D = NP * 16;
            if (which == 0) hbarE[(int64_t)b * D + p * 16 + cc] = 0.5f * tot / (float)e;
            else hbarV[(int64_t)b * D + p * 16 + cc] = tot / (float)m[6];
        }
    }
}

static int g_bwd_nb_global = 1;  // tune knob "bwd_nb_global", see edge_bwd_kernel (NBG)
void set_bwd_nb_global(int on) { g_bwd_nb_global = on ? 1 : 0; }
static int g_fwd_h_hbm = 1;      // tune knob "fwd_h_hbm": the large size class of the forward keeps H in HBM (two workgroups per CU)
void set_fwd_h_hbm(int on) { g_fwd_h_hbm = on ? 1 : 0; }

// tune knob fold_layer1 = 2 (lab rule, not the default): fold only when the minibatch's largest graph fits HALF the LDS.  It
// dates from when the folded kernels had no H-in-HBM / list-in-global forms and ran their large size class at one workgroup
// per CU; they have both now (edge_fwd_kernel<.., FOLD, HLDS = false>, edge_bwd_kernel<.., FOLD, .., NBG>): DHM 115.5k ->
// 119.0k samples/s, profiles/archive/r03_lab_fold_two_per_cu.log
bool edge_fold_pays(const MbView &mb) {
    return edge_lds_bytes(mb.max_n, mb.max_inc, false, false, true, true) <= LDS_HALF &&
           edge_lds_bytes(mb.max_n, mb.max_inc, true, false, true, true) <= LDS_HALF;
}
bool edge_fold_ok(const MbView &mb) {
    return edge_lds_bytes(mb.max_n, mb.max_inc, false, false, true, true) <= LDS_LIMIT &&
           edge_lds_bytes(mb.max_n, mb.max_inc, true, false, true, true) <= LDS_LIMIT;
}

int launch_edge_fwd(const PackedView &pk, const MbView &mb, int D, bool last, const float *PQ, const float *bias,
                    const float *Hin, float *Hout, float *hbarV, float *hbarE, const float *Ccur, float *FE,
                    hipStream_t st, Profiler *prof, const FoldArgs *fold, int fe_full, const uint8_t *pqflag) {
    const int NP = D / 16;
    const int nfb = 2 * D / 64;                // 64-column blocks of the P/Q GEMM's output
    if (fold) pqflag = nullptr;                // (a folded layer's P/Q never comes from the GEMM)
    const FoldArgs fa = fold ? *fold : FoldArgs{nullptr, nullptr, nullptr, nullptr, nullptr};
    if (fold && (last || !edge_fold_ok(mb)))
        return fail(UPAMD_E_LIMIT, "edge_fwd: the folded first layer needs the staged size class and a later layer behind it");
    const int began = prof_begin(prof, "edge_fwd", st, 0.0, 0.0);
    dim3 grid((unsigned)(mb.B * NP)), block(EDGE_THREADS);
    // one launch of a given (stage, lds, fit) configuration
    auto go = [&](bool stage, int64_t lds, int aux_cap, int fit, bool hlds = true) -> int {
#define UPAMD_EF(L_, S_, F_, H_, D_)                                                                                  \
    do {                                                                                                              \
        if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void *>(&edge_fwd_kernel<L_, S_, F_, H_, D_>), lds)) return rc_;  \
        hipLaunchKernelGGL((edge_fwd_kernel<L_, S_, F_, H_, D_>), grid, block, (size_t)lds, st, pk, mb, NP, PQ, bias, Hin, \
                           Hout, hbarV, hbarE, Ccur, FE, aux_cap, fit, fa, fe_full, pqflag, nfb);                     \
    } while (0)
        const bool dma = pqflag != nullptr;
        if (fold && !hlds) UPAMD_EF(false, true, true, false, false);
        else if (fold) UPAMD_EF(false, true, true, true, false);
        else if (last && stage && !hlds) { if (dma) UPAMD_EF(true, true, false, false, true); else UPAMD_EF(true, true, false, false, false); }
        else if (last && stage) { if (dma) UPAMD_EF(true, true, false, true, true); else UPAMD_EF(true, true, false, true, false); }
        else if (last) UPAMD_EF(true, false, false, true, false);
        else if (stage && !hlds) { if (dma) UPAMD_EF(false, true, false, false, true); else UPAMD_EF(false, true, false, false, false); }
        else if (stage) { if (dma) UPAMD_EF(false, true, false, true, true); else UPAMD_EF(false, true, false, true, false); }
        else UPAMD_EF(false, false, false, true, false);
#undef UPAMD_EF
        UPAMD_HIP(hipGetLastError());
        return 0;
    };
    const int64_t lds_max = edge_lds_bytes(mb.max_n, mb.max_inc, false, last, true);
    int rc = 0;
    if (lds_max <= LDS_HALF) {
        // every graph of the minibatch fits two workgroups per CU: one launch (+ the row's candidate lists on the last layer)
        // the last layer asks for the whole half: what a graph leaves of it holds its row's candidate lists
        const int64_t lds = (last && FE) ? LDS_HALF : lds_max;
        rc = go(true, lds, (int)lds, 0);
    } else {
        // a few large graphs must not cost every workgroup its neighbour on the CU: the graphs that fit in half the
        // LDS run in a two-per-CU launch, the rest in a second launch -- with H left in HBM if that keeps them at two
        // workgroups per CU (DHM-sized graphs: ~350 .. ~530 nodes), else staged with up to the whole LDS, or un-staged;
        // a workgroup of the wrong class exits at once
        rc = go(true, LDS_HALF, (int)LDS_HALF, (int)LDS_HALF);
        if (rc == 0) {
            const int64_t lds_noh = edge_lds_bytes(mb.max_n, mb.max_inc, false, last, true, false);
            if (g_fwd_h_hbm && lds_noh <= LDS_HALF) {      // (the folded layer too: its H_0 tiles then go straight to the output slice)
                rc = go(true, LDS_HALF, (int)LDS_HALF, -(int)LDS_HALF, false);
            } else if (lds_max <= LDS_LIMIT) {
                rc = go(true, lds_max, (int)lds_max, -(int)LDS_HALF);
            } else {
                const int64_t lds = edge_lds_bytes(mb.max_n, mb.max_inc, false, last, false);
                if (lds > LDS_LIMIT) return fail(UPAMD_E_LIMIT, "edge_fwd: graph too large for LDS (n=%d, 2e=%d)", mb.max_n, mb.max_inc);
                rc = go(false, lds, (int)lds, -(int)LDS_HALF);
            }
        }
    }
    prof_end(prof, "edge_fwd", st, began);
    return rc;
}

// ------------------------------------------------------------------------------------------
// backward w.r.t. P and Q (node-centric, the tanh's are recomputed instead of stored):
//   dS_v = G_v / (deg_v + 1e-6) (+ 1/2 dhbarE / e on the last layer)
//   dm_(v,u) = dS_v + dS_u (+ the pointer-head gradient of that edge on the last layer)
//   dP_v = sum_u 1/2 dm (1 - tanh^2(P_v + Q_u + b)),   dQ_v = sum_u 1/2 dm (1 - tanh^2(P_u + Q_v + b))
// The pointer-head term touches only the row's candidate edges: it is added from the packer's per-node
// candidate-incidence lists after the main walk, so the main loop stays branch-free.
// ------------------------------------------------------------------------------------------
// NBG: the neighbour ids are walked from global memory (L1 / L2 hits: every panel's workgroup of the graph reads the same
// list, a node's ids are consecutive) instead of LDS.  Without the 4 e bytes of the list a DHM-sized graph (up to ~400 nodes)
// fits HALF the LDS with P/Q and dS staged: two workgroups per CU instead of one for the backward's large size class.
template <bool LAST, bool STAGE, bool FOLD, bool DMA = false, bool NBG = false>
__global__ __launch_bounds__(EDGE_THREADS) __attribute__((amdgpu_num_sgpr(72), amdgpu_waves_per_eu(8, 8))) void edge_bwd_kernel(PackedView pk, MbView mb, int NP,
                                                                const float *__restrict__ PQ,
                                                                const float *__restrict__ bias,
                                                                const float *__restrict__ G,
                                                                const float *__restrict__ dhbarE, int ld_dhbarE,
                                                                const float *__restrict__ dMhe, float *__restrict__ dPQ,
                                                                float *__restrict__ dbias_part, int aux_cap, int fit,
                                                                FoldArgs fa, const uint8_t *__restrict__ pqflag, int nfb) {
    static_assert(!FOLD || STAGE, "the folded first layer computes its slice into LDS");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x / NP, p = blockIdx.x % NP;
    const int32_t *m = mb.rows + (int64_t)b * UPAMD_META_STRIDE;      // one scalar load: meta row + minibatch offsets
    const int n = m[0], e = m[1];
    if (fit != 0) {                                    // size classes, see launch_edge_fwd
        const int64_t need = edge_lds_bytes(n, 2 * e, true, LAST, true);
        if (fit > 0 ? need > fit : need <= -fit) return;
    }
    const int64_t o = m[14], M = mb.M;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int ca = 2 * (lane & 7), g = lane >> 3;      // this lane's two columns (ca, ca+1); node slot within the wave
    const EdgeLds L = carve(smem, n, e, STAGE, true, true, !NBG);

    const float *Pg = PQ + ((int64_t)(2 * p) * M + o) * 16;
    const float *Qg = PQ + ((int64_t)(2 * p + 1) * M + o) * 16;
    const float *Gg = G + ((int64_t)p * M + o) * 16;
    const int32_t *rpg = pk.rowptr + m[13];
    const uint32_t *nbg = reinterpret_cast<const uint32_t *>(pk.inc_nbr + 2 * (int64_t)m[10]);
    const uint16_t *og = pk.order + m[9];
    const float2 bc = make_float2(C2 * bias[p * 16 + ca], C2 * bias[p * 16 + ca + 1]);
    static_assert(!DMA || (STAGE && !FOLD), "the LDS-DMA stage-in belongs to the staged, not folded kernels");
    static_assert(!NBG || DMA || FOLD, "the list-in-global walk belongs to the LDS-DMA and the folded kernels");
    constexpr bool dma = DMA;                                  // LDS-DMA stage-in (see dma_pq_slice)
    const bool batched = STAGE && !dma && fits_batched(n, e);
    bool ok = false;
    if (dma) {
        dma_pq_slice(L.PQ, Pg, Qg, n);
        dma_x_slice(L.X, Gg, n);                               // raw G; turned into dS in place below
        ListRegs lr;
        lists_load(lr, rpg, nbg, og, nullptr, n, NBG ? 0 : e);
        // degrees of the G rows whose chunks THIS lane has requested (wave-instruction t = w, w + 16: node 16 t + lane / 4)
        const int v0 = 16 * w + (lane >> 2), v1 = v0 + 16 * EDGE_WAVES;
        const int d00 = rpg[v0 < n ? v0 : 0], d01 = rpg[v0 < n ? v0 + 1 : 0];
        const int d10 = rpg[v1 < n ? v1 : 0], d11 = rpg[v1 < n ? v1 + 1 : 0];
        float4 ex4 = make_float4(0.f, 0.f, 0.f, 0.f);          // the lane's four columns are the same on every trip
        if (LAST) {
            const float4 dh = *reinterpret_cast<const float4 *>(dhbarE + (int64_t)b * ld_dhbarE + p * 16 + (tid & 3) * 4);
            ex4 = make_float4(0.5f * dh.x / (float)e, 0.5f * dh.y / (float)e, 0.5f * dh.z / (float)e, 0.5f * dh.w / (float)e);
        }
        const int bad = dma_flags_bad(pqflag, nfb, o, n, p);
        lists_commit(lr, L, rpg, nbg, og, nullptr, n, NBG ? 0 : e);
        wait_vmcnt<0>();                                       // this wave's DMA has landed: its own chunks may be rewritten
        float4 *x4 = reinterpret_cast<float4 *>(L.X);
        if (v0 < n) {
            const float inv = __builtin_amdgcn_rcpf((float)(d01 - d00) + 1e-6f);
            const float4 g0 = x4[4 * v0 + (lane & 3)];
            x4[4 * v0 + (lane & 3)] = make_float4(fmaf(g0.x, inv, ex4.x), fmaf(g0.y, inv, ex4.y), fmaf(g0.z, inv, ex4.z), fmaf(g0.w, inv, ex4.w));
        }
        if (v1 < n) {
            const float inv = __builtin_amdgcn_rcpf((float)(d11 - d10) + 1e-6f);
            const float4 g1 = x4[4 * v1 + (lane & 3)];
            x4[4 * v1 + (lane & 3)] = make_float4(fmaf(g1.x, inv, ex4.x), fmaf(g1.y, inv, ex4.y), fmaf(g1.z, inv, ex4.z), fmaf(g1.w, inv, ex4.w));
        }
        for (int v = v1 + 16 * EDGE_WAVES; v < n; v += 16 * EDGE_WAVES) {      // graphs above 512 nodes
            const float inv = __builtin_amdgcn_rcpf((float)(rpg[v + 1] - rpg[v]) + 1e-6f);
            const float4 gg = x4[4 * v + (lane & 3)];
            x4[4 * v + (lane & 3)] = make_float4(fmaf(gg.x, inv, ex4.x), fmaf(gg.y, inv, ex4.y), fmaf(gg.z, inv, ex4.z), fmaf(gg.w, inv, ex4.w));
        }
        ok = !bad && fmaxf(fabsf(bc.x), fabsf(bc.y)) <= EF_BIAS_LIMIT;
    } else if (batched) {
        // one memory round trip (see fits_batched): P/Q, G, the degrees of the G rows, the lists -- then commit
        const float4 *p4 = reinterpret_cast<const float4 *>(Pg), *q4 = reinterpret_cast<const float4 *>(Qg);
        const float4 *g4 = reinterpret_cast<const float4 *>(Gg);
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        const int i0 = tid, i1 = tid + EDGE_THREADS;
        const bool in0 = i0 < n * 4, in1 = i1 < n * 4;
        // unconditional loads from clamped (always valid) indices, see edge_fwd_kernel
        const int c0 = in0 ? i0 : 0, c1 = in1 ? i1 : 0;
        const float4 g0 = g4[c0], g1 = g4[c1];
        float4 p0 = z4, q0 = z4, p1 = z4, q1 = z4;
        if (!FOLD) {
            p0 = p4[c0]; q0 = q4[c0]; p1 = p4[c1]; q1 = q4[c1];
        }
        const int d00 = rpg[c0 >> 2], d01 = rpg[(c0 >> 2) + 1];     // row pointers of the G rows (their degrees)
        const int d10 = rpg[c1 >> 2], d11 = rpg[(c1 >> 2) + 1];
        const uint32_t nb0 = nbg[(!NBG && i0 < e) ? i0 : 0], nb1 = nbg[(!NBG && i1 < e) ? i1 : 0];
        const int rRp = rpg[tid <= n ? tid : 0];
        const uint32_t rOrd = og[tid < n ? tid : 0];
        float4 ex4 = z4;                                      // the thread's four columns are the same on both trips
        if (LAST) {
            const float4 dh = *reinterpret_cast<const float4 *>(dhbarE + (int64_t)b * ld_dhbarE + p * 16 + (tid & 3) * 4);
            ex4 = make_float4(0.5f * dh.x / (float)e, 0.5f * dh.y / (float)e, 0.5f * dh.z / (float)e, 0.5f * dh.w / (float)e);
        }
        float mx = 0.f;
        if (FOLD) mx = fold_fill<false>(fa, M, o, n, p, L.PQ, L.X, true);
        if (in0) {
            if (!FOLD) mx = put_pq_exp(L.PQ, i0, p0, q0);
            const float inv = __builtin_amdgcn_rcpf((float)(d01 - d00) + 1e-6f);
            reinterpret_cast<float4 *>(L.X)[i0] = make_float4(fmaf(g0.x, inv, ex4.x), fmaf(g0.y, inv, ex4.y),
                                                              fmaf(g0.z, inv, ex4.z), fmaf(g0.w, inv, ex4.w));
        }
        if (in1) {
            if (!FOLD) mx = fmaxf(mx, put_pq_exp(L.PQ, i1, p1, q1));
            const float inv = __builtin_amdgcn_rcpf((float)(d11 - d10) + 1e-6f);
            reinterpret_cast<float4 *>(L.X)[i1] = make_float4(fmaf(g1.x, inv, ex4.x), fmaf(g1.y, inv, ex4.y),
                                                              fmaf(g1.z, inv, ex4.z), fmaf(g1.w, inv, ex4.w));
        }
        if (!NBG && i0 < e) reinterpret_cast<uint32_t *>(L.nb)[i0] = nb0;
        if (!NBG && i1 < e) reinterpret_cast<uint32_t *>(L.nb)[i1] = nb1;
        if (tid <= n) L.rp[tid] = rRp;
        if (tid < n) L.ord[tid] = (uint16_t)rOrd;
        ok = mx <= EF_LIMIT && fmaxf(fabsf(bc.x), fabsf(bc.y)) <= EF_BIAS_LIMIT;
    } else {
        for (int i = tid; i <= n; i += EDGE_THREADS) L.rp[i] = rpg[i];
        for (int i = tid; !NBG && i < e; i += EDGE_THREADS) reinterpret_cast<uint32_t *>(L.nb)[i] = nbg[i];
        for (int i = tid; i < n; i += EDGE_THREADS) L.ord[i] = og[i];
    }
    // last layer: the per-node candidate-incidence pointers (aux_cap >= 0) and, when they fit, the lists themselves
    // are staged too -- otherwise every node pass would pay a global round trip just to learn it has no candidate
    const bool heads_on = LAST && dMhe != nullptr && m[2] > 0;
    const int32_t *hpg = pk.hinc_ptr + m[13];
    const uint16_t *hnb = pk.hinc_nbr + 2 * (int64_t)m[11];
    const uint16_t *hhe = pk.hinc_he + 2 * (int64_t)m[11];
    // Decided per WORKGROUP from what its own graph leaves of the launch's dynamic LDS (`aux_cap` bytes; the launch is sized for
    // the minibatch's largest graph, which used to switch all of this off for every graph): first the pointers, then the lists,
    // then -- by LDS-DMA -- the row's slice of the pointer-head gradient dM itself, so that the candidate pass below is LDS-only.
    const int nhc = heads_on ? m[2] : 0;
    const int64_t aux_avail = (int64_t)aux_cap - (L.aux - smem);
    const int64_t aux_fixed = a16(((int64_t)n + 1) * 4), aux_lists = a16((int64_t)nhc * 8);
    const bool hp_lds = heads_on && aux_avail >= aux_fixed, hl_lds = hp_lds && aux_avail >= aux_fixed + aux_lists;
    const bool dm_lds = DMA && hl_lds && aux_avail >= aux_fixed + aux_lists + (int64_t)nhc * 64;
    int *a_hp = reinterpret_cast<int *>(L.aux);
    uint16_t *a_hnb = reinterpret_cast<uint16_t *>(L.aux + aux_fixed);
    uint16_t *a_hhe = a_hnb + 2 * nhc;
    float *a_dm = reinterpret_cast<float *>(L.aux + aux_fixed + aux_lists);
    if (DMA && dm_lds) {
        dma_x_slice(a_dm, dMhe + ((int64_t)p * mb.Nhe + m[15]) * 16, nhc);
    }
    if (hp_lds)
        for (int i = tid; i <= n; i += EDGE_THREADS) a_hp[i] = hpg[i];
    if (hl_lds) {
        for (int i = tid; i < m[2]; i += EDGE_THREADS) {       // two entries (both endpoints) per candidate
            reinterpret_cast<uint32_t *>(a_hnb)[i] = reinterpret_cast<const uint32_t *>(hnb)[i];
            reinterpret_cast<uint32_t *>(a_hhe)[i] = reinterpret_cast<const uint32_t *>(hhe)[i];
        }
    }
    if (DMA && dm_lds) wait_vmcnt<0>();    // (the barrier below covers the other waves' parts)
    bool ef = false;                       // LDS holds the exp form (workgroup-uniform), see stage_pq_exp
    if (STAGE) {
        if (!batched && !dma) {
            if (FOLD) ok = fold_fill<false>(fa, M, o, n, p, L.PQ, L.X, true) <= EF_LIMIT && fmaxf(fabsf(bc.x), fabsf(bc.y)) <= EF_BIAS_LIMIT;
            else ok = stage_pq_exp(L.PQ, Pg, Qg, n) && fmaxf(fabsf(bc.x), fabsf(bc.y)) <= EF_BIAS_LIMIT;
        }
        ef = !__syncthreads_or(ok ? 0 : 1);
        if (!ef) {
            if (dma) {
                ef = dma_fixup(L.PQ, pqflag, nfb, o, n, p, EF_LIMIT, fmaxf(fabsf(bc.x), fabsf(bc.y)) <= EF_BIAS_LIMIT);
            } else {
                if (FOLD) fold_fill<false>(fa, M, o, n, p, L.PQ, L.X, false);
                else stage_pq(L.PQ, Pg, Qg, n);
                __syncthreads();
            }
        }
    } else {
        __syncthreads();
    }
    float2 extra = make_float2(0.f, 0.f);     // same for every node of the graph; depends on the lane's columns only
    if (LAST) {
        const float2 dh = *reinterpret_cast<const float2 *>(dhbarE + (int64_t)b * ld_dhbarE + p * 16 + ca);
        extra = make_float2(0.5f * dh.x / (float)e, 0.5f * dh.y / (float)e);
    }
    if (STAGE && !batched && !dma) {
        float4 ex4 = make_float4(0.f, 0.f, 0.f, 0.f);       // the thread's four columns are the same on every trip
        if (LAST) {
            const float4 dh = *reinterpret_cast<const float4 *>(dhbarE + (int64_t)b * ld_dhbarE + p * 16 + (tid & 3) * 4);
            ex4 = make_float4(0.5f * dh.x / (float)e, 0.5f * dh.y / (float)e, 0.5f * dh.z / (float)e, 0.5f * dh.w / (float)e);
        }
        const float4 *g4 = reinterpret_cast<const float4 *>(Gg);
        for (int i = tid; i < n * 4; i += EDGE_THREADS) {
            const int v = i >> 2;
            const float inv = __builtin_amdgcn_rcpf((float)(L.rp[v + 1] - L.rp[v]) + 1e-6f);
            const float4 gg = g4[i];
            reinterpret_cast<float4 *>(L.X)[i] = make_float4(fmaf(gg.x, inv, ex4.x), fmaf(gg.y, inv, ex4.y),
                                                             fmaf(gg.z, inv, ex4.z), fmaf(gg.w, inv, ex4.w));
        }
        __syncthreads();
    }
    auto pq4 = [&](int u) -> float4 {
        if (STAGE) return *reinterpret_cast<const float4 *>(L.PQ + u * 16 + ca);
        const float4 x = *reinterpret_cast<const float4 *>(((lane & 4) ? Qg : Pg) + u * 16 + 4 * (lane & 3));      // pair order
        if (pqflag && !pqflag[((o + u) >> 6) * nfb + (p >> 1)])     // the GEMM stored this block in exp form
            return make_float4(__builtin_amdgcn_logf(x.x), __builtin_amdgcn_logf(x.y), __builtin_amdgcn_logf(x.z), __builtin_amdgcn_logf(x.w));
        return make_float4(C2 * x.x, C2 * x.y, C2 * x.z, C2 * x.w);
    };
    auto ds2 = [&](int u) -> float2 {
        if (STAGE) return *reinterpret_cast<const float2 *>(L.X + u * 16 + ca);
        const float2 gg = *reinterpret_cast<const float2 *>(Gg + u * 16 + ca);
        const float inv = __builtin_amdgcn_rcpf((float)(L.rp[u + 1] - L.rp[u]) + 1e-6f);
        return make_float2(fmaf(gg.x, inv, extra.x), fmaf(gg.y, inv, extra.y));
    };
    const float *dMg = heads_on ? dMhe + ((int64_t)p * mb.Nhe + m[15]) * 16 + ca : nullptr;
    float2 sumdP = make_float2(0.f, 0.f), sumdQ = make_float2(0.f, 0.f);
    auto walk = [&](auto efc) {
        constexpr bool EF = decltype(efc)::value;
        auto r = [](float a, float nb) -> float { return EF ? rcp1p_mul(a, nb) : rcp1p_exp2(a + nb); };
        const float2 eb = make_float2(EF ? __builtin_amdgcn_exp2f(bc.x) : bc.x, EF ? __builtin_amdgcn_exp2f(bc.y) : bc.y);
        auto fold = [&](float x, float bb) -> float { return EF ? x * bb : x + bb; };
        const int nchunks = (n + 7) >> 3;
        // chunks are dealt to the waves in serpentine order (0..15, 31..16, 32..47, ...): the degree-sorted chunks get
        // shorter and shorter, so plain round-robin would give wave 0 the longest chunk of every round
        for (int rnd = 0, j = w; rnd * EDGE_WAVES < nchunks; ++rnd, j = rnd * EDGE_WAVES + ((rnd & 1) ? EDGE_WAVES - 1 - w : w)) {
            if (j >= nchunks) continue;
            const int vi = 8 * j + g;
            const bool valid = vi < n;
            const int v = L.ord[valid ? vi : n - 1];
            int k = L.rp[v];
            const int k1 = valid ? L.rp[v + 1] : k;
            const float4 own = pq4(v);
            const float pv0 = fold(own.x, eb.x), qv0 = fold(own.z, eb.x), pv1 = fold(own.y, eb.y), qv1 = fold(own.w, eb.y);
            const float2 sv = ds2(v);
            // sums of dm * (r - r^2) per column; 1 - tanh^2 = 4 (r - r^2)
            float aP0 = 0.f, aQ0 = 0.f, aP1 = 0.f, aQ1 = 0.f;
            // Exp form: everything NEGATED -- nr = rcp(-(1 + E)) = -r from the negated own-side factors, nr^2 + nr = -(r - r^2) is then
            // a plain FMA, the sums come out as -sum and the sign goes into the final scale.  Bit-identical to r - r^2 summed
            // with the positive sign (negation is exact), and four v_xor_b32 per incidence less: the packed FMAs take no
            // negation modifier, and the walk is bound by its VALU instruction count (profiles/archive/r03_lab_shared_reciprocal.log)
            // The own-side factors are kept as (column ca, column ca + 1) PAIRS: the neighbour's (Q, Q) and (P, P) are register
            // pairs of its ds_read_b128, so t, nr^2 + nr and the sums are two-wide v_pk_fma_f32 each (16 VALU instructions per
            // incidence instead of 22).
            const f32x2e nP = {-pv0, -pv1}, nQ = {-qv0, -qv1}, m1 = {-1.0f, -1.0f};
            f32x2e accP = {0.f, 0.f}, accQ = {0.f, 0.f};
            auto add = [&](const float4 &nb, float dm0, float dm1) {
                if (EF) {
                    const f32x2e tP = __builtin_elementwise_fma(nP, (f32x2e){nb.z, nb.w}, m1);
                    const f32x2e tQ = __builtin_elementwise_fma(nQ, (f32x2e){nb.x, nb.y}, m1);
                    const f32x2e rP = {__builtin_amdgcn_rcpf(tP.x), __builtin_amdgcn_rcpf(tP.y)};
                    const f32x2e rQ = {__builtin_amdgcn_rcpf(tQ.x), __builtin_amdgcn_rcpf(tQ.y)};
                    const f32x2e dm = {dm0, dm1};
                    accP = __builtin_elementwise_fma(dm, __builtin_elementwise_fma(rP, rP, rP), accP);
                    accQ = __builtin_elementwise_fma(dm, __builtin_elementwise_fma(rQ, rQ, rQ), accQ);
                } else {
                    const float r1 = r(pv0, nb.z), r2 = r(qv0, nb.x), r3 = r(pv1, nb.w), r4 = r(qv1, nb.y);
                    aP0 = fmaf(dm0, fmaf(-r1, r1, r1), aP0);
                    aQ0 = fmaf(dm0, fmaf(-r2, r2, r2), aQ0);
                    aP1 = fmaf(dm1, fmaf(-r3, r3, r3), aP1);
                    aQ1 = fmaf(dm1, fmaf(-r4, r4, r4), aQ1);
                }
            };
            const uint16_t *nb16 = reinterpret_cast<const uint16_t *>(nbg);
            int un = NBG ? nb16[k < k1 ? k : 0] : 0;           // (NBG: the next id is requested one trip ahead)
            if (NBG) {
                for (; k < k1; ++k) {
                    const int u = un;
                    un = nb16[k + 1 < k1 ? k + 1 : k];
                    const float2 su = ds2(u);
                    add(pq4(u), sv.x + su.x, sv.y + su.y);
                }
            } else {
                // (one induction variable: the list pointer is also the loop's counter)
                for (const uint16_t *q = L.nb + k, *qe = L.nb + k1; q < qe; ++q) {
                    const int u = *q;
                    const float2 su = ds2(u);
                    add(pq4(u), sv.x + su.x, sv.y + su.y);
                }
            }
            if (heads_on && valid) {
                // candidate gradients live in global memory: fetch HB of them per trip so their latencies overlap;
                // a padding entry has dm = 0 and adds exactly nothing
                constexpr int HB = 3;
                const int hk1 = hp_lds ? a_hp[v + 1] : hpg[v + 1];
                for (int hk = hp_lds ? a_hp[v] : hpg[v]; hk < hk1; hk += HB) {
                    float2 dmh[HB];
                    int uu[HB];
#pragma unroll
                    for (int i = 0; i < HB; ++i) {
                        const bool in = hk + i < hk1;
                        const int kk = in ? hk + i : hk;
                        const int he = hl_lds ? a_hhe[kk] : hhe[kk];
                        uu[i] = hl_lds ? a_hnb[kk] : hnb[kk];
                        dmh[i] = dm_lds ? *reinterpret_cast<const float2 *>(a_dm + he * 16 + ca)
                                        : *reinterpret_cast<const float2 *>(dMg + (int64_t)he * 16);
                        if (!in) dmh[i] = make_float2(0.f, 0.f);
                    }
#pragma unroll
                    for (int i = 0; i < HB; ++i) add(pq4(uu[i]), dmh[i].x, dmh[i].y);
                }
            }
            if (EF) {
                aP0 = accP.x; aP1 = accP.y;
                aQ0 = accQ.x; aQ1 = accQ.y;
            }
            if (valid) {
                constexpr float SC = EF ? -2.f : 2.f;                                     // 1/2 * 4 (exp form: of the negated sums)
                const float2 dP = make_float2(SC * aP0, SC * aP1), dQ = make_float2(SC * aQ0, SC * aQ1);
                // pair order: (dP_ca, dP_ca+1, dQ_ca, dQ_ca+1) is chunk (lane & 3) of the node's row in panel 2p + ((lane >> 2) & 1)
                *reinterpret_cast<float4 *>(dPQ + ((int64_t)(2 * p + ((lane >> 2) & 1)) * M + o + v) * 16 + 4 * (lane & 3)) =
                    make_float4(dP.x, dP.y, dQ.x, dQ.y);
                sumdP.x += dP.x; sumdP.y += dP.y;
                sumdQ.x += dQ.x; sumdQ.y += dQ.y;
            }
        }
    };
    if (ef) walk(std::true_type{});
    else walk(std::false_type{});
    // per-graph column sums of dP and dQ (bias gradient = sum dP; layer 1 also needs sum dQ), P/Q pair order
#pragma unroll
    for (int sft = 8; sft <= 32; sft <<= 1) {
        sumdP.x += __shfl_xor(sumdP.x, sft); sumdP.y += __shfl_xor(sumdP.y, sft);
        sumdQ.x += __shfl_xor(sumdQ.x, sft); sumdQ.y += __shfl_xor(sumdQ.y, sft);
    }
    if (g == 0) {
        *reinterpret_cast<float2 *>(L.red + (w * 2 + 0) * 16 + ca) = sumdP;
        *reinterpret_cast<float2 *>(L.red + (w * 2 + 1) * 16 + ca) = sumdQ;
    }
    __syncthreads();
    if (tid < 32) {
        const int which = tid >> 4, cc = tid & 15;
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < EDGE_WAVES; ++q) tot += L.red[(q * 2 + which) * 16 + cc];
        dbias_part[(int64_t)b * (NP * 32) + p * 32 + pq_pos(which, cc)] = tot;
    }
}

int launch_edge_bwd(const PackedView &pk, const MbView &mb, int D, bool last, const float *PQ, const float *bias,
                    const float *G, const float *dhbarE, int ld_dhbarE, const float *dMhe, float *dPQ,
                    float *dbias_part, hipStream_t st, Profiler *prof, const FoldArgs *fold, const uint8_t *pqflag) {
    const int NP = D / 16;
    const int nfb = 2 * D / 64;
    if (fold) pqflag = nullptr;
    const FoldArgs fa = fold ? *fold : FoldArgs{nullptr, nullptr, nullptr, nullptr, nullptr};
    if (fold && (last || !edge_fold_ok(mb)))
        return fail(UPAMD_E_LIMIT, "edge_bwd: the folded first layer needs the staged size class and a later layer behind it");
    const int began = prof_begin(prof, "edge_bwd", st, 0.0, 0.0);
    dim3 grid((unsigned)(mb.B * NP)), block(EDGE_THREADS);
    auto go = [&](bool stage, int64_t lds, int aux_cap, int fit, bool nbg = false) -> int {
#define UPAMD_EB(L_, S_, F_, D_, N_)                                                                                  \
    do {                                                                                                              \
        if (int rc_ = ensure_dynamic_lds(reinterpret_cast<const void *>(&edge_bwd_kernel<L_, S_, F_, D_, N_>), lds)) return rc_;  \
        hipLaunchKernelGGL((edge_bwd_kernel<L_, S_, F_, D_, N_>), grid, block, (size_t)lds, st, pk, mb, NP, PQ, bias, G,  \
                           dhbarE, ld_dhbarE, dMhe, dPQ, dbias_part, aux_cap, fit, fa, pqflag, nfb);                 \
    } while (0)
        const bool dma = pqflag != nullptr;
        if (nbg) {
            if (fold) UPAMD_EB(false, true, true, false, true);
            else if (last) UPAMD_EB(true, true, false, true, true);
            else UPAMD_EB(false, true, false, true, true);
        } else
        if (fold) UPAMD_EB(false, true, true, false, false);
        else if (last && stage) { if (dma) UPAMD_EB(true, true, false, true, false); else UPAMD_EB(true, true, false, false, false); }
        else if (last) UPAMD_EB(true, false, false, false, false);
        else if (stage) { if (dma) UPAMD_EB(false, true, false, true, false); else UPAMD_EB(false, true, false, false, false); }
        else UPAMD_EB(false, false, false, false, false);
#undef UPAMD_EB
        UPAMD_HIP(hipGetLastError());
        return 0;
    };
    const int64_t lds_max = edge_lds_bytes(mb.max_n, mb.max_inc, true, last, true);
    int rc = 0;
    if (lds_max <= LDS_HALF) {
        // the last layer asks for the whole half: what a graph leaves of it holds its candidate pointers / lists / dM slice
        const int64_t lds = (last && dMhe) ? LDS_HALF : lds_max;
        rc = go(true, lds, (int)lds, 0);
    } else {                                                      // size classes, see launch_edge_fwd
        rc = go(true, LDS_HALF, (int)LDS_HALF, (int)LDS_HALF);
        if (rc == 0) {
            const int64_t lds_nonb = edge_lds_bytes(mb.max_n, mb.max_inc, true, last, true, true, false);
            if ((pqflag || fold) && g_bwd_nb_global && lds_nonb <= LDS_HALF_HARD) {
                rc = go(true, LDS_HALF_HARD, (int)LDS_HALF_HARD, -(int)LDS_HALF, true);      // neighbour ids from global memory: still two per CU
            } else if (lds_max <= LDS_LIMIT) {
                rc = go(true, lds_max, (int)lds_max, -(int)LDS_HALF);
            } else {
                const int64_t lds = edge_lds_bytes(mb.max_n, mb.max_inc, true, last, false);
                if (lds > LDS_LIMIT) return fail(UPAMD_E_LIMIT, "edge_bwd: graph too large for LDS (n=%d, 2e=%d)", mb.max_n, mb.max_inc);
                rc = go(false, lds, (int)lds, -(int)LDS_HALF);
            }
        }
    }
    prof_end(prof, "edge_bwd", st, began);
    return rc;
}

}  // namespace upamd
