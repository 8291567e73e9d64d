// GCN message passing on the ragged batch (gfx950, wave64): forward and backward of one layer.
//
// Reference math (urban_planning/models/state_encoder.py:110-148,194-197): for a live edge (i,j)
//   m_ij = 1/2 [ tanh(W [h_i;h_j] + b) + tanh(W [h_j;h_i] + b) ],   h_v += sum_{e touches v} m_e / (deg_v + 1e-6)
// with W [h_i;h_j] = P_i + Q_j, where P = H Wa^T and Q = H Wb^T (W = [Wa | Wb]) come from the node GEMM.
//
// Work decomposition: one 1024-thread workgroup per (graph, 16-column panel).  In the panel-major
// layout the graph's P, Q and H slices for those 16 columns are contiguous runs of n*64 B; they are
// staged into LDS once with coalesced 16-byte loads (P/Q interleaved so a neighbour costs one
// ds_read_b64), then every gather hits LDS.  A wave handles FOUR nodes at a time -- one per 16-lane
// group, lanes = the 16 columns -- walking that node's incidence list sequentially: a CSR-by-
// destination segment sum in a fixed order, no atomics, no cross-lane reduction, bit-reproducible.
// Nodes are visited in the packer's degree-sorted order so the four lists a wave walks together have
// (nearly) equal length.  16 waves per workgroup x 2 workgroups per CU = the full 8 waves per SIMD.
//
// The kernels are VALU-bound (PMC: SQ_ACTIVE_INST_VALU saturated), so the inner loop is trimmed to the
// transcendental minimum: P and Q are staged PRE-SCALED by 2*log2(e), so with E = 2^(P'_v + Q'_u + b')
//   tanh(x) = 1 - 2 r,  r = 1 / (1 + E)          (v_exp_f32 + v_rcp_f32, abs error ~1e-7, clean saturation)
// the forward only accumulates r (sum of tanh = 2 deg - 2 sum r) and the backward uses
//   1 - tanh^2 = 4 (r - r^2).
#include "kernels.h"

namespace upamd {

#define META(t) (pk.meta + (int64_t)(t) * UPAMD_META_STRIDE)

constexpr int EDGE_THREADS = 1024;
constexpr int EDGE_WAVES = EDGE_THREADS / 64;
constexpr int64_t LDS_LIMIT = 160 * 1024;
constexpr float C2 = 2.8853900817779268f;      // 2 * log2(e)

__device__ __forceinline__ float rcp1p_exp2(float x) {      // 1 / (1 + 2^x)
    return __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x) + 1.0f);
}

static inline int64_t a16(int64_t x) { return (x + 15) / 16 * 16; }

int64_t edge_lds_bytes(int max_n, int max_inc, bool bwd, bool last, bool stage) {
    (void)last;
    int64_t b = 0;
    if (stage) b += (int64_t)max_n * 192;                     // P/Q interleaved (128 B/node) + H or dS (64 B/node)
    b = a16(b);
    b += a16(((int64_t)max_n + 1) * 4);                       // row_ptr
    b += a16((int64_t)max_inc * 2);                           // neighbour ids (u16)
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
};

__device__ __forceinline__ EdgeLds carve(unsigned char *smem, int n, int e, bool stage, bool bwd) {
    EdgeLds L;
    int64_t o = 0;
    L.PQ = reinterpret_cast<float2 *>(smem);
    L.X = reinterpret_cast<float *>(smem + (int64_t)n * 128);
    if (stage) o = (int64_t)n * 192;
    o = (o + 15) / 16 * 16;
    L.rp = reinterpret_cast<int *>(smem + o); o += (((int64_t)n + 1) * 4 + 15) / 16 * 16;
    L.nb = reinterpret_cast<uint16_t *>(smem + o); o += ((int64_t)e * 4 + 15) / 16 * 16;
    L.ord = reinterpret_cast<uint16_t *>(smem + o); o += ((int64_t)n * 2 + 15) / 16 * 16;
    L.nm = reinterpret_cast<uint8_t *>(smem + o);
    if (!bwd) o += ((int64_t)n + 15) / 16 * 16;
    L.red = reinterpret_cast<float *>(smem + o);
    return L;
}

// interleave a graph's P and Q panel slices into LDS, pre-scaled: PQ[v][c] = C2 * (P[v][c], Q[v][c])
__device__ __forceinline__ void stage_pq(float2 *PQl, const float *Pg, const float *Qg, int n) {
    const float4 *p4 = reinterpret_cast<const float4 *>(Pg);
    const float4 *q4 = reinterpret_cast<const float4 *>(Qg);
    for (int i = threadIdx.x; i < n * 4; i += EDGE_THREADS) {
        const float4 pp = p4[i], qq = q4[i];
        float4 *d = reinterpret_cast<float4 *>(PQl + i * 4);       // 4 consecutive (P,Q) pairs = 32 B
        d[0] = make_float4(C2 * pp.x, C2 * qq.x, C2 * pp.y, C2 * qq.y);
        d[1] = make_float4(C2 * pp.z, C2 * qq.z, C2 * pp.w, C2 * qq.w);
    }
}

// ------------------------------------------------------------------------------------------
// forward: H_out = H_in + S / (deg + 1e-6).  The last layer also emits the masked node mean, the edge
// mean (= 1/2 sum_v S_v / e: every message is counted at both of its endpoints) and -- fused, while the
// P/Q slices are still in LDS -- the land-use pointer-head inputs of the row's candidate edges.  The head's
// first Linear acts on [m ; c ; m*c ; m-c] (state_encoder.py:207-210, m = the candidate's last-layer message);
// with W1 = [Wa|Wb|Wc|Wd] that is (Wa+Wd) m + Wc (m*c) + (Wb-Wd) c, so only FE = [m ; m*c] is materialised and
// the c-only term becomes a per-row bias.
// ------------------------------------------------------------------------------------------
template <bool LAST, bool STAGE>
__global__ __launch_bounds__(EDGE_THREADS) void edge_fwd_kernel(PackedView pk, MbView mb, int NP,
                                                                const float *__restrict__ PQ,
                                                                const float *__restrict__ bias,
                                                                const float *__restrict__ Hin, float *__restrict__ Hout,
                                                                float *__restrict__ hbarV, float *__restrict__ hbarE,
                                                                const float *__restrict__ Ccur, float *__restrict__ FE) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x / NP, p = blockIdx.x % NP;
    const int t = mb.idx[b];
    const int32_t *m = META(t);
    const int n = m[0], e = m[1];
    const int64_t o = mb.node_off[b], M = mb.M;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const EdgeLds L = carve(smem, n, e, STAGE, false);

    const float *Pg = PQ + ((int64_t)(2 * p) * M + o) * 16;
    const float *Qg = PQ + ((int64_t)(2 * p + 1) * M + o) * 16;
    const float *Hg = Hin + ((int64_t)p * M + o) * 16;
    float *Ho = Hout + ((int64_t)p * M + o) * 16;
    if (STAGE) {
        stage_pq(L.PQ, Pg, Qg, n);
        const float4 *h4 = reinterpret_cast<const float4 *>(Hg);
        for (int i = tid; i < n * 4; i += EDGE_THREADS) reinterpret_cast<float4 *>(L.X)[i] = h4[i];
    }
    const int32_t *rpg = pk.rowptr + m[13];
    for (int i = tid; i <= n; i += EDGE_THREADS) L.rp[i] = rpg[i];
    const uint32_t *nbg = reinterpret_cast<const uint32_t *>(pk.inc_nbr + 2 * (int64_t)m[10]);
    for (int i = tid; i < e; i += EDGE_THREADS) reinterpret_cast<uint32_t *>(L.nb)[i] = nbg[i];
    const uint16_t *og = pk.order + m[9];
    const uint8_t *nmg = pk.nmask + m[9];
    for (int i = tid; i < n; i += EDGE_THREADS) {
        L.ord[i] = og[i];
        L.nm[i] = nmg[i];
    }
    __syncthreads();

    // scaled (P,Q) of node u for this lane's column
    auto pq = [&](int u) -> float2 {
        if (STAGE) return L.PQ[u * 16 + c];
        return make_float2(C2 * Pg[u * 16 + c], C2 * Qg[u * 16 + c]);
    };
    const float bc = C2 * bias[p * 16 + c];
    float sumS = 0.f, sumH = 0.f;
    const int nchunks = (n + 3) >> 2;
    for (int j = w; j < nchunks; j += EDGE_WAVES) {
        const int vi = 4 * j + g;
        const bool valid = vi < n;
        const int v = L.ord[valid ? vi : n - 1];
        const float2 own = pq(v);
        const float pv = own.x + bc, qv = own.y + bc;
        int k = L.rp[v];
        const int k1 = valid ? L.rp[v + 1] : k;
        const float degf = (float)(k1 - k);
        float accR = 0.f;                  // sum over incidences of r1 + r2
        for (; k < k1; ++k) {
            const float2 nbv = pq(L.nb[k]);
            accR += rcp1p_exp2(pv + nbv.y) + rcp1p_exp2(nbv.x + qv);
        }
        if (valid) {
            const float S = degf - accR;   // 1/2 sum (tanh1 + tanh2) = 1/2 (2 deg - 2 accR)
            const float a = S / (degf + 1e-6f);
            float h;
            if (STAGE) {
                h = L.X[v * 16 + c] + a;
                L.X[v * 16 + c] = h;
            } else {
                h = Hg[v * 16 + c] + a;
                Ho[v * 16 + c] = h;
            }
            if (LAST) {
                sumS += S;
                if (L.nm[v]) sumH += h;
            }
        }
    }
    if (LAST && FE && m[2] > 0) {
        // pointer-head inputs of this row's candidate edges (4 candidates per wave pass)
        const int nh = m[2];
        const int64_t NH = mb.Nhe, q0 = mb.he_off[b];
        const float cc = Ccur[(int64_t)b * (NP * 16) + p * 16 + c];
        for (int q = 4 * w + g; q < nh; q += 4 * EDGE_WAVES) {
            float mm = 0.f;
            if (pk.he_live[m[11] + q]) {
                const float2 vi2 = pq(pk.he_src[m[11] + q]), vj2 = pq(pk.he_dst[m[11] + q]);
                mm = 1.f - (rcp1p_exp2(vi2.x + vj2.y + bc) + rcp1p_exp2(vj2.x + vi2.y + bc));
            }
            const int64_t row = q0 + q;
            FE[((int64_t)p * NH + row) * 16 + c] = mm;
            FE[((int64_t)(NP + p) * NH + row) * 16 + c] = mm * cc;
        }
    }
    if (STAGE) {
        __syncthreads();
        float4 *o4 = reinterpret_cast<float4 *>(Ho);
        for (int i = tid; i < n * 4; i += EDGE_THREADS) o4[i] = reinterpret_cast<const float4 *>(L.X)[i];
    }
    if (LAST) {
        sumS += __shfl_xor(sumS, 16); sumS += __shfl_xor(sumS, 32);
        sumH += __shfl_xor(sumH, 16); sumH += __shfl_xor(sumH, 32);
        if (g == 0) {
            L.red[(w * 2 + 0) * 16 + c] = sumS;
            L.red[(w * 2 + 1) * 16 + c] = sumH;
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

int launch_edge_fwd(const PackedView &pk, const MbView &mb, int D, bool last, const float *PQ, const float *bias,
                    const float *Hin, float *Hout, float *hbarV, float *hbarE, const float *Ccur, float *FE,
                    hipStream_t st, Profiler *prof) {
    const int NP = D / 16;
    bool stage = true;
    int64_t lds = edge_lds_bytes(mb.max_n, mb.max_inc, false, last, true);
    if (lds > LDS_LIMIT) {
        stage = false;
        lds = edge_lds_bytes(mb.max_n, mb.max_inc, false, last, false);
        if (lds > LDS_LIMIT) return fail(UPAMD_E_LIMIT, "edge_fwd: graph too large for LDS (n=%d, 2e=%d)", mb.max_n, mb.max_inc);
    }
    const int began = prof_begin(prof, "edge_fwd", st, 0.0, 0.0);
    dim3 grid((unsigned)(mb.B * NP)), block(EDGE_THREADS);
#define UPAMD_EF(L_, S_)                                                                                              \
    do {                                                                                                              \
        if (lds > 64 * 1024)                                                                                          \
            UPAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&edge_fwd_kernel<L_, S_>),                   \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                     \
        hipLaunchKernelGGL((edge_fwd_kernel<L_, S_>), grid, block, (size_t)lds, st, pk, mb, NP, PQ, bias, Hin, Hout,  \
                           hbarV, hbarE, Ccur, FE);                                                                   \
    } while (0)
    if (last && stage) UPAMD_EF(true, true);
    else if (last) UPAMD_EF(true, false);
    else if (stage) UPAMD_EF(false, true);
    else UPAMD_EF(false, false);
#undef UPAMD_EF
    prof_end(prof, "edge_fwd", st, began);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// backward w.r.t. P and Q (node-centric, the tanh's are recomputed instead of stored):
//   dS_v = G_v / (deg_v + 1e-6) (+ 1/2 dhbarE / e on the last layer)
//   dm_(v,u) = dS_v + dS_u (+ the pointer-head gradient of that edge on the last layer)
//   dP_v = sum_u 1/2 dm (1 - tanh^2(P_v + Q_u + b)),   dQ_v = sum_u 1/2 dm (1 - tanh^2(P_u + Q_v + b))
// The pointer-head term touches only the row's candidate edges: it is added from the packer's per-node
// candidate-incidence lists after the main walk, so the main loop stays branch-free.
// ------------------------------------------------------------------------------------------
template <bool LAST, bool STAGE>
__global__ __launch_bounds__(EDGE_THREADS) void edge_bwd_kernel(PackedView pk, MbView mb, int NP,
                                                                const float *__restrict__ PQ,
                                                                const float *__restrict__ bias,
                                                                const float *__restrict__ G,
                                                                const float *__restrict__ dhbarE, int ld_dhbarE,
                                                                const float *__restrict__ dMhe, float *__restrict__ dPQ,
                                                                float *__restrict__ dbias_part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x / NP, p = blockIdx.x % NP;
    const int t = mb.idx[b];
    const int32_t *m = META(t);
    const int n = m[0], e = m[1];
    const int64_t o = mb.node_off[b], M = mb.M;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const EdgeLds L = carve(smem, n, e, STAGE, true);

    const float *Pg = PQ + ((int64_t)(2 * p) * M + o) * 16;
    const float *Qg = PQ + ((int64_t)(2 * p + 1) * M + o) * 16;
    const float *Gg = G + ((int64_t)p * M + o) * 16;
    const int32_t *rpg = pk.rowptr + m[13];
    for (int i = tid; i <= n; i += EDGE_THREADS) L.rp[i] = rpg[i];
    const uint32_t *nbg = reinterpret_cast<const uint32_t *>(pk.inc_nbr + 2 * (int64_t)m[10]);
    for (int i = tid; i < e; i += EDGE_THREADS) reinterpret_cast<uint32_t *>(L.nb)[i] = nbg[i];
    const uint16_t *og = pk.order + m[9];
    for (int i = tid; i < n; i += EDGE_THREADS) L.ord[i] = og[i];
    if (STAGE) stage_pq(L.PQ, Pg, Qg, n);
    __syncthreads();
    float extra = 0.f;     // same for every node of the graph; depends on the lane's column only
    if (LAST) extra = 0.5f * dhbarE[(int64_t)b * ld_dhbarE + p * 16 + c] / (float)e;
    if (STAGE) {
        for (int i = tid; i < n * 16; i += EDGE_THREADS) {
            const int v = i >> 4;
            float ex = 0.f;
            if (LAST) ex = 0.5f * dhbarE[(int64_t)b * ld_dhbarE + p * 16 + (i & 15)] / (float)e;
            L.X[i] = Gg[i] / ((float)(L.rp[v + 1] - L.rp[v]) + 1e-6f) + ex;
        }
        __syncthreads();
    }
    auto pq = [&](int u) -> float2 {
        if (STAGE) return L.PQ[u * 16 + c];
        return make_float2(C2 * Pg[u * 16 + c], C2 * Qg[u * 16 + c]);
    };
    auto ds = [&](int u) -> float {
        if (STAGE) return L.X[u * 16 + c];
        return Gg[u * 16 + c] / ((float)(L.rp[u + 1] - L.rp[u]) + 1e-6f) + extra;
    };
    const float bc = C2 * bias[p * 16 + c];
    const bool heads_on = LAST && dMhe != nullptr && m[2] > 0;
    const int32_t *hpg = pk.hinc_ptr + m[13];
    const uint16_t *hnb = pk.hinc_nbr + 2 * (int64_t)m[11];
    const uint16_t *hhe = pk.hinc_he + 2 * (int64_t)m[11];
    const float *dMg = heads_on ? dMhe + ((int64_t)p * mb.Nhe + mb.he_off[b]) * 16 + c : nullptr;
    float sumdP = 0.f, sumdQ = 0.f;
    const int nchunks = (n + 3) >> 2;
    for (int j = w; j < nchunks; j += EDGE_WAVES) {
        const int vi = 4 * j + g;
        const bool valid = vi < n;
        const int v = L.ord[valid ? vi : n - 1];
        int k = L.rp[v];
        const int k1 = valid ? L.rp[v + 1] : k;
        const float2 own = pq(v);
        const float pv = own.x + bc, qv = own.y + bc, sv = ds(v);
        float accP = 0.f, accQ = 0.f;     // sums of dm * (r - r^2); 1 - tanh^2 = 4 (r - r^2)
        for (; k < k1; ++k) {
            const int u = L.nb[k];
            const float2 nbv = pq(u);
            const float dm = sv + ds(u);
            const float r1 = rcp1p_exp2(pv + nbv.y), r2 = rcp1p_exp2(nbv.x + qv);
            accP = fmaf(dm, fmaf(-r1, r1, r1), accP);
            accQ = fmaf(dm, fmaf(-r2, r2, r2), accQ);
        }
        if (heads_on && valid) {
            for (int hk = hpg[v]; hk < hpg[v + 1]; ++hk) {
                const float2 nbv = pq(hnb[hk]);
                const float dmh = dMg[(int64_t)hhe[hk] * 16];
                const float r1 = rcp1p_exp2(pv + nbv.y), r2 = rcp1p_exp2(nbv.x + qv);
                accP = fmaf(dmh, fmaf(-r1, r1, r1), accP);
                accQ = fmaf(dmh, fmaf(-r2, r2, r2), accQ);
            }
        }
        if (valid) {
            const float dP = 2.f * accP, dQ = 2.f * accQ;       // 1/2 * 4
            dPQ[((int64_t)(2 * p) * M + o + v) * 16 + c] = dP;
            dPQ[((int64_t)(2 * p + 1) * M + o + v) * 16 + c] = dQ;
            sumdP += dP;
            sumdQ += dQ;
        }
    }
    // per-graph column sums of dP and dQ (bias gradient = sum dP; layer 1 also needs sum dQ), P/Q panel order
    sumdP += __shfl_xor(sumdP, 16); sumdP += __shfl_xor(sumdP, 32);
    sumdQ += __shfl_xor(sumdQ, 16); sumdQ += __shfl_xor(sumdQ, 32);
    if (g == 0) {
        L.red[(w * 2 + 0) * 16 + c] = sumdP;
        L.red[(w * 2 + 1) * 16 + c] = sumdQ;
    }
    __syncthreads();
    if (tid < 32) {
        const int which = tid >> 4, cc = tid & 15;
        float tot = 0.f;
#pragma unroll
        for (int q = 0; q < EDGE_WAVES; ++q) tot += L.red[(q * 2 + which) * 16 + cc];
        dbias_part[(int64_t)b * (NP * 32) + (2 * p + which) * 16 + cc] = tot;
    }
}

int launch_edge_bwd(const PackedView &pk, const MbView &mb, int D, bool last, const float *PQ, const float *bias,
                    const float *G, const float *dhbarE, int ld_dhbarE, const float *dMhe, float *dPQ,
                    float *dbias_part, hipStream_t st, Profiler *prof) {
    const int NP = D / 16;
    bool stage = true;
    int64_t lds = edge_lds_bytes(mb.max_n, mb.max_inc, true, last, true);
    if (lds > LDS_LIMIT) {
        stage = false;
        lds = edge_lds_bytes(mb.max_n, mb.max_inc, true, last, false);
        if (lds > LDS_LIMIT) return fail(UPAMD_E_LIMIT, "edge_bwd: graph too large for LDS (n=%d, 2e=%d)", mb.max_n, mb.max_inc);
    }
    const int began = prof_begin(prof, "edge_bwd", st, 0.0, 0.0);
    dim3 grid((unsigned)(mb.B * NP)), block(EDGE_THREADS);
#define UPAMD_EB(L_, S_)                                                                                              \
    do {                                                                                                              \
        if (lds > 64 * 1024)                                                                                          \
            UPAMD_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&edge_bwd_kernel<L_, S_>),                   \
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                     \
        hipLaunchKernelGGL((edge_bwd_kernel<L_, S_>), grid, block, (size_t)lds, st, pk, mb, NP, PQ, bias, G, dhbarE,  \
                           ld_dhbarE, dMhe, dPQ, dbias_part);                                                         \
    } while (0)
    if (last && stage) UPAMD_EB(true, true);
    else if (last) UPAMD_EB(true, false);
    else if (stage) UPAMD_EB(false, true);
    else UPAMD_EB(false, false);
#undef UPAMD_EB
    prof_end(prof, "edge_bwd", st, began);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

}  // namespace upamd
