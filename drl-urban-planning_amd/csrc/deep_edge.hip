// Edge MLPs with more than one sub-layer (num_edge_fc_layers = K > 1, urban_planning/models/state_encoder.py:59-82,110-130).
//
// With K = 1 the edge MLP factorises onto the nodes (edge.hip) and no per-edge tensor exists.  Behind a second Linear that
// is no longer possible: the sub-layers k = 1 .. K-1 act on one row per EDGE DIRECTION.  The packer's incidence CSR already
// lists every live edge once at each endpoint, so "incidence k of node v with neighbour u" IS the direction v -> u:
//   A_1[k]   = tanh(P_v + Q_u + b_0)                      (P | Q = H Wcat_0^T from the node GEMM, as for K = 1)
//   A_j+1[k] = tanh(A_j[k] W_j^T + b_j),  j = 1 .. K-1     (plain GEMMs over the NI = sum 2e incidence rows, gemm.hip)
//   m_e      = 1/2 (A_K[k] + A_K[rev k])                   (rev k = the same edge seen from the other endpoint)
//   S_v      = sum_{k in inc(v)} m_e(k),   H_out = H_in + S_v / (deg_v + 1e-6)
// Per-incidence tensors are panel-major [D/16][NI][16] like every other per-row tensor.  All sums run over a node's
// incidence list in list order: fixed order, no atomics, bit-reproducible.  No shipped configuration uses K > 1, so these
// kernels are written for clarity (one thread per (row, column) / (node, column)), not tuned like edge.hip.
#include "kernels.h"

namespace upamd {

#define META(t) (pk.meta + (int64_t)(t) * UPAMD_META_STRIDE)

namespace {

__device__ __forceinline__ float tanh_fast(float x) {      // same form as the GEMM epilogue's (abs error ~1e-7)
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}

// sum over the 16 row groups of a 256-thread workgroup (thread = (column c = tid & 15, group tid >> 4)); result valid
// in threads 0 .. 15 (one per column)
__device__ __forceinline__ float colgroup_sum(float v, float *part) {
    __syncthreads();
    part[threadIdx.x] = v;
    __syncthreads();
    float tot = 0.f;
    if (threadIdx.x < 16) {
#pragma unroll
        for (int q = 0; q < 16; ++q) tot += part[q * 16 + threadIdx.x];
    }
    return tot;
}

// ------------------------------------------------------------------------------------------
// Index tables of a minibatch (one workgroup per graph): for every incidence its endpoints as minibatch-wide node rows,
// the incidence of the opposite direction, and for every land-use candidate an incidence of its edge (-1: not a live edge).
// rev: the t-th (v -> u) entry of v's list is paired with the t-th (u -> v) entry of u's list (duplicate edges carry
// identical values, any bijection is right; a self-loop's two entries pair with themselves).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void inc_index_kernel(PackedView pk, MbView mb, int32_t *__restrict__ gsrc,
                                                        int32_t *__restrict__ gdst, int32_t *__restrict__ grev,
                                                        int32_t *__restrict__ cand_inc) {
    const int b = blockIdx.x, t = mb.idx[b];
    const int32_t *m = META(t);
    const int n = m[0], nh = m[2];
    const int32_t *rp = pk.rowptr + m[13];
    const uint16_t *nb = pk.inc_nbr + 2 * (int64_t)m[10];
    const int64_t o = mb.node_off[b], io = mb.inc_off[b];
    for (int v = threadIdx.x; v < n; v += 256) {
        const int k0 = rp[v], k1 = rp[v + 1];
        for (int k = k0; k < k1; ++k) {
            const int u = nb[k];
            int rank = 0;
            for (int j = k0; j < k; ++j) rank += (nb[j] == u) ? 1 : 0;
            int r = k, cnt = 0;                       // (a malformed list would pair the entry with itself)
            const int j1 = rp[u + 1];
            for (int j = rp[u]; j < j1; ++j) {
                if (nb[j] == v) {
                    if (cnt == rank) { r = j; break; }
                    ++cnt;
                }
            }
            gsrc[io + k] = (int32_t)(o + v);
            gdst[io + k] = (int32_t)(o + u);
            grev[io + k] = (int32_t)(io + r);
        }
    }
    const int64_t q0 = mb.he_off[b];
    for (int q = threadIdx.x; q < nh; q += 256) {
        int ci = -1;
        if (pk.he_live[m[11] + q]) {
            const int s = pk.he_src[m[11] + q], d = pk.he_dst[m[11] + q];
            const int j1 = rp[s + 1];
            for (int j = rp[s]; j < j1; ++j)
                if (nb[j] == d) { ci = (int)(io + j); break; }
        }
        cand_inc[q0 + q] = ci;
    }
}

// A_1[k] = tanh(P[src k] + Q[dst k] + b_0): one thread per (panel, incidence, column quad)
__global__ __launch_bounds__(256) void inc_gather_fwd_kernel(const float *__restrict__ PQ, const float *__restrict__ bias,
                                                             const int32_t *__restrict__ gsrc, const int32_t *__restrict__ gdst,
                                                             float *__restrict__ A1, int64_t M, int64_t NI, int NP) {
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= (int64_t)NP * NI * 4) return;
    const int cq = (int)(g & 3);
    const int64_t k = (g >> 2) % NI;
    const int p = (int)((g >> 2) / NI);
    // P/Q pair order (kernels.h): columns 4 cq .. 4 cq + 3 = the pairs 2 cq, 2 cq + 1 = chunks (2 cq) & 3, + 1 of the node's
    // row in panel 2 p + (cq >> 1); a chunk is (P_c, P_c+1, Q_c, Q_c+1)
    const int64_t pan = (int64_t)(2 * p + (cq >> 1)) * M;
    const int ch = 8 * (cq & 1);
    const float4 s0 = *reinterpret_cast<const float4 *>(PQ + (pan + gsrc[k]) * 16 + ch);
    const float4 s1 = *reinterpret_cast<const float4 *>(PQ + (pan + gsrc[k]) * 16 + ch + 4);
    const float4 d0 = *reinterpret_cast<const float4 *>(PQ + (pan + gdst[k]) * 16 + ch);
    const float4 d1 = *reinterpret_cast<const float4 *>(PQ + (pan + gdst[k]) * 16 + ch + 4);
    const float4 pp = make_float4(s0.x, s0.y, s1.x, s1.y), qq = make_float4(d0.z, d0.w, d1.z, d1.w);
    const float4 bb = *reinterpret_cast<const float4 *>(bias + p * 16 + 4 * cq);
    *reinterpret_cast<float4 *>(A1 + ((int64_t)p * NI + k) * 16 + 4 * cq) =
        make_float4(tanh_fast(pp.x + qq.x + bb.x), tanh_fast(pp.y + qq.y + bb.y), tanh_fast(pp.z + qq.z + bb.z),
                    tanh_fast(pp.w + qq.w + bb.w));
}

// H_out = H_in + S / (deg + 1e-6), S_v = 1/2 sum_{k in inc(v)} (A_K[k] + A_K[rev k]).  The last layer also emits the
// masked node mean, the edge mean (1/2 sum_v S_v / e) and the land-use head inputs FE = [m ; m*c] of the row's
// candidates.  One workgroup per (graph, 16-column panel); thread = (column, one of 16 node groups).
template <bool LAST>
__global__ __launch_bounds__(256) void inc_scatter_fwd_kernel(PackedView pk, MbView mb, int NP, const float *__restrict__ AK,
                                                              const int32_t *__restrict__ grev,
                                                              const int32_t *__restrict__ cand_inc,
                                                              const float *__restrict__ Hin, float *__restrict__ Hout,
                                                              float *__restrict__ hbarV, float *__restrict__ hbarE,
                                                              const float *__restrict__ Ccur, float *__restrict__ FE, int fe_full) {
    __shared__ float part[256];
    const int b = blockIdx.x / NP, p = blockIdx.x % NP, t = mb.idx[b];
    const int32_t *m = META(t);
    const int n = m[0], e = m[1];
    const int32_t *rp = pk.rowptr + m[13];
    const uint8_t *nm = pk.nmask + m[9];
    const int64_t o = mb.node_off[b], io = mb.inc_off[b], M = mb.M, NI = mb.NI;
    const int c = threadIdx.x & 15, vg = threadIdx.x >> 4;
    const float *Ap = AK + (int64_t)p * NI * 16 + c;
    float sumS = 0.f, sumH = 0.f;
    for (int v = vg; v < n; v += 16) {
        const int k0 = rp[v], k1 = rp[v + 1];
        float acc = 0.f;
        for (int k = k0; k < k1; ++k) acc += Ap[(io + k) * 16] + Ap[(int64_t)grev[io + k] * 16];
        const float S = 0.5f * acc;
        const int64_t hi = ((int64_t)p * M + o + v) * 16 + c;
        const float h = fmaf(S, __builtin_amdgcn_rcpf((float)(k1 - k0) + 1e-6f), Hin[hi]);
        Hout[hi] = h;
        if (LAST) {
            sumS += S;
            if (nm[v]) sumH += h;
        }
    }
    if (!LAST) return;
    const int D = NP * 16;
    const float totS = colgroup_sum(sumS, part);
    const float totH = colgroup_sum(sumH, part);
    if (threadIdx.x < 16) {
        hbarE[(int64_t)b * D + p * 16 + c] = 0.5f * totS / (float)e;
        hbarV[(int64_t)b * D + p * 16 + c] = totH / (float)m[6];
    }
    const int nh = m[2];
    if (FE && nh > 0) {
        const int64_t NH = mb.Nhe, q0 = mb.he_off[b];
        const float cc = Ccur[(int64_t)b * D + p * 16 + c];
        for (int q = vg; q < nh; q += 16) {
            const int ci = cand_inc[q0 + q];
            const float mm = ci >= 0 ? 0.5f * (Ap[(int64_t)ci * 16] + Ap[(int64_t)grev[ci] * 16]) : 0.f;
            FE[((int64_t)p * NH + q0 + q) * 16 + c] = mm;
            if (fe_full) FE[((int64_t)(NP + p) * NH + q0 + q) * 16 + c] = mm * cc;
        }
    }
}

// Backward seed at the top of the edge MLP:  dS_v = G_v / (deg_v + 1e-6) (+ 1/2 dhbarE / e on the last layer),
//   dA_K[k = v -> u] = 1/2 (dS_v + dS_u + [the pointer-head gradient of that edge on the last layer]),
//   dpre_K = dA_K (1 - A_K^2);  dbpart[b] = its per-graph column sums (the last sub-layer's bias gradient).
// The t-th (v -> u) incidence takes the t-th candidate of v's candidate-incidence list whose neighbour is u, so every
// live candidate is counted exactly once at each of its endpoints (the packer lists it at both).
template <bool LAST>
__global__ __launch_bounds__(256) void inc_seed_bwd_kernel(PackedView pk, MbView mb, int NP, const float *__restrict__ AK,
                                                           const float *__restrict__ G, const float *__restrict__ dhbarE,
                                                           int ld_dhbarE, const float *__restrict__ dMhe,
                                                           float *__restrict__ dpre, float *__restrict__ dbpart) {
    __shared__ float part[256];
    const int b = blockIdx.x / NP, p = blockIdx.x % NP, t = mb.idx[b];
    const int32_t *m = META(t);
    const int n = m[0], e = m[1];
    const int32_t *rp = pk.rowptr + m[13];
    const uint16_t *nb = pk.inc_nbr + 2 * (int64_t)m[10];
    const int32_t *hp = pk.hinc_ptr + m[13];
    const uint16_t *hnb = pk.hinc_nbr + 2 * (int64_t)m[11];
    const uint16_t *hhe = pk.hinc_he + 2 * (int64_t)m[11];
    const int64_t o = mb.node_off[b], io = mb.inc_off[b], M = mb.M, NI = mb.NI, NH = mb.Nhe, q0 = mb.he_off[b];
    const int c = threadIdx.x & 15, vg = threadIdx.x >> 4;
    const bool heads_on = LAST && dMhe != nullptr && m[2] > 0;
    const float extra = LAST ? 0.5f * dhbarE[(int64_t)b * ld_dhbarE + p * 16 + c] / (float)e : 0.f;
    const float *Gp = G + ((int64_t)p * M + o) * 16 + c;
    const float *Ap = AK + ((int64_t)p * NI + io) * 16 + c;
    float *Dp = dpre + ((int64_t)p * NI + io) * 16 + c;
    float accb = 0.f;
    for (int v = vg; v < n; v += 16) {
        const int k0 = rp[v], k1 = rp[v + 1];
        const float dSv = fmaf(Gp[(int64_t)v * 16], __builtin_amdgcn_rcpf((float)(k1 - k0) + 1e-6f), extra);
        for (int k = k0; k < k1; ++k) {
            const int u = nb[k];
            const float dSu = fmaf(Gp[(int64_t)u * 16], __builtin_amdgcn_rcpf((float)(rp[u + 1] - rp[u]) + 1e-6f), extra);
            float dm = dSv + dSu;
            if (heads_on) {
                int rank = 0;
                for (int j = k0; j < k; ++j) rank += (nb[j] == u) ? 1 : 0;
                int cnt = 0;
                const int j1 = hp[v + 1];
                for (int j = hp[v]; j < j1; ++j) {
                    if (hnb[j] == u) {
                        if (cnt == rank) { dm += dMhe[((int64_t)p * NH + q0 + hhe[j]) * 16 + c]; break; }
                        ++cnt;
                    }
                }
            }
            const float a = Ap[(int64_t)k * 16];
            const float d = 0.5f * dm * fmaf(-a, a, 1.0f);
            Dp[(int64_t)k * 16] = d;
            accb += d;
        }
    }
    const float tot = colgroup_sum(accb, part);
    if (threadIdx.x < 16 && dbpart) dbpart[(int64_t)b * (NP * 16) + p * 16 + c] = tot;
}

// dpre = dA (1 - A^2) in place over a graph's incidence rows; dbpart[b] (optional) = its per-graph column sums
__global__ __launch_bounds__(256) void inc_tanh_bwd_kernel(PackedView pk, MbView mb, int NP, const float *__restrict__ A,
                                                           float *__restrict__ dA, float *__restrict__ dbpart) {
    __shared__ float part[256];
    const int b = blockIdx.x / NP, p = blockIdx.x % NP, t = mb.idx[b];
    const int ninc = 2 * META(t)[1];
    const int64_t io = mb.inc_off[b], NI = mb.NI;
    const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const float *Ap = A + ((int64_t)p * NI + io) * 16 + c;
    float *Dp = dA + ((int64_t)p * NI + io) * 16 + c;
    float accb = 0.f;
    for (int r = rg; r < ninc; r += 16) {
        const float a = Ap[(int64_t)r * 16];
        const float d = Dp[(int64_t)r * 16] * fmaf(-a, a, 1.0f);
        Dp[(int64_t)r * 16] = d;
        accb += d;
    }
    const float tot = colgroup_sum(accb, part);
    if (threadIdx.x < 16 && dbpart) dbpart[(int64_t)b * (NP * 16) + p * 16 + c] = tot;
}

// dP_v = sum_{k in inc(v)} dpre_1[k],  dQ_v = sum_{k in inc(v)} dpre_1[rev k]  (dpre_1[k = v -> u] feeds P_v and Q_u),
// written in the P/Q pair order of the node GEMM's output, + the per-graph column sums of dP | dQ (edge.hip's format)
__global__ __launch_bounds__(256) void inc_scatter_bwd_kernel(PackedView pk, MbView mb, int NP, const float *__restrict__ dpre1,
                                                              const int32_t *__restrict__ grev, float *__restrict__ dPQ,
                                                              float *__restrict__ dbias_part) {
    __shared__ float part[256];
    const int b = blockIdx.x / NP, p = blockIdx.x % NP, t = mb.idx[b];
    const int32_t *m = META(t);
    const int n = m[0];
    const int32_t *rp = pk.rowptr + m[13];
    const int64_t o = mb.node_off[b], io = mb.inc_off[b], M = mb.M, NI = mb.NI;
    const int c = threadIdx.x & 15, vg = threadIdx.x >> 4;
    const float *Dp = dpre1 + (int64_t)p * NI * 16 + c;
    float sP = 0.f, sQ = 0.f;
    for (int v = vg; v < n; v += 16) {
        const int k0 = rp[v], k1 = rp[v + 1];
        float dP = 0.f, dQ = 0.f;
        for (int k = k0; k < k1; ++k) {
            dP += Dp[(io + k) * 16];
            dQ += Dp[(int64_t)grev[io + k] * 16];
        }
        const int pos = pq_pos(0, c);           // position of P column c inside the pair's 32 floats; its Q is 2 further
        float *row = dPQ + ((int64_t)(2 * p + (pos >> 4)) * M + o + v) * 16 + (pos & 15);
        row[0] = dP;
        row[2] = dQ;
        sP += dP;
        sQ += dQ;
    }
    const float tP = colgroup_sum(sP, part);
    const float tQ = colgroup_sum(sQ, part);
    if (threadIdx.x < 16) {
        dbias_part[(int64_t)b * (NP * 32) + p * 32 + pq_pos(0, c)] = tP;
        dbias_part[(int64_t)b * (NP * 32) + p * 32 + pq_pos(1, c)] = tQ;
    }
}

}  // namespace

int launch_inc_index(const PackedView &pk, const MbView &mb, int32_t *gsrc, int32_t *gdst, int32_t *grev, int32_t *cand_inc,
                     hipStream_t st) {
    if (!mb.inc_off) return fail(UPAMD_E_INVALID, "num_edge_fc_layers > 1 needs the minibatch's incidence offsets (inc_off_dev, n_inc)");
    hipLaunchKernelGGL(inc_index_kernel, dim3(mb.B), dim3(256), 0, st, pk, mb, gsrc, gdst, grev, cand_inc);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

int launch_inc_gather_fwd(const MbView &mb, int D, const float *PQ, const float *bias, const int32_t *gsrc, const int32_t *gdst,
                          float *A1, hipStream_t st) {
    if (mb.NI <= 0) return 0;
    const int NP = D / 16;
    const int64_t total = (int64_t)NP * mb.NI * 4;
    hipLaunchKernelGGL(inc_gather_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, PQ, bias, gsrc, gdst, A1, mb.M,
                       mb.NI, NP);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

int launch_inc_scatter_fwd(const PackedView &pk, const MbView &mb, int D, bool last, const float *AK, const int32_t *grev,
                           const int32_t *cand_inc, const float *Hin, float *Hout, float *hbarV, float *hbarE,
                           const float *Ccur, float *FE, hipStream_t st, int fe_full) {
    const int NP = D / 16;
    if (last)
        hipLaunchKernelGGL(inc_scatter_fwd_kernel<true>, dim3(mb.B * NP), dim3(256), 0, st, pk, mb, NP, AK, grev, cand_inc, Hin, Hout,
                           hbarV, hbarE, Ccur, FE, fe_full);
    else
        hipLaunchKernelGGL(inc_scatter_fwd_kernel<false>, dim3(mb.B * NP), dim3(256), 0, st, pk, mb, NP, AK, grev, cand_inc, Hin, Hout,
                           hbarV, hbarE, Ccur, FE, fe_full);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

int launch_inc_seed_bwd(const PackedView &pk, const MbView &mb, int D, bool last, const float *AK, const float *G,
                        const float *dhbarE, int ld_dhbarE, const float *dMhe, float *dpre, float *dbpart, hipStream_t st) {
    const int NP = D / 16;
    if (last)
        hipLaunchKernelGGL(inc_seed_bwd_kernel<true>, dim3(mb.B * NP), dim3(256), 0, st, pk, mb, NP, AK, G, dhbarE, ld_dhbarE, dMhe,
                           dpre, dbpart);
    else
        hipLaunchKernelGGL(inc_seed_bwd_kernel<false>, dim3(mb.B * NP), dim3(256), 0, st, pk, mb, NP, AK, G, dhbarE, ld_dhbarE, dMhe,
                           dpre, dbpart);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

int launch_inc_tanh_bwd(const PackedView &pk, const MbView &mb, int D, const float *A, float *dA, float *dbpart, hipStream_t st) {
    hipLaunchKernelGGL(inc_tanh_bwd_kernel, dim3(mb.B * (D / 16)), dim3(256), 0, st, pk, mb, D / 16, A, dA, dbpart);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

int launch_inc_scatter_bwd(const PackedView &pk, const MbView &mb, int D, const float *dpre1, const int32_t *grev, float *dPQ,
                           float *dbias_part, hipStream_t st) {
    hipLaunchKernelGGL(inc_scatter_bwd_kernel, dim3(mb.B * (D / 16)), dim3(256), 0, st, pk, mb, D / 16, dpre1, grev, dPQ, dbias_part);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

}  // namespace upamd
