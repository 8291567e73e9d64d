// Fused small-model path (gcn_node_dim <= 32 -- the dims every shipped YAML uses, hlg.yaml:21-33): ONE workgroup owns one
// graph and runs the whole network on it out of LDS -- numerical encoder, node encoder, every GCN layer, masked means,
// single-query attention, value head, pointer head, the PPO loss seeds and the complete backward -- so an optimizer step is
// three launches (this kernel, the slab reduction, Adam) instead of ~35 dependent ones.
//
// Reference math: urban_planning/models/state_encoder.py:84-214 (encoder), policy.py:19-104 (pointer heads, log-prob /
// entropy), value.py:15-39 (value head), urban_planning_agent.py:326-333,363-371 + khrylib/rl/agents/agent_pg.py:19-23
// (loss); the hand-derived backward is the one of tests/csr_model.py, stage by stage.
//
// This header holds the PER-GRAPH program and nothing target specific: tiny.hip instantiates it as a HIP kernel; the test
// infrastructure (tests/tiny_emul.cpp, TINY_HOST) compiles the very same text with g++ and runs it on the CPU against the
// oracle before any GPU time is spent.  The contract that makes that possible:
//   * all parallelism is "for every index i of a range, independent iterations", written T_FOR(i, N) and closed by T_SYNC()
//     (a workgroup barrier on the GPU, nothing on the host where the range runs sequentially) -- no wave intrinsics, no
//     atomics, no thread-private state that survives a T_SYNC();
//   * an iteration writes only elements it owns; every sum is a serial loop in a fixed order inside one iteration (two-level
//     sums: fixed partial groups, then a fixed combine) => bit-reproducible run to run, like the large-model kernels;
//   * gradients of the parameters go to the workgroup's own slab in global memory (`+=` by the owning iteration); the slabs
//     are added in a fixed order by the reduction launch.
#pragma once
#include <stdint.h>

#ifdef TINY_HOST
#include <math.h>
#define TDEV static inline
#define TMEM inline
#define THD static inline
#define T_FOR(i, N) for (int i = 0; i < (N); ++i)
#define T_SYNC() \
    do {         \
    } while (0)
#define T_TID0 1
static inline float t_exp2(float x) { return exp2f(x); }
static inline float t_rcp(float x) { return 1.0f / x; }
static inline float t_log(float x) { return logf(x); }
#else
#define TDEV __device__ __forceinline__
#define TMEM __device__ __forceinline__
#define THD __host__ __device__ inline
#define T_FOR(i, N) for (int i = (int)threadIdx.x; i < (N); i += (int)blockDim.x)
#define T_SYNC() __syncthreads()
#define T_TID0 (threadIdx.x == 0)
TDEV float t_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
TDEV float t_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
TDEV float t_log(float x) { return __logf(x); }
#endif

namespace upamd_tiny {

constexpr int META = 16;          // UPAMD_META_STRIDE
constexpr int XPAD = 24;          // UPAMD_NODE_PAD
constexpr int MAXMLP = 4;
constexpr int MAXL = 16;
constexpr int NG = 32;            // partial groups of the two-level sums
constexpr int CH = 32;            // pointer-head candidates per chunk
constexpr float C2 = 2.8853900817779268f;     // 2 log2(e)
constexpr float LOG2E = 1.4426950408889634f;
constexpr float PAD_LOGIT = -4294967296.0f;   // -2^32 + 1 in fp32 (policy.py:50,59)

struct Dims {
    int D, L, heads, F, Fn;
    int n_num, num_hidden[MAXMLP];
    int n_value, value_hidden[MAXMLP];
    int h0l, h0r;
    int S_last, W;                // width of the last numerical layer; width of state_value = 3 D + S_last + 3
};
struct Offs {                     // float offsets into the flat parameter buffer (the gradient slab has the same layout)
    int num_w[MAXMLP], num_b[MAXMLP], node_w, node_b, edge_w[MAXL], edge_b[MAXL];
    int inproj_w, inproj_b, outproj_w, outproj_b, q_w, q_b, k_w, k_b, v_w, v_b;
    int value_w[MAXMLP], value_b[MAXMLP], land_w0, land_b0, land_w1, road_w0, road_b0, road_w1;
    int n_floats;
};
enum Mode { FWD = 0, BWD = 1, STEP = 2 };

struct Args {
    // packed replay (upamd_pack_layout sections)
    const int32_t *meta;
    const float *X;
    const uint8_t *nmask;
    const int32_t *rowptr;
    const uint16_t *inc_nbr, *he_src, *he_dst, *rn_node, *hinc_nbr, *hinc_he;
    const int32_t *hinc_ptr;
    const uint8_t *he_live;
    const float *numerical, *cur;
    // minibatch
    int B;
    const int32_t *idx, *he_off, *rn_off;
    Dims d;
    Offs o;
    const float *prm;
    int mode;
    float *value, *logp, *ent;                    // [B]
    float *z_he, *z_rn;                           // optional: candidate logits in minibatch order (action heads)
    const float *dvalue, *dlogp, *dent;           // BWD: seeds
    const int64_t *rows;                          // STEP: replay row of minibatch row b (may be null: b itself)
    const float *adv, *ret, *old_logp, *exps;     // STEP: whole-replay arrays
    float clip_eps, cv, ce, inv_rows, inv_ind;
    float *loss_rows;                             // STEP: [B][4] = (value - ret)^2, min(s1, s2), entropy, 0 of the row
    float *slab;                                  // [G][slab_stride] gradient slabs, one per workgroup
    int64_t slab_stride;
    float *scratch;                               // [G][scratch_stride] per-workgroup global scratch (dM of the candidates)
    int64_t scratch_stride;
    int max_n, max_inc;
};

TDEV float t_tanh(float x) { return 1.0f - 2.0f * t_rcp(t_exp2(C2 * x) + 1.0f); }
TDEV float t_exp(float x) { return t_exp2(x * LOG2E); }
THD int imax(int a, int b) { return a > b ? a : b; }
THD int64_t a4(int64_t x) { return (x + 3) / 4 * 4; }

// ---- LDS plan (floats).  n / inc = the LARGEST graph of the launch: one plan per launch, every graph uses its prefix.
struct Plan {
    int64_t H, X, PQ, rp, nb, alpha, sc, vec, total;      // offsets
    int64_t xsize, vsize;
};
THD int64_t vec_floats(const Dims &d) {
    int64_t u = d.Fn, v = 0;
    for (int i = 0; i < d.n_num; ++i) u += d.num_hidden[i];
    for (int i = 0; i < d.n_value; ++i) v += d.value_hidden[i];
    const int64_t D = d.D, Hd = (int64_t)d.heads * d.D, h0 = imax(d.h0l, d.h0r);
    //     U        cur   C..dC (16 D-vectors)   head vectors (10)   SV, dSV     V, dV      A, M (h0 x D)  const, s, w2..   partials        scalars
    return u + XPAD + 16 * D + 10 * Hd + 2 * a4(d.W) + 2 * a4(v) + 2 * h0 * D + 8 * h0 + (int64_t)NG * imax((int)Hd, 64) + 64 + 256;
}
THD Plan make_plan(const Dims &d, int n, int inc) {
    Plan p;
    const int64_t nD = (int64_t)n * d.D;
    int64_t o = 0;
    p.H = o; o += a4((int64_t)d.L * nD);                                           // H^1 .. H^L ([n][D] each; slot L becomes G)
    p.xsize = imax((int)nD, inc / 2 + imax(n, 1) + CH * (d.D + 2 * imax(d.h0l, d.h0r)));
    p.X = o; o += a4(p.xsize);                                                     // S / dS | head scratch (z, chunk buffers)
    p.PQ = o; o += a4(2 * nD);                                                     // P | Q of a layer; backward: half | d(half)
    p.rp = o; o += a4(n + 1);
    p.nb = o; o += a4((inc + 1) / 2);                                              // u16 neighbour ids
    p.alpha = o; o += a4((int64_t)d.heads * n);
    p.sc = o; o += a4((int64_t)d.heads * n);
    p.vsize = vec_floats(d);
    p.vec = o; o += a4(p.vsize);
    p.total = o;
    return p;
}

// bump allocator over the vec region
struct Bump {
    float *base;
    int64_t used;
    TMEM float *get(int64_t n) {
        float *r = base + used;
        used += a4(n);
        return r;
    }
};

// out[j] = act(bias[j] + sum_k W[j * K + k] * in[k]);  weights from global memory, vectors in LDS (tiny per-sample layers)
TDEV void lin(float *out, const float *in, const float *W, const float *bias, int N, int K, int act, float scale = 1.0f) {
    T_FOR(j, N) {
        float acc = bias ? bias[j] : 0.0f;
        const float *w = W + (int64_t)j * K;
        for (int k = 0; k < K; ++k) acc = fmaf(w[k], in[k], acc);
        acc *= scale;
        out[j] = act ? t_tanh(acc) : acc;
    }
    T_SYNC();
}
// out[k] = sum_j W[j * K + k] * in[j]   (W^T in)
TDEV void lin_t(float *out, const float *in, const float *W, int N, int K, int j0 = 0, int j1 = -1) {
    if (j1 < 0) j1 = N;
    T_FOR(k, K) {
        float acc = 0.0f;
        for (int j = j0; j < j1; ++j) acc = fmaf(W[(int64_t)j * K + k], in[j], acc);
        out[k] = acc;
    }
    T_SYNC();
}
// slab[(j, k)] += a[j] * x[k]  (rank-1 weight gradient of a per-sample layer), and slab_b[j] += a[j]
TDEV void outer_acc(float *gw, float *gb, const float *a, const float *x, int N, int K) {
    T_FOR(i, N * K) {
        const int j = i / K, k = i - j * K;
        gw[i] += a[j] * x[k];
    }
    if (gb) {
        T_FOR(j, N) gb[j] += a[j];
    }
    T_SYNC();
}

// dst[c] = scale * sum_{v < N} f(v, c), c < C: NG fixed partial groups, then a fixed combine
template <class F>
TDEV void colsum(float *dst, float *part, int N, int C, float scale, F f) {
    T_FOR(i, NG * C) {
        const int g = i / C, c = i - g * C;
        float acc = 0.0f;
        for (int v = g; v < N; v += NG) acc += f(v, c);
        part[i] = acc;
    }
    T_SYNC();
    T_FOR(c, C) {
        float acc = 0.0f;
        for (int g = 0; g < NG; ++g) acc += part[g * C + c];
        dst[c] = acc * scale;
    }
    T_SYNC();
}

// ======================================================================================================================
template <int D>
TDEV void graph_program(const Args &A, int b, float *slab, float *gscr, float *lds, const Plan &pl) {
    const Dims &d = A.d;
    const Offs &o = A.o;
    const float *prm = A.prm;
    const int t = A.idx[b];
    const int32_t *m = A.meta + (int64_t)t * META;
    const int n = m[0], e = m[1], stage = m[4], act = m[5];
    const int nc = stage == 0 ? m[2] : (stage == 1 ? m[3] : 0);      // candidates of the row's pointer head
    const int64_t node_off = m[9];
    const float *Xg = A.X + node_off * XPAD;
    const uint8_t *nmg = A.nmask + node_off;
    const int32_t *rpg = A.rowptr + m[13];
    const uint16_t *nbg = A.inc_nbr + 2 * (int64_t)m[10];
    const uint16_t *hsrc = A.he_src + m[11], *hdst = A.he_dst + m[11];
    const uint8_t *hlive = A.he_live + m[11];
    const uint16_t *rnn = A.rn_node + m[12];
    const int32_t *hpg = A.hinc_ptr + m[13];
    const uint16_t *hnb = A.hinc_nbr + 2 * (int64_t)m[11], *hhe = A.hinc_he + 2 * (int64_t)m[11];
    const int L = d.L, Hn = d.heads, dh = D / Hn, inc = 2 * e;
    const int nD = n * D;
    const bool land = stage == 0 && nc > 0, road = stage == 1 && nc > 0;
    const bool bwd = A.mode != FWD;

    float *Hs = lds + pl.H;                      // slot l (1..L) at Hs + (l - 1) * nD
    float *Xr = lds + pl.X;
    float *PQ = lds + pl.PQ;                     // [n][2D]: P columns 0..D-1, Q columns D..2D-1
    int *rp = reinterpret_cast<int *>(lds + pl.rp);
    uint16_t *nb = reinterpret_cast<uint16_t *>(lds + pl.nb);
    float *alpha = lds + pl.alpha, *sc = lds + pl.sc;
    Bump vb{lds + pl.vec, 0};
    auto slotH = [&](int l) -> float * { return Hs + (int64_t)(l - 1) * nD; };

    // ---- vec region
    float *U[MAXMLP + 1];
    U[0] = vb.get(d.Fn);
    for (int i = 0; i < d.n_num; ++i) U[i + 1] = vb.get(d.num_hidden[i]);
    float *cur = vb.get(XPAD), *C = vb.get(D), *q0 = vb.get(D), *q1 = vb.get(D), *ov = vb.get(D), *att = vb.get(D);
    float *hbarV = vb.get(D), *hbarE = vb.get(D);
    float *tk = vb.get(Hn * D), *rr = vb.get(Hn * D), *ss = vb.get(Hn * D), *uu = vb.get(Hn * D);
    float *SV = vb.get(d.W);
    float *V[MAXMLP + 1];
    V[0] = SV;
    for (int i = 0; i < d.n_value; ++i) V[i + 1] = vb.get(d.value_hidden[i]);
    const int h0 = land ? d.h0l : d.h0r;
    float *Aeff = vb.get((int64_t)imax(d.h0l, d.h0r) * D);       // land: (Wa + Wd) + Wc diag(C);  road: R1
    float *cst = vb.get(imax(d.h0l, d.h0r));                    // land: b1 + (Wb - Wd) C;        road: rb1
    float *w2v = vb.get(imax(d.h0l, d.h0r));
    float *part = vb.get((int64_t)NG * imax(Hn * D, 64));
    float *scal = vb.get(64);      // 0 mx, 1 lse, 2 ent, 3 logp, 4.. softmax scratch per head (mx, sum), 16 dvalue 17 dlogp 18 dent
    // backward-only vectors
    float *dSV = vb.get(d.W), *dC = vb.get(D), *dq0 = vb.get(D), *dq1 = vb.get(D), *dov = vb.get(D), *datt_unused = vb.get(D);
    (void)datt_unused;
    float *du = vb.get(Hn * D), *ds = vb.get(Hn * D), *dr = vb.get(Hn * D), *dtk = vb.get(Hn * D);
    float *dVa = vb.get(64), *dVb = vb.get(64);      // ping-pong of the small MLP backward (hidden <= 64)
    float *Mj = vb.get((int64_t)imax(d.h0l, d.h0r) * D), *sj = vb.get(imax(d.h0l, d.h0r)), *dw2 = vb.get(imax(d.h0l, d.h0r));

    // =============================================================================== forward
    // lists + per-sample inputs
    T_FOR(i, n + 1) rp[i] = rpg[i];
    T_FOR(i, inc) nb[i] = nbg[i];
    T_FOR(i, d.Fn) U[0][i] = A.numerical[(int64_t)t * d.Fn + i];
    T_FOR(i, XPAD) cur[i] = A.cur[(int64_t)t * XPAD + i];
    T_SYNC();
    // numerical encoder (state_encoder.py:35-57,187)
    {
        int K = d.Fn;
        for (int i = 0; i < d.n_num; ++i) {
            lin(U[i + 1], U[i], prm + o.num_w[i], prm + o.num_b[i], d.num_hidden[i], K, 1);
            K = d.num_hidden[i];
        }
    }
    // current node through the node encoder (:191), attention query path (:150-156 + MultiheadAttention's q projection)
    lin(C, cur, prm + o.node_w, prm + o.node_b, D, d.F, 0);
    lin(q0, C, prm + o.q_w, prm + o.q_b, D, D, 0);
    const float scale = 1.0f / sqrtf((float)dh);
    lin(q1, q0, prm + o.inproj_w, prm + o.inproj_b, D, D, 0, scale);
    // r_h = Wk^T (Wik[head rows]^T q1[head rows]):  score_j = r_h . h_j   (key-side biases are softmax-shift invariant)
    for (int h = 0; h < Hn; ++h) lin_t(tk + h * D, q1, prm + o.inproj_w + (int64_t)D * D, D, D, h * dh, (h + 1) * dh);
    for (int h = 0; h < Hn; ++h) lin_t(rr + h * D, tk + h * D, prm + o.k_w, D, D);

    // node encoder on every node (:189-190): H^0 into slot 1 (layer 1 updates it in place)
    const float *We = prm + o.node_w, *be = prm + o.node_b;
    auto encode_nodes = [&](float *dst) {
        T_FOR(i, nD) {
            const int v = i / D, c = i - v * D;
            const float *x = Xg + (int64_t)v * XPAD, *w = We + (int64_t)c * d.F;
            float acc = be[c];
            for (int f = 0; f < d.F; ++f) acc = fmaf(w[f], x[f], acc);
            dst[i] = acc;
        }
        T_SYNC();
    };
    // P | Q of layer l from Hin: PQ[v][j] = sum_k Wl[j % D][(j / D) * D + k] Hin[v][k]   (linear_0.weight is [D][2D] = [Wa | Wb])
    auto pq_full = [&](int l, const float *Hin) {
        const float *Wl = prm + o.edge_w[l - 1];
        T_FOR(i, n * 2 * D) {
            const int v = i / (2 * D), j = i - v * 2 * D;
            const float *w = Wl + (int64_t)(j % D) * (2 * D) + (j / D) * D, *h = Hin + (int64_t)v * D;
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < D; ++k) acc = fmaf(w[k], h[k], acc);
            PQ[i] = acc;
        }
        T_SYNC();
    };
    encode_nodes(slotH(1));
    for (int l = 1; l <= L; ++l) {
        const float *Hin = l == 1 ? slotH(1) : slotH(l - 1);
        float *Hout = slotH(l);
        const float *bl = prm + o.edge_b[l - 1];
        const bool last = l == L;
        pq_full(l, Hin);
        // node-centric segment sum (:110-148): S_v = sum over incidences 1/2 [tanh(P_v + Q_u + b) + tanh(P_u + Q_v + b)]
        T_FOR(i, nD) {
            const int v = i / D, c = i - v * D;
            const float pv = PQ[v * 2 * D + c] + bl[c], qv = PQ[v * 2 * D + D + c] + bl[c];
            const int k0 = rp[v], k1 = rp[v + 1];
            float S = 0.0f;
            for (int k = k0; k < k1; ++k) {
                const int u = nb[k];
                S += 0.5f * (t_tanh(pv + PQ[u * 2 * D + D + c]) + t_tanh(PQ[u * 2 * D + c] + qv));
            }
            Hout[i] = Hin[i] + S / ((float)(k1 - k0) + 1e-6f);
            if (last) Xr[i] = S;
        }
        T_SYNC();
    }
    float *HL = slotH(L);
    // masked node mean, edge mean (:179-182,199-200; every message is summed at both endpoints)
    colsum(hbarV, part, n, D, 1.0f / (float)m[6], [&](int v, int c) { return nmg[v] ? HL[v * D + c] : 0.0f; });
    colsum(hbarE, part, n, D, 0.5f / (float)e, [&](int v, int c) { return Xr[v * D + c]; });
    // single-query attention over the node_mask nodes (:150-161)
    for (int h = 0; h < Hn; ++h) {
        float *sch = sc + (int64_t)h * n, *al = alpha + (int64_t)h * n;
        const float *r = rr + h * D;
        T_FOR(v, n) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < D; ++k) acc = fmaf(r[k], HL[v * D + k], acc);
            sch[v] = nmg[v] ? acc : -INFINITY;
        }
        T_SYNC();
        T_FOR(g, NG) {
            float mx = -INFINITY;
            for (int v = g; v < n; v += NG) mx = fmaxf(mx, sch[v]);
            part[g] = mx;
        }
        T_SYNC();
        if (T_TID0) {
            float mx = -INFINITY;
            for (int g = 0; g < NG; ++g) mx = fmaxf(mx, part[g]);
            scal[4] = mx;
        }
        T_SYNC();
        T_FOR(v, n) al[v] = nmg[v] ? t_exp(sch[v] - scal[4]) : 0.0f;
        T_SYNC();
        T_FOR(g, NG) {
            float sum = 0.0f;
            for (int v = g; v < n; v += NG) sum += al[v];
            part[g] = sum;
        }
        T_SYNC();
        if (T_TID0) {
            float sum = 0.0f;
            for (int g = 0; g < NG; ++g) sum += part[g];
            scal[5] = 1.0f / sum;
        }
        T_SYNC();
        T_FOR(v, n) al[v] *= scal[5];
        T_SYNC();
        colsum(ss + h * D, part, n, D, 1.0f, [&](int v, int c) { return al[v] * HL[v * D + c]; });
        // u_h = Wv s_h + bv;   o[head rows] = Wiv[head rows] u_h + biv[head rows]
        lin(uu + h * D, ss + h * D, prm + o.v_w, prm + o.v_b, D, D, 0);
    }
    T_FOR(i, D) {
        const float *w = prm + o.inproj_w + (int64_t)(2 * D + i) * D, *u = uu + (i / dh) * D;
        float acc = prm[o.inproj_b + 2 * D + i];
        for (int k = 0; k < D; ++k) acc = fmaf(w[k], u[k], acc);
        ov[i] = acc;
    }
    T_SYNC();
    lin(att, ov, prm + o.outproj_w, prm + o.outproj_b, D, D, 0);
    // state_value = [h_num | mean nodes | mean edges | attended current node | stage] (:204-205), value head (value.py:15-39)
    T_FOR(i, d.W) {
        float v;
        if (i < d.S_last) v = U[d.n_num][i];
        else if (i < d.S_last + D) v = hbarV[i - d.S_last];
        else if (i < d.S_last + 2 * D) v = hbarE[i - d.S_last - D];
        else if (i < d.S_last + 3 * D) v = att[i - d.S_last - 2 * D];
        else v = (i - d.S_last - 3 * D) == stage ? 1.0f : 0.0f;
        SV[i] = v;
    }
    T_SYNC();
    {
        int K = d.W;
        for (int i = 0; i < d.n_value; ++i) {
            lin(V[i + 1], V[i], prm + o.value_w[i], prm + o.value_b[i], d.value_hidden[i], K, i < d.n_value - 1);
            K = d.value_hidden[i];
        }
    }
    // ---- pointer head of the row's stage (policy.py:45-104); candidates only (a masked slot has probability exactly 0)
    float *z = Xr;                               // [nc] logits
    float *chunk = Xr + a4(imax(nc, 1));         // [CH][D] m (or XR), [CH][h0] hid, [CH][h0] dpre
    const float *PQl = PQ;                       // last layer's P | Q is still in place (the forward wrote it last)
    const float *blL = prm + o.edge_b[L - 1];
    // candidate inputs of a chunk: land = the candidate edge's last-layer message m (0 if not a live edge), road = its node's H^L row
    auto cand_inputs = [&](int q0c, int cn, float *mq) {
        T_FOR(i, cn * D) {
            const int q = q0c + i / D, c = i % D;
            float val;
            if (land) {
                val = 0.0f;
                if (hlive[q]) {
                    const int vi = hsrc[q], vj = hdst[q];
                    val = 0.5f * (t_tanh(PQl[vi * 2 * D + c] + PQl[vj * 2 * D + D + c] + blL[c]) +
                                  t_tanh(PQl[vj * 2 * D + c] + PQl[vi * 2 * D + D + c] + blL[c]));
                }
            } else {
                val = HL[rnn[q] * D + c];
            }
            mq[i] = val;
        }
        T_SYNC();
    };
    auto cand_hidden = [&](int cn, const float *mq, float *hid) {
        T_FOR(i, cn * h0) {
            const int q = i / h0, j = i - q * h0;
            const float *a = Aeff + (int64_t)j * D, *x = mq + (int64_t)q * D;
            float acc = cst[j];
#pragma unroll
            for (int c = 0; c < D; ++c) acc = fmaf(a[c], x[c], acc);
            hid[i] = t_tanh(acc);
        }
        T_SYNC();
    };
    if (land || road) {
        if (land) {
            // W1 [m; c; m*c; m-c] = ((Wa + Wd) + Wc diag(c)) m + (Wb - Wd) c: per-graph effective weight + per-graph bias
            const float *W1 = prm + o.land_w0;
            T_FOR(i, h0 * D) {
                const int j = i / D, c = i - j * D;
                const float *w = W1 + (int64_t)j * 4 * D;
                Aeff[i] = w[c] + w[3 * D + c] + w[2 * D + c] * C[c];
            }
            T_FOR(j, h0) {
                const float *w = W1 + (int64_t)j * 4 * D;
                float acc = prm[o.land_b0 + j];
                for (int c = 0; c < D; ++c) acc = fmaf(w[D + c] - w[3 * D + c], C[c], acc);
                cst[j] = acc;
                w2v[j] = prm[o.land_w1 + j];
            }
        } else {
            T_FOR(i, h0 * D) Aeff[i] = prm[o.road_w0 + i];
            T_FOR(j, h0) {
                cst[j] = prm[o.road_b0 + j];
                w2v[j] = prm[o.road_w1 + j];
            }
        }
        T_SYNC();
        for (int c0 = 0; c0 < nc; c0 += CH) {
            const int cn = nc - c0 < CH ? nc - c0 : CH;
            float *mq = chunk, *hid = chunk + CH * D;
            cand_inputs(c0, cn, mq);
            cand_hidden(cn, mq, hid);
            T_FOR(q, cn) {
                float acc = 0.0f;
                for (int j = 0; j < h0; ++j) acc = fmaf(w2v[j], hid[q * h0 + j], acc);
                z[c0 + q] = acc;
            }
            T_SYNC();
        }
        // log-softmax over the candidates, log-prob of the action, entropy
        T_FOR(g, NG) {
            float mx = -INFINITY;
            for (int q = g; q < nc; q += NG) mx = fmaxf(mx, z[q]);
            part[g] = mx;
        }
        T_SYNC();
        if (T_TID0) {
            float mx = -INFINITY;
            for (int g = 0; g < NG; ++g) mx = fmaxf(mx, part[g]);
            scal[0] = mx;
        }
        T_SYNC();
        T_FOR(g, NG) {
            float sum = 0.0f;
            for (int q = g; q < nc; q += NG) sum += t_exp(z[q] - scal[0]);
            part[g] = sum;
        }
        T_SYNC();
        if (T_TID0) {
            float sum = 0.0f;
            for (int g = 0; g < NG; ++g) sum += part[g];
            scal[1] = scal[0] + t_log(sum);
        }
        T_SYNC();
        T_FOR(g, NG) {
            float pz = 0.0f;
            for (int q = g; q < nc; q += NG) {
                const float lp = z[q] - scal[1];
                pz += t_exp(lp) * lp;
            }
            part[g] = pz;
        }
        T_SYNC();
        if (T_TID0) {
            float pz = 0.0f;
            for (int g = 0; g < NG; ++g) pz += part[g];
            scal[2] = -pz;
            scal[3] = (act >= 0 ? z[act] : PAD_LOGIT) - scal[1];
        }
        T_SYNC();
        float *zout = land ? A.z_he : A.z_rn;
        if (zout) {
            const int64_t zo = land ? A.he_off[b] : A.rn_off[b];
            T_FOR(q, nc) zout[zo + q] = z[q];
        }
    } else {
        // a row of another stage, or without any valid candidate (every logit is the pad constant, whose logsumexp is
        // absorbed in fp32): log-prob = entropy = 0 (policy.py:90-91)
        if (T_TID0) {
            scal[1] = 0.0f;
            scal[2] = 0.0f;
            scal[3] = 0.0f;
        }
        T_SYNC();
    }
    const float value = V[d.n_value][0], logp = scal[3], entr = scal[2];
    if (A.mode != BWD && T_TID0) {
        A.value[b] = value;
        A.logp[b] = logp;
        A.ent[b] = entr;
    }
    if (!bwd) {
        T_SYNC();
        return;
    }

    // =============================================================================== loss seeds
    float gv, gl, ge;
    if (A.mode == STEP) {
        // value loss over all rows, surrogate + entropy over rows with exps != 0 (urban_planning_agent.py:326-333,363-371);
        // torch.min ties / clamp edges as autograd resolves them (dense.hip: ppo_loss_kernel)
        const int64_t tr = A.rows ? A.rows[b] : b;
        const float diff = value - A.ret[tr];
        gv = A.cv * 2.0f * diff * A.inv_rows;
        gl = 0.0f;
        ge = 0.0f;
        float smin = 0.0f, sent = 0.0f;
        if (A.exps[tr] != 0.0f) {
            const float lo = 1.0f - A.clip_eps, hi = 1.0f + A.clip_eps;
            const float ratio = expf(logp - A.old_logp[tr]), Ad = A.adv[tr];
            const float s1 = ratio * Ad, s2 = fminf(fmaxf(ratio, lo), hi) * Ad;
            smin = fminf(s1, s2);
            sent = entr;
            const bool inside = ratio >= lo && ratio <= hi;
            const float dsdr = inside ? Ad : (s1 < s2 ? Ad : 0.0f);
            gl = -dsdr * ratio * A.inv_ind;
            ge = -A.ce * A.inv_ind;
        }
        if (T_TID0) {
            A.loss_rows[(int64_t)b * 4 + 0] = diff * diff;
            A.loss_rows[(int64_t)b * 4 + 1] = smin;
            A.loss_rows[(int64_t)b * 4 + 2] = sent;
            A.loss_rows[(int64_t)b * 4 + 3] = 0.0f;
        }
    } else {
        gv = A.dvalue[b];
        gl = A.dlogp[b];
        ge = A.dent[b];
    }

    // =============================================================================== backward
    float *G = slab;                             // gradient slab of this workgroup (parameter layout)
    // ---- value head (value.py:15-39)
    {
        float *dz = dVa, *dn = dVb;
        if (T_TID0) dz[0] = gv;
        T_SYNC();
        for (int i = d.n_value - 1; i >= 0; --i) {
            const int N = d.value_hidden[i], K = i == 0 ? d.W : d.value_hidden[i - 1];
            if (i < d.n_value - 1) {
                T_FOR(j, N) dz[j] *= 1.0f - V[i + 1][j] * V[i + 1][j];
                T_SYNC();
            }
            outer_acc(G + o.value_w[i], G + o.value_b[i], dz, V[i], N, K);
            float *dst = i == 0 ? dSV : dn;
            lin_t(dst, dz, prm + o.value_w[i], N, K);
            if (i > 0) {
                float *tmp = dz;
                dz = dn;
                dn = tmp;
            }
        }
    }
    const float *dhbarV = dSV + d.S_last, *dhbarE = dSV + d.S_last + D, *datt = dSV + d.S_last + 2 * D;
    // ---- numerical encoder
    {
        float *dz = dVa, *dn = dVb;
        T_FOR(j, d.S_last) dz[j] = dSV[j];
        T_SYNC();
        for (int i = d.n_num - 1; i >= 0; --i) {
            const int N = d.num_hidden[i], K = i == 0 ? d.Fn : d.num_hidden[i - 1];
            T_FOR(j, N) dz[j] *= 1.0f - U[i + 1][j] * U[i + 1][j];
            T_SYNC();
            outer_acc(G + o.num_w[i], G + o.num_b[i], dz, U[i], N, K);
            if (i > 0) {
                lin_t(dn, dz, prm + o.num_w[i], N, K);
                float *tmp = dz;
                dz = dn;
                dn = tmp;
            }
        }
    }
    // ---- attention, the part behind the softmax: out-projection, value projections
    outer_acc(G + o.outproj_w, G + o.outproj_b, datt, ov, D, D);
    lin_t(dov, datt, prm + o.outproj_w, D, D);
    // o[i] = Wiv[i] . u_h(i) + biv[i]:  dbiv += do;  dWiv[i][k] += do[i] u_h(i)[k];  du_h[k] = sum_{i in h} Wiv[i][k] do[i]
    T_FOR(i, D * D) {
        const int r = i / D, k = i - r * D;
        G[o.inproj_w + (int64_t)(2 * D + r) * D + k] += dov[r] * uu[(r / dh) * D + k];
    }
    T_FOR(i, D) G[o.inproj_b + 2 * D + i] += dov[i];
    T_SYNC();
    for (int h = 0; h < Hn; ++h) lin_t(du + h * D, dov, prm + o.inproj_w + (int64_t)2 * D * D, D, D, h * dh, (h + 1) * dh);
    // u_h = Wv s_h + bv:  dbv += sum_h du_h;  dWv[j][k] += sum_h du_h[j] s_h[k];  ds_h = Wv^T du_h
    T_FOR(i, D * D) {
        const int j = i / D, k = i - j * D;
        float acc = 0.0f;
        for (int h = 0; h < Hn; ++h) acc += du[h * D + j] * ss[h * D + k];
        G[o.v_w + i] += acc;
    }
    T_FOR(j, D) {
        float acc = 0.0f;
        for (int h = 0; h < Hn; ++h) acc += du[h * D + j];
        G[o.v_b + j] += acc;
    }
    T_SYNC();
    for (int h = 0; h < Hn; ++h) lin_t(ds + h * D, du + h * D, prm + o.v_w, D, D);
    // ---- attention core per head: t_j = ds . h_j, T = sum alpha t, dscore_j = alpha_j (t_j - T), dr = sum_j dscore_j h_j
    for (int h = 0; h < Hn; ++h) {
        float *tj = sc + (int64_t)h * n;
        const float *al = alpha + (int64_t)h * n, *dsv = ds + h * D;
        T_FOR(v, n) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < D; ++k) acc = fmaf(dsv[k], HL[v * D + k], acc);
            tj[v] = acc;
        }
        T_SYNC();
        T_FOR(g, NG) {
            float sum = 0.0f;
            for (int v = g; v < n; v += NG) sum += al[v] * tj[v];
            part[g] = sum;
        }
        T_SYNC();
        if (T_TID0) {
            float sum = 0.0f;
            for (int g = 0; g < NG; ++g) sum += part[g];
            scal[6] = sum;
        }
        T_SYNC();
        T_FOR(v, n) tj[v] = al[v] * (tj[v] - scal[6]);       // dscore_j (0 on nodes outside the mask: alpha = 0)
        T_SYNC();
        colsum(dr + h * D, part, n, D, 1.0f, [&](int v, int c) { return tj[v] * HL[v * D + c]; });
    }
    // ---- pointer head (before H^L is overwritten by G^L): dz, second + first Linear, candidate inputs
    T_FOR(i, imax(d.h0l, d.h0r) * D) Mj[i] = 0.0f;
    T_FOR(i, imax(d.h0l, d.h0r)) {
        sj[i] = 0.0f;
        dw2[i] = 0.0f;
    }
    T_FOR(i, D) dC[i] = 0.0f;
    T_SYNC();
    float *dXR = PQ;                             // road rows only: [nc][D] gradient of the candidates' H^L rows (P | Q is free)
    float *dMg = gscr;                           // land rows: [nc][D] gradient of the candidates' messages (global scratch)
    if (land || road) {
        const float lse = scal[1], Hent = scal[2];
        for (int c0 = 0; c0 < nc; c0 += CH) {
            const int cn = nc - c0 < CH ? nc - c0 : CH;
            float *mq = chunk, *hid = chunk + CH * D, *dpre = chunk + CH * D + CH * h0;
            cand_inputs(c0, cn, mq);
            cand_hidden(cn, mq, hid);
            // dz_k = dlogp (delta_ka - p_k) - dent p_k (log p_k + H);  dpre[k][j] = dz_k w2[j] (1 - hid^2)
            T_FOR(i, cn * h0) {
                const int q = i / h0, j = i - q * h0;
                const float lp = z[c0 + q] - lse, p = t_exp(lp);
                float dz = -gl * p - ge * p * (lp + Hent);
                if (c0 + q == act) dz += gl;
                const float hv = hid[i];
                dpre[i] = dz * w2v[j] * (1.0f - hv * hv);
                hid[i] = dz * hv;                 // (dz hid: the summand of dw2)
            }
            T_SYNC();
            // running sums over the candidates: dw2, db1 (= s), M[j][c] = sum dpre[k][j] m[k][c]
            T_FOR(j, h0) {
                float a1 = dw2[j], a2 = sj[j];
                for (int q = 0; q < cn; ++q) {
                    a1 += hid[q * h0 + j];
                    a2 += dpre[q * h0 + j];
                }
                dw2[j] = a1;
                sj[j] = a2;
            }
            T_FOR(i, h0 * D) {
                const int j = i / D, c = i - j * D;
                float acc = Mj[i];
                for (int q = 0; q < cn; ++q) acc = fmaf(dpre[q * h0 + j], mq[q * D + c], acc);
                Mj[i] = acc;
            }
            // gradient of the candidate inputs: dm[k][c] = sum_j A[j][c] dpre[k][j]  (land: only live candidates carry it on)
            float *dst = land ? dMg : dXR;
            T_FOR(i, cn * D) {
                const int q = i / D, c = i - q * D;
                float acc = 0.0f;
                for (int j = 0; j < h0; ++j) acc = fmaf(Aeff[j * D + c], dpre[q * h0 + j], acc);
                if (land && !hlive[c0 + q]) acc = 0.0f;
                dst[(int64_t)(c0 + q) * D + c] = acc;
            }
            T_SYNC();
        }
        if (land) {
            // feat = [m; c; m*c; m-c]:  dWa += M,  dWb += s (x) c,  dWc += M * c,  dWd += M - s (x) c;  db1 += s;  dw2
            // dC[c] += sum_j (Wb - Wd)[j][c] s[j] + Wc[j][c] M[j][c]
            const float *W1 = prm + o.land_w0;
            T_FOR(i, h0 * D) {
                const int j = i / D, c = i - j * D;
                float *g = G + o.land_w0 + (int64_t)j * 4 * D;
                const float Mv = Mj[i], sc_ = sj[j] * C[c];
                g[c] += Mv;
                g[D + c] += sc_;
                g[2 * D + c] += Mv * C[c];
                g[3 * D + c] += Mv - sc_;
            }
            T_FOR(j, h0) {
                G[o.land_b0 + j] += sj[j];
                G[o.land_w1 + j] += dw2[j];
            }
            T_FOR(c, D) {
                float acc = 0.0f;
                for (int j = 0; j < h0; ++j) {
                    const float *w = W1 + (int64_t)j * 4 * D;
                    acc += (w[D + c] - w[3 * D + c]) * sj[j] + w[2 * D + c] * Mj[j * D + c];
                }
                dC[c] = acc;
            }
        } else {
            T_FOR(i, h0 * D) G[o.road_w0 + i] += Mj[i];
            T_FOR(j, h0) {
                G[o.road_b0 + j] += sj[j];
                G[o.road_w1 + j] += dw2[j];
            }
        }
        T_SYNC();
    }
    // ---- G^L in place of H^L: masked-mean share + attention terms (+ the road candidates' rows)
    T_FOR(i, nD) {
        const int v = i / D, c = i - v * D;
        float g = nmg[v] ? dhbarV[c] / (float)m[6] : 0.0f;
        for (int h = 0; h < Hn; ++h) g += alpha[(int64_t)h * n + v] * ds[h * D + c] + sc[(int64_t)h * n + v] * rr[h * D + c];
        HL[i] = g;
    }
    T_SYNC();
    if (road) {
        T_FOR(i, nc * D) {                       // road_mask slots are distinct nodes: one writer per element
            const int q = i / D, c = i - q * D;
            HL[rnn[q] * D + c] += dXR[i];
        }
        T_SYNC();
    }
    float *Gn = HL;
    // ---- attention, the query chain:  r_h = Wk^T tk_h,  tk_h = Wik[head rows]^T q1[head rows]
    for (int h = 0; h < Hn; ++h) {
        // dtk_h[j] = sum_d Wk[j][d] dr_h[d]
        T_FOR(j, D) {
            float acc = 0.0f;
            for (int k = 0; k < D; ++k) acc = fmaf(prm[o.k_w + j * D + k], dr[h * D + k], acc);
            dtk[h * D + j] = acc;
        }
        T_SYNC();
    }
    T_FOR(i, D * D) {                            // dWk[j][d] += sum_h tk_h[j] dr_h[d]
        const int j = i / D, k = i - j * D;
        float acc = 0.0f;
        for (int h = 0; h < Hn; ++h) acc += tk[h * D + j] * dr[h * D + k];
        G[o.k_w + i] += acc;
    }
    T_FOR(i, D * D) {                            // dWik[r][j] += q1[r] dtk_h(r)[j]
        const int r = i / D, j = i - r * D;
        G[o.inproj_w + (int64_t)(D + r) * D + j] += q1[r] * dtk[(r / dh) * D + j];
    }
    T_FOR(r, D) {                                // dq1[r] = sum_j Wik[r][j] dtk_h(r)[j]
        const float *w = prm + o.inproj_w + (int64_t)(D + r) * D, *dt = dtk + (r / dh) * D;
        float acc = 0.0f;
        for (int j = 0; j < D; ++j) acc = fmaf(w[j], dt[j], acc);
        dq1[r] = acc * scale;                    // through q1 = (Wiq q0 + biq) * scale
    }
    T_SYNC();
    outer_acc(G + o.inproj_w, G + o.inproj_b, dq1, q0, D, D);
    lin_t(dq0, dq1, prm + o.inproj_w, D, D);
    outer_acc(G + o.q_w, G + o.q_b, dq0, C, D, D);
    T_FOR(c, D) {
        float acc = dC[c];
        for (int j = 0; j < D; ++j) acc = fmaf(prm[o.q_w + j * D + c], dq0[j], acc);
        dC[c] = acc;
    }
    T_SYNC();
    // current node's pass through the node encoder
    outer_acc(G + o.node_w, G + o.node_b, dC, cur, D, d.F);

    // ---- GCN layers, last to first (:110-148,194-197).  Per layer in two column halves: P | Q of the half in PQ[0 .. nD),
    // dP | dQ of the half in PQ[nD .. 2 nD).  Row layout of a half: [v][0 .. D/2) = P columns, [v][D/2 .. D) = Q columns.
    constexpr int HC = D / 2;
    float *PQh = PQ, *dPQh = PQ + nD, *dS = Xr;
    for (int l = L; l >= 1; --l) {
        const bool last = l == L;
        float *Hprev;
        if (l == 1) {
            Hprev = slotH(1);                    // H^1 is dead (layer 2 is done): recompute H^0 in its place
            encode_nodes(Hprev);
        } else {
            Hprev = slotH(l - 1);
        }
        const float *Wl = prm + o.edge_w[l - 1], *bl = prm + o.edge_b[l - 1];
        float *gW = G + o.edge_w[l - 1], *gB = G + o.edge_b[l - 1];
        // dS_v = G_v / (deg_v + 1e-6) (+ 1/2 dhbarE / e on the last layer)
        T_FOR(i, nD) {
            const int v = i / D, c = i - v * D;
            float x = Gn[i] / ((float)(rp[v + 1] - rp[v]) + 1e-6f);
            if (last) x += 0.5f * dhbarE[c] / (float)e;
            dS[i] = x;
        }
        T_SYNC();
        for (int half = 0; half < 2; ++half) {
            const int cb = half * HC;
            // P | Q of the half
            T_FOR(i, nD) {
                const int v = i / D, jj = i - v * D;
                const int side = jj / HC, c = cb + jj % HC;
                const float *w = Wl + (int64_t)c * (2 * D) + side * D, *h = Hprev + (int64_t)v * D;
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < D; ++k) acc = fmaf(w[k], h[k], acc);
                PQh[i] = acc;
            }
            T_SYNC();
            // dP_v = sum_u 1/2 dm (1 - tanh^2(P_v + Q_u + b)),  dQ_v = sum_u 1/2 dm (1 - tanh^2(P_u + Q_v + b)),
            // dm = dS_v + dS_u; the row's candidate edges add their head gradient on the last layer (packer's
            // candidate-incidence lists: neighbour + candidate index per incident live candidate)
            T_FOR(i, n * HC) {
                const int v = i / HC, cc = i - v * HC, c = cb + cc;
                const float bc = bl[c];
                const float pv = PQh[v * D + cc] + bc, qv = PQh[v * D + HC + cc] + bc, sv = dS[v * D + c];
                float aP = 0.0f, aQ = 0.0f;
                for (int k = rp[v]; k < rp[v + 1]; ++k) {
                    const int u = nb[k];
                    const float dm = sv + dS[u * D + c];
                    const float t1 = t_tanh(pv + PQh[u * D + HC + cc]), t2 = t_tanh(PQh[u * D + cc] + qv);
                    aP = fmaf(0.5f * dm, 1.0f - t1 * t1, aP);
                    aQ = fmaf(0.5f * dm, 1.0f - t2 * t2, aQ);
                }
                if (last && land) {
                    for (int k = hpg[v]; k < hpg[v + 1]; ++k) {
                        const int u = hnb[k];
                        const float dm = dMg[(int64_t)hhe[k] * D + c];
                        const float t1 = t_tanh(pv + PQh[u * D + HC + cc]), t2 = t_tanh(PQh[u * D + cc] + qv);
                        aP = fmaf(0.5f * dm, 1.0f - t1 * t1, aP);
                        aQ = fmaf(0.5f * dm, 1.0f - t2 * t2, aQ);
                    }
                }
                dPQh[v * D + cc] = aP;
                dPQh[v * D + HC + cc] = aQ;
            }
            T_SYNC();
            // weight / bias gradient of the half's rows:  dW[c][side * D + k] += sum_v d(side)_v[c] H_v[k];  db[c] += sum_v dP_v[c]
            T_FOR(i, D * D) {
                const int jj = i / D, k = i - jj * D;
                const int side = jj / HC, c = cb + jj % HC;
                float acc = 0.0f;
                for (int v = 0; v < n; ++v) acc = fmaf(dPQh[v * D + jj], Hprev[v * D + k], acc);
                gW[(int64_t)c * (2 * D) + side * D + k] += acc;
            }
            T_FOR(cc, HC) {
                float acc = 0.0f;
                for (int v = 0; v < n; ++v) acc += dPQh[v * D + cc];
                gB[cb + cc] += acc;
            }
            // dgrad in place: G_v[k] += sum_c dP_v[c] Wa[c][k] + dQ_v[c] Wb[c][k]   (the walks read dS, not G)
            T_FOR(i, nD) {
                const int v = i / D, k = i - v * D;
                float acc = Gn[i];
                for (int jj = 0; jj < D; ++jj) {
                    const int side = jj / HC, c = cb + jj % HC;
                    acc = fmaf(dPQh[v * D + jj], Wl[(int64_t)c * (2 * D) + side * D + k], acc);
                }
                Gn[i] = acc;
            }
            T_SYNC();
        }
    }
    // ---- node encoder on every node: dWe += G^0^T X, dbe += colsum(G^0)
    T_FOR(i, D * d.F) {
        const int c = i / d.F, f = i - c * d.F;
        float acc = 0.0f;
        for (int v = 0; v < n; ++v) acc = fmaf(Gn[v * D + c], Xg[(int64_t)v * XPAD + f], acc);
        G[o.node_w + i] += acc;
    }
    T_FOR(c, D) {
        float acc = 0.0f;
        for (int v = 0; v < n; ++v) acc += Gn[v * D + c];
        G[o.node_b + c] += acc;
    }
    T_SYNC();
}

}  // namespace upamd_tiny
