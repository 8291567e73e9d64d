// Fused small-model path (gcn_node_dim <= 32 -- the dims every shipped YAML uses, hlg.yaml:21-33): ONE workgroup owns one
// graph and runs the whole network on it out of LDS -- numerical encoder, node encoder, every GCN layer, masked means,
// single-query attention, value head, pointer head, the PPO loss seeds and the complete backward -- so an optimizer step is
// three launches (this kernel, the slab reduction, Adam) instead of ~35 dependent ones.
//
// Reference math: urban_planning/models/state_encoder.py:84-214 (encoder), policy.py:19-104 (pointer heads, log-prob /
// entropy), value.py:15-39 (value head), urban_planning_agent.py:326-333,363-371 + khrylib/rl/agents/agent_pg.py:19-23
// (loss); the hand-derived backward is the one of tests/csr_model.py, stage by stage.
//
// This header holds the PER-GRAPH program: tiny.hip instantiates it as a HIP kernel; the test infrastructure
// (tests/emul/tiny_emul.cpp, TINY_HOST) compiles the same text with g++ and runs it on the CPU against the oracle before any GPU
// time is spent.  Target-specific are only the macros below and four small #ifdef'd helpers that load the same values another way
// (the meta row, the staging loads of the first phase, ld_row, t_tid).  The contract that makes that possible:
//   * all parallelism is "for every index of a range, independent iterations" -- T_FOR(i, N), or T_FOR_J(j, NJ) { .. T_FOR_V(v,
//     NV, NJ) { .. } } where a thread keeps ONE j (its weight row in registers) and strides over v -- closed by T_SYNC() (a
//     workgroup barrier on the GPU, nothing on the host where the ranges run sequentially); no wave intrinsics, no float
//     atomics (the one atomic is T_FLAG, an OR of a range flag: order independent), no thread-private state that survives a
//     T_SYNC();
//   * a phase may carry SIDE work (T_FOR_SIDE: one layer of a per-sample chain, run by the last wave) next to its own loops
//     (T_FORM*: the same loops on the other waves); on the host both are plain loops, side work first;
//   * an iteration writes only elements it owns; every sum is a serial loop in a fixed order inside one iteration (two-level
//     sums: fixed partial groups, then a fixed combine) => bit-reproducible run to run and for any block size;
//   * gradients of the parameters go to the workgroup's own slab in global memory (`+=` by the owning iteration); the slabs
//     are added in a fixed order by the reduction launch.
//
// What bounds the kernel is the LATENCY of its ~95 dependent phases, not arithmetic (1 MFLOP per graph).  What that took is in
// profiles/archive/r04_lab_tiny_sections.log: weight rows in registers, every serial LDS sum with eight loads in flight, LDS rows loaded
// whole before their FMAs, a per-phase thread id (no cross-phase CSE of address math), staging loads issued together.
#pragma once
#include <stdint.h>

#include <type_traits>

#ifdef TINY_HOST
#include <math.h>
#define TDEV static inline
#define TMEM inline
#define THD static inline
#define T_FOR(i, N) for (int i = 0; i < (N); ++i)
#define T_FOR_J(j, NJ) for (int j = 0; j < (NJ); ++j)
#define T_FOR_V(v, NV, NJ) for (int v = 0; v < (NV); ++v)
#define T_FORM(i, N) T_FOR(i, N)
#define T_FORM_J(j, NJ) T_FOR_J(j, NJ)
#define T_FORM_V(v, NV, NJ) T_FOR_V(v, NV, NJ)
#define T_FOR_SIDE(j, N) for (int j = 0; j < (N); ++j)
#define T_SYNC() \
    do {         \
    } while (0)
#define T_MARK(k) \
    do {          \
    } while (0)
#define T_FLAG(p) (*(p) |= 1)
static inline float t_exp2(float x) { return exp2f(x); }
static inline float t_rcp(float x) { return 1.0f / x; }
static inline float t_log(float x) { return logf(x); }
#else
#define TDEV __device__ __forceinline__
#define TMEM __device__ __forceinline__
#define THD __host__ __device__ inline
// The thread id of a phase is laundered through an empty asm: what a phase derives from it (tid / 16, LDS addresses, ...) is then
// recomputed per phase (a few VALU instructions) instead of being shared across ALL phases by common-subexpression elimination and
// held live for the whole program -- which is what had the 128-VGPR variant spill ~100 registers and serialise loads through scratch.
static __device__ __forceinline__ int t_tid() {
    int t = (int)threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}
static __device__ __forceinline__ int t_tidm() {      // ... of a main thread; beyond every range for the side wave
    const int t = t_tid();
    return t < (int)blockDim.x - 64 ? t : 0x3fffffff;
}
#define T_FOR(i, N) for (int i = t_tid(); i < (N); i += (int)blockDim.x)
// thread -> (j = tid % NJP, first v = tid / NJP), NJP = NJ rounded up to a power of two (it divides the block size)
#define T_FOR_J(j, NJ) for (int j = t_tid() & (upamd_tiny::np2(NJ) - 1), _once = 1; _once && j < (NJ); _once = 0)
#define T_FOR_V(v, NV, NJ) \
    for (int v = t_tid() / upamd_tiny::np2(NJ), _vs = (int)blockDim.x / upamd_tiny::np2(NJ); v < (NV); v += _vs)
// A phase may carry SIDE work: one layer of the per-sample chains (numerical encoder, query path), which depend on nothing the
// graph phases produce until the attention.  The workgroup's last wave does it (T_FOR_SIDE) while the others do the phase's own
// loops (T_FORM*: the same loops over blockDim - 64 threads); the phase's barrier publishes both.  A chain layer is one L2 round
// trip for its weights and a barrier -- ~1.2 us as a phase of its own, nothing underneath a graph phase.
#define T_MAIN ((int)blockDim.x - 64)
#define T_FORM(i, N) for (int i = t_tidm(); i < (N); i += T_MAIN)
#define T_FORM_J(j, NJ) \
    for (int _t = t_tid(), j = _t & (upamd_tiny::np2(NJ) - 1), _once = _t < T_MAIN; _once && j < (NJ); _once = 0)
#define T_FORM_V(v, NV, NJ) for (int v = t_tid() / upamd_tiny::np2(NJ), _vs = T_MAIN / upamd_tiny::np2(NJ); v < (NV); v += _vs)
#define T_FOR_SIDE(j, N) for (int j = t_tid() - T_MAIN; (unsigned)j < (unsigned)(N); j += 64)
#define T_SYNC() __syncthreads()
// section time stamps (100 MHz wall clock) of the FIRST graph of workgroup 0 into A.prof (lab hook, null in production)
#define T_MARK(k)                                                                                \
    do {                                                                                         \
        if (A.prof && blockIdx.x == 0 && b == 0 && threadIdx.x == 0) A.prof[k] = wall_clock64(); \
    } while (0)
// raise a per-graph flag word in LDS (an OR: order independent, so the result does not depend on who gets there first)
#define T_FLAG(p) atomicOr((p), 1)
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
constexpr int64_t LDS_FLOATS = (160 * 1024 - 512) / 4;   // what one workgroup may use of a CU's LDS
constexpr int CHMIN = 32;         // pointer-head candidates per chunk the LDS plan guarantees (more when the graph leaves room)
constexpr float C2 = 2.8853900817779268f;     // 2 log2(e)
constexpr float LOG2E = 1.4426950408889634f;
constexpr float PAD_LOGIT = -4294967296.0f;   // -2^32 + 1 in fp32 (policy.py:50,59)
// Exp form of a layer's P | Q (as in edge.hip): with E = 2^(C2 (P_v + Q_u + b)) = eP_v * eb * eQ_u, tanh(x) = 1 - 2 / (1 + E) needs no
// exponential per incidence -- valid while no factor over / underflows: every |C2 P|, |C2 Q| <= EF_LIMIT and |C2 b| <= EF_BIAS keep
// all products inside [2^-86, 2^86].  A graph whose layer leaves that range raises the layer's flag and walks in the linear form.
constexpr float EF_LIMIT = 40.0f, EF_BIAS = 6.0f;

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
    const uint16_t *inc_nbr, *he_src, *he_dst, *rn_node, *hinc_nbr, *hinc_he, *order;
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
    int max_n, max_inc, max_cand;
    long long *prof;                              // lab hook: section time stamps (see T_MARK), normally null
};

TDEV float t_tanh(float x) { return 1.0f - 2.0f * t_rcp(t_exp2(C2 * x) + 1.0f); }
TDEV float t_exp(float x) { return t_exp2(x * LOG2E); }
THD int imax(int a, int b) { return a > b ? a : b; }
THD int imin(int a, int b) { return a < b ? a : b; }
THD int64_t a4(int64_t x) { return (x + 3) / 4 * 4; }
THD int np2(int x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}

// ---- LDS plan (floats).  n / inc / cand = the LARGEST graph of the launch: one plan per launch, every graph uses its prefix.
struct Plan {
    int64_t H, X, PQ, rp, nb, ord, alpha, sc, nm, vec, total;      // offsets
    int64_t xsize, vsize;
};
// head scratch inside the X region: z [nc] | candidate lists (2 u16 + 1 u8 per candidate -> 1.25 floats, kept at 1.5) | chunk buffers
THD int64_t head_fixed(int nc) { return a4(nc) + a4((3 * (int64_t)nc + 1) / 2); }
THD int part_floats(const Dims &d) {
    const int Hd = d.heads * d.D;
    return imax(imax(NG * 2 * imax(Hd, 32), 4 * d.D * d.D + NG * d.D), 2 * d.D * (XPAD + 1));
}
THD int64_t vec_floats(const Dims &d) {
    int64_t u = d.Fn, v = 0;
    for (int i = 0; i < d.n_num; ++i) u += d.num_hidden[i];
    for (int i = 0; i < d.n_value; ++i) v += d.value_hidden[i];
    const int64_t D = d.D, Hd = (int64_t)d.heads * d.D, h0 = imax(d.h0l, d.h0r);
    //     U        cur   16 D-vectors   10 head vectors   SV, dSV     V, dV      A, M (h0 x D)  const, s, w2..   partials   scalars + slack
    return u + XPAD + 16 * D + 10 * Hd + 2 * a4(d.W) + 2 * a4(v) + 3 * h0 * D + 8 * h0 + part_floats(d) + 64 + 256;
}
THD Plan plan_layout(const Dims &d, int n, int inc, int cand, int64_t x_extra) {
    Plan p;
    const int64_t nD = (int64_t)n * d.D;
    const int64_t per = d.D + 2 * (imax(d.h0l, d.h0r) + 1);                        // chunk floats per candidate
    const int64_t hs = head_fixed(imax(cand, 1)) + (int64_t)CHMIN * per;
    int64_t o = 0;
    p.H = o; o += a4((int64_t)d.L * nD);                                           // H^1 .. H^L ([n][D] each; slot L becomes G)
    p.xsize = (nD > hs ? nD : hs) + x_extra;
    p.X = o; o += a4(p.xsize);                                                     // S / dS | head scratch
    p.PQ = o; o += a4(imax((int)(2 * nD), n * XPAD));                              // P | Q of a layer / staged raw features; backward: half | d(half)
    p.rp = o; o += a4(n + 1);
    p.nb = o; o += a4((inc + 1) / 2 + 1);                                          // u16 neighbour ids
    p.ord = o; o += a4((n + 1) / 2 + 1);                                           // u16 node ids in processing (degree-sorted) order
    p.alpha = o; o += a4((int64_t)d.heads * n);
    p.sc = o; o += a4((int64_t)d.heads * n);
    p.nm = o; o += a4((n + 3) / 4);                                                // node_mask bytes
    p.vsize = vec_floats(d);
    p.vec = o; o += a4(p.vsize);
    p.total = o;
    return p;
}
THD Plan make_plan(const Dims &d, int n, int inc, int cand) {
    // what the launch's largest graph leaves of the LDS goes to the head's chunk buffers, up to one chunk for every candidate
    // (a chunk costs four barriers forward + backward: the candidate-rich graphs were the last to finish)
    const Plan base = plan_layout(d, n, inc, cand, 0);
    const int64_t per = d.D + 2 * (imax(d.h0l, d.h0r) + 1);
    const int64_t want = head_fixed(imax(cand, 1)) + (int64_t)imax(cand, 1) * per - base.xsize;
    const int64_t room = LDS_FLOATS - base.total;
    int64_t extra = want < room ? want : room;
    extra = extra > 0 ? extra & ~(int64_t)3 : 0;
    return plan_layout(d, n, inc, cand, extra);
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

// sum_k w[k] x[k]: the (global-memory) weights in batches of 16 loads in flight (the layers are latency-bound: an L2 round trip per
// batch is what a phase costs)
struct f4 {
    float x, y, z, w;
};
// A row of K floats (16-byte aligned, K % 4 == 0) out of LDS into registers with ALL its 16-byte loads issued before the first use.
// Written as `acc = fmaf(w[k], row[k], acc)` over the LDS pointer, hipcc pairs every 8-byte read with its two FMAs and waits for
// each one: eight serial LDS round trips per 16-float row (seen in the ISA), 3 of a P | Q phase's 7 us.
#ifdef TINY_HOST
template <int K>
TDEV void ld_row(const float *p, float (&x)[K]) {
    for (int k = 0; k < K; ++k) x[k] = p[k];
}
#else
typedef float f4v __attribute__((ext_vector_type(4)));
template <int K>
TDEV void ld_row(const float *p, float (&x)[K]) {
    const f4v *p4 = reinterpret_cast<const f4v *>(p);
    f4v v[K / 4];
#pragma unroll
    for (int q = 0; q < K / 4; ++q) v[q] = p4[q];
    __builtin_amdgcn_sched_barrier(0);           // (keeps the loads together: the scheduler otherwise sinks each to its consumer)
#pragma unroll
    for (int q = 0; q < K / 4; ++q) {
        x[4 * q] = v[q].x;
        x[4 * q + 1] = v[q].y;
        x[4 * q + 2] = v[q].z;
        x[4 * q + 3] = v[q].w;
    }
}
#endif
TDEV float dot_g(const float *w, const float *x, int K, float acc = 0.0f) {
    int k = 0;
    if ((K & 3) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0) {
        // 16-byte loads, 32 floats per round trip (the parameter tensors start on 4-float boundaries: rows of a K % 4 == 0
        // layer are 16-byte aligned)
        const f4 *w4 = reinterpret_cast<const f4 *>(w);
        for (; k + 32 <= K; k += 32) {
            f4 a[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = w4[(k >> 2) + q];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                acc = fmaf(a[q].w, x[k + 4 * q + 3], fmaf(a[q].z, x[k + 4 * q + 2], fmaf(a[q].y, x[k + 4 * q + 1], fmaf(a[q].x, x[k + 4 * q], acc))));
        }
        if (k < K) {                           // last 4 .. 28 floats: clamped loads, masked adds (all in flight together)
            const int rem = (K - k) >> 2;
            f4 a[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = w4[(k >> 2) + (q < rem ? q : rem - 1)];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (q < rem)
                    acc = fmaf(a[q].w, x[k + 4 * q + 3], fmaf(a[q].z, x[k + 4 * q + 2], fmaf(a[q].y, x[k + 4 * q + 1], fmaf(a[q].x, x[k + 4 * q], acc))));
        }
        return acc;
    }
    for (; k + 16 <= K; k += 16) {
        float a[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = w[k + q];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc = fmaf(a[q], x[k + q], acc);
    }
    if (k + 8 <= K) {
        float a[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = w[k + q];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = fmaf(a[q], x[k + q], acc);
        k += 8;
    }
    if (k < K) {                               // tail: clamped loads, masked adds (all in flight together)
        float a[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = w[k + q < K ? k + q : K - 1];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = k + q < K ? fmaf(a[q], x[k + q], acc) : acc;
    }
    return acc;
}
// Serial sums over LDS operands: a loop `acc = fmaf(a[t], b[t], acc)` that waits for its two loads in every iteration costs an LDS
// round trip (~100 cycles) per term, and a phase's few waves cannot hide it.  These keep SB loads of each operand in flight and add
// in index order (the same bits as the plain loop).  A short tail re-reads the last term and masks the add.
constexpr int SB = 8;
TDEV int cnt_s(int n, int first, int step) { return first < n ? (n - first + step - 1) / step : 0; }      // terms of `for (v = first; v < n; v += step)`
TDEV float dot_s(const float *a, int sa, const float *b, int sb, int cnt, float acc) {
    int t = 0;
    for (; t + SB <= cnt; t += SB) {
        float x[SB], y[SB];
#pragma unroll
        for (int q = 0; q < SB; ++q) {
            x[q] = a[(int64_t)(t + q) * sa];
            y[q] = b[(int64_t)(t + q) * sb];
        }
#pragma unroll
        for (int q = 0; q < SB; ++q) acc = fmaf(x[q], y[q], acc);
    }
    if (t < cnt) {
        float x[SB], y[SB];
#pragma unroll
        for (int q = 0; q < SB; ++q) {
            const int tt = t + q < cnt ? t + q : cnt - 1;
            x[q] = a[(int64_t)tt * sa];
            y[q] = b[(int64_t)tt * sb];
        }
#pragma unroll
        for (int q = 0; q < SB; ++q) acc = t + q < cnt ? fmaf(x[q], y[q], acc) : acc;
    }
    return acc;
}
TDEV float sum_s(const float *a, int sa, int cnt, float acc) {
    int t = 0;
    for (; t + SB <= cnt; t += SB) {
        float x[SB];
#pragma unroll
        for (int q = 0; q < SB; ++q) x[q] = a[(int64_t)(t + q) * sa];
#pragma unroll
        for (int q = 0; q < SB; ++q) acc += x[q];
    }
    if (t < cnt) {
        float x[SB];
#pragma unroll
        for (int q = 0; q < SB; ++q) x[q] = a[(int64_t)(t + q < cnt ? t + q : cnt - 1) * sa];
#pragma unroll
        for (int q = 0; q < SB; ++q) acc = t + q < cnt ? acc + x[q] : acc;
    }
    return acc;
}
TDEV float max_s(const float *a, int sa, int cnt, float mx) {
    int t = 0;
    for (; t < cnt; t += SB) {
        float x[SB];
#pragma unroll
        for (int q = 0; q < SB; ++q) x[q] = a[(int64_t)(t + q < cnt ? t + q : cnt - 1) * sa];
#pragma unroll
        for (int q = 0; q < SB; ++q) mx = fmaxf(mx, x[q]);
    }
    return mx;
}
// sum_j W[j * ld + k] in[j], j in [j0, j1): strided weights, 16 loads in flight
TDEV float dot_t(const float *W, int ld, int k, const float *in, int j0, int j1) {
    float acc = 0.0f;
    int j = j0;
    for (; j + 16 <= j1; j += 16) {
        float a[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) a[q] = W[(int64_t)(j + q) * ld + k];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc = fmaf(a[q], in[j + q], acc);
    }
    if (j + 8 <= j1) {
        float a[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = W[(int64_t)(j + q) * ld + k];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = fmaf(a[q], in[j + q], acc);
        j += 8;
    }
    if (j < j1) {
        float a[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = W[(int64_t)(j + q < j1 ? j + q : j1 - 1) * ld + k];
#pragma unroll
        for (int q = 0; q < 8; ++q) acc = j + q < j1 ? fmaf(a[q], in[j + q], acc) : acc;
    }
    return acc;
}

// out[j] = act((bias[j] + sum_k W[j * K + k] * in[k]) * scale);  weights from global memory, vectors in LDS
TDEV void lin(float *out, const float *in, const float *W, const float *bias, int N, int K, int act, float scale = 1.0f) {
    T_FOR(j, N) {
        const float acc = dot_g(W + (int64_t)j * K, in, K, bias ? bias[j] : 0.0f) * scale;
        out[j] = act ? t_tanh(acc) : acc;
    }
    T_SYNC();
}
// One layer of a small MLP backwards, ONE phase: weight gradient gw[j][k] += dz[j] x[k], bias gradient gb[j] += dz[j], and the
// gradient of the layer's input dx[k] = (sum_j W[j][k] dz[j]) * (1 - y[k]^2) (y = that input if it is a tanh output, else null)
TDEV void mlp_layer_bwd(float *gw, float *gb, float *dx, const float *dz, const float *x, const float *W, const float *y, int N,
                        int K) {
    T_FOR(i, N * K + N + (dx ? K : 0)) {
        if (i < N * K) {
            const int j = i / K, k = i - j * K;
            gw[i] += dz[j] * x[k];
        } else if (i < N * K + N) {
            gb[i - N * K] += dz[i - N * K];
        } else {
            const int k = i - N * K - N;
            float g = dot_t(W, K, k, dz, 0, N);
            if (y) g *= 1.0f - y[k] * y[k];
            dx[k] = g;
        }
    }
    T_SYNC();
}
// slab[(j, k)] += a[j] * x[k]  (rank-1 weight gradient of a per-sample layer), and slab_b[j] += a[j]
TDEV void outer_acc(float *gw, float *gb, const float *a, const float *x, int N, int K) {
    T_FOR(i, N * K + (gb ? N : 0)) {
        if (i < N * K) {
            const int j = i / K, k = i - j * K;
            gw[i] += a[j] * x[k];
        } else {
            gb[i - N * K] += a[i - N * K];
        }
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
    const int32_t *mrow = A.meta + (int64_t)t * META;
    int m[META];                                 // the row's 16 ints in ONE round trip (read field by field they were five)
#ifdef TINY_HOST
    for (int q = 0; q < META; ++q) m[q] = mrow[q];
#else
    {
        typedef int i4v __attribute__((ext_vector_type(4)));
        const i4v *m4 = reinterpret_cast<const i4v *>(mrow);
        const i4v a0 = m4[0], a1 = m4[1], a2 = m4[2], a3 = m4[3];
        __builtin_amdgcn_sched_barrier(0);
        m[0] = a0.x; m[1] = a0.y; m[2] = a0.z; m[3] = a0.w; m[4] = a1.x; m[5] = a1.y; m[6] = a1.z; m[7] = a1.w;
        m[8] = a2.x; m[9] = a2.y; m[10] = a2.z; m[11] = a2.w; m[12] = a3.x; m[13] = a3.y; m[14] = a3.z; m[15] = a3.w;
    }
#endif
    const int n = m[0], e = m[1], stage = m[4], act = m[5];
    const int nc = stage == 0 ? m[2] : (stage == 1 ? m[3] : 0);      // candidates of the row's pointer head
    const int64_t node_off = m[9];
    const float *Xg = A.X + node_off * XPAD;
    const uint8_t *nmg = A.nmask + node_off;
    const int32_t *rpg = A.rowptr + m[13];
    const uint16_t *nbg = A.inc_nbr + 2 * (int64_t)m[10];
    const uint16_t *og = A.order + node_off;
    const uint16_t *hsrc = A.he_src + m[11], *hdst = A.he_dst + m[11];
    const uint8_t *hlive = A.he_live + m[11];
    const uint16_t *rnn = A.rn_node + m[12];
    const int32_t *hpg = A.hinc_ptr + m[13];
    const uint16_t *hnb = A.hinc_nbr + 2 * (int64_t)m[11], *hhe = A.hinc_he + 2 * (int64_t)m[11];
    const int L = d.L, Hn = d.heads, dh = D / Hn, F = d.F;
    const int nD = n * D;
    const bool land = stage == 0 && nc > 0, road = stage == 1 && nc > 0;
    const bool bwd = A.mode != FWD;

    float *Hs = lds + pl.H;                      // slot l (1..L) at Hs + (l - 1) * nD
    float *Xr = lds + pl.X;
    float *PQ = lds + pl.PQ;                     // [n][2D]: P columns 0..D-1, Q columns D..2D-1
    float *Xs = PQ;                              // ... or the graph's raw node features [n][XPAD], while P | Q is not needed
    int *rp = reinterpret_cast<int *>(lds + pl.rp);
    uint16_t *nb = reinterpret_cast<uint16_t *>(lds + pl.nb);
    uint16_t *ord = reinterpret_cast<uint16_t *>(lds + pl.ord);      // the packer's degree-sorted order: a wave's nodes walk lists of similar length
    float *alpha = lds + pl.alpha, *sc = lds + pl.sc;
    uint8_t *nm = reinterpret_cast<uint8_t *>(lds + pl.nm);
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
    const int h0 = land ? d.h0l : d.h0r, h0m = imax(d.h0l, d.h0r);
    const int hs = h0 + 1;                        // row stride of the candidates' hidden rows in LDS: odd, so that both a thread per
                                                 // candidate and a thread per hidden unit walk them without bank conflicts
    float *Aeff = vb.get((int64_t)h0m * D);      // land: (Wa + Wd) + Wc diag(C);  road: R1
    float *AeffT = vb.get((int64_t)h0m * D);     // ... and its transpose [D][h0] (a thread per hidden unit reads it stride-1)
    float *cst = vb.get(h0m);                    // land: b1 + (Wb - Wd) C;        road: rb1
    float *w2v = vb.get(h0m);
    float *part = vb.get(part_floats(d));
    float *scal = vb.get(64);      // 0 max logit, 1 lse, 2 entropy, 3 log-prob, 8 + h: 1 / softmax sum of head h
    int *bad = reinterpret_cast<int *>(vb.get(MAXL + 4));      // bad[l] != 0: layer l's P | Q left the exp-form range (linear walk)
    // backward-only vectors
    float *dSV = vb.get(d.W), *dC = vb.get(D), *dq0 = vb.get(D), *dq1 = vb.get(D), *dov = vb.get(D);
    float *du = vb.get(Hn * D), *ds = vb.get(Hn * D), *dr = vb.get(Hn * D), *dtk = vb.get(Hn * D);
    float *dVa = vb.get(64), *dVb = vb.get(64);  // ping-pong of the small MLPs' backward (hidden <= 64)
    float *Mj = vb.get((int64_t)h0m * D), *sj = vb.get(h0m), *dw2 = vb.get(h0m);
    // The node encoder's weight rows (F = 23 floats: unaligned scalar loads, 23 per thread) are staged, padded to XPAD, where
    // encode_nodes reads them: 7.8 -> 3.8 us.  (The same for the GCN layers' aligned 16-float rows bought nothing.)  They live in
    // the partial-sum scratch, which no phase between the staging and encode_nodes uses (forward: the first phases; backward:
    // staged again next to the raw features) -- 1.5 KB that the largest DHM graphs (397 nodes) need to stay in one workgroup's LDS.
    float *weS = part;                           // row c = We[c][0 .. F) then zeros   (D * XPAD <= part_floats)

    // =============================================================================== forward
    T_MARK(0);
    // lists, per-sample inputs, the raw node features
    // (global -> LDS copies: a thread's loads -- up to four, a quarter of the range apart -- are all issued before its first store;
    // a plain strided loop waits out one HBM round trip per iteration, and the 27 KB of raw features were six of them)
    auto copy4 = [&](auto *dst, const auto *src, int cnt) {
        const int Q = (cnt + 3) / 4;
        T_FOR(i, Q) {
            typename std::remove_cv<typename std::remove_reference<decltype(*src)>::type>::type v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = src[i + q * Q < cnt ? i + q * Q : cnt - 1];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (i + q * Q < cnt) dst[i + q * Q] = v[q];
        }
    };
    auto stage_x = [&]() { copy4(reinterpret_cast<f4 *>(Xs), reinterpret_cast<const f4 *>(Xg), n * (XPAD / 4)); };      // (rows of 96 B)
    const uint32_t *nbg2 = reinterpret_cast<const uint32_t *>(nbg);      // (two u16 ids per word; inc = 2 e)
    uint32_t *nb2 = reinterpret_cast<uint32_t *>(nb);
    const f4 *Xg4 = reinterpret_cast<const f4 *>(Xg);
    f4 *Xs4 = reinterpret_cast<f4 *>(Xs);
    const int nx4 = n * (XPAD / 4);
#ifdef TINY_HOST
    for (int i = 0; i < n + 1; ++i) rp[i] = rpg[i];
    for (int i = 0; i < e; ++i) nb2[i] = nbg2[i];
    for (int i = 0; i < n; ++i) ord[i] = og[i];
    for (int i = 0; i < n; ++i) nm[i] = nmg[i];
    for (int i = 0; i < nx4; ++i) Xs4[i] = Xg4[i];
    for (int i = 0; i < D * XPAD; ++i) weS[i] = i % XPAD < F ? prm[o.node_w + (int64_t)(i / XPAD) * F + i % XPAD] : 0.0f;
    for (int i = 0; i < d.Fn; ++i) U[0][i] = A.numerical[(int64_t)t * d.Fn + i];
    for (int i = 0; i < XPAD; ++i) cur[i] = A.cur[(int64_t)t * XPAD + i];
#else
    {
        // EVERY global load of the staging phase is issued before the first LDS store.  Written as one copy loop per array, each loop
        // waited for its own loads before storing -- eight dependent HBM round trips, 38 us of a 250 us graph (barrier trace,
        // tools/lab_trace).  A thread takes elements tid, tid + NT, ... of each array (clamped loads, masked stores); what an array
        // has beyond the fixed count goes through a plain loop afterwards (never at the shipped sizes).
        const int tid = t_tid(), NT = (int)blockDim.x;
        constexpr int K_RP = 2, K_NB = 4, K_ORD = 2, K_X = 4, K_WE = 2;
        int r_rp[K_RP];
        uint32_t r_nb[K_NB];
        uint16_t r_ord[K_ORD];
        uint8_t r_nm[K_ORD];
        f4 r_x[K_X];
        float r_we[K_WE], r_u = 0.0f, r_cur = 0.0f;
        auto at = [&](int k, int cnt) { const int idx = tid + k * NT; return idx < cnt ? idx : (cnt > 0 ? cnt - 1 : 0); };
#pragma unroll
        for (int k = 0; k < K_RP; ++k) r_rp[k] = rpg[at(k, n + 1)];
#pragma unroll
        for (int k = 0; k < K_NB; ++k) r_nb[k] = nbg2[at(k, e)];
#pragma unroll
        for (int k = 0; k < K_ORD; ++k) {
            r_ord[k] = og[at(k, n)];
            r_nm[k] = nmg[at(k, n)];
        }
#pragma unroll
        for (int k = 0; k < K_X; ++k) r_x[k] = Xg4[at(k, nx4)];
#pragma unroll
        for (int k = 0; k < K_WE; ++k) {
            const int idx = at(k, D * XPAD), c = idx / XPAD, f = idx - c * XPAD;
            r_we[k] = prm[o.node_w + (int64_t)c * F + (f < F ? f : 0)];
            if (f >= F) r_we[k] = 0.0f;
        }
        if (tid < d.Fn) r_u = A.numerical[(int64_t)t * d.Fn + tid];
        if (tid < XPAD) r_cur = A.cur[(int64_t)t * XPAD + tid];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < K_RP; ++k)
            if (tid + k * NT < n + 1) rp[tid + k * NT] = r_rp[k];
#pragma unroll
        for (int k = 0; k < K_NB; ++k)
            if (tid + k * NT < e) nb2[tid + k * NT] = r_nb[k];
#pragma unroll
        for (int k = 0; k < K_ORD; ++k)
            if (tid + k * NT < n) {
                ord[tid + k * NT] = r_ord[k];
                nm[tid + k * NT] = r_nm[k];
            }
#pragma unroll
        for (int k = 0; k < K_X; ++k)
            if (tid + k * NT < nx4) Xs4[tid + k * NT] = r_x[k];
#pragma unroll
        for (int k = 0; k < K_WE; ++k)
            if (tid + k * NT < D * XPAD) weS[tid + k * NT] = r_we[k];
        if (tid < d.Fn) U[0][tid] = r_u;
        if (tid < XPAD) cur[tid] = r_cur;
        for (int i = tid + K_RP * NT; i < n + 1; i += NT) rp[i] = rpg[i];
        for (int i = tid + K_NB * NT; i < e; i += NT) nb2[i] = nbg2[i];
        for (int i = tid + K_ORD * NT; i < n; i += NT) {
            ord[i] = og[i];
            nm[i] = nmg[i];
        }
        for (int i = tid + K_X * NT; i < nx4; i += NT) Xs4[i] = Xg4[i];
        for (int i = tid + NT; i < d.Fn; i += NT) U[0][i] = A.numerical[(int64_t)t * d.Fn + i];
    }
#endif
    T_FOR(i, MAXL + 1) bad[i] = 0;
    T_SYNC();
    // ---- per-sample chains of the forward, as SIDE work of the graph phases below (one layer per phase, in this order):
    //   numerical encoder (state_encoder.py:35-57,187);  current node through the node encoder (:191);  attention query path
    //   (:150-156 + MultiheadAttention's q projection);  r_h = Wk^T (Wik[head rows]^T q1[head rows]):  score_j = r_h . h_j
    //   (key-side biases are softmax-shift invariant)
    const float scale = 1.0f / sqrtf((float)dh);
    int side_k = 0;
    const int side_n = d.n_num + 5;
    auto side_step = [&]() {
        const int k = side_k++;
#pragma unroll
        for (int i = 0; i < MAXMLP; ++i)
            if (k == i && i < d.n_num) {
                const int K = i == 0 ? d.Fn : d.num_hidden[i > 0 ? i - 1 : 0];
                T_FOR_SIDE(jj, d.num_hidden[i])
                    U[i + 1][jj] = t_tanh(dot_g(prm + o.num_w[i] + (int64_t)jj * K, U[i], K, prm[o.num_b[i] + jj]));
            }
        const int kc = k - d.n_num;
        if (kc == 0) T_FOR_SIDE(jj, D) C[jj] = dot_g(prm + o.node_w + (int64_t)jj * F, cur, F, prm[o.node_b + jj]);
        if (kc == 1) T_FOR_SIDE(jj, D) q0[jj] = dot_g(prm + o.q_w + (int64_t)jj * D, C, D, prm[o.q_b + jj]);
        if (kc == 2) T_FOR_SIDE(jj, D) q1[jj] = dot_g(prm + o.inproj_w + (int64_t)jj * D, q0, D, prm[o.inproj_b + jj]) * scale;
        if (kc == 3) T_FOR_SIDE(ii, Hn * D) {
            const int h = ii / D, kk = ii - h * D;
            tk[ii] = dot_t(prm + o.inproj_w + (int64_t)D * D, D, kk, q1, h * dh, (h + 1) * dh);
        }
        if (kc == 4) T_FOR_SIDE(ii, Hn * D) {
            const int h = ii / D, kk = ii - h * D;
            rr[ii] = dot_t(prm + o.k_w, D, kk, tk + h * D, 0, D);
        }
    };
    T_MARK(1);
    T_MARK(2);

    // node encoder on every node (:189-190): H^0 from the staged raw features; thread = one output column, its weight row in registers
    const float *be = prm + o.node_b;
    auto encode_nodes = [&](float *dst, bool side) {
        if (side) side_step();
        T_FORM_J(c, D) {
            float w[XPAD];
#pragma unroll
            for (int f = 0; f < XPAD; ++f) w[f] = weS[c * XPAD + f];
            const float bc = be[c];
            T_FORM_V(v, n, D) {
                float x[XPAD];
                ld_row<XPAD>(Xs + (int64_t)v * XPAD, x);
                float acc = bc;
#pragma unroll
                for (int f = 0; f < XPAD; ++f) acc = fmaf(w[f], x[f], acc);
                dst[v * D + c] = acc;
            }
        }
        T_SYNC();
    };
    encode_nodes(slotH(1), true);                // layer 1 updates it in place
    T_MARK(3);
    for (int l = 1; l <= L; ++l) {
        const float *Hin = l == 1 ? slotH(1) : slotH(l - 1);
        float *Hout = slotH(l);
        const float *Wl = prm + o.edge_w[l - 1], *bl = prm + o.edge_b[l - 1];
        const bool last = l == L;
        // P | Q of the layer: PQ[v][j] = sum_k Wl[j % D][(j / D) * D + k] Hin[v][k]   (linear_0.weight is [D][2D] = [Wa | Wb])
        side_step();
        T_FORM_J(j, 2 * D) {
            float w[D];
            const float *wr = Wl + (int64_t)(j % D) * (2 * D) + (j / D) * D;
#pragma unroll
            for (int k = 0; k < D; ++k) w[k] = wr[k];
            float mx = fabsf(bl[j % D]) * (EF_LIMIT / EF_BIAS);          // (the bias limit folded into the same test)
            T_FORM_V(v, n, 2 * D) {
                float h[D];
                ld_row<D>(Hin + (int64_t)v * D, h);
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < D; ++k) acc = fmaf(w[k], h[k], acc);
                PQ[v * 2 * D + j] = acc;
                mx = fmaxf(mx, fabsf(acc));
            }
            if (!(C2 * mx <= EF_LIMIT)) T_FLAG(bad + l);
        }
        T_SYNC();
        const bool ef = bad[l] == 0;
        if (ef) {                                // exp form in place: the walks below then need no exponential per incidence
            T_FOR(i, n * 2 * D) PQ[i] = t_exp2(C2 * PQ[i]);
            T_SYNC();
        }
        // node-centric segment sum (:110-148): S_v = sum over incidences 1/2 [tanh(P_v + Q_u + b) + tanh(P_u + Q_v + b)]
        //   exp form: 1/2 (t1 + t2) = 1 - (r1 + r2),  r = 1 / (1 + E)
        side_step();
        T_FORM_J(c, D) {
            const float bc = bl[c], eb = t_exp2(C2 * bc);
            T_FORM_V(vi, n, D) {
                const int v = ord[vi];
                const int k0 = rp[v], k1 = rp[v + 1];
                float S = 0.0f;
                // four incidences per trip: their neighbour ids, then their rows, are requested together (the walk of a hub is a
                // chain of dependent LDS round trips); a trip past the end re-reads the last entry and adds exactly 0
                if (ef) {
                    const float pv = PQ[v * 2 * D + c] * eb, qv = PQ[v * 2 * D + D + c] * eb;
                    float acc = 0.0f;
                    for (int k = k0; k < k1; k += 4) {
                        int u[4];
                        float a[4], bq[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) u[q] = nb[k + q < k1 ? k + q : k1 - 1];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            a[q] = PQ[u[q] * 2 * D + D + c];
                            bq[q] = PQ[u[q] * 2 * D + c];
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float r = t_rcp(fmaf(pv, a[q], 1.0f)) + t_rcp(fmaf(bq[q], qv, 1.0f));
                            acc += k + q < k1 ? r : 0.0f;
                        }
                    }
                    S = (float)(k1 - k0) - acc;
                } else {
                    const float pv = PQ[v * 2 * D + c] + bc, qv = PQ[v * 2 * D + D + c] + bc;
                    for (int k = k0; k < k1; k += 4) {
                        int u[4];
                        float a[4], bq[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) u[q] = nb[k + q < k1 ? k + q : k1 - 1];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            a[q] = PQ[u[q] * 2 * D + D + c];
                            bq[q] = PQ[u[q] * 2 * D + c];
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float r = 0.5f * (t_tanh(pv + a[q]) + t_tanh(bq[q] + qv));
                            S += k + q < k1 ? r : 0.0f;
                        }
                    }
                }
                Hout[v * D + c] = fmaf(S, t_rcp((float)(k1 - k0) + 1e-6f), Hin[v * D + c]);
                if (last) Xr[v * D + c] = S;
            }
        }
        T_SYNC();
    }
    T_MARK(4);
    float *HL = slotH(L);
    // masked node mean, edge mean (:179-182,199-200; every message is summed at both endpoints): one two-level pass for both
    side_step();
    T_FORM(i, NG * 2 * D) {
        const int g = i / (2 * D), c = i - g * 2 * D;
        float acc = 0.0f;
        if (c < D) {
            const int cnt = cnt_s(n, g, NG);
            for (int t0 = 0; t0 < cnt; t0 += SB) {
                float x[SB];
                uint8_t k8[SB];
#pragma unroll
                for (int q = 0; q < SB; ++q) {
                    const int v = g + (t0 + q < cnt ? t0 + q : cnt - 1) * NG;
                    x[q] = HL[v * D + c];
                    k8[q] = nm[v];
                }
#pragma unroll
                for (int q = 0; q < SB; ++q) acc += (t0 + q < cnt && k8[q]) ? x[q] : 0.0f;
            }
        } else {
            acc = sum_s(Xr + g * D + c - D, NG * D, cnt_s(n, g, NG), acc);
        }
        part[i] = acc;
    }
    T_SYNC();
    side_step();
    T_FORM(c, 2 * D) {
        const float acc = sum_s(part + c, 2 * D, NG, 0.0f);
        if (c < D) hbarV[c] = acc * (1.0f / (float)m[6]);
        else hbarE[c - D] = acc * (0.5f / (float)e);
    }
    T_SYNC();
    while (side_k < side_n) {                    // (chains longer than the phases above: the rest as phases of their own)
        side_step();
        T_SYNC();
    }
    T_MARK(5);
    // single-query attention over the node_mask nodes (:150-161).  alpha is kept UNNORMALISED (exp(score - max)) until the
    // backward, its normaliser as scal[8 + h]
    for (int h = 0; h < Hn; ++h) {
        float *sch = sc + (int64_t)h * n, *al = alpha + (int64_t)h * n;
        const float *r = rr + h * D;
        T_FOR(v, n) {                            // (columns visited from v % D on: consecutive nodes sit on different LDS banks)
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const int kk = (k + v) & (D - 1);
                acc = fmaf(r[kk], HL[v * D + kk], acc);
            }
            sch[v] = nm[v] ? acc : -INFINITY;
        }
        T_SYNC();
        T_FOR(g, NG) {
            part[g] = max_s(sch + g, NG, cnt_s(n, g, NG), -INFINITY);
        }
        T_SYNC();
        T_FOR(v, n) {                            // (every iteration combines the NG partial maxima itself: no extra phase)
            const float mx = max_s(part, 1, NG, -INFINITY);
            al[v] = nm[v] ? t_exp(sch[v] - mx) : 0.0f;
        }
        T_SYNC();
        // partial sums of alpha (column D) and of alpha * H^L (columns 0..D-1)
        T_FOR(i, NG * (D + 1)) {
            const int g = i / (D + 1), c = i - g * (D + 1);
            float acc = 0.0f;
            if (c < D) acc = dot_s(al + g, NG, HL + g * D + c, NG * D, cnt_s(n, g, NG), acc);
            else acc = sum_s(al + g, NG, cnt_s(n, g, NG), acc);
            part[i] = acc;
        }
        T_SYNC();
        T_FOR(c, D + 1) {
            const float sum = sum_s(part + D, D + 1, NG, 0.0f), acc = sum_s(part + (c < D ? c : 0), D + 1, NG, 0.0f);
            const float inv = 1.0f / sum;
            if (c < D) ss[h * D + c] = acc * inv;
            else scal[8 + h] = inv;
        }
        T_SYNC();
        // u_h = Wv s_h + bv;   o[head rows] = Wiv[head rows] u_h + biv[head rows]
        lin(uu + h * D, ss + h * D, prm + o.v_w, prm + o.v_b, D, D, 0);
    }
    T_MARK(6);
    T_FOR(i, D) ov[i] = dot_g(prm + o.inproj_w + (int64_t)(2 * D + i) * D, uu + (i / dh) * D, D, prm[o.inproj_b + 2 * D + i]);
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
    T_MARK(7);
    // ---- pointer head of the row's stage (policy.py:45-104); candidates only (a masked slot has probability exactly 0)
    // head scratch in the X region: z [nc] | the candidate lists | chunk buffers ([CH][D] inputs, [CH][h0] hidden, [CH][h0] dpre)
    const int ncl = imax(nc, 1);
    float *z = Xr;
    uint16_t *cl_a = reinterpret_cast<uint16_t *>(Xr + a4(ncl));                   // land: src | road: node
    uint16_t *cl_b = cl_a + ncl;                                                   // land: dst
    uint8_t *cl_live = reinterpret_cast<uint8_t *>(cl_b + ncl);
    float *chunk = Xr + head_fixed(ncl);
    const int per = D + 2 * (h0m + 1);
    int CH = (int)((pl.xsize - head_fixed(ncl)) / per);
    CH = imin(imax(CH, 1), ncl);
    const float *PQl = PQ;                       // last layer's P | Q is still in place (the forward wrote it last)
    const float *blL = prm + o.edge_b[L - 1];
    // candidate inputs of a chunk: land = the candidate edge's last-layer message m (0 if not a live edge), road = its node's H^L row
    auto cand_inputs = [&](int q0c, int cn, float *mq) {
        const bool efL = bad[L] == 0;           // (the last layer's P | Q is in exp form)
        T_FOR_J(c, D) {
            const float bc = blL[c], eb = t_exp2(C2 * bc);
            T_FOR_V(qq, cn, D) {
                const int q = q0c + qq;
                float val;
                if (land) {
                    val = 0.0f;
                    if (cl_live[q]) {
                        const int vi = cl_a[q], vj = cl_b[q];
                        if (efL)
                            val = 1.0f - (t_rcp(fmaf(PQl[vi * 2 * D + c] * eb, PQl[vj * 2 * D + D + c], 1.0f)) +
                                          t_rcp(fmaf(PQl[vj * 2 * D + c] * eb, PQl[vi * 2 * D + D + c], 1.0f)));
                        else
                            val = 0.5f * (t_tanh(PQl[vi * 2 * D + c] + PQl[vj * 2 * D + D + c] + bc) +
                                          t_tanh(PQl[vj * 2 * D + c] + PQl[vi * 2 * D + D + c] + bc));
                    }
                } else {
                    val = HL[cl_a[q] * D + c];
                }
                mq[qq * D + c] = val;
            }
        }
        T_SYNC();
    };
    auto cand_hidden = [&](int cn, const float *mq, float *hid) {
        T_FOR_J(j, h0) {
            float a[D];
#pragma unroll
            for (int c = 0; c < D; ++c) a[c] = AeffT[c * h0 + j];
            const float cj = cst[j];
            T_FOR_V(q, cn, h0) {
                float x[D];
                ld_row<D>(mq + (int64_t)q * D, x);
                float acc = cj;
#pragma unroll
                for (int c = 0; c < D; ++c) acc = fmaf(a[c], x[c], acc);
                hid[q * hs + j] = t_tanh(acc);
            }
        }
        T_SYNC();
    };
    if (land || road) {
        T_FOR(q, nc) {
            if (land) {
                cl_a[q] = hsrc[q];
                cl_b[q] = hdst[q];
                cl_live[q] = hlive[q];
            } else {
                cl_a[q] = rnn[q];
            }
        }
        if (land) {
            // W1 [m; c; m*c; m-c] = ((Wa + Wd) + Wc diag(c)) m + (Wb - Wd) c: per-graph effective weight + per-graph bias
            const float *W1 = prm + o.land_w0;
            T_FOR(i, h0 * D) {
                const int j = i / D, c = i - j * D;
                const float *w = W1 + (int64_t)j * 4 * D;
                const float av = w[c] + w[3 * D + c] + w[2 * D + c] * C[c];
                Aeff[i] = av;
                AeffT[c * h0 + j] = av;
            }
            T_FOR(j, h0) {
                const float *w = W1 + (int64_t)j * 4 * D;
                float acc = prm[o.land_b0 + j];
#pragma unroll
                for (int c = 0; c < D; ++c) acc = fmaf(w[D + c] - w[3 * D + c], C[c], acc);
                cst[j] = acc;
                w2v[j] = prm[o.land_w1 + j];
            }
        } else {
            T_FOR(i, h0 * D) {
                const float av = prm[o.road_w0 + i];
                Aeff[i] = av;
                AeffT[(i % D) * h0 + i / D] = av;
            }
            T_FOR(j, h0) {
                cst[j] = prm[o.road_b0 + j];
                w2v[j] = prm[o.road_w1 + j];
            }
        }
        T_SYNC();
        T_MARK(21);
        for (int c0 = 0; c0 < nc; c0 += CH) {
            const int cn = imin(nc - c0, CH);
            float *mq = chunk, *hid = chunk + (int64_t)CH * D;      // [CH][D] inputs, [CH][hs] hidden
            cand_inputs(c0, cn, mq);
            if (c0 == 0) T_MARK(22);
            cand_hidden(cn, mq, hid);
            if (c0 == 0) T_MARK(23);
            T_FOR(q, cn) {
                z[c0 + q] = dot_s(w2v, 1, hid + q * hs, 1, h0, 0.0f);
            }
            T_SYNC();
        }
        T_MARK(24);
        // log-softmax over the candidates, log-prob of the action, entropy
        T_FOR(g, NG) {
            part[g] = max_s(z + g, NG, cnt_s(nc, g, NG), -INFINITY);
        }
        T_SYNC();
        T_FOR(g, NG) {
            const float mx = max_s(part, 1, NG, -INFINITY);
            float sum = 0.0f;
            const int cnt = cnt_s(nc, g, NG);
            for (int t0 = 0; t0 < cnt; t0 += SB) {
                float x[SB];
#pragma unroll
                for (int q = 0; q < SB; ++q) x[q] = z[g + (t0 + q < cnt ? t0 + q : cnt - 1) * NG];
#pragma unroll
                for (int q = 0; q < SB; ++q) sum += t0 + q < cnt ? t_exp(x[q] - mx) : 0.0f;
            }
            part[NG + g] = sum;
            if (g == 0) scal[0] = mx;
        }
        T_SYNC();
        T_FOR(g, NG) {
            const float sum = sum_s(part + NG, 1, NG, 0.0f);
            const float lse = scal[0] + t_log(sum);
            float pz = 0.0f;
            const int cnt = cnt_s(nc, g, NG);
            for (int t0 = 0; t0 < cnt; t0 += SB) {
                float x[SB];
#pragma unroll
                for (int q = 0; q < SB; ++q) x[q] = z[g + (t0 + q < cnt ? t0 + q : cnt - 1) * NG];
#pragma unroll
                for (int q = 0; q < SB; ++q) {
                    const float lp = x[q] - lse;
                    pz += t0 + q < cnt ? t_exp(lp) * lp : 0.0f;
                }
            }
            part[2 * NG + g] = pz;
            if (g == 0) scal[1] = lse;
        }
        T_SYNC();
        T_FOR(g, 1) {
            const float pz = sum_s(part + 2 * NG, 1, NG, 0.0f);
            scal[2] = -pz;
            scal[3] = (act >= 0 ? z[act] : PAD_LOGIT) - scal[1];
        }
        float *zout = land ? A.z_he : A.z_rn;
        if (zout) {
            const int64_t zo = land ? A.he_off[b] : A.rn_off[b];
            T_FOR(q, nc) zout[zo + q] = z[q];
        }
        T_SYNC();
    } else {
        // a row of another stage, or without any valid candidate (every logit is the pad constant, whose logsumexp is
        // absorbed in fp32): log-prob = entropy = 0 (policy.py:90-91)
        T_FOR(g, 1) {
            scal[1] = 0.0f;
            scal[2] = 0.0f;
            scal[3] = 0.0f;
        }
        T_SYNC();
    }
    T_MARK(8);
    const float value = V[d.n_value][0], logp = scal[3], entr = scal[2];
    if (A.mode != BWD) {
        T_FOR(g, 1) {
            A.value[b] = value;
            A.logp[b] = logp;
            A.ent[b] = entr;
        }
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
        T_FOR(g, 1) {
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
    T_MARK(9);
    float *G = slab;                             // gradient slab of this workgroup (parameter layout)
    // ---- value head (value.py:15-39): one phase per layer
    {
        float *dz = dVa, *dn = dVb;
        T_FOR(g, 1) dz[0] = gv;
        T_SYNC();
        for (int i = d.n_value - 1; i >= 0; --i) {
            const int N = d.value_hidden[i], K = i == 0 ? d.W : d.value_hidden[i - 1];
            mlp_layer_bwd(G + o.value_w[i], G + o.value_b[i], i == 0 ? dSV : dn, dz, V[i], prm + o.value_w[i], i == 0 ? nullptr : V[i],
                          N, K);
            float *tmp = dz;
            dz = dn;
            dn = tmp;
        }
    }
    const float *dhbarV = dSV + d.S_last, *dhbarE = dSV + d.S_last + D, *datt = dSV + d.S_last + 2 * D;
    T_MARK(10);
    // ---- numerical encoder
    {
        float *dz = dVa, *dn = dVb;
        T_FOR(j, d.S_last) dz[j] = dSV[j] * (1.0f - U[d.n_num][j] * U[d.n_num][j]);
        T_SYNC();
        for (int i = d.n_num - 1; i >= 0; --i) {
            const int N = d.num_hidden[i], K = i == 0 ? d.Fn : d.num_hidden[i - 1];
            mlp_layer_bwd(G + o.num_w[i], G + o.num_b[i], i == 0 ? nullptr : dn, dz, U[i], prm + o.num_w[i], i == 0 ? nullptr : U[i], N,
                          K);
            float *tmp = dz;
            dz = dn;
            dn = tmp;
        }
    }
    T_MARK(11);
    // ---- attention, the part behind the softmax: out-projection (weight gradient + do in one phase), value projections
    mlp_layer_bwd(G + o.outproj_w, G + o.outproj_b, dov, datt, ov, prm + o.outproj_w, nullptr, D, D);
    // o[i] = Wiv[i] . u_h(i) + biv[i]:  dbiv += do;  dWiv[i][k] += do[i] u_h(i)[k];  du_h[k] = sum_{i in h} Wiv[i][k] do[i]
    T_FOR(i, D * D + D + Hn * D) {
        if (i < D * D) {
            const int r = i / D, k = i - r * D;
            G[o.inproj_w + (int64_t)(2 * D + r) * D + k] += dov[r] * uu[(r / dh) * D + k];
        } else if (i < D * D + D) {
            G[o.inproj_b + 2 * D + i - D * D] += dov[i - D * D];
        } else {
            const int ii = i - D * D - D, h = ii / D, k = ii - h * D;
            du[ii] = dot_t(prm + o.inproj_w + (int64_t)2 * D * D, D, k, dov, h * dh, (h + 1) * dh);
        }
    }
    T_SYNC();
    // u_h = Wv s_h + bv:  dbv += sum_h du_h;  dWv[j][k] += sum_h du_h[j] s_h[k];  ds_h = Wv^T du_h
    T_FOR(i, D * D + D + Hn * D) {
        if (i < D * D) {
            const int j = i / D, k = i - j * D;
            float acc = 0.0f;
            for (int h = 0; h < Hn; ++h) acc += du[h * D + j] * ss[h * D + k];
            G[o.v_w + i] += acc;
        } else if (i < D * D + D) {
            const int j = i - D * D;
            float acc = 0.0f;
            for (int h = 0; h < Hn; ++h) acc += du[h * D + j];
            G[o.v_b + j] += acc;
        } else {
            const int ii = i - D * D - D, h = ii / D, k = ii - h * D;
            ds[ii] = dot_t(prm + o.v_w, D, k, du + h * D, 0, D);
        }
    }
    T_SYNC();
    T_MARK(12);
    // ---- attention core per head: t_j = ds . h_j, T = sum alpha t, dscore_j = alpha_j (t_j - T), dr = sum_j dscore_j h_j
    //      (alpha is normalised here: alpha[] held exp(score - max), scal[8 + h] its normaliser)
    for (int h = 0; h < Hn; ++h) {
        float *tj = sc + (int64_t)h * n, *al = alpha + (int64_t)h * n;
        const float *dsv = ds + h * D;
        const float inv = scal[8 + h];
        T_FOR(v, n) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < D; ++k) {
                const int kk = (k + v) & (D - 1);
                acc = fmaf(dsv[kk], HL[v * D + kk], acc);
            }
            tj[v] = acc;
            al[v] *= inv;
        }
        T_SYNC();
        T_FOR(g, NG) {
            part[g] = dot_s(al + g, NG, tj + g, NG, cnt_s(n, g, NG), 0.0f);
        }
        T_SYNC();
        T_FOR(v, n) {
            const float T = sum_s(part, 1, NG, 0.0f);
            tj[v] = al[v] * (tj[v] - T);         // dscore_j (0 on nodes outside the mask: alpha = 0)
        }
        T_SYNC();
        T_FOR(i, NG * D) {
            const int g = i / D, c = i - g * D;
            part[i] = dot_s(tj + g, NG, HL + g * D + c, NG * D, cnt_s(n, g, NG), 0.0f);
        }
        T_SYNC();
        T_FOR(c, D) {
            dr[h * D + c] = sum_s(part + c, D, NG, 0.0f);
        }
        T_SYNC();
    }
    T_MARK(13);
    // ---- pointer head (before H^L is overwritten by G^L): dz, second + first Linear, candidate inputs
    T_FOR(i, h0m * D) Mj[i] = 0.0f;
    T_FOR(i, h0m) {
        sj[i] = 0.0f;
        dw2[i] = 0.0f;
    }
    T_FOR(i, D) dC[i] = 0.0f;
    T_SYNC();
    float *dMg = gscr;                           // land rows: [nc][D] gradient of the candidates' messages (global scratch)
    float *dXR = PQ;                             // road rows only: [nc][D] gradient of the candidates' H^L rows (P | Q is free)
    if (land || road) {
        const float lse = scal[1], Hent = scal[2];
        for (int c0 = 0; c0 < nc; c0 += CH) {
            const int cn = imin(nc - c0, CH);
            float *mq = chunk, *hid = chunk + (int64_t)CH * D, *dpre = chunk + (int64_t)CH * (D + hs);
            if (c0 == 0) T_MARK(25);
            cand_inputs(c0, cn, mq);
            cand_hidden(cn, mq, hid);
            if (c0 == 0) T_MARK(26);
            // dz_k = dlogp (delta_ka - p_k) - dent p_k (log p_k + H);  dpre[k][j] = dz_k w2[j] (1 - hid^2)
            T_FOR(i, cn * h0) {
                const int q = i / h0, j = i - q * h0;
                const float lp = z[c0 + q] - lse, p = t_exp(lp);
                float dz = -gl * p - ge * p * (lp + Hent);
                if (c0 + q == act) dz += gl;
                const float hv = hid[q * hs + j];
                dpre[q * hs + j] = dz * w2v[j] * (1.0f - hv * hv);
                hid[q * hs + j] = dz * hv;        // (dz hid: the summand of dw2)
            }
            T_SYNC();
            if (c0 == 0) T_MARK(27);
            // running sums over the candidates: dw2, db1 (= s), M[j][c] = sum dpre[k][j] m[k][c];  and the gradient of the
            // candidate inputs dm[k][c] = sum_j A[j][c] dpre[k][j]  (land: only live candidates carry it on)
            T_FOR(i, h0 * D + 2 * h0) {                 // (the two plain sums behind the products: waves of their own, one sum each)
                if (i < h0 * D) {
                    const int j = i / D, c = i - j * D;
                    Mj[i] = dot_s(dpre + j, hs, mq + c, D, cn, Mj[i]);
                } else if (i < h0 * D + h0) {
                    const int j = i - h0 * D;
                    dw2[j] = sum_s(hid + j, hs, cn, dw2[j]);
                } else {
                    const int j = i - h0 * D - h0;
                    sj[j] = sum_s(dpre + j, hs, cn, sj[j]);
                }
            }
            float *dst = land ? dMg : dXR;
            T_FOR_J(c, D) {
                T_FOR_V(q, cn, D) {
                    float acc = dot_s(Aeff + c, D, dpre + q * hs, 1, h0, 0.0f);
                    if (land && !cl_live[c0 + q]) acc = 0.0f;
                    dst[(int64_t)(c0 + q) * D + c] = acc;
                }
            }
            T_SYNC();
            if (c0 == 0) T_MARK(28);
        }
        T_MARK(29);
        if (land) {
            // feat = [m; c; m*c; m-c]:  dWa += M,  dWb += s (x) c,  dWc += M * c,  dWd += M - s (x) c;  db1 += s;  dw2
            // dC[c] += sum_j (Wb - Wd)[j][c] s[j] + Wc[j][c] M[j][c]
            const float *W1 = prm + o.land_w0;
            T_FOR(i, h0 * D + h0 + D) {
                if (i < h0 * D) {
                    const int j = i / D, c = i - j * D;
                    float *g = G + o.land_w0 + (int64_t)j * 4 * D;
                    const float Mv = Mj[i], sc_ = sj[j] * C[c];
                    g[c] += Mv;
                    g[D + c] += sc_;
                    g[2 * D + c] += Mv * C[c];
                    g[3 * D + c] += Mv - sc_;
                } else if (i < h0 * D + h0) {
                    const int j = i - h0 * D;
                    G[o.land_b0 + j] += sj[j];
                    G[o.land_w1 + j] += dw2[j];
                } else {
                    const int c = i - h0 * D - h0;
                    float acc = 0.0f;
                    for (int j0 = 0; j0 < h0; j0 += SB) {       // (h0 % 16 == 0; the weights come from L2: all of a batch in flight)
                        float wb[SB], wd[SB], wc[SB], s8[SB], m8[SB];
#pragma unroll
                        for (int q = 0; q < SB; ++q) {
                            const float *w = W1 + (int64_t)(j0 + q) * 4 * D;
                            wb[q] = w[D + c];
                            wd[q] = w[3 * D + c];
                            wc[q] = w[2 * D + c];
                            s8[q] = sj[j0 + q];
                            m8[q] = Mj[(j0 + q) * D + c];
                        }
#pragma unroll
                        for (int q = 0; q < SB; ++q) acc += (wb[q] - wd[q]) * s8[q] + wc[q] * m8[q];
                    }
                    dC[c] = acc;
                }
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
    T_MARK(14);
    // ---- G^L in place of H^L: masked-mean share + attention terms (+ the road candidates' rows)
    T_FOR_J(c, D) {
        const float gm = dhbarV[c] / (float)m[6];
        T_FOR_V(v, n, D) {
            float g = nm[v] ? gm : 0.0f;
            for (int h = 0; h < Hn; ++h) g += alpha[(int64_t)h * n + v] * ds[h * D + c] + sc[(int64_t)h * n + v] * rr[h * D + c];
            HL[v * D + c] = g;
        }
    }
    T_SYNC();
    if (road) {
        T_FOR(i, nc * D) {                       // road_mask slots are distinct nodes: one writer per element
            const int q = i / D, c = i - q * D;
            HL[cl_a[q] * D + c] += dXR[i];
        }
        T_SYNC();
    }
    float *Gn = HL;
    T_MARK(15);
    // ---- attention, the query chain:  r_h = Wk^T tk_h,  tk_h = Wik[head rows]^T q1[head rows]
    T_FOR(i, Hn * D) {                           // dtk_h[j] = sum_d Wk[j][d] dr_h[d]
        const int h = i / D, j = i - h * D;
        dtk[i] = dot_g(prm + o.k_w + (int64_t)j * D, dr + h * D, D);
    }
    T_SYNC();
    T_FOR(i, 2 * D * D + D) {
        if (i < D * D) {                         // dWk[j][d] += sum_h tk_h[j] dr_h[d]
            const int j = i / D, k = i - j * D;
            float acc = 0.0f;
            for (int h = 0; h < Hn; ++h) acc += tk[h * D + j] * dr[h * D + k];
            G[o.k_w + i] += acc;
        } else if (i < 2 * D * D) {              // dWik[r][j] += q1[r] dtk_h(r)[j]
            const int ii = i - D * D, r = ii / D, j = ii - r * D;
            G[o.inproj_w + (int64_t)(D + r) * D + j] += q1[r] * dtk[(r / dh) * D + j];
        } else {                                 // dq1[r] = sum_j Wik[r][j] dtk_h(r)[j], through q1 = (Wiq q0 + biq) * scale
            const int r = i - 2 * D * D;
            dq1[r] = dot_g(prm + o.inproj_w + (int64_t)(D + r) * D, dtk + (r / dh) * D, D) * scale;
        }
    }
    T_SYNC();
    mlp_layer_bwd(G + o.inproj_w, G + o.inproj_b, dq0, dq1, q0, prm + o.inproj_w, nullptr, D, D);
    // q0 = Wq C + bq:  dWq, dbq;  dC += Wq^T dq0 (on top of the land-use head's share)
    T_FOR(i, D * D + 2 * D) {
        if (i < D * D) {
            const int j = i / D, k = i - j * D;
            G[o.q_w + i] += dq0[j] * C[k];
        } else if (i < D * D + D) {
            G[o.q_b + i - D * D] += dq0[i - D * D];
        } else {
            const int c = i - D * D - D;
            dC[c] += dot_t(prm + o.q_w, D, c, dq0, 0, D);
        }
    }
    T_SYNC();
    // current node's pass through the node encoder
    outer_acc(G + o.node_w, G + o.node_b, dC, cur, D, F);
    T_MARK(16);

    // ---- GCN layers, last to first (:110-148,194-197).  Per layer in two column halves: P | Q of the half in PQ[0 .. nD),
    // dP | dQ of the half in PQ[nD .. 2 nD).  Row layout of a half: [v][0 .. D/2) = P columns, [v][D/2 .. D) = Q columns.
    constexpr int HC = D / 2;
    constexpr int NGW = D == 16 ? 4 : 1;         // node groups of the weight-gradient partial sums (NGW * D * D partials)
    float *PQh = PQ, *dPQh = PQ + nD, *dS = Xr;
    for (int l = L; l >= 1; --l) {
        const bool last = l == L;
        float *Hprev;
        if (l == L - 1) T_MARK(17);
        if (l == 1) {
            Hprev = slotH(1);                    // H^1 is dead (layer 2 is done): recompute H^0 in its place
            stage_x();
            T_FOR(i, D * XPAD) weS[i] = i % XPAD < F ? prm[o.node_w + (int64_t)(i / XPAD) * F + i % XPAD] : 0.0f;
            T_SYNC();
            encode_nodes(Hprev, false);
        } else {
            Hprev = slotH(l - 1);
        }
        const float *Wl = prm + o.edge_w[l - 1], *bl = prm + o.edge_b[l - 1];
        float *gW = G + o.edge_w[l - 1], *gB = G + o.edge_b[l - 1];
        // dS_v = G_v / (deg_v + 1e-6) (+ 1/2 dhbarE / e on the last layer)
        T_FOR_J(c, D) {
            const float ex = last ? 0.5f * dhbarE[c] / (float)e : 0.0f;
            T_FOR_V(v, n, D) dS[v * D + c] = fmaf(Gn[v * D + c], t_rcp((float)(rp[v + 1] - rp[v]) + 1e-6f), ex);
        }
        T_SYNC();
        if (last) T_MARK(30);
        const bool ef = bad[l] == 0;             // the forward's verdict on this layer's P | Q (same values: same form)
        for (int half = 0; half < 2; ++half) {
            const int cb = half * HC;
            // P | Q of the half (thread = one of its D columns, weight row in registers), in exp form where the layer allows it
            T_FOR_J(jj, D) {
                float w[D];
                const float *wr = Wl + (int64_t)(cb + jj % HC) * (2 * D) + (jj / HC) * D;
#pragma unroll
                for (int k = 0; k < D; ++k) w[k] = wr[k];
                T_FOR_V(v, n, D) {
                    float h[D];
                    ld_row<D>(Hprev + (int64_t)v * D, h);
                    float acc = 0.0f;
#pragma unroll
                    for (int k = 0; k < D; ++k) acc = fmaf(w[k], h[k], acc);
                    PQh[v * D + jj] = ef ? t_exp2(C2 * acc) : acc;
                }
            }
            T_SYNC();
            if (last && half == 0) T_MARK(31);
            // dP_v = sum_u 1/2 dm (1 - tanh^2(P_v + Q_u + b)),  dQ_v = sum_u 1/2 dm (1 - tanh^2(P_u + Q_v + b)),
            // dm = dS_v + dS_u; the row's candidate edges add their head gradient on the last layer (packer's
            // candidate-incidence lists: neighbour + candidate index per incident live candidate).
            // exp form: 1 - tanh^2 = 4 (r - r^2) with r = 1 / (1 + E)
            T_FOR_J(cc, HC) {
                const int c = cb + cc;
                const float bc = bl[c], eb = t_exp2(C2 * bc);
                T_FOR_V(vi, n, HC) {
                    const int v = ord[vi];
                    const float sv = dS[v * D + c];
                    const float pv = ef ? PQh[v * D + cc] * eb : PQh[v * D + cc] + bc;
                    const float qv = ef ? PQh[v * D + HC + cc] * eb : PQh[v * D + HC + cc] + bc;
                    float aP = 0.0f, aQ = 0.0f;
                    // one edge term: the neighbour's Q / P of this column, edge gradient dm (a masked term has dm = 0)
                    auto term = [&](float qu, float pu, float dm) {
                        if (ef) {
                            const float r1 = t_rcp(fmaf(pv, qu, 1.0f)), r2 = t_rcp(fmaf(pu, qv, 1.0f));
                            aP = fmaf(2.0f * dm, fmaf(-r1, r1, r1), aP);
                            aQ = fmaf(2.0f * dm, fmaf(-r2, r2, r2), aQ);
                        } else {
                            const float t1 = t_tanh(pv + qu), t2 = t_tanh(pu + qv);
                            aP = fmaf(0.5f * dm, 1.0f - t1 * t1, aP);
                            aQ = fmaf(0.5f * dm, 1.0f - t2 * t2, aQ);
                        }
                    };
                    const int k0 = rp[v], k1 = rp[v + 1];
                    for (int k = k0; k < k1; k += 4) {      // four incidences per trip (see the forward walk)
                        int u[4];
                        float qu[4], pu[4], du_[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) u[q] = nb[k + q < k1 ? k + q : k1 - 1];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            qu[q] = PQh[u[q] * D + HC + cc];
                            pu[q] = PQh[u[q] * D + cc];
                            du_[q] = dS[u[q] * D + c];
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) term(qu[q], pu[q], k + q < k1 ? sv + du_[q] : 0.0f);
                    }
                    if (last && land) {
                        for (int k = hpg[v]; k < hpg[v + 1]; ++k) {
                            const int u = hnb[k];
                            term(PQh[u * D + HC + cc], PQh[u * D + cc], dMg[(int64_t)hhe[k] * D + c]);
                        }
                    }
                    dPQh[v * D + cc] = aP;
                    dPQh[v * D + HC + cc] = aQ;
                }
            }
            T_SYNC();
            if (last && half == 0) T_MARK(32);
            // partial sums over fixed node groups of the half's weight gradient (rows c of [Wa | Wb]) and bias gradient, and the
            // dgrad in place: G_v[k] += sum_c dP_v[c] Wa[c][k] + dQ_v[c] Wb[c][k]   (the walks read dS, not G)
            T_FOR(i, NGW * D * D + NG * HC) {
                if (i < NGW * D * D) {
                    const int g = i / (D * D), r = i - g * D * D, jj = r / D, k = r - jj * D;
                    part[i] = dot_s(dPQh + g * D + jj, NGW * D, Hprev + g * D + k, NGW * D, cnt_s(n, g, NGW), 0.0f);
                } else {
                    const int ii = i - NGW * D * D, g = ii / HC, cc = ii - g * HC;
                    part[i] = sum_s(dPQh + g * D + cc, NG * D, cnt_s(n, g, NG), 0.0f);
                }
            }
            T_FOR_J(k, D) {
                float w[D];
#pragma unroll
                for (int jj = 0; jj < D; ++jj) w[jj] = Wl[(int64_t)(cb + jj % HC) * (2 * D) + (jj / HC) * D + k];
                T_FOR_V(v, n, D) {
                    float dp[D];
                    ld_row<D>(dPQh + (int64_t)v * D, dp);
                    float acc = Gn[v * D + k];
#pragma unroll
                    for (int jj = 0; jj < D; ++jj) acc = fmaf(dp[jj], w[jj], acc);
                    Gn[v * D + k] = acc;
                }
            }
            T_SYNC();
            if (last && half == 0) T_MARK(33);
            // fixed-order combine into the slab:  dW[c][side * D + k],  db[c] += sum_v dP_v[c]
            T_FOR(i, D * D + HC) {
                if (i < D * D) {
                    const int jj = i / D, k = i - jj * D;
                    float acc = 0.0f;
                    for (int g = 0; g < NGW; ++g) acc += part[g * D * D + i];
                    gW[(int64_t)(cb + jj % HC) * (2 * D) + (jj / HC) * D + k] += acc;
                } else {
                    const int cc = i - D * D;
                    gB[cb + cc] += sum_s(part + NGW * D * D + cc, HC, NG, 0.0f);
                }
            }
            T_SYNC();
            if (last && half == 0) T_MARK(34);
        }
    }
    T_MARK(19);
    // ---- node encoder on every node: dWe += G^0^T X, dbe += colsum(G^0)   (X staged once more: P | Q is dead)
    stage_x();
    T_SYNC();
    constexpr int NGX = 2;                       // node groups of the partial sums
    T_FOR(i, NGX * D * (F + 1)) {
        const int g = i / (D * (F + 1)), r = i - g * D * (F + 1), c = r / (F + 1), f = r - c * (F + 1);
        if (f < F) part[i] = dot_s(Gn + g * D + c, NGX * D, Xs + g * XPAD + f, NGX * XPAD, cnt_s(n, g, NGX), 0.0f);
        else part[i] = sum_s(Gn + g * D + c, NGX * D, cnt_s(n, g, NGX), 0.0f);
    }
    T_SYNC();
    T_FOR(i, D * (F + 1)) {
        const int c = i / (F + 1), f = i - c * (F + 1);
        float acc = 0.0f;
        for (int g = 0; g < NGX; ++g) acc += part[g * D * (F + 1) + i];
        if (f < F) G[o.node_w + (int64_t)c * F + f] += acc;
        else G[o.node_b + c] += acc;
    }
    T_SYNC();
    T_MARK(20);
}

}  // namespace upamd_tiny
