// Engine: sequences the kernels of one minibatch forward / backward on a stream.
// The order of operations mirrors tests/csr_model.py (the executable spec) step by step.
//
// Launch budget (round 2): the per-sample chains, the parameter preparation and every small reduction are fused
// (chain.hip), so a training step is ~45 launches instead of ~165: what remains is one launch per big GEMM / message-
// passing pass plus a handful of grouped kernels.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <unordered_map>
#include <vector>

#include "kernels.h"

using namespace upamd;

// side stream of the forked step (see fork_side) + its fork / join events
struct SideCtx {
    hipStream_t side = nullptr;
    // a second stream of the same (high) priority for the weight-gradient GEMMs of small minibatches (side_wgrad).  Measured at 256
    // rows (profiles/archive/r03_lab_rccl_side_stream.log): at NORMAL priority it is the fastest form without a process group (2.39 ms; at high
    // priority the weight gradient runs ahead of the caller's dgrad GEMM, the one on the critical path: 2.47 ms) -- but with RCCL
    // initialised a normal-priority stream shares the caller's hardware queue and the cross-stream events then stall it (3.08 ms
    // against 2.55 ms without side_wgrad).  High priority is the setting that holds in both worlds (2.46-2.47 ms).
    hipStream_t side2 = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_a = nullptr, ev_b = nullptr, ev_join2 = nullptr;
    hipEvent_t pool[8] = {};             // round-robin events of the finer-grained hand-overs (stream_after / event_on)
    int pool_next = 0;
    // gradient buckets of the LAST backward enqueued on this caller stream (upamd_grad_buckets): float ranges of the flat
    // gradient buffer in the order they became final, each with the event recorded behind its last writer
    static constexpr int MAX_BUCKETS = 20;
    int n_buckets = 0;
    int64_t b_begin[MAX_BUCKETS] = {}, b_end[MAX_BUCKETS] = {};
    hipEvent_t b_ev[MAX_BUCKETS] = {};
};

struct upamd_engine {
    upamd_model_desc d;
    ParamLayout P;
    Profiler prof;
    // one side context per CALLER stream (created on first use, on the device of that stream): two callers that drive
    // the engine on two streams (an action server next to the learner) do not queue their
    // side chains behind each other.  The engine's entry points are not re-entrant: the host wrapper serialises them.
    std::unordered_map<hipStream_t, SideCtx> sides;
    // workspaces whose LAST forward ran the general kernels (i.e. hold the activations the general backward reads).  The fused
    // small-model forward writes none of them: a backward that takes the general path behind it (tune knob tiny_fused flipped in
    // between) would differentiate through stale memory -- refused instead.  (The fused backward recomputes its forward: any order is fine.)
    std::unordered_map<const void *, int> general_fwd;
    bool general_fwd_complete = true;      // false once the table had to be dropped: a missing entry then proves nothing
};

namespace {

// tune knob "side_wgrad": the weight-gradient GEMM of GCN layer l on the side stream next to the same layer's dgrad GEMM (dP|dQ
// alternates between two buffers).  Measured (profiles/archive/r03_lab_side_streams.log): two big GEMMs sharing the matrix pipe lose 2.5 %
// of the 2048-row step, but at <= 256 rows per step neither GEMM fills the chip (2.9 rounds of workgroups) and running them
// together gains 3 % -- so the default (1) applies it to minibatches of at most SIDE_WGRAD_MAX_NODES nodes; 2 = always, with the
// weight gradient BEHIND the dgrad, i.e. next to the next layer's message passing (-2 %: the walk and the GEMM slow each other
// more than they overlap); 3 = always next to the dgrad; 0 = never
static int g_side_wgrad = 1;
constexpr int64_t SIDE_WGRAD_MAX_NODES = 98304;
static bool side_wgrad_on(int64_t M) { return g_side_wgrad >= 2 || (g_side_wgrad == 1 && M <= SIDE_WGRAD_MAX_NODES); }
constexpr int MAXL = 16;
constexpr int MAXK = UPAMD_MAX_EDGE_FC;
static inline int LK(int l, int k) { return (l - 1) * (MAXK + 1) + k; }      // l = 1 .. L, k = 0 .. MAXK

// workspace slots (float offsets computed by make_plan); "+l" / "+i" ranges are indexed by layer / MLP depth
enum Slot : int {
    S_ROWS = 0, S_WE_PAD, S_WKK, S_WKKT, S_WVV, S_WVVT, S_BVV, S_W1F, S_W1FT, S_WBD, S_WBDT, S_R1T, S_W1C, S_B1C, S_WET, S_WQT, S_WIQT,
    S_WOT, S_XP, S_CURG, S_C, S_HBARV, S_HBARE, S_Q0, S_Q1, S_R, S_ALPHA, S_S, S_O, S_ATT, S_SV, S_CONSTB, S_FE, S_HIDL, S_Z_HE, S_P_HE,
    S_XR, S_HIDR, S_Z_RN, S_P_RN, S_LSE, S_ENTK,
    // backward
    S_DSV, S_DATT, S_DO, S_DS, S_DR, S_DQ1, S_DQ0, S_DC, S_DC_HEAD, S_DCONST, S_DWKK, S_DWVV, S_DBVV, S_DW1F, S_DWBD, S_TN, S_DWC1,
    S_DZ_HE, S_DZ_RN, S_DPREL, S_DFE, S_DMHE, S_DPRER, S_DXR, S_G0, S_G1, S_DPQ, S_DPQ2, S_SLAB_SMALL, S_SLAB_XP1, S_SLAB_XP2, S_SLAB_FE,
    S_SLAB_XR, S_CSP0, S_CSP1, S_CSP2, S_CSP3,
    S_WCAT,                               // + l (0 .. L-1)
    S_WCATT = S_WCAT + MAXL,              // + l
    S_H = S_WCATT + MAXL,                 // + l (0 .. L)
    S_PQ = S_H + MAXL + 1,                // + l (1 .. L)
    S_CS = S_PQ + MAXL + 1,               // + l (1 .. L): column sums of dP | dQ
    S_DBIAS = S_CS + MAXL + 1,            // + l (1 .. L): per-graph partial sums of dP | dQ
    S_SLAB_W = S_DBIAS + MAXL + 1,        // + l (2 .. L): split-K slabs of the layer's weight gradient
    S_DPQL = S_SLAB_W + MAXL + 1,         // + l (1 .. L): per-layer dP | dQ, only when the node-level products are deferred
    S_WNT = S_DPQL + MAXL + 1,            // + i: transposed numerical-encoder weights
    S_WVT = S_WNT + UPAMD_MAX_MLP,        // + i: transposed value-head weights
    S_U = S_WVT + UPAMD_MAX_MLP,          // + i (0 .. n_num)
    S_V = S_U + UPAMD_MAX_MLP + 1,        // + i (1 .. n_value - 1)
    S_DAV = S_V + UPAMD_MAX_MLP + 1,      // + i
    S_DAN = S_DAV + UPAMD_MAX_MLP,        // + i
    // ---- edge MLPs with K > 1 sub-layers (deep_edge.hip); "+ lk" = (l - 1) * MAXK + k
    S_EIDX = S_DAN + UPAMD_MAX_MLP,       // int32: gsrc | gdst | grev [NI] each, then cand_inc [NH]
    S_EDA, S_EDB,                         // [NI, D] ping-pong of the backward through the sub-layers
    S_EA,                                 // + lk, k = 1 .. K: the sub-layer activations A_k [NI, D] (kept for the backward)
    S_EWT = S_EA + MAXL * (MAXK + 1),     // + lk, k = 1 .. K-1: W_k^T
    S_EBP = S_EWT + MAXL * (MAXK + 1),    // + lk, k = 1 .. K-1: per-graph column sums of dpre_k+1 [B, D] (bias gradient of linear_k)
    S_ESLAB = S_EBP + MAXL * (MAXK + 1),  // + lk, k = 1 .. K-1: split-K slabs of dW_k
    S_PQF = S_ESLAB + MAXL * (MAXK + 1),  // + l (1 .. L): bytes, exp-form flags of the layer's P/Q GEMM output (kernels.h: PQ_EXP_LIMIT)
    // fused small-model path (tiny.hip): per-workgroup gradient slabs, per-workgroup global scratch, per-row loss terms
    S_TINY_SLAB = S_PQF + MAXL + 1, S_TINY_SCR, S_TINY_LOSS,
    S_COUNT
};

struct Plan {
    int64_t off[S_COUNT];
    int64_t len[S_COUNT];
    int64_t total = 0;                    // floats
    int64_t small_slab_floats = 0;
};

struct Dims {
    int D, L, heads, dh, F, Fn, S_last, W;   // W = width of state_value
    int Wp;                                  // its row stride in the workspace: W padded to a multiple of 16
    int h0l, h0r;                            // hidden sizes of the two pointer heads
    int maxnum, maxval, maxdim;
    bool mlp;                                // rl-mlp encoder (UPAMD_ENCODER_MLP)
    int K;                                   // num_edge_fc_layers (sub-layers of every edge MLP)
};

Dims dims_of(const upamd_model_desc &d) {
    Dims x;
    x.D = d.D; x.L = d.L; x.heads = d.heads; x.dh = d.D / d.heads; x.F = d.node_dim; x.Fn = d.numerical_dim;
    x.mlp = d.encoder == UPAMD_ENCODER_MLP;
    x.K = x.mlp ? 1 : edge_fc_layers(d);
    x.S_last = d.num_hidden[d.n_num - 1];
    x.W = (x.mlp ? 2 : 3) * d.D + x.S_last + 3;
    x.Wp = (x.W + 15) / 16 * 16;
    x.h0l = d.land_hidden[0];
    x.h0r = d.road_hidden[0];
    x.maxnum = x.Fn;
    for (int i = 0; i < d.n_num; ++i) x.maxnum = std::max(x.maxnum, d.num_hidden[i]);
    x.maxval = 1;
    for (int i = 0; i < d.n_value; ++i) x.maxval = std::max(x.maxval, d.value_hidden[i]);
    x.maxdim = std::max(std::max(x.maxnum, x.maxval), d.D);
    return x;
}

ChainDims chain_dims(const upamd_model_desc &d, const Dims &x, int B) {
    ChainDims c;
    memset(&c, 0, sizeof(c));
    c.B = B; c.D = x.D; c.heads = x.heads; c.dh = x.dh; c.F = x.F; c.Fn = x.Fn;
    c.n_num = d.n_num; c.n_value = d.n_value;
    for (int i = 0; i < d.n_num; ++i) c.num_hidden[i] = d.num_hidden[i];
    for (int i = 0; i < d.n_value; ++i) c.value_hidden[i] = d.value_hidden[i];
    c.S_last = x.S_last; c.W = x.W; c.Wp = x.Wp; c.h0l = x.h0l;
    c.maxnum = x.maxnum; c.maxval = x.maxval; c.maxdim = x.maxdim;
    c.mlp = x.mlp ? 1 : 0;
    c.scale = 1.0f / std::sqrt((float)x.dh);
    return c;
}

// Models whose node-level weight-gradient shapes are too small for the tiled MFMA kernel (D < 32: the reference's
// shipped D = 16) are launch-bound: their five dY^T X products per step become jobs of the one grouped launch that
// already computes the per-sample weight gradients.
static bool defer_node_tn(int D) { return !tn_shape_mfma_ok(2 * D, D); }

void make_plan(const upamd_model_desc &d, const ParamLayout &P, const upamd_minibatch &mb, Plan *pl) {
    const Dims x = dims_of(d);
    const int64_t B = mb.B, M = std::max<int64_t>(mb.n_nodes, 1), NH = std::max<int64_t>(mb.n_he, 1),
                  NR = std::max<int64_t>(mb.n_rn, 1);
    const int D = x.D;
    int64_t off = 0;
    for (int s = 0; s < S_COUNT; ++s) { pl->off[s] = -1; pl->len[s] = 0; }
    auto add = [&](int slot, int64_t n) {
        pl->off[slot] = off;
        pl->len[slot] = n;
        off = align_up(off + std::max<int64_t>(n, 1), 64);
    };
    add(S_ROWS, B * UPAMD_META_STRIDE);     // int32 row descriptors (MbView::rows)
    // ---- prepared parameters
    add(S_WE_PAD, (int64_t)D * 32);
    for (int l = 0; l < x.L; ++l) { add(S_WCAT + l, 2LL * D * D); add(S_WCATT + l, 2LL * D * D); }
    add(S_WKK, (int64_t)D * D); add(S_WKKT, (int64_t)D * D); add(S_WVV, (int64_t)D * D); add(S_WVVT, (int64_t)D * D); add(S_BVV, D);
    add(S_W1F, 2LL * D * x.h0l); add(S_W1FT, 2LL * D * x.h0l); add(S_WBD, (int64_t)D * x.h0l); add(S_WBDT, (int64_t)D * x.h0l);
    add(S_R1T, (int64_t)D * x.h0r);
    add(S_W1C, 2LL * D * 32); add(S_B1C, 2LL * D);      // first GCN layer collapsed onto the raw node features
    {
        int prev = x.Fn;
        for (int i = 0; i < d.n_num; ++i) { add(S_WNT + i, (int64_t)prev * d.num_hidden[i]); prev = d.num_hidden[i]; }
        prev = x.W;
        for (int i = 0; i < d.n_value; ++i) { add(S_WVT + i, (int64_t)prev * d.value_hidden[i]); prev = d.value_hidden[i]; }
    }
    add(S_WET, (int64_t)x.F * D); add(S_WQT, (int64_t)D * D); add(S_WIQT, (int64_t)D * D); add(S_WOT, (int64_t)D * D);
    // ---- forward activations
    add(S_XP, 2 * M * 16);
    add(S_U + 0, B * x.Fn);
    for (int i = 0; i < d.n_num; ++i) add(S_U + i + 1, B * d.num_hidden[i]);
    add(S_CURG, (x.mlp ? 3 : 1) * B * UPAMD_NODE_PAD);      // mlp: + rows of mean edge features + zero rows (node-encoder job)
    add(S_C, B * D);
    for (int l = 0; l <= x.L; ++l) add(S_H + l, M * D);
    for (int l = 1; l <= x.L; ++l) add(S_PQ + l, M * 2 * D);
    for (int l = 1; l <= x.L; ++l) add(S_PQF + l, (gemm_exp_flag_bytes(M, 2 * D) + 3) / 4);
    add(S_HBARV, B * D); add(S_HBARE, B * D);
    add(S_Q0, B * D); add(S_Q1, B * D); add(S_R, B * x.heads * D); add(S_ALPHA, (int64_t)x.heads * M);
    add(S_S, B * x.heads * D); add(S_O, B * D); add(S_ATT, B * D);
    add(S_SV, B * x.Wp);
    for (int i = 1; i < d.n_value; ++i) add(S_V + i, B * d.value_hidden[i - 1]);
    add(S_CONSTB, B * x.h0l);
    add(S_FE, NH * 2 * D); add(S_HIDL, NH * x.h0l); add(S_Z_HE, NH); add(S_P_HE, NH);
    add(S_XR, NR * D); add(S_HIDR, NR * x.h0r); add(S_Z_RN, NR); add(S_P_RN, NR);
    add(S_LSE, B); add(S_ENTK, B);
    // ---- backward
    for (int i = 0; i < d.n_value; ++i) add(S_DAV + i, B * d.value_hidden[i]);
    for (int i = 0; i < d.n_num; ++i) add(S_DAN + i, B * d.num_hidden[i]);
    add(S_DSV, B * x.Wp); add(S_DATT, B * D); add(S_DO, B * D); add(S_DS, B * x.heads * D); add(S_DR, B * x.heads * D);
    add(S_DQ1, B * D); add(S_DQ0, B * D); add(S_DC, (x.mlp ? 3 : 1) * B * D); add(S_DC_HEAD, B * D); add(S_DCONST, B * x.h0l);
    add(S_DWKK, (int64_t)D * D); add(S_DWVV, (int64_t)D * D); add(S_DBVV, D);
    add(S_DW1F, 2LL * D * x.h0l); add(S_DWBD, (int64_t)D * x.h0l); add(S_TN, 2LL * D * 32); add(S_DWC1, 2LL * D * D);
    for (int l = 1; l <= x.L; ++l) { add(S_CS + l, 2LL * D); add(S_DBIAS + l, B * 2 * D); }
    add(S_DZ_HE, NH); add(S_DZ_RN, NR); add(S_DPREL, NH * x.h0l); add(S_DFE, NH * 2 * D); add(S_DMHE, NH * D);
    add(S_DPRER, NR * x.h0r); add(S_DXR, NR * D);
    add(S_G0, M * D); add(S_G1, M * D); add(S_DPQ, M * 2 * D);
    // small models (no MFMA-tiled weight-gradient shapes): the node-level dY^T X products join the step's one grouped
    // launch at the end, so every layer's dP | dQ has to survive until then
    if (defer_node_tn(D))
        for (int l = 1; l <= x.L; ++l) add(S_DPQL + l, M * 2 * D);
    // split-K slabs: every weight-gradient product keeps its own region until the step's final reduction
    auto slab_floats = [](int I, int J, int64_t rows) {
        return (int64_t)(tn_shape_mfma_ok(I, J) ? tn_splits(I, J, rows) : tn_job_splits(rows)) * I * J;
    };
    for (int l = 2; l <= x.L; ++l) add(S_SLAB_W + l, slab_floats(2 * D, D, M));
    add(S_SLAB_XP1, slab_floats(2 * D, 32, M));
    add(S_SLAB_XP2, slab_floats(D, 32, M));
    add(S_SLAB_FE, std::max(slab_floats(2 * D, x.h0l, NH), (int64_t)head_wgrad_groups((int)B) * 2 * D * x.h0l));
    add(S_SLAB_XR, slab_floats(D, x.h0r, NR));
    // per-sample weight gradients (grouped dY^T X): <= 16 row splits of every per-sample weight (+ the collapsed ones)
    pl->small_slab_floats = 16LL * (P.n_floats + 4LL * D * D + 2LL * D * x.h0l + 4096);
    add(S_SLAB_SMALL, pl->small_slab_floats);
    // edge MLPs with K > 1 sub-layers: one row per edge direction (incidence) behind the first Linear
    if (x.K > 1) {
        const int64_t NI = std::max<int64_t>(mb.n_inc, 1);
        add(S_EIDX, 3 * NI + NH);
        add(S_EDA, NI * D); add(S_EDB, NI * D);
        for (int l = 1; l <= x.L; ++l) {
            for (int k = 1; k <= x.K; ++k) add(S_EA + LK(l, k), NI * D);
            for (int k = 1; k < x.K; ++k) {
                add(S_EWT + LK(l, k), (int64_t)D * D);
                add(S_EBP + LK(l, k), B * D);
                add(S_ESLAB + LK(l, k), slab_floats(D, D, NI));
            }
        }
    }
    // per-row sums of the pointer heads' backward (pointer_bwd2): [B][h0] each, reduced over the rows afterwards;
    // 0: land dz*hid, 1: (free), 2: road dz*hid, 3: road dpre  (the land dpre sums are S_DCONST)
    for (int k = 0; k < 4; ++k) add(S_CSP0 + k, B * std::max(std::max(x.h0l, x.h0r), 16));
    if (tiny_supported(d, mb.max_n, mb.max_inc, mb.max_cand)) {
        const int G = tiny_groups((int)B);
        add(S_TINY_SLAB, (int64_t)G * tiny_slab_stride(P));
        add(S_TINY_SCR, (int64_t)G * tiny_scratch_stride(d, mb.max_cand));
        add(S_TINY_LOSS, B * 4);
    }
    // second dP|dQ buffer (layer l's weight gradient on the side stream may still read its own).  LAST slot of the plan on
    // purpose: whether it exists depends on a process-wide knob, and a knob change between a forward and its backward must not
    // move any other slot (it only changes the size the backward asks for, which check_args verifies)
    if (side_wgrad_on(M)) add(S_DPQ2, M * 2 * D);
    pl->total = off;
}

PackedView make_view(const void *packed_dev, const upamd_pack_layout &L) {
    const char *b = static_cast<const char *>(packed_dev);
    PackedView v;
    v.meta = reinterpret_cast<const int32_t *>(b + L.off_meta);
    v.X = reinterpret_cast<const float *>(b + L.off_x);
    v.nmask = reinterpret_cast<const uint8_t *>(b + L.off_nmask);
    v.rowptr = reinterpret_cast<const int32_t *>(b + L.off_rowptr);
    v.inc_nbr = reinterpret_cast<const uint16_t *>(b + L.off_inc_nbr);
    v.he_src = reinterpret_cast<const uint16_t *>(b + L.off_he_src);
    v.he_dst = reinterpret_cast<const uint16_t *>(b + L.off_he_dst);
    v.he_live = reinterpret_cast<const uint8_t *>(b + L.off_he_live);
    v.rn_node = reinterpret_cast<const uint16_t *>(b + L.off_rn_node);
    v.order = reinterpret_cast<const uint16_t *>(b + L.off_order);
    v.hinc_ptr = reinterpret_cast<const int32_t *>(b + L.off_hinc_ptr);
    v.hinc_nbr = reinterpret_cast<const uint16_t *>(b + L.off_hinc_nbr);
    v.hinc_he = reinterpret_cast<const uint16_t *>(b + L.off_hinc_he);
    v.he_sel = L.off_he_sel >= 0 ? reinterpret_cast<const uint16_t *>(b + L.off_he_sel) : nullptr;      // (absent in a replay planned
    v.xbar = L.off_xbar >= 0 ? reinterpret_cast<const float *>(b + L.off_xbar) : nullptr;                //  without the rl-mlp fields)
    v.numerical = reinterpret_cast<const float *>(b + L.off_numerical);
    v.cur = reinterpret_cast<const float *>(b + L.off_cur);
    v.Fn = L.numerical_dim;
    return v;
}

MbView make_mb(const upamd_minibatch &mb) {
    MbView v;
    v.B = mb.B; v.M = mb.n_nodes; v.Nhe = mb.n_he; v.Nrn = mb.n_rn; v.max_n = mb.max_n; v.max_inc = mb.max_inc; v.max_cand = mb.max_cand;
    v.idx = mb.idx_dev; v.node_off = mb.node_off_dev; v.he_off = mb.he_off_dev; v.rn_off = mb.rn_off_dev;
    v.NI = mb.n_inc; v.inc_off = mb.inc_off_dev;
    v.rows = nullptr;
    return v;
}

int check_args(upamd_engine *eng, const void *packed, const upamd_pack_layout *layout, const upamd_minibatch *mb,
               const float *params, void *ws, int64_t ws_bytes, Plan *pl) {
    if (!eng || !packed || !layout || !mb || !params || !ws) return fail(UPAMD_E_INVALID, "null argument");
    if (mb->B <= 0 || mb->n_nodes <= 0) return fail(UPAMD_E_INVALID, "empty minibatch (B=%d, nodes=%lld)", mb->B, (long long)mb->n_nodes);
    if (!mb->idx_dev || !mb->node_off_dev || !mb->he_off_dev || !mb->rn_off_dev) return fail(UPAMD_E_INVALID, "minibatch schedule pointers are null");
    if (eng->d.encoder == UPAMD_ENCODER_MLP && (layout->off_xbar < 0 || layout->off_he_sel < 0))
        return fail(UPAMD_E_INVALID, "this replay was planned without the rl-mlp fields (upamd_pack_plan_ex flag 2): an rl-mlp engine cannot read it");
    if (layout->node_dim != eng->d.node_dim || layout->numerical_dim != eng->d.numerical_dim)
        return fail(UPAMD_E_INVALID, "packed replay feature sizes (%d,%d) do not match the model (%d,%d)", layout->node_dim,
                    layout->numerical_dim, eng->d.node_dim, eng->d.numerical_dim);
    if (eng->d.encoder == UPAMD_ENCODER_SGNN && edge_fc_layers(eng->d) > 1 && (!mb->inc_off_dev || mb->n_inc < 0))
        return fail(UPAMD_E_INVALID, "num_edge_fc_layers > 1 needs the minibatch's incidence offsets (inc_off_dev, n_inc)");
    if (reinterpret_cast<uintptr_t>(ws) % 256 != 0) return fail(UPAMD_E_INVALID, "workspace must be 256-byte aligned");
    make_plan(eng->d, eng->P, *mb, pl);
    if (pl->total * 4 > ws_bytes) return fail(UPAMD_E_WORKSPACE, "workspace too small: need %lld bytes, got %lld", (long long)pl->total * 4, (long long)ws_bytes);
    return 0;
}

// The per-sample chain launches are latency-bound (50-65 us at any batch size, 32 workgroups at 256 rows) and most of them are
// off the critical path: the forward's chain is needed only from the LAST GCN layer on, the backward's chain + grouped
// per-sample weight gradients only by the final reduction.  They run on an engine-owned side stream, forked from / joined to the
// caller's stream with events, underneath the GCN layers' GEMM / message-passing launches (tune knob "side_stream", default on).
static int g_side_stream = 1;
static int g_side_priority = 1;
static int side_ready(upamd_engine *eng, hipStream_t st, SideCtx **out) {
    SideCtx &c = eng->sides[st];
    if (!c.side) {
        // The runtime multiplexes the streams of one priority level onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default) in
        // creation order: with RCCL initialised (its own streams come first) the side stream landed on the SAME hardware queue as
        // the caller's stream and the forked step ran serialised (measured: no overlap, -3 %).  A stream of another priority
        // level lives in that level's own queues; high priority also suits what runs here -- short kernels that gate the caller's
        // stream (tune knob "side_priority": 1 = high (default), 0 = normal, 2 = low).
        int least = 0, greatest = 0;
        UPAMD_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        if (g_side_priority == 0 || least == greatest)
            UPAMD_HIP(hipStreamCreateWithFlags(&c.side, hipStreamNonBlocking));
        else
            UPAMD_HIP(hipStreamCreateWithPriority(&c.side, hipStreamNonBlocking, g_side_priority == 1 ? greatest : least));
        if (g_side_priority == 0 || least == greatest)
            UPAMD_HIP(hipStreamCreateWithFlags(&c.side2, hipStreamNonBlocking));
        else
            UPAMD_HIP(hipStreamCreateWithPriority(&c.side2, hipStreamNonBlocking, g_side_priority == 1 ? greatest : least));
        UPAMD_HIP(hipEventCreateWithFlags(&c.ev_join2, hipEventDisableTiming));
        UPAMD_HIP(hipEventCreateWithFlags(&c.ev_fork, hipEventDisableTiming));
        UPAMD_HIP(hipEventCreateWithFlags(&c.ev_join, hipEventDisableTiming));
        UPAMD_HIP(hipEventCreateWithFlags(&c.ev_a, hipEventDisableTiming));
        UPAMD_HIP(hipEventCreateWithFlags(&c.ev_b, hipEventDisableTiming));
        for (hipEvent_t &e : c.pool) UPAMD_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        for (hipEvent_t &e : c.b_ev) UPAMD_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    *out = &c;
    return 0;
}
// side stream continues from this point of `st`
static int fork_side(SideCtx *c, hipStream_t st) {
    UPAMD_HIP(hipEventRecord(c->ev_fork, st));
    UPAMD_HIP(hipStreamWaitEvent(c->side, c->ev_fork, 0));
    return 0;
}
// `st` continues after everything enqueued on the side stream so far
static int join_side(SideCtx *c, hipStream_t st) {
    UPAMD_HIP(hipEventRecord(c->ev_join, c->side));
    UPAMD_HIP(hipStreamWaitEvent(st, c->ev_join, 0));
    return 0;
}

// `to` continues after everything enqueued on `from` so far (ev is re-recorded: the host enqueues in order)
static int stream_after(hipStream_t to, hipStream_t from, hipEvent_t ev) {
    UPAMD_HIP(hipEventRecord(ev, from));
    UPAMD_HIP(hipStreamWaitEvent(to, ev, 0));
    return 0;
}
// tune knob "side_heads" (default on, needs side_stream): the land-use pointer-head chain -- forward: its first Linear next to the
// attention; backward: softmax / second-Linear backward, feature backward and weight gradient next to the value-head chain and
// the attention backward -- runs on the side stream.  All of these are HBM-bound kernels of 0.1-0.3 ms that used to queue one
// behind the other; the two chains only meet at the last GCN layer's backward (dS from the attention side, dM from the head side).
static int g_side_heads = 1;
static hipEvent_t next_event(SideCtx *c) { return c->pool[c->pool_next++ & 7]; }

// An error return between fork and join must not leave side-stream work running on a workspace the caller may free next:
// the guard drains the side stream unless the join was reached
struct SideGuard {
    SideCtx *c = nullptr;
    bool armed = false;
    ~SideGuard() {
        if (armed && c && c->side) (void)hipStreamSynchronize(c->side);
        if (armed && c && c->side2) (void)hipStreamSynchronize(c->side2);
    }
};

// tune knob "grad_buckets" (default on): the backward finalises the gradient buffer range by range -- heads + attention + value
// head right behind the per-sample products, every GCN layer's weight behind its weight-gradient GEMM, the rest at the end -- and
// records an event per range, so that a data-parallel caller can all-reduce a range while the layers below are still in their
// backward (upamd_grad_buckets / upamd_grad_bucket_wait).  The arithmetic and the order of every sum are those of the
// single-range form (0): the same bits, only the launch that performs a reduction moves.
static int g_grad_buckets = 1;
static int bucket_done(SideCtx *c, int64_t begin, int64_t end, hipStream_t on) {
    if (c->n_buckets >= SideCtx::MAX_BUCKETS) return fail(UPAMD_E_LIMIT, "too many gradient buckets");
    const int k = c->n_buckets++;
    c->b_begin[k] = begin;
    c->b_end[k] = end;
    UPAMD_HIP(hipEventRecord(c->b_ev[k], on));
    return 0;
}

// UPAMD_DEBUG_SYNC=1: synchronise after every launch and name the one that faulted (debugging aid; off by default)
static const bool g_debug_sync = getenv("UPAMD_DEBUG_SYNC") && atoi(getenv("UPAMD_DEBUG_SYNC")) != 0;
static int debug_sync(const char *what) {
    if (!g_debug_sync) return 0;
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) return fail(UPAMD_E_HIP, "device fault after: %s (%s)", what, hipGetErrorString(e));
    fprintf(stderr, "[upamd] ok: %.90s\n", what);
    return 0;
}

#define CK(expr)                                   \
    do {                                           \
        int _rc = (expr);                          \
        if (_rc) return _rc;                       \
        if (g_debug_sync) {                        \
            _rc = debug_sync(#expr);               \
            if (_rc) return _rc;                   \
        }                                          \
    } while (0)

// tune knob "pq_exp" (default on): the P/Q GEMMs of layers 2 .. L store their output in exp form (flagged block by block)
// and the message-passing kernels stage it by LDS-DMA (edge.hip: dma_pq_slice)
static int g_pq_exp = 1;
// the P/Q product of layer l as launched by the forward; the backward re-derives the same decision from it
static GemmNT pq_gemm(const float *H, int64_t M, int D, const float *Wcat, float *PQ) {
    GemmNT g;
    g.A = H; g.M = M; g.K = D; g.lda = 0; g.a_rm = false; g.W = Wcat; g.N = 2 * D; g.ldw = D; g.bias = nullptr; g.R = nullptr;
    g.C = PQ; g.ldc = 0; g.c_rm = false; g.act_tanh = 0; g.alpha = 1.f;
    return g;
}
static bool pq_exp_layer(const GemmNT &g, int l, int K) { return g_pq_exp && K == 1 && l >= 2 && gemm_nt_exp_store_ok(g); }

// slabs[s][I][J] = partial sums of A[rows, I](pm)^T * Bm[rows, J](pm): the tiled split-K MFMA kernel where the shape allows
// (I % 128 == 0), otherwise one grouped-kernel launch with panel-major operands (narrow models: D = 16 ... 64)
// tune knob "fold_layer1" (default on): first GCN layer computed inside the message-passing kernels
static int g_fold_layer1 = 1;
// (1 = fold whenever the slices fit the LDS at all; 2 = only where every graph fits HALF of it.  Measured, round 3: with the
// K = 32 GEMMs instead of the fold's one-workgroup-per-CU large size class DHM minibatches gain 0.4 %, mixed ones lose 1 %
// (profiles/archive/r03_lab_fold_rule.log) -- the default stays 1)
static bool fold_layer1(const MbView &mb, int L, int K) {
    return g_fold_layer1 && K == 1 && L >= 2 && (g_fold_layer1 == 2 ? edge_fold_pays(mb) : edge_fold_ok(mb));
}

int node_tn(const float *A, int I, const float *Bm, int J, int64_t rows, float *slabs, int *S_out, hipStream_t st, Profiler *prof) {
    if (tn_shape_mfma_ok(I, J)) return launch_gemm_tn(A, I, Bm, J, rows, slabs, S_out, st, prof);
    TnJobs tj;
    int rc = tn_add(&tj, A, 0, I, Bm, 0, J, rows, slabs, S_out, 1, 1);
    if (rc) return rc;
    const int began = prof_begin(prof, "gemm_tn_small", st, 2.0 * (double)rows * I * J, 4.0 * (double)rows * (I + J));
    rc = launch_gtn(tj, st);
    prof_end(prof, "gemm_tn_small", st, began);
    return rc;
}

// Slab-reduction jobs of a step: flushed as one launch (or several, when the job table would overflow)
struct Reducer {
    RedJobs jobs;
    int blocks = 0;
    hipStream_t st;
    // forked step: some sources of a reducer that flushes on `st` are produced on the side streams and are ordered only by the
    // step's final join.  A flush forced EARLY by a full job table (deep models: 2 + L (1 + 2 (K - 1)) jobs can pass
    // RED_MAX_JOBS before the join) first makes `st` wait for everything the side streams have been given so far.
    SideCtx *side = nullptr;
    int add(const float *slab, int S, int64_t sstride, int I, int J, int mode, int jkeep, float *dst, int ldd, float *dst2 = nullptr,
            int overwrite = 0) {
        if (jobs.n >= RED_MAX_JOBS) {
            if (side && side->side != st) CK(stream_after(st, side->side, next_event(side)));
            if (side && side->side2 && side->side2 != st) CK(stream_after(st, side->side2, next_event(side)));
            CK(flush());
        }
        return red_add(&jobs, &blocks, slab, S, sstride, I, J, mode, jkeep, dst, ldd, dst2, overwrite);
    }
    int flush() {
        int rc = launch_greduce(jobs, blocks, st);
        jobs.n = 0;
        blocks = 0;
        return rc;
    }
};

// name -> (slot, rows, cols, kind) of the workspace tensors the tests / action heads read
int slot_of_name(const upamd_model_desc &d, const Dims &x, const upamd_minibatch &mb, const std::string &n, int64_t *rows,
                 int64_t *cols, int *kind) {
    const int64_t B = mb.B, M = mb.n_nodes, NH = mb.n_he, NR = mb.n_rn;
    auto num = [&](size_t pos) { return atoi(n.c_str() + pos); };
    *kind = 0;
    // with the first layer folded into the message-passing kernels H0 and PQ1 are never materialised
    const bool folded = !x.mlp && fold_layer1(make_mb(mb), x.L, x.K);
    if (n[0] == 'H' && n.size() > 1 && isdigit(n[1])) {
        *rows = M; *cols = x.D; *kind = 1;
        return (num(1) <= x.L && !(folded && num(1) == 0)) ? S_H + num(1) : -1;
    }
    if (n.rfind("PQ", 0) == 0) {
        *rows = M; *cols = 2 * x.D; *kind = 1;
        return (num(2) >= 1 && num(2) <= x.L && !(folded && num(2) == 1)) ? S_PQ + num(2) : -1;
    }
    if (n == "dPQ") { *rows = M; *cols = 2 * x.D; *kind = 1; return S_DPQ; }
    if (n.rfind("EA", 0) == 0 && n.size() >= 5 && n.find('_') != std::string::npos) {      // "EA<l>_<k>": sub-layer activation A_k of layer l
        const int l = num(2), k = atoi(n.c_str() + n.find('_') + 1);
        *rows = mb.n_inc; *cols = x.D; *kind = 1;
        return (x.K > 1 && l >= 1 && l <= x.L && k >= 1 && k <= x.K) ? S_EA + LK(l, k) : -1;
    }
    if (n == "G0" || n == "G1") { *rows = M; *cols = x.D; *kind = 1; return n == "G0" ? S_G0 : S_G1; }
    if (n == "Xp") { *rows = M; *cols = 32; *kind = 1; return S_XP; }
    if (n == "FE" || n == "dFE") { *rows = NH; *cols = 2 * x.D; *kind = 1; return n == "FE" ? S_FE : S_DFE; }
    if (n == "hidl" || n == "dprel") { *rows = NH; *cols = x.h0l; *kind = 1; return n == "hidl" ? S_HIDL : S_DPREL; }
    if (n == "dMhe") { *rows = NH; *cols = x.D; *kind = 1; return S_DMHE; }
    if (n == "XR" || n == "dXR") { *rows = NR; *cols = x.D; *kind = 1; return n == "XR" ? S_XR : S_DXR; }
    if (n == "hidr" || n == "dprer") { *rows = NR; *cols = x.h0r; *kind = 1; return n == "hidr" ? S_HIDR : S_DPRER; }
    if (n == "z_he" || n == "p_he" || n == "dz_he") { *rows = NH; *cols = 1; return n == "z_he" ? S_Z_HE : (n == "p_he" ? S_P_HE : S_DZ_HE); }
    if (n == "z_rn" || n == "p_rn" || n == "dz_rn") { *rows = NR; *cols = 1; return n == "z_rn" ? S_Z_RN : (n == "p_rn" ? S_P_RN : S_DZ_RN); }
    if (n == "alpha") { *rows = x.heads; *cols = M; return S_ALPHA; }
    if (n == "SV" || n == "dSV") { *rows = B; *cols = x.Wp; return n == "SV" ? S_SV : S_DSV; }      // columns >= W are zero padding
    if (n == "r" || n == "s" || n == "ds" || n == "dr") {
        *rows = B; *cols = (int64_t)x.heads * x.D;
        return n == "r" ? S_R : (n == "s" ? S_S : (n == "ds" ? S_DS : S_DR));
    }
    if (n == "lse" || n == "entk") { *rows = B; *cols = 1; return n == "lse" ? S_LSE : S_ENTK; }
    if (n == "curg") { *rows = B; *cols = UPAMD_NODE_PAD; return S_CURG; }
    if (n == "Wkk" || n == "Wvv" || n == "dWkk" || n == "dWvv") {
        *rows = x.D; *cols = x.D;
        return n == "Wkk" ? S_WKK : (n == "Wvv" ? S_WVV : (n == "dWkk" ? S_DWKK : S_DWVV));
    }
    if (n[0] == 'U' && n.size() > 1 && isdigit(n[1])) {
        const int i = num(1);
        if (i < 0 || i > d.n_num) return -1;
        *rows = B; *cols = i == 0 ? x.Fn : d.num_hidden[i - 1];
        return S_U + i;
    }
    if (n[0] == 'V' && n.size() > 1 && isdigit(n[1])) {
        const int i = num(1);
        if (i < 1 || i >= d.n_value) return -1;
        *rows = B; *cols = d.value_hidden[i - 1];
        return S_V + i;
    }
    *rows = B; *cols = x.D;
    if (n == "C") return S_C;
    if (n == "hbarV") return S_HBARV;
    if (n == "hbarE") return S_HBARE;
    if (n == "q0") return S_Q0;
    if (n == "q1") return S_Q1;
    if (n == "o") return S_O;
    if (n == "att") return S_ATT;
    if (n == "do") return S_DO;
    if (n == "datt") return S_DATT;
    if (n == "dq1") return S_DQ1;
    if (n == "dq0") return S_DQ0;
    if (n == "dC") return S_DC;
    if (n == "dC_head") return S_DC_HEAD;
    return -1;
}

}  // namespace

void upamd::set_fold_layer1(int on) { g_fold_layer1 = on == 2 ? 2 : (on ? 1 : 0); }
void upamd::set_pq_exp(int on) { g_pq_exp = on ? 1 : 0; }
void upamd::set_side_stream(int on) { g_side_stream = on ? 1 : 0; }
void upamd::set_side_priority(int v) { g_side_priority = (v >= 0 && v <= 2) ? v : 1; }
void upamd::set_side_heads(int on) { g_side_heads = on ? 1 : 0; }
void upamd::set_side_wgrad(int on) { g_side_wgrad = (on >= 0 && on <= 3) ? on : 1; }
void upamd::set_grad_buckets(int on) { g_grad_buckets = on ? 1 : 0; }

extern "C" int upamd_engine_create(const upamd_model_desc *desc, upamd_engine **out) {
    if (!out) return fail(UPAMD_E_INVALID, "upamd_engine_create: out is null");
    ParamLayout P;
    int rc = build_param_layout(desc, &P);
    if (rc) return rc;
    if (desc->n_land != 2 || desc->n_road != 2)
        return fail(UPAMD_E_INVALID, "this build supports pointer heads of the form [hidden, 1] only (got %d and %d layers)", desc->n_land, desc->n_road);
    if (desc->L > MAXL) return fail(UPAMD_E_INVALID, "num_gcn_layers > %d", MAXL);
    upamd_engine *e = new upamd_engine();
    e->d = *desc;
    e->P = P;
    *out = e;
    return UPAMD_OK;
}

extern "C" int upamd_profile_reset(upamd_engine *eng) {
    if (!eng) return fail(UPAMD_E_INVALID, "null engine");
    for (auto &kv : eng->prof.stats)
        for (hipEvent_t e : kv.second.ev) (void)hipEventDestroy(e);
    eng->prof.stats.clear();
    return UPAMD_OK;
}

extern "C" void upamd_engine_destroy(upamd_engine *eng) {
    if (!eng) return;
    upamd_profile_reset(eng);
    for (auto &kv : eng->sides) {
        SideCtx &c = kv.second;
        if (!c.side) continue;
        (void)hipStreamSynchronize(c.side);
        if (c.side2) {
            (void)hipStreamSynchronize(c.side2);
            (void)hipStreamDestroy(c.side2);
            (void)hipEventDestroy(c.ev_join2);
        }
        (void)hipEventDestroy(c.ev_fork);
        (void)hipEventDestroy(c.ev_join);
        (void)hipEventDestroy(c.ev_a);
        (void)hipEventDestroy(c.ev_b);
        for (hipEvent_t e : c.pool) (void)hipEventDestroy(e);
        (void)hipStreamDestroy(c.side);
    }
    delete eng;
}

extern "C" int upamd_profile_enable(upamd_engine *eng, int32_t on) {
    if (!eng) return fail(UPAMD_E_INVALID, "null engine");
    eng->prof.on = on != 0;
    return UPAMD_OK;
}

extern "C" int upamd_profile_read(upamd_engine *eng, const char *name, int64_t *launches, double *total_ms,
                                  double *total_flops, double *total_bytes) {
    if (!eng || !name) return fail(UPAMD_E_INVALID, "null argument");
    if (launches) *launches = 0;
    if (total_ms) *total_ms = 0;
    if (total_flops) *total_flops = 0;
    if (total_bytes) *total_bytes = 0;
    auto itk = eng->prof.stats.find(name);
    if (itk == eng->prof.stats.end()) return UPAMD_OK;
    KernelStat *k = &itk->second;
    double ms = 0;
    for (size_t i = 0; i + 1 < k->ev.size(); i += 2) {
        UPAMD_HIP(hipEventSynchronize(k->ev[i + 1]));
        float t = 0;
        UPAMD_HIP(hipEventElapsedTime(&t, k->ev[i], k->ev[i + 1]));
        ms += t;
    }
    if (launches) *launches = k->launches;
    if (total_ms) *total_ms = ms;
    if (total_flops) *total_flops = k->flops;
    if (total_bytes) *total_bytes = k->bytes;
    return UPAMD_OK;
}

extern "C" int upamd_workspace_bytes(upamd_engine *eng, const upamd_minibatch *mb, int32_t training, int64_t *bytes) {
    (void)training;
    if (!eng || !mb || !bytes) return fail(UPAMD_E_INVALID, "null argument");
    Plan pl;
    make_plan(eng->d, eng->P, *mb, &pl);
    *bytes = pl.total * 4 + 256;
    return UPAMD_OK;
}

extern "C" int upamd_ws_tensor(upamd_engine *eng, const upamd_minibatch *mb, const char *name, int64_t *byte_offset,
                               int64_t *rows, int64_t *cols, int32_t *kind) {
    if (!eng || !mb || !name) return fail(UPAMD_E_INVALID, "null argument");
    Plan pl;
    make_plan(eng->d, eng->P, *mb, &pl);
    const Dims x = dims_of(eng->d);
    int64_t r = 0, c = 0;
    int k = 0;
    const int slot = slot_of_name(eng->d, x, *mb, name, &r, &c, &k);
    if (slot < 0 || pl.off[slot] < 0) return fail(UPAMD_E_INVALID, "unknown workspace tensor '%s'", name);
    // the fused small-model path keeps every intermediate in LDS: only the pointer-head logits reach the workspace, and a name
    // that resolves to a slot the general path would have written must not hand back stale memory
    if (tiny_supported(eng->d, mb->max_n, mb->max_inc, mb->max_cand) && slot != S_Z_HE && slot != S_Z_RN)
        return fail(UPAMD_E_INVALID, "workspace tensor '%s' is not materialised by the fused small-model path (this minibatch takes "
                                     "it: only z_he / z_rn are); tune knob tiny_fused = 0 selects the general kernels", name);
    if (byte_offset) *byte_offset = pl.off[slot] * 4;
    if (rows) *rows = r;
    if (cols) *cols = c;
    if (kind) *kind = k;
    return UPAMD_OK;
}

// =============================================================================================
// forward
// =============================================================================================
extern "C" int upamd_forward(upamd_engine *eng, const void *packed_dev, const upamd_pack_layout *layout,
                             const upamd_minibatch *mbp, const float *prm, void *ws_dev, int64_t ws_bytes,
                             float *value_dev, float *logp_dev, float *ent_dev, int32_t keep, void *stream) {
    (void)keep;
    Plan pl;
    CK(check_args(eng, packed_dev, layout, mbp, prm, ws_dev, ws_bytes, &pl));
    if (!value_dev || !logp_dev || !ent_dev) return fail(UPAMD_E_INVALID, "output pointers are null");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const upamd_model_desc &d = eng->d;
    const ParamLayout &P = eng->P;
    const Dims x = dims_of(d);
    const int D = x.D, B = mbp->B;
    const PackedView pk = make_view(packed_dev, *layout);
    MbView mb = make_mb(*mbp);
    float *ws = static_cast<float *>(ws_dev);
    auto W = [&](int slot) { return ws + pl.off[slot]; };
    auto PR = [&](int idx) { return prm + P.off(idx); };
    Profiler *prof = &eng->prof;
    // fused small-model path (tiny.hip): the whole forward of a graph in one workgroup, one launch for the minibatch
    const bool fused_fwd = tiny_supported(d, mbp->max_n, mbp->max_inc, mbp->max_cand);
    if (fused_fwd) {
        eng->general_fwd.erase(ws_dev);
    } else {
        if (eng->general_fwd.size() > 65536) { eng->general_fwd.clear(); eng->general_fwd_complete = false; }
        eng->general_fwd[ws_dev] = 1;
    }
    if (fused_fwd) {
        TinyIO io;
        memset(&io, 0, sizeof(io));
        io.mode = 0;
        io.value = value_dev; io.logp = logp_dev; io.ent = ent_dev;
        io.z_he = W(S_Z_HE); io.z_rn = W(S_Z_RN);           // candidate logits for the action heads (upamd_ws_tensor)
        const int began = prof_begin(prof, "tiny_fwd", st, 0.0, 0.0);
        const int rc = launch_tiny(d, P, pk, mb, prm, io, st);
        prof_end(prof, "tiny_fwd", st, began);
        return rc;
    }
    const float *Win = x.mlp ? prm : PR(P.inproj_w), *bin = x.mlp ? prm : PR(P.inproj_b);      // (unused by the rl-mlp encoder)
    const float *Wiq = Win, *Wik = Win + (int64_t)D * D, *Wiv = Win + 2LL * D * D;
    const float *biq = bin, *biv = bin + 2 * D;
    const bool land = mb.Nhe > 0, road = mb.Nrn > 0;
    // head.hip: the land-use head's first Linear works on the m half of FE alone (the m*c half is then never written)
    const bool fe_half = head_fe_half_ok(D, x.h0l);
    const int fe_full = fe_half ? 0 : 1;
    bool head_on_side = false;            // the land-use head's first Linear was launched on the side stream (joined below)
    // (launch-bound small models gain nothing from the fork: its two event round trips cost more than they hide)
    const bool forked = !x.mlp && g_side_stream != 0 && !defer_node_tn(x.D);
    SideGuard side_guard;
    SideCtx *sc = nullptr;
    if (forked) {
        CK(side_ready(eng, st, &sc));
        side_guard.c = sc;
    }

    if (x.mlp) {
        // ===== rl-mlp encoder (MLPStateEncoder, state_encoder.py:217-308): node encoder only, pooled means, no attention
        {
            PermJobs pj;
            int blocks = 0;
            CK(perm_add(&pj, &blocks, PERM_PAD_COLS, PR(P.node_w), nullptr, W(S_WE_PAD), nullptr, nullptr, nullptr, D, x.F, 32));
            CK(perm_add(&pj, &blocks, PERM_LAND_HEAD, PR(P.land_w[0]), nullptr, W(S_W1F), W(S_WBD), W(S_W1FT), W(S_WBDT), x.h0l, D, 0));
            CK(perm_add(&pj, &blocks, PERM_TRANSPOSE, PR(P.road_w[0]), nullptr, W(S_R1T), nullptr, nullptr, nullptr, x.h0r, D, 0));
            int prev = x.Fn;
            for (int i = 0; i < d.n_num; ++i) {
                CK(perm_add(&pj, &blocks, PERM_TRANSPOSE, PR(P.num_w[i]), nullptr, W(S_WNT + i), nullptr, nullptr, nullptr, d.num_hidden[i], prev, 0));
                prev = d.num_hidden[i];
            }
            prev = x.W;
            for (int i = 0; i < d.n_value; ++i) {
                CK(perm_add(&pj, &blocks, PERM_TRANSPOSE, PR(P.value_w[i]), nullptr, W(S_WVT + i), nullptr, nullptr, nullptr, d.value_hidden[i], prev, 0));
                prev = d.value_hidden[i];
            }
            CK(perm_add(&pj, &blocks, PERM_TRANSPOSE, PR(P.node_w), nullptr, W(S_WET), nullptr, nullptr, nullptr, D, x.F, 0));
            CK(launch_permute(pj, blocks, st));
        }
        const ChainDims cd = chain_dims(d, x, B);
        {
            ChainFwdPre a;
            memset(&a, 0, sizeof(a));
            a.d = cd; a.pk = pk; a.mb = mb;
            a.rows = reinterpret_cast<int32_t *>(W(S_ROWS)); a.Xp = W(S_XP);
            for (int i = 0; i < d.n_num; ++i) { a.WnT[i] = W(S_WNT + i); a.bn[i] = PR(P.num_b[i]); }
            a.WeT = W(S_WET); a.be = PR(P.node_b); a.WbdT = W(S_WBDT); a.b1l = PR(P.land_b0);
            for (int i = 0; i <= d.n_num; ++i) a.U[i] = W(S_U + i);
            a.curg = W(S_CURG); a.C = W(S_C); a.hbarE = W(S_HBARE);
            a.constb = land ? W(S_CONSTB) : nullptr;
            CK(launch_chain_fwd_pre(a, st));
        }
        mb.rows = reinterpret_cast<const int32_t *>(W(S_ROWS));
        CK(launch_gemm_nt(W(S_XP), mb.M, 32, W(S_WE_PAD), D, PR(P.node_b), nullptr, W(S_H + 0), 0, st, prof));
        CK(launch_mlp_pool_fwd(pk, mb, D, W(S_H + 0), PR(P.node_b), W(S_C), W(S_HBARV), land ? W(S_FE) : nullptr, st, fe_full));
        {
            ChainFwdPost a;
            memset(&a, 0, sizeof(a));
            a.d = cd; a.rows = mb.rows;
            a.hbarV = W(S_HBARV); a.hbarE = W(S_HBARE); a.Ulast = W(S_U + d.n_num);
            for (int i = 0; i < d.n_value; ++i) { a.WvT[i] = W(S_WVT + i); a.bv[i] = PR(P.value_b[i]); }
            a.SV = W(S_SV);
            for (int i = 1; i < d.n_value; ++i) a.V[i] = W(S_V + i);
            a.value = value_dev;
            CK(launch_chain_fwd_post(a, st));
        }
    } else {
    // ---- 1. every gather-style parameter preparation in ONE launch
    {
        PermJobs pj;
        int blocks = 0;
        CK(perm_add(&pj, &blocks, PERM_PAD_COLS, PR(P.node_w), nullptr, W(S_WE_PAD), nullptr, nullptr, nullptr, D, x.F, 32));
        for (int l = 0; l < x.L; ++l)
            CK(perm_add(&pj, &blocks, PERM_WCAT, PR(P.edge_w[l]), nullptr, W(S_WCAT + l), W(S_WCATT + l), nullptr, nullptr, D, D, 0));
        CK(perm_add(&pj, &blocks, PERM_LAND_HEAD, PR(P.land_w[0]), nullptr, W(S_W1F), W(S_WBD), W(S_W1FT), W(S_WBDT), x.h0l, D, 0));
        CK(perm_add(&pj, &blocks, PERM_TRANSPOSE, PR(P.road_w[0]), nullptr, W(S_R1T), nullptr, nullptr, nullptr, x.h0r, D, 0));
        int prev = x.Fn;
        for (int i = 0; i < d.n_num; ++i) {
            CK(perm_add(&pj, &blocks, PERM_TRANSPOSE, PR(P.num_w[i]), nullptr, W(S_WNT + i), nullptr, nullptr, nullptr, d.num_hidden[i], prev, 0));
            prev = d.num_hidden[i];
        }
        prev = x.W;
        for (int i = 0; i < d.n_value; ++i) {
            CK(perm_add(&pj, &blocks, PERM_TRANSPOSE, PR(P.value_w[i]), nullptr, W(S_WVT + i), nullptr, nullptr, nullptr, d.value_hidden[i], prev, 0));
            prev = d.value_hidden[i];
        }
        CK(perm_add(&pj, &blocks, PERM_TRANSPOSE, PR(P.node_w), nullptr, W(S_WET), nullptr, nullptr, nullptr, D, x.F, 0));
        CK(perm_add(&pj, &blocks, PERM_TRANSPOSE, PR(P.q_w), nullptr, W(S_WQT), nullptr, nullptr, nullptr, D, D, 0));
        CK(perm_add(&pj, &blocks, PERM_TRANSPOSE, Wiq, nullptr, W(S_WIQT), nullptr, nullptr, nullptr, D, D, 0));
        CK(perm_add(&pj, &blocks, PERM_TRANSPOSE, PR(P.outproj_w), nullptr, W(S_WOT), nullptr, nullptr, nullptr, D, D, 0));
        CK(launch_permute(pj, blocks, st));
    }
    // ---- 2. the collapsed weight products in ONE launch
    //   Wkk = Wik Wk, Wvv = Wiv Wv, bvv = Wiv bv + biv (single-query attention, state_encoder.py:150-161);
    //   W1c = Wcat_1 We, b1c = Wcat_1 be (first GCN layer straight from the raw node features)
    {
        SmmJobs sj;
        int blocks = 0;
        CK(smm_add(&sj, &blocks, D, D, D, Wik, D, 1, PR(P.k_w), D, 1, nullptr, W(S_WKK), D, 0, 1.f, W(S_WKKT), D));
        CK(smm_add(&sj, &blocks, D, D, D, Wiv, D, 1, PR(P.v_w), D, 1, nullptr, W(S_WVV), D, 0, 1.f, W(S_WVVT), D));
        CK(smm_add(&sj, &blocks, 1, D, D, PR(P.v_b), 0, 1, Wiv, 1, D, biv, W(S_BVV), D, 0, 1.f));
        CK(smm_add(&sj, &blocks, 2 * D, 32, D, W(S_WCAT + 0), D, 1, W(S_WE_PAD), 32, 1, nullptr, W(S_W1C), 32, 0, 1.f));
        CK(smm_add(&sj, &blocks, 1, 2 * D, D, PR(P.node_b), 0, 1, W(S_WCAT + 0), 1, D, nullptr, W(S_B1C), 2 * D, 0, 1.f));
        CK(launch_gsmm(sj, blocks, st));
    }
    // ---- 3. per-sample chain before the graph part (+ row descriptors, + node-feature gather)
    const ChainDims cd = chain_dims(d, x, B);
    {
        ChainFwdPre a;
        memset(&a, 0, sizeof(a));
        a.d = cd; a.pk = pk; a.mb = mb;
        a.rows = reinterpret_cast<int32_t *>(W(S_ROWS)); a.Xp = W(S_XP);
        for (int i = 0; i < d.n_num; ++i) { a.WnT[i] = W(S_WNT + i); a.bn[i] = PR(P.num_b[i]); }
        a.WeT = W(S_WET); a.be = PR(P.node_b); a.WqT = W(S_WQT); a.bq = PR(P.q_b); a.WiqT = W(S_WIQT); a.biq = biq;
        a.Wkk = W(S_WKK); a.WbdT = W(S_WBDT); a.b1l = PR(P.land_b0);
        for (int i = 0; i <= d.n_num; ++i) a.U[i] = W(S_U + i);
        a.curg = W(S_CURG); a.C = W(S_C); a.q0 = W(S_Q0); a.q1 = W(S_Q1); a.r = W(S_R);
        a.constb = land ? W(S_CONSTB) : nullptr;
        if (forked) {
            // row descriptors + node features first (the graph part needs them at once); the per-sample layers go to the
            // side stream and are joined in front of the last GCN layer (the first consumer of C, then r)
            a.part = CHAIN_GATHER;
            CK(launch_chain_fwd_pre(a, st));
            CK(fork_side(sc, st));
            side_guard.armed = true;
            a.part = CHAIN_LAYERS;
            CK(launch_chain_fwd_pre(a, sc->side));
        } else {
            CK(launch_chain_fwd_pre(a, st));
        }
    }
    mb.rows = reinterpret_cast<const int32_t *>(W(S_ROWS));
    // ---- 4. node encoder on all nodes (state_encoder.py:189-190) and the GCN layers (state_encoder.py:194-197).
    // Layer 1 reads its P/Q straight from the raw node features:
    // PQ_1 = H_0 Wcat_1^T = Xp (Wcat_1 We)^T + Wcat_1 be  (K = 32 instead of D: saves one full-size node GEMM)
    // With the fold (default) neither H_0 nor PQ_1 ever exists in HBM: the layer-1 message-passing workgroups build
    // their slices from Xp in LDS (edge.hip: fold_fill); otherwise two K = 32 GEMMs write them out.
    const bool fold = fold_layer1(mb, x.L, x.K);
    const FoldArgs fa{W(S_XP), W(S_W1C), W(S_B1C), W(S_WE_PAD), PR(P.node_b)};
    if (!fold) CK(launch_gemm_nt(W(S_XP), mb.M, 32, W(S_WE_PAD), D, PR(P.node_b), nullptr, W(S_H + 0), 0, st, prof));
    // num_edge_fc_layers > 1 (deep_edge.hip): the sub-layers behind the first Linear act on one row per edge direction;
    // index tables (endpoints, opposite direction, candidate -> incidence) once per forward, W_k^T for the backward
    const int64_t NIp = std::max<int64_t>(mb.NI, 1);
    int32_t *gsrc = x.K > 1 ? reinterpret_cast<int32_t *>(W(S_EIDX)) : nullptr;
    int32_t *gdst = gsrc + NIp, *grev = gsrc + 2 * NIp, *cand_inc = gsrc + 3 * NIp;
    if (x.K > 1) {
        CK(launch_inc_index(pk, mb, gsrc, gdst, grev, cand_inc, st));
        PermJobs pj;
        int blocks = 0;
        for (int l = 1; l <= x.L; ++l)
            for (int k = 1; k < x.K; ++k) {
                if (pj.n >= PERM_MAX_JOBS) {
                    CK(launch_permute(pj, blocks, st));
                    pj.n = 0;
                    blocks = 0;
                }
                CK(perm_add(&pj, &blocks, PERM_TRANSPOSE, PR(P.edge_wk[l - 1][k - 1]), nullptr, W(S_EWT + LK(l, k)), nullptr, nullptr,
                            nullptr, D, D, 0));
            }
        CK(launch_permute(pj, blocks, st));
    }
    for (int l = 1; l <= x.L; ++l) {
        const uint8_t *pqflag = nullptr;        // != nullptr: this layer's P/Q is in exp form, block by block
        if (l == 1) {
            if (!fold) CK(launch_gemm_nt(W(S_XP), mb.M, 32, W(S_W1C), 2 * D, W(S_B1C), nullptr, W(S_PQ + 1), 0, st, prof));
        } else {
            GemmNT g = pq_gemm(W(S_H + l - 1), mb.M, D, W(S_WCAT + l - 1), W(S_PQ + l));
            bool used = false;
            if (pq_exp_layer(g, l, x.K)) {
                g.exp_flags = reinterpret_cast<uint8_t *>(W(S_PQF + l));
                g.exp_used = &used;
            }
            CK(launch_gemm_nt_ex(g, st, prof));
            if (g.exp_flags && !used) return fail(UPAMD_E_INVALID, "P/Q GEMM of layer %d did not take the exp-form store it was planned with", l);
            if (used) pqflag = g.exp_flags;
        }
        if (forked && l == x.L) {                            // C (head inputs of the last layer), r, U, constb are ready
            CK(join_side(sc, st));
            side_guard.armed = false;
        }
        if (x.K > 1) {
            // A_1 = tanh(P_src + Q_dst + b_0) per edge direction, A_k+1 = tanh(A_k W_k^T + b_k), then the node segment sum
            CK(launch_inc_gather_fwd(mb, D, W(S_PQ + l), PR(P.edge_b[l - 1]), gsrc, gdst, W(S_EA + LK(l, 1)), st));
            for (int k = 1; k < x.K; ++k)
                CK(launch_gemm_nt(W(S_EA + LK(l, k)), mb.NI, D, PR(P.edge_wk[l - 1][k - 1]), D, PR(P.edge_bk[l - 1][k - 1]), nullptr,
                                  W(S_EA + LK(l, k + 1)), 1, st, prof));
            CK(launch_inc_scatter_fwd(pk, mb, D, l == x.L, W(S_EA + LK(l, x.K)), grev, cand_inc, W(S_H + l - 1), W(S_H + l), W(S_HBARV),
                                      W(S_HBARE), W(S_C), (l == x.L && land) ? W(S_FE) : nullptr, st, fe_full));
            continue;
        }
        // the last layer also writes the land-use pointer-head inputs FE (needs C, computed above)
        CK(launch_edge_fwd(pk, mb, D, l == x.L, W(S_PQ + l), PR(P.edge_b[l - 1]), W(S_H + l - 1), W(S_H + l), W(S_HBARV), W(S_HBARE),
                           W(S_C), (l == x.L && land) ? W(S_FE) : nullptr, st, prof, (l == 1 && fold) ? &fa : nullptr, fe_full, pqflag));
    }
    // the land-use head's first Linear (needs FE of the last layer, C, constb) goes to the side stream next to the attention
    if (forked && g_side_heads && land && !road && fe_half) {
        CK(fork_side(sc, st));
        side_guard.armed = true;
        CK(launch_head_hidden_fwd(pk, mb, D, W(S_FE), W(S_C), W(S_W1F), W(S_CONSTB), W(S_HIDL), sc->side));
        head_on_side = true;
    }
    // ---- 5. attention core, then the per-sample chain after it (out-projection, state_value, value head)
    CK(launch_attn_fwd(pk, mb, D, x.heads, W(S_H + x.L), W(S_R), W(S_ALPHA), W(S_S), st));
    {
        ChainFwdPost a;
        memset(&a, 0, sizeof(a));
        a.d = cd; a.rows = mb.rows;
        a.s = W(S_S); a.hbarV = W(S_HBARV); a.hbarE = W(S_HBARE); a.Ulast = W(S_U + d.n_num);
        a.WvvT = W(S_WVVT); a.bvv = W(S_BVV); a.WoT = W(S_WOT); a.bo = PR(P.outproj_b);
        for (int i = 0; i < d.n_value; ++i) { a.WvT[i] = W(S_WVT + i); a.bv[i] = PR(P.value_b[i]); }
        a.o = W(S_O); a.att = W(S_ATT); a.SV = W(S_SV);
        for (int i = 1; i < d.n_value; ++i) a.V[i] = W(S_V + i);
        a.value = value_dev;
        CK(launch_chain_fwd_post(a, st));
    }
    }
    const float *HL = W(S_H + x.L);
    // ---- 6. pointer heads (policy.py:19-65)
    if (land && !head_on_side) {
        // factorised first Linear: hid = tanh(FE [Wa+Wd | Wc]^T + ((Wb-Wd) c_b + b1)), the bias rows are
        // pre-written into hid and accumulated in place
        if (fe_half) {
            // per-graph effective weight W_b = (Wa + Wd) + Wc diag(c_b) in LDS, hid = tanh(W_b m + constb) in one kernel
            CK(launch_head_hidden_fwd(pk, mb, D, W(S_FE), W(S_C), W(S_W1F), W(S_CONSTB), W(S_HIDL), st));
        } else {
            CK(launch_he_bias_rows(pk, mb, x.h0l, W(S_CONSTB), W(S_HIDL), st));
            CK(launch_gemm_nt(W(S_FE), mb.Nhe, 2 * D, W(S_W1F), x.h0l, nullptr, W(S_HIDL), W(S_HIDL), 1, st, prof));
        }
    }
    if (road) {
        CK(launch_road_gather(pk, mb, D, HL, W(S_XR), st));
        CK(launch_gemm_nt(W(S_XR), mb.Nrn, D, PR(P.road_w[0]), x.h0r, PR(P.road_b0), nullptr, W(S_HIDR), 1, st, prof));
    }
    if (head_on_side) {
        CK(join_side(sc, st));
        side_guard.armed = false;
    }
    // second (bias-free) Linear + masked softmax + log-prob / entropy in one kernel; the entropy is kept for the backward
    CK(launch_pointer_fwd2(pk, mb, W(S_HIDL), PR(P.land_w[1]), x.h0l, W(S_HIDR), PR(P.road_w[1]), x.h0r, W(S_Z_HE), W(S_Z_RN), W(S_P_HE),
                           W(S_P_RN), logp_dev, ent_dev, W(S_LSE), W(S_ENTK), st));
    return UPAMD_OK;
}

// =============================================================================================
// backward
// =============================================================================================
static int backward_impl(upamd_engine *eng, const void *packed_dev, const upamd_pack_layout *layout,
                         const upamd_minibatch *mbp, const float *prm, void *ws_dev, int64_t ws_bytes,
                         const float *dvalue_dev, const float *dlogp_dev, const float *dent_dev,
                         float *grads, hipStream_t st, SideCtx *bk) {
    Plan pl;
    CK(check_args(eng, packed_dev, layout, mbp, prm, ws_dev, ws_bytes, &pl));
    if (!dvalue_dev || !dlogp_dev || !dent_dev || !grads) return fail(UPAMD_E_INVALID, "null seed/grad pointer");
    const upamd_model_desc &d = eng->d;
    const ParamLayout &P = eng->P;
    const Dims x = dims_of(d);
    const int D = x.D, B = mbp->B;
    const PackedView pk = make_view(packed_dev, *layout);
    MbView mb = make_mb(*mbp);
    float *ws = static_cast<float *>(ws_dev);
    auto W = [&](int slot) { return ws + pl.off[slot]; };
    mb.rows = reinterpret_cast<const int32_t *>(W(S_ROWS));     // written by the forward of this minibatch
    auto PR = [&](int idx) { return prm + P.off(idx); };
    auto GR = [&](int idx) { return grads + P.off(idx); };
    Profiler *prof = &eng->prof;
    if (!tiny_supported(d, mbp->max_n, mbp->max_inc, mbp->max_cand) && eng->general_fwd_complete &&
        eng->general_fwd.find(ws_dev) == eng->general_fwd.end())
        return fail(UPAMD_E_INVALID, "upamd_backward: the last forward on this workspace did not run the general kernels (no forward at all, or "
                                     "the fused small-model path -- was tune knob tiny_fused changed between forward and backward?): the "
                                     "activations this backward reads were never written");
    if (tiny_supported(d, mbp->max_n, mbp->max_inc, mbp->max_cand)) {
        // fused small-model path: forward recomputed inside the kernel, backward from the given seeds, slabs -> grads (added)
        TinyIO io;
        memset(&io, 0, sizeof(io));
        io.mode = 1;
        io.dvalue = dvalue_dev; io.dlogp = dlogp_dev; io.dent = dent_dev;
        io.slab = W(S_TINY_SLAB); io.scratch = W(S_TINY_SCR); io.loss_rows = W(S_TINY_LOSS);
        io.grads = grads; io.accumulate = 1;
        const int began = prof_begin(prof, "tiny_bwd", st, 0.0, 0.0);
        const int rc = launch_tiny(d, P, pk, mb, prm, io, st);
        prof_end(prof, "tiny_bwd", st, began);
        return rc;
    }
    const float *Win = x.mlp ? prm : PR(P.inproj_w);                                           // (unused by the rl-mlp encoder)
    const float *Wiq = Win, *Wik = Win + (int64_t)D * D, *Wiv = Win + 2LL * D * D;
    float *gWin = x.mlp ? grads : GR(P.inproj_w), *gbin = x.mlp ? grads : GR(P.inproj_b);
    const float *HL = W(S_H + x.L);
    const bool land = mb.Nhe > 0, road = mb.Nrn > 0;
    const bool fe_half = head_fe_half_ok(D, x.h0l);      // the forward's decision (same process-wide knob)
    const ChainDims cd = chain_dims(d, x, B);
    Reducer red1, red2;             // reduction #1: everything independent; #2: what must follow the collapsed-product gradients
    // the step's ONE grouped dY^T X launch (per-sample weight gradients, and the node-level products of small models);
    // the reductions of its slabs are queued after it has been launched
    TnJobs tj;
    std::vector<std::function<int()>> after_gtn;
    const bool defer = defer_node_tn(x.D);
    // slabs = A[rows, I]^T Bm[rows, J] (panel-major operands) followed by red_add(S) -- at once, or as a grouped job
    hipStream_t tn_stream = st;          // (the side stream for the GCN weight gradients under tune knob side_wgrad)
    auto node_tn_red = [&](const float *A, int I, const float *Bm, int J, int64_t rows, float *slabs,
                           std::function<int(int)> red_add) -> int {
        int Sn = 1;
        if (defer && !tn_shape_mfma_ok(I, J) && tj.n < TN_MAX_JOBS / 4) {
            CK(tn_add(&tj, A, 0, I, Bm, 0, J, rows, slabs, &Sn, 1, 1));
            after_gtn.push_back([red_add, Sn]() { return red_add(Sn); });
            return 0;
        }
        CK(node_tn(A, I, Bm, J, rows, slabs, &Sn, tn_stream, prof));
        return red_add(Sn);
    };
    red1.st = st; red2.st = st;
    // forked step: the big slab sets (the GCN layers' weight gradients, the land-use head's) are reduced on the side stream as
    // soon as their producer has run, underneath the following GEMM / message-passing launches, instead of in the final flush
    Reducer redS;
    // gradient buckets (tune knob grad_buckets, forked step only): `redE` takes every reduction whose destination lies in the
    // attention / value-head / pointer-head part of the flat buffer; it is flushed on the side stream right behind the grouped
    // per-sample products, followed there by the prepared-parameter mappings of that range -- bucket 0 is final before the last
    // GCN layer's backward has finished.  Not forked: `rE` IS red1 and everything happens where it always did.
    Reducer redE;
    const bool early = g_grad_buckets != 0 && g_side_stream != 0 && !defer_node_tn(x.D) && !x.mlp;
    Reducer &rE = early ? redE : red1;

    if (x.mlp) {
        // ===== rl-mlp encoder: value head / numerical encoder -> pooled means + pointer heads -> node encoder
        {
            ChainBwdPost a;
            memset(&a, 0, sizeof(a));
            a.d = cd; a.dvalue = dvalue_dev;
            for (int i = 1; i < d.n_value; ++i) a.V[i] = W(S_V + i);
            for (int i = 0; i <= d.n_num; ++i) a.U[i] = W(S_U + i);
            for (int i = 0; i < d.n_value; ++i) { a.Wv[i] = PR(P.value_w[i]); a.dAv[i] = W(S_DAV + i); }
            for (int i = 0; i < d.n_num; ++i) { a.Wn[i] = PR(P.num_w[i]); a.dAn[i] = W(S_DAN + i); }
            a.dSV = W(S_DSV);
            CK(launch_chain_bwd_post(a, st));
        }
        const float *dSVm = W(S_DSV);
        float *G0 = W(S_G0);
        CK(launch_pointer_bwd2(pk, mb, W(S_Z_HE), W(S_Z_RN), W(S_P_HE), W(S_P_RN), W(S_ENTK), W(S_LSE), dlogp_dev, dent_dev, W(S_HIDL),
                               PR(P.land_w[1]), x.h0l, W(S_HIDR), PR(P.road_w[1]), x.h0r, W(S_DZ_HE), W(S_DZ_RN), W(S_DPREL), W(S_DPRER),
                               land ? W(S_CSP0) : nullptr, land ? W(S_DCONST) : nullptr, road ? W(S_CSP2) : nullptr, road ? W(S_CSP3) : nullptr, st));
        if (land) {
            // dw2 = sum_rows (sum_k dz hid), db1 = sum_rows dconst: the row sums come out of pointer_bwd2
            CK(red1.add(W(S_CSP0), B, x.h0l, 1, x.h0l, 0, x.h0l, GR(P.land_w[1]), x.h0l));
            CK(red1.add(W(S_DCONST), B, x.h0l, 1, x.h0l, 0, x.h0l, GR(P.land_b0), x.h0l));
            if (fe_half) {
                int Sn = 1;
                CK(launch_head_wgrad(pk, mb, D, W(S_FE), W(S_C), W(S_DPREL), W(S_SLAB_FE), &Sn, st));
                CK(red1.add(W(S_SLAB_FE), Sn, 2LL * D * x.h0l, 2 * D, x.h0l, 1, x.h0l, W(S_DW1F), 2 * D, nullptr, 1));
            } else {
                CK(node_tn_red(W(S_FE), 2 * D, W(S_DPREL), x.h0l, mb.Nhe, W(S_SLAB_FE), [&](int Sn) {
                    return red1.add(W(S_SLAB_FE), Sn, 2LL * D * x.h0l, 2 * D, x.h0l, 1, x.h0l, W(S_DW1F), 2 * D, nullptr, 1);
                }));
            }
            // a candidate that is not a live edge has the bias as its embedding: its gradient is kept (-> dbe)
            if (he_feat_bwd_fused_ok(D, x.h0l)) {
                CK(launch_he_feat_bwd_fused(pk, mb, D, W(S_FE), W(S_C), W(S_DPREL), W(S_W1FT), W(S_DMHE), W(S_DC_HEAD), st, 1));
            } else {
                CK(launch_gemm_nt(W(S_DPREL), mb.Nhe, x.h0l, W(S_W1FT), 2 * D, nullptr, nullptr, W(S_DFE), 0, st, prof));
                CK(launch_he_feat_bwd(pk, mb, D, W(S_FE), W(S_C), W(S_DFE), W(S_DMHE), W(S_DC_HEAD), st, 1));
            }
        }
        // G^0 = the node mean's share + every candidate's dM at its selected endpoint; rows [2B, 3B) of the node-encoder
        // job = the per-graph sums of the non-live candidates' dM (bias only: their X rows are zero)
        CK(launch_mlp_pool_bwd(pk, mb, D, dSVm + x.S_last, x.Wp, land ? W(S_DMHE) : nullptr, G0, W(S_DC) + 2LL * B * D, st));
        if (road) {
            CK(red1.add(W(S_CSP2), B, x.h0r, 1, x.h0r, 0, x.h0r, GR(P.road_w[1]), x.h0r));
            CK(red1.add(W(S_CSP3), B, x.h0r, 1, x.h0r, 0, x.h0r, GR(P.road_b0), x.h0r));
            CK(node_tn_red(W(S_XR), D, W(S_DPRER), x.h0r, mb.Nrn, W(S_SLAB_XR), [&](int Sn) {
                return red1.add(W(S_SLAB_XR), Sn, (int64_t)D * x.h0r, D, x.h0r, 1, x.h0r, GR(P.road_w[0]), D);
            }));
            CK(launch_gemm_nt(W(S_DPRER), mb.Nrn, x.h0r, W(S_R1T), D, nullptr, nullptr, W(S_DXR), 0, st, prof));
            CK(launch_road_scatter_add(pk, mb, D, W(S_DXR), G0, st));
        }
        {
            ChainBwdPre a;
            memset(&a, 0, sizeof(a));
            a.d = cd;
            a.dconst = land ? W(S_DCONST) : nullptr; a.dC_head = land ? W(S_DC_HEAD) : nullptr;
            a.Wbd = W(S_WBD); a.dSV = dSVm; a.dC = W(S_DC);
            CK(launch_chain_bwd_pre(a, st));
        }
        // node encoder on all nodes: dWe += G^0^T X, dbe += colsum(G^0) (Xp's column of ones)
        CK(node_tn_red(G0, D, W(S_XP), 32, mb.M, W(S_SLAB_XP2), [&](int Sn) {
            return red1.add(W(S_SLAB_XP2), Sn, (int64_t)D * 32, D, 32, 0, x.F, GR(P.node_w), x.F, GR(P.node_b));
        }));
        {
            float *slab = W(S_SLAB_SMALL);
            int64_t used = 0;
            struct Pending { Reducer *rd; const float *slab; int S, N, K; float *dst; int ldd, overwrite; };
            std::vector<Pending> pending;
            auto job = [&](Reducer &rd, const float *A, int64_t lda, int N, const float *X, int64_t ldx, int K, int64_t rows, float *dst,
                           int ldd, int overwrite) -> int {
                const int splits = tn_job_splits(rows);
                if (used + (int64_t)splits * N * K > pl.small_slab_floats) return fail(UPAMD_E_WORKSPACE, "small-slab region exhausted");
                int Sj = 1;
                CK(tn_add(&tj, A, lda, N, X, ldx, K, rows, slab + used, &Sj));
                pending.push_back(Pending{&rd, slab + used, Sj, N, K, dst, ldd, overwrite});
                used += (int64_t)Sj * N * K;
                return 0;
            };
            for (int i = 0; i < d.n_value; ++i) {
                const int N = d.value_hidden[i];
                const int K = i == 0 ? x.W : d.value_hidden[i - 1];
                const float *X = i == 0 ? W(S_SV) : W(S_V + i);
                CK(job(red1, W(S_DAV + i), N, N, X, i == 0 ? x.Wp : K, K, B, GR(P.value_w[i]), K, 0));
                CK(job(red1, W(S_DAV + i), N, N, nullptr, 0, 1, B, GR(P.value_b[i]), 1, 0));
            }
            for (int i = 0; i < d.n_num; ++i) {
                const int N = d.num_hidden[i];
                const int K = i == 0 ? x.Fn : d.num_hidden[i - 1];
                CK(job(red1, W(S_DAN + i), N, N, W(S_U + i), K, K, B, GR(P.num_w[i]), K, 0));
                CK(job(red1, W(S_DAN + i), N, N, nullptr, 0, 1, B, GR(P.num_b[i]), 1, 0));
            }
            if (land) CK(job(red1, W(S_DCONST), x.h0l, x.h0l, W(S_C), D, D, B, W(S_DWBD), D, 1));
            // the node encoder's three per-sample uses in ONE job over 3B rows: [dC ; dhbarE ; non-live dM sums] x
            // [current node ; mean edge features ; 0]
            CK(job(red2, W(S_DC), D, D, W(S_CURG), UPAMD_NODE_PAD, x.F, 3LL * B, GR(P.node_w), x.F, 0));
            CK(job(red2, W(S_DC), D, D, nullptr, 0, 1, 3LL * B, GR(P.node_b), 1, 0));
            CK(launch_gtn(tj, st));
            for (const Pending &q : pending) CK(q.rd->add(q.slab, q.S, (int64_t)q.N * q.K, q.N, q.K, 0, q.K, q.dst, q.ldd, nullptr, q.overwrite));
            for (auto &f : after_gtn) CK(f());
        }
        CK(red1.flush());
        if (land) {
            PermJobs pj;
            int blocks = 0;
            CK(perm_add(&pj, &blocks, PERM_LAND_SCATTER, W(S_DW1F), W(S_DWBD), GR(P.land_w[0]), nullptr, nullptr, nullptr, x.h0l, D, 0));
            CK(launch_permute(pj, blocks, st));
        }
        CK(red2.flush());
        return UPAMD_OK;
    }
    // The forked step: side stream next to the caller's.  With `heads_side` the fork happens HERE and the pointer-head chain
    // (softmax / second-Linear backward -> feature backward -> weight gradient) runs on the side stream next to the value-head
    // chain and the attention backward on the caller's; the two meet at the last GCN layer's backward (ev_b: dM is ready) and
    // at the per-sample chain (ev_a: dr is ready).  Otherwise the fork is where round 2 had it (step 4).
    const bool forked = g_side_stream != 0 && !defer;
    const bool heads_side = forked && g_side_heads && land && !road && fe_half && he_feat_bwd_fused_ok(D, x.h0l);
    SideGuard side_guard;
    SideCtx *sc = nullptr;
    if (forked) {
        CK(side_ready(eng, st, &sc));
        side_guard.c = sc;
        red1.side = sc;
        red2.side = sc;
    }
    if (heads_side) {
        CK(fork_side(sc, st));
        side_guard.armed = true;
    }
    hipStream_t hs = heads_side ? sc->side : st;      // the stream of the pointer-head chain
    // ---- 1. per-sample chain, the part after the attention: value head, numerical encoder, out-projection, Wvv
    {
        ChainBwdPost a;
        memset(&a, 0, sizeof(a));
        a.d = cd; a.dvalue = dvalue_dev;
        for (int i = 1; i < d.n_value; ++i) a.V[i] = W(S_V + i);
        for (int i = 0; i <= d.n_num; ++i) a.U[i] = W(S_U + i);
        for (int i = 0; i < d.n_value; ++i) { a.Wv[i] = PR(P.value_w[i]); a.dAv[i] = W(S_DAV + i); }
        for (int i = 0; i < d.n_num; ++i) { a.Wn[i] = PR(P.num_w[i]); a.dAn[i] = W(S_DAN + i); }
        a.Wo = PR(P.outproj_w); a.Wvv = W(S_WVV);
        a.dSV = W(S_DSV); a.datt = W(S_DATT); a.dov = W(S_DO); a.ds = W(S_DS);
        CK(launch_chain_bwd_post(a, st));
    }
    const float *dSV = W(S_DSV);                     // [B, W] with row stride Wp
    const float *dhbarV = dSV + x.S_last;            // column slices, ld = Wp
    const float *dhbarE = dSV + x.S_last + D;
    // ---- 2. attention core: writes G^L (mean + attention terms) and dr
    float *G = W(S_G0), *Gn = W(S_G1);
    CK(launch_attn_bwd(pk, mb, D, x.heads, HL, W(S_R), W(S_ALPHA), W(S_S), W(S_DS), dhbarV, x.Wp, G, W(S_DR), st));
    if (heads_side) UPAMD_HIP(hipEventRecord(sc->ev_a, st));      // dr + everything chain_bwd_post wrote: the side chain waits for it below
    // ---- 3. pointer heads: softmax backward + second-Linear backward fused
    CK(launch_pointer_bwd2(pk, mb, W(S_Z_HE), W(S_Z_RN), W(S_P_HE), W(S_P_RN), W(S_ENTK), W(S_LSE), dlogp_dev, dent_dev, W(S_HIDL),
                           PR(P.land_w[1]), x.h0l, W(S_HIDR), PR(P.road_w[1]), x.h0r, W(S_DZ_HE), W(S_DZ_RN), W(S_DPREL), W(S_DPRER),
                               land ? W(S_CSP0) : nullptr, land ? W(S_DCONST) : nullptr, road ? W(S_CSP2) : nullptr, road ? W(S_CSP3) : nullptr, hs));
    if (land) {
        // dw2 = sum_rows (sum_k dz hid), db1 = sum_rows dconst: the row sums come out of pointer_bwd2
        CK(rE.add(W(S_CSP0), B, x.h0l, 1, x.h0l, 0, x.h0l, GR(P.land_w[1]), x.h0l));
        CK(rE.add(W(S_DCONST), B, x.h0l, 1, x.h0l, 0, x.h0l, GR(P.land_b0), x.h0l));
        if (heads_side) {
            // feature backward first (the last GCN layer's backward on the caller's stream waits for its dM), then the weight gradient
            CK(launch_he_feat_bwd_fused(pk, mb, D, W(S_FE), W(S_C), W(S_DPREL), W(S_W1FT), W(S_DMHE), W(S_DC_HEAD), hs));
            CK(stream_after(st, hs, sc->ev_b));
        }
        // dW1f = dpre^T FE (mapped back onto [Wa|Wb|Wc|Wd] after the reduction, together with dWbd = dconst^T C)
        // (with FE = m only: per-graph products, the c part weighted by c_b inside the kernel -- head.hip)
        if (fe_half) {
            int Sn = 1;
            CK(launch_head_wgrad(pk, mb, D, W(S_FE), W(S_C), W(S_DPREL), W(S_SLAB_FE), &Sn, hs));
            if (heads_side) {
                redS.st = hs;
                CK(redS.add(W(S_SLAB_FE), Sn, 2LL * D * x.h0l, 2 * D, x.h0l, 1, x.h0l, W(S_DW1F), 2 * D, nullptr, 1));
                CK(redS.flush());
            } else {
                CK(rE.add(W(S_SLAB_FE), Sn, 2LL * D * x.h0l, 2 * D, x.h0l, 1, x.h0l, W(S_DW1F), 2 * D, nullptr, 1));
            }
        } else {
            CK(node_tn_red(W(S_FE), 2 * D, W(S_DPREL), x.h0l, mb.Nhe, W(S_SLAB_FE), [&](int Sn) {
                return rE.add(W(S_SLAB_FE), Sn, 2LL * D * x.h0l, 2 * D, x.h0l, 1, x.h0l, W(S_DW1F), 2 * D, nullptr, 1);
            }));
        }
        // dFE = dpre W1f, then the feature backward (dMhe for the last GCN layer, dC from the m*c term); with the shipped
        // head width (h0 = 32) one kernel does both and dFE never exists in HBM
        if (heads_side) {
            // (launched above)
        } else if (he_feat_bwd_fused_ok(D, x.h0l)) {
            CK(launch_he_feat_bwd_fused(pk, mb, D, W(S_FE), W(S_C), W(S_DPREL), W(S_W1FT), W(S_DMHE), W(S_DC_HEAD), st));
        } else {
            CK(launch_gemm_nt(W(S_DPREL), mb.Nhe, x.h0l, W(S_W1FT), 2 * D, nullptr, nullptr, W(S_DFE), 0, st, prof));
            CK(launch_he_feat_bwd(pk, mb, D, W(S_FE), W(S_C), W(S_DFE), W(S_DMHE), W(S_DC_HEAD), st));
        }
    }
    if (road) {
        CK(rE.add(W(S_CSP2), B, x.h0r, 1, x.h0r, 0, x.h0r, GR(P.road_w[1]), x.h0r));
        CK(rE.add(W(S_CSP3), B, x.h0r, 1, x.h0r, 0, x.h0r, GR(P.road_b0), x.h0r));
        CK(node_tn_red(W(S_XR), D, W(S_DPRER), x.h0r, mb.Nrn, W(S_SLAB_XR), [&](int Sn) {
            return rE.add(W(S_SLAB_XR), Sn, (int64_t)D * x.h0r, D, x.h0r, 1, x.h0r, GR(P.road_w[0]), D);
        }));
        CK(launch_gemm_nt(W(S_DPRER), mb.Nrn, x.h0r, W(S_R1T), D, nullptr, nullptr, W(S_DXR), 0, st, prof));
        CK(launch_road_scatter_add(pk, mb, D, W(S_DXR), G, st));
    }
    // ---- 3b / 7. every per-sample weight gradient dY^T X in ONE grouped MFMA launch.  Its inputs are complete once the
    // per-sample chain below has run, so in the forked step it is launched on the side stream right behind that chain and
    // only its slab reductions wait for the join (small models put their node-level products into the same launch: not forked)
    struct Pending { Reducer *rd; const float *slab; int S, N, K; float *dst; int ldd, overwrite; };
    std::vector<Pending> pending;           // the slabs are reduced after the grouped launch that fills them
    auto grouped_launch = [&](hipStream_t gs) -> int {
        float *slab = W(S_SLAB_SMALL);
        int64_t used = 0;
        // (A [B][N], X [B][K] | ones) -> dst [N][K] (ld), accumulate or overwrite; bias: X = nullptr
        auto job = [&](Reducer &rd, const float *A, int64_t lda, int N, const float *X, int64_t ldx, int K, float *dst, int ldd,
                       int overwrite) -> int {
            if (tj.n >= TN_MAX_JOBS) return fail(UPAMD_E_LIMIT, "too many per-sample weight-gradient jobs");
            const int splits = tn_job_splits(B);
            if (used + (int64_t)splits * N * K > pl.small_slab_floats) return fail(UPAMD_E_WORKSPACE, "small-slab region exhausted");
            int Sj = 1;
            CK(tn_add(&tj, A, lda, N, X, ldx, K, B, slab + used, &Sj));
            pending.push_back(Pending{&rd, slab + used, Sj, N, K, dst, ldd, overwrite});
            used += (int64_t)Sj * N * K;
            return 0;
        };
        for (int i = 0; i < d.n_value; ++i) {       // value head (value.py:15-34)
            const int N = d.value_hidden[i];
            const int K = i == 0 ? x.W : d.value_hidden[i - 1];
            const float *X = i == 0 ? W(S_SV) : W(S_V + i);
            CK(job(rE, W(S_DAV + i), N, N, X, i == 0 ? x.Wp : K, K, GR(P.value_w[i]), K, 0));
            CK(job(rE, W(S_DAV + i), N, N, nullptr, 0, 1, GR(P.value_b[i]), 1, 0));
        }
        for (int i = 0; i < d.n_num; ++i) {         // numerical encoder
            const int N = d.num_hidden[i];
            const int K = i == 0 ? x.Fn : d.num_hidden[i - 1];
            CK(job(rE, W(S_DAN + i), N, N, W(S_U + i), K, K, GR(P.num_w[i]), K, 0));
            CK(job(rE, W(S_DAN + i), N, N, nullptr, 0, 1, GR(P.num_b[i]), 1, 0));
        }
        CK(job(rE, W(S_DATT), D, D, W(S_O), D, D, GR(P.outproj_w), D, 0));
        CK(job(rE, W(S_DATT), D, D, nullptr, 0, 1, GR(P.outproj_b), 1, 0));
        for (int h = 0; h < x.heads; ++h) {
            // dWvv[h-slice,:] = do[:,h-slice]^T s[:,h,:] ;  dWkk[h-slice,:] = q1[:,h-slice]^T dr[:,h,:]
            CK(job(rE, W(S_DO) + h * x.dh, D, x.dh, W(S_S) + (int64_t)h * D, (int64_t)x.heads * D, D, W(S_DWVV) + (int64_t)h * x.dh * D, D, 1));
            CK(job(rE, W(S_Q1) + h * x.dh, D, x.dh, W(S_DR) + (int64_t)h * D, (int64_t)x.heads * D, D, W(S_DWKK) + (int64_t)h * x.dh * D, D, 1));
        }
        CK(job(rE, W(S_DO), D, D, nullptr, 0, 1, W(S_DBVV), 1, 1));
        CK(job(rE, W(S_DQ1), D, D, W(S_Q0), D, D, gWin, D, 0));                    // in_proj, q rows
        CK(job(rE, W(S_DQ1), D, D, nullptr, 0, 1, gbin, 1, 0));
        CK(job(rE, W(S_DQ0), D, D, W(S_C), D, D, GR(P.q_w), D, 0));
        CK(job(rE, W(S_DQ0), D, D, nullptr, 0, 1, GR(P.q_b), 1, 0));
        if (land) CK(job(rE, W(S_DCONST), x.h0l, x.h0l, W(S_C), D, D, W(S_DWBD), D, 1));
        // the current node's pass through the node encoder: after the collapsed-product gradients (same destination)
        CK(job(red2, W(S_DC), D, D, W(S_CURG), UPAMD_NODE_PAD, x.F, GR(P.node_w), x.F, 0));
        CK(job(red2, W(S_DC), D, D, nullptr, 0, 1, GR(P.node_b), 1, 0));
        CK(launch_gtn(tj, gs));
        return 0;
    };
    // ---- 4. per-sample chain, the part before the graph: dr -> dq1 -> dq0 -> dC (+ the land-use head's two dC terms)
    if (heads_side) {
        UPAMD_HIP(hipStreamWaitEvent(sc->side, sc->ev_a, 0));      // dr, datt, do, ... of the caller's stream
    } else if (forked) {
        CK(fork_side(sc, st));
        side_guard.armed = true;
    }
    {
        ChainBwdPre a;
        memset(&a, 0, sizeof(a));
        a.d = cd; a.dr = W(S_DR);
        a.dconst = land ? W(S_DCONST) : nullptr; a.dC_head = land ? W(S_DC_HEAD) : nullptr;
        a.WkkT = W(S_WKKT); a.Wiq = Wiq; a.Wq = PR(P.q_w); a.Wbd = W(S_WBD);
        a.dq1 = W(S_DQ1); a.dq0 = W(S_DQ0); a.dC = W(S_DC);
        CK(launch_chain_bwd_pre(a, forked ? sc->side : st));
    }
    if (forked) CK(grouped_launch(sc->side));
    // the attention mappings Wkk = Wik Wk, Wvv = Wiv Wv, bvv = Wiv bv + biv back onto the stored parameters
    auto attention_smm = [&](SmmJobs *sj, int *blocks) -> int {
        // Wkk = Wik Wk:  dWik += dWkk Wk^T,  dWk += Wik^T dWkk
        CK(smm_add(sj, blocks, D, D, D, W(S_DWKK), D, 1, PR(P.k_w), 1, D, nullptr, gWin + (int64_t)D * D, D, 1, 1.f));
        CK(smm_add(sj, blocks, D, D, D, Wik, 1, D, W(S_DWKK), D, 1, nullptr, GR(P.k_w), D, 1, 1.f));
        // Wvv = Wiv Wv, bvv = Wiv bv + biv:  dWiv += dWvv Wv^T + dbvv (x) bv,  dWv += Wiv^T dWvv,  dbv += Wiv^T dbvv
        CK(smm_add(sj, blocks, D, D, D, W(S_DWVV), D, 1, PR(P.v_w), 1, D, nullptr, gWin + 2LL * D * D, D, 1, 1.f, nullptr, 0, W(S_DBVV), PR(P.v_b)));
        CK(smm_add(sj, blocks, D, D, D, Wiv, 1, D, W(S_DWVV), D, 1, nullptr, GR(P.v_w), D, 1, 1.f));
        CK(smm_add(sj, blocks, 1, D, D, W(S_DBVV), 0, 1, Wiv, D, 1, nullptr, GR(P.v_b), D, 1, 1.f));
        return 0;
    };
    auto land_scatter = [&](hipStream_t ps) -> int {
        PermJobs pj;
        int blocks = 0;
        CK(perm_add(&pj, &blocks, PERM_LAND_SCATTER, W(S_DW1F), W(S_DWBD), GR(P.land_w[0]), nullptr, nullptr, nullptr, x.h0l, D, 0));
        return launch_permute(pj, blocks, ps);
    };
    const bool early_done = early && forked;
    if (early_done) {
        // ---- bucket 0 on the side stream: every producer of the range has been enqueued on it (or on the caller's stream
        // before the fork / before ev_a); same jobs, same per-destination order as the final flush of the single-range form
        redE.st = sc->side;
        for (const Pending &q : pending)
            if (q.rd == &redE) CK(redE.add(q.slab, q.S, (int64_t)q.N * q.K, q.N, q.K, 0, q.K, q.dst, q.ldd, nullptr, q.overwrite));
        CK(redE.flush());
        if (land) CK(land_scatter(sc->side));
        {
            SmmJobs sj;
            int blocks = 0;
            CK(attention_smm(&sj, &blocks));
            CK(launch_gsmm(sj, blocks, sc->side));
        }
        CK(redE.add(W(S_DBVV), 1, 0, 1, D, 0, D, gbin + 2 * D, D));
        CK(redE.flush());
        CK(bucket_done(bk, P.tensors[P.inproj_w].offset, P.n_floats, sc->side));
    }
    // ---- 5. GCN layers, last to first
    const bool fold = fold_layer1(mb, x.L, x.K);      // the forward's decision (same minibatch): PQ_1 was never written
    const FoldArgs fa{W(S_XP), W(S_W1C), W(S_B1C), W(S_WE_PAD), PR(P.node_b)};
    bool g1_done = false, used_side2 = false;
    hipEvent_t tn_done = nullptr;
    const bool wgrad_side = forked && side_wgrad_on(mb.M) && !defer && x.K == 1 && pl.off[S_DPQ2] >= 0;
    hipEvent_t wgrad_done[MAXL + 2] = {};
    for (int l = x.L; l >= 1; --l) {
        const bool last = (l == x.L);
        float *dPQ = defer ? W(S_DPQL + l) : ((wgrad_side && ((x.L - l) & 1)) ? W(S_DPQ2) : W(S_DPQ));
        // this layer's dP|dQ goes into the buffer layer l + 2 used: its weight gradient (side stream) must have read it
        if (wgrad_side && l + 2 <= x.L && wgrad_done[l + 2]) UPAMD_HIP(hipStreamWaitEvent(st, wgrad_done[l + 2], 0));
        if (x.K > 1) {
            // backward through the sub-layers on the per-incidence rows (deep_edge.hip), top to bottom:
            //   dpre_K = 1/2 (dS_src + dS_dst [+ head term]) (1 - A_K^2);  for j = K-1 .. 1 (linear_j: A_j -> A_j+1):
            //   db_j = colsum dpre_j+1,  dW_j = dpre_j+1^T A_j,  dpre_j = (dpre_j+1 W_j) (1 - A_j^2);  then dP | dQ per node
            const int64_t NIp = std::max<int64_t>(mb.NI, 1);
            const int32_t *grev = reinterpret_cast<const int32_t *>(W(S_EIDX)) + 2 * NIp;
            float *cur = W(S_EDA), *other = W(S_EDB);
            CK(launch_inc_seed_bwd(pk, mb, D, last, W(S_EA + LK(l, x.K)), G, dhbarE, x.Wp, (last && land) ? W(S_DMHE) : nullptr, cur,
                                   W(S_EBP + LK(l, x.K - 1)), st));
            for (int j = x.K - 1; j >= 1; --j) {
                CK(red1.add(W(S_EBP + LK(l, j)), B, D, 1, D, 0, D, GR(P.edge_bk[l - 1][j - 1]), D));
                int Sn = 1;
                CK(node_tn(cur, D, W(S_EA + LK(l, j)), D, mb.NI, W(S_ESLAB + LK(l, j)), &Sn, st, prof));
                CK(red1.add(W(S_ESLAB + LK(l, j)), Sn, (int64_t)D * D, D, D, 0, D, GR(P.edge_wk[l - 1][j - 1]), D));
                CK(launch_gemm_nt(cur, mb.NI, D, W(S_EWT + LK(l, j)), D, nullptr, nullptr, other, 0, st, prof));
                CK(launch_inc_tanh_bwd(pk, mb, D, W(S_EA + LK(l, j)), other, j > 1 ? W(S_EBP + LK(l, j - 1)) : nullptr, st));
                std::swap(cur, other);
            }
            CK(launch_inc_scatter_bwd(pk, mb, D, cur, grev, dPQ, W(S_DBIAS + l), st));
        } else {
            // the forward's decision for this layer (same minibatch, same knobs): P/Q in exp form with block flags
            const bool expf = l >= 2 && pq_exp_layer(pq_gemm(W(S_H + l - 1), mb.M, D, W(S_WCAT + l - 1), W(S_PQ + l)), l, x.K);
            CK(launch_edge_bwd(pk, mb, D, last, W(S_PQ + l), PR(P.edge_b[l - 1]), G, dhbarE, x.Wp, (last && land) ? W(S_DMHE) : nullptr,
                               dPQ, W(S_DBIAS + l), st, prof, (l == 1 && fold) ? &fa : nullptr,
                               expf ? reinterpret_cast<const uint8_t *>(W(S_PQF + l)) : nullptr));
        }
        // column sums of dP | dQ over the minibatch (P/Q panel order); the layer's bias gradient is the P half
        // (layer bucket: reduced next to the layer's weight-gradient slabs instead of in the final flush)
        const bool layer_early = early_done && l > 1 && x.K == 1 && g_side_heads && tn_shape_mfma_ok(2 * D, D);
        if (!layer_early) CK(red1.add(W(S_DBIAS + l), B, 2LL * D, 1, 2 * D, 3, 2 * D, GR(P.edge_b[l - 1]), 0, W(S_CS + l)));
        if (l > 1) {
            // side_wgrad = 1: the weight gradient starts next to this layer's dgrad; 2: behind it, i.e. next to the NEXT layer's
            // message-passing backward (the dgrad is launched first and the side stream waits for it)
            if (wgrad_side && g_side_wgrad == 2) {
                CK(launch_gemm_nt(dPQ, mb.M, 2 * D, W(S_WCATT + l - 1), D, nullptr, G, Gn, 0, st, prof));
                std::swap(G, Gn);
            }
            if (wgrad_side) {
                CK(stream_after(sc->side2, st, next_event(sc)));     // dP|dQ of this layer is complete (2: and its dgrad has run)
                tn_stream = sc->side2;
                used_side2 = true;
            }
            const bool red_side = forked && g_side_heads && !defer && tn_shape_mfma_ok(2 * D, D);
            if (red_side && !wgrad_side) tn_done = next_event(sc);
            CK(node_tn_red(dPQ, 2 * D, W(S_H + l - 1), D, mb.M, W(S_SLAB_W + l), [&, l](int Sn) {
                if (red_side) {
                    // (the slabs are complete once the product has run: reduce them on the side stream right behind it)
                    if (!wgrad_side) CK(stream_after(sc->side, st, tn_done));
                    redS.st = wgrad_side ? sc->side2 : sc->side;
                    CK(redS.add(W(S_SLAB_W + l), Sn, 2LL * D * D, 2 * D, D, 2, D, GR(P.edge_w[l - 1]), 2 * D));
                    if (layer_early) CK(redS.add(W(S_DBIAS + l), B, 2LL * D, 1, 2 * D, 3, 2 * D, GR(P.edge_b[l - 1]), 0, W(S_CS + l)));
                    CK(redS.flush());
                    if (layer_early)
                        CK(bucket_done(bk, P.tensors[P.edge_w[l - 1]].offset, P.tensors[P.edge_b[l - 1] + 1].offset, redS.st));
                    return 0;
                }
                return red1.add(W(S_SLAB_W + l), Sn, 2LL * D * D, 2 * D, D, 2, D, GR(P.edge_w[l - 1]), 2 * D);
            }));
            if (wgrad_side) {
                tn_stream = st;
                wgrad_done[l] = next_event(sc);
                UPAMD_HIP(hipEventRecord(wgrad_done[l], sc->side2));
            }
            if (!(wgrad_side && g_side_wgrad == 2)) {
                CK(launch_gemm_nt(dPQ, mb.M, 2 * D, W(S_WCATT + l - 1), D, nullptr, G, Gn, 0, st, prof));
                std::swap(G, Gn);
            }
            if (l == 2 && forked && g_side_heads && !defer && x.K == 1) {
                // G^1 is complete: the node encoder's HBM-bound G^1^T Xp product (J = 32) goes to the side stream, next to the
                // first layer's message-passing backward and its own dPQ_1^T Xp product
                CK(stream_after(sc->side, st, next_event(sc)));
                tn_stream = sc->side;
                CK(node_tn_red(G, D, W(S_XP), 32, mb.M, W(S_SLAB_XP2), [&](int Sn) {
                    return red1.add(W(S_SLAB_XP2), Sn, (int64_t)D * 32, D, 32, 0, x.F, GR(P.node_w), x.F, GR(P.node_b));
                }));
                tn_stream = st;
                g1_done = true;
            }
        } else {
            // layer 1: H_0 = Xp We^T + be, so dWcat_1 = dPQ_1^T H_0 = (dPQ_1^T Xp) We^T + colsum(dPQ_1) (x) be --
            // a J = 32 reduction over the nodes instead of a full-size weight-gradient GEMM.  Tn = dPQ_1^T Xp
            CK(node_tn_red(dPQ, 2 * D, W(S_XP), 32, mb.M, W(S_SLAB_XP1), [&](int Sn) {
                return red1.add(W(S_SLAB_XP1), Sn, 2LL * D * 32, 2 * D, 32, 0, 32, W(S_TN), 32, nullptr, 1);
            }));
        }
    }
    // ---- 6. node encoder.  G^0 = G^1 + dPQ_1 Wcat_1 is never formed (it is only needed for the encoder's own
    // gradients): dWe = G^0^T X = G^1^T X + Wcat_1^T (dPQ_1^T X),  dbe = colsum(G^1) + Wcat_1^T colsum(dPQ_1).
    // G holds G^1 here.  Xp's column 31 is all ones, so column 31 of G^1^T Xp is colsum(G^1): straight into dbe
    // (forked step: launched on the side stream inside the layer loop, as soon as G^1 existed)
    if (!g1_done)
        CK(node_tn_red(G, D, W(S_XP), 32, mb.M, W(S_SLAB_XP2), [&](int Sn) {
            return red1.add(W(S_SLAB_XP2), Sn, (int64_t)D * 32, D, 32, 0, x.F, GR(P.node_w), x.F, GR(P.node_b));
        }));
    // ---- 7. (not forked: the grouped launch here) then its slab reductions, and the deferred node-level ones
    if (forked) {
        CK(join_side(sc, st));
        if (used_side2) CK(stream_after(st, sc->side2, sc->ev_join2));
        side_guard.armed = false;
    } else {
        CK(grouped_launch(st));
    }
    for (const Pending &q : pending)
        if (!(early_done && q.rd == &redE)) CK(q.rd->add(q.slab, q.S, (int64_t)q.N * q.K, q.N, q.K, 0, q.K, q.dst, q.ldd, nullptr, q.overwrite));
    for (auto &f : after_gtn) CK(f());
    // ---- 8. reduction #1: every split-K slab / partial sum of the step, fixed order
    if (early && !early_done) CK(redE.flush());      // (unreachable today: early implies forked; kept so that no job can be lost)
    CK(red1.flush());
    // ---- 9. gradients of the prepared parameters mapped back onto the stored ones
    if (land && !early_done) CK(land_scatter(st));
    {
        SmmJobs sj;
        int blocks = 0;
        if (!early_done) CK(attention_smm(&sj, &blocks));
        // first GCN layer: dWcat_1 = Tn We^T + cs1 (x) be;  dWe += Wcat_1^T Tn;  dbe += Wcat_1^T cs1   (Tn = dPQ_1^T Xp)
        CK(smm_add(&sj, &blocks, 2 * D, D, 32, W(S_TN), 32, 1, W(S_WE_PAD), 1, 32, nullptr, W(S_DWC1), D, 0, 1.f, nullptr, 0, W(S_CS + 1), PR(P.node_b)));
        CK(smm_add(&sj, &blocks, D, x.F, 2 * D, W(S_WCAT + 0), 1, D, W(S_TN), 32, 1, nullptr, GR(P.node_w), x.F, 1, 1.f));
        CK(smm_add(&sj, &blocks, 1, D, 2 * D, W(S_CS + 1), 0, 1, W(S_WCAT + 0), D, 1, nullptr, GR(P.node_b), D, 1, 1.f));
        CK(launch_gsmm(sj, blocks, st));
    }
    // ---- 10. reduction #2: dWcat_1 un-permuted into the layer's weight, dbiv += dbvv, the current node's encoder pass
    CK(red2.add(W(S_DWC1), 1, 0, 2 * D, D, 2, D, GR(P.edge_w[0]), 2 * D));
    if (!early_done) CK(red2.add(W(S_DBVV), 1, 0, 1, D, 0, D, gbin + 2 * D, D));
    CK(red2.flush());
    if (bk->n_buckets > 0) {
        // the rest of the buffer: everything in front of the earliest bucket (numerical + node encoder, first GCN layer, and the
        // layers whose weight gradient was not reduced on a side stream)
        int64_t lo = P.n_floats;
        for (int k = 0; k < bk->n_buckets; ++k) lo = std::min(lo, bk->b_begin[k]);
        if (lo > 0) CK(bucket_done(bk, 0, lo, st));
    }
    return UPAMD_OK;
}

extern "C" int upamd_backward(upamd_engine *eng, const void *packed_dev, const upamd_pack_layout *layout,
                              const upamd_minibatch *mbp, const float *prm, void *ws_dev, int64_t ws_bytes,
                              const float *dvalue_dev, const float *dlogp_dev, const float *dent_dev,
                              float *grads, void *stream) {
    if (!eng) return fail(UPAMD_E_INVALID, "engine is null");
    hipStream_t st = static_cast<hipStream_t>(stream);
    SideCtx *bk = nullptr;
    CK(side_ready(eng, st, &bk));
    bk->n_buckets = 0;
    int rc = backward_impl(eng, packed_dev, layout, mbp, prm, ws_dev, ws_bytes, dvalue_dev, dlogp_dev, dent_dev, grads, st, bk);
    if (rc) {
        bk->n_buckets = 0;
        return rc;
    }
    if (bk->n_buckets == 0) CK(bucket_done(bk, 0, eng->P.n_floats, st));      // paths that finalise everything at the end
    return UPAMD_OK;
}

extern "C" int upamd_grad_buckets(upamd_engine *eng, void *stream, int32_t cap, int32_t *n_out, int64_t *begin, int64_t *end) {
    if (!eng || !n_out) return fail(UPAMD_E_INVALID, "upamd_grad_buckets: null argument");
    auto it = eng->sides.find(static_cast<hipStream_t>(stream));
    const int n = it == eng->sides.end() ? 0 : it->second.n_buckets;
    *n_out = n;
    if (n > cap) return fail(UPAMD_E_LIMIT, "upamd_grad_buckets: %d buckets, room for %d", n, cap);
    for (int k = 0; k < n; ++k) {
        if (begin) begin[k] = it->second.b_begin[k];
        if (end) end[k] = it->second.b_end[k];
    }
    return UPAMD_OK;
}

extern "C" int upamd_grad_bucket_wait(upamd_engine *eng, void *stream, int32_t k, void *waiter) {
    if (!eng) return fail(UPAMD_E_INVALID, "engine is null");
    auto it = eng->sides.find(static_cast<hipStream_t>(stream));
    if (it == eng->sides.end() || k < 0 || k >= it->second.n_buckets)
        return fail(UPAMD_E_INVALID, "upamd_grad_bucket_wait: no bucket %d recorded on that stream", k);
    UPAMD_HIP(hipStreamWaitEvent(static_cast<hipStream_t>(waiter), it->second.b_ev[k], 0));
    return UPAMD_OK;
}

// =============================================================================================
// fused optimizer-step front end of small models (tiny.hip)
// =============================================================================================
extern "C" int upamd_step_fused_ok(upamd_engine *eng, const upamd_minibatch *mb) {
    if (!eng || !mb) return 0;
    return tiny_supported(eng->d, mb->max_n, mb->max_inc, mb->max_cand) ? 1 : 0;
}

extern "C" int upamd_step_fused(upamd_engine *eng, const void *packed_dev, const upamd_pack_layout *layout, const upamd_minibatch *mbp,
                                const float *prm, void *ws_dev, int64_t ws_bytes, const int64_t *rows_dev, const float *adv_dev,
                                const float *ret_dev, const float *old_logp_dev, const float *exps_dev, float clip_eps, float cv,
                                float ce, float inv_rows, float inv_ind, float *value_dev, float *logp_dev, float *ent_dev,
                                float *grads_dev, float *losses_dev, void *stream) {
    Plan pl;
    CK(check_args(eng, packed_dev, layout, mbp, prm, ws_dev, ws_bytes, &pl));
    if (!adv_dev || !ret_dev || !old_logp_dev || !exps_dev || !value_dev || !logp_dev || !ent_dev || !grads_dev || !losses_dev)
        return fail(UPAMD_E_INVALID, "upamd_step_fused: null argument");
    if (!tiny_supported(eng->d, mbp->max_n, mbp->max_inc, mbp->max_cand))
        return fail(UPAMD_E_INVALID, "upamd_step_fused: this model / minibatch is not covered by the fused small-model path "
                                     "(ask upamd_step_fused_ok first and use upamd_forward / upamd_ppo_loss_rows / upamd_backward)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    float *ws = static_cast<float *>(ws_dev);
    const PackedView pk = make_view(packed_dev, *layout);
    const MbView mb = make_mb(*mbp);
    TinyIO io;
    memset(&io, 0, sizeof(io));
    io.mode = 2;
    io.value = value_dev; io.logp = logp_dev; io.ent = ent_dev;
    io.rows = rows_dev; io.adv = adv_dev; io.ret = ret_dev; io.old_logp = old_logp_dev; io.exps = exps_dev;
    io.clip_eps = clip_eps; io.cv = cv; io.ce = ce; io.inv_rows = inv_rows; io.inv_ind = inv_ind;
    io.slab = ws + pl.off[S_TINY_SLAB]; io.scratch = ws + pl.off[S_TINY_SCR]; io.loss_rows = ws + pl.off[S_TINY_LOSS];
    io.grads = grads_dev; io.accumulate = 0; io.losses = losses_dev;
    Profiler *prof = &eng->prof;
    const int began = prof_begin(prof, "tiny_step", st, 0.0, 0.0);
    const int rc = launch_tiny(eng->d, eng->P, pk, mb, prm, io, st);
    prof_end(prof, "tiny_step", st, began);
    return rc;
}
