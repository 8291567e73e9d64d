// Engine: sequences the kernels of one minibatch forward / backward on a stream.
// The order of operations mirrors tests/csr_model.py (the executable spec) step by step.
#include <cmath>
#include <cstring>
#include <map>

#include "kernels.h"

using namespace upamd;

struct upamd_engine {
    upamd_model_desc d;
    ParamLayout P;
    Profiler prof;
};

namespace {

struct Plan {
    std::map<std::string, int64_t> off;     // float offsets
    std::map<std::string, int64_t> len;
    int64_t total = 0;                      // floats
};

struct Dims {
    int D, L, heads, dh, F, Fn, S_last, W;   // W = width of state_value
    int Wp;                                  // its row stride: W padded to a multiple of 16 so the value head runs on MFMA
    int h0l, h0r;                            // hidden sizes of the two pointer heads
    int maxdim;                              // widest per-sample activation
};

Dims dims_of(const upamd_model_desc &d) {
    Dims x;
    x.D = d.D; x.L = d.L; x.heads = d.heads; x.dh = d.D / d.heads; x.F = d.node_dim; x.Fn = d.numerical_dim;
    x.S_last = d.num_hidden[d.n_num - 1];
    x.W = 3 * d.D + x.S_last + 3;
    x.Wp = (x.W + 15) / 16 * 16;
    x.h0l = d.land_hidden[0];
    x.h0r = d.road_hidden[0];
    x.maxdim = std::max(x.Wp, std::max(x.Fn, d.D));
    for (int i = 0; i < d.n_num; ++i) x.maxdim = std::max(x.maxdim, d.num_hidden[i]);
    for (int i = 0; i < d.n_value; ++i) x.maxdim = std::max(x.maxdim, d.value_hidden[i]);
    return x;
}

int64_t slab_floats(const Dims &x, int64_t B, int64_t M, int64_t Nhe, int64_t Nrn) {
    int64_t s = 0;
    s = std::max<int64_t>(s, (int64_t)tn_splits(2 * x.D, x.D, M) * 2 * x.D * x.D);          // GCN weight grads
    s = std::max<int64_t>(s, (int64_t)tn_splits(2 * x.D, 32, M) * 2 * x.D * 32);            // node encoder (G^1 and dPQ_1 parts)
    s = std::max<int64_t>(s, (int64_t)tn_splits(2 * x.D, x.h0l, Nhe) * 2 * x.D * x.h0l);    // land head
    s = std::max<int64_t>(s, (int64_t)tn_splits(x.D, x.h0r, Nrn) * x.D * x.h0r);            // road head
    s = std::max<int64_t>(s, (int64_t)tn_splits(x.D, x.D, B) * x.D * x.D);                  // per-sample D x D layers
    s = std::max<int64_t>(s, (int64_t)smm_splits((int)B) * x.maxdim * x.maxdim);            // small per-sample layers (split-K)
    s = std::max<int64_t>(s, (int64_t)smm_splits(2 * x.D) * x.D * 32);                       // node-encoder products over K = 2D
    return s;
}

void make_plan(const upamd_model_desc &d, const upamd_minibatch &mb, Plan *pl) {
    const Dims x = dims_of(d);
    const int64_t B = mb.B, M = std::max<int64_t>(mb.n_nodes, 1), NH = std::max<int64_t>(mb.n_he, 1),
                  NR = std::max<int64_t>(mb.n_rn, 1);
    const int D = x.D;
    int64_t off = 0;
    auto add = [&](const std::string &name, int64_t n) {
        pl->off[name] = off;
        pl->len[name] = n;
        off = align_up(off + std::max<int64_t>(n, 1), 64);
    };
    add("rows", B * UPAMD_META_STRIDE);     // int32 row descriptors (MbView::rows)
    add("We_pad", (int64_t)D * 32);
    for (int l = 0; l < x.L; ++l) {
        add("Wcat" + std::to_string(l), 2LL * D * D);
        add("WcatT" + std::to_string(l), 2LL * D * D);
    }
    add("Wkk", (int64_t)D * D); add("Wvv", (int64_t)D * D); add("bvv", D);
    add("W1f", 2LL * D * x.h0l); add("W1fT", 2LL * D * x.h0l); add("Wbd", (int64_t)D * x.h0l); add("R1T", (int64_t)D * x.h0r);
    add("constb", B * x.h0l); add("dconst", B * x.h0l);
    add("wt", (int64_t)x.maxdim * x.maxdim);          // transposed-weight scratch of the per-sample layers
    add("W1c", 2LL * D * 32); add("b1c", 2LL * D);    // first GCN layer collapsed onto the raw node features
    add("dWc1", 2LL * D * D);                         // backward of that collapse
    add("Xp", 2 * M * 16);
    add("U0", B * x.Fn);
    for (int i = 0; i < d.n_num; ++i) add("U" + std::to_string(i + 1), B * d.num_hidden[i]);
    add("curg", B * UPAMD_NODE_PAD);
    add("C", B * D);
    for (int l = 0; l <= x.L; ++l) add("H" + std::to_string(l), M * D);
    for (int l = 1; l <= x.L; ++l) add("PQ" + std::to_string(l), M * 2 * D);
    add("hbarV", B * D); add("hbarE", B * D);
    add("q0", B * D); add("q1", B * D); add("r", B * x.heads * D); add("alpha", (int64_t)x.heads * M);
    add("s", B * x.heads * D); add("o", B * D); add("att", B * D);
    add("SV", B * x.Wp);
    add("Vw0p", (int64_t)d.value_hidden[0] * x.Wp);     // first value-head weight, columns zero-padded to Wp
    for (int i = 0; i < d.n_value; ++i) add("V" + std::to_string(i + 1), B * d.value_hidden[i]);
    add("FE", NH * 2 * D); add("hidl", NH * x.h0l); add("z_he", NH); add("p_he", NH);
    add("XR", NR * D); add("hidr", NR * x.h0r); add("z_rn", NR); add("p_rn", NR);
    add("lse", B); add("entk", B);
    // backward temporaries
    add("dzA", B * x.maxdim); add("dzB", B * x.maxdim); add("dnA", B * x.maxdim); add("dnB", B * x.maxdim);
    add("datt", B * D);
    add("do", B * D); add("ds", B * x.heads * D); add("dr", B * x.heads * D);
    add("dq1", B * D); add("dq0", B * D); add("dC", B * D); add("dC_head", B * D);
    // accumulate-into scratch of the backward, one contiguous block zeroed by a single memset ("zero_end" marks its end)
    add("dWkk", (int64_t)D * D); add("dWvv", (int64_t)D * D); add("dbvv", D);
    add("dW1f", 2LL * D * x.h0l); add("dWbd", (int64_t)D * x.h0l); add("Tn", 2LL * D * 32);
    for (int l = 1; l <= x.L; ++l) add("cs" + std::to_string(l), 2LL * D);     // column sums of dP | dQ per layer
    add("zero_end", 1);
    add("dz_he", NH); add("dz_rn", NR); add("dprel", NH * x.h0l); add("dFE", NH * 2 * D); add("dMhe", NH * D);
    add("dprer", NR * x.h0r); add("dXR", NR * D);
    add("G0", M * D); add("G1", M * D); add("dPQ", M * 2 * D); add("dbias_part", B * 2 * D);
    add("slabs", slab_floats(x, B, M, NH, NR));
    const int64_t maxrows = std::max(M, std::max(NH, NR));
    add("cs_part", (int64_t)colsum_pm_blocks(maxrows) * std::max(4 * D, 64));
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
    v.numerical = reinterpret_cast<const float *>(b + L.off_numerical);
    v.cur = reinterpret_cast<const float *>(b + L.off_cur);
    v.Fn = L.numerical_dim;
    return v;
}

MbView make_mb(const upamd_minibatch &mb) {
    MbView v;
    v.B = mb.B; v.M = mb.n_nodes; v.Nhe = mb.n_he; v.Nrn = mb.n_rn; v.max_n = mb.max_n; v.max_inc = mb.max_inc;
    v.idx = mb.idx_dev; v.node_off = mb.node_off_dev; v.he_off = mb.he_off_dev; v.rn_off = mb.rn_off_dev;
    v.rows = nullptr;
    return v;
}

int check_args(upamd_engine *eng, const void *packed, const upamd_pack_layout *layout, const upamd_minibatch *mb,
               const float *params, void *ws, int64_t ws_bytes, Plan *pl) {
    if (!eng || !packed || !layout || !mb || !params || !ws) return fail(UPAMD_E_INVALID, "null argument");
    if (mb->B <= 0 || mb->n_nodes <= 0) return fail(UPAMD_E_INVALID, "empty minibatch (B=%d, nodes=%lld)", mb->B, (long long)mb->n_nodes);
    if (!mb->idx_dev || !mb->node_off_dev || !mb->he_off_dev || !mb->rn_off_dev) return fail(UPAMD_E_INVALID, "minibatch schedule pointers are null");
    if (layout->node_dim != eng->d.node_dim || layout->numerical_dim != eng->d.numerical_dim)
        return fail(UPAMD_E_INVALID, "packed replay feature sizes (%d,%d) do not match the model (%d,%d)", layout->node_dim,
                    layout->numerical_dim, eng->d.node_dim, eng->d.numerical_dim);
    if (reinterpret_cast<uintptr_t>(ws) % 256 != 0) return fail(UPAMD_E_INVALID, "workspace must be 256-byte aligned");
    make_plan(eng->d, *mb, pl);
    if (pl->total * 4 > ws_bytes) return fail(UPAMD_E_WORKSPACE, "workspace too small: need %lld bytes, got %lld", (long long)pl->total * 4, (long long)ws_bytes);
    return 0;
}

#define CK(expr)            \
    do {                    \
        int _rc = (expr);   \
        if (_rc) return _rc; \
    } while (0)

// Row-major [rows, .] linear algebra of the per-sample layers.  Large, well-shaped products go to the MFMA
// kernels (row-major operand variants); everything else to the generic strided kernel.
struct Lin {
    hipStream_t st;
    Profiler *prof;
    float *slabs;      // split-K scratch
    float *wt;         // transposed-weight scratch
    static constexpr int MIN_ROWS = 256;

    // Y[R,N] = scale * act(X[R,K] W[N,K]^T + b)
    int nt(const float *X, int64_t ldx, int R, int K, const float *W, int64_t ldw, const float *b, int N, float *Y,
           int64_t ldy, int act, float scale) const {
        GemmNT g{X, R, K, ldx, true, W, N, ldw, b, nullptr, Y, ldy, true, act, scale};
        if (R >= MIN_ROWS && gemm_nt_mfma_ok(g)) return launch_gemm_nt_ex(g, st, prof);
        return launch_smm(R, N, K, X, ldx, 1, W, 1, ldw, b, Y, ldy, 0, act, scale, st);
    }
    // Y[R,N] = X[R,K] Wm[K,N]   (Wm row-major dense, ld = N)
    int nn(const float *X, int64_t ldx, int R, int K, const float *Wm, int N, float *Y, int64_t ldy) const {
        GemmNT g{X, R, K, ldx, true, Wm, N, N, nullptr, nullptr, Y, ldy, true, 0, 1.f};
        g.w_kn = true;                       // the kernel reads the [K][N] weight as it is
        if (R >= MIN_ROWS && gemm_nt_mfma_ok(g)) return launch_gemm_nt_ex(g, st, prof);
        return launch_smm(R, N, K, X, ldx, 1, Wm, N, 1, nullptr, Y, ldy, 0, 0, 1.f, st);
    }
    // dW[N,K] += dY[R,N]^T X[R,K];  db[N] += colsum(dY)
    // (keep < K: X's trailing columns are zero padding, dW is [N, keep])
    int tn_acc(const float *dY, int64_t ldy, int R, int N, const float *X, int64_t ldx, int K, float *dW, float *db,
               int keep = -1) const {
        if (keep < 0) keep = K;
        GemmTN g{dY, N, ldy, X, K, ldx, R, true, slabs};
        if (R >= MIN_ROWS && gemm_tn_mfma_ok(g)) {
            int S = 1;
            CK(launch_gemm_tn_ex(g, &S, st, prof));
            CK(launch_reduce_slabs(slabs, S, N, K, 0, keep, dW, keep, st));
        } else if (R >= 64) {       // reduction over rows, small output: split-K, fixed-order reduce
            int S = 1;
            CK(launch_smm_splitk(N, K, R, dY, 1, ldy, X, ldx, 1, slabs, &S, st));
            CK(launch_reduce_slabs(slabs, S, N, K, 0, keep, dW, keep, st));
        } else {
            CK(launch_smm(N, keep, R, dY, 1, ldy, X, ldx, 1, nullptr, dW, keep, 1, 0, 1.f, st));
        }
        if (db) CK(launch_colsum_rm(dY, R, N, ldy, db, st));
        return 0;
    }
    // Y[R,N] += X[R,K] W[N,K]^T
    int nt_acc(const float *X, int64_t ldx, int R, int K, const float *W, int64_t ldw, int N, float *Y, int64_t ldy) const {
        GemmNT g{X, R, K, ldx, true, W, N, ldw, nullptr, Y, Y, ldy, true, 0, 1.f};
        if (R >= MIN_ROWS && gemm_nt_mfma_ok(g)) return launch_gemm_nt_ex(g, st, prof);
        return launch_smm(R, N, K, X, ldx, 1, W, 1, ldw, nullptr, Y, ldy, 1, 0, 1.f, st);
    }
};

}  // namespace

extern "C" int upamd_engine_create(const upamd_model_desc *desc, upamd_engine **out) {
    if (!out) return fail(UPAMD_E_INVALID, "upamd_engine_create: out is null");
    ParamLayout P;
    int rc = build_param_layout(desc, &P);
    if (rc) return rc;
    if (desc->n_land != 2 || desc->n_road != 2)
        return fail(UPAMD_E_INVALID, "this build supports pointer heads of the form [hidden, 1] only (got %d and %d layers)", desc->n_land, desc->n_road);
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
    make_plan(eng->d, *mb, &pl);
    *bytes = pl.total * 4 + 256;
    return UPAMD_OK;
}

extern "C" int upamd_ws_tensor(upamd_engine *eng, const upamd_minibatch *mb, const char *name, int64_t *byte_offset,
                               int64_t *rows, int64_t *cols, int32_t *kind) {
    if (!eng || !mb || !name) return fail(UPAMD_E_INVALID, "null argument");
    Plan pl;
    make_plan(eng->d, *mb, &pl);
    auto it = pl.off.find(name);
    if (it == pl.off.end()) return fail(UPAMD_E_INVALID, "unknown workspace tensor '%s'", name);
    const Dims x = dims_of(eng->d);
    const std::string n(name);
    int64_t r = 0, c = 0;
    int k = 0;
    const int64_t B = mb->B, M = mb->n_nodes, NH = mb->n_he, NR = mb->n_rn;
    if (n[0] == 'H' && isdigit(n[1])) { r = M; c = x.D; k = 1; }
    else if (n.rfind("PQ", 0) == 0 || n == "dPQ") { r = M; c = 2 * x.D; k = 1; }
    else if (n == "G0" || n == "G1") { r = M; c = x.D; k = 1; }
    else if (n == "Xp") { r = M; c = 32; k = 1; }
    else if (n == "FE" || n == "dFE") { r = NH; c = 2 * x.D; k = 1; }
    else if (n == "hidl" || n == "dprel") { r = NH; c = x.h0l; k = 1; }
    else if (n == "dMhe") { r = NH; c = x.D; k = 1; }
    else if (n == "XR" || n == "dXR") { r = NR; c = x.D; k = 1; }
    else if (n == "hidr" || n == "dprer") { r = NR; c = x.h0r; k = 1; }
    else if (n == "z_he" || n == "p_he" || n == "dz_he") { r = NH; c = 1; }
    else if (n == "z_rn" || n == "p_rn" || n == "dz_rn") { r = NR; c = 1; }
    else if (n == "alpha") { r = x.heads; c = M; }
    else if (n == "SV") { r = B; c = x.Wp; }      // columns >= W are zero padding
    else if (n == "r" || n == "s" || n == "ds" || n == "dr") { r = B; c = (int64_t)x.heads * x.D; }
    else if (n == "lse" || n == "entk") { r = B; c = 1; }
    else if (n == "U0") { r = B; c = x.Fn; }
    else if (n == "curg") { r = B; c = UPAMD_NODE_PAD; }
    else if (n == "Wkk" || n == "Wvv" || n == "dWkk" || n == "dWvv") { r = x.D; c = x.D; }
    else if (n[0] == 'U' || n[0] == 'V') { r = B; c = pl.len[n] / std::max<int64_t>(B, 1); }
    else { r = B; c = x.D; }     // C, hbarV, hbarE, q0, q1, o, att, do, dq*, dC*
    if (byte_offset) *byte_offset = it->second * 4;
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
    auto W = [&](const std::string &n) { return ws + pl.off.at(n); };
    auto PR = [&](int idx) { return prm + P.off(idx); };
    Profiler *prof = &eng->prof;
    const Lin lin{st, prof, W("slabs"), W("wt")};
    // row descriptors first: every per-graph kernel below (and the backward) reads them
    CK(launch_gather_rows(pk, mb, reinterpret_cast<int32_t *>(W("rows")), st));
    mb.rows = reinterpret_cast<const int32_t *>(W("rows"));

    // -- per-step weight preparation (tiny)
    CK(launch_pad_cols(PR(P.node_w), D, x.F, 32, W("We_pad"), st));
    for (int l = 0; l < x.L; ++l)
        CK(launch_prep_wcat(PR(P.edge_w[l]), D, W("Wcat" + std::to_string(l)), W("WcatT" + std::to_string(l)), st));
    const float *Win = PR(P.inproj_w), *bin = PR(P.inproj_b);
    const float *Wiq = Win, *Wik = Win + (int64_t)D * D, *Wiv = Win + 2LL * D * D;
    const float *biq = bin, *biv = bin + 2 * D;
    // Wkk = Wik Wk, Wvv = Wiv Wv, bvv = Wiv bv + biv
    CK(lin.nn(Wik, D, D, D, PR(P.k_w), D, W("Wkk"), D));
    CK(lin.nn(Wiv, D, D, D, PR(P.v_w), D, W("Wvv"), D));
    CK(launch_smm(1, D, D, PR(P.v_b), D, 1, Wiv, 1, D, biv, W("bvv"), D, 0, 0, 1.f, st));

    // -- inputs
    CK(launch_gather_inputs(pk, mb, W("Xp"), W("U0"), W("curg"), st));
    // numerical encoder (state_encoder.py:35-57,187)
    {
        int prev = x.Fn;
        for (int i = 0; i < d.n_num; ++i) {
            CK(lin.nt(W("U" + std::to_string(i)), prev, B, prev, PR(P.num_w[i]), prev, PR(P.num_b[i]), d.num_hidden[i],
                      W("U" + std::to_string(i + 1)), d.num_hidden[i], 1, 1.f));
            prev = d.num_hidden[i];
        }
    }
    // node encoder on all nodes and on the current node (state_encoder.py:189-191)
    CK(launch_gemm_nt(W("Xp"), mb.M, 32, W("We_pad"), D, PR(P.node_b), nullptr, W("H0"), 0, st, prof));
    CK(launch_smm(B, D, x.F, W("curg"), UPAMD_NODE_PAD, 1, PR(P.node_w), 1, x.F, PR(P.node_b), W("C"), D, 0, 0, 1.f, st));
    // GCN layers (state_encoder.py:194-197).  Layer 1 reads its P/Q straight from the raw node features:
    // PQ_1 = H_0 Wcat_1^T = Xp (Wcat_1 We)^T + Wcat_1 be  (K = 32 instead of D: saves one full-size node GEMM)
    CK(lin.nn(W("Wcat0"), D, 2 * D, D, W("We_pad"), 32, W("W1c"), 32));
    CK(launch_smm(1, 2 * D, D, PR(P.node_b), D, 1, W("Wcat0"), 1, D, nullptr, W("b1c"), 2 * D, 0, 0, 1.f, st));
    for (int l = 1; l <= x.L; ++l) {
        const std::string sl = std::to_string(l);
        if (l == 1)
            CK(launch_gemm_nt(W("Xp"), mb.M, 32, W("W1c"), 2 * D, W("b1c"), nullptr, W("PQ1"), 0, st, prof));
        else
            CK(launch_gemm_nt(W("H" + std::to_string(l - 1)), mb.M, D, W("Wcat" + std::to_string(l - 1)), 2 * D, nullptr, nullptr,
                              W("PQ" + sl), 0, st, prof));
        // the last layer also writes the land-use pointer-head inputs FE (needs C, computed above)
        CK(launch_edge_fwd(pk, mb, D, l == x.L, W("PQ" + sl), PR(P.edge_b[l - 1]), W("H" + std::to_string(l - 1)), W("H" + sl),
                           W("hbarV"), W("hbarE"), W("C"), (l == x.L && mb.Nhe > 0) ? W("FE") : nullptr, st, prof));
    }
    const float *HL = W("H" + std::to_string(x.L));
    // attention (state_encoder.py:150-161)
    const float scale = 1.0f / std::sqrt((float)x.dh);
    CK(lin.nt(W("C"), D, B, D, PR(P.q_w), D, PR(P.q_b), D, W("q0"), D, 0, 1.f));
    CK(lin.nt(W("q0"), D, B, D, Wiq, D, biq, D, W("q1"), D, 0, scale));
    for (int h = 0; h < x.heads; ++h)   // r[b,h,:] = q1[b, h-slice] @ Wkk[h-slice, :]
        CK(lin.nn(W("q1") + h * x.dh, D, B, x.dh, W("Wkk") + (int64_t)h * x.dh * D, D, W("r") + (int64_t)h * D,
                  (int64_t)x.heads * D));
    CK(launch_attn_fwd(pk, mb, D, x.heads, HL, W("r"), W("alpha"), W("s"), st));
    for (int h = 0; h < x.heads; ++h)   // o[b, h-slice] = s[b,h,:] @ Wvv[h-slice,:]^T + bvv[h-slice]
        CK(lin.nt(W("s") + (int64_t)h * D, (int64_t)x.heads * D, B, D, W("Wvv") + (int64_t)h * x.dh * D, D, W("bvv") + h * x.dh,
                  x.dh, W("o") + h * x.dh, D, 0, 1.f));
    CK(lin.nt(W("o"), D, B, D, PR(P.outproj_w), D, PR(P.outproj_b), D, W("att"), D, 0, 1.f));
    // value head (value.py:15-39)
    CK(launch_assemble_sv(pk, mb, D, x.S_last, W("U" + std::to_string(d.n_num)), W("hbarV"), W("hbarE"), W("att"), W("SV"), x.Wp, st));
    CK(launch_pad_cols(PR(P.value_w[0]), d.value_hidden[0], x.W, x.Wp, W("Vw0p"), st));
    {
        const float *prevp = W("SV");
        int prev = x.Wp;
        for (int i = 0; i < d.n_value; ++i) {
            float *out = (i == d.n_value - 1) ? value_dev : W("V" + std::to_string(i + 1));
            CK(lin.nt(prevp, prev, B, prev, i == 0 ? W("Vw0p") : PR(P.value_w[i]), prev, PR(P.value_b[i]), d.value_hidden[i], out,
                      d.value_hidden[i], i < d.n_value - 1, 1.f));
            prevp = out;
            prev = d.value_hidden[i];
        }
    }
    // pointer heads (policy.py:19-65)
    if (mb.Nhe > 0) {
        // factorised first Linear: hid = tanh(FE [Wa+Wd | Wc]^T + ((Wb-Wd) c_b + b1)), the bias rows are
        // pre-written into hid and accumulated in place
        CK(launch_prep_land_head(PR(P.land_w[0]), D, x.h0l, W("W1f"), W("Wbd"), st));
        CK(lin.nt(W("C"), D, B, D, W("Wbd"), D, PR(P.land_b0), x.h0l, W("constb"), x.h0l, 0, 1.f));
        CK(launch_he_bias_rows(pk, mb, x.h0l, W("constb"), W("hidl"), st));
        CK(launch_gemm_nt(W("FE"), mb.Nhe, 2 * D, W("W1f"), x.h0l, nullptr, W("hidl"), W("hidl"), 1, st, prof));
        CK(launch_rowdot_pm(W("hidl"), mb.Nhe, x.h0l, PR(P.land_w[1]), W("z_he"), st));
    }
    if (mb.Nrn > 0) {
        CK(launch_road_gather(pk, mb, D, HL, W("XR"), st));
        CK(launch_gemm_nt(W("XR"), mb.Nrn, D, PR(P.road_w[0]), x.h0r, PR(P.road_b0), nullptr, W("hidr"), 1, st, prof));
        CK(launch_rowdot_pm(W("hidr"), mb.Nrn, x.h0r, PR(P.road_w[1]), W("z_rn"), st));
    }
    CK(launch_pointer_fwd(pk, mb, W("z_he"), W("z_rn"), W("p_he"), W("p_rn"), logp_dev, ent_dev, W("lse"), st));
    // the entropy is needed again by the backward: keep a copy next to lse
    UPAMD_HIP(hipMemcpyAsync(W("entk"), ent_dev, sizeof(float) * (size_t)B, hipMemcpyDeviceToDevice, st));
    return UPAMD_OK;
}

// =============================================================================================
// backward
// =============================================================================================
extern "C" int upamd_backward(upamd_engine *eng, const void *packed_dev, const upamd_pack_layout *layout,
                              const upamd_minibatch *mbp, const float *prm, void *ws_dev, int64_t ws_bytes,
                              const float *dvalue_dev, const float *dlogp_dev, const float *dent_dev,
                              float *grads, void *stream) {
    Plan pl;
    CK(check_args(eng, packed_dev, layout, mbp, prm, ws_dev, ws_bytes, &pl));
    if (!dvalue_dev || !dlogp_dev || !dent_dev || !grads) return fail(UPAMD_E_INVALID, "null seed/grad pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const upamd_model_desc &d = eng->d;
    const ParamLayout &P = eng->P;
    const Dims x = dims_of(d);
    const int D = x.D, B = mbp->B;
    const PackedView pk = make_view(packed_dev, *layout);
    MbView mb = make_mb(*mbp);
    float *ws = static_cast<float *>(ws_dev);
    auto W = [&](const std::string &n) { return ws + pl.off.at(n); };
    mb.rows = reinterpret_cast<const int32_t *>(W("rows"));     // written by the forward of this minibatch
    auto PR = [&](int idx) { return prm + P.off(idx); };
    auto GR = [&](int idx) { return grads + P.off(idx); };
    Profiler *prof = &eng->prof;
    const Lin lin{st, prof, W("slabs"), W("wt")};
    const float *Win = PR(P.inproj_w);
    const float *Wiq = Win, *Wik = Win + (int64_t)D * D, *Wiv = Win + 2LL * D * D;
    float *gWin = GR(P.inproj_w), *gbin = GR(P.inproj_b);
    const float *HL = W("H" + std::to_string(x.L));

    UPAMD_HIP(hipMemsetAsync(W("dWkk"), 0, sizeof(float) * (size_t)(W("zero_end") - W("dWkk")), st));      // every accumulate-into scratch
    // ---- value head
    float *dzA = W("dzA"), *dzB = W("dzB");
    {
        const float *dz = dvalue_dev;     // dz of the last layer is the seed itself ([B,1])
        int64_t ldz = 1;
        for (int i = d.n_value - 1; i >= 0; --i) {
            const int N = d.value_hidden[i];
            const int K = (i == 0) ? x.Wp : d.value_hidden[i - 1];
            const float *Xin = (i == 0) ? W("SV") : W("V" + std::to_string(i));
            if (i < d.n_value - 1) CK(launch_tanh_bwd(const_cast<float *>(dz), W("V" + std::to_string(i + 1)), (int64_t)B * N, st));
            // first layer: SV rows are Wp wide (zero padded); only the W real columns of the gradient are kept
            CK(lin.tn_acc(dz, ldz, B, N, Xin, K, K, GR(P.value_w[i]), GR(P.value_b[i]), i == 0 ? x.W : K));
            float *dnext = (dz == dzA) ? dzB : dzA;
            CK(lin.nn(dz, ldz, B, N, i == 0 ? W("Vw0p") : PR(P.value_w[i]), K, dnext, K));
            dz = dnext;
            ldz = K;
        }
        if (dz != dzA) UPAMD_HIP(hipMemcpyAsync(dzA, dz, sizeof(float) * (size_t)B * x.Wp, hipMemcpyDeviceToDevice, st));
    }
    const float *dSV = dzA;                          // [B, W] with row stride Wp
    const float *dhbarV = dSV + x.S_last;            // column slices, ld = Wp
    const float *dhbarE = dSV + x.S_last + D;

    // ---- numerical encoder backward
    {
        float *bufA = W("dnA"), *bufB = W("dnB");
        // compact the dUlast column slice of dSV into a dense [B, S_last] buffer
        UPAMD_HIP(hipMemcpy2DAsync(bufA, sizeof(float) * x.S_last, dSV, sizeof(float) * x.Wp, sizeof(float) * x.S_last, B,
                                   hipMemcpyDeviceToDevice, st));
        float *dz = bufA;
        for (int i = d.n_num - 1; i >= 0; --i) {
            const int N = d.num_hidden[i];
            const int K = (i == 0) ? x.Fn : d.num_hidden[i - 1];
            CK(launch_tanh_bwd(dz, W("U" + std::to_string(i + 1)), (int64_t)B * N, st));
            CK(lin.tn_acc(dz, N, B, N, W("U" + std::to_string(i)), K, K, GR(P.num_w[i]), GR(P.num_b[i])));
            if (i > 0) {
                float *dnext = (dz == bufA) ? bufB : bufA;
                CK(lin.nn(dz, N, B, N, PR(P.num_w[i]), K, dnext, K));
                dz = dnext;
            }
        }
    }

    // ---- attention, dense part (datt compacted so the MFMA path sees aligned rows)
    UPAMD_HIP(hipMemcpy2DAsync(W("datt"), sizeof(float) * D, dSV + x.S_last + 2 * D, sizeof(float) * x.Wp, sizeof(float) * D, B,
                               hipMemcpyDeviceToDevice, st));
    const float *datt = W("datt");
    CK(lin.tn_acc(datt, D, B, D, W("o"), D, D, GR(P.outproj_w), GR(P.outproj_b)));
    CK(lin.nn(datt, D, B, D, PR(P.outproj_w), D, W("do"), D));
    CK(launch_colsum_rm(W("do"), B, D, D, W("dbvv"), st));
    for (int h = 0; h < x.heads; ++h) {
        // dWvv[h-slice,:] += do[:,h-slice]^T s[:,h,:]
        CK(lin.tn_acc(W("do") + h * x.dh, D, B, x.dh, W("s") + (int64_t)h * D, (int64_t)x.heads * D, D,
                      W("dWvv") + (int64_t)h * x.dh * D, nullptr));
        // ds[:,h,:] = do[:,h-slice] Wvv[h-slice,:]
        CK(lin.nn(W("do") + h * x.dh, D, B, x.dh, W("Wvv") + (int64_t)h * x.dh * D, D, W("ds") + (int64_t)h * D,
                  (int64_t)x.heads * D));
    }
    // ---- attention core: writes G^L (mean + attention terms) and dr
    float *G = W("G0"), *Gn = W("G1");
    CK(launch_attn_bwd(pk, mb, D, x.heads, HL, W("r"), W("alpha"), W("s"), W("ds"), dhbarV, x.Wp, G, W("dr"), st));
    for (int h = 0; h < x.heads; ++h) {
        // dq1[:,h-slice] = dr[:,h,:] Wkk[h-slice,:]^T
        CK(lin.nt(W("dr") + (int64_t)h * D, (int64_t)x.heads * D, B, D, W("Wkk") + (int64_t)h * x.dh * D, D, nullptr, x.dh,
                  W("dq1") + h * x.dh, D, 0, 1.f));
        // dWkk[h-slice,:] += q1[:,h-slice]^T dr[:,h,:]
        CK(lin.tn_acc(W("q1") + h * x.dh, D, B, x.dh, W("dr") + (int64_t)h * D, (int64_t)x.heads * D, D,
                      W("dWkk") + (int64_t)h * x.dh * D, nullptr));
    }
    const float scale = 1.0f / std::sqrt((float)x.dh);
    CK(launch_scale(W("dq1"), (int64_t)B * D, scale, st));                                  // dpre = dq1 * scale
    CK(lin.tn_acc(W("dq1"), D, B, D, W("q0"), D, D, gWin, gbin));                            // in_proj, q rows
    CK(lin.nn(W("dq1"), D, B, D, Wiq, D, W("dq0"), D));
    CK(lin.tn_acc(W("dq0"), D, B, D, W("C"), D, D, GR(P.q_w), GR(P.q_b)));
    CK(lin.nn(W("dq0"), D, B, D, PR(P.q_w), D, W("dC"), D));
    // collapsed products: Wkk = Wik Wk ; Wvv = Wiv Wv ; bvv = Wiv bv + biv
    CK(lin.nt_acc(W("dWkk"), D, D, D, PR(P.k_w), D, D, gWin + (int64_t)D * D, D));                                   // dWik += dWkk Wk^T
    CK(lin.tn_acc(Wik, D, D, D, W("dWkk"), D, D, GR(P.k_w), nullptr));                                               // dWk  += Wik^T dWkk
    CK(lin.nt_acc(W("dWvv"), D, D, D, PR(P.v_w), D, D, gWin + 2LL * D * D, D));                                      // dWiv += dWvv Wv^T
    CK(launch_smm(D, D, 1, W("dbvv"), 1, 1, PR(P.v_b), 1, 1, nullptr, gWin + 2LL * D * D, D, 1, 0, 1.f, st));         // dWiv += dbvv (x) bv
    CK(lin.tn_acc(Wiv, D, D, D, W("dWvv"), D, D, GR(P.v_w), nullptr));                                               // dWv  += Wiv^T dWvv
    CK(launch_smm(1, D, D, W("dbvv"), D, 1, Wiv, D, 1, nullptr, GR(P.v_b), D, 1, 0, 1.f, st));                         // dbv  += Wiv^T dbvv
    CK(launch_axpy(gbin + 2 * D, W("dbvv"), D, 1.f, st));                                                            // dbiv += dbvv

    // ---- pointer heads
    CK(launch_pointer_bwd(pk, mb, W("z_he"), W("z_rn"), W("p_he"), W("p_rn"), W("entk"), W("lse"), dlogp_dev, dent_dev,
                          W("dz_he"), W("dz_rn"), st));
    int S = 1;
    if (mb.Nhe > 0) {
        CK(launch_colsum_pm(W("hidl"), mb.Nhe, x.h0l, W("dz_he"), W("cs_part"), GR(P.land_w[1]), st));          // dw2 += sum dz * hid
        CK(launch_rowdot_bwd_pm(W("hidl"), mb.Nhe, x.h0l, PR(P.land_w[1]), W("dz_he"), W("dprel"), st));
        CK(launch_colsum_pm(W("dprel"), mb.Nhe, x.h0l, nullptr, W("cs_part"), GR(P.land_b0), st));
        // dW1f = dpre^T FE, mapped back onto [Wa|Wb|Wc|Wd] together with dWbd = dconst^T C
        CK(launch_gemm_tn(W("FE"), 2 * D, W("dprel"), x.h0l, mb.Nhe, W("slabs"), &S, st, prof));
        CK(launch_reduce_slabs(W("slabs"), S, 2 * D, x.h0l, 1, x.h0l, W("dW1f"), 2 * D, st));
        CK(launch_he_segsum(pk, mb, x.h0l, W("dprel"), W("dconst"), st));
        CK(lin.tn_acc(W("dconst"), x.h0l, B, x.h0l, W("C"), D, D, W("dWbd"), nullptr));
        CK(launch_land_head_w_scatter(W("dW1f"), W("dWbd"), D, x.h0l, GR(P.land_w[0]), st));
        // dC from the bias term, then dFE = dpre W1f and the feature backward
        CK(lin.nn(W("dconst"), x.h0l, B, x.h0l, W("Wbd"), D, W("dC_head"), D));
        CK(launch_axpy(W("dC"), W("dC_head"), (int64_t)B * D, 1.f, st));
        CK(launch_transpose(W("W1f"), x.h0l, 2 * D, W("W1fT"), st));
        CK(launch_gemm_nt(W("dprel"), mb.Nhe, x.h0l, W("W1fT"), 2 * D, nullptr, nullptr, W("dFE"), 0, st, prof));
        CK(launch_he_feat_bwd(pk, mb, D, W("FE"), W("C"), W("dFE"), W("dMhe"), W("dC_head"), st));
        CK(launch_axpy(W("dC"), W("dC_head"), (int64_t)B * D, 1.f, st));
    }
    if (mb.Nrn > 0) {
        CK(launch_colsum_pm(W("hidr"), mb.Nrn, x.h0r, W("dz_rn"), W("cs_part"), GR(P.road_w[1]), st));
        CK(launch_rowdot_bwd_pm(W("hidr"), mb.Nrn, x.h0r, PR(P.road_w[1]), W("dz_rn"), W("dprer"), st));
        CK(launch_colsum_pm(W("dprer"), mb.Nrn, x.h0r, nullptr, W("cs_part"), GR(P.road_b0), st));
        CK(launch_gemm_tn(W("XR"), D, W("dprer"), x.h0r, mb.Nrn, W("slabs"), &S, st, prof));
        CK(launch_reduce_slabs(W("slabs"), S, D, x.h0r, 1, x.h0r, GR(P.road_w[0]), D, st));
        CK(launch_transpose(PR(P.road_w[0]), x.h0r, D, W("R1T"), st));
        CK(launch_gemm_nt(W("dprer"), mb.Nrn, x.h0r, W("R1T"), D, nullptr, nullptr, W("dXR"), 0, st, prof));
        CK(launch_road_scatter_add(pk, mb, D, W("dXR"), G, st));
    }

    // ---- GCN layers, last to first
    for (int l = x.L; l >= 1; --l) {
        const std::string sl = std::to_string(l), sp = std::to_string(l - 1);
        const bool last = (l == x.L);
        CK(launch_edge_bwd(pk, mb, D, last, W("PQ" + sl), PR(P.edge_b[l - 1]), G, dhbarE, x.Wp,
                           (last && mb.Nhe > 0) ? W("dMhe") : nullptr, W("dPQ"), W("dbias_part"), st, prof));
        // column sums of dP | dQ over the minibatch (P/Q panel order); the layer's bias gradient is the P half
        CK(launch_reduce_rows_add(W("dbias_part"), B, 2 * D, W("cs" + sl), st));
        CK(launch_add_p_panels(GR(P.edge_b[l - 1]), W("cs" + sl), D, st));
        if (l > 1) {
            CK(launch_gemm_tn(W("dPQ"), 2 * D, W("H" + sp), D, mb.M, W("slabs"), &S, st, prof));
            CK(launch_reduce_slabs(W("slabs"), S, 2 * D, D, 2, D, GR(P.edge_w[l - 1]), 2 * D, st));
            CK(launch_gemm_nt(W("dPQ"), mb.M, 2 * D, W("WcatT" + sp), D, nullptr, G, Gn, 0, st, prof));
            std::swap(G, Gn);
        } else {
            // layer 1: H_0 = Xp We^T + be, so dWcat_1 = dPQ_1^T H_0 = (dPQ_1^T Xp) We^T + colsum(dPQ_1) (x) be --
            // two J = 32 reductions over the nodes instead of a full-size weight-gradient GEMM
            CK(launch_gemm_tn(W("dPQ"), 2 * D, W("Xp"), 32, mb.M, W("slabs"), &S, st, prof));
            CK(launch_reduce_slabs(W("slabs"), S, 2 * D, 32, 0, 32, W("Tn"), 32, st));
            CK(lin.nt(W("Tn"), 32, 2 * D, 32, W("We_pad"), 32, nullptr, D, W("dWc1"), D, 0, 1.f));       // Tn We^T
            CK(launch_smm(2 * D, D, 1, W("cs1"), 1, 1, PR(P.node_b), 1, 1, nullptr, W("dWc1"), D, 1, 0, 1.f, st));
            CK(launch_reduce_slabs(W("dWc1"), 1, 2 * D, D, 2, D, GR(P.edge_w[0]), 2 * D, st));
        }
    }
    // ---- node encoder.  G^0 = G^1 + dPQ_1 Wcat_1 is never formed (it is only needed for the encoder's own
    // gradients): dWe = G^0^T X = G^1^T X + Wcat_1^T (dPQ_1^T X),  dbe = colsum(G^1) + Wcat_1^T colsum(dPQ_1).
    // G holds G^1 and "dPQ" holds dPQ_1 here; this replaces a full-size dgrad GEMM by two K = M, J = 32 ones.
    CK(launch_gemm_tn(G, D, W("Xp"), 32, mb.M, W("slabs"), &S, st, prof));
    // Xp's column 31 is all ones (gather_inputs), so column 31 of G^1^T Xp is colsum(G^1): straight into dbe
    CK(launch_reduce_slabs(W("slabs"), S, D, 32, 0, x.F, GR(P.node_w), x.F, st, GR(P.node_b)));
    // "Tn" = dPQ_1^T Xp and "cs1" = colsum(dPQ_1) come from the l = 1 iteration of the loop above
    // (split-K: 4 workgroups looping over K = 2D would be pure latency)
    CK(launch_smm_splitk(D, x.F, 2 * D, W("WcatT0"), 2 * D, 1, W("Tn"), 32, 1, W("slabs"), &S, st));
    CK(launch_reduce_slabs(W("slabs"), S, D, x.F, 0, x.F, GR(P.node_w), x.F, st));
    CK(launch_smm_splitk(1, D, 2 * D, W("cs1"), 2 * D, 1, W("WcatT0"), 1, 2 * D, W("slabs"), &S, st));
    CK(launch_reduce_slabs(W("slabs"), S, 1, D, 0, D, GR(P.node_b), D, st));
    CK(lin.tn_acc(W("dC"), D, B, D, W("curg"), UPAMD_NODE_PAD, x.F, GR(P.node_w), GR(P.node_b)));
    return UPAMD_OK;
}
