// Fused small-model path (gcn_node_dim <= 32): the per-graph program of tiny_body.h as a HIP kernel -- one 1024-thread
// workgroup per graph, persistent over the minibatch, everything of a graph in LDS -- plus the fixed-order reduction of the
// per-workgroup gradient slabs.  An optimizer step at the reference's shipped dims (hlg.yaml:21-33: D = 16, L = 2) is THREE
// launches: tiny_kernel<STEP> (forward + loss seeds + backward, ~0.21 ms for 256 rows), tiny_reduce (slabs -> gradients, loss
// scalars, ~7 us), Adam (5 us).
// Replaces, for such models, the ~35 dependent launches of the general path (engine.hip), which stays the path of every
// other model and of graphs too large for one workgroup's LDS.
#include <cstring>

#include "kernels.h"
#include "tiny_body.h"

namespace upamd {

using namespace upamd_tiny;

// threads per workgroup: tune knob "tiny_threads".  1024 (default): 4 waves per SIMD at <= 128 VGPRs each (the program needs ~100
// since the per-phase thread id, see t_tid() in tiny_body.h); the phases are latency-bound and 4 waves hide more of it than the 2
// of the 512-thread variant: ~12 % faster at the reference dims (profiles/archive/r04_lab_tiny_sections.log).
static int g_tiny_threads = 1024;
void set_tiny_threads(int n) { g_tiny_threads = n == 512 ? 512 : 1024; }
constexpr int64_t TINY_LDS_LIMIT = 160 * 1024 - 512;

static long long *g_tiny_prof = nullptr;      // lab hook (upamd_tiny_profile): section time stamps of one graph
void set_tiny_prof(void *buf) { g_tiny_prof = static_cast<long long *>(buf); }
static int g_tiny_fused = 1;       // tune knob "tiny_fused" (default on)
void set_tiny_fused(int on) { g_tiny_fused = on ? 1 : 0; }

template <int D, int NT>
__global__ __launch_bounds__(NT) void tiny_kernel(Args A, Plan pl) {
    extern __shared__ __attribute__((aligned(16))) float tiny_lds[];
    float *slab = A.slab ? A.slab + (int64_t)blockIdx.x * A.slab_stride : nullptr;
    float *gscr = A.scratch ? A.scratch + (int64_t)blockIdx.x * A.scratch_stride : nullptr;
    if (A.mode != FWD) {
        // (16-byte stores: slab_stride is a multiple of 4 floats and the buffer 256-byte aligned)
        f4 *s4 = reinterpret_cast<f4 *>(slab);
        T_FOR(i, (int)(A.slab_stride / 4)) s4[i] = f4{0.0f, 0.0f, 0.0f, 0.0f};
        T_SYNC();
    }
    for (int b = blockIdx.x; b < A.B; b += gridDim.x) graph_program<D>(A, b, slab, gscr, tiny_lds, pl);
}

// grads[i] (+)= sum over the G slabs, in slab order;  the last block forms the four loss scalars from the per-row terms
__global__ __launch_bounds__(256) void tiny_reduce_kernel(const float *__restrict__ slab, int64_t stride, int G, int64_t P,
                                                          float *__restrict__ grads, int accumulate,
                                                          const float *__restrict__ loss_rows, int B, float inv_rows, float inv_ind,
                                                          float cv, float ce, float *__restrict__ losses) {
    if (blockIdx.x == gridDim.x - 1 && losses) {
        __shared__ float red[3][256];
        float tv = 0.f, ts = 0.f, te = 0.f;
        for (int b = threadIdx.x; b < B; b += 256) {
            tv += loss_rows[(int64_t)b * 4 + 0];
            ts += loss_rows[(int64_t)b * 4 + 1];
            te += loss_rows[(int64_t)b * 4 + 2];
        }
        red[0][threadIdx.x] = tv; red[1][threadIdx.x] = ts; red[2][threadIdx.x] = te;
        __syncthreads();
        if (threadIdx.x == 0) {
            float a = 0.f, b2 = 0.f, c = 0.f;
            for (int q = 0; q < 256; ++q) { a += red[0][q]; b2 += red[1][q]; c += red[2][q]; }
            const float vl = a * inv_rows, sl = -b2 * inv_ind, el = -c * inv_ind;
            losses[0] = sl + cv * vl + ce * el;
            losses[1] = vl;
            losses[2] = sl;
            losses[3] = el;
        }
        return;
    }
    // 64 parameters x 4 slab groups per block: a thread sums a contiguous quarter of the slabs (eight loads in flight, slab order),
    // the four partial sums are combined as (q0 + q1) + (q2 + q3).  (One thread per parameter over all 256 slabs: 54 blocks on 256
    // CUs and 32 dependent batches each -- 13 us for 14 MB.)
    __shared__ float part[4][64];
    const int p = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + p;
    const int Gq = (G + 3) / 4, g0 = sg * Gq, g1 = g0 + Gq < G ? g0 + Gq : G;
    float acc = 0.f;
    if (i < P) {
        int g = g0;
        for (; g + 8 <= g1; g += 8) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = slab[(int64_t)(g + q) * stride + i];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc += v[q];
        }
        for (; g < g1; ++g) acc += slab[(int64_t)g * stride + i];
    }
    part[sg][p] = acc;
    __syncthreads();
    if (sg == 0 && i < P) {
        const float tot = (part[0][p] + part[1][p]) + (part[2][p] + part[3][p]);
        grads[i] = accumulate ? grads[i] + tot : tot;
    }
}

static void tiny_fill(const upamd_model_desc &d, const ParamLayout &P, Dims *x, Offs *o) {
    std::memset(x, 0, sizeof(*x));
    std::memset(o, 0, sizeof(*o));
    x->D = d.D; x->L = d.L; x->heads = d.heads; x->F = d.node_dim; x->Fn = d.numerical_dim;
    x->n_num = d.n_num; x->n_value = d.n_value;
    for (int i = 0; i < d.n_num; ++i) x->num_hidden[i] = d.num_hidden[i];
    for (int i = 0; i < d.n_value; ++i) x->value_hidden[i] = d.value_hidden[i];
    x->h0l = d.land_hidden[0]; x->h0r = d.road_hidden[0];
    x->S_last = d.num_hidden[d.n_num - 1];
    x->W = 3 * d.D + x->S_last + 3;
    for (int i = 0; i < d.n_num; ++i) { o->num_w[i] = (int)P.off(P.num_w[i]); o->num_b[i] = (int)P.off(P.num_b[i]); }
    o->node_w = (int)P.off(P.node_w); o->node_b = (int)P.off(P.node_b);
    for (int l = 0; l < d.L && l < MAXL; ++l) { o->edge_w[l] = (int)P.off(P.edge_w[l]); o->edge_b[l] = (int)P.off(P.edge_b[l]); }
    o->inproj_w = (int)P.off(P.inproj_w); o->inproj_b = (int)P.off(P.inproj_b);
    o->outproj_w = (int)P.off(P.outproj_w); o->outproj_b = (int)P.off(P.outproj_b);
    o->q_w = (int)P.off(P.q_w); o->q_b = (int)P.off(P.q_b); o->k_w = (int)P.off(P.k_w); o->k_b = (int)P.off(P.k_b);
    o->v_w = (int)P.off(P.v_w); o->v_b = (int)P.off(P.v_b);
    for (int i = 0; i < d.n_value; ++i) { o->value_w[i] = (int)P.off(P.value_w[i]); o->value_b[i] = (int)P.off(P.value_b[i]); }
    o->land_w0 = (int)P.off(P.land_w[0]); o->land_b0 = (int)P.off(P.land_b0); o->land_w1 = (int)P.off(P.land_w[1]);
    o->road_w0 = (int)P.off(P.road_w[0]); o->road_b0 = (int)P.off(P.road_b0); o->road_w1 = (int)P.off(P.road_w[1]);
    o->n_floats = (int)P.n_floats;
}

// The fused path covers: the SGNN encoder with single-Linear edge MLPs, D = 16 | 32, at least two GCN layers (the backward
// recomputes H^0 in H^1's place), two-layer pointer heads, hidden widths <= 64, and minibatches whose LARGEST graph fits one
// workgroup's LDS next to everything else (D = 16: up to ~400 nodes / ~2300 edges, the DHM community included).
bool tiny_supported(const upamd_model_desc &d, int max_n, int max_inc, int max_cand) {
    if (!g_tiny_fused || max_cand <= 0) return false;
    if (d.encoder != UPAMD_ENCODER_SGNN || edge_fc_layers(d) != 1) return false;
    if (!(d.D == 16 || d.D == 32) || d.L < 2 || d.L > MAXL || d.node_dim > XPAD) return false;
    if (d.n_land != 2 || d.n_road != 2 || d.land_hidden[0] > 64 || d.road_hidden[0] > 64) return false;
    for (int i = 0; i < d.n_num; ++i) if (d.num_hidden[i] > 64) return false;
    for (int i = 0; i < d.n_value; ++i) if (d.value_hidden[i] > 64) return false;
    if (d.numerical_dim > 256) return false;
    Dims x;
    Offs o;
    ParamLayout P;
    if (build_param_layout(&d, &P)) return false;
    tiny_fill(d, P, &x, &o);
    return make_plan(x, max_n > 0 ? max_n : 1, max_inc, max_cand).total * 4 <= TINY_LDS_LIMIT;
}

int tiny_groups(int B) { return B < 256 ? B : 256; }
int64_t tiny_scratch_stride(const upamd_model_desc &d, int max_cand) { return align_up(((int64_t)max_cand + 1) * d.D, 64); }
int64_t tiny_slab_stride(const ParamLayout &P) { return align_up(P.n_floats, 64); }

int launch_tiny(const upamd_model_desc &d, const ParamLayout &P, const PackedView &pk, const MbView &mb, const float *prm,
                const TinyIO &io, hipStream_t st) {
    Args A;
    std::memset(&A, 0, sizeof(A));
    A.meta = pk.meta; A.X = pk.X; A.nmask = pk.nmask; A.rowptr = pk.rowptr; A.inc_nbr = pk.inc_nbr;
    A.he_src = pk.he_src; A.he_dst = pk.he_dst; A.rn_node = pk.rn_node; A.hinc_nbr = pk.hinc_nbr; A.hinc_he = pk.hinc_he; A.order = pk.order;
    A.hinc_ptr = pk.hinc_ptr; A.he_live = pk.he_live; A.numerical = pk.numerical; A.cur = pk.cur;
    A.B = mb.B; A.idx = mb.idx; A.he_off = mb.he_off; A.rn_off = mb.rn_off;
    tiny_fill(d, P, &A.d, &A.o);
    A.prm = prm; A.mode = io.mode;
    A.value = io.value; A.logp = io.logp; A.ent = io.ent; A.z_he = io.z_he; A.z_rn = io.z_rn;
    A.dvalue = io.dvalue; A.dlogp = io.dlogp; A.dent = io.dent;
    A.rows = io.rows; A.adv = io.adv; A.ret = io.ret; A.old_logp = io.old_logp; A.exps = io.exps;
    A.clip_eps = io.clip_eps; A.cv = io.cv; A.ce = io.ce; A.inv_rows = io.inv_rows; A.inv_ind = io.inv_ind;
    A.loss_rows = io.loss_rows;
    A.slab = io.slab; A.slab_stride = tiny_slab_stride(P);
    A.scratch = io.scratch; A.scratch_stride = tiny_scratch_stride(d, mb.max_cand);
    A.max_n = mb.max_n; A.max_inc = mb.max_inc; A.max_cand = mb.max_cand;
    A.prof = g_tiny_prof;
    const Plan pl = make_plan(A.d, mb.max_n > 0 ? mb.max_n : 1, mb.max_inc, mb.max_cand);
    const int64_t lds = pl.total * 4;
    if (lds > TINY_LDS_LIMIT) return fail(UPAMD_E_LIMIT, "fused small-model path: a graph of %d nodes / %d incidences needs %lld bytes of LDS", mb.max_n, mb.max_inc, (long long)lds);
    if (io.mode != FWD && (!io.slab || !io.scratch)) return fail(UPAMD_E_INVALID, "fused small-model path: no slab / scratch workspace");
    const int G = tiny_groups(mb.B);
#define UPAMD_TINY(D_, NT_)                                                                                        \
    do {                                                                                                           \
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void *>(&tiny_kernel<D_, NT_>), lds)) return rc;    \
        hipLaunchKernelGGL((tiny_kernel<D_, NT_>), dim3(G), dim3(NT_), (size_t)lds, st, A, pl);                     \
    } while (0)
    if (d.D == 16) { if (g_tiny_threads == 512) UPAMD_TINY(16, 512); else UPAMD_TINY(16, 1024); }
    else { if (g_tiny_threads == 512) UPAMD_TINY(32, 512); else UPAMD_TINY(32, 1024); }
#undef UPAMD_TINY
    UPAMD_HIP(hipGetLastError());
    if (io.mode != FWD) {
        const int64_t Pn = P.n_floats;
        const unsigned blocks = (unsigned)((Pn + 63) / 64) + 1;
        hipLaunchKernelGGL(tiny_reduce_kernel, dim3(blocks), dim3(256), 0, st, io.slab, A.slab_stride, G, Pn, io.grads,
                           io.accumulate, io.loss_rows, mb.B, io.inv_rows, io.inv_ind, io.cv, io.ce,
                           io.mode == STEP ? io.losses : nullptr);
        UPAMD_HIP(hipGetLastError());
    }
    return 0;
}

}  // namespace upamd
