// TEST INFRASTRUCTURE (CPU): runs the per-graph program of the fused small-model path (csrc/tiny_body.h -- the very text the HIP
// kernel is compiled from) on the host, one "workgroup" per graph with its ranges executed sequentially, so that the math and
// the hand-derived backward can be checked against the oracle without a GPU (tests/test_tiny_emul.py).  Not part of the product.
#define TINY_HOST 1
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../drl-urban-planning_amd/csrc/tiny_body.h"
#include "../../include/upamd.h"

using namespace upamd_tiny;

extern "C" {

int tiny_emul_sizeof_dims() { return (int)sizeof(Dims); }
int tiny_emul_sizeof_offs() { return (int)sizeof(Offs); }
// LDS bytes of the plan for a launch whose largest graph has (n nodes, inc incidences, cand candidates), before the chunk buffers
// take what is left; and the budget of one workgroup
long long tiny_emul_plan_base_bytes(const Dims *d, int n, int inc, int cand) { return plan_layout(*d, n, inc, cand, 0).total * 4; }
long long tiny_emul_lds_budget_bytes() { return LDS_FLOATS * 4; }

// grads (out, [n_floats]): sum of the per-workgroup slabs in workgroup order;  losses (out, [4]): STEP mode only
int tiny_emul_run(const void *packed, const upamd_pack_layout *L, int B, const int32_t *idx, const int32_t *he_off,
                  const int32_t *rn_off, int max_n, int max_inc, int max_cand, const Dims *dims, const Offs *offs, const float *prm, int mode,
                  int groups, float *value, float *logp, float *ent, float *z_he, float *z_rn, const float *dvalue,
                  const float *dlogp, const float *dent, const int64_t *rows, const float *adv, const float *ret,
                  const float *old_logp, const float *exps, float clip_eps, float cv, float ce, float inv_rows, float inv_ind,
                  float *grads, float *losses, int64_t *lds_bytes_out) {
    const char *base = static_cast<const char *>(packed);
    Args A;
    memset(&A, 0, sizeof(A));
    A.meta = reinterpret_cast<const int32_t *>(base + L->off_meta);
    A.X = reinterpret_cast<const float *>(base + L->off_x);
    A.nmask = reinterpret_cast<const uint8_t *>(base + L->off_nmask);
    A.rowptr = reinterpret_cast<const int32_t *>(base + L->off_rowptr);
    A.inc_nbr = reinterpret_cast<const uint16_t *>(base + L->off_inc_nbr);
    A.he_src = reinterpret_cast<const uint16_t *>(base + L->off_he_src);
    A.he_dst = reinterpret_cast<const uint16_t *>(base + L->off_he_dst);
    A.he_live = reinterpret_cast<const uint8_t *>(base + L->off_he_live);
    A.rn_node = reinterpret_cast<const uint16_t *>(base + L->off_rn_node);
    A.hinc_ptr = reinterpret_cast<const int32_t *>(base + L->off_hinc_ptr);
    A.hinc_nbr = reinterpret_cast<const uint16_t *>(base + L->off_hinc_nbr);
    A.hinc_he = reinterpret_cast<const uint16_t *>(base + L->off_hinc_he);
    A.order = reinterpret_cast<const uint16_t *>(base + L->off_order);
    A.numerical = reinterpret_cast<const float *>(base + L->off_numerical);
    A.cur = reinterpret_cast<const float *>(base + L->off_cur);
    A.B = B; A.idx = idx; A.he_off = he_off; A.rn_off = rn_off;
    A.d = *dims; A.o = *offs; A.prm = prm; A.mode = mode;
    A.value = value; A.logp = logp; A.ent = ent; A.z_he = z_he; A.z_rn = z_rn;
    A.dvalue = dvalue; A.dlogp = dlogp; A.dent = dent;
    A.rows = rows; A.adv = adv; A.ret = ret; A.old_logp = old_logp; A.exps = exps;
    A.clip_eps = clip_eps; A.cv = cv; A.ce = ce; A.inv_rows = inv_rows; A.inv_ind = inv_ind;
    A.max_n = max_n; A.max_inc = max_inc; A.max_cand = max_cand;
    const Plan pl = make_plan(A.d, max_n, max_inc, max_cand);
    if (lds_bytes_out) *lds_bytes_out = pl.total * 4;
    const int G = groups < B ? groups : B;
    const int64_t P = offs->n_floats;
    std::vector<float> slabs((size_t)G * P, 0.f), scratch((size_t)G * ((size_t)max_cand + 1) * dims->D, 0.f), loss_rows((size_t)B * 4, 0.f);
    std::vector<float> lds((size_t)pl.total + 64);
    A.slab = slabs.data(); A.slab_stride = P;
    A.scratch = scratch.data(); A.scratch_stride = ((int64_t)max_cand + 1) * dims->D;
    A.loss_rows = loss_rows.data();
    for (int wg = 0; wg < G; ++wg) {
        for (int b = wg; b < B; b += G) {
            // poison the LDS image so that a read of something this graph did not write shows up as a NaN
            for (auto &x : lds) x = __builtin_nanf("");
            if (dims->D == 16) graph_program<16>(A, b, A.slab + wg * A.slab_stride, A.scratch + wg * A.scratch_stride, lds.data(), pl);
            else if (dims->D == 32) graph_program<32>(A, b, A.slab + wg * A.slab_stride, A.scratch + wg * A.scratch_stride, lds.data(), pl);
            else return -1;
        }
    }
    if (mode != FWD && grads) {
        for (int64_t i = 0; i < P; ++i) {
            float acc = 0.f;
            for (int wg = 0; wg < G; ++wg) acc += slabs[(size_t)wg * P + i];
            grads[i] = acc;
        }
    }
    if (mode == STEP && losses) {
        float tv = 0.f, ts = 0.f, te = 0.f;
        for (int b = 0; b < B; ++b) { tv += loss_rows[(size_t)b * 4]; ts += loss_rows[(size_t)b * 4 + 1]; te += loss_rows[(size_t)b * 4 + 2]; }
        const float vl = tv * inv_rows, sl = -ts * inv_ind, el = -te * inv_ind;
        losses[0] = sl + cv * vl + ce * el; losses[1] = vl; losses[2] = sl; losses[3] = el;
    }
    return 0;
}
}
