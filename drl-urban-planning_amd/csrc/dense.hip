// The PPO minibatch math (gfx950): loss + per-row seeds (with the minibatch gathers and zero_grad), GAE, Adam,
// first-step gradient clipping.
#include "kernels.h"

namespace upamd {

// ------------------------------------------------------------------------------------------
// PPO loss + seeds (urban_planning_agent.py:326-333,363-371; agent_pg.py:19-23).  Single
// workgroup, fixed-order tree reductions.  torch.min ties / clamp edges follow autograd:
// inside the clip range both branches are equal and the gradient is A*ratio; outside it the
// gradient flows only if the unclipped branch is the smaller one.
// ------------------------------------------------------------------------------------------
// `rows` (optional): adv / ret / old_logp / exps are whole-replay arrays and minibatch row b is replay row rows[b]
// (saves the caller four gather launches per step).  Blocks 1.. zero `zero[0..nzero)` (the gradient buffer of the step).
__global__ __launch_bounds__(1024) void ppo_loss_kernel(int B, const float *__restrict__ value,
                                                        const float *__restrict__ logp, const float *__restrict__ ent,
                                                        const int64_t *__restrict__ rows,
                                                        const float *__restrict__ adv, const float *__restrict__ ret,
                                                        const float *__restrict__ old_logp, const float *__restrict__ exps,
                                                        float clip_eps, float cv, float ce, float inv_rows, float inv_ind,
                                                        float *__restrict__ dvalue, float *__restrict__ dlogp,
                                                        float *__restrict__ dent, float *__restrict__ losses,
                                                        float *__restrict__ zero, int64_t nzero) {
    if (blockIdx.x > 0) {
        const int64_t stride = (int64_t)(gridDim.x - 1) * 1024;
        for (int64_t i = (int64_t)(blockIdx.x - 1) * 1024 + threadIdx.x; i < nzero; i += stride) zero[i] = 0.f;
        return;
    }
    __shared__ float red[3][16];
    float sv = 0.f, ss = 0.f, se = 0.f;
    const float lo = 1.f - clip_eps, hi = 1.f + clip_eps;
    for (int b = threadIdx.x; b < B; b += 1024) {
        const int64_t t = rows ? rows[b] : b;
        const float diff = value[b] - ret[t];
        sv = fmaf(diff, diff, sv);
        dvalue[b] = cv * 2.f * diff * inv_rows;
        float gl = 0.f, ge = 0.f;
        if (exps[t] != 0.f) {
            const float ratio = expf(logp[b] - old_logp[t]);
            const float A = adv[t];
            const float s1 = ratio * A;
            const float s2 = fminf(fmaxf(ratio, lo), hi) * A;
            ss += fminf(s1, s2);
            se += ent[b];
            const bool inside = ratio >= lo && ratio <= hi;
            const float dsdr = inside ? A : (s1 < s2 ? A : 0.f);
            gl = -dsdr * ratio * inv_ind;
            ge = -ce * inv_ind;
        }
        dlogp[b] = gl;
        dent[b] = ge;
    }
    for (int off = 32; off > 0; off >>= 1) {
        sv += __shfl_xor(sv, off);
        ss += __shfl_xor(ss, off);
        se += __shfl_xor(se, off);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = sv; red[1][w] = ss; red[2][w] = se; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float tv = 0.f, ts = 0.f, te = 0.f;
        for (int q = 0; q < 16; ++q) { tv += red[0][q]; ts += red[1][q]; te += red[2][q]; }
        const float vl = tv * inv_rows, sl = -ts * inv_ind, el = -te * inv_ind;
        losses[0] = sl + cv * vl + ce * el;
        losses[1] = vl;
        losses[2] = sl;
        losses[3] = el;
    }
}
int launch_ppo_loss(int B, const float *value, const float *logp, const float *ent, const int64_t *rows, const float *adv,
                    const float *ret, const float *old_logp, const float *exps, float clip_eps, float cv, float ce,
                    float inv_rows, float inv_ind, float *dvalue, float *dlogp, float *dent, float *losses,
                    float *zero, int64_t nzero, hipStream_t st) {
    const unsigned zb = (zero && nzero > 0) ? (unsigned)std::min<int64_t>((nzero + 4095) / 4096, 512) : 0u;
    hipLaunchKernelGGL(ppo_loss_kernel, dim3(1 + zb), dim3(1024), 0, st, B, value, logp, ent, rows, adv, ret, old_logp, exps, clip_eps, cv, ce, inv_rows, inv_ind, dvalue, dlogp, dent, losses, zero, nzero);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// GAE (khrylib/rl/core/common.py:12-21).  Episodes are independent once masks[t] == 0 (all
// carried terms are multiplied by the mask), so every thread that sits on an episode end walks
// its episode backwards with the reference's exact operation order (no FMA contraction).
// ------------------------------------------------------------------------------------------
__global__ void gae_kernel(int64_t T, const float *__restrict__ rewards, const float *__restrict__ masks,
                           const float *__restrict__ values, float gamma, float gt, float *__restrict__ adv,
                           float *__restrict__ ret) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    // segment ends: the last row, or a row whose successor-coupling is cut (masks[t] == 0)
    if (!(t == T - 1 || masks[t] == 0.f)) return;
    float prev_value = 0.f, prev_adv = 0.f;
    for (int64_t i = t; i >= 0; --i) {
        if (i != t && masks[i] == 0.f) break;
        const float m = masks[i], v = values[i];
        const float delta = __fsub_rn(__fadd_rn(rewards[i], __fmul_rn(__fmul_rn(gamma, prev_value), m)), v);
        const float a = __fadd_rn(delta, __fmul_rn(__fmul_rn(gt, prev_adv), m));
        adv[i] = a;
        ret[i] = __fadd_rn(v, a);
        prev_value = v;
        prev_adv = a;
    }
}
int launch_gae(int64_t T, const float *rewards, const float *masks, const float *values, double gamma, double tau,
               float *adv, float *ret, hipStream_t st) {
    if (T <= 0) return 0;
    // python evaluates gamma * tau in double before it meets the float32 tensor (common.py:16)
    hipLaunchKernelGGL(gae_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, st, T, rewards, masks, values, (float)gamma, (float)(gamma * tau), adv, ret);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------
// torch.optim.Adam single-tensor semantics (coupled weight decay, bias corrections, eps outside sqrt)
// ------------------------------------------------------------------------------------------
__global__ void adam_kernel(int64_t n, float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                            float *__restrict__ v, float step_size, float b1, float b2, float one_m_b1, float one_m_b2,
                            float eps, float wd, float bc2_sqrt) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float grad = g[i];
    const float param = p[i];
    if (wd != 0.f) grad = fmaf(wd, param, grad);
    const float mi = m[i] + (grad - m[i]) * one_m_b1;             // exp_avg.lerp_(grad, 1 - beta1)
    const float vi = v[i] * b2 + one_m_b2 * grad * grad;          // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = param - step_size * (mi / denom);                      // param.addcdiv_(exp_avg, denom, -lr / bc1)
}
int launch_adam(int64_t n, float *p, const float *g, float *m, float *v, int step, double lr, double b1, double b2,
                double eps, double wd, hipStream_t st) {
    if (n <= 0) return 0;
    const double bc1 = 1.0 - pow(b1, step), bc2 = 1.0 - pow(b2, step);
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, p, g, m, v, (float)(lr / bc1), (float)b1,
                       (float)b2, (float)(1.0 - b1), (float)(1.0 - b2), (float)eps, (float)wd, (float)sqrt(bc2));
    UPAMD_HIP(hipGetLastError());
    return 0;
}

// All optimizer groups of a step in ONE launch: group q covers [begin, end) with its own bias corrections (its own
// step count); a group with step == 0 is skipped (a head without rows: grad None in the reference).  Optionally the
// four loss scalars behind the gradients are copied out (the caller's per-step log row).
struct AdamGroups {
    int64_t begin[4], end[4];
    float step_size[4], bc2_sqrt[4];
    int n;
};
__global__ void adam_groups_kernel(AdamGroups G, float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                   float *__restrict__ v, float b1, float b2, float one_m_b1, float one_m_b2, float eps,
                                   float wd, const float *__restrict__ loss_src, float *__restrict__ loss_dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (loss_dst && i < 4) loss_dst[i] = loss_src[i];
    int q = -1;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (k < G.n && i >= G.begin[k] && i < G.end[k]) q = k;
    if (q < 0) return;
    float grad = g[i];
    const float param = p[i];
    if (wd != 0.f) grad = fmaf(wd, param, grad);
    const float mi = m[i] + (grad - m[i]) * one_m_b1;
    const float vi = v[i] * b2 + one_m_b2 * grad * grad;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / G.bc2_sqrt[q] + eps;
    p[i] = param - G.step_size[q] * (mi / denom);
}
int launch_adam_groups(int n_groups, const int64_t *begin, const int64_t *end, const int32_t *step, float *p, const float *g,
                       float *m, float *v, double lr, double b1, double b2, double eps, double wd, const float *loss_src,
                       float *loss_dst, hipStream_t st) {
    AdamGroups G;
    G.n = 0;
    int64_t hi = loss_dst ? 4 : 0;
    for (int k = 0; k < n_groups; ++k) {
        if (step[k] <= 0 || end[k] <= begin[k]) continue;
        const double bc1 = 1.0 - pow(b1, step[k]), bc2 = 1.0 - pow(b2, step[k]);
        G.begin[G.n] = begin[k];
        G.end[G.n] = end[k];
        G.step_size[G.n] = (float)(lr / bc1);
        G.bc2_sqrt[G.n] = (float)sqrt(bc2);
        hi = std::max(hi, end[k]);
        ++G.n;
    }
    if (hi <= 0) return 0;
    hipLaunchKernelGGL(adam_groups_kernel, dim3((unsigned)((hi + 255) / 256)), dim3(256), 0, st, G, p, g, m, v, (float)b1,
                       (float)b2, (float)(1.0 - b1), (float)(1.0 - b2), (float)eps, (float)wd, loss_src, loss_dst);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

// sum of squares: per-block partials into scratch, then one block adds them (fixed order) into *out_accum
__global__ __launch_bounds__(256) void sumsq_part_kernel(const float *__restrict__ x, int64_t n, float *__restrict__ scratch) {
    __shared__ float red[4];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc = fmaf(x[i], x[i], acc);
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) scratch[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
__global__ void sumsq_final_kernel(const float *__restrict__ scratch, int nblk, float *__restrict__ out_accum) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float acc = 0.f;
        for (int i = 0; i < nblk; ++i) acc += scratch[i];
        *out_accum += acc;
    }
}
int launch_sumsq(const float *x, int64_t n, float *scratch, float *out_accum, hipStream_t st) {
    if (n <= 0) return 0;
    int nblk = (int)((n + 255) / 256);
    if (nblk > 1024) nblk = 1024;
    hipLaunchKernelGGL(sumsq_part_kernel, dim3(nblk), dim3(256), 0, st, x, n, scratch);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(64), 0, st, scratch, nblk, out_accum);
    UPAMD_HIP(hipGetLastError());
    return 0;
}
// torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to <= 1
__global__ void clip_scale_kernel(float *__restrict__ g, int64_t n, const float *__restrict__ sumsq, float max_norm) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float coef = fminf(max_norm / (sqrtf(*sumsq) + 1e-6f), 1.f);
    g[i] *= coef;
}
int launch_clip_scale(float *g, int64_t n, const float *sumsq, float max_norm, hipStream_t st) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(clip_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g, n, sumsq, max_norm);
    UPAMD_HIP(hipGetLastError());
    return 0;
}


// ------------------------------------------------------------------------------------------
// select_action (urban_planning/models/policy.py:67-85) for a batch of rows, from the RAGGED pointer-head logits: one wave per
// row.  The reference builds a Categorical over the PADDED edge / node slots with the pad constant -2^32 + 1 in the masked ones
// (policy.py:50-52, 59-61): their probability is exactly 0, so the distribution over the row's own candidates is the same
// distribution, and its arg-max (first maximum in slot order: candidates are stored in slot order) the same arg-max.
//   greedy[b] != 0 : arg-max                       (mean_action, :76-77 / :82-83)
//   else           : one draw by inverse CDF with the row's uniform u[b] in [0, 1): the smallest candidate j whose
//                    running sum of exp(z - max) exceeds u * total  (Categorical.sample's distribution; the uniforms come
//                    from the caller's generator)
// A row without any candidate is the reference's uniform Categorical over the padded slots (arg-max: slot 0).
// Output: actions[b] = (edge slot, 0) for a land-use row, (0, node) for a road row, (0, 0) for any other stage.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void select_actions_kernel(const int32_t *__restrict__ meta, const int32_t *__restrict__ he_slot,
                                                            const uint16_t *__restrict__ rn_node, const int32_t *__restrict__ idx,
                                                            const int32_t *__restrict__ he_off, const int32_t *__restrict__ rn_off,
                                                            const float *__restrict__ z_he, const float *__restrict__ z_rn,
                                                            const uint8_t *__restrict__ greedy, const float *__restrict__ uniform,
                                                            float *__restrict__ actions) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int32_t *m = meta + (int64_t)idx[b] * UPAMD_META_STRIDE;
    const int stage = m[4];
    float a0 = 0.f, a1 = 0.f;
    if (stage == 0 || stage == 1) {
        const int lo = stage == 0 ? he_off[b] : rn_off[b];
        const int cnt = (stage == 0 ? he_off[b + 1] : rn_off[b + 1]) - lo;
        const float *z = (stage == 0 ? z_he : z_rn) + lo;
        float pick;
        if (cnt <= 0) {
            const int pad = stage == 0 ? m[8] : m[7];
            pick = greedy[b] ? 0.f : floorf(fminf(uniform[b], 0.99999994f) * (float)(pad > 0 ? pad : 1));
        } else {
            // maximum and its first position
            float mx = -INFINITY;
            int arg = 0x7fffffff;
            for (int j = lane; j < cnt; j += 64) {
                const float v = z[j];
                if (v > mx) { mx = v; arg = j; }
            }
            for (int off = 32; off > 0; off >>= 1) {
                const float om = __shfl_xor(mx, off);
                const int oa = __shfl_xor(arg, off);
                if (om > mx || (om == mx && oa < arg)) { mx = om; arg = oa; }
            }
            int k = arg;
            if (!greedy[b]) {
                float tot = 0.f;
                for (int j = lane; j < cnt; j += 64) tot += __expf(z[j] - mx);
                for (int off = 32; off > 0; off >>= 1) tot += __shfl_xor(tot, off);
                const float target = uniform[b] * tot;
                float run = 0.f;           // sum of the weights of all candidates before this chunk (wave-uniform)
                k = -1;
                for (int j0 = 0; j0 < cnt && k < 0; j0 += 64) {
                    const int j = j0 + lane;
                    const float wgt = j < cnt ? __expf(z[j] - mx) : 0.f;
                    float inc = wgt;       // inclusive prefix sum over the lanes
                    for (int off = 1; off < 64; off <<= 1) {
                        const float o = __shfl_up(inc, off);
                        if (lane >= off) inc += o;
                    }
                    const bool hit = j < cnt && run + inc > target;
                    const unsigned long long mask = __ballot(hit);
                    if (mask) k = j0 + __ffsll((long long)mask) - 1;
                    run += __shfl(inc, 63);
                }
                if (k < 0) {               // rounding left the target at / above the total: the last candidate with weight > 0
                    k = arg;
                    for (int j = cnt - 1; j >= 0; --j)
                        if (__expf(z[j] - mx) > 0.f) { k = j; break; }
                }
            }
            pick = stage == 0 ? (float)he_slot[m[11] + k] : (float)rn_node[m[12] + k];
        }
        if (stage == 0) a0 = pick; else a1 = pick;
    }
    if (lane == 0) {
        actions[2 * b] = a0;
        actions[2 * b + 1] = a1;
    }
}

int launch_select_actions(int B, const int32_t *meta, const int32_t *he_slot, const uint16_t *rn_node, const int32_t *idx,
                          const int32_t *he_off, const int32_t *rn_off, const float *z_he, const float *z_rn,
                          const uint8_t *greedy, const float *uniform, float *actions, hipStream_t st) {
    hipLaunchKernelGGL(select_actions_kernel, dim3((unsigned)B), dim3(64), 0, st, meta, he_slot, rn_node, idx, he_off, rn_off,
                       z_he, z_rn, greedy, uniform, actions);
    UPAMD_HIP(hipGetLastError());
    return 0;
}

}  // namespace upamd
