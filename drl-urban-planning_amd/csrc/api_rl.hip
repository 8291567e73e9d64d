// C-ABI entry points of the PPO minibatch math (loss, GAE, first-step clip, Adam).
#include <cstring>

#include "kernels.h"

using namespace upamd;

extern "C" int upamd_ppo_loss(int32_t B, const float *value_dev, const float *logp_dev, const float *ent_dev,
                              const float *adv_dev, const float *ret_dev, const float *old_logp_dev,
                              const float *exps_dev, float clip_epsilon, float value_pred_coef, float entropy_coef,
                              float inv_rows, float inv_ind, float *dvalue_dev, float *dlogp_dev, float *dent_dev,
                              float *losses_dev, void *stream) {
    if (B <= 0) return fail(UPAMD_E_INVALID, "upamd_ppo_loss: B must be > 0");
    if (!value_dev || !logp_dev || !ent_dev || !adv_dev || !ret_dev || !old_logp_dev || !exps_dev || !dvalue_dev ||
        !dlogp_dev || !dent_dev || !losses_dev)
        return fail(UPAMD_E_INVALID, "upamd_ppo_loss: null pointer");
    return launch_ppo_loss(B, value_dev, logp_dev, ent_dev, nullptr, adv_dev, ret_dev, old_logp_dev, exps_dev, clip_epsilon,
                           value_pred_coef, entropy_coef, inv_rows, inv_ind, dvalue_dev, dlogp_dev, dent_dev, losses_dev,
                           nullptr, 0, static_cast<hipStream_t>(stream));
}

extern "C" int upamd_ppo_loss_rows(int32_t B, const float *value_dev, const float *logp_dev, const float *ent_dev,
                                   const int64_t *rows_dev, const float *adv_all_dev, const float *ret_all_dev,
                                   const float *old_logp_all_dev, const float *exps_all_dev, float clip_epsilon,
                                   float value_pred_coef, float entropy_coef, float inv_rows, float inv_ind,
                                   float *dvalue_dev, float *dlogp_dev, float *dent_dev, float *losses_dev,
                                   float *zero_dev, int64_t n_zero, void *stream) {
    if (B <= 0 || n_zero < 0) return fail(UPAMD_E_INVALID, "upamd_ppo_loss_rows: B must be > 0, n_zero >= 0");
    if (!value_dev || !logp_dev || !ent_dev || !rows_dev || !adv_all_dev || !ret_all_dev || !old_logp_all_dev ||
        !exps_all_dev || !dvalue_dev || !dlogp_dev || !dent_dev || !losses_dev || (n_zero > 0 && !zero_dev))
        return fail(UPAMD_E_INVALID, "upamd_ppo_loss_rows: null pointer");
    return launch_ppo_loss(B, value_dev, logp_dev, ent_dev, rows_dev, adv_all_dev, ret_all_dev, old_logp_all_dev,
                           exps_all_dev, clip_epsilon, value_pred_coef, entropy_coef, inv_rows, inv_ind, dvalue_dev,
                           dlogp_dev, dent_dev, losses_dev, zero_dev, n_zero, static_cast<hipStream_t>(stream));
}

extern "C" int upamd_select_actions(const void *packed_dev, const upamd_pack_layout *layout, const upamd_minibatch *mb,
                                    const float *z_he_dev, const float *z_rn_dev, const uint8_t *greedy_dev,
                                    const float *uniform_dev, float *actions_dev, void *stream) {
    if (!packed_dev || !layout || !mb || !greedy_dev || !uniform_dev || !actions_dev)
        return fail(UPAMD_E_INVALID, "upamd_select_actions: null pointer");
    if (mb->B <= 0) return fail(UPAMD_E_INVALID, "upamd_select_actions: B must be > 0");
    if (!mb->idx_dev || !mb->he_off_dev || !mb->rn_off_dev) return fail(UPAMD_E_INVALID, "upamd_select_actions: minibatch schedule pointers are null");
    if ((mb->n_he > 0 && !z_he_dev) || (mb->n_rn > 0 && !z_rn_dev))
        return fail(UPAMD_E_INVALID, "upamd_select_actions: the minibatch has candidates but their logits are null");
    const char *b = static_cast<const char *>(packed_dev);
    return launch_select_actions(mb->B, reinterpret_cast<const int32_t *>(b + layout->off_meta),
                                 reinterpret_cast<const int32_t *>(b + layout->off_he_slot),
                                 reinterpret_cast<const uint16_t *>(b + layout->off_rn_node), mb->idx_dev, mb->he_off_dev,
                                 mb->rn_off_dev, z_he_dev, z_rn_dev, greedy_dev, uniform_dev, actions_dev,
                                 static_cast<hipStream_t>(stream));
}

extern "C" int upamd_gae(int64_t T, const float *rewards_dev, const float *masks_dev, const float *values_dev,
                         double gamma, double tau, float *adv_dev, float *ret_dev, void *stream) {
    if (T <= 0) return fail(UPAMD_E_INVALID, "upamd_gae: T must be > 0");
    if (!rewards_dev || !masks_dev || !values_dev || !adv_dev || !ret_dev) return fail(UPAMD_E_INVALID, "upamd_gae: null pointer");
    return launch_gae(T, rewards_dev, masks_dev, values_dev, gamma, tau, adv_dev, ret_dev, static_cast<hipStream_t>(stream));
}

extern "C" int upamd_clip_first_step(const upamd_model_desc *desc, float *grads_dev, float max_norm,
                                     float *scratch_dev, void *stream) {
    ParamLayout P;
    int rc = build_param_layout(desc, &P);
    if (rc) return rc;
    if (!grads_dev || !scratch_dev) return fail(UPAMD_E_INVALID, "upamd_clip_first_step: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    // encoder range = group 0 minus the value head (the value head is the tail of group 0)
    const int64_t enc_b = P.group_begin[0];
    const int64_t enc_e = P.off(P.value_w[0]);
    const int64_t val_e = P.group_end[0];
    float *acc = scratch_dev + 2048;   // two accumulators behind the per-block partials
    UPAMD_HIP(hipMemsetAsync(acc, 0, 2 * sizeof(float), st));
    // 1) clip_grad_norm_(policy_net.parameters()) : shared encoder + both pointer heads
    rc = launch_sumsq(grads_dev + enc_b, enc_e - enc_b, scratch_dev, acc, st);
    if (rc) return rc;
    rc = launch_sumsq(grads_dev + P.group_begin[1], P.group_end[2] - P.group_begin[1], scratch_dev, acc, st);
    if (rc) return rc;
    rc = launch_clip_scale(grads_dev + enc_b, enc_e - enc_b, acc, max_norm, st);
    if (rc) return rc;
    rc = launch_clip_scale(grads_dev + P.group_begin[1], P.group_end[2] - P.group_begin[1], acc, max_norm, st);
    if (rc) return rc;
    // 2) clip_grad_norm_(value_net.parameters()) : (already scaled) shared encoder + value head
    rc = launch_sumsq(grads_dev + enc_b, val_e - enc_b, scratch_dev, acc + 1, st);
    if (rc) return rc;
    return launch_clip_scale(grads_dev + enc_b, val_e - enc_b, acc + 1, max_norm, st);
}

extern "C" int upamd_adam_step(int64_t begin, int64_t end, float *params_dev, const float *grads_dev, float *m_dev,
                               float *v_dev, int32_t step, double lr, double beta1, double beta2, double eps,
                               double weight_decay, void *stream) {
    if (end < begin || step < 1) return fail(UPAMD_E_INVALID, "upamd_adam_step: bad range or step");
    if (!params_dev || !grads_dev || !m_dev || !v_dev) return fail(UPAMD_E_INVALID, "upamd_adam_step: null pointer");
    return launch_adam(end - begin, params_dev + begin, grads_dev + begin, m_dev + begin, v_dev + begin, step, lr, beta1,
                       beta2, eps, weight_decay, static_cast<hipStream_t>(stream));
}

extern "C" int upamd_adam_groups(int32_t n_groups, const int64_t *begin, const int64_t *end, const int32_t *step,
                                 float *params_dev, const float *grads_dev, float *m_dev, float *v_dev, double lr,
                                 double beta1, double beta2, double eps, double weight_decay, const float *loss_src_dev,
                                 float *loss_dst_dev, void *stream) {
    if (n_groups < 0 || n_groups > 4 || (n_groups > 0 && (!begin || !end || !step)))
        return fail(UPAMD_E_INVALID, "upamd_adam_groups: 0..4 groups with begin/end/step tables");
    for (int k = 0; k < n_groups; ++k)
        if (end[k] < begin[k] || begin[k] < 0 || step[k] < 0) return fail(UPAMD_E_INVALID, "upamd_adam_groups: bad range or step");
    if (!params_dev || !grads_dev || !m_dev || !v_dev || ((loss_dst_dev != nullptr) != (loss_src_dev != nullptr)))
        return fail(UPAMD_E_INVALID, "upamd_adam_groups: null pointer");
    return launch_adam_groups(n_groups, begin, end, step, params_dev, grads_dev, m_dev, v_dev, lr, beta1, beta2, eps,
                              weight_decay, loss_src_dev, loss_dst_dev, static_cast<hipStream_t>(stream));
}

extern "C" int upamd_gemm_nt(const float *A_dev, int64_t M, int32_t K, int64_t lda, int32_t a_row_major, const float *W_dev,
                             int32_t N, int64_t ldw, const float *bias_dev, const float *R_dev, float *C_dev, int64_t ldc,
                             int32_t c_row_major, int32_t act_tanh, float alpha, void *stream) {
    if (!A_dev || !W_dev || !C_dev || M <= 0 || K <= 0 || N <= 0) return fail(UPAMD_E_INVALID, "upamd_gemm_nt: bad argument");
    GemmNT g{A_dev, M, K, lda, a_row_major != 0, W_dev, N, ldw, bias_dev, R_dev, C_dev, ldc, c_row_major != 0, act_tanh, alpha};
    return launch_gemm_nt_ex(g, static_cast<hipStream_t>(stream), nullptr);
}

extern "C" int64_t upamd_gemm_nt_split_scratch_bytes(int32_t N, int32_t K) { return gemm_nt_split_scratch_bytes(N, K); }

extern "C" int upamd_gemm_nt_split(const float *A_dev, int64_t M, int32_t K, const float *W_dev, int32_t N, int64_t ldw,
                                   const float *bias_dev, const float *R_dev, float *C_dev, int32_t act_tanh, float alpha,
                                   int32_t n_products, void *scratch_dev, void *stream) {
    if (!A_dev || !W_dev || !C_dev || M <= 0 || K <= 0 || N <= 0) return fail(UPAMD_E_INVALID, "upamd_gemm_nt_split: bad argument");
    GemmNT g{A_dev, M, K, 0, false, W_dev, N, ldw, bias_dev, R_dev, C_dev, 0, false, act_tanh, alpha};
    return launch_gemm_nt_split(g, scratch_dev, n_products, static_cast<hipStream_t>(stream), nullptr);
}

// lab hook: one wave samples (shader clock, 100 MHz wall clock) pairs while other streams run kernels; the ratio of the
// differences is the effective shader clock under that load
__global__ void clock_probe_kernel(long long *out, int samples, int gap_ticks) {
    if (threadIdx.x != 0) return;
    for (int i = 0; i < samples; ++i) {
        const long long w0 = wall_clock64();
        out[2 * i] = clock64();
        out[2 * i + 1] = w0;
        while (wall_clock64() - w0 < gap_ticks) __builtin_amdgcn_s_sleep(16);
    }
}

extern "C" int upamd_clock_probe(void *out_dev, int32_t samples, int32_t gap_ticks, void *stream) {
    if (!out_dev || samples <= 0) return fail(UPAMD_E_INVALID, "upamd_clock_probe: bad argument");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream),
                       static_cast<long long *>(out_dev), samples, gap_ticks);
    UPAMD_HIP(hipGetLastError());
    return UPAMD_OK;
}

extern "C" int upamd_tiny_profile(void *buf_dev) {
    set_tiny_prof(buf_dev);
    return UPAMD_OK;
}

extern "C" int upamd_tune(const char *name, int32_t value) {
    if (!name) return fail(UPAMD_E_INVALID, "upamd_tune: name is null");
    if (!strcmp(name, "gemm_nt_dma")) { set_gemm_nt_dma_variant(value); return UPAMD_OK; }
    if (!strcmp(name, "gemm_lds_pad")) { set_gemm_lds_pad(value); return UPAMD_OK; }
    if (!strcmp(name, "gemm_stagger_mode")) { set_gemm_stagger(value, -1); return UPAMD_OK; }
    if (!strcmp(name, "gemm_stagger_cycles")) { set_gemm_stagger(-1, value); return UPAMD_OK; }
    if (!strcmp(name, "fold_layer1")) { set_fold_layer1(value); return UPAMD_OK; }
    if (!strcmp(name, "pq_exp")) { set_pq_exp(value); return UPAMD_OK; }
    if (!strcmp(name, "bwd_nb_global")) { set_bwd_nb_global(value); return UPAMD_OK; }
    if (!strcmp(name, "nt_min_wgs")) { set_gemm_nt_min_wgs(value); return UPAMD_OK; }
    if (!strcmp(name, "gemm_split")) { set_gemm_nt_split(value); return UPAMD_OK; }
    if (!strcmp(name, "he_fused")) { set_he_feat_fused(value); return UPAMD_OK; }
    if (!strcmp(name, "side_wgrad")) { set_side_wgrad(value); return UPAMD_OK; }
    if (!strcmp(name, "grad_buckets")) { set_grad_buckets(value); return UPAMD_OK; }
    if (!strcmp(name, "side_priority")) { set_side_priority(value); return UPAMD_OK; }
    if (!strcmp(name, "side_heads")) { set_side_heads(value); return UPAMD_OK; }
    if (!strcmp(name, "side_stream")) { set_side_stream(value); return UPAMD_OK; }
    if (!strcmp(name, "fwd_h_hbm")) { set_fwd_h_hbm(value); return UPAMD_OK; }
    if (!strcmp(name, "fe_half")) { set_fe_half(value); return UPAMD_OK; }
    if (!strcmp(name, "tiny_fused")) { set_tiny_fused(value); return UPAMD_OK; }
    if (!strcmp(name, "tiny_threads")) { set_tiny_threads(value); return UPAMD_OK; }
    return fail(UPAMD_E_INVALID, "upamd_tune: unknown knob '%s'", name);
}

extern "C" int64_t upamd_gemm_tn_scratch_floats(int32_t I, int32_t J, int64_t M) {
    return (int64_t)tn_splits(I, J, M) * I * J;
}

extern "C" int upamd_gemm_tn(const float *A_dev, int32_t I, int64_t lda, const float *B_dev, int32_t J, int64_t ldb, int64_t M,
                             int32_t row_major, float *scratch_dev, float *out_dev, void *stream) {
    if (!A_dev || !B_dev || !scratch_dev || !out_dev || I <= 0 || J <= 0) return fail(UPAMD_E_INVALID, "upamd_gemm_tn: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    GemmTN g{A_dev, I, lda, B_dev, J, ldb, M, row_major != 0, scratch_dev};
    int S = 1;
    int rc = launch_gemm_tn_ex(g, &S, st, nullptr);
    if (rc) return rc;
    UPAMD_HIP(hipMemsetAsync(out_dev, 0, sizeof(float) * (size_t)I * J, st));
    return launch_reduce_slabs(scratch_dev, S, I, J, 0, J, out_dev, J, st);
}
