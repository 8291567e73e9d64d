/*
 * upamd.h -- C ABI of the MI355X-native SGNN policy/value + PPO-update hot path.
 *
 * The reference (tsinghua-fib-lab/DRL-urban-planning) is pure Python/PyTorch and has no FFI
 * or plugin layer for this path: the drop-in boundary is its Python class
 * surface.  This header is the boundary BEHIND that surface -- what a binding for the path
 * (ctypes here) talks to.  Every entry point cites the reference code whose
 * work it replaces (paths relative to the reference repo root).
 *
 * What sits ABOVE this ABI is the reference's Python surface, kept by drl-urban-planning_amd/{models,agent}.py:
 * create_sgnn_model / create_mlp_model, policy_net.forward / select_action / get_log_prob_entropy, value_net(x),
 * update_params(batch, iteration).  One deliberate difference: on a GPU module policy_net.forward(x) returns the
 * reference's Categorical objects (same padded-slot layout, same pad constant) but built from the HIP forward's logits
 * WITHOUT an autograd graph -- gradients flow through get_log_prob_entropy (upamd_forward / upamd_backward), which is
 * the only route the reference's update uses (urban_planning_agent.py:363-371).
 *
 * Conventions: plain C types, pointers and sizes only.  `*_dev` pointers are device (HBM)
 * addresses, everything else is host memory.  `stream` is a hipStream_t passed as void*
 * (NULL = default stream).  Every function returns 0 on success and a negative UPAMD_E_*
 * code on failure; upamd_last_error() then returns a thread-local message.  Nothing aborts.
 * All floating point is fp32 (the reference trains in float32: urban_planning/train.py:47-48).
 */
#ifndef UPAMD_H
#define UPAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: exactly the entry points declared here are exported */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define UPAMD_ABI_VERSION 7

#define UPAMD_OK 0
#define UPAMD_E_INVALID (-1)   /* bad argument / unsupported configuration            */
#define UPAMD_E_HIP (-2)       /* a HIP runtime call failed                            */
#define UPAMD_E_LIMIT (-3)     /* a size limit of the packed format was exceeded       */
#define UPAMD_E_WORKSPACE (-4) /* caller's workspace is too small                      */
#define UPAMD_E_REPLAN (-5)    /* upamd_pack_fill*: the masks-only plan does not hold for these states; plan with exact = 1 */

#define UPAMD_MAX_MLP 4        /* max depth of each small MLP (hidden-size lists)      */
#define UPAMD_META_STRIDE 16   /* int32 words per state in the meta table              */
#define UPAMD_NODE_PAD 24      /* packed node-feature row width (node_dim <= 24)       */
#define UPAMD_MAX_EDGE_FC 4     /* max num_edge_fc_layers (sub-layers of one edge MLP)  */

int upamd_abi_version(void);
const char *upamd_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Model description == the reference's YAML specs (urban_planning/cfg/exp_cfg/real/hlg.yaml:21-33)
 * read by create_sgnn_model (urban_planning/models/model.py:8-19).
 * Constraints of this build: D % 16 == 0, D % heads == 0, num_edge_fc_layers <= UPAMD_MAX_EDGE_FC,
 * node_dim <= 24, policy-head hidden sizes are multiples of 16 and end in 1.
 * ------------------------------------------------------------------------------------------ */
#define UPAMD_ENCODER_SGNN 0   /* SGNNStateEncoder  (urban_planning/models/state_encoder.py:7-214,   --agent rl-sgnn) */
#define UPAMD_ENCODER_MLP 1    /* MLPStateEncoder   (urban_planning/models/state_encoder.py:217-308, --agent rl-mlp): no
                                  message passing, no attention; L / heads are ignored                                  */
#define UPAMD_MLP_TYPE_COLS 14 /* city_config.NUM_TYPES + 1 one-hot type columns                                        */
#define UPAMD_MLP_FEASIBLE 1   /* city_config.FEASIBLE                                                                  */

typedef struct upamd_model_desc {
    int32_t node_dim;                        /* agent.node_dim (23)                               */
    int32_t numerical_dim;                   /* agent.numerical_feature_size (52)                 */
    int32_t D;                               /* gcn_node_dim                                      */
    int32_t L;                               /* num_gcn_layers                                    */
    int32_t heads;                           /* num_attention_heads                               */
    int32_t n_num;                           /* len(state_encoder_hidden_size)                    */
    int32_t num_hidden[UPAMD_MAX_MLP];
    int32_t n_land;                          /* len(policy_land_use_head_hidden_size), last == 1  */
    int32_t land_hidden[UPAMD_MAX_MLP];
    int32_t n_road;                          /* len(policy_road_head_hidden_size), last == 1      */
    int32_t road_hidden[UPAMD_MAX_MLP];
    int32_t n_value;                         /* len(value_head_hidden_size), last == 1            */
    int32_t value_hidden[UPAMD_MAX_MLP];
    int32_t encoder;                         /* UPAMD_ENCODER_SGNN | UPAMD_ENCODER_MLP             */
    int32_t edge_fc_layers;                  /* num_edge_fc_layers (state_encoder.py:59-82); 0 is read as 1.  1 (every shipped
                                                config): the factorised P/Q message passing.  > 1: the sub-layers behind the
                                                first one act on per-incidence tensors (one row per edge direction)     */
} upamd_model_desc;

/* Flat fp32 parameter buffer layout.  Tensor names are the reference's de-duplicated
 * state_dict keys ("shared_net.node_encoder.weight", "policy_land_use_head.land_use_linear_0.weight",
 * "value_head.linear_0.weight", ...; urban_planning/models/{state_encoder,policy,value}.py).
 * group: 0 = shared encoder + value head, 1 = land-use head, 2 = road head (the three sets
 * torch.optim.Adam treats independently because a head that was not evaluated has grad None,
 * urban_planning/models/policy.py:48,57).  Offsets are in floats and 4-float aligned;
 * n_floats includes the alignment padding. */
int upamd_param_count(const upamd_model_desc *desc, int64_t *n_floats, int32_t *n_tensors);
int upamd_param_info(const upamd_model_desc *desc, int32_t index, char *name_out, int32_t name_cap,
                     int64_t *offset, int32_t *rows, int32_t *cols, int32_t *group);
/* float ranges [begin,end) of the three optimizer groups inside the flat buffer */
int upamd_param_groups(const upamd_model_desc *desc, int64_t begin_out[3], int64_t end_out[3]);

/* ------------------------------------------------------------------------------------------
 * Replay packer (host).  Replaces the per-minibatch `tensorfy` + `batch_data` of the reference
 * (urban_planning/agents/urban_planning_agent.py:16-20, urban_planning/models/state_encoder.py:163-177):
 * the T padded 9-field states of one PPO iteration (wire format:
 * urban_planning/envs/observation_extractor.py:207-228) are converted ONCE into a ragged/CSR
 * batch that goes to HBM in a single copy.
 *
 * ptrs: [9][T] host addresses of the 9 fields of every state, field-major:
 *   0 numerical f32[numerical_dim]   1 node_features f32[pad_n][node_dim]  2 edge_index i64[pad_e][2]
 *   3 current_node f32[node_dim]     4 node_mask u8[pad_n]                 5 edge_mask u8[pad_e]
 *   6 land_use_mask u8[pad_e]        7 road_mask u8[pad_n]                 8 stage f32[3]
 * actions: f32[T][2] (indices into the PADDED edge / node order, urban_planning/models/policy.py:93,99).
 * meta (out, int32[T][UPAMD_META_STRIDE]):
 *   0 n  1 e  2 n_head_edges  3 n_road_nodes  4 stage  5 action (candidate index or -1)
 *   6 n_node_mask  7 pad_n  8 pad_e  9 node_off  10 edge_off  11 he_off  12 rn_off  13 rowptr_off
 * ------------------------------------------------------------------------------------------ */
typedef struct upamd_pack_layout {
    int64_t T;
    int64_t total_nodes, total_edges, total_he, total_rn;
    int32_t node_dim, numerical_dim;
    /* byte offsets of the sections inside the packed buffer (each 256-byte aligned) */
    int64_t off_meta;      /* int32 [T][UPAMD_META_STRIDE]                                      */
    int64_t off_x;         /* f32   [total_nodes][UPAMD_NODE_PAD]                                */
    int64_t off_nmask;     /* u8    [total_nodes]                                                */
    int64_t off_rowptr;    /* int32 [total_nodes + T]   per graph n+1 local incidence offsets    */
    int64_t off_inc_nbr;   /* u16   [2*total_edges]     neighbour (local node id) per incidence  */
    int64_t off_he_src;    /* u16   [total_he]                                                   */
    int64_t off_he_dst;    /* u16   [total_he]                                                   */
    int64_t off_he_live;   /* u8    [total_he]          0 if the candidate edge is not a live edge */
    int64_t off_he_slot;   /* int32 [total_he]          padded edge slot of the candidate        */
    int64_t off_rn_node;   /* u16   [total_rn]          candidate node (== padded slot)          */
    int64_t off_numerical; /* f32   [T][numerical_dim]                                           */
    int64_t off_cur;       /* f32   [T][UPAMD_NODE_PAD]                                          */
    int64_t off_order;     /* u16   [total_nodes]       per graph: node ids sorted by degree (descending,
                                                         stable) -- the processing order of the edge kernels */
    int64_t off_hinc_ptr;  /* int32 [total_nodes + T]   per graph n+1 offsets into the candidate-incidence lists */
    int64_t off_hinc_nbr;  /* u16   [2*total_he]        per node: neighbour across each incident LIVE candidate edge */
    int64_t off_hinc_he;   /* u16   [2*total_he]        ... and that candidate's local index                         */
    /* inputs of the rl-mlp encoder (urban_planning/models/state_encoder.py:263-282): an edge is represented by the raw
     * features of ONE endpoint -- the second one if that node's type (arg-max of the first NUM_TYPES+1 = 14 feature
     * columns) is FEASIBLE (= 1, urban_planning/envs/city_config.py:24,53), else the first one */
    int64_t off_he_sel;    /* u16   [total_he]          the selected endpoint of every candidate edge               */
    int64_t off_xbar;      /* f32   [T][UPAMD_NODE_PAD] mean over the live edges of the selected endpoint's features */
    int64_t total_bytes;
} upamd_pack_layout;

int upamd_pack_plan(int64_t T, const uint64_t *ptrs, const int32_t *pad_n, const int32_t *pad_e,
                    const float *actions, int32_t node_dim, int32_t numerical_dim, int32_t n_threads,
                    int32_t *meta, upamd_pack_layout *layout);
/* States given as COMPACT WIRE RECORDS (the lossless trimmed form of the same nine arrays that the
 * rollout workers write into shared memory instead of pickling padded tuples through a queue, khrylib/rl/agents/agent.py:92-97):
 * rec_ptrs[t] / rec_sizes[t] = address and byte size of record t.  Fills the [9][T] field-address table and the per-state
 * array extents (the record's trimmed row counts) that upamd_pack_plan* / upamd_pack_fill* take -- no per-field host objects. */
int upamd_record_table(int64_t T, const uint64_t *rec_ptrs, const int64_t *rec_sizes, int32_t node_dim, int32_t numerical_dim,
                       uint64_t *ptrs, int32_t *pad_n, int32_t *pad_e);
/* exact = 1: upamd_pack_plan.  exact = 0: the counting pass reads the masks only (not the int64 edge list: a 16th of the host bytes)
 * and takes every graph's extent n from its node / road masks -- true for every state the extractor emits, whose edges join live
 * nodes (observation_extractor.py:84-132).  upamd_pack_fill* verifies each live endpoint against that extent and returns
 * UPAMD_E_REPLAN if one lies beyond it; the caller then plans again with exact = 1 (the host wrapper does).
 * `exact` is a flag word: bit 0 = exact counting pass, bit 1 (value 2) = plan WITHOUT the two sections only the rl-mlp encoder reads
 * (off_he_sel = off_xbar = -1; upamd_pack_fill* then skips the per-edge feature mean, two thirds of a state's fill time) -- an SGNN
 * engine never touches them, an rl-mlp engine refuses such a replay. */
int upamd_pack_plan_ex(int64_t T, const uint64_t *ptrs, const int32_t *pad_n, const int32_t *pad_e, const float *actions,
                       int32_t node_dim, int32_t numerical_dim, int32_t n_threads, int32_t exact, int32_t *meta,
                       upamd_pack_layout *layout);
int upamd_pack_fill(int64_t T, const uint64_t *ptrs, const int32_t *meta, const upamd_pack_layout *layout,
                    int32_t n_threads, void *out);
/* The same for the states [t_begin, t_end) only: every section of the packed buffer is state-major, so a range of states is ONE
 * contiguous byte range per section -- the host wrapper packs chunk k + 1 while chunk k is on its way to HBM and chunk k - 1 is
 * in the value / old-log-prob pre-pass (urban_planning_agent.py:256-264, :283-292).  Filling [0, T) range by range gives the bytes
 * upamd_pack_fill writes. */
int upamd_pack_fill_range(int64_t T, const uint64_t *ptrs, const int32_t *meta, const upamd_pack_layout *layout,
                          int64_t t_begin, int64_t t_end, int32_t n_threads, void *out);

/* ------------------------------------------------------------------------------------------
 * Engine: forward / backward of the whole policy+value network over one minibatch of graphs.
 * Replaces SGNNStateEncoder.forward (urban_planning/models/state_encoder.py:184-214),
 * UrbanPlanningPolicy.get_log_prob_entropy (urban_planning/models/policy.py:87-104),
 * UrbanPlanningValue.forward (urban_planning/models/value.py:36-39) and their autograd backward.
 * The shared encoder is evaluated ONCE per call (the reference evaluates it twice per optimizer
 * step, urban_planning_agent.py:330-332; the gradients are identical).
 * ------------------------------------------------------------------------------------------ */
typedef struct upamd_engine upamd_engine;

typedef struct upamd_minibatch {
    int32_t B;                /* rows (graphs) in this minibatch (this rank's share)               */
    int64_t n_nodes;          /* sum of n over the rows                                            */
    int64_t n_he;             /* sum of head-edge candidates                                       */
    int64_t n_rn;             /* sum of road-node candidates                                       */
    int32_t max_n;            /* max n over the rows                                               */
    int32_t max_inc;          /* max 2*e over the rows                                             */
    const int32_t *idx_dev;       /* [B]   state ids into the packed replay                        */
    const int32_t *node_off_dev;  /* [B+1] prefix sums of n in minibatch order                     */
    const int32_t *he_off_dev;    /* [B+1]                                                         */
    const int32_t *rn_off_dev;    /* [B+1]                                                         */
    /* only read when the model has edge_fc_layers > 1 (may be 0 / NULL otherwise): */
    int64_t n_inc;                /* sum of 2*e over the rows (incidences = edge directions)           */
    const int32_t *inc_off_dev;   /* [B+1] prefix sums of 2*e in minibatch order                       */
    /* read by the fused small-model path (upamd_step_fused_ok): max over the rows of the row's pointer-head candidates
     * (n_head_edges for a land-use row, n_road_nodes for a road row); 0 = not provided -> the general path is taken */
    int32_t max_cand;
} upamd_minibatch;

int upamd_engine_create(const upamd_model_desc *desc, upamd_engine **out);
void upamd_engine_destroy(upamd_engine *eng);

/* bytes of scratch HBM a minibatch of this shape needs (training=1 keeps activations) */
int upamd_workspace_bytes(upamd_engine *eng, const upamd_minibatch *mb, int32_t training, int64_t *bytes);

/* value_dev/logp_dev/ent_dev: f32[B].  keep != 0 leaves the activations in ws for upamd_backward. */
int upamd_forward(upamd_engine *eng, const void *packed_dev, const upamd_pack_layout *layout,
                  const upamd_minibatch *mb, const float *params_dev, void *ws_dev, int64_t ws_bytes,
                  float *value_dev, float *logp_dev, float *ent_dev, int32_t keep, void *stream);

/* seeds dvalue/dlogp/dent: f32[B] (dLoss/d output).  grads_dev: flat buffer (same layout as the
 * parameters), ACCUMULATED into (zero it first).  Must follow upamd_forward(keep=1) on the same ws. */
int upamd_backward(upamd_engine *eng, const void *packed_dev, const upamd_pack_layout *layout,
                   const upamd_minibatch *mb, const float *params_dev, void *ws_dev, int64_t ws_bytes,
                   const float *dvalue_dev, const float *dlogp_dev, const float *dent_dev,
                   float *grads_dev, void *stream);

/* Gradient buckets of the LAST upamd_backward enqueued on `stream` (data parallelism, SURVEY.md section 8e: the reference has one
 * optimizer.step() per minibatch, urban_planning_agent.py:334-337, and no collective; the north-star adds ONE all-reduce of the
 * flat gradient buffer per step).  The backward finalises that buffer range by range -- attention + value head + pointer heads
 * right behind the per-sample products, each GCN layer's weight + bias behind that layer's weight-gradient GEMM, the rest (numerical
 * + node encoder, first layer) at its end -- and records one event per range: begin[k], end[k]) are float offsets in readiness
 * order, disjoint, together [0, n_floats).  upamd_grad_bucket_wait makes `waiter_stream` wait for range k, so the caller can
 * all-reduce it on a communication stream while the layers below are still in their backward.  Paths that finalise everything at
 * the end (small / rl-mlp models, tune knob "grad_buckets" = 0) report one range.  Same sums in the same order either way: a
 * bucketed step is bit-identical to a single-range one.  Not updated by upamd_step_fused (one range by construction). */
int upamd_grad_buckets(upamd_engine *eng, void *stream, int32_t cap, int32_t *n_out, int64_t *begin, int64_t *end);
int upamd_grad_bucket_wait(upamd_engine *eng, void *stream, int32_t k, void *waiter_stream);

/* Small models (gcn_node_dim <= 32 -- the dims of every shipped YAML, hlg.yaml:21-33 -- single-Linear edge MLPs, graphs that fit one
 * workgroup's LDS): ONE launch runs forward + PPO loss seeds + backward of the whole minibatch, one workgroup per graph, and a second
 * one adds the per-workgroup gradient slabs in a fixed order.  Replaces the upamd_forward / upamd_ppo_loss_rows / upamd_backward
 * sequence of one optimizer step (urban_planning_agent.py:326-337) for such models -- same arguments, same results (same tolerances
 * against the reference), `grads_dev` is OVERWRITTEN (no zero_grad needed), `losses_dev` gets the four loss scalars.
 * upamd_step_fused_ok returns 1 when the engine's model and this minibatch are covered (upamd_forward / upamd_backward then take the
 * fused kernels too, and upamd_ws_tensor only serves "z_he" / "z_rn"), else 0. */
int upamd_step_fused_ok(upamd_engine *eng, const upamd_minibatch *mb);
int upamd_step_fused(upamd_engine *eng, const void *packed_dev, const upamd_pack_layout *layout, const upamd_minibatch *mb,
                     const float *params_dev, void *ws_dev, int64_t ws_bytes, const int64_t *rows_dev, const float *adv_dev,
                     const float *ret_dev, const float *old_logp_dev, const float *exps_dev, float clip_eps, float cv, float ce,
                     float inv_rows, float inv_ind, float *value_dev, float *logp_dev, float *ent_dev, float *grads_dev,
                     float *losses_dev, void *stream);

/* byte offset / shape of a named intermediate inside ws after upamd_forward(keep=1) -- for the
 * stage-by-stage parity tests.  kind: 0 = row-major [rows][cols], 1 = panel-major [cols/16][rows][16]. */
int upamd_ws_tensor(upamd_engine *eng, const upamd_minibatch *mb, const char *name, int64_t *byte_offset,
                    int64_t *rows, int64_t *cols, int32_t *kind);

/* Rollout inference: UrbanPlanningPolicy.select_action (urban_planning/models/policy.py:67-85) for every
 * row of a minibatch after upamd_forward -- per row the arg-max (greedy_dev[b] != 0: mean_action, :76-77 / :82-83) or one draw from
 * the Categorical over the row's pointer-head candidates, by inverse CDF with the row's uniform_dev[b] in [0, 1) (the reference draws
 * with Categorical.sample on the CPU generator, :78-79 / :84-85: the same distribution, another stream).  z_he_dev / z_rn_dev are the
 * ragged logits upamd_ws_tensor names "z_he" / "z_rn" (land-use rows' / road rows' candidates in minibatch order); the masked slots
 * of the reference's padded logits hold the pad constant -2^32 + 1 (policy.py:50-52, 59-61), i.e. probability exactly 0, so the
 * distribution over the candidates alone is the reference's.  A row without candidates is the reference's uniform Categorical over
 * its padded slots.  actions_dev: f32 [B][2] = (padded edge slot, 0) for a land-use row, (0, node) for a road row -- what
 * select_action returns (:70-83).  One launch, one wave per row. */
int upamd_select_actions(const void *packed_dev, const upamd_pack_layout *layout, const upamd_minibatch *mb,
                         const float *z_he_dev, const float *z_rn_dev, const uint8_t *greedy_dev, const float *uniform_dev,
                         float *actions_dev, void *stream);

/* ------------------------------------------------------------------------------------------
 * PPO minibatch math.
 * ------------------------------------------------------------------------------------------ */

/* Clipped-surrogate + value + entropy loss and its per-row seeds.  Replaces
 * ppo_entropy_loss (urban_planning/agents/urban_planning_agent.py:363-371), value_loss
 * (khrylib/rl/agents/agent_pg.py:19-23) and the loss assembly (:333).
 * inv_rows = 1/(global minibatch rows), inv_ind = 1/(global count of exps != 0): with one rank
 * these are 1/B and 1/|ind|; with data parallelism every rank passes the GLOBAL counts and the
 * partial losses/gradients are summed by the all-reduce.
 * losses_dev: f32[4] = {loss, value_loss, surr_loss, entropy_loss} (this rank's partial sums). */
int upamd_ppo_loss(int32_t B, const float *value_dev, const float *logp_dev, const float *ent_dev,
                   const float *adv_dev, const float *ret_dev, const float *old_logp_dev,
                   const float *exps_dev, float clip_epsilon, float value_pred_coef, float entropy_coef,
                   float inv_rows, float inv_ind, float *dvalue_dev, float *dlogp_dev, float *dent_dev,
                   float *losses_dev, void *stream);

/* The same loss for a minibatch given as ROW INDICES into the replay-wide arrays (what the reference's
 * `advantages[ind]`, `returns[ind]`, `fixed_log_probs[ind]`, `exps[ind]` gathers do, urban_planning_agent.py:316-321),
 * and -- fused, it is the launch in front of the backward -- the zeroing of the step's gradient buffer
 * (optimizer.zero_grad(), :335): zero_dev[0 .. n_zero) = 0.  rows_dev: int64[B]. */
int upamd_ppo_loss_rows(int32_t B, const float *value_dev, const float *logp_dev, const float *ent_dev,
                        const int64_t *rows_dev, const float *adv_all_dev, const float *ret_all_dev,
                        const float *old_logp_all_dev, const float *exps_all_dev, float clip_epsilon,
                        float value_pred_coef, float entropy_coef, float inv_rows, float inv_ind,
                        float *dvalue_dev, float *dlogp_dev, float *dent_dev, float *losses_dev,
                        float *zero_dev, int64_t n_zero, void *stream);

/* Generalised advantage estimation, bit-exact with the reference's Python loop
 * (khrylib/rl/core/common.py:5-26): trajectories are concatenated, masks[t]==0 ends an episode. */
int upamd_gae(int64_t T, const float *rewards_dev, const float *masks_dev, const float *values_dev,
              double gamma, double tau, float *adv_dev, float *ret_dev, void *stream);

/* The reference's gradient clipping as it actually executes (khrylib/rl/agents/agent_ppo.py:43-46 with
 * the generator lists of urban_planning_agent.py:46): clip_grad_norm_(policy params, max_norm) then
 * clip_grad_norm_(value params, max_norm), both effective on the first optimizer step of a process only.
 * The caller invokes this exactly once (first step).  scratch_dev: >= 4096 floats. */
int upamd_clip_first_step(const upamd_model_desc *desc, float *grads_dev, float max_norm,
                          float *scratch_dev, void *stream);

/* torch.optim.Adam (coupled L2 weight decay, urban_planning_agent.py:148-149) on the flat buffers,
 * one call per optimizer group.  `step` is that group's 1-based step count. */
int upamd_adam_step(int64_t begin, int64_t end, float *params_dev, const float *grads_dev, float *m_dev,
                    float *v_dev, int32_t step, double lr, double beta1, double beta2, double eps,
                    double weight_decay, void *stream);

/* optimizer.step() for all groups of a step in one launch (same arithmetic as upamd_adam_step per group):
 * host tables begin/end/step[n_groups <= 4]; step[k] == 0 skips group k (a head without rows in the minibatch has
 * grad None in the reference and torch's Adam skips it).  loss_src_dev/loss_dst_dev (both or neither): the four loss
 * scalars are copied out by the same launch (the caller's per-step log row). */
int upamd_adam_groups(int32_t n_groups, const int64_t *begin, const int64_t *end, const int32_t *step,
                      float *params_dev, const float *grads_dev, float *m_dev, float *v_dev, double lr,
                      double beta1, double beta2, double eps, double weight_decay, const float *loss_src_dev,
                      float *loss_dst_dev, void *stream);

/* ------------------------------------------------------------------------------------------
 * The two fp32-MFMA GEMM building blocks of the engine, exposed for kernel-level parity tests and
 * micro-benchmarks.  They stand in for the reference's nn.Linear products (and their autograd) on
 * [nodes, D] / [edges, 2D] tensors (urban_planning/models/state_encoder.py:19,59-82,110-130).
 * Layouts: row_major = 1 -> [rows][ld]; 0 -> panel-major [cols/16][rows][16] (ld ignored).
 * ------------------------------------------------------------------------------------------ */
/* C[M,N] = alpha * act(A[M,K] * W[N,K]^T + bias[N] + R[M,N]);  W row-major with leading dimension ldw */
int upamd_gemm_nt(const float *A_dev, int64_t M, int32_t K, int64_t lda, int32_t a_row_major, const float *W_dev,
                  int32_t N, int64_t ldw, const float *bias_dev, const float *R_dev, float *C_dev, int64_t ldc,
                  int32_t c_row_major, int32_t act_tanh, float alpha, void *stream);

/* OPT-IN variant of upamd_gemm_nt for panel-major operands (not used by upamd_forward / upamd_backward): the same fp32
 * product computed on the bf16 matrix pipe from an exact three-way bf16 split of both operands, n_products = 6 (terms
 * below 2^-26 relative dropped) or 9 (all partial products); fp32 accumulation.  scratch_dev: device buffer of
 * upamd_gemm_nt_split_scratch_bytes(N, K) bytes for the split weight planes.  N % 128 == 0, K % 32 == 0. */
int64_t upamd_gemm_nt_split_scratch_bytes(int32_t N, int32_t K);
int upamd_gemm_nt_split(const float *A_dev, int64_t M, int32_t K, const float *W_dev, int32_t N, int64_t ldw,
                        const float *bias_dev, const float *R_dev, float *C_dev, int32_t act_tanh, float alpha,
                        int32_t n_products, void *scratch_dev, void *stream);
/* out[I,J] (row-major, overwritten) = A[M,I]^T * B[M,J], deterministic split-K through scratch_dev
 * (upamd_gemm_tn_scratch_floats(I, J, M) floats) */
int64_t upamd_gemm_tn_scratch_floats(int32_t I, int32_t J, int64_t M);
int upamd_gemm_tn(const float *A_dev, int32_t I, int64_t lda, const float *B_dev, int32_t J, int64_t ldb, int64_t M,
                  int32_t row_major, float *scratch_dev, float *out_dev, void *stream);

/* Process-wide kernel-lab knobs (tools/gemm_lab.py, tests, bench.py's UPAMD_TUNE): select a kernel configuration by name; no
 * reference counterpart.  Defaults in brackets; every setting computes the same results (the tests run both sides).
 *   "gemm_nt_dma" [1]   0 = register-staged gemm_nt, k > 0 = LDS-DMA configuration k of the plain panel-major launches
 *   "gemm_split"  [0]   6 | 9 = node GEMMs as fp32-equivalent split-bf16 products (see upamd_gemm_nt_split)
 *   "fold_layer1" [1]   first GCN layer computed inside the message-passing kernels (H_0 / PQ_1 never in HBM)
 *   "he_fused"    [1]   land-use head feature backward with its K = 32 product inside the kernel (no dFE tensor)
 *   "side_stream" [1]   per-sample chains + grouped per-sample weight gradients on an engine-owned side stream
 *   "fwd_h_hbm"   [1]   forward of graphs too big for two workgroups per CU keeps H in HBM instead of LDS
 *   "fe_half"     [1]   land-use head's first Linear on the candidate messages m alone (no m*c tensor; hidden = 32, D % 32 == 0, D <= 256)
 *   "pq_exp"      [1]   P/Q GEMMs of layers 2..L store 2^(C2 x) block by block (flag bytes), the message-passing kernels stage
 *                       their slices by LDS-DMA; 0 = plain P/Q, register-staged slices
 *   "bwd_nb_global" [1] backward of graphs too big for two workgroups per CU walks the neighbour ids from global memory
 *   "fold_layer1" = 2   fold only where every graph of the minibatch fits half the LDS
 *   "nt_min_wgs"  [128] workgroups a gemm_nt launch must have before the 128-wide N tile is used (tests: 1)
 *   "tiny_fused"  [1]   models with D <= 32 run the fused one-workgroup-per-graph kernels (tiny.hip); 0 = the general path
 *   "tiny_threads" [1024] threads per workgroup of the fused small-model kernels (1024: 4 waves/SIMD | 512: no scratch)
 *   "side_heads"  [1]   land-use pointer-head chain (forward: first Linear; backward: softmax / feature / weight-gradient kernels) on the side stream
 *   "side_wgrad"  [1]   GCN weight-gradient GEMMs on a second side stream: 1 = for minibatches of <= 98304 nodes, 0 never, 2 behind the
 *                       layer's dgrad GEMM, 3 always
 *   "grad_buckets" [1]  upamd_backward finalises the gradient buffer range by range (upamd_grad_buckets); 0 = everything in the final flush
 *   "side_priority" [1] priority level of the side streams created from now on: 1 high, 0 normal, 2 low
 *   "gemm_lds_pad", "gemm_stagger_mode", "gemm_stagger_cycles": residency / first-round stagger of the LDS-DMA gemm_nt */
int upamd_tune(const char *name, int32_t value);
/* Lab hook: buf_dev = int64[32] (or NULL to switch off): the fused small-model kernel's workgroup 0 writes 100 MHz wall-clock stamps
 * at the section boundaries of its first graph (tools/r04_diag_tiny.py prints the section times). */
int upamd_tiny_profile(void *buf_dev);
/* Lab hook: one wave writes `samples` pairs (shader-clock counter, 100 MHz wall-clock counter) into out_dev (int64[2 * samples]),
 * `gap_ticks` wall-clock ticks apart; launched on a side stream it measures the effective shader clock under load. */
int upamd_clock_probe(void *out_dev, int32_t samples, int32_t gap_ticks, void *stream);

/* ------------------------------------------------------------------------------------------
 * Per-kernel timing of the dominant kernels (HIP events on the launch stream), for bench.py.
 * ------------------------------------------------------------------------------------------ */
int upamd_profile_enable(upamd_engine *eng, int32_t on);
/* name: kernel instance, e.g. "gemm_nt_128" | "gemm_nt_32" | "gemm_tn_128" | "gemm_tn_32" | "edge_fwd" | "edge_bwd"
 * (zero launches if that instance never ran).  Synchronises the recorded events. */
int upamd_profile_read(upamd_engine *eng, const char *name, int64_t *launches, double *total_ms,
                       double *total_flops, double *total_bytes);
int upamd_profile_reset(upamd_engine *eng);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* UPAMD_H */
