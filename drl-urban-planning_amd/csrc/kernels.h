// Launchers of the HIP kernels (gfx950).  Host-callable, asynchronous on `stream`.
//
// Data layouts
//   row-major   [rows][ld]                        -- per-sample tensors ([B, .]), weights
//   panel-major [cols/16][rows][16]  ("pm")       -- every per-node / per-candidate tensor.
//     A graph's 16-column slice is one contiguous run (n*64 B), which is what the edge kernels
//     stage into LDS and what the GEMM A-tile loads stream.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <map>
#include <string>

#include "upamd_internal.h"

namespace upamd {

// "Pair order" of the 2D P | Q columns of a GCN layer (rows of Wcat, columns of PQ / dPQ, entries of the bias partial sums):
// position c' = 32 p + 4 j + t holds   P column 16 p + 2 j + t (t = 0, 1)   or   Q column 16 p + 2 j + (t - 2) (t = 2, 3).
// In the panel-major PQ tensor the 16 floats of a node in panel 2p (2p + 1) are then exactly the LDS image the
// message-passing walk wants for column pairs 0..3 (4..7): [P_2j, P_2j+1, Q_2j, Q_2j+1] per 16-byte chunk, so the edge
// kernels stage a slice with plain 16-byte copies (LDS-DMA) and write dP | dQ of two columns with one 16-byte store.
__host__ __device__ inline int pq_col(int cp) { return (cp >> 5) * 16 + ((cp >> 2) & 7) * 2 + (cp & 1); }
__host__ __device__ inline int pq_side(int cp) { return (cp >> 1) & 1; }      // 0 = P (Wa), 1 = Q (Wb)
// position of (side, column)
__host__ __device__ inline int pq_pos(int side, int col) { return (col >> 4) * 32 + ((col >> 1) & 7) * 4 + side * 2 + (col & 1); }

// LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave instruction, lane i's 16 bytes land at the wave-uniform LDS base + 16 i;
// no staging VGPRs, no ds_write pass).  Completion is tracked by vmcnt only.
typedef __attribute__((address_space(1))) const void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
constexpr float PQ_C2 = 2.8853900817779268f;      // 2 * log2(e): tanh(x) = 1 - 2 / (1 + 2^(PQ_C2 x))
// The P/Q GEMM of a GCN layer may store its output in EXP FORM, 2^(PQ_C2 x), which is what the message-passing kernels
// stage into LDS anyway (edge.hip): a 64-row x 64-column output block whose largest |PQ_C2 x| is <= PQ_EXP_LIMIT is
// written as 2^(PQ_C2 x) with flag 0, any other block as plain x with flag 1.  flags: [2 * ceil(M / 128)][N / 64] bytes.
constexpr float PQ_EXP_LIMIT = 28.f;              // = edge.hip's EF_LIMIT_FWD: products of two factors and the bias stay finite

struct PackedView {   // device pointers into the packed replay (upamd_pack_layout)
    const int32_t *meta;
    const float *X;
    const uint8_t *nmask;
    const int32_t *rowptr;
    const uint16_t *inc_nbr, *he_src, *he_dst, *rn_node, *order, *hinc_nbr, *hinc_he, *he_sel;
    const int32_t *hinc_ptr;
    const uint8_t *he_live;
    const float *numerical, *cur, *xbar;
    int Fn;
};

struct MbView {       // one minibatch (device schedule arrays)
    int B;
    int64_t M, Nhe, Nrn;
    int max_n, max_inc, max_cand;
    const int32_t *idx, *node_off, *he_off, *rn_off;
    int64_t NI;                 // incidences (edge directions) of the minibatch; only used when num_edge_fc_layers > 1
    const int32_t *inc_off;     // [B+1] prefix sums of 2e (may be null when K = 1)
    // [B][16] row descriptors gathered once per forward: columns 0..13 = the state's meta row, 14 = node_off[b],
    // 15 = he_off[b].  One scalar load per workgroup instead of the idx -> meta -> offsets chain.
    const int32_t *rows;
};

struct KernelStat {
    int64_t launches = 0;
    double flops = 0, bytes = 0;
    std::vector<hipEvent_t> ev;   // start/stop pairs
};

struct Profiler {
    bool on = false;
    std::map<std::string, KernelStat> stats;   // keyed by kernel instance name ("gemm_nt_128", "edge_fwd", ...)
};
// HIP-event bracket around one launch on `st` (no-ops unless prof->on); returns 1 if recording
int prof_begin(Profiler *prof, const char *name, hipStream_t st, double flops, double bytes);
// A launch that asks for more than 64 KB of dynamic LDS needs hipFuncAttributeMaxDynamicSharedMemorySize raised first; the
// largest value already set is remembered per (device, kernel), so the runtime call happens once, not per launch.
int ensure_dynamic_lds(const void *kernel, int64_t lds_bytes);
void prof_end(Profiler *prof, const char *name, hipStream_t st, int began);

// ---- gemm.hip --------------------------------------------------------------------------
// C[M,N](pm) = act( A[M,K](pm) * W[N,K]^T (row-major, ld = K) + bias[N] + R[M,N](pm) )
int launch_gemm_nt(const float *A, int64_t M, int K, const float *W, int N, const float *bias, const float *R,
                   float *C, int act_tanh, hipStream_t st, Profiler *prof);
// general form: A / C (and R) either panel-major or row-major (ld in floats), W row-major with ld = ldw
struct GemmNT {
    const float *A; int64_t M; int K; int64_t lda; bool a_rm;
    const float *W; int N; int64_t ldw;
    const float *bias; const float *R;
    float *C; int64_t ldc; bool c_rm;
    int act_tanh; float alpha;
    bool w_kn = false;          // W is given as [K][N] (row stride ldw) instead of [N][K]: C = A W, no transpose needed
    // exp-form store (see PQ_EXP_LIMIT): honoured only by the LDS-DMA kernel with 64 x 64 wave blocks and only without
    // bias / residual / activation; launch_gemm_nt_ex reports through *exp_used whether the flags were written
    uint8_t *exp_flags = nullptr;
    bool *exp_used = nullptr;
};
int64_t gemm_exp_flag_bytes(int64_t M, int N);
bool gemm_nt_exp_store_ok(const GemmNT &g);      // a launch of g with exp_flags set WILL write them (same choice as the launcher)
bool gemm_nt_mfma_ok(const GemmNT &g);
void set_gemm_nt_dma_variant(int v);
void set_gemm_nt_min_wgs(int v);            // tune knob "nt_min_wgs": workgroups a launch must have before the 128-wide N tile is used
void set_gemm_stagger(int mode, int cycles);
void set_gemm_lds_pad(int bytes);   // first-residency-round stagger of the GEMM workgroups (-1 = keep)       // kernel-lab knob: LDS-DMA configuration of the plain panel-major launches
int launch_gemm_nt_ex(const GemmNT &g, hipStream_t st, Profiler *prof);
// opt-in: the same product on the bf16 matrix pipe by exact 3-way operand splitting (gemm_split.hip)
void set_gemm_nt_split(int nprod);          // tune knob: 0 (default, exact fp32 MFMA) | 6 | 9 partial products
bool gemm_nt_split_ok(const GemmNT &g);
int64_t gemm_nt_split_scratch_bytes(int N, int K);
int launch_gemm_nt_split(const GemmNT &g, void *planes, int nprod, hipStream_t st, Profiler *prof);
struct GemmTN {
    const float *A; int I; int64_t lda;
    const float *Bm; int J; int64_t ldb;
    int64_t M; bool in_rm;      // both inputs panel-major, or both row-major [M][ld]
    float *slabs;
};
bool gemm_tn_mfma_ok(const GemmTN &g);
int launch_gemm_tn_ex(const GemmTN &g, int *S_out, hipStream_t st, Profiler *prof);
// slabs[S][I][J] = partial sums over row chunks of A[M,I](pm)^T * Bm[M,J](pm);  returns S via *S_out
int tn_splits(int I, int J, int64_t M);
bool tn_shape_mfma_ok(int I, int J);       // the tiled split-K kernel handles I % 128 == 0, J % 32 == 0; other shapes go to gtn_kernel
int launch_gemm_tn(const float *A, int I, const float *Bm, int J, int64_t M, float *slabs, int *S_out,
                   hipStream_t st, Profiler *prof);
// dst (+)= sum_s slabs[s]:  mode 0: dst[i*ldd + j], j < jkeep;  mode 1: dst[j*ldd + i];
// mode 2: GCN un-permute, slab row j' (P/Q pair order) -> dst[pq_col(j')*ldd + pq_side(j')*J + k]
int launch_reduce_slabs(const float *slabs, int S, int I, int J, int mode, int jkeep, float *dst, int ldd,
                        hipStream_t st, float *last_col_dst = nullptr);

// ---- graph.hip -------------------------------------------------------------------------
__host__ __device__ int64_t edge_lds_bytes(int max_n, int max_inc, bool bwd, bool last, bool stage, bool hlds = true, bool nblds = true);
void set_bwd_nb_global(int on);            // tune knob "bwd_nb_global" (default on): large size class of the backward walks the neighbour ids from global memory
// First GCN layer folded into the message-passing stage-in (edge.hip: fold_fill): the workgroup computes its P/Q (and
// H_0) slice from the raw node features.  Xp: panel-major [2][M][16]; W1c = Wcat_1 We [2D][32] (rows in P/Q pair
// order), b1c [2D]; We (zero-padded) [D][32], be [D].
struct FoldArgs {
    const float *Xp, *W1c, *b1c, *We, *be;
};
bool edge_fold_ok(const MbView &mb);        // every graph of the minibatch fits the staged (LDS-resident) size classes
bool edge_fold_pays(const MbView &mb);      // ... and fits half the LDS (two workgroups per CU), where the fold beats the K = 32 GEMMs
void set_fwd_h_hbm(int on);                // tune knob: large-graph size class of the forward with H left in HBM (default on)
void set_grad_buckets(int on);             // tune knob "grad_buckets" (default 1): the backward finalises the gradient buffer range by range (engine.hip)
void set_side_wgrad(int on);               // tune knob "side_wgrad" (default 1 = adaptive: on for minibatches of <= 98304 nodes; 0 never, 2 behind the dgrad, 3 always): GCN weight-gradient GEMMs on a side stream
void set_side_priority(int v);             // tune knob "side_priority": priority level of the side streams created from now on (1 high, 0 normal, 2 low)
void set_side_heads(int on);               // tune knob "side_heads" (default on): the land-use pointer-head chain on the side stream
void set_side_stream(int on);              // tune knob "side_stream" (default on): per-sample chains + grouped weight gradients on an engine-owned side stream
void set_pq_exp(int on);                   // tune knob "pq_exp" (default on): exp-form P/Q from the GEMM epilogue + LDS-DMA stage-in
void set_fold_layer1(int on);              // tune knob: compute the first GCN layer inside the message-passing kernels
int launch_edge_fwd(const PackedView &pk, const MbView &mb, int D, bool last, const float *PQ, const float *bias,
                    const float *Hin, float *Hout, float *hbarV, float *hbarE, const float *Ccur, float *FE,
                    hipStream_t st, Profiler *prof, const FoldArgs *fold = nullptr, int fe_full = 1, const uint8_t *pqflag = nullptr);
int launch_edge_bwd(const PackedView &pk, const MbView &mb, int D, bool last, const float *PQ, const float *bias,
                    const float *G, const float *dhbarE, int ld_dhbarE, const float *dMhe, float *dPQ,
                    float *dbias_part, hipStream_t st, Profiler *prof, const FoldArgs *fold = nullptr, const uint8_t *pqflag = nullptr);
int launch_attn_fwd(const PackedView &pk, const MbView &mb, int D, int heads, const float *HL, const float *r,
                    float *alpha, float *s, hipStream_t st);
int launch_attn_bwd(const PackedView &pk, const MbView &mb, int D, int heads, const float *HL, const float *r,
                    const float *alpha, const float *s, const float *ds, const float *dhbarV, int ld_dhbarV, float *GL, float *dr,
                    hipStream_t st);
int launch_he_feat_bwd(const PackedView &pk, const MbView &mb, int D, const float *FE, const float *C,
                       const float *dFE, float *dMhe, float *dC_head, hipStream_t st, int keep_dead = 0);
// the same with dFE = dpre W1f (h0 == 32) computed inside the kernel on the matrix cores: no dFE tensor in HBM
bool he_feat_bwd_fused_ok(int D, int h0);
void set_he_feat_fused(int on);            // tune knob "he_fused" (default on)
int launch_he_feat_bwd_fused(const PackedView &pk, const MbView &mb, int D, const float *FE, const float *C, const float *dprel,
                             const float *W1fT, float *dMhe, float *dC_head, hipStream_t st, int keep_dead = 0);
// ---- head.hip: the land-use head's first Linear on the candidate messages m alone (FE keeps only its m half: the m*c half
// is a per-graph change of the weight).  fe_half_ok: h0 == 32, D % 32 == 0, D <= 256 (tune knob "fe_half", default on)
bool head_fe_half_ok(int D, int h0);
void set_fe_half(int on);
int head_wgrad_groups(int B);
int launch_head_hidden_fwd(const PackedView &pk, const MbView &mb, int D, const float *FE, const float *C, const float *W1f,
                           const float *constb, float *hid, hipStream_t st);
// slab[G][2D][32]: per-workgroup shares of dpre^T m (rows 0 .. D-1) and of its c_b-weighted version (rows D .. 2D-1)
int launch_head_wgrad(const PackedView &pk, const MbView &mb, int D, const float *FE, const float *C, const float *dprel,
                      float *slab, int *S_out, hipStream_t st);
// rl-mlp encoder (state_encoder.py:284-308): masked node mean of H^0 and the land-use head inputs gathered from H^0 rows
int launch_mlp_pool_fwd(const PackedView &pk, const MbView &mb, int D, const float *H0, const float *be, const float *C,
                        float *hbarV, float *FE, hipStream_t st, int fe_full = 1);
// ... and its backward: G^0 = dhbarV / n_mask on masked nodes + the candidates' dM routed to their selected endpoint;
// candidates that are not live edges (m == bias) contribute to dbe_extra[b][D]
int launch_mlp_pool_bwd(const PackedView &pk, const MbView &mb, int D, const float *dhbarV, int ld, const float *dMhe, float *G0,
                        float *dbe_extra, hipStream_t st);
int launch_he_bias_rows(const PackedView &pk, const MbView &mb, int h0, const float *constb, float *hid, hipStream_t st);
int launch_road_gather(const PackedView &pk, const MbView &mb, int D, const float *HL, float *XR, hipStream_t st);
int launch_road_scatter_add(const PackedView &pk, const MbView &mb, int D, const float *dXR, float *GL, hipStream_t st);
// masked softmax over each row's candidate list: logp, entropy (+ probabilities kept for backward)

// ---- deep_edge.hip: edge MLPs with K > 1 sub-layers (per-incidence tensors, panel-major [D/16][NI][16]) -----------------
int launch_inc_index(const PackedView &pk, const MbView &mb, int32_t *gsrc, int32_t *gdst, int32_t *grev, int32_t *cand_inc,
                     hipStream_t st);
int launch_inc_gather_fwd(const MbView &mb, int D, const float *PQ, const float *bias, const int32_t *gsrc, const int32_t *gdst,
                          float *A1, hipStream_t st);
int launch_inc_scatter_fwd(const PackedView &pk, const MbView &mb, int D, bool last, const float *AK, const int32_t *grev,
                           const int32_t *cand_inc, const float *Hin, float *Hout, float *hbarV, float *hbarE,
                           const float *Ccur, float *FE, hipStream_t st, int fe_full = 1);
int launch_inc_seed_bwd(const PackedView &pk, const MbView &mb, int D, bool last, const float *AK, const float *G,
                        const float *dhbarE, int ld_dhbarE, const float *dMhe, float *dpre, float *dbpart, hipStream_t st);
int launch_inc_tanh_bwd(const PackedView &pk, const MbView &mb, int D, const float *A, float *dA, float *dbpart, hipStream_t st);
int launch_inc_scatter_bwd(const PackedView &pk, const MbView &mb, int D, const float *dpre1, const int32_t *grev, float *dPQ,
                           float *dbias_part, hipStream_t st);

// ---- chain.hip: fused small kernels (job tables are passed BY VALUE as kernel arguments, <= 4 KB each) ---------------
enum { PERM_PAD_COLS = 0, PERM_TRANSPOSE, PERM_WCAT, PERM_LAND_HEAD, PERM_LAND_SCATTER };
constexpr int PERM_MAX_JOBS = 32, SMM_MAX_JOBS = 20, TN_MAX_JOBS = 40, RED_MAX_JOBS = 56;
struct PermJob { const float *src, *src2; float *dst, *dst2, *dst3, *dst4; int kind, rows, cols, aux, blk_begin, pad_; };
struct PermJobs { int n = 0; PermJob j[PERM_MAX_JOBS]; };
int perm_add(PermJobs *P, int *blocks, int kind, const float *src, const float *src2, float *dst, float *dst2, float *dst3,
             float *dst4, int rows, int cols, int aux);
int launch_permute(const PermJobs &P, int blocks, hipStream_t st);

struct SmmJob { const float *A, *B, *bias, *u, *v; float *C, *CT; int64_t sa0, sa1, sb0, sb1, ldc, ldct; int I, J, K, accumulate;
                float scale; int blk_begin, tiles_j, pad_; };
struct SmmJobs { int n = 0; SmmJob j[SMM_MAX_JOBS]; };
// C[i*ldc + j] (=|+=) scale * (sum_k A[i*sa0 + k*sa1] B[k*sb0 + j*sb1] + bias[j]) + u[i] v[j];  CT (optional) = C^T
int smm_add(SmmJobs *P, int *blocks, int I, int J, int K, const float *A, int64_t sa0, int64_t sa1, const float *B, int64_t sb0,
            int64_t sb1, const float *bias, float *C, int64_t ldc, int accumulate, float scale, float *CT = nullptr,
            int64_t ldct = 0, const float *u = nullptr, const float *v = nullptr);
int launch_gsmm(const SmmJobs &P, int blocks, hipStream_t st);

struct ChainDims {
    int B, D, heads, dh, F, Fn, n_num, num_hidden[UPAMD_MAX_MLP], n_value, value_hidden[UPAMD_MAX_MLP];
    int S_last, W, Wp, h0l, maxnum, maxval, maxdim;
    int mlp;            // rl-mlp encoder: no attention path; state_value = [h_num ; mean nodes ; mean edges ; stage]
    float scale;        // 1 / sqrt(D / heads)
};
struct ChainFwdPre {
    ChainDims d; PackedView pk; MbView mb;
    int32_t *rows; float *Xp;
    const float *WnT[UPAMD_MAX_MLP], *bn[UPAMD_MAX_MLP];      // [K][N] transposed weights
    const float *WeT, *be, *WqT, *bq, *WiqT, *biq, *Wkk, *WbdT, *b1l;
    float *U[UPAMD_MAX_MLP + 1], *curg, *C, *q0, *q1, *r, *constb;
    float *hbarE;        // mlp: We xbar_E + be  (mean over the live edges of the encoded selected endpoint)
    // what this launch covers: CHAIN_ALL, or the two halves of a forked forward -- CHAIN_GATHER (row descriptors + node-feature
    // gather: what the graph part needs first) and CHAIN_LAYERS (the per-sample layers, needed only from the last GCN layer on)
    int part;
};
enum { CHAIN_ALL = 0, CHAIN_GATHER = 1, CHAIN_LAYERS = 2 };
struct ChainFwdPost {
    ChainDims d; const int32_t *rows;
    const float *s, *hbarV, *hbarE, *Ulast, *WvvT, *bvv, *WoT, *bo, *WvT[UPAMD_MAX_MLP], *bv[UPAMD_MAX_MLP];
    float *o, *att, *SV, *V[UPAMD_MAX_MLP + 1], *value;
};
struct ChainBwdPost {
    ChainDims d;
    const float *dvalue, *V[UPAMD_MAX_MLP + 1], *U[UPAMD_MAX_MLP + 1];
    const float *Wv[UPAMD_MAX_MLP], *Wn[UPAMD_MAX_MLP], *Wo, *Wvv;      // [N][K] weights as stored
    float *dAv[UPAMD_MAX_MLP], *dAn[UPAMD_MAX_MLP], *dSV, *datt, *dov, *ds;
};
struct ChainBwdPre {
    ChainDims d;
    const float *dr, *dconst, *dC_head, *WkkT, *Wiq, *Wq, *Wbd;
    const float *dSV;    // mlp: dhbarE (a column slice of dSV) is copied behind dC: rows [B, 2B) of the node-encoder job
    float *dq1, *dq0, *dC;
};
int launch_chain_fwd_pre(const ChainFwdPre &a, hipStream_t st);
int launch_chain_fwd_post(const ChainFwdPost &a, hipStream_t st);
int launch_chain_bwd_post(const ChainBwdPost &a, hipStream_t st);
int launch_chain_bwd_pre(const ChainBwdPre &a, hipStream_t st);

struct TnJob { const float *A, *X; float *slab; int64_t lda, ldx, rows_total; int N, K, rows, tiles_n, tiles_k, splits, chunk, wave_begin,
               a_pm, x_pm; };
struct TnJobs { int n = 0; int total_waves = 0; TnJob j[TN_MAX_JOBS]; };
int tn_job_splits(int64_t rows);
// slab[split][n][k] = sum_rows A[row*lda + n] * X[row*ldx + k]   (X == nullptr: ones, K = 1)
// a_pm / x_pm: the operand is panel-major [cols/16][rows][16] (ld ignored) instead of row-major
int tn_add(TnJobs *P, const float *A, int64_t lda, int N, const float *X, int64_t ldx, int K, int64_t rows, float *slab, int *S_out,
           int a_pm = 0, int x_pm = 0);
int launch_gtn(const TnJobs &P, hipStream_t st);

struct RedJob { const float *slab; float *dst, *dst2; int64_t sstride; int S, I, J, mode, jkeep, ldd, overwrite, blk_begin; };
struct RedJobs { int n = 0; RedJob j[RED_MAX_JOBS]; };
int red_add(RedJobs *P, int *blocks, const float *slab, int S, int64_t sstride, int I, int J, int mode, int jkeep, float *dst,
            int ldd, float *dst2 = nullptr, int overwrite = 0);
int launch_greduce(const RedJobs &P, int blocks, hipStream_t st);

int launch_pointer_fwd2(const PackedView &pk, const MbView &mb, const float *hidl, const float *w2l, int h0l, const float *hidr,
                        const float *w2r, int h0r, float *z_he, float *z_rn, float *p_he, float *p_rn, float *logp, float *ent,
                        float *lse, float *ent_keep, hipStream_t st);
int launch_pointer_bwd2(const PackedView &pk, const MbView &mb, const float *z_he, const float *z_rn, const float *p_he,
                        const float *p_rn, const float *ent, const float *lse, const float *dlogp, const float *dent,
                        const float *hidl, const float *w2l, int h0l, const float *hidr, const float *w2r, int h0r, float *dz_he,
                        float *dz_rn, float *dprel, float *dprer, float *rs_dzh_l, float *rs_dpre_l, float *rs_dzh_r,
                        float *rs_dpre_r, hipStream_t st);

// ---- dense.hip -------------------------------------------------------------------------
int launch_ppo_loss(int B, const float *value, const float *logp, const float *ent, const int64_t *rows, const float *adv,
                    const float *ret, const float *old_logp, const float *exps, float clip_eps, float cv, float ce,
                    float inv_rows, float inv_ind, float *dvalue, float *dlogp, float *dent, float *losses,
                    float *zero, int64_t nzero, hipStream_t st);
int launch_select_actions(int B, const int32_t *meta, const int32_t *he_slot, const uint16_t *rn_node, const int32_t *idx,
                          const int32_t *he_off, const int32_t *rn_off, const float *z_he, const float *z_rn,
                          const uint8_t *greedy, const float *uniform, float *actions, hipStream_t st);
int launch_gae(int64_t T, const float *rewards, const float *masks, const float *values, double gamma, double tau,
               float *adv, float *ret, hipStream_t st);
int launch_adam(int64_t n, float *p, const float *g, float *m, float *v, int step, double lr, double b1, double b2,
                double eps, double wd, hipStream_t st);
int launch_adam_groups(int n_groups, const int64_t *begin, const int64_t *end, const int32_t *step, float *p, const float *g,
                       float *m, float *v, double lr, double b1, double b2, double eps, double wd, const float *loss_src,
                       float *loss_dst, hipStream_t st);
int launch_sumsq(const float *x, int64_t n, float *scratch, float *out_accum, hipStream_t st);
int launch_clip_scale(float *g, int64_t n, const float *sumsq, float max_norm, hipStream_t st);

// ---- tiny.hip: fused small-model path (D <= 32; one workgroup per graph, the whole network out of LDS) --------------------
struct TinyIO {
    int mode;                                   // 0 forward, 1 backward from seeds (forward recomputed), 2 forward + PPO loss + backward
    float *value, *logp, *ent, *z_he, *z_rn;
    const float *dvalue, *dlogp, *dent;
    const int64_t *rows;
    const float *adv, *ret, *old_logp, *exps;
    float clip_eps, cv, ce, inv_rows, inv_ind;
    float *loss_rows, *slab, *scratch;          // workspace: [B][4], [groups][slab_stride], [groups][scratch_stride]
    float *grads;                               // flat gradient buffer (accumulate != 0: added to, else overwritten)
    int accumulate;
    float *losses;                              // mode 2: the four loss scalars
};
void set_tiny_fused(int on);                    // tune knob "tiny_fused" (default on)
void set_tiny_prof(void *dev_buf);              // lab hook: int64[32] section time stamps (100 MHz) of workgroup 0's first graph
void set_tiny_threads(int n);                   // tune knob "tiny_threads": 1024 (default) | 512 threads per workgroup
bool tiny_supported(const upamd_model_desc &d, int max_n, int max_inc, int max_cand);
int tiny_groups(int B);
int64_t tiny_scratch_stride(const upamd_model_desc &d, int max_cand);
int64_t tiny_slab_stride(const ParamLayout &P);
int launch_tiny(const upamd_model_desc &d, const ParamLayout &P, const PackedView &pk, const MbView &mb, const float *prm,
                const TinyIO &io, hipStream_t st);

}  // namespace upamd
