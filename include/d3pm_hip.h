/* d3pm_hip.h -- C ABI of the MI355X (gfx950) D3PM codec-token sampler.
 *
 * The upstream project has no FFI layer: its boundary is the Python module API
 * (vall_e.vall_e.get_model / AR.generate_audio).  This header is the C-ABI that
 * sits *below* that Python surface; each entry point names the reference statements it replaces
 * (paths relative to /root/reference/vall_e/vall_e/).
 *
 * Conventions
 *   - the library keeps NO mutable process-wide state (thread-compatible): schedule choices travel in `d3pm_tuning`
 *     through the shape structs, timing hooks in a caller-owned `d3pm_prof`; the only thread-local is the error string;
 *   - every pointer marked "device" is HBM memory owned by the caller (e.g. a torch tensor's
 *     data_ptr()); nothing is allocated, freed or synchronised inside the library;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*), so calls are capturable
 *     in a hipGraph and ordered with the caller's other work on that stream;
 *   - return value: 0 = ok, <0 = error (see D3PM_E_*); nothing throws across the ABI;
 *     d3pm_last_error() gives a thread-local message for the last failure;
 *   - `dtype` selects storage *and* arithmetic of the denoiser: F32 (exact-f32 FMA path, the
 *     1e-3 logits-parity mode), F16 (the only dtype the reference sampler runs in) or BF16.
 *     Accumulation is always fp32; every op output is rounded to `dtype` where the reference's
 *     eager model rounds.  The posterior/sampling arithmetic is fp16 in every mode, as upstream
 *     (its tables are hard-cast to fp16, ar_discrete.py:257-277).
 *   - token ids are int32, row-major [batch][canvas]; activations are [batch*canvas][d_model].
 */
#ifndef D3PM_HIP_H
#define D3PM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: only what this header declares is exported */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

/* 2: + d3pm_ce_loss_rows, the fp8 entry points (d3pm_*_fp8), D3PM_FLAG_SEED_IN_HBM, tuning knobs 2..3, GEMM variant 5
 *    (additions only) */
/* 3: + d3pm_op_final_sample, d3pm_op_cond_embed, the fp32 training ops (d3pm_op_*_f32), d3pm_op_linear_rowpanel / _lnpro,
 *    d3pm_prof_read_class, d3pm_debug_gemm_clock, tuning knobs 4..11, GEMM variants 6..8; the kernel class D3PM_K_GEMM_LN took
 *    the value 4, so "every class" (D3PM_K_COUNT) is now 5 */
/* 4: the tuning knobs moved from process-wide state (d3pm_set_tuning) into `d3pm_tuning`, reached through the shape structs
 *    (and a trailing argument of d3pm_op_linear / d3pm_op_attention); the profiling hooks became a handle (d3pm_prof);
 *    the library keeps no mutable global state besides the thread-local error string.  The experiment-only entry points
 *    (d3pm_set_tuning's ablation arms, d3pm_op_final_sample, d3pm_op_linear_lnpro, d3pm_debug_gemm_clock) left the product:
 *    they live in libd3pm_hip_ab.so, include/d3pm_hip_ab.h. */
/* 5: + d3pm_op_attention_pair; new VALUES of existing tuning fields (row_panel bit 3, attn_query_groups 4, attn_cross_resident
 *    4 / 5); layouts unchanged */
/* 6: LayerNorm folded into the projections (d3pm_fold_block, d3pm_weights.fold, d3pm_fold_bytes / d3pm_fold_build,
 *    d3pm_op_linear_fold / d3pm_op_linear_stats / d3pm_op_row_stats); d3pm_tuning gained regime_batch and ln_fold (layout change);
 *    d3pm_workspace_bytes grew by the row-moment buffer and the fp8 path's scale slot */
#define D3PM_ABI_VERSION 6

enum { D3PM_F32 = 0, D3PM_F16 = 1, D3PM_BF16 = 2 };

enum {
  D3PM_OK = 0,
  D3PM_E_ARG = -1,        /* null pointer / inconsistent sizes                     */
  D3PM_E_WORKSPACE = -2,  /* workspace smaller than d3pm_workspace_bytes()         */
  D3PM_E_HIP = -3,        /* a HIP runtime call or kernel launch failed            */
  D3PM_E_SHAPE = -4       /* shape not supported by any kernel in this build       */
};

/* flags for d3pm_sample_loop / d3pm_denoise_step */
enum {
  D3PM_FLAG_GREEDY = 1,        /* argmax of the posterior without Gumbel noise   */
  D3PM_FLAG_FORCE_GENERIC = 2, /* never take the MFMA kernels (cross-check / debugging)          */
  D3PM_FLAG_SEED_IN_HBM = 4    /* d3pm_sample_loop: `seed` is the address of a uint64 in HBM, read by the sampling
                                  kernel at run time -- a captured HIP graph of the loop can then be replayed with a
                                  new seed (measured: no faster than eager launches, 66.6 vs 66.3 ms per utterance) */
};

/* Schedule choices.  A NULL `tuning` pointer anywhere means d3pm_tuning_default().  The struct is read during the call and never
 * retained; two threads may use different tunings at the same time.  The GEMM / workspace / row-panel fields select between
 * schedules that give bit-identical results.  Three choices change the ORDER of a floating-point accumulation, i.e. results that
 * agree to rounding noise, not bit for bit: the attention instruction shape (attn_query_groups 32 / 33, attn_cross_resident 4 / 5,
 * and -- in auto mode -- the batch regime, see regime_batch) and ln_fold.  Within one setting of those a result never depends on
 * the batch an utterance rides in, nor on how a batch is split over ranks or streams.
 *   gemm_variant       0 = auto (default): the big-tile persistent schedule when the shape divides into whole tiles that fill >= 85 %
 *                          of their rounds over the CUs -- 192 x 128 tiles (four waves, two workgroups per CU) first, then 192 x 256,
 *                          then 96 x 512 (eight waves, one workgroup per CU); 192 x 128 also for mid-size batches that give at
 *                          least every (long K: every second) CU a tile; else the latency schedule (64 x 64 tiles, whole-K operand
 *                          panels in flight) for M <= 1536 rows, the 128 x 128 throughput schedule otherwise;
 *                      2 = always the throughput schedule (one LDS stage, 4 workgroups per CU; persistent over the 128 x 128
 *                          tiles when M and N are multiples of 128 and there are >= 512 tiles, else as 5);
 *                      3 = always the round-1 latency schedule (128 x 128 tiles, two stages, asm DMA prefetch);
 *                      4 = always the latency schedule (64 x 64 tiles, every DMA piece of K <= 512 in flight at once);
 *                      5 = throughput schedule with one tile per workgroup (the non-persistent form of 2);
 *                      6 / 7 / 8 = as auto, but only the 192 x 256 / 96 x 512 / 192 x 128 (two 4-wave workgroups per CU) big
 *                          tile is considered, for any shape made of whole tiles.
 *   gemm_persist_slots resident workgroups of the persistent 128 x 128 schedule, a multiple of 8 (default 1024 = 4 per CU).
 *   lat_tile           tile of the latency GEMM: 0 = auto (fewest rounds over the 256 CUs, then most workgroups),
 *                      1 / 2 / 3 = always 64 x 64 / 96 x 64 / 32 x 64.
 *   attn_query_groups  schedule of the MFMA self-attention: 0 = auto -- the software-pipelined 32 x 32 x 16 kernel (one 32-query
 *                      group per wave, three workgroups per CU) for whole 128-query blocks and 64-key tiles once that grid has two
 *                      workgroups per CU, else the 16 x 16 x 32 kernel with 2 sixteen-query groups per wave when that still gives
 *                      >= 4 workgroups per CU, else 1; 1 or 2 = the 16 x 16 x 32 kernel with that many groups; 4 = the key-split
 *                      kernel (four waves share 32 queries and each walks every fourth 64-key tile, the partial softmax states
 *                      combined at the end; measured 2 % faster at one utterance and slower from two on, so never automatic);
 *                      32 = the 32 x 32 x 16 kernel wherever it applies; 33 = as 32 without the software pipeline.  (The schedules
 *                      accumulate in different orders: results agree to rounding noise, not bit for bit; within one schedule a
 *                      result does not depend on the batch it rides in.)
 *   attn_pair_sequential  a paired attention launch (text + prompt cross-attention) on the tile-by-tile kernel: 2 = every
 *                      workgroup runs both problems one after the other; 0 = the second half of the grid takes problem 2;
 *                      1 (default) = auto: sequential while that still leaves >= 2 workgroups per CU.
 *   attn_cross_resident  the cross-attention pair of a block (<= 64 and <= 256 keys) with every K / V tile of both problems
 *                      fetched into LDS once per (utterance, head) by a workgroup of eight waves that then walks that head's
 *                      256-query blocks: 2 = always, 0 = never (tile-by-tile kernel), 1 (default) = auto: when there is at least
 *                      one such workgroup per CU (batch >= 11 at 768 rows and 8 heads); 3 = as 2 with one query block per workgroup;
 *                      4 / 5 = as 2 on the 16 x 16 x 32 / the 32 x 32 x 16 instruction.  The resident kernel of 1 / 2 / 3 is the
 *                      32 x 32 x 16 one (the software-pipelined 32-key block of the self-attention kernel); the 16 x 16 x 32 form
 *                      (4) is bit-identical to the tile-by-tile kernel, the two instruction shapes accumulate in different orders.
 *   row_panel          bit mask of the block's projections that run as row-panel launches (d3pm_op_linear_rowpanel) when
 *                      d_model = 512, the dtype is 16-bit and batch * canvas is a multiple of 96: 1 = self-attention
 *                      out-projection + norm2 | norm22, 2 = both cross-attention out-projections + norm3 / FiLM, 4 = fc2 + the
 *                      next block's norm1; 8 = at one or two utterances (batch * canvas <= 2048 rows, where no row panel applies)
 *                      both cross-attention out-projections as ONE launch of the latency GEMM: two products through one resident
 *                      weight panel, the first result kept in registers (same bits as the two launches it replaces).
 *                      Default 10: in the sampler's loop the fc2 form (4) is slower fused, and since the plain projections run as two
 *                      4-wave workgroups per CU the self-attention form (1) is too (102.6 k vs 103.7 k tokens/s, round 3); the dual
 *                      cross-attention form (2) and the latency form (8) pay.
 *   workspace_alias    1 (default) = the packed qkv rows, the MLP hidden rows and the logits of an iteration share one
 *                      workspace region (never live together); 0 = separate regions.  d3pm_workspace_bytes and the step /
 *                      loop calls must see the same value.
 *   regime_batch       the batch size the AUTOMATIC attention choices (attn_query_groups = 0, attn_cross_resident = 1) are made
 *                      for: 0 (default) = the batch of the call itself; > 0 = as if the call carried that many utterances.  The two
 *                      instruction shapes accumulate in different orders, so a caller that splits one logical batch over ranks
 *                      or streams (vall_e/vall_e/dp.py, AR.generate_audio(streams=)) passes the GLOBAL batch here and every shard
 *                      then takes the kernels -- and produces the bits -- of the unsplit batch.
 *   ln_fold            1 (default) = when d3pm_weights.fold is given, the LayerNorm-fed projections (norm1 -> QKV, norm2 | norm22 ->
 *                      cross-attention queries, norm3 + FiLM -> fc1) read the raw residual stream and apply the LayerNorm in their
 *                      epilogue from row moments the producing projection left behind (d3pm_fold_block): no LayerNorm launch and
 *                      no row-panel launch in a block (one small launch per evaluation rebuilds fc1's FiLM-folded weights for
 *                      the timestep); 0 = the stand-alone / row-panel LayerNorm launches (the eager rounding points).  Ignored (= 0) for fp32 models, D3PM_FLAG_FORCE_GENERIC and the fp8 entry points.
 *   prof               optional timing hooks (d3pm_prof_create below), NULL = none. */
struct d3pm_prof;
typedef struct d3pm_tuning {
  int32_t gemm_variant, gemm_persist_slots, lat_tile;
  int32_t attn_query_groups, attn_pair_sequential, attn_cross_resident;
  int32_t row_panel, workspace_alias;
  int32_t regime_batch, ln_fold;
  struct d3pm_prof *prof;
} d3pm_tuning;
void d3pm_tuning_default(d3pm_tuning *t);

typedef struct d3pm_shape {
  int32_t d_model;    /* 32 upstream (ar_discrete.py:208); 512 asked by get_model (__init__.py:26) */
  int32_t n_heads;    /* 16 upstream (ar_discrete.py:238)                                          */
  int32_t n_layers;   /* 8 upstream  (ar_discrete.py:238)                                          */
  int32_t canvas;     /* frames per utterance the denoiser sees, 448 upstream (:704)               */
  int32_t s_text;     /* phoneme keys, 50 upstream (:714)                                          */
  int32_t s_prompt;   /* prompt keys, 398 upstream (:726)                                          */
  int32_t n_classes;  /* 1025 (:255)                                                               */
  int32_t mask_id;    /* absorbing id 512 (:332)                                                   */
  int32_t timesteps;  /* 100 (:207); the loop runs t = timesteps-1 .. 1 (:750)                     */
  int32_t dtype;      /* D3PM_F32 / D3PM_F16 / D3PM_BF16                                           */
  int32_t n_q;        /* quantizer levels the D3PM generates jointly: 0 or 1 = level 0 only, as upstream (:699-709,
                         SURVEY.md section 0 #4).  n_q > 1 is this build's extension (SURVEY section 8d config 2,
                         BASELINE.json configs[1] "x 8 quantizers"; no reference counterpart): token grids are
                         [batch][canvas][n_q], a frame's input embedding is the sum of its n_q level embeddings
                         (resps_emb [n_q][n_classes][d], as MultiEmbedding sums the prompt levels, base.py:244-274), `final`
                         is [n_q * n_classes][d] and every (frame, level) is sampled like a level-0 token, level l > 0 on
                         Philox stream 16 + l.  With n_q = 1 every entry point is bit-identical to the level-0 path. */
  const d3pm_tuning *tuning;   /* schedule choices of the calls made with this shape, or NULL (defaults)   */
} d3pm_shape;

/* One DiT block's parameters (ar_discrete.py:103-124), device pointers, elements of `dtype`,
 * torch layouts: Linear weight [out][in]; MultiheadAttention in_proj [3d][d] (q|k|v rows).
 * cross_attn2.* is dead upstream (the prompt attention re-uses cross_attn, :142) and not passed. */
typedef struct d3pm_block_weights {
  const void *norm1_w, *norm1_b;          /* [d]               LayerNorm eps 1e-6 (:108)  */
  const void *attn_in_w, *attn_in_b;      /* [3d][d], [3d]     self-attention (:109)      */
  const void *attn_out_w, *attn_out_b;    /* [d][d], [d]                                  */
  const void *norm2_w, *norm2_b;          /* text-query LN (:112)                         */
  const void *norm22_w, *norm22_b;        /* prompt-query LN (:117)                       */
  const void *cross_in_w, *cross_in_b;    /* [3d][d], [3d]     cross_attn (:113)          */
  const void *cross_out_w, *cross_out_b;  /* [d][d], [d]                                  */
  const void *norm3_w, *norm3_b;          /* (:121)                                       */
  const void *fc1_w, *fc1_b;              /* [4d][d], [4d]     timm Mlp (:123)            */
  const void *fc2_w, *fc2_b;              /* [d][4d], [d]                                 */
  const void *tfc_w, *tfc_b;              /* [2d][d], [2d]     timestep_fc (:124)         */
} d3pm_block_weights;

/* A block's LayerNorms folded into the projections they feed (ar_discrete.py:131-132, 136-142, 145-159), built by
 * d3pm_fold_build from d3pm_block_weights; device pointers into the caller's `storage`.  For a projection
 * y = LN(x) W^T + b:  w = rn(W o gamma) in the model dtype,  s[n] = sum_k w[n][k] and b[n] = sum_k W[n][k] beta[k] + b[n] in fp32, so
 * that y = rstd_r (x_r . w^T - mean_r s) + b.  fc1 carries FiLM (gamma_k rn(1 + scale_t[k]), beta_k rn(1 + scale_t[k]) + shift_t[k]),
 * which depends on the timestep: its folded copy is rebuilt for t at the top of every denoiser evaluation, into the workspace
 * (one launch for all layers), and is not part of this struct. */
typedef struct d3pm_fold_block {
  const void *qkv_w;  const float *qkv_s, *qkv_b;     /* [3d][d]; [3d]                 norm1 -> attn in-projection                 */
  const void *q2_w;   const float *q2_s, *q2_b;       /* [2d][d]; [2d]                 norm2 -> q rows | norm22 -> the same q rows  */
} d3pm_fold_block;

typedef struct d3pm_weights {
  const void *resps_emb;                  /* [n_classes][d]  (:212); n_q > 1: [n_q][n_classes][d] */
  const void *time_emb;                   /* [timesteps+1][d] (:213)                      */
  const void *final_w, *final_b;          /* [n_classes][d], [n_classes] (:240); n_q > 1: [n_q * n_classes][d], [n_q * n_classes] */
  const d3pm_block_weights *blocks;       /* HOST array of n_layers entries               */
  const d3pm_fold_block *fold;            /* HOST array of n_layers entries (d3pm_fold_build), or NULL: stand-alone LayerNorms */
} d3pm_weights;

/* Block-scaled fp8 ("MX": OCP e4m3 codes + one e8m0 power-of-two scale per 32 elements along K) copies of a block's
 * projection weights for the fp8 fast path (BASELINE.json configs[4]; built by the caller, e.g. _hip.quantize_mx):
 *   codes  [N][K] one byte per element, row-major;
 *   scales [N][4][K / 128] bytes, value 2^(byte - 127): scales[n][g][s] applies to elements [128 s + 32 g, 128 s + 32 g + 32)
 *          of row n (g-major so that the matrix instruction's lane fetches four consecutive k-steps of its block as one dword;
 *          csrc/d3pm_mx.hip).
 * cross_in covers the q rows only ([d][d]).  fc2_w8 / fc2_scale may be NULL: then fc1 writes a 16-bit hidden layer and fc2
 * stays a 16-bit GEMM. */
typedef struct d3pm_fp8_block_weights {
  const void *attn_in_w8;  const void *attn_in_scale;    /* [3d][d], [3d][4][d/128]   */
  const void *cross_in_w8; const void *cross_in_scale;   /* [d][d],  [d][4][d/128]    */
  const void *fc1_w8;      const void *fc1_scale;        /* [4d][d], [4d][4][d/128]   */
  const void *fc2_w8;      const void *fc2_scale;        /* [d][4d], [d][4][4d/128] or NULL */
} d3pm_fp8_block_weights;

/* fp16 bit patterns of the absorbing-state schedule, HOST arrays.
 * betas[timesteps+1]; d,c,dbar,cbar[timesteps]:  Q_t = d_t I + c_t 1 e_M^T,  Qbar_t likewise. */
typedef struct d3pm_schedule {
  int32_t timesteps;
  const uint16_t *d, *c, *dbar, *cbar;
} d3pm_schedule;

int d3pm_abi_version(void);
const char *d3pm_last_error(void);

/* Replaces AR.cosine_beta_schedule + the 3 x [T,1025,1025] fp16 table build
 * (ar_discrete.py:257,268-277,286-304,315-334) by the 4 fp16 scalars per step they reduce to
 * (SURVEY.md §8a a14).  Pure host arithmetic; all outputs are HOST arrays of fp16 bit patterns. */
int d3pm_schedule_build(int timesteps, uint16_t *betas /*[timesteps+1]*/, uint16_t *d, uint16_t *c,
                        uint16_t *dbar, uint16_t *cbar /*[timesteps] each*/);

/* Bytes of device scratch the step/loop entry points need for `batch` utterances. */
size_t d3pm_workspace_bytes(const d3pm_shape *shape, int batch);

/* Step-invariant precomputation ----------------------------------------------------------- */

/* film[t][layer][2d] = timestep_fc_layer(time_emb[t])  (ar_discrete.py:145,752) for every t:
 * the FiLM scale/shift depends on weights and t only, so it is tabulated once per weight set. */
int d3pm_film_table(const d3pm_shape *shape, const d3pm_weights *w, void *film /*device*/,
                    void *stream);

/* LayerNorm folded into the projections: bytes of device storage for the tables of `shape` (0 when the shape does not qualify:
 * 16-bit dtype and d_model a multiple of 256), and the build; blocks_out is a HOST array of n_layers entries that receives
 * pointers into `storage` (pass it as d3pm_weights.fold).  Once per weight set; 5 d^2 elements + 10 d floats per layer. */
size_t d3pm_fold_bytes(const d3pm_shape *shape);
int d3pm_fold_build(const d3pm_shape *shape, const d3pm_weights *w, void *storage /*device*/, size_t storage_bytes,
                    d3pm_fold_block *blocks_out /*host*/, void *stream);

/* K/V projections of the step-invariant conditions with cross_attn's k/v rows
 * (the `k`/`v` halves of F.multi_head_attention_forward's in-projection for the calls at
 * ar_discrete.py:138,142).  cond_* are device [batch][S][d]; kv_* receive [n_layers][batch][S][2d]. */
int d3pm_cond_kv(const d3pm_shape *shape, const d3pm_weights *w, int batch, const void *cond_text,
                 const void *cond_prompt, void *kv_text, void *kv_prompt, void *stream);

/* Condition encoders (once per utterance) ------------------------------------------------------
 * Replaces the statements before the loop of AR.generate_audio (ar_discrete.py:736-746): text_emb / proms_emb
 * (base.py:244-274) gathers, the sinusoidal position add (:84-92 incl. the position-0 quirk for text), two
 * post-norm nn.TransformerEncoderLayer (nhead 16, ReLU FFN 2048, LayerNorm eps 1e-5, :216-230) and the
 * timm Mlp(SiLU) of each of `encodertext` / `encoder2`.  Dropout is identity (inference). */
typedef struct d3pm_encoder_layer_weights {
  const void *in_w, *in_b;      /* self_attn.in_proj [3d][d], [3d] */
  const void *out_w, *out_b;    /* self_attn.out_proj [d][d], [d]  */
  const void *lin1_w, *lin1_b;  /* linear1 [d_ff][d], [d_ff]       */
  const void *lin2_w, *lin2_b;  /* linear2 [d][d_ff], [d]          */
  const void *norm1_w, *norm1_b, *norm2_w, *norm2_b;
} d3pm_encoder_layer_weights;

typedef struct d3pm_encoder_weights {
  const d3pm_encoder_layer_weights *layers;   /* HOST array of n_layers entries */
  int32_t n_layers, n_heads, d_ff, mlp_hidden;
  const void *fc1_w, *fc1_b;                  /* [mlp_hidden][d], [mlp_hidden]  */
  const void *fc2_w, *fc2_b;                  /* [d][mlp_hidden], [d]           */
} d3pm_encoder_weights;

typedef struct d3pm_cond_weights {
  const void *text_emb;    /* [n_classes][d]            (ar_discrete.py:210) */
  const void *proms_emb;   /* [n_levels][n_classes][d]  (:211)               */
  const void *pe_text0;    /* [d]           sinusoid of position 0, model dtype */
  const void *pe_prompt;   /* [s_prompt][d] sinusoid table, model dtype         */
  int32_t n_levels;
  d3pm_encoder_weights text_encoder;     /* encodertext (:216-222) */
  d3pm_encoder_weights prompt_encoder;   /* encoder2    (:224-230) */
} d3pm_cond_weights;

size_t d3pm_cond_workspace_bytes(const d3pm_shape *shape, const d3pm_cond_weights *cw, int batch);

/* text device int32 [batch][s_text] (zero padded), prompt device int32 [batch][s_prompt][n_levels];
 * cond_text [batch][s_text][d], cond_prompt [batch][s_prompt][d] of `dtype` (feed d3pm_cond_kv). */
int d3pm_encode_conditions(const d3pm_shape *shape, const d3pm_cond_weights *cw, int batch, const int32_t *text,
                           const int32_t *prompt, void *cond_text, void *cond_prompt, void *workspace,
                           size_t workspace_bytes, void *stream);

/* One denoiser evaluation --------------------------------------------------------------------
 * Replaces the loop body at ar_discrete.py:752-776: resps_emb gather, n_layers x DiTBlock.forward
 * (:126-161, incl. torch's multi_head_attention_forward need_weights branch and timm Mlp), final
 * Linear on the masked hidden state.  x_t device int32 [batch][canvas]; frame_mask device uint8
 * [canvas]; logits_out device [batch*canvas][n_classes] of `dtype`; hidden_out optional
 * [batch*canvas][d] (state after the last block, before the final mask) or NULL;
 * only_layers >= 0 stops after that many blocks (block-level parity tests). */
int d3pm_denoise_step(const d3pm_shape *shape, const d3pm_weights *w, int batch, const int32_t *x_t,
                      const uint8_t *frame_mask, int t, const void *film, const void *kv_text,
                      const void *kv_prompt, void *workspace, size_t workspace_bytes,
                      void *logits_out, void *hidden_out, int only_layers, uint32_t flags,
                      void *stream);

/* Replaces AR.p_sample / q_posterior_logits / _at / _at_onehot (ar_discrete.py:337-420):
 * fp16 softmax of the x0-logits, closed-form fact1/fact2, log+log in fp16, fp32 Gumbel add from
 * the Philox stream (seed, row = (utt0+b)*canvas+frame, t), first-index argmax.
 * logits device [batch*canvas][n_classes] of logits_dtype; posterior_out optional device fp16
 * bit patterns [batch*canvas][n_classes] or NULL. */
int d3pm_posterior_sample(const d3pm_shape *shape, int batch, const void *logits, int logits_dtype,
                          const int32_t *x_t, int32_t *x_next, int t, const d3pm_schedule *sched,
                          uint64_t seed, uint32_t utt0, uint32_t flags, uint16_t *posterior_out,
                          void *stream);

/* Replaces the whole reverse process of AR.generate_audio after the condition encoders
 * (ar_discrete.py:748-780): for t = t_start .. t_stop+1: denoise step + posterior sample.
 * x device int32 [batch][canvas], in: x_T, out: x_{t_stop}.  trace optional device int32
 * [t_start-t_stop][batch][canvas] receiving x after every step, or NULL. */
int d3pm_sample_loop(const d3pm_shape *shape, const d3pm_weights *w, int batch, int32_t *x,
                     const uint8_t *frame_mask, int t_start, int t_stop, const void *film,
                     const void *kv_text, const void *kv_prompt, const d3pm_schedule *sched,
                     uint64_t seed, uint32_t utt0, uint32_t flags, void *workspace,
                     size_t workspace_bytes, int32_t *trace, void *stream);

/* The fp8 fast path (BASELINE.json configs[4]): same contracts as d3pm_denoise_step / d3pm_sample_loop, but the K = d_model
 * projections fed by a LayerNorm -- norm1 -> QKV, norm2|norm22 -> the merged cross-attention query projection, norm3(+FiLM) ->
 * fc1 -- and (when fc2_w8 is given) fc2 run on the block-scaled matrix instruction v_mfma_scale_f32_16x16x128_f8f6f4 with
 * e4m3 operands: LayerNorm rows are quantised on the fly in blocks of 32 (power-of-two scales), fc1's GELU epilogue writes
 * its output directly in that format for fc2, weights come from `fp8_blocks` (a HOST array of n_layers entries); fp32
 * accumulation, 16-bit residual stream.  Requires d_model = 512, a 16-bit model dtype and batch * canvas a multiple of 192;
 * D3PM_E_SHAPE otherwise (and D3PM_E_ARG under D3PM_FLAG_FORCE_GENERIC): a number labelled fp8 is never a 16-bit run.  The
 * reference has no such mode: tests report agreement against the 16-bit path. */
int d3pm_denoise_step_fp8(const d3pm_shape *shape, const d3pm_weights *weights,
                          const d3pm_fp8_block_weights *fp8_blocks, int batch, const int32_t *x_t,
                          const uint8_t *frame_mask, int t, const void *film, const void *kv_text,
                          const void *kv_prompt, void *workspace, size_t workspace_bytes, void *logits_out,
                          void *hidden_out, int only_layers, uint32_t flags, void *stream);
int d3pm_sample_loop_fp8(const d3pm_shape *shape, const d3pm_weights *weights,
                         const d3pm_fp8_block_weights *fp8_blocks, int batch, int32_t *x,
                         const uint8_t *frame_mask, int t_start, int t_stop, const void *film, const void *kv_text,
                         const void *kv_prompt, const d3pm_schedule *sched, uint64_t seed, uint32_t utt0,
                         uint32_t flags, void *workspace, size_t workspace_bytes, int32_t *trace, void *stream);

/* Replaces AR.q_sample / q_probs (ar_discrete.py:467-502): forward noising of x0 at step t with
 * Philox stream 1.  x0, x_out device int32 [batch][canvas]. */
int d3pm_q_sample(const d3pm_shape *shape, int batch, const int32_t *x0, int32_t *x_out,
                  const uint8_t *frame_mask, int t, const d3pm_schedule *sched, uint64_t seed,
                  uint32_t utt0, void *stream);

/* Training-side loss of one denoiser evaluation (AR.forward, ar_discrete.py:683-690; SURVEY.md §8f row 3, forward
 * half): row_loss[b][i] = logsumexp(x) - x[target] with x = logits[b][i][:] * mask[i], target = targets[b][i] (already
 * multiplied by the mask by the caller), fp32.  A padded frame (mask 0) yields log(n_classes).  logits device
 * [batch][canvas][n_classes] contiguous in `logits_dtype`; targets device int32 [batch][canvas]; row_loss device fp32
 * [batch][canvas].  The reference's loss of one step is the mean of these rows; the backward pass is not part of
 * this library. */
int d3pm_ce_loss_rows(const d3pm_shape *shape, int batch, const void *logits, int logits_dtype,
                      const int32_t *targets, const uint8_t *frame_mask, float *row_loss, void *stream);

/* The raw uniform stream the two samplers consume (what the reference draws with torch.rand,
 * ar_discrete.py:402,480): out device fp32 [rows][n_classes] for global rows row0.. at step t. */
int d3pm_uniform(uint64_t seed, int t, uint32_t row0, int rows, int n_classes, int stream_id,
                 float *out, void *stream);

/* Stock NAR model: quantizer levels 1..7 given level 0 ----------------------------
 * Replaces one pass of NAR.forward's inference loop (nar.py:76-101) = Base.forward (base.py:403-499) at
 * `quant_levels = level`: input assembly [text | sep | prompt | sep | response] + sinusoid, n_layers x
 * {AdaLN -> masked attention -> residual, AdaLN -> GELU FFN -> residual} (:161-234), classifier, and the
 * temperature sampling of the response rows (Gumbel-max over Philox stream 2 instead of torch's multinomial). */
typedef struct d3pm_nar_shape {
  int32_t d_model, n_heads, n_layers, n_tokens, n_prom_levels, n_resp_levels, dtype;
  const d3pm_tuning *tuning;   /* as in d3pm_shape */
} d3pm_nar_shape;

typedef struct d3pm_nar_block_weights {
  const void *attn_norm_emb;        /* blocks.i.attn.norm.emb.weight [n_resp_levels][2d]  (AdaLN :139) */
  const void *to_qkv_w;             /* [3d][d], no bias (:100)                                         */
  const void *to_out_w, *to_out_b;  /* [d][d], [d] (:101)                                              */
  const void *ffn_norm_emb;         /* blocks.i.ffn.norm.emb.weight                                    */
  const void *ffn0_w, *ffn0_b;      /* [4d][d], [4d] (:216)                                            */
  const void *ffn3_w, *ffn3_b;      /* [d][4d], [d]  (:219)                                            */
} d3pm_nar_block_weights;

typedef struct d3pm_nar_weights {
  const void *text_emb;             /* [n_tokens][d]                 (:336) */
  const void *proms_emb;            /* [n_prom_levels][n_tokens][d]  (:339) */
  const void *resps_emb;            /* [n_resp_levels][n_tokens][d]  (:340) */
  const void *sep;                  /* [d]                           (:344) */
  const void *classifier_w, *classifier_b;   /* [n_tokens][d], [n_tokens] (:360) */
  const void *pe;                   /* [pe_rows][d] sinusoid table in the model dtype (:38-89) */
  int32_t pe_rows;
  const d3pm_nar_block_weights *blocks;      /* HOST array of n_layers entries */
} d3pm_nar_weights;

size_t d3pm_nar_workspace_bytes(const d3pm_nar_shape *shape, int batch, int t_max);

/* lens device int32 [batch][3] = (t_text, t_prompt, t_response); text device int32 [batch][tt_max];
 * prom device int32 [batch][tp_max][n_prom_levels] (-1 = level absent); resp device int32
 * [batch][tr_max][n_resp_levels+1], levels 0..level given, level+1 written; t_max >= max(t_text+t_prompt+t_resp+2);
 * logits_out optional device [batch][t_max][n_tokens] of `dtype` (all rows, before sampling) or NULL. */
int d3pm_nar_level(const d3pm_nar_shape *shape, const d3pm_nar_weights *w, int batch, int t_max, const int32_t *lens,
                   const int32_t *text, int tt_max, const int32_t *prom, int tp_max, int32_t *resp, int tr_max,
                   int level, float temperature, uint64_t seed, uint32_t utt0, uint32_t flags, void *workspace,
                   size_t workspace_bytes, void *logits_out, void *stream);

/* Single-operator entry points (kernel-level parity tests and micro-benchmarks).  `family`:
 * 0 = auto (MFMA when the shape tiles, else generic), 1 = generic FMA kernels, 2 = MFMA (fails with
 * D3PM_E_SHAPE when unsupported).  Same contracts as inside the denoiser:
 *   linear:    Y[M][N] = epilogue(X[M][K] . W[N][K]^T + bias), act 0 none / 1 exact-erf GELU,
 *              optional residuals R1 (+R2) [M][N] and row mask (see csrc/d3pm_kernels.h)
 *   attention: per (utterance, head) softmax(rn(q*scale) . k^T) . v with Q [B*Tq][ldq], K/V [B*S][ldkv]
 *   layernorm: Y = LN(X)*w + b over the last dim (eps 1e-6), optional FiLM vector [2d] */
int d3pm_op_linear(int dtype, int family, const void *X, int ldx, const void *W, const void *bias, void *Y,
                   int ldy, const void *R1, const void *R2, int ldr, const uint8_t *row_mask,
                   int mask_period, int M, int N, int K, int act, const d3pm_tuning *tuning, void *stream);
/* Block-scaled fp8 single ops (BASELINE.json configs[4]; no reference counterpart: the reference is fp16 only).  Formats as
 * in d3pm_fp8_block_weights: codes [rows][K], scales [rows][4][K / 128].
 *   quantize_mx:   any 16-bit X [M][K] (K a multiple of 128) -> X8, SX; scale of a block = the smallest power of two with
 *                  absmax / scale <= 448, codes = rn_e4m3(x / scale) (never saturate);
 *   layernorm_mx:  Y8 [M][512], SX [M][4][4] = quantize_mx(LN(X) * w + b [FiLM]), the 16-bit LayerNorm result being exactly
 *                  d3pm_op_layernorm's;
 *   linear_mx:     epilogue(sum_k X8 2^sx . W8 2^sw + bias) with fp32 accumulation: act 0 none / 1 GELU, optional residual
 *                  R1 [M][N] and row mask as d3pm_op_linear, 16-bit output Y [M][ldy] in `out_dtype` -- or, with Y8 / SY given
 *                  (R1 = row_mask = NULL, Y ignored), the output itself in the MX format: Y8 [M][N], SY [M][4][N / 128].
 *                  M a multiple of 192, N of 128, K of 512.  tuning->gemm_variant 6 / 8 pins the 192 x 256 / 192 x 128 tile. */
int d3pm_op_quantize_mx(int dtype, const void *X, int ldx, void *X8, void *SX, int M, int K, void *stream);
int d3pm_op_layernorm_mx(int dtype, const void *X, void *Y8, void *SX, const void *w, const void *b, const void *film, int M,
                         int d, float eps, void *stream);
int d3pm_op_linear_mx(int out_dtype, const void *X8, int ldx, const void *SX, const void *W8, const void *SW, const void *bias,
                      void *Y, int ldy, const void *R1, int ldr, const uint8_t *row_mask, int mask_period, void *Y8, void *SY,
                      int M, int N, int K, int act, const d3pm_tuning *tuning, void *stream);
/* The folded-LayerNorm pieces as single ops (MFMA family, 16-bit; d3pm_fold_block above):
 *   row_stats     per row and 32-column part the pair (sum x, sum x^2), fp32 -- what a producing projection leaves behind -- in the
 *                 layout [ceil(M / 16)][d / 32][16][2] (the 16 rows of a row block side by side: one part of one row block is one
 *                 128-byte line): the pair of (row, part) starts at float (((row / 16) * (d / 32) + part) * 16 + row % 16) * 2;
 *   linear_stats  d3pm_op_linear with a residual (R1 [+ R2] [+ row mask]) that also writes the moments of the rows it stores (N = d);
 *   linear_fold   Y = act(rstd_r (X_r . Wf^T - mean_r fold_s) + fold_b) with the moments of X's rows from `stats_in` (layout of row_stats, K columns)
 *                 (X = the raw rows, Wf / fold_s / fold_b as d3pm_fold_build makes them); act 0 / 1 (GELU). */
int d3pm_op_row_stats(int dtype, const void *X, int ldx, int M, int d, float *stats, void *stream);
int d3pm_op_linear_stats(int dtype, const void *X, int ldx, const void *W, const void *bias, void *Y, int ldy, const void *R1,
                         const void *R2, int ldr, const uint8_t *row_mask, int mask_period, int M, int N, int K, float *stats_out,
                         const d3pm_tuning *tuning, void *stream);
int d3pm_op_linear_fold(int dtype, const void *X, int ldx, const void *Wf, const float *fold_s, const float *fold_b,
                        const float *stats_in, float eps, void *Y, int ldy, int M, int N, int K, int act, const d3pm_tuning *tuning,
                        void *stream);
/* One projection's tables as d3pm_fold_build makes them (tests): W [N][K], bias [N] or NULL, gamma / beta [K], film NULL or
 * (scale [K] | shift [K]) -> Wf [N][K], fold_s / fold_b [N]. */
int d3pm_op_fold_weights(int dtype, const void *W, const void *bias, const void *gamma, const void *beta, const void *film, int N, int K,
                         void *Wf, float *fold_s, float *fold_b, void *stream);
int d3pm_op_attention(int dtype, int family, const void *Q, int ldq, const void *K, const void *V, int ldkv,
                      void *O, int ldo, int B, int Tq, int S, int H, int hd, float scale, const d3pm_tuning *tuning,
                      void *stream);
/* The text and prompt cross-attentions of a DiT block (ar_discrete.py:138,142) as the block launches them: two independent
 * problems with the same B / Tq / H / hd -- queries Q1 / Q2 [B][Tq][ldq], keys and values K / V [B][S][ldkv] with S1 / S2 keys,
 * outputs O1 / O2 [B][Tq][ldo] -- in ONE launch (tuning->attn_cross_resident / attn_pair_sequential pick the schedule). */
int d3pm_op_attention_pair(int dtype, const void *Q1, const void *K1, const void *V1, void *O1, int S1, const void *Q2,
                           const void *K2, const void *V2, void *O2, int S2, int ldq, int ldkv, int ldo, int B, int Tq, int H,
                           int hd, float scale, const d3pm_tuning *tuning, void *stream);
int d3pm_op_layernorm(int dtype, const void *X, void *Y, const void *w, const void *b, const void *film,
                      int M, int d, float eps, void *stream);
/* Row-panel projection (d_model = 512): a Linear whose output is added to the residual stream, together with the LayerNorm(s)
 * the block applies to the new rows next, in ONE launch -- a workgroup owns whole rows (96 x 512 tiles), so the row moments
 * are reduced on chip.  Three forms, each bit-identical to the launches it replaces (d3pm_op_linear, then d3pm_op_layernorm):
 *   self-attention out-projection  (ar_discrete.py:132-136): X2 = film = row_mask = NULL, both LayerNorms:
 *       Y = rn(R1 + rn(X W^T + bias)),  ln_y = norm2(Y),  ln2_y = norm22(Y);
 *   both cross-attention out-projections (:138-143, the SAME weights): X2 given, film given, ln2_* = row_mask = NULL:
 *       Y = rn(rn(R1 + rn(X W^T + bias)) + rn(X2 W^T + bias)),  ln_y = FiLM(norm3(Y))  (:145-156);
 *   MLP down-projection (:159-161): row_mask given, X2 = film = ln2_* = NULL:
 *       Y = rn(R1 + rn(X W^T + bias)) * mask[row % mask_period],  ln_y = norm1 of the NEXT block (:131).
 * X (and X2) [M][ldx], W [512][K], Y / R1 / ln_y / ln2_y [M][512] (Y may alias R1), M a multiple of 96, K a multiple of 128.
 * fp8 fast path: with ln_sx (and ln2_sx) given -- first two forms only -- the LayerNorm rows leave in the block-scaled fp8
 * format (d3pm_op_quantize_mx of the 16-bit LayerNorm result, bit for bit): ln_y / ln2_y are then code buffers [M][512] bytes
 * and ln_sx / ln2_sx receive the scales [M][4][4].  D3PM_E_SHAPE for any other combination. */
int d3pm_op_linear_rowpanel(int dtype, const void *X, const void *X2, int ldx, const void *W, const void *bias, void *Y,
                            const void *R1, const uint8_t *row_mask, int mask_period, int M, int K, const void *ln_w,
                            const void *ln_b, void *ln_y, const void *ln2_w, const void *ln2_b, void *ln2_y,
                            const void *film, float eps, void *ln_sx, void *ln2_sx, void *stream);

/* Timing hooks for bench.py's roofline object.  A d3pm_prof attached to a tuning (d3pm_tuning.prof) makes every launch of the
 * kernel class `kclass` (D3PM_K_*; D3PM_K_COUNT = every class) made INSIDE d3pm_sample_loop with that tuning be bracketed by a
 * hipEvent pair on the launch stream -- in every 16th diffusion iteration only, so that the event pairs do not perturb the
 * timed region they measure; launches outside the loop (condition encoders, cond-K/V projections) are never bracketed.
 * d3pm_prof_read_class synchronises the events of one class and returns its launch count, total milliseconds and the
 * algorithmic flops / bytes of those launches; d3pm_prof_read returns the sums over the classes and resets.  A d3pm_prof is
 * owned by the caller (create / destroy) and must not be shared between threads that launch concurrently. */
/* D3PM_K_GEMM_LN: the launches that are a projection AND the LayerNorm(s) next to it (d3pm_op_linear_rowpanel) */
enum { D3PM_K_GEMM = 0, D3PM_K_ATTN = 1, D3PM_K_SAMPLE = 2, D3PM_K_LN = 3, D3PM_K_GEMM_LN = 4, D3PM_K_COUNT = 5 };
int d3pm_prof_create(int kclass, int max_events, struct d3pm_prof **out);
int d3pm_prof_read_class(struct d3pm_prof *prof, int kclass, int *launches, double *total_ms, double *flops, double *bytes);
int d3pm_prof_read(struct d3pm_prof *prof, int *launches, double *total_ms, double *flops, double *bytes);
int d3pm_prof_destroy(struct d3pm_prof *prof);

/* ---- Training side: gradients of the ops of AR.forward ------------------------------------------
 * The reference obtains them from autograd over ar_discrete.py:588-694 (DiT blocks :126-161, final Linear :776, masked
 * cross-entropy :683-690) inside engine.backward (utils/engines.py:144-147).  Here each is a single-op entry on fp32
 * tensors (the F32 precision mode); vall_e/vall_e/train.py replays the forward with a stash and walks it backwards.
 * All pointers are device pointers owned by the caller; nothing allocates or synchronises; 0 = ok.
 *
 * d3pm_op_matmul_f32      C[i][j] = beta C[i][j] + sum_k A[i sai + k sak] B[k sbk + j sbj]; rows of C whose row_mask entry
 *                         (row % mask_period) is 0 receive 0 (+ beta C).  dX = dY W and dW += dY^T X of every nn.Linear.
 * d3pm_op_colsum_f32      out[j] = beta out[j] + sum_i X[i][j]                              (bias gradients)
 * d3pm_op_act_bwd_f32     dU = dM act'(U), act = 1 exact-erf GELU (nn.GELU, :123), 2 ReLU, 3 SiLU (condition encoders)
 * d3pm_op_mask_rows_f32   X[i][:] = 0 where mask[i % period] == 0                           (x * mask, :128,161)
 * d3pm_op_layernorm_bwd_f32  LayerNorm (+ FiLM out = y (1 + film[c]) + film[d + c], :145-156) backward of one [M][d] block:
 *                         dX (overwritten or accumulated), dw / db / dfilm accumulated (fp32 atomics)
 * d3pm_op_attention_bwd_f32  softmax(scale q k^T) v backward (nn.MultiheadAttention core, :132,138,142): dQ overwritten,
 *                         dK / dV = beta_kv * old + new; stats = 2 B H Tq floats of scratch
 * d3pm_op_ce_bwd_f32      dlogits = mask (softmax(logits mask) - onehot(target)) gscale    (:683-690)
 * d3pm_op_embed_f32 / d3pm_op_embed_bwd_f32  nn.Embedding gather with the row mask, and its scatter-add (padding_idx row skipped) */
/* The condition-side embeddings as a single op (ar_discrete.py:736-741): which = 0 text rows W[tok] + pe0 (`tables` [n_classes][d],
 * `pe` [d]); which = 1 prompt rows sum_l W[l][tok_l] + pe[s] (`tokens` [rows][n_levels], -1 = level absent; `tables`
 * [n_levels][n_classes][d]; `pe` [s_prompt][d]).  Used by the training step, which needs the encoder inputs it stashes. */
int d3pm_op_cond_embed(int dtype, int which, const int32_t *tokens, int n_levels, const void *tables, const void *pe, void *y, int rows,
                       int s_prompt, int d, int n_classes, void *stream);
int d3pm_op_matmul_f32(const float *A, long sai, long sak, const float *B, long sbk, long sbj, float *C, int ldc, int M, int N, int K,
                       float beta, const uint8_t *row_mask, int mask_period, void *stream);
int d3pm_op_colsum_f32(const float *X, int ldx, int M, int N, float *out, float beta, void *stream);
int d3pm_op_act_bwd_f32(const float *U, const float *dM, float *dU, size_t n, int act, void *stream);
int d3pm_op_mask_rows_f32(float *X, int ldx, int M, int N, const uint8_t *mask, int period, void *stream);
int d3pm_op_layernorm_bwd_f32(const float *X, const float *dOut, const float *w, const float *b, const float *film, float eps, int M,
                              int d, float *dX, int accumulate_dx, float *dw, float *db, float *dfilm, void *stream);
int d3pm_op_attention_bwd_f32(const float *Q, int ldq, const float *K, const float *V, int ldkv, const float *dO, int ldo, float *dQ,
                              int lddq, float *dK, float *dV, int lddkv, float *stats, int B, int Tq, int S, int H, int hd, float scale,
                              float beta_kv, void *stream);
int d3pm_op_ce_bwd_f32(const float *logits, int ldl, const int32_t *targets, const uint8_t *frame_mask, int canvas, int rows, int K,
                       float gscale, float *dlogits, int ldd, void *stream);
int d3pm_op_embed_f32(const int32_t *tok, const uint8_t *mask, int period, const float *table, float *Y, int rows, int d, int n_classes,
                      void *stream);
int d3pm_op_embed_bwd_f32(const int32_t *tok, const uint8_t *mask, int period, const float *dY, float *dTable, int rows, int d,
                          int n_classes, int padding_idx, void *stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* D3PM_HIP_H */
