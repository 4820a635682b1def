// d3pm_kernels.h -- host-side launchers of the denoiser kernels (internal to the library).
// Two families implement the same contracts:
//   generic  (d3pm_generic.hip)  any shape, f32/f16/bf16, fp32 FMA arithmetic -- parity mode,
//                                MFMA-hostile shapes (head_dim 2 of the upstream model), cross-check
//   mfma     (d3pm_mfma_*.hip)   f16/bf16, LDS-tiled MFMA 16x16x32, shapes that tile by 128/64
#pragma once
#include "d3pm_common.h"

namespace d3pm {

enum { ACT_NONE = 0, ACT_GELU = 1, ACT_RELU = 2, ACT_SILU = 3 };

// Schedule choices travel with the arguments of a launch (include/d3pm_hip.h: d3pm_tuning); nullptr = the defaults.
// The library has no mutable global state: the defaults are a constant.
inline const d3pm_tuning& tune_of(const d3pm_tuning* t) {
  static const d3pm_tuning kDefault = {/*gemm_variant*/ 0, /*gemm_persist_slots*/ 1024, /*lat_tile*/ 0, /*attn_query_groups*/ 0,
                                       /*attn_pair_sequential*/ 1, /*attn_cross_resident*/ 1, /*row_panel*/ 10, /*workspace_alias*/ 1,
                                       /*regime_batch*/ 0, /*ln_fold*/ 1, /*prof*/ nullptr};
  return t ? *t : kDefault;
}

#ifdef D3PM_ABLATIONS
// libd3pm_hip_ab.so only (include/d3pm_hip_ab.h): process-wide knobs of the experiments that did not ship
struct AbKnobs { int big_mode = 1, attn_arm = 0, ring = 0, gelu_table = 0, ln_prologue = 0, fused_final_sample = 0; };
AbKnobs& ab_knobs();
#endif

// Y[M][N] = epilogue(X[M][K] . W[N][K]^T + bias)      (torch.nn.functional.linear layout)
// epilogue, with rn() = round to the storage dtype exactly where the eager reference rounds:
//   v = rn(acc + bias); if act: v = rn(act(v))   (exact-erf GELU, ReLU or SiLU);
//   if R1 && R2: v = rn(rn(R1 + R2) + v)  else if R1: v = rn(R1 + v);
//   if row_mask: v = v * row_mask[row % mask_period]
struct LinearArgs {
  const void* X = nullptr; int ldx = 0;
  const void* W = nullptr;
  const void* bias = nullptr;
  void* Y = nullptr; int ldy = 0;
  const void* R1 = nullptr; const void* R2 = nullptr; int ldr = 0;
  const uint8_t* row_mask = nullptr; int mask_period = 1;
  int M = 0, N = 0, K = 0;
  int act = ACT_NONE;
  const d3pm_tuning* tune = nullptr;
  // LayerNorm folded into this projection (MFMA family only, d3pm_mfma_tile.h): X = the raw residual rows [M][K = d_model], W = W o gamma,
  // `bias` unused; v = rn(rstd_r (acc - mean_r fold_s[n]) + fold_b[n]) [then act] with the row moments from stats_in [M][K / 32][2]
  const float* fold_s = nullptr; const float* fold_b = nullptr; const float* stats_in = nullptr; float fold_eps = 1e-6f;
  // moments of the rows this launch stores (N = d_model, after residual / mask): stats_out [M][N / 32][2] partial (sum, sum of squares)
  float* stats_out = nullptr;
};
// which LayerNorm-fold combinations the MFMA epilogues instantiate: LNF [+ GELU] without residual / mask; STATS with R1, R1 + R2, R1 + mask
inline bool fold_args_ok(const LinearArgs& a) {
  auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) % 16) == 0; };
  if (a.fold_s) {
    if (!a.fold_b || !a.stats_in || a.R1 || a.row_mask || a.stats_out || (a.act != ACT_NONE && a.act != ACT_GELU)) return false;
    if (a.K % 256 != 0 || a.N % 4 != 0 || !al16(a.fold_s) || !al16(a.fold_b) || !al16(a.stats_in)) return false;
  } else if (a.fold_b || a.stats_in) {
    return false;
  }
  if (a.stats_out) {
    if (!a.R1 || a.act != ACT_NONE || a.N % 32 != 0 || (reinterpret_cast<uintptr_t>(a.stats_out) % 8) != 0) return false;
  }
  return true;
}

// O[b][i][h*hd+c] = sum_j P[i][j] V[b][j][h*hd+c],  P = rn(softmax(rn(rn(q*scale) . k)))
// element (b, row, h, c) of Q lives at Q + (b*Tq+row)*ldq + h*hd + c; K/V likewise with S, ldkv.
struct AttnArgs {
  const void* Q = nullptr; int ldq = 0;
  const void* K = nullptr; const void* V = nullptr; int ldkv = 0;
  void* O = nullptr; int ldo = 0;
  int B = 0, Tq = 0, S = 0, H = 0, hd = 0;
  float scale = 1.f;
  // optional per-utterance number of valid keys (device int32[B]): keys >= key_len[b] are masked out
  // (the key-padding mask of the stock NAR attention, base.py:118-124); S is then the padded key count
  const int32_t* key_len = nullptr;
  // optional second, independent problem with the same B / Tq / H / hd launched in the same grid
  // (the text and prompt cross-attentions of one DiT block): its own Q, K/V (S2 keys) and output
  const void* Q2 = nullptr; const void* K2 = nullptr; const void* V2 = nullptr; void* O2 = nullptr; int S2 = 0;
  const d3pm_tuning* tune = nullptr;
};

// Y = LN(X) * w + b (eps), optionally FiLM: Y = rn(rn(LN * rn(1 + film[c])) + film[d + c])
// second output (w2/b2/Y2) optional: a second LayerNorm of the same rows (norm2 + norm22).
struct LayerNormArgs {
  const void* X = nullptr; void* Y = nullptr;
  const void* w = nullptr; const void* b = nullptr;
  const void* w2 = nullptr; const void* b2 = nullptr; void* Y2 = nullptr;
  const void* film = nullptr;
  int M = 0, d = 0; float eps = 1e-6f;
  // optional gather in front (the first block's norm1 directly on the token embedding, ar_discrete.py:753,127,131): row m of the
  // input is table[tokens[m]] (zeros where frame_mask[m % canvas] == 0) with X = the table; the gathered rows also go to Xout
  const int32_t* tokens = nullptr; const uint8_t* frame_mask = nullptr; int canvas = 0, n_classes = 0; void* Xout = nullptr;
};

// what the row-panel projection (d3pm_mfma_gemm_big.hip) does to the rows it has just finished: N = d_model = 512
struct RowPanelFuse {
  const void* X2 = nullptr;                                  // second operand through the same weights (dual out-projection)
  const void* lnw = nullptr; const void* lnb = nullptr; void* lny = nullptr;        // LayerNorm of the new rows
  const void* lnw2 = nullptr; const void* lnb2 = nullptr; void* lny2 = nullptr;     // a second LayerNorm of the same rows
  const void* film = nullptr;                                // FiLM (scale | shift, 2 x 512) on the first
  float eps = 1e-6f;
  // fp8 fast path: with sx (and sx2) set, lny (lny2) receive the LayerNorm rows in the block-scaled fp8 format of d3pm_mx.hip --
  // codes [M][512] bytes -- and sx (sx2) their scales [M][4][4]
  void* sx = nullptr; void* sx2 = nullptr;
};

// LayerNorm applied to the operand rows inside the latency GEMM (d3pm_mfma_gemm_lat.hip): X is then the residual stream.
// period > 0: M = 2 * period rows, rows >= period re-read source row m - period under the second LayerNorm (w2, b2).
struct LnPrologue {
  const void* w = nullptr; const void* b = nullptr; const void* w2 = nullptr; const void* b2 = nullptr;
  const void* film = nullptr; float eps = 1e-6f; int period = 0;
};

struct EmbedArgs {
  const int32_t* tokens = nullptr; const uint8_t* frame_mask = nullptr; int canvas = 0;
  const void* table = nullptr; void* Y = nullptr; int M = 0, d = 0, n_classes = 0;
  int n_q = 1;      // > 1: tokens [M][n_q], table [n_q][n_classes][d], row = rn(sum over the levels) (d3pm_shape.n_q)
};

// per-step scalars of the closed-form posterior (host-built from d3pm_schedule)
struct PosteriorConsts {
  float log_f1_zero;   // log16(rn16(0 + eps))
  float log_f1_d;      // log16(rn16(d_t + eps))      x_t != M, j == x_t
  float log_f1_c;      // log16(rn16(c_t + eps))      x_t == M, j != M
  float log_f1_one;    // log16(rn16(1 + eps))        x_t == M, j == M
  float dbar_prev;     // dbar_{t-1}
  float cbar_prev;     // cbar_{t-1}
  int t;
};

struct SampleArgs {
  const void* logits = nullptr; int logits_dtype = D3PM_F16; int ldl = 0;
  const int32_t* x_t = nullptr; int32_t* x_next = nullptr; int32_t* x_next2 = nullptr;
  uint16_t* posterior_out = nullptr;
  int rows = 0, n_classes = 0, mask_id = 0, canvas = 0;
  uint64_t seed = 0; uint32_t row0 = 0; int greedy = 0;
  const uint64_t* seed_hbm = nullptr;   // when set, the kernel reads the seed from HBM (lets a captured HIP graph be replayed with a new seed)
  int n_q = 1;      // > 1: row r = (frame row r / n_q, level r % n_q); Philox row = row0 + frame row, stream = level ? 16 + level : 0
  PosteriorConsts pc{};
};

int generic_linear(int dtype, const LinearArgs& a, hipStream_t s);
int generic_attention(int dtype, const AttnArgs& a, hipStream_t s);
int generic_layernorm(int dtype, const LayerNormArgs& a, hipStream_t s);
int embed_tokens(int dtype, const EmbedArgs& a, hipStream_t s);
// condition-side embeddings: text rows = rn(W[tok] + pe0); prompt rows = rn(rn(sum_l W[l][tok_l]) + pe[s])
// head_dim hd -> 2 hd re-layout around an attention call (d3pm_headpad.hip): [rows][groups][hd] <-> [rows][groups][2 hd], zeros above
int pad_heads(const void* in, void* out, long long rows, int groups, int hd, hipStream_t s);
int unpad_heads(const void* in, void* out, long long rows, int groups, int hd, hipStream_t s);
int cond_embed_text(int dtype, const int32_t* tok, const void* table, const void* pe0, void* y, int rows, int d,
                    int n_classes, hipStream_t s);
int cond_embed_prompt(int dtype, const int32_t* codes, int n_levels, const void* tables, const void* pe, void* y,
                      int rows, int s_prompt, int d, int n_classes, hipStream_t s);
int posterior_sample(const SampleArgs& a, hipStream_t s);
// What the sampler launch of iteration t can prepare for iteration t - 1 (folded-LayerNorm path, n_q = 1; d3pm_sample.hip): the
// embedding row of every id it has just drawn + the row's moments (ar_discrete.py:753,127), and -- in workgroups behind the
// sampler's -- fc1 of every block under norm3 + FiLM(t - 1) (:145-159).  Two launches less per iteration, same bits.
struct NextIterPrep {
  int dtype = D3PM_BF16;
  const void* table = nullptr; void* x = nullptr; float* stats = nullptr; const uint8_t* frame_mask = nullptr; int d = 0;
  const d3pm_block_weights* blocks = nullptr; int n_layers = 0; const void* film_t = nullptr;      // film_t: row t - 1 of the FiLM table
  void* Wf = nullptr; float* s_out = nullptr; float* b_out = nullptr;
};
bool posterior_sample_prep_supported(const SampleArgs& a, const NextIterPrep& n);
int posterior_sample_prep(const SampleArgs& a, const NextIterPrep& n, hipStream_t s);

// fp8 fast path on the block-scaled MFMA (d3pm_mx.hip): e4m3 codes [rows][K] + e8m0 block scales [rows][4][K / 128]
//   Y[M][N] = epilogue(sum_k X8 2^sx . W8 2^sw + bias), epilogue as LinearArgs (plain, GELU, R1, R1 + mask); with Y8 / SY set the
//   output is written in the MX format itself (codes [M][N] + scales [M][4][N / 128]: the next projection's operand) instead of Y
struct MxLinearArgs {
  const void* X8 = nullptr; int ldx = 0; const void* SX = nullptr;
  const void* W8 = nullptr; const void* SW = nullptr;
  const void* bias = nullptr;
  void* Y = nullptr; int ldy = 0;
  const void* R1 = nullptr; int ldr = 0;
  const uint8_t* row_mask = nullptr; int mask_period = 1;
  void* Y8 = nullptr; void* SY = nullptr;
  int M = 0, N = 0, K = 0;
  int act = ACT_NONE;
  const d3pm_tuning* tune = nullptr;
};
bool mx_linear_supported(int dtype, const MxLinearArgs& a);
int mx_linear(int dtype, const MxLinearArgs& a, hipStream_t s);
int layernorm_mx(int dtype, const void* x, uint8_t* y8, uint8_t* sx, const void* w, const void* b, const void* film,
                 const void* w2, const void* b2, uint8_t* y8_2, uint8_t* sx_2, int M, int d, float eps, hipStream_t s);
int quantize_mx(int dtype, const void* x, int ldx, uint8_t* y8, uint8_t* sx, int M, int K, hipStream_t s);

// MFMA family: return D3PM_E_SHAPE when the shape does not fit (caller falls back to generic)
bool mfma_linear_supported(int dtype, const LinearArgs& a);
int mfma_linear(int dtype, const LinearArgs& a, hipStream_t s);
bool mfma_attention_supported(int dtype, const AttnArgs& a);
int mfma_attention(int dtype, const AttnArgs& a, hipStream_t s);
bool mfma_attention32_supported(int dtype, const AttnArgs& a);   // d3pm_mfma_attn32.hip: self-attention on the 32 x 32 x 16 instruction
int mfma_attention32(int dtype, const AttnArgs& a, hipStream_t s);
bool mfma_attention32_cross_supported(int dtype, const AttnArgs& a);   // the resident cross-attention pair on the same instruction
int mfma_attention32_cross(int dtype, const AttnArgs& a, int n_qsplit, hipStream_t s);
// d3pm_mfma_attn_lat.hip: one or two utterances -- the key tiles of a 32-query group split over the four waves of a workgroup
bool mfma_attention_split_supported(int dtype, const AttnArgs& a);
int mfma_attention_split(int dtype, const AttnArgs& a, hipStream_t s);
#ifdef D3PM_ABLATIONS
int read_attn32_stamps(unsigned long long* out, int n);
#endif
#ifdef D3PM_ABLATIONS
bool panel64_ln_supported(int dtype, const LinearArgs& a, const LnPrologue& ln);
bool ln_prologue_linear_applies(int dtype, const LinearArgs& a, const LnPrologue& ln);   // would mfma_linear pick the latency GEMM?
int ln_prologue_linear(int dtype, const LinearArgs& a, const LnPrologue& ln, hipStream_t s);
#endif
// latency GEMM, two products through one weight panel: Y = rn(rn(R1 + rn(X W^T + b)) + rn(X2 W^T + b)) (d3pm_mfma_gemm_lat.hip)
bool panel64_dual_supported(int dtype, const LinearArgs& a, const void* X2);
int panel64_dual(int dtype, const LinearArgs& a, const void* X2, hipStream_t s);
// big-tile GEMM, the same two products at throughput batch sizes (d3pm_mfma_gemm_big.hip); stats_out optional
bool big_dual_supported(int dtype, const LinearArgs& a, const void* X2);
int big_dual(int dtype, const LinearArgs& a, const void* X2, hipStream_t s);
bool row_panel_supported(int dtype, const LinearArgs& a, const RowPanelFuse& f);
int row_panel_linear(int dtype, const LinearArgs& a, const RowPanelFuse& f, hipStream_t s);
bool fast_layernorm_supported(int dtype, const LayerNormArgs& a);
int fast_layernorm(int dtype, const LayerNormArgs& a, hipStream_t s);

PosteriorConsts make_posterior_consts(const d3pm_schedule* sched, int t);

// LayerNorm folded into the projections (d3pm_fold.hip): weight preparation and the non-GEMM producers of row moments
bool fold_shape_ok(int dtype, int d);
int fold_rows_launch(int dtype, const void* W, const void* bias, const void* gamma, const void* beta, const void* film, long film_ld,
                     int n_rows, int n_t, int K, void* Wf, float* s_out, float* b_out, hipStream_t s);
int fold_fc1_step_launch(int dtype, const d3pm_block_weights* blocks, int n_layers, const void* film_t, int d, void* Wf, float* s_out,
                         float* b_out, hipStream_t s);
int row_stats_launch(int dtype, const void* x, int ldx, int M, int d, float* stats, hipStream_t s);
int embed_tokens_stats(int dtype, const EmbedArgs& a, float* stats, hipStream_t s);

// ---- stock NAR model (levels 1..7): input assembly, AdaLN, temperature sampling (d3pm_nar.hip) ----
struct NarEmbedArgs {
  const int32_t* lens = nullptr;                 // [B][3] = (t_text, t_prompt, t_response)
  const int32_t* text = nullptr; int tt_max = 0;  // [B][tt_max]
  const int32_t* prom = nullptr; int tp_max = 0; int n_prom_levels = 8;   // [B][tp_max][n_prom_levels], -1 = level absent
  const int32_t* resp = nullptr; int tr_max = 0; int resp_stride = 8; int n_given = 1;   // [B][tr_max][resp_stride]
  const void *w_text = nullptr, *w_prom = nullptr, *w_resp = nullptr, *sep = nullptr, *pe = nullptr;
  void* x = nullptr; uint8_t* row_mask = nullptr; int32_t* key_len = nullptr;
  int batch = 0, t_max = 0, d = 0, n_tokens = 0;
};
int nar_embed(int dtype, const NarEmbedArgs& a, hipStream_t s);
int adaln(int dtype, const void* x, void* y, const void* emb_row, const uint8_t* row_mask, int M, int d, hipStream_t s);
int nar_sample(int dtype, const void* logits, int ldl, const int32_t* lens, int32_t* resp, int tr_max, int resp_stride,
               int t_max, int n_tokens, int level, float temperature, uint64_t seed, uint32_t utt0, int greedy, int batch,
               hipStream_t s);

}  // namespace d3pm
