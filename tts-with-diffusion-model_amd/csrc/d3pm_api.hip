// d3pm_api.hip -- extern "C" entry points (include/d3pm_hip.h) and the per-step launch sequence.
//
// One denoiser evaluation = the loop body of AR.generate_audio (ar_discrete.py:752-776):
//   embed -> n_layers x { LN1, QKV GEMM, self-attention, out GEMM(+res), LN2/LN22, 2 x Q GEMM,
//   2 x cross-attention against the cached condition K/V, 2 x out GEMM(+res), LN3+FiLM,
//   fc1 GEMM(+GELU), fc2 GEMM(+res, *mask) } -> final GEMM -> posterior/sample.
// Nothing here allocates or synchronises; everything is enqueued on the caller's stream.
#include <cstdarg>
#include <cstdio>
#include <new>
#include <vector>

#include "d3pm_kernels.h"
#ifdef D3PM_ABLATIONS
#include "../../include/d3pm_hip_ab.h"
#endif

namespace d3pm {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int q_sample_launch(const d3pm_shape*, int, const int32_t*, int32_t*, const uint8_t*, int, const d3pm_schedule*,
                    uint64_t, uint32_t, hipStream_t);
int uniform_launch(uint64_t, int, uint32_t, int, int, int, float*, hipStream_t);
int ce_loss_launch(int, const void*, int, const int32_t*, const uint8_t*, int, int, int, float*, hipStream_t);
#ifdef D3PM_ABLATIONS
bool final_sample_supported(int dtype, int n_classes, int d, const void* X, int ldx, const void* W);
int final_sample(int dtype, const void* X, int ldx, const void* W, const void* bias, int d, const SampleArgs& a, hipStream_t s);
int read_big_gemm_stamp(unsigned long long* out);
// libd3pm_hip_ab.so only: the knobs of the experiments (include/d3pm_hip_ab.h).  Process-wide on purpose -- an A/B script toggles
// them between interleaved arms; the product library does not contain this object.
AbKnobs& ab_knobs() {
  static AbKnobs k;
  return k;
}
#endif

}  // namespace d3pm

// ---- profiling hooks (bench.py roofline object): a caller-owned handle, reached through d3pm_tuning.prof -----------------------
// Only launches made from inside d3pm_sample_loop are ever bracketed (sample_now is false outside it): the condition
// encoders and the cond-K/V projections run once per utterance and are not part of any class's per-launch figures.
struct d3pm_prof {
  int kclass = -1;                // D3PM_K_* one class, D3PM_K_COUNT every class
  std::vector<hipEvent_t> ev;     // pairs
  std::vector<int> cls;           // class of pair i
  int used = 0;
  double flops[D3PM_K_COUNT] = {}, bytes[D3PM_K_COUNT] = {};
  int stride = 16;       // only the iterations with t % stride == 0 are bracketed (event pairs cost ~3 us each: all launches 6 % of the step, every 8th 2.5 %)
  bool sample_now = false;
};

namespace d3pm {

// what a launch sequence carries besides its arguments: the caller's schedule choices and (optionally) its timing hooks
struct Ctx {
  const d3pm_tuning* tune;
  d3pm_prof* prof;
  explicit Ctx(const d3pm_tuning* t) : tune(t), prof(t ? t->prof : nullptr) {}
};

struct ProfScope {
  bool on;
  d3pm_prof* p;
  hipStream_t s;
  ProfScope(const Ctx& cx, int kclass, hipStream_t st, double flops, double bytes) : p(cx.prof), s(st) {
    on = p && (p->kclass == kclass || p->kclass == D3PM_K_COUNT) && p->sample_now && p->used + 2 <= static_cast<int>(p->ev.size());
    if (on) {
      (void)hipEventRecord(p->ev[p->used], s);
      p->cls[p->used / 2] = kclass;
      p->flops[kclass] += flops;
      p->bytes[kclass] += bytes;
    }
  }
  ~ProfScope() {
    if (on) {
      (void)hipEventRecord(p->ev[p->used + 1], s);
      p->used += 2;
    }
  }
};

// ---- kernel-family dispatch ----------------------------------------------------------------
static int run_linear(const Ctx& cx, int dtype, LinearArgs a, uint32_t flags, hipStream_t s) {
  const size_t es = dtype_size(dtype);
  a.tune = cx.tune;
  ProfScope p(cx, D3PM_K_GEMM, s, 2.0 * a.M * a.N * a.K,
              es * (static_cast<double>(a.M) * a.K + static_cast<double>(a.N) * a.K +
                    static_cast<double>(a.M) * a.N * (1 + (a.R1 ? 1 : 0) + (a.R2 ? 1 : 0))));
  if (!(flags & D3PM_FLAG_FORCE_GENERIC) && mfma_linear_supported(dtype, a)) return mfma_linear(dtype, a, s);
  return generic_linear(dtype, a, s);
}
#ifdef D3PM_ABLATIONS
// D3PM_AB_LN_PROLOGUE: at one or two utterances the LayerNorm-fed projections normalise their operand rows themselves.
// LayerNorm + projection in one launch of the latency GEMM (d3pm_mfma_gemm_lat.hip); a.X is the un-normalised stream
static int run_ln_linear(const Ctx& cx, int dtype, LinearArgs a, const LnPrologue& ln, hipStream_t s) {
  const size_t es = dtype_size(dtype);
  a.tune = cx.tune;
  ProfScope p(cx, D3PM_K_GEMM_LN, s, 2.0 * a.M * a.N * a.K,
              es * (static_cast<double>(ln.period ? ln.period : a.M) * a.K + static_cast<double>(a.N) * a.K + static_cast<double>(a.M) * a.N));
  return ln_prologue_linear(dtype, a, ln, s);
}
static bool ln_prologue_applies(const Ctx& cx, int dtype, LinearArgs a, const LnPrologue& ln) {
  a.tune = cx.tune;
  return ln_prologue_linear_applies(dtype, a, ln);
}
#else
// the LayerNorm-prologue form of the latency GEMM was measured slower and lives in libd3pm_hip_ab.so only (include/d3pm_hip_ab.h)
static int run_ln_linear(const Ctx&, int, const LinearArgs&, const LnPrologue&, hipStream_t) { return D3PM_E_SHAPE; }
static bool ln_prologue_applies(const Ctx&, int, const LinearArgs&, const LnPrologue&) { return false; }
#endif
// projection onto the residual stream + the LayerNorm(s) of the new rows, one launch (d3pm_mfma_gemm_big.hip)
static int run_row_panel(const Ctx& cx, int dtype, const LinearArgs& a, const RowPanelFuse& f, hipStream_t s) {
  const size_t es = dtype_size(dtype);
  const double prods = f.X2 ? 2.0 : 1.0, mn = static_cast<double>(a.M) * a.N;
  ProfScope p(cx, D3PM_K_GEMM_LN, s, prods * 2.0 * a.M * a.N * a.K,
              es * (prods * a.M * a.K + static_cast<double>(a.N) * a.K + mn * (3.0 + (f.lny2 ? 1.0 : 0.0))));
  return row_panel_linear(dtype, a, f, s);
}
static int run_attention(const Ctx& cx, int dtype, AttnArgs a, uint32_t flags, hipStream_t s) {
  a.tune = cx.tune;
  // algorithmic bytes: queries in + outputs out (of BOTH problems of a paired launch: rounds 1-3 counted one, which made the
  // pair look like 1.3x wasted traffic -- PMC says 118.8 MB against 118.6) + keys and values
  ProfScope p(cx, D3PM_K_ATTN, s, 4.0 * a.B * a.H * a.Tq * static_cast<double>(a.S + a.S2) * a.hd,
              dtype_size(dtype) * ((a.Q2 ? 4.0 : 2.0) * a.B * a.Tq * a.H * a.hd + 2.0 * a.B * (a.S + a.S2) * a.H * a.hd));
  if (!(flags & D3PM_FLAG_FORCE_GENERIC) && mfma_attention_supported(dtype, a)) return mfma_attention(dtype, a, s);
  if (a.Q2) {   // the generic kernel takes one problem per launch
    AttnArgs first = a, second = a;
    first.Q2 = first.K2 = first.V2 = nullptr; first.O2 = nullptr; first.S2 = 0;
    second = first;
    second.Q = a.Q2; second.K = a.K2; second.V = a.V2; second.O = a.O2; second.S = a.S2;
    int rc = generic_attention(dtype, first, s);
    return rc != D3PM_OK ? rc : generic_attention(dtype, second, s);
  }
  return generic_attention(dtype, a, s);
}
static int run_layernorm(const Ctx& cx, int dtype, const LayerNormArgs& a, uint32_t flags, hipStream_t s) {
  ProfScope p(cx, D3PM_K_LN, s, 0.0, dtype_size(dtype) * static_cast<double>(a.M) * a.d * (a.Y2 ? 3.0 : 2.0));
  if (!(flags & D3PM_FLAG_FORCE_GENERIC) && fast_layernorm_supported(dtype, a)) return fast_layernorm(dtype, a, s);
  return generic_layernorm(dtype, a, s);
}

// ---- workspace carve-up ----------------------------------------------------------------------
struct Workspace {
  char *x, *h, *h2, *qkv, *att, *att2, *mlp, *logits;
  float* stats;       // [n][d / 32][2] row moments of the residual stream (LayerNorm folded into the projections, d3pm_mfma_tile.h)
  char* fc1f;         // [L][4d][d] fc1 under norm3 + FiLM(t), rebuilt per evaluation (d3pm_fold.hip); then fp32 [L][4d] s and b'
  float *fc1f_s, *fc1f_b;
  uint8_t* mxs;       // fp8 fast path: block scales [2n][d / 32] of the LayerNorm rows (a slot of their own: nothing else ever lives here)
  size_t total;
};
static size_t align256(size_t v) { return (v + 255) & ~static_cast<size_t>(255); }
// internal logits rows are padded to a multiple of 8 elements (16-B aligned rows for vector stores/loads)
static int logits_ld(const d3pm_shape& sh) { return (sh.n_classes + 7) & ~7; }
// quantizer levels generated jointly (d3pm_shape.n_q; 0 and 1 = the upstream level-0 path)
static int levels(const d3pm_shape& sh) { return sh.n_q > 1 ? sh.n_q : 1; }
static Workspace carve(const d3pm_shape& sh, int batch, char* base) {
  const size_t es = dtype_size(sh.dtype), n = static_cast<size_t>(batch) * sh.canvas, d = sh.d_model;
  Workspace w{};
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align256(bytes); return p; };
  w.x = take(n * d * es);
  w.h = take(n * d * es);
  w.h2 = take(n * d * es);           // adjacent to h: norm2 | norm22 outputs feed ONE [2n, d] query projection
  // Regions that are never live together share memory (D3PM_TUNE_WORKSPACE_ALIAS): 350 -> 200 MB per 32 utterances beside a
  // 256-MB Infinity Cache, +1.6 % tokens/s measured for the first pair alone (profiles/round2_c_ab_throughput.txt):
  //   packed qkv rows -> cross-attention queries -> MLP hidden rows -> logits of the iteration;
  //   norm22 output (dead once the query projection ran) -> prompt cross-attention output
  if (tune_of(sh.tuning).workspace_alias) {
    const size_t big = n * 4 * d * es, lg = n * levels(sh) * logits_ld(sh) * es;
    w.mlp = take(big > lg ? big : lg);
    w.qkv = w.mlp;
    w.logits = w.mlp;
    w.att = take(n * d * es);
    w.att2 = w.h2;
  } else {
    w.qkv = take(n * 3 * d * es);
    w.att = take(n * d * es);
    w.att2 = take(n * d * es);
    w.mlp = take(n * 4 * d * es);
    w.logits = take(n * levels(sh) * logits_ld(sh) * es);
  }
  w.stats = reinterpret_cast<float*>(take(((n + 15) & ~static_cast<size_t>(15)) * ((d + 31) / 32) * 2 * sizeof(float)));
  w.mxs = reinterpret_cast<uint8_t*>(take(2 * n * ((d + 31) / 32)));
  if (fold_shape_ok(sh.dtype, sh.d_model)) {
    const size_t L = sh.n_layers;
    w.fc1f = take(L * 4 * d * d * es);
    w.fc1f_s = reinterpret_cast<float*>(take(L * 4 * d * sizeof(float)));
    w.fc1f_b = reinterpret_cast<float*>(take(L * 4 * d * sizeof(float)));
  }
  w.total = off;
  return w;
}

static int check_shape(const d3pm_shape* sh, int batch) {
  D3PM_REQUIRE(sh, D3PM_E_ARG, "null shape");
  D3PM_REQUIRE(batch > 0 && sh->d_model > 0 && sh->n_heads > 0 && sh->d_model % sh->n_heads == 0 && sh->n_layers > 0 &&
                   sh->canvas > 0 && sh->s_text > 0 && sh->s_prompt > 0 && sh->n_classes > 1 && sh->mask_id >= 0 &&
                   sh->mask_id < sh->n_classes && sh->timesteps >= 2 && sh->n_q >= 0 && sh->n_q <= 16,
               D3PM_E_ARG, "inconsistent d3pm_shape");
  D3PM_REQUIRE(sh->dtype == D3PM_F32 || sh->dtype == D3PM_F16 || sh->dtype == D3PM_BF16, D3PM_E_ARG, "bad dtype %d",
               sh->dtype);
  return D3PM_OK;
}

#define D3PM_TRY(expr)            \
  do {                            \
    int rc_ = (expr);             \
    if (rc_ != D3PM_OK) return rc_; \
  } while (0)

static const char* at(const void* p, size_t elems, size_t es) { return static_cast<const char*>(p) + elems * es; }
static char* at(void* p, size_t elems, size_t es) { return static_cast<char*>(p) + elems * es; }

// The block sequence with the LayerNorms folded into the projections (ar_discrete.py:126-161; d3pm_mfma_tile.h EPI_LNF / EPI_STATS):
//   embed (+ moments) -> n_layers x { QKV <- x [norm1 folded], self-attention, out-projection + x (+ moments),
//   merged query projection <- x [norm2 | norm22 folded, N = 2d], paired cross-attention, both out-projections + x (+ moments),
//   fc1 + GELU <- x [norm3 + FiLM(t) folded], fc2 + x, frame mask (+ moments) }: ten launches per block, none of them a LayerNorm.
static int denoiser_blocks_folded(const d3pm_shape& sh, const d3pm_weights& w, int batch, const int32_t* x_t, const uint8_t* frame_mask,
                                  int t, const void* film, const void* kv_text, const void* kv_prompt, const Workspace& ws, int layers,
                                  hipStream_t s, bool prepared) {
  const int dt = sh.dtype, d = sh.d_model, H = sh.n_heads, hd = d / H, T = sh.canvas, n = batch * T;
  const Ctx cx(sh.tuning);
  const size_t es = dtype_size(dt);
  const float scale = static_cast<float>(std::sqrt(1.0 / static_cast<double>(hd)));
  EmbedArgs e;
  e.tokens = x_t; e.frame_mask = frame_mask; e.canvas = T; e.table = w.resps_emb; e.Y = ws.x;
  e.M = n; e.d = d; e.n_classes = sh.n_classes; e.n_q = levels(sh);
  if (!prepared) {      // (inside the loop the previous iteration's sampler launch has done both: posterior_sample_prep)
    {
      ProfScope p(cx, D3PM_K_LN, s, 0.0, es * static_cast<double>(n) * d * 2.0);
      D3PM_TRY(embed_tokens_stats(dt, e, ws.stats, s));
    }
    {   // fc1 of every block under norm3 + FiLM(t): the weights this evaluation's fc1 launches read
      ProfScope p(cx, D3PM_K_LN, s, 0.0, es * 2.0 * layers * 4.0 * d * d);
      D3PM_TRY(fold_fc1_step_launch(dt, w.blocks, layers, at(film, static_cast<size_t>(t) * sh.n_layers * 2 * d, es), d, ws.fc1f, ws.fc1f_s,
                                    ws.fc1f_b, s));
    }
  }
  auto folded = [&](const void* Wf, const float* fs, const float* fb, void* Y, int N, int act) -> int {
    LinearArgs g;
    g.X = ws.x; g.ldx = d; g.W = Wf; g.Y = Y; g.ldy = N; g.M = n; g.N = N; g.K = d; g.act = act;
    g.fold_s = fs; g.fold_b = fb; g.stats_in = ws.stats; g.fold_eps = 1e-6f;
    D3PM_REQUIRE(mfma_linear_supported(dt, g), D3PM_E_SHAPE, "folded LayerNorm projection %d x %d x %d not supported", n, N, d);
    return run_linear(cx, dt, g, 0, s);
  };
  for (int l = 0; l < layers; ++l) {
    const d3pm_block_weights& b = w.blocks[l];
    const d3pm_fold_block& f = w.fold[l];
    // ---- self-attention ----
    D3PM_TRY(folded(f.qkv_w, f.qkv_s, f.qkv_b, ws.qkv, 3 * d, ACT_NONE));
    AttnArgs a;
    a.Q = ws.qkv; a.ldq = 3 * d; a.K = at(ws.qkv, d, es); a.V = at(ws.qkv, 2 * d, es); a.ldkv = 3 * d;
    a.O = ws.att; a.ldo = d; a.B = batch; a.Tq = T; a.S = T; a.H = H; a.hd = hd; a.scale = scale;
    D3PM_TRY(run_attention(cx, dt, a, 0, s));
    LinearArgs g;
    g.X = ws.att; g.ldx = d; g.W = b.attn_out_w; g.bias = b.attn_out_b; g.Y = ws.x; g.ldy = d;
    g.R1 = ws.x; g.ldr = d; g.M = n; g.N = d; g.K = d; g.stats_out = ws.stats;
    D3PM_TRY(run_linear(cx, dt, g, 0, s));
    // ---- cross-attention: q_text | q_prompt are the two halves of ONE [n][2d] projection of x (the same q rows under norm2 / norm22)
    D3PM_TRY(folded(f.q2_w, f.q2_s, f.q2_b, ws.qkv, 2 * d, ACT_NONE));
    {
      const void* kvt = at(kv_text, static_cast<size_t>(l) * batch * sh.s_text * 2 * d, es);
      const void* kvp = at(kv_prompt, static_cast<size_t>(l) * batch * sh.s_prompt * 2 * d, es);
      a = AttnArgs();
      a.Q = ws.qkv; a.ldq = 2 * d; a.K = kvt; a.V = at(kvt, d, es); a.ldkv = 2 * d; a.O = ws.att; a.ldo = d;
      a.B = batch; a.Tq = T; a.S = sh.s_text; a.H = H; a.hd = hd; a.scale = scale;
      a.Q2 = at(ws.qkv, d, es); a.K2 = kvp; a.V2 = at(kvp, d, es); a.O2 = ws.att2; a.S2 = sh.s_prompt;
      D3PM_TRY(run_attention(cx, dt, a, 0, s));
    }
    // ---- x = (x + o_text) + o_prompt, rounded at each add like the eager sum ----
    g = LinearArgs();
    g.X = ws.att; g.ldx = d; g.W = b.cross_out_w; g.bias = b.cross_out_b; g.Y = ws.x; g.ldy = d; g.R1 = ws.x; g.ldr = d;
    g.M = n; g.N = d; g.K = d; g.stats_out = ws.stats;
    if ((tune_of(sh.tuning).row_panel & 8) && tune_of(sh.tuning).gemm_variant == 0 && panel64_dual_supported(dt, g, ws.att2)) {
      g.tune = cx.tune;      // one or two utterances: both products through one resident weight panel (same bits as the two launches)
      ProfScope p(cx, D3PM_K_GEMM, s, 2.0 * 2.0 * g.M * g.N * g.K, es * (2.0 * g.M * g.K + static_cast<double>(g.N) * g.K + 2.0 * g.M * g.N));
      D3PM_TRY(panel64_dual(dt, g, ws.att2, s));
    } else if ((tune_of(sh.tuning).row_panel & 2) && (g.tune = cx.tune, big_dual_supported(dt, g, ws.att2))) {
      // throughput batches: the same two products through one tile of the ordinary big-tile launch (same bits again)
      ProfScope p(cx, D3PM_K_GEMM, s, 2.0 * 2.0 * g.M * g.N * g.K, es * (2.0 * g.M * g.K + static_cast<double>(g.N) * g.K + 2.0 * g.M * g.N));
      D3PM_TRY(big_dual(dt, g, ws.att2, s));
    } else {
      g.Y = ws.h; g.R1 = nullptr; g.stats_out = nullptr;          // o_text -> h (free: no LayerNorm output lives there any more)
      D3PM_TRY(run_linear(cx, dt, g, 0, s));
      g = LinearArgs();
      g.X = ws.att2; g.ldx = d; g.W = b.cross_out_w; g.bias = b.cross_out_b; g.Y = ws.x; g.ldy = d;
      g.R1 = ws.x; g.R2 = ws.h; g.ldr = d; g.M = n; g.N = d; g.K = d; g.stats_out = ws.stats;
      D3PM_TRY(run_linear(cx, dt, g, 0, s));
    }
    // ---- FiLM-modulated MLP: the (layer, t) copy of fc1 carries norm3 and the modulation ----
    const size_t ln = static_cast<size_t>(l) * 4 * d;
    D3PM_TRY(folded(at(ws.fc1f, ln * d, es), ws.fc1f_s + ln, ws.fc1f_b + ln, ws.mlp, 4 * d, ACT_GELU));
    g = LinearArgs();
    g.X = ws.mlp; g.ldx = 4 * d; g.W = b.fc2_w; g.bias = b.fc2_b; g.Y = ws.x; g.ldy = d; g.R1 = ws.x; g.ldr = d;
    g.row_mask = frame_mask; g.mask_period = T; g.M = n; g.N = d; g.K = 4 * d; g.stats_out = ws.stats;
    D3PM_TRY(run_linear(cx, dt, g, 0, s));
  }
  return D3PM_OK;
}

// hidden state after `layers` blocks is left in ws.x
// is this evaluation taking the folded-LayerNorm launch sequence (denoiser_blocks_folded)?
static bool fold_active(const d3pm_shape& sh, const d3pm_weights& w, uint32_t flags, const d3pm_fp8_block_weights* f8) {
  return !f8 && w.fold && tune_of(sh.tuning).ln_fold && !(flags & D3PM_FLAG_FORCE_GENERIC) && fold_shape_ok(sh.dtype, sh.d_model);
}

static int denoiser_blocks(const d3pm_shape& sh, const d3pm_weights& w, int batch, const int32_t* x_t,
                           const uint8_t* frame_mask, int t, const void* film, const void* kv_text,
                           const void* kv_prompt, const Workspace& ws, int layers, uint32_t flags, hipStream_t s,
                           const d3pm_fp8_block_weights* f8 = nullptr, bool prepared = false) {
  const int dt = sh.dtype, d = sh.d_model, H = sh.n_heads, hd = d / H, T = sh.canvas;
  const int n = batch * T;
  const Ctx cx(sh.tuning);
  const size_t es = dtype_size(dt);
  // fp8 fast path (BASELINE.json configs[4]): the three LayerNorm-fed K = d projections take e4m3 operands; the e4m3 rows
  // and their scales live where the 16-bit LayerNorm outputs would (ws.h | ws.h2 are adjacent: 2 n d 2 bytes)
  const bool use8 = f8 != nullptr;
  if (use8) {   // the *_fp8 entry points never fall back to the 16-bit kernels silently: a number labelled fp8 is fp8
    D3PM_REQUIRE(!(flags & D3PM_FLAG_FORCE_GENERIC), D3PM_E_ARG, "fp8 fast path: D3PM_FLAG_FORCE_GENERIC selects the 16-bit generic kernels");
    D3PM_REQUIRE(d == 512 && (dt == D3PM_F16 || dt == D3PM_BF16) && n % 192 == 0 && ws.h2 == at(ws.h, static_cast<size_t>(n) * d, dtype_size(dt)),
                 D3PM_E_SHAPE, "fp8 fast path needs d_model = 512, a 16-bit model dtype and batch * canvas (%d) a multiple of 192", n);
  }
  // MX operands of the LayerNorm-fed projections live where the 16-bit LayerNorm outputs would: codes [2n][512] fill ws.h.
  // The shared qkv | q | hidden | logits region (n x 4d elements = 4096 n bytes) also holds fc1's MX output: codes [n][2048] at
  // its start and scales [n][64] at byte 2048 n.  The LayerNorm block scales [2n][16] have a workspace slot of their own (ws.mxs):
  // they are read by a persistent GEMM for the whole launch, so they must not share bytes with anything that launch writes (a
  // 16-bit fc1 output [n][2048] x 2 B covers the whole shared region).
  uint8_t* x8 = reinterpret_cast<uint8_t*>(ws.h);
  uint8_t* h8 = reinterpret_cast<uint8_t*>(ws.mlp);
  uint8_t* sh8 = h8 + static_cast<size_t>(n) * 4 * d;
  uint8_t* sx8 = ws.mxs;
  auto mx_gemm = [&](const uint8_t* X8, int ldx8, const uint8_t* SX8, const void* W8, const void* SW8, const void* bias, void* Y, int ldy,
                     const void* R1, const uint8_t* mask, int period, uint8_t* Y8, uint8_t* SY, int M, int N, int K, int act) -> int {
    MxLinearArgs m;
    m.X8 = X8; m.ldx = ldx8; m.SX = SX8; m.W8 = W8; m.SW = SW8; m.bias = bias; m.Y = Y; m.ldy = ldy; m.R1 = R1; m.ldr = ldy;
    m.row_mask = mask; m.mask_period = period; m.Y8 = Y8; m.SY = SY; m.M = M; m.N = N; m.K = K; m.act = act; m.tune = cx.tune;
    D3PM_REQUIRE(mx_linear_supported(dt, m), D3PM_E_SHAPE, "fp8 fast path: block-scaled GEMM %d x %d x %d not supported", M, N, K);
    ProfScope p(cx, D3PM_K_GEMM, s, 2.0 * M * N * K,
                1.03125 * (static_cast<double>(M) * K + static_cast<double>(N) * K) + static_cast<double>(M) * N * (Y8 ? 1.03125 : (R1 ? 2.0 : 1.0) * es));
    return mx_linear(dt, m, s);
  };
  const float scale = static_cast<float>(std::sqrt(1.0 / static_cast<double>(hd)));

  // LayerNorm folded into the projections (d3pm_tuning.ln_fold, d3pm_fold_block): every LayerNorm-fed projection reads the raw
  // residual stream and normalises in its epilogue; every projection that lands on the residual stream leaves the row moments
  if (fold_active(sh, w, flags, f8))
    return denoiser_blocks_folded(sh, w, batch, x_t, frame_mask, t, film, kv_text, kv_prompt, ws, layers, s, prepared);

  EmbedArgs e;
  e.tokens = x_t; e.frame_mask = frame_mask; e.canvas = T; e.table = w.resps_emb; e.Y = ws.x;
  e.M = n; e.d = d; e.n_classes = sh.n_classes; e.n_q = levels(sh);
  // the first block's norm1 reads the embedding rows straight from the table and writes x beside its own output: one launch and
  // one pass over x less per iteration (same bits: the gather is a copy)
  bool embed_fused = false;
  if (!use8 && !(flags & D3PM_FLAG_FORCE_GENERIC) && layers > 0 && levels(sh) == 1) {
    LayerNormArgs ln0;
    ln0.X = w.resps_emb; ln0.Y = ws.h; ln0.w = w.blocks[0].norm1_w; ln0.b = w.blocks[0].norm1_b; ln0.M = n; ln0.d = d; ln0.eps = 1e-6f;
    ln0.tokens = x_t; ln0.frame_mask = frame_mask; ln0.canvas = T; ln0.n_classes = sh.n_classes; ln0.Xout = ws.x;
    if (fast_layernorm_supported(dt, ln0)) {
      ProfScope p(cx, D3PM_K_LN, s, 0.0, dtype_size(dt) * static_cast<double>(n) * d * 3.0);
      D3PM_TRY(fast_layernorm(dt, ln0, s));
      embed_fused = true;
    }
  }
  if (!embed_fused) D3PM_TRY(embed_tokens(dt, e, s));

  // row-panel launches (D3PM_TUNE_ROW_PANEL): a projection that lands on the residual stream also writes the LayerNorm(s) the
  // block applies to the new rows next -- same bits, one launch and one pass over x less each
  // (one 96-row tile per workgroup: only when the tiles fill >= 85 % of whole rounds over the 256 CUs, as for the other big tiles)
  const long long rp_tiles = n / 96, rp_rounds = (rp_tiles + 255) / 256;
  const bool rp_fills = n % 96 == 0 && rp_tiles * 5 >= 256 * 4 && rp_tiles * 100 >= rp_rounds * 256 * 85;
  const int panel = (!(flags & D3PM_FLAG_FORCE_GENERIC) && d == 512 && rp_fills && (dt == D3PM_F16 || dt == D3PM_BF16))
                        ? (tune_of(sh.tuning).row_panel & (use8 ? 3 : 7)) : 0;      // fp8: fc2 is a block-scaled GEMM of its own
  bool norm1_done = embed_fused;   // norm1(x) of this block is already in ws.h (the embedding launch, or the previous block's fc2)
  // the opposite regime (one or two utterances, latency GEMM): LayerNorm runs as the prologue of the projection it feeds
#ifdef D3PM_ABLATIONS
  const bool lnpro_on = ab_knobs().ln_prologue != 0;
#else
  const bool lnpro_on = false;
#endif
  const bool lnpro = lnpro_on && !use8 && !(flags & D3PM_FLAG_FORCE_GENERIC) && d == 512 && (dt == D3PM_F16 || dt == D3PM_BF16);

  for (int l = 0; l < layers; ++l) {
    const d3pm_block_weights& b = w.blocks[l];
    // ---- self-attention ----
    LayerNormArgs ln;
    ln.X = ws.x; ln.Y = ws.h; ln.w = b.norm1_w; ln.b = b.norm1_b; ln.M = n; ln.d = d; ln.eps = 1e-6f;
    LinearArgs g;
    g.X = ws.h; g.ldx = d; g.W = b.attn_in_w; g.bias = b.attn_in_b; g.Y = ws.qkv; g.ldy = 3 * d;
    g.M = n; g.N = 3 * d; g.K = d;
    if (use8) {
      {
        ProfScope p(cx, D3PM_K_LN, s, 0.0, static_cast<double>(n) * d * (es + 1.03125));
        D3PM_TRY(layernorm_mx(dt, ws.x, x8, sx8, b.norm1_w, b.norm1_b, nullptr, nullptr, nullptr, nullptr, nullptr, n, d, 1e-6f, s));
      }
      D3PM_TRY(mx_gemm(x8, d, sx8, f8[l].attn_in_w8, f8[l].attn_in_scale, b.attn_in_b, ws.qkv, 3 * d, nullptr, nullptr, 1, nullptr, nullptr,
                       n, 3 * d, d, ACT_NONE));
    } else {
      LnPrologue lp;
      lp.w = ln.w; lp.b = ln.b; lp.eps = ln.eps;
      LinearArgs gx = g;
      gx.X = ws.x;
      if (!norm1_done && lnpro && ln_prologue_applies(cx, dt, gx, lp)) {
        D3PM_TRY(run_ln_linear(cx, dt, gx, lp, s));
      } else {
        if (!norm1_done) D3PM_TRY(run_layernorm(cx, dt, ln, flags, s));
        D3PM_TRY(run_linear(cx, dt, g, flags, s));
      }
    }
    norm1_done = false;
    AttnArgs a;
    a.Q = ws.qkv; a.ldq = 3 * d; a.K = at(ws.qkv, d, es); a.V = at(ws.qkv, 2 * d, es); a.ldkv = 3 * d;
    a.O = ws.att; a.ldo = d; a.B = batch; a.Tq = T; a.S = T; a.H = H; a.hd = hd; a.scale = scale;
    D3PM_TRY(run_attention(cx, dt, a, flags, s));
    g = LinearArgs();
    g.X = ws.att; g.ldx = d; g.W = b.attn_out_w; g.bias = b.attn_out_b; g.Y = ws.x; g.ldy = d;
    g.R1 = ws.x; g.ldr = d; g.M = n; g.N = d; g.K = d;
    // ---- cross-attention: text keys with LN2 queries, prompt keys with LN22 queries, SAME weights ----
    ln = LayerNormArgs();
    ln.X = ws.x; ln.Y = ws.h; ln.w = b.norm2_w; ln.b = b.norm2_b; ln.Y2 = ws.h2; ln.w2 = b.norm22_w; ln.b2 = b.norm22_b;
    ln.M = n; ln.d = d; ln.eps = 1e-6f;
    RowPanelFuse rp;
    rp.lnw = ln.w; rp.lnb = ln.b; rp.lny = ln.Y; rp.lnw2 = ln.w2; rp.lnb2 = ln.b2; rp.lny2 = ln.Y2; rp.eps = ln.eps;
    if (use8) {      // the LayerNorm rows leave the row-panel launch as MX codes + block scales: the query projection's operand
      rp.lny = x8; rp.lny2 = x8 + static_cast<size_t>(n) * d; rp.sx = sx8; rp.sx2 = sx8 + static_cast<size_t>(n) * 16;
    }
    const bool norm2_fused = (panel & 1) && row_panel_supported(dt, g, rp);
    if (norm2_fused) D3PM_TRY(run_row_panel(cx, dt, g, rp, s));
    else D3PM_TRY(run_linear(cx, dt, g, flags, s));
    char* q_text = ws.qkv;
    char* q_prom = at(ws.qkv, static_cast<size_t>(n) * d, es);
    if (use8) {
      // MX rows of norm2(x) | norm22(x) stacked [2n][d] (fills ws.h), block scales [2n][16]: ONE GEMM for both queries
      if (!norm2_fused) {
        ProfScope p(cx, D3PM_K_LN, s, 0.0, static_cast<double>(n) * d * (es + 2 * 1.03125));
        D3PM_TRY(layernorm_mx(dt, ws.x, x8, sx8, b.norm2_w, b.norm2_b, nullptr, b.norm22_w, b.norm22_b,
                              x8 + static_cast<size_t>(n) * d, sx8 + static_cast<size_t>(n) * 16, n, d, 1e-6f, s));
      }
      D3PM_TRY(mx_gemm(x8, d, sx8, f8[l].cross_in_w8, f8[l].cross_in_scale, b.cross_in_b, q_text, d, nullptr, nullptr, 1, nullptr, nullptr,
                       2 * n, d, d, ACT_NONE));
    } else if (ws.h2 == at(ws.h, static_cast<size_t>(n) * d, es)) {
      // both query projections share cross_attn's q rows: LN2|LN22 outputs and q_text|q_prompt are adjacent in
      // the workspace, so the pair is ONE [2n, d] x [d, d] GEMM (twice the workgroups of either alone)
      g = LinearArgs();
      g.X = ws.h; g.ldx = d; g.W = b.cross_in_w; g.bias = b.cross_in_b; g.Y = q_text; g.ldy = d; g.M = 2 * n; g.N = d; g.K = d;
      LnPrologue lp;
      lp.w = ln.w; lp.b = ln.b; lp.w2 = ln.w2; lp.b2 = ln.b2; lp.eps = ln.eps; lp.period = n;
      LinearArgs gx = g;
      gx.X = ws.x;
      if (!norm2_fused && lnpro && ln_prologue_applies(cx, dt, gx, lp)) {
        D3PM_TRY(run_ln_linear(cx, dt, gx, lp, s));
      } else {
        if (!norm2_fused) D3PM_TRY(run_layernorm(cx, dt, ln, flags, s));
        D3PM_TRY(run_linear(cx, dt, g, flags, s));
      }
    } else {
      if (!norm2_fused) D3PM_TRY(run_layernorm(cx, dt, ln, flags, s));
      for (int which = 0; which < 2; ++which) {
        g = LinearArgs();
        g.X = which ? ws.h2 : ws.h; g.ldx = d; g.W = b.cross_in_w; g.bias = b.cross_in_b;
        g.Y = which ? q_prom : q_text; g.ldy = d; g.M = n; g.N = d; g.K = d;
        D3PM_TRY(run_linear(cx, dt, g, flags, s));
      }
    }
    {   // text and prompt cross-attention are independent: one paired launch
      const void* kvt = at(kv_text, static_cast<size_t>(l) * batch * sh.s_text * 2 * d, es);
      const void* kvp = at(kv_prompt, static_cast<size_t>(l) * batch * sh.s_prompt * 2 * d, es);
      a = AttnArgs();
      a.Q = q_text; a.ldq = d; a.K = kvt; a.V = at(kvt, d, es); a.ldkv = 2 * d; a.O = ws.att; a.ldo = d;
      a.B = batch; a.Tq = T; a.S = sh.s_text; a.H = H; a.hd = hd; a.scale = scale;
      a.Q2 = q_prom; a.K2 = kvp; a.V2 = at(kvp, d, es); a.O2 = ws.att2; a.S2 = sh.s_prompt;
      D3PM_TRY(run_attention(cx, dt, a, flags, s));
    }
    // ---- both out-projections, then the FiLM-modulated MLP ----
    ln = LayerNormArgs();
    ln.X = ws.x; ln.Y = ws.h; ln.w = b.norm3_w; ln.b = b.norm3_b; ln.M = n; ln.d = d; ln.eps = 1e-6f;
    ln.film = at(film, (static_cast<size_t>(t) * sh.n_layers + l) * 2 * d, es);
    g = LinearArgs();
    g.X = ws.att; g.ldx = d; g.W = b.cross_out_w; g.bias = b.cross_out_b; g.Y = ws.x; g.ldy = d; g.R1 = ws.x; g.ldr = d;
    g.M = n; g.N = d; g.K = d;
    rp = RowPanelFuse();
    rp.X2 = ws.att2; rp.lnw = ln.w; rp.lnb = ln.b; rp.lny = ln.Y; rp.film = ln.film; rp.eps = ln.eps;
    if (use8) { rp.lny = x8; rp.sx = sx8; }
    const bool norm3_fused = (panel & 2) && row_panel_supported(dt, g, rp);
    if (norm3_fused) {
      D3PM_TRY(run_row_panel(cx, dt, g, rp, s));
    } else if ((tune_of(sh.tuning).row_panel & 8) && !(flags & D3PM_FLAG_FORCE_GENERIC) && tune_of(sh.tuning).gemm_variant == 0 &&
               panel64_dual_supported(dt, g, ws.att2)) {
      // one or two utterances: both out-projections in ONE launch of the latency GEMM (the weight panel is resident in LDS; o_text
      // stays in registers): x = (x + o_text) + o_prompt with the roundings of the two-launch form below
      g.tune = cx.tune;
      ProfScope p(cx, D3PM_K_GEMM, s, 2.0 * 2.0 * g.M * g.N * g.K, es * (2.0 * g.M * g.K + static_cast<double>(g.N) * g.K + 2.0 * g.M * g.N));
      D3PM_TRY(panel64_dual(dt, g, ws.att2, s));
    } else {
      // o_text -> h (free now); x = (x + o_text) + o_prompt, rounded at each add like the eager sum
      g = LinearArgs();
      g.X = ws.att; g.ldx = d; g.W = b.cross_out_w; g.bias = b.cross_out_b; g.Y = ws.h; g.ldy = d; g.M = n; g.N = d; g.K = d;
      D3PM_TRY(run_linear(cx, dt, g, flags, s));
      g = LinearArgs();
      g.X = ws.att2; g.ldx = d; g.W = b.cross_out_w; g.bias = b.cross_out_b; g.Y = ws.x; g.ldy = d;
      g.R1 = ws.x; g.R2 = ws.h; g.ldr = d; g.M = n; g.N = d; g.K = d;
      D3PM_TRY(run_linear(cx, dt, g, flags, s));
    }
    g = LinearArgs();
    g.X = ws.h; g.ldx = d; g.W = b.fc1_w; g.bias = b.fc1_b; g.Y = ws.mlp; g.ldy = 4 * d; g.M = n; g.N = 4 * d; g.K = d;
    g.act = ACT_GELU;
    const bool fc2_mx = use8 && f8[l].fc2_w8 && f8[l].fc2_scale;
    if (use8) {
      if (!norm3_fused) {
        ProfScope p(cx, D3PM_K_LN, s, 0.0, static_cast<double>(n) * d * (es + 1.03125));
        D3PM_TRY(layernorm_mx(dt, ws.x, x8, sx8, b.norm3_w, b.norm3_b, ln.film, nullptr, nullptr, nullptr, nullptr, n, d, 1e-6f, s));
      }
      // with an MX fc2 the GELU epilogue writes the hidden layer as codes + block scales (half the bytes out, half in again)
      D3PM_TRY(mx_gemm(x8, d, sx8, f8[l].fc1_w8, f8[l].fc1_scale, b.fc1_b, ws.mlp, 4 * d, nullptr, nullptr, 1, fc2_mx ? h8 : nullptr,
                       fc2_mx ? sh8 : nullptr, n, 4 * d, d, ACT_GELU));
    } else {
      LnPrologue lp;
      lp.w = ln.w; lp.b = ln.b; lp.film = ln.film; lp.eps = ln.eps;
      LinearArgs gx = g;
      gx.X = ws.x;
      if (!norm3_fused && lnpro && ln_prologue_applies(cx, dt, gx, lp)) {
        D3PM_TRY(run_ln_linear(cx, dt, gx, lp, s));
      } else {
        if (!norm3_fused) D3PM_TRY(run_layernorm(cx, dt, ln, flags, s));
        D3PM_TRY(run_linear(cx, dt, g, flags, s));
      }
    }
    g = LinearArgs();
    g.X = ws.mlp; g.ldx = 4 * d; g.W = b.fc2_w; g.bias = b.fc2_b; g.Y = ws.x; g.ldy = d; g.R1 = ws.x; g.ldr = d;
    g.row_mask = frame_mask; g.mask_period = T; g.M = n; g.N = d; g.K = 4 * d;
    rp = RowPanelFuse();
    if (l + 1 < layers) { rp.lnw = w.blocks[l + 1].norm1_w; rp.lnb = w.blocks[l + 1].norm1_b; rp.lny = ws.h; rp.eps = 1e-6f; }
    if (fc2_mx) {
      D3PM_TRY(mx_gemm(h8, 4 * d, sh8, f8[l].fc2_w8, f8[l].fc2_scale, b.fc2_b, ws.x, d, ws.x, frame_mask, T, nullptr, nullptr, n, d, 4 * d,
                       ACT_NONE));
    } else if ((panel & 4) && l + 1 < layers && row_panel_supported(dt, g, rp)) {
      D3PM_TRY(run_row_panel(cx, dt, g, rp, s));
      norm1_done = true;
    } else {
      D3PM_TRY(run_linear(cx, dt, g, flags, s));
    }
  }
  return D3PM_OK;
}

// the fused final + sampler kernel takes over wherever the final projection would have run on the MFMA family
static bool fused_final_sample_applies(const d3pm_shape& sh, const d3pm_weights& w, const Workspace& ws, uint32_t flags) {
#ifdef D3PM_ABLATIONS
  return ab_knobs().fused_final_sample && !(flags & D3PM_FLAG_FORCE_GENERIC) && sh.d_model >= 64 && levels(sh) == 1 &&
         final_sample_supported(sh.dtype, sh.n_classes, sh.d_model, ws.x, sh.d_model, w.final_w);
#else
  return false;      // built, bit-identical, 253 us vs 30 + 87 us: lives in libd3pm_hip_ab.so only (include/d3pm_hip_ab.h)
#endif
}

static int final_logits(const d3pm_shape& sh, const d3pm_weights& w, int batch, const Workspace& ws, void* logits,
                        int ldl, uint32_t flags, hipStream_t s) {
  // x is already multiplied by the frame mask at the end of every block (ar_discrete.py:161,773).
  // n_q > 1: one projection per level (final_w [n_q][n_classes][d]) into the level's [ldl]-wide slot of a frame's n_q * ldl logits
  const Ctx cx(sh.tuning);
  const size_t es = dtype_size(sh.dtype);
  for (int l = 0; l < levels(sh); ++l) {
    LinearArgs g;
    g.X = ws.x; g.ldx = sh.d_model; g.W = at(w.final_w, static_cast<size_t>(l) * sh.n_classes * sh.d_model, es);
    g.bias = at(w.final_b, static_cast<size_t>(l) * sh.n_classes, es); g.Y = at(logits, static_cast<size_t>(l) * ldl, es);
    g.ldy = levels(sh) * ldl; g.M = batch * sh.canvas; g.N = sh.n_classes; g.K = sh.d_model;
    D3PM_TRY(run_linear(cx, sh.dtype, g, flags, s));
  }
  return D3PM_OK;
}

}  // namespace d3pm

using namespace d3pm;

extern "C" {

int d3pm_abi_version(void) { return D3PM_ABI_VERSION; }
const char* d3pm_last_error(void) { return g_err; }

size_t d3pm_workspace_bytes(const d3pm_shape* shape, int batch) {
  if (check_shape(shape, batch) != D3PM_OK) return 0;
  return carve(*shape, batch, nullptr).total;
}

int d3pm_film_table(const d3pm_shape* sh, const d3pm_weights* w, void* film, void* stream) {
  D3PM_TRY(check_shape(sh, 1));
  D3PM_REQUIRE(w && w->blocks && w->time_emb && film, D3PM_E_ARG, "d3pm_film_table: null pointer");
  const int d = sh->d_model;
  for (int l = 0; l < sh->n_layers; ++l) {
    LinearArgs g;
    g.X = w->time_emb; g.ldx = d; g.W = w->blocks[l].tfc_w; g.bias = w->blocks[l].tfc_b;
    g.Y = at(film, static_cast<size_t>(l) * 2 * d, dtype_size(sh->dtype)); g.ldy = sh->n_layers * 2 * d;
    g.M = sh->timesteps + 1; g.N = 2 * d; g.K = d;
    D3PM_TRY(generic_linear(sh->dtype, g, static_cast<hipStream_t>(stream)));
  }
  return D3PM_OK;
}

// ---- LayerNorm folded into the projections: the tables ---------------------------------------------------------------
struct FoldLayout { size_t qkv_w, q2_w, qkv_s, qkv_b, q2_s, q2_b, per_layer; };
static FoldLayout fold_layout(const d3pm_shape& sh) {
  const size_t es = dtype_size(sh.dtype), d = sh.d_model;
  FoldLayout L{};
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t at_ = off; off += align256(bytes); return at_; };
  L.qkv_w = take(3 * d * d * es); L.q2_w = take(2 * d * d * es);
  L.qkv_s = take(3 * d * 4); L.qkv_b = take(3 * d * 4); L.q2_s = take(2 * d * 4); L.q2_b = take(2 * d * 4);
  L.per_layer = off;
  return L;
}

size_t d3pm_fold_bytes(const d3pm_shape* sh) {
  if (check_shape(sh, 1) != D3PM_OK || !fold_shape_ok(sh->dtype, sh->d_model)) return 0;
  return fold_layout(*sh).per_layer * static_cast<size_t>(sh->n_layers);
}

int d3pm_fold_build(const d3pm_shape* sh, const d3pm_weights* w, void* storage, size_t storage_bytes, d3pm_fold_block* out, void* stream) {
  D3PM_TRY(check_shape(sh, 1));
  D3PM_REQUIRE(w && w->blocks && storage && out, D3PM_E_ARG, "d3pm_fold_build: null pointer");
  D3PM_REQUIRE(fold_shape_ok(sh->dtype, sh->d_model), D3PM_E_SHAPE, "d3pm_fold_build: needs a 16-bit dtype and d_model a multiple of 256");
  const FoldLayout L = fold_layout(*sh);
  D3PM_REQUIRE(storage_bytes >= L.per_layer * sh->n_layers, D3PM_E_WORKSPACE, "d3pm_fold_build: storage %zu < required %zu", storage_bytes,
               L.per_layer * static_cast<size_t>(sh->n_layers));
  D3PM_REQUIRE(reinterpret_cast<uintptr_t>(storage) % 256 == 0, D3PM_E_ARG, "d3pm_fold_build: storage must be 256-byte aligned");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int dt = sh->dtype, d = sh->d_model;
  const size_t es = dtype_size(dt);
  for (int l = 0; l < sh->n_layers; ++l) {
    const d3pm_block_weights& b = w->blocks[l];
    char* base = static_cast<char*>(storage) + L.per_layer * l;
    auto f32 = [&](size_t off) { return reinterpret_cast<float*>(base + off); };
    D3PM_TRY(fold_rows_launch(dt, b.attn_in_w, b.attn_in_b, b.norm1_w, b.norm1_b, nullptr, 0, 3 * d, 1, d, base + L.qkv_w, f32(L.qkv_s), f32(L.qkv_b), s));
    // cross_attn's q rows under norm2 (text queries) and under norm22 (prompt queries): the two halves of one [2d][d] operand
    D3PM_TRY(fold_rows_launch(dt, b.cross_in_w, b.cross_in_b, b.norm2_w, b.norm2_b, nullptr, 0, d, 1, d, base + L.q2_w, f32(L.q2_s), f32(L.q2_b), s));
    D3PM_TRY(fold_rows_launch(dt, b.cross_in_w, b.cross_in_b, b.norm22_w, b.norm22_b, nullptr, 0, d, 1, d, base + L.q2_w + static_cast<size_t>(d) * d * es,
                              f32(L.q2_s) + d, f32(L.q2_b) + d, s));
    d3pm_fold_block& o = out[l];
    o.qkv_w = base + L.qkv_w; o.qkv_s = f32(L.qkv_s); o.qkv_b = f32(L.qkv_b);
    o.q2_w = base + L.q2_w; o.q2_s = f32(L.q2_s); o.q2_b = f32(L.q2_b);
  }
  return D3PM_OK;
}

int d3pm_op_fold_weights(int dtype, const void* W, const void* bias, const void* gamma, const void* beta, const void* film, int N, int K,
                         void* Wf, float* fold_s, float* fold_b, void* stream) {
  D3PM_REQUIRE(W && gamma && beta && Wf && fold_s && fold_b && N > 0 && K > 0 && K % 8 == 0 && (dtype == D3PM_F16 || dtype == D3PM_BF16),
               D3PM_E_ARG, "d3pm_op_fold_weights: bad arguments");
  return fold_rows_launch(dtype, W, bias, gamma, beta, film, 0, N, 1, K, Wf, fold_s, fold_b, static_cast<hipStream_t>(stream));
}

int d3pm_op_row_stats(int dtype, const void* X, int ldx, int M, int d, float* stats, void* stream) {
  D3PM_REQUIRE(X && stats && M > 0 && d > 0 && d % 32 == 0 && ldx % 8 == 0 && (dtype == D3PM_F16 || dtype == D3PM_BF16), D3PM_E_ARG,
               "d3pm_op_row_stats: bad arguments");
  return row_stats_launch(dtype, X, ldx, M, d, stats, static_cast<hipStream_t>(stream));
}

int d3pm_op_linear_stats(int dtype, const void* X, int ldx, const void* W, const void* bias, void* Y, int ldy, const void* R1,
                         const void* R2, int ldr, const uint8_t* row_mask, int mask_period, int M, int N, int K, float* stats_out,
                         const d3pm_tuning* tuning, void* stream) {
  D3PM_REQUIRE(X && W && Y && R1 && stats_out && M > 0 && N > 0 && K > 0, D3PM_E_ARG, "d3pm_op_linear_stats: bad arguments");
  LinearArgs g;
  g.tune = tuning;
  g.X = X; g.ldx = ldx; g.W = W; g.bias = bias; g.Y = Y; g.ldy = ldy; g.R1 = R1; g.R2 = R2; g.ldr = ldr;
  g.row_mask = row_mask; g.mask_period = mask_period > 0 ? mask_period : 1; g.M = M; g.N = N; g.K = K; g.stats_out = stats_out;
  D3PM_REQUIRE(mfma_linear_supported(dtype, g), D3PM_E_SHAPE, "d3pm_op_linear_stats: needs the MFMA family (16-bit, K %% 64 == 0, N %% 32 == 0)");
  return mfma_linear(dtype, g, static_cast<hipStream_t>(stream));
}

int d3pm_op_linear_fold(int dtype, const void* X, int ldx, const void* Wf, const float* fold_s, const float* fold_b, const float* stats_in,
                        float eps, void* Y, int ldy, int M, int N, int K, int act, const d3pm_tuning* tuning, void* stream) {
  D3PM_REQUIRE(X && Wf && fold_s && fold_b && stats_in && Y && M > 0 && N > 0 && K > 0, D3PM_E_ARG, "d3pm_op_linear_fold: bad arguments");
  LinearArgs g;
  g.tune = tuning;
  g.X = X; g.ldx = ldx; g.W = Wf; g.Y = Y; g.ldy = ldy; g.M = M; g.N = N; g.K = K; g.act = act;
  g.fold_s = fold_s; g.fold_b = fold_b; g.stats_in = stats_in; g.fold_eps = eps;
  D3PM_REQUIRE(mfma_linear_supported(dtype, g), D3PM_E_SHAPE,
               "d3pm_op_linear_fold: needs the MFMA family (16-bit), K a multiple of 256, N of 4, act 0 / 1 and 16-byte aligned tables");
  return mfma_linear(dtype, g, static_cast<hipStream_t>(stream));
}

int d3pm_cond_kv(const d3pm_shape* sh, const d3pm_weights* w, int batch, const void* cond_text, const void* cond_prompt,
                 void* kv_text, void* kv_prompt, void* stream) {
  D3PM_TRY(check_shape(sh, batch));
  D3PM_REQUIRE(w && w->blocks && cond_text && cond_prompt && kv_text && kv_prompt, D3PM_E_ARG, "d3pm_cond_kv: null pointer");
  const int d = sh->d_model;
  const size_t es = dtype_size(sh->dtype);
  const Ctx cx(sh->tuning);
  for (int l = 0; l < sh->n_layers; ++l)
    for (int which = 0; which < 2; ++which) {
      const int S = which ? sh->s_prompt : sh->s_text;
      LinearArgs g;
      g.X = which ? cond_prompt : cond_text; g.ldx = d;
      g.W = at(w->blocks[l].cross_in_w, static_cast<size_t>(d) * d, es);   // k|v rows of the packed in-projection
      g.bias = at(w->blocks[l].cross_in_b, d, es);
      g.Y = at(which ? kv_prompt : kv_text, static_cast<size_t>(l) * batch * S * 2 * d, es); g.ldy = 2 * d;
      g.M = batch * S; g.N = 2 * d; g.K = d;
      D3PM_TRY(run_linear(cx, sh->dtype, g, 0, static_cast<hipStream_t>(stream)));
    }
  return D3PM_OK;
}

// ---- condition encoders ----------------------------------------------------------------------------
struct CondWs { char *x, *tmp, *qkv, *att, *ff, *qkv_pad, *att_pad; size_t total; };
#ifndef D3PM_ENC_HEAD_PAD
#define D3PM_ENC_HEAD_PAD 1      // A/B builds (tools/build_variant.py): 0 = the encoders' 32-wide heads stay on the generic attention kernel
#endif
// the encoder's self-attention on the 64-wide MFMA kernels through zero-padded heads (d3pm_headpad.hip)
static bool encoder_pads_heads(const d3pm_shape& sh, const d3pm_encoder_weights& e) {
  return D3PM_ENC_HEAD_PAD != 0 && (sh.dtype == D3PM_F16 || sh.dtype == D3PM_BF16) && e.n_heads > 0 && sh.d_model == 32 * e.n_heads;
}
static CondWs carve_cond(const d3pm_shape& sh, const d3pm_cond_weights& cw, int batch, char* base) {
  const size_t es = dtype_size(sh.dtype), d = sh.d_model;
  const size_t n = static_cast<size_t>(batch) * (sh.s_prompt > sh.s_text ? sh.s_prompt : sh.s_text);
  auto wide = [](const d3pm_encoder_weights& e) { return static_cast<size_t>(e.d_ff > e.mlp_hidden ? e.d_ff : e.mlp_hidden); };
  const size_t ffw = wide(cw.text_encoder) > wide(cw.prompt_encoder) ? wide(cw.text_encoder) : wide(cw.prompt_encoder);
  CondWs w{};
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align256(bytes); return p; };
  w.x = take(n * d * es);
  w.tmp = take(n * d * es);
  w.qkv = take(n * 3 * d * es);
  w.att = take(n * d * es);
  w.ff = take(n * ffw * es);
  if (encoder_pads_heads(sh, cw.text_encoder) || encoder_pads_heads(sh, cw.prompt_encoder)) {
    w.qkv_pad = take(n * 6 * d * es);
    w.att_pad = take(n * 2 * d * es);
  }
  w.total = off;
  return w;
}

// x (ws.x, [rows][d]) -> out ([rows][d]); `seq` rows per utterance
static int run_encoder(const d3pm_shape& sh, const d3pm_encoder_weights& e, int batch, int seq, const CondWs& ws, void* out,
                       hipStream_t s) {
  const int dt = sh.dtype, d = sh.d_model, n = batch * seq, hd = d / e.n_heads;
  const size_t es = dtype_size(dt);
  const Ctx cx(sh.tuning);
  for (int l = 0; l < e.n_layers; ++l) {
    const d3pm_encoder_layer_weights& w = e.layers[l];
    LinearArgs g;
    g.X = ws.x; g.ldx = d; g.W = w.in_w; g.bias = w.in_b; g.Y = ws.qkv; g.ldy = 3 * d; g.M = n; g.N = 3 * d; g.K = d;
    D3PM_TRY(run_linear(cx, dt, g, 0, s));
    AttnArgs a;
    a.Q = ws.qkv; a.ldq = 3 * d; a.K = at(ws.qkv, d, es); a.V = at(ws.qkv, 2 * d, es); a.ldkv = 3 * d; a.O = ws.att; a.ldo = d;
    a.B = batch; a.Tq = seq; a.S = seq; a.H = e.n_heads; a.hd = hd; a.scale = static_cast<float>(std::sqrt(1.0 / hd));
    if (encoder_pads_heads(sh, e) && ws.qkv_pad) {
      D3PM_TRY(pad_heads(ws.qkv, ws.qkv_pad, n, 3 * e.n_heads, hd, s));
      AttnArgs p = a;
      p.Q = ws.qkv_pad; p.K = at(ws.qkv_pad, 2 * d, es); p.V = at(ws.qkv_pad, 4 * d, es); p.ldq = p.ldkv = 6 * d;
      p.O = ws.att_pad; p.ldo = 2 * d; p.hd = 2 * hd;                // scale stays 1 / sqrt(hd)
      D3PM_TRY(run_attention(cx, dt, p, 0, s));
      D3PM_TRY(unpad_heads(ws.att_pad, ws.att, n, e.n_heads, hd, s));
    } else {
      D3PM_TRY(run_attention(cx, dt, a, 0, s));
    }
    g = LinearArgs();   // x + self_attn(x), then post-norm
    g.X = ws.att; g.ldx = d; g.W = w.out_w; g.bias = w.out_b; g.Y = ws.tmp; g.ldy = d; g.R1 = ws.x; g.ldr = d; g.M = n; g.N = d; g.K = d;
    D3PM_TRY(run_linear(cx, dt, g, 0, s));
    LayerNormArgs ln;
    ln.X = ws.tmp; ln.Y = ws.x; ln.w = w.norm1_w; ln.b = w.norm1_b; ln.M = n; ln.d = d; ln.eps = 1e-5f;
    D3PM_TRY(run_layernorm(cx, dt, ln, 0, s));
    g = LinearArgs();   // FFN: linear2(relu(linear1(x)))
    g.X = ws.x; g.ldx = d; g.W = w.lin1_w; g.bias = w.lin1_b; g.Y = ws.ff; g.ldy = e.d_ff; g.M = n; g.N = e.d_ff; g.K = d; g.act = ACT_RELU;
    D3PM_TRY(run_linear(cx, dt, g, 0, s));
    g = LinearArgs();
    g.X = ws.ff; g.ldx = e.d_ff; g.W = w.lin2_w; g.bias = w.lin2_b; g.Y = ws.tmp; g.ldy = d; g.R1 = ws.x; g.ldr = d; g.M = n; g.N = d; g.K = e.d_ff;
    D3PM_TRY(run_linear(cx, dt, g, 0, s));
    ln = LayerNormArgs();
    ln.X = ws.tmp; ln.Y = ws.x; ln.w = w.norm2_w; ln.b = w.norm2_b; ln.M = n; ln.d = d; ln.eps = 1e-5f;
    D3PM_TRY(run_layernorm(cx, dt, ln, 0, s));
  }
  LinearArgs g;   // timm Mlp: fc2(silu(fc1(x)))
  g.X = ws.x; g.ldx = d; g.W = e.fc1_w; g.bias = e.fc1_b; g.Y = ws.ff; g.ldy = e.mlp_hidden; g.M = n; g.N = e.mlp_hidden; g.K = d; g.act = ACT_SILU;
  D3PM_TRY(run_linear(cx, dt, g, 0, s));
  g = LinearArgs();
  g.X = ws.ff; g.ldx = e.mlp_hidden; g.W = e.fc2_w; g.bias = e.fc2_b; g.Y = out; g.ldy = d; g.M = n; g.N = d; g.K = e.mlp_hidden;
  return run_linear(cx, dt, g, 0, s);
}

static bool encoder_ok(const d3pm_encoder_weights& e, int d) {
  return e.layers && e.n_layers > 0 && e.n_heads > 0 && d % e.n_heads == 0 && e.d_ff > 0 && e.mlp_hidden > 0 && e.fc1_w &&
         e.fc1_b && e.fc2_w && e.fc2_b;
}

size_t d3pm_cond_workspace_bytes(const d3pm_shape* sh, const d3pm_cond_weights* cw, int batch) {
  if (check_shape(sh, batch) != D3PM_OK || !cw) return 0;
  return carve_cond(*sh, *cw, batch, nullptr).total;
}

int d3pm_encode_conditions(const d3pm_shape* sh, const d3pm_cond_weights* cw, int batch, const int32_t* text,
                           const int32_t* prompt, void* cond_text, void* cond_prompt, void* workspace,
                           size_t workspace_bytes, void* stream) {
  D3PM_TRY(check_shape(sh, batch));
  D3PM_REQUIRE(cw && text && prompt && cond_text && cond_prompt && workspace && cw->text_emb && cw->proms_emb &&
                   cw->pe_text0 && cw->pe_prompt && cw->n_levels > 0,
               D3PM_E_ARG, "d3pm_encode_conditions: null pointer");
  D3PM_REQUIRE(encoder_ok(cw->text_encoder, sh->d_model) && encoder_ok(cw->prompt_encoder, sh->d_model), D3PM_E_ARG,
               "d3pm_encode_conditions: incomplete encoder weights");
  CondWs ws = carve_cond(*sh, *cw, batch, static_cast<char*>(workspace));
  D3PM_REQUIRE(workspace_bytes >= ws.total, D3PM_E_WORKSPACE, "workspace %zu < required %zu", workspace_bytes, ws.total);
  hipStream_t s = static_cast<hipStream_t>(stream);
  D3PM_TRY(cond_embed_text(sh->dtype, text, cw->text_emb, cw->pe_text0, ws.x, batch * sh->s_text, sh->d_model, sh->n_classes, s));
  D3PM_TRY(run_encoder(*sh, cw->text_encoder, batch, sh->s_text, ws, cond_text, s));
  D3PM_TRY(cond_embed_prompt(sh->dtype, prompt, cw->n_levels, cw->proms_emb, cw->pe_prompt, ws.x, batch * sh->s_prompt,
                             sh->s_prompt, sh->d_model, sh->n_classes, s));
  return run_encoder(*sh, cw->prompt_encoder, batch, sh->s_prompt, ws, cond_prompt, s);
}

static int denoise_step_impl(const d3pm_shape* sh, const d3pm_weights* w, int batch, const int32_t* x_t,
                             const uint8_t* frame_mask, int t, const void* film, const void* kv_text, const void* kv_prompt,
                             void* workspace, size_t workspace_bytes, void* logits_out, void* hidden_out, int only_layers,
                             uint32_t flags, void* stream, const d3pm_fp8_block_weights* f8) {
  D3PM_TRY(check_shape(sh, batch));
  D3PM_REQUIRE(w && w->blocks && x_t && frame_mask && film && kv_text && kv_prompt && workspace, D3PM_E_ARG,
               "d3pm_denoise_step: null pointer");
  D3PM_REQUIRE(t >= 0 && t <= sh->timesteps, D3PM_E_ARG, "t=%d outside [0,%d]", t, sh->timesteps);
  Workspace ws = carve(*sh, batch, static_cast<char*>(workspace));
  D3PM_REQUIRE(workspace_bytes >= ws.total, D3PM_E_WORKSPACE, "workspace %zu < required %zu", workspace_bytes, ws.total);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int layers = (only_layers >= 0 && only_layers < sh->n_layers) ? only_layers : sh->n_layers;
  D3PM_TRY(denoiser_blocks(*sh, *w, batch, x_t, frame_mask, t, film, kv_text, kv_prompt, ws, layers, flags, s, f8));
  if (hidden_out)
    D3PM_CHECK_HIP(hipMemcpyAsync(hidden_out, ws.x, static_cast<size_t>(batch) * sh->canvas * sh->d_model * dtype_size(sh->dtype),
                                  hipMemcpyDeviceToDevice, s));
  if (logits_out) {
    const size_t es = dtype_size(sh->dtype);
    D3PM_TRY(final_logits(*sh, *w, batch, ws, ws.logits, logits_ld(*sh), flags, s));
    D3PM_CHECK_HIP(hipMemcpy2DAsync(logits_out, sh->n_classes * es, ws.logits, logits_ld(*sh) * es, sh->n_classes * es,
                                    static_cast<size_t>(batch) * sh->canvas * levels(*sh), hipMemcpyDeviceToDevice, s));
  }
  return D3PM_OK;
}

int d3pm_denoise_step(const d3pm_shape* sh, const d3pm_weights* w, int batch, const int32_t* x_t,
                      const uint8_t* frame_mask, int t, const void* film, const void* kv_text, const void* kv_prompt,
                      void* workspace, size_t workspace_bytes, void* logits_out, void* hidden_out, int only_layers,
                      uint32_t flags, void* stream) {
  return denoise_step_impl(sh, w, batch, x_t, frame_mask, t, film, kv_text, kv_prompt, workspace, workspace_bytes, logits_out,
                           hidden_out, only_layers, flags, stream, nullptr);
}

int d3pm_denoise_step_fp8(const d3pm_shape* sh, const d3pm_weights* w, const d3pm_fp8_block_weights* fp8_blocks, int batch,
                          const int32_t* x_t, const uint8_t* frame_mask, int t, const void* film, const void* kv_text,
                          const void* kv_prompt, void* workspace, size_t workspace_bytes, void* logits_out, void* hidden_out,
                          int only_layers, uint32_t flags, void* stream) {
  D3PM_REQUIRE(fp8_blocks, D3PM_E_ARG, "d3pm_denoise_step_fp8: null fp8 weights");
  return denoise_step_impl(sh, w, batch, x_t, frame_mask, t, film, kv_text, kv_prompt, workspace, workspace_bytes, logits_out,
                           hidden_out, only_layers, flags, stream, fp8_blocks);
}

int d3pm_posterior_sample(const d3pm_shape* sh, int batch, const void* logits, int logits_dtype, const int32_t* x_t,
                          int32_t* x_next, int t, const d3pm_schedule* sched, uint64_t seed, uint32_t utt0,
                          uint32_t flags, uint16_t* posterior_out, void* stream) {
  D3PM_TRY(check_shape(sh, batch));
  D3PM_REQUIRE(logits && x_t && x_next && sched && sched->d && sched->c && sched->dbar && sched->cbar, D3PM_E_ARG,
               "d3pm_posterior_sample: null pointer");
  D3PM_REQUIRE(t >= 0 && t < sched->timesteps, D3PM_E_ARG, "t=%d outside the schedule", t);
  SampleArgs a;
  a.logits = logits; a.logits_dtype = logits_dtype; a.ldl = sh->n_classes; a.x_t = x_t; a.x_next = x_next;
  a.posterior_out = posterior_out; a.rows = batch * sh->canvas * levels(*sh); a.n_classes = sh->n_classes; a.mask_id = sh->mask_id;
  a.n_q = levels(*sh);
  a.canvas = sh->canvas; a.seed = seed; a.row0 = utt0 * static_cast<uint32_t>(sh->canvas);
  a.greedy = (flags & D3PM_FLAG_GREEDY) ? 1 : 0; a.pc = make_posterior_consts(sched, t);
  return posterior_sample(a, static_cast<hipStream_t>(stream));
}

static int sample_loop_impl(const d3pm_shape* sh, const d3pm_weights* w, int batch, int32_t* x, const uint8_t* frame_mask,
                            int t_start, int t_stop, const void* film, const void* kv_text, const void* kv_prompt,
                            const d3pm_schedule* sched, uint64_t seed, uint32_t utt0, uint32_t flags, void* workspace,
                            size_t workspace_bytes, int32_t* trace, void* stream, const d3pm_fp8_block_weights* f8) {
  D3PM_TRY(check_shape(sh, batch));
  D3PM_REQUIRE(w && w->blocks && x && frame_mask && film && kv_text && kv_prompt && sched && workspace, D3PM_E_ARG,
               "d3pm_sample_loop: null pointer");
  D3PM_REQUIRE(t_start < sched->timesteps && t_start <= sh->timesteps && t_stop >= 0 && t_stop <= t_start, D3PM_E_ARG,
               "bad step range %d..%d", t_start, t_stop);
  Workspace ws = carve(*sh, batch, static_cast<char*>(workspace));
  D3PM_REQUIRE(workspace_bytes >= ws.total, D3PM_E_WORKSPACE, "workspace %zu < required %zu", workspace_bytes, ws.total);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int rows = batch * sh->canvas;
  const Ctx cx(sh->tuning);
  bool prepared = false;      // the previous iteration's sampler launch has already embedded x_t and folded fc1 for this t
  for (int t = t_start; t > t_stop; --t) {
    if (cx.prof) cx.prof->sample_now = (t % cx.prof->stride) == 0;
    D3PM_TRY(denoiser_blocks(*sh, *w, batch, x, frame_mask, t, film, kv_text, kv_prompt, ws, sh->n_layers, flags, s, f8, prepared));
    prepared = false;
    SampleArgs a;
    a.logits = ws.logits; a.logits_dtype = sh->dtype; a.ldl = logits_ld(*sh); a.x_t = x; a.x_next = x;
    a.x_next2 = trace ? trace + static_cast<size_t>(t_start - t) * rows * levels(*sh) : nullptr;
    a.rows = rows * levels(*sh); a.n_q = levels(*sh); a.n_classes = sh->n_classes; a.mask_id = sh->mask_id; a.canvas = sh->canvas; a.seed = seed;
    if (flags & D3PM_FLAG_SEED_IN_HBM) a.seed_hbm = reinterpret_cast<const uint64_t*>(static_cast<uintptr_t>(seed));
    a.row0 = utt0 * static_cast<uint32_t>(sh->canvas); a.greedy = (flags & D3PM_FLAG_GREEDY) ? 1 : 0;
    a.pc = make_posterior_consts(sched, t);
    const double fin_flops = 2.0 * rows * sh->n_classes * sh->d_model;
    (void)fin_flops;
#ifdef D3PM_ABLATIONS
    if (fused_final_sample_applies(*sh, *w, ws, flags)) {
      // final projection + posterior + draw in one kernel: the logits stay on chip (d3pm_final_sample.hip)
      ProfScope p(cx, D3PM_K_SAMPLE, s, fin_flops,
                  dtype_size(sh->dtype) * (static_cast<double>(rows) * sh->d_model + static_cast<double>(sh->n_classes) * sh->d_model) + 8.0 * rows);
      D3PM_TRY(final_sample(sh->dtype, ws.x, sh->d_model, w->final_w, w->final_b, sh->d_model, a, s));
    } else
#endif
    {
      D3PM_TRY(final_logits(*sh, *w, batch, ws, ws.logits, logits_ld(*sh), flags, s));
      ProfScope p(cx, D3PM_K_SAMPLE, s, 0.0,
                  static_cast<double>(rows) * levels(*sh) * (sh->n_classes * dtype_size(sh->dtype) + 8.0));
      NextIterPrep nx;
      if (t - 1 > t_stop && fold_active(*sh, *w, flags, f8)) {
        const size_t es = dtype_size(sh->dtype);
        nx.dtype = sh->dtype; nx.table = w->resps_emb; nx.x = ws.x; nx.stats = ws.stats; nx.frame_mask = frame_mask; nx.d = sh->d_model;
        nx.blocks = w->blocks; nx.n_layers = sh->n_layers;
        nx.film_t = at(film, static_cast<size_t>(t - 1) * sh->n_layers * 2 * sh->d_model, es);
        nx.Wf = ws.fc1f; nx.s_out = ws.fc1f_s; nx.b_out = ws.fc1f_b;
      }
      if (nx.table && posterior_sample_prep_supported(a, nx)) {
        D3PM_TRY(posterior_sample_prep(a, nx, s));       // + the embedding rows, their moments and the fc1 fold of iteration t - 1
        prepared = true;
      } else {
        D3PM_TRY(posterior_sample(a, s));
      }
    }
  }
  if (cx.prof) cx.prof->sample_now = false;
  return D3PM_OK;
}

int d3pm_sample_loop(const d3pm_shape* sh, const d3pm_weights* w, int batch, int32_t* x, const uint8_t* frame_mask,
                     int t_start, int t_stop, const void* film, const void* kv_text, const void* kv_prompt,
                     const d3pm_schedule* sched, uint64_t seed, uint32_t utt0, uint32_t flags, void* workspace,
                     size_t workspace_bytes, int32_t* trace, void* stream) {
  return sample_loop_impl(sh, w, batch, x, frame_mask, t_start, t_stop, film, kv_text, kv_prompt, sched, seed, utt0, flags,
                          workspace, workspace_bytes, trace, stream, nullptr);
}

int d3pm_sample_loop_fp8(const d3pm_shape* sh, const d3pm_weights* w, const d3pm_fp8_block_weights* fp8_blocks, int batch,
                         int32_t* x, const uint8_t* frame_mask, int t_start, int t_stop, const void* film,
                         const void* kv_text, const void* kv_prompt, const d3pm_schedule* sched, uint64_t seed, uint32_t utt0,
                         uint32_t flags, void* workspace, size_t workspace_bytes, int32_t* trace, void* stream) {
  D3PM_REQUIRE(fp8_blocks, D3PM_E_ARG, "d3pm_sample_loop_fp8: null fp8 weights");
  return sample_loop_impl(sh, w, batch, x, frame_mask, t_start, t_stop, film, kv_text, kv_prompt, sched, seed, utt0, flags,
                          workspace, workspace_bytes, trace, stream, fp8_blocks);
}

int d3pm_q_sample(const d3pm_shape* sh, int batch, const int32_t* x0, int32_t* x_out, const uint8_t* frame_mask, int t,
                  const d3pm_schedule* sched, uint64_t seed, uint32_t utt0, void* stream) {
  D3PM_TRY(check_shape(sh, batch));
  D3PM_REQUIRE(x0 && x_out && frame_mask && sched && sched->dbar && sched->cbar, D3PM_E_ARG, "d3pm_q_sample: null pointer");
  D3PM_REQUIRE(levels(*sh) == 1, D3PM_E_SHAPE, "d3pm_q_sample: the training side covers the upstream level-0 model only (n_q = 1)");
  D3PM_REQUIRE(t >= 0 && t < sched->timesteps, D3PM_E_ARG, "t=%d outside the schedule", t);
  return q_sample_launch(sh, batch, x0, x_out, frame_mask, t, sched, seed, utt0, static_cast<hipStream_t>(stream));
}

int d3pm_ce_loss_rows(const d3pm_shape* sh, int batch, const void* logits, int logits_dtype, const int32_t* targets,
                      const uint8_t* frame_mask, float* row_loss, void* stream) {
  D3PM_TRY(check_shape(sh, batch));
  D3PM_REQUIRE(logits && targets && frame_mask && row_loss, D3PM_E_ARG, "d3pm_ce_loss_rows: null pointer");
  D3PM_REQUIRE(levels(*sh) == 1, D3PM_E_SHAPE, "d3pm_ce_loss_rows: the training side covers the upstream level-0 model only (n_q = 1)");
  return ce_loss_launch(logits_dtype, logits, sh->n_classes, targets, frame_mask, sh->canvas, batch * sh->canvas,
                        sh->n_classes, row_loss, static_cast<hipStream_t>(stream));
}

int d3pm_uniform(uint64_t seed, int t, uint32_t row0, int rows, int n_classes, int stream_id, float* out, void* stream) {
  D3PM_REQUIRE(out && rows > 0 && n_classes > 0, D3PM_E_ARG, "d3pm_uniform: bad arguments");
  return uniform_launch(seed, t, row0, rows, n_classes, stream_id, out, static_cast<hipStream_t>(stream));
}

// ---- stock NAR model -------------------------------------------------------------------------------------
struct NarWs { char *x, *h, *qkv, *att, *ffn, *logits; uint8_t* mask; int32_t* key_len; size_t total; };
static NarWs carve_nar(const d3pm_nar_shape& sh, int batch, int t_max, char* base) {
  const size_t es = dtype_size(sh.dtype), n = static_cast<size_t>(batch) * t_max, d = sh.d_model;
  NarWs w{};
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off += align256(bytes); return p; };
  w.x = take(n * d * es);
  w.h = take(n * d * es);
  w.qkv = take(n * 3 * d * es);
  w.att = take(n * d * es);
  w.ffn = take(n * 4 * d * es);
  w.logits = take(n * sh.n_tokens * es);
  w.mask = reinterpret_cast<uint8_t*>(take(n));
  w.key_len = reinterpret_cast<int32_t*>(take(static_cast<size_t>(batch) * 4));
  w.total = off;
  return w;
}
static int check_nar(const d3pm_nar_shape* sh, int batch, int t_max) {
  D3PM_REQUIRE(sh && batch > 0 && t_max > 0 && sh->d_model > 0 && sh->n_heads > 0 && sh->d_model % sh->n_heads == 0 &&
                   sh->n_layers > 0 && sh->n_tokens > 1 && sh->n_prom_levels > 0 && sh->n_resp_levels > 0,
               D3PM_E_ARG, "inconsistent d3pm_nar_shape");
  D3PM_REQUIRE(sh->dtype == D3PM_F32 || sh->dtype == D3PM_F16 || sh->dtype == D3PM_BF16, D3PM_E_ARG, "bad dtype %d", sh->dtype);
  return D3PM_OK;
}

size_t d3pm_nar_workspace_bytes(const d3pm_nar_shape* sh, int batch, int t_max) {
  if (check_nar(sh, batch, t_max) != D3PM_OK) return 0;
  return carve_nar(*sh, batch, t_max, nullptr).total;
}

int d3pm_nar_level(const d3pm_nar_shape* sh, const d3pm_nar_weights* w, int batch, int t_max, const int32_t* lens,
                   const int32_t* text, int tt_max, const int32_t* prom, int tp_max, int32_t* resp, int tr_max, int level,
                   float temperature, uint64_t seed, uint32_t utt0, uint32_t flags, void* workspace, size_t workspace_bytes,
                   void* logits_out, void* stream) {
  D3PM_TRY(check_nar(sh, batch, t_max));
  D3PM_REQUIRE(w && w->blocks && w->text_emb && w->proms_emb && w->resps_emb && w->sep && w->classifier_w && w->classifier_b &&
                   w->pe && lens && text && prom && resp && workspace,
               D3PM_E_ARG, "d3pm_nar_level: null pointer");
  D3PM_REQUIRE(level >= 0 && level < sh->n_resp_levels && temperature > 0.f && w->pe_rows >= t_max && tt_max > 0 && tp_max > 0 &&
                   tr_max > 0,
               D3PM_E_ARG, "d3pm_nar_level: bad level / temperature / sizes");
  NarWs ws = carve_nar(*sh, batch, t_max, static_cast<char*>(workspace));
  D3PM_REQUIRE(workspace_bytes >= ws.total, D3PM_E_WORKSPACE, "workspace %zu < required %zu", workspace_bytes, ws.total);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int dt = sh->dtype, d = sh->d_model, n = batch * t_max, hd = d / sh->n_heads, stride = sh->n_resp_levels + 1;
  const size_t es = dtype_size(dt);
  const Ctx cx(sh->tuning);

  NarEmbedArgs e;
  e.lens = lens; e.text = text; e.tt_max = tt_max; e.prom = prom; e.tp_max = tp_max; e.n_prom_levels = sh->n_prom_levels;
  e.resp = resp; e.tr_max = tr_max; e.resp_stride = stride; e.n_given = level + 1;
  e.w_text = w->text_emb; e.w_prom = w->proms_emb; e.w_resp = w->resps_emb; e.sep = w->sep; e.pe = w->pe;
  e.x = ws.x; e.row_mask = ws.mask; e.key_len = ws.key_len; e.batch = batch; e.t_max = t_max; e.d = d; e.n_tokens = sh->n_tokens;
  D3PM_TRY(nar_embed(dt, e, s));

  for (int l = 0; l < sh->n_layers; ++l) {
    const d3pm_nar_block_weights& b = w->blocks[l];
    // x = (x + to_out(attention(AdaLN(x) * m)) * m) * m
    D3PM_TRY(adaln(dt, ws.x, ws.h, at(b.attn_norm_emb, static_cast<size_t>(level) * 2 * d, es), ws.mask, n, d, s));
    LinearArgs g;
    g.X = ws.h; g.ldx = d; g.W = b.to_qkv_w; g.Y = ws.qkv; g.ldy = 3 * d; g.M = n; g.N = 3 * d; g.K = d;
    D3PM_TRY(run_linear(cx, dt, g, flags, s));
    AttnArgs a;
    a.Q = ws.qkv; a.ldq = 3 * d; a.K = at(ws.qkv, d, es); a.V = at(ws.qkv, 2 * d, es); a.ldkv = 3 * d; a.O = ws.att; a.ldo = d;
    a.B = batch; a.Tq = t_max; a.S = t_max; a.H = sh->n_heads; a.hd = hd; a.scale = 1.0f / std::sqrt(static_cast<float>(hd));
    a.key_len = ws.key_len;
    D3PM_TRY(run_attention(cx, dt, a, flags, s));
    g = LinearArgs();
    g.X = ws.att; g.ldx = d; g.W = b.to_out_w; g.bias = b.to_out_b; g.Y = ws.x; g.ldy = d; g.R1 = ws.x; g.ldr = d;
    g.row_mask = ws.mask; g.mask_period = n; g.M = n; g.N = d; g.K = d;
    D3PM_TRY(run_linear(cx, dt, g, flags, s));
    // x = (x + ffn(AdaLN(x) * m)) * m
    D3PM_TRY(adaln(dt, ws.x, ws.h, at(b.ffn_norm_emb, static_cast<size_t>(level) * 2 * d, es), ws.mask, n, d, s));
    g = LinearArgs();
    g.X = ws.h; g.ldx = d; g.W = b.ffn0_w; g.bias = b.ffn0_b; g.Y = ws.ffn; g.ldy = 4 * d; g.M = n; g.N = 4 * d; g.K = d; g.act = ACT_GELU;
    D3PM_TRY(run_linear(cx, dt, g, flags, s));
    g = LinearArgs();
    g.X = ws.ffn; g.ldx = 4 * d; g.W = b.ffn3_w; g.bias = b.ffn3_b; g.Y = ws.x; g.ldy = d; g.R1 = ws.x; g.ldr = d;
    g.row_mask = ws.mask; g.mask_period = n; g.M = n; g.N = d; g.K = 4 * d;
    D3PM_TRY(run_linear(cx, dt, g, flags, s));
  }
  LinearArgs g;
  g.X = ws.x; g.ldx = d; g.W = w->classifier_w; g.bias = w->classifier_b; g.Y = ws.logits; g.ldy = sh->n_tokens; g.M = n;
  g.N = sh->n_tokens; g.K = d;
  D3PM_TRY(run_linear(cx, dt, g, flags, s));
  if (logits_out)
    D3PM_CHECK_HIP(hipMemcpyAsync(logits_out, ws.logits, static_cast<size_t>(n) * sh->n_tokens * es, hipMemcpyDeviceToDevice, s));
  return nar_sample(dt, ws.logits, sh->n_tokens, lens, resp, tr_max, stride, t_max, sh->n_tokens, level, temperature, seed, utt0,
                    (flags & D3PM_FLAG_GREEDY) ? 1 : 0, batch, s);
}

int d3pm_op_linear(int dtype, int family, const void* X, int ldx, const void* W, const void* bias, void* Y, int ldy,
                   const void* R1, const void* R2, int ldr, const uint8_t* row_mask, int mask_period, int M, int N, int K,
                   int act, const d3pm_tuning* tuning, void* stream) {
  D3PM_REQUIRE(X && W && Y && M > 0 && N > 0 && K > 0, D3PM_E_ARG, "d3pm_op_linear: bad arguments");
  const Ctx cx(tuning);
  LinearArgs g;
  g.tune = tuning;
  g.X = X; g.ldx = ldx; g.W = W; g.bias = bias; g.Y = Y; g.ldy = ldy; g.R1 = R1; g.R2 = R2; g.ldr = ldr;
  g.row_mask = row_mask; g.mask_period = mask_period > 0 ? mask_period : 1; g.M = M; g.N = N; g.K = K; g.act = act;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (family == 1) return generic_linear(dtype, g, s);
  if (family == 2) {
    D3PM_REQUIRE(mfma_linear_supported(dtype, g), D3PM_E_SHAPE, "d3pm_op_linear: shape not supported by the MFMA kernel");
    return mfma_linear(dtype, g, s);
  }
  return run_linear(cx, dtype, g, 0, s);
}

int d3pm_op_quantize_mx(int dtype, const void* X, int ldx, void* X8, void* SX, int M, int K, void* stream) {
  D3PM_REQUIRE(X && X8 && SX && M > 0 && K > 0, D3PM_E_ARG, "d3pm_op_quantize_mx: bad arguments");
  return quantize_mx(dtype, X, ldx, static_cast<uint8_t*>(X8), static_cast<uint8_t*>(SX), M, K, static_cast<hipStream_t>(stream));
}

int d3pm_op_layernorm_mx(int dtype, const void* X, void* Y8, void* SX, const void* w, const void* b, const void* film, int M, int d,
                         float eps, void* stream) {
  D3PM_REQUIRE(X && Y8 && SX && w && b && M > 0, D3PM_E_ARG, "d3pm_op_layernorm_mx: bad arguments");
  return layernorm_mx(dtype, X, static_cast<uint8_t*>(Y8), static_cast<uint8_t*>(SX), w, b, film, nullptr, nullptr, nullptr, nullptr, M, d,
                      eps, static_cast<hipStream_t>(stream));
}

int d3pm_op_linear_mx(int out_dtype, const void* X8, int ldx, const void* SX, const void* W8, const void* SW, const void* bias, void* Y,
                      int ldy, const void* R1, int ldr, const uint8_t* row_mask, int mask_period, void* Y8, void* SY, int M, int N, int K,
                      int act, const d3pm_tuning* tuning, void* stream) {
  D3PM_REQUIRE(X8 && SX && W8 && SW && (Y || Y8) && M > 0 && N > 0 && K > 0, D3PM_E_ARG, "d3pm_op_linear_mx: bad arguments");
  MxLinearArgs m;
  m.X8 = X8; m.ldx = ldx; m.SX = SX; m.W8 = W8; m.SW = SW; m.bias = bias; m.Y = Y; m.ldy = ldy; m.R1 = R1; m.ldr = ldr;
  m.row_mask = row_mask; m.mask_period = mask_period > 0 ? mask_period : 1; m.Y8 = Y8; m.SY = SY; m.M = M; m.N = N; m.K = K; m.act = act;
  m.tune = tuning;
  D3PM_REQUIRE(mx_linear_supported(out_dtype, m), D3PM_E_SHAPE,
               "d3pm_op_linear_mx: needs M a multiple of 192, N of 128, K of 512, a 16-bit output type, 16-byte aligned operands and "
               "one of the epilogues plain / GELU / R1 / R1 + mask (MX output: plain / GELU)");
  return mx_linear(out_dtype, m, static_cast<hipStream_t>(stream));
}

int d3pm_op_attention(int dtype, int family, const void* Q, int ldq, const void* K, const void* V, int ldkv, void* O,
                      int ldo, int B, int Tq, int S, int H, int hd, float scale, const d3pm_tuning* tuning, void* stream) {
  D3PM_REQUIRE(Q && K && V && O && B > 0 && Tq > 0 && S > 0 && H > 0 && hd > 0, D3PM_E_ARG, "d3pm_op_attention: bad arguments");
  const Ctx cx(tuning);
  AttnArgs a;
  a.tune = tuning;
  a.Q = Q; a.ldq = ldq; a.K = K; a.V = V; a.ldkv = ldkv; a.O = O; a.ldo = ldo; a.B = B; a.Tq = Tq; a.S = S; a.H = H;
  a.hd = hd; a.scale = scale;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (family == 1) return generic_attention(dtype, a, s);
  if (family == 2) {
    D3PM_REQUIRE(mfma_attention_supported(dtype, a), D3PM_E_SHAPE, "d3pm_op_attention: shape not supported by the MFMA kernel");
    return mfma_attention(dtype, a, s);
  }
  return run_attention(cx, dtype, a, 0, s);
}

int d3pm_op_attention_pair(int dtype, const void* Q1, const void* K1, const void* V1, void* O1, int S1, const void* Q2, const void* K2,
                           const void* V2, void* O2, int S2, int ldq, int ldkv, int ldo, int B, int Tq, int H, int hd, float scale,
                           const d3pm_tuning* tuning, void* stream) {
  D3PM_REQUIRE(Q1 && K1 && V1 && O1 && Q2 && K2 && V2 && O2 && B > 0 && Tq > 0 && S1 > 0 && S2 > 0 && H > 0 && hd > 0, D3PM_E_ARG,
               "d3pm_op_attention_pair: bad arguments");
  const Ctx cx(tuning);
  AttnArgs a;
  a.tune = tuning;
  a.Q = Q1; a.ldq = ldq; a.K = K1; a.V = V1; a.ldkv = ldkv; a.O = O1; a.ldo = ldo; a.B = B; a.Tq = Tq; a.S = S1; a.H = H;
  a.hd = hd; a.scale = scale;
  a.Q2 = Q2; a.K2 = K2; a.V2 = V2; a.O2 = O2; a.S2 = S2;
  return run_attention(cx, dtype, a, 0, static_cast<hipStream_t>(stream));
}

int d3pm_op_layernorm(int dtype, const void* X, void* Y, const void* w, const void* b, const void* film, int M, int d,
                      float eps, void* stream) {
  D3PM_REQUIRE(X && Y && w && b && M > 0 && d > 0, D3PM_E_ARG, "d3pm_op_layernorm: bad arguments");
  LayerNormArgs ln;
  ln.X = X; ln.Y = Y; ln.w = w; ln.b = b; ln.film = film; ln.M = M; ln.d = d; ln.eps = eps;
  const Ctx cx(nullptr);
  return run_layernorm(cx, dtype, ln, 0, static_cast<hipStream_t>(stream));
}

#ifdef D3PM_ABLATIONS
int d3pm_op_linear_lnpro(int dtype, const void* X, const void* W, const void* bias, void* Y, int M, int N, int act, const void* ln_w,
                         const void* ln_b, const void* ln2_w, const void* ln2_b, const void* film, float eps, void* stream) {
  D3PM_REQUIRE(X && W && Y && ln_w && ln_b && M > 0 && N > 0, D3PM_E_ARG, "d3pm_op_linear_lnpro: bad arguments");
  LinearArgs g;
  g.X = X; g.ldx = 512; g.W = W; g.bias = bias; g.Y = Y; g.ldy = N; g.M = M; g.N = N; g.K = 512; g.act = act;
  LnPrologue lp;
  lp.w = ln_w; lp.b = ln_b; lp.w2 = ln2_w; lp.b2 = ln2_b; lp.film = film; lp.eps = eps; lp.period = ln2_w ? M / 2 : 0;
  D3PM_REQUIRE(panel64_ln_supported(dtype, g, lp), D3PM_E_SHAPE,
               "d3pm_op_linear_lnpro: needs a 16-bit dtype, 16-byte aligned operands, N a multiple of 8, act 0 / 1 and, with a second "
               "LayerNorm, M = 2 x a multiple of 64 rows and no FiLM");
  return ln_prologue_linear(dtype, g, lp, static_cast<hipStream_t>(stream));
}

#endif

int d3pm_op_linear_rowpanel(int dtype, const void* X, const void* X2, int ldx, const void* W, const void* bias, void* Y, const void* R1,
                            const uint8_t* row_mask, int mask_period, int M, int K, const void* ln_w, const void* ln_b, void* ln_y,
                            const void* ln2_w, const void* ln2_b, void* ln2_y, const void* film, float eps, void* ln_sx, void* ln2_sx,
                            void* stream) {
  D3PM_REQUIRE(X && W && bias && Y && R1 && ln_w && ln_b && ln_y && M > 0 && K > 0, D3PM_E_ARG, "d3pm_op_linear_rowpanel: bad arguments");
  LinearArgs g;
  g.X = X; g.ldx = ldx; g.W = W; g.bias = bias; g.Y = Y; g.ldy = 512; g.R1 = R1; g.ldr = 512; g.row_mask = row_mask;
  g.mask_period = mask_period > 0 ? mask_period : 1; g.M = M; g.N = 512; g.K = K;
  RowPanelFuse f;
  f.X2 = X2; f.lnw = ln_w; f.lnb = ln_b; f.lny = ln_y; f.lnw2 = ln2_w; f.lnb2 = ln2_b; f.lny2 = ln2_y; f.film = film; f.eps = eps;
  f.sx = ln_sx; f.sx2 = ln2_sx;
  D3PM_REQUIRE(row_panel_supported(dtype, g, f), D3PM_E_SHAPE,
               "d3pm_op_linear_rowpanel: needs a 16-bit dtype, M a multiple of 96, K a multiple of 128 (>= 256), 16-byte aligned "
               "operands and one of the three fused forms (include/d3pm_hip.h)");
  return row_panel_linear(dtype, g, f, static_cast<hipStream_t>(stream));
}

#ifdef D3PM_ABLATIONS
int d3pm_op_final_sample(const d3pm_shape* sh, const d3pm_weights* w, int batch, const void* hidden, const int32_t* x_t,
                         int32_t* x_next, int t, const d3pm_schedule* sched, uint64_t seed, uint32_t utt0, uint32_t flags,
                         void* stream) {
  D3PM_TRY(check_shape(sh, batch));
  D3PM_REQUIRE(w && w->final_w && hidden && x_t && x_next && sched && sched->d && sched->c && sched->dbar && sched->cbar, D3PM_E_ARG,
               "d3pm_op_final_sample: null pointer");
  D3PM_REQUIRE(t >= 0 && t < sched->timesteps, D3PM_E_ARG, "t=%d outside the schedule", t);
  D3PM_REQUIRE(final_sample_supported(sh->dtype, sh->n_classes, sh->d_model, hidden, sh->d_model, w->final_w), D3PM_E_SHAPE,
               "d3pm_op_final_sample: needs a 16-bit model, 1025 classes and d_model a multiple of 32 (<= 1024)");
  SampleArgs a;
  a.x_t = x_t; a.x_next = x_next; a.rows = batch * sh->canvas; a.n_classes = sh->n_classes; a.mask_id = sh->mask_id;
  a.canvas = sh->canvas; a.seed = seed; a.row0 = utt0 * static_cast<uint32_t>(sh->canvas);
  a.greedy = (flags & D3PM_FLAG_GREEDY) ? 1 : 0; a.pc = make_posterior_consts(sched, t);
  return final_sample(sh->dtype, hidden, sh->d_model, w->final_w, w->final_b, sh->d_model, a, static_cast<hipStream_t>(stream));
}

#endif

int d3pm_op_cond_embed(int dtype, int which, const int32_t* tokens, int n_levels, const void* tables, const void* pe, void* y, int rows,
                       int s_prompt, int d, int n_classes, void* stream) {
  D3PM_REQUIRE(tokens && tables && pe && y && rows > 0 && d > 0 && n_classes > 0, D3PM_E_ARG, "d3pm_op_cond_embed: bad arguments");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (which == 0) return cond_embed_text(dtype, tokens, tables, pe, y, rows, d, n_classes, s);
  D3PM_REQUIRE(n_levels > 0 && s_prompt > 0, D3PM_E_ARG, "d3pm_op_cond_embed: bad prompt arguments");
  return cond_embed_prompt(dtype, tokens, n_levels, tables, pe, y, rows, s_prompt, d, n_classes, s);
}

void d3pm_tuning_default(d3pm_tuning* t) {
  if (t) *t = tune_of(nullptr);
}

#ifdef D3PM_ABLATIONS
int d3pm_ab_set(int knob, int value) {
  AbKnobs& k = ab_knobs();
  static const int big_modes[] = {0, 1, 3, 5, 9, 17, 32, 33, 81, 145, 209, 257, 465, 513, 1025, 2049, 4129, 8193, 16385, 24577, 32769};
  if (knob == D3PM_AB_GEMM_BIG_MODE) {
    for (int m : big_modes)
      if (m == value) { k.big_mode = value; return D3PM_OK; }
  }
  if (knob == D3PM_AB_ATTN_ARM && (value == 0 || value == 3 || (value >= 100 && value <= 164) || value == 228 || value == 201 || value == 202 || value == 300 || value == 301 || (value >= 320 && value <= 324))) { k.attn_arm = value; return D3PM_OK; }
  if (knob == D3PM_AB_GEMM_RING && (value == 0 || value == 1)) { k.ring = value; return D3PM_OK; }
  if (knob == D3PM_AB_GELU_TABLE && (value == 0 || value == 1)) { k.gelu_table = value; return D3PM_OK; }
  if (knob == D3PM_AB_LN_PROLOGUE && (value == 0 || value == 1)) { k.ln_prologue = value; return D3PM_OK; }
  if (knob == D3PM_AB_FUSED_FINAL_SAMPLE && (value == 0 || value == 1)) { k.fused_final_sample = value; return D3PM_OK; }
  set_error("d3pm_ab_set: unknown knob %d / value %d", knob, value);
  return D3PM_E_ARG;
}

int d3pm_debug_gemm_clock(unsigned long long* clocks_and_ticks) {
  D3PM_REQUIRE(clocks_and_ticks, D3PM_E_ARG, "d3pm_debug_gemm_clock: null pointer");
  return read_big_gemm_stamp(clocks_and_ticks);
}

int d3pm_debug_attn32_stamps(unsigned long long* out, int n) {
  D3PM_REQUIRE(out && n > 0, D3PM_E_ARG, "d3pm_debug_attn32_stamps: bad arguments");
  return read_attn32_stamps(out, n);
}
#endif

int d3pm_prof_create(int kclass, int max_events, d3pm_prof** out) {
  D3PM_REQUIRE(out && kclass >= 0 && kclass <= D3PM_K_COUNT && max_events > 0, D3PM_E_ARG, "d3pm_prof_create: bad arguments");
  d3pm_prof* p = new (std::nothrow) d3pm_prof();
  D3PM_REQUIRE(p, D3PM_E_ARG, "d3pm_prof_create: out of host memory");
  p->ev.resize(static_cast<size_t>(max_events) * 2);
  p->cls.assign(static_cast<size_t>(max_events), -1);
  for (size_t i = 0; i < p->ev.size(); ++i) {
    if (hipEventCreate(&p->ev[i]) != hipSuccess) {
      set_error("d3pm_prof_create: hipEventCreate failed");
      for (size_t j = 0; j < i; ++j) (void)hipEventDestroy(p->ev[j]);     // the events created so far
      delete p;
      return D3PM_E_HIP;
    }
  }
  p->kclass = kclass;
  *out = p;
  return D3PM_OK;
}

int d3pm_prof_read_class(d3pm_prof* p, int kclass, int* launches, double* total_ms, double* flops, double* bytes) {
  D3PM_REQUIRE(p && kclass >= 0 && kclass < D3PM_K_COUNT, D3PM_E_ARG, "d3pm_prof_read_class: bad arguments");
  double ms = 0;
  int n = 0;
  for (int i = 0; i + 1 < p->used; i += 2) {
    if (p->cls[i / 2] != kclass) continue;
    D3PM_CHECK_HIP(hipEventSynchronize(p->ev[i + 1]));
    float m = 0;
    D3PM_CHECK_HIP(hipEventElapsedTime(&m, p->ev[i], p->ev[i + 1]));
    ms += m;
    ++n;
  }
  if (launches) *launches = n;
  if (total_ms) *total_ms = ms;
  if (flops) *flops = p->flops[kclass];
  if (bytes) *bytes = p->bytes[kclass];
  return D3PM_OK;
}

int d3pm_prof_read(d3pm_prof* p, int* launches, double* total_ms, double* flops, double* bytes) {
  D3PM_REQUIRE(p, D3PM_E_ARG, "d3pm_prof_read: null handle");
  int n = 0;
  double ms = 0, fl = 0, by = 0;
  for (int c = 0; c < D3PM_K_COUNT; ++c) {
    int nc = 0;
    double mc = 0, fc = 0, bc = 0;
    D3PM_TRY(d3pm_prof_read_class(p, c, &nc, &mc, &fc, &bc));
    n += nc; ms += mc; fl += fc; by += bc;
  }
  if (launches) *launches = n;
  if (total_ms) *total_ms = ms;
  if (flops) *flops = fl;
  if (bytes) *bytes = by;
  p->used = 0;
  for (int c = 0; c < D3PM_K_COUNT; ++c) p->flops[c] = p->bytes[c] = 0;
  return D3PM_OK;
}

int d3pm_prof_destroy(d3pm_prof* p) {
  if (!p) return D3PM_OK;
  for (auto& e : p->ev) (void)hipEventDestroy(e);
  delete p;
  return D3PM_OK;
}

}  // extern "C"
