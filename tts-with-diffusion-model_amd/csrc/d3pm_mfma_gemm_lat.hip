// d3pm_mfma_gemm_lat.hip -- latency GEMM for one or two utterances (M <= 1536 rows): 64 x 64 tiles, whole-K panels.
//
//   Y[M][N] = epilogue(X[M][K] . W[N][K]^T + bias)      same contract and epilogue as d3pm_mfma_gemm.hip
//
// replaces the nn.Linear / MultiheadAttention projections of DiTBlock.forward
// (/root/reference/vall_e/vall_e/ar_discrete.py:132,138,142,159,776) when ONE utterance is sampled (the p50-latency
// half of the BASELINE.json metric).  At M = 768 every 128 x 128 launch costs ~10 us whatever its size
// (profiles/round1_i_microbench.txt): 24 .. 72 workgroups on 256 CUs, each walking its eight k-steps one DMA round
// trip after the other.  The time is latency, so this kernel removes the chain instead of shortening its links:
//   * 64 x 64 output tiles: 96 .. 384 workgroups for the block's projections at M = 768;
//   * the operand panels of a tile for 256 k (X [64][256] + W [64][256] = 64 KiB) fit one LDS buffer, and there are two:
//     for K = 512 EVERY DMA piece of the tile is issued before the first wait (32 per wave, all in flight together),
//     so the kernel pays one memory round trip, not eight; longer K (fc2: 2048) streams 256-k rounds through the
//     two buffers;
//   * within a round the four k-steps run back to back, no barriers (the whole round has landed);
//   * same swizzled 128-byte-row LDS image per 64-k sub-tile, same D = W_frag . X_frag^T orientation, same k order
//     and the same epilogue code as the other schedules: bit-identical results.
//   * LayerNorm prologue (LNPRO, K = d_model = 512): the projections fed by a LayerNorm (ar_discrete.py:131-132 norm1 -> qkv,
//     :136-142 norm2 | norm22 -> the cross-attention queries, :145-159 norm3 + FiLM -> fc1) take the RESIDUAL STREAM as their
//     operand: the tile's 64 whole rows are in LDS anyway, so each wave normalises 16 of them in place -- lane L owns the
//     8-element chunk L of a row, exactly layernorm_vec's layout, same arithmetic and reduction order, same bits -- before the
//     MFMAs start.  Every column tile of a row block redoes the rows' LayerNorm (8 .. 32 x redundant, ~1 us), which at one
//     utterance is far cheaper than the separate launch it removes: 18 of the 67 launches of a diffusion iteration.
//   * two products through one weight panel (DUAL, K = d_model = 512, 32 x 64 tiles): the text and prompt cross-attention outputs
//     both go through cross_attn.out_proj (ar_discrete.py:138,142).  The tile's W panel is in LDS for the whole kernel anyway, so
//     the second operand's panel (X2, 32 KiB) rides in beside it and a second pass of the same k-steps gives
//     x' = rn(rn(R1 + rn(X W^T + b)) + rn(X2 W^T + b)) -- the R1 + R2 epilogue of the two-launch form with R2 taken from
//     registers: same bits, one launch (and one round trip of the intermediate through HBM) less per DiT block.
#include "d3pm_kernels.h"
#include "d3pm_mfma_tile.h"

namespace d3pm {
namespace {

[[maybe_unused]] constexpr int LT = 64;                       // tile rows and columns of the base geometry
constexpr int KC = 256;                      // k per round
// Tile geometries (TM x TN, four waves as 2 x 2): 64 x 64 is the base; 96 x 64 turns the 288 / 384 tiles of the qkv / fc1
// projections of one utterance (two rounds over 256 CUs at one workgroup per CU) into 192 / 256 (one round, one DMA flight);
// 32 x 64 gives the N = 512 projections (out-proj, fc2: 96 tiles of 64 x 64) 192 workgroups with half the X panel each.
template <int TM, int TN> struct LatGeom {
  static constexpr int XSUB = TM * ROW_BYTES, WSUB = TN * ROW_BYTES;     // one [rows][64 k] sub-tile of each operand
  static constexpr int XOPER = (KC / BK) * XSUB, WOPER = (KC / BK) * WSUB;
  static constexpr int BUF = XOPER + WOPER;                              // X + W of a round
};
constexpr int BUF_BYTES = LatGeom<64, 64>::BUF;     // 64 KiB (the LayerNorm prologue runs on the base geometry)
constexpr int SUB_BYTES = LatGeom<64, 64>::XSUB;

__device__ __forceinline__ int xcd_remap_lat(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <typename T> struct LnProArgs {
  const T* w; const T* b; const T* w2; const T* b2; const T* film; float eps; int period;
};

template <typename T, int EPI, bool LNPRO = false, int TM = 64, int TN = 64, bool DUAL = false>
__global__ __launch_bounds__(256, 1) void gemm_mfma_panel64(const T* __restrict__ X, int ldx, const T* __restrict__ W,
                                                            const T* __restrict__ bias, T* Y, int ldy, const T* R1,
                                                            const T* R2, int ldr, const uint8_t* __restrict__ row_mask,
                                                            int mask_period, int M, int N, int K, int n_tiles,
                                                            const uint16_t* __restrict__ gelu_tab_g, LnProArgs<T> ln,
                                                            EpiFold ef, const T* __restrict__ X2 = nullptr) {
  using G = LatGeom<TM, TN>;
  static_assert(!LNPRO || (TM == 64 && TN == 64), "the LayerNorm prologue is written for the base geometry");
  static_assert(!DUAL || (!LNPRO && (EPI & ~EPI_STATS) == EPI_R1 && TN == 64), "two products: whole tiles, K = 2 rounds, x' = (R1 + h) + y2");
  static_assert(TM % 32 == 0 && TN % 64 == 0, "four waves as 2 x 2, column blocks regrouped in pairs");
  constexpr int XP = TM / 32, WP = TN / 32;       // DMA pieces (8 rows) per wave and sub-tile = MFMA row / column blocks per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const uint16_t* gelu_tab = nullptr;
  if constexpr ((EPI & EPI_GELU) != 0 && std::is_same<T, bf16>::value && 2 * G::BUF + GELU_TAB_BYTES <= 160 * 1024) {
    if (gelu_tab_g) {                       // bf16 GELU by table lookup (d3pm_mfma_tile.h); published by the first barrier
      gelu_table_to_lds(gelu_tab_g, smem + 2 * G::BUF, tid, 256);
      gelu_tab = reinterpret_cast<const uint16_t*>(smem + 2 * G::BUF);
    }
  }
  const int bid = xcd_remap_lat(blockIdx.x, gridDim.x);
  const int m0 = (bid / n_tiles) * TM, n0 = (bid % n_tiles) * TN;
  const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem));
  // DMA piece (sub-tile s, rows 8j .. 8j+7): a wave takes the pieces j = wave + 4 i of every sub-tile of both operands
  // (swizzle key (row >> 1) & 7: one per-lane offset per piece)
  const int lrow = lane >> 3;
  const T* gx[XP];
  const T* gw[WP];
#pragma unroll
  for (int i = 0; i < XP; ++i) {
    const int row = 8 * (wave + 4 * i) + lrow;
    int mr = m0 + row;
    mr = mr < M ? mr : M - 1;                // ragged edges: clamped loads, predicated stores
    if constexpr (LNPRO) {
      if (ln.period) mr %= ln.period;        // rows >= period: the same source rows under the second LayerNorm (norm2 | norm22)
    }
    gx[i] = X + static_cast<size_t>(mr) * ldx + ((lane & 7) ^ ((row >> 1) & 7)) * 8;
  }
#pragma unroll
  for (int i = 0; i < WP; ++i) {
    const int row = 8 * (wave + 4 * i) + lrow;
    int nr = n0 + row;
    nr = nr < N ? nr : N - 1;
    gw[i] = W + static_cast<size_t>(nr) * K + ((lane & 7) ^ ((row >> 1) & 7)) * 8;
  }
  // the epilogue's operands (bias, residuals, frame mask) are requested FIRST: they are older than every DMA piece, so the
  // counted waits below cover them, and they have landed long before the last MFMA instead of starting a new round trip there
  EpiPre<T, WP, XP> pre;
  if constexpr (!DUAL) epilogue_prefetch<T, EPI, WP, XP>(pre, bias, R1, R2, ldr, row_mask, mask_period, M, N, m0 + wm * (TM / 2), n0 + wn * (TN / 2), lane, &ef);
  [[maybe_unused]] Pack8<T> dual_r1;
  [[maybe_unused]] EpiPre<T, WP, XP> dual_pre;
  if constexpr (DUAL) {
    epilogue_prefetch<T, 0, WP, XP>(dual_pre, bias, nullptr, nullptr, ldr, nullptr, 1, M, N, m0 + wm * (TM / 2), n0 + wn * (TN / 2), lane);
    dual_r1 = *reinterpret_cast<const Pack8<T>*>(R1 + static_cast<size_t>(m0 + wm * (TM / 2) + (lane & 15)) * ldr + n0 + wn * (TN / 2) + epilogue_nq(lane));
  }
  constexpr int PIECES = (KC / BK) * (XP + WP);       // per wave and round
  constexpr int PIECES2 = DUAL ? 2 * (KC / BK) * XP : 0;   // the second operand's panel (both rounds), issued last
  auto issue_round = [&](int r, int buf) {
    const uint32_t base = lds_base + buf * G::BUF;
#pragma unroll
    for (int s = 0; s < KC / BK; ++s) {
#pragma unroll
      for (int i = 0; i < XP; ++i) glds16_asm(gx[i] + r * KC + s * BK, base + s * G::XSUB + (wave + 4 * i) * 1024);
#pragma unroll
      for (int i = 0; i < WP; ++i) glds16_asm(gw[i] + r * KC + s * BK, base + G::XOPER + s * G::WSUB + (wave + 4 * i) * 1024);
    }
  };
  floatx4 acc[WP][XP];
#pragma unroll
  for (int a = 0; a < WP; ++a)
#pragma unroll
    for (int b = 0; b < XP; ++b) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fch = lane >> 4;
  const int rounds = K / KC;                 // K is a multiple of 256 (launcher)
  issue_round(0, 0);
  if (rounds > 1) issue_round(1, 1);
  if constexpr (DUAL) {                      // K = 2 rounds (launcher): X2's whole-K panel behind the two buffers
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int s = 0; s < KC / BK; ++s)
#pragma unroll
        for (int i = 0; i < XP; ++i) {
          const int row = 8 * (wave + 4 * i) + lrow;       // whole tiles (launcher): no clamp
          glds16_asm(X2 + static_cast<size_t>(m0 + row) * ldx + ((lane & 7) ^ ((row >> 1) & 7)) * 8 + r * KC + s * BK,
                     lds_base + 2 * G::BUF + r * G::XOPER + s * G::XSUB + (wave + 4 * i) * 1024);
        }
  }
  if constexpr (LNPRO) {                     // K = 512: both rounds are the whole rows
    const bool second = ln.period && m0 >= ln.period;
    const T* lw = second ? ln.w2 : ln.w;
    const T* lb = second ? ln.b2 : ln.b;
    const Pack8<T> wv = *reinterpret_cast<const Pack8<T>*>(lw + lane * 8), bv = *reinterpret_cast<const Pack8<T>*>(lb + lane * 8);
    Pack8<T> sc, sh;
    if (ln.film) { sc = *reinterpret_cast<const Pack8<T>*>(ln.film + lane * 8); sh = *reinterpret_cast<const Pack8<T>*>(ln.film + 512 + lane * 8); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // every wave's pieces of both rounds have landed
    // chunk c = lane of a row: round c >> 5, sub-tile (c >> 3) & 3, 16-byte chunk c & 7 of the swizzled 128-byte row
    char* const chunk_base = smem + (lane >> 5) * BUF_BYTES + ((lane >> 3) & 3) * SUB_BYTES;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
      const int row = wave * 16 + i;
      Pack8<T>* px = reinterpret_cast<Pack8<T>*>(chunk_base + lds_off(row, lane & 7));
      const Pack8<T> raw = *px;
      float v[8], sum = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[e] = static_cast<float>(raw.v[e]); sum += v[e]; }
      const float mean = wave_sum_up(sum) / 512.0f;
      float q = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float t = v[e] - mean; q += t * t; }
      const float rstd = rsqrtf(wave_sum_up(q) / 512.0f + ln.eps);
      Pack8<T> o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o.v[e] = static_cast<T>((v[e] - mean) * rstd * static_cast<float>(wv.v[e]) + static_cast<float>(bv.v[e]));
      if (ln.film) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float gg = rn<T>(1.0f + static_cast<float>(sc.v[e]));
          o.v[e] = static_cast<T>(rn<T>(static_cast<float>(o.v[e]) * gg) + static_cast<float>(sh.v[e]));
        }
      }
      *px = o;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  // the four k-steps of one round: X sub-tiles at bx, W sub-tiles at bw
  auto round_product = [&](const char* bx, const char* bw) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < KC / BK; ++s)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        uint4 fx[XP], fw[WP];
#pragma unroll
        for (int q = 0; q < XP; ++q) fx[q] = *reinterpret_cast<const uint4*>(bx + s * G::XSUB + lds_off(wm * (TM / 2) + q * 16 + frow, ks * 4 + fch));
#pragma unroll
        for (int q = 0; q < WP; ++q) fw[q] = *reinterpret_cast<const uint4*>(bw + s * G::WSUB + lds_off(wn * (TN / 2) + q * 16 + frow, ks * 4 + fch));
#pragma unroll
        for (int nt = 0; nt < WP; ++nt)
#pragma unroll
          for (int mt = 0; mt < XP; ++mt) acc[nt][mt] = mma<T>(fw[nt], fx[mt], acc[nt][mt]);
      }
  };
  for (int r = 0; r < rounds; ++r) {
    const int buf = r & 1;
    if (r + 1 < rounds) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES + PIECES2) : "memory");   // the next round's pieces may stay in flight
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES2) : "memory");
    __builtin_amdgcn_s_barrier();            // every wave's pieces of round r have landed (LNPRO: and every row is normalised)
    __builtin_amdgcn_sched_barrier(0);
    const char* bx = smem + buf * G::BUF;
    round_product(bx, bx + G::XOPER);
    __builtin_amdgcn_sched_barrier(0);
    if (r + 2 < rounds) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();          // every wave has finished reading this buffer
      issue_round(r + 2, buf);
    }
  }
  if constexpr (DUAL) {
    // h = rn(X W^T + b) packed in registers, then the second product over the same W panel (both rounds are still in LDS)
    static_assert(XP == 1 && WP == 2, "32 x 64 tiles: one 16-byte group per lane");
    uintx4 h1p[1], y2p[1];
    epilogue_store<T, 0, WP, XP, true, true, false, true>(acc, bias, Y, ldy, nullptr, nullptr, ldr, nullptr, 1, M, N, m0 + wm * (TM / 2), n0 + wn * (TN / 2), lane, h1p, nullptr, &dual_pre);
#pragma unroll
    for (int a = 0; a < WP; ++a)
#pragma unroll
      for (int b = 0; b < XP; ++b) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();            // every wave's X2 pieces have landed
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 2; ++r) round_product(smem + 2 * G::BUF + r * G::XOPER, smem + r * G::BUF + G::XOPER);
    __builtin_amdgcn_sched_barrier(0);
    epilogue_store<T, 0, WP, XP, true, true, false, true>(acc, bias, Y, ldy, nullptr, nullptr, ldr, nullptr, 1, M, N, m0 + wm * (TM / 2), n0 + wn * (TN / 2), lane, y2p, nullptr, &dual_pre);
    // x' = rn(rn(R1 + h) + y2): the R1 + R2 epilogue of the two-launch form with R2 = h from registers
    const Pack8<T> r1 = dual_r1;
    const Pack8<T> hh = __builtin_bit_cast(Pack8<T>, h1p[0]), yy = __builtin_bit_cast(Pack8<T>, y2p[0]);
    Pack8<T> o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o.v[i] = static_cast<T>(rn<T>(static_cast<float>(r1.v[i]) + static_cast<float>(hh.v[i])) + static_cast<float>(yy.v[i]));
    *reinterpret_cast<Pack8<T>*>(Y + static_cast<size_t>(m0 + wm * (TM / 2) + (lane & 15)) * ldy + n0 + wn * (TN / 2) + epilogue_nq(lane)) = o;
    if constexpr ((EPI & EPI_STATS) != 0) {      // the moments of the new rows, in the order every epilogue uses (d3pm_mfma_tile.h: emit_stats)
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = static_cast<float>(o.v[i]);
      part_stats_store(v, ef.stats_out, static_cast<size_t>(m0 + wm * (TM / 2) + (lane & 15)), N, n0 + wn * (TN / 2), lane >> 4, true);
    }
    return;
  }
  epilogue_store<T, EPI, WP, XP, false, false, false, true>(acc, bias, Y, ldy, R1, R2, ldr, row_mask, mask_period, M, N, m0 + wm * (TM / 2),
                                                            n0 + wn * (TN / 2), lane, nullptr, gelu_tab, &pre, &ef);
}

inline bool aligned16l(const void* p) { return (reinterpret_cast<uintptr_t>(p) % 16) == 0; }

}  // namespace

bool panel64_linear_supported(int dtype, const LinearArgs& a) {
  if (dtype != D3PM_F16 && dtype != D3PM_BF16) return false;
  if (a.M < 1 || a.N < 16 || a.K < KC || a.K % KC != 0) return false;
  if (a.ldx % 8 != 0 || a.ldy % 8 != 0 || !aligned16l(a.X) || !aligned16l(a.W) || !aligned16l(a.Y)) return false;
  if (a.R1 && (a.ldr % 8 != 0 || a.N % 8 != 0 || !aligned16l(a.R1))) return false;
  if (a.R2 && (!a.R1 || !aligned16l(a.R2))) return false;
  const bool gelu = a.act == ACT_GELU, r1 = a.R1 != nullptr, r2 = a.R2 != nullptr, mk = a.row_mask != nullptr;
  if (a.act != ACT_NONE && !gelu) return false;
  if (gelu && (r1 || mk)) return false;       // instantiated epilogues: plain, GELU, R1, R1+R2, R1+mask; LNF, LNF+GELU; R1 / R2 / R1+mask + STATS
  if (mk && (!r1 || r2)) return false;
  return fold_args_ok(a);
}

#ifdef D3PM_ABLATIONS
const uint16_t* gelu_table_device(hipStream_t s);
int gelu_table_enabled();

// the LayerNorm-prologue form: X is the un-normalised residual stream [period or M][512]
bool panel64_ln_supported(int dtype, const LinearArgs& a, const LnPrologue& ln) {
  if (!panel64_linear_supported(dtype, a) || a.K != 512 || a.R1 || a.row_mask || !ln.w || !ln.b) return false;
  if ((ln.w2 != nullptr) != (ln.b2 != nullptr) || (ln.w2 != nullptr) != (ln.period > 0)) return false;
  if (ln.period && (ln.period % LT != 0 || a.M != 2 * ln.period || ln.film)) return false;
  return aligned16l(ln.w) && aligned16l(ln.b) && aligned16l(ln.w2) && aligned16l(ln.b2) && aligned16l(ln.film);
}
#endif

// d3pm_tuning.lat_tile: 0 auto, 1 / 2 / 3 = always 64 x 64 / 96 x 64 / 32 x 64.  Auto: the geometry with the fewest rounds over the
// 256 CUs (one workgroup per CU), then the one with the most workgroups (the time of a launch here is the latency of one
// workgroup's chain -- DMA flight, k-steps, epilogue -- so a round less or a shorter chain is what pays; flops do not matter)
static int lat_geometry(const LinearArgs& a) {
  const int g_lat_tile = tune_of(a.tune).lat_tile;
  if (g_lat_tile >= 1 && g_lat_tile <= 3) return g_lat_tile - 1;
  static const int tm[3] = {64, 96, 32};
  int best = 0;
  long long best_rounds = 0, best_tiles = 0;
  for (int g = 0; g < 3; ++g) {
    const long long tiles = static_cast<long long>((a.M + tm[g] - 1) / tm[g]) * ((a.N + 63) / 64), rounds = (tiles + 255) / 256;
    if (g == 0 || rounds < best_rounds || (rounds == best_rounds && tiles > best_tiles)) { best = g; best_rounds = rounds; best_tiles = tiles; }
  }
  return best;
}

template <typename U, int E, bool LN, int TM, int TN>
static int panel64_launch(const LinearArgs& a, const LnPrologue* lnp, const uint16_t* tab, hipStream_t s) {
  using G = LatGeom<TM, TN>;
  constexpr size_t kMaxLds = 2 * G::BUF + GELU_TAB_BYTES <= 160 * 1024 ? 2 * G::BUF + GELU_TAB_BYTES : 2 * G::BUF;
  if (2 * G::BUF + GELU_TAB_BYTES > 160 * 1024) tab = nullptr;            // no room for the table beside this geometry's panels
  const size_t lds = 2 * G::BUF + (tab ? GELU_TAB_BYTES : 0);
  const int n_tiles = (a.N + TN - 1) / TN, m_tiles = (a.M + TM - 1) / TM;
  D3PM_LDS_ATTR((&gemm_mfma_panel64<U, E, LN, TM, TN>), kMaxLds);
  LnProArgs<U> la{};
  if (lnp) la = LnProArgs<U>{static_cast<const U*>(lnp->w), static_cast<const U*>(lnp->b), static_cast<const U*>(lnp->w2),
                             static_cast<const U*>(lnp->b2), static_cast<const U*>(lnp->film), lnp->eps, lnp->period};
  gemm_mfma_panel64<U, E, LN, TM, TN><<<dim3(static_cast<unsigned>(n_tiles * m_tiles)), dim3(256), lds, s>>>(
      static_cast<const U*>(a.X), a.ldx, static_cast<const U*>(a.W), static_cast<const U*>(a.bias), static_cast<U*>(a.Y), a.ldy,
      static_cast<const U*>(a.R1), static_cast<const U*>(a.R2), a.ldr, a.row_mask, a.mask_period, a.M, a.N, a.K, n_tiles, tab, la,
      epi_fold_of(a));
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

template <typename U, int E> static int panel64_geometry(const LinearArgs& a, const uint16_t* tab, hipStream_t s) {
  switch (lat_geometry(a)) {
    case 1: return panel64_launch<U, E, false, 96, 64>(a, nullptr, tab, s);
    case 2: return panel64_launch<U, E, false, 32, 64>(a, nullptr, tab, s);
    default: return panel64_launch<U, E, false, 64, 64>(a, nullptr, tab, s);
  }
}

// x' = rn(rn(R1 + rn(X W^T + b)) + rn(X2 W^T + b)) in one launch (the two cross-attention out-projections of a DiT block at one or
// two utterances): whole 32 x 64 tiles, K = 512 (both operand panels and the weight panel resident: 128 KiB)
bool panel64_dual_supported(int dtype, const LinearArgs& a, const void* X2) {
  if (!panel64_linear_supported(dtype, a) || !X2 || !aligned16l(X2)) return false;
  if (a.K != 2 * KC || a.M % 32 != 0 || a.N % 64 != 0 || a.M > 2048) return false;
  if (a.fold_s || a.M % 32 != 0) return false;
  return a.R1 && !a.R2 && !a.row_mask && a.act == ACT_NONE && a.bias;
}

int panel64_dual(int dtype, const LinearArgs& a, const void* X2, hipStream_t s) {
  using G = LatGeom<32, 64>;
  constexpr size_t lds = 2 * G::BUF + 2 * G::XOPER;
  const int n_tiles = a.N / 64, m_tiles = a.M / 32;
  auto go = [&](auto* tag) -> int {
    using U = std::remove_pointer_t<decltype(tag)>;
    if (a.stats_out) {
      D3PM_LDS_ATTR((&gemm_mfma_panel64<U, EPI_R1 | EPI_STATS, false, 32, 64, true>), lds);
      gemm_mfma_panel64<U, EPI_R1 | EPI_STATS, false, 32, 64, true><<<dim3(static_cast<unsigned>(n_tiles * m_tiles)), dim3(256), lds, s>>>(
          static_cast<const U*>(a.X), a.ldx, static_cast<const U*>(a.W), static_cast<const U*>(a.bias), static_cast<U*>(a.Y), a.ldy,
          static_cast<const U*>(a.R1), nullptr, a.ldr, nullptr, 1, a.M, a.N, a.K, n_tiles, nullptr, LnProArgs<U>{}, epi_fold_of(a),
          static_cast<const U*>(X2));
    } else {
      D3PM_LDS_ATTR((&gemm_mfma_panel64<U, EPI_R1, false, 32, 64, true>), lds);
      gemm_mfma_panel64<U, EPI_R1, false, 32, 64, true><<<dim3(static_cast<unsigned>(n_tiles * m_tiles)), dim3(256), lds, s>>>(
          static_cast<const U*>(a.X), a.ldx, static_cast<const U*>(a.W), static_cast<const U*>(a.bias), static_cast<U*>(a.Y), a.ldy,
          static_cast<const U*>(a.R1), nullptr, a.ldr, nullptr, 1, a.M, a.N, a.K, n_tiles, nullptr, LnProArgs<U>{}, EpiFold{},
          static_cast<const U*>(X2));
    }
    D3PM_LAUNCH_CHECK();
    return D3PM_OK;
  };
  return dtype == D3PM_F16 ? go(static_cast<f16*>(nullptr)) : go(static_cast<bf16*>(nullptr));
}

int panel64_linear(int dtype, const LinearArgs& a, hipStream_t s, const LnPrologue* lnp) {
#ifdef D3PM_ABLATIONS
  const uint16_t* tab = (a.act == ACT_GELU && dtype == D3PM_BF16 && gelu_table_enabled()) ? gelu_table_device(s) : nullptr;
#else
  const uint16_t* tab = nullptr;
#endif
  const int epi = (a.act == ACT_GELU ? EPI_GELU : 0) | (a.R1 ? (a.R2 ? EPI_R2 : EPI_R1) : 0) | (a.row_mask ? EPI_MASK : 0) |
                  (a.fold_s ? EPI_LNF : 0) | (a.stats_out ? EPI_STATS : 0);
  auto go = [&](auto* tag) -> int {
    using U = std::remove_pointer_t<decltype(tag)>;
#ifdef D3PM_ABLATIONS
    if (lnp) {
      switch (epi) {
        case 0: return panel64_launch<U, 0, true, 64, 64>(a, lnp, tab, s);
        case EPI_GELU: return panel64_launch<U, EPI_GELU, true, 64, 64>(a, lnp, tab, s);
        default: return D3PM_E_SHAPE;
      }
    }
#else
    if (lnp) return D3PM_E_SHAPE;
#endif
    switch (epi) {
      case 0: return panel64_geometry<U, 0>(a, tab, s);
      case EPI_GELU: return panel64_geometry<U, EPI_GELU>(a, tab, s);
      case EPI_R1: return panel64_geometry<U, EPI_R1>(a, tab, s);
      case EPI_R2: return panel64_geometry<U, EPI_R2>(a, tab, s);
      case EPI_R1 | EPI_MASK: return panel64_geometry<U, EPI_R1 | EPI_MASK>(a, tab, s);
      case EPI_LNF: return panel64_geometry<U, EPI_LNF>(a, tab, s);
      case EPI_LNF | EPI_GELU: return panel64_geometry<U, EPI_LNF | EPI_GELU>(a, tab, s);
      case EPI_R1 | EPI_STATS: return panel64_geometry<U, EPI_R1 | EPI_STATS>(a, tab, s);
      case EPI_R2 | EPI_STATS: return panel64_geometry<U, EPI_R2 | EPI_STATS>(a, tab, s);
      case EPI_R1 | EPI_MASK | EPI_STATS: return panel64_geometry<U, EPI_R1 | EPI_MASK | EPI_STATS>(a, tab, s);
      default: break;
    }
    return D3PM_E_SHAPE;
  };
  return dtype == D3PM_F16 ? go(static_cast<f16*>(nullptr)) : go(static_cast<bf16*>(nullptr));
}

}  // namespace d3pm
