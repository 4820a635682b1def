// d3pm_mfma_gemm_big.hip -- big-tile persistent MFMA GEMM for the DiT projections at throughput batch sizes.
//
//   Y[M][N] = epilogue(X[M][K] . W[N][K]^T + bias)      same contract and epilogue as d3pm_mfma_gemm.hip
//
// replaces the nn.Linear / MultiheadAttention projections of DiTBlock.forward
// (/root/reference/vall_e/vall_e/ar_discrete.py:132,138,142,159) when the batch is large enough to fill the chip.
//
// Why a second structure.  With K = d_model = 512 the 128 x 128 kernels of d3pm_mfma_gemm.hip move one byte from L2
// into LDS per 64 flop; the measured L2 -> LDS rate of a CU (~32 B/clk) then caps them at half the MFMA rate
// (profiles/round1_*: 577 .. 853 TFLOP/s by shape, 0.30 of peak over the loop).  The lever is bytes per flop:
//   * a wave owns a 96 x 64 sub-tile (6 x 4 MFMA tiles of 16 x 16, 96 accumulator registers); a workgroup is 2 x 2 of them (192 x 128,
//     77 flop per staged byte, TWO workgroups per CU: the default since round 3 -- in the sampler's loop two independent
//     workgroups hide cold operands and each other's epilogue better than one, big_linear_tile below) or eight (192 x 256 / 96 x 512,
//     110 / 81 flop per staged byte, one workgroup per CU);
//   * the tile shapes divide the bench workload exactly: M = 32 utterances x 768 canvas rows = 128 x 192 rows, N = 512 / 1024 /
//     1536 / 2048 = 4 / 8 / 12 / 16 x 128 columns, i.e. 512 / 1024 / 1536 / 2048 tiles = 1 / 2 / 3 / 4 whole rounds over the 512
//     resident workgroups -- the 128 x 128 / 256 x 256 grids left a quarter of the chip idle in the last round (DESIGN_HISTORY.md
//     section 3);
//   * two LDS stages of (TM + TN) x 128 B; the next k-step's 1-KiB DMA pieces (global_load_lds_dwordx4 from inline
//     asm, invisible to hipcc's wait counters) are issued one per four MFMAs inside the current k-step, so the
//     memory pipeline's issue back-pressure hides under the partner wave's MFMAs; one s_barrier per k-step with a
//     counted vmcnt; the stream of k-steps runs across tile boundaries (the next tile's first k-step is in flight
//     under the epilogue) and the epilogue's 16-byte stores stay in flight into the next tile;
//   * same swizzled 128-byte-row LDS image, same D = W_frag . X_frag^T orientation and the same epilogue code as the
//     128 x 128 kernels (d3pm_mfma_tile.h): the accumulation order over k is identical, results are bit-identical.
#include "d3pm_kernels.h"
#include "d3pm_mfma_tile.h"
#include "d3pm_mx.h"

namespace d3pm {
namespace {

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{})
template <class F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// one 16-byte fragment read whose completion is waited for by hand (lds_wait): hipcc would otherwise sink each read
// down to the instruction before its first use (it minimises registers), which exposes the LDS latency in every group
__device__ __forceinline__ void lds_read16(uintx4& dst, uint32_t addr, int off) {   // off: a constant after inlining
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off));
}
// wait until at most N of this wave's LDS reads are outstanding; the "+v" operands are the fragments this makes valid,
// so that no MFMA that consumes them can be scheduled above the wait
template <int N> __device__ __forceinline__ void lds_wait(uintx4& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }
template <int N> __device__ __forceinline__ void lds_wait(uintx4& a, uintx4& b, uintx4& c, uintx4& d, uintx4& e) {
  asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e) : "n"(N));
}
// DMA piece without the M0 save / restore of glds16_asm_s.  M0 is a reserved register that hipcc only writes right in
// front of its own M0-reading instructions (LDS-DMA builtins, s_movrel, GWS), and gemm_mfma_big contains none of those
// (checked in the .s: no m0 outside these statements), so the value left behind is never observed.
__device__ __forceinline__ void glds16_m0(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

// MODE: 1 is the shipped schedule; every other value is instantiated in the A/B library only (-DD3PM_ABLATIONS, include/d3pm_hip_ab.h).
// bit 0 = hand-scheduled fragment reads (asm, counted lgkmcnt); bit 3 (8) = deferred stores: a tile's 12 output stores
// per wave are issued a few per k-step inside the NEXT tile's main loop (all CUs finish their tiles together, so stores
// issued in the epilogue arrive as one chip-wide burst that the HBM write path drains at ~5.5 TB/s while every MFMA pipe
// idles: 4.5 us per 25 MB round; needs K >= 512); the other bits are timing-only ablations (WRONG
// results): 16 = no DMA pieces in the k-loop, 32 = no fragment reads / MFMAs, 64 = no waits / barriers, 128 = the MFMAs
// take their operands from registers (no LDS reads)
__device__ unsigned long long g_big_stamp[4];   // MODE bit 8: {shader clocks, 100 MHz ticks} of block 0 (clock under load)

// Row-panel fusion (FUSE != 0; 96 x 512 tiles, d_model = 512 only: a workgroup then owns whole rows of the residual stream):
//   bit 0  a SECOND product with the same weights runs through the same tile before the epilogue -- the text and prompt
//          cross-attention outputs both go through cross_attn.out_proj (ar_discrete.py:138,142): h = rn(X W^T + b) stays
//          packed in registers, then x' = rn(rn(R1 + h) + rn(X2 W^T + b)): one launch, no h round trip through HBM;
//   bit 1  LayerNorm of the finished rows (the NEXT op of the block, ar_discrete.py:131,136,153) in the epilogue: the row
//          moments are reduced in registers, across lanes and across the eight waves (LDS) in the exact order of
//          layernorm_vec (wave_sum_up), so the output is bit-identical to the stand-alone LayerNorm launch it replaces;
//   bit 2  a second LayerNorm of the same rows (norm2 | norm22);   bit 3  FiLM on the first (norm3, :145-156);
//   bit 5  (fp8 fast path) the LayerNorm outputs leave in the block-scaled fp8 format of d3pm_mx.hip instead of 16 bit: lny / lny2
//          are then code buffers [M][512] bytes and sx / sx2 their block scales [M][4][4]; the value that is quantised is the 16-bit
//          LayerNorm result bit for bit (the four lanes that own 32 consecutive columns of a row own one MX block).
template <typename T> struct RowPanelArgs {
  const T* X2; const T* lnw; const T* lnb; const T* lnw2; const T* lnb2; const T* film; T* lny; T* lny2; float eps;
  uint8_t* sx; uint8_t* sx2;
};
constexpr int FUSE_DUAL = 1, FUSE_LN = 2, FUSE_LN2 = 4, FUSE_FILM = 8, FUSE_ABL_NOLN = 16, FUSE_MX = 32;   // 16: timing-only (LayerNorm arithmetic skipped)

template <typename T, int EPI, int WM, int WN, int MODE, int FUSE = 0>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm_mfma_big(const T* __restrict__ X, int ldx, const T* __restrict__ W,
                                                        const T* __restrict__ bias, T* Y, int ldy, const T* R1,
                                                        const T* R2, int ldr, const uint8_t* __restrict__ row_mask,
                                                        int mask_period, int M, int N, int K, int n_tiles,
                                                        int tiles_total, const uint16_t* __restrict__ gelu_tab_g,
                                                        RowPanelArgs<T> rp, EpiFold ef) {
  constexpr int NW = WM * WN, TM = 96 * WM, TN = 64 * WN;   // 8 waves: one workgroup per CU; 4 waves: two
  constexpr int XD = TM / 8, WD = TN / 8;              // 1-KiB DMA pieces (8 rows x 128 B) per k-step and operand
  constexpr int XPW = (XD + NW - 1) / NW, WPW = WD / NW;   // pieces per wave
  constexpr int NDMA = XPW + WPW;                      // 7 (192 x 256), 10 (96 x 512: waves 4..7 repeat an X piece), 10 (192 x 128)
  static_assert(NW % 2 == 0 && WD % NW == 0 && (XD % NW == 0 || (XD % NW) % 2 == 0), "piece distribution keeps the parity of the wave");
  constexpr int X_BYTES = TM * ROW_BYTES, STAGE = (TM + TN) * ROW_BYTES;
  static_assert(NDMA <= 12, "one DMA piece per group of four MFMAs");
  constexpr bool kHand = (MODE & 1) != 0;
  constexpr int ABL = (MODE >> 4) & 3;
  constexpr bool kNoSync = (MODE & 64) != 0, kNoReads = (MODE & 128) != 0, kStamp = (MODE & 256) != 0;
  constexpr bool kPrioYoung = (MODE & 2) != 0, kPrioMfma = (MODE & 4) != 0, kDrip = (MODE & 8) != 0;
  // MODE bit 15: the epilogue's operands (bias, residual rows, frame mask) of a tile are requested at the top of the tile instead
  // of behind its last MFMA -- one exposed round trip less per tile, and for one-round launches (fc2: one tile per CU) the 25 MB
  // residual read runs under the k-loop instead of after it.  NPRE = the vector-memory operations this certainly adds behind
  // the previous tile's stores (a LOWER bound keeps the counted waits safe: they may only wait for more): 12 residual + 6 mask loads
  constexpr bool kPreEpi = (MODE & 32768) != 0 && FUSE == 0 && (EPI & EPI_R2) == 0;
  // EPI_LNF (K = 512: the launcher): the row moments a lane needs for its six row blocks are 12 x 16 bytes per tile -- 48 KB per
  // workgroup tile beside its 320 KB of operands, through the same CU <-> L2 path -- and a round trip the epilogue would start
  // with.  Loaded where the epilogue needs them they cost qkv 36.7 -> 45.1 us and fc1 69.7 -> 84.8 us in the loop; requested at
  // the top of the tile they stalled the second k-step instead (every k-step waits with vmcnt(0), vector-memory operations
  // retire in order).  The one place such a load can hide is ACROSS an epilogue: the moments of the NEXT row panel are requested
  // at the start of the epilogue of the last tile of this one, from inline asm (exactly 12 loads) behind the tile's own
  // per-column vectors (8 asm loads), the epilogue waits with vmcnt(24) -- its vectors have landed, the moments may stay in
  // flight -- and the next tile's first k-step, which waits for everything older than the 12 output stores anyway, finds them
  // landed; they are reduced to the two scalars per row right behind that k-step.
  constexpr bool kLnfPre = (EPI & EPI_LNF) != 0;
  constexpr int NPRE = !kPreEpi ? 0 : ((EPI & EPI_R1) ? 12 : 0) + ((EPI & EPI_MASK) ? 6 : 0);
  static_assert((MODE & 32768) == 0 || (!kDrip && (MODE & 1)), "epilogue prefetch rides in the hand-placed schedule");
  constexpr bool kDual = (FUSE & FUSE_DUAL) != 0, kLn = (FUSE & FUSE_LN) != 0, kLn2 = (FUSE & FUSE_LN2) != 0, kFilm = (FUSE & FUSE_FILM) != 0;
  constexpr bool kMx = (FUSE & FUSE_MX) != 0;
  static_assert(!kMx || kLn, "the MX output is the LayerNorm output");
  static_assert(FUSE == 0 || !kDrip, "fusions ride in the hand-placed schedule without deferred stores");
  static_assert(FUSE == 0 || FUSE == FUSE_DUAL || (WM == 1 && WN == 8 && (MODE & 1)), "row-panel fusion: 96 x 512 tiles, hand-placed schedule");
  static_assert((EPI & (EPI_LNF | EPI_STATS)) == 0 || ((FUSE == 0 || (FUSE == FUSE_DUAL && (EPI & EPI_LNF) == 0)) && !kDrip && !kPreEpi),
                "the folded-LayerNorm epilogues ride in the plain launches (and the row moments in the dual out-projection)");
  static_assert(!kLn2 || kLn, "the second LayerNorm shares the moments of the first");
  static_assert(!kDrip || kHand, "deferred stores ride in the hand-placed schedule");
  constexpr int SPS = 12 - NDMA;                        // deferred stores per k-step: the MFMA groups behind the last DMA piece
  constexpr int DRIP_STEPS = kDrip ? (12 + SPS - 1) / SPS : 0;          // 3 (192 x 256) or 6
  constexpr int PEEL = kDrip ? ((DRIP_STEPS + 1 + 1) & ~1) : 2;           // k-steps written out per tile (even)
  unsigned long long stamp_c = 0, stamp_r = 0;
  if constexpr (kStamp) { stamp_c = __builtin_amdgcn_s_memtime(); stamp_r = __builtin_amdgcn_s_memrealtime(); }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  // XCD x = blockIdx & 7 owns a contiguous range of tiles (the n-tiles of one X panel then share an L2)
  const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
  const int tq = tiles_total >> 3, tr = tiles_total & 7;
  const int lo = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq, cnt = tq + (xcd < tr ? 1 : 0);
  int t = blockIdx.x >> 3;
  // Tile walk: workgroup slot s of an XCD takes tiles s, s + per_xcd, .. of the XCD's range, so the n-tiles of one row panel run at
  // the same time on neighbouring CUs and share the panel's X rows in L2 (fetched from the fabric once).  A walk over CONSECUTIVE
  // tiles (one workgroup = 3-4 n-tiles of one panel, its row scalars fetched once) was measured for the folded launches: +1.2 %
  // while the output rows still went through L2, nothing once they left as `sc1` stores (106.8 k vs 106.7 k tokens/s) -- and it
  // re-fetched each panel 2-3 times from the fabric (PMC 93.6 MB per launch against 49 / 30 algorithmic), so it is not kept.
  const int t_end = cnt, t_step = per_xcd;
  if (t >= t_end) return;                                            // block-uniform
  if constexpr ((MODE & (8192 | 16384)) != 0) {
    // A/B: de-synchronise the chip-wide output burst -- the workgroups of every second XCD start later by a fraction of a tile
    // (bits 13 / 14 / both: ~2 / 4 / 6 us; one s_sleep 127 measured ~3.1 us), so that one group's stores drain while the other
    // group streams operands
    if (xcd & 1) {
      constexpr int units = ((MODE >> 13) & 3) * 2;
      for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(40);          // ~1 us each
    }
  }
  const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem));
  // DMA piece j of an operand tile = rows 8j .. 8j+7; a wave takes pieces j = wave + NW p, so the swizzle key
  // (row >> 1) & 7 = (4 (j & 1) + (lane >> 4)) & 7 is the same for all of its pieces: one per-lane offset per operand
  const int lrow = lane >> 3, logical = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
  const uint32_t ox = static_cast<uint32_t>(lrow * ldx + logical * 8) * 2u;
  const uint32_t ow = static_cast<uint32_t>(lrow * K + logical * 8) * 2u;
  auto dma = [&](int p, const T* px, const T* pw, uint32_t stage) __attribute__((always_inline)) {   // p: unrolled constant
    if (p < XPW) {
      int j = wave + NW * p;
      if (XD % NW != 0 && j >= XD) j = (XD / NW) * NW + wave % (XD % NW);   // same parity as `wave`: a harmless repeat
      if (kHand) glds16_m0(px + static_cast<size_t>(8 * j) * ldx, ox, stage + j * 1024);
      else glds16_asm_s(px + static_cast<size_t>(8 * j) * ldx, ox, stage + j * 1024);
    } else {
      const int j = wave + NW * (p - XPW);
      if (kHand) glds16_m0(pw + static_cast<size_t>(8 * j) * K, ow, stage + X_BYTES + j * 1024);
      else glds16_asm_s(pw + static_cast<size_t>(8 * j) * K, ow, stage + X_BYTES + j * 1024);
    }
  };
  // fragment addresses: row = base + 16 q + (lane & 15); the swizzle key (row >> 1) & 7 = (lane & 15) >> 1 because
  // every base is a multiple of 16
  const int frow = lane & 15, fch = lane >> 4, fkey = (frow >> 1) & 7;
  const int fo0 = frow * ROW_BYTES + ((fch ^ fkey) << 4), fo1 = frow * ROW_BYTES + (((4 + fch) ^ fkey) << 4);
  const char* const fx_base = smem + wm * 96 * ROW_BYTES;
  const char* const fw_base = smem + X_BYTES + wn * 64 * ROW_BYTES;

  // bf16 GELU epilogue: the lookup table rides behind the two stages (only where it fits: 192 x 256 tiles)
  const uint16_t* gelu_tab = nullptr;
  if constexpr ((EPI & EPI_GELU) != 0 && std::is_same<T, bf16>::value && 2 * STAGE + GELU_TAB_BYTES <= 160 * 1024 && NW == 8) {
    if (gelu_tab_g) {
      gelu_table_to_lds(gelu_tab_g, smem + 2 * STAGE, tid, NW * 64);
      gelu_tab = reinterpret_cast<const uint16_t*>(smem + 2 * STAGE);     // the first k-step's barrier publishes it
    }
  }
  if constexpr (kPrioYoung) {      // the later-dispatched half of an 8-wave workgroup loses issue arbitration on its SIMD
    if (wave >= NW / 2) __builtin_amdgcn_s_setprio(1);
  }
  const int nk = K / BK;                                             // even (checked by the launcher)
  int tile = lo + t;
  const T* sx = X + static_cast<size_t>((tile / n_tiles) * TM) * ldx;
  const T* sw = W + static_cast<size_t>((tile % n_tiles) * TN) * K;
#pragma unroll
  for (int p = 0; p < NDMA; ++p) dma(p, sx, sw, lds_base);          // first k-step of the first tile
  typedef float float2v __attribute__((ext_vector_type(2)));
  [[maybe_unused]] float2v lst[6][4];                                // EPI_LNF: the moments of the tile about to start, in flight
  auto moments_request = [&](int mrow0) __attribute__((always_inline)) {
    // parts 4 g .. 4 g + 3 of row (lane & 15) of each of the six 16-row blocks: [row block][part][row][2] (stats_index), 24 loads
    const float* sp = ef.stats_in + stats_index(static_cast<size_t>(mrow0 + wm * 96 + (lane & 15)), (lane >> 4) * 4, 16);
#pragma unroll
    for (int mt = 0; mt < 6; ++mt) {
      asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(lst[mt][0]) : "v"(sp + mt * 512) : "memory");
      asm volatile("global_load_dwordx2 %0, %1, off offset:128" : "=v"(lst[mt][1]) : "v"(sp + mt * 512) : "memory");
      asm volatile("global_load_dwordx2 %0, %1, off offset:256" : "=v"(lst[mt][2]) : "v"(sp + mt * 512) : "memory");
      asm volatile("global_load_dwordx2 %0, %1, off offset:384" : "=v"(lst[mt][3]) : "v"(sp + mt * 512) : "memory");
    }
  };
  if constexpr (kLnfPre) moments_request((tile / n_tiles) * TM);     // the first tile's: covered by the wait below
  [[maybe_unused]] RowScalars<6> rows;
  [[maybe_unused]] bool fresh = true;                                // lst holds moments that have not been reduced yet
  // the first step of a tile waits with vmcnt(12): behind an epilogue the DMA pieces are older than its 12 stores; the very
  // first tile has no stores behind its pieces, so they are waited for here
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  uintx4 pend[12];                // deferred-store mode: the previous tile, finished and packed
#pragma unroll
  for (int j = 0; j < 12; ++j) pend[j] = uintx4{0u, 0u, 0u, 0u};
  T* pend_y = Y;
  bool pending = false;
  for (;;) {
    floatx4 acc[4][6];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 6; ++b) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int m0 = (tile / n_tiles) * TM, n0 = (tile % n_tiles) * TN;
    [[maybe_unused]] EpiPre<T, 4, 6> pre;
    if constexpr (kPreEpi) epilogue_prefetch<T, EPI, 4, 6>(pre, bias, R1, R2, ldr, row_mask, mask_period, M, N, m0 + wm * 96, n0 + wn * 64, lane);
    // what follows this tile (block-uniform); the last tile re-reads its own first k-step: valid memory, never used
    const int t_next = t + t_step;
    const bool more = t_next < t_end;
    const int tile_next = more ? lo + t_next : tile;
    const T* sx_next = X + static_cast<size_t>((tile_next / n_tiles) * TM) * ldx;
    const T* sw_next = W + static_cast<size_t>((tile_next % n_tiles) * TN) * K;

    // one k-step on stage S while the DMA pieces of the following k-step (px, pw) go to stage S ^ 1.  WAITN: what the wait at
    // the top may leave in flight (the stores issued behind the previous k-step's last DMA piece); [ST0, ST0 + STN): the
    // deferred stores of the previous tile that this k-step issues
    auto step = [&](auto S_, auto WAITN_, auto ST0_, auto STN_, const T* px, const T* pw) __attribute__((always_inline)) {
      constexpr int S = decltype(S_)::value, WAITN = decltype(WAITN_)::value, ST0 = decltype(ST0_)::value, STN = decltype(STN_)::value;
      const char* bx = fx_base + S * STAGE;
      const char* bw = fw_base + S * STAGE;
      const uint32_t nxt = lds_base + (S ^ 1) * STAGE;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (!kNoSync) {
        if constexpr (WAITN > 0) {
          if (!kDrip || pending) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();      // every wave's pieces of this k-step have landed; stage S ^ 1 is no longer read
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (ABL == 2 && (MODE & 4096) != 0) {
        // timing probe: the same pieces through registers (global_load_dwordx4 -> ds_write_b128) instead of LDS-DMA
        uintx4 stg[NDMA];
#pragma unroll
        for (int p = 0; p < NDMA; ++p) {
          if (p < XPW) {
            int j = wave + NW * p;
            if constexpr (XD % NW != 0) { if (j >= XD) j = (XD / NW) * NW + wave % (XD % NW); }
            stg[p] = *reinterpret_cast<const uintx4*>(reinterpret_cast<const char*>(px + static_cast<size_t>(8 * j) * ldx) + ox);
          } else {
            const int j = wave + NW * (p - XPW);
            stg[p] = *reinterpret_cast<const uintx4*>(reinterpret_cast<const char*>(pw + static_cast<size_t>(8 * j) * K) + ow);
          }
        }
#pragma unroll
        for (int p = 0; p < NDMA; ++p) {
          int j = p < XPW ? wave + NW * p : wave + NW * (p - XPW);
          if constexpr (XD % NW != 0) { if (p < XPW && j >= XD) j = (XD / NW) * NW + wave % (XD % NW); }
          *reinterpret_cast<uintx4*>(smem + (S ^ 1) * STAGE + (p < XPW ? 0 : X_BYTES) + j * 1024 + lane * 16) = stg[p];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      } else if constexpr (ABL == 2) {
#pragma unroll
        for (int g = 0; g < NDMA; ++g) dma(g, px, pw, nxt);
      } else if constexpr (kHand) {
        // issue order of the 20 fragment reads of a k-step: W0..3 X0 X1 | g0: X2 | g1: X3 | g2: X4 W'0 | g3: X5 W'1 | g4: X6 W'2 |
        // g5: X7 W'3 | g6: X8 | g7: X9 | g8: X10 | g9: X11 (X six row blocks per k-half, W' = the second k-half's W fragments);
        // group g consumes X_g (and W at g = 0, W' at g = 6): the counts below are the reads younger than what it needs
        const uint32_t ax0 = lds_base + S * STAGE + wm * 96 * ROW_BYTES + fo0, ax1 = ax0 - fo0 + fo1;
        const uint32_t aw0 = lds_base + S * STAGE + X_BYTES + wn * 64 * ROW_BYTES + fo0, aw1 = aw0 - fo0 + fo1;
        uintx4 fw0[4], fw1[4], fx[3];
        if constexpr (kNoReads) {
          const uintx4 junk = {0x3f803f80u + lane * 0x10001u, 0x3f003e80u ^ (lane << 7), 0xbf80bf00u + S, 0x3e803f00u ^ lane};
          for (int q = 0; q < 4; ++q) { fw0[q] = junk + q; fw1[q] = junk * (q + 3); }
          fx[0] = junk + 7; fx[1] = junk * 5; fx[2] = junk + 11;
          for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(fw0[q]), "+v"(fw1[q]));
          asm volatile("" : "+v"(fx[0]), "+v"(fx[1]), "+v"(fx[2]));
        } else {
          static_for<4>([&](auto NT) { lds_read16(fw0[NT.value], aw0, NT.value * 16 * ROW_BYTES); });
          lds_read16(fx[0], ax0, 0);
          lds_read16(fx[1], ax0, 16 * ROW_BYTES);
        }
        static_for<12>([&](auto G) {
          constexpr int g = G.value, ks = g / 6, mt = g % 6;
          if constexpr (!kNoReads) {
            if constexpr (g + 2 < 12) lds_read16(fx[(g + 2) % 3], (g + 2) / 6 ? ax1 : ax0, ((g + 2) % 6) * 16 * ROW_BYTES);
            if constexpr (ks == 0 && mt >= 2) lds_read16(fw1[mt - 2], aw1, (mt - 2) * 16 * ROW_BYTES);
          }
          if constexpr ((MODE & 2048) != 0) {                // A/B: every piece of the next k-step issued in front of the first MFMA group
            if constexpr (g == 0 && ABL != 1) {
#pragma unroll
              for (int p = 0; p < NDMA; ++p) dma(p, px, pw, nxt);
            }
          } else if constexpr (g < NDMA && ABL != 1) dma(g, px, pw, nxt);
          constexpr int kWait[12] = {2, 2, 3, 4, 5, 5, 1, 2, 2, 2, 1, 0};
          if constexpr (kNoReads) {}
          else if constexpr (g == 0) lds_wait<kWait[g]>(fw0[0], fw0[1], fw0[2], fw0[3], fx[0]);
          else if constexpr (g == 6) lds_wait<kWait[g]>(fw1[0], fw1[1], fw1[2], fw1[3], fx[g % 3]);
          else lds_wait<kWait[g]>(fx[g % 3]);
          if constexpr (kPrioMfma) __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc[nt][mt] = mma<T>(__builtin_bit_cast(uint4, ks ? fw1[nt] : fw0[nt]), __builtin_bit_cast(uint4, fx[g % 3]), acc[nt][mt]);
          if constexpr (kPrioMfma) __builtin_amdgcn_s_setprio(0);
          if constexpr (kDrip && g >= NDMA && g - NDMA < STN) {
            constexpr int j = ST0 + g - NDMA;          // store j = (row block j / 2, column pair j % 2) of the previous tile
            if (pending) *reinterpret_cast<uintx4*>(pend_y + static_cast<size_t>((j / 2) * 16) * ldy + (j % 2) * 32) = pend[j];
          }
        });
      } else {
      uint4 fw[2][4], fx[3];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) fw[0][nt] = *reinterpret_cast<const uint4*>(bw + nt * 16 * ROW_BYTES + fo0);
      fx[0] = *reinterpret_cast<const uint4*>(bx + fo0);
      fx[1] = *reinterpret_cast<const uint4*>(bx + 16 * ROW_BYTES + fo0);
#pragma unroll
      for (int g = 0; g < 12; ++g) {       // group g: k-half g / 6, row block g % 6, four MFMAs
        const int ks = g / 6, mt = g % 6;
        if (g + 2 < 12) {
          const int g2 = g + 2;
          fx[g2 % 3] = *reinterpret_cast<const uint4*>(bx + (g2 % 6) * 16 * ROW_BYTES + (g2 / 6 ? fo1 : fo0));
        }
        if (ks == 0 && mt >= 2) fw[1][mt - 2] = *reinterpret_cast<const uint4*>(bw + (mt - 2) * 16 * ROW_BYTES + fo1);
        if (g < NDMA && ABL != 1) dma(g, px, pw, nxt);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt][mt] = mma<T>(fw[ks][nt], fx[g % 3], acc[nt][mt]);
      }
      }
      __builtin_amdgcn_sched_barrier(0);
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    uintx4 h1p[kDual ? 12 : 1];
    // the nk k-steps of one product: operand rows `ax`, weights `sw`; (nx, nw) = the first k-step of whatever follows
    auto run_k = [&](const T* ax, const T* nx, const T* nw, auto FIRSTW) __attribute__((always_inline)) {
      step(I0{}, FIRSTW, I0{}, I0{}, ax + BK, sw + BK);
      if constexpr (kLnfPre) {
        // the first k-step waited for everything older than the previous tile's stores: this tile's moments, requested in front of
        // those stores, have landed (the "+v" operands keep every use behind that wait)
        if (fresh)
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) {
          asm volatile("" : "+v"(lst[mt][0]), "+v"(lst[mt][1]), "+v"(lst[mt][2]), "+v"(lst[mt][3]));
          float a = 0.f, q = 0.f;
#pragma unroll
          for (int i = 0; i < 4; ++i) { a += lst[mt][i][0]; q += lst[mt][i][1]; }
          a = add_xor16(a); q = add_xor16(q);
          fold_row_scalars(add_xor32(a), add_xor32(q), 512, ef.eps, rows.ra[mt], rows.rc[mt]);
        }
      }
      step(I1{}, I0{}, I0{}, I0{}, ax + 2 * BK, sw + 2 * BK);
      for (int kt = 2; kt < nk; kt += 2) {                  // nk is even and >= 4
        step(I0{}, I0{}, I0{}, I0{}, ax + (kt + 1) * BK, sw + (kt + 1) * BK);
        const bool last = kt + 2 >= nk;
        step(I1{}, I0{}, I0{}, I0{}, last ? nx : ax + (kt + 2) * BK, last ? nw : sw + (kt + 2) * BK);
      }
    };
    if constexpr (kDrip) {
      static_for<PEEL>([&](auto SI) {                     // the k-steps that carry the previous tile's stores, written out
        constexpr int si = SI.value;
        constexpr int cnt = 12 - si * SPS < 0 ? 0 : (12 - si * SPS < SPS ? 12 - si * SPS : SPS);
        constexpr int prev = si == 0 ? 0 : (12 - (si - 1) * SPS < 0 ? 0 : (12 - (si - 1) * SPS < SPS ? 12 - (si - 1) * SPS : SPS));
        const bool lastk = si + 1 >= nk;
        step(std::integral_constant<int, si & 1>{}, std::integral_constant<int, prev>{}, std::integral_constant<int, si * SPS>{},
             std::integral_constant<int, cnt>{}, lastk ? sx_next : sx + (si + 1) * BK, lastk ? sw_next : sw + (si + 1) * BK);
      });
      for (int kt = PEEL; kt < nk; kt += 2) {               // nk is even and >= PEEL
        step(I0{}, I0{}, I0{}, I0{}, sx + (kt + 1) * BK, sw + (kt + 1) * BK);
        const bool last = kt + 2 >= nk;
        step(I1{}, I0{}, I0{}, I0{}, last ? sx_next : sx + (kt + 2) * BK, last ? sw_next : sw + (kt + 2) * BK);
      }
    } else if constexpr (kDual) {
      // two products through the same weights, one loop so that the k-step code exists once: phase 0 ends with its result
      // packed in registers (h1p), phase 1 runs into the common epilogue below
      const T* sx2 = rp.X2 + static_cast<size_t>(m0) * ldx;
      const T* ax = sx;
#pragma unroll 1
      for (int ph = 0; ph < 2; ++ph) {
        const bool mid = ph == 0;
        run_k(ax, mid ? sx2 : sx_next, mid ? sw : sw_next, std::integral_constant<int, 12>{});
        if (mid) {
          epilogue_store<T, 0, 4, 6, true, true>(acc, bias, Y, ldy, nullptr, nullptr, ldr, nullptr, 1, M, N, m0 + wm * 96, n0 + wn * 64, lane, h1p);
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // nothing but the bias loads follows these DMA pieces: wait for all
          ax = sx2;
        }
      }
    } else {
      run_k(sx, sx_next, sw_next, std::integral_constant<int, 12 + NPRE>{});
    }
    if constexpr (FUSE != 0) {
      uintx4 xp[12];
      if constexpr (kDual) {
        // x' = rn(rn(R1 + h) + rn(X2 W^T + b)): the R1 + R2 epilogue of the two-launch form with R2 = h taken from registers
        epilogue_store<T, 0, 4, 6, true, true>(acc, bias, Y, ldy, nullptr, nullptr, ldr, nullptr, 1, M, N, m0 + wm * 96, n0 + wn * 64, lane, xp);
        const T* r1row = R1 + static_cast<size_t>(m0 + wm * 96 + (lane & 15)) * ldr + n0 + wn * 64 + epilogue_nq(lane);
        Pack8<T> r1p[12];
#pragma unroll
        for (int j = 0; j < 12; ++j) r1p[j] = *reinterpret_cast<const Pack8<T>*>(r1row + static_cast<size_t>((j / 2) * 16) * ldr + (j % 2) * 32);
#pragma unroll
        for (int j = 0; j < 12; ++j) {
          const Pack8<T> hh = __builtin_bit_cast(Pack8<T>, h1p[j]), yy = __builtin_bit_cast(Pack8<T>, xp[j]);
          Pack8<T> o;
#pragma unroll
          for (int i = 0; i < 8; ++i)
            o.v[i] = static_cast<T>(rn<T>(static_cast<float>(r1p[j].v[i]) + static_cast<float>(hh.v[i])) + static_cast<float>(yy.v[i]));
          xp[j] = __builtin_bit_cast(uintx4, o);
        }
      } else {
        epilogue_store<T, EPI, 4, 6, true, true>(acc, bias, Y, ldy, R1, R2, ldr, row_mask, mask_period, M, N, m0, n0 + wn * 64, lane, xp);
      }
      // ---- the finished rows: store x', then LayerNorm them in place (layernorm_vec's arithmetic, bit for bit)
      const int g = lane >> 4, nq = epilogue_nq(lane);
      T* yrow = Y + static_cast<size_t>(m0 + wm * 96 + (lane & 15)) * ldy + n0 + wn * 64 + nq;
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        T* yp = yrow + static_cast<size_t>((j / 2) * 16) * ldy + (j % 2) * 32;
        if constexpr (FUSE == FUSE_DUAL) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(yp), "v"(xp[j]) : "memory");   // as the plain epilogue (kNts)
        else *reinterpret_cast<uintx4*>(yp) = xp[j];
      }
      if constexpr ((EPI & EPI_STATS) != 0) {        // the row moments of the new rows for the folded LayerNorm that reads them next
#pragma unroll
        for (int j = 0; j < 12; ++j) {
          const Pack8<T> p = __builtin_bit_cast(Pack8<T>, xp[j]);
          float v[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = static_cast<float>(p.v[i]);
          part_stats_store(v, ef.stats_out, static_cast<size_t>(m0 + wm * 96 + (j / 2) * 16 + (lane & 15)), N, n0 + wn * 64 + (j % 2) * 32, g, true);
        }
      }
      if constexpr (kLn && (FUSE & FUSE_ABL_NOLN) != 0) {      // timing-only: the stores of the LayerNorm outputs without their arithmetic
#pragma unroll
        for (int np = 0; np < 2; ++np)
#pragma unroll
          for (int mt = 0; mt < 6; ++mt) {
            const size_t off = static_cast<size_t>(m0 + mt * 16 + (lane & 15)) * 512 + n0 + wn * 64 + np * 32 + nq;
            *reinterpret_cast<uintx4*>(rp.lny + off) = xp[mt * 2 + np];
            if constexpr (kLn2) *reinterpret_cast<uintx4*>(rp.lny2 + off) = xp[mt * 2 + np];
          }
      } else if constexpr (kLn) {
        float* red = reinterpret_cast<float*>(smem + 2 * STAGE);              // [2][96 rows][8 waves] partial sums
        auto unpack = [&](int j, float (&v)[8]) __attribute__((always_inline)) {
          const Pack8<T> p = __builtin_bit_cast(Pack8<T>, xp[j]);
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = static_cast<float>(p.v[i]);
        };
        // a row's 512 columns = 64 chunks of 8; this lane holds chunks c = 8 wave + 4 np + 2 (g & 1) + (g >> 1) of rows
        // (lane & 15) + 16 mt.  wave_sum_up's tree over the chunk index: bit 0 = lane ^ 32, bit 1 = lane ^ 16, bit 2 = np,
        // bits 3..5 = the wave (through LDS), each level adding the two halves exactly as the 64-lane butterfly does
        auto row_total = [&](float (&part)[6][2], int which, float (&tot)[6]) __attribute__((always_inline)) {
#pragma unroll
          for (int mt = 0; mt < 6; ++mt) {
            float a0 = part[mt][0], a1 = part[mt][1];
            a0 = add_xor32(a0); a1 = add_xor32(a1);         // v[l] + v[l ^ 32], then ^ 16 (d3pm_common.h: same bits as __shfl_xor)
            a0 = add_xor16(a0); a1 = add_xor16(a1);
            const float w8 = a0 + a1;
            if (g == 0) red[(which * 96 + mt * 16 + (lane & 15)) * 8 + wave] = w8;
          }
          __syncthreads();
#pragma unroll
          for (int mt = 0; mt < 6; ++mt) {
            const floatx4 lo4 = *reinterpret_cast<const floatx4*>(red + (which * 96 + mt * 16 + (lane & 15)) * 8);
            const floatx4 hi4 = *reinterpret_cast<const floatx4*>(red + (which * 96 + mt * 16 + (lane & 15)) * 8 + 4);
            tot[mt] = ((lo4[0] + lo4[1]) + (lo4[2] + lo4[3])) + ((hi4[0] + hi4[1]) + (hi4[2] + hi4[3]));
          }
        };
        float part[6][2], mean[6], rstd[6];
#pragma unroll
        for (int mt = 0; mt < 6; ++mt)
#pragma unroll
          for (int np = 0; np < 2; ++np) {
            float v[8], s1 = 0.f;
            unpack(mt * 2 + np, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) s1 += v[i];
            part[mt][np] = s1;
          }
        row_total(part, 0, mean);
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) mean[mt] = mean[mt] / 512.0f;
#pragma unroll
        for (int mt = 0; mt < 6; ++mt)
#pragma unroll
          for (int np = 0; np < 2; ++np) {
            float v[8], q = 0.f;
            unpack(mt * 2 + np, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float tt = v[i] - mean[mt]; q += tt * tt; }
            part[mt][np] = q;
          }
        row_total(part, 1, rstd);
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) rstd[mt] = rsqrtf(rstd[mt] / 512.0f + rp.eps);
        if constexpr (kMx) {
          // the LayerNorm result(s) as MX rows: per row block both 32-column halves are finished, quantised and stored together
          Pack8<T> wv[2], bv[2], sc[2], sh[2], w2v[2], b2v[2];
#pragma unroll
          for (int np = 0; np < 2; ++np) {
            const int col = n0 + wn * 64 + np * 32 + nq;
            wv[np] = *reinterpret_cast<const Pack8<T>*>(rp.lnw + col); bv[np] = *reinterpret_cast<const Pack8<T>*>(rp.lnb + col);
            if constexpr (kFilm) { sc[np] = *reinterpret_cast<const Pack8<T>*>(rp.film + col); sh[np] = *reinterpret_cast<const Pack8<T>*>(rp.film + 512 + col); }
            if constexpr (kLn2) { w2v[np] = *reinterpret_cast<const Pack8<T>*>(rp.lnw2 + col); b2v[np] = *reinterpret_cast<const Pack8<T>*>(rp.lnb2 + col); }
          }
          uint8_t* const y8 = reinterpret_cast<uint8_t*>(rp.lny);
          uint8_t* const y8b = reinterpret_cast<uint8_t*>(rp.lny2);
          const int col0 = n0 + wn * 64, blk = (col0 >> 5) + (lane >> 4);          // the block whose scale byte lanes g = 0 / 1 write
#pragma unroll
          for (int mt = 0; mt < 6; ++mt) {
            uint2 cd[2], cd2[2];
            uint32_t sb[2], sb2[2];
#pragma unroll
            for (int np = 0; np < 2; ++np) {
              float v[8], o[8], o2[8];
              unpack(mt * 2 + np, v);
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                const float nrm = (v[i] - mean[mt]) * rstd[mt];
                o[i] = rn<T>(nrm * static_cast<float>(wv[np].v[i]) + static_cast<float>(bv[np].v[i]));
                if constexpr (kFilm) {
                  const float gg = rn<T>(1.0f + static_cast<float>(sc[np].v[i]));
                  o[i] = rn<T>(rn<T>(o[i] * gg) + static_cast<float>(sh[np].v[i]));
                }
                if constexpr (kLn2) o2[i] = rn<T>(nrm * static_cast<float>(w2v[np].v[i]) + static_cast<float>(b2v[np].v[i]));
              }
              cd[np] = mx_block_quantise(o, sb[np]);
              if constexpr (kLn2) cd2[np] = mx_block_quantise(o2, sb2[np]);
            }
            const size_t row = static_cast<size_t>(m0 + mt * 16 + (lane & 15));
            mx_store_row64(cd[0], cd[1], y8 + row * 512 + col0, lane);
            if constexpr (kLn2) mx_store_row64(cd2[0], cd2[1], y8b + row * 512 + col0, lane);
            if (lane < 32) {
              rp.sx[row * 16 + (blk & 3) * 4 + (blk >> 2)] = static_cast<uint8_t>(lane < 16 ? sb[0] : sb[1]);
              if constexpr (kLn2) rp.sx2[row * 16 + (blk & 3) * 4 + (blk >> 2)] = static_cast<uint8_t>(lane < 16 ? sb2[0] : sb2[1]);
            }
          }
        } else {
#pragma unroll
        for (int np = 0; np < 2; ++np) {
          const int col = n0 + wn * 64 + np * 32 + nq;
          const Pack8<T> wv = *reinterpret_cast<const Pack8<T>*>(rp.lnw + col), bv = *reinterpret_cast<const Pack8<T>*>(rp.lnb + col);
          Pack8<T> sc, sh, w2v, b2v;
          if constexpr (kFilm) { sc = *reinterpret_cast<const Pack8<T>*>(rp.film + col); sh = *reinterpret_cast<const Pack8<T>*>(rp.film + 512 + col); }
          if constexpr (kLn2) { w2v = *reinterpret_cast<const Pack8<T>*>(rp.lnw2 + col); b2v = *reinterpret_cast<const Pack8<T>*>(rp.lnb2 + col); }
#pragma unroll
          for (int mt = 0; mt < 6; ++mt) {
            float v[8], nrm[8];
            unpack(mt * 2 + np, v);
            Pack8<T> o;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              nrm[i] = (v[i] - mean[mt]) * rstd[mt];
              o.v[i] = static_cast<T>(nrm[i] * static_cast<float>(wv.v[i]) + static_cast<float>(bv.v[i]));
            }
            if constexpr (kFilm) {
#pragma unroll
              for (int i = 0; i < 8; ++i) {
                const float gg = rn<T>(1.0f + static_cast<float>(sc.v[i]));
                o.v[i] = static_cast<T>(rn<T>(static_cast<float>(o.v[i]) * gg) + static_cast<float>(sh.v[i]));
              }
            }
            const size_t off = static_cast<size_t>(m0 + mt * 16 + (lane & 15)) * 512 + col;
            *reinterpret_cast<Pack8<T>*>(rp.lny + off) = o;
            if constexpr (kLn2) {
              Pack8<T> o2;
#pragma unroll
              for (int i = 0; i < 8; ++i) o2.v[i] = static_cast<T>(nrm[i] * static_cast<float>(w2v.v[i]) + static_cast<float>(b2v.v[i]));
              *reinterpret_cast<Pack8<T>*>(rp.lny2 + off) = o2;
            }
          }
        }
        }   // 16-bit LayerNorm outputs
      }
    } else {
      bool kept = false;
      if constexpr (kDrip) {         // keep the finished tile in registers: its stores go out inside the next tile's first k-steps
        if (more) {
          epilogue_store<T, EPI, 4, 6, true, true>(acc, bias, Y, ldy, R1, R2, ldr, row_mask, mask_period, M, N, m0 + wm * 96,
                                                   n0 + wn * 64, lane, pend, gelu_tab);
          pend_y = Y + static_cast<size_t>(m0 + wm * 96 + (lane & 15)) * ldy + n0 + wn * 64 + epilogue_nq(lane);
          pending = true;
          kept = true;
        }
      }
      if constexpr (kLnfPre) {
        floatx4 bq[4], sq[4];
        const int ncol = n0 + wn * 64 + (lane >> 4) * 4;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bq[nt]) : "v"(ef.b + ncol + nt * 16) : "memory");
          asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(sq[nt]) : "v"(ef.s + ncol + nt * 16) : "memory");
        }
        fresh = more && (tile_next / n_tiles) != (tile / n_tiles);      // the next tile reads other rows
        if (fresh) {
          moments_request((tile_next / n_tiles) * TM);
          asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
          asm volatile("" : "+v"(bq[nt]), "+v"(sq[nt]));
#pragma unroll
          for (int r = 0; r < 4; ++r) { pre.bv[nt][r] = bq[nt][r]; pre.sv[nt][r] = sq[nt][r]; }
        }
        epilogue_store<T, EPI, 4, 6, true, false, true, true>(acc, bias, Y, ldy, R1, R2, ldr, row_mask, mask_period, M, N, m0 + wm * 96,
                                                              n0 + wn * 64, lane, nullptr, gelu_tab, &pre, &ef, &rows);
      } else if (!kept)
        // output rows leave as `sc1` stores (kNts): they are not read again by this launch, and left in the XCD's L2 they displace the
        // operand panels the other tiles of the launch still stream (in the loop: 103.8 k -> 106.2 k tokens/s with the stand-alone
        // LayerNorms, 104.5 k -> 110.5 k with them folded; MODE bit 9 of the A/B library used to try `nt`, which keeps the line in L2)
        epilogue_store<T, EPI, 4, 6, true, false, true, kPreEpi>(acc, bias, Y, ldy, R1, R2, ldr, row_mask, mask_period, M, N, m0 + wm * 96,
                                                                 n0 + wn * 64, lane, nullptr, gelu_tab, &pre, &ef);
    }
    if (!more) break;
    t = t_next;
    tile = tile_next;
    sx = sx_next;
    sw = sw_next;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the unused look-ahead pieces must not outlive the workgroup's LDS
  if constexpr (kStamp) {
    if (blockIdx.x == 0 && tid == 0) {
      g_big_stamp[0] = __builtin_amdgcn_s_memtime() - stamp_c;
      g_big_stamp[1] = __builtin_amdgcn_s_memrealtime() - stamp_r;
    }
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) % 16) == 0; }

}  // namespace

// Tile geometries: id 1 = 96 x 512 (1 x 8 waves), 2 = 192 x 256 (2 x 4 waves), 3 = 192 x 128 (2 x 2 waves, two workgroups per CU)
static void big_geometry(int id, int& tm, int& tn, int& waves) {
  tm = id == 1 ? 96 : 192;
  tn = id == 1 ? 512 : id == 2 ? 256 : 128;
  waves = id == 3 ? 4 : 8;
}

#ifdef D3PM_ABLATIONS
const uint16_t* gelu_table_device(hipStream_t s);
// D3PM_AB_GELU_TABLE: bf16 GELU epilogues read an LDS table (d3pm_mfma_tile.h).  Not shipped: measured SLOWER on MI355X -- fc1 +
// GELU at M = 24576 with 192 x 256 tiles 86.4 us with the table vs 72.6 us with the polynomial (67.9 us for the shipped 192 x 128
// geometry); 64 random 2-byte LDS reads per instruction cost more than the 17 vector instructions they replace
int gelu_table_enabled() { return ab_knobs().gelu_table; }
#else
static int gelu_table_enabled() { return 0; }
#endif

// 0 = not applicable, else the geometry id.  `want` (tuning knob): 0 auto, 1 / 2 / 3 forced.
int big_linear_tile(int dtype, const LinearArgs& a, int want) {
  if (dtype != D3PM_F16 && dtype != D3PM_BF16) return 0;
  if (a.K < 4 * BK || a.K % (2 * BK) != 0 || a.M < 96 || a.N < 128) return 0;
  if (a.ldx % 8 != 0 || a.ldy % 8 != 0 || !aligned16(a.X) || !aligned16(a.W) || !aligned16(a.Y)) return 0;
  if (a.R1 && (a.ldr % 8 != 0 || !aligned16(a.R1))) return 0;
  if (a.R2 && (!a.R1 || !aligned16(a.R2))) return 0;
  if (!fold_args_ok(a) || (a.fold_s && a.K != 512)) return 0;      // the big tile prefetches the 16 moment parts of a 512-wide row
  const bool gelu = a.act == ACT_GELU, r1 = a.R1 != nullptr, r2 = a.R2 != nullptr, mk = a.row_mask != nullptr;
  if (a.act != ACT_NONE && !gelu) return 0;
  if (gelu && (r1 || mk)) return 0;                       // instantiated epilogues: plain, GELU, R1, R1+R2, R1+mask
  if (mk && (!r1 || r2)) return 0;
  if (a.ldx >= (1 << 24) || a.K >= (1 << 24)) return 0;       // 32-bit per-lane DMA offsets
  auto fits = [&](int id, bool forced) {
    int tm, tn, waves;
    big_geometry(id, tm, tn, waves);
    if (a.M % tm != 0 || a.N % tn != 0) return false;
    if (forced) return true;                                        // tuning knob / kernel tests: any shape of whole tiles
    const long long slots = 256 * (8 / waves);
    const long long tiles = static_cast<long long>(a.M / tm) * (a.N / tn), rounds = (tiles + slots - 1) / slots;
    return tiles * 5 >= slots * 4 && tiles * 100 >= rounds * slots * 85;   // >= 85 % of the slot-rounds do work
  };
  if (want >= 1 && want <= 3) return fits(want, true) ? want : 0;
  // Under the GELU epilogue 192 x 128 tiles win clearly: its VALU work only overlaps with MFMAs when a second workgroup shares the
  // CU (two 4-wave workgroups per CU).  For the other K = 512 projections the microbenchmark cannot tell 192 x 128 from 192 x 256
  // (qkv 43.3 vs 44.5 us, merged q 30.5 vs 31.4: profiles/round3_v_ab_gemm_geometries.txt) but the sampler's loop can, where the
  // operands arrive cold from the previous launch and two independent workgroups per CU hide that better than one: GEMM class
  // 1317 -> 1241 us per iteration, 97.8 k -> 101.2 k tokens/s with 192 x 128 everywhere (interleaved arms of bench.py --tune
  // gemm_variant=8, profiles/round3_v_ab_tune_geom.txt)
  // -- and fc2 (K = 2048) likewise: with 192 x 128 for the K = 512 projections only the class is at 1294 us, with fc2 too at 1249
  // (profiles/round3_w_ab_tune_geom.txt).  So: two 4-wave workgroups per CU wherever that geometry fills its rounds.
  if (fits(3, false)) return 3;
  if (fits(2, false)) return 2;
  if (fits(3, false)) return 3;
  if (fits(1, false)) return 1;
  // Mid-size batches (the VCTK config: 32 x 384 rows; 8 .. 16 utterances of the libritts shape) leave the chip partly idle with
  // every big tile, yet 192 x 128 tiles of four waves still beat the 128 x 128 kernels there once at least half (long K) or all
  // of the CUs get one: measured at M = 12288 / 6144 (tests/ab_gemm_shapes.py, profiles/round3_f_ab_gemm_shapes.txt): qkv 24.1 vs
  // 26.6 / 15.6 vs 17.2 us, out-projection 12.8 vs 15.8 us, fc2 31.8 vs 39.0 / 29.5 vs 35.4 us
  if (a.M % 192 == 0 && a.N % 128 == 0) {
    const long long tiles = static_cast<long long>(a.M / 192) * (a.N / 128);
    if (tiles >= (a.K >= 1024 ? 128 : 256)) return 3;
  }
  return 0;
}

// kernel MODE template argument: 1 = the shipped schedule (hand-placed reads); every other value is an experiment that
// exists in libd3pm_hip_ab.so only (include/d3pm_hip_ab.h, D3PM_AB_GEMM_BIG_MODE)
#ifdef D3PM_ABLATIONS
int big_gemm_mode() { return ab_knobs().big_mode; }
int read_big_gemm_stamp(unsigned long long* out) {
  D3PM_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_big_stamp), 2 * sizeof(unsigned long long)));
  return D3PM_OK;
}
#endif

template <typename U, int E, int WM, int WN, int MD>
static int big_launch(const LinearArgs& a, int n_tiles, int tiles_total, dim3 grid, size_t lds, hipStream_t s) {
  const uint16_t* tab = nullptr;
#ifdef D3PM_ABLATIONS
  if ((E & EPI_GELU) && std::is_same<U, bf16>::value && WM * WN == 8 && lds + GELU_TAB_BYTES <= 160 * 1024 && gelu_table_enabled()) {
    tab = gelu_table_device(s);
    if (tab) lds += GELU_TAB_BYTES;
  }
#endif
  D3PM_LDS_ATTR((&gemm_mfma_big<U, E, WM, WN, MD>), 160 * 1024);
  gemm_mfma_big<U, E, WM, WN, MD><<<grid, dim3(WM * WN * 64), lds, s>>>(
      static_cast<const U*>(a.X), a.ldx, static_cast<const U*>(a.W), static_cast<const U*>(a.bias), static_cast<U*>(a.Y), a.ldy,
      static_cast<const U*>(a.R1), static_cast<const U*>(a.R2), a.ldr, a.row_mask, a.mask_period, a.M, a.N, a.K, n_tiles,
      tiles_total, tab, RowPanelArgs<U>{}, epi_fold_of(a));
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

template <typename U, int E>
static int big_launch_geometry(int id, const LinearArgs& a, int n_tiles, int tiles_total, dim3 grid, size_t lds, hipStream_t s) {
#ifndef D3PM_ABLATIONS
  if (id == 1) return big_launch<U, E, 1, 8, 1>(a, n_tiles, tiles_total, grid, lds, s);
  if (id == 2) return big_launch<U, E, 2, 4, 1>(a, n_tiles, tiles_total, grid, lds, s);
  return big_launch<U, E, 2, 2, 1>(a, n_tiles, tiles_total, grid, lds, s);
#else
  const int md = big_gemm_mode();
  if (E == 0 && id == 3 && md >= 16) {
    switch (md) {
      case 17: return big_launch<U, 0, 2, 2, 17>(a, n_tiles, tiles_total, grid, lds, s);
      case 33: return big_launch<U, 0, 2, 2, 33>(a, n_tiles, tiles_total, grid, lds, s);
      case 209: return big_launch<U, 0, 2, 2, 209>(a, n_tiles, tiles_total, grid, lds, s);
      default: break;
    }
  }
  if (E == 0 && id == 2 && (md == 3 || md == 5)) {
    if (md == 3) return big_launch<U, 0, 2, 4, 3>(a, n_tiles, tiles_total, grid, lds, s);
    return big_launch<U, 0, 2, 4, 5>(a, n_tiles, tiles_total, grid, lds, s);
  }
  if (E == 0 && id == 2 && md >= 16) {      // timing-only ablation builds (tests/ab_gemm.py): plain epilogue, 192 x 256 only
    switch (md) {
      case 17: return big_launch<U, 0, 2, 4, 17>(a, n_tiles, tiles_total, grid, lds, s);
      case 32: return big_launch<U, 0, 2, 4, 32>(a, n_tiles, tiles_total, grid, lds, s);
      case 81: return big_launch<U, 0, 2, 4, 81>(a, n_tiles, tiles_total, grid, lds, s);
      case 145: return big_launch<U, 0, 2, 4, 145>(a, n_tiles, tiles_total, grid, lds, s);
      case 209: return big_launch<U, 0, 2, 4, 209>(a, n_tiles, tiles_total, grid, lds, s);
      case 257: return big_launch<U, 0, 2, 4, 257>(a, n_tiles, tiles_total, grid, lds, s);
      case 465: return big_launch<U, 0, 2, 4, 465>(a, n_tiles, tiles_total, grid, lds, s);
      default: break;
    }
  }
  if (md == 4129 && E == 0 && id == 2) return big_launch<U, 0, 2, 4, 4129>(a, n_tiles, tiles_total, grid, lds, s);   // timing probe
  if (md == 33 && E == 0 && id == 2) return big_launch<U, 0, 2, 4, 33>(a, n_tiles, tiles_total, grid, lds, s);       // its LDS-DMA twin
  if (md == 2049 && E == 0) {      // A/B: all DMA pieces of a k-step issued at its top (tests/ab_gemm.py), plain epilogue only
    if (id == 2) return big_launch<U, 0, 2, 4, 2049>(a, n_tiles, tiles_total, grid, lds, s);
  }
  if (md == 8193 || md == 16385 || md == 24577) {      // staggered XCD groups (A/B: tests/ab_gemm.py), 192 x 256 and plain / R1 epilogues only
    if (id == 2 && md == 8193) return big_launch<U, E, 2, 4, 8193>(a, n_tiles, tiles_total, grid, lds, s);
    if (id == 2 && md == 16385) return big_launch<U, E, 2, 4, 16385>(a, n_tiles, tiles_total, grid, lds, s);
    if (id == 2 && md == 24577) return big_launch<U, E, 2, 4, 24577>(a, n_tiles, tiles_total, grid, lds, s);
  }
  if (md == 32769) {    // A/B: epilogue operands prefetched at the top of the tile (tests/ab_gemm.py)
    if (id == 1) return big_launch<U, E, 1, 8, 32769>(a, n_tiles, tiles_total, grid, lds, s);
    if (id == 2) return big_launch<U, E, 2, 4, 32769>(a, n_tiles, tiles_total, grid, lds, s);
    return big_launch<U, E, 2, 2, 32769>(a, n_tiles, tiles_total, grid, lds, s);
  }
  if (md == 513) {      // hand-placed schedule with non-temporal output stores (A/B: tests/ab_gemm.py)
    if (id == 1) return big_launch<U, E, 1, 8, 513>(a, n_tiles, tiles_total, grid, lds, s);
    if (id == 2) return big_launch<U, E, 2, 4, 513>(a, n_tiles, tiles_total, grid, lds, s);
    return big_launch<U, E, 2, 2, 513>(a, n_tiles, tiles_total, grid, lds, s);
  }
  const bool hand = (md & 1) != 0;
  // deferred stores need a next tile to hide in (more tiles than persistent workgroups) and K >= 8 k-steps
  const bool drip = md == 9 && a.K >= 8 * BK && tiles_total > static_cast<int>(grid.x);
  if (id == 1) return drip ? big_launch<U, E, 1, 8, 9>(a, n_tiles, tiles_total, grid, lds, s) : hand ? big_launch<U, E, 1, 8, 1>(a, n_tiles, tiles_total, grid, lds, s) : big_launch<U, E, 1, 8, 0>(a, n_tiles, tiles_total, grid, lds, s);
  if (id == 2) return drip ? big_launch<U, E, 2, 4, 9>(a, n_tiles, tiles_total, grid, lds, s) : hand ? big_launch<U, E, 2, 4, 1>(a, n_tiles, tiles_total, grid, lds, s) : big_launch<U, E, 2, 4, 0>(a, n_tiles, tiles_total, grid, lds, s);
  return drip ? big_launch<U, E, 2, 2, 9>(a, n_tiles, tiles_total, grid, lds, s) : hand ? big_launch<U, E, 2, 2, 1>(a, n_tiles, tiles_total, grid, lds, s) : big_launch<U, E, 2, 2, 0>(a, n_tiles, tiles_total, grid, lds, s);
#endif
}

// the folded-LayerNorm epilogues (EPI_LNF / EPI_STATS, d3pm_mfma_tile.h) run on the shipped schedule of each geometry only
template <typename U, int E>
static int big_launch_fold(int id, const LinearArgs& a, int n_tiles, int tiles_total, dim3 grid, size_t lds, hipStream_t s) {
  if (id == 1) return big_launch<U, E, 1, 8, 1>(a, n_tiles, tiles_total, grid, lds, s);
  if (id == 2) return big_launch<U, E, 2, 4, 1>(a, n_tiles, tiles_total, grid, lds, s);
  return big_launch<U, E, 2, 2, 1>(a, n_tiles, tiles_total, grid, lds, s);
}

// Both cross-attention out-projections (ar_discrete.py:138,142: the SAME weights) in one launch of the ordinary big-tile geometry:
// a second product through the tile, the first result kept packed in 48 registers, x' = rn(rn(R1 + rn(X W^T + b)) + rn(X2 W^T + b))
// -- the bits of the two launches it replaces -- and, with stats_out, the row moments of x' for the folded norm3 of fc1.
bool big_dual_supported(int dtype, const LinearArgs& a, const void* X2) {
  if (!X2 || !aligned16(X2) || !a.R1 || a.R2 || a.row_mask || a.act != ACT_NONE || !a.bias || a.fold_s) return false;
  return big_linear_tile(dtype, a, tune_of(a.tune).gemm_variant == 6 ? 2 : tune_of(a.tune).gemm_variant == 8 ? 3 : 0) >= 2;
}

template <typename U, int E, int WM, int WN>
static int big_dual_launch(const LinearArgs& a, const void* X2, int n_tiles, int tiles_total, dim3 grid, size_t lds, hipStream_t s) {
  D3PM_LDS_ATTR((&gemm_mfma_big<U, E, WM, WN, 1, FUSE_DUAL>), 160 * 1024);
  RowPanelArgs<U> rp{};
  rp.X2 = static_cast<const U*>(X2);
  gemm_mfma_big<U, E, WM, WN, 1, FUSE_DUAL><<<grid, dim3(WM * WN * 64), lds, s>>>(
      static_cast<const U*>(a.X), a.ldx, static_cast<const U*>(a.W), static_cast<const U*>(a.bias), static_cast<U*>(a.Y), a.ldy,
      static_cast<const U*>(a.R1), nullptr, a.ldr, nullptr, 1, a.M, a.N, a.K, n_tiles, tiles_total, nullptr, rp, epi_fold_of(a));
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

int big_dual(int dtype, const LinearArgs& a, const void* X2, hipStream_t s) {
  const int id = big_linear_tile(dtype, a, tune_of(a.tune).gemm_variant == 6 ? 2 : tune_of(a.tune).gemm_variant == 8 ? 3 : 0);
  int tm, tn, waves;
  big_geometry(id, tm, tn, waves);
  const int n_tiles = a.N / tn, tiles_total = (a.M / tm) * n_tiles;
  const int slots = 256 * (8 / waves), want = (tiles_total + 7) & ~7;
  const dim3 grid(static_cast<unsigned>(want < slots ? want : slots));
  const size_t lds = 2 * static_cast<size_t>(tm + tn) * ROW_BYTES;
  auto go = [&](auto* tag) -> int {
    using U = std::remove_pointer_t<decltype(tag)>;
    if (a.stats_out)
      return id == 2 ? big_dual_launch<U, EPI_R2 | EPI_STATS, 2, 4>(a, X2, n_tiles, tiles_total, grid, lds, s)
                     : big_dual_launch<U, EPI_R2 | EPI_STATS, 2, 2>(a, X2, n_tiles, tiles_total, grid, lds, s);
    return id == 2 ? big_dual_launch<U, EPI_R2, 2, 4>(a, X2, n_tiles, tiles_total, grid, lds, s)
                   : big_dual_launch<U, EPI_R2, 2, 2>(a, X2, n_tiles, tiles_total, grid, lds, s);
  };
  return dtype == D3PM_F16 ? go(static_cast<f16*>(nullptr)) : go(static_cast<bf16*>(nullptr));
}

int big_linear(int dtype, const LinearArgs& a, int id, hipStream_t s) {
  int tm, tn, waves;
  big_geometry(id, tm, tn, waves);
  const int n_tiles = a.N / tn, tiles_total = (a.M / tm) * n_tiles;
  const int slots = 256 * (8 / waves), want = (tiles_total + 7) & ~7;
  const dim3 grid(static_cast<unsigned>(want < slots ? want : slots));
  const size_t lds = 2 * static_cast<size_t>(tm + tn) * ROW_BYTES;
  const int epi = (a.act == ACT_GELU ? EPI_GELU : 0) | (a.R1 ? (a.R2 ? EPI_R2 : EPI_R1) : 0) | (a.row_mask ? EPI_MASK : 0) |
                  (a.fold_s ? EPI_LNF : 0) | (a.stats_out ? EPI_STATS : 0);
  auto go = [&](auto* tag) -> int {
    using U = std::remove_pointer_t<decltype(tag)>;
    switch (epi) {
      case EPI_LNF: return big_launch_fold<U, EPI_LNF>(id, a, n_tiles, tiles_total, grid, lds, s);
      case EPI_LNF | EPI_GELU: return big_launch_fold<U, EPI_LNF | EPI_GELU>(id, a, n_tiles, tiles_total, grid, lds, s);
      case EPI_R1 | EPI_STATS: return big_launch_fold<U, EPI_R1 | EPI_STATS>(id, a, n_tiles, tiles_total, grid, lds, s);
      case EPI_R2 | EPI_STATS: return big_launch_fold<U, EPI_R2 | EPI_STATS>(id, a, n_tiles, tiles_total, grid, lds, s);
      case EPI_R1 | EPI_MASK | EPI_STATS: return big_launch_fold<U, EPI_R1 | EPI_MASK | EPI_STATS>(id, a, n_tiles, tiles_total, grid, lds, s);
      case 0: return big_launch_geometry<U, 0>(id, a, n_tiles, tiles_total, grid, lds, s);
      case EPI_GELU: return big_launch_geometry<U, EPI_GELU>(id, a, n_tiles, tiles_total, grid, lds, s);
      case EPI_R1: return big_launch_geometry<U, EPI_R1>(id, a, n_tiles, tiles_total, grid, lds, s);
      case EPI_R2: return big_launch_geometry<U, EPI_R2>(id, a, n_tiles, tiles_total, grid, lds, s);
      case EPI_R1 | EPI_MASK: return big_launch_geometry<U, EPI_R1 | EPI_MASK>(id, a, n_tiles, tiles_total, grid, lds, s);
      default: break;
    }
    return D3PM_E_SHAPE;
  };
  return dtype == D3PM_F16 ? go(static_cast<f16*>(nullptr)) : go(static_cast<bf16*>(nullptr));
}

// ---- row-panel fusion: projection onto the residual stream + the LayerNorm(s) that read it next, one launch ----------
// Instantiated: out-proj + x -> norm2 | norm22 (EPI_R1, two LayerNorms); both cross-attention out-projections + x -> norm3 with
// FiLM (EPI_R2 from registers); fc2 + x, frame mask -> the next block's norm1 (EPI_R1 | EPI_MASK).
static int row_panel_kind(const LinearArgs& a, const RowPanelFuse& f) {
  const bool r1 = a.R1 != nullptr, mk = a.row_mask != nullptr, dual = f.X2 != nullptr, ln2 = f.lnw2 != nullptr, film = f.film != nullptr;
  if (!r1 || a.R2 || !f.lnw || !f.lnb || !f.lny || a.act != ACT_NONE) return 0;
  if (ln2 != (f.lnb2 != nullptr) || ln2 != (f.lny2 != nullptr)) return 0;
  if (!dual && !mk && ln2 && !film) return 1;
  if (dual && !mk && !ln2 && film) return 2;
  if (!dual && mk && !ln2 && !film) return 3;
  return 0;
}

bool row_panel_supported(int dtype, const LinearArgs& a, const RowPanelFuse& f) {
  if (dtype != D3PM_F16 && dtype != D3PM_BF16) return false;
  if (a.N != 512 || a.ldy != 512 || a.ldr != 512 || a.M < 96 || a.M % 96 != 0 || a.K < 4 * BK || a.K % (2 * BK) != 0) return false;
  if (a.ldx % 8 != 0 || a.ldx >= (1 << 24) || a.K >= (1 << 24) || !a.bias) return false;
  for (const void* p : {a.X, a.W, static_cast<const void*>(a.Y), a.R1, a.bias, f.X2, f.lnw, f.lnb, f.lnw2, f.lnb2, f.film,
                        static_cast<const void*>(f.lny), static_cast<const void*>(f.lny2)})
    if (!aligned16(p)) return false;
  if ((f.sx != nullptr) != (f.sx2 != nullptr) && f.lnw2) return false;     // both LayerNorm outputs in the same format
  if (f.sx && row_panel_kind(a, f) == 3) return false;                       // MX outputs: the two forms the fp8 fast path uses
  return row_panel_kind(a, f) != 0;
}

template <typename U, int E, int FUSE>
static int row_panel_launch(const LinearArgs& a, const RowPanelFuse& f, hipStream_t s) {
  static_assert((FUSE & FUSE_MX) == 0 || (FUSE & FUSE_LN) != 0, "MX output = LayerNorm output");
  const int tiles_total = a.M / 96, want = (tiles_total + 7) & ~7;
  const dim3 grid(static_cast<unsigned>(want < 256 ? want : 256));
  const size_t lds = 2 * static_cast<size_t>(96 + 512) * ROW_BYTES + 2 * 96 * 8 * sizeof(float);
  D3PM_LDS_ATTR((&gemm_mfma_big<U, E, 1, 8, 1, FUSE>), 160 * 1024);
  RowPanelArgs<U> rp{static_cast<const U*>(f.X2), static_cast<const U*>(f.lnw), static_cast<const U*>(f.lnb),
                     static_cast<const U*>(f.lnw2), static_cast<const U*>(f.lnb2), static_cast<const U*>(f.film),
                     static_cast<U*>(f.lny), static_cast<U*>(f.lny2), f.eps, static_cast<uint8_t*>(f.sx), static_cast<uint8_t*>(f.sx2)};
  gemm_mfma_big<U, E, 1, 8, 1, FUSE><<<grid, dim3(512), lds, s>>>(
      static_cast<const U*>(a.X), a.ldx, static_cast<const U*>(a.W), static_cast<const U*>(a.bias), static_cast<U*>(a.Y), a.ldy,
      static_cast<const U*>(a.R1), nullptr, a.ldr, a.row_mask, a.mask_period, a.M, a.N, a.K, 1, tiles_total, nullptr, rp, EpiFold{});
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

int row_panel_linear(int dtype, const LinearArgs& a, const RowPanelFuse& f, hipStream_t s) {
  const int kind = row_panel_kind(a, f);
  auto go = [&](auto* tag) -> int {
    using U = std::remove_pointer_t<decltype(tag)>;
    switch (kind) {
      case 1:
        if (f.sx) return row_panel_launch<U, EPI_R1, FUSE_LN | FUSE_LN2 | FUSE_MX>(a, f, s);
#ifdef D3PM_ABLATIONS
        if (big_gemm_mode() == 1025) return row_panel_launch<U, EPI_R1, FUSE_LN | FUSE_LN2 | FUSE_ABL_NOLN>(a, f, s);
#endif
        return row_panel_launch<U, EPI_R1, FUSE_LN | FUSE_LN2>(a, f, s);
      case 2: return f.sx ? row_panel_launch<U, EPI_R2, FUSE_DUAL | FUSE_LN | FUSE_FILM | FUSE_MX>(a, f, s)
                          : row_panel_launch<U, EPI_R2, FUSE_DUAL | FUSE_LN | FUSE_FILM>(a, f, s);
      case 3: return row_panel_launch<U, EPI_R1 | EPI_MASK, FUSE_LN>(a, f, s);
      default: break;
    }
    set_error("row-panel projection: unsupported epilogue combination");
    return D3PM_E_SHAPE;
  };
  return dtype == D3PM_F16 ? go(static_cast<f16*>(nullptr)) : go(static_cast<bf16*>(nullptr));
}

}  // namespace d3pm
