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
//   * one workgroup of EIGHT waves per CU owns a 192 x 256 (or 96 x 512) output tile -- 110 (81) flop per staged
//     byte -- and each wave a 96 x 64 sub-tile (6 x 4 MFMA tiles of 16 x 16, 96 accumulator registers);
//   * the tile shapes divide the bench workload exactly: M = 32 utterances x 768 rows = 128 x 192 rows, N = 512 /
//     1536 / 2048 = 2 / 6 / 8 x 256 columns, so every projection is a whole number of rounds over the 256 CUs
//     (256 / 512 / 768 / 1024 tiles) -- the 128 x 128 / 256 x 256 grids left a quarter of the chip idle in the last
//     round;
//   * two LDS stages of (TM + TN) x 128 B; the next k-step's 1-KiB DMA pieces (global_load_lds_dwordx4 from inline
//     asm, invisible to hipcc's wait counters) are issued one per four MFMAs inside the current k-step, so the
//     memory pipeline's issue back-pressure hides under the partner wave's MFMAs; one s_barrier per k-step with a
//     counted vmcnt; the stream of k-steps runs across tile boundaries (the next tile's first k-step is in flight
//     under the epilogue) and the epilogue's 16-byte stores stay in flight into the next tile;
//   * same swizzled 128-byte-row LDS image, same D = W_frag . X_frag^T orientation and the same epilogue code as the
//     128 x 128 kernels (d3pm_mfma_tile.h): the accumulation order over k is identical, results are bit-identical.
#include "d3pm_kernels.h"
#include "d3pm_mfma_tile.h"

namespace d3pm {
namespace {

template <typename T, int EPI, int WM>
__global__ __launch_bounds__(512, 2) void gemm_mfma_big(const T* __restrict__ X, int ldx, const T* __restrict__ W,
                                                        const T* __restrict__ bias, T* Y, int ldy, const T* R1,
                                                        const T* R2, int ldr, const uint8_t* __restrict__ row_mask,
                                                        int mask_period, int M, int N, int K, int n_tiles,
                                                        int tiles_total) {
  constexpr int WN = 8 / WM, TM = 96 * WM, TN = 64 * WN;
  constexpr int XD = TM / 8, WD = TN / 8;              // 1-KiB DMA pieces (8 rows x 128 B) per k-step and operand
  constexpr int XPW = (XD + 7) / 8, WPW = WD / 8;      // pieces per wave
  constexpr int NDMA = XPW + WPW;                      // 7 (192 x 256) or 10 (96 x 512; waves 4..7 repeat an X piece)
  constexpr int X_BYTES = TM * ROW_BYTES, STAGE = (TM + TN) * ROW_BYTES;
  static_assert(NDMA <= 12, "one DMA piece per group of four MFMAs");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  // XCD x = blockIdx & 7 owns a contiguous range of tiles (the n-tiles of one X panel then share an L2)
  const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
  const int tq = tiles_total >> 3, tr = tiles_total & 7;
  const int lo = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq, cnt = tq + (xcd < tr ? 1 : 0);
  int t = blockIdx.x >> 3;
  if (t >= cnt) return;                                              // block-uniform
  const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem));
  // DMA piece j of an operand tile = rows 8j .. 8j+7; a wave takes pieces j = wave + 8p, so the swizzle key
  // (row >> 1) & 7 = (4 (j & 1) + (lane >> 4)) & 7 is the same for all of its pieces: one per-lane offset per operand
  const int lrow = lane >> 3, logical = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
  const uint32_t ox = static_cast<uint32_t>(lrow * ldx + logical * 8) * 2u;
  const uint32_t ow = static_cast<uint32_t>(lrow * K + logical * 8) * 2u;
  auto dma = [&](int p, const T* px, const T* pw, uint32_t stage) __attribute__((always_inline)) {   // p: unrolled constant
    if (p < XPW) {
      int j = wave + 8 * p;
      if (XD % 8 != 0 && j >= XD) j = (XD / 8) * 8 + (wave & 3);     // same parity as `wave`: a harmless repeat
      glds16_asm_s(px + static_cast<size_t>(8 * j) * ldx, ox, stage + j * 1024);
    } else {
      const int j = wave + 8 * (p - XPW);
      glds16_asm_s(pw + static_cast<size_t>(8 * j) * K, ow, stage + X_BYTES + j * 1024);
    }
  };
  // fragment addresses: row = base + 16 q + (lane & 15); the swizzle key (row >> 1) & 7 = (lane & 15) >> 1 because
  // every base is a multiple of 16
  const int frow = lane & 15, fch = lane >> 4, fkey = (frow >> 1) & 7;
  const int fo0 = frow * ROW_BYTES + ((fch ^ fkey) << 4), fo1 = frow * ROW_BYTES + (((4 + fch) ^ fkey) << 4);
  const char* const fx_base = smem + wm * 96 * ROW_BYTES;
  const char* const fw_base = smem + X_BYTES + wn * 64 * ROW_BYTES;

  const int nk = K / BK;                                             // even (checked by the launcher)
  int tile = lo + t;
  const T* sx = X + static_cast<size_t>((tile / n_tiles) * TM) * ldx;
  const T* sw = W + static_cast<size_t>((tile % n_tiles) * TN) * K;
#pragma unroll
  for (int p = 0; p < NDMA; ++p) dma(p, sx, sw, lds_base);          // first k-step of the first tile
  // the first step of a tile waits with vmcnt(12): behind an epilogue the DMA pieces are older than its 12 stores; the very
  // first tile has no stores behind its pieces, so they are waited for here
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (;;) {
    floatx4 acc[4][6];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 6; ++b) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int m0 = (tile / n_tiles) * TM, n0 = (tile % n_tiles) * TN;
    // what follows this tile (block-uniform); the last tile re-reads its own first k-step: valid memory, never used
    const int t_next = t + per_xcd;
    const bool more = t_next < cnt;
    const int tile_next = more ? lo + t_next : tile;
    const T* sx_next = X + static_cast<size_t>((tile_next / n_tiles) * TM) * ldx;
    const T* sw_next = W + static_cast<size_t>((tile_next % n_tiles) * TN) * K;

    // one k-step on stage S while the DMA pieces of the following k-step (px, pw) go to stage S ^ 1
    auto step = [&](const int S, const bool first, const T* px, const T* pw) __attribute__((always_inline)) {   // S, first: constants
      const char* bx = fx_base + S * STAGE;
      const char* bw = fw_base + S * STAGE;
      const uint32_t nxt = lds_base + (S ^ 1) * STAGE;
      __builtin_amdgcn_sched_barrier(0);
      if (first) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();        // every wave's pieces of this k-step have landed; stage S ^ 1 is no longer read
      __builtin_amdgcn_sched_barrier(0);
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
        if (g < NDMA) dma(g, px, pw, nxt);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt][mt] = mma<T>(fw[ks][nt], fx[g % 3], acc[nt][mt]);
      }
      __builtin_amdgcn_sched_barrier(0);
    };

    step(0, true, sx + BK, sw + BK);
    step(1, false, sx + 2 * BK, sw + 2 * BK);
    for (int kt = 2; kt < nk; kt += 2) {                 // nk is even and >= 4
      step(0, false, sx + (kt + 1) * BK, sw + (kt + 1) * BK);
      const bool last = kt + 2 >= nk;
      step(1, false, last ? sx_next : sx + (kt + 2) * BK, last ? sw_next : sw + (kt + 2) * BK);
    }
    epilogue_store<T, EPI, 4, 6, true>(acc, bias, Y, ldy, R1, R2, ldr, row_mask, mask_period, M, N, m0 + wm * 96,
                                       n0 + wn * 64, lane);
    if (!more) break;
    t = t_next;
    tile = tile_next;
    sx = sx_next;
    sw = sw_next;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the unused look-ahead pieces must not outlive the workgroup's LDS
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) % 16) == 0; }

}  // namespace

// 0 = not applicable; 1 = 96 x 512 tiles, 2 = 192 x 256 tiles.  `want` (tuning knob): 0 auto, 1 / 2 forced.
int big_linear_tile(int dtype, const LinearArgs& a, int want) {
  if (dtype != D3PM_F16 && dtype != D3PM_BF16) return 0;
  if (a.K < 4 * BK || a.K % (2 * BK) != 0 || a.M < 96 || a.N < 256) return 0;
  if (a.ldx % 8 != 0 || a.ldy % 8 != 0 || !aligned16(a.X) || !aligned16(a.W) || !aligned16(a.Y)) return 0;
  if (a.R1 && (a.ldr % 8 != 0 || !aligned16(a.R1))) return 0;
  if (a.R2 && (!a.R1 || !aligned16(a.R2))) return 0;
  const bool gelu = a.act == ACT_GELU, r1 = a.R1 != nullptr, r2 = a.R2 != nullptr, mk = a.row_mask != nullptr;
  if (a.act != ACT_NONE && !gelu) return 0;
  if (gelu && (r1 || mk)) return 0;                       // instantiated epilogues: plain, GELU, R1, R1+R2, R1+mask
  if (mk && (!r1 || r2)) return 0;
  if (a.ldx >= (1 << 24) || a.K >= (1 << 24)) return 0;       // 32-bit per-lane DMA offsets
  auto fits = [&](int wm, bool forced) {
    const int tm = 96 * wm, tn = 64 * (8 / wm);
    if (a.M % tm != 0 || a.N % tn != 0) return false;
    if (forced) return true;                                        // tuning knob / kernel tests: any shape of whole tiles
    const long long tiles = static_cast<long long>(a.M / tm) * (a.N / tn), rounds = (tiles + 255) / 256;
    return tiles >= 200 && tiles * 100 >= rounds * 256 * 85;        // >= 85 % of the CU-rounds do work
  };
  if (want == 1 || want == 2) return fits(want, true) ? want : 0;
  if (fits(2, false)) return 2;
  return fits(1, false) ? 1 : 0;
}

int big_linear(int dtype, const LinearArgs& a, int wm, hipStream_t s) {
  const int tm = 96 * wm, tn = 64 * (8 / wm);
  const int n_tiles = a.N / tn, tiles_total = (a.M / tm) * n_tiles;
  const int want = (tiles_total + 7) & ~7;
  const dim3 grid(static_cast<unsigned>(want < 256 ? want : 256)), block(512);
  const size_t lds = 2 * static_cast<size_t>(tm + tn) * ROW_BYTES;
  const int epi = (a.act == ACT_GELU ? EPI_GELU : 0) | (a.R1 ? (a.R2 ? EPI_R2 : EPI_R1) : 0) | (a.row_mask ? EPI_MASK : 0);
#define D3PM_BIG(E, WMV)                                                                                              \
  do {                                                                                                                \
    static bool attr_set = false;                                                                                     \
    if (!attr_set) {                                                                                                  \
      D3PM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mfma_big<U, E, WMV>),                    \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                    \
      attr_set = true;                                                                                                \
    }                                                                                                                 \
    gemm_mfma_big<U, E, WMV><<<grid, block, lds, s>>>(static_cast<const U*>(a.X), a.ldx, static_cast<const U*>(a.W),  \
        static_cast<const U*>(a.bias), static_cast<U*>(a.Y), a.ldy, static_cast<const U*>(a.R1),                     \
        static_cast<const U*>(a.R2), a.ldr, a.row_mask, a.mask_period, a.M, a.N, a.K, n_tiles, tiles_total);         \
    return D3PM_OK;                                                                                                   \
  } while (0)
#define D3PM_BIG_WM(E)         \
  do {                         \
    if (wm == 2) D3PM_BIG(E, 2); \
    else D3PM_BIG(E, 1);       \
  } while (0)
  auto go = [&](auto* tag) -> int {
    using U = std::remove_pointer_t<decltype(tag)>;
    switch (epi) {
      case 0: D3PM_BIG_WM(0);
      case EPI_GELU: D3PM_BIG_WM(EPI_GELU);
      case EPI_R1: D3PM_BIG_WM(EPI_R1);
      case EPI_R2: D3PM_BIG_WM(EPI_R2);
      case EPI_R1 | EPI_MASK: D3PM_BIG_WM(EPI_R1 | EPI_MASK);
      default: break;
    }
    return D3PM_E_SHAPE;
  };
  int rc = dtype == D3PM_F16 ? go(static_cast<f16*>(nullptr)) : go(static_cast<bf16*>(nullptr));
#undef D3PM_BIG_WM
#undef D3PM_BIG
  if (rc != D3PM_OK) return rc;
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

}  // namespace d3pm
