// d3pm_mfma_gemm_ring.hip -- 192 x 256 persistent MFMA GEMM with a five-slot ring of 32-k operand slabs.
//
//   Y[M][N] = epilogue(X[M][K] . W[N][K]^T + bias)      same contract and epilogue as d3pm_mfma_gemm.hip
//
// replaces the nn.Linear / MultiheadAttention projections of DiTBlock.forward
// (/root/reference/vall_e/vall_e/ar_discrete.py:132,138,142,159) at throughput batch sizes (experimental schedule, A/B knob).
//
// An experiment that did NOT pay (kept as D3PM_TUNE_GEMM_VARIANT = 9, bit-identical, for the record): two 57-KB stages fill the
// LDS of the big-tile kernel (d3pm_mfma_gemm_big.hip), so more operand bytes in flight need finer slabs -- here a slab is 32 k
// (64-byte LDS rows, (192 + 256) x 64 B = 28 KB), five slabs ring through 140 KB, a k-step still consumes two of them behind ONE
// barrier, and three are in flight (85 KB) instead of two.  Measured (tests/ab_gemm.py 6m33 9m33 6m1 9m1): the stream alone
// 40.1 us against 36.9, the qkv GEMM 51.6 us against 42.3 -- the operand stream of one 8-wave workgroup is bound by its
// issue -> land -> barrier cadence (~55 GB/s per CU), not by the bytes it keeps in flight.
//   * slab h lives in slot h % 5; k-step g reads slabs 2g, 2g+1 and, behind its barrier, issues slabs 2g+3, 2g+4 into the slots
//     that k-step g-1 has just released; the stream of slabs runs across tile boundaries;
//   * 64-byte rows: 16-byte chunk c of row r sits at chunk c ^ ((r >> 2) & 3) -- a 16-row fragment read covers one contiguous
//     KiB, conflict-free; a DMA piece is 16 rows x 64 B;
//   * same D = W_frag . X_frag^T orientation, same k order and the same epilogue code as the other schedules: bit-identical.
// Compiled into libd3pm_hip_ab.so only (-DD3PM_ABLATIONS; include/d3pm_hip_ab.h): built, measured, not shipped.
#ifdef D3PM_ABLATIONS
#include "d3pm_kernels.h"
#include "d3pm_mfma_tile.h"

namespace d3pm {
namespace {

constexpr int RTM = 192, RTN = 256, RNW = 8;
constexpr int HROW = 64;                                   // bytes per row of a 32-k slab
constexpr int XH = RTM * HROW, SLOT = (RTM + RTN) * HROW;  // 12288, 28672
constexpr int NSLOT = 5;
constexpr int XPC = RTM / 16, WPC = RTN / 16, HPC = XPC + WPC;   // 1-KiB pieces per slab: 12 + 16 = 28

template <typename T, int EPI, bool kProbe = false>   // kProbe: timing only, the operand stream without fragment reads / MFMAs
__global__ __launch_bounds__(RNW * 64, 2) void gemm_mfma_ring(const T* __restrict__ X, int ldx, const T* __restrict__ W,
                                                              const T* __restrict__ bias, T* Y, int ldy, const T* R1, const T* R2,
                                                              int ldr, const uint8_t* __restrict__ row_mask, int mask_period, int M,
                                                              int N, int K, int n_tiles, int tiles_total) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  // XCD x = blockIdx & 7 owns a contiguous range of tiles (as the big-tile kernel)
  const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
  const int tq = tiles_total >> 3, tr = tiles_total & 7;
  const int lo = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq, cnt = tq + (xcd < tr ? 1 : 0);
  int t = blockIdx.x >> 3;
  if (t >= cnt) return;
  const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem));
  const int nh = K / 32;                                   // slabs per tile (even: the launcher checks K % 64 == 0)

  // ---- the fetch cursor: the next slab to issue, walking tiles and k independently of the compute cursor
  int ft = t;                                              // index into this workgroup's tile list
  int fh = 0;                                              // slab within that tile
  int fslot = 0;                                           // ring slot of that slab
  auto tile_of = [&](int ti) { return lo + (ti < cnt ? ti : t); };     // past the end: valid memory, never used
  int ftile = tile_of(ft);
  const T* fx = X + static_cast<size_t>((ftile / n_tiles) * RTM) * ldx;
  const T* fw = W + static_cast<size_t>((ftile % n_tiles) * RTN) * K;
  auto advance = [&]() {
    fslot = fslot == NSLOT - 1 ? 0 : fslot + 1;
    if (++fh == nh) {
      fh = 0;
      ft += per_xcd;
      ftile = tile_of(ft);
      fx = X + static_cast<size_t>((ftile / n_tiles) * RTM) * ldx;
      fw = W + static_cast<size_t>((ftile % n_tiles) * RTN) * K;
    }
  };
  // a DMA piece = 16 rows x 64 B: lane -> row lane >> 2, physical chunk lane & 3 holding logical chunk (lane & 3) ^ ((row >> 2) & 3)
  const int rip = lane >> 2, logical = (lane & 3) ^ ((rip >> 2) & 3);
  const uint32_t ox = static_cast<uint32_t>(rip * ldx + logical * 8) * 2u;
  const uint32_t ow = static_cast<uint32_t>(rip * K + logical * 8) * 2u;
  auto piece = [&](int q) __attribute__((always_inline)) {   // piece q of the slab under the cursor (wave-uniform q)
    const uint32_t dst = lds_base + fslot * SLOT;
    if (q < XPC) glds16_asm_s(fx + static_cast<size_t>(16 * q) * ldx + fh * 32, ox, dst + q * 1024);
    else glds16_asm_s(fw + static_cast<size_t>(16 * (q - XPC)) * K + fh * 32, ow, dst + XH + (q - XPC) * 1024);
  };
  // prologue: slabs 0, 1, 2 (84 pieces over the 8 waves)
  for (int s = 0; s < 3; ++s) {
    for (int q = wave; q < HPC; q += RNW) piece(q);
    advance();
  }
  // per k-step a wave issues 7 of the 56 pieces of a slab pair: p = wave + 8 i; p < 28 -> first slab
  auto issue_pair = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 3; ++i) piece(wave + 8 * i);                     // p = wave .. wave + 16 < 28
    if (wave < 4) piece(wave + 24);
    advance();
    if (wave >= 4) piece(wave + 24 - HPC);                               // p = wave + 24 >= 28
#pragma unroll
    for (int i = 4; i < 7; ++i) piece(wave + 8 * i - HPC);
    advance();
  };

  const int frow = lane & 15, fch = lane >> 4, fkey = (frow >> 2) & 3;
  const int fo = frow * HROW + ((fch ^ fkey) << 4);
  int cslot = 0;                                           // ring slot of the first slab of the current k-step
  int tile = lo + t;
  bool first = true;                                       // very first k-step of the workgroup: nothing but DMA pieces behind
  for (;;) {
    floatx4 acc[4][6];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 6; ++b) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int m0 = (tile / n_tiles) * RTM, n0 = (tile % n_tiles) * RTN;
    for (int g = 0; g < nh / 2; ++g) {
      // slabs 2g, 2g+1 have landed once at most this wave's youngest pieces (the second slab of the last pair: 3 or 4; behind an
      // epilogue its 12 stores) are outstanding
      if (g == 0 && !first) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      first = false;
      __builtin_amdgcn_s_barrier();                        // every wave's pieces have landed; the previous k-step's slots are free
      issue_pair();
#pragma unroll
      for (int ks = 0; ks < (kProbe ? 0 : 2); ++ks) {
        int slot = cslot + ks;
        slot = slot >= NSLOT ? slot - NSLOT : slot;
        const char* bx = smem + slot * SLOT + (wm * 96) * HROW + fo;
        const char* bw = smem + slot * SLOT + XH + (wn * 64) * HROW + fo;
        uint4 fwv[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) fwv[nt] = *reinterpret_cast<const uint4*>(bw + nt * 16 * HROW);
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) {
          const uint4 fxv = *reinterpret_cast<const uint4*>(bx + mt * 16 * HROW);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc[nt][mt] = mma<T>(fwv[nt], fxv, acc[nt][mt]);
        }
      }
      cslot += 2;
      cslot = cslot >= NSLOT ? cslot - NSLOT : cslot;
    }
    epilogue_store<T, EPI, 4, 6, true>(acc, bias, Y, ldy, R1, R2, ldr, row_mask, mask_period, M, N, m0 + wm * 96, n0 + wn * 64, lane);
    t += per_xcd;
    if (t >= cnt) break;
    tile = lo + t;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // look-ahead pieces must not outlive the workgroup's LDS
}

inline bool aligned16r(const void* p) { return (reinterpret_cast<uintptr_t>(p) % 16) == 0; }

}  // namespace

bool ring_linear_supported(int dtype, const LinearArgs& a) {
  if (dtype != D3PM_F16 && dtype != D3PM_BF16) return false;
  if (a.M % RTM != 0 || a.N % RTN != 0 || a.K < 256 || a.K % 64 != 0) return false;
  if (a.ldx % 8 != 0 || a.ldy % 8 != 0 || !aligned16r(a.X) || !aligned16r(a.W) || !aligned16r(a.Y)) return false;
  if (a.R1 && (a.ldr % 8 != 0 || !aligned16r(a.R1))) return false;
  if (a.R2 && (!a.R1 || !aligned16r(a.R2))) return false;
  if (a.ldx >= (1 << 24) || a.K >= (1 << 24)) return false;
  const bool r1 = a.R1 != nullptr, r2 = a.R2 != nullptr, mk = a.row_mask != nullptr;
  if (a.act != ACT_NONE) return false;                     // instantiated epilogues: plain, R1, R1+R2, R1+mask
  if (mk && (!r1 || r2)) return false;
  return true;
}

int big_gemm_mode();

int ring_linear(int dtype, const LinearArgs& a, hipStream_t s) {
  const int n_tiles = a.N / RTN, tiles_total = (a.M / RTM) * n_tiles, want = (tiles_total + 7) & ~7;
  const dim3 grid(static_cast<unsigned>(want < 256 ? want : 256)), block(RNW * 64);
  const size_t lds = static_cast<size_t>(NSLOT) * SLOT;
  const int epi = (a.R1 ? (a.R2 ? EPI_R2 : EPI_R1) : 0) | (a.row_mask ? EPI_MASK : 0);
#define D3PM_RING(E)                                                                                                     \
  do {                                                                                                                   \
    D3PM_LDS_ATTR((&gemm_mfma_ring<U, E>), NSLOT * SLOT);                                                                \
    gemm_mfma_ring<U, E><<<grid, block, lds, s>>>(static_cast<const U*>(a.X), a.ldx, static_cast<const U*>(a.W),          \
        static_cast<const U*>(a.bias), static_cast<U*>(a.Y), a.ldy, static_cast<const U*>(a.R1), static_cast<const U*>(a.R2), \
        a.ldr, a.row_mask, a.mask_period, a.M, a.N, a.K, n_tiles, tiles_total);                                          \
    return D3PM_OK;                                                                                                      \
  } while (0)
  auto go = [&](auto* tag) -> int {
    using U = std::remove_pointer_t<decltype(tag)>;
    if (epi == 0 && big_gemm_mode() == 33) {               // timing probe (tests/ab_gemm.py 9m33)
      static bool attr_probe = false;
      if (!attr_probe) {
        D3PM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mfma_ring<U, 0, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, NSLOT * SLOT));
        attr_probe = true;
      }
      gemm_mfma_ring<U, 0, true><<<grid, block, lds, s>>>(static_cast<const U*>(a.X), a.ldx, static_cast<const U*>(a.W),
          static_cast<const U*>(a.bias), static_cast<U*>(a.Y), a.ldy, nullptr, nullptr, a.ldr, nullptr, 1, a.M, a.N, a.K, n_tiles, tiles_total);
      return D3PM_OK;
    }
    switch (epi) {
      case 0: D3PM_RING(0);
      case EPI_R1: D3PM_RING(EPI_R1);
      case EPI_R2: D3PM_RING(EPI_R2);
      case EPI_R1 | EPI_MASK: D3PM_RING(EPI_R1 | EPI_MASK);
      default: break;
    }
    return D3PM_E_SHAPE;
  };
#undef D3PM_RING
  const int rc = dtype == D3PM_F16 ? go(static_cast<f16*>(nullptr)) : go(static_cast<bf16*>(nullptr));
  if (rc != D3PM_OK) return rc;
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

}  // namespace d3pm

#endif  // D3PM_ABLATIONS
