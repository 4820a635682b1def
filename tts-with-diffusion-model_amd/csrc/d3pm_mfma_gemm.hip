// d3pm_mfma_gemm.hip -- LDS-tiled MFMA GEMM for the DiT projections on gfx950 (f16 / bf16).
//
//   Y[M][N] = epilogue(X[M][K] . W[N][K]^T + bias)      both operands K-contiguous (torch Linear layout)
//
// replaces every nn.Linear / MultiheadAttention in/out projection of DiTBlock.forward
// (/root/reference/vall_e/vall_e/ar_discrete.py:132,138,142,159) and the final Linear (:776) when the
// shape tiles (K % 64 == 0, 16-byte aligned rows); everything else goes to linear_tiled (generic).
//
// Structure (one workgroup = 4 wave64 = 128 x 128 output tile, K-step 64):
//   * direct-to-LDS staging (global_load_lds_dwordx4, 1 KiB per wave-instruction), no staging VGPRs;
//   * LDS rows are 128 B (64 k); 16-B chunk c of row r lives at chunk c ^ ((r >> 1) & 7): applied to the
//     per-lane SOURCE address of the DMA and undone on the ds_read_b128 fragment reads -- bank-conflict free
//     (checked exhaustively against the gfx950 lane groups; SQ_LDS_BANK_CONFLICT = 0 in profiles/);
//   * v_mfma_f32_16x16x32_{f16,bf16}: each wave owns 64 x 64 = 4 x 4 tiles, fp32 accumulators;
//   * the MFMA is issued as D = W_frag . X_frag^T so that a lane ends up with 4 consecutive output
//     columns of one row; the epilogue regroups them to 8 per lane (v_permlane16_swap) so that bias /
//     GELU / residual / mask run on registers and residual loads and stores are 16 B per lane;
//     epilogue rounding points are the eager model's (d3pm_kernels.h);
//   * tiles are walked in an XCD-aware order (the n-tiles of one X panel share an L2).
// Three schedules of the same arithmetic (bit-identical results):
//   throughput, persistent (default for shapes of whole tiles, >= 512 tiles): ONE 32-KiB LDS stage, FOUR
//       workgroups per CU hide each other's DMA latency, each walks the tiles of its XCD with the next tile's
//       first k-step in flight under the epilogue -- 845 / 555 / 700 / 950 TFLOP/s on the qkv / proj / fc1+GELU /
//       fc2 shapes at M = 24576 (profiles/round1_h_microbench.txt);
//   throughput, one tile per workgroup (ragged shapes such as the 1025-class final layer, 720 TFLOP/s);
//   latency (M <= 1536, one or two utterances): two stages, the next K-tile's DMA issued from inline asm so
//       that it stays in flight under the MFMAs (hipcc otherwise drains it before the first ds_read),
//       counted vmcnt + raw s_barrier; ~10 % shorter kernels when a CU holds a single workgroup.
// What bounds them (ablation builds, 24576 x 1536 x 512 bf16): main loop alone 36.5 us, the 75 MB of stores alone
// 13.5 us, the kernel 45 us -- loads (604 MB of L2 -> LDS, 64 flop per byte at 128 x 128) and stores queue on the
// same CU <-> L2 path, so they add; staggering the co-resident workgroups changes nothing in steady state.
// Measured and rejected in round 1 (same tests, same shapes; numbers in DESIGN.md §3): register staging
// (scratch spills, 200 TF/s), 256x128 / 256x256 tiles with 8 / 16 waves (480-530), 256x256 two-stage prefetch
// (450), K-step 32 two-stage prefetch at 4 workgroups per CU (550), 128x64 tiles at 6 per CU (550),
// K split four ways inside a 16-wave workgroup for M = 768 (slower: the fixed per-kernel cost dominates),
// an X-stationary schedule for K = 512 (X fragments resident in 128 VGPRs, W streamed through a 2 x 64 KiB ring,
// one 8-wave workgroup per CU: correct, 580 vs 845 TF/s -- prologue, DMA issue, epilogue and MFMA phases of the
// single workgroup run back to back), a persistent 256 x 256 x 64 tile with eight waves and two 64-KiB stages
// (half the L2 bytes, correct, 583 vs 845 TF/s: 2.25 tiles per CU round up to 3 and the 128-KiB store tail of a
// tile is not overlapped; with one barrier per k-step, DMA issued inside the MFMA sequence and the packed outputs
// stored under the next tile's MFMAs 707 vs 852 on qkv, 561 vs 680 on fc1+GELU), 8-row x 128-byte store regrouping
// by DPP (no change: the store path is not segment-bound), 192 x 128 tiles at three workgroups per CU in the
// persistent schedule (17 % fewer L2 bytes, bit-identical, 51.2 vs 51.6 us on qkv in an interleaved A/B: no change).
// M and N tails are handled by clamped loads and predicated stores; K must be a multiple of 64.
#include "d3pm_kernels.h"
#include "d3pm_mfma_tile.h"

namespace d3pm {
namespace {

constexpr int BM = 128, BN = 128;
constexpr int TILE_BYTES = BM * ROW_BYTES;              // 16 KiB per operand tile


// Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2).  Remap the linear block
// id so that each XCD walks a contiguous range of tiles: the n-tiles of one 128-row X panel then hit the same
// L2 instead of fetching the panel from the Infinity Cache 8 times.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}


// ---- throughput schedule: direct-to-LDS staging (global_load_lds_dwordx4) ---------------------------------
// Each wave-instruction lands 1 KiB = 8 rows x 128 B linearly in LDS (wave-uniform base + lane*16), so the
// XOR swizzle is applied to the per-lane SOURCE address (logical chunk = lane&7 ^ f(row)) and undone by the
// same lds_off() on the fragment reads.  No staging VGPRs, no ds_write pass.

template <typename T, int EPI>
__global__ __launch_bounds__(256, 4) void gemm_mfma_128_glds(const T* __restrict__ X, int ldx, const T* __restrict__ W,
                                                             const T* __restrict__ bias, T* Y, int ldy, const T* R1,
                                                             const T* R2, int ldr, const uint8_t* __restrict__ row_mask,
                                                             int mask_period, int M, int N, int K, int n_tiles, EpiFold ef) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = bid % n_tiles, tile_m = bid / n_tiles;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const T* gx[4];
  const T* gw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    const int logical = (lane & 7) ^ ((row >> 1) & 7);
    int mr = m0 + row, nr = n0 + row;
    mr = mr < M ? mr : M - 1;
    nr = nr < N ? nr : N - 1;
    gx[i] = X + static_cast<size_t>(mr) * ldx + logical * 8;
    gw[i] = W + static_cast<size_t>(nr) * K + logical * 8;
  }
  auto issue = [&](int kt, int buf) {
    char* base = smem + buf * 2 * TILE_BYTES + (wave * 4) * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((glb_void)(gx[i] + kt * BK), (lds_void)(base + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_void)(gw[i] + kt * BK), (lds_void)(base + TILE_BYTES + i * 1024), 16, 0, 0);
    }
  };
  floatx4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fch = lane >> 4;
  const int nk = K / BK;
  for (int kt = 0; kt < nk; ++kt) {
    const char* bufA = smem;
    const char* bufB = bufA + TILE_BYTES;
    issue(kt, 0);
    __syncthreads();   // drains the DMA (vmcnt(0)) and publishes the tile
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 fx[4], fw[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        fx[t] = *reinterpret_cast<const uint4*>(bufA + lds_off(wm * 64 + t * 16 + frow, ks * 4 + fch));
        fw[t] = *reinterpret_cast<const uint4*>(bufB + lds_off(wn * 64 + t * 16 + frow, ks * 4 + fch));
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = mma<T>(fw[nt], fx[mt], acc[nt][mt]);
    }
    __syncthreads();
  }
#ifndef D3PM_EXP_NOSTORE
  epilogue_store<T, EPI>(acc, bias, Y, ldy, R1, R2, ldr, row_mask, mask_period, M, N, m0 + wm * 64, n0 + wn * 64, lane, nullptr, nullptr, static_cast<const EpiPre<T, 4, 4>*>(nullptr), &ef);
#else
  {  // ablation build (tests/bench_kernels.py), never shipped: keep every accumulator live, store nothing
    float sum = 0.f;
    for (int a = 0; a < 4; ++a)
      for (int b = 0; b < 4; ++b) sum += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
    if (sum == 12345.f) Y[0] = static_cast<T>(1.f);
  }
#endif
}


// ---- latency schedule: direct-to-LDS staging issued from inline asm (glds16_asm), next tile in flight under the MFMAs ---

template <typename T, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_mfma_128_pf(const T* __restrict__ X, int ldx, const T* __restrict__ W,
                                                           const T* __restrict__ bias, T* Y, int ldy, const T* R1,
                                                           const T* R2, int ldr, const uint8_t* __restrict__ row_mask,
                                                           int mask_period, int M, int N, int K, int n_tiles, EpiFold ef) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int bid = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_n = bid % n_tiles, tile_m = bid / n_tiles;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
      (__attribute__((address_space(3))) char*)smem));

  const T* gx[4];
  const T* gw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    const int logical = (lane & 7) ^ ((row >> 1) & 7);
    int mr = m0 + row, nr = n0 + row;
    mr = mr < M ? mr : M - 1;
    nr = nr < N ? nr : N - 1;
    gx[i] = X + static_cast<size_t>(mr) * ldx + logical * 8;
    gw[i] = W + static_cast<size_t>(nr) * K + logical * 8;
  }
  auto issue = [&](int kt, int buf) {
    const uint32_t base = lds_base + buf * 2 * TILE_BYTES + (wave * 4) * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16_asm(gx[i] + kt * BK, base + i * 1024);
      glds16_asm(gw[i] + kt * BK, base + TILE_BYTES + i * 1024);
    }
  };

  floatx4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fch = lane >> 4;
  const int nk = K / BK;
  issue(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    const char* bufA = smem + (kt & 1) * 2 * TILE_BYTES;
    const char* bufB = bufA + TILE_BYTES;
    if (kt + 1 < nk) {
      issue(kt + 1, (kt + 1) & 1);                       // buffer last read in iteration kt-1 (barrier B below)
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // this wave's 8 DMAs of tile kt have landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                        // barrier A: every wave's share of tile kt is in LDS
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 fx[4], fw[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        fx[t] = *reinterpret_cast<const uint4*>(bufA + lds_off(wm * 64 + t * 16 + frow, ks * 4 + fch));
        fw[t] = *reinterpret_cast<const uint4*>(bufB + lds_off(wn * 64 + t * 16 + frow, ks * 4 + fch));
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = mma<T>(fw[nt], fx[mt], acc[nt][mt]);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // all fragment reads of tile kt retired
    __builtin_amdgcn_s_barrier();                        // barrier B: tile kt's buffer may be overwritten
  }
#ifndef D3PM_EXP_NOSTORE
  epilogue_store<T, EPI>(acc, bias, Y, ldy, R1, R2, ldr, row_mask, mask_period, M, N, m0 + wm * 64, n0 + wn * 64, lane, nullptr, nullptr, static_cast<const EpiPre<T, 4, 4>*>(nullptr), &ef);
#else
  {  // ablation build (tests/bench_kernels.py), never shipped: keep every accumulator live, store nothing
    float sum = 0.f;
    for (int a = 0; a < 4; ++a)
      for (int b = 0; b < 4; ++b) sum += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
    if (sum == 12345.f) Y[0] = static_cast<T>(1.f);
  }
#endif
}


// ---- persistent throughput schedule over whole 128 x 128 tiles: gemm_mfma_128_persist --------------------------
// Measured on the one-tile-per-workgroup kernel (24576 x 1536 x 512 bf16): main loop alone 36.5 us, the 75 MB of
// stores alone 13.5 us, together 47.9 us -- the four co-resident workgroups run in phase (same start, same tile
// time, 2.25 rounds), so every CU alternates between "all loading / MFMA" and "all storing".  Here a workgroup stays
// resident and walks the tiles of its XCD: the first k-step of the NEXT tile is put in flight before the epilogue,
// and the epilogue's stores are left in flight into the next tile's first k-step (counted vmcnt: the 8 stores of an
// interior wave tile are the youngest operations, everything older -- the DMA -- has landed).  Ragged edge tiles go
// through gemm_mfma_128_glds in a second launch, so this kernel carries no clamps and no predicated epilogue.
// DMA sources are a uniform tile base (SGPR pair) plus a tile-invariant 32-bit per-lane offset.

template <typename T, int EPI>
__global__ __launch_bounds__(256, 4) void gemm_mfma_128_persist(const T* __restrict__ X, int ldx, const T* __restrict__ W,
                                                                const T* __restrict__ bias, T* Y, int ldy, const T* R1,
                                                                const T* R2, int ldr, const uint8_t* __restrict__ row_mask,
                                                                int mask_period, int M, int N, int K, int n_tiles,
                                                                EpiFold ef, int tiles_total) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // XCD x = blockIdx & 7 owns a contiguous range of tiles (the n-tiles of one X panel then share an L2)
  const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
  const int tq = tiles_total >> 3, tr = tiles_total & 7;
  const int lo = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq, cnt = tq + (xcd < tr ? 1 : 0);
  int t = blockIdx.x >> 3;
  if (t >= cnt) return;                                            // block-uniform
  const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem)) +
                            wave * 4096;
  // per-lane DMA offsets: DMA i of a wave covers rows (wave * 4 + i) * 8 .. + 7; the swizzle key (row >> 1) & 7 only
  // depends on i & 1, so two offsets per operand serve every i and the 16-row steps ride on the uniform base pointer
  uint32_t ox[2], ow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    const int logical = (lane & 7) ^ ((row >> 1) & 7);
    ox[i] = static_cast<uint32_t>(row * ldx + logical * 8) * 2u;
    ow[i] = static_cast<uint32_t>(row * K + logical * 8) * 2u;
  }
  auto issue = [&](const T* px, const T* pw) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16_asm_s(px + static_cast<size_t>((i >> 1) * 16) * ldx, ox[i & 1], lds_base + i * 1024);
      glds16_asm_s(pw + static_cast<size_t>((i >> 1) * 16) * K, ow[i & 1], lds_base + TILE_BYTES + i * 1024);
    }
  };
  const int frow = lane & 15, fch = lane >> 4;
  const int nk = K / BK;
  const char* bufA = smem;
  const char* bufB = bufA + TILE_BYTES;
  int tile = lo + t;
  const T* sx = X + static_cast<size_t>((tile / n_tiles) * BM) * ldx;
  const T* sw = W + static_cast<size_t>((tile % n_tiles) * BN) * K;
  issue(sx, sw);
  bool first = true;
  for (;;) {
    floatx4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk; ++kt) {
      if (kt > 0) issue(sx + kt * BK, sw + kt * BK);
      if (kt == 0 && !first) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // previous tile's 8 stores stay in flight
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                      // every wave's share of the k-step is in LDS
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        uint4 fx[4], fw[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          fx[q] = *reinterpret_cast<const uint4*>(bufA + lds_off(wm * 64 + q * 16 + frow, ks * 4 + fch));
          fw[q] = *reinterpret_cast<const uint4*>(bufB + lds_off(wn * 64 + q * 16 + frow, ks * 4 + fch));
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) acc[nt][mt] = mma<T>(fw[nt], fx[mt], acc[nt][mt]);
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // all fragment reads retired
      __builtin_amdgcn_s_barrier();                       // the stage may be overwritten
    }
    const int m0 = (tile / n_tiles) * BM, n0 = (tile % n_tiles) * BN;
    t += per_xcd;
    const bool more = t < cnt;                            // block-uniform
    if (more) {
      tile = lo + t;
      sx = X + static_cast<size_t>((tile / n_tiles) * BM) * ldx;
      sw = W + static_cast<size_t>((tile % n_tiles) * BN) * K;
      issue(sx, sw);
    }
    epilogue_store<T, EPI, 4, 4, true>(acc, bias, Y, ldy, R1, R2, ldr, row_mask, mask_period, M, N, m0 + wm * 64,
                                       n0 + wn * 64, lane, nullptr, nullptr, static_cast<const EpiPre<T, 4, 4>*>(nullptr), &ef);
    if (!more) break;
    first = false;
  }
}


inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

#ifdef D3PM_ABLATIONS
__device__ uint16_t g_gelu_bf16[GELU_TAB_ENTRIES];
__global__ void fill_gelu_table() {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= GELU_TAB_ENTRIES) return;
  const uint32_t sign = i / (GELU_TAB_NE * 128), rem = i % (GELU_TAB_NE * 128);
  const uint32_t bits = (sign << 15) | ((rem / 128 + GELU_TAB_E0) << 7) | (rem % 128);
  const float v = __uint_as_float(bits << 16);
  g_gelu_bf16[i] = static_cast<uint16_t>(__float_as_uint(rn<bf16>(gelu_erf(v))) >> 16);
}
#endif

}  // namespace

#ifdef D3PM_ABLATIONS
// Device address of the bf16 GELU table (d3pm_mfma_tile.h), filled on first use on `s` (a static of the library: no
// allocation; the fill kernel is idempotent, so a capture that happens to contain it replays harmlessly).
const uint16_t* gelu_table_device(hipStream_t s) {
  static const uint16_t* ptr = nullptr;
  if (!ptr) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_gelu_bf16)) != hipSuccess) return nullptr;
    fill_gelu_table<<<(GELU_TAB_ENTRIES + 255) / 256, 256, 0, s>>>();
    if (hipGetLastError() != hipSuccess) return nullptr;
    ptr = static_cast<const uint16_t*>(p);
  }
  return ptr;
}
#endif

bool mfma_linear_supported(int dtype, const LinearArgs& a) {
  if (dtype != D3PM_F16 && dtype != D3PM_BF16) return false;
  if (a.M < 1 || a.N < 1 || a.K < BK || a.K % BK != 0) return false;
  if (a.ldx % 8 != 0 || a.ldy % 8 != 0 || !aligned(a.X, 16) || !aligned(a.W, 16) || !aligned(a.Y, 16)) return false;
  if (a.R1 && (a.ldr % 8 != 0 || a.N % 8 != 0 || !aligned(a.R1, 16))) return false;
  if (a.R2 && !aligned(a.R2, 16)) return false;
  if (static_cast<long long>(a.M) * a.N < 128 * 128) return false;     // not worth a 128^2 tile
  const bool gelu = a.act == ACT_GELU, r1 = a.R1 != nullptr, r2 = a.R2 != nullptr, mk = a.row_mask != nullptr;
  if (!fold_args_ok(a)) return false;
  if (a.act == ACT_RELU || a.act == ACT_SILU) return !r1 && !mk && !a.fold_s && !a.stats_out;   // condition-encoder FFN epilogues
  if (r2 && !r1) return false;
  // instantiated epilogues: plain, GELU, R1, R1+R2, R1+mask
  if (gelu && (r1 || mk)) return false;
  if (mk && (!r1 || r2)) return false;
  return true;
}

// d3pm_tuning.gemm_variant: 0 auto, 2 throughput (128 x 128 persistent over whole tiles), 3 the round-1 latency schedule
// (128 x 128, two stages), 4 the latency schedule (64 x 64 tiles, whole-K panels), 5 throughput with one 128 x 128 tile per
// workgroup, 6 / 7 / 8 big tiles (192 x 256 / 96 x 512 / 192 x 128, d3pm_mfma_gemm_big.hip) wherever they apply, else as auto
// without big tiles.  gemm_persist_slots: resident workgroups of the persistent schedule (1024 = 4 per CU x 256 CUs).
bool panel64_linear_supported(int dtype, const LinearArgs& a);
int panel64_linear(int dtype, const LinearArgs& a, hipStream_t s, const LnPrologue* ln = nullptr);
int big_linear_tile(int dtype, const LinearArgs& a, int want);
int big_linear(int dtype, const LinearArgs& a, int id, hipStream_t s);

#ifdef D3PM_ABLATIONS
bool ring_linear_supported(int dtype, const LinearArgs& a);
int ring_linear(int dtype, const LinearArgs& a, hipStream_t s);
// The LayerNorm-prologue form exists for the latency schedule only: true where mfma_linear would pick that schedule anyway
bool ln_prologue_linear_applies(int dtype, const LinearArgs& a, const LnPrologue& ln) {
  const int variant = tune_of(a.tune).gemm_variant;
  const bool autosel = variant == 0;
  if (!(variant == 4 || (autosel && a.M <= 1536))) return false;
  if (autosel && big_linear_tile(dtype, a, 0)) return false;
  return panel64_ln_supported(dtype, a, ln);
}
int ln_prologue_linear(int dtype, const LinearArgs& a, const LnPrologue& ln, hipStream_t s) { return panel64_linear(dtype, a, s, &ln); }
#endif

int mfma_linear(int dtype, const LinearArgs& a, hipStream_t s) {
  // All schedules accumulate in the same order, so the choice never changes a bit of the result.
  const bool ffn_act = a.act == ACT_RELU || a.act == ACT_SILU;
  const d3pm_tuning& tn = tune_of(a.tune);
  const int g_gemm_variant = tn.gemm_variant, g_persist_slots = tn.gemm_persist_slots >= 8 ? (tn.gemm_persist_slots & ~7) : 1024;
#ifdef D3PM_ABLATIONS
  if (ab_knobs().ring && ring_linear_supported(dtype, a)) return ring_linear(dtype, a, s);   // experimental ring schedule
#endif
  const bool autosel = g_gemm_variant == 0 || (g_gemm_variant >= 6 && g_gemm_variant <= 8);
  if (autosel) {
    const int id = big_linear_tile(dtype, a, g_gemm_variant == 6 ? 2 : g_gemm_variant == 7 ? 1 : g_gemm_variant == 8 ? 3 : 0);
    if (id) return big_linear(dtype, a, id, s);
  }
  // one or two utterances: 64 x 64 tiles with whole-K panels in flight (d3pm_mfma_gemm_lat.hip); 4 forces it for any M
  if ((g_gemm_variant == 4 || (autosel && a.M <= 1536)) && panel64_linear_supported(dtype, a)) return panel64_linear(dtype, a, s);
  const bool latency = !ffn_act && (g_gemm_variant == 3 || (autosel && a.M <= 1536));
  const int n_tiles = (a.N + BN - 1) / BN, m_tiles = (a.M + BM - 1) / BM;
  // shapes made of whole tiles go through the persistent kernel once there are enough tiles to fill the chip twice
  // over (a ragged edge would need a second launch that costs more than persistence gains: measured on N = 1025)
  const bool persist = !latency && !ffn_act && (autosel || g_gemm_variant == 2) && a.M % BM == 0 &&
                       a.N % BN == 0 && static_cast<long long>(n_tiles) * m_tiles >= 2 * 256 &&
                       static_cast<long long>(BM) * a.ldx * 2 < (1ll << 31) && static_cast<long long>(BN) * a.K * 2 < (1ll << 31);
  const size_t lds = (latency ? 4 : 2) * TILE_BYTES;   // 64 KiB: two workgroups per CU; 32 KiB: four
  const int tiles_total = n_tiles * m_tiles, want = (tiles_total + 7) & ~7;
  const dim3 grid(static_cast<unsigned>(persist ? (want < g_persist_slots ? want : g_persist_slots) : tiles_total)), block(256);
  const int epi = (a.act == ACT_GELU ? EPI_GELU : a.act == ACT_RELU ? EPI_RELU : a.act == ACT_SILU ? EPI_SILU : 0) |
                  (a.R1 ? (a.R2 ? EPI_R2 : EPI_R1) : 0) | (a.row_mask ? EPI_MASK : 0) | (a.fold_s ? EPI_LNF : 0) | (a.stats_out ? EPI_STATS : 0);
  const EpiFold ef = epi_fold_of(a);

#define D3PM_GEMM(KERNEL, ...)                                                                                  \
  do {                                                                                                          \
    D3PM_LDS_ATTR((&KERNEL), 4 * TILE_BYTES);                                                                   \
    KERNEL<<<grid, block, lds, s>>>(static_cast<const U*>(a.X), a.ldx, static_cast<const U*>(a.W),             \
                                    static_cast<const U*>(a.bias), static_cast<U*>(a.Y), a.ldy,                 \
                                    static_cast<const U*>(a.R1), static_cast<const U*>(a.R2), a.ldr, a.row_mask, \
                                    a.mask_period, a.M, a.N, a.K, n_tiles, ef, ##__VA_ARGS__);                  \
    return D3PM_OK;                                                                                             \
  } while (0)
#define D3PM_GEMM_EPI(E)                                                \
  do {                                                                  \
    if (latency) D3PM_GEMM((gemm_mfma_128_pf<U, E>));                   \
    else if (persist) D3PM_GEMM((gemm_mfma_128_persist<U, E>), tiles_total); \
    else D3PM_GEMM((gemm_mfma_128_glds<U, E>));                         \
  } while (0)
  auto go = [&](auto* tag) -> int {
    using U = std::remove_pointer_t<decltype(tag)>;
    switch (epi) {   // the epilogues the denoiser and the condition encoders use
      case 0: D3PM_GEMM_EPI(0);
      case EPI_GELU: D3PM_GEMM_EPI(EPI_GELU);
      case EPI_R1: D3PM_GEMM_EPI(EPI_R1);
      case EPI_R2: D3PM_GEMM_EPI(EPI_R2);
      case EPI_R1 | EPI_MASK: D3PM_GEMM_EPI(EPI_R1 | EPI_MASK);
      case EPI_LNF: D3PM_GEMM_EPI(EPI_LNF);                       // LayerNorm folded into the projection (d3pm_mfma_tile.h)
      case EPI_LNF | EPI_GELU: D3PM_GEMM_EPI(EPI_LNF | EPI_GELU);
      case EPI_R1 | EPI_STATS: D3PM_GEMM_EPI(EPI_R1 | EPI_STATS);   // residual rows + their moments for the next folded LayerNorm
      case EPI_R2 | EPI_STATS: D3PM_GEMM_EPI(EPI_R2 | EPI_STATS);
      case EPI_R1 | EPI_MASK | EPI_STATS: D3PM_GEMM_EPI(EPI_R1 | EPI_MASK | EPI_STATS);
      case EPI_RELU: D3PM_GEMM((gemm_mfma_128_glds<U, EPI_RELU>));
      case EPI_SILU: D3PM_GEMM((gemm_mfma_128_glds<U, EPI_SILU>));
      default: break;
    }
    return D3PM_E_SHAPE;
  };
#undef D3PM_GEMM_EPI
  int rc = dtype == D3PM_F16 ? go(static_cast<f16*>(nullptr)) : go(static_cast<bf16*>(nullptr));
  if (rc != D3PM_OK) return rc;
#undef D3PM_GEMM
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

}  // namespace d3pm
