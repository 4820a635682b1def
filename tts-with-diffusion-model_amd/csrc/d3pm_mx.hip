// d3pm_mx.hip -- the fp8 fast path of BASELINE.json configs[4] on the block-scaled matrix instruction of gfx950:
//
//     v_mfma_scale_f32_16x16x128_f8f6f4   D[16][16] += sum_k A[i][k] 2^(sa - 127) . B[k][j] 2^(sb - 127),  K = 128
//
// with OCP e4m3 operands and one e8m0 scale (a power of two) per 32 contraction elements ("MX" blocks).  It retires 128 k per
// instruction at twice the cycles of the 16-bit 16x16x32 form, i.e. twice the 16-bit MFMA rate, AND its operands are half
// the bytes -- the 16-bit projections of this model are bound by the L2 -> LDS operand stream, not by the matrix pipe
// (DESIGN.md section 3).  No reference counterpart exists (the reference is fp16 only): the kernel tests pin these kernels to
// an fp32 evaluation of the SAME codes and scales (products of two e4m3 numbers and power-of-two scales are exact in fp32),
// and the model-level tests report agreement with the 16-bit path.
//
// What the instruction does with its operands was measured, not assumed (tools/probe_mx.hip,
// profiles/round3_a_probe_mx_scaled_mfma.txt): lane (r = lane & 15, g = lane >> 4) supplies row r (A) / column r (B); its 32
// operand bytes are TWO 16-byte halves that belong to different 32-element blocks; the block scale is taken from byte
// `op_sel` of the scale register of lane r + 16 q, where q is the block: half 0 of lanes g = 0, 1 -> q 0, half 0 of g = 2, 3
// -> q 1, half 1 of g = 0, 1 -> q 2, half 1 of g = 2, 3 -> q 3.  A and B use the same map, so any permutation of k that is
// applied to both operands AND keeps blocks together is free.  Here half h of lane group g is the 16-byte chunk 4 h + g of a
// 128-byte LDS row (the read pattern of the 16-bit kernels: conflict-free with the (row >> 1) & 7 swizzle), so block q is
// simply elements [32 q, 32 q + 32) of the 128-element k-step: natural MX blocks.
//
// Formats (all row-major, one byte per element):
//   codes   X8 [M][K]   e4m3
//   scales  SX [M][4][K / 128]   e8m0 (value 2^(byte - 127)):  SX[m][g][s] scales elements [128 s + 32 g, 128 s + 32 g + 32) of
//           row m -- stored g-major so that the lane (r, g) of an MFMA fetches the scales of FOUR consecutive k-steps of its
//           (row, block) as one dword and selects among them with op_sel; no shuffles, no unpacking
//   weights likewise: W8 [N][K], SW [N][4][K / 128]  (quantised once on the host: _hip.quantize_mx)
// Scale rule (both sides): amax of the block = 1.m x 2^E  ->  scale 2^(E - 8) when 1.m <= 1.75, else 2^(E - 7): the smallest
// power of two with amax / scale <= 448 (the largest e4m3 value); exact integer arithmetic on the fp32 bits, codes =
// rn_e4m3(x / scale) never saturate.
#include "d3pm_kernels.h"
#include "d3pm_mfma_tile.h"
#include "d3pm_mx.h"

namespace d3pm {
namespace {

typedef int intx8 __attribute__((ext_vector_type(8)));

template <typename T> struct alignas(16) Vec8 { T v[8]; };

// lane owns 8 consecutive elements of a 512-wide row (chunk `lane`): blocks of 32 = 4 consecutive lanes.  Writes the 8 codes and
// the row's 16 scale bytes (K = 512: SX[m][g][s], byte g * 4 + s <-> block 4 s + g)
__device__ __forceinline__ void store_mx_row512(const float (&o)[8], uint8_t* dst8, uint8_t* sx_row, int lane) {
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(o[i]));
  amax = fmaxf(amax, __shfl_xor(amax, 1, kWave));
  amax = fmaxf(amax, __shfl_xor(amax, 2, kWave));
  const uint32_t sb = mx_scale_byte(amax);
  *reinterpret_cast<uint2*>(dst8) = mx_pack8(o, mx_inv_scale(sb));
  // byte j = g * 4 + s of the row's scales comes from block 4 s + g = lanes 4 (4 s + g) ..: lane j fetches it
  const int j = lane & 15, src = 4 * (4 * (j & 3) + (j >> 2));
  const uint32_t mine = __shfl(sb, src, kWave);
  if (lane < 16) sx_row[lane] = static_cast<uint8_t>(mine);
}

// ---- LayerNorm (+ FiLM) -> MX row: layernorm_vec's arithmetic up to the 16-bit result, then blocks of 32 ------------------
// optional second LayerNorm of the same rows (w2, b2 -> y8_2, sx_2), as layernorm_vec's dual output (norm2 | norm22)
template <typename T>
__global__ __launch_bounds__(256) void layernorm_mx_rows(const T* __restrict__ x, uint8_t* __restrict__ y8, uint8_t* __restrict__ sx,
                                                         const T* __restrict__ w, const T* __restrict__ b,
                                                         const T* __restrict__ film, const T* __restrict__ w2,
                                                         const T* __restrict__ b2, uint8_t* __restrict__ y8_2,
                                                         uint8_t* __restrict__ sx_2, int M, float eps) {
  constexpr int d = 512;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int bidx;      // XCD-aware order, as layernorm_vec: each XCD takes a contiguous range of rows
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    bidx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int row = bidx * 4 + wave;
  if (row >= M) return;
  const Vec8<T> raw = *reinterpret_cast<const Vec8<T>*>(x + static_cast<size_t>(row) * d + lane * 8);
  float v[8], s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { v[i] = static_cast<float>(raw.v[i]); s += v[i]; }
  const float mean = wave_sum_up(s) / static_cast<float>(d);
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { const float t = v[i] - mean; q += t * t; }
  const float rstd = rsqrtf(wave_sum_up(q) / static_cast<float>(d) + eps);
  const int col = lane * 8;
  const Vec8<T> wv = *reinterpret_cast<const Vec8<T>*>(w + col), bv = *reinterpret_cast<const Vec8<T>*>(b + col);
  float o[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = rn<T>((v[i] - mean) * rstd * static_cast<float>(wv.v[i]) + static_cast<float>(bv.v[i]));
  if (film) {
    const Vec8<T> sc = *reinterpret_cast<const Vec8<T>*>(film + col), sh = *reinterpret_cast<const Vec8<T>*>(film + d + col);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float g = rn<T>(1.0f + static_cast<float>(sc.v[i]));
      o[i] = rn<T>(rn<T>(o[i] * g) + static_cast<float>(sh.v[i]));
    }
  }
  store_mx_row512(o, y8 + static_cast<size_t>(row) * d + col, sx + static_cast<size_t>(row) * 16, lane);
  if (y8_2) {
    const Vec8<T> w2v = *reinterpret_cast<const Vec8<T>*>(w2 + col), b2v = *reinterpret_cast<const Vec8<T>*>(b2 + col);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = rn<T>((v[i] - mean) * rstd * static_cast<float>(w2v.v[i]) + static_cast<float>(b2v.v[i]));
    store_mx_row512(o, y8_2 + static_cast<size_t>(row) * d + col, sx_2 + static_cast<size_t>(row) * 16, lane);
  }
}

// ---- any 16-bit [M][K] -> MX (K a multiple of 128): one wave per 512-element piece of a row (kernel tests, attention output) --
template <typename T>
__global__ __launch_bounds__(256) void quantize_mx_rows(const T* __restrict__ x, int ldx, uint8_t* __restrict__ y8, uint8_t* __restrict__ sx,
                                                        int M, int K) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int pieces = (K + 511) / 512, ks = K / 128;
  const long long id = static_cast<long long>(blockIdx.x) * 4 + wave;
  if (id >= static_cast<long long>(M) * pieces) return;
  const int row = static_cast<int>(id / pieces), piece = static_cast<int>(id % pieces), col = piece * 512 + lane * 8;
  float o[8];
  const bool live = col < K;
  if (live) {
    const Vec8<T> raw = *reinterpret_cast<const Vec8<T>*>(x + static_cast<size_t>(row) * ldx + col);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = static_cast<float>(raw.v[i]);
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = 0.f;
  }
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(o[i]));
  amax = fmaxf(amax, __shfl_xor(amax, 1, kWave));
  amax = fmaxf(amax, __shfl_xor(amax, 2, kWave));
  const uint32_t sb = mx_scale_byte(amax);
  if (live) {
    *reinterpret_cast<uint2*>(y8 + static_cast<size_t>(row) * K + col) = mx_pack8(o, mx_inv_scale(sb));
    if ((lane & 3) == 0) {
      const int blk = col >> 5, s = blk >> 2, g = blk & 3;
      sx[(static_cast<size_t>(row) * 4 + g) * ks + s] = static_cast<uint8_t>(sb);
    }
  }
}

// ---- block-scaled GEMM, big-tile persistent schedule --------------------------------------------------------------------------
//   Y[M][N] = epilogue(sum_k X8[m][k] 2^(sx - 127) . W8[n][k] 2^(sw - 127) + bias)
// The structure of gemm_mfma_big (d3pm_mfma_gemm_big.hip): one workgroup of eight waves per CU on a 192 x 256 tile (or two of
// four waves on 192 x 128), wave tile 96 x 64 = 6 x 4 MFMA tiles, two LDS stages of (TM + TN) x 128 B, the next k-step's 1-KiB DMA
// pieces issued from inline asm between the MFMAs of the current one, one s_barrier per k-step with a counted vmcnt, the
// stream of k-steps running across tile boundaries, the 16-bit kernels' epilogue (d3pm_mfma_tile.h).  A k-step is 128
// one-byte elements = the same 128-byte LDS rows, swizzle and DMA pattern; it issues 24 scaled MFMAs (the time of the 16-bit
// k-step's 48) for twice the contraction depth, so a projection needs half the k-steps, DMA pieces and barriers.
// OUT8 (fc1 -> fc2): the epilogue quantises its own output -- a lane group of four owns 32 consecutive columns of a row after
// the regrouping of epilogue_store, exactly one MX block -- and writes codes + block scales instead of 16-bit values: the
// consumer's operand format, no row-wide reduction, half the output bytes (fp32 -> e4m3 directly, tanh-form GELU: see the epilogue).
template <typename T, int EPI, int WM, int WN, bool OUT8>
__global__ __launch_bounds__(WM * WN * 64, 2) void gemm_mx_big(const uint8_t* __restrict__ X, int ldx, const uint8_t* __restrict__ SX,
                                                              const uint8_t* __restrict__ W, const uint8_t* __restrict__ SW,
                                                              const T* __restrict__ bias, T* Y, int ldy, const T* R1, int ldr,
                                                              const uint8_t* __restrict__ row_mask, int mask_period, uint8_t* Y8,
                                                              uint8_t* SY, int M, int N, int K, int n_tiles, int tiles_total) {
  constexpr int NW = WM * WN, TM = 96 * WM, TN = 64 * WN;
  constexpr int XD = TM / 8, WD = TN / 8;                  // 1-KiB DMA pieces (8 rows x 128 B) per k-step and operand
  constexpr int XPW = (XD + NW - 1) / NW, WPW = WD / NW;   // pieces per wave
  constexpr int NDMA = XPW + WPW;                          // 7 (192 x 256), 10 (192 x 128)
  static_assert(NW % 2 == 0 && WD % NW == 0 && XD % NW == 0, "piece distribution keeps the parity of the wave");
  static_assert(NDMA <= 12, "one DMA piece per pair of MFMAs");
  constexpr int X_BYTES = TM * ROW_BYTES, STAGE = (TM + TN) * ROW_BYTES;
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
  // DMA piece j of an operand tile = rows 8j .. 8j+7; a wave takes pieces j = wave + NW p, so the swizzle key
  // (row >> 1) & 7 = (4 (j & 1) + (lane >> 4)) & 7 is the same for all of its pieces: one per-lane offset per operand
  const int frow_ = lane & 15, fch_ = lane >> 4;
  const int lrow = lane >> 3, logical = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
  const uint32_t ox = static_cast<uint32_t>(lrow * ldx + logical * 16);
  const uint32_t ow = static_cast<uint32_t>(lrow * K + logical * 16);
  auto dma = [&](int p, const uint8_t* px, const uint8_t* pw, uint32_t stage) __attribute__((always_inline)) {   // p: unrolled constant
    if (p < XPW) {
      const int j = wave + NW * p;
      glds16_asm_s(px + static_cast<size_t>(8 * j) * ldx, ox, stage + j * 1024);
    } else {
      const int j = wave + NW * (p - XPW);
      glds16_asm_s(pw + static_cast<size_t>(8 * j) * K, ow, stage + X_BYTES + j * 1024);
    }
  };
  // block scales of the lane's fragment rows: dword c of the (row, g) run = k-steps 4 c .. 4 c + 3.  Issued from asm like the DMA
  // pieces (a compiler-visible load would make hipcc place its own vmcnt waits, which cannot see the pieces in flight), right
  // behind the last MFMA that reads the previous chunk's dwords; valid after the next step-top wait (they are older than
  // everything that wait leaves in flight).  One scalar base per operand + a per-lane offset per 16-row block.
  const uint32_t osc = static_cast<uint32_t>((frow_ * 4 + fch_) * (K / 128)), srow = static_cast<uint32_t>(64 * (K / 128));
  auto load_scales = [&](int (&dw)[4], int (&dx)[6], const uint8_t* wbase, const uint8_t* xbase) __attribute__((always_inline)) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) asm volatile("global_load_dword %0, %1, %2" : "=v"(dw[nt]) : "v"(osc + nt * srow), "s"(wbase) : "memory");
#pragma unroll
    for (int mt = 0; mt < 6; ++mt) asm volatile("global_load_dword %0, %1, %2" : "=v"(dx[mt]) : "v"(osc + mt * srow), "s"(xbase) : "memory");
  };
  // fragment addresses: row = base + 16 q + (lane & 15); the swizzle key (row >> 1) & 7 = (lane & 15) >> 1 because every base
  // is a multiple of 16.  Half 0 of lane group g = chunk g, half 1 = chunk 4 + g (header: natural MX blocks)
  const int frow = lane & 15, fch = lane >> 4, fkey = (frow >> 1) & 7;
  const int fo0 = frow * ROW_BYTES + ((fch ^ fkey) << 4), fo1 = frow * ROW_BYTES + (((4 + fch) ^ fkey) << 4);
  const char* const fx_base = smem + wm * 96 * ROW_BYTES;
  const char* const fw_base = smem + X_BYTES + wn * 64 * ROW_BYTES;
  const int nk = K / 128, nchunk = nk / 4;                 // k-steps; a scale dword covers four of them
  const size_t ks_bytes = static_cast<size_t>(nk);         // bytes of one (row, g) scale run
  int tile = lo + t;
  const uint8_t* sx = X + static_cast<size_t>((tile / n_tiles) * TM) * ldx;
  const uint8_t* sw = W + static_cast<size_t>((tile % n_tiles) * TN) * K;
  int scw[4], scx[6];                                                 // scale dwords of the current chunk (four k-steps)
  load_scales(scw, scx, SW + static_cast<size_t>((tile % n_tiles) * TN + wn * 64) * 4 * ks_bytes,
              SX + static_cast<size_t>((tile / n_tiles) * TM + wm * 96) * 4 * ks_bytes);
#pragma unroll
  for (int p = 0; p < NDMA; ++p) dma(p, sx, sw, lds_base);          // first k-step of the first tile
  // the first step of a tile waits with vmcnt(NST): behind an epilogue the DMA pieces are older than its NST stores; the very
  // first tile has no stores behind its pieces, so they are waited for here
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (;;) {
    floatx4 acc[4][6];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 6; ++b) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int m0 = (tile / n_tiles) * TM, n0 = (tile % n_tiles) * TN;
    const int t_next = t + per_xcd;
    const bool more = t_next < cnt;
    const int tile_next = more ? lo + t_next : tile;
    const uint8_t* sx_next = X + static_cast<size_t>((tile_next / n_tiles) * TM) * ldx;
    const uint8_t* sw_next = W + static_cast<size_t>((tile_next % n_tiles) * TN) * K;
    // scale runs of this wave's rows (wave-uniform bases; the lane's (row, g) offset is `osc`)
    const uint8_t* swb = SW + static_cast<size_t>(n0 + wn * 64) * 4 * ks_bytes;
    const uint8_t* sxb = SX + static_cast<size_t>(m0 + wm * 96) * 4 * ks_bytes;
    const uint8_t* swb_next = SW + static_cast<size_t>((tile_next % n_tiles) * TN + wn * 64) * 4 * ks_bytes;
    const uint8_t* sxb_next = SX + static_cast<size_t>((tile_next / n_tiles) * TM + wm * 96) * 4 * ks_bytes;

    // one k-step on stage S with scale byte OPS of the chunk's scale dwords, while the DMA pieces of the following k-step
    // (px, pw) go to stage S ^ 1.  WAITN: what the wait at the top may leave in flight (the stores behind the previous tile)
    // (nwb, nxb): with OPS == 3 (last step of a chunk) the scale runs whose dword the following chunk needs
    auto step = [&](auto S_, auto OPS_, auto WAITN_, const uint8_t* px, const uint8_t* pw, const uint8_t* nwb, const uint8_t* nxb) __attribute__((always_inline)) {
      constexpr int S = decltype(S_)::value, OPS = decltype(OPS_)::value, WAITN = decltype(WAITN_)::value;
      const char* bx = fx_base + S * STAGE;
      const char* bw = fw_base + S * STAGE;
      const uint32_t nxt = lds_base + (S ^ 1) * STAGE;
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAITN) : "memory");      // (also: the scale dwords loaded behind the previous step)
      __builtin_amdgcn_s_barrier();      // every wave's pieces of this k-step have landed; stage S ^ 1 is no longer read
      __builtin_amdgcn_sched_barrier(0);
      uintx4 fw[4][2], fx[2][2];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        fw[nt][0] = *reinterpret_cast<const uintx4*>(bw + nt * 16 * ROW_BYTES + fo0);
        fw[nt][1] = *reinterpret_cast<const uintx4*>(bw + nt * 16 * ROW_BYTES + fo1);
      }
      fx[0][0] = *reinterpret_cast<const uintx4*>(bx + fo0);
      fx[0][1] = *reinterpret_cast<const uintx4*>(bx + fo1);
#pragma unroll
      for (int g = 0; g < 12; ++g) {       // group g: row block g / 2, column-block pair g % 2: two MFMAs
        const int mt = g >> 1, np = g & 1;
        if (np == 0 && mt + 1 < 6) {
          fx[(mt + 1) & 1][0] = *reinterpret_cast<const uintx4*>(bx + (mt + 1) * 16 * ROW_BYTES + fo0);
          fx[(mt + 1) & 1][1] = *reinterpret_cast<const uintx4*>(bx + (mt + 1) * 16 * ROW_BYTES + fo1);
        }
        if (g < NDMA) dma(g, px, pw, nxt);
        const uintx4 xl = fx[mt & 1][0], xh = fx[mt & 1][1];
        const intx8 xf = {static_cast<int>(xl[0]), static_cast<int>(xl[1]), static_cast<int>(xl[2]), static_cast<int>(xl[3]),
                          static_cast<int>(xh[0]), static_cast<int>(xh[1]), static_cast<int>(xh[2]), static_cast<int>(xh[3])};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int nt = 2 * np + j;
          const uintx4 wl = fw[nt][0], wh = fw[nt][1];
          const intx8 wf = {static_cast<int>(wl[0]), static_cast<int>(wl[1]), static_cast<int>(wl[2]), static_cast<int>(wl[3]),
                            static_cast<int>(wh[0]), static_cast<int>(wh[1]), static_cast<int>(wh[2]), static_cast<int>(wh[3])};
          acc[nt][mt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf, xf, acc[nt][mt], 0, 0, OPS, scw[nt], OPS, scx[mt]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (OPS == 3) {
        load_scales(scw, scx, nwb, nxb);        // nothing reads the old dwords any more
        __builtin_amdgcn_sched_barrier(0);
      }
    };

    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using INST = std::integral_constant<int, 12>;       // stores of an epilogue per wave (MX output: 6 code + 6 scale stores)
    for (int c = 0; c < nchunk; ++c) {
      const int k0 = c * 512;
      const bool last = c + 1 == nchunk;
      const uint8_t* nwb = last ? swb_next : swb + (c + 1) * 4;
      const uint8_t* nxb = last ? sxb_next : sxb + (c + 1) * 4;
      if (c == 0) step(I0{}, I0{}, INST{}, sx + k0 + 128, sw + k0 + 128, nwb, nxb);
      else step(I0{}, I0{}, I0{}, sx + k0 + 128, sw + k0 + 128, nwb, nxb);
      step(I1{}, I1{}, I0{}, sx + k0 + 256, sw + k0 + 256, nwb, nxb);
      step(I0{}, I2{}, I0{}, sx + k0 + 384, sw + k0 + 384, nwb, nxb);
      step(I1{}, I3{}, I0{}, last ? sx_next : sx + k0 + 512, last ? sw_next : sw + k0 + 512, nwb, nxb);
    }
    if constexpr (OUT8) {
      // epilogue with an MX output (the hidden layer between fc1 and fc2): bias + activation in fp32, then per (row, 32 columns)
      // block: absmax over the four lanes that own it, scale byte, eight codes per lane.  The values go from fp32 straight to e4m3
      // -- no 16-bit rounding in between, it would be noise below the 3 mantissa bits that survive -- and GELU takes its tanh form
      //     gelu(v) ~ v / (1 + 2^(-2 log2(e) sqrt(2/pi) (v + 0.044715 v^3)))        (|error| < 5e-4 absolute, e4m3 keeps 6e-2 relative)
      // two transcendentals and four FMAs instead of the 16-instruction exact-erf chain of the 16-bit epilogue: this epilogue is
      // what the launch waits for once the matrix work takes half the time (profiles/round3_d_ab_mx_fp8_gemm.txt).
      // Column pair np of this wave = block (wn * 2 + np) % 4 of k-step (n0 + wn * 64) / 128 of the consumer's K = N.
      const int g = lane >> 4, nks = N / 128, col0 = n0 + wn * 64;
      float bv[4][4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        Pack4<T> pb{};
        if (bias) pb = *reinterpret_cast<const Pack4<T>*>(bias + col0 + nt * 16 + g * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) bv[nt][r] = bias ? static_cast<float>(pb.v[r]) : 0.f;
      }
#pragma unroll
      for (int mt = 0; mt < 6; ++mt) {
        uint2 cd[2];
        uint32_t sb[2];
#pragma unroll
        for (int np = 0; np < 2; ++np) {
          float v[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            float x = acc[2 * np + (r >> 2)][mt][r & 3] + bv[2 * np + (r >> 2)][r & 3];
            if constexpr ((EPI & EPI_GELU) != 0) {
              const float u = x * __builtin_fmaf(x * x, -2.302208198f * 0.044715f, -2.302208198f);      // -2 log2(e) sqrt(2/pi) (x + 0.044715 x^3)
              x = x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u));
            }
            v[r] = x;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) swap_rows16(v[r], v[4 + r]);      // lane: 8 consecutive columns at 32 np + nq (d3pm_mfma_tile.h)
          cd[np] = mx_block_quantise(v, sb[np]);
        }
        const int row = m0 + wm * 96 + mt * 16 + (lane & 15);
        mx_store_row64(cd[0], cd[1], Y8 + static_cast<size_t>(row) * N + col0, lane);
        // the row's two block scales: lanes g = 0 / 1 write the byte of half 0 / 1 (one store instruction)
        if (lane < 32) {
          const int blk = (col0 >> 5) + (lane >> 4);
          SY[(static_cast<size_t>(row) * 4 + (blk & 3)) * nks + (blk >> 2)] = static_cast<uint8_t>(lane < 16 ? sb[0] : sb[1]);
        }
      }
    } else {
      // `sc1` output stores, as in gemm_mfma_big: the rows are not read again by this launch and must not displace its operands in L2
      epilogue_store<T, EPI, 4, 6, true, false, true>(acc, bias, Y, ldy, R1, nullptr, ldr, row_mask, mask_period, M, N, m0 + wm * 96,
                                                      n0 + wn * 64, lane);
    }
    if (!more) break;
    t = t_next;
    tile = tile_next;
    sx = sx_next;
    sw = sw_next;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the unused look-ahead pieces must not outlive the workgroup's LDS
}

inline bool aligned16m(const void* p) { return (reinterpret_cast<uintptr_t>(p) % 16) == 0; }

}  // namespace

// geometry: 2 = 192 x 256 (eight waves, one workgroup per CU), 3 = 192 x 128 (four waves, two per CU)
static int mx_geometry(const MxLinearArgs& a) {
  const int want = tune_of(a.tune).gemm_variant;          // 6 / 8: only the 192 x 256 / 192 x 128 tile (as for the 16-bit big tiles)
  if (want == 6) return (a.M % 192 == 0 && a.N % 256 == 0) ? 2 : 0;
  if (want == 8) return (a.M % 192 == 0 && a.N % 128 == 0) ? 3 : 0;
  auto fits = [&](int tm, int tn, int slots) {
    if (a.M % tm != 0 || a.N % tn != 0) return false;
    const long long tiles = static_cast<long long>(a.M / tm) * (a.N / tn), rounds = (tiles + slots - 1) / slots;
    return tiles * 100 >= rounds * slots * 85 || tiles >= 4 * slots;      // >= 85 % of the slot-rounds do work
  };
  // two 4-wave workgroups per CU first, as for the 16-bit big tiles (d3pm_mfma_gemm_big.hip, big_linear_tile): in the sampler's loop
  // the operands arrive cold and a second workgroup hides that (fp8 50-step path 224.0 k -> 227.3 k tokens/s,
  // profiles/round3_w_fp8_geom.txt); under the GELU epilogue its VALU work also overlaps with the other workgroup's MFMAs
  if (fits(192, 128, 512)) return 3;
  if (fits(192, 256, 256)) return 2;
  if (a.M % 192 == 0 && a.N % 256 == 0) return 2;
  if (a.M % 192 == 0 && a.N % 128 == 0) return 3;
  return 0;
}

bool mx_linear_supported(int dtype, const MxLinearArgs& a) {
  if (dtype != D3PM_F16 && dtype != D3PM_BF16) return false;
  if (a.M < 192 || a.N < 128 || a.K < 512 || a.K % 512 != 0 || a.ldx % 16 != 0) return false;
  if (!a.X8 || !a.SX || !a.W8 || !a.SW || !aligned16m(a.X8) || !aligned16m(a.W8)) return false;
  if (static_cast<long long>(a.ldx) * 8 * 64 >= (1ll << 31) || static_cast<long long>(a.K) * 8 * 64 >= (1ll << 31)) return false;
  if (a.Y8) {                                     // MX output: plain or GELU epilogue, no residual
    if (!a.SY || a.R1 || a.row_mask || !aligned16m(a.Y8) || a.N % 128 != 0) return false;
  } else {
    if (!a.Y || a.ldy % 8 != 0 || !aligned16m(a.Y)) return false;
    if (a.R1 && (a.ldr % 8 != 0 || !aligned16m(a.R1))) return false;
    if (a.row_mask && !a.R1) return false;
    if (a.act == ACT_GELU && (a.R1 || a.row_mask)) return false;
  }
  if (a.act != ACT_NONE && a.act != ACT_GELU) return false;
  // the kernel is persistent and re-reads the block scales of its operands for every tile: nothing it writes may share bytes with
  // them (or with the operand codes)
  auto overlaps = [](const void* p, size_t np, const void* q, size_t nq) {
    const uintptr_t a0 = reinterpret_cast<uintptr_t>(p), b0 = reinterpret_cast<uintptr_t>(q);
    return p && q && a0 < b0 + nq && b0 < a0 + np;
  };
  const size_t M = a.M, N = a.N, K = a.K, es = 2;
  const struct { const void* p; size_t n; } ins[] = {{a.SX, M * (K / 32)}, {a.SW, N * (K / 32)}, {a.X8, (M - 1) * a.ldx + K}, {a.W8, N * K}};
  const struct { const void* p; size_t n; } outs[] = {{a.Y8 ? nullptr : a.Y, ((M - 1) * a.ldy + N) * es}, {a.Y8, M * N}, {a.SY, M * (N / 32)}};
  for (const auto& in : ins)
    for (const auto& out : outs)
      if (overlaps(in.p, in.n, out.p, out.n)) return false;
  return mx_geometry(a) != 0;
}

template <typename U, int E, int WM, int WN, bool OUT8>
static int mx_launch(const MxLinearArgs& a, hipStream_t s) {
  constexpr int TM = 96 * WM, TN = 64 * WN, NWAVE = WM * WN;
  const int n_tiles = a.N / TN, tiles_total = (a.M / TM) * n_tiles;
  const int slots = 256 * (8 / NWAVE), want = (tiles_total + 7) & ~7;
  const dim3 grid(static_cast<unsigned>(want < slots ? want : slots));
  const size_t lds = 2 * static_cast<size_t>(TM + TN) * ROW_BYTES;
  D3PM_LDS_ATTR((&gemm_mx_big<U, E, WM, WN, OUT8>), 160 * 1024);
  gemm_mx_big<U, E, WM, WN, OUT8><<<grid, dim3(NWAVE * 64), lds, s>>>(
      static_cast<const uint8_t*>(a.X8), a.ldx, static_cast<const uint8_t*>(a.SX), static_cast<const uint8_t*>(a.W8),
      static_cast<const uint8_t*>(a.SW), static_cast<const U*>(a.bias), static_cast<U*>(a.Y), a.ldy, static_cast<const U*>(a.R1), a.ldr,
      a.row_mask, a.mask_period, static_cast<uint8_t*>(a.Y8), static_cast<uint8_t*>(a.SY), a.M, a.N, a.K, n_tiles, tiles_total);
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

int mx_linear(int dtype, const MxLinearArgs& a, hipStream_t s) {
  const int id = mx_geometry(a);
  const bool gelu = a.act == ACT_GELU, out8 = a.Y8 != nullptr;
  const int epi = (gelu ? EPI_GELU : 0) | (a.R1 ? EPI_R1 : 0) | (a.row_mask ? EPI_MASK : 0);
  auto go = [&](auto* tag) -> int {
    using U = std::remove_pointer_t<decltype(tag)>;
    if (out8) {
      if (id == 2) return gelu ? mx_launch<U, EPI_GELU, 2, 4, true>(a, s) : mx_launch<U, 0, 2, 4, true>(a, s);
      return gelu ? mx_launch<U, EPI_GELU, 2, 2, true>(a, s) : mx_launch<U, 0, 2, 2, true>(a, s);
    }
    switch (epi) {
      case 0: return id == 2 ? mx_launch<U, 0, 2, 4, false>(a, s) : mx_launch<U, 0, 2, 2, false>(a, s);
      case EPI_GELU: return id == 2 ? mx_launch<U, EPI_GELU, 2, 4, false>(a, s) : mx_launch<U, EPI_GELU, 2, 2, false>(a, s);
      case EPI_R1: return id == 2 ? mx_launch<U, EPI_R1, 2, 4, false>(a, s) : mx_launch<U, EPI_R1, 2, 2, false>(a, s);
      case EPI_R1 | EPI_MASK: return id == 2 ? mx_launch<U, EPI_R1 | EPI_MASK, 2, 4, false>(a, s) : mx_launch<U, EPI_R1 | EPI_MASK, 2, 2, false>(a, s);
      default: break;
    }
    return D3PM_E_SHAPE;
  };
  return dtype == D3PM_F16 ? go(static_cast<f16*>(nullptr)) : go(static_cast<bf16*>(nullptr));
}

int layernorm_mx(int dtype, const void* x, uint8_t* y8, uint8_t* sx, const void* w, const void* b, const void* film,
                 const void* w2, const void* b2, uint8_t* y8_2, uint8_t* sx_2, int M, int d, float eps, hipStream_t s) {
  D3PM_REQUIRE(d == 512 && (dtype == D3PM_F16 || dtype == D3PM_BF16), D3PM_E_SHAPE, "layernorm_mx: d = 512, 16-bit input only");
  const dim3 grid((M + 3) / 4), block(256);
  if (dtype == D3PM_F16)
    layernorm_mx_rows<f16><<<grid, block, 0, s>>>(static_cast<const f16*>(x), y8, sx, static_cast<const f16*>(w),
                                                  static_cast<const f16*>(b), static_cast<const f16*>(film),
                                                  static_cast<const f16*>(w2), static_cast<const f16*>(b2), y8_2, sx_2, M, eps);
  else
    layernorm_mx_rows<bf16><<<grid, block, 0, s>>>(static_cast<const bf16*>(x), y8, sx, static_cast<const bf16*>(w),
                                                   static_cast<const bf16*>(b), static_cast<const bf16*>(film),
                                                   static_cast<const bf16*>(w2), static_cast<const bf16*>(b2), y8_2, sx_2, M, eps);
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

int quantize_mx(int dtype, const void* x, int ldx, uint8_t* y8, uint8_t* sx, int M, int K, hipStream_t s) {
  D3PM_REQUIRE((dtype == D3PM_F16 || dtype == D3PM_BF16) && K % 128 == 0 && ldx % 8 == 0 && aligned16m(x), D3PM_E_SHAPE,
               "quantize_mx: 16-bit input, K a multiple of 128, 16-byte aligned rows");
  const long long waves = static_cast<long long>(M) * ((K + 511) / 512);
  const dim3 grid(static_cast<unsigned>((waves + 3) / 4)), block(256);
  if (dtype == D3PM_F16) quantize_mx_rows<f16><<<grid, block, 0, s>>>(static_cast<const f16*>(x), ldx, y8, sx, M, K);
  else quantize_mx_rows<bf16><<<grid, block, 0, s>>>(static_cast<const bf16*>(x), ldx, y8, sx, M, K);
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

}  // namespace d3pm
