// d3pm_mfma_attn32.hip -- self-attention of a DiT block on the 32 x 32 x 16 matrix instruction (head_dim 64, f16 / bf16), gfx950.
//
// Replaces the need_weights branch of torch's multi_head_attention_forward as called by DiTBlock.forward
// (/root/reference/vall_e/vall_e/ar_discrete.py:132) at throughput batch sizes; d3pm_mfma_attn.hip keeps every other case
// (cross-attention, ragged keys, key lengths, one or two utterances).
//
// Why a second kernel.  At head_dim 64 a score costs 256 flop of matrix work and one v_exp_f32, and an MFMA holds its SIMD's
// vector issue port for 8 cycles whatever its shape: 8 of the 16 cycles of v_mfma_f32_16x16x32, 8 of the 32 of
// v_mfma_f32_32x32x16 (MI355X_MICROARCH.md, constants table).  With the 16 x 16 instruction the softmax of one wave therefore
// cannot run under the matrix products of its SIMD partner -- measured in round 2: products alone 34.5 us, softmax alone 28.9,
// together 56.5 -- with the 32 x 32 instruction three quarters of the issue slots stay open.  Same flash-style numerics as
// attn_mfma_hd64 (scores in the log2 domain relative to a deferred running reference, un-normalised 16-bit probabilities into
// the second product, fp32 row sums divided out at the end) except that the row sum adds the fp32 probabilities on the VALU
// (a ones-row on this instruction would cost half a P.V product).
//
// Workgroup = 4 wave64 = 128 queries of one (utterance, head), two workgroups per CU; a wave owns 32 queries and walks the
// keys in tiles of 64:
//   * S^T = K . Q^T: A = K fragment from LDS (lane: key l & 31, 8 head-dim elements 16 ks + 8 (l >> 5)), B = Q fragment held
//     in registers; a lane ends up with 2 x 16 scores of ONE query (keys 32 kb + 8 j + 4 (l >> 5) + r), so the row maximum is
//     in-lane plus one v_permlane32_swap;
//   * the exponentiated scores, eight consecutive registers at a time, ARE the B operand of O^T += V^T . P (contraction slot
//     8 (l >> 5) + i <-> key 16 t + 8 (i >> 2) + 4 (l >> 5) + (i & 3), the same permutation on both operands);
//   * V stays row-major in LDS and is read column-major with ds_read_b64_tr_b16 (4 keys x 16 columns per 16-lane group);
//   * K / V tiles are staged global -> registers -> LDS one tile ahead, one barrier per tile; 128-byte LDS rows with the
//     16-byte chunks XOR-swizzled (K: (row >> 1) & 7 -- the b128 fragment reads of 32 rows x one chunk; V: ((row >> 1) & 1)
//     << 2 -- the transposed reads of 4 rows x 64 bytes), both conflict-free by construction.
#include "d3pm_kernels.h"

namespace d3pm {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef short short4v __attribute__((ext_vector_type(4)));

constexpr int HD = 64, BKV = 64, ROWB = 128;
constexpr int TILE = BKV * ROWB;   // 8 KiB per K or V tile
constexpr float kDefer = 8.0f;     // log2 of the largest un-normalised probability tolerated before m_ref is raised

template <typename T> __device__ __forceinline__ floatx16 mma32(uint4 a, uint4 b, floatx16 c);
template <> __device__ __forceinline__ floatx16 mma32<f16>(uint4 a, uint4 b, floatx16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ floatx16 mma32<bf16>(uint4 a, uint4 b, floatx16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ int k_off(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int v_off(int row, int chunk) { return row * ROWB + ((chunk ^ (((row >> 1) & 1) << 2)) << 4); }

template <typename T> __device__ __forceinline__ uint32_t pack2(float a, float b) {
  typedef float float2v __attribute__((ext_vector_type(2)));
  typedef T pair __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((float2v){a, b}, pair));
}

// the value of lane l ^ 32 (v_permlane32_swap of two copies leaves one holding the low half twice, the other the high half twice)
__device__ __forceinline__ void halves(float x, float& lo, float& hi) {
  lo = x;
  hi = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
}

// Output rows of the 32 x 32 kernels.  A lane holds, per 8-column group k = 4 db + j of its query's 64 output columns, columns
// 8 k + 4 hh .. + 3 (hh = lane >> 5): the natural store is eight 8-byte stores per lane, and that tail is bound by the number of
// store INSTRUCTIONS, not by bytes (MI355X_MICROARCH.md, 'attention epilogue store tail').  v_permlane32_swap between the packed
// registers of groups k and k + 1 (lanes 32-63 of the first trade places with lanes 0-31 of the second) leaves lanes 0-31 with
// columns 8 k .. 8 k + 7 and lanes 32-63 with 8 k + 8 .. 8 k + 15: four 16-byte stores, same bytes, same addresses.
template <typename T>
__device__ __forceinline__ void store_rows32(T* row_base /* O + row * ldo + h * 64, 16-byte aligned */, const floatx16 (&acc)[2], float inv, int hh,
                                             bool live) {
  T* const p = row_base + (hh ? 8 : 0);
#pragma unroll
  for (int k = 0; k < 8; k += 2) {
    const int db = k >> 2, j = k & 3;
    uint32_t ax = pack2<T>(acc[db][4 * j] * inv, acc[db][4 * j + 1] * inv), ay = pack2<T>(acc[db][4 * j + 2] * inv, acc[db][4 * j + 3] * inv);
    uint32_t bx = pack2<T>(acc[db][4 * j + 4] * inv, acc[db][4 * j + 5] * inv), by = pack2<T>(acc[db][4 * j + 6] * inv, acc[db][4 * j + 7] * inv);
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ax), "+v"(bx));
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(ay), "+v"(by));
    // (an `sc1` store here -- the GEMM epilogues' way of keeping their operands in L2 -- measured slower: 45.0 vs 41.0 us per launch)
    if (live) *reinterpret_cast<uint4*>(p + 8 * k) = uint4{ax, ay, bx, by};
  }
}

// STAMP (A/B library only): wave 0 of the first workgroup and of one in the middle of the grid records the shader clock at eight
// points of every key tile (d3pm_debug_attn32_stamps); see the stamp() calls for the points
constexpr int kStampTiles = 12, kStampPoints = 8;
__device__ unsigned long long g_attn32_stamp[2 * kStampTiles * kStampPoints];

template <typename T, int STAMP = 0>   // 1: coarse (start / end of every 32nd workgroup: shader clocks + 100 MHz ticks), 2: per tile
__global__ __launch_bounds__(256, 2) void attn32_hd64(const T* __restrict__ Q, int ldq, const T* __restrict__ Kp,
                                                      const T* __restrict__ Vp, int ldkv, T* __restrict__ O, int ldo, int Tq,
                                                      int S, float scale, int H, int n_qblocks) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * TILE];   // [buffer][K tile | V tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long stamp_c = 0, stamp_r = 0;
  if constexpr (STAMP == 1) { stamp_c = __builtin_amdgcn_s_memtime(); stamp_r = __builtin_amdgcn_s_memrealtime(); }
  // XCD-aware order: each XCD gets a contiguous range of (utterance, head, query-block) ids, so the query blocks that share
  // one K / V share one L2
  int bid;
  {
    const int nblocks = gridDim.x, q = nblocks >> 3, r = nblocks & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int qb = bid % n_qblocks, h = (bid / n_qblocks) % H, b = bid / (n_qblocks * H);
  const int q0 = (qb * 4 + wave) * 32;
  const int qn = lane & 31, hh = lane >> 5;
  const T* Kb = Kp + static_cast<size_t>(b) * S * ldkv + h * HD;
  const T* Vb = Vp + static_cast<size_t>(b) * S * ldkv + h * HD;

  // Q fragments (B operand of S^T = K . Q^T): lane holds q[query][16 ks + 8 hh .. +7], pre-scaled by sqrt(1/hd) * log2(e)
  const float qscale = scale * 1.4426950408889634f;
  uint4 qf[4];
  {
    const T* qp = Q + (static_cast<size_t>(b) * Tq + q0 + qn) * ldq + h * HD + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      typedef T tvec8 __attribute__((ext_vector_type(8)));
      const tvec8 e = __builtin_bit_cast(tvec8, *reinterpret_cast<const uint4*>(qp + ks * 16));
      qf[ks] = uint4{pack2<T>(static_cast<float>(e[0]) * qscale, static_cast<float>(e[1]) * qscale),
                     pack2<T>(static_cast<float>(e[2]) * qscale, static_cast<float>(e[3]) * qscale),
                     pack2<T>(static_cast<float>(e[4]) * qscale, static_cast<float>(e[5]) * qscale),
                     pack2<T>(static_cast<float>(e[6]) * qscale, static_cast<float>(e[7]) * qscale)};
    }
  }

  // staging: thread -> 2 x 16 B of the K tile and 2 x 16 B of the V tile (rows r0s and r0s + 32, chunk chs)
  const int r0s = tid >> 3, chs = tid & 7;
  const int ko0 = k_off(r0s, chs), ko1 = k_off(r0s + 32, chs), vo0 = v_off(r0s, chs), vo1 = v_off(r0s + 32, chs);
  struct Staged { uint4 k0, k1, v0, v1; };
  const uint32_t lo0 = static_cast<uint32_t>(r0s * ldkv + chs * 8) * 2u, lo1 = lo0 + static_cast<uint32_t>(32 * ldkv) * 2u;
  auto load_tile = [=](int tile) -> Staged {
    Staged st;
    const char* kt = reinterpret_cast<const char*>(Kb + static_cast<size_t>(tile) * BKV * ldkv);
    const char* vt = reinterpret_cast<const char*>(Vb + static_cast<size_t>(tile) * BKV * ldkv);
    st.k0 = *reinterpret_cast<const uint4*>(kt + lo0);
    st.k1 = *reinterpret_cast<const uint4*>(kt + lo1);
    st.v0 = *reinterpret_cast<const uint4*>(vt + lo0);
    st.v1 = *reinterpret_cast<const uint4*>(vt + lo1);
    return st;
  };
  auto store_tile = [=](char* base, const Staged& st) {
    *reinterpret_cast<uint4*>(base + ko0) = st.k0;
    *reinterpret_cast<uint4*>(base + ko1) = st.k1;
    *reinterpret_cast<uint4*>(base + TILE + vo0) = st.v0;
    *reinterpret_cast<uint4*>(base + TILE + vo1) = st.v1;
  };

  const int n_tiles = S / BKV;
  Staged st = load_tile(0);
  store_tile(smem, st);
  __syncthreads();

  float m_ref = 0.f, lsum = 0.f;
  floatx16 negm, acc_o[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { negm[i] = 0.f; acc_o[0][i] = 0.f; acc_o[1][i] = 0.f; }

  // per-lane LDS offsets of the fragment reads, computed once (the XOR swizzles only touch address bits that the tile-local
  // constants do not):
  //   K (b128): row 32 kb + qn, chunk 2 ks + hh                                   -> ok[ks] + kb * 4096
  //   V (tr b64): row 16 t + 8 sub + 4 hh + (qi >> 2), columns 32 db + 16 cb + 4 (qi & 3) -> ov[db] + (16 t + 8 sub) * 128
  int ok[4], ov[2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) ok[ks] = k_off(qn, 2 * ks + hh);
  {
    const int qi = lane & 15, cb = (lane >> 4) & 1;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int col = 32 * db + 16 * cb + 4 * (qi & 3);
      ov[db] = v_off(4 * hh + (qi >> 2), col >> 3) + (col & 7) * 2;
    }
  }

  const int stamp_slot = blockIdx.x == 0 ? 0 : (blockIdx.x == (gridDim.x / 16) * 8 + 3 ? 1 : -1);
  auto stamp = [&](int tile, int point) __attribute__((always_inline)) {
    if constexpr (STAMP == 2) {
      __builtin_amdgcn_sched_barrier(0);
      if (stamp_slot >= 0 && tid == 0 && tile < kStampTiles)
        g_attn32_stamp[(stamp_slot * kStampTiles + tile) * kStampPoints + point] = __builtin_amdgcn_s_memtime();
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto do_tile = [&](int tile, auto BUF_) __attribute__((always_inline)) {
    constexpr int BUF = decltype(BUF_)::value;
    stamp(tile, 0);                                     // top of the tile (behind the previous tile's barrier)
    const char* kb_ = smem + BUF * 2 * TILE;
    const char* vb_ = kb_ + TILE;
    const bool more = tile + 1 < n_tiles;
    if (more) st = load_tile(tile + 1);
    stamp(tile, 1);                                     // the next tile's global loads are issued

    // ---- S^T tile: 64 keys x 32 queries; the accumulator starts at -m_ref, so the scores leave the matrix pipe relative to
    // the running reference
    floatx16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const uint4 kf = *reinterpret_cast<const uint4*>(kb_ + ok[ks] + kb * 32 * ROWB);
        s[kb] = mma32<T>(kf, qf[ks], ks == 0 ? negm : s[kb]);
      }
    float mx = fmaxf(s[0][0], s[1][0]);
#pragma unroll
    for (int i = 1; i < 16; ++i) mx = fmaxf(fmaxf(mx, s[0][i]), s[1][i]);
    // Deferred maximum: m_ref only moves when some score of the wave exceeds it by more than 2^kDefer (or on the first tile)
    if (tile == 0 || __any(mx > kDefer)) {               // wave-uniform
      float a, c;
      halves(mx, a, c);                                  // lanes l and l ^ 32 share a query
      mx = fmaxf(a, c);
      const float delta = tile == 0 ? mx : fmaxf(mx, 0.f);
#pragma unroll
      for (int i = 0; i < 16; ++i) { s[0][i] -= delta; s[1][i] -= delta; }
      if (tile != 0) {
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        lsum *= alpha;
#pragma unroll
        for (int i = 0; i < 16; ++i) { acc_o[0][i] *= alpha; acc_o[1][i] *= alpha; }
      }
      m_ref += delta;
#pragma unroll
      for (int i = 0; i < 16; ++i) negm[i] = -m_ref;
    }
    stamp(tile, 2);                                     // scores available: K reads, 8 MFMAs, the maximum (+ a rare rescale)
    uint4 pf[4];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s[kb][i] = __builtin_amdgcn_exp2f(s[kb][i]);
      float t0 = 0.f, t1 = 0.f;
#pragma unroll
      for (int i = 0; i < 16; i += 2) { t0 += s[kb][i]; t1 += s[kb][i + 1]; }
      lsum += t0 + t1;
#pragma unroll
      for (int u = 0; u < 2; ++u)
        pf[2 * kb + u] = uint4{pack2<T>(s[kb][8 * u + 0], s[kb][8 * u + 1]), pack2<T>(s[kb][8 * u + 2], s[kb][8 * u + 3]),
                               pack2<T>(s[kb][8 * u + 4], s[kb][8 * u + 5]), pack2<T>(s[kb][8 * u + 6], s[kb][8 * u + 7])};
    }

    stamp(tile, 3);                                     // exponentials, row sum, packing
    // ---- O^T += V^T . P ----
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        typedef short4v __attribute__((address_space(3))) * lds_ptr;
        const short4v va = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(vb_ + ov[db] + (16 * t) * ROWB));
        const short4v vc = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(vb_ + ov[db] + (16 * t + 8) * ROWB));
        const uint2 lo = __builtin_bit_cast(uint2, va), hi = __builtin_bit_cast(uint2, vc);
        acc_o[db] = mma32<T>(uint4{lo.x, lo.y, hi.x, hi.y}, pf[t], acc_o[db]);
      }
    stamp(tile, 4);                                     // V reads and the 8 P.V MFMAs issued
    if constexpr (STAMP == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp(tile, 5);                                     // the next tile's global loads have returned
    if (more) store_tile(smem + (BUF ^ 1) * 2 * TILE, st);
    stamp(tile, 6);                                     // its LDS stores issued
    __syncthreads();
    stamp(tile, 7);                                     // barrier passed
  };
  for (int tile = 0; tile < n_tiles; tile += 2) {
    do_tile(tile, std::integral_constant<int, 0>{});
    if (tile + 1 < n_tiles) do_tile(tile + 1, std::integral_constant<int, 1>{});
  }

  float la, lc;
  halves(lsum, la, lc);
  const float inv = 1.0f / (la + lc);
  store_rows32<T>(O + (static_cast<size_t>(b) * Tq + q0 + qn) * ldo + h * HD, acc_o, inv, hh, true);
  if constexpr (STAMP == 1) {
    if (tid == 0 && (blockIdx.x & 31) == 0 && (blockIdx.x >> 5) < 48) {
      unsigned long long* o = g_attn32_stamp + (blockIdx.x >> 5) * 4;
      o[0] = stamp_c; o[1] = __builtin_amdgcn_s_memtime(); o[2] = stamp_r; o[3] = __builtin_amdgcn_s_memrealtime();
    }
  }
}

// ---- the software-pipelined form ---------------------------------------------------------------------------------------
// One wave's instruction stream carries matrix and vector work of DIFFERENT 32-key blocks side by side: two waves on a SIMD do
// not overlap one's MFMAs with the other's exponentials (they add up: round 2), but inside one stream the 24 free cycles of
// every 32 x 32 x 16 MFMA take vector instructions for nothing.  Per 32-key block j (two per key tile):
//   phase 1: S(j + 1) = K(j + 1) . Q^T [4 MFMAs]  beside  exp2 of S(j), its packing into P(j), the V(j) fragment reads
//   phase 2: O^T += V(j)^T . P(j)     [4 MFMAs]  beside  the rest of exp2 / packing, the row sum of P(j), the K(j + 2)
//            fragment reads and the maximum of S(j + 1)
// and then the deferred-maximum test on S(j + 1) (rarely taken: rescale O, l and S(j + 1), raise m_ref) -- behind P.V(j) and
// the sum of P(j), in front of the product of block j + 2, so every number is the one the plain walk computes.
// The groups are pinned with sched_barrier(0) (hipcc otherwise hoists the exponentials in front of the products).  K(j + 1)
// of the next tile is read half a tile before that tile's turn: three LDS buffers, tiles staged two ahead, still one barrier
// per tile.
// ABL != 0: timing-only ablations for tools/probe_attn32.hip (WRONG results): 1 no exponentials, 2 no S products, 4 no P.V
// products, 8 no V fragment reads, 16 no K fragment reads, 32 no staging / barrier after the prologue, 64 no maximum / test,
// 128 no packing, 256 no row sum
// NW = waves (32-query groups) per workgroup: 4 = 128 queries, three workgroups per CU; 6 = 192 queries, two workgroups per CU --
// the same twelve waves per CU, but every staged K / V tile serves half as many queries again (a third fewer bytes L2 -> LDS and
// staging instructions per query; waves 4 and 5 do not stage).  A query's arithmetic does not depend on NW: bit-identical.
template <typename T, int STAMP = 0, int ABL = 0, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 3 : 2) void attn32p_hd64(const T* __restrict__ Q, int ldq, const T* __restrict__ Kp,
                                                       const T* __restrict__ Vp, int ldkv, T* __restrict__ O, int ldo, int Tq,
                                                       int S, float scale, int H, int n_qblocks) {
  __shared__ __attribute__((aligned(16))) char smem[3 * 2 * TILE];   // [buffer][K tile | V tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned long long stamp_c = 0, stamp_r = 0;
  if constexpr (STAMP == 1) { stamp_c = __builtin_amdgcn_s_memtime(); stamp_r = __builtin_amdgcn_s_memrealtime(); }
  int bid;
  {
    const int nblocks = gridDim.x, q = nblocks >> 3, r = nblocks & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int qb = bid % n_qblocks, h = (bid / n_qblocks) % H, b = bid / (n_qblocks * H);
  const int q0 = (qb * NW + wave) * 32;
  const bool stager = NW == 4 || wave < 4;               // wave-uniform: 256 threads stage a tile
  const int qn = lane & 31, hh = lane >> 5;
  const T* Kb = Kp + static_cast<size_t>(b) * S * ldkv + h * HD;
  const T* Vb = Vp + static_cast<size_t>(b) * S * ldkv + h * HD;

  const float qscale = scale * 1.4426950408889634f;
  uint4 qf[4];
  {
    const T* qp = Q + (static_cast<size_t>(b) * Tq + q0 + qn) * ldq + h * HD + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      typedef T tvec8 __attribute__((ext_vector_type(8)));
      const tvec8 e = __builtin_bit_cast(tvec8, *reinterpret_cast<const uint4*>(qp + ks * 16));
      qf[ks] = uint4{pack2<T>(static_cast<float>(e[0]) * qscale, static_cast<float>(e[1]) * qscale),
                     pack2<T>(static_cast<float>(e[2]) * qscale, static_cast<float>(e[3]) * qscale),
                     pack2<T>(static_cast<float>(e[4]) * qscale, static_cast<float>(e[5]) * qscale),
                     pack2<T>(static_cast<float>(e[6]) * qscale, static_cast<float>(e[7]) * qscale)};
    }
  }

  const int r0s = tid >> 3, chs = tid & 7;
  const int ko0 = k_off(r0s, chs), ko1 = k_off(r0s + 32, chs), vo0 = v_off(r0s, chs), vo1 = v_off(r0s + 32, chs);
  struct Staged { uint4 k0, k1, v0, v1; };
  const uint32_t lo0 = static_cast<uint32_t>(r0s * ldkv + chs * 8) * 2u, lo1 = lo0 + static_cast<uint32_t>(32 * ldkv) * 2u;
  auto load_tile = [=](int tile) -> Staged {
    Staged st;
    const char* kt = reinterpret_cast<const char*>(Kb + static_cast<size_t>(tile) * BKV * ldkv);
    const char* vt = reinterpret_cast<const char*>(Vb + static_cast<size_t>(tile) * BKV * ldkv);
    st.k0 = *reinterpret_cast<const uint4*>(kt + lo0);
    st.k1 = *reinterpret_cast<const uint4*>(kt + lo1);
    st.v0 = *reinterpret_cast<const uint4*>(vt + lo0);
    st.v1 = *reinterpret_cast<const uint4*>(vt + lo1);
    return st;
  };
  auto store_tile = [=](char* base, const Staged& st) {
    *reinterpret_cast<uint4*>(base + ko0) = st.k0;
    *reinterpret_cast<uint4*>(base + ko1) = st.k1;
    *reinterpret_cast<uint4*>(base + TILE + vo0) = st.v0;
    *reinterpret_cast<uint4*>(base + TILE + vo1) = st.v1;
  };

  const int n_tiles = S / BKV;
  if (stager) {
    Staged st = load_tile(0);
    store_tile(smem, st);
    if (n_tiles > 1) {
      st = load_tile(1);
      store_tile(smem + 2 * TILE, st);
    }
  }
  __syncthreads();

  int ok[4], ov[2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) ok[ks] = k_off(qn, 2 * ks + hh);
  {
    const int qi = lane & 15, cb = (lane >> 4) & 1;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int col = 32 * db + 16 * cb + 4 * (qi & 3);
      ov[db] = v_off(4 * hh + (qi >> 2), col >> 3) + (col & 7) * 2;
    }
  }

  float m_ref = 0.f;
  float ls[4] = {0.f, 0.f, 0.f, 0.f};
  floatx16 negm, acc_o[2], sA, sB;
#pragma unroll
  for (int i = 0; i < 16; ++i) { negm[i] = 0.f; acc_o[0][i] = 0.f; acc_o[1][i] = 0.f; }
  uint4 kf[4];
  auto read_k = [&](const char* kblock) __attribute__((always_inline)) {   // the four K fragments of a 32-key block
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kf[ks] = *reinterpret_cast<const uint4*>(kblock + ok[ks]);
  };
  auto row_max = [&](const floatx16& s) __attribute__((always_inline)) -> float {
    float mx = fmaxf(s[0], s[1]);
#pragma unroll
    for (int i = 2; i < 16; i += 2) mx = fmaxf(fmaxf(mx, s[i]), s[i + 1]);
    return mx;
  };

  // ---- prologue: the scores of block 0 and the first reference
  read_k(smem);
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) sA = mma32<T>(kf[ks], qf[ks], ks == 0 ? negm : sA);
  read_k(smem + 32 * ROWB);
  {
    float a, c;
    halves(row_max(sA), a, c);
    const float delta = fmaxf(a, c);
#pragma unroll
    for (int i = 0; i < 16; ++i) sA[i] -= delta;
    m_ref = delta;
#pragma unroll
    for (int i = 0; i < 16; ++i) negm[i] = -m_ref;
  }

#define SB() __builtin_amdgcn_sched_barrier(0)
#define EXP2(x) ((ABL & 1) ? (x) + 1.0f : __builtin_amdgcn_exp2f(x))
  // one block: sc = S(j) relative to m_ref; NEXT: sn <- S(j + 1), kf holds K(j + 1) on entry and K(j + 2) (from knext) on exit;
  // vblock = the 32 V rows of block j
  // STG: which half of the tile staged two ahead this block carries through registers (0 none, 1 the K rows, 2 the V rows): a
  // half lives in 8 registers for one block instead of the whole tile in 16 for two (three waves per SIMD: 168 registers)
  auto block = [&](floatx16& sc, floatx16& sn, const char* vblock, const char* knext, auto NEXT_, auto STG_, int stile, char* sdst)
      __attribute__((always_inline)) {
    constexpr bool NEXT = decltype(NEXT_)::value;
    constexpr int STG = (ABL & 32) ? 0 : decltype(STG_)::value;
    typedef short4v __attribute__((address_space(3))) * lds_ptr;
    uint4 vf[2], pf, h0, h1;
    auto read_v = [&](int t, int db) __attribute__((always_inline)) {
      if constexpr (ABL & 8) {
        vf[db] = uint4{0x3c003c00u + t, 0x3c003800u + db, 0x38003c00u, 0x3c003c00u ^ static_cast<uint32_t>(lane)};
      } else {
        const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(vblock + ov[db] + (16 * t) * ROWB)));
        const uint2 hi = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(vblock + ov[db] + (16 * t + 8) * ROWB)));
        vf[db] = uint4{lo.x, lo.y, hi.x, hi.y};
      }
    };
    // Vector work per MFMA gap is sized from tools/probe_issue2.hip: beside one v_mfma_f32_32x32x16 (32 cycles) fit three v_exp_f32
    // or v_cvt_pk (8 - 10 cycles each) or six plain adds; v_pk_add_f32 / v_pk_mul_f32 do NOT overlap with the matrix pipe (three
    // per gap: 62 cycles per gap), so the row sum is four chains of plain v_add_f32 from asm (hipcc would pair them)
    auto e4 = [&](int i) __attribute__((always_inline)) {
      sc[i] = EXP2(sc[i]); sc[i + 1] = EXP2(sc[i + 1]); sc[i + 2] = EXP2(sc[i + 2]); sc[i + 3] = EXP2(sc[i + 3]);
      if constexpr (!(ABL & 256)) {
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(ls[0]) : "v"(sc[i]));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(ls[1]) : "v"(sc[i + 1]));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(ls[2]) : "v"(sc[i + 2]));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(ls[3]) : "v"(sc[i + 3]));
      }
    };
    auto pack8 = [&](int i) __attribute__((always_inline)) {
      if constexpr (ABL & 128) pf = uint4{__builtin_bit_cast(uint32_t, sc[i]), __builtin_bit_cast(uint32_t, sc[i + 2]), __builtin_bit_cast(uint32_t, sc[i + 4]), __builtin_bit_cast(uint32_t, sc[i + 6])};
      else pf = uint4{pack2<T>(sc[i], sc[i + 1]), pack2<T>(sc[i + 2], sc[i + 3]), pack2<T>(sc[i + 4], sc[i + 5]), pack2<T>(sc[i + 6], sc[i + 7])};
    };
    auto pv = [&](int db, int slot) __attribute__((always_inline)) {
      if constexpr (ABL & 4) acc_o[db][slot] += __builtin_bit_cast(float, pf.x) + __builtin_bit_cast(float, vf[db].y);
      else acc_o[db] = mma32<T>(vf[db], pf, acc_o[db]);
    };
    SB();
    if constexpr (STG != 0) {      // unconditional in the tile index (a branch on it splits the block and hipcc sinks the vector work
      // behind it): past the end the last tile is staged again, into a buffer that is no longer read
      if (NW == 4 || stager) {
        const char* gt = reinterpret_cast<const char*>((STG == 1 ? Kb : Vb) + static_cast<size_t>(stile < n_tiles ? stile : n_tiles - 1) * BKV * ldkv);
        h0 = *reinterpret_cast<const uint4*>(gt + lo0);
        h1 = *reinterpret_cast<const uint4*>(gt + lo1);
      }
    }
    read_v(0, 0);
    read_v(0, 1);
    SB();
    // ---- phase 1
    if constexpr (NEXT) { if constexpr (ABL & 2) sn = negm; else sn = mma32<T>(kf[0], qf[0], negm); }
    e4(0);
    SB();
    if constexpr (NEXT && !(ABL & 2)) sn = mma32<T>(kf[1], qf[1], sn);
    e4(4);
    SB();
    if constexpr (NEXT && !(ABL & 2)) sn = mma32<T>(kf[2], qf[2], sn);
    pack8(0);
    SB();
    if constexpr (NEXT && !(ABL & 2)) sn = mma32<T>(kf[3], qf[3], sn);
    e4(8);
    SB();
    // ---- phase 2
    pv(0, 0);
    read_v(1, 0);                                          // the fragment register is free once the product is issued
    e4(12);
    SB();
    pv(1, 0);
    read_v(1, 1);
    if constexpr (NEXT && !(ABL & 16)) read_k(knext);
    pack8(8);
    SB();
    pv(0, 1);
    float mx = 0.f;
    if constexpr (NEXT && !(ABL & 64)) mx = row_max(sn);
    SB();
    pv(1, 1);
    if constexpr (STG != 0) {
      if (NW == 4 || stager) {
        *reinterpret_cast<uint4*>(sdst + (STG == 1 ? ko0 : TILE + vo0)) = h0;
        *reinterpret_cast<uint4*>(sdst + (STG == 1 ? ko1 : TILE + vo1)) = h1;
      }
    }
    SB();
    if constexpr (NEXT && !(ABL & 64)) {
      if (__builtin_expect(__any(mx > kDefer), 0)) {       // wave-uniform, rare
        float a, c;
        halves(mx, a, c);                                  // lanes l and l ^ 32 share a query
        const float delta = fmaxf(fmaxf(a, c), 0.f);
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        m_ref += delta;
        const float nm = -m_ref;
        // in place, from asm: as plain C++ the updates become loop-carried phis that hipcc resolves with sixteen v_mov_b64 per
        // block on the COMMON path
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          asm volatile("v_sub_f32 %0, %0, %1" : "+v"(sn[i]) : "v"(delta));
          asm volatile("v_mul_f32 %0, %0, %1" : "+v"(acc_o[0][i]) : "v"(alpha));
          asm volatile("v_mul_f32 %0, %0, %1" : "+v"(acc_o[1][i]) : "v"(alpha));
          asm volatile("v_mov_b32 %0, %1" : "+v"(negm[i]) : "v"(nm));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(ls[i]) : "v"(alpha));
      }
    }
    SB();
  };
  using Yes = std::integral_constant<bool, true>;
  using No = std::integral_constant<bool, false>;
  using StK = std::integral_constant<int, 1>;
  using StV = std::integral_constant<int, 2>;
  using St0 = std::integral_constant<int, 0>;

  int cur = 0, nx1 = 2 * TILE, nx2 = 4 * TILE;               // byte offsets of the buffers of tiles i, i + 1, i + 2
  // every tile but the last: both blocks have a successor (one straight-line body: with the last tile's shorter second block
  // inside the loop hipcc moves the output accumulators between two register sets in every block)
  for (int i = 0; i + 1 < n_tiles; ++i) {
    const char* vb = smem + cur + TILE;
    block(sA, sB, vb, smem + nx1, Yes{}, StK{}, i + 2, smem + nx2);                     // keys 0..31 of tile i; then K(tile i + 1, first half)
    block(sB, sA, vb + 32 * ROWB, smem + nx1 + 32 * ROWB, Yes{}, StV{}, i + 2, smem + nx2);
    if constexpr (!(ABL & 32)) __syncthreads();
    const int t = cur;
    cur = nx1; nx1 = nx2; nx2 = t;
  }
  {
    const char* vb = smem + cur + TILE;
    block(sA, sB, vb, smem, Yes{}, St0{}, 0, smem);           // (the K fragments read here are not used)
    block(sB, sA, vb + 32 * ROWB, smem, No{}, St0{}, 0, smem);
  }
#undef SB
#undef EXP2

  float la, lc;
  halves((ls[0] + ls[1]) + (ls[2] + ls[3]), la, lc);
  const float inv = 1.0f / (la + lc);
  store_rows32<T>(O + (static_cast<size_t>(b) * Tq + q0 + qn) * ldo + h * HD, acc_o, inv, hh, true);
  if constexpr (STAMP == 1) {
    if (tid == 0 && (blockIdx.x & 31) == 0 && (blockIdx.x >> 5) < 48) {
      unsigned long long* o = g_attn32_stamp + (blockIdx.x >> 5) * 4;
      o[0] = stamp_c; o[1] = __builtin_amdgcn_s_memtime(); o[2] = stamp_r; o[3] = __builtin_amdgcn_s_memrealtime();
    }
  }
}

// ---- the cross-attention pair of a block on the same instruction: attn32_cross_hd64 ------------------------------------------
// attn_cross_hd64 (d3pm_mfma_attn.hip) keeps every K / V tile of the text (<= 64 keys) and prompt (<= 256 keys) problems of one
// (utterance, head) in LDS and walks that head's 256-query blocks with no barrier; what its launch then waits for is the issue
// port of each SIMD (16 x 16 x 32: ~1300 issue cycles per 64-key tile and wave, matrix and vector work adding up).  This kernel
// is that residency with the software-pipelined 32-key block of attn32p_hd64: a wave owns 32 queries, the products of block
// j + 1 and the exponentials of block j share one instruction stream.  Differences from the self-attention kernel: no staging
// and no barrier inside the walk (the image is resident), key counts that are not multiples of 32 -- the LAST block's product
// starts from -m_ref on its valid keys and from -inf on the others (cmask: the C operand does the masking, no select in the
// stream), blocks past the last valid key are not walked -- and two problems per query block with the next phase's queries
// fetched under the current one.  K swizzle as attn_cross_hd64; V swizzle ((row >> 1) & 1) << 2 as attn32p_hd64 reads it.
typedef const __attribute__((address_space(1))) void* glb_ptr32_t;
typedef __attribute__((address_space(3))) void* lds_ptr32_t;

template <typename T>
__global__ __launch_bounds__(512, 2) void attn32_cross_hd64(const T* __restrict__ Q1, const T* __restrict__ K1, const T* __restrict__ V1,
                                                            T* __restrict__ O1, int S1, const T* __restrict__ Q2,
                                                            const T* __restrict__ K2, const T* __restrict__ V2, T* __restrict__ O2,
                                                            int S2, int ldq, int ldkv, int ldo, int Tq, float scale, int H,
                                                            int n_qblocks, int n_qsplit) {
  extern __shared__ __attribute__((aligned(16))) char smem_x[];   // [tile][K | V], 16 KiB per tile: tile 0 = text, 1.. = prompt
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid;
  {
    const int nblocks = gridDim.x, q = nblocks >> 3, r = nblocks & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int qs = bid % n_qsplit, h = (bid / n_qsplit) % H, b = bid / (n_qsplit * H);
  const int nt2 = (S2 + BKV - 1) / BKV, n_tiles = 1 + nt2;
  {   // every K / V piece of both problems: 16 pieces (1 KiB = 8 rows x 128 B) per tile, dealt over the 8 waves
    const int lrow = lane >> 3, cpos = lane & 7;
    const int total = n_tiles * 16;
    for (int p = wave; p < total; p += 8) {                 // wave-uniform trip count
      const int tile = p >> 4, which = (p >> 3) & 1, j = p & 7;   // which: 0 = K, 1 = V
      const int row = 8 * j + lrow;
      const int logical = which == 0 ? (cpos ^ ((row >> 1) & 7)) : (cpos ^ (((row >> 1) & 1) << 2));
      const T* base = tile == 0 ? (which == 0 ? K1 : V1) : (which == 0 ? K2 : V2);
      const int S = tile == 0 ? S1 : S2;
      int key = (tile == 0 ? 0 : (tile - 1) * BKV) + row;
      key = key < S ? key : S - 1;
      const T* src = base + (static_cast<size_t>(b) * S + key) * ldkv + h * HD + logical * 8;
      __builtin_amdgcn_global_load_lds((glb_ptr32_t)src, (lds_ptr32_t)(smem_x + tile * 2 * TILE + which * TILE + j * 1024), 16, 0, 0);
    }
  }
  const int qn = lane & 31, hh = lane >> 5;
  const float qscale = scale * 1.4426950408889634f;
  int ok[4], ov[2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) ok[ks] = k_off(qn, 2 * ks + hh);
  {
    const int qi = lane & 15, cb = (lane >> 4) & 1;
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      const int col = 32 * db + 16 * cb + 4 * (qi & 3);
      ov[db] = v_off(4 * hh + (qi >> 2), col >> 3) + (col & 7) * 2;
    }
  }
  uint4 qraw[4];
  auto fetch_q = [&](int qb, int prob) __attribute__((always_inline)) {
    const T* Q = prob == 0 ? Q1 : Q2;
    int qrow = (qb * 8 + wave) * 32 + qn;
    qrow = qrow < Tq ? qrow : Tq - 1;
    const T* qp = Q + (static_cast<size_t>(b) * Tq + qrow) * ldq + h * HD + hh * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qraw[ks] = *reinterpret_cast<const uint4*>(qp + ks * 16);
  };
  fetch_q(qs, 0);
  bool landed = false;
#define SB() __builtin_amdgcn_sched_barrier(0)
  for (int qb = qs; qb < n_qblocks; qb += n_qsplit) {
  for (int prob = 0; prob < 2; ++prob) {
    T* O = prob == 0 ? O1 : O2;
    const int S = prob == 0 ? S1 : S2, t0 = prob == 0 ? 0 : 1;
    const int nb = (S + 31) >> 5;                            // 32-key blocks with at least one valid key
    const char* const pbase = smem_x + t0 * 2 * TILE;        // block j: K rows at pbase + (j >> 1) * 2 TILE + (j & 1) * 32 rows, V + TILE
    uint4 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      typedef T tvec8 __attribute__((ext_vector_type(8)));
      const tvec8 e = __builtin_bit_cast(tvec8, qraw[ks]);
      qf[ks] = uint4{pack2<T>(static_cast<float>(e[0]) * qscale, static_cast<float>(e[1]) * qscale),
                     pack2<T>(static_cast<float>(e[2]) * qscale, static_cast<float>(e[3]) * qscale),
                     pack2<T>(static_cast<float>(e[4]) * qscale, static_cast<float>(e[5]) * qscale),
                     pack2<T>(static_cast<float>(e[6]) * qscale, static_cast<float>(e[7]) * qscale)};
    }
    if (prob == 0) fetch_q(qb, 1);
    else if (qb + n_qsplit < n_qblocks) fetch_q(qb + n_qsplit, 0);
    if (!landed) {                                           // one wait for the whole workgroup's K / V image
      __syncthreads();                                       // (drains this wave's DMA pieces, then the barrier)
      landed = true;
    }
    // the last block's product starts from cmask - m_ref: 0 on its valid keys (lane: keys 8 j + 4 hh + r <-> element 4 j + r), -inf on the rest
    floatx16 cmask, negm, acc_o[2], sA, sB;
    {
      const int lim = S - 32 * (nb - 1) - 4 * hh;            // this lane's key 8 j + r of the last block exists iff 8 j + r < lim
#pragma unroll
      for (int i = 0; i < 16; ++i) cmask[i] = (8 * (i >> 2) + (i & 3) < lim) ? 0.f : -INFINITY;
    }
    float m_ref = 0.f;
    float ls[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) { negm[i] = 0.f; acc_o[0][i] = 0.f; acc_o[1][i] = 0.f; }
    uint4 kf[4];
    auto kblock = [&](int j) __attribute__((always_inline)) -> const char* { return pbase + (j >> 1) * 2 * TILE + (j & 1) * 32 * ROWB; };
    auto read_k = [&](const char* kb) __attribute__((always_inline)) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) kf[ks] = *reinterpret_cast<const uint4*>(kb + ok[ks]);
    };
    auto row_max = [&](const floatx16& s) __attribute__((always_inline)) -> float {
      float mx = fmaxf(s[0], s[1]);
#pragma unroll
      for (int i = 2; i < 16; i += 2) mx = fmaxf(fmaxf(mx, s[i]), s[i + 1]);
      return mx;
    };
    // ---- prologue: the scores of block 0 (masked if it is also the last one) and the first reference
    read_k(kblock(0));
    {
      floatx16 c0 = negm;
      if (nb == 1) c0 = cmask;                               // wave-uniform
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) sA = mma32<T>(kf[ks], qf[ks], ks == 0 ? c0 : sA);
    }
    if (nb > 1) read_k(kblock(1));
    {
      float a, c;
      halves(row_max(sA), a, c);
      const float delta = fmaxf(a, c);                       // finite: block 0 has a valid key
#pragma unroll
      for (int i = 0; i < 16; ++i) sA[i] -= delta;
      m_ref = delta;
#pragma unroll
      for (int i = 0; i < 16; ++i) negm[i] = -m_ref;
    }
    // one block (attn32p_hd64's, without staging): sc = S(j) relative to m_ref; NEXT: sn <- S(j + 1) from kf (K(j + 1), read by the
    // previous block), LAST: that product is the last block's and starts from cmask - m_ref; READK: kf <- K(j + 2) from knext
    auto block = [&](floatx16& sc, floatx16& sn, const char* vblock, const char* knext, auto NEXT_, auto LAST_, auto READK_)
        __attribute__((always_inline)) {
      constexpr bool NEXT = decltype(NEXT_)::value, LAST = decltype(LAST_)::value, READK = decltype(READK_)::value;
      typedef short4v __attribute__((address_space(3))) * lds_ptr;
      uint4 vf[2], pf;
      auto read_v = [&](int t, int db) __attribute__((always_inline)) {
        const uint2 lo = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(vblock + ov[db] + (16 * t) * ROWB)));
        const uint2 hi = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(vblock + ov[db] + (16 * t + 8) * ROWB)));
        vf[db] = uint4{lo.x, lo.y, hi.x, hi.y};
      };
      auto e4 = [&](int i) __attribute__((always_inline)) {
        sc[i] = __builtin_amdgcn_exp2f(sc[i]); sc[i + 1] = __builtin_amdgcn_exp2f(sc[i + 1]);
        sc[i + 2] = __builtin_amdgcn_exp2f(sc[i + 2]); sc[i + 3] = __builtin_amdgcn_exp2f(sc[i + 3]);
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(ls[0]) : "v"(sc[i]));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(ls[1]) : "v"(sc[i + 1]));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(ls[2]) : "v"(sc[i + 2]));
        asm volatile("v_add_f32 %0, %0, %1" : "+v"(ls[3]) : "v"(sc[i + 3]));
      };
      auto pack8 = [&](int i) __attribute__((always_inline)) {
        pf = uint4{pack2<T>(sc[i], sc[i + 1]), pack2<T>(sc[i + 2], sc[i + 3]), pack2<T>(sc[i + 4], sc[i + 5]), pack2<T>(sc[i + 6], sc[i + 7])};
      };
      auto pv = [&](int db) __attribute__((always_inline)) { acc_o[db] = mma32<T>(vf[db], pf, acc_o[db]); };
      floatx16 cn = negm;
      if constexpr (NEXT && LAST) {
#pragma unroll
        for (int i = 0; i < 16; ++i) cn[i] = negm[i] + cmask[i];
      }
      SB();
      read_v(0, 0);
      read_v(0, 1);
      SB();
      // ---- phase 1
      if constexpr (NEXT) sn = mma32<T>(kf[0], qf[0], cn);
      e4(0);
      SB();
      if constexpr (NEXT) sn = mma32<T>(kf[1], qf[1], sn);
      e4(4);
      SB();
      if constexpr (NEXT) sn = mma32<T>(kf[2], qf[2], sn);
      pack8(0);
      SB();
      if constexpr (NEXT) sn = mma32<T>(kf[3], qf[3], sn);
      e4(8);
      SB();
      // ---- phase 2
      pv(0);
      read_v(1, 0);
      e4(12);
      SB();
      pv(1);
      read_v(1, 1);
      if constexpr (READK) read_k(knext);
      pack8(8);
      SB();
      pv(0);
      float mx = 0.f;
      if constexpr (NEXT) mx = row_max(sn);
      SB();
      pv(1);
      SB();
      if constexpr (NEXT) {
        if (__builtin_expect(__any(mx > kDefer), 0)) {       // wave-uniform, rare
          float a, c;
          halves(mx, a, c);
          const float delta = fmaxf(fmaxf(a, c), 0.f);
          const float alpha = __builtin_amdgcn_exp2f(-delta);
          m_ref += delta;
          const float nm = -m_ref;
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            asm volatile("v_sub_f32 %0, %0, %1" : "+v"(sn[i]) : "v"(delta));      // -inf (a masked key) stays -inf
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(acc_o[0][i]) : "v"(alpha));
            asm volatile("v_mul_f32 %0, %0, %1" : "+v"(acc_o[1][i]) : "v"(alpha));
            asm volatile("v_mov_b32 %0, %1" : "+v"(negm[i]) : "v"(nm));
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(ls[i]) : "v"(alpha));
        }
      }
      SB();
    };
    using Yes = std::integral_constant<bool, true>;
    using No = std::integral_constant<bool, false>;
    auto vblk = [&](int j) __attribute__((always_inline)) -> const char* { return kblock(j) + TILE; };
    int j = 0;
    for (; j + 3 < nb; j += 2) {                              // both blocks: a successor that is not the last, and a K block to read
      block(sA, sB, vblk(j), kblock(j + 2), Yes{}, No{}, Yes{});
      block(sB, sA, vblk(j + 1), kblock(j + 3), Yes{}, No{}, Yes{});
    }
    const int rest = nb - j;                                  // 1, 2 or 3 (wave-uniform)
    if (rest == 1) {
      block(sA, sB, vblk(j), pbase, No{}, No{}, No{});
    } else if (rest == 2) {
      block(sA, sB, vblk(j), pbase, Yes{}, Yes{}, No{});
      block(sB, sA, vblk(j + 1), pbase, No{}, No{}, No{});
    } else {
      block(sA, sB, vblk(j), kblock(j + 2), Yes{}, No{}, Yes{});
      block(sB, sA, vblk(j + 1), pbase, Yes{}, Yes{}, No{});
      block(sA, sB, vblk(j + 2), pbase, No{}, No{}, No{});
    }
    float la, lc;
    halves((ls[0] + ls[1]) + (ls[2] + ls[3]), la, lc);
    const float inv = 1.0f / (la + lc);
    const int qrow = (qb * 8 + wave) * 32 + qn;              // (the swaps need every lane; only the store is predicated)
    store_rows32<T>(O + (static_cast<size_t>(b) * Tq + (qrow < Tq ? qrow : Tq - 1)) * ldo + h * HD, acc_o, inv, hh, qrow < Tq);
  }
  }   // query blocks
#undef SB
}

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

}  // namespace

// whole 128-query blocks and whole 64-key tiles of a single problem without key lengths: the self-attention of a DiT block
bool mfma_attention32_supported(int dtype, const AttnArgs& a) {
  if (dtype != D3PM_F16 && dtype != D3PM_BF16) return false;
  if (a.hd != HD || a.Q2 != nullptr || a.key_len != nullptr) return false;
  if (a.Tq < 128 || a.Tq % 128 || a.S < BKV || a.S % BKV) return false;
  if (a.ldq % 8 || a.ldkv % 8 || a.ldo % 8) return false;      // 16-byte output stores
  return aligned(a.Q, 16) && aligned(a.K, 16) && aligned(a.V, 16) && aligned(a.O, 16);
}

#ifdef D3PM_ABLATIONS
int read_attn32_stamps(unsigned long long* out, int n) {
  const int total = 2 * kStampTiles * kStampPoints;
  D3PM_CHECK_HIP(hipDeviceSynchronize());
  D3PM_CHECK_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_attn32_stamp), sizeof(unsigned long long) * (n < total ? n : total)));
  return D3PM_OK;
}
#endif

// the cross-attention pair of a block with both K / V images resident (<= 64 text keys, <= 256 prompt keys), no key lengths
bool mfma_attention32_cross_supported(int dtype, const AttnArgs& a) {
  if (dtype != D3PM_F16 && dtype != D3PM_BF16) return false;
  if (a.hd != HD || a.Q2 == nullptr || a.key_len != nullptr) return false;
  if (a.S < 1 || a.S > BKV || a.S2 < 1 || a.S2 > 4 * BKV || a.Tq < 1) return false;
  if (a.ldq % 8 || a.ldkv % 8 || a.ldo % 8) return false;      // 16-byte output stores
  return aligned(a.Q, 16) && aligned(a.K, 16) && aligned(a.V, 16) && aligned(a.O, 16) && aligned(a.Q2, 16) && aligned(a.K2, 16) &&
         aligned(a.V2, 16) && aligned(a.O2, 16);
}

int mfma_attention32_cross(int dtype, const AttnArgs& a, int n_qsplit, hipStream_t s) {
  const int n_qblocks = (a.Tq + 255) / 256;
  const dim3 grid(static_cast<unsigned>(n_qsplit * a.H * a.B)), block(512);
  const size_t lds = static_cast<size_t>(1 + (a.S2 + BKV - 1) / BKV) * 2 * TILE;
  auto go = [&](auto* tag) -> int {
    using U = std::remove_pointer_t<decltype(tag)>;
    D3PM_LDS_ATTR((&attn32_cross_hd64<U>), 5 * 2 * TILE);
    attn32_cross_hd64<U><<<grid, block, lds, s>>>(static_cast<const U*>(a.Q), static_cast<const U*>(a.K), static_cast<const U*>(a.V),
                                                  static_cast<U*>(a.O), a.S, static_cast<const U*>(a.Q2), static_cast<const U*>(a.K2),
                                                  static_cast<const U*>(a.V2), static_cast<U*>(a.O2), a.S2, a.ldq, a.ldkv, a.ldo, a.Tq,
                                                  a.scale, a.H, n_qblocks, n_qsplit);
    D3PM_LAUNCH_CHECK();
    return D3PM_OK;
  };
  return dtype == D3PM_F16 ? go(static_cast<f16*>(nullptr)) : go(static_cast<bf16*>(nullptr));
}

int mfma_attention32(int dtype, const AttnArgs& a, hipStream_t s) {
  const int n_qblocks = a.Tq / 128;
  const dim3 grid(static_cast<unsigned>(n_qblocks * a.H * a.B)), block(256);
#ifdef D3PM_ABLATIONS
  if (ab_knobs().attn_arm >= 321 && ab_knobs().attn_arm <= 324 && dtype == D3PM_BF16) {      // coarse stamps: 321 the plain walk, 322 the pipelined one; 323 / 324: the same at ONE workgroup per CU (idle dynamic LDS)
    const int arm = ab_knobs().attn_arm;
    const size_t pad = arm >= 323 ? 72 * 1024 : 0;
    D3PM_LDS_ATTR((&attn32_hd64<bf16, 1>), 96 * 1024);
    D3PM_LDS_ATTR((&attn32p_hd64<bf16, 1>), 96 * 1024);
    if (arm == 321 || arm == 323)
      attn32_hd64<bf16, 1><<<grid, block, pad, s>>>(static_cast<const bf16*>(a.Q), a.ldq, static_cast<const bf16*>(a.K), static_cast<const bf16*>(a.V),
                                                  a.ldkv, static_cast<bf16*>(a.O), a.ldo, a.Tq, a.S, a.scale, a.H, n_qblocks);
    else
      attn32p_hd64<bf16, 1><<<grid, block, pad, s>>>(static_cast<const bf16*>(a.Q), a.ldq, static_cast<const bf16*>(a.K), static_cast<const bf16*>(a.V),
                                                   a.ldkv, static_cast<bf16*>(a.O), a.ldo, a.Tq, a.S, a.scale, a.H, n_qblocks);
    D3PM_LAUNCH_CHECK();
    return D3PM_OK;
  }
  if (ab_knobs().attn_arm == 320 && dtype == D3PM_BF16) {      // the stamped build (timing probe)
    attn32_hd64<bf16, 2><<<grid, block, 0, s>>>(static_cast<const bf16*>(a.Q), a.ldq, static_cast<const bf16*>(a.K), static_cast<const bf16*>(a.V),
                                                   a.ldkv, static_cast<bf16*>(a.O), a.ldo, a.Tq, a.S, a.scale, a.H, n_qblocks);
    D3PM_LAUNCH_CHECK();
    return D3PM_OK;
  }
#endif
  if (tune_of(a.tune).attn_query_groups != 33) {           // 33: the plain walk (A/B against the pipelined one)
#ifdef D3PM_ABLATIONS
    // A/B library only (attn_query_groups = 35): 192-query workgroups of six waves, two per CU -- a third fewer K / V bytes staged
    // per query, bit-identical, measured SLOWER (65.8 vs 52.8 us at 32 x 768, 23.0 vs 19.4 at 32 x 384: tests/ab_attn32.py,
    // profiles/round3_t_ab_attn32.txt): the staged bytes are not what this kernel waits for
    if (tune_of(a.tune).attn_query_groups == 35 && a.Tq % 192 == 0) {
      const dim3 grid6(static_cast<unsigned>(a.Tq / 192 * a.H * a.B)), block6(384);
      if (dtype == D3PM_F16)
        attn32p_hd64<f16, 0, 0, 6><<<grid6, block6, 0, s>>>(static_cast<const f16*>(a.Q), a.ldq, static_cast<const f16*>(a.K), static_cast<const f16*>(a.V),
                                                            a.ldkv, static_cast<f16*>(a.O), a.ldo, a.Tq, a.S, a.scale, a.H, a.Tq / 192);
      else
        attn32p_hd64<bf16, 0, 0, 6><<<grid6, block6, 0, s>>>(static_cast<const bf16*>(a.Q), a.ldq, static_cast<const bf16*>(a.K), static_cast<const bf16*>(a.V),
                                                             a.ldkv, static_cast<bf16*>(a.O), a.ldo, a.Tq, a.S, a.scale, a.H, a.Tq / 192);
      D3PM_LAUNCH_CHECK();
      return D3PM_OK;
    }
#endif
    if (dtype == D3PM_F16)
      attn32p_hd64<f16><<<grid, block, 0, s>>>(static_cast<const f16*>(a.Q), a.ldq, static_cast<const f16*>(a.K), static_cast<const f16*>(a.V),
                                               a.ldkv, static_cast<f16*>(a.O), a.ldo, a.Tq, a.S, a.scale, a.H, n_qblocks);
    else
      attn32p_hd64<bf16><<<grid, block, 0, s>>>(static_cast<const bf16*>(a.Q), a.ldq, static_cast<const bf16*>(a.K), static_cast<const bf16*>(a.V),
                                                a.ldkv, static_cast<bf16*>(a.O), a.ldo, a.Tq, a.S, a.scale, a.H, n_qblocks);
    D3PM_LAUNCH_CHECK();
    return D3PM_OK;
  }
  if (dtype == D3PM_F16)
    attn32_hd64<f16><<<grid, block, 0, s>>>(static_cast<const f16*>(a.Q), a.ldq, static_cast<const f16*>(a.K), static_cast<const f16*>(a.V),
                                            a.ldkv, static_cast<f16*>(a.O), a.ldo, a.Tq, a.S, a.scale, a.H, n_qblocks);
  else
    attn32_hd64<bf16><<<grid, block, 0, s>>>(static_cast<const bf16*>(a.Q), a.ldq, static_cast<const bf16*>(a.K), static_cast<const bf16*>(a.V),
                                             a.ldkv, static_cast<bf16*>(a.O), a.ldo, a.Tq, a.S, a.scale, a.H, n_qblocks);
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

}  // namespace d3pm
