// d3pm_mfma_attn.hip -- flash-style attention on MFMA for head_dim 64 (f16 / bf16), gfx950.
//
// Replaces the need_weights branch of torch's multi_head_attention_forward as called by
// DiTBlock.forward (/root/reference/vall_e/vall_e/ar_discrete.py:132 self, :138 text, :142 prompt):
// q*sqrt(1/hd) -> bmm -> softmax -> bmm.  The [T,S] score matrix and the head-averaged weights the
// reference materialises (and throws away) never exist here.
//
// Workgroup = 4 wave64 = 64 (QG=1) or 128 (QG=2) queries of one (utterance, head); each wave owns 16*QG queries and walks
// the keys in tiles of 64 with an online softmax.
//   * S^T = K.Q^T by v_mfma_f32_16x16x32 (K fragment = A operand from LDS, Q fragment = B operand
//     held in registers for the whole kernel): a lane holds 4 consecutive keys x 4 key tiles of ONE
//     query, so max / sum run in-lane plus two xor-shuffles (lanes l, l^16, l^32, l^48 share a query);
//   * the exponentiated scores of two key tiles, converted to 16 bit, ARE the B operand of the
//     P.V product (contraction index permuted consistently on both operands): no LDS round trip;
//   * V stays row-major in LDS and is read column-major with ds_read_b64_tr_b16 (4 keys x 16
//     columns per 16-lane group) as the A operand, giving O^T: a lane owns 4 consecutive output
//     columns of its own query, so the softmax rescale needs no cross-lane traffic;
//   * K and V tiles are staged global -> registers -> LDS (16 B per lane) one tile ahead, one
//     barrier per tile; 128-B LDS rows, 16-B chunks XOR-swizzled (K: (row>>1)&7 for the b128
//     fragment reads, V: ((row>>1)&3)<<1 for the transposed reads) -- conflict-free by construction.
// Numerics (flash-style, NOT the eager rounding points -- the generic attention_rows keeps those): scores stay
// fp32 in the log2 domain, the probabilities enter the second MFMA un-normalised (2^(s-m) <= 1, rounded to
// the storage dtype), the row sum is accumulated in fp32 and divided out at the end.  Differs from the eager
// round(bmm) -> softmax -> round -> bmm chain by rounding noise only (precision modes).
// The softmax is VALU-bound at head_dim 64 (256 flop of MFMA per score), so it is kept to ~6 VALU issue
// slots per score: max, sub, v_exp_f32, add, half a cvt_pk; the O rescale is skipped when no maximum moved
// and the key-padding compare/select only runs on a ragged last tile.
#include "d3pm_kernels.h"

namespace d3pm {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));

constexpr int HD = 64, BKV = 64, ROWB = 128;
constexpr int TILE = BKV * ROWB;   // 8 KiB per K or V tile

template <typename T> __device__ __forceinline__ floatx4 mma(uint4 a, uint4 b, floatx4 c);
template <> __device__ __forceinline__ floatx4 mma<f16>(uint4 a, uint4 b, floatx4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ floatx4 mma<bf16>(uint4 a, uint4 b, floatx4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ int k_off(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int v_off(int row, int chunk) { return row * ROWB + ((chunk ^ (((row >> 1) & 3) << 1)) << 4); }

template <class F> __device__ __forceinline__ void static_for4(F&& f) {
  f(std::integral_constant<int, 0>{}); f(std::integral_constant<int, 1>{}); f(std::integral_constant<int, 2>{}); f(std::integral_constant<int, 3>{});
}

constexpr float kDefer = 8.0f;   // log2 of the largest un-normalised probability tolerated before m_ref is raised

// max over the four lanes l, l^16, l^32, l^48 on the VALU (no LDS round trip): after v_permlane16_swap of two copies
// one holds the even 16-lane rows twice and the other the odd rows twice; v_permlane32_swap does the same for halves.
// Inline asm: see the note on the two-result builtin in d3pm_mfma_gemm.hip; s_nop 1 = the VALU-write hazard.
__device__ __forceinline__ float max_over_query_lanes(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  x = fmaxf(a, b);
  a = x;
  b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}

// two floats -> one packed 16-bit pair; the vector conversion lowers to a single v_cvt_pk_{bf16,f16}_f32
template <typename T> __device__ __forceinline__ uint32_t pack2(float a, float b) {
  typedef float float2v __attribute__((ext_vector_type(2)));
  typedef T pair __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((float2v){a, b}, pair));
}

// Output rows of the 16 x 16 kernels.  A lane holds, per 16-column block dt of its query's 64 output columns, columns
// 16 dt + 4 g .. + 3 (g = lane >> 4): four 8-byte stores per query group, a tail bound by the number of store INSTRUCTIONS
// (MI355X_MICROARCH.md, 'attention epilogue store tail').  v_permlane16_swap between the packed registers of blocks dt and dt + 1
// (odd 16-lane rows of the first trade places with the even rows of the second -- the regrouping of the GEMM epilogue,
// d3pm_mfma_tile.h) leaves a lane with 8 consecutive columns (dt + (g & 1)) 16 + (g >> 1) 8 .. + 7: two 16-byte stores, same
// bytes, same addresses.  Every lane takes part in the swaps; only the store is predicated.
template <typename T>
__device__ __forceinline__ void store_rows16(T* row_base /* O + row * ldo + h * 64, 16-byte aligned */, const floatx4 (&acc)[4], float inv, int g,
                                             bool live) {
#pragma unroll
  for (int dt = 0; dt < 4; dt += 2) {
    uint32_t ax = pack2<T>(acc[dt][0] * inv, acc[dt][1] * inv), ay = pack2<T>(acc[dt][2] * inv, acc[dt][3] * inv);
    uint32_t bx = pack2<T>(acc[dt + 1][0] * inv, acc[dt + 1][1] * inv), by = pack2<T>(acc[dt + 1][2] * inv, acc[dt + 1][3] * inv);
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(ax), "+v"(bx));
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(ay), "+v"(by));
    if (live) *reinterpret_cast<uint4*>(row_base + (dt + (g & 1)) * 16 + (g >> 1) * 8) = uint4{ax, ay, bx, by};
  }
}

// ABL != 0 is instantiated in the A/B library only (-DD3PM_ABLATIONS): timing-only ablation builds (results wrong by construction; tests/ab_attn.py): 1 no v_exp, 2 no maximum / rescale logic, 4 no K/V
// staging after the first tile, 8 no barriers, 16 no P.V product, 32 no Q.K product
// fragment reads whose completion is waited for by hand (ABL bit 6: every K and V fragment of a tile issued at the top of
// the tile, counted lgkmcnt before each consumer) -- hipcc sinks a plain LDS load to the instruction before its first use
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_read_b128(u32x4& dst, uint32_t addr, int off) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off));
}
__device__ __forceinline__ void lds_read_tr_b64(u32x2& dst, uint32_t addr, int off) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off));
}
template <int N> __device__ __forceinline__ void lds_wait2(u32x4& a, u32x4& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N)); }
template <int N> __device__ __forceinline__ void lds_wait_v(u32x2 (&v)[2][4][2]) {
  asm volatile("s_waitcnt lgkmcnt(%16)"
               : "+v"(v[0][0][0]), "+v"(v[0][0][1]), "+v"(v[0][1][0]), "+v"(v[0][1][1]), "+v"(v[0][2][0]), "+v"(v[0][2][1]), "+v"(v[0][3][0]),
                 "+v"(v[0][3][1]), "+v"(v[1][0][0]), "+v"(v[1][0][1]), "+v"(v[1][1][0]), "+v"(v[1][1][1]), "+v"(v[1][2][0]), "+v"(v[1][2][1]),
                 "+v"(v[1][3][0]), "+v"(v[1][3][1])
               : "n"(N));
}

// one 1-KiB direct-to-LDS piece (8 rows x 128 B: lane -> row lane >> 3, 16-byte chunk lane & 7) issued from asm: invisible to
// hipcc's wait counters (a compiler-visible DMA is drained before the next ds_read), so the wait is placed by hand
__device__ __forceinline__ void dma_piece(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

template <typename T, int QG, bool PAIR, int ABL = 0>   // QG groups of 16 queries per wave (K/V fragments are read once per wave and reused)
__global__ __launch_bounds__(256, QG == 1 ? 4 : 2) void attn_mfma_hd64(const T* __restrict__ Q, int ldq, const T* __restrict__ Kp,
                                                      const T* __restrict__ Vp, int ldkv, T* __restrict__ O, int ldo,
                                                      int Tq, int S, float scale, int H, int n_qblocks,
                                                      const T* __restrict__ Q2, const T* __restrict__ K2,
                                                      const T* __restrict__ V2, T* __restrict__ O2, int S2, int n_first,
                                                      const int32_t* __restrict__ key_len) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 2 * TILE];   // [buffer][K tile | V tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // 1-D grid, XCD-aware order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a contiguous
  // range of (utterance, head, query-block) ids -- the query blocks that share one K/V then share one L2
  // (PMC: L2 hit rate of the plain 3-D grid was 42 %).
  int bid;
  {
    const int nblocks = gridDim.x, q = nblocks >> 3, r = nblocks & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // Paired launch (the text and prompt cross-attentions of a block: same queries' rows, different Q / K / V): n_first
  // < 0 -> every workgroup runs problem 1 and then problem 2 for its query block, sharing its fixed costs (half the
  // workgroups; the 50-key text problem alone is a single tile per workgroup); n_first >= 0 -> the second half of the
  // grid takes problem 2.
  const bool sequential = PAIR && n_first < 0;
  if (!sequential && bid >= n_first) {   // block-uniform
    bid -= n_first;
    Q = Q2; Kp = K2; Vp = V2; O = O2; S = S2;
  }
  const int qb = bid % n_qblocks, h = (bid / n_qblocks) % H, b = bid / (n_qblocks * H);
  for (int prob = 0; prob < (PAIR ? 2 : 1); ++prob) {
  if (prob == 1) { Q = Q2; Kp = K2; Vp = V2; O = O2; S = S2; }
  const int S_pad = S;                                 // row stride of the K/V batches
  if (key_len) S = min(key_len[b], S_pad);             // valid keys of this utterance (>= 1)
  const int q0 = (qb * 4 + wave) * (16 * QG);
  const int qi = lane & 15, g = lane >> 4;
  const T* Kb = Kp + static_cast<size_t>(b) * S_pad * ldkv + h * HD;
  const T* Vb = Vp + static_cast<size_t>(b) * S_pad * ldkv + h * HD;

  // Q fragments (B operand of S^T = K.Q^T): lane holds q[query][32*ks + 8g .. +7], pre-scaled by
  // sqrt(1/hd) * log2(e) so that the scores come out of the MFMA in the log2 domain (softmax via v_exp_f32)
  const float qscale = scale * 1.4426950408889634f;
  uint4 qf[QG][2];
#pragma unroll
  for (int qg = 0; qg < QG; ++qg) {
    int qrow = q0 + qg * 16 + qi;
    qrow = qrow < Tq ? qrow : Tq - 1;
    const T* qp = Q + (static_cast<size_t>(b) * Tq + qrow) * ldq + h * HD;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      typedef T tvec8 __attribute__((ext_vector_type(8)));
      const tvec8 e = __builtin_bit_cast(tvec8, *reinterpret_cast<const uint4*>(qp + ks * 32 + g * 8));
      qf[qg][ks] = uint4{pack2<T>(static_cast<float>(e[0]) * qscale, static_cast<float>(e[1]) * qscale),
                         pack2<T>(static_cast<float>(e[2]) * qscale, static_cast<float>(e[3]) * qscale),
                         pack2<T>(static_cast<float>(e[4]) * qscale, static_cast<float>(e[5]) * qscale),
                         pack2<T>(static_cast<float>(e[6]) * qscale, static_cast<float>(e[7]) * qscale)};
    }
  }

  // staging: thread -> 2 x 16 B of the K tile and 2 x 16 B of the V tile (rows r0s and r0s+32, chunk chs);
  // plain scalars / by-value struct so that the in-flight tile lives in VGPRs, not in scratch
  const int r0s = tid >> 3, chs = tid & 7;
  const int ko0 = k_off(r0s, chs), ko1 = k_off(r0s + 32, chs), vo0 = v_off(r0s, chs), vo1 = v_off(r0s + 32, chs);
  struct Staged { uint4 k0, k1, v0, v1; };
  // whole tiles: a wave-uniform tile base (scalar registers) + two per-lane byte offsets that never change, so a tile costs
  // no vector address arithmetic; only a ragged last tile clamps its rows
  const uint32_t lo0 = static_cast<uint32_t>(r0s * ldkv + chs * 8) * 2u, lo1 = lo0 + static_cast<uint32_t>(32 * ldkv) * 2u;
  auto load_tile = [=](int tile) -> Staged {
    Staged st;
    const char* kt = reinterpret_cast<const char*>(Kb + static_cast<size_t>(tile) * BKV * ldkv);
    const char* vt = reinterpret_cast<const char*>(Vb + static_cast<size_t>(tile) * BKV * ldkv);
    if ((tile + 1) * BKV <= S) {
      st.k0 = *reinterpret_cast<const uint4*>(kt + lo0);
      st.k1 = *reinterpret_cast<const uint4*>(kt + lo1);
      st.v0 = *reinterpret_cast<const uint4*>(vt + lo0);
      st.v1 = *reinterpret_cast<const uint4*>(vt + lo1);
    } else {
      int a = tile * BKV + r0s, c = a + 32;
      a = a < S ? a : S - 1;
      c = c < S ? c : S - 1;
      st.k0 = *reinterpret_cast<const uint4*>(Kb + static_cast<size_t>(a) * ldkv + chs * 8);
      st.k1 = *reinterpret_cast<const uint4*>(Kb + static_cast<size_t>(c) * ldkv + chs * 8);
      st.v0 = *reinterpret_cast<const uint4*>(Vb + static_cast<size_t>(a) * ldkv + chs * 8);
      st.v1 = *reinterpret_cast<const uint4*>(Vb + static_cast<size_t>(c) * ldkv + chs * 8);
    }
    return st;
  };
  auto store_tile = [=](char* base, const Staged& st) {
    *reinterpret_cast<uint4*>(base + ko0) = st.k0;
    *reinterpret_cast<uint4*>(base + ko1) = st.k1;
    *reinterpret_cast<uint4*>(base + TILE + vo0) = st.v0;
    *reinterpret_cast<uint4*>(base + TILE + vo1) = st.v1;
  };

  const int n_tiles = (S + BKV - 1) / BKV;
  // ABL bit 7 (same results): K / V tiles go global -> LDS directly, four 1-KiB pieces per wave and tile (pieces wave, wave + 4
  // = K rows, wave + 8, wave + 12 = V rows), issued at the top of the previous tile; a lane fetches the 16-byte chunk that the
  // swizzled image wants at its position
  constexpr bool kDma = (ABL & 128) != 0;
  const uint32_t lds0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem));
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto dma_tile = [&](int tile, int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int which = i >> 1, j = wave_u + 4 * (i & 1), row = 8 * j + (lane >> 3), cpos = lane & 7;
      const int logical = which == 0 ? (cpos ^ ((row >> 1) & 7)) : (cpos ^ (((row >> 1) & 3) << 1));
      int key = tile * BKV + row;
      key = key < S ? key : S - 1;
      const T* src = (which == 0 ? Kb : Vb) + static_cast<size_t>(key) * ldkv + logical * 8;
      dma_piece(src, lds0 + buf * 2 * TILE + which * TILE + j * 1024);
    }
  };
  Staged st;
  if constexpr (kDma) {
    dma_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    st = load_tile(0);
    store_tile(smem, st);
  }
  __syncthreads();

  float m_ref[QG];
  floatx4 negm[QG], acc_o[QG][4], acc_l[QG];
  const uint32_t one2 = pack2<T>(1.0f, 1.0f);
  const uint4 ones = uint4{one2, one2, one2, one2};
#pragma unroll
  for (int qg = 0; qg < QG; ++qg) {
    m_ref[qg] = 0.f;
    negm[qg] = floatx4{0.f, 0.f, 0.f, 0.f};
    acc_l[qg] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc_o[qg][dt] = floatx4{0.f, 0.f, 0.f, 0.f};
  }

  // per-lane LDS offsets of the fragment reads, computed once: the XOR swizzle only touches address bits that the tile-local
  // row / column-block constants do not, so every read of a tile is one of these plus an instruction immediate
  //   K (b128): row 16 kt + qi, chunk 4 ks + g          -> ok[ks] + kt * 2048
  //   V (tr b64): row 32 kb2 + 16 half + 4 g + (qi >> 2), columns 16 dt + 4 (qi & 3) -> ov[dt] + kb2 * 4096 + half * 2048
  int ok[2], ov[4];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) ok[ks] = k_off(qi, ks * 4 + g);
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    const int col = dt * 16 + 4 * (qi & 3);
    ov[dt] = v_off(4 * g + (qi >> 2), col >> 3) + (col & 7) * 2;
  }
  // two tiles per trip so that the buffer a tile reads (and the one it fills) is a compile-time constant
  auto do_tile = [&](int tile, auto BUF_) __attribute__((always_inline)) {
    constexpr int BUF = decltype(BUF_)::value;
    const char* kb = smem + BUF * 2 * TILE;
    const char* vb = kb + TILE;
    const bool more = tile + 1 < n_tiles;
    if constexpr (kDma) {
      if (more) dma_tile(tile + 1, BUF ^ 1);           // BUF ^ 1 was last read in the previous tile, behind its barrier
    } else {
      if (more && !(ABL & 4)) st = load_tile(tile + 1);
    }

    // ---- S^T tile: 64 keys x (16 QG) queries per wave; each K fragment feeds QG MFMAs ----
    // The accumulator starts at -m_ref (one register quad per query group, shared by the four key tiles as the C
    // operand), so the scores leave the matrix pipe already relative to the running reference: no subtraction.
    floatx4 s[QG][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        if constexpr (ABL & 32) {
#pragma unroll
          for (int qg = 0; qg < QG; ++qg) s[qg][kt] = negm[qg] + floatx4{0.1f * kt, 0.2f, 0.3f * tile, 0.4f};
        } else if constexpr ((ABL & 64) == 0) {
        uint4 kf = *reinterpret_cast<const uint4*>(kb + ok[ks] + kt * 16 * ROWB);
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) s[qg][kt] = mma<T>(kf, qf[qg][ks], ks == 0 ? negm[qg] : s[qg][kt]);
        }
      }
    u32x2 vfr[2][4][2];
    if constexpr ((ABL & 64) != 0) {
      const uint32_t lbase = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem)) + BUF * 2 * TILE;
      u32x4 kfr[4][2];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) lds_read_b128(kfr[kt][ks], lbase + ok[ks], kt * 16 * ROWB);
      // the LDS counter holds 15 operations: the 16 V reads go out four at a time behind each key block's MFMAs, so that
      // they return under the softmax; counts = reads younger than the fragments a step consumes
      static_for4([&](auto KT) {
        constexpr int kt = decltype(KT)::value;
        constexpr int kWaitK[4] = {6, 8, 10, 11};
        lds_wait2<kWaitK[kt]>(kfr[kt][0], kfr[kt][1]);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int qg = 0; qg < QG; ++qg)
            s[qg][kt] = mma<T>(__builtin_bit_cast(uint4, kfr[kt][ks]), qf[qg][ks], ks == 0 ? negm[qg] : s[qg][kt]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          constexpr int idx0 = kt * 4;
          const int idx = idx0 + j, kb2 = idx >> 3, dt = (idx >> 1) & 3, half = idx & 1;   // constants after unrolling
          lds_read_tr_b64(vfr[kb2][dt][half], lbase + TILE + ov[dt], (2 * kb2 + half) * 16 * ROWB);
        }
      });
    }
    // lane holds scores (log2 domain, minus m_ref) of its query for keys tile*64 + kt*16 + 4g + r
    const bool ragged = (tile == n_tiles - 1) && (S & (BKV - 1));   // wave-uniform: only the last tile can be partial
    uint4 pf[QG][2];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
      if (ragged) {
        const int lim = S - tile * BKV - 4 * g;              // keys of this lane's column that exist: compare with constants
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[qg][kt][r] = (kt * 16 + r < lim) ? s[qg][kt][r] : -INFINITY;
      }
      float mx = fmaxf(s[qg][0][0], s[qg][0][1]);          // linear chain: hipcc folds pairs into v_max3_f32
      mx = fmaxf(fmaxf(mx, s[qg][0][2]), s[qg][0][3]);
#pragma unroll
      for (int kt = 1; kt < 4; ++kt) {
        mx = fmaxf(fmaxf(mx, s[qg][kt][0]), s[qg][kt][1]);
        mx = fmaxf(fmaxf(mx, s[qg][kt][2]), s[qg][kt][3]);
      }
      // Deferred maximum: m_ref only moves when some score of the wave exceeds it by more than 2^kDefer (or on the
      // first tile), so the common tile does neither the cross-lane maximum nor the rescale of O; probabilities are
      // then at most 2^kDefer instead of 1, which neither fp32 sums nor 16-bit P notice (normalised by the same sum).
      if (!(ABL & 2) && (tile == 0 || __any(mx > kDefer))) {               // wave-uniform
        mx = max_over_query_lanes(mx);                     // lanes l, l^16, l^32, l^48 share a query
        const float delta = tile == 0 ? mx : fmaxf(mx, 0.f);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[qg][kt][r] -= delta;
        if (tile != 0) {                                   // on the first tile O and l are still zero
          const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc_l[qg][r] *= alpha;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc_o[qg][dt][r] *= alpha;
        }
        m_ref[qg] += delta;
        negm[qg] = floatx4{-m_ref[qg], -m_ref[qg], -m_ref[qg], -m_ref[qg]};
      }
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[qg][kt][r] = (ABL & 1) ? s[qg][kt][r] : __builtin_amdgcn_exp2f(s[qg][kt][r]);
      // contraction index j<4 -> key tile 2kb, j>=4 -> key tile 2kb+1 (same permutation as the V reads)
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2) {
        const floatx4 pa = s[qg][2 * kb2], pb = s[qg][2 * kb2 + 1];
        pf[qg][kb2] = uint4{pack2<T>(pa[0], pa[1]), pack2<T>(pa[2], pa[3]), pack2<T>(pb[0], pb[1]), pack2<T>(pb[2], pb[3])};
      }
    }

    // ---- O^T += V^T . P^T ; each transposed V fragment feeds QG MFMAs ----
    if constexpr (ABL & 16) {
#pragma unroll
      for (int qg = 0; qg < QG; ++qg)
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2) {
          acc_o[qg][kb2][0] += __builtin_bit_cast(float, pf[qg][kb2].x); acc_o[qg][kb2][1] += __builtin_bit_cast(float, pf[qg][kb2].y);
          acc_o[qg][kb2][2] += __builtin_bit_cast(float, pf[qg][kb2].z); acc_o[qg][kb2][3] += __builtin_bit_cast(float, pf[qg][kb2].w);
          acc_l[qg][0] += 1.0f;
        }
    } else {
    if constexpr ((ABL & 64) != 0) lds_wait_v<0>(vfr);
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        // 16-lane group g, lane qi: address of row (key0 + qi>>2), columns 16dt + 4(qi&3) .. +3
        uint4 vf;
        if constexpr ((ABL & 64) != 0) {
          vf = uint4{vfr[kb2][dt][0].x, vfr[kb2][dt][0].y, vfr[kb2][dt][1].x, vfr[kb2][dt][1].y};
        } else {
          typedef short4v __attribute__((address_space(3))) * lds_ptr;
          short4v va = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(vb + ov[dt] + (2 * kb2) * 16 * ROWB));
          short4v vc = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(vb + ov[dt] + (2 * kb2 + 1) * 16 * ROWB));
          uint2 lo = __builtin_bit_cast(uint2, va), hi = __builtin_bit_cast(uint2, vc);
          vf = uint4{lo.x, lo.y, hi.x, hi.y};
        }
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) acc_o[qg][dt] = mma<T>(vf, pf[qg][kb2], acc_o[qg][dt]);
      }
    // row sums on the matrix pipe: ones^T . P^T adds the (rounded) probabilities of all 64 keys of the tile for
    // the lane's query into every register of acc_l -- 2 MFMAs instead of 16 v_add + the final shuffles
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
      for (int qg = 0; qg < QG; ++qg) acc_l[qg] = mma<T>(ones, pf[qg][kb2], acc_l[qg]);
    }
    if constexpr (kDma) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the next tile have landed
    } else {
      if (more && !(ABL & 4)) store_tile(smem + (BUF ^ 1) * 2 * TILE, st);
    }
    if constexpr (!(ABL & 8)) __syncthreads();
  };
  for (int tile = 0; tile < n_tiles; tile += 2) {
    do_tile(tile, std::integral_constant<int, 0>{});
    if (tile + 1 < n_tiles) do_tile(tile + 1, std::integral_constant<int, 1>{});
  }

#pragma unroll
  for (int qg = 0; qg < QG; ++qg) {
    const float inv = 1.0f / acc_l[qg][0];
    const int qrow = q0 + qg * 16 + qi;
    store_rows16<T>(O + (static_cast<size_t>(b) * Tq + (qrow < Tq ? qrow : Tq - 1)) * ldo + h * HD, acc_o[qg], inv, g, qrow < Tq);
  }
  }   // problems of a sequential pair
}

// ---- cross-attention pair with short key sequences: attn_cross_hd64 ---------------------------------------------------
// The text (<= 64 keys) and prompt (<= 256 keys) cross-attentions of a DiT block (ar_discrete.py:138,142) are two tiny
// problems per query block: at B = 32 the tile-by-tile kernel above spends 49 us on 13.8 GFLOP (measured in situ) -- five
// dependent load -> LDS -> barrier rounds per workgroup at two workgroups per CU, i.e. pure latency.  Here EVERY K / V tile of
// both problems (<= 80 KiB) is fetched into LDS by direct-to-LDS DMA at kernel entry, all pieces in flight at once, the
// workgroup waits once and then runs the five tiles back to back with no further barrier.  Eight waves x 32 queries share
// the tiles (256 queries per workgroup).  Arithmetic per tile is the code of attn_mfma_hd64 (same rounding, same order).
typedef const __attribute__((address_space(1))) void* glb_ptr_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// PIPE: software-pipelined walk -- the S^T = K . Q^T products of tile i + 1 are issued BEFORE the softmax of tile i (a second
// set of score registers), so the matrix pipe works under the exponentials of the same wave instead of waiting for them: with
// every K / V tile resident a wave's time per tile is its own dependency chain (LDS reads -> 16 MFMAs -> max -> 32 v_exp ->
// V reads -> 20 MFMAs), which two waves per SIMD do not cover.  Exactness: the prefetched product starts from the -m_ref of the
// moment; when the softmax of tile i then moves m_ref (always on the first tile, otherwise only if a score exceeds it by 2^8)
// the product of tile i + 1 is issued again from the new -m_ref, so every number is the one the plain walk computes.
template <typename T, bool PIPE = false>
__global__ __launch_bounds__(512, 2) void attn_cross_hd64(const T* __restrict__ Q1, const T* __restrict__ K1, const T* __restrict__ V1,
                                                          T* __restrict__ O1, int S1, const T* __restrict__ Q2,
                                                          const T* __restrict__ K2, const T* __restrict__ V2, T* __restrict__ O2,
                                                          int S2, int ldq, int ldkv, int ldo, int Tq, float scale, int H,
                                                          int n_qblocks, int n_qsplit) {
  constexpr int QG = 2, MAXT = 5;                            // 1 text tile + up to 4 prompt tiles
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [tile][K | V], 16 KiB per tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid;
  {
    const int nblocks = gridDim.x, q = nblocks >> 3, r = nblocks & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  // a workgroup keeps the K / V image of its (utterance, head) and walks the query blocks qs, qs + n_qsplit, ..: at the
  // throughput batch n_qsplit = 1 -- one workgroup per CU, one DMA flight per (utterance, head) instead of one per query block
  const int qs = bid % n_qsplit, h = (bid / n_qsplit) % H, b = bid / (n_qsplit * H);
  const int nt2 = (S2 + BKV - 1) / BKV, n_tiles = 1 + nt2;
  // ---- every K / V piece of both problems: 16 pieces (1 KiB = 8 rows x 128 B) per tile, dealt over the 8 waves
  {
    const int lrow = lane >> 3, cpos = lane & 7;
    const int total = n_tiles * 16;
    for (int p = wave; p < total; p += 8) {                 // wave-uniform trip count
      const int tile = p >> 4, which = (p >> 3) & 1, j = p & 7;   // which: 0 = K, 1 = V
      const int row = 8 * j + lrow;
      const int logical = which == 0 ? (cpos ^ ((row >> 1) & 7)) : (cpos ^ (((row >> 1) & 3) << 1));
      const T* base = tile == 0 ? (which == 0 ? K1 : V1) : (which == 0 ? K2 : V2);
      const int S = tile == 0 ? S1 : S2;
      int key = (tile == 0 ? 0 : (tile - 1) * BKV) + row;
      key = key < S ? key : S - 1;
      const T* src = base + (static_cast<size_t>(b) * S + key) * ldkv + h * HD + logical * 8;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(smem + tile * 2 * TILE + which * TILE + j * 1024), 16, 0, 0);
    }
  }
  const int qi = lane & 15, g = lane >> 4;
  const float qscale = scale * 1.4426950408889634f;
  const uint32_t one2 = pack2<T>(1.0f, 1.0f);
  const uint4 ones = uint4{one2, one2, one2, one2};
  bool landed = false;
  // the queries of a (query block, problem) phase are fetched one phase ahead: raw rows in registers while the previous phase
  // computes, scaled and packed at the top of their own phase
  uint4 qraw[QG][2];
  auto fetch_q = [&](int qb, int prob) __attribute__((always_inline)) {
    const T* Q = prob == 0 ? Q1 : Q2;
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
      int qrow = (qb * 8 + wave) * (16 * QG) + qg * 16 + qi;
      qrow = qrow < Tq ? qrow : Tq - 1;
      const T* qp = Q + (static_cast<size_t>(b) * Tq + qrow) * ldq + h * HD;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) qraw[qg][ks] = *reinterpret_cast<const uint4*>(qp + ks * 32 + g * 8);
    }
  };
  if constexpr (!PIPE) fetch_q(qs, 0);
  for (int qb = qs; qb < n_qblocks; qb += n_qsplit) {
  const int q0 = (qb * 8 + wave) * (16 * QG);
  for (int prob = 0; prob < 2; ++prob) {
    if constexpr (PIPE) fetch_q(qb, prob);       // the pipelined walk spends its registers on the second score set: no query prefetch
    T* O = prob == 0 ? O1 : O2;
    const int S = prob == 0 ? S1 : S2, t0 = prob == 0 ? 0 : 1, nt = prob == 0 ? 1 : nt2;
    uint4 qf[QG][2];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        typedef T tvec8 __attribute__((ext_vector_type(8)));
        const tvec8 e = __builtin_bit_cast(tvec8, qraw[qg][ks]);
        qf[qg][ks] = uint4{pack2<T>(static_cast<float>(e[0]) * qscale, static_cast<float>(e[1]) * qscale),
                           pack2<T>(static_cast<float>(e[2]) * qscale, static_cast<float>(e[3]) * qscale),
                           pack2<T>(static_cast<float>(e[4]) * qscale, static_cast<float>(e[5]) * qscale),
                           pack2<T>(static_cast<float>(e[6]) * qscale, static_cast<float>(e[7]) * qscale)};
      }
    if constexpr (!PIPE) {
      if (prob == 0) fetch_q(qb, 1);
      else if (qb + n_qsplit < n_qblocks) fetch_q(qb + n_qsplit, 0);
    }
    if (!landed) {                                           // one wait for the whole workgroup's K / V image
      __syncthreads();                                       // (drains this wave's DMA pieces, then the barrier)
      landed = true;
    }
    float m_ref[QG];
    floatx4 negm[QG], acc_o[QG][4], acc_l[QG];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
      m_ref[qg] = 0.f;
      negm[qg] = floatx4{0.f, 0.f, 0.f, 0.f};
      acc_l[qg] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) acc_o[qg][dt] = floatx4{0.f, 0.f, 0.f, 0.f};
    }
    // S^T of one tile from the current -m_ref
    auto qk_tile = [&](int tile, floatx4 (&s)[QG][4]) __attribute__((always_inline)) {
      const char* kb = smem + (t0 + tile) * 2 * TILE;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          uint4 kf = *reinterpret_cast<const uint4*>(kb + k_off(kt * 16 + qi, ks * 4 + g));
#pragma unroll
          for (int qg = 0; qg < QG; ++qg) s[qg][kt] = mma<T>(kf, qf[qg][ks], ks == 0 ? negm[qg] : s[qg][kt]);
        }
    };
    // softmax of one tile's scores -> the 16-bit probabilities as PV operands; returns whether m_ref moved (wave-uniform)
    auto softmax_tile = [&](int tile, floatx4 (&s)[QG][4], uint4 (&pf)[QG][2]) __attribute__((always_inline)) -> bool {
      const bool ragged = (tile == nt - 1) && (S & (BKV - 1));
      const int key_base = tile * BKV + 4 * g;
      bool moved = false;
#pragma unroll
      for (int qg = 0; qg < QG; ++qg) {
        if (ragged) {
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[qg][kt][r] = (key_base + kt * 16 + r < S) ? s[qg][kt][r] : -INFINITY;
        }
        float mx = fmaxf(s[qg][0][0], s[qg][0][1]);
        mx = fmaxf(fmaxf(mx, s[qg][0][2]), s[qg][0][3]);
#pragma unroll
        for (int kt = 1; kt < 4; ++kt) {
          mx = fmaxf(fmaxf(mx, s[qg][kt][0]), s[qg][kt][1]);
          mx = fmaxf(fmaxf(mx, s[qg][kt][2]), s[qg][kt][3]);
        }
        if (tile == 0 || __any(mx > kDefer)) {
          moved = true;
          mx = max_over_query_lanes(mx);
          const float delta = tile == 0 ? mx : fmaxf(mx, 0.f);
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[qg][kt][r] -= delta;
          if (tile != 0) {
            const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc_l[qg][r] *= alpha;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc_o[qg][dt][r] *= alpha;
          }
          m_ref[qg] += delta;
          negm[qg] = floatx4{-m_ref[qg], -m_ref[qg], -m_ref[qg], -m_ref[qg]};
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[qg][kt][r] = __builtin_amdgcn_exp2f(s[qg][kt][r]);
#pragma unroll
        for (int kb2 = 0; kb2 < 2; ++kb2) {
          const floatx4 pa = s[qg][2 * kb2], pb = s[qg][2 * kb2 + 1];
          pf[qg][kb2] = uint4{pack2<T>(pa[0], pa[1]), pack2<T>(pa[2], pa[3]), pack2<T>(pb[0], pb[1]), pack2<T>(pb[2], pb[3])};
        }
      }
      return moved;
    };
    auto pv_tile = [&](int tile, const uint4 (&pf)[QG][2]) __attribute__((always_inline)) {
      const char* vb = smem + (t0 + tile) * 2 * TILE + TILE;
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const int col = dt * 16 + 4 * (qi & 3);
          const int r0 = (2 * kb2) * 16 + 4 * g + (qi >> 2), r1 = r0 + 16;
          typedef short4v __attribute__((address_space(3))) * lds_ptr;
          short4v va = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(vb + v_off(r0, col >> 3) + (col & 7) * 2));
          short4v vc = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(vb + v_off(r1, col >> 3) + (col & 7) * 2));
          uint2 lo = __builtin_bit_cast(uint2, va), hi = __builtin_bit_cast(uint2, vc);
          const uint4 vf = uint4{lo.x, lo.y, hi.x, hi.y};
#pragma unroll
          for (int qg = 0; qg < QG; ++qg) acc_o[qg][dt] = mma<T>(vf, pf[qg][kb2], acc_o[qg][dt]);
        }
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) acc_l[qg] = mma<T>(ones, pf[qg][kb2], acc_l[qg]);
    };
    if constexpr (PIPE) {
      floatx4 sa[QG][4], sb[QG][4];
      uint4 pf[QG][2];
      auto walk = [&](int tile, floatx4 (&cur)[QG][4], floatx4 (&nxt)[QG][4]) __attribute__((always_inline)) {
        const bool more = tile + 1 < nt;
        if (more && tile > 0) qk_tile(tile + 1, nxt);       // ahead of the exponentials of `tile` (the first tile always moves m_ref)
        const bool moved = softmax_tile(tile, cur, pf);
        if (more && (tile == 0 || moved)) qk_tile(tile + 1, nxt);       // m_ref moved: the product (again) from the new -m_ref
        pv_tile(tile, pf);
      };
      qk_tile(0, sa);
      for (int tile = 0; tile < nt; tile += 2) {           // two tiles per trip: the score sets trade places without copies
        walk(tile, sa, sb);
        if (tile + 1 >= nt) break;
        walk(tile + 1, sb, sa);
      }
    } else {
      for (int tile = 0; tile < nt; ++tile) {
        floatx4 s[QG][4];
        uint4 pf[QG][2];
        qk_tile(tile, s);
        softmax_tile(tile, s, pf);
        pv_tile(tile, pf);
      }
    }
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
      const float inv = 1.0f / acc_l[qg][0];
      const int qrow = q0 + qg * 16 + qi;
      store_rows16<T>(O + (static_cast<size_t>(b) * Tq + (qrow < Tq ? qrow : Tq - 1)) * ldo + h * HD, acc_o[qg], inv, g, qrow < Tq);
    }
  }
  }   // query blocks
  (void)MAXT;
}

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

}  // namespace

bool mfma_attention_supported(int dtype, const AttnArgs& a) {
  if (dtype != D3PM_F16 && dtype != D3PM_BF16) return false;
  if (a.hd != HD || a.S < 1 || a.Tq < 1) return false;
  if (a.ldq % 8 || a.ldkv % 8 || a.ldo % 8) return false;      // 16-byte output stores
  if (a.Q2 && !(a.S2 >= 1 && aligned(a.Q2, 16) && aligned(a.K2, 16) && aligned(a.V2, 16) && aligned(a.O2, 16))) return false;
  return aligned(a.Q, 16) && aligned(a.K, 16) && aligned(a.V, 16) && aligned(a.O, 16);
}

// d3pm_tuning.attn_query_groups: 0 = auto: two 16-query groups per wave (each K / V fragment read from LDS feeds two MFMAs;
// 63.8 vs 65.8 us on the 768 x 768 x 256-head self-attention, 27.7 vs 30.1 us on the 225-key prompt attention) once that still
// leaves >= 4 workgroups per CU, one group otherwise (a single utterance is 48 workgroups of two groups: latency regime).
// attn_pair_sequential: 0 never, 1 auto (when the paired grid still has >= 2 workgroups per CU), 2 always.
// attn_cross_resident: cross-attention pair with every K / V tile resident in LDS (attn_cross_hd64): 0 never, 1 auto, 2 always.
// Auto = when its grid (256 queries per workgroup) has at least one workgroup per CU: at one utterance it is 24 workgroups of
// eight waves and the tile-by-tile kernel's 192 small ones finish sooner (p50 latency 46.7 -> 42.2 ms,
// profiles/round2_d_latency_ab.txt)

int mfma_attention(int dtype, const AttnArgs& a, hipStream_t s) {
  const d3pm_tuning& tn = tune_of(a.tune);
  const int g_attn_cross_resident = tn.attn_cross_resident, g_attn_pair_seq = tn.attn_pair_sequential;
#ifdef D3PM_ABLATIONS
  const int g_attn_qg = (ab_knobs().attn_arm && ab_knobs().attn_arm < 300) ? ab_knobs().attn_arm : tn.attn_query_groups;      // arms >= 3: include/d3pm_hip_ab.h
#else
  const int g_attn_qg = (tn.attn_query_groups == 1 || tn.attn_query_groups == 2) ? tn.attn_query_groups : 0;
#endif
  // The 32 x 32 x 16 kernel of d3pm_mfma_attn32.hip wherever it applies (whole 128-query blocks and 64-key tiles of a single
  // problem: the self-attention of a DiT block) once its grid has two workgroups per CU (measured: 55.3 vs 65.3 us at 32
  // utterances x 768 frames, 19.5 vs 20.3 at 32 x 384, 28.8 vs 30.8 at 16 x 768; tests/ab_attn32.py); below that the 64-query
  // workgroups of the kernel below finish sooner.  attn_query_groups: 0 auto, 32 always, 33 always and the plain walk, 1 / 2 never.
  {
    // regime_batch: the batch the automatic choice is made for (a shard of a split batch takes the kernels of the whole batch)
    const long long wgs32 = static_cast<long long>(a.Tq / 128) * a.H * (tn.regime_batch > 0 ? tn.regime_batch : a.B);
    const int q = tn.attn_query_groups;
#ifdef D3PM_ABLATIONS
    const bool arm35 = q == 35;      // A/B library: the pipelined kernel with 192-query workgroups (d3pm_mfma_attn32.hip)
#else
    const bool arm35 = false;
#endif
    if ((q == 32 || q == 33 || arm35 || (q == 0 && wgs32 >= 512)) && mfma_attention32_supported(dtype, a)) return mfma_attention32(dtype, a, s);
  }
  const long long cross_wgs = static_cast<long long>((a.Tq + 255) / 256) * a.H * (tn.regime_batch > 0 ? tn.regime_batch : a.B);
  if ((g_attn_cross_resident >= 2 || (g_attn_cross_resident == 1 && cross_wgs >= 256)) && a.Q2 != nullptr && a.key_len == nullptr &&
      a.S <= BKV && a.S2 <= 4 * BKV) {
    const int n_qblocks = (a.Tq + 255) / 256;
    int n_qsplit = (256 + a.H * a.B - 1) / (a.H * a.B);            // workgroups per (utterance, head): enough to cover the CUs
    n_qsplit = n_qsplit < 1 ? 1 : n_qsplit > n_qblocks ? n_qblocks : n_qsplit;
    if (g_attn_cross_resident == 3) n_qsplit = n_qblocks;              // tuning: one query block per workgroup (the first form)
    // the same residency on the 32 x 32 x 16 instruction with the software-pipelined block of d3pm_mfma_attn32.hip (the shipped
    // form: 34.9 vs 35.8 us per pair in the microbenchmark, 43.1 vs 44.1 us per attention launch in the loop, tests/ab_cross32.py,
    // profiles/round3_r_ab_cross32.txt; it takes over at the batch size where the self-attention changes instruction shape too):
    // 5 always, 4 = the 16 x 16 x 32 form (bit-identical to the tile-by-tile kernel)
    constexpr bool kCross32 = true;
    if ((g_attn_cross_resident == 5 || (kCross32 && g_attn_cross_resident != 4)) && mfma_attention32_cross_supported(dtype, a))
      return mfma_attention32_cross(dtype, a, n_qsplit, s);
    const dim3 grid(static_cast<unsigned>(n_qsplit * a.H * a.B)), block(512);
    const size_t lds = static_cast<size_t>(1 + (a.S2 + BKV - 1) / BKV) * 2 * TILE;
#define D3PM_CROSS(T, PIPE)                                                                                                  \
    do {                                                                                                                     \
      D3PM_LDS_ATTR((&attn_cross_hd64<T, PIPE>), 5 * 2 * TILE);                                                              \
      attn_cross_hd64<T, PIPE><<<grid, block, lds, s>>>(static_cast<const T*>(a.Q), static_cast<const T*>(a.K),               \
          static_cast<const T*>(a.V), static_cast<T*>(a.O), a.S, static_cast<const T*>(a.Q2), static_cast<const T*>(a.K2),    \
          static_cast<const T*>(a.V2), static_cast<T*>(a.O2), a.S2, a.ldq, a.ldkv, a.ldo, a.Tq, a.scale, a.H, n_qblocks, n_qsplit); \
    } while (0)
#ifdef D3PM_ABLATIONS
    if (ab_knobs().attn_arm == 300) {      // A/B: the software-pipelined walk
      if (dtype == D3PM_F16) D3PM_CROSS(f16, true); else D3PM_CROSS(bf16, true);
    } else
#endif
    if (dtype == D3PM_F16) D3PM_CROSS(f16, false); else D3PM_CROSS(bf16, false);
#undef D3PM_CROSS
    D3PM_LAUNCH_CHECK();
    return D3PM_OK;
  }
  // Opt-in (attn_query_groups = 4): the key-split kernel of d3pm_mfma_attn_lat.hip -- four waves share 32 queries and take every
  // fourth key tile.  Measured at one utterance 0.8 ms of 38.8 faster (p50), at two utterances 2 ms slower (tests/ab_latency.py,
  // profiles/round3_o_ab_latency.txt): the launch is bound by the K / V bytes each workgroup streams through its CU (196 KB per
  // (head, query group) at 768 keys), not by the length of the tile chain, so it is not an automatic choice -- one utterance keeps
  // the schedule, and therefore the bits, of the batches up to ten.
  if (tn.attn_query_groups == 4 && mfma_attention_split_supported(dtype, a)) return mfma_attention_split(dtype, a, s);
  const long long wgs2 = static_cast<long long>((a.Tq + 127) / 128) * a.H * a.B * (a.Q2 ? 2 : 1);
  const int qg = g_attn_qg == 0 ? (wgs2 >= 4 * 256 ? 2 : 1) : g_attn_qg >= 100 ? 2 : (g_attn_qg == 3 && (dtype != D3PM_BF16 || a.Q2)) ? 2 : g_attn_qg, per_block = 64 * qg;
  const int n_qblocks = (a.Tq + per_block - 1) / per_block;
  const int n_blocks1 = n_qblocks * a.H * a.B;
  const bool seq = a.Q2 != nullptr && (g_attn_pair_seq == 2 || (g_attn_pair_seq == 1 && n_blocks1 >= 512));
  const int n_first = seq ? -1 : n_blocks1;
  dim3 grid(static_cast<unsigned>(n_blocks1) * ((a.Q2 && !seq) ? 2 : 1)), block(256);
#define D3PM_ATTN(T, QG, PAIR)                                                                                       \
  attn_mfma_hd64<T, QG, PAIR><<<grid, block, 0, s>>>(static_cast<const T*>(a.Q), a.ldq, static_cast<const T*>(a.K),  \
                                                     static_cast<const T*>(a.V), a.ldkv, static_cast<T*>(a.O), a.ldo, a.Tq, \
                                                     a.S, a.scale, a.H, n_qblocks, static_cast<const T*>(a.Q2),           \
                                                     static_cast<const T*>(a.K2), static_cast<const T*>(a.V2),           \
                                                     static_cast<T*>(a.O2), a.S2, n_first, a.key_len)
#define D3PM_ATTN_QG(T)                                                                    \
  do {                                                                                     \
    if (seq) { if (qg == 1) D3PM_ATTN(T, 1, true); else D3PM_ATTN(T, 2, true); }           \
    else { if (qg == 1) D3PM_ATTN(T, 1, false); else D3PM_ATTN(T, 2, false); }             \
  } while (0)
#ifdef D3PM_ABLATIONS
  if (g_attn_qg == 3 && dtype == D3PM_BF16 && !a.Q2) {        // three query groups per wave (A/B: fewer LDS reads per MFMA, two waves per SIMD)
    D3PM_ATTN(bf16, 3, false);
  } else if (g_attn_qg >= 200 && dtype == D3PM_BF16 && !a.Q2) {      // occupancy probe: the shipped QG = 2 kernel with idle dynamic LDS
    const size_t pad = g_attn_qg == 201 ? 32 * 1024 : 96 * 1024;   // 201: two workgroups per CU, 202: one (three without)
    D3PM_LDS_ATTR((&attn_mfma_hd64<bf16, 2, false, 0>), 96 * 1024);
    attn_mfma_hd64<bf16, 2, false, 0><<<grid, block, pad, s>>>(static_cast<const bf16*>(a.Q), a.ldq, static_cast<const bf16*>(a.K),
        static_cast<const bf16*>(a.V), a.ldkv, static_cast<bf16*>(a.O), a.ldo, a.Tq, a.S, a.scale, a.H, n_qblocks, nullptr, nullptr, nullptr,
        nullptr, 0, n_first, a.key_len);
  } else if (g_attn_qg >= 100 && dtype == D3PM_BF16 && !a.Q2) {      // timing-only ablations of the QG = 2 kernel
#define D3PM_ABL(A) case A: attn_mfma_hd64<bf16, 2, false, A><<<grid, block, 0, s>>>(static_cast<const bf16*>(a.Q), a.ldq, static_cast<const bf16*>(a.K), \
      static_cast<const bf16*>(a.V), a.ldkv, static_cast<bf16*>(a.O), a.ldo, a.Tq, a.S, a.scale, a.H, n_qblocks, nullptr, nullptr, nullptr, nullptr, 0, n_first, a.key_len); break
    switch (g_attn_qg - 100) {
      D3PM_ABL(64); D3PM_ABL(128); D3PM_ABL(1); D3PM_ABL(2); D3PM_ABL(3); D3PM_ABL(4); D3PM_ABL(12); D3PM_ABL(16); D3PM_ABL(32); D3PM_ABL(48); D3PM_ABL(15); D3PM_ABL(60);
      default: D3PM_ATTN(bf16, 2, false);
    }
#undef D3PM_ABL
  } else
#endif
  if (dtype == D3PM_F16) D3PM_ATTN_QG(f16); else D3PM_ATTN_QG(bf16);
#undef D3PM_ATTN_QG
#undef D3PM_ATTN
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

}  // namespace d3pm
