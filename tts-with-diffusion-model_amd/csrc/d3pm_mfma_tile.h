// d3pm_mfma_tile.h -- pieces shared by the MFMA GEMM kernels (d3pm_mfma_gemm.hip, d3pm_mfma_gemm_big.hip):
// fragment types, the swizzled 128-byte-row LDS image, the direct-to-LDS DMA helpers and the epilogue that turns
// fp32 accumulators into bias / activation / residual / mask -ed 16-byte stores with the eager rounding points.
#pragma once
#include <type_traits>

#include "d3pm_kernels.h"

namespace d3pm {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int BK = 64;                                  // k-step: one 128-byte LDS row per tile row
constexpr int ROW_BYTES = BK * 2;

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * ROW_BYTES + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <typename T> __device__ __forceinline__ floatx4 mma(uint4 a, uint4 b, floatx4 c);
template <> __device__ __forceinline__ floatx4 mma<f16>(uint4 a, uint4 b, floatx4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ floatx4 mma<bf16>(uint4 a, uint4 b, floatx4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// erf for the GELU epilogue: erf(x) = sign(x) (1 - 2^(-t P(t))), t = min(|x|, 3.95), P a degree-8 fit of
// -log2(erfc(t))/t (fit error 2e-9, fp32 evaluation within 1.4 * 2^-24 absolute of the exact erf on [-6, 6]).  One
// straight-line chain of 8 FMAs and one v_exp_f32 instead of libm's two-branch erff: the fc1 epilogue was VALU-bound
// on erff (2500 of its 3400 vector instructions per wave, 60 of the 95 us of the kernel at 24576 x 2048).  What GELU
// needs is absolute accuracy of erf at the 2^-24 grid -- the reference's own 1 + erf(x) cancels against that grid in
// the negative tail -- and there this form is as close to torch's CPU gelu as torch's is to the exact function
// (tools/fit_erf.py: fp16 7.6e-4 / bf16 9e-5 of N(0,1) inputs round differently, torch-vs-exact 6.2e-4 / 9e-5).
// The generic (FMA) family keeps libm's erff: it is the numerical specification and the bit-exact native-shape path.
__device__ __forceinline__ float erf_fit(float x) {
  const float t = fminf(fabsf(x), 3.95f);
  float p = -1.1604810424614698e-05f;
  p = __builtin_fmaf(p, t, 0.00015296436322387308f);
  p = __builtin_fmaf(p, t, -0.000848234398290515f);
  p = __builtin_fmaf(p, t, 0.0022747856564819813f);
  p = __builtin_fmaf(p, t, -8.480551332468167e-05f);
  p = __builtin_fmaf(p, t, -0.027724476531147957f);
  p = __builtin_fmaf(p, t, 0.1483079046010971f);
  p = __builtin_fmaf(p, t, 0.9184429049491882f);
  p = __builtin_fmaf(p, t, 1.6279072761535645f);
  const float e = __builtin_amdgcn_exp2f(-(p * t));
  return __builtin_copysignf(1.0f - e, x);
}
__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erf_fit(v * 0.70710678118654752f)); }

// GELU of a bf16 value is a function of 16 bits: for 2^-14 <= |v| < 8 (bf16 exponent fields 113 .. 129: every value an
// activation takes in practice) the result rn_bf16(gelu_erf(v)) is read from an 8.5 KiB table in LDS -- ~8 vector
// instructions and one ds_read_u16 instead of the ~25 of the polynomial (the GELU epilogue is VALU-bound: 96 elements per
// lane and tile).  The table is filled on the device by gelu_erf itself (gelu_table_device, d3pm_mfma_gemm.hip), so the
// lookup returns bit for bit what the arithmetic path returns; values outside the range take the arithmetic path.
constexpr int GELU_TAB_E0 = 113, GELU_TAB_NE = 17;
constexpr int GELU_TAB_ENTRIES = 2 * GELU_TAB_NE * 128;       // [sign][exponent][mantissa]
constexpr int GELU_TAB_BYTES = GELU_TAB_ENTRIES * 2;          // 8704
__device__ __forceinline__ float gelu_bf16_lookup(float v, const uint16_t* tab) {
  const uint32_t b = __float_as_uint(v) >> 16;                // v is a bf16 value held in a float
  const uint32_t e = ((b >> 7) & 0xFFu) - GELU_TAB_E0;
  if (e < static_cast<uint32_t>(GELU_TAB_NE))
    return __uint_as_float(static_cast<uint32_t>(tab[e * 128u + (b & 127u) + (b >> 15) * (GELU_TAB_NE * 128u)]) << 16);
  return rn<bf16>(gelu_erf(v));
}
// workgroup-wide copy of the table into LDS (caller synchronises before the first lookup)
__device__ __forceinline__ void gelu_table_to_lds(const uint16_t* __restrict__ gtab, char* lds, int tid, int nthreads) {
  for (int i = tid; i < GELU_TAB_BYTES / 16; i += nthreads)
    reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(gtab)[i];
}

// Epilogue.  The MFMA leaves D[n = nt*16 + (lane>>4)*4 + r][m = mt*16 + (lane&15)] in a lane: 4 consecutive
// columns of one row per (nt, mt), i.e. 8-byte stores that touch 32 contiguous bytes per row and instruction.
// v_permlane16_swap between the accumulators of column blocks nt and nt+1 (odd 16-lane rows of the first operand
// trade places with the even rows of the second) regroups them so that a lane owns 8 consecutive columns
//   n = (nt + (g&1))*16 + (g>>1)*8 .. +7,   g = lane>>4,
// after bias + activation (elementwise, so the order does not matter): residuals and the output then move as
// 16-byte accesses, half the store instructions and 64 contiguous bytes per row and instruction (the store tail is
// issue-bound, MI355X_MICROARCH.md constants table).  Values and rounding points are untouched -- the swap only
// changes which lane finishes which element.
// The swap is inline asm on purpose: with hipcc 7.2 the two-result __builtin_amdgcn_permlane16_swap loses its second
// result once the operands are floats in an unrolled loop (both halves read the first output register; seen in the
// .s of a 10-line kernel and as 50 % wrong columns on the GPU).  `s_nop 1` covers the VALU-write -> permlane-read
// hazard the compiler cannot see inside the string; the operands are always produced by the bias add (VALU), never
// directly by an MFMA.
// Residual loads are branch-free (clamped addresses) and batched per row; only the stores are predicated.
// EPI bits: 1 = exact-erf GELU, 2 = one residual (R1), 4 = two residuals (R1, R2), 8 = frame mask, 16 = ReLU, 32 = SiLU
//           64 = LayerNorm folded into this projection (EPI_LNF), 128 = row moments of the output rows (EPI_STATS): below
constexpr int EPI_GELU = 1, EPI_R1 = 2, EPI_R2 = 4, EPI_MASK = 8, EPI_RELU = 16, EPI_SILU = 32, EPI_LNF = 64, EPI_STATS = 128;

// LayerNorm taken out of the launch list algebraically (ar_discrete.py:131-132, 136-142, 153-159: LayerNorm -> Linear).  With
// W' = W o gamma (rounded to the storage type once, at weight-preparation time), s_n = sum_k W'[n][k] and
// b'_n = sum_k W[n][k] beta_k + b_n,
//     LN(x) W^T + b  =  rstd_r (x_r . W'^T  -  mean_r s)  +  b'
// so the projection that CONSUMES a LayerNorm reads the raw residual stream through the unchanged operand path and applies two
// per-row scalars and two per-column vectors in its epilogue (EPI_LNF), and the projection that PRODUCES the residual rows
// (out-projection / fc2 + residual) leaves their moments behind (EPI_STATS): per row and per 32-column part of the row the pair
// (sum y, sum y^2) of the ROUNDED values it stores, fp32, no atomics.  Layout (stats_index): [M / 16][N / 32][16][2] -- the 16 rows
// of an MFMA row block side by side, so that the store of one part by one wave is one whole 128-byte line (row-major it was 16
// partial lines per store instruction and cost 2.5-3 us per producing launch).  A part is what one 16-lane row group
// of any MFMA kernel's epilogue owns after the column regrouping (8 columns in each of 4 lanes), so every tile geometry sums a
// part in the same order and writes the same bits; the consumer adds a row's parts in one fixed order (four lanes x parts_in / 4
// parts each, then two butterfly steps).  var = E[y^2] - mean^2 in fp32: with |mean| <~ 10 sigma the relative error of rstd stays
// below 1e-5, two orders under the 16-bit storage quantum.  This skips the 16-bit rounding of the LayerNorm output (closer to the
// fp32 model than the eager chain); the F32 / generic family keeps the stand-alone LayerNorm.
struct EpiFold {
  const float* s = nullptr;          // [N]  sum_k W'[n][k]
  const float* b = nullptr;          // [N]  folded bias b'
  const float* stats_in = nullptr;   // [M][parts_in][2]  moments of the operand rows (the residual stream, K columns)
  float* stats_out = nullptr;        // [M][N / 32][2]    moments of the rows this launch stores (N = d_model)
  float eps = 1e-6f;
  int parts_in = 0;                  // K / 32, a multiple of 8
};

// float index of (row, part) in a moments buffer of `parts` parts per row: [row / 16][part][row % 16][2]
__host__ __device__ __forceinline__ size_t stats_index(size_t row, int part, int parts) {
  return (((row >> 4) * parts + part) * 16 + (row & 15)) * 2;
}

inline EpiFold epi_fold_of(const LinearArgs& a) {
  EpiFold e;
  e.s = a.fold_s; e.b = a.fold_b; e.stats_in = a.stats_in; e.stats_out = a.stats_out; e.eps = a.fold_eps; e.parts_in = a.K / 32;
  return e;
}

// (sum, sum of squares) of row `row` from its parts: lane group g = lane >> 4 adds parts [g P / 4, (g + 1) P / 4) in order, then the
// four groups are combined by two butterfly steps (a + b == b + a bit for bit, so all four lanes end with the same bits)
__device__ __forceinline__ void fold_row_moments(const float* __restrict__ stats, int parts, size_t row, int g, float& s1, float& s2) {
  typedef float float2v __attribute__((ext_vector_type(2)));
  const int per = parts >> 2;
  float a = 0.f, q = 0.f;
  for (int i = 0; i < per; ++i) {
    const float2v v = *reinterpret_cast<const float2v*>(stats + stats_index(row, g * per + i, parts));
    a += v[0]; q += v[1];
  }
  a = add_xor16(a); q = add_xor16(q);
  s1 = add_xor32(a); s2 = add_xor32(q);
}
// the in-lane half of a part's moments: eight consecutive columns.  a: pairwise tree; q: one fused chain (v7^2 first)
__device__ __forceinline__ void part_moments8(const float (&v)[8], float& a, float& q) {
  a = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  q = v[7] * v[7];
#pragma unroll
  for (int i = 6; i >= 0; --i) q = __builtin_fmaf(v[i], v[i], q);
}

// EPI_STATS: (sum, sum of squares) of the 32 stored values of row m in the 32-column part starting at column `col0` -- eight in this
// lane, the other 24 in the lanes 16 / 32 / 48 away.  The order below is the definition of a part's moments for every kernel
// (and of d3pm_op_row_stats, d3pm_fold.hip).  Executed by the whole wave (lane permutes); `ok` only predicates the store.
__device__ __forceinline__ void part_stats_store(const float (&v)[8], float* __restrict__ stats_out, size_t m, int N, int col0, int g, bool ok) {
  float a, q;
  part_moments8(v, a, q);
  // one butterfly for both moments: after the 16-lane swap the even lane rows hold sums of `a`, the odd ones sums of `q`
  //   rows (a0, a1, a2, a3) | (q0, q1, q2, q3)  --swap16-->  (a0, q0, a2, q2) | (a1, q1, a3, q3)  --add-->  (a01, q01, a23, q23)
  // and after the 32-lane swap of two copies of that: (a01 + a23, q01 + q23, ..): lane group 0 ends with sum, group 1 with sum of squares
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(q));
  float z = a + q, z2 = z;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(z), "+v"(z2));
  z += z2;
  if (g < 2 && ok) stats_out[stats_index(m, col0 >> 5, N >> 5) + g] = z;
}

// per-row scalars of EPI_LNF computed ahead of the epilogue (big-tile GEMM: the moments are requested at the top of a tile and
// reduced under the k-loop, so that the epilogue does not start with a dependent round trip to memory)
template <int MT> struct RowScalars { float ra[MT], rc[MT]; };

// the two per-row scalars of EPI_LNF: y = fma(acc, ra, fma(rc, s_n, b'_n)) with ra = rstd, rc = -mean rstd
__device__ __forceinline__ void fold_row_scalars(float s1, float s2, int d, float eps, float& ra, float& rc) {
  const float inv_d = 1.0f / static_cast<float>(d);
  const float mean = s1 * inv_d;
  const float var = fmaxf(__builtin_fmaf(-mean, mean, s2 * inv_d), 0.f);
  ra = rsqrtf(var + eps);
  rc = -mean * ra;
}

typedef uint32_t uintx4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ uint32_t pack2(float a, float b) {
  typedef T pair __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, pair{static_cast<T>(a), static_cast<T>(b)});
}
template <typename T> struct alignas(8) Pack4 { T v[4]; };
template <typename T> struct alignas(16) Pack8 { T v[8]; };

__device__ __forceinline__ void swap_rows16(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

// kPack (interior tiles only): the MT * NT / 2 finished 16-byte groups go to `packed[mt * (NT / 2) + np]` instead of memory;
// the caller stores group (mt, np) later at Y + (mw0 + (lane & 15) + 16 mt) * ldy + nw0 + epilogue_nq(lane) + 32 np.
__device__ __forceinline__ int epilogue_nq(int lane) { const int g = lane >> 4; return (g & 1) * 16 + (g >> 1) * 8; }

// Operands of the epilogue fetched ahead of the main loop (latency GEMM: at one utterance a launch is a chain of dependent round
// trips, and bias -> residual -> mask behind the last MFMA were three of them).  Addresses are clamped exactly as the loads inside
// epilogue_store clamp them, so the prefetched values are the ones it would have loaded, for interior and ragged tiles alike:
// same values, same arithmetic -- only when the loads are issued changes.
template <typename T, int NT, int MT> struct EpiPre {
  float bv[NT][4];
  Pack8<T> r1[MT][NT / 2], r2[MT][NT / 2];
  float mk[MT];
  float sv[NT][4];             // EPI_LNF: s_n beside b'_n (bv)
  floatx4 st[MT][2];           // EPI_LNF, parts_in == 16 (d_model = 512): the lane's four parts of each of its rows, not yet reduced
};
template <typename T, int EPI, int NT, int MT>
__device__ __forceinline__ void epilogue_prefetch(EpiPre<T, NT, MT>& pre, const T* __restrict__ bias, const T* R1, const T* R2, int ldr,
                                                  const uint8_t* __restrict__ row_mask, int mask_period, int M, int N, int mw0,
                                                  int nw0, int lane, const EpiFold* ef = nullptr) {
  constexpr bool kR1 = (EPI & (EPI_R1 | EPI_R2)) != 0, kR2 = (EPI & EPI_R2) != 0, kMask = (EPI & EPI_MASK) != 0;
  constexpr bool kLnf = (EPI & EPI_LNF) != 0;
  constexpr int NP = NT / 2;
  const int g = lane >> 4, nq = (g & 1) * 16 + (g >> 1) * 8;
  if constexpr (kLnf) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      int n = nw0 + nt * 16 + g * 4;
      n = n < N ? n : N - 4;
      const floatx4 b4 = *reinterpret_cast<const floatx4*>(ef->b + n), s4 = *reinterpret_cast<const floatx4*>(ef->s + n);
#pragma unroll
      for (int r = 0; r < 4; ++r) { pre.bv[nt][r] = b4[r]; pre.sv[nt][r] = s4[r]; }
    }
    if (ef->parts_in == 16) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        typedef float float2v __attribute__((ext_vector_type(2)));
        const int m = mw0 + mt * 16 + (lane & 15);
        const float* p = ef->stats_in + stats_index(static_cast<size_t>(m < M ? m : M - 1), g * 4, 16);      // parts 4g .. 4g + 3: 32 floats apart
        const float2v p0 = *reinterpret_cast<const float2v*>(p), p1 = *reinterpret_cast<const float2v*>(p + 32),
                      p2 = *reinterpret_cast<const float2v*>(p + 64), p3 = *reinterpret_cast<const float2v*>(p + 96);
        pre.st[mt][0] = floatx4{p0[0], p0[1], p1[0], p1[1]};
        pre.st[mt][1] = floatx4{p2[0], p2[1], p3[0], p3[1]};
      }
    }
  } else if (bias == nullptr) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) pre.bv[nt][r] = 0.f;
  } else if ((N & 3) == 0 && (reinterpret_cast<uintptr_t>(bias) & 7) == 0) {
    Pack4<T> pb[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = nw0 + nt * 16 + g * 4;
      pb[nt] = *reinterpret_cast<const Pack4<T>*>(bias + (n < N ? n : N - 4));
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) pre.bv[nt][r] = static_cast<float>(pb[nt].v[r]);
  } else {
    T sb[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nw0 + nt * 16 + g * 4 + r;
        sb[nt][r] = bias[n < N ? n : N - 1];
      }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) pre.bv[nt][r] = static_cast<float>(sb[nt][r]);
  }
  const int n_last = N >= 8 ? N - 8 : 0;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mw0 + mt * 16 + (lane & 15);
    const int mc = m < M ? m : M - 1;
    pre.mk[mt] = 1.f;
    if constexpr (kMask) pre.mk[mt] = row_mask[mc % mask_period] ? 1.f : 0.f;
    if constexpr (kR1) {
#pragma unroll
      for (int np = 0; np < NP; ++np) {
        int nc = nw0 + np * 32 + nq;
        nc = nc < n_last ? nc : n_last;
        pre.r1[mt][np] = *reinterpret_cast<const Pack8<T>*>(R1 + static_cast<size_t>(mc) * ldr + nc);
        if constexpr (kR2) pre.r2[mt][np] = *reinterpret_cast<const Pack8<T>*>(R2 + static_cast<size_t>(mc) * ldr + nc);
      }
    }
  }
}

template <typename T, int EPI, int NT = 4, int MT = 4, bool kInteriorOnly = false, bool kPack = false, bool kNts = false, bool kPre = false>
__device__ __forceinline__ void epilogue_store(floatx4 (&acc)[NT][MT], const T* __restrict__ bias, T* Y, int ldy,
                                               const T* R1, const T* R2, int ldr, const uint8_t* __restrict__ row_mask,
                                               int mask_period, int M, int N, int mw0, int nw0, int lane,
                                               uintx4* packed = nullptr, const uint16_t* gelu_tab = nullptr,
                                               const EpiPre<T, NT, MT>* pre = nullptr,      // pre: read only when kPre
                                               const EpiFold* ef = nullptr,                 // ef: read only under EPI_LNF / EPI_STATS
                                               const RowScalars<MT>* rows = nullptr) {      // EPI_LNF: the row scalars, when the caller has them
  static_assert(!kPack || kInteriorOnly, "packing to registers is for whole tiles");
  static_assert(NT % 2 == 0, "column blocks are regrouped in pairs");
  constexpr bool kGelu = EPI & EPI_GELU, kR1 = (EPI & (EPI_R1 | EPI_R2)) != 0, kR2 = (EPI & EPI_R2) != 0, kMask = (EPI & EPI_MASK) != 0;
  constexpr bool kLnf = (EPI & EPI_LNF) != 0, kStats = (EPI & EPI_STATS) != 0;
  static_assert(!(kLnf && (kR1 || kMask)) && !(kStats && kPack), "instantiated: LNF [+ GELU]; R1 / R2 / R1 + mask [+ STATS]");
  constexpr int NP = NT / 2;
  const int g = lane >> 4;
  const int nq = (g & 1) * 16 + (g >> 1) * 8;           // column of this lane's 8-group inside a 32-column pair
  // bias in the MFMA layout (applied before the regrouping).  Loads are branch-free per lane -- clamped addresses,
  // one batch, one wait: per-element predicated loads compile to sixteen serialized L2 round trips.
  float bv[NT][4];
  [[maybe_unused]] float sv[NT][4], ra[MT], rc[MT];
  if constexpr (kLnf) {
    // per-column vectors b'_n, s_n (fp32) and the two per-row scalars from the producer's partial moments
    if constexpr (kPre) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { bv[nt][r] = pre->bv[nt][r]; sv[nt][r] = pre->sv[nt][r]; }
    } else {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        int n = nw0 + nt * 16 + g * 4;
        n = n < N ? n : N - 4;
        const floatx4 b4 = *reinterpret_cast<const floatx4*>(ef->b + n), s4 = *reinterpret_cast<const floatx4*>(ef->s + n);
#pragma unroll
        for (int r = 0; r < 4; ++r) { bv[nt][r] = b4[r]; sv[nt][r] = s4[r]; }
      }
    }
    const int d_in = ef->parts_in * 32;
    if (rows != nullptr) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) { ra[mt] = rows->ra[mt]; rc[mt] = rows->rc[mt]; }
    } else if (kPre && ef->parts_in == 16) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const floatx4 p0 = pre->st[mt][0], p1 = pre->st[mt][1];
        float a = 0.f, q = 0.f;
        a += p0[0]; q += p0[1]; a += p0[2]; q += p0[3];
        a += p1[0]; q += p1[1]; a += p1[2]; q += p1[3];
        a = add_xor16(a); q = add_xor16(q);
        fold_row_scalars(add_xor32(a), add_xor32(q), d_in, ef->eps, ra[mt], rc[mt]);
      }
    } else {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = mw0 + mt * 16 + (lane & 15);
        float s1, s2;
        fold_row_moments(ef->stats_in, ef->parts_in, static_cast<size_t>(m < M ? m : M - 1), g, s1, s2);
        fold_row_scalars(s1, s2, d_in, ef->eps, ra[mt], rc[mt]);
      }
    }
  } else if constexpr (kPre) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[nt][r] = pre->bv[nt][r];
  } else if (bias == nullptr) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[nt][r] = 0.f;
  } else if ((N & 3) == 0 && (reinterpret_cast<uintptr_t>(bias) & 7) == 0) {
    Pack4<T> pb[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int n = nw0 + nt * 16 + g * 4;
      pb[nt] = *reinterpret_cast<const Pack4<T>*>(bias + (n < N ? n : N - 4));
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[nt][r] = static_cast<float>(pb[nt].v[r]);   // columns >= N are never stored
  } else {
    T sb[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nw0 + nt * 16 + g * 4 + r;
        sb[nt][r] = bias[n < N ? n : N - 1];
      }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[nt][r] = static_cast<float>(sb[nt][r]);
  }
  auto finish = [&](float (&v)[8], int np, int mt) {   // bias + activation in the MFMA layout, then the regrouping
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      if constexpr (kLnf)
        v[r] = rn<T>(__builtin_fmaf(acc[2 * np + (r >> 2)][mt][r & 3], ra[mt], __builtin_fmaf(rc[mt], sv[2 * np + (r >> 2)][r & 3], bv[2 * np + (r >> 2)][r & 3])));
      else
        v[r] = rn<T>(acc[2 * np + (r >> 2)][mt][r & 3] + bv[2 * np + (r >> 2)][r & 3]);
      if constexpr (kGelu) {
        if constexpr (std::is_same<T, bf16>::value) v[r] = gelu_tab ? gelu_bf16_lookup(v[r], gelu_tab) : rn<T>(gelu_erf(v[r]));
        else v[r] = rn<T>(gelu_erf(v[r]));
      }
      if (EPI & EPI_RELU) v[r] = fmaxf(v[r], 0.f);
      if (EPI & EPI_SILU) v[r] = rn<T>(v[r] / (1.0f + expf(-v[r])));
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) swap_rows16(v[r], v[4 + r]);
  };
  auto emit_stats = [&](const float (&v)[8], int m, int np, bool ok) {
    part_stats_store(v, ef->stats_out, static_cast<size_t>(m), N, nw0 + np * 32, g, ok);
  };
  if (kInteriorOnly || (mw0 + MT * 16 <= M && nw0 + NT * 16 <= N)) {
    // interior wave tile (wave-uniform test): no clamps or predicates, one 64-bit row pointer per operand that
    // advances by 16 rows, column offsets as instruction immediates
    const int m = mw0 + (lane & 15);
    T* y = Y + static_cast<size_t>(m) * ldy + nw0 + nq;
    const T* r1 = kR1 ? R1 + static_cast<size_t>(m) * ldr + nw0 + nq : nullptr;
    const T* r2 = kR2 ? R2 + static_cast<size_t>(m) * ldr + nw0 + nq : nullptr;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      float mk = 1.f;
      if constexpr (kPre) mk = pre->mk[mt];
      else if constexpr (kMask) mk = row_mask[(m + mt * 16) % mask_period] ? 1.f : 0.f;
      Pack8<T> p1[NP], p2[NP];
      if (kR1) {
#pragma unroll
        for (int np = 0; np < NP; ++np) {
          if constexpr (kPre) {
            p1[np] = pre->r1[mt][np];
            if (kR2) p2[np] = pre->r2[mt][np];
          } else {
            p1[np] = *reinterpret_cast<const Pack8<T>*>(r1 + np * 32);
            if (kR2) p2[np] = *reinterpret_cast<const Pack8<T>*>(r2 + np * 32);
          }
        }
      }
#pragma unroll
      for (int np = 0; np < NP; ++np) {
        float v[8];
        finish(v, np, mt);
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          if (kR1) {
            float res = static_cast<float>(p1[np].v[r]);
            if (kR2) res = rn<T>(res + static_cast<float>(p2[np].v[r]));
            v[r] = rn<T>(res + v[r]);
          }
          if (kMask) v[r] *= mk;
        }
        if constexpr (kStats) emit_stats(v, m + mt * 16, np, true);
        const uintx4 grp = uintx4{pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]), pack2<T>(v[4], v[5]), pack2<T>(v[6], v[7])};
        if constexpr (kPack) packed[mt * NP + np] = grp;
        else if constexpr (kNts) {
          // streaming output that must not displace the operand panels in this XCD's L2: an `sc1` store leaves no line behind
          // (MI355X_MICROARCH.md, stores of each flavour); from asm, so the caller's hand-counted vmcnt waits cover it
          T* yp = y + np * 32;
          // (`s_nop 1`: on gfx940+ a store of more than 64 bits needs two wait states before its data registers are written again, and hipcc
          // does not see inside the asm statement)
          asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(yp), "v"(grp) : "memory");
        }
        else *reinterpret_cast<uintx4*>(y + np * 32) = grp;
      }
      y += static_cast<size_t>(16) * ldy;
      if (kR1) r1 += static_cast<size_t>(16) * ldr;
      if (kR2) r2 += static_cast<size_t>(16) * ldr;
    }
    return;
  }
  if constexpr (kInteriorOnly) return;   // the persistent schedule is only ever launched over whole tiles
  const int n_last = N >= 8 ? N - 8 : 0;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = mw0 + mt * 16 + (lane & 15);
    const int mc = m < M ? m : M - 1;
    float mk = 1.f;
    if constexpr (kPre) mk = pre->mk[mt];
    else if constexpr (kMask) mk = row_mask[mc % mask_period] ? 1.f : 0.f;
    Pack8<T> p1[NP], p2[NP];
    if (kR1) {   // N % 8 == 0 is guaranteed by mfma_linear_supported when a residual is given
#pragma unroll
      for (int np = 0; np < NP; ++np) {
        if constexpr (kPre) {
          p1[np] = pre->r1[mt][np];
          if (kR2) p2[np] = pre->r2[mt][np];
        } else {
          int nc = nw0 + np * 32 + nq;
          nc = nc < n_last ? nc : n_last;
          p1[np] = *reinterpret_cast<const Pack8<T>*>(R1 + static_cast<size_t>(mc) * ldr + nc);
          if (kR2) p2[np] = *reinterpret_cast<const Pack8<T>*>(R2 + static_cast<size_t>(mc) * ldr + nc);
        }
      }
    }
#pragma unroll
    for (int np = 0; np < NP; ++np) {
      const int n = nw0 + np * 32 + nq;
      float v[8];
      finish(v, np, mt);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        if (kR1) {
          float res = static_cast<float>(p1[np].v[r]);
          if (kR2) res = rn<T>(res + static_cast<float>(p2[np].v[r]));
          v[r] = rn<T>(res + v[r]);
        }
      }
      if constexpr (kStats) {          // N is a multiple of 32 (launcher): a part is stored whole or not at all
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] *= mk;
        emit_stats(v, mc, np, m < M && nw0 + np * 32 < N);
      }
      if (m < M && n < N) {
        T* y = Y + static_cast<size_t>(m) * ldy + n;
        if (n + 7 < N) {
          Pack8<T> o;
#pragma unroll
          for (int r = 0; r < 8; ++r) o.v[r] = static_cast<T>(v[r] * mk);
          *reinterpret_cast<Pack8<T>*>(y) = o;
        } else {
          for (int r = 0; r < 8 && n + r < N; ++r) y[r] = static_cast<T>(v[r] * mk);
        }
      }
    }
  }
}

typedef __attribute__((address_space(3))) void* lds_void;
typedef const __attribute__((address_space(1))) void* glb_void;

// hipcc cannot tell an in-flight LDS-DMA from the LDS reads of the tile being computed and drains it
// (vmcnt(0)) before the first ds_read; issuing the DMA from asm keeps it out of the compiler's counters, so
// the wait is placed by hand: counted vmcnt (8 DMAs per wave per tile stay in flight), raw s_barrier.
__device__ __forceinline__ void glds16_asm(const void* gsrc, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(lds_dst)
               : "memory");
}

__device__ __forceinline__ void glds16_asm_s(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}

}  // namespace
}  // namespace d3pm
