// d3pm_fp8.hip -- the "fast" configuration of BASELINE.json configs[4]: fp8 (OCP e4m3) operands for the two largest
// K = d_model projections of a DiT block -- the QKV in-projection and fc1 (ar_discrete.py:132,159) -- whose input is a
// LayerNorm output.  No reference oracle exists for this mode (the reference is fp16 only); tests pin the kernels to a
// torch fp32 evaluation of the SAME quantised operands and report token agreement against the bf16 path.
//
//   layernorm_fp8_rows   LayerNorm (+ FiLM) exactly as layernorm_vec up to the 16-bit result, then one scale per row
//                        (absmax / 448) and e4m3 codes: X8 [M][512] bytes + sx [M] fp32
//   gemm_fp8_persist     Y = epilogue((X8 . W8^T) * sx[m] * sw[n] + bias): the persistent 128 x 128 schedule of
//                        d3pm_mfma_gemm.hip with a K-step of 128 one-byte elements (the same 128-byte LDS rows, XOR
//                        swizzle and LDS-DMA pattern), v_mfma_f32_16x16x32_fp8_fp8, fp32 accumulation.  Half the
//                        L2 -> LDS bytes, DMA instructions and barriers per flop of the 16-bit kernel, which is bound
//                        by exactly those.  A 16-byte LDS chunk feeds TWO MFMAs (its low and high
//                        8 bytes), i.e. the contraction index is permuted identically on both operands.
#include "d3pm_kernels.h"

namespace d3pm {
namespace {

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef uint32_t uintx4 __attribute__((ext_vector_type(4)));
constexpr int BM = 128, BN = 128, BK8 = 128;            // K-step in fp8 elements = 128 B rows
constexpr int ROW_BYTES = 128, TILE_BYTES = BM * ROW_BYTES;
constexpr float kFp8Max = 448.0f;

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * ROW_BYTES + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ void glds16_asm_s(const void* sbase, uint32_t voff, uint32_t lds_dst) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}
__device__ __forceinline__ void swap_rows16(float& a, float& b) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
template <typename T> __device__ __forceinline__ uint32_t pack2(float a, float b) {
  typedef T pair __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, pair{static_cast<T>(a), static_cast<T>(b)});
}
// same fitted erf as the 16-bit MFMA epilogue (d3pm_mfma_gemm.hip)
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

// ---- LayerNorm -> e4m3 row -------------------------------------------------------------------------------------
template <typename T> struct alignas(16) Vec8 { T v[8]; };

__device__ __forceinline__ void store_fp8_row(const float (&o)[8], uint8_t* dst, float* scale_out, int lane) {
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(o[i]));
  amax = wave_max(amax);
  const float scale = amax > 0.f ? amax / kFp8Max : 1.0f, inv = 1.0f / scale;
  uint32_t lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(o[0] * inv, o[1] * inv, lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(o[2] * inv, o[3] * inv, lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(o[4] * inv, o[5] * inv, hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(o[6] * inv, o[7] * inv, hi, true);
  *reinterpret_cast<uint2*>(dst) = uint2{lo, hi};
  if (lane == 0) *scale_out = scale;
}

// optional second LayerNorm of the same rows (w2, b2 -> y8_2, sx_2), as layernorm_vec's dual output (norm2 | norm22)
template <typename T>
__global__ __launch_bounds__(256) void layernorm_fp8_rows(const T* __restrict__ x, uint8_t* __restrict__ y8, float* __restrict__ sx,
                                                          const T* __restrict__ w, const T* __restrict__ b,
                                                          const T* __restrict__ film, const T* __restrict__ w2,
                                                          const T* __restrict__ b2, uint8_t* __restrict__ y8_2,
                                                          float* __restrict__ sx_2, int M, float eps) {
  constexpr int d = 512;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
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
  store_fp8_row(o, y8 + static_cast<size_t>(row) * d + col, sx + row, lane);
  if (y8_2) {
    const Vec8<T> w2v = *reinterpret_cast<const Vec8<T>*>(w2 + col), b2v = *reinterpret_cast<const Vec8<T>*>(b2 + col);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = rn<T>((v[i] - mean) * rstd * static_cast<float>(w2v.v[i]) + static_cast<float>(b2v.v[i]));
    store_fp8_row(o, y8_2 + static_cast<size_t>(row) * d + col, sx_2 + row, lane);
  }
}

// ---- persistent fp8 GEMM ------------------------------------------------------------------------------------------
constexpr int EPI_GELU = 1;

template <typename T, int EPI>
__global__ __launch_bounds__(256, 4) void gemm_fp8_persist(const uint8_t* __restrict__ X, int ldx, const float* __restrict__ sx,
                                                           const uint8_t* __restrict__ W, const float* __restrict__ sw,
                                                           const T* __restrict__ bias, T* __restrict__ Y, int ldy, int M,
                                                           int N, int K, int n_tiles, int tiles_total) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
  const int tq = tiles_total >> 3, tr = tiles_total & 7;
  const int lo = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq, cnt = tq + (xcd < tr ? 1 : 0);
  int t = blockIdx.x >> 3;
  if (t >= cnt) return;                                            // block-uniform
  const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem)) +
                            wave * 4096;
  uint32_t ox[2], ow[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (wave * 4 + i) * 8 + (lane >> 3);
    const int logical = (lane & 7) ^ ((row >> 1) & 7);
    ox[i] = static_cast<uint32_t>(row * ldx + logical * 16);
    ow[i] = static_cast<uint32_t>(row * K + logical * 16);
  }
  auto issue = [&](const uint8_t* px, const uint8_t* pw) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16_asm_s(px + static_cast<size_t>((i >> 1) * 16) * ldx, ox[i & 1], lds_base + i * 1024);
      glds16_asm_s(pw + static_cast<size_t>((i >> 1) * 16) * K, ow[i & 1], lds_base + TILE_BYTES + i * 1024);
    }
  };
  const int frow = lane & 15, fch = lane >> 4;
  const int nk = K / BK8;
  const char* bufA = smem;
  const char* bufB = bufA + TILE_BYTES;
  int tile = lo + t;
  const uint8_t* px0 = X + static_cast<size_t>((tile / n_tiles) * BM) * ldx;
  const uint8_t* pw0 = W + static_cast<size_t>((tile % n_tiles) * BN) * K;
  issue(px0, pw0);
  bool first = true;
  for (;;) {
    floatx4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};
    for (int kt = 0; kt < nk; ++kt) {
      if (kt > 0) issue(px0 + kt * BK8, pw0 + kt * BK8);
      if (kt == 0 && !first) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // previous tile's 8 stores stay in flight
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
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
          for (int mt = 0; mt < 4; ++mt) {
            // the 16-byte chunk holds this lane's 8 k-values of two 32-wide k-blocks: same permutation on both operands
            const long wl = static_cast<long>((static_cast<unsigned long long>(fw[nt].y) << 32) | fw[nt].x);
            const long wh = static_cast<long>((static_cast<unsigned long long>(fw[nt].w) << 32) | fw[nt].z);
            const long xl = static_cast<long>((static_cast<unsigned long long>(fx[mt].y) << 32) | fx[mt].x);
            const long xh = static_cast<long>((static_cast<unsigned long long>(fx[mt].w) << 32) | fx[mt].z);
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wl, xl, acc[nt][mt], 0, 0, 0);
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(wh, xh, acc[nt][mt], 0, 0, 0);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    const int m0 = (tile / n_tiles) * BM + wm * 64, n0 = (tile % n_tiles) * BN + wn * 64;
    t += per_xcd;
    const bool more = t < cnt;                            // block-uniform
    if (more) {
      tile = lo + t;
      px0 = X + static_cast<size_t>((tile / n_tiles) * BM) * ldx;
      pw0 = W + static_cast<size_t>((tile % n_tiles) * BN) * K;
      issue(px0, pw0);
    }
    // ---- epilogue: D[n = nt*16 + g*4 + r][m = mt*16 + (lane & 15)]; dequantise, bias, activation, regroup, 16-byte stores
    {
      const int g = lane >> 4, mrow = m0 + (lane & 15);
      const int nq = (g & 1) * 16 + (g >> 1) * 8;
      float cs[4][4], bv[4][4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + nt * 16 + g * 4 + r;
          cs[nt][r] = sw[n];
          bv[nt][r] = bias ? static_cast<float>(bias[n]) : 0.f;
        }
      T* y = Y + static_cast<size_t>(mrow) * ldy + n0 + nq;
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) {
        const float rs = sx[mrow + mt * 16];
#pragma unroll
        for (int np = 0; np < 2; ++np) {
          float v[8];
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            const int nt = 2 * np + (r >> 2);
            v[r] = rn<T>(acc[nt][mt][r & 3] * (rs * cs[nt][r & 3]) + bv[nt][r & 3]);
            if (EPI & EPI_GELU) v[r] = rn<T>(gelu_erf(v[r]));
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) swap_rows16(v[r], v[4 + r]);
          *reinterpret_cast<uintx4*>(y + np * 32) =
              uintx4{pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]), pack2<T>(v[4], v[5]), pack2<T>(v[6], v[7])};
        }
        y += static_cast<size_t>(16) * ldy;
      }
    }
    if (!more) break;
    first = false;
  }
}

}  // namespace

bool fp8_linear_supported(int out_dtype, int M, int N, int K, int ldx, int ldy) {
  return (out_dtype == D3PM_F16 || out_dtype == D3PM_BF16) && M % BM == 0 && N % BN == 0 && K % BK8 == 0 && ldx % 16 == 0 &&
         ldy % 8 == 0 && static_cast<long long>(BM) * ldx < (1ll << 31) && static_cast<long long>(BN) * K < (1ll << 31);
}

int fp8_linear(int out_dtype, const uint8_t* X, int ldx, const float* sx, const uint8_t* W, const float* sw, const void* bias,
               void* Y, int ldy, int M, int N, int K, int act, hipStream_t s) {
  const int n_tiles = N / BN, tiles_total = n_tiles * (M / BM), want = (tiles_total + 7) & ~7;
  const dim3 grid(static_cast<unsigned>(want < 1024 ? want : 1024)), block(256);
#define D3PM_FP8(U, E)                                                                                              \
  gemm_fp8_persist<U, E><<<grid, block, 2 * TILE_BYTES, s>>>(X, ldx, sx, W, sw, static_cast<const U*>(bias),       \
                                                              static_cast<U*>(Y), ldy, M, N, K, n_tiles, tiles_total)
  if (out_dtype == D3PM_F16) { if (act == ACT_GELU) D3PM_FP8(f16, EPI_GELU); else D3PM_FP8(f16, 0); }
  else { if (act == ACT_GELU) D3PM_FP8(bf16, EPI_GELU); else D3PM_FP8(bf16, 0); }
#undef D3PM_FP8
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

int layernorm_fp8(int dtype, const void* x, uint8_t* y8, float* sx, const void* w, const void* b, const void* film,
                  const void* w2, const void* b2, uint8_t* y8_2, float* sx_2, int M, int d, float eps, hipStream_t s) {
  D3PM_REQUIRE(d == 512 && (dtype == D3PM_F16 || dtype == D3PM_BF16), D3PM_E_SHAPE, "layernorm_fp8: d = 512, 16-bit input only");
  const dim3 grid((M + 3) / 4), block(256);
  if (dtype == D3PM_F16)
    layernorm_fp8_rows<f16><<<grid, block, 0, s>>>(static_cast<const f16*>(x), y8, sx, static_cast<const f16*>(w),
                                                   static_cast<const f16*>(b), static_cast<const f16*>(film),
                                                   static_cast<const f16*>(w2), static_cast<const f16*>(b2), y8_2, sx_2, M, eps);
  else
    layernorm_fp8_rows<bf16><<<grid, block, 0, s>>>(static_cast<const bf16*>(x), y8, sx, static_cast<const bf16*>(w),
                                                    static_cast<const bf16*>(b), static_cast<const bf16*>(film),
                                                    static_cast<const bf16*>(w2), static_cast<const bf16*>(b2), y8_2, sx_2, M, eps);
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

}  // namespace d3pm
