// d3pm_mfma_gemm.hip -- LDS-tiled MFMA GEMM for the DiT projections on gfx950 (f16 / bf16).
//
//   Y[M][N] = epilogue(X[M][K] . W[N][K]^T + bias)      both operands K-contiguous (torch Linear layout)
//
// replaces every nn.Linear / MultiheadAttention in/out projection of DiTBlock.forward
// (/root/reference/vall_e/vall_e/ar_discrete.py:132,138,142,159) and the final Linear (:776) when the
// shape tiles (K % 64 == 0, 16-byte aligned rows); everything else goes to linear_tiled (generic).
//
// Structure (one workgroup = 4 wave64 = 128 x 128 output tile, K-step 64):
//   * global -> registers -> LDS staging, 16 B per lane, next K-tile's loads issued before the
//     current tile's MFMAs and written to the other LDS buffer after them (one barrier per K-tile);
//   * LDS rows are 128 B (64 k); 16-B chunk c of row r lives at chunk c ^ ((r >> 1) & 7): both the
//     ds_write_b128 of the staging pass and the ds_read_b128 of the MFMA fragments are bank-conflict
//     free (checked exhaustively against the gfx950 lane groups);
//   * v_mfma_f32_16x16x32_{f16,bf16}: each wave owns 64 x 64 = 4 x 4 tiles, fp32 accumulators;
//   * the MFMA is issued as D = W_frag . X_frag^T so that a lane ends up with 4 consecutive output
//     columns of one row: bias / GELU / residual / mask run on registers and the store (and the
//     residual loads) are 8 B per lane;
//   * epilogue rounding points are the eager model's (see d3pm_kernels.h).
// M and N tails are handled by clamped loads and predicated stores; K must be a multiple of 64.
#include "d3pm_kernels.h"

namespace d3pm {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int ROW_BYTES = BK * 2;                       // 128 B per LDS row
constexpr int TILE_BYTES = BM * ROW_BYTES;              // 16 KiB per operand tile

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * ROW_BYTES + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <typename T> __device__ __forceinline__ floatx4 mma(uint4 a, uint4 b, floatx4 c);
template <> __device__ __forceinline__ floatx4 mma<f16>(uint4 a, uint4 b, floatx4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ floatx4 mma<bf16>(uint4 a, uint4 b, floatx4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }

template <typename T> struct Pack4 { T v[4]; };

template <typename T>
__device__ __forceinline__ void epilogue_store(floatx4 (&acc)[4][4], const T* __restrict__ bias, T* Y, int ldy,
                                               const T* R1, const T* R2, int ldr, const uint8_t* __restrict__ row_mask,
                                               int mask_period, int M, int N, int act, int mw0, int nw0, int lane) {
  // epilogue: lane holds D[n = nt*16 + (lane>>4)*4 + r][m = mt*16 + (lane&15)], r = 0..3
  const int nq = (lane >> 4) * 4;
  float bv[4][4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = nw0 + nt * 16 + nq + r;
      bv[nt][r] = (bias && n < N) ? static_cast<float>(bias[n]) : 0.f;
    }
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) {
    const int m = mw0 + mt * 16 + (lane & 15);
    if (m >= M) continue;
    const float mk = row_mask ? (row_mask[m % mask_period] ? 1.f : 0.f) : 1.f;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = nw0 + nt * 16 + nq;
      if (n >= N) continue;
      const bool full = n + 3 < N;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] = rn<T>(acc[nt][mt][r] + bv[nt][r]);
        if (act == ACT_GELU) v[r] = rn<T>(gelu_erf(v[r]));
      }
      if (R1) {
        const T* r1 = R1 + static_cast<size_t>(m) * ldr + n;
        const T* r2 = R2 ? R2 + static_cast<size_t>(m) * ldr + n : nullptr;
        if (full) {
          Pack4<T> p1 = *reinterpret_cast<const Pack4<T>*>(r1), p2{};
          if (r2) p2 = *reinterpret_cast<const Pack4<T>*>(r2);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float res = static_cast<float>(p1.v[r]);
            if (r2) res = rn<T>(res + static_cast<float>(p2.v[r]));
            v[r] = rn<T>(res + v[r]);
          }
        } else {
          for (int r = 0; r < 4 && n + r < N; ++r) {
            float res = static_cast<float>(r1[r]);
            if (r2) res = rn<T>(res + static_cast<float>(r2[r]));
            v[r] = rn<T>(res + v[r]);
          }
        }
      }
      T* y = Y + static_cast<size_t>(m) * ldy + n;
      if (full) {
        Pack4<T> o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o.v[r] = static_cast<T>(v[r] * mk);
        *reinterpret_cast<Pack4<T>*>(y) = o;
      } else {
        for (int r = 0; r < 4 && n + r < N; ++r) y[r] = static_cast<T>(v[r] * mk);
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256, 2) void gemm_mfma_128(const T* __restrict__ X, int ldx, const T* __restrict__ W,
                                                        const T* __restrict__ bias, T* Y, int ldy, const T* R1,
                                                        const T* R2, int ldr, const uint8_t* __restrict__ row_mask,
                                                        int mask_period, int M, int N, int K, int act, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][A tile | B tile]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile_n = blockIdx.x % n_tiles, tile_m = blockIdx.x / n_tiles;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // staging assignment: 4 x 16 B of the X tile and 4 x 16 B of the W tile per thread
  const T* gx[4];
  const T* gw[4];
  int soff[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int c = tid + 256 * i, row = c >> 3, ch = c & 7;
    int mr = m0 + row, nr = n0 + row;
    mr = mr < M ? mr : M - 1;
    nr = nr < N ? nr : N - 1;
    gx[i] = X + static_cast<size_t>(mr) * ldx + ch * 8;
    gw[i] = W + static_cast<size_t>(nr) * K + ch * 8;
    soff[i] = lds_off(row, ch);
  }
  uint4 rx[4], rw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    rx[i] = *reinterpret_cast<const uint4*>(gx[i]);
    rw[i] = *reinterpret_cast<const uint4*>(gw[i]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    *reinterpret_cast<uint4*>(smem + soff[i]) = rx[i];
    *reinterpret_cast<uint4*>(smem + TILE_BYTES + soff[i]) = rw[i];
  }
  __syncthreads();

  floatx4 acc[4][4];   // [nt][mt]: rows of D index n, columns index m
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fch = lane >> 4;
  const int nk = K / BK;
  for (int kt = 0; kt < nk; ++kt) {
    const char* bufA = smem + (kt & 1) * 2 * TILE_BYTES;
    const char* bufB = bufA + TILE_BYTES;
    const bool more = kt + 1 < nk;
    if (more) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        rx[i] = *reinterpret_cast<const uint4*>(gx[i] + (kt + 1) * BK);
        rw[i] = *reinterpret_cast<const uint4*>(gw[i] + (kt + 1) * BK);
      }
    }
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
    if (more) {
      char* nb = smem + ((kt + 1) & 1) * 2 * TILE_BYTES;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        *reinterpret_cast<uint4*>(nb + soff[i]) = rx[i];
        *reinterpret_cast<uint4*>(nb + TILE_BYTES + soff[i]) = rw[i];
      }
    }
    __syncthreads();
  }

  epilogue_store<T>(acc, bias, Y, ldy, R1, R2, ldr, row_mask, mask_period, M, N, act, m0 + wm * 64, n0 + wn * 64, lane);
}


// ---- variant 2: direct-to-LDS staging (global_load_lds_dwordx4) -------------------------------------
// Each wave-instruction lands 1 KiB = 8 rows x 128 B linearly in LDS (wave-uniform base + lane*16), so the
// XOR swizzle is applied to the per-lane SOURCE address (logical chunk = lane&7 ^ f(row)) and undone by the
// same lds_off() on the fragment reads.  No staging VGPRs, no ds_write pass.
typedef __attribute__((address_space(3))) void* lds_void;
typedef const __attribute__((address_space(1))) void* glb_void;

template <typename T, int NBUF>   // NBUF 2: next tile's DMA issued before the MFMAs; NBUF 1: 32 KiB LDS, 4 workgroups per CU
__global__ __launch_bounds__(256, NBUF == 1 ? 4 : 2) void gemm_mfma_128_glds(const T* __restrict__ X, int ldx, const T* __restrict__ W,
                                                             const T* __restrict__ bias, T* Y, int ldy, const T* R1,
                                                             const T* R2, int ldr, const uint8_t* __restrict__ row_mask,
                                                             int mask_period, int M, int N, int K, int act, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile_n = blockIdx.x % n_tiles, tile_m = blockIdx.x / n_tiles;
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
  if (NBUF == 2) {
    issue(0, 0);
    __syncthreads();
  }

  floatx4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};

  const int frow = lane & 15, fch = lane >> 4;
  const int nk = K / BK;
  for (int kt = 0; kt < nk; ++kt) {
    const char* bufA = smem + (NBUF == 2 ? (kt & 1) : 0) * 2 * TILE_BYTES;
    const char* bufB = bufA + TILE_BYTES;
    if (NBUF == 2) {
      if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
    } else {
      issue(kt, 0);
      __syncthreads();
    }
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
  epilogue_store<T>(acc, bias, Y, ldy, R1, R2, ldr, row_mask, mask_period, M, N, act, m0 + wm * 64, n0 + wn * 64, lane);
}

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

}  // namespace

bool mfma_linear_supported(int dtype, const LinearArgs& a) {
  if (dtype != D3PM_F16 && dtype != D3PM_BF16) return false;
  if (a.M < 1 || a.N < 1 || a.K < BK || a.K % BK != 0) return false;
  if (a.ldx % 8 != 0 || a.ldy % 4 != 0 || !aligned(a.X, 16) || !aligned(a.W, 16) || !aligned(a.Y, 8)) return false;
  if (a.R1 && (a.ldr % 4 != 0 || !aligned(a.R1, 8))) return false;
  if (a.R2 && !aligned(a.R2, 8)) return false;
  if (static_cast<long long>(a.M) * a.N < 128 * 128) return false;     // not worth a 128^2 tile
  return true;
}

static int g_gemm_variant = 0;   // 0 = register staging, 1 = direct-to-LDS (2 buffers), 2 = direct-to-LDS (1 buffer, 4 WG/CU)
void set_gemm_variant(int v) { g_gemm_variant = v; }

int mfma_linear(int dtype, const LinearArgs& a, hipStream_t s) {
  const int n_tiles = (a.N + BN - 1) / BN, m_tiles = (a.M + BM - 1) / BM;
  const size_t lds = (g_gemm_variant == 2 ? 2 : 4) * TILE_BYTES;   // 64 KiB: two workgroups per CU; 32 KiB: four
  dim3 grid(static_cast<unsigned>(n_tiles) * m_tiles), block(256);
#define D3PM_GEMM(...)                                                                                        \
  do {                                                                                                        \
    static bool attr_set = false;                                                                             \
    if (!attr_set) {                                                                                          \
      D3PM_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&__VA_ARGS__),                         \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES));        \
      attr_set = true;                                                                                        \
    }                                                                                                         \
    using T = std::remove_const_t<std::remove_pointer_t<decltype(tag)>>;                                      \
    __VA_ARGS__<<<grid, block, lds, s>>>(static_cast<const T*>(a.X), a.ldx, static_cast<const T*>(a.W),      \
                                         static_cast<const T*>(a.bias), static_cast<T*>(a.Y), a.ldy,          \
                                         static_cast<const T*>(a.R1), static_cast<const T*>(a.R2), a.ldr,     \
                                         a.row_mask, a.mask_period, a.M, a.N, a.K, a.act, n_tiles);           \
  } while (0)
  auto go = [&](auto* tag) -> int {
    using U = std::remove_pointer_t<decltype(tag)>;
    if (g_gemm_variant == 1) D3PM_GEMM(gemm_mfma_128_glds<U, 2>);
    else if (g_gemm_variant == 2) D3PM_GEMM(gemm_mfma_128_glds<U, 1>);
    else D3PM_GEMM(gemm_mfma_128<U>);
    return D3PM_OK;
  };
  int rc = dtype == D3PM_F16 ? go(static_cast<f16*>(nullptr)) : go(static_cast<bf16*>(nullptr));
  if (rc != D3PM_OK) return rc;
#undef D3PM_GEMM
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

}  // namespace d3pm
