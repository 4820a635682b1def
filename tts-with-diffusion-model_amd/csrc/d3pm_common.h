// d3pm_common.h -- element types, rounding helpers, wave64 reductions and the Philox stream
// shared by every gfx950 kernel of the D3PM sampler.  Wavefront width is 64 throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/d3pm_hip.h"

namespace d3pm {

using f16 = _Float16;
using bf16 = __bf16;

constexpr int kWave = 64;

// ---- element access: every activation/weight is float, f16 or bf16; math is fp32 -----------
template <typename T> __device__ __forceinline__ float ldf(const T* p) { return static_cast<float>(*p); }
template <typename T> __device__ __forceinline__ void stf(T* p, float v) { *p = static_cast<T>(v); }
// value of `v` after the store-rounding the eager reference applies at each op output
template <typename T> __device__ __forceinline__ float rn(float v) { return static_cast<float>(static_cast<T>(v)); }
template <> __device__ __forceinline__ float rn<float>(float v) { return v; }
__device__ __forceinline__ float rn16(float v) { return static_cast<float>(static_cast<f16>(v)); }

// ---- wave64 reductions ---------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
  return v;
}
// v[l] + v[l ^ 32] and v[l] + v[l ^ 16] without the LDS crossbar: v_permlane{32,16}_swap of two copies of v leaves one register
// holding the even halves (rows) twice and the other the odd ones, and a + b is bit for bit b + a -- the butterfly step of
// __shfl_xor(v, 32 / 16) at a few cycles instead of a ds_bpermute round trip (~120).  `s_nop 1`: VALU write -> permlane read.
__device__ __forceinline__ float add_xor32(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float add_xor16(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
template <int CTRL> __device__ __forceinline__ float add_dpp(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// the same sum with the butterfly walked from lane distance 1 up to 32.  The vectorised LayerNorms use this order: lane L
// holds the 8-element chunk L of a 512-wide row, and the GEMM epilogue that normalises a finished row in place
// (d3pm_mfma_gemm_big.hip, row-panel kernel) holds chunk 8 wave + 4 np + 2 (lane bit 4) + (lane bit 5) -- low chunk bits
// inside a wave, high bits across waves -- so walking low to high lets it reproduce this tree exactly: same bits out.
// Every step is v[l] + v[l ^ off] as with __shfl_xor, on the DPP / permlane paths instead of six ds_bpermute round trips (at one
// utterance a LayerNorm launch is a chain of dependent latencies, and the two reductions were a quarter of it): off = 1, 2 are
// quad permutes; after them the four lanes of a quad hold the same bits, so the lane off = 4 away may be ANY lane of the
// neighbouring quad (row_half_mirror: l ^ 7) and likewise off = 8 any lane of the other half row (row_mirror: l ^ 15).
__device__ __forceinline__ float wave_sum_up(float v) {
  v = add_dpp<0xB1>(v);    // quad_perm [1, 0, 3, 2]: l ^ 1
  v = add_dpp<0x4E>(v);    // quad_perm [2, 3, 0, 1]: l ^ 2
  v = add_dpp<0x141>(v);   // row_half_mirror
  v = add_dpp<0x140>(v);   // row_mirror
  v = add_xor16(v);
  return add_xor32(v);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, kWave));
  return v;
}
// argmax with first-index tie-break (torch.argmax on CPU returns the first maximal index)
__device__ __forceinline__ void wave_argmax(float& v, int& idx) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    float ov = __shfl_xor(v, off, kWave);
    int oi = __shfl_xor(idx, off, kWave);
    if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
  }
}

// ---- Philox4x32-10 (Salmon et al. 2011), counter-based: no state, any (row, class, t) in O(1) --
struct Philox {
  static constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
  __host__ __device__ static inline void round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    uint64_t p0 = static_cast<uint64_t>(M0) * c[0];
    uint64_t p1 = static_cast<uint64_t>(M1) * c[2];
    uint32_t hi0 = static_cast<uint32_t>(p0 >> 32), lo0 = static_cast<uint32_t>(p0);
    uint32_t hi1 = static_cast<uint32_t>(p1 >> 32), lo1 = static_cast<uint32_t>(p1);
    uint32_t n0 = hi1 ^ c[1] ^ k0, n2 = hi0 ^ c[3] ^ k1;
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
  }
  __host__ __device__ static inline void gen(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
      round(c, k0, k1);
      k0 += W0;
      k1 += W1;
    }
  }
};
// stream definition (oracle/philox.py): counter = (class>>2, global row, t, stream), word = class&3
__host__ __device__ inline void noise4(uint64_t seed, uint32_t group, uint32_t row, uint32_t t,
                                       uint32_t stream, float (&u)[4]) {
  uint32_t c[4] = {group, row, t, stream};
  Philox::gen(c, static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
#pragma unroll
  for (int i = 0; i < 4; ++i) u[i] = static_cast<float>(c[i] >> 8) * 5.9604644775390625e-08f;  // 2^-24
}
// -log(-log(clamp(u, FLT_MIN, 1)))  (ar_discrete.py:416-417)
__device__ __forceinline__ float gumbel(float u) {
  u = fminf(fmaxf(u, 1.17549435e-38f), 1.0f);
  return -logf(-logf(u));
}

// ---- launch / error plumbing (host) ----------------------------------------------------------
void set_error(const char* fmt, ...);
#define D3PM_CHECK_HIP(expr)                                                        \
  do {                                                                              \
    hipError_t e_ = (expr);                                                         \
    if (e_ != hipSuccess) {                                                         \
      ::d3pm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return D3PM_E_HIP;                                                            \
    }                                                                               \
  } while (0)
#define D3PM_LAUNCH_CHECK() D3PM_CHECK_HIP(hipGetLastError())
#define D3PM_REQUIRE(cond, code, ...)          \
  do {                                         \
    if (!(cond)) {                             \
      ::d3pm::set_error(__VA_ARGS__);          \
      return (code);                           \
    }                                          \
  } while (0)

inline size_t dtype_size(int dtype) { return dtype == D3PM_F32 ? 4 : 2; }

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to a (kernel, device) pair.  Every launch site keeps one bit per device of the
// process (an idempotent memo, set with atomics: two threads racing on it both set the attribute, which is harmless) so that a
// process driving several GPUs sets it on each of them and the common case costs one relaxed load.
inline hipError_t lds_attr_for_device(const void* fn, size_t bytes, std::atomic<uint64_t>& done) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
  if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
  return e;
}
#define D3PM_LDS_ATTR(fn, bytes)                                                                              \
  do {                                                                                                        \
    static std::atomic<uint64_t> lds_attr_done_{0};                                                           \
    D3PM_CHECK_HIP(::d3pm::lds_attr_for_device(reinterpret_cast<const void*>(fn), (bytes), lds_attr_done_)); \
  } while (0)

}  // namespace d3pm
