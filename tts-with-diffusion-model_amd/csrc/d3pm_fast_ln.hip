// d3pm_fast_ln.hip -- vectorised LayerNorm (+ optional second LayerNorm of the same rows, + FiLM)
// for 16-bit activations with d_model a multiple of 512: one wave64 per row, every lane keeps its
// 8 (16, ..) contiguous elements in registers, so each row is read once (16 B per lane, fully
// coalesced) and written once.  HBM-bound: 2 (3 with the dual output) x N x d x 2 bytes per launch.
// Same arithmetic contract as layernorm_rows (two-pass fp32 moments, eager rounding points).
// Replaces nn.LayerNorm at ar_discrete.py:131,136,140,153 and the FiLM modulation at :146-156.
#include "d3pm_kernels.h"

namespace d3pm {
namespace {

template <typename T> struct Vec8 { T v[8]; };

template <typename T, int CH>   // CH = d / 512 chunks of 8 elements per lane
__global__ __launch_bounds__(256) void layernorm_vec(const T* __restrict__ x, T* __restrict__ y,
                                                     const T* __restrict__ w, const T* __restrict__ b,
                                                     const T* __restrict__ w2, const T* __restrict__ b2,
                                                     T* __restrict__ y2, const T* __restrict__ film, int M, float eps,
                                                     const int32_t* __restrict__ tok, const uint8_t* __restrict__ frame_mask,
                                                     int canvas, int n_classes, T* __restrict__ xout) {
  constexpr int d = CH * 512;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a contiguous range of rows -- the ranges
  // the GEMM tiles that wrote these rows and the ones that read the result are dealt in (LayerNorm class 74 -> 70 us per iteration
  // in the loop: part of its input is still in the XCD's own L2; profiles/round3_l2_ab_ln_xcd.txt)
  int bidx;
  {
    const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    bidx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int row = bidx * 4 + wave;
  if (row >= M) return;
  const T* xr = x + static_cast<size_t>(row) * d;
  bool live = true;
  if (tok) {                                  // x is the embedding table: this row = table[token] * frame mask (embed_rows_vec)
    int id = tok[row];
    id = id < 0 ? 0 : (id >= n_classes ? n_classes - 1 : id);
    live = frame_mask[row % canvas] != 0;
    xr = x + static_cast<size_t>(id) * d;
  }
  // d = 512: the per-column parameters are fetched beside the row, not behind the two reductions (at one utterance a launch is a
  // chain of dependent round trips: this removes one of them; at throughput batches the launch is memory-bound either way)
  [[maybe_unused]] Vec8<T> pw, pb, pw2, pb2, psc, psh;
  if constexpr (CH == 1) {
    pw = *reinterpret_cast<const Vec8<T>*>(w + lane * 8); pb = *reinterpret_cast<const Vec8<T>*>(b + lane * 8);
    if (film) { psc = *reinterpret_cast<const Vec8<T>*>(film + lane * 8); psh = *reinterpret_cast<const Vec8<T>*>(film + d + lane * 8); }
    if (y2) { pw2 = *reinterpret_cast<const Vec8<T>*>(w2 + lane * 8); pb2 = *reinterpret_cast<const Vec8<T>*>(b2 + lane * 8); }
  }
  float v[CH][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    Vec8<T> raw = *reinterpret_cast<const Vec8<T>*>(xr + (c * 64 + lane) * 8);
    if (tok) {
      if (!live) raw = Vec8<T>{};
      *reinterpret_cast<Vec8<T>*>(xout + static_cast<size_t>(row) * d + (c * 64 + lane) * 8) = raw;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[c][i] = static_cast<float>(raw.v[i]); s += v[c][i]; }
  }
  const float mean = wave_sum_up(s) / static_cast<float>(d);
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) { float t = v[c][i] - mean; q += t * t; }
  const float rstd = rsqrtf(wave_sum_up(q) / static_cast<float>(d) + eps);
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int col = (c * 64 + lane) * 8;
    Vec8<T> wv, bv, o;
    if constexpr (CH == 1) { wv = pw; bv = pb; }
    else { wv = *reinterpret_cast<const Vec8<T>*>(w + col); bv = *reinterpret_cast<const Vec8<T>*>(b + col); }
    float n[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      n[i] = (v[c][i] - mean) * rstd;
      o.v[i] = static_cast<T>(n[i] * static_cast<float>(wv.v[i]) + static_cast<float>(bv.v[i]));
    }
    if (film) {
      Vec8<T> sc, sh;
      if constexpr (CH == 1) { sc = psc; sh = psh; }
      else { sc = *reinterpret_cast<const Vec8<T>*>(film + col); sh = *reinterpret_cast<const Vec8<T>*>(film + d + col); }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        float g = rn<T>(1.0f + static_cast<float>(sc.v[i]));
        o.v[i] = static_cast<T>(rn<T>(static_cast<float>(o.v[i]) * g) + static_cast<float>(sh.v[i]));
      }
    }
    *reinterpret_cast<Vec8<T>*>(y + static_cast<size_t>(row) * d + col) = o;
    if (y2) {
      Vec8<T> w2v, b2v, o2;
      if constexpr (CH == 1) { w2v = pw2; b2v = pb2; }
      else { w2v = *reinterpret_cast<const Vec8<T>*>(w2 + col); b2v = *reinterpret_cast<const Vec8<T>*>(b2 + col); }
#pragma unroll
      for (int i = 0; i < 8; ++i) o2.v[i] = static_cast<T>(n[i] * static_cast<float>(w2v.v[i]) + static_cast<float>(b2v.v[i]));
      *reinterpret_cast<Vec8<T>*>(y2 + static_cast<size_t>(row) * d + col) = o2;
    }
  }
}

template <typename T> int launch(const LayerNormArgs& a, hipStream_t s) {
  dim3 grid((a.M + 3) / 4), block(256);
#define D3PM_LN(CH)                                                                                          \
  layernorm_vec<T, CH><<<grid, block, 0, s>>>(static_cast<const T*>(a.X), static_cast<T*>(a.Y),              \
                                              static_cast<const T*>(a.w), static_cast<const T*>(a.b),        \
                                              static_cast<const T*>(a.w2), static_cast<const T*>(a.b2),      \
                                              static_cast<T*>(a.Y2), static_cast<const T*>(a.film), a.M, a.eps,  \
                                              a.tokens, a.frame_mask, a.canvas, a.n_classes, static_cast<T*>(a.Xout))
  switch (a.d / 512) {
    case 1: D3PM_LN(1); break;
    case 2: D3PM_LN(2); break;
    case 4: D3PM_LN(4); break;
    default: set_error("fast layernorm: unsupported d=%d", a.d); return D3PM_E_SHAPE;
  }
#undef D3PM_LN
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

}  // namespace

bool fast_layernorm_supported(int dtype, const LayerNormArgs& a) {
  if (dtype != D3PM_F16 && dtype != D3PM_BF16) return false;
  if (a.d != 512 && a.d != 1024 && a.d != 2048) return false;
  auto al = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) % 16) == 0; };
  if (a.tokens && !(a.frame_mask && a.canvas > 0 && a.n_classes > 0 && a.Xout && al(a.Xout))) return false;
  return al(a.X) && al(a.Y) && al(a.w) && al(a.b) && al(a.w2) && al(a.b2) && al(a.Y2) && al(a.film);
}

int fast_layernorm(int dtype, const LayerNormArgs& a, hipStream_t s) {
  return dtype == D3PM_F16 ? launch<f16>(a, s) : launch<bf16>(a, s);
}

}  // namespace d3pm
