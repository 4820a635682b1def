// d3pm_fold_rows.h -- per-row device routines of the folded-LayerNorm preparation, shared by d3pm_fold.hip (stand-alone
// launches) and d3pm_sample.hip (the sampler launch of iteration t prepares iteration t - 1: embedding gather + row moments of the
// rows it has just sampled, and fc1 of every block under norm3 + FiLM(t - 1), in the workgroups behind the sampler's).
#pragma once
#include "d3pm_kernels.h"

namespace d3pm {
namespace {

template <typename T> struct Vec8 { T v[8]; };

// float index of (row, part): [row / 16][part][row % 16][2] -- d3pm_mfma_tile.h has the same function for the GEMM epilogues
__device__ __forceinline__ size_t stats_index_dev(size_t row, int part, int parts) { return (((row >> 4) * parts + part) * 16 + (row & 15)) * 2; }

// fc1 of every layer at one timestep: row r = (layer r / n_rows, output n = r % n_rows); film_t = film[t] = [L][2K]
struct FoldStepPtrs { const void* W[16]; const void* bias[16]; const void* gamma[16]; const void* beta[16]; };

// one wave, one output row r of fc1 o norm3 o FiLM(t): W' row, s, b' (the arithmetic of fold_rows with FiLM, statement for statement)
template <typename T>
__device__ __forceinline__ void fold_layer_row(const FoldStepPtrs& p, const T* __restrict__ film_t, int n_rows, int K, int r, int lane,
                                               T* __restrict__ Wf, float* __restrict__ s_out, float* __restrict__ b_out) {
  const int l = r / n_rows, n = r % n_rows;
  const T* wrow = static_cast<const T*>(p.W[l]) + static_cast<size_t>(n) * K;
  const T* gamma = static_cast<const T*>(p.gamma[l]);
  const T* beta = static_cast<const T*>(p.beta[l]);
  const T* frow = film_t + static_cast<size_t>(l) * 2 * K;
  T* orow = Wf + static_cast<size_t>(r) * K;
  float s = 0.f, b = 0.f;
  for (int k = lane * 8; k < K; k += 512) {
    const Vec8<T> w8 = *reinterpret_cast<const Vec8<T>*>(wrow + k), g8 = *reinterpret_cast<const Vec8<T>*>(gamma + k),
                  b8 = *reinterpret_cast<const Vec8<T>*>(beta + k), sc8 = *reinterpret_cast<const Vec8<T>*>(frow + k),
                  sh8 = *reinterpret_cast<const Vec8<T>*>(frow + K + k);
    Vec8<T> o8;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float w = static_cast<float>(w8.v[i]);
      float g = static_cast<float>(g8.v[i]), c = static_cast<float>(b8.v[i]);
      const float gg = rn<T>(1.0f + static_cast<float>(sc8.v[i]));
      g *= gg;
      c = __builtin_fmaf(c, gg, static_cast<float>(sh8.v[i]));
      o8.v[i] = static_cast<T>(w * g);
      s += static_cast<float>(o8.v[i]);
      b = __builtin_fmaf(w, c, b);
    }
    *reinterpret_cast<Vec8<T>*>(orow + k) = o8;
  }
  s = wave_sum(s);
  b = wave_sum(b);
  if (lane == 0) {
    s_out[r] = s;
    b_out[r] = b + static_cast<float>(static_cast<const T*>(p.bias[l])[n]);
  }
}

// the in-lane and in-quad half of a 32-column part's moments: lane L of a pass owns one 16-byte chunk, a quad owns a part
template <typename T>
__device__ __forceinline__ void part_moments(const Vec8<T>& raw, float& a, float& q) {
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = static_cast<float>(raw.v[i]);
  a = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));      // = part_moments8 of d3pm_mfma_tile.h
  q = v[7] * v[7];
#pragma unroll
  for (int i = 6; i >= 0; --i) q = __builtin_fmaf(v[i], v[i], q);
  // the order of the GEMM epilogues (d3pm_mfma_tile.h part_stats_store: columns 0-7 + 16-23, then + (8-15 + 24-31)), so that the
  // moments of a row are the same bits whichever kernel produced them
  a = add_dpp<0x4E>(a); q = add_dpp<0x4E>(q);     // lane ^ 2
  a = add_dpp<0xB1>(a); q = add_dpp<0xB1>(q);     // lane ^ 1
}

// one wave, one canvas row: x[row] = table[id] (zeros on a padded frame), and the row's moments (embed_rows_vec + row_stats)
template <typename T>
__device__ __forceinline__ void embed_row_stats(const T* __restrict__ table, int id, bool live, T* __restrict__ y, int row, int d, int n_classes,
                                                float* __restrict__ stats, int lane) {
  typedef float float2v __attribute__((ext_vector_type(2)));
  id = id < 0 ? 0 : (id >= n_classes ? n_classes - 1 : id);
  const Vec8<T>* src = reinterpret_cast<const Vec8<T>*>(table + static_cast<size_t>(id) * d);
  Vec8<T>* dst = reinterpret_cast<Vec8<T>*>(y + static_cast<size_t>(row) * d);
  const int parts = d >> 5;
  for (int c = lane; c < (d >> 3); c += kWave) {
    Vec8<T> raw = src[c];
    if (!live) raw = Vec8<T>{};
    dst[c] = raw;
    float a, q;
    part_moments(raw, a, q);
    if ((lane & 3) == 0) *reinterpret_cast<float2v*>(stats + stats_index_dev(static_cast<size_t>(row), c >> 2, parts)) = float2v{a, q};
  }
}

}  // namespace
}  // namespace d3pm
