// d3pm_nar.hip -- kernels specific to the stock NAR model that fills quantizer levels 1..7 after the D3PM
// sampler.  GEMMs, attention and the launch plumbing are shared with the denoiser.
//
// Replaces (paths under /root/reference/vall_e/vall_e/):
//   nar_embed_rows   Base.forward's input assembly base.py:441-449: Embedding / MultiEmbedding (:237-274) of the
//                    text | sep | prompt | sep | response segments (_join :277-286), zero padding to the longest
//                    utterance (list_to_tensor :19-35) and SinusodialEmbedding.add_pe (:84-92)
//   adaln_rows       AdaLN.forward base.py:136-158 followed by the `* m` of PrenormResidual.forward :195
//   nar_sample_rows  `Categorical(logits=h / T).sample()` base.py:487-494, as a Gumbel-max over the Philox stream 2
//                    (same distribution; torch's multinomial stream cannot be reproduced on a GPU)
#include "d3pm_kernels.h"

namespace d3pm {
namespace {

// one workgroup per row (b, pos) of the padded [B][T_max] grid
template <typename T>
__global__ void nar_embed_rows(const int32_t* __restrict__ lens, const int32_t* __restrict__ text, int tt_max,
                               const int32_t* __restrict__ prom, int tp_max, int n_prom_levels,
                               const int32_t* __restrict__ resp, int tr_max, int resp_stride, int n_given,
                               const T* __restrict__ w_text, const T* __restrict__ w_prom, const T* __restrict__ w_resp,
                               const T* __restrict__ sep, const T* __restrict__ pe, T* __restrict__ x,
                               uint8_t* __restrict__ row_mask, int32_t* __restrict__ key_len, int t_max, int d,
                               int n_tokens) {
  const int row = blockIdx.x, b = row / t_max, pos = row % t_max;
  const int tt = lens[b * 3], tp = lens[b * 3 + 1], tr = lens[b * 3 + 2];
  const int total = tt + 1 + tp + 1 + tr;
  if (threadIdx.x == 0) {
    row_mask[row] = pos < total ? 1 : 0;
    if (pos == 0) key_len[b] = total;
  }
  T* xr = x + static_cast<size_t>(row) * d;
  if (pos >= total) {
    for (int c = threadIdx.x; c < d; c += blockDim.x) xr[c] = static_cast<T>(0.f);
    return;
  }
  auto clampid = [&](int id) { return id < 0 ? 0 : (id >= n_tokens ? n_tokens - 1 : id); };
  const T* per = pe + static_cast<size_t>(pos) * d;
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    float v;
    if (pos < tt) {
      v = ldf(w_text + static_cast<size_t>(clampid(text[b * tt_max + pos])) * d + c);
    } else if (pos == tt || pos == tt + 1 + tp) {
      v = ldf(sep + c);
    } else if (pos < tt + 1 + tp) {
      const int32_t* cr = prom + (static_cast<size_t>(b) * tp_max + (pos - tt - 1)) * n_prom_levels;
      float acc = 0.f;
      for (int l = 0; l < n_prom_levels; ++l) {
        if (cr[l] < 0) continue;   // level absent
        acc += ldf(w_prom + (static_cast<size_t>(l) * n_tokens + clampid(cr[l])) * d + c);
      }
      v = rn<T>(acc);
    } else {
      const int32_t* cr = resp + (static_cast<size_t>(b) * tr_max + (pos - tt - tp - 2)) * resp_stride;
      float acc = 0.f;
      for (int l = 0; l < n_given; ++l) acc += ldf(w_resp + (static_cast<size_t>(l) * n_tokens + clampid(cr[l])) * d + c);
      v = rn<T>(acc);
    }
    stf(xr + c, v + ldf(per + c));
  }
}

// one wave per row: y = (exp(lg) * (c * (1 - k*h) * h) + beta) * mask, h = LayerNorm(x) without affine, every
// intermediate rounded to the storage dtype like the eager expression chain
template <typename T>
__global__ void adaln_rows(const T* __restrict__ x, T* __restrict__ y, const T* __restrict__ emb_row,
                           const uint8_t* __restrict__ row_mask, int M, int d, float eps, float k, float cc) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + wave;
  if (row >= M) return;
  T* yr = y + static_cast<size_t>(row) * d;
  if (!row_mask[row]) {
    for (int c = lane; c < d; c += kWave) yr[c] = static_cast<T>(0.f);
    return;
  }
  const T* xr = x + static_cast<size_t>(row) * d;
  float s = 0.f;
  for (int c = lane; c < d; c += kWave) s += ldf(xr + c);
  const float mean = wave_sum(s) / static_cast<float>(d);
  float q = 0.f;
  for (int c = lane; c < d; c += kWave) {
    float t = ldf(xr + c) - mean;
    q += t * t;
  }
  const float rstd = rsqrtf(wave_sum(q) / static_cast<float>(d) + eps);
  for (int c = lane; c < d; c += kWave) {
    const float h = rn<T>((ldf(xr + c) - mean) * rstd);
    float a = rn<T>(k * h);
    a = rn<T>(1.0f - a);
    a = rn<T>(cc * a);
    a = rn<T>(a * h);
    const float g = rn<T>(expf(ldf(emb_row + c)));
    stf(yr + c, rn<T>(g * a) + ldf(emb_row + d + c));
  }
}

// the same with 16-byte accesses for d_model a multiple of 512 (16-bit storage): lane l owns elements 8l .. 8l+7 of
// every 512-element chunk, the row is read once and kept in registers
template <typename T> struct alignas(16) Row8 { T v[8]; };

template <typename T, int CH>
__global__ __launch_bounds__(256) void adaln_rows_vec(const T* __restrict__ x, T* __restrict__ y, const T* __restrict__ emb_row,
                                                      const uint8_t* __restrict__ row_mask, int M, float eps, float k, float cc) {
  constexpr int d = CH * 512;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= M) return;
  T* yr = y + static_cast<size_t>(row) * d;
  if (!row_mask[row]) {
    Row8<T> z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z.v[i] = static_cast<T>(0.f);
#pragma unroll
    for (int c = 0; c < CH; ++c) *reinterpret_cast<Row8<T>*>(yr + (c * 64 + lane) * 8) = z;
    return;
  }
  const T* xr = x + static_cast<size_t>(row) * d;
  float v[CH][8], s = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const Row8<T> raw = *reinterpret_cast<const Row8<T>*>(xr + (c * 64 + lane) * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[c][i] = static_cast<float>(raw.v[i]); s += v[c][i]; }
  }
  const float mean = wave_sum(s) / static_cast<float>(d);
  float q = 0.f;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) { const float t = v[c][i] - mean; q += t * t; }
  const float rstd = rsqrtf(wave_sum(q) / static_cast<float>(d) + eps);
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int col = (c * 64 + lane) * 8;
    const Row8<T> lg = *reinterpret_cast<const Row8<T>*>(emb_row + col), be = *reinterpret_cast<const Row8<T>*>(emb_row + d + col);
    Row8<T> o;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float h = rn<T>((v[c][i] - mean) * rstd);
      float a = rn<T>(k * h);
      a = rn<T>(1.0f - a);
      a = rn<T>(cc * a);
      a = rn<T>(a * h);
      const float g = rn<T>(expf(static_cast<float>(lg.v[i])));
      o.v[i] = static_cast<T>(rn<T>(g * a) + static_cast<float>(be.v[i]));
    }
    *reinterpret_cast<Row8<T>*>(yr + col) = o;
  }
}

// one wave per response frame: argmax_j rn(logit_j / T) + gumbel(u_j); writes level `level + 1` of the frame
template <typename T>
__global__ void nar_sample_rows(const T* __restrict__ logits, int ldl, const int32_t* __restrict__ lens,
                                int32_t* __restrict__ resp, int tr_max, int resp_stride, int t_max, int n_tokens,
                                int level, float temperature, uint64_t seed, uint32_t utt0,
                                int greedy, int batch) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int idx = blockIdx.x * (blockDim.x >> 6) + wave;
  if (idx >= batch * tr_max) return;
  const int b = idx / tr_max, f = idx % tr_max;
  const int tt = lens[b * 3], tp = lens[b * 3 + 1], tr = lens[b * 3 + 2];
  if (f >= tr || tt + tp + 2 + f >= t_max) return;   // a sequence longer than the padded grid is truncated, never read past
  const T* lr = logits + (static_cast<size_t>(b) * t_max + (tt + tp + 2 + f)) * ldl;
  const int groups = (n_tokens + 3) >> 2;
  int best_j = 0;
  float best_v = -INFINITY;
  for (int g = lane; g < groups; g += kWave) {
    float u[4];
    if (!greedy) noise4(seed, static_cast<uint32_t>(g), (utt0 + static_cast<uint32_t>(b)) * 65536u + static_cast<uint32_t>(f),
                        static_cast<uint32_t>(level), 2u, u);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int j = g * 4 + w;
      if (j >= n_tokens) continue;
      float v = rn<T>(ldf(lr + j) / temperature);
      if (!greedy) v += gumbel(u[w]);
      if (v > best_v) { best_v = v; best_j = j; }
    }
  }
  wave_argmax(best_v, best_j);
  if (lane == 0) resp[(static_cast<size_t>(b) * tr_max + f) * resp_stride + level + 1] = best_j;
}

template <typename F> int dispatch(int dtype, F&& f) {
  switch (dtype) {
    case D3PM_F32: return f(static_cast<float*>(nullptr));
    case D3PM_F16: return f(static_cast<f16*>(nullptr));
    case D3PM_BF16: return f(static_cast<bf16*>(nullptr));
  }
  set_error("unknown dtype %d", dtype);
  return D3PM_E_ARG;
}

}  // namespace

int nar_embed(int dtype, const NarEmbedArgs& a, hipStream_t s) {
  return dispatch(dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    nar_embed_rows<T><<<a.batch * a.t_max, a.d >= 256 ? 256 : 64, 0, s>>>(
        a.lens, a.text, a.tt_max, a.prom, a.tp_max, a.n_prom_levels, a.resp, a.tr_max, a.resp_stride, a.n_given,
        static_cast<const T*>(a.w_text), static_cast<const T*>(a.w_prom), static_cast<const T*>(a.w_resp),
        static_cast<const T*>(a.sep), static_cast<const T*>(a.pe), static_cast<T*>(a.x), a.row_mask, a.key_len, a.t_max, a.d,
        a.n_tokens);
    D3PM_LAUNCH_CHECK();
    return D3PM_OK;
  });
}

int adaln(int dtype, const void* x, void* y, const void* emb_row, const uint8_t* row_mask, int M, int d, hipStream_t s) {
  return dispatch(dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    auto al = [](const void* p) { return reinterpret_cast<uintptr_t>(p) % 16 == 0; };
    const bool vec = sizeof(T) == 2 && (d == 512 || d == 1024) && al(x) && al(y) && al(emb_row);
    if (vec && d == 1024)
      adaln_rows_vec<T, 2><<<(M + 3) / 4, 256, 0, s>>>(static_cast<const T*>(x), static_cast<T*>(y), static_cast<const T*>(emb_row),
                                                       row_mask, M, 1e-5f, 0.1f, 2.0f);
    else if (vec)
      adaln_rows_vec<T, 1><<<(M + 3) / 4, 256, 0, s>>>(static_cast<const T*>(x), static_cast<T*>(y), static_cast<const T*>(emb_row),
                                                       row_mask, M, 1e-5f, 0.1f, 2.0f);
    else
      adaln_rows<T><<<(M + 3) / 4, 256, 0, s>>>(static_cast<const T*>(x), static_cast<T*>(y), static_cast<const T*>(emb_row),
                                                row_mask, M, d, 1e-5f, 0.1f, 2.0f);
    D3PM_LAUNCH_CHECK();
    return D3PM_OK;
  });
}

int nar_sample(int dtype, const void* logits, int ldl, const int32_t* lens, int32_t* resp, int tr_max, int resp_stride,
               int t_max, int n_tokens, int level, float temperature, uint64_t seed, uint32_t utt0, int greedy, int batch,
               hipStream_t s) {
  return dispatch(dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    const int rows = batch * tr_max;
    nar_sample_rows<T><<<(rows + 3) / 4, 256, 0, s>>>(static_cast<const T*>(logits), ldl, lens, resp, tr_max, resp_stride, t_max,
                                                      n_tokens, level, temperature, seed, utt0, greedy, batch);
    D3PM_LAUNCH_CHECK();
    return D3PM_OK;
  });
}

}  // namespace d3pm
