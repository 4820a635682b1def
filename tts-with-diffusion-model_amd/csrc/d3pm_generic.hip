// d3pm_generic.hip -- shape-agnostic kernels of the DiT denoiser (any d_model / head_dim / dtype).
//
// Arithmetic contract: fp32 FMA accumulation, one rounding to the storage dtype at every point
// where the reference's eager model materialises a tensor (see d3pm_kernels.h).  This family is
// the numerical spec of the library: the F32 mode runs on it (logits within 1e-3 of the fp32
// reference), the upstream-native shape (d=32, 16 heads of head_dim 2: MFMA-hostile) runs on it,
// and the MFMA family is cross-checked against it on the GPU.
//
// Reference statements covered (paths under /root/reference/vall_e/vall_e/):
//   embed_rows      ar_discrete.py:753 + the x*mask at :127-128
//   layernorm_rows  nn.LayerNorm calls :131,136,140,153 and the FiLM at :146-156
//   linear_tiled    every nn.Linear / MHA in-out projection (:132,138,142,145,159,776)
//   attention_rows  torch multi_head_attention_forward, need_weights branch (q*sqrt(1/hd), bmm,
//                   softmax, bmm), called at :132,138,142
#include "d3pm_kernels.h"

namespace d3pm {
namespace {

// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void embed_rows(const int32_t* __restrict__ tok, const uint8_t* __restrict__ frame_mask,
                           int canvas, const T* __restrict__ table, T* __restrict__ y, int M, int d,
                           int n_classes) {
  int row = blockIdx.x;
  if (row >= M) return;
  int id = tok[row];
  id = id < 0 ? 0 : (id >= n_classes ? n_classes - 1 : id);
  bool live = frame_mask[row % canvas] != 0;
  const T* src = table + static_cast<size_t>(id) * d;
  T* dst = y + static_cast<size_t>(row) * d;
  for (int c = threadIdx.x; c < d; c += blockDim.x) dst[c] = live ? src[c] : static_cast<T>(0.f);
}

// n_q > 1 (d3pm_shape.n_q, this build's extension): a frame's row is the sum of its n_q level embeddings, accumulated in fp32 in
// level order and rounded once -- what MultiEmbedding does for the prompt levels (base.py:244-274)
template <typename T>
__global__ void embed_levels_rows(const int32_t* __restrict__ tok, const uint8_t* __restrict__ frame_mask, int canvas,
                                  const T* __restrict__ tables, T* __restrict__ y, int M, int d, int n_classes, int n_q) {
  const int row = blockIdx.x;
  if (row >= M) return;
  const bool live = frame_mask[row % canvas] != 0;
  T* dst = y + static_cast<size_t>(row) * d;
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    float acc = 0.f;
    for (int l = 0; l < n_q; ++l) {
      int id = tok[static_cast<size_t>(row) * n_q + l];
      id = id < 0 ? 0 : (id >= n_classes ? n_classes - 1 : id);
      acc += static_cast<float>(tables[(static_cast<size_t>(l) * n_classes + id) * d + c]);
    }
    dst[c] = live ? static_cast<T>(acc) : static_cast<T>(0.f);
  }
}

// the same gather with 16-byte accesses: one wave per row, four rows per workgroup (rows of 16-byte multiples)
template <typename T>
__global__ __launch_bounds__(256) void embed_rows_vec(const int32_t* __restrict__ tok, const uint8_t* __restrict__ frame_mask,
                                                      int canvas, const T* __restrict__ table, T* __restrict__ y, int M,
                                                      int d, int n_classes) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= M) return;
  int id = tok[row];
  id = id < 0 ? 0 : (id >= n_classes ? n_classes - 1 : id);
  const bool live = frame_mask[row % canvas] != 0;
  const uint4* src = reinterpret_cast<const uint4*>(table + static_cast<size_t>(id) * d);
  uint4* dst = reinterpret_cast<uint4*>(y + static_cast<size_t>(row) * d);
  const int n16 = d * static_cast<int>(sizeof(T)) / 16;
  for (int c = lane; c < n16; c += kWave) dst[c] = live ? src[c] : uint4{0u, 0u, 0u, 0u};
}

// ------------------------------------------------------------------------------------------
// one wave per row; two-pass moments in fp32
template <typename T>
__global__ void layernorm_rows(const T* __restrict__ x, T* __restrict__ y, const T* __restrict__ w,
                               const T* __restrict__ b, const T* __restrict__ w2,
                               const T* __restrict__ b2, T* __restrict__ y2,
                               const T* __restrict__ film, int M, int d, float eps) {
  int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int row = blockIdx.x * (blockDim.x >> 6) + wave;
  if (row >= M) return;
  const T* xr = x + static_cast<size_t>(row) * d;
  float s = 0.f;
  for (int c = lane; c < d; c += kWave) s += ldf(xr + c);
  float mean = wave_sum(s) / static_cast<float>(d);
  float v = 0.f;
  for (int c = lane; c < d; c += kWave) {
    float t = ldf(xr + c) - mean;
    v += t * t;
  }
  float rstd = rsqrtf(wave_sum(v) / static_cast<float>(d) + eps);
  for (int c = lane; c < d; c += kWave) {
    float n = (ldf(xr + c) - mean) * rstd;
    float o = rn<T>(n * ldf(w + c) + ldf(b + c));
    if (film) {
      float sc = rn<T>(1.0f + ldf(film + c));
      o = rn<T>(rn<T>(o * sc) + ldf(film + d + c));
    }
    stf(y + static_cast<size_t>(row) * d + c, o);
    if (y2) stf(y2 + static_cast<size_t>(row) * d + c, n * ldf(w2 + c) + ldf(b2 + c));
  }
}

// ------------------------------------------------------------------------------------------
// 64x64 output tile / 256 threads, 4x4 per thread, BK = 16, fp32 FMA chain in k order.
constexpr int LT = 64, LK = 16;

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }

template <typename T>
__global__ __launch_bounds__(256) void linear_tiled(const T* __restrict__ X, int ldx,
                                                    const T* __restrict__ W,
                                                    const T* __restrict__ bias, T* Y,
                                                    int ldy, const T* R1, const T* R2, int ldr,
                                                    const uint8_t* __restrict__ row_mask,
                                                    int mask_period, int M, int N, int K, int act) {
  __shared__ float As[LK][LT + 1];
  __shared__ float Bs[LK][LT + 1];
  const int tid = threadIdx.x;
  const int m0 = blockIdx.y * LT, n0 = blockIdx.x * LT;
  const int lr = tid >> 2, lk = (tid & 3) * 4;   // loader: row 0..63, k offset 0,4,8,12
  const int ty = tid >> 4, tx = tid & 15;        // compute: rows ty*4.., cols tx*4..
  float acc[4][4] = {};
  for (int k0 = 0; k0 < K; k0 += LK) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int k = k0 + lk + i;
      int m = m0 + lr, n = n0 + lr;
      As[lk + i][lr] = (m < M && k < K) ? ldf(X + static_cast<size_t>(m) * ldx + k) : 0.f;
      Bs[lk + i][lr] = (n < N && k < K) ? ldf(W + static_cast<size_t>(n) * K + k) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < LK; ++k) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = As[k][ty * 4 + i]; b[i] = Bs[k][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + ty * 4 + i;
    if (m >= M) continue;
    float mk = row_mask ? (row_mask[m % mask_period] ? 1.f : 0.f) : 1.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int n = n0 + tx * 4 + j;
      if (n >= N) continue;
      float v = rn<T>(acc[i][j] + (bias ? ldf(bias + n) : 0.f));
      if (act == ACT_GELU) v = rn<T>(gelu_erf(v));
      else if (act == ACT_RELU) v = fmaxf(v, 0.f);
      else if (act == ACT_SILU) v = rn<T>(v / (1.0f + expf(-v)));
      if (R1) {
        float r = ldf(R1 + static_cast<size_t>(m) * ldr + n);
        if (R2) r = rn<T>(r + ldf(R2 + static_cast<size_t>(m) * ldr + n));
        v = rn<T>(r + v);
      }
      stf(Y + static_cast<size_t>(m) * ldy + n, v * mk);
    }
  }
}

// ------------------------------------------------------------------------------------------
// one wave per query row, lanes over keys for the scores / softmax, lanes over (column, key
// group) for P.V.  Scores, normalised probabilities and the output are rounded to the storage
// dtype like the eager bmm/softmax/bmm chain.  LDS: per wave [S] floats + [hd] floats.
template <typename T> struct alignas(16) Vec16 { T v[16 / sizeof(T)]; };

template <typename T>
__global__ __launch_bounds__(256) void attention_rows(const T* __restrict__ Q, int ldq,
                                                      const T* __restrict__ Kp,
                                                      const T* __restrict__ Vp, int ldkv,
                                                      T* __restrict__ O, int ldo, int Tq, int S_pad,
                                                      int hd, float scale, int q_per_wave,
                                                      const int32_t* __restrict__ key_len) {
  extern __shared__ float smem[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  float* sc = smem + static_cast<size_t>(wave) * (S_pad + hd);
  float* qs = sc + S_pad;
  const int b = blockIdx.z, h = blockIdx.y;
  const int S = key_len ? min(key_len[b], S_pad) : S_pad;     // valid keys of this utterance
  const int hdp = hd < kWave ? hd : kWave;       // columns handled per pass
  const int G = kWave / hdp;                      // key groups per pass
  const T* Kb = Kp + static_cast<size_t>(b) * S_pad * ldkv + h * hd;
  const T* Vb = Vp + static_cast<size_t>(b) * S_pad * ldkv + h * hd;
  constexpr int VEC = 16 / static_cast<int>(sizeof(T));
  const bool vec_ok = hd % VEC == 0 && ldkv % VEC == 0 && (reinterpret_cast<uintptr_t>(Kp) & 15) == 0;   // block-uniform
  for (int qi = 0; qi < q_per_wave; ++qi) {
    const int i = (blockIdx.x * nw + wave) * q_per_wave + qi;
    const bool active = i < Tq;                    // wave-uniform; barriers stay block-uniform
    const T* qr = Q + (static_cast<size_t>(b) * Tq + (active ? i : 0)) * ldq + h * hd;
    if (active)
      for (int c = lane; c < hd; c += kWave) qs[c] = rn<T>(ldf(qr + c) * scale);
    __syncthreads();
    if (active) {
      float mx = -INFINITY;
      for (int j = lane; j < S; j += kWave) {
        const T* kr = Kb + static_cast<size_t>(j) * ldkv;
        float s = 0.f;
        if (vec_ok) {   // same FMA chain in the same order, the key row just arrives 16 bytes at a time
          for (int c = 0; c < hd; c += VEC) {
            const Vec16<T> kv = *reinterpret_cast<const Vec16<T>*>(kr + c);
#pragma unroll
            for (int e = 0; e < VEC; ++e) s = fmaf(qs[c + e], static_cast<float>(kv.v[e]), s);
          }
        } else {
          for (int c = 0; c < hd; ++c) s = fmaf(qs[c], ldf(kr + c), s);
        }
        s = rn<T>(s);
        sc[j] = s;
        mx = fmaxf(mx, s);
      }
      mx = wave_max(mx);
      float sum = 0.f;
      for (int j = lane; j < S; j += kWave) {
        float e = expf(sc[j] - mx);
        sc[j] = e;
        sum += e;
      }
      sum = wave_sum(sum);
      for (int j = lane; j < S; j += kWave) sc[j] = rn<T>(sc[j] / sum);
    }
    __syncthreads();
    if (active) {
      T* orow = O + (static_cast<size_t>(b) * Tq + i) * ldo + h * hd;
      for (int cc = 0; cc < hd; cc += kWave) {
        int c = cc + (lane % hdp), grp = lane / hdp;
        float acc = 0.f;
        for (int j = grp; j < S; j += G) acc = fmaf(sc[j], ldf(Vb + static_cast<size_t>(j) * ldkv + c), acc);
        for (int off = hdp; off < kWave; off <<= 1) acc += __shfl_xor(acc, off, kWave);
        if (grp == 0) stf(orow + c, acc);
      }
    }
    __syncthreads();
  }
}

// text condition rows: W[tok] + PE(position 0) (the x.shape[0] quirk, ar_discrete.py:89,741)
template <typename T>
__global__ void cond_text_rows(const int32_t* __restrict__ tok, const T* __restrict__ table, const T* __restrict__ pe0,
                               T* __restrict__ y, int rows, int d, int n_classes) {
  int row = blockIdx.x;
  if (row >= rows) return;
  int id = tok[row];
  id = id < 0 ? 0 : (id >= n_classes ? n_classes - 1 : id);
  for (int c = threadIdx.x; c < d; c += blockDim.x)
    stf(y + static_cast<size_t>(row) * d + c, ldf(table + static_cast<size_t>(id) * d + c) + ldf(pe0 + c));
}

// prompt condition rows: sum over quantizer levels (fp32, one rounding: base.py:255-274) + PE(position) (ar_discrete.py:745)
template <typename T>
__global__ void cond_prompt_rows(const int32_t* __restrict__ codes, int n_levels, const T* __restrict__ tables,
                                 const T* __restrict__ pe, T* __restrict__ y, int rows, int s_prompt, int d, int n_classes) {
  int row = blockIdx.x;
  if (row >= rows) return;
  const int32_t* cr = codes + static_cast<size_t>(row) * n_levels;
  const T* per = pe + static_cast<size_t>(row % s_prompt) * d;
  for (int c = threadIdx.x; c < d; c += blockDim.x) {
    float acc = 0.f;
    for (int l = 0; l < n_levels; ++l) {
      int id = cr[l];
      if (id < 0) continue;                       // level absent in the prompt: contributes nothing (zero one-hot row)
      id = id >= n_classes ? n_classes - 1 : id;
      acc += ldf(tables + (static_cast<size_t>(l) * n_classes + id) * d + c);
    }
    stf(y + static_cast<size_t>(row) * d + c, rn<T>(acc) + ldf(per + c));
  }
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

int embed_tokens(int dtype, const EmbedArgs& a, hipStream_t s) {
  return dispatch(dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    if (a.n_q > 1) {
      embed_levels_rows<T><<<a.M, a.d >= 256 ? 256 : 64, 0, s>>>(a.tokens, a.frame_mask, a.canvas, static_cast<const T*>(a.table),
                                                              static_cast<T*>(a.Y), a.M, a.d, a.n_classes, a.n_q);
      D3PM_LAUNCH_CHECK();
      return D3PM_OK;
    }
    const bool vec = (a.d * sizeof(T)) % 16 == 0 && reinterpret_cast<uintptr_t>(a.table) % 16 == 0 &&
                     reinterpret_cast<uintptr_t>(a.Y) % 16 == 0 && a.d * sizeof(T) >= 256;
    if (vec) {
      embed_rows_vec<T><<<(a.M + 3) / 4, 256, 0, s>>>(a.tokens, a.frame_mask, a.canvas, static_cast<const T*>(a.table),
                                                      static_cast<T*>(a.Y), a.M, a.d, a.n_classes);
    } else {
      int threads = a.d >= 256 ? 256 : (a.d >= 128 ? 128 : 64);
      embed_rows<T><<<a.M, threads, 0, s>>>(a.tokens, a.frame_mask, a.canvas, static_cast<const T*>(a.table),
                                            static_cast<T*>(a.Y), a.M, a.d, a.n_classes);
    }
    D3PM_LAUNCH_CHECK();
    return D3PM_OK;
  });
}

int cond_embed_text(int dtype, const int32_t* tok, const void* table, const void* pe0, void* y, int rows, int d,
                    int n_classes, hipStream_t s) {
  return dispatch(dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    cond_text_rows<T><<<rows, d >= 256 ? 256 : 64, 0, s>>>(tok, static_cast<const T*>(table), static_cast<const T*>(pe0),
                                                        static_cast<T*>(y), rows, d, n_classes);
    D3PM_LAUNCH_CHECK();
    return D3PM_OK;
  });
}

int cond_embed_prompt(int dtype, const int32_t* codes, int n_levels, const void* tables, const void* pe, void* y,
                      int rows, int s_prompt, int d, int n_classes, hipStream_t s) {
  return dispatch(dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    cond_prompt_rows<T><<<rows, d >= 256 ? 256 : 64, 0, s>>>(codes, n_levels, static_cast<const T*>(tables),
                                                          static_cast<const T*>(pe), static_cast<T*>(y), rows, s_prompt,
                                                          d, n_classes);
    D3PM_LAUNCH_CHECK();
    return D3PM_OK;
  });
}

int generic_layernorm(int dtype, const LayerNormArgs& a, hipStream_t s) {
  return dispatch(dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    int rows_per_block = 4;
    layernorm_rows<T><<<(a.M + rows_per_block - 1) / rows_per_block, rows_per_block * kWave, 0, s>>>(
        static_cast<const T*>(a.X), static_cast<T*>(a.Y), static_cast<const T*>(a.w),
        static_cast<const T*>(a.b), static_cast<const T*>(a.w2), static_cast<const T*>(a.b2),
        static_cast<T*>(a.Y2), static_cast<const T*>(a.film), a.M, a.d, a.eps);
    D3PM_LAUNCH_CHECK();
    return D3PM_OK;
  });
}

int generic_linear(int dtype, const LinearArgs& a, hipStream_t s) {
  return dispatch(dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    dim3 grid((a.N + LT - 1) / LT, (a.M + LT - 1) / LT);
    linear_tiled<T><<<grid, 256, 0, s>>>(static_cast<const T*>(a.X), a.ldx, static_cast<const T*>(a.W),
                                         static_cast<const T*>(a.bias), static_cast<T*>(a.Y), a.ldy,
                                         static_cast<const T*>(a.R1), static_cast<const T*>(a.R2), a.ldr,
                                         a.row_mask, a.mask_period, a.M, a.N, a.K, a.act);
    D3PM_LAUNCH_CHECK();
    return D3PM_OK;
  });
}

int generic_attention(int dtype, const AttnArgs& a, hipStream_t s) {
  bool pow2 = (a.hd & (a.hd - 1)) == 0;
  D3PM_REQUIRE((a.hd <= kWave && pow2) || a.hd % kWave == 0, D3PM_E_SHAPE,
               "generic attention needs head_dim a power of two <= 64 or a multiple of 64 (got %d)", a.hd);
  return dispatch(dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    const int nw = 4, qpw = 4;
    dim3 grid((a.Tq + nw * qpw - 1) / (nw * qpw), a.H, a.B);
    size_t lds = static_cast<size_t>(nw) * (a.S + a.hd) * sizeof(float);
    D3PM_REQUIRE(lds <= 64 * 1024, D3PM_E_SHAPE, "generic attention: %d keys exceed the LDS budget", a.S);
    attention_rows<T><<<grid, nw * kWave, lds, s>>>(static_cast<const T*>(a.Q), a.ldq, static_cast<const T*>(a.K),
                                                   static_cast<const T*>(a.V), a.ldkv, static_cast<T*>(a.O),
                                                   a.ldo, a.Tq, a.S, a.hd, a.scale, qpw, a.key_len);
    D3PM_LAUNCH_CHECK();
    return D3PM_OK;
  });
}

}  // namespace d3pm
