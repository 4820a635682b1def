// d3pm_fold.hip -- LayerNorm folded into the projection that consumes it: weight preparation and the row-moment producers
// that are not GEMM epilogues.
//
// DiTBlock.forward applies every LayerNorm directly in front of a Linear (/root/reference/vall_e/vall_e/ar_discrete.py:131-132
// norm1 -> attn in-projection, :136-142 norm2 | norm22 -> cross_attn's query rows, :145-159 norm3 + FiLM -> mlp.fc1), so
//     LN(x) W^T + b = rstd_r (x_r . W'^T - mean_r s) + b',   W' = W o gamma,  s_n = sum_k W'[n][k],  b' = W beta + b
// (d3pm_mfma_tile.h, EPI_LNF).  With FiLM the per-column factor is gamma_k rn(1 + scale_t[k]) and the constant
// beta_k rn(1 + scale_t[k]) + shift_t[k]: it depends on the timestep, so fc1's W' is rebuilt for the timestep at the top of every
// denoiser evaluation -- ONE launch for all layers, 2 x 12 MB through the chip at d = 512 (~6 us), into the caller's workspace.
// (A table of every (layer, t) copy, 1.27 GB, was measured first: each fc1 launch then streamed its 2 MB from HBM cold and ran
// 7.5 us slower than with the weights resident in the Infinity Cache -- 45 us per iteration against the 6 of rebuilding.)
// W' is rounded to the storage type once; s is summed over the ROUNDED W' (so that x . W' - mean s cancels exactly what the matrix
// pipe accumulated), b' over the unrounded products, both in fp32.
#include "d3pm_kernels.h"
#include "d3pm_fold_rows.h"

namespace d3pm {
namespace {

// one wave per output row (t, n): W row n [K], gamma / beta [K], optional FiLM row film + t * film_ld = (scale [K] | shift [K])
template <typename T>
__global__ __launch_bounds__(256) void fold_rows(const T* __restrict__ W, const T* __restrict__ bias, const T* __restrict__ gamma,
                                                 const T* __restrict__ beta, const T* __restrict__ film, long film_ld, int n_rows,
                                                 int n_t, int K, T* __restrict__ Wf, float* __restrict__ s_out, float* __restrict__ b_out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long r = static_cast<long>(blockIdx.x) * 4 + wave;
  if (r >= static_cast<long>(n_rows) * n_t) return;
  const int t = static_cast<int>(r / n_rows), n = static_cast<int>(r % n_rows);
  const T* wrow = W + static_cast<size_t>(n) * K;
  const T* frow = film ? film + static_cast<size_t>(t) * film_ld : nullptr;
  T* orow = Wf + static_cast<size_t>(r) * K;
  float s = 0.f, b = 0.f;
  for (int k = lane * 8; k < K; k += 512) {
    const Vec8<T> w8 = *reinterpret_cast<const Vec8<T>*>(wrow + k), g8 = *reinterpret_cast<const Vec8<T>*>(gamma + k),
                  b8 = *reinterpret_cast<const Vec8<T>*>(beta + k);
    Vec8<T> sc8{}, sh8{}, o8;
    if (frow) { sc8 = *reinterpret_cast<const Vec8<T>*>(frow + k); sh8 = *reinterpret_cast<const Vec8<T>*>(frow + K + k); }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float w = static_cast<float>(w8.v[i]);
      float g = static_cast<float>(g8.v[i]), c = static_cast<float>(b8.v[i]);
      if (frow) {                         // FiLM as the eager model applies it: (1 + scale) rounded to the storage type first (:146-156)
        const float gg = rn<T>(1.0f + static_cast<float>(sc8.v[i]));
        g *= gg;
        c = __builtin_fmaf(c, gg, static_cast<float>(sh8.v[i]));
      }
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
    b_out[r] = b + (bias ? static_cast<float>(bias[n]) : 0.f);
  }
}

// fc1 of every layer at one timestep, one launch (d3pm_fold_rows.h)
template <typename T>
__global__ __launch_bounds__(256) void fold_rows_layers(FoldStepPtrs p, const T* __restrict__ film_t, int n_rows, int n_layers, int K,
                                                        T* __restrict__ Wf, float* __restrict__ s_out, float* __restrict__ b_out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + wave;
  if (r >= n_rows * n_layers) return;
  fold_layer_row<T>(p, film_t, n_rows, K, r, lane, Wf, s_out, b_out);
}

// (sum, sum of squares) of every 32-column part of every row, in the layout of d3pm_mfma_tile.h stats_index.  One wave per row,
// lane L of pass j owns the 16-byte chunk 64 j + L; the four lanes of a quad own one part.  Used where the residual rows do not
// come out of a GEMM epilogue: the token embedding in front of the first block (ar_discrete.py:753), and as a stand-alone op.
template <typename T>
__global__ __launch_bounds__(256) void row_stats(const T* __restrict__ x, int ldx, int M, int d, float* __restrict__ stats) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= M) return;
  typedef float float2v __attribute__((ext_vector_type(2)));
  const int parts = d >> 5;
  for (int c = lane; c < (d >> 3); c += kWave) {
    const Vec8<T> raw = *reinterpret_cast<const Vec8<T>*>(x + static_cast<size_t>(row) * ldx + c * 8);
    float a, q;
    part_moments(raw, a, q);
    if ((lane & 3) == 0) *reinterpret_cast<float2v*>(stats + stats_index_dev(static_cast<size_t>(row), c >> 2, parts)) = float2v{a, q};
  }
}

// embed_rows_vec (d3pm_generic.hip) + the moments of the gathered rows in the same pass
template <typename T>
__global__ __launch_bounds__(256) void embed_rows_stats(const int32_t* __restrict__ tok, const uint8_t* __restrict__ frame_mask,
                                                        int canvas, const T* __restrict__ table, T* __restrict__ y, int M, int d,
                                                        int n_classes, float* __restrict__ stats) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= M) return;
  embed_row_stats<T>(table, tok[row], frame_mask[row % canvas] != 0, y, row, d, n_classes, stats, lane);
}

}  // namespace

bool fold_shape_ok(int dtype, int d) { return (dtype == D3PM_F16 || dtype == D3PM_BF16) && d >= 256 && d % 256 == 0; }

int fold_rows_launch(int dtype, const void* W, const void* bias, const void* gamma, const void* beta, const void* film, long film_ld,
                     int n_rows, int n_t, int K, void* Wf, float* s_out, float* b_out, hipStream_t s) {
  const long rows = static_cast<long>(n_rows) * n_t;
  const dim3 grid(static_cast<unsigned>((rows + 3) / 4)), block(256);
  if (dtype == D3PM_F16)
    fold_rows<f16><<<grid, block, 0, s>>>(static_cast<const f16*>(W), static_cast<const f16*>(bias), static_cast<const f16*>(gamma),
                                          static_cast<const f16*>(beta), static_cast<const f16*>(film), film_ld, n_rows, n_t, K,
                                          static_cast<f16*>(Wf), s_out, b_out);
  else
    fold_rows<bf16><<<grid, block, 0, s>>>(static_cast<const bf16*>(W), static_cast<const bf16*>(bias), static_cast<const bf16*>(gamma),
                                           static_cast<const bf16*>(beta), static_cast<const bf16*>(film), film_ld, n_rows, n_t, K,
                                           static_cast<bf16*>(Wf), s_out, b_out);
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

// fc1 of every block under norm3 + FiLM(t) (ar_discrete.py:145-159): blocks HOST array; film_t = row t of the FiLM table
int fold_fc1_step_launch(int dtype, const d3pm_block_weights* blocks, int n_layers, const void* film_t, int d, void* Wf, float* s_out,
                         float* b_out, hipStream_t s) {
  if (n_layers > 16) {           // beyond the pointer table of one launch: layer by layer
    const size_t es = dtype_size(dtype);
    for (int l = 0; l < n_layers; ++l) {
      const int rc = fold_rows_launch(dtype, blocks[l].fc1_w, blocks[l].fc1_b, blocks[l].norm3_w, blocks[l].norm3_b,
                                      static_cast<const char*>(film_t) + static_cast<size_t>(l) * 2 * d * es, 0, 4 * d, 1, d,
                                      static_cast<char*>(Wf) + static_cast<size_t>(l) * 4 * d * d * es, s_out + static_cast<size_t>(l) * 4 * d,
                                      b_out + static_cast<size_t>(l) * 4 * d, s);
      if (rc != D3PM_OK) return rc;
    }
    return D3PM_OK;
  }
  FoldStepPtrs p{};
  for (int l = 0; l < n_layers; ++l) { p.W[l] = blocks[l].fc1_w; p.bias[l] = blocks[l].fc1_b; p.gamma[l] = blocks[l].norm3_w; p.beta[l] = blocks[l].norm3_b; }
  const dim3 grid(static_cast<unsigned>((4 * d * n_layers + 3) / 4)), block(256);
  if (dtype == D3PM_F16)
    fold_rows_layers<f16><<<grid, block, 0, s>>>(p, static_cast<const f16*>(film_t), 4 * d, n_layers, d, static_cast<f16*>(Wf), s_out, b_out);
  else
    fold_rows_layers<bf16><<<grid, block, 0, s>>>(p, static_cast<const bf16*>(film_t), 4 * d, n_layers, d, static_cast<bf16*>(Wf), s_out, b_out);
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

int row_stats_launch(int dtype, const void* x, int ldx, int M, int d, float* stats, hipStream_t s) {
  const dim3 grid(static_cast<unsigned>((M + 3) / 4)), block(256);
  if (dtype == D3PM_F16) row_stats<f16><<<grid, block, 0, s>>>(static_cast<const f16*>(x), ldx, M, d, stats);
  else row_stats<bf16><<<grid, block, 0, s>>>(static_cast<const bf16*>(x), ldx, M, d, stats);
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

int embed_tokens_stats(int dtype, const EmbedArgs& a, float* stats, hipStream_t s) {
  if (a.n_q > 1) {                 // level-summed rows (extension): the generic gather, then the moments of the rows it wrote
    int rc = embed_tokens(dtype, a, s);
    return rc != D3PM_OK ? rc : row_stats_launch(dtype, a.Y, a.d, a.M, a.d, stats, s);
  }
  const dim3 grid(static_cast<unsigned>((a.M + 3) / 4)), block(256);
  if (dtype == D3PM_F16)
    embed_rows_stats<f16><<<grid, block, 0, s>>>(a.tokens, a.frame_mask, a.canvas, static_cast<const f16*>(a.table), static_cast<f16*>(a.Y),
                                                 a.M, a.d, a.n_classes, stats);
  else
    embed_rows_stats<bf16><<<grid, block, 0, s>>>(a.tokens, a.frame_mask, a.canvas, static_cast<const bf16*>(a.table),
                                                  static_cast<bf16*>(a.Y), a.M, a.d, a.n_classes, stats);
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

}  // namespace d3pm
