// d3pm_sample.hip -- D3PM absorbing-state transition / posterior / categorical sampling kernels.
//
// Replaces (paths under /root/reference/vall_e/vall_e/):
//   posterior_sample_rows  AR.p_sample + q_posterior_logits + _at + _at_onehot (ar_discrete.py:337-420)
//   q_sample_rows          AR.q_sample + q_probs                               (ar_discrete.py:467-502)
//   uniform_rows           the torch.rand draws at ar_discrete.py:402,480 (Philox stream instead)
//
// The reference multiplies one-hot / softmax rows into dense [1025,1025] fp16 tables; every table
// is d*I + c*1 e_M^T with row M = e_M (proven on the reference's tables in
// tests/golden/make_golden.py), so per row the work is O(K):
//     fact1_j = d_t [j==x] (x != M)      |  c_t (j != M), 1 (j == M)        (x == M)
//     fact2_j = rn16(p_j * dbar_{t-1}) (j != M),  rn16(cbar_{t-1} * sum_{k!=M} p_k + p_M) (j == M)
//     out_j   = rn16(log16(rn16(fact1_j+eps)) + log16(rn16(fact2_j+eps)))
//     x_{t-1} = argmax_j fp32(out_j) + gumbel(u_j)
// with p = rn16(softmax_fp32(rn16 logits)).  All fp16 rounding points of the eager fp16 model are
// kept; HBM traffic per row is the K logits in and one id out (plus 4 B of x_t).
//
// Mapping: one wave per row; lane l owns class groups g = l, l+64, .. (4 consecutive classes per
// Philox call), i.e. 5 groups x 4 classes = 20 registers of logits for K = 1025.
#include <cmath>

#include "d3pm_kernels.h"
#include "d3pm_fold_rows.h"
#include "d3pm_sample_row.h"

namespace d3pm {
namespace {

template <typename T>
__global__ __launch_bounds__(256) void posterior_sample_rows(
    const T* __restrict__ logits, int ldl, const int32_t* x_t, int32_t* x_next,
    int32_t* x_next2, uint16_t* __restrict__ post_out, int rows, int K, int mask_id,
    uint64_t seed, const uint64_t* __restrict__ seed_hbm, uint32_t row0, int greedy, PosteriorConsts pc, int n_q) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + wave;
  if (row >= rows) return;
  if (seed_hbm) seed = *seed_hbm;
  // n_q > 1 (d3pm_shape.n_q): row = frame row * n_q + level; the level-0 token of a frame draws the noise the level-0-only
  // path draws, level l > 0 draws from Philox stream 16 + l at the same (frame row, t)
  const int frow = n_q > 1 ? row / n_q : row, level = row - frow * (n_q > 1 ? n_q : 1);
  const uint32_t strm = level ? 16u + static_cast<uint32_t>(level) : 0u;
  int best_j;
  if (K == 1025 && mask_id < 1024 && !post_out)        // kernel-uniform: the predicate-free routine for the reference's class count (same bits)
    best_j = sample_row_1025<T>(logits + static_cast<size_t>(row) * ldl, mask_id, x_t[row], seed, row0 + static_cast<uint32_t>(frow), greedy, pc, lane, strm);
  else
    best_j = sample_row<T>(logits + static_cast<size_t>(row) * ldl, K, mask_id, x_t[row], seed, row0 + static_cast<uint32_t>(frow), greedy, pc,
                           post_out ? post_out + static_cast<size_t>(row) * K : nullptr, lane, strm);
  if (lane == 0) {
    x_next[row] = best_j;
    if (x_next2) x_next2[row] = best_j;
  }
}

// The sampler of iteration t and the preparation of iteration t - 1 in one launch (NextIterPrep, d3pm_kernels.h): workgroups
// [0, sample_blocks) draw x_{t-1} for four rows each and at once gather those rows' embeddings into the residual stream with
// their moments (the id is in every lane after the wave argmax); the workgroups behind them rebuild fc1 o norm3 o FiLM(t - 1) of
// every block.  The two halves are independent (one VALU-bound, one a 24-MB stream), so the launch costs the longer of them: at one
// utterance 12.4 + 8.7 + 5.0 us of launches become ~13, at 32 utterances 94 + 10.8 + 10.5 become ~97.
template <typename T>
__global__ __launch_bounds__(256) void posterior_sample_prep_rows(
    const T* __restrict__ logits, int ldl, const int32_t* x_t, int32_t* x_next, int32_t* x_next2, int rows, int K, int mask_id,
    uint64_t seed, const uint64_t* __restrict__ seed_hbm, uint32_t row0, int greedy, PosteriorConsts pc, int canvas, int sample_blocks,
    const T* __restrict__ table, T* __restrict__ xres, float* __restrict__ stats, const uint8_t* __restrict__ frame_mask, int d,
    FoldStepPtrs fp, const T* __restrict__ film_t, int n_layers, T* __restrict__ Wf, float* __restrict__ s_out, float* __restrict__ b_out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (static_cast<int>(blockIdx.x) >= sample_blocks) {
    const int r = (blockIdx.x - sample_blocks) * 4 + wave;
    if (r < 4 * d * n_layers) fold_layer_row<T>(fp, film_t, 4 * d, d, r, lane, Wf, s_out, b_out);
    return;
  }
  const int row = blockIdx.x * 4 + wave;
  if (row >= rows) return;
  if (seed_hbm) seed = *seed_hbm;
  const int best_j = (K == 1025 && mask_id < 1024)
      ? sample_row_1025<T>(logits + static_cast<size_t>(row) * ldl, mask_id, x_t[row], seed, row0 + static_cast<uint32_t>(row), greedy, pc, lane, 0u)
      : sample_row<T>(logits + static_cast<size_t>(row) * ldl, K, mask_id, x_t[row], seed, row0 + static_cast<uint32_t>(row), greedy, pc, nullptr, lane, 0u);
  if (lane == 0) {
    x_next[row] = best_j;
    if (x_next2) x_next2[row] = best_j;
  }
  embed_row_stats<T>(table, best_j, frame_mask[row % canvas] != 0, xres, row, d, K, stats, lane);
}

// forward noising: logits are log16(rn16(row_of_Qbar_t + eps)) with at most three distinct values
__global__ __launch_bounds__(256) void q_sample_rows(const int32_t* __restrict__ x0, int32_t* __restrict__ out,
                                                     const uint8_t* __restrict__ frame_mask, int canvas,
                                                     int rows, int K, int mask_id, uint64_t seed,
                                                     uint32_t row0, int t, float log_dbar, float log_cbar,
                                                     float log_zero, float log_one) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + wave;
  if (row >= rows) return;
  const int x = x0[row];
  const int groups = (K + 3) >> 2;
  int best_j = 0;
  float best_v = -INFINITY;
  for (int g = lane; g < groups; g += kWave) {
    float u[4];
    noise4(seed, static_cast<uint32_t>(g), row0 + static_cast<uint32_t>(row), static_cast<uint32_t>(t), 1u, u);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      int j = g * 4 + w;
      if (j >= K) continue;
      float l;
      if (x == mask_id) l = (j == mask_id) ? log_one : log_zero;
      else l = (j == x) ? log_dbar : (j == mask_id ? log_cbar : log_zero);
      float v = l + gumbel(u[w]);
      if (v > best_v) { best_v = v; best_j = j; }
    }
  }
  wave_argmax(best_v, best_j);
  if (lane == 0) out[row] = frame_mask[row % canvas] ? best_j : 0;
}

__global__ void uniform_rows(uint64_t seed, int t, uint32_t row0, int rows, int K, int stream_id,
                             float* __restrict__ out) {
  const int groups = (K + 3) >> 2;
  size_t idx = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= static_cast<size_t>(rows) * groups) return;
  int r = static_cast<int>(idx / groups), g = static_cast<int>(idx % groups);
  float u[4];
  noise4(seed, static_cast<uint32_t>(g), row0 + static_cast<uint32_t>(r), static_cast<uint32_t>(t),
         static_cast<uint32_t>(stream_id), u);
  for (int w = 0; w < 4; ++w)
    if (g * 4 + w < K) out[static_cast<size_t>(r) * K + g * 4 + w] = u[w];
}

}  // namespace

float host_h2f(uint16_t h);
float host_log16(float fact);

int posterior_sample(const SampleArgs& a, hipStream_t s) {
  D3PM_REQUIRE(a.n_classes <= kWave * kMaxGroupsPerLane * 4, D3PM_E_SHAPE,
               "posterior_sample supports up to %d classes", kWave * kMaxGroupsPerLane * 4);
  const int rpb = 4;
  dim3 grid((a.rows + rpb - 1) / rpb), block(rpb * kWave);
#define D3PM_PS(T)                                                                                      \
  posterior_sample_rows<T><<<grid, block, 0, s>>>(static_cast<const T*>(a.logits), a.ldl, a.x_t, a.x_next, \
                                                  a.x_next2, a.posterior_out, a.rows, a.n_classes,        \
                                                  a.mask_id, a.seed, a.seed_hbm, a.row0, a.greedy, a.pc, a.n_q)
  switch (a.logits_dtype) {
    case D3PM_F32: D3PM_PS(float); break;
    case D3PM_F16: D3PM_PS(f16); break;
    case D3PM_BF16: D3PM_PS(bf16); break;
    default: set_error("unknown logits dtype %d", a.logits_dtype); return D3PM_E_ARG;
  }
#undef D3PM_PS
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

bool posterior_sample_prep_supported(const SampleArgs& a, const NextIterPrep& n) {
  return a.n_q == 1 && !a.posterior_out && a.logits_dtype == n.dtype && (n.dtype == D3PM_F16 || n.dtype == D3PM_BF16) && n.n_layers <= 16 &&
         n.d % 256 == 0 && a.n_classes <= kWave * kMaxGroupsPerLane * 4 && n.table && n.x && n.stats && n.blocks && n.film_t && n.Wf;
}

int posterior_sample_prep(const SampleArgs& a, const NextIterPrep& n, hipStream_t s) {
  FoldStepPtrs p{};
  for (int l = 0; l < n.n_layers; ++l) {
    p.W[l] = n.blocks[l].fc1_w; p.bias[l] = n.blocks[l].fc1_b; p.gamma[l] = n.blocks[l].norm3_w; p.beta[l] = n.blocks[l].norm3_b;
  }
  const int sample_blocks = (a.rows + 3) / 4, fold_blocks = (4 * n.d * n.n_layers + 3) / 4;
  const dim3 grid(static_cast<unsigned>(sample_blocks + fold_blocks)), block(256);
#define D3PM_PSP(T)                                                                                                                 \
  posterior_sample_prep_rows<T><<<grid, block, 0, s>>>(static_cast<const T*>(a.logits), a.ldl, a.x_t, a.x_next, a.x_next2, a.rows, a.n_classes, \
      a.mask_id, a.seed, a.seed_hbm, a.row0, a.greedy, a.pc, a.canvas, sample_blocks, static_cast<const T*>(n.table), static_cast<T*>(n.x),    \
      n.stats, n.frame_mask, n.d, p, static_cast<const T*>(n.film_t), n.n_layers, static_cast<T*>(n.Wf), n.s_out, n.b_out)
  if (n.dtype == D3PM_F16) D3PM_PSP(f16); else D3PM_PSP(bf16);
#undef D3PM_PSP
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

int q_sample_launch(const d3pm_shape* sh, int batch, const int32_t* x0, int32_t* out, const uint8_t* frame_mask,
                    int t, const d3pm_schedule* sched, uint64_t seed, uint32_t utt0, hipStream_t s) {
  const int rows = batch * sh->canvas, rpb = 4;
  float ld = host_log16(host_h2f(sched->dbar[t])), lc = host_log16(host_h2f(sched->cbar[t]));
  q_sample_rows<<<(rows + rpb - 1) / rpb, rpb * kWave, 0, s>>>(x0, out, frame_mask, sh->canvas, rows, sh->n_classes,
                                                               sh->mask_id, seed, utt0 * sh->canvas, t, ld, lc,
                                                               host_log16(0.f), host_log16(1.f));
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

// Training-side loss rows (ar_discrete.py:683-690): x = logits * mask (a padded frame becomes an all-zero row, i.e. a
// uniform prediction), target = x0 * mask, row loss = logsumexp(x) - x[target]; the caller takes the mean over the
// canvas.  One wave per row, fp32 arithmetic on the logits as stored.
template <typename T>
__global__ __launch_bounds__(256) void ce_loss_rows(const T* __restrict__ logits, int ldl, const int32_t* __restrict__ targets,
                                                    const uint8_t* __restrict__ frame_mask, int canvas, int rows, int K,
                                                    float* __restrict__ row_loss) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + wave;
  if (row >= rows) return;
  if (!frame_mask[row % canvas]) {
    if (lane == 0) row_loss[row] = logf(static_cast<float>(K));
    return;
  }
  const T* lr = logits + static_cast<size_t>(row) * ldl;
  float mx = -INFINITY;
  for (int j = lane; j < K; j += kWave) mx = fmaxf(mx, static_cast<float>(lr[j]));
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < K; j += kWave) sum += expf(static_cast<float>(lr[j]) - mx);
  sum = wave_sum(sum);
  int tg = targets[row];
  tg = tg < 0 ? 0 : (tg >= K ? K - 1 : tg);
  if (lane == 0) row_loss[row] = mx + logf(sum) - static_cast<float>(lr[tg]);
}

int ce_loss_launch(int dtype, const void* logits, int ldl, const int32_t* targets, const uint8_t* frame_mask, int canvas,
                   int rows, int K, float* row_loss, hipStream_t s) {
  const dim3 grid((rows + 3) / 4), block(256);
  switch (dtype) {
    case D3PM_F32: ce_loss_rows<float><<<grid, block, 0, s>>>(static_cast<const float*>(logits), ldl, targets, frame_mask, canvas, rows, K, row_loss); break;
    case D3PM_F16: ce_loss_rows<f16><<<grid, block, 0, s>>>(static_cast<const f16*>(logits), ldl, targets, frame_mask, canvas, rows, K, row_loss); break;
    case D3PM_BF16: ce_loss_rows<bf16><<<grid, block, 0, s>>>(static_cast<const bf16*>(logits), ldl, targets, frame_mask, canvas, rows, K, row_loss); break;
    default: set_error("unknown logits dtype %d", dtype); return D3PM_E_ARG;
  }
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

int uniform_launch(uint64_t seed, int t, uint32_t row0, int rows, int K, int stream_id, float* out, hipStream_t s) {
  size_t n = static_cast<size_t>(rows) * ((K + 3) / 4);
  uniform_rows<<<static_cast<unsigned>((n + 255) / 256), 256, 0, s>>>(seed, t, row0, rows, K, stream_id, out);
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

}  // namespace d3pm
