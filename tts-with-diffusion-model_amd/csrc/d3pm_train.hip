// d3pm_train.hip -- backward-pass building blocks of the training-side D3PM forward.
//
// The reference trains the denoiser with autograd over AR.forward (/root/reference/vall_e/vall_e/ar_discrete.py:588-694:
// q_sample -> DiT blocks :126-161 -> final Linear :776 -> masked cross-entropy :683-690) inside a DeepSpeed engine
// (/root/reference/vall_e/utils/engines.py:144-147 backward + data-parallel gradient all-reduce).  Here the gradient of
// every op of that forward is a hand-written kernel behind a single-op C entry (include/d3pm_hip.h, d3pm_op_*_bwd); the
// host side (vall_e/vall_e/train.py) replays the forward with a stash and walks it backwards, and the data-parallel
// reduction is one bucketed torch.distributed all-reduce (RCCL) of the flat gradient.
//
// Scope: fp32 tensors (the F32 precision mode: the gradient check is against torch.autograd over the oracle in fp32),
// generic kernels sized for clarity -- the sampler is this build's hot path, training is a "next" row -- except the matrix
// products (dX = dY W, dW += dY^T X of every nn.Linear), which run on the fp32 matrix instruction with unchanged bits.  Every kernel accumulates in fp32 in a fixed order except the per-column reductions that use fp32 atomics
// (LayerNorm / embedding gradients), which are order-independent to rounding.
#include <cmath>

#include "d3pm_kernels.h"

namespace d3pm {
namespace {

// C[i][j] = beta * C[i][j] + sum_k A(i,k) * B(k,j),  A(i,k) = A[i*sai + k*sak], B(k,j) = B[k*sbk + j*sbj]
// (any of the four transposition combinations through the strides); 32 x 32 tile per 256 threads, k-step 16
__global__ __launch_bounds__(256) void matmul_strided(const float* __restrict__ A, long sai, long sak, const float* __restrict__ B,
                                                      long sbk, long sbj, float* C, int ldc, int M, int N, int K, float beta,
                                                      const uint8_t* __restrict__ row_mask, int mask_period) {
  __shared__ float sa[16][33], sb[16][33];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  float acc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
  for (int k0 = 0; k0 < K; k0 += 16) {
    for (int e = threadIdx.x; e < 16 * 32; e += 256) {
      const int kk = e >> 5, r = e & 31;
      const int i = i0 + r, j = j0 + r, k = k0 + kk;
      sa[kk][r] = (i < M && k < K) ? A[i * sai + k * sak] : 0.f;
      sb[kk][r] = (j < N && k < K) ? B[k * sbk + j * sbj] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const float a0 = sa[kk][ty], a1 = sa[kk][ty + 16], b0 = sb[kk][tx], b1 = sb[kk][tx + 16];
      acc[0][0] = fmaf(a0, b0, acc[0][0]);
      acc[0][1] = fmaf(a0, b1, acc[0][1]);
      acc[1][0] = fmaf(a1, b0, acc[1][0]);
      acc[1][1] = fmaf(a1, b1, acc[1][1]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int i = i0 + ty + 16 * a, j = j0 + tx + 16 * b;
      if (i < M && j < N) {
        float v = acc[a][b];
        if (row_mask) v *= row_mask[i % mask_period] ? 1.f : 0.f;
        float* c = C + static_cast<size_t>(i) * ldc + j;
        *c = beta == 0.f ? v : fmaf(beta, *c, v);
      }
    }
}

// The same product on the matrix pipe: v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate) is bit for bit a k-ordered fmaf
// chain (MI355X_MICROARCH.md "FP32-input MFMA"), i.e. exactly the chain of matmul_strided above -- same gradients to the last
// bit, at the fp32 matrix rate (157 TFLOP/s peak) instead of one FMA per lane and LDS read.  128 x 128 tile per 256 threads
// (wave = 64 x 64 = 2 x 2 MFMA tiles), k-step 16 through LDS ([k][row] images, padded: the fragment of a lane is ONE float --
// A[i = lane & 31][k = lane >> 5], B[k = lane >> 5][j = lane & 31]); the global reads walk whichever operand index is
// contiguous in memory (all four transposition combinations arrive through the strides).
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void matmul_strided_mfma(const float* __restrict__ A, long sai, long sak, const float* __restrict__ B,
                                                           long sbk, long sbj, float* C, int ldc, int M, int N, int K, float beta,
                                                           const uint8_t* __restrict__ row_mask, int mask_period) {
  constexpr int TM = 128, TN = 128, BKS = 16, LD = 132;          // LD: 128 + 4 floats of padding (conflict-free b32 reads and writes)
  __shared__ float sa[BKS][LD], sb[BKS][LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i0 = blockIdx.y * TM, j0 = blockIdx.x * TN;
  const int wi = (wave >> 1) * 64, wj = (wave & 1) * 64;
  const int fr = lane & 31, fk = lane >> 5;
  floatx16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const bool a_k_contig = sak == 1, b_k_contig = sbk == 1;
  for (int k0 = 0; k0 < K; k0 += BKS) {
#pragma unroll
    for (int q = 0; q < (BKS * TM) / 256; ++q) {
      const int e = tid + 256 * q;
      {
        const int kk = a_k_contig ? (e & 15) : (e >> 7), r = a_k_contig ? (e >> 4) : (e & 127);
        const int i = i0 + r, k = k0 + kk;
        sa[kk][r] = (i < M && k < K) ? A[i * sai + k * sak] : 0.f;
      }
      {
        const int kk = b_k_contig ? (e & 15) : (e >> 7), r = b_k_contig ? (e >> 4) : (e & 127);
        const int j = j0 + r, k = k0 + kk;
        sb[kk][r] = (j < N && k < K) ? B[k * sbk + j * sbj] : 0.f;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k2 = 0; k2 < BKS / 2; ++k2) {
      const float a0 = sa[2 * k2 + fk][wi + fr], a1 = sa[2 * k2 + fk][wi + 32 + fr];
      const float b0 = sb[2 * k2 + fk][wj + fr], b1 = sb[2 * k2 + fk][wj + 32 + fr];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
  // C/D of the 32 x 32 tile: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + wi + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk, j = j0 + wj + b * 32 + fr;
        if (i < M && j < N) {
          float v = acc[a][b][r];
          if (row_mask) v *= row_mask[i % mask_period] ? 1.f : 0.f;
          float* c = C + static_cast<size_t>(i) * ldc + j;
          *c = beta == 0.f ? v : fmaf(beta, *c, v);
        }
      }
}

// out[j] = beta * out[j] + sum_i X[i][j]
__global__ __launch_bounds__(256) void colsum_rows(const float* __restrict__ X, int ldx, int M, int N, float* out, float beta) {
  __shared__ float part[4][64];
  const int c = threadIdx.x & 63, slice = threadIdx.x >> 6, j = blockIdx.x * 64 + c;
  float s = 0.f;
  if (j < N)
    for (int i = slice; i < M; i += 4) s += X[static_cast<size_t>(i) * ldx + j];
  part[slice][c] = s;
  __syncthreads();
  if (slice == 0 && j < N) {
    s = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
    out[j] = beta == 0.f ? s : fmaf(beta, out[j], s);
  }
}

// dU = dM * act'(U): act 1 exact-erf GELU, 2 ReLU, 3 SiLU
__global__ void act_bwd_k(const float* __restrict__ U, const float* __restrict__ dM, float* __restrict__ dU, size_t n, int act) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float u = U[i];
  float g;
  if (act == ACT_GELU) {
    const float cdf = 0.5f * (1.0f + erff(u * 0.70710678118654752f));
    const float pdf = 0.3989422804014327f * expf(-0.5f * u * u);
    g = cdf + u * pdf;
  } else if (act == ACT_RELU) {
    g = u > 0.f ? 1.f : 0.f;
  } else {
    const float sg = 1.0f / (1.0f + expf(-u));
    g = sg * (1.0f + u * (1.0f - sg));
  }
  dU[i] = dM[i] * g;
}

__global__ void mask_rows_k(float* X, int ldx, int M, int N, const uint8_t* __restrict__ mask, int period) {
  const size_t idx = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= static_cast<size_t>(M) * N) return;
  const int i = static_cast<int>(idx / N), j = static_cast<int>(idx % N);
  if (!mask[i % period]) X[static_cast<size_t>(i) * ldx + j] = 0.f;
}

// LayerNorm (+ optional FiLM) backward, one wave per row:  y = xhat * w + b,  out = film ? y * (1 + scale) + shift : y
//   dy = film ? dOut * (1 + scale) : dOut;   dscale += sum_rows dOut * y;   dshift += sum_rows dOut
//   dw += sum_rows dy * xhat;   db += sum_rows dy
//   dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * w;   dX = accumulate ? dX + dx : dx
__global__ __launch_bounds__(256) void layernorm_bwd_rows(const float* __restrict__ X, const float* __restrict__ dOut,
                                                          const float* __restrict__ w, const float* __restrict__ b,
                                                          const float* __restrict__ film, float eps, int M, int d, float* dX,
                                                          int accumulate, float* dw, float* db, float* dfilm) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= M) return;
  const float* x = X + static_cast<size_t>(row) * d;
  const float* go = dOut + static_cast<size_t>(row) * d;
  float s = 0.f;
  for (int c = lane; c < d; c += kWave) s += x[c];
  const float mean = wave_sum(s) / d;
  float v = 0.f;
  for (int c = lane; c < d; c += kWave) { const float t = x[c] - mean; v += t * t; }
  const float rstd = 1.0f / sqrtf(wave_sum(v) / d + eps);
  float sg = 0.f, sgx = 0.f;
  for (int c = lane; c < d; c += kWave) {
    const float xh = (x[c] - mean) * rstd;
    const float dy = film ? go[c] * (1.0f + film[c]) : go[c];
    const float g = dy * w[c];
    sg += g;
    sgx += g * xh;
    atomicAdd(dw + c, dy * xh);
    atomicAdd(db + c, dy);
    if (film) {
      atomicAdd(dfilm + c, go[c] * (xh * w[c] + b[c]));
      atomicAdd(dfilm + d + c, go[c]);
    }
  }
  const float mg = wave_sum(sg) / d, mgx = wave_sum(sgx) / d;
  float* dx = dX + static_cast<size_t>(row) * d;
  for (int c = lane; c < d; c += kWave) {
    const float xh = (x[c] - mean) * rstd;
    const float dy = film ? go[c] * (1.0f + film[c]) : go[c];
    const float r = rstd * (dy * w[c] - mg - xh * mgx);
    dx[c] = accumulate ? dx[c] + r : r;
  }
}

// Attention backward, pass 1: one wave per (utterance, head, query).  P = softmax(scale * q . k^T) recomputed;
// stats[.][0] = logsumexp of the scaled scores, stats[.][1] = D = dO . O = sum_j P_j (dO . v_j);
// dQ = scale * sum_j P_j ((dO . v_j) - D) k_j
__global__ __launch_bounds__(256) void attn_bwd_q_rows(const float* __restrict__ Q, int ldq, const float* __restrict__ Kp,
                                                       const float* __restrict__ Vp, int ldkv, const float* __restrict__ dO, int ldo,
                                                       float* __restrict__ dQ, int lddq, float* __restrict__ stats, int B, int Tq,
                                                       int S, int H, int hd, float scale) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long idx = static_cast<long>(blockIdx.x) * 4 + wave;
  if (idx >= static_cast<long>(B) * H * Tq) return;
  const int i = static_cast<int>(idx % Tq), h = static_cast<int>((idx / Tq) % H), b = static_cast<int>(idx / (static_cast<long>(Tq) * H));
  const float* q = Q + (static_cast<size_t>(b) * Tq + i) * ldq + h * hd;
  const float* go = dO + (static_cast<size_t>(b) * Tq + i) * ldo + h * hd;
  const float* kb = Kp + static_cast<size_t>(b) * S * ldkv + h * hd;
  const float* vb = Vp + static_cast<size_t>(b) * S * ldkv + h * hd;
  float mx = -INFINITY;
  for (int j = lane; j < S; j += kWave) {
    float sc = 0.f;
    for (int c = 0; c < hd; ++c) sc = fmaf(q[c] * scale, kb[static_cast<size_t>(j) * ldkv + c], sc);
    mx = fmaxf(mx, sc);
  }
  mx = wave_max(mx);
  float l = 0.f, dsum = 0.f;
  for (int j = lane; j < S; j += kWave) {
    float sc = 0.f, dp = 0.f;
    for (int c = 0; c < hd; ++c) {
      sc = fmaf(q[c] * scale, kb[static_cast<size_t>(j) * ldkv + c], sc);
      dp = fmaf(go[c], vb[static_cast<size_t>(j) * ldkv + c], dp);
    }
    const float e = expf(sc - mx);
    l += e;
    dsum += e * dp;
  }
  l = wave_sum(l);
  const float D = wave_sum(dsum) / l, lse = mx + logf(l);
  if (lane == 0) { stats[idx * 2] = lse; stats[idx * 2 + 1] = D; }
  float* dq = dQ + (static_cast<size_t>(b) * Tq + i) * lddq + h * hd;
  for (int c0 = 0; c0 < hd; ++c0) {
    float part = 0.f;
    for (int j = lane; j < S; j += kWave) {
      float sc = 0.f, dp = 0.f;
      for (int c = 0; c < hd; ++c) {
        sc = fmaf(q[c] * scale, kb[static_cast<size_t>(j) * ldkv + c], sc);
        dp = fmaf(go[c], vb[static_cast<size_t>(j) * ldkv + c], dp);
      }
      const float p = expf(sc - lse);
      part = fmaf(p * (dp - D), kb[static_cast<size_t>(j) * ldkv + c0], part);
    }
    part = wave_sum(part);
    if (lane == 0) dq[c0] = part * scale;
  }
}

// pass 2: one wave per (utterance, head, key):  dV_j = sum_i P_ij dO_i,  dK_j = scale * sum_i P_ij ((dO_i . v_j) - D_i) q_i
__global__ __launch_bounds__(256) void attn_bwd_kv_rows(const float* __restrict__ Q, int ldq, const float* __restrict__ Kp,
                                                        const float* __restrict__ Vp, int ldkv, const float* __restrict__ dO, int ldo,
                                                        float* __restrict__ dK, float* __restrict__ dV, int lddkv,
                                                        const float* __restrict__ stats, int B, int Tq, int S, int H, int hd,
                                                        float scale, float beta) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long idx = static_cast<long>(blockIdx.x) * 4 + wave;
  if (idx >= static_cast<long>(B) * H * S) return;
  const int j = static_cast<int>(idx % S), h = static_cast<int>((idx / S) % H), b = static_cast<int>(idx / (static_cast<long>(S) * H));
  const float* k = Kp + (static_cast<size_t>(b) * S + j) * ldkv + h * hd;
  const float* v = Vp + (static_cast<size_t>(b) * S + j) * ldkv + h * hd;
  const float* qb = Q + static_cast<size_t>(b) * Tq * ldq + h * hd;
  const float* gb = dO + static_cast<size_t>(b) * Tq * ldo + h * hd;
  const float* st = stats + (static_cast<size_t>(b) * H + h) * Tq * 2;
  float* dk = dK + (static_cast<size_t>(b) * S + j) * lddkv + h * hd;
  float* dv = dV + (static_cast<size_t>(b) * S + j) * lddkv + h * hd;
  for (int c0 = 0; c0 < hd; ++c0) {
    float pk = 0.f, pv = 0.f;
    for (int i = lane; i < Tq; i += kWave) {
      float sc = 0.f, dp = 0.f;
      for (int c = 0; c < hd; ++c) {
        sc = fmaf(qb[static_cast<size_t>(i) * ldq + c] * scale, k[c], sc);
        dp = fmaf(gb[static_cast<size_t>(i) * ldo + c], v[c], dp);
      }
      const float p = expf(sc - st[i * 2]);
      pv = fmaf(p, gb[static_cast<size_t>(i) * ldo + c0], pv);
      pk = fmaf(p * (dp - st[i * 2 + 1]), qb[static_cast<size_t>(i) * ldq + c0], pk);
    }
    pk = wave_sum(pk);
    pv = wave_sum(pv);
    if (lane == 0) {
      dk[c0] = beta == 0.f ? pk * scale : fmaf(beta, dk[c0], pk * scale);
      dv[c0] = beta == 0.f ? pv : fmaf(beta, dv[c0], pv);
    }
  }
}

// d(row loss)/d(logits) of ce_loss_rows: x = logits * mask, dlogits = mask * (softmax(x) - onehot(target)) * gscale
__global__ __launch_bounds__(256) void ce_bwd_rows(const float* __restrict__ logits, int ldl, const int32_t* __restrict__ targets,
                                                   const uint8_t* __restrict__ frame_mask, int canvas, int rows, int K, float gscale,
                                                   float* __restrict__ dlogits, int ldd) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + wave;
  if (row >= rows) return;
  float* dr = dlogits + static_cast<size_t>(row) * ldd;
  if (!frame_mask[row % canvas]) {
    for (int j = lane; j < K; j += kWave) dr[j] = 0.f;
    return;
  }
  const float* lr = logits + static_cast<size_t>(row) * ldl;
  float mx = -INFINITY;
  for (int j = lane; j < K; j += kWave) mx = fmaxf(mx, lr[j]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < K; j += kWave) sum += expf(lr[j] - mx);
  sum = wave_sum(sum);
  int tg = targets[row];
  tg = tg < 0 ? 0 : (tg >= K ? K - 1 : tg);
  for (int j = lane; j < K; j += kWave) dr[j] = (expf(lr[j] - mx) / sum - (j == tg ? 1.f : 0.f)) * gscale;
}

// Y[row] = mask ? table[tok[row]] : 0 (tok 0 = padding row, zero by construction);  backward: dTable[tok] += mask * dY[row], tok != 0
__global__ void embed_rows_f32(const int32_t* __restrict__ tok, const uint8_t* __restrict__ mask, int period, const float* __restrict__ table,
                               float* __restrict__ Y, int rows, int d, int n_classes) {
  const size_t idx = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= static_cast<size_t>(rows) * d) return;
  const int r = static_cast<int>(idx / d), c = static_cast<int>(idx % d);
  int t = tok[r];
  t = t < 0 ? 0 : (t >= n_classes ? n_classes - 1 : t);
  Y[idx] = (!mask || mask[r % period]) ? table[static_cast<size_t>(t) * d + c] : 0.f;
}
__global__ void embed_bwd_rows(const int32_t* __restrict__ tok, const uint8_t* __restrict__ mask, int period, const float* __restrict__ dY,
                               float* dTable, int rows, int d, int n_classes, int padding_idx) {
  const size_t idx = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (idx >= static_cast<size_t>(rows) * d) return;
  const int r = static_cast<int>(idx / d), c = static_cast<int>(idx % d);
  const int t = tok[r];
  if (t < 0 || t >= n_classes || t == padding_idx || (mask && !mask[r % period])) return;
  atomicAdd(dTable + static_cast<size_t>(t) * d + c, dY[idx]);
}

}  // namespace
}  // namespace d3pm

using namespace d3pm;

extern "C" {

int d3pm_op_matmul_f32(const float* A, long sai, long sak, const float* B, long sbk, long sbj, float* C, int ldc, int M, int N, int K,
                       float beta, const uint8_t* row_mask, int mask_period, void* stream) {
  D3PM_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && ldc >= N, D3PM_E_ARG, "d3pm_op_matmul_f32: bad arguments");
  if (M >= 64 && N >= 64) {      // the matrix-pipe form (same bits); small products keep the 32 x 32 FMA tiles
    dim3 grid((N + 127) / 128, (M + 127) / 128);
    matmul_strided_mfma<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(A, sai, sak, B, sbk, sbj, C, ldc, M, N, K, beta, row_mask,
                                                                            mask_period > 0 ? mask_period : 1);
  } else {
    dim3 grid((N + 31) / 32, (M + 31) / 32);
    matmul_strided<<<grid, 256, 0, static_cast<hipStream_t>(stream)>>>(A, sai, sak, B, sbk, sbj, C, ldc, M, N, K, beta, row_mask,
                                                                       mask_period > 0 ? mask_period : 1);
  }
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

int d3pm_op_colsum_f32(const float* X, int ldx, int M, int N, float* out, float beta, void* stream) {
  D3PM_REQUIRE(X && out && M > 0 && N > 0, D3PM_E_ARG, "d3pm_op_colsum_f32: bad arguments");
  colsum_rows<<<(N + 63) / 64, 256, 0, static_cast<hipStream_t>(stream)>>>(X, ldx, M, N, out, beta);
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

int d3pm_op_act_bwd_f32(const float* U, const float* dM, float* dU, size_t n, int act, void* stream) {
  D3PM_REQUIRE(U && dM && dU && n > 0 && act >= ACT_GELU && act <= ACT_SILU, D3PM_E_ARG, "d3pm_op_act_bwd_f32: bad arguments");
  act_bwd_k<<<static_cast<unsigned>((n + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(U, dM, dU, n, act);
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

int d3pm_op_mask_rows_f32(float* X, int ldx, int M, int N, const uint8_t* mask, int period, void* stream) {
  D3PM_REQUIRE(X && mask && M > 0 && N > 0 && period > 0, D3PM_E_ARG, "d3pm_op_mask_rows_f32: bad arguments");
  const size_t n = static_cast<size_t>(M) * N;
  mask_rows_k<<<static_cast<unsigned>((n + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(X, ldx, M, N, mask, period);
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

int d3pm_op_layernorm_bwd_f32(const float* X, const float* dOut, const float* w, const float* b, const float* film, float eps, int M,
                              int d, float* dX, int accumulate_dx, float* dw, float* db, float* dfilm, void* stream) {
  D3PM_REQUIRE(X && dOut && w && b && dX && dw && db && M > 0 && d > 0 && (!film || dfilm), D3PM_E_ARG,
               "d3pm_op_layernorm_bwd_f32: bad arguments");
  layernorm_bwd_rows<<<(M + 3) / 4, 256, 0, static_cast<hipStream_t>(stream)>>>(X, dOut, w, b, film, eps, M, d, dX, accumulate_dx, dw,
                                                                                db, dfilm);
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

int d3pm_op_attention_bwd_f32(const float* Q, int ldq, const float* K, const float* V, int ldkv, const float* dO, int ldo, float* dQ,
                              int lddq, float* dK, float* dV, int lddkv, float* stats, int B, int Tq, int S, int H, int hd, float scale,
                              float beta_kv, void* stream) {
  D3PM_REQUIRE(Q && K && V && dO && dQ && dK && dV && stats && B > 0 && Tq > 0 && S > 0 && H > 0 && hd > 0, D3PM_E_ARG,
               "d3pm_op_attention_bwd_f32: bad arguments");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const long nq = static_cast<long>(B) * H * Tq, nk = static_cast<long>(B) * H * S;
  attn_bwd_q_rows<<<static_cast<unsigned>((nq + 3) / 4), 256, 0, s>>>(Q, ldq, K, V, ldkv, dO, ldo, dQ, lddq, stats, B, Tq, S, H, hd, scale);
  attn_bwd_kv_rows<<<static_cast<unsigned>((nk + 3) / 4), 256, 0, s>>>(Q, ldq, K, V, ldkv, dO, ldo, dK, dV, lddkv, stats, B, Tq, S, H, hd,
                                                                      scale, beta_kv);
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

int d3pm_op_ce_bwd_f32(const float* logits, int ldl, const int32_t* targets, const uint8_t* frame_mask, int canvas, int rows, int K,
                       float gscale, float* dlogits, int ldd, void* stream) {
  D3PM_REQUIRE(logits && targets && frame_mask && dlogits && rows > 0 && K > 1 && canvas > 0, D3PM_E_ARG, "d3pm_op_ce_bwd_f32: bad arguments");
  ce_bwd_rows<<<(rows + 3) / 4, 256, 0, static_cast<hipStream_t>(stream)>>>(logits, ldl, targets, frame_mask, canvas, rows, K, gscale,
                                                                            dlogits, ldd);
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

int d3pm_op_embed_f32(const int32_t* tok, const uint8_t* mask, int period, const float* table, float* Y, int rows, int d, int n_classes,
                      void* stream) {
  D3PM_REQUIRE(tok && table && Y && rows > 0 && d > 0 && n_classes > 0, D3PM_E_ARG, "d3pm_op_embed_f32: bad arguments");
  const size_t n = static_cast<size_t>(rows) * d;
  embed_rows_f32<<<static_cast<unsigned>((n + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(tok, mask, period > 0 ? period : 1,
                                                                                                      table, Y, rows, d, n_classes);
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

int d3pm_op_embed_bwd_f32(const int32_t* tok, const uint8_t* mask, int period, const float* dY, float* dTable, int rows, int d,
                          int n_classes, int padding_idx, void* stream) {
  D3PM_REQUIRE(tok && dY && dTable && rows > 0 && d > 0 && n_classes > 0, D3PM_E_ARG, "d3pm_op_embed_bwd_f32: bad arguments");
  const size_t n = static_cast<size_t>(rows) * d;
  embed_bwd_rows<<<static_cast<unsigned>((n + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(tok, mask, period > 0 ? period : 1, dY,
                                                                                                      dTable, rows, d, n_classes, padding_idx);
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

}  // extern "C"
