// d3pm_sample_row.h -- the per-row D3PM posterior + Gumbel-max draw, shared by the stand-alone sampler
// (d3pm_sample.hip: logits from HBM) and the fused final-projection + sampler (d3pm_final_sample.hip: logits from LDS).
// Arithmetic and rounding points: see the header of d3pm_sample.hip (/root/reference/vall_e/vall_e/ar_discrete.py:337-420).
#pragma once
#include <cmath>

#include "d3pm_kernels.h"

namespace d3pm {
namespace {

constexpr int kMaxGroupsPerLane = 5;   // supports n_classes <= 64*5*4 = 1280
constexpr float kEps = 1.0e-6f;        // self.eps (ar_discrete.py:276), added in fp32 opmath then rounded

// One wave draws x_{t-1} of one row.  `lr[j]` are the row's K logits in the model dtype (any address space);
// returns the sampled id in every lane.  `post_row` (optional) receives the fp16 posterior logits of the row.
template <typename T, typename P>
__device__ __forceinline__ int sample_row(P lr, int K, int mask_id, int x, uint64_t seed, uint32_t grow, int greedy,
                                          const PosteriorConsts& pc, uint16_t* post_row, int lane, uint32_t stream = 0u) {
  const int groups = (K + 3) >> 2;
  float z[kMaxGroupsPerLane][4];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < kMaxGroupsPerLane; ++i) {
    int g = lane + i * kWave;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      int j = g * 4 + w;
      float v = (g < groups && j < K) ? rn16(static_cast<float>(lr[j])) : -INFINITY;
      z[i][w] = v;
      mx = fmaxf(mx, v);
    }
  }
  int best_j = 0;
  float best_v = -INFINITY;
  if (pc.t == 0) {
    // t == 0: model logits are used as they are and no noise is added (ar_discrete.py:407,413)
#pragma unroll
    for (int i = 0; i < kMaxGroupsPerLane; ++i)
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        int j = (lane + i * kWave) * 4 + w;
        if (j < K && z[i][w] > best_v) { best_v = z[i][w]; best_j = j; }
      }
  } else {
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxGroupsPerLane; ++i)
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        float e = expf(z[i][w] - mx);   // exp(-inf) = 0 for the padding classes
        z[i][w] = e;
        sum += e;
      }
    sum = wave_sum(sum);
    float s_other = 0.f, p_mask = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxGroupsPerLane; ++i)
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        int j = (lane + i * kWave) * 4 + w;
        float p = rn16(z[i][w] / sum);
        z[i][w] = p;
        if (j == mask_id) p_mask = p; else s_other += p;
      }
    s_other = wave_sum(s_other);
    p_mask = wave_sum(p_mask);
    const float f2_mask = rn16(fmaf(s_other, pc.cbar_prev, p_mask));
    const bool x_is_mask = (x == mask_id);
#pragma unroll
    for (int i = 0; i < kMaxGroupsPerLane; ++i) {
      int g = lane + i * kWave;
      if (g >= groups) continue;
      float u[4];
      if (!greedy) noise4(seed, static_cast<uint32_t>(g), grow, static_cast<uint32_t>(pc.t), stream, u);
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        int j = g * 4 + w;
        if (j >= K) continue;
        float lf1 = x_is_mask ? (j == mask_id ? pc.log_f1_one : pc.log_f1_c)
                              : (j == x ? pc.log_f1_d : pc.log_f1_zero);
        float f2 = (j == mask_id) ? f2_mask : rn16(z[i][w] * pc.dbar_prev);
        float lf2 = rn16(logf(rn16(f2 + kEps)));
        float out = rn16(lf1 + lf2);
        if (post_row) post_row[j] = __builtin_bit_cast(uint16_t, static_cast<f16>(out));
        float v = greedy ? out : out + gumbel(u[w]);
        if (v > best_v) { best_v = v; best_j = j; }   // ascending j per lane keeps the first maximum
      }
    }
  }
  wave_argmax(best_v, best_j);
  return best_j;
}

// The same routine for n_classes = 1025 = 4 x 256 + 1 (1024 codec ids + the mask id: every model of the reference, ar_discrete.py:255)
// without a predicate in it.  The general routine above walks five passes of 64 groups x 4 classes and tests every group and every
// class against K -- 121 exec-mask branches in the compiled kernel, and a fifth pass that exists for ONE class (1024, lane 0).
// Here passes 0..3 cover classes 0..1023 unconditionally and class 1024 is a tail element of lane 0 (the other lanes carry a -inf
// logit, whose exp is 0: what they added in the general routine as well).  Every operation on a class, the order in which a lane
// accumulates its partial sums (ascending class, the tail last) and the first-index argmax are those of sample_row: same bits.
#ifndef D3PM_SAMPLER_EARLY_OUT
#define D3PM_SAMPLER_EARLY_OUT 1      // A/B builds (tools/build_variant.py NAME -DD3PM_SAMPLER_EARLY_OUT=0): the full routine for every row
#endif
template <typename T, typename P>
__device__ __forceinline__ int sample_row_1025(P lr, int mask_id, int x, uint64_t seed, uint32_t grow, int greedy,
                                               const PosteriorConsts& pc, int lane, uint32_t stream = 0u, bool early_out = D3PM_SAMPLER_EARLY_OUT != 0) {
  constexpr int K = 1025;
  float z[4][4], zt;
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      z[i][w] = rn16(static_cast<float>(lr[(lane + i * kWave) * 4 + w]));
      mx = fmaxf(mx, z[i][w]);
    }
  zt = lane == 0 ? rn16(static_cast<float>(lr[K - 1])) : -INFINITY;
  mx = fmaxf(mx, zt);
  int best_j = 0;
  float best_v = -INFINITY;
  if (pc.t == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int w = 0; w < 4; ++w)
        if (z[i][w] > best_v) { best_v = z[i][w]; best_j = (lane + i * kWave) * 4 + w; }
    if (zt > best_v) { best_v = zt; best_j = K - 1; }
  } else {
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        z[i][w] = expf(z[i][w] - mx);
        sum += z[i][w];
      }
    zt = expf(zt - mx);
    sum += zt;
    sum = wave_sum(sum);
    const bool x_is_mask = (x == mask_id);
    // score of class j from its probability: the reference's out_j (+ Gumbel noise); f2m = fact2 of the mask class
    auto one = [&](int j, float p, float u, float f2m) __attribute__((always_inline)) {
      const float lf1 = x_is_mask ? (j == mask_id ? pc.log_f1_one : pc.log_f1_c) : (j == x ? pc.log_f1_d : pc.log_f1_zero);
      const float f2 = (j == mask_id) ? f2m : rn16(p * pc.dbar_prev);
      const float lf2 = rn16(logf(rn16(f2 + kEps)));
      const float out = rn16(lf1 + lf2);
      return greedy ? out : out + gumbel(u);
    };
    // the uniforms of the row's 1025 classes: groups lane + 64 i (words 0..3) and word 0 of group 256
    float u[4][4], ut[4] = {0.5f, 0.5f, 0.5f, 0.5f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int w = 0; w < 4; ++w) u[i][w] = 0.5f;
      if (!greedy) noise4(seed, static_cast<uint32_t>(lane + i * kWave), grow, static_cast<uint32_t>(pc.t), stream, u[i]);
    }
    if (!greedy) noise4(seed, 256u, grow, static_cast<uint32_t>(pc.t), stream, ut);      // group 256 = classes 1024..1027: word 0
    if (!x_is_mask && !greedy && early_out) {
      // A REVEALED row (x_t != mask) keeps its token unless another class wins the Gumbel race, and every other class starts
      // log(eps) = -13.8 behind (fact1 = 0 off the diagonal: ar_discrete.py:337-420).  Instead of the posterior of all
      // 1025 classes (a division, three logs and six roundings each) the kept token's exact score is compared with an UPPER bound
      // of every other class's, built from the same monotone operations on upper bounds of their inputs:
      //   p_j <= rn16(1 / sum)  (the largest exponential is exp(0) = 1);   gumbel(u_j) <= gumbel(max_j u_j);   and for the mask class
      //   fact2_M <= rn16(1.001 cbar + rn16(1 / sum))  (sum_{k != M} p_k <= 1.001: 1025 roundings to fp16) with its own uniform.
      // If the kept token clears both bounds by 2^-5 (four fp16 quanta at this magnitude; whatever logf does in its last bit is four
      // orders below that) it is the argmax the full routine returns, whatever the other scores are; otherwise the full routine
      // runs on the values already in registers.  Same ids as without the test, by construction; what is skipped is ~55 % of the
      // row's vector work (the uniforms themselves, ~40 %, are still drawn: the stream is part of the contract).
      float um = fmaxf(fmaxf(fmaxf(u[0][0], u[0][1]), fmaxf(u[0][2], u[0][3])), fmaxf(fmaxf(u[1][0], u[1][1]), fmaxf(u[1][2], u[1][3])));
      um = fmaxf(um, fmaxf(fmaxf(fmaxf(u[2][0], u[2][1]), fmaxf(u[2][2], u[2][3])), fmaxf(fmaxf(u[3][0], u[3][1]), fmaxf(u[3][2], u[3][3]))));
      um = wave_max(lane == 0 ? fmaxf(um, ut[0]) : um);
      // (exponential, uniform) of a class id that is the same in every lane: pass / word picked by wave-uniform selects, then the owner
      // lane's copy (class 1024: group 256 = lane 0's tail)
      auto at = [&](int j, float& ej, float& uj) __attribute__((always_inline)) {
        const int g = j >> 2, pass = g >> 6, word = j & 3, owner = g & 63;
        float ev = zt, uv = ut[0];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            const bool hit = pass == i && word == w;   // wave-uniform
            ev = hit ? z[i][w] : ev;
            uv = hit ? u[i][w] : uv;
          }
        ej = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ev), owner));
        uj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, uv), owner));
      };
      const int xu = __builtin_amdgcn_readfirstlane(x), mu = __builtin_amdgcn_readfirstlane(mask_id);
      float e_x, u_x, e_m, u_m;
      at(xu, e_x, u_x);
      at(mu, e_m, u_m);
      (void)e_m;
      const float v_x = one(xu, rn16(e_x / sum), u_x, 0.f);      // the kept token's score exactly as the full routine computes it (x != M)
      const float p_ub = rn16(1.0f / sum);
      const float lf2_ub = rn16(logf(rn16(rn16(p_ub * pc.dbar_prev) + kEps)));
      const float others_ub = rn16(pc.log_f1_zero + lf2_ub) + gumbel(um);
      const float lf2m_ub = rn16(logf(rn16(rn16(fmaf(1.001f, pc.cbar_prev, p_ub)) + kEps)));
      const float mask_ub = rn16(pc.log_f1_zero + lf2m_ub) + gumbel(u_m);
      if (fmaxf(others_ub, mask_ub) + 0.03125f < v_x) return xu;      // wave-uniform: every operand is
    }
    float s_other = 0.f, p_mask = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float p = rn16(z[i][w] / sum);
        z[i][w] = p;
        const bool is_mask = (lane + i * kWave) * 4 + w == mask_id;
        p_mask = is_mask ? p : p_mask;
        s_other += is_mask ? 0.f : p;           // (+ 0 leaves the partial sum as it is: the general routine skips the add)
      }
    zt = rn16(zt / sum);
    s_other += zt;                              // class 1024 is never the mask id (512)
    s_other = wave_sum(s_other);
    p_mask = wave_sum(p_mask);
    const float f2_mask = rn16(fmaf(s_other, pc.cbar_prev, p_mask));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int g = lane + i * kWave;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const float v = one(g * 4 + w, z[i][w], u[i][w], f2_mask);
        if (v > best_v) { best_v = v; best_j = g * 4 + w; }   // ascending j per lane keeps the first maximum
      }
    }
    {
      const float v = one(K - 1, zt, ut[0], f2_mask);
      if (lane == 0 && v > best_v) { best_v = v; best_j = K - 1; }
    }
  }
  wave_argmax(best_v, best_j);
  return best_j;
}

}  // namespace
}  // namespace d3pm
