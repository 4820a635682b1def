// d3pm_schedule.cpp -- host arithmetic of the absorbing-state D3PM schedule.
//
// Replaces AR.cosine_beta_schedule (ar_discrete.py:286-304, called with timesteps+1 at :257),
// _get_absorbing_transition_mat (:315-334) and the fp16 tensordot chain that builds the cumulative
// tables (:268-277).  Every table the reference materialises ([T,1025,1025] fp16, 3 x 210 MB) has
// the form  d*I + c*1 e_M^T  with row M = e_M, so the whole schedule is four fp16 scalars per step:
//     d_t    = rn16(1 - beta16_t)               c_t    = beta16_t
//     dbar_t = rn16(dbar_{t-1} * d_t)           cbar_t = rn16(rn32(dbar_{t-1} * c_t + cbar_{t-1}))
// (fp32 accumulate, round to fp16 -- what an fp16 matmul with fp32 accumulation yields for the two
// non-zero terms of each dot product).  tests/test_schedule.py pins these against the scalars read
// out of the reference's own dense tables (tests/golden/tables_t100.npz).
#include <cmath>
#include <cstring>

#include "d3pm_common.h"

#include "d3pm_kernels.h"

namespace d3pm {
namespace {
// ---- host-side fp16 helpers (schedule constants) ------------------------------------------------
inline float h2f(uint16_t h) {
  uint32_t sign = (h & 0x8000u) << 16, exp = (h >> 10) & 0x1f, man = h & 0x3ffu, bits;
  if (exp == 0) {
    if (man == 0) bits = sign;
    else {
      int e = -1;
      do { ++e; man <<= 1; } while (!(man & 0x400u));
      bits = sign | ((127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
    }
  } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
  else bits = sign | ((exp - 15 + 127) << 23) | (man << 13);
  float f;
  __builtin_memcpy(&f, &bits, 4);
  return f;
}
inline uint16_t f2h(float f) {   // round-to-nearest-even, subnormals kept
  uint32_t x;
  __builtin_memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  int32_t exp = static_cast<int32_t>((x >> 23) & 0xff) - 127 + 15;
  uint32_t man = x & 0x7fffffu;
  if (((x >> 23) & 0xff) == 0xff) return sign | 0x7c00u | (man ? 0x200u : 0);
  if (exp >= 31) return sign | 0x7c00u;
  if (exp <= 0) {
    if (exp < -10) return sign;
    man |= 0x800000u;
    uint32_t shift = 14 - exp;
    uint32_t half = man >> shift, rem = man & ((1u << shift) - 1), mid = 1u << (shift - 1);
    if (rem > mid || (rem == mid && (half & 1))) ++half;
    return sign | half;
  }
  uint32_t half = (exp << 10) | (man >> 13), rem = man & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (half & 1))) ++half;
  return sign | half;
}
inline float rn16h(float v) { return h2f(f2h(v)); }
inline float log16_of(float fact) { return rn16h(logf(rn16h(fact + 1.0e-6f))); }



// double -> fp16, round-to-nearest-even in one step (no intermediate float rounding)
uint16_t d2h(double v) {
  uint64_t x;
  __builtin_memcpy(&x, &v, 8);
  uint16_t sign = static_cast<uint16_t>((x >> 48) & 0x8000u);
  int32_t e = static_cast<int32_t>((x >> 52) & 0x7ff);
  uint64_t man = x & 0xfffffffffffffull;
  if (e == 0x7ff) return sign | 0x7c00u | (man ? 0x200u : 0);
  if (e == 0 && man == 0) return sign;
  int32_t exp = e - 1023 + 15;
  if (exp >= 31) return sign | 0x7c00u;
  man |= 1ull << 52;                       // 53-bit significand
  int shift = 42;                          // keep 11 bits (1 implicit + 10)
  if (exp <= 0) {
    shift += 1 - exp;
    exp = 0;
    if (shift > 54) return sign;
  }
  uint64_t q = man >> shift, rem = man & ((1ull << shift) - 1), mid = 1ull << (shift - 1);
  if (rem > mid || (rem == mid && (q & 1))) ++q;
  // q carries the implicit bit when exp > 0: adding (exp-1)<<10 folds mantissa overflow into the exponent
  uint32_t h = exp > 0 ? static_cast<uint32_t>(((exp - 1) << 10) + q) : static_cast<uint32_t>(q);
  if (h >= 0x7c00u) h = 0x7c00u;
  return sign | static_cast<uint16_t>(h);
}
}  // namespace

float host_h2f(uint16_t h) { return h2f(h); }
uint16_t host_f2h(float f) { return f2h(f); }
float host_log16(float fact) { return log16_of(fact); }

// log16(rn16(fact1 + eps)) takes four values per step and fact2 needs dbar/cbar of step t-1
PosteriorConsts make_posterior_consts(const d3pm_schedule* s, int t) {
  PosteriorConsts pc{};
  pc.t = t;
  pc.log_f1_zero = log16_of(0.f);
  pc.log_f1_one = log16_of(1.f);
  if (t > 0) {
    pc.log_f1_d = log16_of(h2f(s->d[t]));
    pc.log_f1_c = log16_of(h2f(s->c[t]));
    pc.dbar_prev = h2f(s->dbar[t - 1]);
    pc.cbar_prev = h2f(s->cbar[t - 1]);
  }
  return pc;
}
}  // namespace d3pm

using namespace d3pm;

extern "C" int d3pm_schedule_build(int timesteps, uint16_t* betas, uint16_t* d, uint16_t* c, uint16_t* dbar,
                                   uint16_t* cbar) {
  D3PM_REQUIRE(timesteps >= 2 && betas && d && c && dbar && cbar, D3PM_E_ARG, "d3pm_schedule_build: bad arguments");
  // cosine_beta_schedule(n = timesteps+1): steps = n+1 grid points linspace(0, steps, steps)
  const int n = timesteps + 1, steps = n + 1;
  const double s = 0.008, step = static_cast<double>(steps) / static_cast<double>(steps - 1);
  double prev = 0.0, first = 0.0;
  for (int i = 0; i < steps; ++i) {
    double x = (i == steps - 1) ? static_cast<double>(steps) : static_cast<double>(i) * step;
    double a = std::cos(((x / steps) + s) / (1 + s) * M_PI * 0.5);
    a = a * a;
    if (i == 0) first = a;
    a = a / first;
    if (i > 0) {
      double beta = 1.0 - a / prev;
      beta = beta < 0.0 ? 0.0 : (beta > 0.999 ? 0.999 : beta);
      betas[i - 1] = host_f2h(static_cast<float>(beta));   // torch: double -> Half goes through float
    }
    prev = a;
  }
  for (int t = 0; t < timesteps; ++t) {
    const double b = static_cast<double>(host_h2f(betas[t]));
    d[t] = d2h(1.0 - b);
    c[t] = betas[t];
    // row M of Q_t is (1-beta)+beta rounded to fp16; the closed form needs it to be exactly 1
    const uint16_t one = d2h(static_cast<double>(host_h2f(d[t])) + b);
    D3PM_REQUIRE(one == 0x3c00u, D3PM_E_SHAPE,
                 "schedule step %d: absorbing row is not exactly 1 in fp16; closed form does not apply", t);
    if (t == 0) {
      dbar[0] = d[0];
      cbar[0] = c[0];
    } else {
      const float db = host_h2f(dbar[t - 1]);
      dbar[t] = host_f2h(db * host_h2f(d[t]));
      const float prod = db * host_h2f(c[t]);               // exact in fp32 (11 x 11 bit significands)
      cbar[t] = host_f2h(prod + host_h2f(cbar[t - 1]));
    }
  }
  return D3PM_OK;
}
