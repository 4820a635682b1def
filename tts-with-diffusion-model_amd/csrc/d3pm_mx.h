// d3pm_mx.h -- device helpers of the block-scaled fp8 ("MX") format shared by the quantisers, the MX GEMM epilogue
// (d3pm_mx.hip) and the row-panel GEMM's LayerNorm epilogue (d3pm_mfma_gemm_big.hip).  Format and scale rule: d3pm_mx.hip.
#pragma once
#include "d3pm_common.h"

namespace d3pm {
namespace {

// e8m0 byte of a block's absolute maximum: amax = 1.m x 2^E -> 2^(E - 8) when 1.m <= 1.75, else 2^(E - 7): the smallest power of
// two with amax / scale <= 448.  Exact integer arithmetic on the fp32 bits (the host side: _hip.mx_scale_bytes).
__device__ __forceinline__ uint32_t mx_scale_byte(float amax) {
  const uint32_t b = __float_as_uint(amax);
  int e = static_cast<int>(b >> 23) - 8 + ((b & 0x7FFFFFu) > 0x600000u ? 1 : 0);
  e = e < 1 ? 1 : (e > 254 ? 254 : e);               // amax = 0 (or denormal) -> the smallest scale: every code is 0
  return static_cast<uint32_t>(e);
}
__device__ __forceinline__ float mx_inv_scale(uint32_t byte) { return __uint_as_float((254u - byte) << 23); }   // 2^(127 - byte), exact

__device__ __forceinline__ uint2 mx_pack8(const float (&o)[8], float inv) {
  uint32_t lo = 0, hi = 0;
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(o[0] * inv, o[1] * inv, lo, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(o[2] * inv, o[3] * inv, lo, true);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(o[4] * inv, o[5] * inv, hi, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(o[6] * inv, o[7] * inv, hi, true);
  return uint2{lo, hi};
}

// The MFMA epilogue layout after the regrouping of epilogue_store (d3pm_mfma_tile.h): for a row block mt and the two 32-column
// halves np = 0, 1 of the wave's 64 columns, lane (r = lane & 15, g = lane >> 4) holds 8 consecutive values at columns
// 32 np + nq(g), nq = 0, 16, 8, 24 for g = 0..3 -- the four lanes of a row own one MX block of 32 columns.
// mx_block_quantise: absmax over those four lanes -> scale byte (the same in all four), the lane's eight codes.
__device__ __forceinline__ uint2 mx_block_quantise(const float (&v)[8], uint32_t& scale_byte) {
  float amax = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(v[i]));
  amax = fmaxf(amax, __shfl_xor(amax, 16, kWave));
  amax = fmaxf(amax, __shfl_xor(amax, 32, kWave));
  scale_byte = mx_scale_byte(amax);
  return mx_pack8(v, mx_inv_scale(scale_byte));
}
// mx_store_row64: the codes of both halves (c0: np = 0, c1: np = 1) of one row block -> 64 contiguous bytes per row as ONE 16-byte
// store per lane: lanes l and l + 32 (g and g + 2: adjacent 8-column groups) trade halves with v_permlane32_swap so that the
// lower lane ends up with 16 consecutive columns of half 0 and the upper lane with 16 of half 1.  `row64` = address of the row's
// first of the 64 columns.
__device__ __forceinline__ void mx_store_row64(uint2 c0, uint2 c1, uint8_t* row64, int lane) {
  // swaps first-operand[lanes 32..63] with second-operand[lanes 0..31]; s_nop 1 = the VALU-write -> permlane-read hazard
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c0.x), "+v"(c1.x));
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(c0.y), "+v"(c1.y));
  // lower lanes: c0 = own half 0, c1 = partner's half 0 (the next 8 columns); upper lanes: c0 = partner's half 1, c1 = own half 1
  const int g = lane >> 4;
  const int col = g < 2 ? g * 16 : 32 + (g - 2) * 16;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  *reinterpret_cast<u32x4*>(row64 + col) = u32x4{c0.x, c0.y, c1.x, c1.y};
}

}  // namespace
}  // namespace d3pm
