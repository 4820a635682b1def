// probe_issue.hip -- issue-rate calibration on one wave per SIMD (gfx950):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_issue.hip -o tools/_bin/probe_issue
// Each variant runs ITER iterations of a fixed instruction mix in one wave per SIMD (256 threads per CU, one workgroup per CU)
// and reports shader cycles per iteration (s_memtime) and the clock (s_memrealtime).
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define SB() __builtin_amdgcn_sched_barrier(0)

__device__ unsigned long long g_out[256 * 4];

// MODE 0: 8 MFMA 32x32x16, 4 independent accumulators        1: 8 MFMA, ONE accumulator (dependent chain)
//      2: 8 x (MFMA + 3 v_exp)     3: 8 x (MFMA + 6 v_add)     4: 8 x (MFMA + 4 v_exp)   5: 24 v_exp only   6: 48 v_add only
//      7: 8 x (MFMA 16x16x32 x2 + 3 v_exp)   8: 16 MFMA 16x16x32 independent   9: 8 x (MFMA + 2 v_cvt_pk + 2 v_exp)
//      10: 8 x (MFMA + 3 v_exp) with the accumulators read by v_max afterwards (dependency on results each iteration)
template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(float* sink, int iters) {
  const int lane = threadIdx.x & 63;
  floatx16 acc[4];
  floatx4 acc4[8];
  for (int a = 0; a < 4; ++a)
    for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
  for (int a = 0; a < 8; ++a) acc4[a] = floatx4{0.f, 0.f, 0.f, 0.f};
  bf16x8 va, vb;
  for (int i = 0; i < 8; ++i) { va[i] = static_cast<__bf16>(0.001f * (lane + i)); vb[i] = static_cast<__bf16>(0.002f * (lane - i)); }
  float e[24];
  for (int i = 0; i < 24; ++i) e[i] = -0.01f * (lane + i);
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    SB();
    if constexpr (MODE == 0) {
#pragma unroll
      for (int m = 0; m < 8; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, acc[m & 3], 0, 0, 0);
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int m = 0; m < 8; ++m) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, acc[0], 0, 0, 0);
    } else if constexpr (MODE == 2 || MODE == 4 || MODE == 10) {
      constexpr int NE = MODE == 4 ? 4 : 3;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, acc[m & 3], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < NE; ++k) e[(m * NE + k) % 24] = __builtin_amdgcn_exp2f(e[(m * NE + k) % 24]);
        SB();
      }
      if constexpr (MODE == 10) {
        float mx = acc[0][0];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int i = 0; i < 16; ++i) mx = fmaxf(mx, acc[a][i]);
        e[0] += mx * 1e-30f;
      }
    } else if constexpr (MODE == 3) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, acc[m & 3], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 6; ++k) e[(m * 6 + k) % 24] += 1.5f;
        SB();
      }
    } else if constexpr (MODE == 5) {
#pragma unroll
      for (int k = 0; k < 24; ++k) e[k] = __builtin_amdgcn_exp2f(e[k]);
    } else if constexpr (MODE == 6) {
#pragma unroll
      for (int k = 0; k < 48; ++k) e[k % 24] += 1.5f;
    } else if constexpr (MODE == 7) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        acc4[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vb, acc4[m], 0, 0, 0);
        acc4[(m + 4) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb, va, acc4[(m + 4) & 7], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < 3; ++k) e[(m * 3 + k) % 24] = __builtin_amdgcn_exp2f(e[(m * 3 + k) % 24]);
        SB();
      }
    } else if constexpr (MODE == 8) {
#pragma unroll
      for (int m = 0; m < 16; ++m) acc4[m & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vb, acc4[m & 7], 0, 0, 0);
    } else if constexpr (MODE == 9) {
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, acc[m & 3], 0, 0, 0);
        e[(2 * m) % 24] = __builtin_amdgcn_exp2f(e[(2 * m) % 24]);
        e[(2 * m + 1) % 24] = __builtin_amdgcn_exp2f(e[(2 * m + 1) % 24]);
        typedef float float2v __attribute__((ext_vector_type(2)));
        typedef __bf16 pair __attribute__((ext_vector_type(2)));
        const pair p0 = __builtin_convertvector((float2v){e[(2 * m) % 24], e[(2 * m + 1) % 24]}, pair);
        const pair p1 = __builtin_convertvector((float2v){e[(2 * m + 2) % 24], e[(2 * m + 3) % 24]}, pair);
        e[(2 * m + 8) % 24] += __builtin_bit_cast(float, p0) + __builtin_bit_cast(float, p1);
        SB();
      }
    }
    SB();
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int a = 0; a < 4; ++a)
    for (int i = 0; i < 16; ++i) s += acc[a][i];
  for (int a = 0; a < 8; ++a) s += acc4[a][0] + acc4[a][3];
  for (int i = 0; i < 24; ++i) s += e[i];
  if (s == 123.456f) sink[threadIdx.x] = s;
  if (threadIdx.x == 0) {
    g_out[blockIdx.x * 4 + 0] = c0; g_out[blockIdx.x * 4 + 1] = c1; g_out[blockIdx.x * 4 + 2] = r0; g_out[blockIdx.x * 4 + 3] = r1;
  }
}

template <int MODE> static void run(const char* name, float* sink) {
  const int iters = 2000;
  probe<MODE><<<256, 256>>>(sink, iters);
  probe<MODE><<<256, 256>>>(sink, iters);
  CK(hipDeviceSynchronize());
  unsigned long long h[256 * 4];
  CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_out), sizeof(h)));
  double cyc = 0, us = 0;
  for (int b = 0; b < 256; ++b) { cyc += static_cast<double>(h[4 * b + 1] - h[4 * b]); us += (h[4 * b + 3] - h[4 * b + 2]) / 100.0; }
  printf("%-64s %8.1f cycles / iteration   %.2f GHz\n", name, cyc / 256 / iters, cyc / us / 1e3);
  fflush(stdout);
}

int main() {
  float* sink;
  CK(hipMalloc(&sink, 4096));
  run<0>("8 MFMA 32x32x16, four accumulators", sink);
  run<1>("8 MFMA 32x32x16, one accumulator (chain)", sink);
  run<8>("16 MFMA 16x16x32, eight accumulators", sink);
  run<5>("24 v_exp_f32", sink);
  run<6>("48 v_add_f32", sink);
  run<2>("8 x (MFMA 32x32x16 + 3 v_exp)", sink);
  run<4>("8 x (MFMA 32x32x16 + 4 v_exp)", sink);
  run<3>("8 x (MFMA 32x32x16 + 6 v_add)", sink);
  run<9>("8 x (MFMA 32x32x16 + 2 v_exp + 2 v_cvt_pk + 2 v_add)", sink);
  run<7>("8 x (2 MFMA 16x16x32 + 3 v_exp)", sink);
  run<10>("8 x (MFMA 32x32x16 + 3 v_exp), then v_max over the results", sink);
  return 0;
}
