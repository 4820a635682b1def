// probe_issue2.hip -- per-instruction issue cost on gfx950, alone and in the gaps of v_mfma_f32_32x32x16_bf16, at one and two
// waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_issue2.hip -o tools/_bin/probe_issue2
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define SB() __builtin_amdgcn_sched_barrier(0)

__device__ unsigned long long g_out[256 * 4];

enum { OP_ADD_INL, OP_ADD_LIT, OP_MAX3, OP_PK_ADD, OP_CVT_PK, OP_EXP, OP_MOV64, OP_FMA, OP_MAX, OP_MUL, OP_NONE };

template <int OP> __device__ __forceinline__ void op(float& a, float& b, float& c, f2& p, f2& q) {
  if constexpr (OP == OP_ADD_INL) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(a));
  if constexpr (OP == OP_ADD_LIT) asm volatile("v_add_f32 %0, 0x3fc00000, %0" : "+v"(a));
  if constexpr (OP == OP_MAX3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
  if constexpr (OP == OP_PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(q));
  if constexpr (OP == OP_CVT_PK) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(a) : "v"(b), "v"(c));
  if constexpr (OP == OP_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(a));
  if constexpr (OP == OP_MOV64) asm volatile("v_mov_b64 %0, %1" : "=v"(p) : "v"(q));
  if constexpr (OP == OP_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
  if constexpr (OP == OP_MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a) : "v"(b));
  if constexpr (OP == OP_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a) : "v"(b));
}

// PER: ops per MFMA gap (MFMAS = 8 per iteration; MFMAS = 0: 48 ops alone)
template <int OP, int MFMAS, int PER>
__global__ __launch_bounds__(512, 1) void probe(float* sink, int iters) {
  const int lane = threadIdx.x & 63;
  floatx16 acc[4];
  for (int a = 0; a < 4; ++a)
    for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
  bf16x8 va, vb;
  for (int i = 0; i < 8; ++i) { va[i] = static_cast<__bf16>(0.001f * (lane + i)); vb[i] = static_cast<__bf16>(0.002f * (lane - i)); }
  float e[8], b = 0.5f + lane, c = 0.25f * lane;
  f2 p[8], q = {1.0f, 2.0f};
  for (int i = 0; i < 8; ++i) { e[i] = -0.01f * (lane + i); p[i] = f2{e[i], e[i]}; }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    SB();
    if constexpr (MFMAS == 0) {
#pragma unroll
      for (int k = 0; k < 48; ++k) op<OP>(e[k & 7], b, c, p[k & 7], q);
    } else {
#pragma unroll
      for (int m = 0; m < MFMAS; ++m) {
        acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, acc[m & 3], 0, 0, 0);
#pragma unroll
        for (int k = 0; k < PER; ++k) op<OP>(e[(m * PER + k) & 7], b, c, p[(m * PER + k) & 7], q);
        SB();
      }
    }
    SB();
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int a = 0; a < 4; ++a)
    for (int i = 0; i < 16; ++i) s += acc[a][i];
  for (int i = 0; i < 8; ++i) s += e[i] + p[i][0] + p[i][1];
  if (s == 123.456f) sink[threadIdx.x] = s;
  if (threadIdx.x == 0) {
    g_out[blockIdx.x * 4 + 0] = c0; g_out[blockIdx.x * 4 + 1] = c1; g_out[blockIdx.x * 4 + 2] = r0; g_out[blockIdx.x * 4 + 3] = r1;
  }
}

template <int OP, int MFMAS, int PER> static double run1(float* sink, int threads) {
  const int iters = 1000;
  probe<OP, MFMAS, PER><<<256, threads>>>(sink, iters);
  probe<OP, MFMAS, PER><<<256, threads>>>(sink, iters);
  CK(hipDeviceSynchronize());
  unsigned long long h[256 * 4];
  CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_out), sizeof(h)));
  double cyc = 0;
  for (int b = 0; b < 256; ++b) cyc += static_cast<double>(h[4 * b + 1] - h[4 * b]);
  return cyc / 256 / iters;
}

template <int OP> static void run(const char* name, float* sink) {
  const double a1 = run1<OP, 0, 0>(sink, 256), a2 = run1<OP, 0, 0>(sink, 512);
  const double g3 = run1<OP, 8, 3>(sink, 256), g6 = run1<OP, 8, 6>(sink, 256), g6w2 = run1<OP, 8, 6>(sink, 512);
  printf("%-22s alone: %5.2f cyc/op (1 wave/SIMD) %5.2f (2 waves, per wave) | 8 x (MFMA + 3 ops): %6.1f  8 x (MFMA + 6 ops): %6.1f cyc/iter; two waves: %6.1f\n",
         name, a1 / 48, a2 / 48, g3, g6, g6w2);
  fflush(stdout);
}

int main() {
  float* sink;
  CK(hipMalloc(&sink, 4096));
  printf("8 MFMA alone: %.1f cycles / iteration (1 wave per SIMD), %.1f (2 waves per SIMD, per wave)\n", run1<OP_NONE, 8, 0>(sink, 256), run1<OP_NONE, 8, 0>(sink, 512));
  run<OP_ADD_INL>("v_add_f32 (inline)", sink);
  run<OP_ADD_LIT>("v_add_f32 (literal)", sink);
  run<OP_MUL>("v_mul_f32", sink);
  run<OP_MAX>("v_max_f32", sink);
  run<OP_FMA>("v_fma_f32", sink);
  run<OP_MAX3>("v_max3_f32", sink);
  run<OP_PK_ADD>("v_pk_add_f32", sink);
  run<OP_CVT_PK>("v_cvt_pk_bf16_f32", sink);
  run<OP_EXP>("v_exp_f32", sink);
  run<OP_MOV64>("v_mov_b64", sink);
  return 0;
}
