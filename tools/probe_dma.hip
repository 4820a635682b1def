// probe_dma.hip -- issue cost of the direct-to-LDS loads and of LDS fragment reads among MFMAs (gfx950), one and two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_dma.hip -o tools/_bin/probe_dma
// Every iteration: 48 x v_mfma_f32_16x16x32_bf16 (the k-step of a 96 x 64 wave tile) with NP direct-to-LDS pieces (1 KiB per
// wave-instruction, from an L2-resident 4 MiB buffer) and NR ds_read_b128 spread between them.  FORM: 0 global_load_lds_dwordx4
// (SGPR base + VGPR offset), 1 buffer_load_dwordx4 ... lds (resource + VGPR offset), 2 global_load_dwordx4 into registers +
// ds_write_b128 (register staging).
#include <cstdio>
#include <cstdlib>
#include <hip/hip_runtime.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define SB() __builtin_amdgcn_sched_barrier(0)

__device__ unsigned long long g_out[256 * 2];

template <int FORM, int NP, int NR>
__global__ __launch_bounds__(512, 1) void probe(const char* __restrict__ src, float* sink, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  floatx4 acc[24];
  for (int a = 0; a < 24; ++a) acc[a] = floatx4{0.f, 0.f, 0.f, 0.f};
  bf16x8 va, vb;
  for (int i = 0; i < 8; ++i) { va[i] = static_cast<__bf16>(0.001f * (lane + i)); vb[i] = static_cast<__bf16>(0.002f * (lane - i)); }
  const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)smem)) + wave * 16384;
  const uint32_t voff = lane * 16;
  const char* base = src + (blockIdx.x & 15) * 262144 + wave * 16384;
  i32x4 rsrc;
  {
    const uintptr_t p = reinterpret_cast<uintptr_t>(base);
    rsrc[0] = static_cast<int>(p & 0xffffffffu);
    rsrc[1] = static_cast<int>((p >> 32) & 0xffffu);
    rsrc[2] = 0x7fffffff;
    rsrc[3] = 0x00020000;
  }
  u32x4 frag[4] = {};
  u32x4 stg[NP > 0 ? NP : 1];
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const int slot = (it & 1) * 8192;
    SB();
#pragma unroll
    for (int g = 0; g < 12; ++g) {
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[(g * 4 + m) % 24] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, vb, acc[(g * 4 + m) % 24], 0, 0, 0);
      if (g < NP) {
        if constexpr (FORM == 0) {
          asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:0" ::"v"(voff + g * 1024), "s"(base), "s"(lds_base + slot + (g & 7) * 1024) : "memory");
        } else if constexpr (FORM == 1) {
          asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff + g * 1024), "s"(rsrc), "s"(lds_base + slot + (g & 7) * 1024) : "memory");
        } else {
          asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(stg[g]) : "v"(voff + g * 1024), "s"(base) : "memory");
        }
      }
      if constexpr (FORM == 2) {
        if (g >= 12 - NP) {      // the pieces loaded in the PREVIOUS half of the step go to LDS (a full step of latency would need more registers)
          asm volatile("s_waitcnt vmcnt(%2)\n\tds_write_b128 %0, %1" ::"v"(lds_base + slot + ((g - (12 - NP)) & 7) * 1024 + voff), "v"(stg[g - (12 - NP)]), "n"(0) : "memory");
        }
      }
      if (g * NR / 12 != (g + 1) * NR / 12) {
#pragma unroll
        for (int r = g * NR / 12; r < (g + 1) * NR / 12; ++r)
          asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(frag[r & 3]) : "v"(lds_base + voff), "n"((r & 7) * 1024));
      }
      SB();
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    SB();
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int a = 0; a < 24; ++a) s += acc[a][0] + acc[a][3];
  for (int r = 0; r < 4; ++r) s += __builtin_bit_cast(float, frag[r][0]);
  if (s == 123.456f) sink[threadIdx.x] = s;
  if (threadIdx.x == 0) { g_out[blockIdx.x * 2] = c0; g_out[blockIdx.x * 2 + 1] = c1; }
}

template <int FORM, int NP, int NR> static double run1(const char* src, float* sink, int threads) {
  const int iters = 400;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<FORM, NP, NR>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
  probe<FORM, NP, NR><<<256, threads, 128 * 1024>>>(src, sink, iters);
  probe<FORM, NP, NR><<<256, threads, 128 * 1024>>>(src, sink, iters);
  CK(hipDeviceSynchronize());
  unsigned long long h[512];
  CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_out), sizeof(h)));
  double cyc = 0;
  for (int b = 0; b < 256; ++b) cyc += static_cast<double>(h[2 * b + 1] - h[2 * b]);
  return cyc / 256 / iters;
}

template <int FORM, int NP, int NR> static void run(const char* name, const char* src, float* sink) {
  printf("%-60s %7.1f cycles per 48-MFMA step (1 wave / SIMD)   %7.1f (2 waves / SIMD, per wave)\n", name, run1<FORM, NP, NR>(src, sink, 256),
         run1<FORM, NP, NR>(src, sink, 512));
  fflush(stdout);
}

int main() {
  char* src;
  float* sink;
  CK(hipMalloc(&src, 8 << 20));
  CK(hipMemset(src, 1, 8 << 20));
  CK(hipMalloc(&sink, 4096));
  run<0, 0, 0>("48 MFMA 16x16x32 alone", src, sink);
  run<0, 0, 20>("+ 20 ds_read_b128", src, sink);
  run<0, 7, 0>("+ 7 global_load_lds_dwordx4", src, sink);
  run<1, 7, 0>("+ 7 buffer_load_dwordx4 lds", src, sink);
  run<2, 7, 0>("+ 7 global_load_dwordx4 -> regs -> ds_write_b128", src, sink);
  run<0, 7, 20>("+ 7 global_load_lds_dwordx4 + 20 ds_read_b128", src, sink);
  run<1, 7, 20>("+ 7 buffer_load_dwordx4 lds + 20 ds_read_b128", src, sink);
  run<2, 7, 20>("+ 7 (load -> regs -> ds_write_b128) + 20 ds_read_b128", src, sink);
  run<0, 10, 20>("+ 10 global_load_lds_dwordx4 + 20 ds_read_b128", src, sink);
  run<0, 4, 20>("+ 4 global_load_lds_dwordx4 + 20 ds_read_b128", src, sink);
  return 0;
}
