// probe_attn32.hip -- stand-alone timing probe of the 32 x 32 x 16 self-attention kernels (GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I tts-with-diffusion-model_amd/csrc tools/probe_attn32.hip -o /tmp/probe_attn32
// Runs the plain and the pipelined walk and the pipelined walk's timing-only ablations at one and two workgroups per CU, with the
// start / end stamps of every 32nd workgroup: median shader cycles per workgroup, clock, kernel span.
#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../tts-with-diffusion-model_amd/csrc/d3pm_mfma_attn32.hip"

namespace d3pm {
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
}
}  // namespace d3pm

using namespace d3pm;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <typename K> static void run(const char* name, K kernel, size_t pad, const bf16* qkv, bf16* o, int B, int T, int H) {
  const int d = H * 64, n_qblocks = T / 128;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
  const dim3 grid(n_qblocks * H * B), block(256);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) kernel<<<grid, block, pad, 0>>>(qkv, 3 * d, qkv + d, qkv + 2 * d, 3 * d, o, d, T, T, 0.125f, H, n_qblocks);
  CK(hipEventRecord(e0));
  const int reps = 10;
  for (int i = 0; i < reps; ++i) kernel<<<grid, block, pad, 0>>>(qkv, 3 * d, qkv + d, qkv + 2 * d, 3 * d, o, d, T, T, 0.125f, H, n_qblocks);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long st[192];
  CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_attn32_stamp), sizeof(st)));
  std::vector<double> cyc, ghz;
  for (int i = 0; i < 48; ++i) {
    const double c = static_cast<double>(st[4 * i + 1] - st[4 * i]), us = (st[4 * i + 3] - st[4 * i + 2]) / 100.0;
    cyc.push_back(c);
    ghz.push_back(c / us / 1e3);
  }
  std::sort(cyc.begin(), cyc.end());
  std::sort(ghz.begin(), ghz.end());
  printf("%-58s %7.1f us/launch | cycles per workgroup: min %6.0f median %6.0f max %6.0f | per 64-key tile (median) %5.0f | %.2f GHz\n", name,
         ms / reps * 1e3, cyc.front(), cyc[24], cyc.back(), cyc[24] / (T / 64), ghz[24]);
  fflush(stdout);
}

int main() {
  const int B = 32, T = 768, H = 8, d = 512;
  const size_t n = static_cast<size_t>(B) * T * 3 * d;
  std::vector<uint16_t> h(n);
  uint32_t x = 12345;
  for (size_t i = 0; i < n; ++i) {      // bf16 values in about [-2, 2)
    x = x * 1664525u + 1013904223u;
    const float f = (static_cast<int>(x >> 8) % 4096 - 2048) / 1024.0f;
    uint32_t u;
    memcpy(&u, &f, 4);
    h[i] = static_cast<uint16_t>(u >> 16);
  }
  bf16 *qkv, *o;
  CK(hipMalloc(&qkv, n * 2));
  CK(hipMalloc(&o, static_cast<size_t>(B) * T * d * 2));
  CK(hipMemcpy(qkv, h.data(), n * 2, hipMemcpyHostToDevice));
  const size_t one = 72 * 1024;
  const size_t two = 16 * 1024;          // + 48 KiB static: two workgroups of the pipelined kernel per CU (it fits three)
#define RUN(NAME, ...) run(NAME " [max wg/cu]", __VA_ARGS__, 0, qkv, o, B, T, H); run(NAME " [2 wg/cu]", __VA_ARGS__, two, qkv, o, B, T, H); run(NAME " [1 wg/cu]", __VA_ARGS__, one, qkv, o, B, T, H)
  RUN("plain walk", attn32_hd64<bf16, 1>);
  RUN("pipelined walk", attn32p_hd64<bf16, 1, 0>);
  RUN("pipelined: no exp", attn32p_hd64<bf16, 1, 1>);
  RUN("pipelined: no S products", attn32p_hd64<bf16, 1, 2>);
  RUN("pipelined: no P.V products", attn32p_hd64<bf16, 1, 4>);
  RUN("pipelined: no products at all", attn32p_hd64<bf16, 1, 6>);
  RUN("pipelined: no V reads", attn32p_hd64<bf16, 1, 8>);
  RUN("pipelined: no K reads", attn32p_hd64<bf16, 1, 16>);
  RUN("pipelined: no K / V reads", attn32p_hd64<bf16, 1, 24>);
  RUN("pipelined: no staging, no barrier", attn32p_hd64<bf16, 1, 32>);
  RUN("pipelined: no maximum / test", attn32p_hd64<bf16, 1, 64>);
  RUN("pipelined: no packing", attn32p_hd64<bf16, 1, 128>);
  RUN("pipelined: no row sum", attn32p_hd64<bf16, 1, 256>);
  RUN("pipelined: no exp / max / pack / sum (products only)", attn32p_hd64<bf16, 1, 1 | 64 | 128 | 256>);
  RUN("pipelined: products only, no LDS reads, no staging", attn32p_hd64<bf16, 1, 1 | 64 | 128 | 256 | 24 | 32>);
  RUN("pipelined: vector work only (no products, reads, staging)", attn32p_hd64<bf16, 1, 6 | 24 | 32>);
  return 0;
}
