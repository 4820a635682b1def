// d3pm_headpad.hip -- head_dim 32 -> 64 re-layout around the condition encoders' self-attention.
//
// The reference's condition encoders are nn.TransformerEncoderLayer(d_model, nhead = 16) (/root/reference/vall_e/vall_e/
// ar_discrete.py:216-230): at d_model = 512 their heads are 32 wide, and the MFMA attention kernels are written for 64 (the DiT
// blocks' width).  Until round 4 those eight attention calls per reverse process ran on the generic FMA kernel (attention_rows):
// 0.06 .. 0.64 ms each at 32 utterances -- 2.8 ms of a 224 ms step, and 7 ms at the VCTK prompt length.  A 32-wide head padded with
// zeros to 64 gives the same scores (the extra products are exact zeros) and an output whose upper 32 columns are zero, so the packed
// projection rows are spread to [rows][3][H][64], the 64-wide kernels run, and the first 32 columns of each head are gathered back:
// two copies of a few MB instead of a scalar-FMA attention.  Scale stays 1 / sqrt(32) (passed explicitly).
#include "d3pm_kernels.h"

namespace d3pm {
namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// in: [rows][groups][hd_in] 16-bit, out: [rows][groups][2 * hd_in]; one 16-byte chunk per thread, zeros in the upper half
__global__ __launch_bounds__(256) void pad_heads_rows(const u32x4* __restrict__ in, u32x4* __restrict__ out, long long chunks_out,
                                                      int chunks_per_head_in) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= chunks_out) return;
  const int cph_out = 2 * chunks_per_head_in;
  const long long head = i / cph_out;
  const int c = static_cast<int>(i - head * cph_out);
  u32x4 v = {0u, 0u, 0u, 0u};
  if (c < chunks_per_head_in) v = in[head * chunks_per_head_in + c];
  out[i] = v;
}

// in: [rows][groups][2 * hd] -> out: [rows][groups][hd] (the first hd columns of every head)
__global__ __launch_bounds__(256) void unpad_heads_rows(const u32x4* __restrict__ in, u32x4* __restrict__ out, long long chunks_out,
                                                        int chunks_per_head_out) {
  const long long i = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= chunks_out) return;
  const long long head = i / chunks_per_head_out;
  const int c = static_cast<int>(i - head * chunks_per_head_out);
  out[i] = in[head * 2 * chunks_per_head_out + c];
}

}  // namespace

// rows x groups heads of `hd` 16-bit elements (hd a multiple of 8; 16-byte aligned buffers)
int pad_heads(const void* in, void* out, long long rows, int groups, int hd, hipStream_t s) {
  D3PM_REQUIRE(in && out && rows > 0 && groups > 0 && hd > 0 && hd % 8 == 0, D3PM_E_ARG, "pad_heads: bad arguments");
  const long long chunks = rows * groups * (2 * hd / 8);
  pad_heads_rows<<<dim3(static_cast<unsigned>((chunks + 255) / 256)), dim3(256), 0, s>>>(
      static_cast<const u32x4*>(in), static_cast<u32x4*>(out), chunks, hd / 8);
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

int unpad_heads(const void* in, void* out, long long rows, int groups, int hd, hipStream_t s) {
  D3PM_REQUIRE(in && out && rows > 0 && groups > 0 && hd > 0 && hd % 8 == 0, D3PM_E_ARG, "unpad_heads: bad arguments");
  const long long chunks = rows * groups * (hd / 8);
  unpad_heads_rows<<<dim3(static_cast<unsigned>((chunks + 255) / 256)), dim3(256), 0, s>>>(
      static_cast<const u32x4*>(in), static_cast<u32x4*>(out), chunks, hd / 8);
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

}  // namespace d3pm
