// d3pm_final_sample.hip -- final projection + D3PM posterior + Gumbel-max draw in ONE kernel: the [rows][1025] logits
// never reach HBM.
//
// Replaces, per diffusion iteration (paths under /root/reference/vall_e/vall_e/):
//   logits = final(x * mask)                         ar_discrete.py:773-776   (nn.Linear d -> 1025)
//   x_{t-1} = p_sample(logits, t, x_t)               ar_discrete.py:401-420 with q_posterior_logits :347-375,
//                                                    _at :337-345, _at_onehot :377-400
// The two-launch form (d3pm_mfma_gemm*.hip then d3pm_sample.hip) writes 24576 x 1032 logits (50.7 MB at the bench shape)
// and reads them back: 100 MB and a kernel boundary per iteration for numbers that are consumed once.
//
// A workgroup (4 waves) owns 32 canvas rows and all 1025 classes:
//   phase 0  the 32 x d hidden rows go to LDS once (16-byte chunks XOR-swizzled by the row, conflict-free fragment reads);
//   phase 1  logits^T = W . x^T on the matrix cores: wave w owns classes 256 w .. 256 w + 255 (16 MFMA tiles of 16
//            classes; wave 3 also the tile that holds class 1024), 2 row tiles each -> 34 accumulator tiles; the weight
//            fragments are read straight from L2 into registers (a class row is consumed by ONE wave: staging it in LDS
//            would buy no reuse) while the x fragments of the k-step come from LDS (2 reads per 34 MFMAs);
//            same v_mfma_f32_16x16x32, same ascending-k accumulation as the stand-alone GEMM: identical fp32 sums;
//   phase 2  z = round_to_model_dtype(acc + bias) -> LDS [32][1040] (the x panel is dead by then: same memory);
//   phase 3  each wave draws 8 rows with the very code of the stand-alone sampler (d3pm_sample_row.h), reading the row's
//            logits from LDS instead of HBM: same grouping of classes over lanes, same reduction order -> the ids are
//            bit-identical to the two-launch path (tests/test_gpu_kernels.py).
// Two workgroups share a CU (66.5 KB of LDS each), so one's sampling arithmetic (VALU, transcendentals, Philox) runs
// under the other's MFMAs.
// Compiled into libd3pm_hip_ab.so only (-DD3PM_ABLATIONS; include/d3pm_hip_ab.h): built, measured, not shipped.
#ifdef D3PM_ABLATIONS
#include "d3pm_kernels.h"
#include "d3pm_mfma_tile.h"
#include "d3pm_sample_row.h"

namespace d3pm {
namespace {

constexpr int FS_ROWS = 32;            // canvas rows per workgroup
constexpr int FS_ZLD = 1040;           // row stride of the logits image in LDS (elements)
constexpr int FS_CLASSES = 1025;

template <typename T>
__global__ __launch_bounds__(256, 2) void final_sample_fused(const T* __restrict__ X, int ldx, const T* __restrict__ W,
                                                             const T* __restrict__ bias, const int32_t* x_t, int32_t* x_next,
                                                             int32_t* x_next2, int rows, int d, int mask_id, uint64_t seed,
                                                             const uint64_t* __restrict__ seed_hbm, uint32_t row0, int greedy,
                                                             PosteriorConsts pc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * FS_ROWS;
  const int row_bytes = d * 2;
  // ---- phase 0: hidden rows -> LDS, chunk c of row r at position c ^ (r & 7)
  const int cpr = d >> 3;
  for (int idx = tid; idx < FS_ROWS * cpr; idx += 256) {
    const int r = idx / cpr, c = idx - r * cpr;
    int mr = m0 + r;
    mr = mr < rows ? mr : rows - 1;
    const uint4 v = *reinterpret_cast<const uint4*>(X + static_cast<size_t>(mr) * ldx + c * 8);
    *reinterpret_cast<uint4*>(smem + r * row_bytes + ((c ^ (r & 7)) << 4)) = v;
  }
  __syncthreads();
  // ---- phase 1: logits^T tiles
  floatx4 acc[17][2];
#pragma unroll
  for (int a = 0; a < 17; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fch = lane >> 4;
  const T* wb = W + static_cast<size_t>(wave * 256 + frow) * d + fch * 8;          // class rows of this wave's 16 tiles
  const T* wlast = W + static_cast<size_t>(FS_CLASSES - 1) * d + fch * 8;           // tile 64: only class 1024 is real
  const int nks = d >> 5;
  for (int ks = 0; ks < nks; ++ks) {
    uint4 xf[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
      xf[mt] = *reinterpret_cast<const uint4*>(smem + (mt * 16 + frow) * row_bytes + (((ks * 4 + fch) ^ (frow & 7)) << 4));
#pragma unroll
    for (int nt = 0; nt < 16; ++nt) {
      const uint4 a = *reinterpret_cast<const uint4*>(wb + static_cast<size_t>(nt * 16) * d + ks * 32);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = mma<T>(a, xf[mt], acc[nt][mt]);
    }
    if (wave == 3) {
      const uint4 a = *reinterpret_cast<const uint4*>(wlast + ks * 32);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) acc[16][mt] = mma<T>(a, xf[mt], acc[16][mt]);
    }
  }
  __syncthreads();                                        // the x panel is dead: its memory becomes the logits image
  // ---- phase 2: z = rn(acc + bias) in the model dtype -> LDS [32][1040]
  T* zs = reinterpret_cast<T*>(smem);
#pragma unroll
  for (int nt = 0; nt < 17; ++nt) {
    if (nt == 16 && wave != 3) break;
    const int j0 = (nt < 16 ? wave * 256 + nt * 16 : 1024) + fch * 4;          // 4 consecutive classes of this lane
    if (j0 >= FS_ZLD) continue;
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = bias ? static_cast<float>(bias[j0 + r < FS_CLASSES ? j0 + r : FS_CLASSES - 1]) : 0.f;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      Pack4<T> o;
#pragma unroll
      for (int r = 0; r < 4; ++r) o.v[r] = static_cast<T>(acc[nt][mt][r] + bv[r]);
      *reinterpret_cast<Pack4<T>*>(zs + (mt * 16 + frow) * FS_ZLD + j0) = o;
    }
  }
  __syncthreads();
  // ---- phase 3: one wave per row, 8 rows per wave
  if (seed_hbm) seed = *seed_hbm;
  typedef const __attribute__((address_space(3))) T* lds_row;
  for (int rr = 0; rr < FS_ROWS / 4; ++rr) {
    const int rl = wave * (FS_ROWS / 4) + rr, row = m0 + rl;
    if (row >= rows) break;                               // wave-uniform
    const int best_j = sample_row<T>((lds_row)(zs + rl * FS_ZLD), FS_CLASSES, mask_id, x_t[row], seed,
                                     row0 + static_cast<uint32_t>(row), greedy, pc, nullptr, lane);
    if (lane == 0) {
      x_next[row] = best_j;
      if (x_next2) x_next2[row] = best_j;
    }
  }
}

}  // namespace

bool final_sample_supported(int dtype, int n_classes, int d, const void* X, int ldx, const void* W) {
  if (dtype != D3PM_F16 && dtype != D3PM_BF16) return false;
  if (n_classes != FS_CLASSES || d < 32 || d % 32 != 0 || FS_ROWS * d * 2 > FS_ROWS * FS_ZLD * 2) return false;
  if (ldx % 8 != 0) return false;
  return (reinterpret_cast<uintptr_t>(X) % 16) == 0 && (reinterpret_cast<uintptr_t>(W) % 16) == 0;
}

// x: hidden rows [rows][d] (already multiplied by the frame mask, ar_discrete.py:161,773); W [1025][d]; ids -> a.x_next
int final_sample(int dtype, const void* X, int ldx, const void* W, const void* bias, int d, const SampleArgs& a, hipStream_t s) {
  const dim3 grid(static_cast<unsigned>((a.rows + FS_ROWS - 1) / FS_ROWS)), block(256);
  const size_t lds = static_cast<size_t>(FS_ROWS) * FS_ZLD * 2;
#define D3PM_FS(U)                                                                                                        \
  do {                                                                                                                    \
    D3PM_LDS_ATTR((&final_sample_fused<U>), 80 * 1024);                                                                   \
    final_sample_fused<U><<<grid, block, lds, s>>>(static_cast<const U*>(X), ldx, static_cast<const U*>(W),               \
                                                   static_cast<const U*>(bias), a.x_t, a.x_next, a.x_next2, a.rows, d,    \
                                                   a.mask_id, a.seed, a.seed_hbm, a.row0, a.greedy, a.pc);                \
  } while (0)
  if (dtype == D3PM_F16) D3PM_FS(f16);
  else D3PM_FS(bf16);
#undef D3PM_FS
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

}  // namespace d3pm

#endif  // D3PM_ABLATIONS
