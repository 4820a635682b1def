// d3pm_mfma_attn_lat.hip -- attention of a DiT block at one or two utterances: the key tiles of a query group are split over the
// four waves of a workgroup (head_dim 64, f16 / bf16), gfx950.
//
// An opt-in schedule (d3pm_tuning.attn_query_groups = 4) of the need_weights branch of torch's multi_head_attention_forward as
// called by DiTBlock.forward (/root/reference/vall_e/vall_e/ar_discrete.py:132 self, :138 text, :142 prompt) for the latency
// regime -- the p50 half of BASELINE.json's metric.  There attn_mfma_hd64 (d3pm_mfma_attn.hip) runs 96 workgroups whose four waves
// walk the SAME twelve key tiles of an utterance one after the other behind one barrier each.  Here the four waves of a workgroup
// share ONE group of 32 queries and each takes every fourth key tile:
//   * a wave stages its own tiles (global -> registers -> its private 16 KiB of LDS, the next tile's loads in flight under the
//     current tile's arithmetic) and reads back only what it wrote, so the walk needs NO workgroup barrier -- LDS operations of
//     one wave execute in order;
//   * per tile the arithmetic is attn_mfma_hd64's (S^T = K . Q^T on v_mfma_f32_16x16x32 from -m_ref, deferred maximum, the
//     exponentiated scores as the B operand of O^T += V^T . P^T with V read column-major by ds_read_b64_tr_b16, row sums on the
//     matrix pipe), two 16-query groups per wave;
//   * at the end the four partial results (m_w, l_w, O_w: flash-style partial softmax states) meet in LDS and wave w finishes 16
//     of the 64 output columns: O = sum_w 2^(m_w - M) O_w / sum_w 2^(m_w - M) l_w, M = max_w m_w.
// The chain is three tiles (self-attention, 768 keys) or one (the 50-key text and 225-key prompt problems of a block, which ride
// in one launch as the two halves of the grid) plus the combine.  Measured (tests/ab_latency.py, same process, interleaved arms):
// p50 of one utterance 38.8 vs 39.6 ms, of two utterances 57.5 vs 55.5 ms -- the kernel itself takes the 8.6 us of the one it
// replaces, because what a launch waits for is the K / V image every workgroup streams through its CU (196 KB at 768 keys, about
// 3 us at the 60-70 GB/s one CU reads from L2), not the tile chain.  Hence opt-in: the automatic choice keeps one utterance on the
// schedule -- and the bits -- of the batches up to ten.  Same numerics class as the other flash-style kernels (fp32 scores in the
// log2 domain, un-normalised 16-bit probabilities, fp32 row sums): results agree with them to rounding noise, not bit for bit.
#include "d3pm_kernels.h"

namespace d3pm {
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));

constexpr int HD = 64, BKV = 64, ROWB = 128, NS = 4;
constexpr int TILE = BKV * ROWB;   // 8 KiB per K or V tile
constexpr int OSTR = 68;           // floats per query row of a partial O in LDS (64 + 4: 16-byte reads of 16 rows spread over the banks)
constexpr float kDefer = 8.0f;

template <typename T> __device__ __forceinline__ floatx4 mma(uint4 a, uint4 b, floatx4 c);
template <> __device__ __forceinline__ floatx4 mma<f16>(uint4 a, uint4 b, floatx4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ floatx4 mma<bf16>(uint4 a, uint4 b, floatx4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ int k_off(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }
__device__ __forceinline__ int v_off(int row, int chunk) { return row * ROWB + ((chunk ^ (((row >> 1) & 3) << 1)) << 4); }
__device__ __forceinline__ float max_over_query_lanes(float x) {      // lanes l, l ^ 16, l ^ 32, l ^ 48 share a query
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  x = fmaxf(a, b);
  a = x;
  b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}
template <typename T> __device__ __forceinline__ uint32_t pack2(float a, float b) {
  typedef float float2v __attribute__((ext_vector_type(2)));
  typedef T pair __attribute__((ext_vector_type(2)));
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((float2v){a, b}, pair));
}

template <typename T>
__global__ __launch_bounds__(256, 2) void attn_split_hd64(const T* __restrict__ Q, int ldq, const T* __restrict__ Kp,
                                                          const T* __restrict__ Vp, int ldkv, T* __restrict__ O, int ldo, int Tq, int S,
                                                          float scale, int H, int n_qblocks, const T* __restrict__ Q2,
                                                          const T* __restrict__ K2, const T* __restrict__ V2, T* __restrict__ O2,
                                                          int S2, int n_first) {
  constexpr int QG = 2;
  __shared__ __attribute__((aligned(16))) char smem[NS * 2 * TILE];   // [wave][K tile | V tile]; afterwards the partial results
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid;
  {
    const int nblocks = gridDim.x, q = nblocks >> 3, r = nblocks & 7, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  if (bid >= n_first) {   // block-uniform: the second half of a paired grid takes problem 2
    bid -= n_first;
    Q = Q2; Kp = K2; Vp = V2; O = O2; S = S2;
  }
  const int qb = bid % n_qblocks, h = (bid / n_qblocks) % H, b = bid / (n_qblocks * H);
  const int q0 = qb * (16 * QG);                         // every wave of the workgroup works on these queries
  const int qi = lane & 15, g = lane >> 4;
  const T* Kb = Kp + static_cast<size_t>(b) * S * ldkv + h * HD;
  const T* Vb = Vp + static_cast<size_t>(b) * S * ldkv + h * HD;

  const float qscale = scale * 1.4426950408889634f;
  uint4 qf[QG][2];
#pragma unroll
  for (int qg = 0; qg < QG; ++qg) {
    int qrow = q0 + qg * 16 + qi;
    qrow = qrow < Tq ? qrow : Tq - 1;
    const T* qp = Q + (static_cast<size_t>(b) * Tq + qrow) * ldq + h * HD;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      typedef T tvec8 __attribute__((ext_vector_type(8)));
      const tvec8 e = __builtin_bit_cast(tvec8, *reinterpret_cast<const uint4*>(qp + ks * 32 + g * 8));
      qf[qg][ks] = uint4{pack2<T>(static_cast<float>(e[0]) * qscale, static_cast<float>(e[1]) * qscale),
                         pack2<T>(static_cast<float>(e[2]) * qscale, static_cast<float>(e[3]) * qscale),
                         pack2<T>(static_cast<float>(e[4]) * qscale, static_cast<float>(e[5]) * qscale),
                         pack2<T>(static_cast<float>(e[6]) * qscale, static_cast<float>(e[7]) * qscale)};
    }
  }

  // wave-private staging: lane -> rows 8 i + (lane >> 3), i = 0..7, 16-byte chunk lane & 7 of the K tile and of the V tile
  const int lrow = lane >> 3, chs = lane & 7;
  // the tile in flight lives in sixteen named-by-index registers (static indices only: a by-value struct of arrays went to scratch)
  uint4 sk[8], sv[8];
  auto load_tile = [&](int tile) __attribute__((always_inline)) {
    if ((tile + 1) * BKV <= S) {
      const char* kt = reinterpret_cast<const char*>(Kb + (static_cast<size_t>(tile) * BKV + lrow) * ldkv + chs * 8);
      const char* vt = reinterpret_cast<const char*>(Vb + (static_cast<size_t>(tile) * BKV + lrow) * ldkv + chs * 8);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        sk[i] = *reinterpret_cast<const uint4*>(kt + static_cast<size_t>(8 * i) * ldkv * 2);
        sv[i] = *reinterpret_cast<const uint4*>(vt + static_cast<size_t>(8 * i) * ldkv * 2);
      }
    } else {                                             // ragged last tile: rows past the end repeat the last key (masked below)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        int key = tile * BKV + 8 * i + lrow;
        key = key < S ? key : S - 1;
        sk[i] = *reinterpret_cast<const uint4*>(Kb + static_cast<size_t>(key) * ldkv + chs * 8);
        sv[i] = *reinterpret_cast<const uint4*>(Vb + static_cast<size_t>(key) * ldkv + chs * 8);
      }
    }
  };
  char* const wbase = smem + wave * 2 * TILE;
  auto store_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      *reinterpret_cast<uint4*>(wbase + k_off(8 * i + lrow, chs)) = sk[i];
      *reinterpret_cast<uint4*>(wbase + TILE + v_off(8 * i + lrow, chs)) = sv[i];
    }
  };

  const int n_tiles = (S + BKV - 1) / BKV;
  float m_ref[QG];
  floatx4 negm[QG], acc_o[QG][4], acc_l[QG];
  const uint32_t one2 = pack2<T>(1.0f, 1.0f);
  const uint4 ones = uint4{one2, one2, one2, one2};
#pragma unroll
  for (int qg = 0; qg < QG; ++qg) {
    m_ref[qg] = 0.f;
    negm[qg] = floatx4{0.f, 0.f, 0.f, 0.f};
    acc_l[qg] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc_o[qg][dt] = floatx4{0.f, 0.f, 0.f, 0.f};
  }
  int ok[2], ov[4];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) ok[ks] = k_off(qi, ks * 4 + g);
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    const int col = dt * 16 + 4 * (qi & 3);
    ov[dt] = v_off(4 * g + (qi >> 2), col >> 3) + (col & 7) * 2;
  }

  const bool any_tile = wave < n_tiles;                  // wave-uniform
#pragma unroll
  for (int i = 0; i < 8; ++i) { sk[i] = uint4{0u, 0u, 0u, 0u}; sv[i] = uint4{0u, 0u, 0u, 0u}; }
  if (any_tile) load_tile(wave);
  bool first = true;
  for (int tile = wave; tile < n_tiles; tile += NS) {    // wave-uniform trip count; no workgroup barrier inside
    store_tile();                                        // behind the previous tile's fragment reads (one wave's LDS operations run in order)
    if (tile + NS < n_tiles) load_tile(tile + NS);
    const char* kb = wbase;
    const char* vb = wbase + TILE;
    floatx4 s[QG][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const uint4 kf = *reinterpret_cast<const uint4*>(kb + ok[ks] + kt * 16 * ROWB);
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) s[qg][kt] = mma<T>(kf, qf[qg][ks], ks == 0 ? negm[qg] : s[qg][kt]);
      }
    const bool ragged = (tile == n_tiles - 1) && (S & (BKV - 1));
    uint4 pf[QG][2];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
      if (ragged) {
        const int lim = S - tile * BKV - 4 * g;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[qg][kt][r] = (kt * 16 + r < lim) ? s[qg][kt][r] : -INFINITY;
      }
      float mx = fmaxf(s[qg][0][0], s[qg][0][1]);
      mx = fmaxf(fmaxf(mx, s[qg][0][2]), s[qg][0][3]);
#pragma unroll
      for (int kt = 1; kt < 4; ++kt) {
        mx = fmaxf(fmaxf(mx, s[qg][kt][0]), s[qg][kt][1]);
        mx = fmaxf(fmaxf(mx, s[qg][kt][2]), s[qg][kt][3]);
      }
      if (first || __any(mx > kDefer)) {                 // wave-uniform
        mx = max_over_query_lanes(mx);
        const float delta = first ? mx : fmaxf(mx, 0.f);
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) s[qg][kt][r] -= delta;
        if (!first) {
          const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
          for (int r = 0; r < 4; ++r) acc_l[qg][r] *= alpha;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc_o[qg][dt][r] *= alpha;
        }
        m_ref[qg] += delta;
        negm[qg] = floatx4{-m_ref[qg], -m_ref[qg], -m_ref[qg], -m_ref[qg]};
      }
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[qg][kt][r] = __builtin_amdgcn_exp2f(s[qg][kt][r]);
#pragma unroll
      for (int kb2 = 0; kb2 < 2; ++kb2) {
        const floatx4 pa = s[qg][2 * kb2], pb = s[qg][2 * kb2 + 1];
        pf[qg][kb2] = uint4{pack2<T>(pa[0], pa[1]), pack2<T>(pa[2], pa[3]), pack2<T>(pb[0], pb[1]), pack2<T>(pb[2], pb[3])};
      }
    }
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        typedef short4v __attribute__((address_space(3))) * lds_ptr;
        const short4v va = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(vb + ov[dt] + (2 * kb2) * 16 * ROWB));
        const short4v vc = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(vb + ov[dt] + (2 * kb2 + 1) * 16 * ROWB));
        const uint2 lo = __builtin_bit_cast(uint2, va), hi = __builtin_bit_cast(uint2, vc);
        const uint4 vf = uint4{lo.x, lo.y, hi.x, hi.y};
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) acc_o[qg][dt] = mma<T>(vf, pf[qg][kb2], acc_o[qg][dt]);
      }
#pragma unroll
    for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
      for (int qg = 0; qg < QG; ++qg) acc_l[qg] = mma<T>(ones, pf[qg][kb2], acc_l[qg]);
    first = false;
  }

  // ---- combine the four partial softmax states.  LDS is reused: every wave is past its last fragment read at the barrier
  __syncthreads();
  float* const cm = reinterpret_cast<float*>(smem);                 // [wave][QG][16] running references (log2 domain)
  float* const cl = cm + NS * QG * 16;                              // [wave][QG][16] row sums
  float* const co = cl + NS * QG * 16;                              // [wave][QG][16][OSTR] un-normalised outputs
#pragma unroll
  for (int qg = 0; qg < QG; ++qg) {
    if (g == 0) {
      cm[(wave * QG + qg) * 16 + qi] = any_tile ? m_ref[qg] : -INFINITY;
      cl[(wave * QG + qg) * 16 + qi] = acc_l[qg][0];
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      *reinterpret_cast<floatx4*>(co + ((wave * QG + qg) * 16 + qi) * OSTR + dt * 16 + 4 * g) = acc_o[qg][dt];
  }
  __syncthreads();
#pragma unroll
  for (int qg = 0; qg < QG; ++qg) {
    float mw[NS], M = -INFINITY;
#pragma unroll
    for (int w = 0; w < NS; ++w) { mw[w] = cm[(w * QG + qg) * 16 + qi]; M = fmaxf(M, mw[w]); }
    float L = 0.f;
    floatx4 o = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < NS; ++w) {
      const float sc = __builtin_amdgcn_exp2f(mw[w] - M);            // 0 for a wave without tiles (m = -inf; wave 0 always has one)
      L += sc * cl[(w * QG + qg) * 16 + qi];
      const floatx4 ow = *reinterpret_cast<const floatx4*>(co + ((w * QG + qg) * 16 + qi) * OSTR + wave * 16 + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] += sc * ow[r];
    }
    const float inv = 1.0f / L;
    const int qrow = q0 + qg * 16 + qi;
    if (qrow < Tq) {
      T* op = O + (static_cast<size_t>(b) * Tq + qrow) * ldo + h * HD + wave * 16 + 4 * g;
      *reinterpret_cast<uint2*>(op) = uint2{pack2<T>(o[0] * inv, o[1] * inv), pack2<T>(o[2] * inv, o[3] * inv)};
    }
  }
}

inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

}  // namespace

// one problem or a pair, no key lengths; the caller (mfma_attention) decides by grid size when this schedule is the one to run
bool mfma_attention_split_supported(int dtype, const AttnArgs& a) {
  if (dtype != D3PM_F16 && dtype != D3PM_BF16) return false;
  if (a.hd != HD || a.key_len != nullptr || a.S < 1 || a.Tq < 1) return false;
  if (a.ldq % 8 || a.ldkv % 8 || a.ldo % 4) return false;
  if (a.Q2 && !(a.S2 >= 1 && aligned(a.Q2, 16) && aligned(a.K2, 16) && aligned(a.V2, 16) && aligned(a.O2, 8))) return false;
  return aligned(a.Q, 16) && aligned(a.K, 16) && aligned(a.V, 16) && aligned(a.O, 8);
}

int mfma_attention_split(int dtype, const AttnArgs& a, hipStream_t s) {
  const int n_qblocks = (a.Tq + 31) / 32;
  const int n_first = n_qblocks * a.H * a.B;
  const dim3 grid(static_cast<unsigned>(n_first) * (a.Q2 ? 2 : 1)), block(256);
  if (dtype == D3PM_F16)
    attn_split_hd64<f16><<<grid, block, 0, s>>>(static_cast<const f16*>(a.Q), a.ldq, static_cast<const f16*>(a.K), static_cast<const f16*>(a.V),
                                                a.ldkv, static_cast<f16*>(a.O), a.ldo, a.Tq, a.S, a.scale, a.H, n_qblocks,
                                                static_cast<const f16*>(a.Q2), static_cast<const f16*>(a.K2), static_cast<const f16*>(a.V2),
                                                static_cast<f16*>(a.O2), a.S2, n_first);
  else
    attn_split_hd64<bf16><<<grid, block, 0, s>>>(static_cast<const bf16*>(a.Q), a.ldq, static_cast<const bf16*>(a.K), static_cast<const bf16*>(a.V),
                                                 a.ldkv, static_cast<bf16*>(a.O), a.ldo, a.Tq, a.S, a.scale, a.H, n_qblocks,
                                                 static_cast<const bf16*>(a.Q2), static_cast<const bf16*>(a.K2), static_cast<const bf16*>(a.V2),
                                                 static_cast<bf16*>(a.O2), a.S2, n_first);
  D3PM_LAUNCH_CHECK();
  return D3PM_OK;
}

}  // namespace d3pm
