// probe_mx.hip -- what v_mfma_scale_f32_16x16x128_f8f6f4 does with its operands on gfx950 (run once on the GPU box;
// the findings are recorded in csrc/d3pm_fp8.hip and DESIGN.md).  hipcc --offload-arch=gfx950 -O2 tools/probe_mx.hip -o probe_mx
//
//   D[16][16] += sum_k A[i][k] * 2^(sa - 127) * B[k][j] * 2^(sb - 127),  K = 128, e4m3 operands (cbsz = blgp = 0)
//
// Questions: (1) which (row, k) does byte b of lane l's 32-byte A operand hold; (2) whose scale byte applies to it;
// (3) what op_sel selects; (4) the C/D layout.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef int intx8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int OPA, int OPB>
__global__ void mx(const intx8* a, const intx8* b, const int* sa, const int* sb, floatx4* d) {
  const int l = threadIdx.x;
  floatx4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[l], b[l], c, 0, 0, OPA, sa[l], OPB, sb[l]);
  d[l] = c;
}

static uint8_t e4m3(int v) {   // small integers 0, 1, 2, 3, 4 ... exactly representable
  static const uint8_t code[] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4a, 0x4c, 0x4e, 0x50};
  return code[v];
}

struct Dev {
  uint8_t *a, *b; int *sa, *sb; float* d;
};

static void run(Dev& dv, const std::vector<uint8_t>& A, const std::vector<uint8_t>& B, const std::vector<int>& SA,
                const std::vector<int>& SB, int opa, int opb, float (&D)[64][4]) {
  hipMemcpy(dv.a, A.data(), 64 * 32, hipMemcpyHostToDevice);
  hipMemcpy(dv.b, B.data(), 64 * 32, hipMemcpyHostToDevice);
  hipMemcpy(dv.sa, SA.data(), 64 * 4, hipMemcpyHostToDevice);
  hipMemcpy(dv.sb, SB.data(), 64 * 4, hipMemcpyHostToDevice);
  auto A_ = reinterpret_cast<const intx8*>(dv.a);
  auto B_ = reinterpret_cast<const intx8*>(dv.b);
  auto D_ = reinterpret_cast<floatx4*>(dv.d);
  if (opa == 0 && opb == 0) mx<0, 0><<<1, 64>>>(A_, B_, dv.sa, dv.sb, D_);
  else if (opa == 1 && opb == 0) mx<1, 0><<<1, 64>>>(A_, B_, dv.sa, dv.sb, D_);
  else if (opa == 2 && opb == 0) mx<2, 0><<<1, 64>>>(A_, B_, dv.sa, dv.sb, D_);
  else if (opa == 3 && opb == 0) mx<3, 0><<<1, 64>>>(A_, B_, dv.sa, dv.sb, D_);
  else mx<0, 1><<<1, 64>>>(A_, B_, dv.sa, dv.sb, D_);
  hipDeviceSynchronize();
  hipMemcpy(D, dv.d, 64 * 16, hipMemcpyDeviceToHost);
}

int main() {
  Dev dv;
  hipMalloc(&dv.a, 64 * 32); hipMalloc(&dv.b, 64 * 32); hipMalloc(&dv.sa, 256); hipMalloc(&dv.sb, 256); hipMalloc(&dv.d, 1024);
  float D[64][4];
  std::vector<uint8_t> A(64 * 32), B(64 * 32);
  std::vector<int> SA(64, 127), SB(64, 127);

  // (4) C/D layout + (1) rows: A[lane l] = value (l & 15) + 1 in byte 0 only ... use B = all ones
  // experiment 1: A all ones, B all ones -> every D = 128
  std::fill(A.begin(), A.end(), e4m3(1)); std::fill(B.begin(), B.end(), e4m3(1));
  run(dv, A, B, SA, SB, 0, 0, D);
  printf("exp1 all-ones: D[0][0]=%g D[63][3]=%g (expect 128)\n", D[0][0], D[63][3]);

  // experiment 2: only lane L of A non-zero (all 32 bytes = 1), B all ones: which D rows/cols light up, with what value
  for (int L : {0, 5, 16, 37, 63}) {
    std::fill(A.begin(), A.end(), 0);
    for (int i = 0; i < 32; ++i) A[L * 32 + i] = e4m3(1);
    run(dv, A, B, SA, SB, 0, 0, D);
    printf("exp2 A-lane %2d only:", L);
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (D[l][r] != 0.f) { printf(" D[lane %d][reg %d]=%g", l, r, D[l][r]); if (l > 3) { l = 64; break; } }
    printf("\n");
  }
  // experiment 3: only lane L of B non-zero
  std::fill(A.begin(), A.end(), e4m3(1));
  for (int L : {0, 5, 16, 37}) {
    std::fill(B.begin(), B.end(), 0);
    for (int i = 0; i < 32; ++i) B[L * 32 + i] = e4m3(1);
    run(dv, A, B, SA, SB, 0, 0, D);
    printf("exp3 B-lane %2d only: nonzero D at", L);
    int n = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (D[l][r] != 0.f && n++ < 6) printf(" [lane %d][reg %d]=%g", l, r, D[l][r]);
    printf(" (%d entries)\n", n);
  }
  // experiment 4: k mapping inside a lane: A lane 0 byte i = 1 only, B lane Lb byte j = 1 only: D != 0 iff same k
  printf("exp4 k-match (A lane La byte i) x (B lane Lb byte j): pairs with D != 0\n");
  for (int La : {0, 16, 32, 48}) for (int i : {0, 1, 4, 15, 16, 31}) {
    std::fill(A.begin(), A.end(), 0); A[La * 32 + i] = e4m3(1);
    for (int Lb : {0, 16, 32, 48}) for (int j : {0, 1, 4, 15, 16, 31}) {
      std::fill(B.begin(), B.end(), 0); B[Lb * 32 + j] = e4m3(1);
      run(dv, A, B, SA, SB, 0, 0, D);
      float s = 0; for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) s += D[l][r];
      if (s != 0.f) printf("   A(l%d,b%d) ~ B(l%d,b%d): %g\n", La, i, Lb, j, s);
    }
  }
  // experiment 5: scales: A, B all ones; scale_a byte0 of lane Ls = 128 (x2): which D change
  std::fill(A.begin(), A.end(), e4m3(1)); std::fill(B.begin(), B.end(), e4m3(1));
  for (int Ls : {0, 5, 16, 37, 63}) {
    std::fill(SA.begin(), SA.end(), 127); SA[Ls] = 128;
    run(dv, A, B, SA, SB, 0, 0, D);
    printf("exp5 scale_a lane %2d = 2x:", Ls);
    int n = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (D[l][r] != 128.f && n++ < 4) printf(" D[lane %d][reg %d]=%g", l, r, D[l][r]);
    printf(" (%d entries differ)\n", n);
  }
  std::fill(SA.begin(), SA.end(), 127);
  for (int Ls : {0, 5, 16, 37}) {
    std::fill(SB.begin(), SB.end(), 127); SB[Ls] = 129;
    run(dv, A, B, SA, SB, 0, 0, D);
    printf("exp5 scale_b lane %2d = 4x:", Ls);
    int n = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) if (D[l][r] != 128.f && n++ < 4) printf(" D[lane %d][reg %d]=%g", l, r, D[l][r]);
    printf(" (%d entries differ)\n", n);
  }
  std::fill(SB.begin(), SB.end(), 127);
  // experiment 6: does the scale of lane L apply exactly to lane L's own 32 values?  A only lane La non-zero; scale lane Ls = 2x
  for (int La : {0, 16, 37}) for (int Ls : {0, 16, 32, 48, 37, 5, 21}) {
    std::fill(A.begin(), A.end(), 0);
    for (int i = 0; i < 32; ++i) A[La * 32 + i] = e4m3(1);
    std::fill(SA.begin(), SA.end(), 127); SA[Ls] = 128;
    run(dv, A, B, SA, SB, 0, 0, D);
    float mx_ = 0; for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) mx_ = D[l][r] > mx_ ? D[l][r] : mx_;
    printf("exp6 A-lane %2d, scale-lane %2d 2x: max D = %g (32 = not applied, 64 = applied)\n", La, Ls, mx_);
  }
  // experiment 7: op_sel picks the byte: scale word = 0x7f808182 -> bytes (lsb first) 0x82 0x81 0x80 0x7f = 8x 4x 2x 1x
  std::fill(A.begin(), A.end(), e4m3(1));
  std::fill(SA.begin(), SA.end(), 0x7f808182);
  for (int op = 0; op < 4; ++op) {
    run(dv, A, B, SA, SB, op, 0, D);
    printf("exp7 op_sel_a = %d: D = %g (128 x {8,4,2,1} = 1024 / 512 / 256 / 128 for byte 0 / 1 / 2 / 3)\n", op, D[0][0]);
  }
  std::fill(SA.begin(), SA.end(), 127);
  std::fill(SB.begin(), SB.end(), 0x7f808182);
  run(dv, A, B, SA, SB, 0, 1, D);
  printf("exp7 op_sel_b = 1: D = %g (expect 512)\n", D[0][0]);
  std::fill(SB.begin(), SB.end(), 127);
  // experiment 8: byte-level map.  For operand lane L = (row r, group g) and byte b: which scale lane r + 16 q (q = 0..3) scales it
  std::fill(B.begin(), B.end(), e4m3(1));
  for (int which = 0; which < 2; ++which)
    for (int L : {3, 19, 35, 51}) {
      printf("exp8 %c lane %2d (row %d, group %d): scale group of byte 0..31 = ", which ? 'B' : 'A', L, L & 15, L >> 4);
      for (int b = 0; b < 32; ++b) {
        std::vector<uint8_t>& OP = which ? B : A;
        std::vector<uint8_t>& OTHER = which ? A : B;
        std::fill(OTHER.begin(), OTHER.end(), e4m3(1));
        std::fill(OP.begin(), OP.end(), 0);
        OP[L * 32 + b] = e4m3(1);
        int found = -1;
        for (int q = 0; q < 4; ++q) {
          std::fill(SA.begin(), SA.end(), 127); std::fill(SB.begin(), SB.end(), 127);
          (which ? SB : SA)[(L & 15) + 16 * q] = 128;
          run(dv, A, B, SA, SB, 0, 0, D);
          float mx_ = 0; for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) mx_ = D[l][r] > mx_ ? D[l][r] : mx_;
          if (mx_ == 2.f) found = q;
        }
        printf("%d", found);
      }
      printf("\n");
    }
  return 0;
}
