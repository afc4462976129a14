// Accuracy of an fp32 GEMM computed on v_mfma_f32_32x32x16_bf16 with both operands split into three bf16 pieces (x = h + m + l
// exactly, truncation split) and six of the nine cross products kept -- against float64, next to the plain fp32 MFMA
// (v_mfma_f32_32x32x2_f32) on the same data.  One wave, C[32][32] = A[32][K] * B[K][32].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split8(const float* x, v4u& H, v4u& M, v4u& L) {
  unsigned h[8], m[8], l[8];
  for (int i = 0; i < 8; ++i) {
    const unsigned xb = __builtin_bit_cast(unsigned, x[i]);
    const unsigned hb = xb & 0xFFFF0000u;
    const float r1 = x[i] - __builtin_bit_cast(float, hb);
    const unsigned mb = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
    const float r2 = r1 - __builtin_bit_cast(float, mb);
    h[i] = hb; m[i] = mb; l[i] = __builtin_bit_cast(unsigned, r2);
  }
  for (int i = 0; i < 4; ++i) {
    H[i] = __builtin_amdgcn_perm(h[2 * i + 1], h[2 * i], 0x07060302u);
    M[i] = __builtin_amdgcn_perm(m[2 * i + 1], m[2 * i], 0x07060302u);
    L[i] = __builtin_amdgcn_perm(l[2 * i + 1], l[2 * i], 0x07060302u);
  }
}

__global__ void k(const float* A, const float* B, float* Csplit, float* Cf32, int K, int nterms) {
  const int lane = threadIdx.x, r = lane & 31, kb = lane >> 5;
  v16f acc = {0}, acc2 = {0};
  for (int k0 = 0; k0 < K; k0 += 16) {
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = A[r * K + k0 + kb * 8 + i]; b[i] = B[(k0 + kb * 8 + i) * 32 + r]; }
    v4u ah, am, al, bh, bm, bl;
    split8(a, ah, am, al); split8(b, bh, bm, bl);
#define MF(x, y) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, x), __builtin_bit_cast(v8bf, y), acc, 0, 0, 0)
    if (nterms >= 6) { MF(al, bh); MF(ah, bl); MF(am, bm); }
    if (nterms >= 3) { MF(am, bh); MF(ah, bm); }
    MF(ah, bh);
    for (int i = 0; i < 8; ++i) {       // fp32 MFMA: K = 2 per instruction: k = kb (lanes 0-31: even k, 32-63: odd)
      const float av = A[r * K + k0 + 2 * i + kb], bv = B[(k0 + 2 * i + kb) * 32 + r];
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc2, 0, 0, 0);
    }
  }
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * kb;
    Csplit[row * 32 + r] = acc[i];
    Cf32[row * 32 + r] = acc2[i];
  }
}

int main() {
  for (int K : {288, 2304}) {
    std::vector<float> A(32 * K), B(K * 32);
    srand(1);
    for (auto& v : A) v = fmaxf(0.f, (float)rand() / RAND_MAX * 2.f - 0.7f);      // ReLU-like activations
    for (auto& v : B) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
    float *dA, *dB, *dC, *dD;
    (void)hipMalloc(&dA, A.size() * 4); (void)hipMalloc(&dB, B.size() * 4); (void)hipMalloc(&dC, 4096); (void)hipMalloc(&dD, 4096);
    (void)hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    for (int nt : {1, 3, 6}) {
      k<<<1, 64>>>(dA, dB, dC, dD, K, nt);
      std::vector<float> C(1024), D(1024);
      (void)hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost); (void)hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
      double e1 = 0, e2 = 0, mx = 0, e3 = 0;
      for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
          double s = 0; float f = 0.f;
          for (int kk = 0; kk < K; ++kk) { s += (double)A[i * K + kk] * B[kk * 32 + j]; f = fmaf(A[i * K + kk], B[kk * 32 + j], f); }
          e1 = fmax(e1, fabs(C[i * 32 + j] - s)); e2 = fmax(e2, fabs(D[i * 32 + j] - s)); e3 = fmax(e3, fabs((double)f - s)); mx = fmax(mx, fabs(s));
        }
      printf("K=%4d terms=%d: max|err| / max|C|  split-bf16 %.2e   fp32 MFMA %.2e   host fmaf chain %.2e\n", K, nt, e1 / mx, e2 / mx, e3 / mx);
    }
  }
  return 0;
}
