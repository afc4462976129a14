// Does a plain VALU instruction stream co-execute with MFMAs on gfx950?  One wave per SIMD (256 threads per CU-resident block) or two;
// each wave runs ITER iterations of {NM MFMAs on independent accumulators, NV v_fma_f32 on independent registers}.
//   hipcc --offload-arch=gfx950 -O3 mfma_valu_coexec.hip -o coexec && ./coexec
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

template <int KIND, int NM, int NV>
__global__ void __launch_bounds__(256) k(float* out, int iters, float s) {
  v16f acc[4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float a = threadIdx.x * 1e-3f, b = s;
  v8bf pa, pb;
  for (int j = 0; j < 8; ++j) { pa[j] = (__bf16)(a + j); pb[j] = (__bf16)(b + j); }
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = a + j;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      if (KIND == 0) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m & 3], 0, 0, 0);
      else acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pb, acc[m & 3], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NV; ++q) v[(m * NV + q) & 7] = __builtin_fmaf(v[(m * NV + q) & 7], s, 1.0f);
    }
  }
  float r = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) r += acc[i][j];
  for (int j = 0; j < 8; ++j) r += v[j];
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int KIND, int NM, int NV>
void run(const char* name, int wgs_per_cu) {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  const int iters = 20000, grid = 256 * wgs_per_cu;
  k<KIND, NM, NV><<<grid, 256>>>(out, 100, 1.0001f);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<KIND, NM, NV><<<grid, 256>>>(out, iters, 1.0001f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // per wave per iteration
  const double ns_per_iter = ms * 1e6 / iters;
  printf("%-28s waves/SIMD=%d  MFMA/iter=%d VALU/MFMA=%d : %8.1f ns/iter  -> %6.1f ns per MFMA-slot per SIMD\n", name, wgs_per_cu, NM, NV,
         ns_per_iter, ns_per_iter / (NM * wgs_per_cu));
  hipFree(out);
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<0, 8, 0>("f32 mfma 32x32x2", w);
    run<0, 8, 2>("f32 mfma 32x32x2", w);
    run<0, 8, 4>("f32 mfma 32x32x2", w);
    run<0, 8, 8>("f32 mfma 32x32x2", w);
    run<1, 8, 0>("bf16 mfma 32x32x16", w);
    run<1, 8, 2>("bf16 mfma 32x32x16", w);
    run<1, 8, 4>("bf16 mfma 32x32x16", w);
    run<1, 8, 8>("bf16 mfma 32x32x16", w);
    run<1, 8, 16>("bf16 mfma 32x32x16", w);
  }
  return 0;
}
