// What does the steady-state loop of conv_gemm16_kernel (vv_conv_bf16.hip) cost, piece by piece?  Per "tap" a wave issues 8
// v_mfma_f32_32x32x16_bf16 on 8 independent accumulators (wave tile 4 x 2), MODE & 1: + 4 ds_read_b128 (A fragments, conflict-free
// 16 B per lane), MODE & 2: + 2 buffer_load_b128 (B fragments from an L2-resident panel), pinned between the MFMAs like the kernel.
//   hipcc --offload-arch=gfx950 -O3 gemm16_loop.hip -o g16loop && ./g16loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

template <int MODE, int WPS>
__global__ void __launch_bounds__(256, WPS) k(const float* __restrict__ panel, float* out, int taps, const float* __restrict__ init) {
  __shared__ float4 lds[2048];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 2048; i += 256) lds[i] = reinterpret_cast<const float4*>(init)[i];
  __syncthreads();
  v16f acc[8];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  const v4f* ldsA = reinterpret_cast<const v4f*>(lds);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(panel), 0, 0x7FFFFFFF, 0x00020000);
  v4f fa[2][4], fb[3][2];
  for (int m = 0; m < 4; ++m) fa[0][m] = ldsA[lane + 64 * m];
  for (int s = 0; s < 3; ++s) for (int n = 0; n < 2; ++n) fb[s][n] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (s * 2 + n) * 1024, 0);
  int soff = 0;
  for (int t0 = 0; t0 < taps; t0 += 6) {
#pragma unroll
    for (int tt = 0; tt < 6; ++tt) {
      const int ca = tt & 1, cb = tt % 3;
      int piece = 0;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          acc[m * 2 + n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, fa[ca][m]), __builtin_bit_cast(v8bf, fb[cb][n]), acc[m * 2 + n], 0, 0, 0);
          if (piece < 2) {
            if (MODE & 2) fb[(tt + 2) % 3][piece] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, soff + piece * 1024, 0);
          } else if (piece < 6) {
            if (MODE & 1) fa[ca ^ 1][piece - 2] = ldsA[lane + 64 * (piece - 2) + ((tt * 7) & 15) * 64];
          }
          ++piece;
          __builtin_amdgcn_sched_barrier(0);
        }
      soff = (soff + 2048) & 0xFFFFF;       // walk a 1 MB panel
    }
  }
  float r = 0.f;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) r += acc[i][j];
  out[blockIdx.x * 256 + tid] = r;
}

template <int MODE, int WPS>
void run(const char* name, const float* panel, float* out, const float* init) {
  const int taps = 6 * 2000, grid = 256 * WPS;
  k<MODE, WPS><<<grid, 256>>>(panel, out, 60, init);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<MODE, WPS><<<grid, 256>>>(panel, out, taps, init);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)grid * 4 * taps * 8 * 32768.0;
  printf("%-44s %d WG/CU: %8.3f ms  %7.1f TF/s  (%.3f of 2.5 PF)  %.1f cycles@2.4GHz per MFMA per SIMD\n", name, WPS, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500.0,
         ms * 1e-3 * 2.4e9 / ((double)taps * 8 * WPS));
}

int main() {
  float *panel, *out, *init;
  hipMalloc(&panel, 2 << 20); hipMalloc(&out, 256 * 4 * 256 * 4); hipMalloc(&init, 2048 * 16);
  float* h = (float*)malloc(2 << 20);
  unsigned short* hs = (unsigned short*)h;
  srand(1);
  for (int i = 0; i < (1 << 20); ++i) hs[i] = 0x3C00 + (rand() & 0x3FF) | ((rand() & 1) << 15);     // bf16 values around +-1
  hipMemcpy(panel, h, 2 << 20, hipMemcpyHostToDevice);
  hipMemcpy(init, h, 2048 * 16, hipMemcpyHostToDevice);
  run<0, 1>("MFMA only", panel, out, init);
  run<0, 2>("MFMA only", panel, out, init);
  run<1, 1>("MFMA + 4 ds_read_b128 / 8", panel, out, init);
  run<1, 2>("MFMA + 4 ds_read_b128 / 8", panel, out, init);
  run<2, 1>("MFMA + 2 buffer_load_b128 / 8", panel, out, init);
  run<2, 2>("MFMA + 2 buffer_load_b128 / 8", panel, out, init);
  run<3, 1>("MFMA + LDS + B loads (the kernel's loop)", panel, out, init);
  run<3, 2>("MFMA + LDS + B loads (the kernel's loop)", panel, out, init);
  return 0;
}
