// Cost model check for a Winograd F(2x2,3x3) chunk loop on the bf16 matrix cores with fp32 operands split into three bf16 pieces
// (x = h + m + l exactly; products hh, hm, mh, mm, hl, lh kept: error <= 3 * 2^-24 per product).  Synthetic: LDS holds arbitrary
// data, no global traffic; per 16-channel chunk and wave the real work of that design -- 32 patch reads, the B^T d B transform,
// the three-way split + packing of 64 V values, 24 panel reads, 48 v_mfma_f32_32x32x16_bf16 -- between two barriers.
//   hipcc --offload-arch=gfx950 -O3 wino_split_loop.hip -o wsl && ./wsl
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split8(const v4f lo, const v4f hi, v4u& H, v4u& M, v4u& L) {
  const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
  unsigned h[8], m[8], l[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const unsigned xb = __builtin_bit_cast(unsigned, x[i]);
    const unsigned hb = xb & 0xFFFF0000u;
    const float r1 = x[i] - __builtin_bit_cast(float, hb);
    const unsigned mb = __builtin_bit_cast(unsigned, r1) & 0xFFFF0000u;
    const float r2 = r1 - __builtin_bit_cast(float, mb);
    h[i] = hb; m[i] = mb; l[i] = __builtin_bit_cast(unsigned, r2);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    H[i] = __builtin_amdgcn_perm(h[2 * i + 1], h[2 * i], 0x07060302u);
    M[i] = __builtin_amdgcn_perm(m[2 * i + 1], m[2 * i], 0x07060302u);
    L[i] = __builtin_amdgcn_perm(l[2 * i + 1], l[2 * i], 0x07060302u);
  }
}

template <int MODE>     // 0: split-bf16 (16-channel chunk)   1: fp32 MFMA reference loop (two 8-channel chunks of the shipped kernel's shape)
__global__ void __launch_bounds__(256, 2) k(float* out, int chunks) {
  extern __shared__ float4 lds4[];
  const v4f* ldsA = reinterpret_cast<const v4f*>(lds4);
  const v4f* ldsB = ldsA + 2048;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int xh = wave & 1;
  for (int i = tid; i < 2048 + 3072; i += 256) lds4[i] = make_float4(i * 1e-3f, 1.f + i * 1e-4f, 0.5f, 0.25f);
  __syncthreads();
  v16f acc[2][4];
  for (int x = 0; x < 2; ++x) for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) acc[x][n][i] = 0.f;
  const int pbase = (wave >> 1) * 700 + l31 * 10 + half * 2;
  for (int c = 0; c < chunks; ++c) {
    __syncthreads();
    if (tid < 64) lds4[5000 + tid] = make_float4(c, 0, 0, 0);      // a few commit-like writes
    __syncthreads();
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const int xi = 2 * xh + x;
      const int a1 = xi == 0 ? 0 : (xi == 2 ? 2 : 1), a2 = xi == 3 ? 3 : (xi == 2 ? 1 : 2);
      const float sg = xi == 1 ? 1.f : -1.f;
      if (MODE == 0) {
        v4f R[4][2];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            const v4f d1 = ldsA[pbase + (a1 * 34 + b) * 5 + q], d2 = ldsA[pbase + (a2 * 34 + b) * 5 + q];
            R[b][q] = d1 + sg * d2;
          }
        v4f V[4][2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          V[0][q] = R[0][q] - R[2][q]; V[1][q] = R[1][q] + R[2][q]; V[2][q] = R[2][q] - R[1][q]; V[3][q] = R[1][q] - R[3][q];
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          v4u H, M, L;
          split8(V[n][0], V[n][1], H, M, L);
          const v4f bh = ldsB[((xi * 4 + n) * 6 + 0 + half) * 32 + l31], bm = ldsB[((xi * 4 + n) * 6 + 2 + half) * 32 + l31],
                    bl = ldsB[((xi * 4 + n) * 6 + 4 + half) * 32 + l31];
          const v8bf ah = __builtin_bit_cast(v8bf, H), am = __builtin_bit_cast(v8bf, M), al = __builtin_bit_cast(v8bf, L);
          const v8bf Bh = __builtin_bit_cast(v8bf, bh), Bm = __builtin_bit_cast(v8bf, bm), Bl = __builtin_bit_cast(v8bf, bl);
          acc[x][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, Bh, acc[x][n], 0, 0, 0);
          acc[x][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, Bl, acc[x][n], 0, 0, 0);
          acc[x][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, Bm, acc[x][n], 0, 0, 0);
          acc[x][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, Bh, acc[x][n], 0, 0, 0);
          acc[x][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, Bm, acc[x][n], 0, 0, 0);
          acc[x][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, Bh, acc[x][n], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {            // two 8-channel chunks' worth (barriers of the second not modelled)
          v4f R[4];
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const v4f d1 = ldsA[pbase + (a1 * 34 + b) * 5 + q], d2 = ldsA[pbase + (a2 * 34 + b) * 5 + q];
            R[b] = d1 + sg * d2;
          }
          v4f V[4];
          V[0] = R[0] - R[2]; V[1] = R[1] + R[2]; V[2] = R[2] - R[1]; V[3] = R[1] - R[3];
#pragma unroll
          for (int n = 0; n < 4; ++n) {
            const v4f u = ldsB[((xi * 4 + n) * 6 + q * 2 + half) * 32 + l31];
            acc[x][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[n].x, u.x, acc[x][n], 0, 0, 0);
            acc[x][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[n].y, u.y, acc[x][n], 0, 0, 0);
            acc[x][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[n].z, u.z, acc[x][n], 0, 0, 0);
            acc[x][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[n].w, u.w, acc[x][n], 0, 0, 0);
          }
        }
      }
    }
  }
  float r = 0.f;
  for (int x = 0; x < 2; ++x) for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) r += acc[x][n][i];
  out[blockIdx.x * 256 + tid] = r;
}

template <int MODE>
void run(const char* name) {
  float* out; (void)hipMalloc(&out, 512 * 256 * 4);
  const int chunks = 2000;
  const size_t lds = (2048 + 3072 + 128) * 16;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  k<MODE><<<512, 256, lds>>>(out, 10);
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  k<MODE><<<512, 256, lds>>>(out, chunks);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %7.2f us per 16 channels (two workgroups per CU)\n", name, ms * 1e3 / chunks);
  (void)hipFree(out);
}

int main() {
  run<1>("fp32 MFMA Winograd loop (2 x 8 channels)");
  run<0>("bf16 x3 split Winograd loop (16 channels)");
  return 0;
}
