// Probe of ds_read_b64_tr_b16 (gfx950): which element does lane l / slot j receive, given per-lane addresses?
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/tr16_probe.hip -o /tmp/tr16_probe && /tmp/tr16_probe
// LDS holds lds[i] = i (16-bit).  Case A: lane i points at 8-byte chunk i (contiguous).  Case B: lane 4j+q of a 16-lane group
// points at [pixel j][channel quad q] of a pixel-major tile with 64 B per pixel, group g -> channel half g&1, pixels 8*(g>>1)+j
// (the addressing of vv_wgrad_bf16.hip).  Expected for B: lane l, slot s = element index (8*(l>>5) + s)*32 + (l&31).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void probe(short* out, int mode) {
  __shared__ short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  int byteoff;
  if (mode == 0) byteoff = l * 8;
  else {
    const int g = l >> 4, j = (l >> 2) & 3, q = l & 3;
    byteoff = (8 * (g >> 1) + j) * 64 + (g & 1) * 32 + q * 8;
  }
  const v4s v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)((__attribute__((address_space(3))) char*)lds + byteoff));
  for (int s = 0; s < 4; ++s) out[l * 4 + s] = v[s];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  short h[256];
  for (int mode = 0; mode < 2; ++mode) {
    probe<<<1, 64>>>(d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int s = 0; s < 4; ++s) {
      const int exp = mode == 0 ? (l & 15) + s * 16 + (l >> 4) * 64 : (8 * (l >> 5) + s) * 32 + (l & 31);
      if (h[l * 4 + s] != exp) ++bad;
    }
    printf("mode %d mismatches vs model: %d\n", mode, bad);
    if (bad) for (int l = 0; l < 64; l += 5) printf("  lane %2d: %d %d %d %d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  return 0;
}
