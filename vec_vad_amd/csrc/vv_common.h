// Shared device helpers for the gfx950 kernels of libvecvad_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/vecvad_hip.h"

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

#define VV_WG 256

// compile-time counted loop: f(std::integral_constant<int, I>) for I in [B, E)
template <int B, int E, typename F>
__device__ __forceinline__ void vv_static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    vv_static_for<B + 1, E>(f);
  }
}

// hipGetLastError() is sticky per thread: clear anything an earlier, unrelated runtime call left behind so that
// VV_CHECK_LAUNCH reports only this launch.
#define VV_LAUNCH(...)                  \
  do {                                  \
    (void)hipGetLastError();            \
    hipLaunchKernelGGL(__VA_ARGS__);    \
  } while (0)

// the hipError_t travels in the return value (bits 8..): no last-error variable anywhere in the library
#define VV_HIP_STATUS(e) (VV_ERR_LAUNCH | ((int)(e) << 8))
#define VV_CHECK_LAUNCH()                                   \
  do {                                                      \
    hipError_t e__ = hipGetLastError();                     \
    if (e__ != hipSuccess) return VV_HIP_STATUS(e__);       \
  } while (0)

// compute units of the current device (vv_elem.hip, cached): the persistent kernels size their grids with it (one / two / VV_RING16_OCC
// workgroups per CU), the host policies (vec_vad_amd/bank.py) their k-splits and Winograd F(4x4) routing -- 256 on an MI355X in SPX mode,
// fewer on a partitioned device (CPX) or another part
extern "C" int vv_num_cus(void);

// vv_conv_bf16.hip: the GEMM-shaped 3x3 kernel of the all-bf16 launches (forward: VV_CONV_ALLSRC_BF16, data gradient: also
// VV_CONV_SRC_BF16; output bf16).  vv_conv_mfma routes to it, vv_conv_ntiles2 reports its (256-pixel) tiles, under ONE predicate.
int vv_conv_gemm16(const vv_conv_params* p, hipStream_t st);
inline bool vv_gemm16_flags(int kind, int flags) {
  return kind == VV_CONV3 && (flags & VV_CONV_BF16) && (flags & VV_CONV_OUT_BF16) && (flags & (VV_CONV_SRC_BF16 | VV_CONV_ALLSRC_BF16)) &&
         !(flags & VV_CONV_NO_GEMM16);
}

// vv_conv_ring16.hip (round 5): the all-bf16 3x3 launches of the 32x32 level with at most 32 input channels on persistent workgroups
// with an LDS-DMA halo prefetch; vv_conv_mfma routes to it (bit-identical results), VV_CONV_NO_RING keeps the launch where it was.
int vv_conv_ring16(const vv_conv_params* p, hipStream_t st);
bool vv_conv_ring16_ok(const vv_conv_params* p);

// DPP row_ror:N -- the value of the lane N places further round this lane's row of 16
template <int N>
__device__ __forceinline__ float vv_dpp_ror(const float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xF, 0xF, false));
}

// XCD-aware work-item remap: the dispatcher places block b on XCD b%8 (observed, MI355X_MICROARCH.md);
// give every XCD one contiguous chunk of the work list so that workgroups sharing a UNet's weight panel
// share an L2.  nper = ceil(total/8); grid = 8*nper; caller drops w >= total.
__device__ __forceinline__ int vv_xcd_remap(int bid, int nper) { return (bid & 7) * nper + (bid >> 3); }

__device__ __forceinline__ float4 vv_relu4(float4 v) {
  v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
  return v;
}
__device__ __forceinline__ float4 vv_act4(float4 v, float4 a, float4 b) {
  v.x = fmaxf(fmaf(a.x, v.x, b.x), 0.f);
  v.y = fmaxf(fmaf(a.y, v.y, b.y), 0.f);
  v.z = fmaxf(fmaf(a.z, v.z, b.z), 0.f);
  v.w = fmaxf(fmaf(a.w, v.w, b.w), 0.f);
  return v;
}
__device__ __forceinline__ float4 vv_max4(float4 p, float4 q) {
  p.x = fmaxf(p.x, q.x); p.y = fmaxf(p.y, q.y); p.z = fmaxf(p.z, q.z); p.w = fmaxf(p.w, q.w);
  return p;
}

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
// (x.lo + y.hi, x.lo - y.hi) in ONE v_pk_add_f32 (operand-half selection + sign on the high result); the compiler needs three
// instructions for the same expression.  The result must not feed an MFMA within the next two issue slots (the compiler does
// not see the VALU -> MFMA operand hazard through inline asm): every call site has several instructions in between.
__device__ __forceinline__ v2f vv_pk_lo_pm_hi(const v2f x, const v2f y) {
  v2f r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y));
  return r;
}

// Butterfly sum over aligned groups of LPP (8 | 16) lanes, result in every lane -- the same pairings as d += __shfl_xor(d, 1 | 2 |
// 4 | 8) (so the same bits), but as DPP operand modifiers of three / four VALU adds instead of ds_bpermute round trips through the
// LDS crossbar: xor 1 / xor 2 = quad permutes; once a quad is uniform, row_half_mirror (l <-> 7 - l) fetches the other quad of the
// 8-lane group and row_mirror (l <-> 15 - l) the other half of the 16-lane row.
template <int LPP>
__device__ __forceinline__ float vv_group_sum(float d) {
  static_assert(LPP == 8 || LPP == 16, "8 or 16 lanes per group");
  auto dpp = [](const float v, const auto CTRL) -> float {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(CTRL)::value, 0xF, 0xF, false));
  };
  d += dpp(d, std::integral_constant<int, 0xB1>{});        // quad_perm [1,0,3,2]
  d += dpp(d, std::integral_constant<int, 0x4E>{});        // quad_perm [2,3,0,1]
  d += dpp(d, std::integral_constant<int, 0x141>{});       // row_half_mirror
  if constexpr (LPP == 16) d += dpp(d, std::integral_constant<int, 0x140>{});      // row_mirror
  return d;
}

// 4 floats -> 4 bf16 (round to nearest even), packed in channel order
__device__ __forceinline__ uint2 vv_pack_bf16x4(float4 v) {
  const v2bf lo = __builtin_convertvector((v2f){v.x, v.y}, v2bf);
  const v2bf hi = __builtin_convertvector((v2f){v.z, v.w}, v2bf);
  uint2 o;
  o.x = __builtin_bit_cast(unsigned, lo);
  o.y = __builtin_bit_cast(unsigned, hi);
  return o;
}

__device__ __forceinline__ float4 vv_unpack_bf16x4(uint2 u) {
  return make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xFFFF0000u),
                     __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xFFFF0000u));
}

// 8 bf16 (16 bytes) <-> 8 floats
struct vv_f8 { float v[8]; };
__device__ __forceinline__ vv_f8 vv_unpack_bf16x8(const uint4 u) {
  vv_f8 r;
  r.v[0] = __builtin_bit_cast(float, u.x << 16); r.v[1] = __builtin_bit_cast(float, u.x & 0xFFFF0000u);
  r.v[2] = __builtin_bit_cast(float, u.y << 16); r.v[3] = __builtin_bit_cast(float, u.y & 0xFFFF0000u);
  r.v[4] = __builtin_bit_cast(float, u.z << 16); r.v[5] = __builtin_bit_cast(float, u.z & 0xFFFF0000u);
  r.v[6] = __builtin_bit_cast(float, u.w << 16); r.v[7] = __builtin_bit_cast(float, u.w & 0xFFFF0000u);
  return r;
}
__device__ __forceinline__ uint4 vv_pack_bf16x8(const vv_f8& f) {
  const uint2 lo = vv_pack_bf16x4(make_float4(f.v[0], f.v[1], f.v[2], f.v[3]));
  const uint2 hi = vv_pack_bf16x4(make_float4(f.v[4], f.v[5], f.v[6], f.v[7]));
  return make_uint4(lo.x, lo.y, hi.x, hi.y);
}

// Resolved (per group) description of how a convolution reads its input.
struct VVSrc {
  const float* p0; int cs0, co0;
  const float* a; const float* b;
  const float* p1; int cs1, co1;
  const int* chmap;
  int csplit;
  int mode;
  int SH, SW;   // resolution of the conv-input coordinate space (for VV_IN_POOL the tensor itself is 2SH x 2SW)
  int B;
};

template <typename P>
__device__ __forceinline__ VVSrc vv_make_src(const P& p, int g, int SH, int SW) {
  VVSrc s;
  s.p0 = p.src0.ptr + (int64_t)g * p.src0.gstride; s.cs0 = p.src0.cstride; s.co0 = p.src0.coff;
  s.a = p.a ? p.a + (int64_t)g * p.ab_gstride : nullptr;
  s.b = p.b ? p.b + (int64_t)g * p.ab_gstride : nullptr;
  s.p1 = p.src1.ptr ? p.src1.ptr + (int64_t)g * p.src1.gstride : nullptr; s.cs1 = p.src1.cstride; s.co1 = p.src1.coff;
  s.chmap = p.chmap ? p.chmap + (int64_t)g * p.CinP : nullptr;
  s.csplit = p.csplit; s.mode = p.in_mode; s.SH = SH; s.SW = SW; s.B = p.B;
  return s;
}

// 4 consecutive input channels [c, c+4) of conv-input pixel (img, y, x); caller guarantees in-bounds coordinates.
__device__ __forceinline__ float4 vv_fetch4(const VVSrc& s, int img, int y, int x, int c) {
  switch (s.mode) {
    case VV_IN_PLAIN: {
      const float* q = s.p0 + ((int64_t)(img * s.SH + y) * s.SW + x) * s.cs0 + s.co0 + c;
      return *reinterpret_cast<const float4*>(q);
    }
    case VV_IN_ACT: {
      const float* q = s.p0 + ((int64_t)(img * s.SH + y) * s.SW + x) * s.cs0 + s.co0 + c;
      return vv_act4(*reinterpret_cast<const float4*>(q), *reinterpret_cast<const float4*>(s.a + c),
                     *reinterpret_cast<const float4*>(s.b + c));
    }
    case VV_IN_POOL: {
      const int W2 = 2 * s.SW;
      const float* q = s.p0 + ((int64_t)(img * 2 * s.SH + 2 * y) * W2 + 2 * x) * s.cs0 + s.co0 + c;
      const float4 a = *reinterpret_cast<const float4*>(s.a + c);
      const float4 b = *reinterpret_cast<const float4*>(s.b + c);
      float4 v00 = vv_act4(*reinterpret_cast<const float4*>(q), a, b);
      float4 v01 = vv_act4(*reinterpret_cast<const float4*>(q + s.cs0), a, b);
      float4 v10 = vv_act4(*reinterpret_cast<const float4*>(q + (int64_t)W2 * s.cs0), a, b);
      float4 v11 = vv_act4(*reinterpret_cast<const float4*>(q + (int64_t)(W2 + 1) * s.cs0), a, b);
      return vv_max4(vv_max4(v00, v01), vv_max4(v10, v11));
    }
    case VV_IN_CAT: {
      if (c < s.csplit) {
        const float* q = s.p0 + ((int64_t)(img * s.SH + y) * s.SW + x) * s.cs0 + s.co0 + c;
        return vv_act4(*reinterpret_cast<const float4*>(q), *reinterpret_cast<const float4*>(s.a + c),
                       *reinterpret_cast<const float4*>(s.b + c));
      }
      const float* q = s.p1 + ((int64_t)(img * s.SH + y) * s.SW + x) * s.cs1 + s.co1 + (c - s.csplit);
      return *reinterpret_cast<const float4*>(q);
    }
    default: {  // VV_IN_CUBE
      const float* q = s.p0 + ((int64_t)(img * s.SH + y) * s.SW + x) * s.cs0 + s.co0;
      float4 v;
      int m0 = s.chmap[c], m1 = s.chmap[c + 1], m2 = s.chmap[c + 2], m3 = s.chmap[c + 3];
      v.x = m0 >= 0 ? q[m0] : 0.f;
      v.y = m1 >= 0 ? q[m1] : 0.f;
      v.z = m2 >= 0 ? q[m2] : 0.f;
      v.w = m3 >= 0 ? q[m3] : 0.f;
      return v;
    }
  }
}

// Cooperative load of a [NI][HH][HW][NCH] tile (NCH channels starting at c0, LDS pixel stride S floats) whose
// top-left conv-input coordinate is (y0, x0) for images img0..img0+NI-1.  Out-of-image pixels are zero
// (the convolution's zero padding applies to the post-activation tensor).
template <int NI, int HH, int HW, int S, int NCH>
__device__ __forceinline__ void vv_stage_tile(float* lds, const VVSrc& s, int img0, int y0, int x0, int c0, int tid,
                                              int cmax = 1 << 30) {
  constexpr int Q = NCH / 4;
  constexpr int NITEMS = NI * HH * HW * Q;
  constexpr int BATCH = 6;                       // loads in flight per thread before the first LDS write
  constexpr int NIT = (NITEMS + VV_WG - 1) / VV_WG;
#pragma unroll 1
  for (int k0 = 0; k0 < NIT; k0 += BATCH) {
    float4 v[BATCH];
#pragma unroll
    for (int k = 0; k < BATCH; ++k) {
      const int it = tid + (k0 + k) * VV_WG;
      const int q = it % Q;
      const int hp = it / Q;
      const int hx = hp % HW;
      const int t = hp / HW;
      const int hy = t % HH;
      const int im = t / HH;
      const int img = img0 + im, y = y0 + hy, x = x0 + hx;
      v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (it < NITEMS && img < s.B && (unsigned)y < (unsigned)s.SH && (unsigned)x < (unsigned)s.SW && c0 + q * 4 < cmax)
        v[k] = vv_fetch4(s, img, y, x, c0 + q * 4);
    }
#pragma unroll
    for (int k = 0; k < BATCH; ++k) {
      const int it = tid + (k0 + k) * VV_WG;
      if (it < NITEMS) *reinterpret_cast<float4*>(lds + (it / Q) * S + (it % Q) * 4) = v[k];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// VVStagerB: register-staged, software-pipelined tile loader.  prefetch() issues every load of a tile into registers
// (they stay in flight under the caller's MFMA phase), commit() applies the deferred BatchNorm+ReLU and writes LDS.
// (a) per-item pixel offsets / halo rows are computed ONCE per workgroup and (b) loads are raw buffer loads: 32-bit voffset + SGPR descriptor, and out-of-image items get an offset beyond num_records so
// the hardware bounds check returns 0.0f (the convolution's zero padding) without a branch.  Per item and tile the
// address math is ~5 VALU instructions instead of ~30 (the staging code was issue-bound, not latency-bound).
// VV_IN_POOL / VV_IN_CUBE items (4-tap max-pool, channel gather) keep the generic immediate path.
template <int NI, int HH, int HW, int S, int NCH, int NTH = VV_WG>      // NTH: threads of the workgroup
struct VVStagerB {
  static constexpr int Q = NCH / 4;
  static constexpr int NITEMS = NI * HH * HW * Q;
  static constexpr int NIT = (NITEMS + NTH - 1) / NTH;
  static_assert(NTH % Q == 0 && NIT <= 32, "stager geometry");
  static constexpr unsigned OOB = 0x80000000u;
  float4 r[NIT];
  int pix[NIT];        // (im*SH + hy)*SW + hx relative to the tile origin, or INT_MIN/2 when the column is never valid
  short hy[NIT], im[NIT];
  float4 sa, sb;
  unsigned valid;
  bool act;

  __device__ __forceinline__ void init(const VVSrc& s, int x0, int tid) {
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int it = tid + k * NTH;
      const int hp = it / Q;
      const int hx = hp % HW;
      const int t = hp / HW;
      hy[k] = (short)(t % HH);
      im[k] = (short)(t / HH);
      const int x = x0 + hx;
      const bool ok = (NITEMS % NTH == 0 || it < NITEMS) && (unsigned)x < (unsigned)s.SW;
      pix[k] = ok ? (im[k] * s.SH + hy[k]) * s.SW + hx : -(1 << 30);
    }
  }

  // x0 must be the value given to init() (all tiles of a workgroup share their column origin)
  __device__ __forceinline__ void prefetch(const VVSrc& s, int img0, int y0, int x0, int c0, int tid, int cmax = 1 << 30) {
    const int q = tid % Q;
    const int c = c0 + q * 4;
    valid = 0;
    act = (s.mode == VV_IN_ACT) || (s.mode == VV_IN_CAT && c < s.csplit);
    if (act && c < cmax) {
      sa = *reinterpret_cast<const float4*>(s.a + c);
      sb = *reinterpret_cast<const float4*>(s.b + c);
    }
    if (s.mode == VV_IN_POOL || s.mode == VV_IN_CUBE) {
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        const int img = img0 + im[k], y = y0 + hy[k], x = x0 + ((tid + k * NTH) / Q) % HW;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pix[k] >= 0 && img < s.B && (unsigned)y < (unsigned)s.SH && c < cmax) v = vv_fetch4(s, img, y, x, c);
        r[k] = v;
      }
      return;
    }
    // All threads of a workgroup read the same source in one call (chunks / ci-tiles never straddle the concat split);
    // make that provable (readfirstlane) so the buffer descriptor lives in SGPRs -- a lane-varying descriptor would be
    // executed as a serialising waterfall loop per load.
    const bool second = __builtin_amdgcn_readfirstlane((int)((s.mode == VV_IN_CAT) && c0 >= s.csplit)) != 0;
    const float* base = second ? s.p1 + s.co1 - s.csplit : s.p0 + s.co0;
    const int cs = second ? s.cs1 : s.cs0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7FFFFFFF, 0x00020000);
    const int tile = (img0 * s.SH + y0) * s.SW + x0;      // may be negative (top halo row of image 0)
    const bool cok = c < cmax;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int y = y0 + hy[k];
      const bool ok = cok && (unsigned)y < (unsigned)s.SH && (img0 + im[k]) < s.B && pix[k] >= 0;
      const unsigned off = ok ? (unsigned)((tile + pix[k]) * cs + c) * 4u : OOB;
      valid |= ok ? (1u << k) : 0u;
      const v4f v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
      r[k] = make_float4(v.x, v.y, v.z, v.w);
    }
  }

  // ---- piecewise interface: the same loads / LDS writes, one item per call with a compile-time item index, for
  // callers that spread the staging of the NEXT tile over the MFMA loop of the current one (begin() once per tile, then
  // load_piece<0..NIT-1>, later commit_piece<0..NIT-1>).  VV_IN_PLAIN / VV_IN_ACT / VV_IN_CAT sources only (the engine
  // materialises pooled and frame-erased inputs, so these are the modes the kernels see).  live = false turns every item into an out-of-range load (zeros,
  // no memory traffic): the tail of the pipeline stays branch-free.
  const float* base_;
  int cs_, c_, tile_, img0_, y0_;
  bool cok_;
  __device__ __forceinline__ void begin(const VVSrc& s, int img0, int y0, int x0, int c0, int tid, int cmax, bool live) {
    const int q = tid % Q;
    c_ = c0 + q * 4;
    valid = 0;
    act = (s.mode == VV_IN_ACT) || (s.mode == VV_IN_CAT && c_ < s.csplit);
    if (act && c_ < cmax) {
      sa = *reinterpret_cast<const float4*>(s.a + c_);
      sb = *reinterpret_cast<const float4*>(s.b + c_);
    }
    const bool second = __builtin_amdgcn_readfirstlane((int)((s.mode == VV_IN_CAT) && c0 >= s.csplit)) != 0;
    base_ = second ? s.p1 + s.co1 - s.csplit : s.p0 + s.co0;
    cs_ = second ? s.cs1 : s.cs0;
    tile_ = (img0 * s.SH + y0) * s.SW + x0;
    img0_ = img0; y0_ = y0;
    cok_ = live && c_ < cmax;
  }
  template <int K>
  __device__ __forceinline__ void load_piece(const VVSrc& s, int x0, int tid) {
    const int y = y0_ + hy[K];
    const bool ok = cok_ && (unsigned)y < (unsigned)s.SH && (img0_ + im[K]) < s.B && pix[K] >= 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base_), 0, 0x7FFFFFFF, 0x00020000);
    const unsigned off = ok ? (unsigned)((tile_ + pix[K]) * cs_ + c_) * 4u : OOB;
    valid |= ok ? (1u << K) : 0u;
    const v4f v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
    r[K] = make_float4(v.x, v.y, v.z, v.w);
  }
  template <int K>
  __device__ __forceinline__ void commit_piece(float* lds, int tid) const {
    const int it = tid + K * NTH;
    if (NITEMS % NTH == 0 || it < NITEMS) {
      float4 v = r[K];
      if (act && ((valid >> K) & 1u)) v = vv_act4(v, sa, sb);
      *reinterpret_cast<float4*>(lds + (it / Q) * S + (tid % Q) * 4) = v;
    }
  }

  // bf16 SOURCE (plain tensors only: a dy that BatchNorm backward stored as bf16): the same items are 8 B = 4 channels
  // each, loaded into r[k].x / r[k].y as bit patterns and copied to LDS as they are.  Compile-time choice of the caller.
  __device__ __forceinline__ void prefetch16(const VVSrc& s, int img0, int y0, int x0, int c0, int tid, int cmax = 1 << 30) {
    const int q = tid % Q;
    const int c = c0 + q * 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s.p0 + (s.co0 >> 1)), 0, 0x7FFFFFFF, 0x00020000);
    const int tile = (img0 * s.SH + y0) * s.SW + x0;
    const bool cok = c < cmax;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int y = y0 + hy[k];
      const bool ok = cok && (unsigned)y < (unsigned)s.SH && (img0 + im[k]) < s.B && pix[k] >= 0;
      const unsigned off = ok ? (unsigned)((tile + pix[k]) * s.cs0 + c) * 2u : OOB;
      const v2f v = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0));
      r[k] = make_float4(v.x, v.y, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void commit_raw16(float* lds, int tid) const {
    const int q = tid % Q;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int it = tid + k * NTH;
      if (NITEMS % NTH == 0 || it < NITEMS)
        *reinterpret_cast<uint2*>(lds + (it / Q) * S + q * 2) = make_uint2(__builtin_bit_cast(unsigned, r[k].x), __builtin_bit_cast(unsigned, r[k].y));
    }
  }

  // ALL sources bf16 (pre-BN conv outputs, pooled / frame-erased inputs, transposed-conv outputs stored as bf16; modes PLAIN / ACT /
  // CAT): 8-byte items; the deferred BatchNorm+ReLU is applied on unpacked values at commit, plain items are copied.
  __device__ __forceinline__ void prefetch16x(const VVSrc& s, int img0, int y0, int x0, int c0, int tid, int cmax = 1 << 30) {
    const int q = tid % Q;
    const int c = c0 + q * 4;
    valid = 0;
    act = (s.mode == VV_IN_ACT) || (s.mode == VV_IN_CAT && c < s.csplit);
    if (act && c < cmax) {
      sa = *reinterpret_cast<const float4*>(s.a + c);
      sb = *reinterpret_cast<const float4*>(s.b + c);
    }
    const bool second = __builtin_amdgcn_readfirstlane((int)((s.mode == VV_IN_CAT) && c0 >= s.csplit)) != 0;
    const float* base = second ? s.p1 : s.p0;
    const int co = second ? s.co1 - s.csplit : s.co0;       // element offset of channel 0 of this call's source
    const int cs = second ? s.cs1 : s.cs0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7FFFFFFF, 0x00020000);
    const int tile = (img0 * s.SH + y0) * s.SW + x0;
    const bool cok = c < cmax;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int y = y0 + hy[k];
      const bool ok = cok && (unsigned)y < (unsigned)s.SH && (img0 + im[k]) < s.B && pix[k] >= 0;
      const unsigned off = ok ? (unsigned)((tile + pix[k]) * cs + co + c) * 2u : OOB;
      valid |= ok ? (1u << k) : 0u;
      const v2f v = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0));
      r[k] = make_float4(v.x, v.y, 0.f, 0.f);
    }
  }
  __device__ __forceinline__ void commit16(float* lds, int tid) const {
    const int q = tid % Q;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int it = tid + k * NTH;
      if (NITEMS % NTH == 0 || it < NITEMS) {
        uint2 h = make_uint2(__builtin_bit_cast(unsigned, r[k].x), __builtin_bit_cast(unsigned, r[k].y));
        if (act && ((valid >> k) & 1u)) h = vv_pack_bf16x4(vv_act4(vv_unpack_bf16x4(h), sa, sb));
        *reinterpret_cast<uint2*>(lds + (it / Q) * S + q * 2) = h;
      }
    }
  }

  // The same with 16-byte items (8 channels): declare the loader with NCH = (channels per chunk) / 2, so that Q = channels / 8 items
  // per pixel; c0 and cmax are still in channels.
  float4 sa2, sb2;
  __device__ __forceinline__ void prefetch16w(const VVSrc& s, int img0, int y0, int x0, int c0, int tid, int cmax = 1 << 30) {
    const int q = tid % Q;
    const int c = c0 + q * 8;
    valid = 0;
    act = (s.mode == VV_IN_ACT) || (s.mode == VV_IN_CAT && c < s.csplit);
    if (act && c < cmax) {
      sa = *reinterpret_cast<const float4*>(s.a + c); sa2 = *reinterpret_cast<const float4*>(s.a + c + 4);
      sb = *reinterpret_cast<const float4*>(s.b + c); sb2 = *reinterpret_cast<const float4*>(s.b + c + 4);
    }
    const bool second = __builtin_amdgcn_readfirstlane((int)((s.mode == VV_IN_CAT) && c0 >= s.csplit)) != 0;
    const float* base = second ? s.p1 : s.p0;
    const int co = second ? s.co1 - s.csplit : s.co0;
    const int cs = second ? s.cs1 : s.cs0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7FFFFFFF, 0x00020000);
    const int tile = (img0 * s.SH + y0) * s.SW + x0;
    const bool cok = c < cmax;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int y = y0 + hy[k];
      const bool ok = cok && (unsigned)y < (unsigned)s.SH && (img0 + im[k]) < s.B && pix[k] >= 0;
      const unsigned off = ok ? (unsigned)((tile + pix[k]) * cs + co + c) * 2u : OOB;
      valid |= ok ? (1u << k) : 0u;
      const v4f v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
      r[k] = make_float4(v.x, v.y, v.z, v.w);
    }
  }
  __device__ __forceinline__ void commit16w(float* lds, int tid) const {
    const int q = tid % Q;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int it = tid + k * NTH;
      if (NITEMS % NTH == 0 || it < NITEMS) {
        uint4 h = make_uint4(__builtin_bit_cast(unsigned, r[k].x), __builtin_bit_cast(unsigned, r[k].y),
                             __builtin_bit_cast(unsigned, r[k].z), __builtin_bit_cast(unsigned, r[k].w));
        if (act && ((valid >> k) & 1u)) {
          const uint2 lo = vv_pack_bf16x4(vv_act4(vv_unpack_bf16x4(make_uint2(h.x, h.y)), sa, sb));
          const uint2 hi = vv_pack_bf16x4(vv_act4(vv_unpack_bf16x4(make_uint2(h.z, h.w)), sa2, sb2));
          h = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
        *reinterpret_cast<uint4*>(lds + (it / Q) * S + q * 4) = h;
      }
    }
  }

  // bf16 operand path: the same items, rounded to nearest-even bf16 (v_cvt_pk_bf16_f32) after the deferred BatchNorm+ReLU and
  // written as 8 B (4 channels) per item; S is still the pixel stride in floats (4 B units).
  __device__ __forceinline__ void commit_bf16(float* lds, int tid) const {
    const int q = tid % Q;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int it = tid + k * NTH;
      if (NITEMS % NTH == 0 || it < NITEMS) {
        float4 v = r[k];
        if (act && ((valid >> k) & 1u)) v = vv_act4(v, sa, sb);
        *reinterpret_cast<uint2*>(lds + (it / Q) * S + q * 2) = vv_pack_bf16x4(v);
      }
    }
  }

  __device__ __forceinline__ void commit(float* lds, int tid) const {
    const int q = tid % Q;
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int it = tid + k * NTH;
      if (NITEMS % NTH == 0 || it < NITEMS) {
        float4 v = r[k];
        if (act && ((valid >> k) & 1u)) v = vv_act4(v, sa, sb);
        *reinterpret_cast<float4*>(lds + (it / Q) * S + q * 4) = v;
      }
    }
  }
};
