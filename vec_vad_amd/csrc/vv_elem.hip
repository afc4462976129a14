// HBM-bound kernels of the UNet bank: weight packing, BatchNorm finalise / backward, 1x1 output conv + squared
// error (loss / per-cube score), fused Adam, cube adapter and layout converters.  All are grouped over the G
// independent UNets (blockIdx.y or .z = group), vectorised to 16 B per lane and use fixed-order reductions
// (wave shuffles + LDS + a second pass), never float atomics, so results are bitwise reproducible.
#include "vv_common.h"

namespace {

// ------------------------------------------------------------------------------------------------ pack
// One workgroup = one tile of 8 K values x 32 N values x 9 taps, staged through LDS: the source is read as contiguous runs (72
// floats per n when the filter tensor is n-major -- modes 0 / 3 --, 288 floats per k when it is k-major -- modes 1 / 2) and every
// tap leaves as contiguous 512-byte runs of the panel.  (Round 2 gathered single floats at a 36-byte stride straight from global
// memory: 547 MB of HBM reads for 86 MB of weights in the 10-UNet bank, 119 us per step.)
__global__ void __launch_bounds__(VV_WG)
pack_weights_kernel(const vv_pack_entry* __restrict__ table, const float* __restrict__ params,
                    const int64_t params_gstride, float* __restrict__ packed, const int64_t packed_gstride) {
  __shared__ float tile[8 * 32 * 9 + 8];
  const vv_pack_entry e = table[blockIdx.y];
  const int g = blockIdx.z;
  const float* src = params + (int64_t)g * params_gstride + e.src_off;
  float* dst = packed + (int64_t)g * packed_gstride + e.dst_off;
  const bool b16 = (e.mode & 4) != 0;        // bf16 panel [tap][KP/16][2][N][8] for the bf16-operand kernels (K step = 16 channels)
  const int m = e.mode & 3;
  const bool kmajor = m == 1 || m == 2;      // source index (k * N + n) * 9 + tap'  (else (n * K + k) * 9 + tap)
  const bool flip = m == 1;                  // data gradient: spatially flipped filter
  const int NB = e.N >> 5, KB = e.KP >> 3;
  const int tid = threadIdx.x;
  for (int blk = blockIdx.x; blk < KB * NB; blk += gridDim.x) {
    const int kb = blk / NB, nb = blk % NB;
    const int k0 = kb * 8, n0 = nb * 32;
    __syncthreads();                         // the previous tile has been written out
    // the tile is kept in source order (linear, conflict-free stores); the write phase gathers (tap, k, n) from it
    for (int i = tid; i < 8 * 32 * 9; i += VV_WG) {
      int64_t si;
      bool ok;
      if (kmajor) { const int kk = i / 288; si = ((int64_t)(k0 + kk) * e.N + n0) * 9 + (i % 288); ok = k0 + kk < e.K; }
      else { const int nn = i / 72, r = i % 72; si = ((int64_t)(n0 + nn) * e.K + k0) * 9 + r; ok = k0 + r / 9 < e.K; }
      tile[i] = ok ? src[si] : 0.f;
    }
    __syncthreads();
    auto at = [&](const int tap, const int kk, const int nn) -> float {
      return kmajor ? tile[kk * 288 + nn * 9 + (flip ? 8 - tap : tap)] : tile[nn * 72 + kk * 9 + tap];
    };
    if (b16) {
      // k = ks*16 + half*8 + e8: this tile is one (ks, half); per tap 32 n x 8 e8 bf16 = 512 contiguous bytes
      const int ks = kb >> 1, half = kb & 1;
      __bf16* d16 = reinterpret_cast<__bf16*>(dst);
      for (int i = tid; i < 9 * 256; i += VV_WG) {
        const int tap = i >> 8, r = i & 255, nn = r >> 3, e8 = r & 7;
        d16[((((int64_t)tap * (e.KP >> 4) + ks) * 2 + half) * e.N + n0 + nn) * 8 + e8] = (__bf16)at(tap, e8, nn);
      }
    } else {
      // k = kq*8 + half*4 + j: this tile is one kq, both halves; per (tap, half) 32 n x 4 j floats = 512 contiguous bytes
      for (int i = tid; i < 9 * 256; i += VV_WG) {
        const int tap = i >> 8, r = i & 255, half = r >> 7, nn = (r >> 2) & 31, j = r & 3;
        dst[((((int64_t)tap * KB + kb) * 2 + half) * e.N + n0 + nn) * 4 + j] = at(tap, half * 4 + j, nn);
      }
    }
  }
}

// Sum of the per-tile partials [n][2][C] of one channel over tiles part, part+NP, ...: four independent fp64 accumulators so
// that the (L2-resident, latency-bound) loads of consecutive tiles overlap.  Fixed order -> deterministic.  The callers run
// 32 channels x NP = 32 partial-sum lanes per block (1024 threads): these launches are pure latency, up to 1024 tiles deep.
constexpr int VV_NP = 32;
__device__ __forceinline__ void vv_sum_partials(const float* __restrict__ st, const int n, const int C, const int part,
                                                double& s1, double& s2) {
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0, b3 = 0.0;
  const int64_t step = (int64_t)2 * C;
  int t = part;
  for (; t + 3 * VV_NP < n; t += 4 * VV_NP) {
    const float* q = st + t * step;
    const float x0 = q[0], y0 = q[C], x1 = q[VV_NP * step], y1 = q[VV_NP * step + C];
    const float x2 = q[2 * VV_NP * step], y2 = q[2 * VV_NP * step + C], x3 = q[3 * VV_NP * step], y3 = q[3 * VV_NP * step + C];
    a0 += (double)x0; b0 += (double)y0;
    a1 += (double)x1; b1 += (double)y1;
    a2 += (double)x2; b2 += (double)y2;
    a3 += (double)x3; b3 += (double)y3;
  }
  for (; t < n; t += VV_NP) {
    a0 += (double)st[t * step];
    b0 += (double)st[t * step + C];
  }
  s1 = (a0 + a1) + (a2 + a3);
  s2 = (b0 + b1) + (b2 + b3);
}

// ------------------------------------------------------------------------------------------------ BN finalise
// grid (C/32, G); 1024 threads = 32 channels x 32 partial-sum lanes; fp64 accumulation of the fp32 tile partials.
__global__ void __launch_bounds__(32 * VV_NP)
bn_finalize_kernel(const int C, const int ntiles, const double count, const int train, const float momentum,
                   const float eps, const float* __restrict__ stats, const int64_t stats_gstride,
                   const float* __restrict__ gamma, const float* __restrict__ beta, const int64_t param_gstride,
                   float* __restrict__ rmean, float* __restrict__ rvar, const int64_t buf_gstride,
                   float* __restrict__ a, float* __restrict__ b, float* __restrict__ mean, float* __restrict__ invstd,
                   const int64_t ab_gstride) {
  __shared__ double sh[2][VV_NP][32];
  const int g = blockIdx.y;
  const int cl = threadIdx.x & 31, part = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  if (train) {
    double s1 = 0.0, s2 = 0.0;
    if (c < C) {
      const float* st = stats + (int64_t)g * stats_gstride + c;
      vv_sum_partials(st, ntiles, C, part, s1, s2);
    }
    sh[0][part][cl] = s1;
    sh[1][part][cl] = s2;
  }
  __syncthreads();
  if (part != 0 || c >= C) return;
  const float gm = gamma[(int64_t)g * param_gstride + c], bt = beta[(int64_t)g * param_gstride + c];
  float* rm = rmean + (int64_t)g * buf_gstride + c;
  float* rv = rvar + (int64_t)g * buf_gstride + c;
  double mu, var;
  if (train) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < VV_NP; ++k) { s1 += sh[0][k][cl]; s2 += sh[1][k][cl]; }
    mu = s1 / count;
    var = s2 / count - mu * mu;
    if (var < 0.0) var = 0.0;
    const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
    *rm = (float)((1.0 - (double)momentum) * (double)*rm + (double)momentum * mu);
    *rv = (float)((1.0 - (double)momentum) * (double)*rv + (double)momentum * unb);
  } else {
    mu = (double)*rm;
    var = (double)*rv;
  }
  const double is = 1.0 / sqrt(var + (double)eps);
  const int64_t o = (int64_t)g * ab_gstride + c;
  a[o] = (float)((double)gm * is);
  b[o] = (float)((double)bt - mu * (double)gm * is);
  mean[o] = (float)mu;
  invstd[o] = (float)is;
}

// ------------------------------------------------------------------------------------------------ BN backward
// dz = (dA [+ maxpool-routed dPool]) * [z > 0]; pass 0 = partial sums of dz and dz*xhat per block of bp pixels,
// pass 1 = dy.  bp = bn_bp(C) = 8192 / C (256 for the 32-channel layers ... 32 for 256 channels): a workgroup's 256 lanes cover
// 1024 / C pixels per iteration, so every thread runs 8 dependent load rounds whatever the width -- with 256-pixel blocks the
// 256-channel layers ran 64 rounds on a sixth of the CUs (34 us for 25 MB at B = 256, and the same 33 us at B = 32).
// PASS 0: only the per-block partial sums of dz and dz*xhat (dz is NOT written);
// PASS 1: recompute dz the same way and write dy = gamma*invstd*(dz - c1 - xhat*c2) -- one HBM pass less than
//         materialising dz first and rewriting it.
template <bool POOL, int PASS>
__global__ void __launch_bounds__(VV_WG)
bn_bwd_reduce_kernel(const vv_bnbwd_params p, const int nblk, const int bp, const float* __restrict__ gamma,
                     const int64_t param_gstride, const float* __restrict__ scratch) {
  __shared__ float sh[2][VV_WG * 4];
  const int g = blockIdx.y, blk = blockIdx.x;
  const int C = p.C, Q4 = C >> 2, PL = VV_WG / Q4;
  const int q = threadIdx.x % Q4, pl = threadIdx.x / Q4;
  const int c = q * 4;
  const int64_t M = (int64_t)p.B * p.H * p.W;
  const float* __restrict__ y = p.y + (int64_t)g * p.y_gstride;
  float* __restrict__ dz = p.dz + (int64_t)g * p.dz_gstride;
  const float* __restrict__ dA = p.dA.ptr + (int64_t)g * p.dA.gstride + p.dA.coff;
  const int dcs = p.dA.cstride;
  // VV_BNBWD_DA_BF16: dA / dpool hold bf16 elements (activation gradients stored by the bf16 kernels)
  const bool da16 = (p.flags & VV_BNBWD_DA_BF16) != 0;
  const unsigned short* __restrict__ dAh = reinterpret_cast<const unsigned short*>(p.dA.ptr + (int64_t)g * p.dA.gstride) + p.dA.coff;
  auto ldA = [&](const int64_t pix) -> float4 {
    if (da16) return vv_unpack_bf16x4(*reinterpret_cast<const uint2*>(dAh + pix * dcs + c));
    return *reinterpret_cast<const float4*>(dA + pix * dcs + c);
  };
  const int64_t abo = (int64_t)g * p.ab_gstride + c;
  const float4 a4 = *reinterpret_cast<const float4*>(p.a + abo), b4 = *reinterpret_cast<const float4*>(p.b + abo);
  const float4 m4 = *reinterpret_cast<const float4*>(p.mean + abo), i4 = *reinterpret_cast<const float4*>(p.invstd + abo);
  float4 s1 = make_float4(0, 0, 0, 0), s2 = make_float4(0, 0, 0, 0);
  float4 gk = make_float4(0, 0, 0, 0), c1 = gk, c2 = gk;
  if constexpr (PASS == 1) {
    const float4 gm = *reinterpret_cast<const float4*>(gamma + (int64_t)g * param_gstride + c);
    gk = make_float4(gm.x * i4.x, gm.y * i4.y, gm.z * i4.z, gm.w * i4.w);
    c1 = *reinterpret_cast<const float4*>(scratch + (int64_t)g * 2 * C + c);
    c2 = *reinterpret_cast<const float4*>(scratch + (int64_t)g * 2 * C + C + c);
  }

  const bool y16 = (p.flags & VV_BNBWD_Y_BF16) != 0;
  auto ldY = [&](const int64_t pix) -> float4 {
    if (y16) return vv_unpack_bf16x4(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(y) + pix * C + c));
    return *reinterpret_cast<const float4*>(y + pix * C + c);
  };
  auto one = [&](const int64_t pix, float4 d, const float4 yv) {
    float4 z;
    z.x = fmaf(a4.x, yv.x, b4.x); z.y = fmaf(a4.y, yv.y, b4.y); z.z = fmaf(a4.z, yv.z, b4.z); z.w = fmaf(a4.w, yv.w, b4.w);
    d.x = z.x > 0.f ? d.x : 0.f; d.y = z.y > 0.f ? d.y : 0.f; d.z = z.z > 0.f ? d.z : 0.f; d.w = z.w > 0.f ? d.w : 0.f;
    const float4 xh = make_float4((yv.x - m4.x) * i4.x, (yv.y - m4.y) * i4.y, (yv.z - m4.z) * i4.z, (yv.w - m4.w) * i4.w);
    if constexpr (PASS == 0) {
      s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
      s2.x = fmaf(d.x, xh.x, s2.x); s2.y = fmaf(d.y, xh.y, s2.y); s2.z = fmaf(d.z, xh.z, s2.z); s2.w = fmaf(d.w, xh.w, s2.w);
    } else {
      d.x = gk.x * (d.x - c1.x - xh.x * c2.x); d.y = gk.y * (d.y - c1.y - xh.y * c2.y);
      d.z = gk.z * (d.z - c1.z - xh.z * c2.z); d.w = gk.w * (d.w - c1.w - xh.w * c2.w);
      if (p.flags & VV_BNBWD_DZ_BF16) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(dz) + pix * C + c) = vv_pack_bf16x4(d);
      else *reinterpret_cast<float4*>(dz + pix * C + c) = d;
    }
  };

  const int64_t nchunks = (M + bp - 1) / bp;
  // persistent workgroups: nblk (<= VV_BN_MAXBLK per UNet) of them walk the chunks blk, blk + nblk, ... -- the per-channel
  // constants are loaded once per workgroup instead of once per 256 pixels, and neighbouring workgroups stream neighbouring memory
  for (int64_t chunk = blk; chunk < nchunks; chunk += nblk) {
  if constexpr (!POOL) {
    // (four pixels' loads in flight per thread: measured -15 % on the apply pass in round 2 and again in round 3 -- not kept)
    for (int i = pl; i < bp; i += PL) {
      const int64_t pix = chunk * bp + i;
      if (pix < M) one(pix, ldA(pix), ldY(pix));
    }
  } else {
    // unit of work = one 2x2 pooling window (first maximum wins ties, like at::max_pool2d)
    const int H2 = p.H >> 1, W2 = p.W >> 1;
    const int64_t MW = M >> 2;
    const float* __restrict__ dP = p.dpool + (int64_t)g * p.dpool_gstride;
    for (int i = pl; i < (bp >> 2); i += PL) {
      const int64_t wi = chunk * (bp >> 2) + i;
      if (wi >= MW) continue;
      const int wx = (int)(wi % W2);
      const int64_t t = wi / W2;
      const int wy = (int)(t % H2);
      const int64_t img = t / H2;
      const int64_t p00 = (img * p.H + 2 * wy) * p.W + 2 * wx;
      const int64_t px[4] = {p00, p00 + 1, p00 + p.W, p00 + p.W + 1};
      float4 zz[4], yq[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float4 yv = yq[k] = ldY(px[k]);
        zz[k].x = fmaxf(fmaf(a4.x, yv.x, b4.x), 0.f); zz[k].y = fmaxf(fmaf(a4.y, yv.y, b4.y), 0.f);
        zz[k].z = fmaxf(fmaf(a4.z, yv.z, b4.z), 0.f); zz[k].w = fmaxf(fmaf(a4.w, yv.w, b4.w), 0.f);
      }
      int ix = 0, iy = 0, iz = 0, iw = 0;
      float bx = zz[0].x, by = zz[0].y, bz = zz[0].z, bw = zz[0].w;
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        if (zz[k].x > bx) { bx = zz[k].x; ix = k; }
        if (zz[k].y > by) { by = zz[k].y; iy = k; }
        if (zz[k].z > bz) { bz = zz[k].z; iz = k; }
        if (zz[k].w > bw) { bw = zz[k].w; iw = k; }
      }
      const float4 dp = da16 ? vv_unpack_bf16x4(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(dP) + wi * C + c))
                             : *reinterpret_cast<const float4*>(dP + wi * C + c);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float4 d = ldA(px[k]);
        d.x += ix == k ? dp.x : 0.f; d.y += iy == k ? dp.y : 0.f; d.z += iz == k ? dp.z : 0.f; d.w += iw == k ? dp.w : 0.f;
        one(px[k], d, yq[k]);
      }
    }
  }
  }      // chunk loop
  if constexpr (PASS == 1) return;
  // block reduction over the PL pixel lanes in fixed order
  float* r1 = sh[0];
  float* r2 = sh[1];
  *reinterpret_cast<float4*>(r1 + (pl * Q4 + q) * 4) = s1;
  *reinterpret_cast<float4*>(r2 + (pl * Q4 + q) * 4) = s2;
  __syncthreads();
  for (int ch = threadIdx.x; ch < C; ch += VV_WG) {      // C <= 256: one channel per thread; nf = 64 has 512-channel layers
    float t1 = 0.f, t2 = 0.f;
    for (int k = 0; k < PL; ++k) { t1 += r1[k * C + ch]; t2 += r2[k * C + ch]; }
    float* o = p.partial + ((int64_t)(g * nblk + blk) * 2) * C + ch;
    o[0] = t1;
    o[C] = t2;
  }
}

// The same two passes when y, dA, dpool and dy are ALL stored as bf16 (mixed precision): 8 channels = 16 bytes per lane and tensor
// instead of 4 channels = 8 bytes, the same 256-pixel blocks and partial-sum layout.  fp32 arithmetic as above.
template <bool POOL, int PASS>
__global__ void __launch_bounds__(VV_WG)
bn_bwd16_kernel(const vv_bnbwd_params p, const int nblk, const int bp, const float* __restrict__ gamma,
                const int64_t param_gstride, const float* __restrict__ scratch) {
  __shared__ float sh[2][VV_WG * 8];
  const int g = blockIdx.y, blk = blockIdx.x;
  const int C = p.C, Q8 = C >> 3, PL = VV_WG / Q8;
  const int q = threadIdx.x % Q8, pl = threadIdx.x / Q8;
  const int c = q * 8;
  const int64_t M = (int64_t)p.B * p.H * p.W;
  const unsigned short* __restrict__ yh = reinterpret_cast<const unsigned short*>(p.y + (int64_t)g * p.y_gstride);
  unsigned short* __restrict__ dzh = reinterpret_cast<unsigned short*>(p.dz + (int64_t)g * p.dz_gstride);
  const unsigned short* __restrict__ dAh = reinterpret_cast<const unsigned short*>(p.dA.ptr + (int64_t)g * p.dA.gstride) + p.dA.coff;
  const int dcs = p.dA.cstride;
  const int64_t abo = (int64_t)g * p.ab_gstride + c;
  float a[8], b[8], m[8], iv[8], s1[8], s2[8], gk[8], c1[8], c2[8];
  auto ld8f = [](const float* q, float* o) {          // 8 consecutive floats, 16-byte aligned: two 16-byte loads
    const float4 lo = *reinterpret_cast<const float4*>(q), hi = *reinterpret_cast<const float4*>(q + 4);
    o[0] = lo.x; o[1] = lo.y; o[2] = lo.z; o[3] = lo.w; o[4] = hi.x; o[5] = hi.y; o[6] = hi.z; o[7] = hi.w;
  };
  ld8f(p.a + abo, a); ld8f(p.b + abo, b); ld8f(p.mean + abo, m); ld8f(p.invstd + abo, iv);
#pragma unroll
  for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; gk[j] = 0.f; c1[j] = 0.f; c2[j] = 0.f; }
  if constexpr (PASS == 1) {
    ld8f(gamma + (int64_t)g * param_gstride + c, gk);
    ld8f(scratch + (int64_t)g * 2 * C + c, c1);
    ld8f(scratch + (int64_t)g * 2 * C + C + c, c2);
#pragma unroll
    for (int j = 0; j < 8; ++j) gk[j] *= iv[j];
  }
  auto ld8 = [&](const unsigned short* q8) -> vv_f8 { return vv_unpack_bf16x8(*reinterpret_cast<const uint4*>(q8)); };
  auto one = [&](const int64_t pix, vv_f8 d, const vv_f8& yv) {
    vv_f8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float z = fmaf(a[j], yv.v[j], b[j]);
      const float dj = z > 0.f ? d.v[j] : 0.f;
      const float xh = (yv.v[j] - m[j]) * iv[j];
      if constexpr (PASS == 0) { s1[j] += dj; s2[j] = fmaf(dj, xh, s2[j]); }
      else o.v[j] = gk[j] * (dj - c1[j] - xh * c2[j]);
    }
    if constexpr (PASS == 1) *reinterpret_cast<uint4*>(dzh + pix * C + c) = vv_pack_bf16x8(o);
  };
  const int64_t nchunks = (M + bp - 1) / bp;
  for (int64_t chunk = blk; chunk < nchunks; chunk += nblk) {       // persistent workgroups, see bn_bwd_reduce_kernel
  if constexpr (!POOL) {
    // bp = 4 * PL for every channel count the bank has (bn_bp): each thread owns four pixels; all eight 16-byte loads are issued
    // before the first use (one pixel at a time made every workgroup a chain of four HBM round trips)
    for (int i0 = pl; i0 < bp; i0 += 4 * PL) {
      uint4 qa[4], qy[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t pix = chunk * bp + i0 + u * PL;
        const bool ok = i0 + u * PL < bp && pix < M;
        qa[u] = ok ? *reinterpret_cast<const uint4*>(dAh + pix * dcs + c) : make_uint4(0, 0, 0, 0);
        qy[u] = ok ? *reinterpret_cast<const uint4*>(yh + pix * C + c) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t pix = chunk * bp + i0 + u * PL;
        if (i0 + u * PL < bp && pix < M) one(pix, vv_unpack_bf16x8(qa[u]), vv_unpack_bf16x8(qy[u]));
      }
    }
  } else {
    const int H2 = p.H >> 1, W2 = p.W >> 1;
    const int64_t MW = M >> 2;
    const unsigned short* __restrict__ dPh = reinterpret_cast<const unsigned short*>(p.dpool + (int64_t)g * p.dpool_gstride);
    for (int i = pl; i < (bp >> 2); i += PL) {
      const int64_t wi = chunk * (bp >> 2) + i;
      if (wi >= MW) continue;
      const int wx = (int)(wi % W2);
      const int64_t t = wi / W2;
      const int wy = (int)(t % H2);
      const int64_t img = t / H2;
      const int64_t p00 = (img * p.H + 2 * wy) * p.W + 2 * wx;
      const int64_t px[4] = {p00, p00 + 1, p00 + p.W, p00 + p.W + 1};
      vv_f8 yq[4];
      int best[8];
      float bv[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        yq[k] = ld8(yh + px[k] * C + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float zz = fmaxf(fmaf(a[j], yq[k].v[j], b[j]), 0.f);
          if (k == 0 || zz > bv[j]) { bv[j] = zz; best[j] = k; }      // first maximum wins ties, like at::max_pool2d
        }
      }
      const vv_f8 dp = ld8(dPh + wi * C + c);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        vv_f8 d = ld8(dAh + px[k] * dcs + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) d.v[j] += best[j] == k ? dp.v[j] : 0.f;
        one(px[k], d, yq[k]);
      }
    }
  }
  }      // chunk loop
  if constexpr (PASS == 1) return;
  float* r1 = sh[0];
  float* r2 = sh[1];
#pragma unroll
  for (int j = 0; j < 8; ++j) { r1[pl * C + c + j] = s1[j]; r2[pl * C + c + j] = s2[j]; }
  __syncthreads();
  if (threadIdx.x < C) {
    float t1 = 0.f, t2 = 0.f;
    for (int k = 0; k < PL; ++k) { t1 += r1[k * C + threadIdx.x]; t2 += r2[k * C + threadIdx.x]; }
    float* o = p.partial + ((int64_t)(g * nblk + blk) * 2) * C + threadIdx.x;
    o[0] = t1;
    o[C] = t2;
  }
}

// phase 2a: sum partials -> dbeta, dgamma, c1 = mean(dz), c2 = mean(dz*xhat)
__global__ void __launch_bounds__(32 * VV_NP)
bn_bwd_sum_kernel(const int C, const int nblk, const double M, const float* __restrict__ partial,
                  float* __restrict__ dgamma, float* __restrict__ dbeta, const int64_t grad_gstride,
                  float* __restrict__ scratch) {
  __shared__ double sh[2][VV_NP][32];
  const int g = blockIdx.y;
  const int cl = threadIdx.x & 31, part = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    const float* st = partial + (int64_t)g * nblk * 2 * C + c;
    vv_sum_partials(st, nblk, C, part, s1, s2);
  }
  sh[0][part][cl] = s1;
  sh[1][part][cl] = s2;
  __syncthreads();
  if (part != 0 || c >= C) return;
  s1 = 0.0; s2 = 0.0;
#pragma unroll
  for (int k = 0; k < VV_NP; ++k) { s1 += sh[0][k][cl]; s2 += sh[1][k][cl]; }
  dbeta[(int64_t)g * grad_gstride + c] = (float)s1;
  dgamma[(int64_t)g * grad_gstride + c] = (float)s2;
  scratch[(int64_t)g * 2 * C + c] = (float)(s1 / M);
  scratch[(int64_t)g * 2 * C + C + c] = (float)(s2 / M);
}

// ------------------------------------------------------------------------------------------------ output conv
// grid (B, G): one block = one cube (HW pixels) of one UNet; CC / 4 lanes per pixel, 4 channels per lane (CC = features_root:
// 32 in every shipped config.cfg, 64 = the default of SelfCompleteNet1raw1of, model/unet.py:563).
template <int CC>
__global__ void __launch_bounds__(VV_WG)
outconv_fwd_kernel(const vv_outconv_params p, const int total, const int nper) {
  constexpr int LPP = CC / 4, NPG = VV_WG / LPP;
  __shared__ float red[4];
  // work item = (cube, UNet), the G UNets of a cube adjacent in one XCD's chunk of the list: they read the same target pixels
  // (3 of the cube's 15 channels / 2 of the flow's each), which then cross HBM once instead of once per UNet
  const int wi = vv_xcd_remap(blockIdx.x, nper);
  if (wi >= total) return;
  const int g = wi % p.G, cube = wi / p.G;
  const int tid = threadIdx.x, sub = tid % LPP, pg = tid / LPP;
  const int c = sub * 4;
  const int C = CC, oc = p.oc[g];
  const int64_t abo = (int64_t)g * p.ab_gstride + c;
  const float4 a4 = *reinterpret_cast<const float4*>(p.a + abo), b4 = *reinterpret_cast<const float4*>(p.b + abo);
  float4 wv[4];
  float bias[4];
#pragma unroll
  for (int co = 0; co < 4; ++co) {
    if (co < oc) {
      wv[co] = *reinterpret_cast<const float4*>(p.w + (int64_t)g * p.param_gstride + co * C + c);
      bias[co] = p.bias[(int64_t)g * p.param_gstride + co];
    } else {
      wv[co] = make_float4(0, 0, 0, 0);
      bias[co] = 0.f;
    }
  }
  const int tsrc = p.tgt_src[g], tco = p.tgt_coff[g];
  const float* tgt = tsrc == 0 ? p.tgt0 : p.tgt1;
  const int tcs = tsrc == 0 ? p.tgt0_cstride : p.tgt1_cstride;
  const float gs = p.gscale ? p.gscale[g] : 0.f;
  const float* __restrict__ y = p.y + (int64_t)g * p.y_gstride;
  const int64_t MB = (int64_t)p.B * p.HW;
  float sse = 0.f;
  // four pixel groups per trip, their loads issued before the first shuffle: one load in flight per lane made the kernel a chain of
  // HBM round trips (the stores inside the body keep the compiler from hoisting the loads itself)
  auto ldy = [&](const int i) -> float4 {
    const int64_t pix = (int64_t)cube * p.HW + i;
    return (p.pad0 & 1) ? vv_unpack_bf16x4(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(y) + pix * C + c))
                        : *reinterpret_cast<const float4*>(y + pix * C + c);      // pad0 bit 0: y holds bf16 elements
  };
  // the target pixel (3 / 2 floats) of lane sub == 0, loaded with the batch: read inside the body it sat behind the previous
  // pixel group's stores, one exposed HBM round trip per body (32 per wave and cube)
  auto ldt = [&](const int i) -> float4 {
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sub == 0) {
      const float* q = tgt + ((int64_t)cube * p.HW + i) * tcs + tco;
      t.x = q[0];
      if (oc > 1) t.y = q[1];
      if (oc > 2) t.z = q[2];
      if (oc > 3) t.w = q[3];
    }
    return t;
  };
  auto body = [&](const int i, const float4 yq, const float4 tq) {
    const int64_t pix = (int64_t)cube * p.HW + i;
    const float4 v = vv_act4(yq, a4, b4);
    float o[4];
#pragma unroll
    for (int co = 0; co < 4; ++co) {
      float d = v.x * wv[co].x;
      d = fmaf(v.y, wv[co].y, d); d = fmaf(v.z, wv[co].z, d); d = fmaf(v.w, wv[co].w, d);
      d = vv_group_sum<LPP>(d);
      o[co] = d + bias[co];
    }
    if (sub == 0) {
      float4 ov = make_float4(0, 0, 0, 0), dv = make_float4(0, 0, 0, 0);
      float e[4] = {0, 0, 0, 0};
      const float tv[4] = {tq.x, tq.y, tq.z, tq.w};
#pragma unroll
      for (int co = 0; co < 4; ++co)
        if (co < oc) {
          e[co] = o[co] - tv[co];
          sse = fmaf(e[co], e[co], sse);
        } else {
          o[co] = 0.f;
        }
      ov = make_float4(o[0], o[1], o[2], o[3]);
      if (p.out4) *reinterpret_cast<float4*>(p.out4 + ((int64_t)g * MB + pix) * 4) = ov;      // NULL: nobody reads the reconstruction (fused train / scoring steps)
      if (p.dout4) {
        dv = make_float4(gs * e[0], gs * e[1], gs * e[2], gs * e[3]);
        *reinterpret_cast<float4*>(p.dout4 + ((int64_t)g * MB + pix) * 4) = dv;
      }
    }
  };
  for (int i0 = pg; i0 < p.HW; i0 += 4 * NPG) {
    float4 yq4[4], tq4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + u * NPG < p.HW) { yq4[u] = ldy(i0 + u * NPG); tq4[u] = ldt(i0 + u * NPG); }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + u * NPG < p.HW) body(i0 + u * NPG, yq4[u], tq4[u]);
  }
  // block reduce sse (only sub==0 lanes hold data): wave reduce then 4 waves
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) sse += __shfl_xor(sse, off);
  if ((tid & 63) == 0) red[tid >> 6] = sse;
  __syncthreads();
  if (tid == 0) p.score[(int64_t)g * p.B + cube] = (red[0] + red[1]) + (red[2] + red[3]);
}

// backward: grid (B, G); partial [G][B][4*CC + 4]
template <int CC>
__global__ void __launch_bounds__(VV_WG)
outconv_bwd_kernel(const int B, const int HW, const int C_, const float* __restrict__ dout4, const float* __restrict__ y,
                   const int64_t y_gstride, const float* __restrict__ a, const float* __restrict__ b,
                   const int64_t ab_gstride, const float* __restrict__ w, const int64_t param_gstride,
                   float* __restrict__ dA, const int64_t dA_gstride, float* __restrict__ partial,
                   const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ bnpart,
                   const int dA_bf16) {      // flags: bit 0 = dA stored as bf16, bit 1 = y holds bf16 elements
  constexpr int C = CC, LPP = CC / 4, NPG = VV_WG / LPP, NOUT = 4 * CC + 4;
  (void)C_;
  __shared__ float sh[NPG][LPP * 16 + 4];
  const int g = blockIdx.y, cube = blockIdx.x;
  const int tid = threadIdx.x, sub = tid % LPP, pg = tid / LPP;
  const int c = sub * 4;
  const int64_t abo = (int64_t)g * ab_gstride + c;
  const float4 a4 = *reinterpret_cast<const float4*>(a + abo), b4 = *reinterpret_cast<const float4*>(b + abo);
  float4 wv[4];
#pragma unroll
  for (int co = 0; co < 4; ++co) wv[co] = *reinterpret_cast<const float4*>(w + (int64_t)g * param_gstride + co * C + c);
  // NOTE: rows co >= oc of `w` belong to the next parameter (bias) -- harmless because dout4 is 0 there.
  const float* __restrict__ yg = y + (int64_t)g * y_gstride;
  float* __restrict__ dAg = dA + (int64_t)g * dA_gstride;
  const int64_t MB = (int64_t)B * HW;
  float4 dw[4];
  float db[4] = {0, 0, 0, 0};
#pragma unroll
  for (int co = 0; co < 4; ++co) dw[co] = make_float4(0, 0, 0, 0);
  // bnpart: this block also leaves the BatchNorm-backward partial sums of the layer in front of the output conv (sum of
  // g = dA * [act > 0] and of g * xhat per channel), so that layer needs no reduction pass over dA and y
  float4 m4 = make_float4(0, 0, 0, 0), i4 = m4, s1 = m4, s2 = m4;
  if (bnpart) { m4 = *reinterpret_cast<const float4*>(mean + abo); i4 = *reinterpret_cast<const float4*>(invstd + abo); }
  // four pixel groups per trip with their loads up front (see outconv_fwd_kernel)
  auto ldd = [&](const int i) -> float4 { return *reinterpret_cast<const float4*>(dout4 + ((int64_t)g * MB + (int64_t)cube * HW + i) * 4); };
  auto ldy = [&](const int i) -> float4 {
    const int64_t pix = (int64_t)cube * HW + i;
    return (dA_bf16 & 2) ? vv_unpack_bf16x4(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(yg) + pix * C + c))
                         : *reinterpret_cast<const float4*>(yg + pix * C + c);
  };
  auto body = [&](const int i, const float4 d, const float4 yv) {
    const int64_t pix = (int64_t)cube * HW + i;
    const float4 v = vv_act4(yv, a4, b4);
    const float dd[4] = {d.x, d.y, d.z, d.w};
    float4 o = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int co = 0; co < 4; ++co) {
      o.x = fmaf(dd[co], wv[co].x, o.x); o.y = fmaf(dd[co], wv[co].y, o.y);
      o.z = fmaf(dd[co], wv[co].z, o.z); o.w = fmaf(dd[co], wv[co].w, o.w);
      dw[co].x = fmaf(dd[co], v.x, dw[co].x); dw[co].y = fmaf(dd[co], v.y, dw[co].y);
      dw[co].z = fmaf(dd[co], v.z, dw[co].z); dw[co].w = fmaf(dd[co], v.w, dw[co].w);
      db[co] += dd[co];
    }
    if (dA_bf16 & 1) {                     // stored as bf16 (mixed precision): the sums below are those of the stored values
      const uint2 h = vv_pack_bf16x4(o);
      *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(dAg) + pix * C + c) = h;
      o = vv_unpack_bf16x4(h);
    } else {
      *reinterpret_cast<float4*>(dAg + pix * C + c) = o;
    }
    if (bnpart) {
      const float4 gq = make_float4(v.x > 0.f ? o.x : 0.f, v.y > 0.f ? o.y : 0.f, v.z > 0.f ? o.z : 0.f, v.w > 0.f ? o.w : 0.f);
      s1.x += gq.x; s1.y += gq.y; s1.z += gq.z; s1.w += gq.w;
      s2.x = fmaf(gq.x, (yv.x - m4.x) * i4.x, s2.x); s2.y = fmaf(gq.y, (yv.y - m4.y) * i4.y, s2.y);
      s2.z = fmaf(gq.z, (yv.z - m4.z) * i4.z, s2.z); s2.w = fmaf(gq.w, (yv.w - m4.w) * i4.w, s2.w);
    }
  };
  for (int i0 = pg; i0 < HW; i0 += 4 * NPG) {
    float4 d4[4], y4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + u * NPG < HW) { d4[u] = ldd(i0 + u * NPG); y4[u] = ldy(i0 + u * NPG); }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + u * NPG < HW) body(i0 + u * NPG, d4[u], y4[u]);
  }
#pragma unroll
  for (int co = 0; co < 4; ++co) *reinterpret_cast<float4*>(&sh[pg][sub * 16 + co * 4]) = dw[co];
  if (sub == 0) {
#pragma unroll
    for (int co = 0; co < 4; ++co) sh[pg][LPP * 16 + co] = db[co];
  }
  __syncthreads();
  float* out = partial + ((int64_t)g * B + cube) * NOUT;
  for (int e = tid; e < NOUT; e += VV_WG) {          // [co][cc] weight-gradient sums, then the 4 bias-gradient sums; fixed order
    float s = 0.f;
    if (e < 4 * CC) {
      const int co = e / CC, cc = e % CC;
      for (int k = 0; k < NPG; ++k) s += sh[k][(cc >> 2) * 16 + co * 4 + (cc & 3)];
    } else {
      for (int k = 0; k < NPG; ++k) s += sh[k][LPP * 16 + (e - 4 * CC)];
    }
    out[e] = s;
  }
  if (!bnpart) return;
  __syncthreads();
  *reinterpret_cast<float4*>(&sh[pg][sub * 8]) = s1;
  *reinterpret_cast<float4*>(&sh[pg][sub * 8 + 4]) = s2;
  __syncthreads();
  if (tid < 2 * C) {                       // threads [0, C) -> sum g, [C, 2C) -> sum g*xhat; fixed order over the pixel lanes
    const int which = tid / CC, cc = tid % CC;
    float s = 0.f;
    for (int k = 0; k < NPG; ++k) s += sh[k][(cc >> 2) * 8 + which * 4 + (cc & 3)];
    bnpart[((int64_t)(g * B + cube) * 2 + which) * C + cc] = s;
  }
}

// grid (G); 1024 threads: NOUT = 4*CC + 4 outputs x NPL partial-sum lanes (7 at CC = 32), fixed order, fp64
template <int CC>
__global__ void __launch_bounds__(1024)
outconv_bwd_reduce_kernel(const int C_, const int nblk, const float* __restrict__ partial, const int* __restrict__ oc,
                          float* __restrict__ dW, float* __restrict__ db, const int64_t grad_gstride) {
  constexpr int NOUT = 4 * CC + 4, NPL = 1024 / NOUT, C = CC;
  (void)C_;
  __shared__ double sh[NPL][NOUT];
  const int g = blockIdx.x, tid = threadIdx.x;
  const int e = tid % NOUT, part = tid / NOUT;
  if (part < NPL) {
    double s0 = 0.0, s1 = 0.0;
    int k = part;
    for (; k + NPL < nblk; k += 2 * NPL) {
      s0 += (double)partial[((int64_t)g * nblk + k) * NOUT + e];
      s1 += (double)partial[((int64_t)g * nblk + k + NPL) * NOUT + e];
    }
    if (k < nblk) s0 += (double)partial[((int64_t)g * nblk + k) * NOUT + e];
    sh[part][e] = s0 + s1;
  }
  __syncthreads();
  if (tid >= NOUT) return;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < NPL; ++k) s += sh[k][tid];
  const int n = oc[g];
  if (tid < 4 * CC) {
    const int co = tid / CC, cc = tid % CC;
    if (co < n) dW[(int64_t)g * grad_gstride + co * C + cc] = (float)s;
  } else if (tid - 4 * CC < n) {
    db[(int64_t)g * grad_gstride + (tid - 4 * CC)] = (float)s;
  }
}

// ------------------------------------------------------------------------------------------------ bias grad
__global__ void __launch_bounds__(VV_WG)
bias_grad_stage1(const int64_t M, const int C, const float* __restrict__ dy, const int64_t dy_gstride, const int cstride,
                 const int coff, float* __restrict__ scratch, const int nblk) {
  __shared__ float sh[VV_WG * 4];
  const int g = blockIdx.y, blk = blockIdx.x;
  const int Q4 = C >> 2, PL = VV_WG / Q4;
  const int q = threadIdx.x % Q4, pl = threadIdx.x / Q4;
  const float* src = dy + (int64_t)g * dy_gstride + coff + q * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
  for (int i = pl; i < 1024; i += PL) {
    const int64_t pix = (int64_t)blk * 1024 + i;
    if (pix < M) {
      const float4 v = *reinterpret_cast<const float4*>(src + pix * cstride);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  *reinterpret_cast<float4*>(sh + (pl * Q4 + q) * 4) = s;
  __syncthreads();
  if (threadIdx.x < C) {
    float t = 0.f;
    for (int k = 0; k < PL; ++k) t += sh[k * C + threadIdx.x];
    scratch[((int64_t)g * nblk + blk) * C + threadIdx.x] = t;
  }
}
__global__ void __launch_bounds__(VV_WG)
bias_grad_stage2(const int C, const int nblk, const float* __restrict__ scratch, float* __restrict__ db,
                 const int64_t grad_gstride) {
  const int g = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += VV_WG) {
    double s = 0.0;
    for (int k = 0; k < nblk; ++k) s += (double)scratch[((int64_t)g * nblk + k) * C + c];
    db[(int64_t)g * grad_gstride + c] = (float)s;
  }
}

// Bias gradient of the transposed conv from the per-tile column sums the data-gradient kernel of the consuming (concat) layer
// leaves in its `stats` partials: db[j] = sum over pixels of d(cat)[pixel][coff + j] = sum over tiles of partial[tile][0][coff + j].
// Same fixed-order fp64 reduction as the BatchNorm partials.
__global__ void __launch_bounds__(32 * VV_NP)
bias_from_partials_kernel(const int C, const int ntiles, const int coff, const int n, const float* __restrict__ partial,
                          const int64_t partial_gstride, float* __restrict__ db, const int64_t grad_gstride) {
  __shared__ double sh[VV_NP][32];
  const int g = blockIdx.y;
  const int cl = threadIdx.x & 31, part = threadIdx.x >> 5;
  const int j = blockIdx.x * 32 + cl;
  double s1 = 0.0, s2 = 0.0;
  if (j < n) vv_sum_partials(partial + (int64_t)g * partial_gstride + coff + j, ntiles, C, part, s1, s2);
  sh[part][cl] = s1;
  __syncthreads();
  if (part != 0 || j >= n) return;
  s1 = 0.0;
#pragma unroll
  for (int k = 0; k < VV_NP; ++k) s1 += sh[k][cl];
  db[(int64_t)g * grad_gstride + j] = (float)s1;
}

// ------------------------------------------------------------------------------------------------ Adam
__global__ void __launch_bounds__(VV_WG)
adam_kernel(const int64_t n4, float4* __restrict__ p, const float4* __restrict__ gr, float4* __restrict__ m,
            float4* __restrict__ v, const float step_size, const float beta1, const float beta2, const float eps,
            const float bc2_sqrt, const float gscale) {
  for (int64_t i = (int64_t)blockIdx.x * VV_WG + threadIdx.x; i < n4; i += (int64_t)gridDim.x * VV_WG) {
    float4 pv = p[i], gv = gr[i], mv = m[i], vv = v[i];
    float* pp = &pv.x; float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gk = gp[k] * gscale;
      mp[k] = mp[k] + (gk - mp[k]) * (1.f - beta1);                 // exp_avg.lerp_(grad, 1-beta1)
      vp[k] = fmaf(1.f - beta2, gk * gk, vp[k] * beta2);            // exp_avg_sq.mul_(beta2).addcmul_(g,g,1-beta2)
      const float denom = __fsqrt_rn(vp[k]) / bc2_sqrt + eps;
      pp[k] = pp[k] - step_size * (mp[k] / denom);
    }
    p[i] = pv; m[i] = mv; v[i] = vv;
  }
}

// Step scalars of Adam in device memory, so that a captured step (hipGraph) replays with the right bias corrections: one thread
// advances the step counter and leaves sc[0] = lr / (1 - beta1^t) (float / float like vv_adam), sc[1] = sqrt(1 - beta2^t).
// beta1 / beta2 arrive as the doubles torch.optim.Adam computes with.
__global__ void adam_tick_kernel(int64_t* __restrict__ t_dev, const float lr, const double beta1, const double beta2,
                                 float* __restrict__ sc) {
  if (threadIdx.x | blockIdx.x) return;
  const int64_t t = t_dev[0] + 1;
  t_dev[0] = t;
  const float bc1 = (float)(1.0 - pow(beta1, (double)t));
  sc[0] = __fdiv_rn(lr, bc1);
  sc[1] = (float)sqrt(1.0 - pow(beta2, (double)t));
}

// Adam over param / m / v [G][U] with the gradients in BUCKET-MAJOR layout: bucket k = columns [bound[k], bound[k+1]) of every
// UNet, stored contiguously as [G][bound[k+1]-bound[k]] at float offset G*bound[k] (one in-place all-reduce per bucket).
struct adam_buckets { int64_t bound4[9]; int nb; };       // bounds in float4 units
__global__ void __launch_bounds__(VV_WG)
adam_bucketed_kernel(const int G, const int64_t U4, const adam_buckets bk, float4* __restrict__ p, const float4* __restrict__ gr,
                     float4* __restrict__ m, float4* __restrict__ v, const float* __restrict__ sc, const float beta1,
                     const float beta2, const float eps, const float gscale) {
  const float step_size = sc[0], bc2_sqrt = sc[1];
  const int64_t n4 = (int64_t)G * U4;
  for (int64_t i = (int64_t)blockIdx.x * VV_WG + threadIdx.x; i < n4; i += (int64_t)gridDim.x * VV_WG) {
    const int64_t g = i / U4, col = i - g * U4;
    int k = 0;
#pragma unroll
    for (int q = 1; q < 8; ++q) k += (q < bk.nb && col >= bk.bound4[q]) ? 1 : 0;
    const int64_t lo = bk.bound4[k], w = bk.bound4[k + 1] - lo;
    float4 pv = p[i], gv = gr[(int64_t)G * lo + g * w + (col - lo)], mv = m[i], vv = v[i];
    float* pp = &pv.x; float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float gk = gp[q] * gscale;
      mp[q] = mp[q] + (gk - mp[q]) * (1.f - beta1);
      vp[q] = fmaf(1.f - beta2, gk * gk, vp[q] * beta2);
      const float denom = __fsqrt_rn(vp[q]) / bc2_sqrt + eps;
      pp[q] = pp[q] - step_size * (mp[q] / denom);
    }
    p[i] = pv; m[i] = mv; v[i] = vv;
  }
}

// ------------------------------------------------------------------------------------------------ adapters
__global__ void __launch_bounds__(VV_WG)
cube_gather_kernel(const int B, const int T, const int Tf, const int HW, const int64_t* __restrict__ idx,
                   const uint8_t* __restrict__ raw, const float* __restrict__ flow, float* __restrict__ x,
                   float* __restrict__ xof) {
  const int64_t e = (int64_t)blockIdx.x * VV_WG + threadIdx.x;
  if (e >= (int64_t)B * HW) return;
  const int b = (int)(e / HW), pix = (int)(e % HW);
  const int64_t n = idx ? idx[b] : b;
  if (raw) {
    const uint8_t* r = raw + (n * T * HW + pix) * 3;
    float* o = x + e * (3 * T);
    for (int t = 0; t < T; ++t) {
      const uint8_t* q = r + (int64_t)t * HW * 3;
      o[t * 3 + 0] = __fdiv_rn((float)q[0], 255.f);
      o[t * 3 + 1] = __fdiv_rn((float)q[1], 255.f);
      o[t * 3 + 2] = __fdiv_rn((float)q[2], 255.f);
    }
  }
  if (flow) {
    const float* f = flow + (n * Tf * HW + pix) * 2;
    float* o = xof + e * (2 * Tf);
    for (int t = 0; t < Tf; ++t) {
      const float2 q = *reinterpret_cast<const float2*>(f + (int64_t)t * HW * 2);
      o[t * 2 + 0] = q.x;
      o[t * 2 + 1] = q.y;
    }
  }
}

__global__ void __launch_bounds__(VV_WG)
nchw_to_nhwc_kernel(const int B, const int C, const int HW, const float* __restrict__ src, float* __restrict__ dst) {
  const int64_t e = (int64_t)blockIdx.x * VV_WG + threadIdx.x;
  if (e >= (int64_t)B * HW) return;
  const int64_t b = e / HW, pix = e % HW;
  for (int c = 0; c < C; ++c) dst[e * C + c] = src[(b * C + c) * HW + pix];
}
__global__ void __launch_bounds__(VV_WG)
out4_to_nchw_kernel(const int B, const int HW, const int oc, const float* __restrict__ out4, float* __restrict__ dst,
                    const int Ctot, const int choff) {
  const int64_t e = (int64_t)blockIdx.x * VV_WG + threadIdx.x;
  if (e >= (int64_t)B * HW) return;
  const int64_t b = e / HW, pix = e % HW;
  const float4 v = *reinterpret_cast<const float4*>(out4 + e * 4);
  const float vv[4] = {v.x, v.y, v.z, v.w};
  for (int c = 0; c < oc; ++c) dst[(b * Ctot + choff + c) * HW + pix] = vv[c];
}
__global__ void __launch_bounds__(VV_WG)
nchw_to_out4_kernel(const int B, const int HW, const int oc, const float* __restrict__ src, const int Ctot,
                    const int choff, float* __restrict__ out4) {
  const int64_t e = (int64_t)blockIdx.x * VV_WG + threadIdx.x;
  if (e >= (int64_t)B * HW) return;
  const int64_t b = e / HW, pix = e % HW;
  float vv[4] = {0, 0, 0, 0};
  for (int c = 0; c < oc; ++c) vv[c] = src[(b * Ctot + choff + c) * HW + pix];
  *reinterpret_cast<float4*>(out4 + e * 4) = make_float4(vv[0], vv[1], vv[2], vv[3]);
}

// pooled activation: out[g][b, y, x, c] = max over the 2x2 window of relu(a*y+b)   (MaxPool2d(2) o ReLU o BatchNorm)
__global__ void __launch_bounds__(VV_WG)
pool_act_kernel(const int64_t n4, const int C, const int H2, const int W2, const float* __restrict__ y, const int64_t y_gstride,
                const float* __restrict__ a, const float* __restrict__ b, const int64_t ab_gstride, float* __restrict__ out,
                const int64_t out_gstride, const int io16) {     // io16: y and out hold bf16 elements (mixed precision)
  const int g = blockIdx.y;
  const int Q4 = C >> 2;
  const float* yg = y + (int64_t)g * y_gstride;
  float* og = out + (int64_t)g * out_gstride;
  for (int64_t e = (int64_t)blockIdx.x * VV_WG + threadIdx.x; e < n4; e += (int64_t)gridDim.x * VV_WG) {
    const int c = (int)(e % Q4) * 4;
    const int64_t pix = e / Q4;                        // pooled pixel index (b, y2, x2)
    const int x2 = (int)(pix % W2);
    const int64_t t = pix / W2;
    const int y2 = (int)(t % H2);
    const int64_t img = t / H2;
    const int W = 2 * W2;
    const float* q = yg + (((img * 2 * H2 + 2 * y2) * W + 2 * x2) * (int64_t)C + c);
    const float4 a4 = *reinterpret_cast<const float4*>(a + (int64_t)g * ab_gstride + c);
    const float4 b4 = *reinterpret_cast<const float4*>(b + (int64_t)g * ab_gstride + c);
    if (io16) {
      const unsigned short* qh = reinterpret_cast<const unsigned short*>(yg) + (((img * 2 * H2 + 2 * y2) * W + 2 * x2) * (int64_t)C + c);
      const float4 w00 = vv_act4(vv_unpack_bf16x4(*reinterpret_cast<const uint2*>(qh)), a4, b4);
      const float4 w01 = vv_act4(vv_unpack_bf16x4(*reinterpret_cast<const uint2*>(qh + C)), a4, b4);
      const float4 w10 = vv_act4(vv_unpack_bf16x4(*reinterpret_cast<const uint2*>(qh + (int64_t)W * C)), a4, b4);
      const float4 w11 = vv_act4(vv_unpack_bf16x4(*reinterpret_cast<const uint2*>(qh + (int64_t)(W + 1) * C)), a4, b4);
      *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(og) + e * 4) = vv_pack_bf16x4(vv_max4(vv_max4(w00, w01), vv_max4(w10, w11)));
      continue;
    }
    const float4 v00 = vv_act4(*reinterpret_cast<const float4*>(q), a4, b4);
    const float4 v01 = vv_act4(*reinterpret_cast<const float4*>(q + C), a4, b4);
    const float4 v10 = vv_act4(*reinterpret_cast<const float4*>(q + (int64_t)W * C), a4, b4);
    const float4 v11 = vv_act4(*reinterpret_cast<const float4*>(q + (int64_t)(W + 1) * C), a4, b4);
    *reinterpret_cast<float4*>(og + e * 4) = vv_max4(vv_max4(v00, v01), vv_max4(v10, v11));
  }
}

// frame erasure: out[g][pixel][k] = chmap[g][k] >= 0 ? cube[pixel][chmap[g][k]] : 0   (model/unet.py:178-183)
__global__ void __launch_bounds__(VV_WG)
cube_erase_kernel(const int G, const int64_t npix, const int Cc, const int CP, const float* __restrict__ cube,
                  const int* __restrict__ chmap, float* __restrict__ out, const int64_t out_gstride, const int out16) {
  // one thread per float4 of an output pixel (CP / 4 threads per pixel) for ALL G UNets: the pixel's 60 bytes are read once
  // (through L1 by its CP / 4 neighbours) and feed G stores, each wave writing 0.5 / 1 KB contiguous per UNet
  const int Q = CP >> 2;
  const int64_t t = (int64_t)blockIdx.x * VV_WG + threadIdx.x;
  const int64_t e = t / Q;
  const int k = (int)(t % Q) * 4;
  if (e >= npix) return;
  const float* q = cube + e * Cc;
  float src[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) src[i] = i < Cc ? q[i] : 0.f;
  for (int g = 0; g < G; ++g) {
    const int* m = chmap + (int64_t)g * CP;
    const int m0 = m[k], m1 = m[k + 1], m2 = m[k + 2], m3 = m[k + 3];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 16; ++i) {          // register-resident select (a dynamic index would spill the pixel to scratch)
      v.x = m0 == i ? src[i] : v.x; v.y = m1 == i ? src[i] : v.y; v.z = m2 == i ? src[i] : v.z; v.w = m3 == i ? src[i] : v.w;
    }
    if (out16) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(out + (int64_t)g * out_gstride) + e * CP + k) = vv_pack_bf16x4(v);
    else *reinterpret_cast<float4*>(out + (int64_t)g * out_gstride + e * CP + k) = v;
  }
}

inline int nblocks(int64_t n, int cap = 1 << 20) {
  int64_t b = (n + VV_WG - 1) / VV_WG;
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// Forward AND backward of the 1x1 output conv in one pass over y (round 4): the fused train step runs vv_outconv_bwd right behind
// vv_outconv_fwd with d(loss)/d(out) = gscale * (out - target) -- nothing in between can change it -- so the activation tensor in
// front of the output conv (the largest of the network: 32 channels at full resolution, 335 MB per launch in BASELINE config 4)
// crosses HBM once instead of twice and the d(out) tensor never exists.  Per pixel: outconv_fwd_kernel's arithmetic, then
// outconv_bwd_kernel's with d = gscale * (out - target) taken from registers; same operation order as the two kernels -> the same
// bits in score, dA, the weight / bias gradient partials and the BatchNorm-backward partial sums.
template <int CC>
__global__ void __launch_bounds__(VV_WG)
outconv_fwdbwd_kernel(const vv_outconv_params p, float* __restrict__ dA, const int64_t dA_gstride, float* __restrict__ partial,
                      const float* __restrict__ mean, const float* __restrict__ invstd, float* __restrict__ bnpart,
                      const int flags, const int total, const int nper) {
  constexpr int C = CC, LPP = CC / 4, NPG = VV_WG / LPP, NOUT = 4 * CC + 4;
  __shared__ float sh[NPG][LPP * 16 + 4];
  __shared__ float red[4];
  const int wi = vv_xcd_remap(blockIdx.x, nper);
  if (wi >= total) return;
  const int g = wi % p.G, cube = wi / p.G;
  const int tid = threadIdx.x, sub = tid % LPP, pg = tid / LPP;
  const int c = sub * 4;
  const int oc = p.oc[g];
  const int64_t abo = (int64_t)g * p.ab_gstride + c;
  const float4 a4 = *reinterpret_cast<const float4*>(p.a + abo), b4 = *reinterpret_cast<const float4*>(p.b + abo);
  float4 wv[4];          // forward rows (rows >= oc: zero)
  float4 wb[4];          // backward rows as outconv_bwd_kernel reads them (rows >= oc hold the next parameter: d is 0 there)
  float bias[4];
#pragma unroll
  for (int co = 0; co < 4; ++co) {
    wb[co] = *reinterpret_cast<const float4*>(p.w + (int64_t)g * p.param_gstride + co * C + c);
    if (co < oc) { wv[co] = wb[co]; bias[co] = p.bias[(int64_t)g * p.param_gstride + co]; }
    else { wv[co] = make_float4(0, 0, 0, 0); bias[co] = 0.f; }
  }
  const int tsrc = p.tgt_src[g], tco = p.tgt_coff[g];
  const float* tgt = tsrc == 0 ? p.tgt0 : p.tgt1;
  const int tcs = tsrc == 0 ? p.tgt0_cstride : p.tgt1_cstride;
  const float gs = p.gscale[g];
  const float* __restrict__ y = p.y + (int64_t)g * p.y_gstride;
  float* __restrict__ dAg = dA + (int64_t)g * dA_gstride;
  const int64_t MB = (int64_t)p.B * p.HW;
  float sse = 0.f;
  float4 dw[4];
  float db[4] = {0, 0, 0, 0};
#pragma unroll
  for (int co = 0; co < 4; ++co) dw[co] = make_float4(0, 0, 0, 0);
  float4 m4 = make_float4(0, 0, 0, 0), i4 = m4, s1 = m4, s2 = m4;
  if (bnpart) { m4 = *reinterpret_cast<const float4*>(mean + abo); i4 = *reinterpret_cast<const float4*>(invstd + abo); }
  auto ldy = [&](const int i) -> float4 {
    const int64_t pix = (int64_t)cube * p.HW + i;
    return (p.pad0 & 1) ? vv_unpack_bf16x4(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(y) + pix * C + c))
                        : *reinterpret_cast<const float4*>(y + pix * C + c);
  };
  auto ldt = [&](const int i) -> float4 {      // every lane of the pixel reads the target (the backward half needs d in all of them)
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    const float* q = tgt + ((int64_t)cube * p.HW + i) * tcs + tco;
    t.x = q[0];
    if (oc > 1) t.y = q[1];
    if (oc > 2) t.z = q[2];
    if (oc > 3) t.w = q[3];
    return t;
  };
  auto body = [&](const int i, const float4 yv, const float4 tq) {
    const int64_t pix = (int64_t)cube * p.HW + i;
    const float4 v = vv_act4(yv, a4, b4);
    float o[4];
#pragma unroll
    for (int co = 0; co < 4; ++co) {
      float d = v.x * wv[co].x;
      d = fmaf(v.y, wv[co].y, d); d = fmaf(v.z, wv[co].z, d); d = fmaf(v.w, wv[co].w, d);
      d = vv_group_sum<LPP>(d);
      o[co] = d + bias[co];
    }
    const float tv[4] = {tq.x, tq.y, tq.z, tq.w};
    float e[4] = {0, 0, 0, 0}, dd[4];
#pragma unroll
    for (int co = 0; co < 4; ++co) {
      if (co < oc) e[co] = o[co] - tv[co]; else o[co] = 0.f;
      dd[co] = gs * e[co];
    }
    if (sub == 0) {
#pragma unroll
      for (int co = 0; co < 4; ++co) if (co < oc) sse = fmaf(e[co], e[co], sse);
      if (p.out4) *reinterpret_cast<float4*>(p.out4 + ((int64_t)g * MB + pix) * 4) = make_float4(o[0], o[1], o[2], o[3]);
      if (p.dout4) *reinterpret_cast<float4*>(p.dout4 + ((int64_t)g * MB + pix) * 4) = make_float4(dd[0], dd[1], dd[2], dd[3]);
    }
    float4 q = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int co = 0; co < 4; ++co) {
      q.x = fmaf(dd[co], wb[co].x, q.x); q.y = fmaf(dd[co], wb[co].y, q.y);
      q.z = fmaf(dd[co], wb[co].z, q.z); q.w = fmaf(dd[co], wb[co].w, q.w);
      dw[co].x = fmaf(dd[co], v.x, dw[co].x); dw[co].y = fmaf(dd[co], v.y, dw[co].y);
      dw[co].z = fmaf(dd[co], v.z, dw[co].z); dw[co].w = fmaf(dd[co], v.w, dw[co].w);
      db[co] += dd[co];
    }
    if (flags & 1) {
      const uint2 h = vv_pack_bf16x4(q);
      *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(dAg) + pix * C + c) = h;
      q = vv_unpack_bf16x4(h);
    } else {
      *reinterpret_cast<float4*>(dAg + pix * C + c) = q;
    }
    if (bnpart) {
      const float4 gq = make_float4(v.x > 0.f ? q.x : 0.f, v.y > 0.f ? q.y : 0.f, v.z > 0.f ? q.z : 0.f, v.w > 0.f ? q.w : 0.f);
      s1.x += gq.x; s1.y += gq.y; s1.z += gq.z; s1.w += gq.w;
      s2.x = fmaf(gq.x, (yv.x - m4.x) * i4.x, s2.x); s2.y = fmaf(gq.y, (yv.y - m4.y) * i4.y, s2.y);
      s2.z = fmaf(gq.z, (yv.z - m4.z) * i4.z, s2.z); s2.w = fmaf(gq.w, (yv.w - m4.w) * i4.w, s2.w);
    }
  };
  for (int i0 = pg; i0 < p.HW; i0 += 4 * NPG) {
    float4 yq4[4], tq4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + u * NPG < p.HW) { yq4[u] = ldy(i0 + u * NPG); tq4[u] = ldt(i0 + u * NPG); }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (i0 + u * NPG < p.HW) body(i0 + u * NPG, yq4[u], tq4[u]);
  }
  // per-cube squared error (outconv_fwd_kernel's reduction)
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) sse += __shfl_xor(sse, off);
  if ((tid & 63) == 0) red[tid >> 6] = sse;
  // weight / bias gradient partials and BatchNorm-backward partial sums (outconv_bwd_kernel's reductions)
#pragma unroll
  for (int co = 0; co < 4; ++co) *reinterpret_cast<float4*>(&sh[pg][sub * 16 + co * 4]) = dw[co];
  if (sub == 0) {
#pragma unroll
    for (int co = 0; co < 4; ++co) sh[pg][LPP * 16 + co] = db[co];
  }
  __syncthreads();
  if (tid == 0) p.score[(int64_t)g * p.B + cube] = (red[0] + red[1]) + (red[2] + red[3]);
  float* out = partial + ((int64_t)g * p.B + cube) * NOUT;
  for (int e = tid; e < NOUT; e += VV_WG) {
    float s = 0.f;
    if (e < 4 * CC) {
      const int co = e / CC, cc = e % CC;
      for (int k = 0; k < NPG; ++k) s += sh[k][(cc >> 2) * 16 + co * 4 + (cc & 3)];
    } else {
      for (int k = 0; k < NPG; ++k) s += sh[k][LPP * 16 + (e - 4 * CC)];
    }
    out[e] = s;
  }
  if (!bnpart) return;
  __syncthreads();
  *reinterpret_cast<float4*>(&sh[pg][sub * 8]) = s1;
  *reinterpret_cast<float4*>(&sh[pg][sub * 8 + 4]) = s2;
  __syncthreads();
  if (tid < 2 * C) {
    const int which = tid / CC, cc = tid % CC;
    float s = 0.f;
    for (int k = 0; k < NPG; ++k) s += sh[k][(cc >> 2) * 8 + which * 4 + (cc & 3)];
    bnpart[((int64_t)(g * p.B + cube) * 2 + which) * C + cc] = s;
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------ output conv, 8 channels per lane
// The three launches of the 1x1 output conv (forward: MODE 0, backward from a stored d(out): MODE 1, both in one pass: MODE 2) as ONE
// kernel body with CC / 8 lanes per pixel (4 at features_root 32) instead of CC / 4: the first form spent 900 wave-instructions per
// 32 pixels, 72 of them the packed multiply-adds that do the work -- the rest an 8-lane butterfly per output channel (48 DPP moves +
// 48 hazard nops + adds), the error terms recomputed on all eight lanes, address arithmetic and ~50 branches on run-time flags.
// Here a lane holds 8 channels (one 16-byte load of a bf16 y), the butterfly is the two quad permutes, the element types of y / dA
// are template parameters.  The three modes share every per-pixel expression and every reduction order, so the fused launch leaves
// the bits of the two separate ones (tests/test_gpu_unet.py::test_fused_outconv_forward_backward_bitwise_equal_to_two_launches).
//   per pixel:  v = relu(a y + b);  out[co] = sum_c v[c] W[co][c] + bias[co]  (8 sequential fmas per lane, then the butterfly);
//               e = out - target, sse += e^2 (lane 0 of the pixel), d = gscale e;
//               dA[c] = sum_co d[co] W[co][c];  dW[co][c] += d[co] v[c];  db[co] += d[co];
//               BatchNorm-backward sums of the layer in front: g = dA [v > 0] (of the STORED dA), sum g, sum g * xhat.
//   per cube:   lanes' sums over their pixels -> LDS -> one thread per output adds the NPG pixel lanes in order.
template <int LPP>
__device__ __forceinline__ float vv_group_sum48(float d) {
  static_assert(LPP == 4 || LPP == 8, "4 or 8 lanes per pixel");
  d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0xB1, 0xF, 0xF, false));      // quad_perm [1,0,3,2]
  d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x4E, 0xF, 0xF, false));      // quad_perm [2,3,0,1]
  if constexpr (LPP == 8) d += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, d), 0x141, 0xF, 0xF, false));   // row_half_mirror
  return d;
}

template <int CC, int MODE, bool Y16, bool DA16>
__global__ void __launch_bounds__(VV_WG)
outconv8_kernel(const vv_outconv_params p, const float* __restrict__ dout4_in, float* __restrict__ dA, const int64_t dA_gstride,
                float* __restrict__ partial, const float* __restrict__ mean, const float* __restrict__ invstd,
                float* __restrict__ bnpart, const int total, const int nper) {
  constexpr int C = CC, LPP = CC / 8, NPG = VV_WG / LPP, NOUT = 4 * CC + 4, UN = 4;
  constexpr bool FWD = MODE != 1, BWD = MODE != 0;
  __shared__ float sh[BWD ? NPG : 1][BWD ? LPP * 32 + 4 : 1];
  __shared__ float red[4];
  int g, cube;
  if constexpr (MODE == 1) { g = blockIdx.y; cube = blockIdx.x; }
  else {
    // work item = (cube, UNet), the G UNets of a cube adjacent in one XCD's chunk of the list: they read the same target pixels
    const int wi = vv_xcd_remap(blockIdx.x, nper);
    if (wi >= total) return;
    g = wi % p.G; cube = wi / p.G;
  }
  const int tid = threadIdx.x, sub = tid % LPP, pg = tid / LPP;
  const int c = sub * 8;
  const int oc = FWD ? p.oc[g] : 4;
  const int64_t abo = (int64_t)g * p.ab_gstride + c;
  float a[8], b[8], wv[4][8], bias[4];
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = p.a[abo + j]; b[j] = p.b[abo + j]; }
#pragma unroll
  for (int co = 0; co < 4; ++co) {
    // (forward modes: rows >= oc are zero; the backward-only launch does not know oc and reads the rows as they are -- they belong
    //  to the next parameter, and d is 0 there: the same sums)
    const bool on = co < oc;
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[co][j] = on ? p.w[(int64_t)g * p.param_gstride + co * C + c + j] : 0.f;
    bias[co] = (FWD && on) ? p.bias[(int64_t)g * p.param_gstride + co] : 0.f;
  }
  int tcs = 0;
  const float* tgt = nullptr;
  float gs = 0.f;
  if constexpr (FWD) {
    const int tsrc = p.tgt_src[g];
    tgt = (tsrc == 0 ? p.tgt0 : p.tgt1) + p.tgt_coff[g];
    tcs = tsrc == 0 ? p.tgt0_cstride : p.tgt1_cstride;
    gs = p.gscale ? p.gscale[g] : 0.f;
  }
  const int64_t MB = (int64_t)p.B * p.HW;
  const int64_t pix0 = (int64_t)cube * p.HW;
  const float* __restrict__ yf = p.y + (int64_t)g * p.y_gstride;
  const unsigned short* __restrict__ yh = reinterpret_cast<const unsigned short*>(yf);
  float* __restrict__ dAf = BWD ? dA + (int64_t)g * dA_gstride : nullptr;
  unsigned short* __restrict__ dAh = reinterpret_cast<unsigned short*>(dAf);
  float sse = 0.f;
  float dw[BWD ? 4 : 1][8], db[4] = {0.f, 0.f, 0.f, 0.f}, m[8], iv[8], s1[8], s2[8];
  const bool bnp = BWD && bnpart != nullptr;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    m[j] = bnp ? mean[abo + j] : 0.f; iv[j] = bnp ? invstd[abo + j] : 0.f; s1[j] = 0.f; s2[j] = 0.f;
    if constexpr (BWD) { dw[0][j] = 0.f; dw[1][j] = 0.f; dw[2][j] = 0.f; dw[3][j] = 0.f; }
  }
  struct Y8 { uint4 lo, hi; };              // 8 channels of y: bf16 in lo | fp32 in lo, hi
  auto ldy = [&](const int i) -> Y8 {
    Y8 r;
    if constexpr (Y16) { r.lo = *reinterpret_cast<const uint4*>(yh + (pix0 + i) * C + c); r.hi = r.lo; }
    else {
      r.lo = *reinterpret_cast<const uint4*>(yf + (pix0 + i) * C + c);
      r.hi = *reinterpret_cast<const uint4*>(yf + (pix0 + i) * C + c + 4);
    }
    return r;
  };
  auto y8 = [&](const Y8& r) -> vv_f8 {
    if constexpr (Y16) return vv_unpack_bf16x8(r.lo);
    else {
      vv_f8 f;
      f.v[0] = __builtin_bit_cast(float, r.lo.x); f.v[1] = __builtin_bit_cast(float, r.lo.y); f.v[2] = __builtin_bit_cast(float, r.lo.z);
      f.v[3] = __builtin_bit_cast(float, r.lo.w); f.v[4] = __builtin_bit_cast(float, r.hi.x); f.v[5] = __builtin_bit_cast(float, r.hi.y);
      f.v[6] = __builtin_bit_cast(float, r.hi.z); f.v[7] = __builtin_bit_cast(float, r.hi.w);
      return f;
    }
  };
  // the target pixel (forward modes; every lane of the pixel reads it: the backward half needs d in all of them) or the stored d(out).
  // Channels >= oc: the load is clamped to the last real channel and the value dropped -- no branch, nothing read past the pixel
  const int k1 = oc > 1 ? 1 : oc - 1, k2 = oc > 2 ? 2 : oc - 1, k3 = oc > 3 ? 3 : oc - 1;
  auto ldt = [&](const int i) -> float4 {
    if constexpr (FWD) {
      const float* q = tgt + (pix0 + i) * tcs;
      return make_float4(q[0], q[k1], q[k2], q[k3]);
    } else {
      return *reinterpret_cast<const float4*>(dout4_in + ((int64_t)g * MB + pix0 + i) * 4);
    }
  };
  auto body = [&](const int i, const Y8& yr, const float4 tq) {
    const int64_t pix = pix0 + i;
    const vv_f8 yv = y8(yr);
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float z = fmaf(a[j], yv.v[j], b[j]); v[j] = z > 0.f ? z : 0.f; }
    float dd[4];
    if constexpr (FWD) {
      float o[4];
#pragma unroll
      for (int co = 0; co < 4; ++co) {
        float d = v[0] * wv[co][0];
#pragma unroll
        for (int j = 1; j < 8; ++j) d = fmaf(v[j], wv[co][j], d);
        o[co] = d;
      }
      // the four butterflies step by step (independent chains back to back: no DPP hazard stalls)
#pragma unroll
      for (int co = 0; co < 4; ++co) o[co] = vv_group_sum48<LPP>(o[co]) + bias[co];
      const float tv[4] = {tq.x, oc > 1 ? tq.y : 0.f, oc > 2 ? tq.z : 0.f, oc > 3 ? tq.w : 0.f};
      float e[4];
#pragma unroll
      for (int co = 0; co < 4; ++co) { e[co] = o[co] - tv[co]; dd[co] = gs * e[co]; }      // rows >= oc: 0 - 0
      if (sub == 0) {
#pragma unroll
        for (int co = 0; co < 4; ++co) sse = fmaf(e[co], e[co], sse);
        if (p.out4) *reinterpret_cast<float4*>(p.out4 + ((int64_t)g * MB + pix) * 4) = make_float4(o[0], o[1], o[2], o[3]);
        if (p.dout4) *reinterpret_cast<float4*>(p.dout4 + ((int64_t)g * MB + pix) * 4) = make_float4(dd[0], dd[1], dd[2], dd[3]);
      }
    } else {
      dd[0] = tq.x; dd[1] = tq.y; dd[2] = tq.z; dd[3] = tq.w;
    }
    if constexpr (BWD) {
      float q[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) q[j] = 0.f;
#pragma unroll
      for (int co = 0; co < 4; ++co) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { q[j] = fmaf(dd[co], wv[co][j], q[j]); dw[co][j] = fmaf(dd[co], v[j], dw[co][j]); }
        db[co] += dd[co];
      }
      if constexpr (DA16) {                 // stored as bf16 (mixed precision): the sums below are those of the stored values
        vv_f8 qf;
#pragma unroll
        for (int j = 0; j < 8; ++j) qf.v[j] = q[j];
        const uint4 h = vv_pack_bf16x8(qf);
        *reinterpret_cast<uint4*>(dAh + pix * C + c) = h;
        qf = vv_unpack_bf16x8(h);
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = qf.v[j];
      } else {
        *reinterpret_cast<float4*>(dAf + pix * C + c) = make_float4(q[0], q[1], q[2], q[3]);
        *reinterpret_cast<float4*>(dAf + pix * C + c + 4) = make_float4(q[4], q[5], q[6], q[7]);
      }
      if (bnp) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float gq = v[j] > 0.f ? q[j] : 0.f;
          s1[j] += gq;
          s2[j] = fmaf(gq, (yv.v[j] - m[j]) * iv[j], s2[j]);
        }
      }
    }
  };
  // UN pixel groups per trip, their loads issued before the first use (one load in flight per lane = a chain of HBM round trips)
  for (int i0 = pg; i0 < p.HW; i0 += UN * NPG) {
    Y8 yq[UN];
    float4 tq[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u)
      if (i0 + u * NPG < p.HW) { yq[u] = ldy(i0 + u * NPG); tq[u] = ldt(i0 + u * NPG); }
#pragma unroll
    for (int u = 0; u < UN; ++u)
      if (i0 + u * NPG < p.HW) body(i0 + u * NPG, yq[u], tq[u]);
  }
  if constexpr (FWD) {
    // per-cube squared error (only the sub == 0 lanes hold data): wave reduce, then the four waves
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sse += __shfl_xor(sse, off);
    if ((tid & 63) == 0) red[tid >> 6] = sse;
  }
  if constexpr (BWD) {
#pragma unroll
    for (int co = 0; co < 4; ++co)
#pragma unroll
      for (int j = 0; j < 8; ++j) sh[pg][sub * 32 + co * 8 + j] = dw[co][j];
    if (sub == 0) {
#pragma unroll
      for (int co = 0; co < 4; ++co) sh[pg][LPP * 32 + co] = db[co];
    }
  }
  __syncthreads();
  if constexpr (FWD) {
    if (tid == 0) p.score[(int64_t)g * p.B + cube] = (red[0] + red[1]) + (red[2] + red[3]);
  }
  if constexpr (BWD) {
    // [co][cc] weight-gradient sums, then the 4 bias-gradient sums of this cube; fixed order over the pixel lanes
    float* out = partial + ((int64_t)g * p.B + cube) * NOUT;
    for (int e = tid; e < NOUT; e += VV_WG) {
      float s = 0.f;
      if (e < 4 * CC) {
        const int co = e / CC, cc = e % CC;
        for (int k = 0; k < NPG; ++k) s += sh[k][(cc >> 3) * 32 + co * 8 + (cc & 7)];
      } else {
        for (int k = 0; k < NPG; ++k) s += sh[k][LPP * 32 + (e - 4 * CC)];
      }
      out[e] = s;
    }
    if (!bnp) return;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) { sh[pg][sub * 16 + j] = s1[j]; sh[pg][sub * 16 + 8 + j] = s2[j]; }
    __syncthreads();
    if (tid < 2 * C) {                     // threads [0, C): sum g, [C, 2C): sum g * xhat
      const int which = tid / CC, cc = tid % CC;
      float s = 0.f;
      for (int k = 0; k < NPG; ++k) s += sh[k][(cc >> 3) * 16 + which * 8 + (cc & 7)];
      bnpart[((int64_t)(g * p.B + cube) * 2 + which) * C + cc] = s;
    }
  }
}

template <int MODE>
static int launch_outconv8(const vv_outconv_params& p, const float* dout4_in, float* dA, int64_t dA_gstride, float* partial,
                           const float* mean, const float* invstd, float* bnpart, bool y16, bool da16, hipStream_t st) {
  const int total = p.B * p.G, nper = (total + 7) / 8;
  const dim3 grid = MODE == 1 ? dim3(p.B, p.G) : dim3(nper * 8);
#define VV_OC8(CC_, Y_, D_)                                                                                                        \
  VV_LAUNCH((outconv8_kernel<CC_, MODE, Y_, D_>), grid, dim3(VV_WG), 0, st, p, dout4_in, dA, dA_gstride, partial, mean, invstd, bnpart, \
            total, nper)
  if (!y16) return VV_ERR_UNSUPPORTED;      // (fp32 y: the callers keep the 4-channels-per-lane kernels, see vv_outconv_fwd)
  if (p.C == 32) {
    if (da16) VV_OC8(32, true, true); else VV_OC8(32, true, false);
  } else if (p.C == 64) {
    if (da16) VV_OC8(64, true, true); else VV_OC8(64, true, false);
  } else return VV_ERR_UNSUPPORTED;      /* features_root 32 (every shipped config) or 64 (SelfCompleteNet1raw1of's default) */
#undef VV_OC8
  VV_CHECK_LAUNCH();
  return VV_OK;
}

extern "C" int vv_pack_weights(const vv_pack_entry* table_dev, int32_t nentries, int32_t G, const float* params,
                               int64_t params_gstride, float* packed, int64_t packed_gstride, int32_t max_elems,
                               vv_stream stream) {
  if (!table_dev || !params || !packed || nentries <= 0) return VV_ERR_BAD_ARG;
  int bx = nblocks(max_elems, 64);
  VV_LAUNCH(pack_weights_kernel, dim3(bx, nentries, G), dim3(VV_WG), 0, (hipStream_t)stream, table_dev, params,
                     params_gstride, packed, packed_gstride);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

extern "C" int vv_bn_finalize(int32_t G, int32_t C, int32_t ntiles, int64_t count, int32_t train, float momentum,
                              float eps, const float* stats, int64_t stats_gstride, const float* gamma,
                              const float* beta, int64_t param_gstride, float* running_mean, float* running_var,
                              int64_t buf_gstride, float* a, float* b, float* mean, float* invstd,
                              int64_t ab_gstride, vv_stream stream) {
  if (!gamma || !beta || !running_mean || !running_var || !a || !b || !mean || !invstd) return VV_ERR_BAD_ARG;
  if (train && !stats) return VV_ERR_BAD_ARG;
  VV_LAUNCH(bn_finalize_kernel, dim3((C + 31) / 32, G), dim3(32 * VV_NP), 0, (hipStream_t)stream, C, ntiles,
                     (double)count, train, momentum, eps, stats, stats_gstride, gamma, beta, param_gstride,
                     running_mean, running_var, buf_gstride, a, b, mean, invstd, ab_gstride);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

// every tensor of the BatchNorm backward stored as bf16, 16-byte aligned 8-channel items: the wide kernel
static inline bool bn_all16(const vv_bnbwd_params* p) {
  const int all = VV_BNBWD_DZ_BF16 | VV_BNBWD_DA_BF16 | VV_BNBWD_Y_BF16;
  return (p->flags & all) == all && p->C % 8 == 0 && VV_WG % (p->C / 8) == 0 && p->dA.cstride % 8 == 0 && p->dA.coff % 8 == 0;
}

static inline int bn_bp(int C) {
  int bp = 8192 / (C > 0 ? C : 1);
  bp = bp > 256 ? 256 : (bp < 16 ? 16 : bp);
  return bp & ~3;                       // whole 2x2 pooling windows
}
// the all-bf16 kernel (8 channels per lane, four pixels' loads in flight) likes blocks twice as large: swept 4096 / 8192 / 16384 per
// C on BASELINE config 4: 2.86 / 2.36 / 2.26 ms of BatchNorm backward per step (the fp32 kernel: 1.08 / 1.08 / 1.11)
static inline int bn_bp16(int C) {
  int bp = 16384 / (C > 0 ? C : 1);
  bp = bp > 512 ? 512 : (bp < 16 ? 16 : bp);
  return bp & ~3;
}

// workgroups (= partial-sum blocks) per UNet: one per chunk of bn_bp(C) pixels, at most VV_BN_MAXBLK persistent ones.  Measured
// with 256 persistent workgroups per UNet: the passes over the 32x32 tensors already run at 5.5-6 TB/s with one chunk per
// workgroup and lost 10-15 % in fp32 -- the cap is therefore far above any chunk count the bank produces.
constexpr int VV_BN_MAXBLK = 1 << 20;
extern "C" int vv_bn_bwd_nblk(int32_t B, int32_t H, int32_t W, int32_t C) {
  const int64_t M = (int64_t)B * H * W;
  const int bp = bn_bp(C);
  const int64_t n = (M + bp - 1) / bp;
  return (int)(n < VV_BN_MAXBLK ? n : VV_BN_MAXBLK);
}

// pixel block / block count of a launch with these parameters (<= vv_bn_bwd_nblk, which sizes the partial-sum buffer)
static inline int bn_bp_of(const vv_bnbwd_params* p) { return bn_all16(p) ? bn_bp16(p->C) : bn_bp(p->C); }
static inline int bn_nblk_of(const vv_bnbwd_params* p) {
  const int64_t M = (int64_t)p->B * p->H * p->W;
  const int bp = bn_bp_of(p);
  const int64_t n = (M + bp - 1) / bp;
  return (int)(n < VV_BN_MAXBLK ? n : VV_BN_MAXBLK);
}

extern "C" int vv_bn_bwd_reduce(const vv_bnbwd_params* p, vv_stream stream) {
  if (!p || !p->y || !p->dA.ptr || !p->dz || !p->partial) return VV_ERR_BAD_ARG;
  if (p->C % 4 || p->C > 1024 || VV_WG % (p->C / 4)) return VV_ERR_UNSUPPORTED;
  const int nblk = bn_nblk_of(p);
  if (bn_all16(p)) {
    if (p->dpool)
      VV_LAUNCH((bn_bwd16_kernel<true, 0>), dim3(nblk, p->G), dim3(VV_WG), 0, (hipStream_t)stream, *p, nblk, bn_bp_of(p), nullptr, 0, nullptr);
    else
      VV_LAUNCH((bn_bwd16_kernel<false, 0>), dim3(nblk, p->G), dim3(VV_WG), 0, (hipStream_t)stream, *p, nblk, bn_bp_of(p), nullptr, 0, nullptr);
  } else if (p->dpool)
    VV_LAUNCH((bn_bwd_reduce_kernel<true, 0>), dim3(nblk, p->G), dim3(VV_WG), 0, (hipStream_t)stream, *p, nblk, bn_bp_of(p), nullptr, 0, nullptr);
  else
    VV_LAUNCH((bn_bwd_reduce_kernel<false, 0>), dim3(nblk, p->G), dim3(VV_WG), 0, (hipStream_t)stream, *p, nblk, bn_bp_of(p), nullptr, 0, nullptr);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

extern "C" int vv_bn_bwd_apply(const vv_bnbwd_params* p, const float* gamma, int64_t param_gstride, float* dgamma,
                               float* dbeta, int64_t grad_gstride, float* scratch, vv_stream stream) {
  if (!p || !p->y || !p->dA.ptr || !p->dz || !p->partial || !gamma || !dgamma || !dbeta || !scratch) return VV_ERR_BAD_ARG;
  // VV_BNBWD_PARTIALS_PER_CUBE: the partials were left by vv_outconv_bwd (one block per cube), not by vv_bn_bwd_reduce
  // VV_BNBWD_PARTIALS_PER_TILE: ... by the Winograd data-gradient launch that produced dA (one block per pixel tile)
  const int nblk = (p->flags & VV_BNBWD_PARTIALS_PER_CUBE) ? p->B
                   : (p->flags & VV_BNBWD_PARTIALS_PER_TILE) ? vv_wino_ntiles(p->B, p->H)
                   : (p->flags & VV_BNBWD_PARTIALS_PER_TILE44) ? vv_wino44_ntiles(p->B, p->H)
                   : (p->flags & VV_BNBWD_PARTIALS_PER_CTILE) ? vv_conv_ntiles(p->B, p->H, p->W)
                   : (p->flags & VV_BNBWD_PARTIALS_PER_TTILE) ? vv_convt_dgrad_ntiles(p->B, p->H, p->W, 0) : bn_nblk_of(p);
  if (nblk <= 0) return VV_ERR_BAD_ARG;
  const int nblk_apply = bn_nblk_of(p);
  const int64_t M = (int64_t)p->B * p->H * p->W;
  VV_LAUNCH(bn_bwd_sum_kernel, dim3((p->C + 31) / 32, p->G), dim3(32 * VV_NP), 0, (hipStream_t)stream, p->C, nblk, (double)M,
            p->partial, dgamma, dbeta, grad_gstride, scratch);
  VV_CHECK_LAUNCH();
  if (bn_all16(p)) {
    if (p->dpool)
      VV_LAUNCH((bn_bwd16_kernel<true, 1>), dim3(nblk_apply, p->G), dim3(VV_WG), 0, (hipStream_t)stream, *p, nblk_apply, bn_bp_of(p), gamma, param_gstride, scratch);
    else
      VV_LAUNCH((bn_bwd16_kernel<false, 1>), dim3(nblk_apply, p->G), dim3(VV_WG), 0, (hipStream_t)stream, *p, nblk_apply, bn_bp_of(p), gamma, param_gstride, scratch);
  } else if (p->dpool)
    VV_LAUNCH((bn_bwd_reduce_kernel<true, 1>), dim3(nblk_apply, p->G), dim3(VV_WG), 0, (hipStream_t)stream, *p, nblk_apply, bn_bp_of(p), gamma, param_gstride, scratch);
  else
    VV_LAUNCH((bn_bwd_reduce_kernel<false, 1>), dim3(nblk_apply, p->G), dim3(VV_WG), 0, (hipStream_t)stream, *p, nblk_apply, bn_bp_of(p), gamma, param_gstride, scratch);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

extern "C" int vv_outconv_fwd(const vv_outconv_params* p, vv_stream stream) {
  if (!p || !p->y || !p->a || !p->b || !p->w || !p->bias || !p->oc || !p->tgt_src || !p->tgt_coff || !p->score || !p->tgt0)
    return VV_ERR_BAD_ARG;
  // bf16 y: the 8-channels-per-lane kernel (the 4-per-lane form is instruction-bound there: 315 -> 192 us for the fused launch of
  // BASELINE config 4).  fp32 y: the 4-per-lane kernels stay -- 100 registers less per lane, five waves per SIMD instead of two, and
  // with 32 bytes of y per lane the launch is bound by loads in flight, not by instructions (77 us against 93 - 104 us).  All three
  // launches of a precision use the same family, so fused and separate launches keep identical bits.
  if (p->pad0 & 1) return launch_outconv8<0>(*p, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, true, false, (hipStream_t)stream);
  const int total = p->B * p->G, nper = (total + 7) / 8;
  if (p->C == 32) VV_LAUNCH(outconv_fwd_kernel<32>, dim3(nper * 8), dim3(VV_WG), 0, (hipStream_t)stream, *p, total, nper);
  else if (p->C == 64) VV_LAUNCH(outconv_fwd_kernel<64>, dim3(nper * 8), dim3(VV_WG), 0, (hipStream_t)stream, *p, total, nper);
  else return VV_ERR_UNSUPPORTED;      /* features_root 32 (every shipped config) or 64 (SelfCompleteNet1raw1of's default) */
  VV_CHECK_LAUNCH();
  return VV_OK;
}

extern "C" int vv_outconv_fwdbwd(const vv_outconv_params* p, float* dA, int64_t dA_gstride, float* partial, const float* mean,
                                 const float* invstd, float* bnpart, int32_t flags, vv_stream stream) {
  if (!p || !p->y || !p->a || !p->b || !p->w || !p->bias || !p->oc || !p->tgt_src || !p->tgt_coff || !p->score || !p->tgt0 ||
      !p->gscale || !dA || !partial)
    return VV_ERR_BAD_ARG;
  if (bnpart && (!mean || !invstd)) return VV_ERR_BAD_ARG;
  if (((flags & 2) != 0) != ((p->pad0 & 1) != 0)) return VV_ERR_BAD_ARG;      // one y tensor: both halves must agree on its element type
  if (flags & 2) return launch_outconv8<2>(*p, nullptr, dA, dA_gstride, partial, mean, invstd, bnpart, true, (flags & 1) != 0, (hipStream_t)stream);
  const int total = p->B * p->G, nper = (total + 7) / 8;
  if (p->C == 32) VV_LAUNCH(outconv_fwdbwd_kernel<32>, dim3(nper * 8), dim3(VV_WG), 0, (hipStream_t)stream, *p, dA, dA_gstride, partial, mean, invstd, bnpart, flags, total, nper);
  else if (p->C == 64) VV_LAUNCH(outconv_fwdbwd_kernel<64>, dim3(nper * 8), dim3(VV_WG), 0, (hipStream_t)stream, *p, dA, dA_gstride, partial, mean, invstd, bnpart, flags, total, nper);
  else return VV_ERR_UNSUPPORTED;
  VV_CHECK_LAUNCH();
  return VV_OK;
}

extern "C" int vv_outconv_bwd_nblk(int32_t B, int32_t HW) { (void)HW; return B; }

extern "C" int vv_outconv_bwd(int32_t G, int32_t B, int32_t HW, int32_t C, const float* dout4, const float* y,
                              int64_t y_gstride, const float* a, const float* b, int64_t ab_gstride, const float* w,
                              int64_t param_gstride, float* dA, int64_t dA_gstride, float* partial,
                              const float* mean, const float* invstd, float* bnpart, int32_t dA_bf16, vv_stream stream) {
  if (!dout4 || !y || !a || !b || !w || !dA || !partial) return VV_ERR_BAD_ARG;
  if (bnpart && (!mean || !invstd)) return VV_ERR_BAD_ARG;
  if (dA_bf16 & 2) {
    vv_outconv_params p = {};
    p.G = G; p.B = B; p.HW = HW; p.C = C;
    p.y = y; p.y_gstride = y_gstride; p.a = a; p.b = b; p.ab_gstride = ab_gstride;
    p.w = w; p.param_gstride = param_gstride;
    return launch_outconv8<1>(p, dout4, dA, dA_gstride, partial, mean, invstd, bnpart, true, (dA_bf16 & 1) != 0, (hipStream_t)stream);
  }
  if (C == 32)
    VV_LAUNCH(outconv_bwd_kernel<32>, dim3(B, G), dim3(VV_WG), 0, (hipStream_t)stream, B, HW, C, dout4, y, y_gstride,
              a, b, ab_gstride, w, param_gstride, dA, dA_gstride, partial, mean, invstd, bnpart, dA_bf16);
  else if (C == 64)
    VV_LAUNCH(outconv_bwd_kernel<64>, dim3(B, G), dim3(VV_WG), 0, (hipStream_t)stream, B, HW, C, dout4, y, y_gstride,
              a, b, ab_gstride, w, param_gstride, dA, dA_gstride, partial, mean, invstd, bnpart, dA_bf16);
  else return VV_ERR_UNSUPPORTED;
  VV_CHECK_LAUNCH();
  return VV_OK;
}

extern "C" int vv_outconv_bwd_reduce(int32_t G, int32_t C, int32_t nblk, const float* partial, const int32_t* oc,
                                     float* dW, float* db, int64_t grad_gstride, vv_stream stream) {
  if (!partial || !oc || !dW || !db) return VV_ERR_BAD_ARG;
  if (C == 32) VV_LAUNCH(outconv_bwd_reduce_kernel<32>, dim3(G), dim3(1024), 0, (hipStream_t)stream, C, nblk, partial, oc, dW, db, grad_gstride);
  else if (C == 64) VV_LAUNCH(outconv_bwd_reduce_kernel<64>, dim3(G), dim3(1024), 0, (hipStream_t)stream, C, nblk, partial, oc, dW, db, grad_gstride);
  else return VV_ERR_UNSUPPORTED;
  VV_CHECK_LAUNCH();
  return VV_OK;
}

extern "C" int vv_bias_grad(int32_t G, int64_t M, int32_t C, const float* dy, int64_t dy_gstride, int32_t cstride,
                            int32_t coff, float* scratch, float* db, int64_t grad_gstride, vv_stream stream) {
  if (!dy || !scratch || !db) return VV_ERR_BAD_ARG;
  if (C > VV_WG || C % 4 || VV_WG % (C / 4) || cstride % 4 || coff % 4) return VV_ERR_UNSUPPORTED;
  const int nblk = (int)((M + 1023) / 1024);
  VV_LAUNCH(bias_grad_stage1, dim3(nblk, G), dim3(VV_WG), 0, (hipStream_t)stream, M, C, dy, dy_gstride, cstride,
                     coff, scratch, nblk);
  VV_CHECK_LAUNCH();
  VV_LAUNCH(bias_grad_stage2, dim3(G), dim3(VV_WG), 0, (hipStream_t)stream, C, nblk, scratch, db, grad_gstride);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

extern "C" int vv_bias_from_partials(int32_t G, int32_t C, int32_t ntiles, int32_t coff, int32_t n, const float* partial,
                                     int64_t partial_gstride, float* db, int64_t grad_gstride, vv_stream stream) {
  if (!partial || !db || n <= 0 || coff < 0 || coff + n > C || ntiles <= 0) return VV_ERR_BAD_ARG;
  VV_LAUNCH(bias_from_partials_kernel, dim3((n + 31) / 32, G), dim3(32 * VV_NP), 0, (hipStream_t)stream, C, ntiles, coff, n, partial,
            partial_gstride, db, grad_gstride);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

extern "C" int vv_adam(int64_t n, float* param, const float* grad, float* m, float* v, float lr, float beta1,
                       float beta2, float eps, float bias_corr1, float bias_corr2_sqrt, float grad_scale,
                       vv_stream stream) {
  if (!param || !grad || !m || !v || (n & 3)) return VV_ERR_BAD_ARG;
  const int64_t n4 = n >> 2;
  VV_LAUNCH(adam_kernel, dim3(nblocks(n4, 4096)), dim3(VV_WG), 0, (hipStream_t)stream, n4, (float4*)param,
                     (const float4*)grad, (float4*)m, (float4*)v, lr / bias_corr1, beta1, beta2, eps, bias_corr2_sqrt,
                     grad_scale);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

namespace {
__global__ void counter_add_kernel(int64_t* __restrict__ p, const int n, const int64_t inc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] += inc;
}
}  // namespace

extern "C" int vv_counter_add(int64_t* counters, int32_t n, int64_t inc, vv_stream stream) {
  if (!counters || n <= 0) return VV_ERR_BAD_ARG;
  VV_LAUNCH(counter_add_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, counters, n, inc);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

extern "C" int vv_adam_tick(int64_t* t_dev, float lr, double beta1, double beta2, float* sc_dev, vv_stream stream) {
  if (!t_dev || !sc_dev) return VV_ERR_BAD_ARG;
  VV_LAUNCH(adam_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t_dev, lr, beta1, beta2, sc_dev);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

extern "C" int vv_adam_bucketed(int32_t G, int64_t U, int32_t nb, const int64_t* bounds, float* param, const float* grad,
                                float* m, float* v, const float* sc_dev, float beta1, float beta2, float eps, float grad_scale,
                                vv_stream stream) {
  if (!param || !grad || !m || !v || !sc_dev || !bounds || (U & 3) || nb < 1 || nb > 8) return VV_ERR_BAD_ARG;
  adam_buckets bk;
  bk.nb = nb;
  for (int k = 0; k <= nb; ++k) {
    if ((bounds[k] & 3) || (k && bounds[k] <= bounds[k - 1])) return VV_ERR_BAD_ARG;
    bk.bound4[k] = bounds[k] >> 2;
  }
  if (bounds[0] != 0 || bounds[nb] != U) return VV_ERR_BAD_ARG;
  for (int k = nb + 1; k < 9; ++k) bk.bound4[k] = bk.bound4[nb];
  const int64_t n4 = (int64_t)G * (U >> 2);
  VV_LAUNCH(adam_bucketed_kernel, dim3(nblocks(n4, 4096)), dim3(VV_WG), 0, (hipStream_t)stream, G, U >> 2, bk, (float4*)param,
            (const float4*)grad, (float4*)m, (float4*)v, sc_dev, beta1, beta2, eps, grad_scale);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

extern "C" int vv_cube_gather(int32_t B, int32_t T, int32_t Tf, int32_t HW, const int64_t* idx, const uint8_t* raw,
                              const float* flow, float* x, float* xof, vv_stream stream) {
  if ((!raw && !flow) || (raw && !x) || (flow && !xof)) return VV_ERR_BAD_ARG;
  VV_LAUNCH(cube_gather_kernel, dim3(nblocks((int64_t)B * HW)), dim3(VV_WG), 0, (hipStream_t)stream, B, T, Tf,
                     HW, idx, raw, flow, x, xof);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

extern "C" int vv_nchw_to_nhwc(int32_t B, int32_t C, int32_t HW, const float* src, float* dst, vv_stream stream) {
  if (!src || !dst) return VV_ERR_BAD_ARG;
  VV_LAUNCH(nchw_to_nhwc_kernel, dim3(nblocks((int64_t)B * HW)), dim3(VV_WG), 0, (hipStream_t)stream, B, C, HW,
                     src, dst);
  VV_CHECK_LAUNCH();
  return VV_OK;
}
extern "C" int vv_out4_to_nchw(int32_t B, int32_t HW, int32_t oc, const float* out4, float* dst, int32_t Ctot,
                               int32_t choff, vv_stream stream) {
  if (!out4 || !dst || oc > 4) return VV_ERR_BAD_ARG;
  VV_LAUNCH(out4_to_nchw_kernel, dim3(nblocks((int64_t)B * HW)), dim3(VV_WG), 0, (hipStream_t)stream, B, HW, oc,
                     out4, dst, Ctot, choff);
  VV_CHECK_LAUNCH();
  return VV_OK;
}
extern "C" int vv_nchw_to_out4(int32_t B, int32_t HW, int32_t oc, const float* src, int32_t Ctot, int32_t choff,
                               float* out4, vv_stream stream) {
  if (!out4 || !src || oc > 4) return VV_ERR_BAD_ARG;
  VV_LAUNCH(nchw_to_out4_kernel, dim3(nblocks((int64_t)B * HW)), dim3(VV_WG), 0, (hipStream_t)stream, B, HW, oc,
                     src, Ctot, choff, out4);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

extern "C" int vv_pool_act(int32_t G, int32_t B, int32_t H2, int32_t W2, int32_t C, const float* y, int64_t y_gstride,
                           const float* a, const float* b, int64_t ab_gstride, float* out, int64_t out_gstride,
                           int32_t io_bf16, vv_stream stream) {
  if (!y || !a || !b || !out || C % 4) return VV_ERR_BAD_ARG;
  const int64_t n4 = (int64_t)B * H2 * W2 * C / 4;
  VV_LAUNCH(pool_act_kernel, dim3(nblocks(n4, 4096), G), dim3(VV_WG), 0, (hipStream_t)stream, n4, C, H2, W2, y, y_gstride, a,
            b, ab_gstride, out, out_gstride, io_bf16);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

extern "C" int vv_cube_erase(int32_t G, int64_t npix, int32_t Cc, int32_t CP, const float* cube, const int32_t* chmap,
                             float* out, int64_t out_gstride, int32_t out_bf16, vv_stream stream) {
  if (!cube || !chmap || !out || CP % 4 || Cc > 16) return VV_ERR_BAD_ARG;      // (15 channels: 5 frames x RGB, vad_datasets.py:159)
  VV_LAUNCH(cube_erase_kernel, dim3(nblocks(npix * (CP / 4))), dim3(VV_WG), 0, (hipStream_t)stream, G, npix, Cc, CP, cube, chmap, out,
            out_gstride, out_bf16);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

// ------------------------------------------------------------------------------------------------ eval mode: fold BatchNorm
__global__ void __launch_bounds__(VV_WG)
fold_bn_kernel(const vv_fold_entry* __restrict__ table, const float* __restrict__ params, const int64_t pg,
               const float* __restrict__ bufs, const int64_t bg, const float eps, float* __restrict__ folded, const int64_t fg) {
  const vv_fold_entry e = table[blockIdx.y];
  const int g = blockIdx.z;
  const float* P = params + (int64_t)g * pg;
  const float* Bf = bufs + (int64_t)g * bg;
  float* F = folded + (int64_t)g * fg;
  const int64_t n = (int64_t)e.cout * e.row;
  for (int64_t i = (int64_t)blockIdx.x * VV_WG + threadIdx.x; i < n; i += (int64_t)gridDim.x * VV_WG) {
    const int c = (int)(i / e.row);
    const double a = (double)P[e.g_off + c] / sqrt((double)Bf[e.rv_off + c] + (double)eps);
    F[e.w_off + i] = (float)a * P[e.w_off + i];
    if (i < e.cout) {            // thread i < cout also writes bias i
      const double ai = (double)P[e.g_off + i] / sqrt((double)Bf[e.rv_off + i] + (double)eps);
      F[e.b_off + i] = (float)(ai * ((double)P[e.b_off + i] - (double)Bf[e.rm_off + i]) + (double)P[e.beta_off + i]);
    }
  }
}

extern "C" int vv_fold_bn(const vv_fold_entry* table_dev, int32_t nentries, int32_t G, const float* params, int64_t params_gstride,
                          const float* bufs, int64_t bufs_gstride, float eps, float* folded, int64_t folded_gstride,
                          vv_stream stream) {
  if (!table_dev || !params || !bufs || !folded || nentries <= 0 || G <= 0) return VV_ERR_BAD_ARG;
  VV_LAUNCH(fold_bn_kernel, dim3(64, nentries, G), dim3(VV_WG), 0, (hipStream_t)stream, table_dev, params, params_gstride, bufs,
            bufs_gstride, eps, folded, folded_gstride);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

extern "C" const char* vv_version(void) { return "vecvad_hip 0.1 (gfx950)"; }

extern "C" int vv_abi_sizeof(int32_t which) {
  switch (which) {
    case 0: return (int)sizeof(vv_view);
    case 1: return (int)sizeof(vv_conv_params);
    case 2: return (int)sizeof(vv_wgrad_params);
    case 3: return (int)sizeof(vv_pack_entry);
    case 4: return (int)sizeof(vv_reduce_entry);
    case 5: return (int)sizeof(vv_fold_entry);
    case 6: return (int)sizeof(vv_bnbwd_params);
    case 7: return (int)sizeof(vv_outconv_params);
    case 8: return (int)sizeof(vv_conv2d_params);
  }
  return -1;
}

extern "C" const char* vv_status_string(int status) {
  switch (status & 0xff) {
    case VV_OK: return "VV_OK";
    case VV_ERR_BAD_ARG: return "VV_ERR_BAD_ARG";
    case VV_ERR_UNSUPPORTED: return "VV_ERR_UNSUPPORTED";
    case VV_ERR_LAUNCH: return (status >> 8) ? hipGetErrorString((hipError_t)(status >> 8)) : "VV_ERR_LAUNCH";
    default: return "unknown vv_status";
  }
}

extern "C" int vv_num_cus(void) {
  static int cached[16] = {0};           // per device ordinal; hipDeviceGetAttribute is legal during stream capture
  int dev = 0, ncu = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (dev >= 0 && dev < 16 && cached[dev] > 0) return cached[dev];
  if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || ncu <= 0) return 256;
  if (dev >= 0 && dev < 16) cached[dev] = ncu;
  return ncu;
}

extern "C" int vv_device_arch_ok(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  const char* a = prop.gcnArchName;
  return (a[0] == 'g' && a[1] == 'f' && a[2] == 'x' && a[3] == '9' && a[4] == '5' && a[5] == '0') ? 1 : 0;
}
