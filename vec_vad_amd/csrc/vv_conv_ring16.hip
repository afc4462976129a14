// 3x3 convolution of the all-bf16 bank on the 32x32 level (BASELINE config 4; model/unet.py:4-20,410-556: inc, up3.conv and their data
// gradients) for layers with at most 32 input channels: persistent workgroups, the halo tile of the NEXT pixel tile in flight by
// LDS-DMA while the current one is multiplied, the whole filter resident in registers.
//
// Why: these launches move 0.5 - 1 GB for 2 - 4 us of matrix work per CU and pixel tile; conv_mfma_kernel<.., BF> keeps one 11 KB
// chunk per workgroup in flight (registers -> LDS, three workgroups per CU: ~26 KB per CU = 3.4 TB/s by Little's law, round 4)
// and lives ~2 us of HBM latency before its first MFMA.  Here (the pattern of wino_ring_kernel, vv_wino.hip):
//   * a workgroup walks a contiguous run of 256-pixel tiles of one (UNet, N tile); the 18 (9 taps x 2 K steps) filter fragments of
//     that (UNet, N tile) stay in registers for the run -- no register-destination load in the tile loop;
//   * the halo tile (10 x 34 pixels x CinP bf16 = 21.8 KB) of tile t + 1 goes global -> LDS by DMA (buffer_load_dwordx4 ... lds, no
//     staging registers) right after the barrier that opens tile t; counted s_waitcnt vmcnt (the output stores of tile t are
//     younger and stay outstanding) + raw s_barrier;
//   * the producing layer's BatchNorm + ReLU in place in LDS by the lane that transferred the 16 bytes (vv_act4 on the unpacked
//     values, rounded back with v_cvt_pk_bf16_f32: the same values conv_mfma_kernel commits);
//   * two halo buffers; the output tile of the epilogue (bf16, 16-byte items per lane like conv_mfma_kernel's) aliases the buffer
//     the finished tile was read from -- 49 KB of LDS per workgroup.
// LDS image: lane-linear as the DMA needs it, slot = (halo pixel) * PPX + piece, 16 bytes = 8 channels per piece; the piece index
// is XOR-swizzled ON THE SOURCE ADDRESS with bits of the halo column (f = (hx >> 2) & 3 for four pieces per pixel, (hx >> 3) & 1 for
// two) so that the sixteen lanes ds_read_b128 services together -- sixteen pixels of a row, one piece -- hit sixteen different
// bank groups for all three column shifts (checked exhaustively).
// Arithmetic and its order are conv_mfma_kernel<8, 32, 1, 1, VV_CONV3, 16, true, 2 | 3>'s: chunk of 16 channels -> tap -> MFMA per
// 32-pixel row block, the same epilogue (bias, round to bf16, column sums of the ROUNDED values, fused BatchNorm-backward sums): the
// output, `stats` and `bn_partial` are bit-identical (tests/test_gpu_bf16.py::test_bf16_ring_conv_bitwise_equal).
#include "vv_common.h"

#ifndef VV_RING16_MIN
#define VV_RING16_MIN (4 * 512)      // work items (pixel tiles x N tiles x UNets) from which this kernel takes a launch
#endif

namespace {

__device__ __forceinline__ void vv_lds_barrier16() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// KS: 16-channel K steps (CinP / 16: 1 | 2).  BNF: vv_conv_params.bn_partial (the BatchNorm-backward sums of the layer whose
// activation gradient this launch produces).
#ifndef VV_RING16_NBUF
#define VV_RING16_NBUF 2             // halo buffers of 24 KB (measured: 3 buffers = 2 tiles ahead is no faster; 4 = one workgroup per CU, 1.3x slower)
#endif
#ifndef VV_RING16_OCC
#define VV_RING16_OCC 2              // workgroups per CU the register budget is set for (3: 168 registers, spills with K = 32)
#endif
template <int KS, bool BNF>
__global__ void __launch_bounds__(VV_WG, VV_RING16_OCC)
conv_ring16_kernel(const vv_conv_params p, const int NT, const int NN, const int total, const int ipw) {
  constexpr int H_ = 32, TH = 8, PARTS = 4, HHT = TH + 2, HW = 34;
  constexpr int PPX = 2 * KS;                               // 16-byte pieces per pixel
  constexpr int NSLOT = HHT * HW * PPX;                     // 680 | 1360
  constexpr int NDMA = (NSLOT + 255) / 256;                 // DMA instructions per wave and tile: 3 | 6
  constexpr int BUF16 = NDMA * 256;                         // 16-byte slots per halo buffer: 768 | 1536
  constexpr int TN = 32, ORS = TN + 8;                      // output tile row: 32 channels + 16 B of padding, in bf16 elements
  constexpr int OUT16 = (256 * ORS * 2 + 15) / 16;          // 1280 slots (20 KB): aliases a halo buffer ...
  constexpr int BUFA = BUF16 > OUT16 ? BUF16 : OUT16;       // ... so a buffer is at least that large
  constexpr int NBUF = VV_RING16_NBUF;                      // halo buffers: tile t is multiplied while tiles t+1 .. t+NBUF-1 are in flight
  constexpr int SP16 = NBUF * BUFA;                         // [2][4][32] floats: column sums of the four waves
  constexpr int BN16 = SP16 + 64;                           // BNF: [2][4][64] floats: the four waves' BatchNorm-backward sums
  constexpr int AB16 = BN16 + (BNF ? 128 : 0);              // [a | b][32] floats of the producing layer's BatchNorm (activated source)
  constexpr int TB16 = AB16 + 16;                           // BNF: [a | b | mean | invstd][32] floats of the layer the sums belong to
  constexpr int STORES_MIN = 4;                             // output stores every wave issues per tile
  constexpr int VPT = STORES_MIN + (BNF ? 4 : 0);           // VMEM instructions besides the DMAs every wave issues per tile, at least
  constexpr int VMW = (NBUF - 1) * VPT + (NBUF - 2) * NDMA; // ... and how many of everything are younger than a tile's DMAs when they must have landed
  static_assert(NBUF >= 2 && VMW <= 63, "ring geometry");
  constexpr int QN = TN / 8, NOUT = 256 * QN / VV_WG;       // 16-byte output items: 4 per pixel, 4 per thread
  __shared__ uint4 lds16[TB16 + (BNF ? 32 : 0)];            // the ONLY __shared__ object (<= 51 KB: three workgroups per CU)
  float* ldsf = reinterpret_cast<float*>(lds16);
  const v4f* ldsA = reinterpret_cast<const v4f*>(lds16);
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds16;

  const int w_begin = blockIdx.x * ipw;
  const int w_end = w_begin + ipw < total ? w_begin + ipw : total;
  if (w_begin >= w_end) return;

  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int Cout = p.Cout;
  const bool act_mode = p.in_mode == VV_IN_ACT;
  const int cs = p.src0.cstride;                            // bf16 elements between pixels of the source

  // ---- this lane's DMA items (k = 0 .. NDMA-1): slot -> (halo row, halo column, piece position); the piece of the source it
  //      holds is piece position ^ f(halo column)
  int rel[NDMA];               // byte offset inside the halo tile (relative to its pixel (0, 0) = image pixel (y0, -1))
  int yx[NDMA];                // halo row | column-in-image flag << 8 | slot-in-use flag << 9 | piece << 10
#pragma unroll
  for (int k = 0; k < NDMA; ++k) {
    const int sl = (wave * NDMA + k) * 64 + lane;
    const int hp = sl / PPX, jp = sl % PPX;
    const int hy = hp / HW, hx = hp % HW;
    const int f = PPX == 4 ? (hx >> 2) & 3 : (hx >> 3) & 1;
    const int j = jp ^ f;
    const bool used = sl < NSLOT;
    const bool xin = used && (unsigned)(hx - 1) < (unsigned)H_;
    rel[k] = ((hy * H_ + hx - 1) * cs + j * 8) * 2;
    yx[k] = hy | (xin ? 256 : 0) | (used ? 512 : 0) | (j << 10);
  }

  auto decode = [&](const int w, int& g, int& nn, int& pt) {
    pt = w % NT;
    const int t = w / NT;
    nn = t % NN;
    g = t / NN;
  };
  auto advance = [&](int& g, int& nn, int& pt) {
    if (++pt == NT) {
      pt = 0;
      if (++nn == NN) { nn = 0; ++g; }
    }
  };
  int gc, nc, ptc;
  decode(w_begin, gc, nc, ptc);
  int gd = gc, nd = nc, ptd = ptc, wd = w_begin;           // DMA cursor: one tile ahead
  int bufd = 0;                                            // halo buffer the next DMA set fills
  auto dma_tile = [&]() {
    const bool live = wd < w_end;
    const int img = ptd >> 2, part = ptd & 3;
    const int y0 = part * TH - 1;
    const int tileoff = ((img * H_ + y0) * H_) * cs * 2;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.src0.ptr + (int64_t)(live ? gd : gc) * p.src0.gstride), 0, 0x7FFFFFFF, 0x00020000);
    const int co2 = p.src0.coff * 2;
#pragma unroll
    for (int k = 0; k < NDMA; ++k) {
      const int y = y0 + (yx[k] & 255);
      const bool ok = live && (yx[k] & 256) && (unsigned)y < (unsigned)H_;
      const unsigned voff = ok ? (unsigned)(tileoff + rel[k] + co2) : 0x80000000u;
      const unsigned dst = lds_base + (unsigned)((bufd * BUFA + (wave * NDMA + k) * 64) * 16);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(dst), "v"(voff), "s"(rs) : "memory", "m0");
    }
    bufd = bufd + 1 == NBUF ? 0 : bufd + 1;
    advance(gd, nd, ptd);
    ++wd;
  };
#pragma unroll
  for (int i = 0; i < NBUF - 1; ++i) dma_tile();           // the first NBUF - 1 tiles

  // ---- this lane's rows in the tile: wave w multiplies rows 2w, 2w+1 (two 32-pixel row blocks), lane = column
  int abase[2];                // 16-byte slot of (row 2w + m + dy = 0, column l31 + dx = 0), piece position for piece 0
#pragma unroll
  for (int m = 0; m < 2; ++m) abase[m] = ((2 * wave + m) * HW + l31) * PPX;
  auto rdA = [&](const int buf, const int m, const int tap, const int ks) -> v4f {
    const int dy = tap / 3, dx = tap % 3;
    const int hx = l31 + dx;
    const int f = PPX == 4 ? (hx >> 2) & 3 : (hx >> 3) & 1;
    return ldsA[buf * BUFA + abase[m] + (dy * HW + dx) * PPX + ((ks * 2 + half) ^ f)];
  };

  v4f fb[9][KS];
  float bias = 0.f;
  int g_have = -1, n_have = -1;
  const int ocs = p.out.cstride;
  int bufc = 0;
  unsigned short* lo16 = reinterpret_cast<unsigned short*>(lds16);
  float* sp = ldsf + SP16 * 4;

  for (int w = w_begin; w < w_end; ++w) {
    const int co0 = nc * TN;
    if (gc != g_have || nc != n_have) {
      // ---- a new (UNet, N tile): filter fragments -> registers, bias, BatchNorm tables -> LDS (rare; drains the DMAs once)
      const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(p.w + (int64_t)gc * p.w_gstride), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          fb[tap][ks] = __builtin_amdgcn_raw_buffer_load_b128(rsW, (unsigned)((((tap * KS + ks) * 2 + half) * Cout + co0 + l31) * 16), 0, 0);
      bias = p.bias ? p.bias[(int64_t)gc * p.bias_gstride + co0 + l31] : 0.f;
      if (act_mode && gc != g_have && tid < 2 * 16 * KS / 4) {        // a, b of the source's 16 KS channels as float4
        const int n4 = 16 * KS / 4;
        const float* src = (tid < n4 ? p.a : p.b) + (int64_t)gc * p.ab_gstride + (tid % n4) * 4;
        reinterpret_cast<float4*>(lds16)[AB16 + (tid < n4 ? 0 : 8) + tid % n4] = *reinterpret_cast<const float4*>(src);
      }
      if constexpr (BNF) {
        if (tid < 128) {
          const float* tb = tid < 32 ? p.bn_a : (tid < 64 ? p.bn_b : (tid < 96 ? p.bn_mean : p.bn_invstd));
          ldsf[TB16 * 4 + tid] = tb[(int64_t)gc * p.bn_gstride + co0 + (tid & 31)];
        }
      }
      __builtin_amdgcn_s_waitcnt(0);             // (the builtin: the compiler's own load scoreboard must see it)
      vv_lds_barrier16();
      g_have = gc; n_have = nc;
    }
    const int img = ptc >> 2, part = ptc & 3;
    const int ty0 = part * TH;

    // ---- this tile's halo has landed (this wave's pieces); activate them in place; meet the other waves
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VMW) : "memory");
    if (act_mode) {
#pragma unroll
      for (int k = 0; k < NDMA; ++k) {
        const int y = ty0 - 1 + (yx[k] & 255);
        if ((yx[k] & 256) && (unsigned)y < (unsigned)H_) {
          const int sl = bufc * BUFA + (wave * NDMA + k) * 64 + lane;
          const int j = yx[k] >> 10;
          const float4* ab = reinterpret_cast<const float4*>(lds16) + AB16;
          const float4 sa = ab[2 * j], sa2 = ab[2 * j + 1], sb = ab[8 + 2 * j], sb2 = ab[8 + 2 * j + 1];
          uint4 h = lds16[sl];
          const uint2 lo = vv_pack_bf16x4(vv_act4(vv_unpack_bf16x4(make_uint2(h.x, h.y)), sa, sb));
          const uint2 hi = vv_pack_bf16x4(vv_act4(vv_unpack_bf16x4(make_uint2(h.z, h.w)), sa2, sb2));
          lds16[sl] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
      }
    }
    vv_lds_barrier16();
    // ---- every wave is past the epilogue of the tile before: its buffer takes the tile NBUF - 1 ahead
    dma_tile();

    // BNF: this thread's z items (the pixels and 8 channels it will store), requested ahead of the MFMA phase
    uint4 zq[BNF ? NOUT : 1];
    if constexpr (BNF) {
      const unsigned short* zh = reinterpret_cast<const unsigned short*>(p.bn_z + (int64_t)gc * p.bn_z_gstride) + co0 + (tid % QN) * 8;
#pragma unroll
      for (int k = 0; k < NOUT; ++k) {
        const int pp = (tid + k * VV_WG) / QN;
        zq[k] = *reinterpret_cast<const uint4*>(zh + ((int64_t)(img * H_ + ty0 + pp / 32) * H_ + pp % 32) * Cout);
      }
    }

    // ---- 2 row blocks x 9 taps x KS steps, conv_mfma_kernel's order (chunk of 16 channels -> tap -> row block)
    v16f acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
    {
      constexpr int NIT = 9 * KS;
      v4f fa[2][2];
#pragma unroll
      for (int m = 0; m < 2; ++m) fa[0][m] = rdA(bufc, m, 0, 0);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int ks = it / 9, tap = it % 9;
        const int cur = it & 1, nxt = cur ^ 1;
        if (it + 1 < NIT) {
#pragma unroll
          for (int m = 0; m < 2; ++m) fa[nxt][m] = rdA(bufc, m, (it + 1) % 9, (it + 1) / 9);
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, fa[cur][m]), __builtin_bit_cast(v8bf, fb[tap][ks]),
                                                           acc[m], 0, 0, 0);
      }
    }

    // ---- epilogue (conv_mfma_kernel's bf16 output-tile form): bias, round, the tile through LDS, 16-byte stores, column sums
    vv_lds_barrier16();                                     // every wave is done reading this tile's halo: its buffer is the output tile
    unsigned short* lo = lo16 + bufc * BUFA * 8;
    const float relu_lo = (p.pad0 & VV_CONV_RELU) ? 0.f : -__builtin_inff();
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * half;
        const int pp = wave * 64 + m * 32 + row;
        float v = acc[m][i] + bias;
        v = v < relu_lo ? relu_lo : v;
        const __bf16 hv = (__bf16)v;
        lo[pp * ORS + l31] = __builtin_bit_cast(unsigned short, hv);
        v = (float)hv;
        s1 += v; s2 = fmaf(v, v, s2);
      }
    vv_lds_barrier16();
    __bf16* outh = reinterpret_cast<__bf16*>(p.out.ptr + (int64_t)gc * p.out.gstride) + p.out.coff;
    __bf16* obase = outh + co0 + (tid % QN) * 8;
    if (p.out1.ptr && co0 + (tid % QN) * 8 >= p.osplit)
      obase = reinterpret_cast<__bf16*>(p.out1.ptr + (int64_t)gc * p.out1.gstride) + p.out1.coff + (co0 + (tid % QN) * 8 - p.osplit);
    float bna[8], bnb[8], bnm[8], bni[8], bs1[8], bs2[8];
    if constexpr (BNF) {
      const float* tb = ldsf + TB16 * 4 + (tid % QN) * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) { bna[j] = tb[j]; bnb[j] = tb[32 + j]; bnm[j] = tb[64 + j]; bni[j] = tb[96 + j]; bs1[j] = 0.f; bs2[j] = 0.f; }
    }
#pragma unroll
    for (int k = 0; k < NOUT; ++k) {
      const int it = tid + k * VV_WG;
      const int pp = it / QN;
      const uint4 v = *reinterpret_cast<const uint4*>(lo + pp * ORS + (tid % QN) * 8);
      *reinterpret_cast<uint4*>(obase + ((int64_t)(img * H_ + ty0 + pp / 32) * H_ + pp % 32) * ocs) = v;
      if constexpr (BNF) {
        const unsigned dw[4] = {v.x, v.y, v.z, v.w}, zw[4] = {zq[k].x, zq[k].y, zq[k].z, zq[k].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const v2f z2 = {__builtin_bit_cast(float, zw[j] << 16), __builtin_bit_cast(float, zw[j] & 0xFFFF0000u)};
          const v2f d2 = {__builtin_bit_cast(float, dw[j] << 16), __builtin_bit_cast(float, dw[j] & 0xFFFF0000u)};
          const v2f a2 = {bna[2 * j], bna[2 * j + 1]}, b2 = {bnb[2 * j], bnb[2 * j + 1]};
          const v2f on = __builtin_elementwise_fma(a2, z2, b2);
          const v2f dj = {on.x > 0.f ? d2.x : 0.f, on.y > 0.f ? d2.y : 0.f};
          v2f t1 = {bs1[2 * j], bs1[2 * j + 1]}, t2 = {bs2[2 * j], bs2[2 * j + 1]};
          t1 += dj;
          t2 = __builtin_elementwise_fma(dj, z2, t2);
          bs1[2 * j] = t1.x; bs1[2 * j + 1] = t1.y; bs2[2 * j] = t2.x; bs2[2 * j + 1] = t2.y;
        }
      }
    }
    if constexpr (BNF) {
      // the 64 threads with this channel group: 16 lanes of each wave (lane % QN), then the four waves through LDS
      float* bl = ldsf + BN16 * 4;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        bs2[j] = bni[j] * (bs2[j] - bnm[j] * bs1[j]);
        bs1[j] += vv_dpp_ror<4>(bs1[j]); bs2[j] += vv_dpp_ror<4>(bs2[j]);
        bs1[j] += vv_dpp_ror<8>(bs1[j]); bs2[j] += vv_dpp_ror<8>(bs2[j]);
        bs1[j] += __shfl_xor(bs1[j], 16); bs2[j] += __shfl_xor(bs2[j], 16);
        bs1[j] += __shfl_xor(bs1[j], 32); bs2[j] += __shfl_xor(bs2[j], 32);
      }
      if (lane < QN) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { bl[wave * 2 * TN + lane * 8 + j] = bs1[j]; bl[wave * 2 * TN + TN + lane * 8 + j] = bs2[j]; }
      }
      vv_lds_barrier16();
      if (tid < 2 * TN) {
        const float t = (bl[tid] + bl[2 * TN + tid]) + (bl[4 * TN + tid] + bl[6 * TN + tid]);
        p.bn_partial[((int64_t)(gc * NT + ptc) * 2) * Cout + (tid / TN) * Cout + co0 + tid % TN] = t;
      }
    }
    if (p.stats) {
      s1 += __shfl_xor(s1, 32);
      s2 += __shfl_xor(s2, 32);
      if (half == 0) {
        sp[wave * 32 + l31] = s1;
        sp[4 * TN + wave * 32 + l31] = s2;
      }
      vv_lds_barrier16();
      if (tid < TN) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) {
          t1 += sp[wv * 32 + tid];
          t2 += sp[4 * TN + wv * 32 + tid];
        }
        float* st = p.stats + ((int64_t)(gc * NT + ptc) * 2) * Cout + co0 + tid;
        st[0] = t1;
        st[Cout] = t2;
      }
    }
    bufc = bufc + 1 == NBUF ? 0 : bufc + 1;
    advance(gc, nc, ptc);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the tail's dummy DMAs (zeros into a halo buffer) have landed
}

template <int KS>
int launch_ring16(const vv_conv_params* p, hipStream_t st) {
  const int NT = p->B * 4;
  const int NN = p->Cout / 32;
  const int total = p->G * NN * NT;
  const int slots = VV_RING16_OCC * vv_num_cus();
  const int ipw = (total + slots - 1) / slots;
  const int nwg = (total + ipw - 1) / ipw;
  if (p->bn_partial)
    VV_LAUNCH((conv_ring16_kernel<KS, true>), dim3(nwg), dim3(VV_WG), 0, st, *p, NT, NN, total, ipw);
  else
    VV_LAUNCH((conv_ring16_kernel<KS, false>), dim3(nwg), dim3(VV_WG), 0, st, *p, NT, NN, total, ipw);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

}  // namespace

// the launches this kernel takes from vv_conv_mfma (which has validated the flags / views): all-bf16 3x3 on the 32x32 level, K <= 32,
// one plain or activated source, 16-byte aligned channel offsets
bool vv_conv_ring16_ok(const vv_conv_params* p) {
  if (p->kind != VV_CONV3 || p->H != 32 || p->W != 32 || (p->CinP != 16 && p->CinP != 32) || p->Cout % 32) return false;
#ifndef VV_RING16_N64
  if (p->Cout % 64 == 0) return false;      // (two N tiles would stage every halo twice: 32 -> 64 measured 318 -> 340 us; conv_mfma_kernel's 64-wide tile keeps it)
#endif
  if (p->pad0 & VV_CONV_NO_RING) return false;
  if (!(p->pad0 & VV_CONV_BF16) || !(p->pad0 & VV_CONV_OUT_BF16) || !(p->pad0 & VV_CONV_ALLSRC_BF16)) return false;
  if (p->in_mode != VV_IN_PLAIN && p->in_mode != VV_IN_ACT) return false;
  if (p->in_mode == VV_IN_ACT && (!p->a || !p->b)) return false;
  if (p->src0.cstride % 8 || p->src0.coff % 8 || p->out.cstride % 8 || p->out.coff % 8) return false;
  if (p->out1.ptr && (p->osplit % 32 || p->out1.coff % 8)) return false;
  if (p->bn_partial && (p->Cout % 64 == 0 || p->stats)) return false;
  if ((int64_t)p->B * 32 * 32 * (p->src0.cstride > p->out.cstride ? p->src0.cstride : p->out.cstride) * 2 >= (1ll << 31)) return false;
  return (int64_t)p->G * (p->Cout / 32) * p->B * 4 >= VV_RING16_MIN;
}

int vv_conv_ring16(const vv_conv_params* p, hipStream_t st) {
  return p->CinP == 16 ? launch_ring16<1>(p, st) : launch_ring16<2>(p, st);
}
