// GEMM-shaped 3x3 convolution for the mixed-precision bank (BASELINE config 4: SelfCompleteNetFull, B = 512, bf16 tensors):
// nn.Conv2d(k3,p1) forward (model/unet.py:10,13) and its data gradient when EVERY tensor of the launch is bf16
// (VV_CONV_BF16 | VV_CONV_OUT_BF16 | VV_CONV_ALLSRC_BF16 or VV_CONV_SRC_BF16), v_mfma_f32_32x32x16_bf16, fp32 accumulation.
//
// Why a second kernel (round 3 measured conv_mfma_kernel<..., BF = true> at 0.29 of the HBM roof AND 0.27 of the bf16 matrix peak,
// with HBM traffic 1.03x algorithmic): that kernel moves every operand through LDS -- per 16-channel chunk a workgroup writes the
// chunk's 18 KB filter panel and its activation tile into LDS and every MFMA reads 1.0 - 1.5 KB fragments back -- and a CU's LDS
// (256 B / clk) cannot feed four matrix pipes that retire a 32x32x16 instruction every 32 cycles that way.  Here:
//   * workgroup tile = 256 pixels x TN (32 | 64 | 128) output channels, 4 waves as WM x WN, wave tile = MR x NR 32x32 blocks
//     (4 x 2 = 128 accumulator registers for TN = 128): per K step a wave reads MR activation fragments from LDS and NR filter
//     fragments for MR * NR MFMAs = 0.5 KB of LDS reads per MFMA instead of 1.5;
//   * the filter NEVER touches LDS: a B fragment of the packed panel [tap][Cin/16][2][Cout][8] is 2 x 512 contiguous bytes, loaded
//     by each wave straight from the L2-resident panel (buffer_load_b128, wave-uniform part of the address in an SGPR) two K steps
//     ahead into a three-deep register ring; FLOP per filter byte = 256 pixels, i.e. <= 16 B / clk / CU of L2 traffic at full rate;
//   * the activation halo tile is double buffered (ONE barrier per 16-channel chunk = per 9 K steps = per 72 MFMAs of a wave): the
//     global loads of chunk c+1 are issued before the MFMA loop of chunk c, the producer's BatchNorm+ReLU is applied in
//     registers on the way into the other buffer after it;
//   * LDS layout: two planes (channels 0-7 | 8-15 of the chunk = the two lane halves of an A fragment), 16 B per pixel and plane,
//     rows padded to HWP pixels, and the 32 rows of an A fragment are assigned to pixels by a per-geometry permutation so that the
//     16 lanes the hardware services together (ds_read_b128: {0-3,12-15,20-27}, {4-11,16-19,28-31}) always hit 16 different
//     16-byte bank groups for every tap shift -- conflict-free on all four pyramid levels without per-pixel padding (the 48-byte
//     pixel stride of the old kernel was conflict-free at 32x32 only);
//   * two workgroups per CU (<= 256 registers, <= 75 KB LDS): the second one's MFMAs cover the first one's commit / barrier / epilogue.
// Epilogue as in conv_mfma_kernel: bias, rounding to bf16, the tile leaves through LDS as 16-byte items, per-tile column sums /
// sums of squares of the STORED values for BatchNorm (or for the transposed conv's bias gradient on data-gradient launches).
// Tiles are the fp32 kernel's (vv_conv_ntiles): 8x32 / 16x16 / 4 images of 8x8 / 16 images of 4x4.
#include "vv_common.h"


namespace {

constexpr unsigned GRP_A = 0x0FF0F00Fu;        // lanes (of 32) that ds_read_b128 services in its first cycle pair

// row rho (0..31) of a 32-pixel block -> pixel of the block.  Lanes of hardware group A take one half of the block, group B the
// other, chosen per geometry so that a group's 16 pixels have 16 different (linear LDS pixel index mod 16):
//   TW >= 16 or TW == 4: first / second 16 pixels of the block (a row of 32, two rows of 16, two 4x4 images with HWP = 12)
//   TW == 8 (HWP = 12) : rows 0,2 / rows 1,3 of the block's four 8-pixel rows
template <int TW>
__device__ __forceinline__ int perm_row(const int rho) {
  const int grp = ((GRP_A >> rho) & 1u) ? 0 : 1;
  const unsigned mk = grp ? ~GRP_A : GRP_A;
  const int idx = __builtin_popcount(mk & ((1u << rho) - 1u));
  if constexpr (TW == 8) return (((idx >> 3) * 2 + grp) << 3) + (idx & 7);
  else return grp * 16 + idx;
}

template <int TH, int TW, int NI>
struct GGeo {
  static constexpr int HH = TH + 2, HW = TW + 2;
  static constexpr int HWP = (TW == 8 || TW == 4) ? 12 : HW;          // padded row length (pixels) in LDS
  static constexpr int NPIX = NI * HH * HW;                           // staged halo pixels
  static constexpr int RAW = NI * HH * HWP;                           // 16-byte slots of one plane
  static constexpr int PLANE = RAW + ((8 - RAW % 16) + 16) % 16;      // == 8 mod 16: the two planes' stores do not collide
  static constexpr int BUF = 2 * PLANE;                               // slots of one buffer (two planes)
};


typedef unsigned v4u_ __attribute__((ext_vector_type(4)));

// A thread group's share of a finished 256-pixel x TN tile on its way out of an LDS output region: NTHR threads own NPX pixels
// (producers: 256 threads, the whole tile; a consumer wave of the register-resident-filter variants: 64 threads, its own 64 pixels).
// Items are 16 bytes = 8 channels of one pixel; a thread's items all hold the SAME 8 channels (NTHR % QN == 0), so the column sums
// for BatchNorm are 16 per-thread accumulators that live ACROSS tiles and are reduced over the threads only when the workgroup
// moves to another UNet / N tile or ends (vv_conv_params.stats rows are summed over tiles by their readers: a tile's row holds
// either zeros or the sums of the run of tiles that ends with it).  Everything tile-independent (item offsets, LDS addresses) is
// computed once: a tile costs one ds_read_b128 + one buffer_store_b128 + 24 VALU per item.
template <int TH, int TW, int NI, int TN, int NTHR, int NPX>
struct OutStore {
  static constexpr int QN = TN / 8, NOUT = NPX * QN / NTHR, ORS = TN + 8;
  static_assert(NTHR % QN == 0 && (NPX * QN) % NTHR == 0, "item geometry");
  int soff[NOUT];          // byte offset of the item relative to the tile's first output element
  int sim[NOUT];           // image of the tile the item belongs to
  int lbase;               // byte offset of item 0 inside an output region
  float s1[8], s2[8];

  // xoff(first channel of the thread's items inside the N tile) -> extra byte offset of the thread's items (second output view)
  template <class XOff>
  __device__ __forceinline__ void init(const int tl, const int px0, const int H, const int W, const int ocs, const XOff xoff) {
    const int qq = tl % QN;
    const int xo = xoff(qq * 8);
#pragma unroll
    for (int k = 0; k < NOUT; ++k) {
      const int pp = px0 + (tl + k * NTHR) / QN;
      const int im = pp / (TH * TW), rr = (pp / TW) % TH, cc = pp % TW;
      soff[k] = (((im * H + rr) * W + cc) * ocs + qq * 8) * 2 + xo;
      sim[k] = im;
    }
    lbase = ((px0 + tl / QN) * ORS + qq * 8) * 2;
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
  }
  // items [K0, K1) of the tile in `region` (its 16-byte aligned LDS address): voffset = soff, soffset = the tile's first element
  template <int K0, int K1>
  __device__ __forceinline__ void store(const char* region, const __amdgpu_buffer_rsrc_t rsO, const int sbase, const int nimg, const bool stats) {
    const bool full = nimg >= NI;                                      // every image of the tile exists (wave-uniform)
#pragma unroll
    for (int k = K0; k < K1; ++k) {
      const uint4 v = *reinterpret_cast<const uint4*>(region + lbase + k * ((NTHR / QN) * ORS * 2));
      const bool ok = full || sim[k] < nimg;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u_, v), rsO, ok ? soff[k] : (int)0x80000000, sbase, 0);
      if (stats) {
        const vv_f8 f = vv_unpack_bf16x8(v);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = ok ? f.v[e] : 0.f;
          s1[e] += x; s2[e] = fmaf(x, x, s2[e]);
        }
      }
    }
  }
  // Sum the accumulators over the group's threads and the FOUR waves of the role (wave index w), write the row of the tile, reset.
  // ex: LDS exchange [2][4][TN] floats + one arrival counter behind it.  No workgroup barrier (the other role is elsewhere): the
  // last of the four waves to arrive adds up.
  template <int NW>
  __device__ __forceinline__ void flush(float* ex, const int w, const int lane, float* row, const int Cout) {
    auto ror = [](const float v, const auto N_) -> float {
      constexpr int ctrl = 0x120 + decltype(N_)::value;                // DPP row_ror:N
      return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, false));
    };
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if constexpr (QN <= 4) { s1[e] += ror(s1[e], std::integral_constant<int, 4>{}); s2[e] += ror(s2[e], std::integral_constant<int, 4>{}); }
      if constexpr (QN <= 8) { s1[e] += ror(s1[e], std::integral_constant<int, 8>{}); s2[e] += ror(s2[e], std::integral_constant<int, 8>{}); }
      s1[e] += __shfl_xor(s1[e], 16); s2[e] += __shfl_xor(s2[e], 16);
      s1[e] += __shfl_xor(s1[e], 32); s2[e] += __shfl_xor(s2[e], 32);
    }
    if (lane < QN) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        ex[w * TN + lane * 8 + e] = s1[e];
        ex[NW * TN + w * TN + lane * 8 + e] = s2[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    unsigned* cnt = reinterpret_cast<unsigned*>(ex + 2 * NW * TN);
    unsigned arrived = 0;
    if (lane == 0) arrived = atomicAdd(cnt, 1u);                       // (LDS operations of a wave execute in order: the partials are there)
    arrived = __builtin_amdgcn_readfirstlane(arrived);
    if ((arrived % NW) == NW - 1) {
      for (int c = lane; c < TN; c += 64) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) {
          t1 += ex[wv * TN + c];
          t2 += ex[NW * TN + wv * TN + c];
        }
        row[c] = t1;
        row[Cout + c] = t2;
      }
    }
  }
};

// A workgroup's position in its strided unit list, unit u = (g * NT + pt) * NN + nn.  Stepping by the stride costs a few scalar
// adds: decomposing u with run-time divisors cost the single-issue producer waves ~150 instructions per tile and cursor, as
// much as a whole tile's loads on the 32x32 level.
struct UnitCur {
  int u, g, pt, nn;
  __device__ __forceinline__ void init(const int u_, const int NN, const int NT) {
    u = u_; nn = u_ % NN;
    const int q = u_ / NN;
    pt = q % NT; g = q / NT;
  }
};
struct UnitStep {
  int du, dg, dpt, dnn, NN, NT;
  __device__ __forceinline__ void init(const int du_, const int NN_, const int NT_) {
    du = du_; NN = NN_; NT = NT_; dnn = du_ % NN_;
    const int q = du_ / NN_;
    dpt = q % NT_; dg = q / NT_;
  }
  __device__ __forceinline__ void step(UnitCur& c) const {
    c.u += du;
    c.nn += dnn;
    const int c1 = c.nn >= NN ? 1 : 0;
    c.nn -= c1 ? NN : 0;
    c.pt += dpt + c1;
    const int c2 = c.pt >= NT ? 1 : 0;
    c.pt -= c2 ? NT : 0;
    c.g += dg + c2;
  }
};

// ---------------------------------------------------------------------------------------------------------------------------
// Persistent producer / consumer form (the default).  What the first form above could not hide (measured, profiles/README.md r04):
//   * a wave's memory instructions return IN ORDER and share one counter (vmcnt): a filter fragment requested from L2 two K steps
//     ahead cannot be consumed before the halo loads of the next chunk, issued just before it, are back from HBM -- every chunk
//     stalled for an HBM round trip; output stores sit in the same queue;
//   * prologue (first halo round trip) and epilogue (tile -> LDS -> 64 KB of stores) of every 256-pixel tile were exposed.
// Here a workgroup is 8 waves = 4 CONSUMERS (MFMA + filter fragments from L2 + LDS fragment reads: nothing in their queue ever
// goes to HBM) and 4 PRODUCERS (everything that does: halo loads RS chunks ahead into RS register sets, BatchNorm+ReLU, commit
// to the LDS tile ring, and the finished tile's stores + BatchNorm partial sums out of an LDS output region), one s_barrier per
// chunk.  One workgroup per CU walks a strided list of (UNet, pixel tile, N tile) units, and the chunk pipeline runs ACROSS tiles:
// the next tile's first chunks are in flight while the current tile finishes, its stores leave while the next tile computes.
// The consumers' MFMA takes the FILTER fragment as its row operand and the pixels as its columns: a lane then holds 4 consecutive
// output channels of ONE pixel per accumulator quad, so the finished tile goes to LDS as packed 8-byte items (32 ds_write_b64 for
// 128 accumulators; the first version, lane = channel, needed 128 two-byte writes and ~9 VALU per value: 10 k cycles per tile, as
// long as the MFMA loop on the 32x32 level), the bias is the accumulators' initial value, and the column sums for BatchNorm are
// taken by the producers from the stored bf16 tile on its way out.
//   CK   16 | 32 channels per chunk (32: two K steps per tap, four LDS planes; the 32x32 level, where a tile has 1 - 2 chunks)
//   NB   LDS tile buffers (3 where they fit: the producers may run a chunk ahead of the barrier), RS register sets in flight
//   BRES > 0: the launch's whole filter slice (9 * BRES / 16 fragments, TN = 32, Cin = BRES <= 64) stays in consumer registers
//   PW   producer waves (4 | 8: the HBM-bound 32x32 level needs twice the loads in flight and twice the issue slots)
//   OWN  the consumer waves store their own rows (resident filter, PW == 4); else the producers store every tile
template <int TH, int TW, int NI, int WM, int MR, int NR, int CK, int OUTB, int NB, int RS, int BRES, int PW, bool OWN>
__global__ void __launch_bounds__(256 + 64 * PW, PW == 8 ? 3 : 2)
conv_gemm16p_kernel(const vv_conv_params p, const int NT, const int NN, const int total, const int nper) {
  using G_ = GGeo<TH, TW, NI>;
  constexpr int HH = G_::HH, HW = G_::HW, HWP = G_::HWP, NPIX = G_::NPIX, PLANE = G_::PLANE;
  constexpr int KS = CK / 16;                                          // K steps per tap and chunk
  constexpr int BUF = 2 * KS * PLANE;                                  // slots of one buffer: planes [kk][half]
  constexpr int WN = 4 / WM, TN = WN * NR * 32;
  static_assert(WM * MR * 32 == 256 && TH * TW * NI == 256, "256-pixel tiles");
  constexpr int NTH = 64 * PW;                                         // producer threads
  static_assert(!OWN || BRES > 0, "own-row stores come with the resident filter");
  constexpr int QI = 2 * KS;                                           // 16-byte items per pixel and chunk
  constexpr int NITEMS = NPIX * QI, NIT = (NITEMS + NTH - 1) / NTH;
  constexpr int ORS = TN + 8;                                          // bf16 per row of the output tile (16 B pad)
  constexpr int OUT4 = (256 * ORS * 2 + 15) / 16;
  constexpr int NWS = OWN ? 4 : PW;                                    // waves of the role that stores
  constexpr int EX4 = (2 * NWS * TN * 4 + 15) / 16 + 1;               // partial sums of the storing waves [2][NWS][TN] + arrival counter
  constexpr int BI4 = 4 * TN * 4 / 16;                                 // bias of the tiles in flight [4][TN] (staged by the producers)
  __shared__ uint4 lds4[NB * BUF + OUTB * OUT4 + EX4 + BI4];
  uint4* const ldsO = lds4 + NB * BUF;
  float* const ldsX = reinterpret_cast<float*>(lds4 + NB * BUF + OUTB * OUT4);
  float* const ldsBias = reinterpret_cast<float*>(lds4 + NB * BUF + OUTB * OUT4 + EX4);
  constexpr int D = NB - 1;                                            // chunk f + D is committed while chunk f is computed

  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool producer = wave >= 4;
  // this workgroup's units: XCD x (= blockIdx & 7, where the dispatcher puts the block) owns units [x nper, (x+1) nper), its
  // workgroups take them strided -- a UNet's filter panel stays in one L2
  const int xcd = blockIdx.x & 7, wgx = gridDim.x >> 3;
  const int u0 = xcd * nper + (blockIdx.x >> 3);
  const int uend = min((xcd + 1) * nper, total);
  if (u0 >= uend) return;
  const int ntile = (uend - u0 + wgx - 1) / wgx;
  // every field of the parameter block that is used below, as scalars: a `cond ? p.src1.x : p.src0.x` on the struct makes the
  // compiler keep a copy of the whole block in scratch -- and every scratch access is a VMEM instruction whose wait (vmcnt(0)) also
  // waits for the halo loads in flight (measured: the producers twice as slow)
  const int H = p.H, W = p.W, PB = p.B, pflags = p.pad0;
  const int mode = p.in_mode, csplit = p.csplit;
  const float* const s0ptr = p.src0.ptr; const int64_t s0g = p.src0.gstride; const int s0cs = p.src0.cstride, s0co = p.src0.coff;
  const float* const s1ptr = p.src1.ptr; const int64_t s1g = p.src1.gstride; const int s1cs = p.src1.cstride, s1co = p.src1.coff;
  const float* const pa_ = p.a; const float* const pb_ = p.b; const int64_t abg = p.ab_gstride;
  const float* const pw_ = p.w; const int64_t wgs = p.w_gstride;
  const float* const pbias = p.bias; const int64_t biasg = p.bias_gstride;
  float* const optr = p.out.ptr; const int64_t og = p.out.gstride; const int ocs = p.out.cstride, oco = p.out.coff;
  // second output view (channels >= osplit; same pixel and group strides, host-checked).  An N tile lies inside one half
  // (osplit % TN == 0: the tile's base moves, o1d = byte distance of the views) or is the whole layer (TN == 2 osplit: the
  // threads of the upper half carry the distance in their item offsets)
  const int osplit = p.out1.ptr ? p.osplit : 0x40000000;
  const int o1d = p.out1.ptr ? (int)((reinterpret_cast<const char*>(p.out1.ptr) - reinterpret_cast<const char*>(p.out.ptr)) + ((int64_t)p.out1.coff - oco) * 2) : 0;
  auto xoff = [&](const int ch) -> int { return (TN > osplit && ch >= osplit) ? o1d - osplit * 2 : 0; };
  auto tile_xoff = [&](const int nn) -> int { return (TN <= osplit && nn * TN >= osplit) ? o1d - osplit * 2 : 0; };
  float* const pstats = p.stats;
  constexpr int TPI = TW / TH;                                         // tiles per image: H == W == TW on every level (vv_conv_gemm16)
  static_assert(TW % TH == 0 && (TPI & (TPI - 1)) == 0, "a tile is TH full rows of a TW x TW image");
  UnitStep ustep;
  ustep.init(wgx, NN, NT);
  const int Cout = p.Cout, CinP = p.CinP, KGT = CinP >> 4;
  const int nchunk = CinP / CK;
  const int F = ntile * nchunk;                                        // chunk iterations of this workgroup
#define PT(acc_)
#define PT0()

  if (producer) {
    // =========================================================== producers
    const int t = tid - 256;
    static_assert(NTH % QI == 0, "one channel group per thread");
    const int q = t % QI;                                              // (kk, half) of this thread's items: channel offset q*8
    unsigned slot[NIT];
    int hyx[NIT];                                                      // hy | hx << 8 | im << 16
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
      const int it = t + k * NTH;
      const int hp = it / QI;
      const int hx = hp % HW, tt = hp / HW, hy = tt % HH, im = tt / HH;
      const bool inr = NITEMS % NTH == 0 || it < NITEMS;
      slot[k] = inr ? (unsigned)(q * PLANE + (im * HH + hy) * HWP + hx) : 0xFFFFFFFFu;
      hyx[k] = hy | (hx << 8) | (im << 16);
    }
    uint4 r[RS][NIT];
    float4 sa[RS], sa2[RS], sb[RS], sb2[RS];
    unsigned vbits[RS];                                                // bit k: item k is a pixel that exists (else zero padding / cube >= B)
    bool act[RS];
    float rbias[RS];                                                   // bias of the tile whose FIRST chunk the set holds (thread t < TN)
    int bslot[RS];                                                     // its slot in the bias ring, or -1
    UnitCur lu;                                                        // load cursor: unit, chunk, flat chunk index, tile ordinal
    lu.init(u0, NN, NT);
    int lc = 0, lf = 0, lt = 0;
    // per-tile state of the load cursor (recomputed when it enters a tile): pixel index and padding mask of every item, the
    // sources' descriptors.  Per chunk and item that leaves 3 VALU: offset = (pixel * 2 cs + 2 (co + c)) | (padding ? 1 << 31 : 0)
    // -- an offset past num_records makes the buffer load return zeros, no branch.
    unsigned pix[NIT], oob[NIT];
    __amdgpu_buffer_rsrc_t rs0, rs1;
    const float* ta_ = nullptr; const float* tb_ = nullptr;
    int tg = 0, tnn = 0;
    auto issue = [&](const auto SET) {
      constexpr int S_ = decltype(SET)::value;
      if (lf >= F) return;
      if (lc == 0) {
        tnn = lu.nn;
        tg = lu.g;
        const int img0 = (lu.pt / TPI) * NI, trem = lu.pt % TPI;
        const int oy0 = trem * TH - 1, ox0 = -1;
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
          const int hy = hyx[k] & 255, hx = (hyx[k] >> 8) & 255, im = hyx[k] >> 16;
          const int y = oy0 + hy, x = ox0 + hx, img = img0 + im;
          const bool ok = slot[k] != 0xFFFFFFFFu && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W && img < PB;
          pix[k] = ok ? (unsigned)((img * H + y) * W + x) : 0u;
          oob[k] = ok ? 0u : 0x80000000u;
        }
        rs0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s0ptr + (int64_t)tg * s0g), 0, 0x7FFFFFFF, 0x00020000);
        rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((s1ptr ? s1ptr : s0ptr) + (int64_t)tg * s1g), 0, 0x7FFFFFFF, 0x00020000);
        ta_ = pa_ ? pa_ + (int64_t)tg * abg : nullptr;
        tb_ = pb_ ? pb_ + (int64_t)tg * abg : nullptr;
      }
      const int c0 = lc * CK;
      const int c = c0 + q * 8;
      const bool second = __builtin_amdgcn_readfirstlane((int)(mode == VV_IN_CAT && c0 >= csplit)) != 0;
      act[S_] = (mode == VV_IN_ACT) || (mode == VV_IN_CAT && !second);
      if (act[S_]) {
        sa[S_] = *reinterpret_cast<const float4*>(ta_ + c); sa2[S_] = *reinterpret_cast<const float4*>(ta_ + c + 4);
        sb[S_] = *reinterpret_cast<const float4*>(tb_ + c); sb2[S_] = *reinterpret_cast<const float4*>(tb_ + c + 4);
      }
      const unsigned cs2 = (unsigned)(second ? s1cs : s0cs) * 2u;
      const unsigned cb2 = (unsigned)((second ? s1co - csplit : s0co) + c) * 2u;
      if (second) {
#pragma unroll
        for (int k = 0; k < NIT; ++k)
          r[S_][k] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs1, (__umul24(pix[k], cs2) + cb2) | oob[k], 0, 0));
      } else {
#pragma unroll
        for (int k = 0; k < NIT; ++k)
          r[S_][k] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs0, (__umul24(pix[k], cs2) + cb2) | oob[k], 0, 0));
      }
      unsigned vb = 0;
#pragma unroll
      for (int k = 0; k < NIT; ++k) vb |= (oob[k] >> 31) << k;
      vbits[S_] = ~vb;
      bslot[S_] = -1;
      if (lc == 0 && pbias) {
        bslot[S_] = lt & 3;
        if (t < TN) rbias[S_] = pbias[(int64_t)tg * biasg + tnn * TN + t];
      }
      ++lf;
      if (++lc == nchunk) { lc = 0; ustep.step(lu); ++lt; }
    };
    int cf = 0;                                                        // flat index of the next chunk to commit
    auto commit = [&](const auto SET) {
      constexpr int S_ = decltype(SET)::value;
      if (cf >= F) return;
      const int boff = (cf % NB) * BUF;
      ++cf;
      if (bslot[S_] >= 0 && t < TN) ldsBias[bslot[S_] * TN + t] = rbias[S_];
#pragma unroll
      for (int k = 0; k < NIT; ++k) {
        if (NITEMS % NTH == 0 || k < NIT - 1 || slot[k] != 0xFFFFFFFFu) {
          uint4 h = r[S_][k];
          if (act[S_]) {                   // (wave-uniform)  padding items were loaded as zeros and must stay zeros: mask, no branch
            const uint2 lo = vv_pack_bf16x4(vv_act4(vv_unpack_bf16x4(make_uint2(h.x, h.y)), sa[S_], sb[S_]));
            const uint2 hi = vv_pack_bf16x4(vv_act4(vv_unpack_bf16x4(make_uint2(h.z, h.w)), sa2[S_], sb2[S_]));
            const unsigned vm = 0u - ((vbits[S_] >> k) & 1u);
            h = make_uint4(lo.x & vm, lo.y & vm, hi.x & vm, hi.y & vm);
          }
          lds4[boff + slot[k]] = h;
        }
      }
    };
    // ---- finished tiles leave through the output regions (skipped when the consumers store their own rows: BRES > 0)
    OutStore<TH, TW, NI, TN, NTH, 256> outs;
    outs.init(t, 0, H, W, ocs, xoff);
    constexpr int NOUT = OutStore<TH, TW, NI, TN, NTH, 256>::NOUT;
    constexpr int NPARTS = NOUT >= 4 ? 4 : NOUT, KPP = NOUT / NPARTS;  // a tile's items leave in NPARTS parts, spread over the iterations
    int pend = NPARTS;                                                 // next part of the tile being stored (NPARTS: none)
    UnitCur su;                                                        // unit / output region of the next tile to be stored
    su.init(u0, NN, NT);
    int sreg = 0;
    int st_sbase = 0, st_nimg = 0, st_row = 0;
    bool st_flush = false;
    __amdgpu_buffer_rsrc_t rsO;
    auto begin_tile = [&]() {
      const int nn = su.nn, pt = su.pt, g = su.g;
      const int img0 = (pt / TPI) * NI, trem = pt % TPI;
      const int ty0 = trem * TH, tx0 = 0;
      rsO = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(optr + (int64_t)g * og) + (int64_t)oco * 2, 0, 0x7FFFFFFF, 0x00020000);
      st_sbase = (((img0 * H + ty0) * W + tx0) * ocs + nn * TN) * 2 + tile_xoff(nn);
      st_nimg = PB - img0;
      st_row = ((g * NT + pt) * 2) * Cout + nn * TN;
      UnitCur un = su;                                                 // the run of tiles ends when the next unit is another UNet / N tile
      ustep.step(un);
      st_flush = un.u >= uend || un.nn != nn || un.g != g;
      pend = 0;
    };
    auto do_parts = [&](int n) {
      const char* region = reinterpret_cast<const char*>(ldsO + sreg * OUT4);
      for (; n > 0 && pend < NPARTS; --n, ++pend) {
        if (pend == 0) outs.template store<0, KPP>(region, rsO, st_sbase, st_nimg, pstats != nullptr);
        if constexpr (NPARTS > 1) { if (pend == 1) outs.template store<KPP, 2 * KPP>(region, rsO, st_sbase, st_nimg, pstats != nullptr); }
        if constexpr (NPARTS > 2) {
          if (pend == 2) outs.template store<2 * KPP, 3 * KPP>(region, rsO, st_sbase, st_nimg, pstats != nullptr);
          if (pend == 3) outs.template store<3 * KPP, 4 * KPP>(region, rsO, st_sbase, st_nimg, pstats != nullptr);
        }
      }
      if (pend == NPARTS && n >= 0) {                                  // tile done: its stats row, next tile / region
        if (pstats) {
          if (st_flush) outs.template flush<PW>(ldsX, wave - 4, lane, pstats + st_row, Cout);
          else if (t < TN) { pstats[st_row + t] = 0.f; pstats[st_row + Cout + t] = 0.f; }
        }
        ustep.step(su); sreg = (sreg + 1) % OUTB;
        pend = NPARTS + 1;
      }
    };
    const int ppi = (NPARTS + (nchunk > 2 ? nchunk - 2 : 0)) / (nchunk > 1 ? nchunk - 1 : 1);      // parts per iteration: done one iteration early
    if (t == 0 && !OWN) *reinterpret_cast<unsigned*>(ldsX + 2 * NWS * TN) = 0u;
#pragma unroll
    for (int k = 0; k < RS; ++k) { bslot[k] = -1; rbias[k] = 0.f; act[k] = false; }
    // set (f + D) % RS (compile time: the loop is unrolled by RS) of iteration f holds chunk f + D.  Written out by hand: generic
    // lambdas nested in vv_static_for made the compiler keep the captured scalars in scratch (every scratch access = a VMEM
    // instruction whose vmcnt(0) wait drains the halo loads in flight).
    const std::integral_constant<int, 0> s0{};
    const std::integral_constant<int, 1> s1{};
    const std::integral_constant<int, 2 % RS> s2{};
    issue(s0);
    issue(s1);
    if constexpr (RS == 3) issue(s2);
    if constexpr (D >= 1) { commit(s0); issue(s0); }
    if constexpr (D >= 2) { commit(s1); issue(s1); }
    __syncthreads();
    int f = 0;
#define VV_PROD_ITER(SETC)                                                                                              \
    {                                                                                                                   \
      PT0();                                                                                                            \
      /* a tile that ended with iteration f - 1 is stored first (its output region is free again before the consumers' next epilogue) */ \
      if constexpr (!OWN) {                                                                                             \
        if (f > 0 && f % nchunk == 0) begin_tile();                                                                     \
        if (pend < NPARTS) do_parts(ppi);                                                                               \
      }                                                                                                                 \
      PT(tB);                                                                                                           \
      commit(SETC);                                                                                                     \
      PT(tA);                                                                                                           \
      issue(SETC);                                                                                                      \
      PT(tD);                                                                                                           \
      __syncthreads();                                                                                                  \
      PT(tC);                                                                                                           \
      ++f;                                                                                                              \
    }
    while (f < F) {
      // iteration f with f % RS == 0: set D % RS, then (D + 1) % RS, ...
      if constexpr (RS == 2) {
        if constexpr (D == 1) { VV_PROD_ITER(s1) if (f >= F) break; VV_PROD_ITER(s0) }
        else { VV_PROD_ITER(s0) if (f >= F) break; VV_PROD_ITER(s1) }
      } else {
        static_assert(RS == 2 || D == 2, "three register sets come with three LDS buffers");
        VV_PROD_ITER(s2) if (f >= F) break; VV_PROD_ITER(s0) if (f >= F) break; VV_PROD_ITER(s1)
      }
    }
#undef VV_PROD_ITER
    if constexpr (!OWN) {
      if (pend < NPARTS) do_parts(NPARTS);                             // (cannot happen: a tile's parts end before the next tile does)
      begin_tile();                                                    // the last tile
      do_parts(NPARTS);
    }
    return;
  }

  // =========================================================== consumers
  const int wm = wave / WN, wn = wave % WN;
  __builtin_amdgcn_s_setprio(1);
  int abase[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    const int pp = (wm * MR + m) * 32 + perm_row<TW>(l31);
    const int im = pp / (TH * TW), rr = (pp / TW) % TH, cc = pp % TW;
    abase[m] = half * PLANE + (im * HH + rr) * HWP + cc;
  }
  const v4f* ldsA = reinterpret_cast<const v4f*>(lds4);
  const int brow = 2 * Cout * 16;
  // filter panel of a unit: descriptor (SGPRs) + this lane's offset
  const int g00 = u0 / (NN * NT), nn00 = u0 % NN;
  auto panel = [&](const UnitCur& c, __amdgpu_buffer_rsrc_t& rs, unsigned& bvo) {
    const bool in = c.u < uend;                                        // past the end: any valid panel (loads are never consumed)
    const int nn = in ? c.nn : nn00, g = in ? c.g : g00;
    rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pw_ + (int64_t)g * wgs), 0, 0x7FFFFFFF, 0x00020000);
    bvo = (unsigned)(half * Cout + nn * TN + wn * (NR * 32) + l31) * 16u;
  };
  __amdgpu_buffer_rsrc_t rsW, rsN;
  unsigned bvo, bvoN;
  UnitCur cu, cnx;                                                     // this tile's unit, the unit whose panel is fetched next
  cu.init(u0, NN, NT);
  cnx = cu;
  panel(cnx, rsW, bvo);
  ustep.step(cnx);
  panel(cnx, rsN, bvoN);

  // accumulators: acc[m][n][i] = pixel (block m, column l31) x output channel n*32 + (i & 3) + 8 (i >> 2) + 4 half
  v16f acc[MR][NR];
  constexpr int NS = 9 * KS;                 // K steps per chunk: s = kk * 9 + tap (the accumulation order of KS 16-channel chunks)
  constexpr int RB = 3, PDB = 2;
  static_assert(NS % RB == 0, "ring turns per chunk");
  constexpr int NBR = BRES > 0 ? 9 * BRES / 16 : 1;
  static_assert(BRES == 0 || NR == 1, "resident filter: 32-wide N tiles");
  v4f fb[RB][NR];
  v4f fbr[NBR];                                                        // resident filter [kg][tap]
  constexpr int AR = NR >= 2 ? 1 : 2;                                  // A fragment register sets (see chunk())
  v4f fa[AR][MR];
  auto loadB = [&](const __amdgpu_buffer_rsrc_t& rs, const unsigned vo, const int s, const int kg0, const int n) -> v4f {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, vo, ((s % 9) * KGT + kg0 + (s / 9)) * brow + n * 512, 0);
  };
  auto aoff = [&](const int s) -> int { const int tap = s % 9, kk = s / 9; return kk * 2 * PLANE + (tap / 3) * HWP + (tap % 3); };

  // one chunk = NS K steps out of buffer `buf`.  FIRST: the tile's first chunk (accumulators start from the bias); LAST: its last
  // one (the B prefetch of steps NS, NS + 1 goes to the NEXT unit's panel).  KG (resident filter only): compile-time chunk index.
  auto chunk = [&](const int ci, const int buf, const auto FIRST_, const auto LAST_, const auto KG_) {
    constexpr bool first_chunk = decltype(FIRST_)::value, last_chunk = decltype(LAST_)::value;
    constexpr int kgc = decltype(KG_)::value;
    const int boff = buf * BUF;
    const int kg0 = ci * KS;
#pragma unroll
    for (int m = 0; m < MR; ++m) fa[0][m] = ldsA[boff + abase[m] + aoff(0)];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int cb = s % RB;
      const int sn = s + 1, s2 = s + PDB;
      const int ca = AR == 2 ? (s & 1) : 0, cn = AR == 2 ? (ca ^ 1) : 0;
#pragma unroll
      for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int n = 0; n < NR; ++n) {
          v4f bfrag;
          if constexpr (BRES > 0) bfrag = fbr[(kgc * KS + s / 9) * 9 + s % 9];
          else bfrag = fb[cb][n];
          if (first_chunk && s == 0) {     // a tile's first step starts from the instruction's inline-constant 0
            const v16f z = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, bfrag), __builtin_bit_cast(v8bf, fa[ca][m]), z, 0, 0, 0);
          } else
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, bfrag), __builtin_bit_cast(v8bf, fa[ca][m]), acc[m][n], 0, 0, 0);
          // A fragments of the next step.  An LDS return into registers that an MFMA issued just before still reads as its COLUMN
          // operand stalls the wave for that instruction (measured: 67 instead of 42 cycles per MFMA with the read right behind
          // its last reader), so: AR == 2 -- a second register set (the 64- and 32-wide N tiles have the registers);
          // AR == 1 -- same registers, one ROW of MFMAs later: fragment m - 1 behind row m, fragment MR - 1 of THIS step behind row 0
          if (n == NR - 1) {
            if constexpr (AR == 2) {
              if (sn < NS) fa[cn][m] = ldsA[boff + abase[m] + aoff(sn)];
            } else {
              if (m >= 1 && sn < NS) fa[0][m - 1] = ldsA[boff + abase[m - 1] + aoff(sn)];
              if (m == 0 && s > 0) fa[0][MR - 1] = ldsA[boff + abase[MR - 1] + aoff(s)];
            }
          }
          // B fragments of step s + 2 behind the first MFMAs (three-deep ring)
          if constexpr (BRES == 0) {
            if (n == 0 && m < NR) {
              if (s2 < NS) fb[s2 % RB][m] = loadB(rsW, bvo, s2, kg0, m);
              else if constexpr (last_chunk) fb[s2 % RB][m] = loadB(rsN, bvoN, s2 - NS, 0, m);
              else fb[s2 % RB][m] = loadB(rsW, bvo, s2 - NS, kg0 + KS, m);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
    }
  };
  // finished tile -> output region `reg`: 4 consecutive channels of the lane's pixel per accumulator quad = one 8-byte LDS write
  auto epilogue = [&](const int reg, const int tile) {
    unsigned short* lo = reinterpret_cast<unsigned short*>(ldsO + reg * OUT4) + wn * (NR * 32) + 4 * half;
    const bool relu = (pflags & VV_CONV_RELU) != 0;
    float4 bq[NR][4];
#pragma unroll
    for (int n = 0; n < NR; ++n)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bq[n][j] = pbias ? *reinterpret_cast<const float4*>(ldsBias + (tile & 3) * TN + wn * (NR * 32) + n * 32 + 8 * j + 4 * half)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      const int pp = (wm * MR + m) * 32 + perm_row<TW>(l31);
      unsigned short* lr = lo + pp * ORS;
#pragma unroll
      for (int n = 0; n < NR; ++n)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float4 v = make_float4(acc[m][n][4 * j] + bq[n][j].x, acc[m][n][4 * j + 1] + bq[n][j].y, acc[m][n][4 * j + 2] + bq[n][j].z,
                                 acc[m][n][4 * j + 3] + bq[n][j].w);
          if (relu) { v.x = v.x < 0.f ? 0.f : v.x; v.y = v.y < 0.f ? 0.f : v.y; v.z = v.z < 0.f ? 0.f : v.z; v.w = v.w < 0.f ? 0.f : v.w; }
          *reinterpret_cast<uint2*>(lr + n * 32 + 8 * j) = vv_pack_bf16x4(v);
        }
    }
  };

  // register-resident filter (32-wide N tiles, WN == 1): a consumer wave owns 64 whole pixel rows of the output region, nothing in its
  // memory queue is ever waited for, and it idles at the barrier most of the time (the launch is HBM-bound) -- it stores its own
  // rows and keeps their column sums; the producers only load.
  OutStore<TH, TW, NI, TN, 64, 64> couts;
  if constexpr (OWN) {
    static_assert(!OWN || (WM == 4 && MR == 2), "own-row stores: one wave = 64 pixels x the whole N tile");
    couts.init(lane, wm * 64, H, W, ocs, xoff);
    if (tid == 0) *reinterpret_cast<unsigned*>(ldsX + 2 * NWS * TN) = 0u;
  }
  auto store_own = [&](const UnitCur& c, const int reg) {
    const int nn = c.nn, pt = c.pt, g = c.g;
    const int img0 = (pt / TPI) * NI, trem = pt % TPI;
    const int ty0 = trem * TH, tx0 = 0;
    const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(optr + (int64_t)g * og) + (int64_t)oco * 2, 0,
                                                                          0x7FFFFFFF, 0x00020000);
    const int sbase = (((img0 * H + ty0) * W + tx0) * ocs + nn * TN) * 2 + tile_xoff(nn);
    couts.template store<0, OutStore<TH, TW, NI, TN, 64, 64>::NOUT>(reinterpret_cast<const char*>(ldsO + reg * OUT4), rsO, sbase, PB - img0,
                                                                    pstats != nullptr);
    if (pstats) {
      const int row = ((g * NT + pt) * 2) * Cout + nn * TN;
      UnitCur un = c;
      ustep.step(un);
      if (un.u >= uend || un.nn != nn || un.g != g) couts.template flush<4>(ldsX, wave, lane, pstats + row, Cout);
      else if (wave == 0 && lane < TN) { pstats[row + lane] = 0.f; pstats[row + Cout + lane] = 0.f; }
    }
  };
  if constexpr (BRES > 0) {
    vv_static_for<0, NBR>([&](auto I) { constexpr int i = decltype(I)::value; fbr[i] = loadB(rsW, bvo, i % 9, i / 9, 0); });
  } else {
#pragma unroll
    for (int s = 0; s < PDB; ++s)
#pragma unroll
      for (int n = 0; n < NR; ++n) fb[s][n] = loadB(rsW, bvo, s, 0, n);
  }
  __syncthreads();
  const std::true_type yes{};
  const std::false_type no{};
  const std::integral_constant<int, 0> k0{};
  int f = 0, reg = 0, tile = 0;
  int gcur = u0 / (NN * NT);
  for (; cu.u < uend; ustep.step(cu)) {
    if constexpr (BRES > 0) {
      // the whole Cin is a few compile-time chunks: chunk index static, filter in registers (re-read when the UNet changes)
      const int g = cu.g;
      if (g != gcur) {
        gcur = g;
        panel(cu, rsW, bvo);
        vv_static_for<0, NBR>([&](auto I) { constexpr int i = decltype(I)::value; fbr[i] = loadB(rsW, bvo, i % 9, i / 9, 0); });
      }
      constexpr int NCH = BRES / CK;
      vv_static_for<0, NCH>([&](auto CI) {
        constexpr int ci = decltype(CI)::value;
        PT0();
        chunk(ci, f % NB, std::integral_constant<bool, ci == 0>{}, std::integral_constant<bool, ci == NCH - 1>{}, CI);
        PT(tA);
        if constexpr (ci == NCH - 1) {
          epilogue(reg, tile);
          if constexpr (OWN) store_own(cu, reg);
          PT(tB);
          reg = (reg + 1) % OUTB;
          ++tile;
        }
        ++f;
        __syncthreads();
        PT(tC);
      });
    } else {
      if (nchunk == 1) {
        PT0();
        chunk(0, f % NB, yes, yes, k0);
        PT(tA);
      } else {
        PT0();
        chunk(0, f % NB, yes, no, k0);
        PT(tA);
        ++f;
        __syncthreads();
        PT(tC);
        for (int ci = 1; ci + 1 < nchunk; ++ci, ++f) {
          chunk(ci, f % NB, no, no, k0);
          PT(tA);
          __syncthreads();
          PT(tC);
        }
        chunk(nchunk - 1, f % NB, no, yes, k0);
        PT(tA);
      }
      epilogue(reg, tile);
      PT(tB);
      reg = (reg + 1) % OUTB;
      ++tile;
      ++f;
      rsW = rsN; bvo = bvoN;
      ustep.step(cnx);
      panel(cnx, rsN, bvoN);
      __syncthreads();
      PT(tC);
    }
  }
}

template <int TH, int TW, int NI, int WM, int MR, int NR, int CK, int OUTB, int NB, int RS, int BRES, int PW = 4>
int launch_p(const vv_conv_params* p, hipStream_t st, const int ncu) {
  constexpr int TN = (4 / WM) * NR * 32;
  constexpr bool OWN = BRES > 0 && PW == 4;
  const int NT = ((p->B + NI - 1) / NI) * (p->H / TH) * (p->W / TW);
  const int NN = p->Cout / TN;
  const int total = p->G * NN * NT;
  const int nper = (total + 7) / 8;
  int grid = ncu < nper * 8 ? ncu : nper * 8;
  grid = (grid + 7) & ~7;
  VV_LAUNCH((conv_gemm16p_kernel<TH, TW, NI, WM, MR, NR, CK, OWN ? 1 : OUTB, NB, RS, BRES, PW, OWN>), dim3(grid), dim3(256 + 64 * PW), 0, st, *p, NT,
            NN, total, nper);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

// NB3: three LDS tile buffers fit beside the output region(s) of every N-tile width at this level
template <int TH, int TW, int NI, int CK, bool NB3, int RS, int PW = 4>
int dispatch_p(const vv_conv_params* p, hipStream_t st) {
  const int ncu = vv_num_cus();                                        // one persistent workgroup per CU
  constexpr int NB = NB3 ? 3 : 2;
  // (eight producer waves = three waves per SIMD, 168 registers: the 128-wide tiles and the 64-channel resident filter do not fit)
  if constexpr (PW == 4) {
    if (p->Cout % 128 == 0 && p->CinP / CK >= 2) return launch_p<TH, TW, NI, 2, 4, 2, CK, 1, NB, RS, 0, PW>(p, st, ncu);
  }
  if (p->Cout % 64 == 0) return launch_p<TH, TW, NI, 2, 4, 1, CK, 2, NB, RS, 0, PW>(p, st, ncu);
  // 32-wide N tiles: the filter slice of a <= 64-channel layer stays in registers
  if constexpr (PW == 4) {
    if (p->CinP == 64 && 64 % CK == 0) return launch_p<TH, TW, NI, 4, 2, 1, CK, 2, NB, RS, 64, PW>(p, st, ncu);
  }
  if (p->CinP == 32 && 32 % CK == 0) return launch_p<TH, TW, NI, 4, 2, 1, CK, 2, NB, RS, 32, PW>(p, st, ncu);
  if (p->CinP == 16 && CK == 16) return launch_p<TH, TW, NI, 4, 2, 1, CK, 2, NB, RS, 16, PW>(p, st, ncu);
  return launch_p<TH, TW, NI, 4, 2, 1, CK, 2, NB, RS, 0, PW>(p, st, ncu);
}

}  // namespace

int vv_conv_gemm16(const vv_conv_params* p, hipStream_t st) {
  if (p->H != p->W || p->CinP % 16 || p->Cout % 32) return VV_ERR_UNSUPPORTED;
  if (p->bn_partial) return VV_ERR_UNSUPPORTED;      // the BatchNorm-backward sums are an epilogue of conv_mfma_kernel's 32-wide launches only
  if (p->in_mode != VV_IN_PLAIN && p->in_mode != VV_IN_ACT && p->in_mode != VV_IN_CAT) return VV_ERR_UNSUPPORTED;
  // 16-byte items of 8 channels: every channel offset / stride a multiple of 8 elements
  if (p->src0.cstride % 8 || p->src0.coff % 8 || p->out.cstride % 8 || p->out.coff % 8) return VV_ERR_BAD_ARG;
  if (p->in_mode == VV_IN_CAT && (p->csplit % 16 || p->src1.cstride % 8 || p->src1.coff % 8 || !p->src1.ptr)) return VV_ERR_BAD_ARG;
  if (p->in_mode != VV_IN_PLAIN && (!p->a || !p->b)) return VV_ERR_BAD_ARG;
  if (p->out1.ptr) {      // second output view (validated by vv_conv_mfma): one buffer descriptor per UNet covers both tensors
    const int64_t d = (reinterpret_cast<const char*>(p->out1.ptr) - reinterpret_cast<const char*>(p->out.ptr)) +
                      ((int64_t)p->out1.coff - p->out.coff) * 2;
    const int64_t span = (int64_t)p->B * p->H * p->W * p->out.cstride * 2;
    if (d < 0 || d + span >= 0x7FFFFFFFll || p->Cout != 2 * p->osplit || (p->osplit & (p->osplit - 1))) return VV_ERR_UNSUPPORTED;
  }
  const bool ck32 = p->CinP % 32 == 0 && (p->in_mode != VV_IN_CAT || p->csplit % 32 == 0);
  switch (p->H) {      // (32x32: vv_conv_mfma keeps those launches on conv_mfma_kernel, see the header)
    case 16: return ck32 ? dispatch_p<16, 16, 1, 32, false, 2>(p, st) : dispatch_p<16, 16, 1, 16, true, 2>(p, st);
    case 8: return ck32 ? dispatch_p<8, 8, 4, 32, false, 2>(p, st) : dispatch_p<8, 8, 4, 16, true, 2>(p, st);
    case 4: return dispatch_p<4, 4, 16, 16, false, 2>(p, st);
  }
  return VV_ERR_UNSUPPORTED;
}
