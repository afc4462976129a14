// MFMA weight-gradient kernels (gfx950, v_mfma_f32_32x32x2_f32, exact fp32, deterministic split-K).
//
//   dW[tap][ci][co] = sum_pixels act[pixel (+tap)][ci] * dy[pixel (*2-1+tap)][co]
// GEMM view per tap: M = ci (32 per workgroup), N = co (32 per workgroup), K = pixels.  One workgroup owns a
// (ci-tile, co-tile, k-split) triple, walks its share of the pixel tiles, and each of its 4 waves keeps all 9 taps
// of the 32x32 tile in accumulators (9 x 16 VGPRs) over a quarter of every pixel tile.  Both operands come from LDS
// with ds_read_b32: lanes 0-31 read 32 consecutive channels of pixel k, lanes 32-63 of pixel k+1 (conflict-free).
// The layer input is re-materialised on load from the producer's pre-BN tensor (BatchNorm+ReLU / max-pool / concat /
// frame erasure), exactly as in the forward kernel, so no post-activation tensor is ever stored.
// Tiles are software-pipelined through registers (loads of tile t+1 fly under tile t's MFMAs).  The 4 waves are
// summed through LDS in fixed order, each workgroup writes one slab; vv_wgrad_reduce adds the slabs in a fixed order
// (bitwise reproducible) and scatters into the PyTorch parameter layout.
//
// Replaces the autograd weight gradients of nn.Conv2d / nn.ConvTranspose2d (model/unet.py:10,13,54; cuDNN).
#include <cstdlib>
#include "vv_common.h"

namespace {

template <int TH, int TW, int NI, int KIND>
__global__ void __launch_bounds__(VV_WG, 1)
wgrad_mfma_kernel(const vv_wgrad_params p, const int NT, const int NCI, const int NCO, const int total, const int nper) {
  constexpr bool CT = KIND != VV_CONV3;
  constexpr int TP = TH * TW * NI;          // pixels (GEMM-K) per tile
  constexpr int PPW = TP / 4;               // per wave
  // act tile / dy tile geometry
  constexpr int AHH = CT ? TH : TH + 2, AHW = CT ? TW : TW + 2;
  constexpr int BHH = CT ? 2 * TH + 1 : TH, BHW = CT ? 2 * TW + 1 : TW;
  constexpr int ASZ = NI * AHH * AHW * 32, BSZ = NI * BHH * BHW * 32, TSZ = ASZ + BSZ;
  // Two LDS tile buffers where they fit: the staging of tile t+1 (loads, BN+ReLU, LDS writes) is spread over the MFMA loop of tile t.
  // 3x3 convolutions always; transposed convolutions with the 64-pixel tiles of round 6 (their dy halo tile is (2TH+1) x (2TW+1) pixels:
  // 50 KB per buffer at 64 pixels, 88 - 99 KB at 128, where the one-buffer form below stays: commit and MFMA phases take turns).
  constexpr bool DB = 2 * TSZ * 4 <= 160 * 1024;
  __shared__ float lds[DB ? 2 * TSZ : TSZ];
  float* lA = lds;
  float* lB = lds + ASZ;

  int w = vv_xcd_remap(blockIdx.x, nper);
  if (w >= total) return;
  const int KS = p.ksplit;
  const int ks = w % KS; w /= KS;
  const int cot = w % NCO; w /= NCO;
  const int cit = w % NCI;
  const int g = w / NCI;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, l31 = lane & 31;
  const int H = p.H, W = p.W;
  const int tilesX = W / TW, tilesY = H / TH, tpi = tilesX * tilesY;

  const VVSrc sa = vv_make_src(p, g, H, W);
  VVSrc sb;
  sb.p0 = p.dy.ptr + (int64_t)g * p.dy.gstride; sb.cs0 = p.dy.cstride; sb.co0 = p.dy.coff;
  sb.a = sb.b = nullptr; sb.p1 = nullptr; sb.cs1 = sb.co1 = 0; sb.chmap = nullptr; sb.csplit = 0;
  sb.mode = VV_IN_PLAIN; sb.SH = CT ? 2 * H : H; sb.SW = CT ? 2 * W : W; sb.B = p.B;

  v16f acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  // software pipeline over this workgroup's pixel tiles: the global loads of tile t+1 are in flight (in registers)
  // while tile t runs on the matrix cores; one workgroup per CU (up to 512 VGPRs per lane), so nothing else hides them.
  VVStagerB<NI, AHH, AHW, 32, 32> stA;
  VVStagerB<NI, BHH, BHW, 32, 32> stB;
  static_assert(TW >= 4, "tiles span whole rows");
  auto issue = [&](const int pt) {
    const int img0 = (pt / tpi) * NI;
    const int trem = pt % tpi;
    const int ty0 = (trem / tilesX) * TH, tx0 = (trem % tilesX) * TW;
    stA.prefetch(sa, img0, CT ? ty0 : ty0 - 1, CT ? tx0 : tx0 - 1, cit * 32, tid, p.CinP);
    stB.prefetch(sb, img0, CT ? 2 * ty0 - 1 : ty0, CT ? 2 * tx0 - 1 : tx0, cot * 32, tid);
  };
  // every tile of this kernel spans full image rows (TW == W), so the column origin is tile independent
  stA.init(sa, CT ? 0 : -1, tid);
  stB.init(sb, CT ? -1 : 0, tid);
  const int dbg = p.pad0;                   // bring-up switch (0 in production): 1 = skip staging (timing experiments)

  // operand addressing of k-step kk (pixel pair 2kk, 2kk+1 of this wave's quarter of the tile)
  auto addr = [&](const float* tA, const float* tB, const int kk, const float*& pa, const float*& pb) {
    const int pp = wave * PPW + 2 * kk + half;
    const int im = pp / (TH * TW), r = (pp / TW) % TH, c = pp % TW;
    if constexpr (CT) {
      pa = tA + pp * 32 + l31;
      pb = tB + ((im * BHH + 2 * r) * BHW + 2 * c) * 32 + l31;
    } else {
      pa = tA + ((im * AHH + r) * AHW + c) * 32 + l31;
      pb = tB + pp * 32 + l31;
    }
  };
  auto rd = [&](const float* pa, const float* pb, const int t) -> float {      // the tap-shifted operand of tap t
    if constexpr (CT) return pb[((t / 3) * BHW + (t % 3)) * 32];
    else return pa[((t / 3) * AHW + (t % 3)) * 32];
  };
  auto rd1 = [&](const float* pa, const float* pb) -> float {                   // the un-shifted operand
    if constexpr (CT) return pa[0];
    else return pb[0];
  };
  auto mfma1 = [&](const int t, const float sh, const float un) {
    if constexpr (CT) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(un, sh, acc[t], 0, 0, 0);   // A = act, B = dy(tap)
    else acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(sh, un, acc[t], 0, 0, 0);                // A = act(tap), B = dy
  };
  constexpr int NK = PPW / 2;
  static_assert(NK % 2 == 0, "k-loop is unrolled by two");

  if constexpr (DB) {
    // ---- double-buffered pipeline.  Slot s = kk*9 + t is "MFMA t of k-step kk, one ds_read of step kk+1, one staging
    // piece".  Slots [0, NP) issue the buffer loads of the next tile into registers, slots [C0, C0+NP) apply BN+ReLU and
    // write them to the OTHER LDS buffer; one barrier per tile.  The workgroup's last tile stages a dead tile (all
    // loads out of range) so the loop stays branch-free.
    constexpr int NPA = decltype(stA)::NIT, NPB = decltype(stB)::NIT, NP = NPA + NPB;
    constexpr int NSLOT = NK * 9, C0 = NSLOT - NP - 9;
    static_assert(C0 >= 2 * NP, "not enough MFMA slots between load issue and commit");
    auto begin_tile = [&](const int pt, const bool live) {
      const int img0 = (pt / tpi) * NI;
      const int trem = pt % tpi;
      const int ty0 = (trem / tilesX) * TH, tx0 = (trem % tilesX) * TW;
      stA.begin(sa, img0, CT ? ty0 : ty0 - 1, CT ? tx0 : tx0 - 1, cit * 32, tid, p.CinP, live);
      stB.begin(sb, img0, CT ? 2 * ty0 - 1 : ty0, CT ? 2 * tx0 - 1 : tx0, cot * 32, tid, 1 << 30, live);
    };
    begin_tile(ks, true);
    vv_static_for<0, NPA>([&](auto K) { stA.template load_piece<K.value>(sa, -1, tid); });
    vv_static_for<0, NPB>([&](auto K) { stB.template load_piece<K.value>(sb, 0, tid); });
    vv_static_for<0, NPA>([&](auto K) { stA.template commit_piece<K.value>(lA, tid); });
    vv_static_for<0, NPB>([&](auto K) { stB.template commit_piece<K.value>(lB, tid); });
    __syncthreads();
    int cur = 0;
    for (int pt = ks; pt < NT; pt += KS) {
      const float* tA = lds + cur * TSZ;
      const float* tB = tA + ASZ;
      float* nA = lds + (cur ^ 1) * TSZ;
      float* nB = nA + ASZ;
      const bool live = (pt + KS < NT) && !(dbg & 2);
      begin_tile(live ? pt + KS : ks, live);
      float a[2][9], b[2];
      const float *pa, *pb;
      addr(tA, tB, 0, pa, pb);
      b[0] = rd1(pa, pb);
#pragma unroll
      for (int t = 0; t < 9; ++t) a[0][t] = rd(pa, pb, t);
      vv_static_for<0, NK>([&](auto KK) {
        constexpr int kk = KK.value, c = kk & 1, n = c ^ 1;
        addr(tA, tB, kk + 1 < NK ? kk + 1 : 0, pa, pb);
        __builtin_amdgcn_sched_barrier(0);
        vv_static_for<0, 9>([&](auto T) {
          constexpr int t = T.value, slot = kk * 9 + t;
          mfma1(t, a[c][t], b[c]);
          if constexpr (kk + 1 < NK) {
            if constexpr (t == 0) b[n] = rd1(pa, pb);
            a[n][t] = rd(pa, pb, t);
          }
          if constexpr (slot < NPA) stA.template load_piece<slot>(sa, -1, tid);
          else if constexpr (slot < NP) stB.template load_piece<slot - NPA>(sb, 0, tid);
          else if constexpr (slot >= C0 && slot < C0 + NPA) stA.template commit_piece<slot - C0>(nA, tid);
          else if constexpr (slot >= C0 + NPA && slot < C0 + NP) stB.template commit_piece<slot - C0 - NPA>(nB, tid);
          __builtin_amdgcn_sched_barrier(0);
        });
      });
      __syncthreads();                        // next buffer complete, current buffer no longer read
      if (!(dbg & 2)) cur ^= 1;             // dbg 2: stage the first tile only (k-loop on real data)
    }
  } else {
  if (!(dbg & 1)) issue(ks);
  for (int pt = ks; pt < NT; pt += KS) {
    if (!(dbg & 1) && !((dbg & 2) && pt != ks)) {      // dbg 2: stage the first tile only (k-loop on real data)
      if (pt != ks) __syncthreads();          // every wave is done reading the previous tile
      stA.commit(lA, tid);
      stB.commit(lB, tid);
      __syncthreads();
      if (pt + KS < NT) issue(pt + KS);
    }

    // k-loop over pixel pairs.  The LDS operands of step k+1 are read into a second register set while step k runs:
    // each MFMA is followed by exactly one ds_read of the next step (pinned with sched_barrier), so the matrix pipe
    // never waits behind a block of LDS issue slots and no ds_read is waited on right after it was issued.
    float a0[9], a1[9], b0, b1;
    const float *pa, *pb;
    addr(lA, lB, 0, pa, pb);
    b0 = rd1(pa, pb);
#pragma unroll
    for (int t = 0; t < 9; ++t) a0[t] = rd(pa, pb, t);
#pragma unroll 1
    for (int kk = 0; kk < NK; kk += 2) {
      addr(lA, lB, kk + 1, pa, pb);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        mfma1(t, a0[t], b0);
        if (t == 0) b1 = rd1(pa, pb);
        a1[t] = rd(pa, pb, t);
        __builtin_amdgcn_sched_barrier(0);
      }
      addr(lA, lB, kk + 2 < NK ? kk + 2 : 0, pa, pb);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        mfma1(t, a1[t], b1);
        if (t == 0) b0 = rd1(pa, pb);
        a0[t] = rd(pa, pb, t);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  }

  // cross-wave (K-split) reduction through LDS in fixed wave order, then one slab per workgroup:
  // slab = (cit*NCO + cot)*KS + ks ; layout [tap][ci(32)][co(32)]
  static_assert(TSZ >= 9 * 1024, "LDS too small for the slab");
  __syncthreads();
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = (i & 3) + 8 * (i >> 2) + 4 * half;
          float* q = lds + t * 1024 + row * 32 + l31;
          *q = wv ? *q + acc[t][i] : acc[t][i];
        }
    }
    __syncthreads();
  }
  float4* out = reinterpret_cast<float4*>(p.partial + (int64_t)g * p.partial_gstride +
                                          ((int64_t)((cit * NCO + cot) * KS + ks)) * (9 * 1024));
  const float4* l4 = reinterpret_cast<const float4*>(lds);
#pragma unroll
  for (int j = 0; j < 9; ++j) out[tid + j * VV_WG] = l4[tid + j * VV_WG];
}

// ---------------------------------------------------------------------------------------------------------------
// Winograd F(2x2,3x3) weight gradient of the 3x3 / stride 1 / pad 1 convolution:
//     dU[xi,nu][ci][co] = sum over 2x2 output tiles of  V[xi,nu][tile][ci] * dM[xi,nu][tile][co],     dg = G^T dU G
// with V = B^T d B (the forward's input transform of the 4x4 input patch) and dM = A dY A^T (the 2x2 output gradients of the
// tile): 16 multiply-adds per tile (4 pixels) and (ci, co) pair instead of 36 -- 2.25x fewer MFMA cycles than the direct form
// above.  Same (32 ci x 32 co x k-split) workgroups and slabs, but the GEMM-K index is the 2x2 tile: lanes 0-31 take tile 2j,
// lanes 32-63 tile 2j+1, lane&31 = channel, so both operands are built per lane from ds_read_b32 (conflict free: consecutive
// lanes = consecutive channels).
// 256 threads, wave = xi.  A wave keeps the 4 GEMMs (nu = 0..3) of its xi row in 64 accumulator registers and walks ALL 2x2
// tiles of a staged unit, so three workgroups share a CU (<= 168 registers per lane, <= 53 KB of LDS) and every SIMD has three
// independent instruction streams: operand reads, transforms and barrier waits of one run under the MFMAs of the others.
// (Round 1 / 2 history, profiles/README.md: forms with all 16 GEMMs per wave -- 256 accumulators, one wave per SIMD -- and with
// 8 GEMMs per wave -- two per SIMD -- ran at 0.37-0.49 of the fp32 MFMA peak; this one at 0.50-0.61.)
//   staged unit = 64 pixels (16 tiles = 8 k-steps of a tile pair) of full image width: UH rows x W x UNI images; the k-split
//   tile (TH x W x NI, what vv_wgrad_ntiles counts) is a run of (TH / UH) * (NI / UNI) units.  Two LDS buffers per workgroup;
//   the loads of the next unit go out under the first MFMAs of this one, their LDS writes under the last ones.
//   Operands of k-step j+1 are built under the MFMAs of step j into the other of two register sets: 6 ds_read2_b32, 8 packed
//   VALU instructions per 4 MFMAs.  Signs that would cost an instruction are folded into the epilogue: row xi = 3 uses +dY[1][.]
//   instead of -dY[1][.], column nu = 3 uses +t1 instead of -t1, so acc[nu] = s(xi) s(nu) dU[xi][nu] with s(3) = -1.
//   Epilogue: each wave folds its columns (dU G), the four xi rows meet in LDS, wave w finishes accumulator rows 4w..4w+3 of
//   all 9 taps (G^T .) and writes them to the workgroup's slab -- the layout vv_wgrad_reduce expects.
template <int TH, int TW, int NI, int UH, int UNI>
__global__ void __launch_bounds__(256, 3)
wgrad_wino_kernel(const vv_wgrad_params p, const int NT, const int NCI, const int NCO, const int total, const int nper) {
  constexpr int NTH = 256, W_ = TW;
  constexpr int AHH = UH + 2, AHW = W_ + 2;
  constexpr int ASZ = UNI * AHH * AHW * 32, BSZ = UNI * UH * W_ * 32, USZ = ASZ + BSZ;      // floats
  static_assert(UNI * UH * W_ == 64, "a staged unit is 64 pixels");
  constexpr int RB = TH / UH, UPT = RB * (NI / UNI);                       // row blocks / units per k-split tile
  constexpr int TXT = W_ / 2, TYT = UH / 2, TPI = TXT * TYT;               // 2x2 tiles per image of a unit
  constexpr int NKS = 8;                                                   // k-steps (tile pairs) per unit
  constexpr int EXF = 4 * 3 * 16 * 64;                                     // epilogue exchange [xi][b][16 regs][64 lanes]
  constexpr int LSZ = 2 * USZ > EXF ? 2 * USZ : EXF;
  static_assert(LSZ * 4 * 3 <= 160 * 1024, "three workgroups per CU");
  constexpr int NIA = UNI * AHH * AHW * 8, NIB = 64 * 8;                   // float4 items
  constexpr int NPA = (NIA + NTH - 1) / NTH, NPB = NIB / NTH, NP = NPA + NPB;
  constexpr int NSLOT = NKS * 4, C0 = NSLOT - NP - 1;
  static_assert(C0 >= NP + 8, "not enough MFMA slots between load issue and commit");
  constexpr unsigned OOB = 0x80000000u;
  constexpr bool ROWDYN = RB > 1;        // units that are not whole images: top / bottom halo row validity changes per unit
  __shared__ float lds[LSZ];

  int w = vv_xcd_remap(blockIdx.x, nper);
  if (w >= total) return;
  const int KS = p.ksplit;
  const int ks = w % KS; w /= KS;
  const int cot = w % NCO; w /= NCO;
  const int cit = w % NCI;
  const int g = w / NCI;

  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int xi = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H;
  const int tpi = H / TH;                                                  // k-split tiles per image (full width)

  // ---- sources: layer input (re-materialised: BatchNorm+ReLU / concat) and dy
  const VVSrc sa = vv_make_src(p, g, H, W_);
  const int q8 = tid & 7, cA = cit * 32 + q8 * 4, cB = cot * 32 + q8 * 4;
  const bool cok = cA < p.CinP;
  const bool act = (sa.mode == VV_IN_ACT) || (sa.mode == VV_IN_CAT && cA < sa.csplit);
  float4 sca = make_float4(1.f, 1.f, 1.f, 1.f), scb = make_float4(0.f, 0.f, 0.f, 0.f);
  if (act && cok) {
    sca = *reinterpret_cast<const float4*>(sa.a + cA);
    scb = *reinterpret_cast<const float4*>(sa.b + cA);
  }
  const bool second = __builtin_amdgcn_readfirstlane((int)((sa.mode == VV_IN_CAT) && cit * 32 >= sa.csplit)) != 0;
  const float* baseA = second ? sa.p1 + sa.co1 - sa.csplit : sa.p0 + sa.co0;
  const int csA = second ? sa.cs1 : sa.cs0;
  const float* baseB = p.dy.ptr + (int64_t)g * p.dy.gstride + p.dy.coff;
  const int csB = p.dy.cstride;

  // ---- staging items (once per workgroup): byte offset inside the source relative to the unit's origin pixel, LDS slot
  unsigned voffA[NPA], voffB[NPB];
  int slotA[NPA];
  unsigned topm = 0, botm = 0;           // items in the top / bottom halo row (ROWDYN)
  int imA[NPA];                          // image of the item inside the unit (UNI > 1)
#pragma unroll
  for (int k = 0; k < NPA; ++k) {
    const int it = tid + k * NTH;
    const int hp = it >> 3;
    const int hx = hp % AHW, t = hp / AHW;
    const int hy = t % AHH, im = t / AHH;
    bool ok = (NIA % NTH == 0 || it < NIA) && cok && (unsigned)(hx - 1) < (unsigned)W_;
    if (!ROWDYN) ok = ok && hy != 0 && hy != AHH - 1;                      // whole images: the halo rows are padding
    topm |= (hy == 0) ? (1u << k) : 0u;
    botm |= (hy == AHH - 1) ? (1u << k) : 0u;
    imA[k] = im;
    voffA[k] = ok ? (unsigned)(((im * H + hy) * W_ + hx) * csA + cA) * 4u : OOB;
    slotA[k] = (NIA % NTH == 0 || it < NIA) ? hp * 32 + q8 * 4 : -1;
  }
#pragma unroll
  for (int k = 0; k < NPB; ++k) {
    const int px = (tid + k * NTH) >> 3;                                   // pixel of the unit: [im][y][x]
    const int im = px / (UH * W_), rem = px % (UH * W_);
    voffB[k] = (unsigned)((im * H * W_ + rem) * csB + cB) * 4u;
  }

  // ---- unit walk: k-split tiles ks, ks + KS, ... < NT, UPT units each
  int pt_n = ks, u_n = 0;                // the unit whose loads are issued next
  float4 r[NP];
  unsigned validA = 0;                   // of the unit in r[] (commit applies the activation to real pixels only)
  __amdgpu_buffer_rsrc_t rsA, rsB;
  unsigned kill = 0;
  bool liveB[NPB];
  auto begin_unit = [&]() {              // scalar part of the next unit's addresses; afterwards pt_n / u_n point one further
    const bool live = pt_n < NT;
    const int pt = live ? pt_n : ks;
    const int img0 = (pt / tpi) * NI + (u_n / RB) * UNI;
    const int uy0 = (pt % tpi) * TH + (u_n % RB) * UH;
    // origin pixel of the halo tile: (img0, uy0 - 1, -1); may lie before the tensor (top row of image 0): 64-bit base
    const int64_t oa = ((int64_t)(img0 * H + uy0 - 1) * W_ - 1) * csA;
    const int64_t ob = ((int64_t)(img0 * H + uy0) * W_) * csB;
    rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(baseA + oa), 0, 0x7FFFFFFF, 0x00020000);
    rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(baseB + ob), 0, 0x7FFFFFFF, 0x00020000);
    const bool img_ok = live && img0 < p.B;                                // UNI == 1: the whole unit
    kill = img_ok ? 0u : 0xFFFFFFFFu;
    if (ROWDYN) {
      if (uy0 == 0) kill |= topm;
      if (uy0 + UH == H) kill |= botm;
    }
    if (UNI > 1) {
#pragma unroll
      for (int k = 0; k < NPA; ++k)
        if (img0 + imA[k] >= p.B) kill |= 1u << k;
    }
#pragma unroll
    for (int k = 0; k < NPB; ++k) {
      const int im = ((tid + k * NTH) >> 3) / (UH * W_);
      liveB[k] = live && img0 + im < p.B;
    }
    validA = 0;
    if (++u_n == UPT) { u_n = 0; pt_n += KS; }
  };
  auto load_piece = [&](auto K) {
    constexpr int k = K.value;
    if constexpr (k < NPA) {
      const bool ok = !((kill >> k) & 1u) && voffA[k] != OOB;
      validA |= ok ? (1u << k) : 0u;
      const v4f v = __builtin_amdgcn_raw_buffer_load_b128(rsA, ok ? voffA[k] : OOB, 0, 0);
      r[k] = make_float4(v.x, v.y, v.z, v.w);
    } else {
      const v4f v = __builtin_amdgcn_raw_buffer_load_b128(rsB, liveB[k - NPA] ? voffB[k - NPA] : OOB, 0, 0);
      r[k] = make_float4(v.x, v.y, v.z, v.w);
    }
  };
  auto commit_piece = [&](auto K, float* buf) {
    constexpr int k = K.value;
    if constexpr (k < NPA) {
      if (NIA % NTH == 0 || k < NPA - 1 || slotA[k] >= 0) {
        float4 v = r[k];
        if (act && ((validA >> k) & 1u)) v = vv_act4(v, sca, scb);
        *reinterpret_cast<float4*>(buf + slotA[k]) = v;
      }
    } else {
      *reinterpret_cast<float4*>(buf + ASZ + ((tid + (k - NPA) * NTH) >> 3) * 32 + q8 * 4) = r[k];
    }
  };

  // ---- operands.  B^T row xi: d[a1] + sg d[a2];  A row xi: c0 dY[0][.] + c1 dY[1][.]  (xi = 3: +dY[1] instead of -dY[1], folded)
  const int a1 = xi == 0 ? 0 : (xi == 2 ? 2 : 1), a2 = xi == 3 ? 3 : (xi == 2 ? 1 : 2);
  const float sg = xi == 1 ? 1.f : -1.f;
  const float c0 = xi == 3 ? 0.f : 1.f, c1 = xi == 0 ? 0.f : (xi == 2 ? -1.f : 1.f);
  const int ao1 = a1 * AHW * 32, ao2 = a2 * AHW * 32;
  // lane origin inside a unit buffer: tile pair j -> tiles 2j + half; TXT is even, so `half` only moves one tile column
  const int laneA = half * 64 + l31, laneB = ASZ + half * 64 + l31;
  v2f V03[2], V12[2], T[2], M12[2];      // two register sets: step j+1 is built while the MFMAs of step j run
  v2f pa01, pa23, pb01, pb23, dq0, dq1;
  auto read_step = [&](const float* ub, auto J) {
    constexpr int j = J.value;
    constexpr int tt = 2 * j;
    constexpr int im = tt / TPI, rem = tt % TPI, tyl = rem / TXT, txl = rem % TXT;
    const float* pA = ub + laneA + ((im * AHH + 2 * tyl) * AHW + 2 * txl) * 32;
    const float* pB = ub + laneB + ((im * UH + 2 * tyl) * W_ + 2 * txl) * 32;
    pa01 = (v2f){pA[ao1], pA[ao1 + 32]};
    pa23 = (v2f){pA[ao1 + 64], pA[ao1 + 96]};
    pb01 = (v2f){pA[ao2], pA[ao2 + 32]};
    pb23 = (v2f){pA[ao2 + 64], pA[ao2 + 96]};
    dq0 = (v2f){pB[0], pB[32]};
    dq1 = (v2f){pB[W_ * 32], pB[W_ * 32 + 32]};
  };
  auto xform_step = [&](auto SET) {
    constexpr int st = SET.value;
    const v2f r01 = __builtin_elementwise_fma((v2f){sg, sg}, pb01, pa01);
    const v2f r23 = __builtin_elementwise_fma((v2f){sg, sg}, pb23, pa23);
    V03[st] = r01 - r23;                                // (r0 - r2, r1 - r3)
    V12[st] = vv_pk_lo_pm_hi(r23, r01);                 // (r2 + r1, r2 - r1)
    const v2f t = __builtin_elementwise_fma((v2f){c1, c1}, dq1, c0 * dq0);
    T[st] = t;                                          // (M0, -M3)
    M12[st] = vv_pk_lo_pm_hi(t, t);                     // (t0 + t1, t0 - t1)
  };

  v16f acc[4];
#pragma unroll
  for (int n = 0; n < 4; ++n)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;

  // ---- first unit
  begin_unit();
  vv_static_for<0, NP>([&](auto K) { load_piece(K); });
  vv_static_for<0, NP>([&](auto K) { commit_piece(K, lds); });
  __syncthreads();
  const int nunits = ((NT - ks + KS - 1) / KS) * UPT;
  int cur = 0;
  for (int un = 0; un < nunits; ++un) {
    const float* ub = lds + cur * USZ;
    float* nb = lds + (cur ^ 1) * USZ;
    begin_unit();                        // the next unit (an unmapped one after the last: loads return zeros, harmless)
    read_step(ub, std::integral_constant<int, 0>{});
    xform_step(std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    vv_static_for<0, NKS>([&](auto JJ) {
      constexpr int j = JJ.value, st = j & 1;
      constexpr bool more = j + 1 < NKS;
      vv_static_for<0, 4>([&](auto NN) {
        constexpr int nu = NN.value, slot = j * 4 + nu;
        const float va = nu == 0 ? V03[st].x : (nu == 1 ? V12[st].x : (nu == 2 ? V12[st].y : V03[st].y));
        const float mb = nu == 0 ? T[st].x : (nu == 1 ? M12[st].x : (nu == 2 ? M12[st].y : T[st].y));
        acc[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(va, mb, acc[nu], 0, 0, 0);
        if constexpr (more && nu == 0) read_step(ub, std::integral_constant<int, j + 1>{});
        if constexpr (more && nu == 2) xform_step(std::integral_constant<int, st ^ 1>{});
        if constexpr (slot < NP) load_piece(std::integral_constant<int, slot>{});
        else if constexpr (slot >= C0 && slot < C0 + NP) commit_piece(std::integral_constant<int, slot - C0>{}, nb);
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    __syncthreads();                     // next buffer complete, current buffer no longer read
    cur ^= 1;
  }

  // ---- epilogue.  dg = G^T dU G,  G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1].  acc[nu] = s(xi) s(nu) dU[xi][nu], s(3) = -1.
  const float sx = xi == 3 ? -1.f : 1.f;
  float* ex = lds + lane;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float u0 = acc[0][i], u1 = acc[1][i], u2 = acc[2][i], u3 = -acc[3][i];
    const float hs = 0.5f * (u1 + u2);
    ex[((xi * 3 + 0) * 16 + i) * 64] = sx * (u0 + hs);
    ex[((xi * 3 + 1) * 16 + i) * 64] = sx * (0.5f * (u1 - u2));
    ex[((xi * 3 + 2) * 16 + i) * 64] = sx * (u3 + hs);
  }
  __syncthreads();
  float* out = p.partial + (int64_t)g * p.partial_gstride + ((int64_t)((cit * NCO + cot) * KS + ks)) * (9 * 1024);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = xi * 4 + j;
    const int row = j + 8 * xi + 4 * half;
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const float d0 = ex[((0 * 3 + b) * 16 + i) * 64], d1 = ex[((1 * 3 + b) * 16 + i) * 64];
      const float d2 = ex[((2 * 3 + b) * 16 + i) * 64], d3 = ex[((3 * 3 + b) * 16 + i) * 64];
      const float hs = 0.5f * (d1 + d2);
      out[(0 * 3 + b) * 1024 + row * 32 + l31] = d0 + hs;
      out[(1 * 3 + b) * 1024 + row * 32 + l31] = 0.5f * (d1 - d2);
      out[(2 * 3 + b) * 1024 + row * 32 + l31] = d3 + hs;
    }
  }
}

__global__ void __launch_bounds__(VV_WG)
wgrad_reduce_kernel(const int kind, const int Cin, const int Cout, const int NCO, const int nslab,
                    const float* __restrict__ partial, const int64_t partial_gstride, float* __restrict__ grad,
                    const int64_t grad_gstride) {
  const int quarter = blockIdx.x & 3;
  const int tap = (blockIdx.x >> 2) % 9;
  const int tile = (blockIdx.x >> 2) / 9;
  const int g = blockIdx.y;
  const int cit = tile / NCO, cot = tile % NCO;
  const int e = quarter * VV_WG + threadIdx.x;
  const float* src = partial + (int64_t)g * partial_gstride + (int64_t)tile * nslab * (9 * 1024) + tap * 1024 + e;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = 0;
  for (; k + 4 <= nslab; k += 4) {
    s0 += src[(int64_t)(k + 0) * (9 * 1024)];
    s1 += src[(int64_t)(k + 1) * (9 * 1024)];
    s2 += src[(int64_t)(k + 2) * (9 * 1024)];
    s3 += src[(int64_t)(k + 3) * (9 * 1024)];
  }
  for (; k < nslab; ++k) s0 += src[(int64_t)k * (9 * 1024)];
  const float s = (s0 + s1) + (s2 + s3);
  float* dst = grad + (int64_t)g * grad_gstride;
  const int ci = cit * 32 + (e >> 5), co = cot * 32 + (e & 31);
  if (ci < Cin && co < Cout) {
    if (kind == VV_CONV3) dst[((int64_t)co * Cin + ci) * 9 + tap] = s;
    else dst[((int64_t)ci * Cout + co) * 9 + tap] = s;
  }
}

// The same reduction for SEVERAL layers in one launch (all weight gradients of a gradient bucket: up to 13 launches of 8-20 us
// become one): blockIdx.x runs over the concatenated block lists of the table's entries.
__global__ void __launch_bounds__(VV_WG)
wgrad_reduce_grouped_kernel(const vv_reduce_entry* __restrict__ table, const int n, const float* __restrict__ partial,
                            const int64_t partial_gstride, float* __restrict__ grads) {
  int e = 0;
  for (int i = 1; i < n; ++i) e = (int)blockIdx.x >= table[i].block_start ? i : e;      // entries are sorted by block_start
  const vv_reduce_entry t = table[e];
  const int bx = blockIdx.x - t.block_start;
  const int quarter = bx & 3;
  const int tap = (bx >> 2) % 9;
  const int tile = (bx >> 2) / 9;
  const int g = blockIdx.y;
  const int cit = tile / t.NCO, cot = tile % t.NCO;
  const int el = quarter * VV_WG + threadIdx.x;
  const float* src = partial + (int64_t)g * partial_gstride + t.part_off + (int64_t)tile * t.nslab * (9 * 1024) + tap * 1024 + el;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = 0;
  for (; k + 4 <= t.nslab; k += 4) {
    s0 += src[(int64_t)(k + 0) * (9 * 1024)];
    s1 += src[(int64_t)(k + 1) * (9 * 1024)];
    s2 += src[(int64_t)(k + 2) * (9 * 1024)];
    s3 += src[(int64_t)(k + 3) * (9 * 1024)];
  }
  for (; k < t.nslab; ++k) s0 += src[(int64_t)k * (9 * 1024)];
  const float s = (s0 + s1) + (s2 + s3);
  float* dst = grads + t.grad_off + (int64_t)g * t.grad_gstride;
  const int ci = cit * 32 + (el >> 5), co = cot * 32 + (el & 31);
  if (ci < t.Cin && co < t.Cout) {
    if (t.kind == VV_CONV3) dst[((int64_t)co * t.Cin + ci) * 9 + tap] = s;
    else dst[((int64_t)ci * t.Cout + co) * 9 + tap] = s;
  }
}

// VV_WGRADT_TILE64 (default 1): the transposed convs' weight gradient walks 64-pixel k-split tiles on the double-buffered pipeline;
// 0 = the 128-pixel tiles / one LDS buffer of rounds 1 - 5 (A/B on one box; read once per process)
inline bool wgradT_tile64() {
  static const int v = [] { const char* e = getenv("VV_WGRADT_TILE64"); return (e && e[0] == '0') ? 0 : 1; }();
  return v != 0;
}

struct WGeo { int TH, TW, NI; };
inline bool wgeo(int kind, int H, int W, WGeo* t) {
  if (H != W) return false;
  if (kind == VV_CONV3) {
    if (H == 32) { *t = {8, 32, 1}; return true; }
    if (H == 16) { *t = {16, 16, 1}; return true; }
    if (H == 8) { *t = {8, 8, 2}; return true; }
    if (H == 4) { *t = {4, 4, 8}; return true; }
  } else if (wgradT_tile64()) {
    if (H == 16) { *t = {4, 16, 1}; return true; }
    if (H == 8) { *t = {4, 8, 2}; return true; }
    if (H == 4) { *t = {4, 4, 4}; return true; }
  } else {
    if (H == 16) { *t = {8, 16, 1}; return true; }
    if (H == 8) { *t = {8, 8, 2}; return true; }
    if (H == 4) { *t = {4, 4, 8}; return true; }
  }
  return false;
}

template <int TH, int TW, int NI, int UH, int UNI>
int launch_ww(const vv_wgrad_params* p, hipStream_t st) {
  if (p->W != TW) return VV_ERR_UNSUPPORTED;
  const int NT = ((p->B + NI - 1) / NI) * (p->H / TH) * (p->W / TW);
  const int NCI = (p->CinP + 31) / 32, NCO = p->Cout / 32;
  const int total = p->G * NCI * NCO * p->ksplit;
  const int nper = (total + 7) / 8;
  VV_LAUNCH((wgrad_wino_kernel<TH, TW, NI, UH, UNI>), dim3(nper * 8), dim3(256), 0, st, *p, NT, NCI, NCO, total, nper);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

template <int TH, int TW, int NI, int KIND>
int launch_w(const vv_wgrad_params* p, hipStream_t st) {
  const int NT = ((p->B + NI - 1) / NI) * (p->H / TH) * (p->W / TW);
  const int NCI = (p->CinP + 31) / 32, NCO = p->Cout / 32;
  const int total = p->G * NCI * NCO * p->ksplit;
  const int nper = (total + 7) / 8;
  VV_LAUNCH((wgrad_mfma_kernel<TH, TW, NI, KIND>), dim3(nper * 8), dim3(VV_WG), 0, st, *p, NT, NCI, NCO, total,
                     nper);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

}  // namespace

extern "C" int vv_wgrad_ntiles(int32_t kind, int32_t B, int32_t H, int32_t W) {
  WGeo t;
  if (!wgeo(kind, H, W, &t)) return -1;
  return ((B + t.NI - 1) / t.NI) * (H / t.TH) * (W / t.TW);
}

extern "C" int vv_wgrad_mfma(const vv_wgrad_params* p, vv_stream stream) {
  if (!p || !p->src0.ptr || !p->dy.ptr || !p->partial) return VV_ERR_BAD_ARG;
  if (p->Cout % 32 || p->ksplit < 1) return VV_ERR_BAD_ARG;
  if (p->kind == VV_CONV3 && (p->in_mode == VV_IN_POOL || p->in_mode == VV_IN_CUBE))
    return VV_ERR_UNSUPPORTED;    // feed the materialised tensor (vv_pool_act / vv_cube_erase) as VV_IN_PLAIN
  hipStream_t st = (hipStream_t)stream;
  if (p->kind == VV_CONV3 && (p->pad0 & 256)) {           // Winograd F(2x2,3x3) form (same k-split tiles, same slabs)
    switch (p->H) {
      case 32: return launch_ww<8, 32, 1, 2, 1>(p, st);
      case 16: return launch_ww<16, 16, 1, 4, 1>(p, st);
      case 8: return launch_ww<8, 8, 2, 8, 1>(p, st);
      case 4: return launch_ww<4, 4, 8, 4, 4>(p, st);
    }
    return VV_ERR_UNSUPPORTED;
  }
  if (p->kind == VV_CONV3) {
    switch (p->H) {
      case 32: return launch_w<8, 32, 1, VV_CONV3>(p, st);
      case 16: return launch_w<16, 16, 1, VV_CONV3>(p, st);
      case 8: return launch_w<8, 8, 2, VV_CONV3>(p, st);
      case 4: return launch_w<4, 4, 8, VV_CONV3>(p, st);
    }
  } else if (wgradT_tile64()) {
    switch (p->H) {
      case 16: return launch_w<4, 16, 1, VV_CONVT_FWD>(p, st);
      case 8: return launch_w<4, 8, 2, VV_CONVT_FWD>(p, st);
      case 4: return launch_w<4, 4, 4, VV_CONVT_FWD>(p, st);
    }
  } else {
    switch (p->H) {
      case 16: return launch_w<8, 16, 1, VV_CONVT_FWD>(p, st);
      case 8: return launch_w<8, 8, 2, VV_CONVT_FWD>(p, st);
      case 4: return launch_w<4, 4, 8, VV_CONVT_FWD>(p, st);
    }
  }
  return VV_ERR_UNSUPPORTED;
}

extern "C" int vv_wgrad_reduce(int32_t kind, int32_t G, int32_t Cin, int32_t CinP, int32_t Cout,
                               int32_t nslab_per_tile, const float* partial, int64_t partial_gstride, float* grad,
                               int64_t grad_gstride, vv_stream stream) {
  if (!partial || !grad) return VV_ERR_BAD_ARG;
  const int NCI = (CinP + 31) / 32, NCO = Cout / 32;
  VV_LAUNCH(wgrad_reduce_kernel, dim3(NCI * NCO * 9 * 4, G), dim3(VV_WG), 0, (hipStream_t)stream, kind, Cin, Cout,
                     NCO, nslab_per_tile, partial, partial_gstride, grad, grad_gstride);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

extern "C" int vv_wgrad_reduce_grouped(const vv_reduce_entry* table_dev, int32_t nentries, int32_t total_blocks, int32_t G,
                                       const float* partial, int64_t partial_gstride, float* grads, vv_stream stream) {
  if (!table_dev || !partial || !grads || nentries <= 0 || total_blocks <= 0) return VV_ERR_BAD_ARG;
  VV_LAUNCH(wgrad_reduce_grouped_kernel, dim3(total_blocks, G), dim3(VV_WG), 0, (hipStream_t)stream, table_dev, nentries, partial,
            partial_gstride, grads);
  VV_CHECK_LAUNCH();
  return VV_OK;
}
