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
  // 3x3 convolutions: two LDS tile buffers, the staging of tile t+1 (loads, BN+ReLU, LDS writes) is spread over the MFMA
  // loop of tile t.  Transposed convolutions (larger dy halo tiles, 4 % of the weight-gradient time) keep one buffer.
  constexpr bool DB = !CT && 2 * TSZ * 4 <= 160 * 1024;
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
      stA.begin(sa, img0, ty0 - 1, tx0 - 1, cit * 32, tid, p.CinP, live);
      stB.begin(sb, img0, ty0, tx0, cot * 32, tid, 1 << 30, live);
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
// above.  Same workgroup / staging structure (32 ci x 32 co x k-split, 4 waves over the tiles of each staged pixel tile, two
// LDS buffers, staging of the next tile interleaved into the MFMA stream), but the GEMM-K index is the 2x2 tile: lanes 0-31
// take tile 2j, lanes 32-63 tile 2j+1, lane&31 = channel, so both operands are built per lane from ds_read_b32 (conflict
// free: consecutive lanes = consecutive channels): one xi row of V (8 patch reads, 8 adds) and of dM (<= 5 ops) per group
// of 4 MFMAs, written into the registers of the row that was consumed one group earlier.  The 16 accumulators (256
// registers) are folded to the 9 filter taps in the epilogue (dg = G^T dU G is linear, so it commutes with the k-split
// sum) and leave through the same slab / vv_wgrad_reduce path as the direct kernel.
template <int TH, int TW, int NI>
__global__ void __launch_bounds__(VV_WG, 1)
wgrad_wino_kernel(const vv_wgrad_params p, const int NT, const int NCI, const int NCO, const int total, const int nper) {
  constexpr int AHH = TH + 2, AHW = TW + 2;
  constexpr int ASZ = NI * AHH * AHW * 32, BSZ = NI * TH * TW * 32, TSZ = ASZ + BSZ;
  static_assert(2 * TSZ * 4 <= 160 * 1024, "two LDS buffers");
  constexpr int TXT = TW / 2, TYT = TH / 2, TPI = TXT * TYT;       // 2x2 tiles per image inside a staged pixel tile
  constexpr int NTL = NI * TPI;                                    // tiles per staged pixel tile
  constexpr int TPWV = NTL / 4;                                    // per wave
  constexpr int NKS = TPWV / 2;                                    // k-steps (tile pairs) per wave and staged tile
  static_assert(NTL % 8 == 0, "tile pairs per wave");
  __shared__ float lds[2 * TSZ];

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
  sb.mode = VV_IN_PLAIN; sb.SH = H; sb.SW = W; sb.B = p.B;

  v16f acc[16];
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  VVStagerB<NI, AHH, AHW, 32, 32> stA;
  VVStagerB<NI, TH, TW, 32, 32> stB;
  stA.init(sa, -1, tid);
  stB.init(sb, 0, tid);
  constexpr int NPA = decltype(stA)::NIT, NPB = decltype(stB)::NIT, NP = NPA + NPB;
  constexpr int NSLOT = NKS * 16, C0 = NSLOT - NP - 4;
  static_assert(C0 >= NP, "not enough MFMA slots between load issue and commit");
  auto begin_tile = [&](const int pt, const bool live) {
    const int img0 = (pt / tpi) * NI;
    const int trem = pt % tpi;
    const int ty0 = (trem / tilesX) * TH, tx0 = (trem % tilesX) * TW;
    stA.begin(sa, img0, ty0 - 1, tx0 - 1, cit * 32, tid, p.CinP, live);
    stB.begin(sb, img0, ty0, tx0, cot * 32, tid, 1 << 30, live);
  };

  // operand rows: V[xi][nu] / M[xi][nu] of the k-step in flight; row xi of the NEXT step overwrites row xi of this one
  float V[4][4], M[4][4], dq[2][2];
  const float *pA, *pB;                 // patch origin / dy origin of the step whose rows are being generated
  auto step_addr = [&](const float* tA, const float* tB, const int j) {
    const int tt = wave * TPWV + 2 * j + half;
    const int im = tt / TPI, rem = tt % TPI;
    const int tyl = rem / TXT, txl = rem % TXT;
    pA = tA + ((im * AHH + 2 * tyl) * AHW + 2 * txl) * 32 + l31;
    pB = tB + ((im * TH + 2 * tyl) * TW + 2 * txl) * 32 + l31;
  };
  auto read_dy = [&]() {
    dq[0][0] = pB[0]; dq[0][1] = pB[32];
    dq[1][0] = pB[TW * 32]; dq[1][1] = pB[TW * 32 + 32];
  };
  // Row xi of V = B^T d B and of dM = A dY A^T is produced in two stages one MFMA group apart: read_row issues its 8 patch
  // reads into pr[], xform_row (a group later, the reads have landed) does the arithmetic.
  //   B^T rows: d0-d2, d1+d2, d2-d1, d1-d3          A rows: (1,0) (1,1) (1,-1) (0,-1)
  float pr[8];
  auto read_row = [&](auto XI, auto HALF) {             // HALF 0: patch row a1, HALF 1: patch row a2
    constexpr int xi = XI.value;
    constexpr int a1 = xi == 0 ? 0 : (xi == 2 ? 2 : 1), a2 = xi == 3 ? 3 : (xi == 2 ? 1 : 2);
    constexpr int a = HALF.value ? a2 : a1;
#pragma unroll
    for (int b = 0; b < 4; ++b) pr[HALF.value * 4 + b] = pA[(a * AHW + b) * 32];
  };
  auto xform_row = [&](auto XI) {
    constexpr int xi = XI.value;
    float r[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) r[b] = xi == 1 ? pr[b] + pr[4 + b] : pr[b] - pr[4 + b];
    V[xi][0] = r[0] - r[2];
    V[xi][1] = r[1] + r[2];
    V[xi][2] = r[2] - r[1];
    V[xi][3] = r[1] - r[3];
    float t0, t1;                       // T[xi][q] = sum_p A[xi][p] dY[p][q]
    if constexpr (xi == 0) { t0 = dq[0][0]; t1 = dq[0][1]; }
    else if constexpr (xi == 1) { t0 = dq[0][0] + dq[1][0]; t1 = dq[0][1] + dq[1][1]; }
    else if constexpr (xi == 2) { t0 = dq[0][0] - dq[1][0]; t1 = dq[0][1] - dq[1][1]; }
    else { t0 = -dq[1][0]; t1 = -dq[1][1]; }
    M[xi][0] = t0;
    M[xi][1] = t0 + t1;
    M[xi][2] = t0 - t1;
    M[xi][3] = -t1;
  };
  auto gen_row = [&](auto XI) {                         // both stages back to back (start of a staged tile only)
    read_row(XI, std::integral_constant<int, 0>{});
    read_row(XI, std::integral_constant<int, 1>{});
    xform_row(XI);
  };

  begin_tile(ks, true);
  vv_static_for<0, NPA>([&](auto K) { stA.template load_piece<K.value>(sa, -1, tid); });
  vv_static_for<0, NPB>([&](auto K) { stB.template load_piece<K.value>(sb, 0, tid); });
  vv_static_for<0, NPA>([&](auto K) { stA.template commit_piece<K.value>(lds, tid); });
  vv_static_for<0, NPB>([&](auto K) { stB.template commit_piece<K.value>(lds + ASZ, tid); });
  __syncthreads();
  int cur = 0;
  for (int pt = ks; pt < NT; pt += KS) {
    const float* tA = lds + cur * TSZ;
    const float* tB = tA + ASZ;
    float* nA = lds + (cur ^ 1) * TSZ;
    float* nB = nA + ASZ;
    const bool live = pt + KS < NT;
    begin_tile(live ? pt + KS : ks, live);
    // rows 0..2 of the first step, and the reads of its row 3 (transformed under the first MFMA group)
    step_addr(tA, tB, 0);
    read_dy();
    gen_row(std::integral_constant<int, 0>{});
    gen_row(std::integral_constant<int, 1>{});
    gen_row(std::integral_constant<int, 2>{});
    read_row(std::integral_constant<int, 3>{}, std::integral_constant<int, 0>{});
    read_row(std::integral_constant<int, 3>{}, std::integral_constant<int, 1>{});
    vv_static_for<0, NKS>([&](auto JJ) {
      constexpr int j = JJ.value;
      constexpr bool more = j + 1 < NKS;
      vv_static_for<0, 16>([&](auto SS) {
        constexpr int sl = SS.value, grp = sl >> 2, nu = sl & 3, slot = j * 16 + sl;
        acc[sl] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[grp][nu], M[grp][nu], acc[sl], 0, 0, 0);
        // under group g: the row whose reads were issued one group earlier is transformed -- row 3 of this step (g = 0) or
        // row g-1 of the next step -- into registers consumed at least one group ago; then the reads of the following row.
        constexpr int rowa = grp == 0 ? 3 : grp - 1, rowb = grp;         // rowb: row of step j+1 whose reads are issued
        if constexpr (nu == 0 && (grp == 0 || more)) xform_row(std::integral_constant<int, rowa>{});
        if constexpr (more) {
          if constexpr (nu == 1) {
            if constexpr (grp == 0) { step_addr(tA, tB, j + 1); read_dy(); }
            read_row(std::integral_constant<int, rowb>{}, std::integral_constant<int, 0>{});
          }
          if constexpr (nu == 2) read_row(std::integral_constant<int, rowb>{}, std::integral_constant<int, 1>{});
        }
        if constexpr (slot < NPA) stA.template load_piece<slot>(sa, -1, tid);
        else if constexpr (slot < NP) stB.template load_piece<slot - NPA>(sb, 0, tid);
        else if constexpr (slot >= C0 && slot < C0 + NPA) stA.template commit_piece<slot - C0>(nA, tid);
        else if constexpr (slot >= C0 + NPA && slot < C0 + NP) stB.template commit_piece<slot - C0 - NPA>(nB, tid);
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    __syncthreads();                        // next buffer complete, current buffer no longer read
    cur ^= 1;
  }

  // ---- epilogue: dg = G^T dU G per (ci, co),  G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1]; then the direct kernel's slab path
  v16f tap[9];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    float c[3][4];                          // G^T dU  (rows)
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      const float u0 = acc[0 * 4 + nu][i], u1 = acc[1 * 4 + nu][i], u2 = acc[2 * 4 + nu][i], u3 = acc[3 * 4 + nu][i];
      c[0][nu] = u0 + 0.5f * (u1 + u2);
      c[1][nu] = 0.5f * (u1 - u2);
      c[2][nu] = u3 + 0.5f * (u1 + u2);
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      tap[a * 3 + 0][i] = c[a][0] + 0.5f * (c[a][1] + c[a][2]);
      tap[a * 3 + 1][i] = 0.5f * (c[a][1] - c[a][2]);
      tap[a * 3 + 2][i] = c[a][3] + 0.5f * (c[a][1] + c[a][2]);
    }
  }
  static_assert(2 * TSZ >= 9 * 1024, "LDS too small for the slab");
  __syncthreads();
  for (int wv = 0; wv < 4; ++wv) {
    if (wave == wv) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = (i & 3) + 8 * (i >> 2) + 4 * half;
          float* q = lds + t * 1024 + row * 32 + l31;
          *q = wv ? *q + tap[t][i] : tap[t][i];
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
// Eight-wave form of the Winograd weight gradient: 512 threads, wave = (K quarter kq of the staged tile's 2x2 tiles) x
// (xi half xh).  A wave keeps only the 8 GEMMs of its xi half (128 accumulator registers, <= 256 registers per lane), so
// two waves share every SIMD and one's operand arithmetic / barrier waits run under the other's MFMAs -- the four-wave
// form above has nobody to hide them (measured MFMA utilisation 0.5 even with the staging switched off).
template <int TH, int TW, int NI>
__global__ void __launch_bounds__(512, 1)
wgrad_wino8_kernel(const vv_wgrad_params p, const int NT, const int NCI, const int NCO, const int total, const int nper) {
  constexpr int NTH = 512;
  constexpr int AHH = TH + 2, AHW = TW + 2;
  constexpr int ASZ = NI * AHH * AHW * 32, BSZ = NI * TH * TW * 32, TSZ = ASZ + BSZ;
  static_assert(2 * TSZ * 4 <= 160 * 1024, "two LDS buffers");
  constexpr int TXT = TW / 2, TYT = TH / 2, TPI = TXT * TYT;       // 2x2 tiles per image inside a staged pixel tile
  constexpr int NTL = NI * TPI;                                    // tiles per staged pixel tile
  constexpr int TPWV = NTL / 4;                                    // per K quarter
  constexpr int NKS = TPWV / 2;                                    // k-steps (tile pairs) per wave and staged tile
  static_assert(NTL % 8 == 0, "tile pairs per wave");
  constexpr int LSZ = 2 * TSZ > 4 * 9 * 1024 ? 2 * TSZ : 4 * 9 * 1024;      // epilogue: one tap slab per K quarter
  __shared__ float lds[LSZ];

  int w = vv_xcd_remap(blockIdx.x, nper);
  if (w >= total) return;
  const int KS = p.ksplit;
  const int ks = w % KS; w /= KS;
  const int cot = w % NCO; w /= NCO;
  const int cit = w % NCI;
  const int g = w / NCI;

  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: everything derived from it lives in SGPRs
  const int kq = wave >> 1, xh = wave & 1;
  const int H = p.H, W = p.W;
  const int tilesX = W / TW, tilesY = H / TH, tpi = tilesX * tilesY;

  const VVSrc sa = vv_make_src(p, g, H, W);
  VVSrc sb;
  sb.p0 = p.dy.ptr + (int64_t)g * p.dy.gstride; sb.cs0 = p.dy.cstride; sb.co0 = p.dy.coff;
  sb.a = sb.b = nullptr; sb.p1 = nullptr; sb.cs1 = sb.co1 = 0; sb.chmap = nullptr; sb.csplit = 0;
  sb.mode = VV_IN_PLAIN; sb.SH = H; sb.SW = W; sb.B = p.B;

  v16f acc[2][4];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[x][n][i] = 0.f;

  VVStagerB<NI, AHH, AHW, 32, 32, NTH> stA;
  VVStagerB<NI, TH, TW, 32, 32, NTH> stB;
  stA.init(sa, -1, tid);
  stB.init(sb, 0, tid);
  constexpr int NPA = decltype(stA)::NIT, NPB = decltype(stB)::NIT, NP = NPA + NPB;
  constexpr int NSLOT = NKS * 8, C0 = NSLOT - NP - 2;
  static_assert(C0 >= NP, "not enough MFMA slots between load issue and commit");
  auto begin_tile = [&](const int pt, const bool live) {
    const int img0 = (pt / tpi) * NI;
    const int trem = pt % tpi;
    const int ty0 = (trem / tilesX) * TH, tx0 = (trem % tilesX) * TW;
    stA.begin(sa, img0, ty0 - 1, tx0 - 1, cit * 32, tid, p.CinP, live);
    stB.begin(sb, img0, ty0, tx0, cot * 32, tid, 1 << 30, live);
  };

  // this wave's two xi rows (x = 0, 1 -> xi = 2 xh + x):   B^T row xi: d[a1] + sg d[a2]      A row xi: (c0, c1)
  //   xi 0: d0 - d2, (1, 0)     1: d1 + d2, (1, 1)     2: d2 - d1, (1, -1)     3: d1 - d3, (0, -1)
  int ao1[2], ao2[2];
  float sg[2], c0[2], c1[2];
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    const int xi = 2 * xh + x;
    const int a1 = xi == 0 ? 0 : (xi == 2 ? 2 : 1), a2 = xi == 3 ? 3 : (xi == 2 ? 1 : 2);
    ao1[x] = a1 * AHW * 32;
    ao2[x] = a2 * AHW * 32;
    sg[x] = xi == 1 ? 1.f : -1.f;
    c0[x] = xi == 3 ? 0.f : 1.f;
    c1[x] = xi == 0 ? 0.f : (xi == 1 ? 1.f : -1.f);
  }
  float V[2][4], M[2][4], dq[2][2], pr[8];
  const float *pA, *pB;
  auto step_addr = [&](const float* tA, const float* tB, const int j) {
    const int tt = kq * TPWV + 2 * j + half;
    const int im = tt / TPI, rem = tt % TPI;
    const int tyl = rem / TXT, txl = rem % TXT;
    pA = tA + ((im * AHH + 2 * tyl) * AHW + 2 * txl) * 32 + l31;
    pB = tB + ((im * TH + 2 * tyl) * TW + 2 * txl) * 32 + l31;
  };
  auto read_dy = [&]() {
    dq[0][0] = pB[0]; dq[0][1] = pB[32];
    dq[1][0] = pB[TW * 32]; dq[1][1] = pB[TW * 32 + 32];
  };
  auto read_row = [&](const int x) {
#pragma unroll
    for (int b = 0; b < 4; ++b) { pr[b] = pA[ao1[x] + b * 32]; pr[4 + b] = pA[ao2[x] + b * 32]; }
  };
  auto xform_row = [&](const int x) {
    float r[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) r[b] = fmaf(sg[x], pr[4 + b], pr[b]);
    V[x][0] = r[0] - r[2];
    V[x][1] = r[1] + r[2];
    V[x][2] = r[2] - r[1];
    V[x][3] = r[1] - r[3];
    const float t0 = fmaf(c1[x], dq[1][0], c0[x] * dq[0][0]), t1 = fmaf(c1[x], dq[1][1], c0[x] * dq[0][1]);
    M[x][0] = t0;
    M[x][1] = t0 + t1;
    M[x][2] = t0 - t1;
    M[x][3] = -t1;
  };

  begin_tile(ks, true);
  vv_static_for<0, NPA>([&](auto K) { stA.template load_piece<K.value>(sa, -1, tid); });
  vv_static_for<0, NPB>([&](auto K) { stB.template load_piece<K.value>(sb, 0, tid); });
  vv_static_for<0, NPA>([&](auto K) { stA.template commit_piece<K.value>(lds, tid); });
  vv_static_for<0, NPB>([&](auto K) { stB.template commit_piece<K.value>(lds + ASZ, tid); });
  __syncthreads();
  int cur = 0;
  for (int pt = ks; pt < NT; pt += KS) {
    const float* tA = lds + cur * TSZ;
    const float* tB = tA + ASZ;
    float* nA = lds + (cur ^ 1) * TSZ;
    float* nB = nA + ASZ;
    const bool live = pt + KS < NT;
    begin_tile(live ? pt + KS : ks, live);
    // first step of the staged tile: row 0 directly, reads of row 1 (transformed under the first MFMA group)
    step_addr(tA, tB, 0);
    read_dy();
    read_row(0);
    xform_row(0);
    read_row(1);
    vv_static_for<0, NKS>([&](auto JJ) {
      constexpr int j = JJ.value;
      constexpr bool more = j + 1 < NKS;
      vv_static_for<0, 8>([&](auto SS) {
        constexpr int sl = SS.value, x = sl >> 2, nu = sl & 3, slot = j * 8 + sl;
        acc[x][nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[x][nu], M[x][nu], acc[x][nu], 0, 0, 0);
        // under group 0: row 1 of this step is transformed (reads issued one group earlier), then the reads of row 0 of the
        // next step; under group 1: row 0 of the next step is transformed, then the reads of its row 1
        if constexpr (x == 0) {
          if constexpr (nu == 0) xform_row(1);
          if constexpr (nu == 1 && more) { step_addr(tA, tB, j + 1); read_row(0); }
        } else if constexpr (more) {
          if constexpr (nu == 0) { read_dy(); }
          if constexpr (nu == 1) { xform_row(0); read_row(1); }
        }
        if constexpr (slot < NPA) stA.template load_piece<slot>(sa, -1, tid);
        else if constexpr (slot < NP) stB.template load_piece<slot - NPA>(sb, 0, tid);
        else if constexpr (slot >= C0 && slot < C0 + NPA) stA.template commit_piece<slot - C0>(nA, tid);
        else if constexpr (slot >= C0 + NPA && slot < C0 + NP) stB.template commit_piece<slot - C0 - NPA>(nB, tid);
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    __syncthreads();                        // next buffer complete, current buffer no longer read
    cur ^= 1;
  }

  // ---- epilogue.  dg = G^T dU G,  G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1]: each wave folds ITS xi half into 9 tap partials
  // (linear), xi halves meet in LDS (one slab per K quarter), then the four slabs are summed in fixed order into the
  // workgroup's global slab -- the layout vv_wgrad_reduce expects.
  __syncthreads();
  float* slab = lds + kq * 9 * 1024;
  // G^T[a][xi] for this wave's rows xi = 2xh, 2xh+1:   xh = 0: (1, .5) (0, .5) (0, .5)      xh = 1: (.5, 0) (-.5, 0) (.5, 1)
  const float ga[3][2] = {{xh ? 0.5f : 1.f, xh ? 0.f : 0.5f}, {xh ? -0.5f : 0.f, xh ? 0.f : 0.5f}, {xh ? 0.5f : 0.f, xh ? 1.f : 0.5f}};
  for (int ph = 1; ph >= 0; --ph) {         // xi half 1 writes, barrier, xi half 0 adds
    if (xh == ph) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * half;
        float* q = slab + row * 32 + l31;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          float c[4];
#pragma unroll
          for (int nu = 0; nu < 4; ++nu) c[nu] = ga[a][0] * acc[0][nu][i] + ga[a][1] * acc[1][nu][i];
          const float h = 0.5f * (c[1] + c[2]);
          const float t0 = c[0] + h, t1 = 0.5f * (c[1] - c[2]), t2 = c[3] + h;
          if (ph) { q[(a * 3 + 0) * 1024] = t0; q[(a * 3 + 1) * 1024] = t1; q[(a * 3 + 2) * 1024] = t2; }
          else { q[(a * 3 + 0) * 1024] += t0; q[(a * 3 + 1) * 1024] += t1; q[(a * 3 + 2) * 1024] += t2; }
        }
      }
    }
    __syncthreads();
  }
  float4* out = reinterpret_cast<float4*>(p.partial + (int64_t)g * p.partial_gstride +
                                          ((int64_t)((cit * NCO + cot) * KS + ks)) * (9 * 1024));
  const float4* l4 = reinterpret_cast<const float4*>(lds);
  for (int e = tid; e < 9 * 256; e += NTH) {
    const float4 s0 = l4[e], s1 = l4[9 * 256 + e], s2 = l4[2 * 9 * 256 + e], s3 = l4[3 * 9 * 256 + e];
    float4 r;
    r.x = (s0.x + s1.x) + (s2.x + s3.x); r.y = (s0.y + s1.y) + (s2.y + s3.y);
    r.z = (s0.z + s1.z) + (s2.z + s3.z); r.w = (s0.w + s1.w) + (s2.w + s3.w);
    out[e] = r;
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

struct WGeo { int TH, TW, NI; };
inline bool wgeo(int kind, int H, int W, WGeo* t) {
  if (H != W) return false;
  if (kind == VV_CONV3) {
    if (H == 32) { *t = {8, 32, 1}; return true; }
    if (H == 16) { *t = {16, 16, 1}; return true; }
    if (H == 8) { *t = {8, 8, 2}; return true; }
    if (H == 4) { *t = {4, 4, 8}; return true; }
  } else {
    if (H == 16) { *t = {8, 16, 1}; return true; }
    if (H == 8) { *t = {8, 8, 2}; return true; }
    if (H == 4) { *t = {4, 4, 8}; return true; }
  }
  return false;
}

template <int TH, int TW, int NI>
int launch_ww(const vv_wgrad_params* p, hipStream_t st) {
  const int NT = ((p->B + NI - 1) / NI) * (p->H / TH) * (p->W / TW);
  const int NCI = (p->CinP + 31) / 32, NCO = p->Cout / 32;
  const int total = p->G * NCI * NCO * p->ksplit;
  const int nper = (total + 7) / 8;
  VV_LAUNCH((wgrad_wino_kernel<TH, TW, NI>), dim3(nper * 8), dim3(VV_WG), 0, st, *p, NT, NCI, NCO, total, nper);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

template <int TH, int TW, int NI>
int launch_ww8(const vv_wgrad_params* p, hipStream_t st) {
  const int NT = ((p->B + NI - 1) / NI) * (p->H / TH) * (p->W / TW);
  const int NCI = (p->CinP + 31) / 32, NCO = p->Cout / 32;
  const int total = p->G * NCI * NCO * p->ksplit;
  const int nper = (total + 7) / 8;
  VV_LAUNCH((wgrad_wino8_kernel<TH, TW, NI>), dim3(nper * 8), dim3(512), 0, st, *p, NT, NCI, NCO, total, nper);
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
  if (p->kind == VV_CONV3 && (p->pad0 & 512)) {           // Winograd form, eight waves (two per SIMD)
    switch (p->H) {
      case 32: return launch_ww8<8, 32, 1>(p, st);
      case 16: return launch_ww8<16, 16, 1>(p, st);
      case 8: return launch_ww8<8, 8, 2>(p, st);
      case 4: return launch_ww8<4, 4, 8>(p, st);
    }
    return VV_ERR_UNSUPPORTED;
  }
  if (p->kind == VV_CONV3 && (p->pad0 & 256)) {           // Winograd F(2x2,3x3) form (same tiles, same slabs)
    // measured per level at B = 256: four waves x 16 GEMMs win on the 32^2 / 16^2 levels (few, long staged tiles), eight
    // waves x 8 GEMMs (two waves per SIMD) on the 8^2 / 4^2 levels (-10 %)
    switch (p->H) {
      case 32: return launch_ww<8, 32, 1>(p, st);
      case 16: return launch_ww<16, 16, 1>(p, st);
      case 8: return launch_ww8<8, 8, 2>(p, st);
      case 4: return launch_ww8<4, 4, 8>(p, st);
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
