/* vecvad_hip.h -- C ABI of libvecvad_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for the VEC_VAD
 * spatio-temporal-cube completion path (UNet bank forward / backward / Adam) and the FlowNet2 native ops.
 *
 * Conventions (the only C-ABI precedent in the reference is the cffi seam of the three
 * FlowNet2 ops, FlowNet2_src/models/components/ops/correlation/src/correlation_cuda.h:1-8,
 * correlation_cuda_kernel.h:5-39, resample2d/src/Resample2d_cuda.c:8-11, channelnorm/src/ChannelNorm_cuda.c:8-11):
 *   - every entry point returns 0 on success and a non-zero vv_status otherwise; nothing aborts or prints;
 *   - the caller owns every buffer (inputs, outputs, scratch); raw device pointers + explicit sizes, no torch types;
 *   - launches are asynchronous on the hipStream_t passed in (pass PyTorch's current stream); no host sync;
 *   - no global mutable state: re-entrant, safe with one process per GPU and with several streams;
 *   - all tensors are fp32; activations are NHWC ("pixel-major, channel-minor"), grouped tensors are [G][...]
 *     with an explicit element stride between the G independent UNets of a bank.
 *
 * Which reference code each entry point replaces is stated next to it (paths relative to the reference root).
 */
#ifndef VECVAD_HIP_H
#define VECVAD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vv_stream; /* hipStream_t */

/* Every entry point returns an int: 0 on success, otherwise (status & 0xff) is one of the codes below and, for VV_ERR_LAUNCH,
 * (status >> 8) is the hipError_t the runtime reported -- the whole error travels in the return value, the library keeps no
 * last-error variable or any other mutable state between calls.  vv_status_string() renders a returned value as text. */
enum vv_status {
  VV_OK = 0,
  VV_ERR_BAD_ARG = 1,      /* unsupported shape / null pointer */
  VV_ERR_LAUNCH = 2,       /* hipGetLastError() != hipSuccess after the launch; the hipError_t sits in bits 8.. of the return */
  VV_ERR_UNSUPPORTED = 3
};

/* ---- how a convolution reads its input ("transform on load"; replaces separate BN/ReLU/pool/cat kernels) ---- */
enum vv_in_mode {
  VV_IN_PLAIN = 0, /* x                                                       */
  VV_IN_ACT = 1,   /* relu(a[c]*x + b[c])      nn.BatchNorm2d + nn.ReLU  model/unet.py:11-12,14-15 */
  VV_IN_POOL = 2,  /* 2x2 max of relu(a*x+b)   + nn.MaxPool2d(2)         model/unet.py:38          */
  VV_IN_CAT = 3,   /* [relu(a*x0+b) , x1]      torch.cat([x2, x1], 1)    model/unet.py:59          */
  VV_IN_CUBE = 4   /* channel gather through chmap (frame erasure)       model/unet.py:178-183     */
};

enum vv_conv_kind {
  VV_CONV3 = 0,     /* 3x3, stride 1, pad 1 (forward, or data-gradient with flipped packed weights)     */
  VV_CONVT_FWD = 1, /* ConvTranspose2d(k3,s2,p1,op1) forward as 4 output-parity phases  model/unet.py:54 */
  VV_CONVT_DGRAD = 2 /* its data gradient = 3x3 stride-2 gather over the 2Hx2W gradient                 */
};

/* Source/destination tensor view: NHWC, channel slice [coff, coff+C) of pixels with `cstride` floats each. */
typedef struct vv_view {
  float* ptr;        /* group 0 base */
  int64_t gstride;   /* floats between groups (0: shared by all groups) */
  int32_t cstride;   /* floats per pixel */
  int32_t coff;      /* first channel of the slice */
} vv_view;

/* MFMA implicit-GEMM convolution (v_mfma_f32_32x32x2_f32, exact fp32).
 * Replaces nn.Conv2d(k3,p1) / nn.ConvTranspose2d(k3,s2,p1,op1) forward and their autograd data-gradients
 * (cuDNN in the reference: model/unet.py:10,13,54) with BatchNorm/ReLU/MaxPool/cat of the producer fused into
 * the load, bias + per-channel sum / sum-of-squares partials (BatchNorm batch statistics) fused into the store. */
/* vv_conv_params.pad0 flag: round both operands to bf16 (nearest even) on their way into LDS and contract on
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- torch.autocast(bfloat16) semantics for the convolution; tensors in
 * HBM, bias, BatchNorm statistics and master weights stay fp32 (BASELINE config 4, "mixed bf16").  `w` must then be a bf16
 * panel (vv_pack_weights with mode | VV_PACK_BF16; CinP % 16 == 0). */
#define VV_CONV_BF16 1
/* with VV_CONV_BF16 and in_mode = VV_IN_PLAIN: src0 holds bf16 elements (same [B,H,W,C] indexing, gstride still in floats) --
 * the data gradient reading a dy that vv_bn_bwd_apply stored as bf16 (VV_BNBWD_DZ_BF16): same values as rounding on load, half
 * the bytes */
#define VV_CONV_SRC_BF16 2
/* with VV_CONV_BF16, VV_CONV3 / VV_CONVT_DGRAD: `out` is stored as bf16 (nearest even; same element indexing, gstride in
 * floats) and `stats` sums the stored values -- activation gradients in bf16, torch.autocast's dtype for them; read back by
 * vv_bn_bwd_* (VV_BNBWD_DA_BF16), by the transposed conv's data / weight gradient (VV_CONV_SRC_BF16 / VV_WGRAD_DY_BF16) */
#define VV_CONV_OUT_BF16 8
/* with VV_CONV_BF16 (any kind; modes PLAIN / ACT / CAT): EVERY source tensor of the launch holds bf16 elements -- pre-BN conv
 * outputs written with VV_CONV_OUT_BF16 (then also allowed on forward and VV_CONVT_FWD launches: torch.autocast's output dtype),
 * pooled / frame-erased inputs written by vv_pool_act / vv_cube_erase with their bf16 switch.  coff / cstride / csplit stay in
 * elements, gstride in floats. */
#define VV_CONV_ALLSRC_BF16 16
/* VV_CONV3 launches of vv_conv_mfma / vv_conv_wino: out = max(conv + bias, 0).  The eval-mode path (test.py:255-257,312-345): with
 * running-statistics BatchNorm the affine map is a constant of the model, so vv_fold_bn folds it into the filter and the bias once per
 * loaded model, the producing convolution applies the ReLU, and every consumer reads its input as VV_IN_PLAIN. */
#define VV_CONV_RELU 32
/* A/B switch: an all-bf16 3x3 launch (VV_CONV_BF16 | VV_CONV_OUT_BF16 | VV_CONV_ALLSRC_BF16 or VV_CONV_SRC_BF16) normally runs the
 * GEMM-shaped kernel of round 4 (vv_conv_bf16.hip, 256-pixel tiles on every level); with this flag it stays on the round-3
 * kernel (128-pixel tiles on the 16x16 / 8x8 / 4x4 levels).  vv_conv_ntiles2 follows the same flags. */
#define VV_CONV_NO_GEMM16 64
/* A/B switch: vv_conv_wino runs the 32x32-level launches with at most 32 input channels on the persistent LDS-DMA ring kernel of
 * round 5 (bit-identical results); with this flag they stay on the per-tile kernel. */
#define VV_CONV_NO_RING 128
typedef struct vv_conv_params {
  int32_t kind;      /* vv_conv_kind */
  int32_t in_mode;   /* vv_in_mode */
  int32_t G, B;      /* groups (UNets), cubes */
  int32_t H, W;      /* CONV3: image size; CONVT_FWD: INPUT size (output is 2H x 2W); CONVT_DGRAD: OUTPUT size
                        (= the transposed conv's input size; the gradient read is 2H x 2W) */
  int32_t Cin;       /* GEMM-K channels actually present */
  int32_t CinP;      /* Cin rounded up to the packing granule (multiple of 16; 8 for CONVT_DGRAD) */
  int32_t Cout;      /* GEMM-N channels (multiple of 32) */
  vv_view src0;      /* primary input */
  const float* a;    /* per-channel scale of src0's BatchNorm (VV_IN_ACT/POOL/CAT), [G][Cact] */
  const float* b;    /* per-channel shift */
  int64_t ab_gstride;
  vv_view src1;      /* VV_IN_CAT: second input (plain) */
  int32_t csplit;    /* VV_IN_CAT: channels [0,csplit) come from src0 */
  int32_t pad0;      /* flags: VV_CONV_BF16 (vv_conv_mfma only) */
  const int32_t* chmap; /* VV_IN_CUBE: [G][CinP] source channel or -1 (zero) */
  const float* w;    /* packed weights [9][CinP/8][2][Cout][4]  (see vv_pack_weights) */
  int64_t w_gstride;
  const float* bias; /* [G][Cout] or NULL */
  int64_t bias_gstride;
  vv_view out;       /* output */
  float* stats;      /* NULL or [G][ntiles][2][Cout] partial sums (sum, sum of squares) over valid pixels */
  /* vv_conv_wino only, data-gradient launches: the first reduction pass of the BatchNorm backward that consumes this output
   * (vv_bn_bwd_reduce) done in the epilogue.  out = dA of the producing layer's activation relu(a z + b); with z that layer's
   * conv output [G][B*H*W][Cout] (pixel stride Cout) the kernel also leaves  sum dz, sum dz * xhat  per pixel tile, dz = dA [a z + b > 0],
   * xhat = (z - mean) invstd, in bn_partial [G][ntiles][2][Cout] -- the layout vv_bn_bwd_apply reads with VV_BNBWD_PARTIALS_PER_TILE.
   * a / b / mean / invstd: [G][Cout] each, group stride bn_gstride.  bn_partial == NULL: off.
   * vv_conv_mfma takes the same fields for all-bf16 3x3 launches with Cout % 64 != 0 on the 32x32 level (z then holds bf16
   * elements; rows per vv_conv_ntiles(B,H,W), read with VV_BNBWD_PARTIALS_PER_CTILE); other launches with bn_partial set are refused. */
  const float* bn_z; int64_t bn_z_gstride;
  const float* bn_a; const float* bn_b; const float* bn_mean; const float* bn_invstd; int64_t bn_gstride;
  float* bn_partial;
  /* vv_conv_mfma, bf16-output 3x3 launches (VV_CONV_BF16 | VV_CONV_OUT_BF16) only: optional second output view.  Output channels
   * [osplit, Cout) go to out1 (as its channels [0, Cout - osplit)), channels [0, osplit) to out: the data gradient of a concat
   * layer leaves as two dense tensors, one per consumer (skip half -> BatchNorm backward, upsampled half -> transposed-conv
   * backward) -- with 32 bf16 channels per half an interleaved pixel row gives each consumer 64 useful bytes of every 128 fetched.
   * osplit a multiple of 32; out1.cstride == out.cstride; out1.gstride == out.gstride.  out1.ptr == NULL: off. */
  vv_view out1;
  int32_t osplit;
  int32_t pad1;
} vv_conv_params;

int vv_conv_mfma(const vv_conv_params* p, vv_stream stream);
/* number of pixel tiles the kernel uses for this (B,H,W): rows of the `stats` partial array */
int vv_conv_ntiles(int32_t B, int32_t H, int32_t W);
/* the same for a launch with these `kind` / pad0 `flags` (the bf16 3x3 kernels use 128-pixel tiles on the 8x8 / 4x4 levels) */
/* pixel tiles of a VV_CONVT_DGRAD launch of vv_conv_mfma (H x W = its output = the transposed conv's INPUT resolution; flags: the
 * launch's pad0, only VV_CONV_BF16 matters) = the rows of bn_partial such a launch leaves; -1 for an unsupported size */
int vv_convt_dgrad_ntiles(int32_t B, int32_t H, int32_t W, int32_t flags);
int vv_conv_ntiles2(int32_t B, int32_t H, int32_t W, int32_t kind, int32_t flags);

/* Weight-gradient (autograd of nn.Conv2d / nn.ConvTranspose2d wrt weight; cuDNN in the reference).
 * dW[tap][ci][co] = sum_pixels act[pixel+tap][ci] * dy[pixel][co]  (CONV3)
 * dW[tap][ci][co] = sum_pixels act[pixel][ci] * dy[2*pixel-1+tap][co] (CONVT)
 * Deterministic split-K: each workgroup-wave writes its own partial slab, vv_wgrad_reduce sums them in fixed order. */
typedef struct vv_wgrad_params {
  int32_t kind;      /* VV_CONV3 or VV_CONVT_FWD (meaning: weight gradient of the transposed conv) */
  int32_t in_mode;   /* how the layer input (act) is read, as in the forward */
  int32_t G, B, H, W; /* H,W: resolution of `act` pixels the GEMM-K runs over (CONVT: the transposed conv's input) */
  int32_t Cin, CinP, Cout;
  int32_t ksplit;    /* pixel-tile split factor */
  vv_view src0; const float* a; const float* b; int64_t ab_gstride;
  vv_view src1; int32_t csplit;
  int32_t pad0;      /* flags: bit 8 (256) = Winograd F(2x2,3x3) form for VV_CONV3 (dU = sum_tiles V^T dM, dg = G^T dU G; same
                        tiles, k-split and slabs, 2.25x fewer MFMA cycles, a few ulp from the direct form; W == H in {32,16,8,4};
                        three workgroups per CU: pick ksplit for 768 workgroup slots); low bits: bring-up */
  const int32_t* chmap;
  vv_view dy;        /* gradient wrt the conv output (CONV3: HxW; CONVT: 2Hx2W) */
  float* partial;    /* [G][nslab][9][32][32] with nslab = (CinP/32... see vv_wgrad_nslab) */
  int64_t partial_gstride;
} vv_wgrad_params;

int vv_wgrad_mfma(const vv_wgrad_params* p, vv_stream stream);
int vv_wgrad_ntiles(int32_t kind, int32_t B, int32_t H, int32_t W);
/* slabs written per (ci-tile, co-tile): ksplit (the 4 waves of a workgroup are summed in LDS first) */

/* Weight gradient of the 3x3 convolution with bf16 operands / fp32 accumulation (mixed precision, BASELINE config 4;
 * torch.autocast(bfloat16) semantics of the autograd weight gradient of nn.Conv2d, model/unet.py:10,13): act (after the
 * deferred BatchNorm+ReLU) and dy are rounded to bf16 on their way into LDS, products accumulate in fp32 on
 * v_mfma_f32_32x32x16_bf16.  Same parameter block as vv_wgrad_mfma (pad0: VV_WGRAD_DY_BF16 only); kind = VV_CONV3 with H = W in {32, 16, 8, 4}, or
 * VV_CONVT_FWD (weight gradient of nn.ConvTranspose2d(k3,s2,p1,op1), model/unet.py:54) with H = W in {16, 8, 4}.
 * One workgroup covers up to 64 ci x 64 co and every ksplit-th pixel tile:  slabs per (ci-tile, co-tile) = ksplit * kw.
 * vv_wgrad_bf16_plan returns 0 when the geometry is not handled (use vv_wgrad_mfma), else 1 and
 *   *ntiles  = pixel tiles (the upper bound of ksplit),
 *   *nblocks = workgroups per group and k-split part,
 *   *kw      = slabs per workgroup and (ci-tile, co-tile) (1 today): pass ksplit * kw to vv_wgrad_reduce. */
#define VV_WGRAD_X_BF16 2    /* vv_wgrad_bf16, pad0: the layer input (src0 / src1) holds bf16 elements; needs VV_WGRAD_DY_BF16 too */
#define VV_WGRAD_DY_BF16 1   /* vv_wgrad_bf16, pad0: dy holds bf16 elements (written by vv_bn_bwd_apply with VV_BNBWD_DZ_BF16) */
int vv_wgrad_bf16(const vv_wgrad_params* p, vv_stream stream);
int vv_wgrad_bf16_plan(int32_t kind, int32_t B, int32_t H, int32_t W, int32_t CinP, int32_t Cout, int32_t* ntiles, int32_t* nblocks,
                       int32_t* kw);

/* Sum the slabs and scatter into PyTorch parameter layout:
 * CONV3 : grad[co][ci][ky][kx]   (nn.Conv2d.weight  [Cout,Cin,3,3])
 * CONVT : grad[ci][co][ky][kx]   (nn.ConvTranspose2d.weight [Cin,Cout,3,3]) */
int vv_wgrad_reduce(int32_t kind, int32_t G, int32_t Cin, int32_t CinP, int32_t Cout, int32_t nslab_per_tile,
                    const float* partial, int64_t partial_gstride, float* grad, int64_t grad_gstride,
                    vv_stream stream);
/* The same for several layers in ONE launch (e.g. every weight gradient of a data-parallel gradient bucket).  Device table, sorted by
 * block_start; entry e owns blocks [block_start, block_start + ceil(CinP/32)*(Cout/32)*36) of the launch; total_blocks = their sum.
 * Slabs of entry e start at partial + g*partial_gstride + part_off; its gradient at grads + grad_off + g*grad_gstride (floats). */
typedef struct vv_reduce_entry {
  int32_t kind, Cin, Cout, NCO, nslab, block_start;
  int64_t part_off, grad_off, grad_gstride;
} vv_reduce_entry;
int vv_wgrad_reduce_grouped(const vv_reduce_entry* table_dev, int32_t nentries, int32_t total_blocks, int32_t G, const float* partial,
                            int64_t partial_gstride, float* grads, vv_stream stream);

/* ---- weight packing: PyTorch OIHW -> MFMA B-operand panels ----
 * mode 0: conv forward      Wp[t=ky*3+kx][k=ci][n=co] = W[co][ci][ky][kx]
 * mode 1: conv data-grad    Wp[t=(2-ky)*3+(2-kx)][k=co][n=ci] = W[co][ci][ky][kx]
 * mode 2: convT forward     Wp[t=ky*3+kx][k=ci][n=co] = Wt[ci][co][ky][kx]
 * mode 3: convT data-grad   Wp[t=ky*3+kx][k=co][n=ci] = Wt[ci][co][ky][kx]
 * packed element (t,k,n) lives at (((t*KP/8 + k/8)*2 + (k%8)/4)*N + n)*4 + k%4 ; rows k >= K are zero.
 * mode | 4 (VV_PACK_BF16): the same panel as bf16 for VV_CONV_BF16 launches (KP % 16 == 0):
 * element (t,k,n) is the bf16 at index (((t*KP/16 + k/16)*2 + (k%16)/8)*N + n)*8 + k%8 from the same dst_off. */
#define VV_PACK_BF16 4
typedef struct vv_pack_entry {
  int64_t src_off;   /* floats from the group's parameter base */
  int64_t dst_off;   /* floats from the group's packed base */
  int32_t mode, K, KP, N;
} vv_pack_entry;
int vv_pack_weights(const vv_pack_entry* table_dev, int32_t nentries, int32_t G, const float* params,
                    int64_t params_gstride, float* packed, int64_t packed_gstride, int32_t max_elems,
                    vv_stream stream);

/* ---- Winograd F(2x2,3x3) form of the same 3x3 / stride 1 / pad 1 convolution (forward and data-gradient) ----
 * vv_conv_wino takes the vv_conv_params of a VV_CONV3 launch with `w` = panels from vv_pack_wino
 * ([16 = xi*4+nu][CinP/8][2][Cout][4], U = G g G^T; mode 0 forward, mode 1 data-gradient i.e. flipped + transposed filter);
 * 2.25x fewer MFMA cycles than vv_conv_mfma, fp32 throughout (results differ from the direct form by a few ulp).
 * in_mode PLAIN / ACT / CAT; CinP % 8 == 0; `stats` has vv_wino_ntiles(B, H) rows per UNet. */
int vv_conv_wino(const vv_conv_params* p, vv_stream stream);
int vv_wino_ntiles(int32_t B, int32_t H);
int vv_pack_wino(const vv_pack_entry* table_dev, int32_t nentries, int32_t G, const float* params, int64_t params_gstride,
                 float* packed, int64_t packed_gstride, int32_t max_kn, vv_stream stream);

/* ---- Winograd F(4x4,3x3) form of the same convolution (round 5; forward and data-gradient) ----
 * 2.25 matrix-core multiply-adds per output pixel and channel pair (F(2x2,3x3): 4, direct: 9), fp32 throughout; the larger transform
 * constants leave the result a few 1e-6 of the tensor's maximum from the direct form (F(2x2): a few 1e-7).
 * vv_conv_wino44 takes the vv_conv_params of a VV_CONV3 launch (same fields and contract as vv_conv_wino, incl. `stats`, the fused
 * BatchNorm-backward sums `bn_partial` and VV_CONV_RELU) with `w` = panels from vv_pack_wino44:
 * [36 = xi*6+nu][CinP/8][2][2][Cout][2], element (t, k, n) at ((((t*CinP/8 + k/8)*2 + (k%4)/2)*2 + (k%8)/4)*Cout + n)*2 + k%2,
 * U = G g G^T (6x6); mode 0 forward, mode 1 data-gradient (flipped + transposed filter).  36*KP*N floats per entry.
 * in_mode PLAIN / ACT / CAT; CinP % 8 == 0; H = W in {32, 16, 8, 4}; `stats` / `bn_partial` have vv_wino44_ntiles(B, H) rows per UNet
 * (a row = 32 tiles of 4x4 pixels). */
int vv_conv_wino44(const vv_conv_params* p, vv_stream stream);
int vv_wino44_ntiles(int32_t B, int32_t H);
int vv_pack_wino44(const vv_pack_entry* table_dev, int32_t nentries, int32_t G, const float* params, int64_t params_gstride,
                   float* packed, int64_t packed_gstride, int32_t max_kn, vv_stream stream);

/* ---- BatchNorm (nn.BatchNorm2d(eps=1e-5, momentum=0.1), model/unet.py:11,14) ----
 * train != 0: batch statistics from the conv's partial sums; writes scale/shift a,b for the consumer's load,
 *             mean / invstd for the backward pass, and updates running_mean / running_var (unbiased) in place.
 * train == 0: a,b from the running statistics. */
int vv_bn_finalize(int32_t G, int32_t C, int32_t ntiles, int64_t count, int32_t train, float momentum, float eps,
                   const float* stats, int64_t stats_gstride, const float* gamma, const float* beta,
                   int64_t param_gstride, float* running_mean, float* running_var, int64_t buf_gstride,
                   float* a, float* b, float* mean, float* invstd, int64_t ab_gstride, vv_stream stream);

/* BatchNorm + ReLU (+ MaxPool, + skip fan-in) backward, phase 1:
 * dz = (dA0 [+ route(dPool)]) * [a*y+b > 0]; writes per-block partial sums of dz and dz*xhat (not dz). */
/* vv_bnbwd_params.flags: vv_bn_bwd_apply writes dy as bf16 (nearest even; [B,H,W,C] with 2-byte elements at the same base,
 * dz_gstride still in floats) for consumers that round it to bf16 anyway (vv_conv_mfma + VV_CONV_SRC_BF16, vv_wgrad_bf16 +
 * VV_WGRAD_DY_BF16) */
#define VV_BNBWD_DZ_BF16 1
#define VV_BNBWD_PARTIALS_PER_CUBE 2   /* vv_bn_bwd_apply: `partial` holds [G][B][2][C] written by vv_outconv_bwd */
#define VV_BNBWD_Y_BF16 8              /* y holds bf16 elements (VV_CONV_OUT_BF16 on the forward launch) */
#define VV_BNBWD_DA_BF16 4             /* dA (and dpool) hold bf16 elements (VV_CONV_OUT_BF16 / vv_outconv_bwd dA_bf16) */
#define VV_BNBWD_PARTIALS_PER_TILE 16  /* vv_bn_bwd_apply: `partial` holds [G][vv_wino_ntiles(B,H)][2][C] written by the data-gradient
                                          launch that produced dA (vv_conv_params.bn_partial): no vv_bn_bwd_reduce pass for that layer */
#define VV_BNBWD_PARTIALS_PER_CTILE 32 /* the same for a data-gradient launch of vv_conv_mfma (all-bf16 tensors): [G][vv_conv_ntiles(B,H,W)][2][C] */
#define VV_BNBWD_PARTIALS_PER_TTILE 128  /* ... of the transposed conv's data gradient (vv_conv_mfma, VV_CONVT_DGRAD, fp32 kernel):
                                           [G][vv_convt_dgrad_ntiles(B,H,W,0)][2][C] */
#define VV_BNBWD_PARTIALS_PER_TILE44 64 /* the same for a data-gradient launch of vv_conv_wino44: [G][vv_wino44_ntiles(B,H)][2][C] */
typedef struct vv_bnbwd_params {
  int32_t G, B, H, W, C;
  int32_t flags;
  const float* y; int64_t y_gstride;             /* pre-BN conv output [B,H,W,C] */
  const float* a; const float* b; const float* mean; const float* invstd; int64_t ab_gstride;
  vv_view dA;                                    /* gradient wrt the post-ReLU activation (same resolution) */
  const float* dpool; int64_t dpool_gstride;     /* NULL or gradient wrt maxpool2(act) [B,H/2,W/2,C] */
  float* dz; int64_t dz_gstride;                 /* out [B,H,W,C] */
  float* partial;                                /* [G][nblk][2][C] */
} vv_bnbwd_params;
int vv_bn_bwd_reduce(const vv_bnbwd_params* p, vv_stream stream);
int vv_bn_bwd_nblk(int32_t B, int32_t H, int32_t W, int32_t C);
/* phase 2: sums the partials (fixed order, fp64), writes dgamma/dbeta, then re-reads dA and y and writes
 * dy = gamma*invstd*(dz - mean(dz) - xhat*mean(dz*xhat)) into p->dz (dz itself is never materialised). */
int vv_bn_bwd_apply(const vv_bnbwd_params* p, const float* gamma, int64_t param_gstride, float* dgamma, float* dbeta,
                    int64_t grad_gstride, float* scratch /* [G][2][C] */, vv_stream stream);

/* ---- output 1x1 conv + squared error (model/unet.py:63-70 ; train.py:385-392,421-426 ; test.py:330-335) ----
 * out[p][co] = bias[co] + sum_c relu(a*y+b)[p][c] * W[co][c];  score[g][cube] = sum (out - target)^2;
 * optional dout[p][co] = gscale[g] * (out - target)  (gradient of lambda * MSELoss(mean)). */
typedef struct vv_outconv_params {
  int32_t G, B, HW, C;           /* HW pixels per cube (1024), C = features_root */
  const float* y; int64_t y_gstride; const float* a; const float* b; int64_t ab_gstride;
  const float* w; const float* bias; int64_t param_gstride;   /* W [oc][C], bias [oc] inside the parameter block */
  const int32_t* oc;             /* [G] output channels (3 raw / 2 flow) */
  const float* tgt0; int32_t tgt0_cstride; int32_t pad0;   /* cube NHWC [B,HW,15]; pad0 bit 0: y holds bf16 elements */
  const float* tgt1; int32_t tgt1_cstride; int32_t pad1;   /* flow NHWC [B,HW,2*T_of] */
  const int32_t* tgt_src;        /* [G] 0: tgt0, 1: tgt1 */
  const int32_t* tgt_coff;       /* [G] first target channel */
  float* out4;                   /* NULL (fused train / scoring steps: nobody reads it) or [G][B*HW][4] reconstruction (channels >= oc are 0) */
  float* score;                  /* [G][B] */
  const float* gscale;           /* NULL or [G] */
  float* dout4;                  /* NULL or [G][B*HW][4] */
} vv_outconv_params;
int vv_outconv_fwd(const vv_outconv_params* p, vv_stream stream);

/* backward of the 1x1 conv: dA[p][c] = sum_co dout[p][co] W[co][c]; dW[co][c] = sum_p dout[p][co] act[p][c];
 * db[co] = sum_p dout[p][co].  partial: [G][nblk][4*C + 4]; reduced by vv_outconv_bwd_reduce.  C = features_root: 32 or 64. */
/* bnpart != NULL: also writes the BatchNorm-backward partial sums of the layer in front of the output conv, [G][B][2][C]
 * (per cube: sum of g = dA * [act > 0], sum of g * xhat; mean / invstd: [G][ab_gstride] like a, b) -- vv_bn_bwd_apply with
 * VV_BNBWD_PARTIALS_PER_CUBE then needs no vv_bn_bwd_reduce pass for that layer. */
int vv_outconv_bwd(int32_t G, int32_t B, int32_t HW, int32_t C, const float* dout4, const float* y,
                   int64_t y_gstride, const float* a, const float* b, int64_t ab_gstride, const float* w,
                   int64_t param_gstride, float* dA, int64_t dA_gstride, float* partial, const float* mean,
                   const float* invstd, float* bnpart, int32_t flags /* bit 0: store dA as bf16 (VV_BNBWD_DA_BF16 consumer); bit 1: y holds bf16 elements */,
                   vv_stream stream);
int vv_outconv_bwd_nblk(int32_t B, int32_t HW);
/* vv_outconv_fwd followed by vv_outconv_bwd with dout = gscale * (out - target), in ONE pass over y (the fused train step: the
 * reference's loss.backward() right behind the forward, train.py:385-402): same score / dA / partial / bnpart bits as the two
 * calls, y read once, d(out) never stored (p->dout4 may be NULL; p->gscale must be given).  flags: bit 0 = dA stored as bf16,
 * bit 1 = y holds bf16 elements (must equal bit 0 of p->pad0). */
int vv_outconv_fwdbwd(const vv_outconv_params* p, float* dA, int64_t dA_gstride, float* partial, const float* mean,
                      const float* invstd, float* bnpart, int32_t flags, vv_stream stream);
int vv_outconv_bwd_reduce(int32_t G, int32_t C, int32_t nblk, const float* partial, const int32_t* oc,
                          float* dW, float* db, int64_t grad_gstride, vv_stream stream);

/* bias gradient of the transposed conv: db[co] = sum_pixels dy[p][co] (two-stage, deterministic) */
int vv_bias_grad(int32_t G, int64_t M, int32_t C, const float* dy, int64_t dy_gstride, int32_t cstride,
                 int32_t coff, float* scratch, float* db, int64_t grad_gstride, vv_stream stream);
/* The same bias gradient without a pass over the tensor: when dy is the output of a vv_conv_mfma / vv_conv_wino launch that was
 * given a `stats` array ([G][ntiles][2][C]: per-tile column sums in slot 0), db[g][j] = sum_tiles stats[g][tile][0][coff + j]
 * (fixed order, fp64).  Used for the transposed conv's bias: its output gradient is a channel slice of the concat layer's data
 * gradient (autograd of nn.ConvTranspose2d bias, model/unet.py:54). */
int vv_bias_from_partials(int32_t G, int32_t C, int32_t ntiles, int32_t coff, int32_t n, const float* partial,
                          int64_t partial_gstride, float* db, int64_t grad_gstride, vv_stream stream);

/* ---- fused Adam over one flat buffer (torch.optim.Adam(eps=1e-7), train.py:376,400-402) ---- */
int vv_adam(int64_t n, float* param, const float* grad, float* m, float* v, float lr, float beta1, float beta2,
            float eps, float bias_corr1, float bias_corr2_sqrt, float grad_scale, vv_stream stream);

/* Graph-replayable form of the same optimiser step (a captured train step must not carry host-computed bias corrections):
 *  vv_adam_tick     : t_dev[0] += 1; sc_dev[0] = lr / (1 - beta1^t), sc_dev[1] = sqrt(1 - beta2^t)   (one thread; beta as doubles)
 *  vv_adam_bucketed : Adam over param / m / v [G][U] reading sc_dev, with the gradient buffer in BUCKET-MAJOR layout -- bucket k =
 *                     columns [bounds[k], bounds[k+1]) of every UNet, contiguous as [G][width_k] at float offset G*bounds[k], so
 *                     that each data-parallel all-reduce (train.py:375) runs in place on one contiguous range.  bounds: HOST array
 *                     of nb+1 multiples of 4, bounds[0] = 0, bounds[nb] = U, nb <= 8. */
int vv_adam_tick(int64_t* t_dev, float lr, double beta1, double beta2, float* sc_dev, vv_stream stream);
/* counters[0..n) += inc on the device: BatchNorm2d's num_batches_tracked of a train-mode forward (model/unet.py:11,14 through
 * torch.nn.BatchNorm2d) -- a first-party launch, so that a captured train step holds no framework kernel */
int vv_counter_add(int64_t* counters, int32_t n, int64_t inc, vv_stream stream);
int vv_adam_bucketed(int32_t G, int64_t U, int32_t nb, const int64_t* bounds, float* param, const float* grad, float* m,
                     float* v, const float* sc_dev, float beta1, float beta2, float eps, float grad_scale, vv_stream stream);

/* ---- cube adapter (vad_datasets.py:130-168: [T,H,W,C] -> [H,W,T*C], uint8 -> float/255) ----
 * raw  uint8 [N][T][HW][3]  -> x   fp32 NHWC [B][HW][3T]   for the cubes idx[0..B)
 * flow fp32  [N][Tf][HW][2] -> xof fp32 NHWC [B][HW][2Tf]                                      */
int vv_cube_gather(int32_t B, int32_t T, int32_t Tf, int32_t HW, const int64_t* idx, const uint8_t* raw,
                   const float* flow, float* x, float* xof, vv_stream stream);

/* Small materialised inputs that keep the MFMA kernels on their fast (plain, pipelined) load path:
 *  vv_pool_act  : out[g][b,y,x,c] = max_{2x2} relu(a*y+b)    nn.MaxPool2d(2) of the activated tensor (model/unet.py:38);
 *                 1/4 of the source tensor, written once in the forward and reused by the weight-gradient.
 *  vv_cube_erase: out[g][pixel][k] = cube[pixel][chmap[g][k]] (or 0): the frame-erased input of UNet g, Cin padded to CP
 *                 (model/unet.py:178-183). */
int vv_pool_act(int32_t G, int32_t B, int32_t H2, int32_t W2, int32_t C, const float* y, int64_t y_gstride, const float* a,
                const float* b, int64_t ab_gstride, float* out, int64_t out_gstride, int32_t io_bf16 /* y and out hold bf16 */,
                vv_stream stream);
int vv_cube_erase(int32_t G, int64_t npix, int32_t Cc, int32_t CP, const float* cube, const int32_t* chmap, float* out,
                  int64_t out_gstride, int32_t out_bf16, vv_stream stream);

/* NCHW <-> NHWC for the module surface (forward(x, x_of) takes NCHW like the reference) */
int vv_nchw_to_nhwc(int32_t B, int32_t C, int32_t HW, const float* src, float* dst, vv_stream stream);
/* gathers channels [0,oc) of out4[g] into NCHW dst[b][choff + ...] ; dst has Ctot channels */
int vv_out4_to_nchw(int32_t B, int32_t HW, int32_t oc, const float* out4, float* dst, int32_t Ctot,
                    int32_t choff, vv_stream stream);
int vv_nchw_to_out4(int32_t B, int32_t HW, int32_t oc, const float* src, int32_t Ctot, int32_t choff,
                    float* out4, vv_stream stream);

/* ---- FlowNet2 native ops (forward only; FlowNet2 is inference-only in VEC_VAD, calc_optical_flow.py:56-57) ----
 * Same argument meaning as the reference's cffi functions, NCHW fp32 contiguous, stream = current stream.
 * Correlation_forward_cuda  (ops/correlation/src/correlation_cuda.c:11-93, correlation_cuda_kernel.cu:10-106):
 *   no rInput scratch is needed (bounds are predicated instead of materialising zero-padded NHWC copies). */
int vv_correlation_fwd(const float* in1, const float* in2, float* out, int32_t B, int32_t C, int32_t H, int32_t W,
                       int32_t pad_size, int32_t kernel_size, int32_t max_displacement, int32_t stride1,
                       int32_t stride2, int32_t corr_type_multiply, vv_stream stream);
int vv_correlation_out_shape(int32_t C, int32_t H, int32_t W, int32_t pad_size, int32_t kernel_size,
                             int32_t max_displacement, int32_t stride1, int32_t stride2, int32_t* oC, int32_t* oH,
                             int32_t* oW);
/* The same op specialised to how FlowNetC uses it (FlowNetC.py:41-47,92-93,120: pad 20, kernel 1, max_displacement 20,
 * stride1 1, stride2 2, then LeakyReLU(0.1), then torch.cat with conv_redir): NHWC maps in (pixel stride `cstride` floats,
 * C % 32 == 0, 16-byte aligned), result * 1/C through y = v > 0 ? v : slope*v written to channels
 * [out_coff, out_coff+441) of an NHWC buffer with pixel stride out_cstride.  W in {64, 128} (512- and 1024-wide
 * inputs); other widths return VV_ERR_UNSUPPORTED -> use vv_correlation_fwd. */
int vv_correlation_nhwc(const float* f1, const float* f2, int32_t cstride, int32_t B, int32_t C, int32_t H, int32_t W,
                        float* out, int32_t out_cstride, int32_t out_coff, float slope, vv_stream stream);
/* Resample2d_cuda_forward (ops/resample2d/src/Resample2d_cuda.c:8-11, Resample2d_kernel.cu:20-66) */
int vv_resample2d_fwd(const float* img, const float* flow, float* out, int32_t B, int32_t C, int32_t H, int32_t W,
                      int32_t fH, int32_t fW, int32_t kernel_size, vv_stream stream);
/* ChannelNorm_cuda_forward (ops/channelnorm/src/ChannelNorm_cuda.c:8-11, ChannelNorm_kernel.cu:19-51) */
int vv_channelnorm_fwd(const float* in, float* out, int32_t B, int32_t C, int32_t H, int32_t W, int32_t norm_deg,
                       vv_stream stream);

/* ---- FlowNet2 conv stack (forward only; FlowNet2_src/models/components/misc.py:8-44, FlowNet{C,S,SD,Fusion}.py) ----
 * Generic NHWC MFMA convolution: kind 0 = nn.Conv2d(k=R in {1,3,5,7}, stride in {1,2}, padding=(R-1)/2),
 * kind 1 = nn.ConvTranspose2d(k4, s2, p1); optional bias; y = v > 0 ? v : slope*v  (slope 0.1 = LeakyReLU, 1.0 = none).
 * src: NHWC with cstride % 4 == 0 and coff % 4 == 0 (pad channels must hold finite values); out: any channel slice of
 * the consumer's concat buffer.  Weights packed by vv_pack_conv2d: [taps][CinP/8][2][CoutP][4], zero padded.
 * kind 2 = the same nn.Conv2d in "row-K" form for the few-channel first layers (FlowNetC.py conv1: 7x7 s2 on 3 channels;
 * FlowNetSD.py conv0: 3x3 s1 on 6): src.cstride = 4 (R = 7, stride 2) or 8 (R = 3, stride 1), src.coff = 0; K runs over the
 * flattened (kx, c) run of one filter row, Cin = CinP = ceil8(R * cstride) (32 / 24), and the panel is packed with taps = R from a
 * weight tensor rearranged to [N][kx * cstride + c][ky] (zeros at the pad channels and beyond kx = R - 1). */
typedef struct vv_conv2d_params {
  int32_t kind, R, stride;
  int32_t B, H, W;          /* input size */
  int32_t Cin, CinP, Cout, CoutP;
  vv_view src;
  const float* w;
  const float* bias;
  float slope;
  int32_t pad0;       /* ksplit: > 1 splits the input-channel loop over that many workgroups; `out` must then be a
                         workspace [ksplit][B*OH*OW][CoutP] (cstride = CoutP, coff = 0) finished by vv_conv2d_splitk_finish */
  vv_view out;
} vv_conv2d_params;
int vv_conv2d_mfma(const vv_conv2d_params* p, vv_stream stream);
/* The same conv(k = 3, stride 1, pad 1) [+ LeakyReLU(slope); slope 1 = none] in Winograd F(2x2, 3x3) form (fp32, a few ulp from the
 * direct form; 2.25x fewer MFMA cycles): FlowNet2's large stride-1 layers (components/misc.py:8-28 as used by FlowNetSD.py:9-103,
 * FlowNetFusion.py:9-64, FlowNetC.py / FlowNetS.py conv3_1 ... conv5_1).  `panel` = vv_pack_wino (mode 0, one entry, G = 1) of the
 * nn.Conv2d weight [Cout][Cin][3][3]: [16][CinP/8][2][Cout][4]; CinP % 8 == 0, Cout % 32 == 0, H % 2 == 0, W % 32 == 0.
 * src: NHWC, pixel stride src_cstride floats (multiple of 4), first channel src_coff; src_elems = floats in the source buffer
 * (reads of the K padding past it return zeros; pad channels inside it must be finite).  out: NHWC slice (out_cstride, out_coff). */
int vv_conv2d_wino(const float* src, int32_t src_cstride, int32_t src_coff, int64_t src_elems, const float* panel, const float* bias,
                   float slope, float* out, int32_t out_cstride, int32_t out_coff, int32_t B, int32_t H, int32_t W, int32_t CinP,
                   int32_t Cout, vv_stream stream);
int vv_conv2d_splitk_finish(const float* ws, int32_t ksplit, int64_t M, int32_t Cout, int32_t CoutP, const float* bias,
                            float slope, float* out, int32_t out_cstride, int32_t out_coff, vv_stream stream);
/* The two-channel flow heads, where a 32-wide MFMA tile would be 94 % padding:
 *  vv_conv3x3_n2  : predict_flow = nn.Conv2d(Cin, 2, 3, 1, 1) (components/misc.py:42-44).  wq = weights repacked as
 *                   [tap 9][C4P][out 2][4] (Cin zero-padded to 4*C4P, C4P*4 a multiple of 32), bias[2] or NULL.
 *  vv_deconv4x4_c2: upsampled_flow = nn.ConvTranspose2d(2, 2, 4, 2, 1) (FlowNetS.py:40-47 ...), w in PyTorch layout
 *                   [2][2][4][4].
 * NHWC in (pixel stride src_cstride, channels from 0), result through y = v > 0 ? v : slope*v into channels
 * [out_coff, out_coff+2) of an NHWC buffer with pixel stride out_cstride. */
int vv_conv3x3_n2(const float* src, int32_t src_cstride, int32_t B, int32_t H, int32_t W, int32_t Cin, const float* wq,
                  int32_t C4P, const float* bias, float slope, float* out, int32_t out_cstride, int32_t out_coff,
                  vv_stream stream);
int vv_deconv4x4_c2(const float* src, int32_t src_cstride, int32_t B, int32_t H, int32_t W, const float* w,
                    const float* bias, float slope, float* out, int32_t out_cstride, int32_t out_coff, vv_stream stream);
/* w: Conv2d [N][K][R][R] (transposed = 0) or ConvTranspose2d [K][N][4][4] (transposed = 1); taps = R*R */
int vv_pack_conv2d(const float* w, float* packed, int32_t taps, int32_t K, int32_t KP, int32_t N, int32_t NP,
                   int32_t transposed, vv_stream stream);
/* nn.Upsample(scale_factor=4) on NCHW planes, times `scale` (flownet2.py:28,34,43-44,76,90,105,122).
 * bilinear: 0 = 'nearest', 1 = 'bilinear' with align_corners=False (torch >= 0.4), 2 = 'bilinear' with align_corners=True
 * (what 'bilinear' meant under the PyTorch 0.3 the reference's README pins for the flow extraction) */
int vv_upsample4(const float* src, float* dst, int32_t BC, int32_t H, int32_t W, int32_t bilinear, float scale,
                 vv_stream stream);

/* ---- eval mode: nn.BatchNorm2d with running statistics folded into the convolution in front of it (model/unet.py:9-16 under
 * net.eval(), test.py:255-257).  For every table entry and group g, with a[c] = gamma[c] / sqrt(running_var[c] + eps) (float64):
 *   folded[w_off + c*row + k] = a[c] * params[w_off + c*row + k]         (row = Cin*9 filter elements per output channel)
 *   folded[b_off + c]         = a[c] * (params[b_off + c] - running_mean[c]) + beta[c]
 * Offsets are in floats inside one group's block; params / folded have params_gstride / folded_gstride floats per group, bufs
 * (running statistics) bufs_gstride.  Entries the table does not name are left untouched (copy them first). */
typedef struct vv_fold_entry {
  int64_t w_off, b_off, g_off, beta_off; /* into the parameter block */
  int64_t rm_off, rv_off;                /* into the buffer block */
  int32_t cout, row;
} vv_fold_entry;
int vv_fold_bn(const vv_fold_entry* table_dev, int32_t nentries, int32_t G, const float* params, int64_t params_gstride,
               const float* bufs, int64_t bufs_gstride, float eps, float* folded, int64_t folded_gstride, vv_stream stream);

/* ---- FlowNet2 plumbing between the sub-networks (FlowNet2_src/models/flownet2.py:65-136), NHWC, no temporaries ----
 * vv_flownet_prep: inputs [B,3,2,H,W] fp32 (0..rgb_max) -> rgb_mean over (frame, H, W) per image and colour (:66-67),
 *   x = (inputs - rgb_mean) / rgb_max (:69), x6 [B,H,W,8] = cat(x1, x2) + 2 zero channels (:70-72), img0 / img1 [B,H,W,4] =
 *   the two frames + 1 zero channel (FlowNetC's siamese stem).  workspace: vv_flownet_prep_workspace_bytes(B) bytes, 16-byte
 *   aligned like every tensor here.
 * vv_warp_pack12: flow = Upsample x4 (mode as vv_upsample4's `bilinear`) of flow2 [B,H/4,W/4,flow_cstride] (channels 0,1)
 *   times `scale` (:76,90); out12 [B,H,W,12] = cat(x6[0:6], Resample2d(img1, flow), flow / div_flow, ChannelNorm(x1 - warped))
 *   (:79-86, 93-100).
 * vv_fusion_pack11: s2 = nearest x4 of s2_flow2 * div_flow (:105), sd = nearest x4 of sd_flow2 / div_flow (:122);
 *   out12 [B,H,W,12] = cat(x1, sd, s2, |sd|, |s2|, |x1 - warp(img1, sd)|, |x1 - warp(img1, s2)|) + 1 zero channel (:132-136). */
int64_t vv_flownet_prep_workspace_bytes(int32_t B);
int vv_flownet_prep(const float* inputs, int32_t B, int32_t H, int32_t W, float rgb_max, void* workspace,
                    int64_t workspace_bytes, float* x6, float* img0, float* img1, vv_stream stream);
int vv_warp_pack12(const float* x6, const float* img1, const float* flow2, int32_t flow_cstride, int32_t B, int32_t H,
                   int32_t W, int32_t mode, float scale, float div_flow, float* out12, vv_stream stream);
int vv_fusion_pack11(const float* x6, const float* img1, const float* s2_flow2, int32_t s2_cstride, const float* sd_flow2,
                     int32_t sd_cstride, int32_t B, int32_t H, int32_t W, float div_flow, float* out12, vv_stream stream);

/* ---- cube extraction (vad_datasets.py:70-93 get_foreground, calc_optical_flow.py:46-59,82) ----
 * One launch crops n boxes out of T decoded frames and resizes each crop with cv2.resize's default INTER_LINEAR
 * arithmetic (uint8: 11-bit fixed point, bit-exact; float32: unfused fp32; exact 2x decimation -> 2x2 area mean;
 * equal size -> copy).
 *   frames [T][H][W][C] uint8 (is_f32 = 0) or float32 (is_f32 = 1)  -- the layout cv2.imread / np.load hand over
 *   crops  int32 [n][4] = x_min, y_min, x_max, y_max with 0 <= min < max <= W|H  (ceil + slice clipping done by the host)
 *   out    [n][T][oh][ow][C], same dtype  (= the [N,T,32,32,C] cube layout of the *_foreground_*.npy files)        */
int vv_crop_resize(const void* frames, int32_t is_f32, int32_t T, int32_t H, int32_t W, int32_t C, const int32_t* crops,
                   int32_t n, int32_t oh, int32_t ow, void* out, vv_stream stream);

/* ---- score aggregation (test.py:330-358,387-399, utils.py:29-41) ----
 * vv_frame_scores: frame_scores[f] = max(frame_scores[f], max over cubes m in [frame_off[f], frame_off[f+1]) with
 *   paints[m] != 0 of  cube_stat[m] < 0 ? big : w_raw*(raw[m]-mu_r)/sd_r + w_of*(of[m]-mu_o)/sd_o ), float64;
 *   stats = double [S][4] (mu_r, sd_r, mu_o, sd_o) indexed by cube_stat[m]; of == NULL drops the flow term (useFlow=False).
 *   The caller initialises frame_scores to -big (the mask background).
 * vv_roc_auc_counts: out3[0] = 2*#{pos>neg} + #{pos==neg}, out3[1] = #pos, out3[2] = #neg; AUC = out3[0]/(2*P*N).     */
int vv_frame_scores(const float* raw, const float* of, const int32_t* frame_off, const int32_t* cube_stat,
                    const double* stats, const uint8_t* paints, double w_raw, double w_of, double big, int32_t n_frames,
                    double* frame_scores, vv_stream stream);
int vv_roc_auc_counts(const double* scores, const uint8_t* labels, int32_t n, uint64_t* out3, vv_stream stream);

/* library self-description */
const char* vv_version(void);
/* text for a value returned by any entry point (a pure function of its argument: pointer to a static string; for
 * VV_ERR_LAUNCH the HIP runtime's own text for the hipError_t carried in bits 8..); never printed by the library */
const char* vv_status_string(int status);
int vv_device_arch_ok(void); /* 1 when the current device is gfx950 */
/* compute units of the current device (256 on an MI355X in SPX mode; 256 when there is no device to ask): what the persistent kernels
 * size their grids with and what a host-side launch policy (k-splits, kernel routing) should use instead of a constant */
int vv_num_cus(void);
/* sizeof of a parameter struct as THIS library was compiled (which: 0 vv_view, 1 vv_conv_params, 2 vv_wgrad_params, 3 vv_pack_entry,
 * 4 vv_reduce_entry, 5 vv_fold_entry, 6 vv_bnbwd_params, 7 vv_outconv_params, 8 vv_conv2d_params; anything else: -1).  A binding
 * (cgo / ctypes) checks its own mirror of the struct against it once at load: vv_conv_params grew at its end in round 4. */
int vv_abi_sizeof(int32_t which);

#ifdef __cplusplus
}
#endif
#endif /* VECVAD_HIP_H */
