// Score aggregation on the device:
//   per-cube reconstruction errors -> z-normalised, weighted cube score -> frame score -> frame-level ROC-AUC
// replacing test.py:330-358 (numpy + one 240x360 float64 mask per cube + torch.save/torch.load per frame) and
// utils.py:29-41 (sklearn roc_curve + auc).  Both kernels are tiny and latency-bound; they exist so that the scores never
// leave HBM between the UNet bank and the final AUC.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vecvad_hip.h"
#include "vv_common.h"

// Products and sums are rounded separately like numpy's: no FMA contraction in this translation unit.
#pragma clang fp contract(off)

namespace {

// The reference paints score m into mask[ceil(y1):ceil(y2), ceil(x1):ceil(x2)] (background -1e5), max-combines the masks
// and later takes mask.max(): that is max over the cubes whose painted rectangle is non-empty, and -1e5 for frames with
// none.  All arithmetic in float64 like numpy's (float32 score - float64 mean) / float64 std.
__global__ void __launch_bounds__(256) frame_score_kernel(const float* __restrict__ raw, const float* __restrict__ of,
                                                          const int32_t* __restrict__ frame_off,
                                                          const int32_t* __restrict__ cube_stat,
                                                          const double* __restrict__ stats,
                                                          const uint8_t* __restrict__ paints, double w_raw, double w_of,
                                                          double big, int n_frames, double* __restrict__ frame_scores) {
  int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n_frames) return;
  double best = frame_scores[f];
  for (int m = frame_off[f]; m < frame_off[f + 1]; ++m) {
    if (!paints[m]) continue;
    int s = cube_stat[m];
    double sc;
    if (s < 0) {
      sc = big;                                     // no model for this block: anomaly by construction (test.py:346-348)
    } else {
      const double* st = stats + 4 * (int64_t)s;
      sc = w_raw * (((double)raw[m] - st[0]) / st[1]);
      if (of) sc = sc + w_of * (((double)of[m] - st[2]) / st[3]);
    }
    best = fmax(best, sc);
  }
  frame_scores[f] = best;
}

// Mann-Whitney form of the ROC-AUC: AUC = (#{(p,n): s_p > s_n} + 0.5 #{s_p == s_n}) / (P N).  Exact integer counts, no
// sort: n^2 compares (n = 2 010 frames for UCSDped2, 40 791 for ShanghaiTech -> < 1 ms).
// out[0] += 2*wins + ties, out[1] = P, out[2] = N (written by block 0).
__global__ void __launch_bounds__(256) auc_count_kernel(const double* __restrict__ scores,
                                                        const uint8_t* __restrict__ labels, int n,
                                                        unsigned long long* __restrict__ out) {
  __shared__ double ss[256];
  __shared__ uint8_t sl[256];
  __shared__ unsigned long long red[4];
  int i = blockIdx.x * 256 + threadIdx.x;
  bool pos = i < n && labels[i] != 0;
  double si = i < n ? scores[i] : 0.0;
  unsigned long long acc = 0, npos = 0;
  for (int base = 0; base < n; base += 256) {
    int j = base + threadIdx.x;
    ss[threadIdx.x] = j < n ? scores[j] : 0.0;
    sl[threadIdx.x] = j < n ? (labels[j] != 0 ? 1 : 0) : 2;
    __syncthreads();
    if (pos) {
      for (int k = 0; k < 256; ++k) {
        if (sl[k] == 0) acc += si > ss[k] ? 2u : (si == ss[k] ? 1u : 0u);
      }
    }
    if (blockIdx.x == 0 && sl[threadIdx.x] == 1) npos += 1;
    __syncthreads();
  }
  // wave reduce, then across the 4 waves
  for (int o = 32; o > 0; o >>= 1) {
    acc += __shfl_down(acc, o);
    npos += __shfl_down(npos, o);
  }
  int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
  __syncthreads();
  if (blockIdx.x == 0) {
    if ((threadIdx.x & 63) == 0) red[wave] = npos;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long p = red[0] + red[1] + red[2] + red[3];
      out[1] = p;
      out[2] = (unsigned long long)n - p;
    }
  }
}

}  // namespace

extern "C" int vv_frame_scores(const float* raw, const float* of, const int32_t* frame_off, const int32_t* cube_stat,
                               const double* stats, const uint8_t* paints, double w_raw, double w_of, double big,
                               int32_t n_frames, double* frame_scores, vv_stream stream) {
  if (!raw || !frame_off || !cube_stat || !stats || !paints || !frame_scores || n_frames < 0) return VV_ERR_BAD_ARG;
  if (n_frames == 0) return VV_OK;
  VV_LAUNCH(frame_score_kernel, dim3((n_frames + 255) / 256), dim3(256), 0, (hipStream_t)stream, raw, of, frame_off,
            cube_stat, stats, paints, w_raw, w_of, big, n_frames, frame_scores);
  VV_CHECK_LAUNCH();
  return VV_OK;
}

extern "C" int vv_roc_auc_counts(const double* scores, const uint8_t* labels, int32_t n, uint64_t* out3,
                                 vv_stream stream) {
  if (!scores || !labels || !out3 || n < 0) return VV_ERR_BAD_ARG;
  {
    const hipError_t e = hipMemsetAsync(out3, 0, 3 * sizeof(uint64_t), (hipStream_t)stream);
    if (e != hipSuccess) return VV_HIP_STATUS(e);
  }
  if (n == 0) return VV_OK;
  VV_LAUNCH(auc_count_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, scores, labels, n,
            (unsigned long long*)out3);
  VV_CHECK_LAUNCH();
  return VV_OK;
}
