"""Cube extraction on the GPU: crop + cv2-style bilinear resize of every box of a frame stack
in ONE launch of ``vv_crop_resize`` (reference vad_datasets.py:70-93 ``get_foreground`` does one ``cv2.resize`` call per
box per frame on the host), and the whole-frame resizes of calc_optical_flow.py:46-59,82.

There is no CPU fallback: without libvecvad_hip.so / a gfx950 device these functions raise.
"""
import math

import numpy as np
import torch

from . import _lib


def boxes_to_crops(bboxes, H, W):
    """``img[..., ceil(y1):ceil(y2), ceil(x1):ceil(x2)]`` of vad_datasets.py:74-76 as int32 ``[n,4]`` (x_min, y_min, x_max,
    y_max) after numpy's clipping of slice ends to the array.  An empty crop raises, as ``cv2.resize`` does on it."""
    crops = np.zeros((len(bboxes), 4), np.int32)
    for i, b in enumerate(bboxes):
        x0, x1 = int(math.ceil(b[0])), int(math.ceil(b[2]))
        y0, y1 = int(math.ceil(b[1])), int(math.ceil(b[3]))
        x0, y0, x1, y1 = max(x0, 0), max(y0, 0), min(x1, W), min(y1, H)
        if x1 <= x0 or y1 <= y0:
            raise ValueError('box %d (%s) selects an empty crop of the %dx%d frame (cv2.resize asserts !ssize.empty())'
                             % (i, list(map(float, b[:4])), H, W))
        crops[i] = (x0, y0, x1, y1)
    return crops


def crop_resize(frames, crops, out_h, out_w):
    """frames: CUDA tensor ``[T,H,W,C]`` uint8 or float32 (contiguous); crops: int32 ``[n,4]`` (numpy or CUDA tensor).
    Returns a CUDA tensor ``[n,T,out_h,out_w,C]`` of the same dtype."""
    if not frames.is_cuda:
        raise _lib.VecVadHipError('crop_resize needs the frames in HBM (CUDA tensor); vec_vad_amd has no CPU path')
    if frames.dtype not in (torch.uint8, torch.float32):
        raise TypeError('crop_resize handles uint8 and float32 frames (the two dtypes the reference resizes), got %s'
                        % frames.dtype)
    if frames.dim() != 4 or not frames.is_contiguous():
        raise ValueError('frames must be a contiguous [T,H,W,C] tensor')
    T, H, W, C = frames.shape
    if not torch.is_tensor(crops):
        crops = torch.from_numpy(np.ascontiguousarray(crops, dtype=np.int32))
    crops = crops.to(device=frames.device, dtype=torch.int32).contiguous()
    n = crops.shape[0]
    out = torch.empty((n, T, out_h, out_w, C), dtype=frames.dtype, device=frames.device)
    if n:
        _lib.check(_lib.lib().vv_crop_resize(frames.data_ptr(), int(frames.dtype == torch.float32), T, H, W, C,
                                            crops.data_ptr(), n, out_h, out_w, out.data_ptr(),
                                            torch.cuda.current_stream(frames.device).cuda_stream), 'vv_crop_resize')
    return out


def resize(img, dsize, device='cuda'):
    """``cv2.resize(img, dsize)`` (default INTER_LINEAR) for an HxW or HxWxC uint8 / float32 numpy image; ``dsize=(w,h)``."""
    img = np.asarray(img)
    squeeze = img.ndim == 2
    a = img[:, :, None] if squeeze else img
    H, W, _ = a.shape
    fr = torch.from_numpy(np.ascontiguousarray(a)).to(device)[None]
    out = crop_resize(fr, np.array([[0, 0, W, H]], np.int32), int(dsize[1]), int(dsize[0]))[0, 0].cpu().numpy()
    return out[:, :, 0] if squeeze else out


def get_foreground(img, bboxes, patch_size, device='cuda'):
    """Drop-in for reference vad_datasets.py:70-93: ``img`` ``[C,H,W]`` or ``[T,C,H,W]`` (numpy) -> ``[n,C,P,P]`` or
    ``[n,T,C,P,P]`` numpy patches.  One upload, one launch, one download for all boxes and frames."""
    img = np.asarray(img)
    single = img.ndim == 3
    fr = img[None] if single else img
    fr = torch.from_numpy(np.ascontiguousarray(np.transpose(fr, [0, 2, 3, 1]))).to(device)
    if len(bboxes) == 0:
        return np.array([])                           # np.array(list()) in the reference
    out = foreground_cubes(fr, bboxes, patch_size)    # [n,T,P,P,C]
    out = out.permute(0, 1, 4, 2, 3).contiguous().cpu().numpy()
    return out[:, 0] if single else out


def foreground_cubes(frames, bboxes, patch_size):
    """Device-resident variant: frames CUDA ``[T,H,W,C]`` -> cubes CUDA ``[n,T,P,P,C]`` (the layout of the saved
    ``*_foreground_*.npy`` cube files and of ``CubeStore``), no host round trip."""
    crops = boxes_to_crops(bboxes, frames.shape[1], frames.shape[2])
    return crop_resize(frames, crops, patch_size, patch_size)
