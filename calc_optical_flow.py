"""Optical-flow extraction driver -- drop-in for the reference's ``calc_optical_flow.py``.

For every frame of a dataset (``unified_dataset_interface(..., context_frame_num=1, border_mode='hard')``) the flow to
the next frame is estimated by FlowNet2 at 512x384 and written, resized back to the frame size WITHOUT rescaling the
vectors, to ``./optical_flow/<same sub-path as the frame>/<frame name>.npy`` as ``[h,w,2]`` float32
(calc_optical_flow.py:12-85).

What runs where: decoded frames are uploaded once per frame triple; both cv2-style resizes (``vv_crop_resize``), the
whole FlowNet2 forward (``vv_conv2d_mfma`` + correlation / resample2d / channelnorm kernels, replayed from a hipGraph)
and the layout changes stay on the GPU; only the final ``[h,w,2]`` field comes back for ``np.save``.
"""
import os

import numpy as np
import torch

from vad_datasets import unified_dataset_interface
from FlowNet2_src import FlowNet2
from vec_vad_amd.extract import crop_resize

FLOW_W, FLOW_H = 512, 384                    # cv2.resize(..., (512, 384)), calc_optical_flow.py:46-58
CHECKPOINT = 'FlowNet2_src/pretrained/FlowNet2_checkpoint.pth.tar'


def pair_of(frame_range):
    """Which two of the three context frames are matched (calc_optical_flow.py:43,61): normally (current, next); at a
    video border, where 'hard' clipping repeats the current frame, the first two entries."""
    if frame_range[1] == frame_range[0] or frame_range[1] == frame_range[2]:
        return 0, 1
    return 1, 2


def load_flownet2(path=CHECKPOINT, device='cuda'):
    """calc_optical_flow.py:15-22: keep the checkpoint entries whose key exists in the model."""
    net = FlowNet2()
    pretrained = torch.load(path, map_location='cpu', weights_only=False)['state_dict']
    own = net.state_dict()
    own.update({k: v for k, v in pretrained.items() if k in own})
    net.load_state_dict(own)
    return net.to(device).eval()


def flow_of_frames(flownet2, frames_chw, frame_range, graphed=True):
    """frames_chw: the dataset item, ``[3,C,H,W]`` (numpy or tensor, uint8 image frames).  Returns ``[H,W,2]`` float32
    (CUDA tensor)."""
    dev = next(flownet2.parameters()).device
    fr = torch.as_tensor(frames_chw).to(dev)
    if fr.dim() != 4 or fr.shape[0] != 3:
        raise ValueError('expected the 3-frame context stack [3,C,H,W] of context_frame_num=1, got %s' % (tuple(fr.shape),))
    a, b = pair_of(frame_range)
    pair = fr[[a, b]].permute(0, 2, 3, 1).contiguous()                       # [2,H,W,C]
    H, W = pair.shape[1], pair.shape[2]
    small = crop_resize(pair, np.array([[0, 0, W, H]], np.int32), FLOW_H, FLOW_W)[0]      # [2,384,512,C]
    if small.shape[3] == 1:
        small = small.expand(-1, -1, -1, 3)                                  # grey frames: same plane three times
    ims = small.permute(3, 0, 1, 2)[None].float().contiguous()               # [1,3,2,384,512], 0..255, BGR
    flow = (flownet2.forward_graphed(ims) if graphed else flownet2(ims))[0]  # [2,384,512]
    flow = flow.permute(1, 2, 0).contiguous()[None]                          # [1,384,512,2]
    return crop_resize(flow, np.array([[0, 0, FLOW_W, FLOW_H]], np.int32), H, W)[0, 0]


def flows_of_frames(flownet2, items, graphed=True):
    """Several frames per launch: ``items`` = [(frames_chw [3,C,H,W], frame_range), ...] of ONE frame size.  FlowNet2 at batch 1
    leaves most of the chip idle on its H/16 ... H/64 levels (a few hundred output pixels per layer); 4 pairs per launch take
    5.3 ms per pair instead of 6.8 on MI355X (1024x448), 8 pairs 5.0.  Returns a list of ``[H,W,2]`` float32 CUDA tensors.
    Per-pair results agree with the one-pair launch to fp32 round-off (the split-K choice of a layer depends on the batch)."""
    dev = next(flownet2.parameters()).device
    ims, sizes = [], []
    for frames_chw, frame_range in items:
        fr = torch.as_tensor(frames_chw).to(dev)
        if fr.dim() != 4 or fr.shape[0] != 3:
            raise ValueError('expected the 3-frame context stack [3,C,H,W] of context_frame_num=1, got %s' % (tuple(fr.shape),))
        a, b = pair_of(frame_range)
        pair = fr[[a, b]].permute(0, 2, 3, 1).contiguous()
        H, W = pair.shape[1], pair.shape[2]
        small = crop_resize(pair, np.array([[0, 0, W, H]], np.int32), FLOW_H, FLOW_W)[0]
        if small.shape[3] == 1:
            small = small.expand(-1, -1, -1, 3)
        ims.append(small.permute(3, 0, 1, 2).float())
        sizes.append((H, W))
    ims = torch.stack(ims).contiguous()                                      # [N,3,2,384,512]
    flow = flownet2.forward_graphed(ims) if graphed else flownet2(ims)       # [N,2,384,512]
    out = []
    for k, (H, W) in enumerate(sizes):
        f = flow[k].permute(1, 2, 0).contiguous()[None]
        out.append(crop_resize(f, np.array([[0, 0, FLOW_W, FLOW_H]], np.int32), H, W)[0, 0])
    return out


def flow_path(dataset, idx, of_root_dir='./optical_flow'):
    """optical_flow/<dataset dir components below the dataset root's parent>/<frame name>.npy (calc_optical_flow.py:13,27-37)."""
    skip = len(dataset.dir.split('/')) - 1
    addr = dataset.all_frame_addr[idx]
    name = addr.split('/')[-1].split('.')[0]
    return os.path.join(of_root_dir, *addr.split('/')[skip:-1]), name + '.npy'


def calc_optical_flow(dataset, flownet2=None, of_root_dir='./optical_flow', log=print, pairs_per_launch=4):
    """pairs_per_launch = 1 is the reference's frame-by-frame loop (calc_optical_flow.py:39-85); the default groups consecutive
    frames of equal size into one FlowNet2 launch (``flows_of_frames``)."""
    if flownet2 is None:
        flownet2 = load_flownet2()

    def save(idx, flow):
        of_path, fname = flow_path(dataset, idx, of_root_dir)
        os.makedirs(of_path, exist_ok=True)
        np.save(os.path.join(of_path, fname), flow.cpu().numpy())

    pend = []          # (idx, frames, frame_range) of one frame size

    def flush():
        if len(pend) == 1:
            save(pend[0][0], flow_of_frames(flownet2, pend[0][1], pend[0][2]))
        elif pend:
            for (idx, _, _), flow in zip(pend, flows_of_frames(flownet2, [(f, r) for _, f, r in pend])):
                save(idx, flow)
        del pend[:]

    for idx in range(len(dataset)):
        log('Calculating optical flow for {}-th frame'.format(idx + 1))
        batch, _ = dataset[idx]
        if pend and tuple(np.shape(batch)) != tuple(np.shape(pend[0][1])):
            flush()
        pend.append((idx, batch, dataset.context_range(idx)))
        if len(pend) >= max(1, pairs_per_launch):
            flush()
    flush()


if __name__ == '__main__':
    # Same example as the reference (calc_optical_flow.py:105-112); change dataset_name for the other datasets.
    dataset_name = 'UCSDped2'
    for mode in ('train', 'test'):
        calc_optical_flow(unified_dataset_interface(dataset_name=dataset_name, dir=os.path.join('raw_datasets', dataset_name),
                                                    context_frame_num=1, mode=mode, border_mode='hard'))
