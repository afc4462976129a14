"""Data-layer surface of the reference's ``vad_datasets.py`` that the cube-completion hot path touches.

In scope (kept name- and behaviour-compatible):
  * ``cube_to_train_dataset``        reference vad_datasets.py:130-168 -- [T,H,W,C] cube -> ([T*C,H,W] input,
                                     [T_of*2,H,W] flow target, [T*C,H,W] copy); uint8 -> float/255, float32 untouched
  * ``frame_size``, ``img_tensor2numpy``, ``img_batch_tensor2numpy``, ``bbox_collate*``   (:16, :27-66)
  * ``CubeStore``                    NEW: the same cubes kept resident on the GPU in their on-disk layout; batches are
                                     gathered + converted by the HIP kernel vv_cube_gather (no per-sample python).
Out of scope for this round: the video frame indexers ``ped_dataset / avenue_dataset /
shanghaiTech_dataset`` and ``get_foreground`` (cv2 file IO + cv2.resize); ``unified_dataset_interface`` raises with
that explanation.  The hot path is fed from the saved cube files (``*_foreground_saved = True`` in config.cfg).
"""
import numpy as np
import torch
from torch.utils.data import Dataset

# (h, w, file_format, scene_num) -- reference vad_datasets.py:16
frame_size = {'UCSDped1': (158, 238, '.tif', 1), 'UCSDped2': (240, 360, '.tif', 1), 'avenue': (360, 640, '.jpg', 1),
              'ShanghaiTech': (480, 856, '.jpg', 1)}


def _to_tensor(pic):
    """torchvision.transforms.ToTensor for ndarrays: HWC -> CHW, uint8 -> float32 / 255, other dtypes unchanged."""
    img = torch.from_numpy(np.ascontiguousarray(np.transpose(pic, (2, 0, 1))))
    if img.dtype == torch.uint8:
        return img.float().div(255)
    return img


transform = _to_tensor


def img_tensor2numpy(img):
    if isinstance(img, np.ndarray):
        return torch.from_numpy(np.transpose(img, [2, 0, 1]))
    return np.transpose(img, [1, 2, 0]).numpy()


def img_batch_tensor2numpy(img_batch):
    if isinstance(img_batch, np.ndarray):
        if img_batch.ndim == 4:
            return torch.from_numpy(np.transpose(img_batch, [0, 3, 1, 2]))
        return torch.from_numpy(np.transpose(img_batch, [0, 1, 4, 2, 3]))
    a = img_batch.numpy()
    if a.ndim == 4:
        return np.transpose(a, [0, 2, 3, 1])
    return np.transpose(a, [0, 1, 3, 4, 2])


def bbox_collate_train(batch):
    return torch.cat([x[0] for x in batch], dim=0), [x[1] for x in batch]


def bbox_collate_test(batch):
    return [x[0] for x in batch], [x[1] for x in batch]


class bbox_collate:
    def __init__(self, mode):
        self.mode = mode

    def collate(self, batch):
        if self.mode == 'train':
            return bbox_collate_train(batch)
        if self.mode == 'test':
            return bbox_collate_test(batch)
        raise NotImplementedError


class cube_to_train_dataset(Dataset):
    """Per-sample adapter with the reference's exact tensor layout (channel index = t*C + c)."""

    def __init__(self, data, target=None, tranform=transform):
        if data.ndim == 4:
            data = data[:, np.newaxis]
        if target is not None and target.ndim == 4:
            target = target[:, np.newaxis]
        self.data, self.target, self.transform = data, target, tranform

    def __len__(self):
        return self.data.shape[0]

    @staticmethod
    def _fold(cube):      # [T,H,W,C] -> [H,W,T*C]
        c = np.transpose(cube, [1, 2, 0, 3])
        return np.reshape(c, (c.shape[0], c.shape[1], -1))

    def __getitem__(self, indice):
        cur = self.data[indice]
        t = self.transform if self.transform is not None else (lambda a: a)
        if self.target is None:
            return t(self._fold(cur[:-1])), t(cur[-1])
        return t(self._fold(cur)), t(self._fold(self.target[indice])), t(self._fold(cur.copy()))


class CubeStore:
    """Device-resident cube set in the reference's on-disk layout: raw uint8 [N,5,32,32,3] (+ flow fp32 [N,Tf,32,32,2]).
    ``UNetBank.set_input_cubes(store.raw, store.flow, idx)`` gathers a batch with one HIP launch."""

    def __init__(self, raw, flow, device='cuda'):
        raw = np.asarray(raw)
        flow = np.asarray(flow)
        if raw.ndim == 4:
            raw = raw[:, None]
        if flow.ndim == 4:
            flow = flow[:, None]
        if raw.dtype != np.uint8:
            raise TypeError('raw cubes are expected as uint8 (cv2.resize of uint8 frames, vad_datasets.py:70-93); '
                            'got %s' % raw.dtype)
        self.raw = torch.from_numpy(np.ascontiguousarray(raw)).to(device)
        self.flow = torch.from_numpy(np.ascontiguousarray(flow, dtype=np.float32)).to(device)
        self.n = raw.shape[0]

    def __len__(self):
        return self.n


def context_range(indice, border_mode, context_frame_num, tot_frame_num, frame_video_idx):
    """Frame indices of the temporal context of frame ``indice`` (reference vad_datasets.py:277-356; the three dataset
    classes hold identical copies).  'elastic' keeps a full centred window by shifting it, 'predict' takes the frames up
    to and including ``indice``, anything else ('hard') clips a centred window; frames that would come from a
    neighbouring video are dropped and the window is refilled by repeating its first / last frame.
    Raises NotImplementedError where the reference does (video shorter than the window)."""
    n, last = context_frame_num, tot_frame_num - 1
    if border_mode == 'elastic':
        # the reference moves ``indice`` itself, so the 'own video' below is the one of the shifted centre
        indice = indice if n <= indice <= last - n else (n if indice < n else last - n)
        start, end, need = indice - n, indice + n, 2 * n + 1
    elif border_mode == 'predict':
        start, end, need = max(indice - n, 0), indice, n + 1
    else:
        start, end, need = max(indice - n, 0), min(indice + n, last), 2 * n + 1
    vids = list(frame_video_idx[start:end + 1])
    pad = need - len(vids)
    if pad > 0:
        vids = [vids[0]] * pad + vids if start == 0 else vids + [vids[-1]] * pad
    rel = [v - frame_video_idx[indice] for v in vids]
    offset = sum(rel)                              # signed count of frames that belong to a neighbouring video
    if rel[0] != 0 and rel[-1] != 0:
        raise NotImplementedError('The video is too short or the context frame number is too large!')
    if pad == 0 and offset == 0:
        return list(range(start, end + 1))
    if border_mode == 'elastic':
        return list(range(start - offset, end - offset + 1))
    if pad > 0 and offset != 0:
        raise NotImplementedError('The video is too short or the context frame number is too large!')
    if border_mode == 'predict':
        idx = list(range(start - offset, end + 1))
        return [idx[0]] * max(abs(offset), pad) + idx
    if offset > 0:
        idx = list(range(start, end - offset + 1))
        return idx + [idx[-1]] * offset
    if offset < 0:
        idx = list(range(start - offset, end + 1))
        return [idx[0]] * (-offset) + idx
    idx = list(range(start, end + 1))
    return [idx[0]] * pad + idx if start == 0 else idx + [idx[-1]] * pad


def unified_dataset_interface(dataset_name, dir, mode='train', context_frame_num=0, border_mode='elastic',
                              file_format=None, all_bboxes=None, patch_size=32):
    raise NotImplementedError('the video frame indexers (reference vad_datasets.py:170-836, cv2 file IO) are out of scope '
                              'for this round (SURVEY.md section 8 f-1); the hot path is fed from saved cube .npy files')
