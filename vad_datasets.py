"""Data-layer surface of the reference's ``vad_datasets.py`` that the cube-completion hot path touches.

Kept name- and behaviour-compatible:
  * ``cube_to_train_dataset``        reference vad_datasets.py:130-168 -- [T,H,W,C] cube -> ([T*C,H,W] input,
                                     [T_of*2,H,W] flow target, [T*C,H,W] copy); uint8 -> float/255, float32 untouched
  * ``frame_size``, ``img_tensor2numpy``, ``img_batch_tensor2numpy``, ``bbox_collate*``   (:16, :27-66)
  * ``get_inputs`` / ``get_foreground`` (:18-25, :70-93) -- the crop + cv2.resize of every box and context frame is ONE
                                     launch of the HIP kernel ``vv_crop_resize`` (vec_vad_amd/extract.py)
  * ``ped_dataset / avenue_dataset / shanghaiTech_dataset / unified_dataset_interface`` (:95-836) -- the frame indexers;
                                     one shared implementation (the reference carries three copies), ``context_range`` is
                                     pinned by reference-generated goldens
  * ``CubeStore``                    NEW: the cubes kept resident on the GPU in their on-disk layout; batches are
                                     gathered + converted by the HIP kernel vv_cube_gather (no per-sample python).
Image DECODING (cv2.imread in the reference) is outside the hot path: ``get_inputs`` uses cv2 when it is importable and
Pillow otherwise (BGR channel order like cv2; JPEG decoders may differ by an LSB -- unpinned, there are no frames here).
"""
import glob
import os
from collections import OrderedDict

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


def get_inputs(file_addr):
    """reference vad_datasets.py:18-25: ``.mat`` -> ['uv'], ``.npy`` -> array, anything else -> BGR uint8 image."""
    file_format = file_addr.split('.')[-1]
    if file_format == 'mat':
        import scipy.io as sio
        return sio.loadmat(file_addr, verify_compressed_data_integrity=False)['uv']
    if file_format == 'npy':
        return np.load(file_addr)
    return _imread(file_addr, gray=False)


def _imread(path, gray):
    try:
        import cv2
        return cv2.imread(path, cv2.IMREAD_GRAYSCALE) if gray else cv2.imread(path)
    except ImportError:
        from PIL import Image
        with Image.open(path) as im:
            if gray:
                return np.array(im.convert('L'))
            return np.ascontiguousarray(np.array(im.convert('RGB'))[:, :, ::-1])


def get_foreground(img, bboxes, patch_size):
    """reference vad_datasets.py:70-93 (``[C,H,W]`` or ``[T,C,H,W]`` numpy in, ``[n,(T,)C,P,P]`` numpy out); all boxes and
    frames go through one ``vv_crop_resize`` launch.  Raises without the HIP library / a GPU (no CPU fallback)."""
    from vec_vad_amd.extract import get_foreground as _gf
    return _gf(img, bboxes, patch_size)


class _frame_dataset(Dataset):
    """Frame indexer shared by the three dataset classes (the reference repeats it per dataset, vad_datasets.py:170-836):
    ``__getitem__(i)`` -> (frame ``[C,H,W]`` | context stack ``[T,C,H,W]`` | foreground patches ``[n,(T,)C,P,P]``, gt)."""

    def __init__(self, dir, mode='train', context_frame_num=0, border_mode='elastic', file_format=None, all_bboxes=None,
                 patch_size=32):
        if mode not in ('train', 'test'):
            raise NotImplementedError
        self.dir = dir
        self.mode = mode
        self.videos = OrderedDict()
        self.all_frame_addr = list()
        self.frame_video_idx = list()
        self.tot_frame_num = 0
        self.context_frame_num = context_frame_num
        self.border_mode = border_mode
        self.file_format = file_format if file_format is not None else self.default_format
        self.all_bboxes = all_bboxes
        self.patch_size = patch_size
        self.return_gt = False
        self.dataset_init()

    # -- layout hooks -------------------------------------------------------------------------------------------
    default_format = '.jpg'

    def _video_dirs(self):
        raise NotImplementedError

    def _load_gt(self):
        pass

    def _gt(self, indice):
        raise NotImplementedError

    def _on_video(self, name, n_frames):
        pass

    # -- shared ---------------------------------------------------------------------------------------------------
    def dataset_init(self):
        for vid, video in enumerate(self._video_dirs(), start=1):
            name = video.split('/')[-1]
            frames = sorted(glob.glob(os.path.join(video, '*' + self.file_format)))
            self.videos[name] = {'path': video, 'frame': frames, 'length': len(frames)}
            self.frame_video_idx += [vid] * len(frames)
            self.all_frame_addr += frames
            self._on_video(name, len(frames))
        self.tot_frame_num = len(self.all_frame_addr)
        if self.mode == 'test':
            self._load_gt()

    def __len__(self):
        return self.tot_frame_num

    def context_range(self, indice):
        return context_range(indice, self.border_mode, self.context_frame_num, self.tot_frame_num, self.frame_video_idx)

    def frames(self, indice):
        """Decoded frame ``[C,H,W]`` (no context) or context stack ``[T,C,H,W]`` of frame ``indice``."""
        if self.context_frame_num == 0:
            return np.transpose(get_inputs(self.all_frame_addr[indice]), [2, 0, 1])
        return np.array([np.transpose(get_inputs(self.all_frame_addr[i]), [2, 0, 1])
                         for i in self.context_range(indice)])

    def __getitem__(self, indice):
        img_batch = self.frames(indice)
        if self.all_bboxes is not None:
            img_batch = get_foreground(img=img_batch, bboxes=self.all_bboxes[indice], patch_size=self.patch_size)
        img_batch = torch.from_numpy(np.ascontiguousarray(img_batch))
        if self.mode == 'test' and self.return_gt:
            return img_batch, torch.from_numpy(self._gt(indice))
        return img_batch, torch.zeros(1)  # to unify the interface

    def cubes_device(self, indice, device='cuda'):
        """NEW: foreground cubes of frame ``indice`` as a device tensor ``[n,T,P,P,C]`` (CubeStore layout), without the
        D2H / H2D round trip of ``__getitem__`` -> ``img_batch_tensor2numpy`` -> ``np.save``."""
        from vec_vad_amd.extract import foreground_cubes
        fr = self.frames(indice)
        fr = fr[None] if fr.ndim == 3 else fr
        fr = torch.from_numpy(np.ascontiguousarray(np.transpose(fr, [0, 2, 3, 1]))).to(device)
        return foreground_cubes(fr, self.all_bboxes[indice], self.patch_size)


class ped_dataset(_frame_dataset):
    """UCSD ped1 / ped2 (reference vad_datasets.py:170-402): ``<dir>/Train/Train*/`` and ``<dir>/Test/Test*/`` frame
    folders, pixel masks in ``Test*_gt/*.bmp``."""
    default_format = '.tif'

    def __init__(self, dir, **kw):
        self.h, self.w = (158, 238) if dir[-1] == '1' else (240, 360)
        self.all_gt_addr = list()
        self.gts = OrderedDict()
        super().__init__(dir, **kw)

    def _video_dirs(self):
        sub = 'Train' if self.mode == 'train' else 'Test'
        dirs = sorted(glob.glob(os.path.join(self.dir, sub, '*')))
        self._gt_dirs = [d for d in dirs if '_gt' in d] if self.mode == 'test' else []
        return [d for d in dirs if '_gt' not in d and sub in d.split('/')[-1]]

    def _load_gt(self):
        self.return_gt = len(self._gt_dirs) > 0
        for gt in self._gt_dirs:
            frames = sorted(glob.glob(os.path.join(gt, '*.bmp')))
            self.gts[gt.split('/')[-1]] = {'gt_frame': frames}
            self.all_gt_addr += frames

    def _gt(self, indice):
        return _imread(self.all_gt_addr[indice], gray=True)


class avenue_dataset(_frame_dataset):
    """CUHK Avenue (reference vad_datasets.py:404-618): ``training|testing/frames/<video>/``, ground truth
    ``ground_truth_demo/testing_label_mask/<k>_label.mat['volLabel']``."""

    def _video_dirs(self):
        sub = 'training' if self.mode == 'train' else 'testing'
        return sorted(glob.glob(os.path.join(self.dir, sub, 'frames', '*')))

    def _load_gt(self):
        gt_dir = os.path.join(self.dir, 'ground_truth_demo', 'testing_label_mask')
        self.return_gt = os.path.exists(gt_dir)
        if self.return_gt:
            import scipy.io as sio
            gts = [sio.loadmat(os.path.join(gt_dir, str(x + 1) + '_label.mat'))['volLabel']
                   for x in range(len(self.videos))]
            self.all_gt = np.concatenate(gts, axis=1)

    def _gt(self, indice):
        return self.all_gt[0, indice]


class shanghaiTech_dataset(_frame_dataset):
    """ShanghaiTech (reference vad_datasets.py:620-836): ``training/videosFrame/<scene>_<clip>/`` and
    ``Testing/frames_part{1,2}/<video>/``; frame-level ground truth ``Testing/test_frame_mask/*.npy``.  ``scene_idx`` is 1
    for every frame (one model for all scenes, :669), ``save_scene_idx`` keeps the real scene of the folder name."""

    def __init__(self, dir, **kw):
        self.save_scene_idx = list()
        self.scene_idx = list()
        self.scene_num = 0
        super().__init__(dir, **kw)
        self.scene_num = len(set(self.scene_idx))

    def _video_dirs(self):
        if self.mode == 'train':
            return sorted(glob.glob(os.path.join(self.dir, 'training', 'videosFrame', '*')))
        base = os.path.join(self.dir, 'Testing', 'frames_part')
        return [v for j in (1, 2) for v in sorted(glob.glob(os.path.join(base + str(j), '*')))]

    def _on_video(self, name, n_frames):
        self.save_scene_idx += [int(name[:2])] * n_frames
        self.scene_idx += [1] * n_frames

    def _load_gt(self):
        gt_dir = os.path.join(self.dir, 'Testing', 'test_frame_mask')
        self.return_gt = os.path.exists(gt_dir)
        if self.return_gt:
            self.all_gt = np.concatenate([np.load(g) for g in sorted(glob.glob(os.path.join(gt_dir, '*')))], axis=0)

    def _gt(self, indice):
        return np.array([self.all_gt[indice]])


def unified_dataset_interface(dataset_name, dir, mode='train', context_frame_num=0, border_mode='elastic',
                              file_format=None, all_bboxes=None, patch_size=32):
    """reference vad_datasets.py:95-113."""
    if file_format is None:
        if dataset_name not in frame_size:
            raise NotImplementedError
        file_format = frame_size[dataset_name][2]
    cls = {'UCSDped1': ped_dataset, 'UCSDped2': ped_dataset, 'avenue': avenue_dataset,
           'ShanghaiTech': shanghaiTech_dataset}.get(dataset_name)
    if cls is None:
        raise NotImplementedError
    return cls(dir=dir, context_frame_num=context_frame_num, mode=mode, border_mode=border_mode, all_bboxes=all_bboxes,
               patch_size=patch_size, file_format=file_format)
