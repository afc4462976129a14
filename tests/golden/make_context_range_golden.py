"""Golden vectors for the temporal-context index logic (reference vad_datasets.py:277-356, ``context_range``; the three
dataset classes carry identical copies).  Runs the REAL reference method on synthetic video layouts (authoring container
only; cv2 / torchvision are stubbed because only index arithmetic is exercised) and stores inputs + outputs as JSON."""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
np.int = int
sys.modules['cv2'] = types.ModuleType('cv2')
tv = types.ModuleType('torchvision')
tvt = types.ModuleType('torchvision.transforms')
tvt.Compose = lambda ts: None
tvt.ToTensor = lambda: None
tv.transforms = tvt
sys.modules['torchvision'] = tv
sys.modules['torchvision.transforms'] = tvt
sys.path.insert(0, '/root/reference')
import vad_datasets as R  # noqa: E402

cases = []
for lens in ([6, 9, 5], [12], [4, 4, 4, 7]):
    fvi = [v for v, n in enumerate(lens) for _ in range(n)]
    for mode in ('elastic', 'predict', 'hard'):
        for ctx in (1, 2, 4):
            obj = object.__new__(R.ped_dataset)
            obj.border_mode, obj.context_frame_num, obj.tot_frame_num, obj.frame_video_idx = mode, ctx, len(fvi), fvi
            for ind in range(len(fvi)):
                try:
                    out = obj.context_range(ind)
                except NotImplementedError:
                    out = [-1]                       # "video too short / context too large"
                cases.append([lens, mode, ctx, ind, [int(v) for v in out]])
json.dump(cases, open(os.path.join(HERE, 'context_range.json'), 'w'))
print(len(cases), 'cases')
