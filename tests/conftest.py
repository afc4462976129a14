import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Small containers (the CPU-only build box: 8 cores shared with other jobs): an OpenMP team as wide as the machine spins against the
# neighbours' load, and the oracle's torch CPU convolutions then take minutes instead of seconds (measured: the same 34 tests in 40 s or
# in 610 s, about every second run; with half the cores still one run in five).  Half the cores AND a passive wait policy: 8 runs of 8
# at 70 - 95 s.  Must happen before torch is imported; a value
# set by the caller wins, and large hosts (the GPU box) keep their default.
if (os.cpu_count() or 8) <= 16:
    os.environ.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // 2)))
    os.environ.setdefault('OMP_WAIT_POLICY', 'passive')      # idle team members sleep instead of spinning on a shared core
    os.environ.setdefault('GOMP_SPINCOUNT', '0')
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu through gpurun)')


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)
