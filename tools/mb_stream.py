"""Sweep of the HBM stream kernels (csrc/lo_prof.hip): triad / copy / read-only x unroll x non-temporal, 1 GiB arrays.
The best copy figure is what bench.py reports as the box's measured ceiling (`roofline.copy_this_box`)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linear_operator_amd import _hip  # noqa: E402

dev = torch.device("cuda", 0)
for mode in ("copy", "triad", "read"):
    for unroll in (1, 2, 4, 8):
        for nt in (0, 1):
            g = _hip.hbm_stream_gbs(dev, mode, unroll=unroll, nt=nt)
            print(f"{mode:6s} unroll {unroll} nt {nt}: {g:8.1f} GB/s", flush=True)
for mode in ("copy", "triad", "read"):
    print(f"default {mode}: {_hip.hbm_stream_gbs(dev, mode):8.1f} GB/s")
for n in (1 << 24, 1 << 26, 1 << 28, 1 << 29):
    print(f"copy n={n}: {_hip.hbm_stream_gbs(dev, 'copy', n_floats=n):8.1f} GB/s")
