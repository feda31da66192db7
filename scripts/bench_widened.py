#!/usr/bin/env python3
"""Throughput of the SURVEY 8f ops on device-resident 1080p maps (algorithmic bytes / time, HIP events)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pysilent_amd import _runtime as rt

B, H, W = 32, 1080, 1920
dev = torch.device("cuda", 0)
value = torch.rand((B, H, W, 1), device=dev) * (torch.rand((B, H, W, 1), device=dev) > 0.5)
color = torch.rand((B, H, W, 3), device=dev) * 255
cells = torch.rand((B, H // 3, W // 3, 1), device=dev) * 255
state = torch.full_like(cells, 8.0)
from pysilent_amd import constant_convolutions as cc
blur = cc.blur_tensor(2, 7)
k333 = cc.midget_rgc(2)
from pysilent_amd.util.normalize import normalize_tensor_positive_negative as _norm
kcs = _norm(cc.center_surround_tensor(2, [1], [1], [1], [-1]))
kend = cc.end_bank(4)
from pysilent_amd.constant_convolutions import edge_orientation_detector as _eod
k7733 = _eod.rgb_2d_edge_tensors()


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts))


px = B * H * W
rows = [
    ("get_centroids 3x3 (value -> distance map + cell totals)", lambda: rt.centroids(value, 3, 3), px * 8 + px // 9 * 4),
    ("get_boosting step (cells of a 3x3 grid, in-place state)", lambda: rt.boosting_step(cells, state, 1.0, 1.0, 1, False),
     cells.numel() * (4 + 4 + 4 + 4 + 4)),
    ("affine_clip (1-channel map)", lambda: rt.affine_clip(value, div=255.0), px * 8),
    ("resize_nearest 1080p -> 655 x 1164", lambda: rt.resize_nearest(value, (655, 1164)), (px + B * 655 * 1164) * 4),
    ("select_peaks (3 ch + value -> peak value)", lambda: rt.select_peaks(color, 0.1, value, want=("peak_value",)),
     px * (12 + 4 + 4 + 4)),
    ("pad_inwards (3 ch)", lambda: rt.pad_inwards(color, 2, 2, 2, 2), px * 24),
    ("value_from_color (3 ch -> 1)", lambda: rt.value_from_color(color), px * 16),
    ("nms3x3 product (3 ch)", lambda: rt.nms3x3(color, "product"), px * 24),
    ("top_value_points (3 ch + value)", lambda: rt.top_value_points(color, 0.1, value), px * (4 + 12 + 4 + 12)),
    ("regulate 7x7 (3 ch)", lambda: rt.regulate(color, blur, 1.0, 0.1), px * 24),
    ("conv2d_same 3x3x3x3 + relu", lambda: rt.conv2d_same(color, k333, relu=True), px * 24),
    ("conv2d_same 3x3x1x1 + relu (gray CS)", lambda: rt.conv2d_same(value, kcs, relu=True), px * 8),
    ("conv2d_same 3x3x1x4 + relu + clip (gray end bank)", lambda: rt.conv2d_same(value, kend, relu=True, clip_hi=255.0), px * 20),
    ("conv2d_same 7x7x3x3 (thick-edge bank)", lambda: rt.conv2d_same(color, k7733), px * 24),
    ("max_value_indices_region (value -> int64 keypoints)", lambda: rt.max_value_indices_region(value, [(H // 2, W // 2)]), px * 4 * 3),
    ("boosting_step visualise (3-channel outputs)", lambda: rt.boosting_step(cells, state, 1.0, 1.0, 1, True), cells.numel() * 40),
]
for name, fn, nbytes in rows:
    ms = timed(fn)
    print("%-58s %7.3f ms  %6.0f GB/s algorithmic (incl. torch.empty of the outputs)" % (name, ms, nbytes / ms / 1e6))
