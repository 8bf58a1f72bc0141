"""Shared helpers for the temporal-filter tests (test infrastructure)."""
import ctypes as C

import numpy as np

from svtav1_hip import abi

PITCH = 64          # the reference's prediction / accumulator blocks are 64x64 (BW x BH) per b64


def block_case(trial, bd, seed=0, ss=1):
    """One 32x32 block against one reference prediction: arrays + the flat parameter block (host pointers)."""
    rng = np.random.default_rng(9100 + trial * 7 + bd + seed)
    is16 = bd > 8
    dt = np.uint16 if is16 else np.uint8
    hi = (1 << bd) - 1
    arrs = {}
    strides_src = [96, 52, 52]
    noise = (1, 3, 8, 20, 60)[trial % 5] * (1 << (bd - 8))
    for pl in range(3):
        sz = 32 if pl == 0 else 32 >> ss
        src = rng.integers(0, hi + 1, size=(sz, strides_src[pl])).astype(dt)
        pred = np.zeros((sz, PITCH), dt)
        pred[:, :sz] = np.clip(src[:, :sz].astype(np.int32) + rng.integers(-noise, noise + 1, size=(sz, sz)), 0, hi)
        pred[:, sz:] = rng.integers(0, hi + 1, size=(sz, PITCH - sz))
        acc = rng.integers(0, 200000, size=(sz, PITCH)).astype(np.uint32)
        cnt = rng.integers(0, 3000, size=(sz, PITCH)).astype(np.uint16)
        arrs[pl] = [src, pred, acc, cnt]
    b = abi.TfBlock()
    for pl in range(3):
        src, pred, acc, cnt = arrs[pl]
        b.src[pl], b.pred[pl], b.accum[pl], b.count[pl] = src.ctypes.data, pred.ctypes.data, acc.ctypes.data, cnt.ctypes.data
        b.src_stride[pl], b.pred_stride[pl] = src.shape[1], PITCH
        b.decay_factor_fp16[pl] = int(rng.integers(1 << 10, 1 << 22))
    b.split = trial % 2
    for i in range(4):
        b.mv_x[i], b.mv_y[i] = int(rng.integers(-200, 201)), int(rng.integers(-200, 201))
        b.block_error[i] = int(rng.integers(0, 1 << (20 if is16 else 16))) << (4 if is16 else 0)
    if trial % 7 == 0:
        b.mv_x[0] = b.mv_y[0] = 0
    b.mv_dist_th = int(rng.choice([1, 16, 64, 300]))
    b.chroma, b.ss_x, b.ss_y, b.is_16bit, b.bit_depth = int(trial % 3 != 2), ss, ss, int(is16), bd
    b.zz_based = int(trial % 4 == 3)
    return b, arrs


def noise_cases():
    """(plane, width, height, stride, bit depth) for the noise estimate: textures with every mix of smooth and edge samples,
    degenerate sizes (no interior / fewer than SMOOTH_THRESHOLD smooth samples) and a frame-sized plane."""
    rng = np.random.default_rng(77)
    k = 0
    for w, h in ((64, 48), (131, 77), (3, 3), (2, 9), (9, 2), (6, 5), (257, 33), (640, 360), (1920, 1080)):
        for bd in (8, 10, 12):
            if w * h > 100000 and bd == 12:
                continue
            stride = w + int(rng.integers(0, 17))
            hi = (1 << bd) - 1
            kind = k % 5
            if kind == 0:       # smooth ramp + mild noise: most samples qualify
                base = np.add.outer(np.arange(h), np.arange(stride)) * (hi / 512.0) % (hi * 0.8)
                img = base + rng.normal(0, 2.0 * (1 << (bd - 8)), size=(h, stride))
            elif kind == 1:     # white noise: almost everything is an edge
                img = rng.integers(0, hi + 1, size=(h, stride))
            elif kind == 2:     # flat: all smooth, Laplacian zero
                img = np.full((h, stride), hi // 3)
            elif kind == 3:     # blocks with hard edges + noise
                img = (np.add.outer(np.arange(h) // 8, np.arange(stride) // 8) % 2) * (hi * 0.6) + rng.normal(0, 3.0 * (1 << (bd - 8)), size=(h, stride))
            else:               # extremes
                img = rng.integers(0, 2, size=(h, stride)) * hi
            img = np.clip(np.rint(img), 0, hi).astype(np.uint8 if bd == 8 else np.uint16)
            k += 1
            yield img, w, h, stride, bd
