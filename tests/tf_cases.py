"""Shared helpers for the temporal-filter tests (test infrastructure)."""
import ctypes as C

import numpy as np

from svtav1_hip import abi

PITCH = 64          # the reference's prediction / accumulator blocks are 64x64 (BW x BH) per b64


def block_case(trial, bd, seed=0):
    """One 32x32 block against one reference prediction: arrays + the flat parameter block (host pointers)."""
    rng = np.random.default_rng(9100 + trial * 7 + bd + seed)
    is16 = bd > 8
    dt = np.uint16 if is16 else np.uint8
    hi = (1 << bd) - 1
    ss = 1
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
