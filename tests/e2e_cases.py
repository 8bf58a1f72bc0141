"""Shared helpers of the end-to-end bitstream gate (test infrastructure): the synthetic clip of SURVEY.md 8d, the
patched reference encoder built by `make -C oracle e2e` (oracle/_ref/e2e/SvtAv1EncApp = the reference's own sources +
tools/reference_hip.patch) and md5 of what it writes."""
import hashlib
import json
import os
import subprocess

import numpy as np

from svtav1_hip import frames

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
APP = os.path.join(ROOT, "oracle", "_ref", "e2e", "SvtAv1EncApp")
HIP_LIB = os.path.join(ROOT, "svt-av1-mod-by-patman_amd", "csrc", "libsvtav1_hip.so")
GOLDEN = os.path.join(ROOT, "tests", "golden", "e2e_md5.json")

# name -> (width, height, frames, bit depth, preset): <= 8 frames of 192x128
CASES = {
    "p12_8bit": (192, 128, 8, 8, 12),
    "p8_8bit": (192, 128, 8, 8, 8),
    "p8_10bit": (192, 128, 6, 10, 8),
}


# larger clips with ragged sizes (neither dimension a multiple of 64 or 16: partial 64x64 / 16x16 blocks at the right and bottom
# edges, several ME / TF / TPL segments) — for the batched (Tier B) paths only: one PCIe round trip per leaf call makes the
# Tier A run of such a clip take minutes
TIER_B_CASES = {
    "p8_8bit_ragged": (424, 232, 17, 8, 8),
    "p10_10bit_ragged": (376, 216, 10, 10, 10),   # preset 10 = tpl level 5: an even number of 16x16 columns / rows, see svt_hip_bind_tpl.c
}
# clips for the in-loop-filter and picture-analysis hooks (row h): presets where each stage runs its search (frame-level
# deblocking with the level search at <= M5 and on base-layer pictures at M6, Wiener at M4 - M8, self-guided restoration at <= M3, CDEF
# everywhere), 8 and 10 bit, <= 10 frames
LF_CASES = {
    "p5_8bit_lf": (256, 192, 6, 8, 5),
    "p6_10bit_lf": (256, 192, 7, 10, 6),   # BASELINE.json configs[3]'s preset
    "p3_8bit_lf": (192, 128, 3, 8, 3),
}
# low-delay prediction structure (`--pred-struct 1`): the temporal filter runs produce_temporally_filtered_pic_ld
LD_CASES = {
    "p8_8bit_ld": (1280, 720, 7, 8, 8),    # the low-delay temporal filter needs >= 720p (derive_tf_params, enc_handle.c:3307-3314)
    "p8_10bit_ld": (1280, 720, 6, 10, 8),
}
EXTRA_ARGS = {"p8_8bit_ld": ["--pred-struct", "1"], "p8_10bit_ld": ["--pred-struct", "1"]}
# presets <= M2 run tf level 1: the temporal filter with 8x8 prediction (enable_8x8_pred)
TF8_CASES = {
    "p2_8bit_tf8": (192, 128, 8, 8, 2),
    "p2_10bit_tf8": (192, 128, 6, 10, 2),
}
ALL_CASES = dict(CASES, **TIER_B_CASES, **LF_CASES, **LD_CASES, **TF8_CASES)


def have_app():
    return os.path.exists(APP)


def write_clip(path, w, h, n, bd, seed=7):
    """Planar I420 (8-bit) / I420p10le: luma = the panning low-pass noise of frames.synthetic_clip, chroma a
    slow gradient so that the chroma transforms / filters see something; 10-bit = 8-bit * 4 + uniform {0..3}."""
    rng = np.random.default_rng(seed + 100)
    luma = frames.synthetic_clip(w, h, n, seed=seed)
    yy, xx = np.mgrid[0:h // 2, 0:w // 2]
    with open(path, "wb") as f:
        for i, y in enumerate(luma):
            u = ((xx * 2 + i * 3) % 256).astype(np.uint8)
            v = ((yy * 3 + 255 - i * 2) % 256).astype(np.uint8)
            for pl in (y, u, v):
                if bd == 8:
                    f.write(pl.tobytes())
                else:
                    f.write((pl.astype(np.uint16) * 4 + rng.integers(0, 4, size=pl.shape, dtype=np.uint16)).astype("<u2").tobytes())


def encode(case, workdir, asm, lp=1, env_extra=None, timeout=900):
    """Runs the encoder; returns ({'ivf': md5, 'recon': md5}, log text)."""
    w, h, n, bd, preset = ALL_CASES[case]
    clip = os.path.join(workdir, f"{case}.yuv")
    if not os.path.exists(clip):
        write_clip(clip, w, h, n, bd)
    ivf, rec = os.path.join(workdir, f"{case}_{asm}.ivf"), os.path.join(workdir, f"{case}_{asm}_rec.yuv")
    for p in (ivf, rec):
        if os.path.exists(p):
            os.remove(p)
    cmd = [APP, "-i", clip, "-w", str(w), "-h", str(h), "--fps", "30", "-n", str(n), "--preset", str(preset),
           "--lp", str(lp), "--asm", asm, "--input-depth", str(bd), "-b", ivf, "-o", rec] + EXTRA_ARGS.get(case, [])
    # SVTAV1_HIP_MIRROR_VERIFY: every hit of the glue's device-resident picture mirrors is compared with the host buffer
    env = dict(os.environ, SVTAV1_HIP_LIB=HIP_LIB, SVTAV1_HIP_MIRROR_VERIFY="1")
    env.update(env_extra or {})
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=timeout)
    if r.returncode != 0:
        raise RuntimeError(f"{' '.join(cmd)} failed ({r.returncode}):\n{r.stdout[-3000:]}")
    md5 = {k: hashlib.md5(open(p, "rb").read()).hexdigest() for k, p in (("ivf", ivf), ("recon", rec))}
    return md5, r.stdout


def golden():
    with open(GOLDEN) as f:
        return json.load(f)


def assert_hip_ran_clean(log):
    """A `--asm hip` encode must have kept its HIP leaves to the end: after a HIP error inside a leaf the library restores the
    C pointers and the encode finishes on the CPU (runtime.cpp, "HIP hot path disabled"), which reproduces the golden md5
    trivially.  Same for a batched hook that declined or failed ("stays on the CPU", "falls back")."""
    for needle in ("HIP hot path disabled", "HIP hot path unavailable", "STALE mirror"):
        assert needle not in log, f"the encode did not stay on the HIP path ({needle!r}):\n" + log[-2000:]
