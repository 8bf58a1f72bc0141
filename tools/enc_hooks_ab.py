#!/usr/bin/env python3
"""Which whole-picture hooks pay inside the patched encoder when everything else runs the reference's x86 intrinsics?
    python tools/enc_hooks_ab.py [1080p|4k] [frames] [short]     (GPU box; prints one line per configuration; `short` = only the
    intrinsics baseline, the open-loop set and all hooks, each twice)"""
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "svt-av1-mod-by-patman_amd"))
from svtav1_hip import frames  # noqa: E402

APP = os.path.join(ROOT, "oracle", "_ref", "e2e", "SvtAv1EncApp")
LIB = os.path.join(ROOT, "svt-av1-mod-by-patman_amd", "csrc", "libsvtav1_hip.so")
HOOKS = ["PA", "ME", "TF", "TPL", "DLF", "CDEF", "LR"]


def main():
    big = len(sys.argv) > 1 and sys.argv[1] == "4k"
    W, H, N, bd = (3840, 2160, 9, 10) if big else (1920, 1080, 33, 8)
    if len(sys.argv) > 2:
        N = int(sys.argv[2])
    short = len(sys.argv) > 3 and sys.argv[3] == "short"
    cores = min(16, len(os.sched_getaffinity(0)))
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        path = os.path.join(tmp, "clip.yuv")
        rng = np.random.default_rng(1)
        with open(path, "wb") as f:
            for y in frames.synthetic_clip(W, H, N, seed=7):
                if bd == 8:
                    f.write(y.tobytes()), f.write(np.full((H // 2) * (W // 2) * 2, 128, np.uint8).tobytes())
                else:
                    f.write((y.astype(np.uint16) * 4 + rng.integers(0, 4, size=y.shape, dtype=np.uint16)).astype("<u2").tobytes())
                    f.write(np.full((H // 2) * (W // 2) * 2, 512, "<u2").tobytes())

        def run(tag, env):
            cmd = [APP, "-i", path, "-w", str(W), "-h", str(H), "--fps", "30", "-n", str(N), "--preset", "8", "--lp", str(cores), "--asm", "hip",
                   "--input-depth", str(bd), "-b", os.path.join(tmp, tag + ".ivf")]
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=dict(os.environ, **env))
            m = re.search(r"Average Speed:\s+([0-9.]+) fps", r.stdout)
            p = re.search(r"PCIe ([0-9.]+) MB up / ([0-9.]+) MB down", r.stdout)
            print(f"{tag:28s} {float(m.group(1)) if m else None} fps   PCIe {p.group(0) if p else '-'}", flush=True)

        base = {"SVTAV1_E2E_SIMD": "2", "SVTAV1_HIP_LIB": LIB, "SVTAV1_HIP_ONLY": "__none__"}
        if short:
            print(f"{W}x{H} {bd}-bit, {N} frames, --preset 8 --lp {cores}", flush=True)
            for _ in range(2):
                run("simd only", {"SVTAV1_E2E_SIMD": "1"})
                run("simd + PA ME TF TPL", dict(base, SVTAV1_HIP_TIERB_PA="1", SVTAV1_HIP_TIERB_ME="1", SVTAV1_HIP_TIERB_TF="1", SVTAV1_HIP_TIERB_TPL="1"))
                run("simd + all", dict(base, **{"SVTAV1_HIP_TIERB_" + h: "1" for h in HOOKS}))
            return
        run("simd only", {"SVTAV1_E2E_SIMD": "1"})
        run("simd + library, no hook", base)
        for h in HOOKS:
            run("simd + " + h, dict(base, **{"SVTAV1_HIP_TIERB_" + h: "1"}))
        run("simd + ME TF", dict(base, SVTAV1_HIP_TIERB_ME="1", SVTAV1_HIP_TIERB_TF="1"))
        run("simd + PA ME TF TPL", dict(base, SVTAV1_HIP_TIERB_PA="1", SVTAV1_HIP_TIERB_ME="1", SVTAV1_HIP_TIERB_TF="1", SVTAV1_HIP_TIERB_TPL="1"))
        run("simd + all", dict(base, **{"SVTAV1_HIP_TIERB_" + h: "1" for h in HOOKS}))


if __name__ == "__main__":
    main()
