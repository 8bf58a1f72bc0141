"""Encoder-level A/B on the GPU box: the patched reference encoder (oracle/_ref/e2e/SvtAv1EncApp) on a 33-frame 1080p clip with
`--asm c`, with its open-loop ME on the GPU, and with ME + temporal filter on the GPU (tools/reference_hip.patch steps 2b / 6b)."""
import os, re, subprocess, sys, tempfile, time
sys.path.insert(0, "svt-av1-mod-by-patman_amd"); sys.path.insert(0, "tests")
import numpy as np
from svtav1_hip import frames
ROOT = os.getcwd()
app = os.path.join(ROOT, "oracle/_ref/e2e/SvtAv1EncApp")
lib = os.path.join(ROOT, "svt-av1-mod-by-patman_amd/csrc/libsvtav1_hip.so")
W, H, N = 1920, 1080, 33
PRESET = sys.argv[1] if len(sys.argv) > 1 else "8"          # usage: python tools/enc_fps_ab.py [preset [lp ...]]
LPS = [int(x) for x in sys.argv[2:] if x != "10bit"] or [16, 4]
BD = 10 if "10bit" in sys.argv else 8      # "10bit" anywhere behind the preset: a 10-bit clip (the 8-bit samples * 4 + noise)
tmp = tempfile.mkdtemp(dir="/tmp")
clip = frames.synthetic_clip(W, H, N, seed=7)
path = os.path.join(tmp, "c.yuv")
rng = np.random.default_rng(11)
with open(path, "wb") as f:
    for y in clip:
        if BD == 8:
            f.write(y.tobytes()); f.write(np.full((H // 2) * (W // 2) * 2, 128, np.uint8).tobytes())
        else:
            f.write((y.astype(np.uint16) * 4 + rng.integers(0, 4, size=y.shape, dtype=np.uint16)).astype("<u2").tobytes())
            f.write(np.full((H // 2) * (W // 2) * 2, 512, "<u2").tobytes())
for lp in LPS:
    for name, asm, env in (("c", "c", {}), ("batched_me", "hip", {"SVTAV1_HIP_LIB": lib, "SVTAV1_HIP_TIERB_ME": "1", "SVTAV1_HIP_ONLY": "__none__"}),
                           ("batched_me_tf", "hip", {"SVTAV1_HIP_LIB": lib, "SVTAV1_HIP_TIERB_ME": "1", "SVTAV1_HIP_TIERB_TF": "1", "SVTAV1_HIP_ONLY": "__none__"}),
                           ("batched_me_tf_tpl", "hip", {"SVTAV1_HIP_LIB": lib, "SVTAV1_HIP_TIERB_ME": "1", "SVTAV1_HIP_TIERB_TF": "1",
                                                         "SVTAV1_HIP_TIERB_TPL": "1", "SVTAV1_HIP_ONLY": "__none__"})):
        t = time.time()
        r = subprocess.run([app, "-i", path, "-w", str(W), "-h", str(H), "--fps", "30", "-n", str(N), "--preset", PRESET, "--input-depth", str(BD), "--lp", str(lp), "--asm", asm,
                            "-b", os.path.join(tmp, name + ".ivf")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, **env))
        m = re.search(r"Average Speed:\s+([0-9.]+) fps", r.stdout)
        g = re.findall(r"svt_hip_bind_(?:me|tf|tpl): \d+ pictures", r.stdout)
        print("preset", PRESET, "lp", lp, name, m.group(1) if m else r.stdout[-300:], "wall", round(time.time() - t, 1), " | ".join(g))
    ref = open(os.path.join(tmp, "c.ivf"), "rb").read()
    print("identical", [open(os.path.join(tmp, n + ".ivf"), "rb").read() == ref for n in ("batched_me", "batched_me_tf", "batched_me_tf_tpl")])
