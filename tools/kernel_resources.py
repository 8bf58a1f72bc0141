#!/usr/bin/env python3
"""Resource usage of the kernels of one .hip file (device assembly metadata): VGPRs, spilled VGPRs, scratch bytes per lane, LDS.
    python tools/kernel_resources.py me_frame.hip [name pattern] [extra hipcc flags...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "svt-av1-mod-by-patman_amd", "csrc")


def main():
    src = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else ""
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "--cuda-device-only", "-S",
           "-I" + os.path.join(ROOT, "include"), os.path.join(CSRC, src), "-o", "-"] + sys.argv[3:]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, cwd=CSRC).stdout
    meta = out[out.rfind("amdhsa.kernels:"):]
    for blk in re.split(r"\n  - \.agpr_count", meta)[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]  # noqa: E731
        name = g("name")
        dem = subprocess.run(["c++filt", name], stdout=subprocess.PIPE, text=True).stdout.strip()
        if pat not in dem:
            continue
        print(f"{dem[:64]:64s} vgpr {g('vgpr_count'):>4s} spill {g('vgpr_spill_count'):>3s} scratch {g('private_segment_fixed_size'):>4s} B  "
              f"lds {g('group_segment_fixed_size'):>6s} B  sgpr {g('sgpr_count')}")


if __name__ == "__main__":
    main()
