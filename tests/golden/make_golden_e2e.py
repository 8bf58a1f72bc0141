#!/usr/bin/env python3
"""Generates tests/golden/e2e_md5.json: md5 of the .ivf bitstream and of the reconstruction the REFERENCE encoder writes
with its C kernels (`--asm c`, `--lp 1`) for the clips of tests/e2e_cases.py.  Needs oracle/_ref/e2e/SvtAv1EncApp
(`make -C oracle ref e2e`, this container only).  The encoder is deterministic across --lp; this script checks
that too (--lp 4 gives the same md5) before writing the file."""
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (os.path.join(ROOT, "tests"), os.path.join(ROOT, "svt-av1-mod-by-patman_amd")):
    sys.path.insert(0, p)
import e2e_cases as E  # noqa: E402


def main():
    out = {}
    with tempfile.TemporaryDirectory(dir=os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None) as d:
        for case in E.ALL_CASES:
            m1, _ = E.encode(case, d, "c", lp=1)
            m4, _ = E.encode(case, d, "c", lp=4)
            assert m1 == m4, f"{case}: --lp 1 and --lp 4 differ"
            out[case] = dict(zip(("width", "height", "frames", "bit_depth", "preset"), E.ALL_CASES[case]), **m1)
            print(case, m1)
    with open(E.GOLDEN, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
