import sys, os, tempfile, re
sys.path.insert(0, "tests"); sys.path.insert(0, "svt-av1-mod-by-patman_amd")
import e2e_cases as E
case = sys.argv[1]
g = E.golden()[case]
for name, env in (("tpl", {"SVTAV1_HIP_TIERB_TPL": "1"}), ("tf", {"SVTAV1_HIP_TIERB_TF": "1"}), ("me", {"SVTAV1_HIP_TIERB_ME": "1"})):
    env = dict(env, SVTAV1_HIP_ONLY="__none__")
    with tempfile.TemporaryDirectory() as d:
        md5, log = E.encode(case, d, "hip", lp=3, env_extra=env)
    print(name, md5["ivf"] == g["ivf"], md5["recon"] == g["recon"], [l for l in log.splitlines() if "svt_hip_bind" in l][-4:])
