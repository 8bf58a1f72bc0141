#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (run by `make -C oracle ref`, only where /root/reference exists).

Builds oracle/_ref/libsvtsimd.a = every x86 INTRINSICS file of the reference (Source/Lib/ASM_{SSE2,SSSE3,SSE4_1,AVX2}/*.c:
plain C with <immintrin.h>) that compiles with gcc -mavx2 and links without the NASM objects (no nasm / yasm in this image,
SURVEY.md 8c), compiled from the sources where they lie, and oracle/_ref/simd_table.inc = for every RTCD pointer, the
function svt_aom_setup_rtcd_internal / svt_aom_setup_common_rtcd_internal would select with the AVX2 flag set
(aom_dsp_rtcd.c:187 ff., common_dsp_rtcd.c:447 ff.: the SET_<ISA list>(ptr, c, impl...) lines), restricted to the functions
that made it into the archive.  ref_harness_simd.c applies that table: bench.py's cpu_baseline then times the reference's
own AVX2 path (kernels whose AVX2 form is NASM — e.g. the dav1d inverse transforms — fall back to the best intrinsics
form present, SSE4.1 or C, and the table records which).
"""
import concurrent.futures as cf
import glob
import os
import re
import subprocess
import sys

REF = os.environ.get("REF", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
OBJ = os.path.join(OUT, "obj_simd")
CC = os.environ.get("CC", "gcc")
LIB = os.path.join(REF, "Source", "Lib")
DIRS = ["ASM_SSE2", "ASM_SSSE3", "ASM_SSE4_1", "ASM_AVX2"]
CFLAGS = ["-O2", "-fPIC", "-ffunction-sections", "-fdata-sections", "-w", "-mavx2", "-msse4.1", "-mssse3", "-DARCH_X86_64=1",
          "-DEXCLUDE_HASH=0", "-DREPRODUCIBLE_BUILDS=0", "-DEN_AVX512_SUPPORT=0",
          f"-I{REF}/Source/API", f"-I{LIB}/Globals", f"-I{LIB}/Codec", f"-I{LIB}/C_DEFAULT", f"-I{REF}/third_party/fastfeat"] + \
         [f"-I{LIB}/{d}" for d in DIRS]
ISA_RANK = {"SSE2": 1, "SSSE3": 2, "SSE41": 3, "AVX2": 4, "AVX512": 9}     # AVX512 is never taken (EN_AVX512_SUPPORT=0)


def compile_one(src):
    obj = os.path.join(OBJ, "simd_" + os.path.basename(src)[:-2] + ".o")
    if os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(src):
        return obj
    r = subprocess.run([CC] + CFLAGS + ["-c", src, "-o", obj], capture_output=True, text=True)
    return obj if r.returncode == 0 else None


def defined(paths):
    out = subprocess.run(["nm", "--defined-only"] + paths, capture_output=True, text=True).stdout
    return {ln.split()[-1] for ln in out.splitlines() if len(ln.split()) == 3 and ln.split()[1] in "TDBRCtdbr"}


def main():
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(s for d in DIRS for s in glob.glob(os.path.join(LIB, d, "*.c")))
    with cf.ThreadPoolExecutor(int(os.environ.get("JOBS", "8"))) as ex:
        objs = [o for o in ex.map(compile_one, srcs) if o]
    base = os.path.join(OUT, "libsvtref.a")
    # drop objects that need NASM symbols: let the linker name them, repeat until the rest links
    # (undefined symbols are reported as warnings: members of the base archive legitimately miss Globals/enc_handle.c)
    for _ in range(12):
        r = subprocess.run([CC, "-shared", "-o", "/dev/null", "-Wl,--no-undefined", "-Wl,--warn-unresolved-symbols",
                            "-Wl,--whole-archive"] + objs + ["-Wl,--no-whole-archive", base, "-lm", "-lpthread"],
                           capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-3000:])
            raise SystemExit("build_simd: trial link failed")
        bad = set(re.findall(r"(\S+simd_\w+\.o): in function", r.stderr))
        if not bad:
            break
        objs = [o for o in objs if o not in bad]
    else:
        raise SystemExit("build_simd: no closed set of intrinsics objects found")
    arch = os.path.join(OUT, "libsvtsimd.a")
    if os.path.exists(arch):
        os.remove(arch)
    subprocess.run(["ar", "rcs", arch] + objs, check=True)
    have = defined(objs)

    # the selection ladder of the reference, evaluated for flags = everything up to AVX2
    rows, seen = [], set()
    for rt in ("aom_dsp_rtcd.c", "common_dsp_rtcd.c"):
        text = open(os.path.join(LIB, "Codec", rt)).read()
        for m in re.finditer(r"\bSET_((?:SSE2|SSSE3|SSE41|AVX2|AVX512)(?:_(?:SSE2|SSSE3|SSE41|AVX2|AVX512))*)\(([^;]*?)\);", text, re.S):
            isas = m.group(1).split("_")
            args = [a.strip() for a in m.group(2).replace("\n", " ").split(",")]
            ptr, impls = args[0], args[2:]
            if len(impls) != len(isas) or ptr in seen:
                continue
            best = None
            for isa, fn in zip(isas, impls):
                if ISA_RANK[isa] <= 4 and fn in have:
                    best = (isa, fn)          # later entries of the ladder are higher ISAs
            want = [i for i in isas if ISA_RANK[i] <= 4][-1] if any(ISA_RANK[i] <= 4 for i in isas) else None
            if best:
                seen.add(ptr)
                rows.append((ptr, best[1], best[0], want))
        # pointers the reference assigns outside the SET_ ladders: `if (flags & HAS_<ISA>) ptr = fn;` (e.g. svt_cdef_filter_block_8xn_16,
        # common_dsp_rtcd.c:800, which svt_cdef_filter_block_avx2 calls: left NULL it crashes every 16-bit CDEF)
        direct = {}
        for m in re.finditer(r"if \(flags & HAS_(SSE2|SSSE3|SSE4_1|AVX2|AVX512F)\)\s+(\w+)\s*=\s*(\w+);", text):
            isa = {"SSE4_1": "SSE41", "AVX512F": "AVX512"}.get(m.group(1), m.group(1))
            if ISA_RANK[isa] <= 4 and m.group(3) in have and m.group(2) not in seen:
                direct[m.group(2)] = (m.group(3), isa)
        for ptr, (fn, isa) in direct.items():
            seen.add(ptr)
            rows.append((ptr, fn, isa, isa))
    with open(os.path.join(OUT, "simd_table.inc"), "w") as f:
        f.write("/* generated by oracle/build_simd.py: {pointer, selected function, its ISA, ISA the reference's full x86 build selects} */\n")
        for ptr, fn, isa, want in rows:
            f.write(f'SIMD_ROW({ptr}, {fn}, "{isa}", "{want}")\n')
    n_full = sum(1 for r in rows if r[2] == r[3])
    print(f"build_simd: {len(objs)} of {len(srcs)} intrinsics files linked; {len(rows)} RTCD pointers get a SIMD function "
          f"({n_full} at the ISA the full x86 build selects)")


if __name__ == "__main__":
    main()
