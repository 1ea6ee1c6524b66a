"""Build libpffft_hip.so (hipcc, gfx950) in-tree.  `python -m pffft_amd.build` or
__graft_entry__.build().  The .so is git-ignored but travels to the GPU box with the snapshot."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpffft_hip.so")
SOURCES = ["pffft_hip.hip", "dma_tu.hip", "conv_tu.hip", "tile_tu.hip", "tile_real_tu.hip", "tileg_tu.hip", "one_tu.hip"] + [f"one_k{f}_tu.hip" for f in (0, 2, 4, 5, 8, 10, 12, 13)] + [f"tile_mr{r}_tu.hip" for r in (3, 5, 9, 15, 25, 27, 45)] + [f"stock_ct_{t}_{h}_gen.hip" for t in ("f32c", "f32r", "f64c", "f64r") for h in "ab"]
OBJDIR = os.path.join(HERE, "..", "build", "obj")
# the PFDSP mixers are their own library, like the reference's PFDSP target (CMakeLists.txt:206)
DSP_LIB = os.path.join(HERE, "libpfdsp_hip.so")
DSP_SRC = os.path.join(CSRC, "pfdsp_hip.hip")
DSP_DEPS = [DSP_SRC, os.path.join(CSRC, "pfdsp_mix.h"), os.path.join(HERE, "..", "include", "pfdsp_hip.h"),
            os.path.join(CSRC, "exports_dsp.map")]


def _deps():
    out = [os.path.join(CSRC, f) for f in os.listdir(CSRC)
           if f.endswith((".h", ".hip", ".map")) and f not in ("pfdsp_hip.hip", "exports_dsp.map")]
    out.append(os.path.join(HERE, "..", "include", "pffft_hip.h"))
    return out


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in _deps())


# The device code of ~2 500 kernels is 90 % of the library: compressed offload bundles (the HIP runtime inflates a bundle when it loads the
# module) take libpffft_hip.so from 27 MB to a few MB.  PFFFT_HIP_NO_COMPRESS=1 builds without (A/B of the load time).
COMPRESS = [] if os.environ.get("PFFFT_HIP_NO_COMPRESS") == "1" else ["--offload-compress"]


def _build_dsp(force: bool, verbose: bool) -> None:
    if not force and os.path.exists(DSP_LIB) and all(os.path.getmtime(f) <= os.path.getmtime(DSP_LIB) for f in DSP_DEPS):
        return
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off"] + COMPRESS + [
           "-shared", "-Wl,--version-script=" + os.path.join(CSRC, "exports_dsp.map"), "-o", DSP_LIB, DSP_SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=CSRC)


def _obj_stale(obj: str) -> bool:
    """True when `obj` is missing or older than any file its depfile (hipcc -MD) lists."""
    dep = obj[:-2] + ".d"
    if not (os.path.exists(obj) and os.path.exists(dep)):
        return True
    t = os.path.getmtime(obj)
    txt = open(dep).read().replace("\\\n", " ")
    files = txt.split(":", 1)[1].split() if ":" in txt else []
    return any((not os.path.exists(f)) or os.path.getmtime(f) > t for f in files)


def build(force: bool = False, verbose: bool = False) -> str:
    _build_dsp(force, verbose)
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=off",
             "-Wno-pass-failed",   # "loop not unrolled" remarks of fully unrollable radix loops hid real diagnostics
             "-mllvm", "-amdgpu-atomic-optimizer-strategy=None"] + COMPRESS
    # PFFFT_HIP_VARIANTS=1: development build - BOTH variants (deposit / direct first stage) of every Stockham plan are
    # instantiated so that the A/B selectors 54 / 55 and tools/tune_stock_df.py can compare them (twice the generated kernels,
    # 10 instead of 5 minutes).  The product build instantiates the adopted one only (tools/gen_stock_plans.hip).
    if os.environ.get("PFFFT_HIP_VARIANTS") == "1":
        flags.append("-DPFFFT_HIP_VARIANTS")
    os.makedirs(OBJDIR, exist_ok=True)
    # one hipcc per translation unit, in parallel (the generated Stockham units hold ~600 kernel instantiations);
    # a unit is recompiled only when one of the files in its depfile changed
    procs, objs = [], []
    for src in SOURCES:
        obj = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        objs.append(obj)
        if not force and not _obj_stale(obj):
            continue
        cmd = [hipcc] + flags + ["-MD", "-MF", obj[:-2] + ".d", "-c", "-o", obj, os.path.join(CSRC, src)]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    # only the C ABI of include/pffft_hip.h becomes a dynamic symbol (exports.map: the kernels' host stubs stay local)
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"),
            "-o", LIB] + objs
    if verbose:
        print(" ".join(link))
    subprocess.run(link, check=True, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
