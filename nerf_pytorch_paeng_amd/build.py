"""Build libmi_nerf.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m nerf_pytorch_paeng_amd.build [--force]

The shared library lands next to this file so that it travels with the repository snapshot to the
GPU box (it is git-ignored, not gpurun-ignored).  Objects are cached under csrc/build/ keyed on
source + header mtimes.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libmi_nerf.so")
SOURCES = ["api.hip", "stages.hip", "mlp_fp32.hip", "mlp_bf16.hip", "mlp_f16s.hip", "mlp_f16s_stash.hip", "dgrad_f16s.hip", "mlp_train.hip", "frames.hip", "pack.cpp"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
         # the MLP kernel's register-resident design needs its k-loops FULLY unrolled (static register indices)
         "-mllvm", "-pragma-unroll-threshold=1000000"]
# mlp_bf16.hip manages the whole AGPR file by hand (explicit a[N] operands in asm statements): hipcc must not park spilled VGPRs there
FILE_FLAGS = {"mlp_bf16.hip": ["-mllvm", "-amdgpu-spill-vgpr-to-agpr=0"], "mlp_f16s.hip": ["-mllvm", "-amdgpu-spill-vgpr-to-agpr=0"],
              "mlp_f16s_stash.hip": ["-mllvm", "-amdgpu-spill-vgpr-to-agpr=0"], "dgrad_f16s.hip": ["-mllvm", "-amdgpu-spill-vgpr-to-agpr=0"]}


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the MI355X path cannot be built (there is no CPU fallback)")
    return exe


def _deps_mtime() -> float:
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(INCLUDE, "mi_nerf.h")]
    return max(os.path.getmtime(h) for h in hs)


def _compile(src: str, force: bool, extra=(), tag: str = "") -> str:
    bdir = os.path.join(CSRC, "build" + tag)
    os.makedirs(bdir, exist_ok=True)
    obj = os.path.join(bdir, src + ".o")
    spath = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(spath)
            and os.path.getmtime(obj) >= _deps_mtime()):
        return obj
    cmd = [_hipcc(), *FLAGS, *FILE_FLAGS.get(src, []), *extra, "-x", "hip", "-c", spath, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj


def build_library(force: bool = False, verbose: bool = False) -> str:
    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), SOURCES))
    if force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"built {LIB} ({os.path.getsize(LIB) / 1024:.0f} KiB)")
    return LIB


def build_variant(tag: str, defines) -> str:
    """A/B variant libmi_nerf_{tag}.so built with extra -D flags (tools/ab_probe.py times it against the shipped one)."""
    lib = os.path.join(HERE, f"libmi_nerf_{tag}.so")
    objs = [_compile(s, False, tuple(defines), "_" + tag) for s in SOURCES]
    r = subprocess.run([_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", lib], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return lib


def build_diag_library() -> str:
    """Diagnostic variant (-DMN_DIAG: s_memtime stamps per kernel segment).  Never shipped or timed."""
    lib = os.path.join(HERE, "libmi_nerf_diag.so")
    objs = [_compile(s, False, ("-DMN_DIAG",), "_diag") for s in SOURCES]
    r = subprocess.run([_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", lib], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return lib


if __name__ == "__main__":
    if "--diag" in sys.argv:
        print(build_diag_library())
    elif "--variant" in sys.argv:                  # python -m ...build --variant TAG -DFOO -DBAR=1
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], [a for a in sys.argv[i + 2:] if a.startswith("-D")]))
    else:
        print(build_library(force="--force" in sys.argv, verbose=True))
