"""Build libmi_nerf.so with hipcc for gfx950 (cross-compiles without a GPU).

    python -m nerf_pytorch_paeng_amd.build [--force]          the shipped library (clean build: ~1 min 20 s on 8 cores)
    python -m nerf_pytorch_paeng_amd.build --variant TAG -DFOO -DBAR=1     an A/B variant (tools/ab_probe.py)
    python -m nerf_pytorch_paeng_amd.build --diag             the -DMN_DIAG variant (s_memtime stamps; never shipped or timed)
    python -m nerf_pytorch_paeng_amd.build --clean            remove every object and every variant

What lands where:
  nerf_pytorch_paeng_amd/libmi_nerf.so (+ .stamp)   the ONE artefact inside the package: git-ignored, NOT gpurun-ignored, so it travels
                                                    with the repository snapshot to the GPU box.  The stamp is a hash of every source,
                                                    header and flag: the library is up to date iff the stamp matches (no mtimes, no
                                                    objects needed -- the GPU box gets neither).
  build_scratch/obj/                                objects of the shipped library (cache; tests/test_packing_cpu.py disassembles them)
  build_scratch/obj_TAG/, build_scratch/libmi_nerf_TAG.so    variants.  build_scratch/ is git-ignored AND gpurun-ignored: a variant is
                                                    built where it is used (tools/ab_probe.py builds the ones it is asked for on the box).
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
SCRATCH = os.path.join(ROOT, "build_scratch")
LIB = os.path.join(HERE, "libmi_nerf.so")
STAMP = LIB + ".stamp"
SOURCES = ["api.hip", "stages.hip", "mlp_fp32.hip", "mlp_fp32_wide.hip", "mlp_bf16.hip", "mlp_f16s.hip", "mlp_f16s_stash.hip", "dgrad_f16s.hip", "mlp_train.hip", "frames.hip", "comm.hip", "pack.cpp"]
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


def _headers():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(INCLUDE, "mi_nerf.h")]


def _digest(paths, extra=()) -> str:
    h = hashlib.sha256()
    for p in paths:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as fh:
            h.update(fh.read())
    h.update(repr(list(extra)).encode())
    return h.hexdigest()


def source_stamp(defines=()) -> str:
    """Hash of everything the library is made of: sources, headers, flags (and a variant's -D list)."""
    return _digest([os.path.join(CSRC, s) for s in SOURCES] + _headers(), [FLAGS, sorted(FILE_FLAGS.items()), list(defines)])


def object_dir(tag: str = "") -> str:
    return os.path.join(SCRATCH, "obj" + ("_" + tag if tag else ""))


def _compile(src: str, force: bool, extra=(), tag: str = "") -> str:
    """One translation unit -> build_scratch/obj[_TAG]/SRC.o, skipped when the object's own stamp (source + headers + flags) matches."""
    bdir = object_dir(tag)
    os.makedirs(bdir, exist_ok=True)
    obj = os.path.join(bdir, src + ".o")
    spath = os.path.join(CSRC, src)
    cmd_flags = [*FLAGS, *FILE_FLAGS.get(src, []), *extra]
    want = _digest([spath] + _headers(), cmd_flags)
    stamp = obj + ".stamp"
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == want:
        return obj
    r = subprocess.run([_hipcc(), *cmd_flags, "-x", "hip", "-c", spath, "-o", obj], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    with open(stamp, "w") as fh:
        fh.write(want)
    return obj


def ensure_object(src: str) -> str:
    """The shipped library's object of one source (compiled on demand): what the object checks of tests/test_packing_cpu.py read."""
    return _compile(src, False)


def _link(objs, lib: str) -> None:
    r = subprocess.run([_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", lib], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")


def build_library(force: bool = False, verbose: bool = False) -> str:
    want = source_stamp()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read() == want:
        if verbose:
            print(f"up to date: {LIB} ({os.path.getsize(LIB) / 1024:.0f} KiB, stamp {want[:16]})")
        return LIB
    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), SOURCES))
    _link(objs, LIB)
    with open(STAMP, "w") as fh:
        fh.write(want)
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) / 1024:.0f} KiB, stamp {want[:16]})")
    return LIB


def variant_path(tag: str) -> str:
    return os.path.join(SCRATCH, f"libmi_nerf_{tag}.so")


def build_variant(tag: str, defines) -> str:
    """A/B variant build_scratch/libmi_nerf_TAG.so built with extra -D flags (tools/ab_probe.py times it against the shipped one)."""
    lib = variant_path(tag)
    want = source_stamp(defines)
    if os.path.exists(lib) and os.path.exists(lib + ".stamp") and open(lib + ".stamp").read() == want:
        return lib
    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(lambda s: _compile(s, False, tuple(defines), tag), SOURCES))
    _link(objs, lib)
    with open(lib + ".stamp", "w") as fh:
        fh.write(want)
    return lib


def build_diag_library() -> str:
    """Diagnostic variant (-DMN_DIAG: s_memtime stamps per kernel segment).  Never shipped or timed."""
    return build_variant("diag", ["-DMN_DIAG"])


def clean() -> None:
    """Remove every object and variant; the shipped library stays."""
    shutil.rmtree(SCRATCH, ignore_errors=True)
    for f in os.listdir(HERE):                             # pre-round-4 layouts
        if f.startswith("libmi_nerf_") and f.endswith(".so"):
            os.remove(os.path.join(HERE, f))
    for d in os.listdir(CSRC):
        if d == "build" or d.startswith("build_"):
            shutil.rmtree(os.path.join(CSRC, d), ignore_errors=True)


if __name__ == "__main__":
    if "--clean" in sys.argv:
        clean()
    elif "--diag" in sys.argv:
        print(build_diag_library())
    elif "--variant" in sys.argv:                  # python -m ...build --variant TAG -DFOO -DBAR=1
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], [a for a in sys.argv[i + 2:] if a.startswith("-D")]))
    else:
        print(build_library(force="--force" in sys.argv, verbose=True))
