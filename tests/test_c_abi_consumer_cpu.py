"""include/mi_nerf.h from plain C: tests/c_abi/consumer.c is compiled with gcc as C99 (-Wall -Werror), linked against libmi_nerf.so and run --
the boundary is a C ABI in fact, not only by `extern "C"`: no C++ in the header, every declared entry resolvable by a C
linker, argument checks answering before any GPU call (no GPU here)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("std", ["c99", "c11"])          # c11: mi_nerf_render_cfg keeps the ABI 3 member name `use_bf16` as an alias of `mode`
def test_header_compiles_as_c99_and_the_library_links_and_answers(tmp_path, std):
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("gcc not found")
    from nerf_pytorch_paeng_amd import _lib
    _lib.lib()                                           # builds / checks the library
    pkg = os.path.dirname(_lib.LIB_PATH)
    exe = str(tmp_path / "consumer")
    r = subprocess.run([gcc, f"-std={std}", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c_abi", "consumer.c"),
                        "-L", pkg, "-lmi_nerf", f"-Wl,-rpath,{pkg}", "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # every entry the header declares is in the library's dynamic symbol table (a C linker needs nothing else)
    syms = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    for name in _lib.SIGNATURES:
        assert f" T {name}\n" in syms, name
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "c_abi consumer ok: ABI 4" in run.stdout


def test_every_entry_point_is_mapped_to_the_reference_in_integration_md():
    """INTEGRATION.md section 3 maps each C entry point to the reference code it replaces (or says that it replaces nothing): every function the
    header declares appears there by name."""
    import re
    with open(os.path.join(ROOT, "include", "mi_nerf.h")) as f:
        names = sorted(set(re.findall(r"\b(mi_nerf_[a-z0-9_]+)\s*\(", f.read())))
    with open(os.path.join(ROOT, "INTEGRATION.md")) as f:
        doc = f.read()
    assert len(names) == 68 and [n for n in names if n not in doc] == []
