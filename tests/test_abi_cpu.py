"""CPU tests of the drop-in boundary: the C-ABI library builds, loads and exports every symbol the header declares;
without a GPU the product fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT, load_pkg

capi = load_pkg("capi")


def _header_functions():
    src = open(os.path.join(ROOT, "include", "ls1hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ls1hip_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = capi.load()
    declared = _header_functions()
    assert declared, "no functions parsed from include/ls1hip.h"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in ls1hip.h but not exported by libls1hip.so"
    # the Python binding covers exactly the header
    assert sorted(capi.SYMBOLS) == declared


def test_version_string():
    assert capi.load().ls1hip_version().decode().startswith("ls1hip")


def test_constants_match_header():
    src = open(os.path.join(ROOT, "include", "ls1hip.h")).read()
    assert int(re.search(r"#define LS1HIP_LEAVING_DOUBLES (\d+)", src).group(1)) == capi.LEAVING_DOUBLES
    assert int(re.search(r"#define LS1HIP_HALO_DOUBLES (\d+)", src).group(1)) == capi.HALO_DOUBLES


def test_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = capi.load()
    ctx = C.c_void_p()
    rc = lib.ls1hip_create(0, C.byref(ctx))
    assert rc != 0 and not ctx.value
    assert b"device" in lib.ls1hip_last_error(None).lower()
    engine = load_pkg("engine")
    with pytest.raises(capi.Ls1HipError):
        engine.DeviceEngine(0)


def test_product_never_imports_oracle():
    pkgdir = os.path.join(ROOT, "ls1-mardyn_amd")
    for dirpath, _, files in os.walk(pkgdir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "ls1_oracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, f


def test_shipped_library_is_not_a_timing_variant():
    """timing variants of the kernels (phase-decomposition builds whose forces are wrong by construction) are compiled
    only by tools/ab_variant.sh under -DLS1_BUILD_VARIANT, which marks the library.  The shipped libls1hip.so must not carry the
    marker symbol, its version string must not say "+variant", the production kernel source must hold no mock switch, and the
    regular Makefile must not be able to set one."""
    import subprocess
    lib = capi.load()
    assert b"variant" not in lib.ls1hip_version()
    syms = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "ls1hip_variant_marker" not in syms
    src = open(os.path.join(ROOT, "ls1-mardyn_amd", "csrc", "kernels_force_verlet.hip")).read()
    body = src.split('#error "variant switches need -DLS1_BUILD_VARIANT', 1)[1]
    assert "_MOCK" not in body, "a timing mock crept back into the production kernel (they live in csrc/variants/)"
    mk = open(os.path.join(ROOT, "ls1-mardyn_amd", "Makefile")).read()
    assert "LS1_BUILD_VARIANT" in mk and "-DLS1_BUILD_VARIANT" not in mk
