"""CPU tests of the boundary: the C-ABI library loads, exports every symbol
include/pt_amd.h declares, and fails loudly without a GPU (no CPU fallback)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "pt_amd.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pt_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_all_exported(pt):
    names = declared_symbols()
    assert len(names) >= 20
    lib = ctypes.CDLL(pt.LIB_PATH)
    for name in names:
        assert hasattr(lib, name), "libpt_amd.so does not export %s" % name
    assert set(names) == set(pt.ABI_SYMBOLS)


def test_header_is_plain_c():
    """The boundary is C: the header compiles as C11 with no C++ or torch types."""
    src = '#include "pt_amd.h"\nint main(void){pt_counters c; pt_camera k; (void)c; (void)k; return 0;}\n'
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-fsyntax-only", "-I",
                    os.path.join(ROOT, "include"), "-x", "c", "-"], input=src.encode(), check=True)


def test_struct_layouts_match_header(pt):
    """ctypes/numpy layouts used by the binding equal the C structs' sizes."""
    src = ('#include <stdio.h>\n#include "pt_amd.h"\nint main(void){printf("%zu %zu %zu %zu\\n",'
           'sizeof(pt_surface),sizeof(pt_material),sizeof(pt_camera),sizeof(pt_counters));return 0;}\n')
    exe = os.path.join(ROOT, "tests", ".abi_sizes")
    try:
        subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), "-x", "c", "-", "-o", exe],
                       input=src.encode(), check=True)
        sizes = [int(x) for x in subprocess.check_output([exe]).split()]
    finally:
        if os.path.exists(exe):
            os.remove(exe)
    assert sizes == [pt.SURFACE.itemsize, pt.MATERIAL.itemsize, pt.CAMERA.itemsize,
                     ctypes.sizeof(pt.Counters)]


def test_no_gpu_fails_loudly(pt):
    """Without a GPU pt_create reports an error; nothing renders on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(pt.PtError) as e:
        pt.Context(0)
    assert e.value.code in (-1, -2)
    assert str(e.value)


def test_product_does_not_touch_the_oracle():
    """Nothing under path-tracer_amd/ names the oracle."""
    pkg = os.path.join(ROOT, "path-tracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hpp", ".hip", ".cpp", ".h", "Makefile")):
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "oracle" not in text.lower(), os.path.join(dirpath, fn)
