"""CPU tests of the drop-in boundary: libmgrt.so loads without a GPU and exports every entry point
include/mgrt.h declares; argument errors are reported through status codes."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from multigrid_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mgrt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    macro = re.search(r"#define MGRT_DECLARE_KERNELS\(R, S\)(.*?)\n\n", text, re.S).group(1)
    kernels = re.findall(r"\b(mgrt_\w+)##S\(", macro)
    rest = text.replace(macro, "")
    plain = re.findall(r"\b(mgrt_\w+)\(", rest)
    names = {k + s for k in kernels for s in ("_f32", "_f64")} | set(plain)
    return names, kernels


def test_header_and_binding_agree():
    names, kernels = declared_symbols()
    assert len(kernels) == 18
    assert names == set(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names, _ = declared_symbols()
    for n in sorted(names):
        assert hasattr(lib, n), n


def test_errors_are_status_codes_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("covers the no-GPU path")
    lib = _lib.load()
    assert lib.mgrt_init(0) == -4  # MGRT_E_NO_DEVICE
    assert b"no HIP device" in lib.mgrt_last_error()
    ptr = C.c_void_p()
    assert lib.mgrt_alloc(C.byref(ptr), 16) == -3  # MGRT_E_NOT_INIT
    with pytest.raises(_lib.MgrtError):
        _lib.check(lib.mgrt_sync(None))


def test_box_cells_host_matches_oracle(mgo):
    for shape, faces in (((5, 6, 7), (2, 3, 3, 2, 2, 3)), ((9, 8), (3, 2, 2, 3, 2, 2)), ((4, 4, 4), (2,) * 6)):
        assert np.array_equal(_lib.box_cells_host(shape, faces), mgo.box_cells(shape, faces))
