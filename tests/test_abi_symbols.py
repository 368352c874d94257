"""CPU-only: the C-ABI library loads and exports every symbol include/nvbio_amd.h declares;
compute entry points fail loudly (no CPU fallback) when there is no GPU."""
import ctypes
import re

import numpy as np
import pytest

import __graft_entry__ as ge


def _declared():
    amd = ge.load_package()
    txt = open(amd.HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nvbio_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported():
    amd = ge.load_package()
    L = ctypes.CDLL(amd.LIB_PATH)
    names = _declared()
    assert len(names) >= 18
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_version_and_error_string():
    amd = ge.load_package()
    assert amd.lib().nvbio_amd_version() == 100
    assert isinstance(amd.lib().nvbio_amd_last_error(), bytes)


def test_no_cpu_fallback():
    """without a GPU every compute call must raise, never silently compute on the host"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    amd = ge.load_package()
    v = amd._View()
    v.length, v.primary = 64, 1
    for i, x in enumerate((0, 16, 32, 48, 64)):
        v.L2[i] = x
    buf = np.zeros(64, dtype=np.uint32)
    addr = buf.ctypes.data
    addr += (32 - addr % 32) % 32
    v.bwt_occ_dev, v.bwt_occ_words = addr, 8
    h = ctypes.c_void_p()
    st = amd.lib().nvbio_fm_index_create(ctypes.byref(v), 0, 0, None, ctypes.byref(h))
    assert st == 5, st                                  # NVBIO_ERR_NO_DEVICE
    assert b"no CPU fallback" in amd.lib().nvbio_amd_last_error() or b"device" in amd.lib().nvbio_amd_last_error()


def test_product_does_not_import_the_oracle():
    import os
    amd = ge.load_package()
    pkg = os.path.dirname(amd.__file__)
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "liboracle" not in src and "nvbio_oracle" not in src, f
