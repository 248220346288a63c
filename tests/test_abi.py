"""The C-ABI library loads and exports every symbol include/*.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    syms = set()
    for h in ("sfm_gpu.h", "sfm_gpu_debug.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        syms.update(re.findall(r"\b(sfm_gpu_[a-z0-9_]+)\s*\(", text))
    return syms


def test_library_exports_declared_symbols():
    from sfm_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build with python -m sfm_amd.build"
    L = ctypes.CDLL(_lib.LIB_PATH)
    syms = _declared_symbols()
    assert len(syms) >= 18
    for s in sorted(syms):
        assert hasattr(L, s), "missing export %s" % s
    L.sfm_gpu_version.restype = ctypes.c_char_p
    assert b"gfx950" in L.sfm_gpu_version()


def test_ratio_test_is_host_side_and_matches_oracle(oracle):
    import numpy as np
    from sfm_amd import matcher
    rng = np.random.default_rng(5)
    m = np.zeros((500, 2), dtype=matcher.DMATCH_DTYPE)
    m["distance"][:, 1] = rng.uniform(100, 400, 500).astype(np.float32)
    m["distance"][:, 0] = (m["distance"][:, 1] * rng.uniform(0.5, 0.9, 500)).astype(np.float32)
    m["distance"][7, 0] = np.float32(0.7) * m["distance"][7, 1]
    c = np.full(500, 2, np.int32)
    c[11] = 1
    keep = matcher.ratio_test(m, c, 0.7)
    om = m.astype(oracle.DMATCH_DTYPE)
    assert keep.tolist() == oracle.ratio_test(om, c, 0.7).tolist()
    assert 11 not in keep.tolist()


def test_no_gpu_context_fails_loudly():
    import pytest
    import torch
    from sfm_amd import matcher
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(matcher.SfmGpuError):
        matcher.Context(0)


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sfm_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "pyoracle" not in text and "sfm_oracle" not in text, f
