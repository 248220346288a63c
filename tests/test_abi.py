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


def test_headers_are_plain_c_and_a_c_host_links(tmp_path):
    """The boundary is for a compiled host: both headers must be valid C99 and
    C++17, and a C program must link against the library and call it (entry
    points that need no GPU)."""
    import subprocess
    from sfm_amd import _lib
    inc = os.path.join(ROOT, "include")
    src = tmp_path / "host.c"
    src.write_text(
        '#include "sfm_gpu.h"\n#include "sfm_gpu_debug.h"\n#include <stdio.h>\n#include <string.h>\n'
        "int main(void) {\n"
        "    sfm_gpu_dmatch m; sfm_gpu_match_params p; memset(&p, 0, sizeof p); (void)m;\n"
        "    if (sizeof(sfm_gpu_dmatch) != 16) return 2;            /* cv::DMatch layout */\n"
        "    if (!strstr(sfm_gpu_version(), \"gfx950\")) return 3;\n"
        "    if (sfm_gpu_image_packed_bytes(1) != 1024u * (128 + 4 + 4 + 8)) return 4;\n"
        "    if (sfm_gpu_create(0, 0, 0) != SFM_GPU_ERR_ARG) return 5; /* null out pointer */\n"
        "    { int32_t keep[2], n = -1; sfm_gpu_dmatch r[4]; int32_t c[2] = {2, 1};\n"
        "      memset(r, 0, sizeof r); r[0].distance = 1.f; r[1].distance = 2.f; r[2].distance = 1.f; r[3].distance = 1.f;\n"
        "      if (sfm_gpu_ratio_test(r, c, 2, 2, 0.7, keep, &n) != SFM_GPU_OK || n != 1 || keep[0] != 0) return 6; }\n"
        '    puts("c host ok");\n    return 0;\n}\n')
    exe = tmp_path / "host"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", inc, str(src), "-o", str(exe),
                           "-L", libdir, "-lsfm_gpu", "-Wl,-rpath," + libdir])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and "c host ok" in out.stdout, (out.returncode, out.stdout, out.stderr)
    cpp = tmp_path / "host.cpp"
    cpp.write_text('#include "sfm_gpu.h"\n#include "sfm_gpu_debug.h"\nint main() { return sfm_gpu_version() ? 0 : 1; }\n')
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-I", inc, str(cpp)])


def test_opencv_facing_sources_parse_against_declaration_only_headers():
    """include/sfm_gpu_cv.hpp and tools/opencv_crosscheck.cpp need OpenCV, which this
    machine lacks.  tests/mock_opencv holds declarations (no bodies, nothing is
    linked or run) of the few OpenCV names the two files use, enough for
    `g++ -fsyntax-only` to find typos, missing includes and calls of C-ABI
    functions with the wrong arguments.  Syntax only: it proves nothing about
    OpenCV's behaviour."""
    import subprocess
    p = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror",
                        "-I", os.path.join(ROOT, "tests", "mock_opencv"), "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tools", "opencv_crosscheck.cpp")], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-4000:]
