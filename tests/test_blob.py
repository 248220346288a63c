"""On-disk descriptor blob (sfm_amd/blob.py): round trips and refusal of
malformed files."""
import os

import numpy as np
import pytest

from sfm_amd import blob


@pytest.mark.parametrize("dtype", [np.float32, np.uint8])
@pytest.mark.parametrize("with_kp", [True, False])
def test_blob_round_trip(tmp_path, dtype, with_kp):
    rng = np.random.default_rng(3)
    d = rng.integers(0, 256, (37, 128)).astype(dtype)
    kp = rng.uniform(0, 4000, (37, 2)).astype(np.float32) if with_kp else None
    K = np.array([[2900.0, 0, 1500], [0, 2880.0, 1000], [0, 0, 1]])
    p = str(tmp_path / "a.sfmd")
    blob.write_blob(p, d, kp, K)
    for mm in (True, False):
        r = blob.read_blob(p, mmap=mm)
        assert r["descriptors"].dtype == dtype and (np.asarray(r["descriptors"]) == d).all()
        assert (r["keypoints"] is None) == (kp is None)
        if kp is not None:
            assert (np.asarray(r["keypoints"]) == kp).all()
        assert (r["K"] == K).all()


def test_blob_empty_and_no_camera(tmp_path):
    p = str(tmp_path / "e.sfmd")
    blob.write_blob(p, np.zeros((0, 128), np.uint8), np.zeros((0, 2), np.float32))
    r = blob.read_blob(p)
    assert r["descriptors"].shape == (0, 128) and r["keypoints"].shape == (0, 2) and r["K"] is None


def test_blob_refuses_malformed(tmp_path):
    p = str(tmp_path / "b.sfmd")
    blob.write_blob(p, np.zeros((5, 128), np.float32))
    raw = open(p, "rb").read()
    open(p, "wb").write(raw[:-1])
    with pytest.raises(ValueError):
        blob.read_blob(p)
    open(p, "wb").write(b"NOTABLOB" + raw[8:])
    with pytest.raises(ValueError):
        blob.read_blob(p)
    open(p, "wb").write(raw[:10])
    with pytest.raises(ValueError):
        blob.read_blob(p)
    with pytest.raises(TypeError):
        blob.write_blob(p, np.zeros((5, 128), np.float64))
    with pytest.raises(ValueError):
        blob.write_blob(p, np.zeros((5, 128), np.uint8), np.zeros((4, 2), np.float32))
