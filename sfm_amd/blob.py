"""On-disk descriptor/keypoint blob: what the step before the matching path
(SIFT extraction, main.cpp:142-145) leaves for it when the two run apart.

One image per file, little endian:
    bytes  0..7    magic  b"SFMDESC1"
    int32  n       rows (keypoints)
    int32  dim     descriptor length (128)
    int32  dtype   0 = float32 (CV_32F, as cv::SIFT emits), 1 = uint8 (CV_8U)
    int32  flags   bit 0: keypoints present
    float64[9]     camera matrix K, row-major (zeros when unknown)
    float32[n][2]  keypoint coordinates (cv::KeyPoint::pt), when present
    dtype[n][dim]  descriptors, densely packed rows
Nothing is executed on load: the reader only checks sizes and maps arrays.
"""
import os
import struct

import numpy as np

MAGIC = b"SFMDESC1"
_HEADER = struct.Struct("<8s4i9d")


def write_blob(path, descriptors, keypoints_xy=None, K=None):
    d = np.ascontiguousarray(descriptors)
    if d.ndim != 2:
        raise ValueError("descriptors must be [n, dim]")
    if d.dtype == np.float32:
        dtype = 0
    elif d.dtype == np.uint8:
        dtype = 1
    else:
        raise TypeError("descriptors must be float32 or uint8")
    n, dim = d.shape
    kp = None
    if keypoints_xy is not None:
        kp = np.ascontiguousarray(keypoints_xy, dtype="<f4").reshape(-1, 2)
        if kp.shape[0] != n:
            raise ValueError("keypoints and descriptors differ in length")
    Kc = np.zeros(9) if K is None else np.ascontiguousarray(K, dtype=np.float64).reshape(9)
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(_HEADER.pack(MAGIC, n, dim, dtype, 1 if kp is not None else 0, *Kc.tolist()))
        if kp is not None:
            f.write(kp.tobytes())
        f.write(d.astype("<f4" if dtype == 0 else np.uint8, copy=False).tobytes())
    os.replace(tmp, path)


def read_blob(path, mmap=True):
    """Returns dict(descriptors, keypoints (or None), K (3x3 or None))."""
    size = os.path.getsize(path)
    with open(path, "rb") as f:
        head = f.read(_HEADER.size)
    if len(head) != _HEADER.size:
        raise ValueError("%s: truncated header" % path)
    magic, n, dim, dtype, flags, *Kv = _HEADER.unpack(head)
    if magic != MAGIC:
        raise ValueError("%s: not a descriptor blob" % path)
    if n < 0 or dim <= 0 or dtype not in (0, 1) or flags & ~1:
        raise ValueError("%s: bad header" % path)
    item = 4 if dtype == 0 else 1
    kp_bytes = n * 8 if flags & 1 else 0
    want = _HEADER.size + kp_bytes + n * dim * item
    if size != want:
        raise ValueError("%s: %d bytes, header says %d" % (path, size, want))
    load = (lambda dt, off, shape: np.memmap(path, dtype=dt, mode="r", offset=off, shape=shape)) if mmap and n > 0 \
        else (lambda dt, off, shape: np.fromfile(path, dtype=dt, offset=off,
                                                 count=int(np.prod(shape))).reshape(shape))
    kp = load("<f4", _HEADER.size, (n, 2)) if flags & 1 else None
    desc = load("<f4" if dtype == 0 else np.uint8, _HEADER.size + kp_bytes, (n, dim))
    K = np.array(Kv, dtype=np.float64).reshape(3, 3)
    return dict(descriptors=desc, keypoints=kp, K=K if K.any() else None)
