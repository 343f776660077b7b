"""The on-disk formats between inference and depth-map filtering, written and read byte-compatibly
with the reference so that either side can be swapped on its own:

    depth      ``depth_est/<view>.pfm``      greyscale PFM, float32, bottom row first, negative scale = little endian
                                             (datasets/data_io.py:7-37 read_pfm, :40-67 save_pfm)
    confidence ``confidence/<view>.npy``     uint8 = floor(photometric_confidence * 255)          (test.py:281-286)
    camera     ``cams/<view>_cam.txt``       "extrinsic" 4x4, blank line, "intrinsic" 3x3, blank line, the 4 values of the
                                             intrinsic slot's last row (depth range)             (test.py:149-166, :102-112)

Host-side numpy only; nothing here touches the GPU.
"""
from __future__ import annotations

import re
import sys
from typing import Tuple

import numpy as np


def read_pfm(filename: str) -> Tuple[np.ndarray, float]:
    """-> (image [H,W] or [H,W,3] float32 with the TOP row first, scale)."""
    with open(filename, "rb") as f:
        header = f.readline().decode("utf-8").rstrip()
        if header not in ("PF", "Pf"):
            raise Exception("Not a PFM file.")
        color = header == "PF"
        m = re.match(r"^(\d+)\s(\d+)\s$", f.readline().decode("utf-8"))
        if not m:
            raise Exception("Malformed PFM header.")
        width, height = int(m.group(1)), int(m.group(2))
        scale = float(f.readline().rstrip())
        endian = "<" if scale < 0 else ">"
        data = np.fromfile(f, endian + "f")
    shape = (height, width, 3) if color else (height, width)
    return np.flipud(np.reshape(data, shape)), abs(scale)


def save_pfm(filename: str, image: np.ndarray, scale: float = 1) -> None:
    if image.dtype.name != "float32":
        raise Exception("Image dtype must be float32.")
    if image.ndim == 3 and image.shape[2] == 3:
        color = True
    elif image.ndim == 2 or (image.ndim == 3 and image.shape[2] == 1):
        color = False
    else:
        raise Exception("Image must have H x W x 3, H x W x 1 or H x W dimensions.")
    image = np.flipud(image)
    little = image.dtype.byteorder == "<" or (image.dtype.byteorder == "=" and sys.byteorder == "little")
    with open(filename, "wb") as f:
        f.write(b"PF\n" if color else b"Pf\n")
        f.write(("%d %d\n" % (image.shape[1], image.shape[0])).encode("utf-8"))
        f.write(("%f\n" % (-scale if little else scale)).encode("utf-8"))
        image.tofile(f)


def save_confidence(filename: str, photometric_confidence: np.ndarray) -> None:
    np.save(filename, (photometric_confidence * 255).astype(np.uint8))


def load_confidence(filename: str) -> np.ndarray:
    return np.load(filename)


def write_cam(filename: str, cam: np.ndarray) -> None:
    """cam [2,4,4]: 0 = extrinsic, 1 = intrinsic in the top-left 3x3 with the depth range in its last row."""
    with open(filename, "w") as f:
        f.write("extrinsic\n")
        for i in range(4):
            f.write("".join(str(cam[0][i][j]) + " " for j in range(4)) + "\n")
        f.write("\nintrinsic\n")
        for i in range(3):
            f.write("".join(str(cam[1][i][j]) + " " for j in range(3)) + "\n")
        f.write("\n" + " ".join(str(cam[1][3][j]) for j in range(4)) + "\n")


def read_camera_parameters(filename: str) -> Tuple[np.ndarray, np.ndarray]:
    """-> (intrinsics [3,3], extrinsics [4,4]) float32."""
    with open(filename) as f:
        lines = [line.rstrip() for line in f.readlines()]
    extrinsics = np.array(" ".join(lines[1:5]).split(), dtype=np.float32).reshape(4, 4)
    intrinsics = np.array(" ".join(lines[7:10]).split(), dtype=np.float32).reshape(3, 3)
    return intrinsics, extrinsics
