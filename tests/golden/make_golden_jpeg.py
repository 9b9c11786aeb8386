#!/usr/bin/env python3
"""Golden vectors for the JPEG ingest stage: JPEG streams and the BGR pixels that
libjpeg-turbo -- the decoder behind the reference's cv2.imdecode(data, IMREAD_COLOR)
(ref: src/duckietown/include/duckietown_utils/jpg.py:21-31) -- decodes them to.

Run in the build container (Pillow 12.2 bundles libjpeg-turbo and decodes with the library's default
settings, the same ones OpenCV's JPEG reader leaves in place: accurate integer IDCT, fancy
upsampling).  Inputs are synthetic (seeded); nothing is taken from /root/reference.

    python tests/golden/make_golden_jpeg.py      -> tests/golden/jpeg_vectors.npz
"""
import hashlib
import io
import os
import sys

import numpy as np
from PIL import Image, features

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from lane_slam_amd import synth  # noqa: E402


def encode(bgr, **kw):
    im = Image.fromarray(bgr[..., ::-1].copy()) if bgr.ndim == 3 else Image.fromarray(bgr)
    b = io.BytesIO()
    im.save(b, "JPEG", **kw)
    return b.getvalue()


def decode(data):
    im = Image.open(io.BytesIO(data))
    return np.ascontiguousarray(np.asarray(im.convert("RGB"))[..., ::-1])


def main():
    assert features.check_feature("libjpeg_turbo")
    rng = np.random.default_rng(2024)
    noise = rng.integers(0, 256, (53, 37, 3), dtype=np.uint8)
    smooth = np.clip(np.add.outer(np.arange(48) * 4, np.arange(64) * 3)[..., None] + np.array([0, 40, 90]), 0, 255).astype(np.uint8)
    lane = synth.make_frame(5)[200:296, 256:400]                  # 96 x 144 crop of a synthetic lane frame
    out = {}
    names = []

    def add(name, data, keep_pixels=True):
        ref = decode(data)
        out["jpeg_" + name] = np.frombuffer(data, np.uint8)
        if keep_pixels:
            out["bgr_" + name] = ref
        out["sha_" + name] = np.frombuffer(hashlib.sha256(ref.tobytes()).digest(), np.uint8)
        out["shape_" + name] = np.array(ref.shape, np.int32)
        names.append(name)

    for iname, img in (("noise", noise), ("smooth", smooth), ("lane", lane)):
        for q in (30, 75, 95):
            for sub, sname in ((0, "444"), (1, "422"), (2, "420")):
                add("%s_q%d_%s" % (iname, q, sname), encode(img, quality=q, subsampling=sub))
    add("noise_opt_420", encode(noise, quality=75, subsampling=2, optimize=True))
    add("lane_rst_420", encode(lane, quality=75, subsampling=2, restart_marker_blocks=5))
    add("lane_rst_444", encode(lane, quality=75, subsampling=0, restart_marker_rows=1))
    add("gray", encode(lane[..., 1].copy(), quality=80))
    add("lane_rgb_444", encode(lane, quality=80, keep_rgb=True))         # Adobe marker, transform 0: components are R, G, B
    add("tiny_3x5_420", encode(noise[:3, :5].copy(), quality=75, subsampling=2))
    add("tiny_1x1_420", encode(noise[:1, :1].copy(), quality=75, subsampling=2))
    add("narrow_9x4_422", encode(noise[:9, :4].copy(), quality=75, subsampling=1))
    # one full camera-sized frame: stream + checksum + a strided sample of the pixels
    full = encode(synth.make_frame(7), quality=75, subsampling=2)
    add("full_640x480_420", full, keep_pixels=False)
    out["sample_full_640x480_420"] = decode(full)[::7, ::5].copy()
    # streams the decoder must refuse (the reference would get None -> ValueError -> frame dropped)
    out["jpeg_progressive"] = np.frombuffer(encode(lane, quality=75, progressive=True), np.uint8)
    out["jpeg_truncated"] = out["jpeg_lane_q75_420"][: len(out["jpeg_lane_q75_420"]) // 2].copy()
    out["names"] = np.array(names)
    path = os.path.join(HERE, "jpeg_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(names), "streams")


if __name__ == "__main__":
    main()
