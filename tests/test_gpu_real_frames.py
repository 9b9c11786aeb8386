"""Real camera content under test: 28 of the reference's 173 Duckiebot camera frames (tests/golden/real_jpegs.npz:
their JPEG byte streams, spread over the five recording sessions; generator tests/golden/make_golden.py: golden_real_jpegs), the
way the reference meets them -- as CompressedImage payloads (ref: src/line_detector/src/line_detector_node.py:141-155):
    JPEG bytes -> device Huffman + IDCT + colour (lf_jpeg_decode_batch_gpu)  ==  the pinned JPEG oracle (libjpeg-turbo vectors)
    -> the whole front end at both geometries                                  ==  the oracle, every field
    -> EDLines over three octaves + LBD (lf_keylines_batch)                    ==  the oracle, every KeyLine field
Real frames are a different workload from the synthetic lane frames (texture: 3x the segments, problems of 9 - 16 k defined
pixels, the BIG problem code of k_lsd_grow, 24 k-entry labelling tables), which three frames could not cover."""
import os

import numpy as np
import pytest

from lane_slam_amd import FrontEnd, default_config
from oracle import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def real(golden_dir):
    z = np.load(os.path.join(golden_dir, "real_jpegs.npz"))
    streams = [bytes(z["jpeg%02d" % k]) for k in range(len(z["names"]))]
    frames = np.stack([O.jpeg_decode(s) for s in streams])              # the pinned oracle decoder
    assert frames.shape == (28, 480, 640, 3)
    return streams, frames


def test_device_jpeg_decode_of_the_camera_streams(real):
    streams, frames = real
    fe = FrontEnd(default_config("fullres"), max_frames=len(streams), max_lines_per_color=64)
    for entropy in ("gpu", "host"):
        got, status = fe.decode_jpeg_batch(streams, n_threads=4, entropy=entropy)
        assert (status == 0).all(), (entropy, status)
        assert np.array_equal(got, frames), entropy
    fe.close()


@pytest.mark.parametrize("geometry", ["fullres", "parity"])
def test_front_end_on_the_camera_streams(real, geometry):
    streams, frames = real
    cfg = default_config(geometry)
    B = len(streams)
    fe = FrontEnd(cfg, max_frames=B, max_lines_per_color=4096)
    o = O.Oracle(cfg)
    # decode on the device straight into the handle's frame buffer, then the front end on the device frames
    buf, nbytes = fe.frames_buffer()
    status = fe.decode_jpeg_batch(streams, n_threads=4, device_ptr=buf)
    assert (status == 0).all()
    seg = fe.process_batch(buf, describe=True, n_frames=B)
    seg2 = fe.process_batch(frames, describe=True)                      # and from host frames: the same
    total = 0
    for f in range(B):
        r = o.process_frame(frames[f], cap=3 * 4096)
        for sg in (seg, seg2):
            s = sg.frame(f)
            assert s.n == r["n"], (f, s.n, r["n"])
            for k in ("lines", "normals", "color", "pixels_normalized", "ground", "keep", "code"):
                assert np.array_equal(getattr(s, k), r[k]), (f, k)
            assert np.array_equal(s.desc, r["desc"], equal_nan=True)
        total += r["n"]
    print("\n%s: %d segments in %d camera frames (%.1f per frame)" % (geometry, total, B, total / B))
    assert total > (60 if geometry == "fullres" else 15) * B // 2
    # k_lsd_grow's other forms (tests/test_gpu_parity.py::test_region_growing_slice_size_does_not_change_results): the row lists of
    # rounds 1 - 3, and the bit plane with so few USED bits that the largest problems of these frames overflow it
    import os
    for form in ("0", "8192"):
        os.environ["LF_GROW_BITMAP"] = form
        try:
            fe2 = FrontEnd(cfg, max_frames=B, max_lines_per_color=4096)
        finally:
            del os.environ["LF_GROW_BITMAP"]
        for _ in range(2):                                                 # the second batch runs on the slice the first one chose
            sg = fe2.process_batch(frames, describe=True)
            assert sg.n == seg.n and np.array_equal(sg.lines, seg.lines) and np.array_equal(sg.code, seg.code)
            assert np.array_equal(sg.frame_offset, seg.frame_offset)
        fe2.close()
    fe.close()


def test_keylines_on_the_camera_streams(real):
    streams, frames = real
    cfg = default_config("fullres")
    B = len(streams)
    o = O.Oracle(cfg)
    gray = np.stack([o.bgr2gray(o.preprocess(f)) for f in frames])
    fe = FrontEnd(cfg, max_frames=B, max_lines_per_color=256)
    k = fe.keylines_batch(frames, n_octaves=3, capacity=B * 6000)
    total = 0
    for f in range(B):
        r = O.octave_keylines(gray[f], 3)
        a, b = int(k["frame_offset"][f]), int(k["frame_offset"][f + 1])
        assert r is not None and b - a == r["n"] and k["frame_status"][f] == 0, (f, b - a, None if r is None else r["n"])
        for name in ("start_end", "in_octave", "angle", "num_pixels", "line_length", "octave", "class_id", "response", "size", "pt", "salience", "code"):
            assert np.array_equal(k[name][a:b], r[name]), (f, name)
        assert np.array_equal(k["desc"][a:b], r["desc"])
        total += r["n"]
    assert k["n"] == total
    print("\n%d KeyLines in %d camera frames, three octaves (%.1f per frame)" % (total, B, total / B))
    fe.close()


def test_opencv32_seed_order_on_the_camera_streams(real):
    """The std::sort seed order (Kinetic's OpenCV) at the reference's own geometry on camera content, and at full resolution on a
    few frames (busy problems: thousands of seeds in a handful of bins)."""
    streams, frames = real
    for geometry, idx in (("parity", range(28)), ("fullres", (0, 9, 17, 25))):
        cfg = default_config(geometry)
        cfg["lsd"]["seed_order"] = "opencv32"
        sel = frames[list(idx)]
        fe = FrontEnd(cfg, max_frames=len(sel), max_lines_per_color=4096)
        o = O.Oracle(cfg)
        seg = fe.process_batch(sel)
        for f in range(len(sel)):
            r = o.process_frame(sel[f], cap=3 * 4096, describe=False)
            s = seg.frame(f)
            assert s.n == r["n"], (geometry, f, s.n, r["n"])
            assert np.array_equal(s.lines, r["lines"]) and np.array_equal(s.color, r["color"]) and np.array_equal(s.keep, r["keep"])
        fe.close()
