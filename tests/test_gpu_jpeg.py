"""GPU parity of the JPEG ingest stage: lf_jpeg_decode_batch through the C ABI against
the committed libjpeg-turbo vectors (bit exact) and against the pinned oracle."""
import hashlib
import io
import os

import numpy as np
import pytest

from lane_slam_amd import FrontEnd, default_config, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vectors(golden_dir):
    return np.load(os.path.join(golden_dir, "jpeg_vectors.npz"))


@pytest.fixture(scope="module")
def fe():
    return FrontEnd(default_config("parity"), max_frames=32, max_lines_per_color=256)


def test_every_golden_stream_decodes_bit_exactly(fe, vectors):
    from lane_slam_amd.jpg import image_cv_from_jpg, jpg_info
    from oracle.oracle import jpeg_decode
    for name in [str(n) for n in vectors["names"]]:
        data = bytes(vectors["jpeg_" + name])
        rows, cols = jpg_info(data)[:2]
        got, status = fe.decode_jpeg_batch([data], rows=rows, cols=cols)
        assert status[0] == 0, name
        assert hashlib.sha256(got[0].tobytes()).digest() == bytes(vectors["sha_" + name]), name
        if "bgr_" + name in vectors:
            assert np.array_equal(got[0], vectors["bgr_" + name]), name
        assert np.array_equal(got[0], jpeg_decode(data)), name
        assert np.array_equal(image_cv_from_jpg(data), got[0])           # the reference-named single-frame helper


def test_mixed_batch_with_failures(fe, vectors):
    """One call, different sampling layouts, plus streams the reference would drop: per-frame status,
    failed frames come back as zeros, the good ones are unaffected."""
    names = ["lane_q75_444", "lane_q30_422", "lane_q95_420", "gray", "lane_rst_420"]
    streams = [bytes(vectors["jpeg_" + n]) for n in names]
    streams.insert(2, bytes(vectors["jpeg_truncated"]))
    streams.insert(4, bytes(vectors["jpeg_progressive"]))
    streams.append(bytes(vectors["jpeg_noise_q75_420"]))                 # decodable, but another size
    streams.append(b"")
    rows, cols = vectors["shape_lane_q75_444"][:2]
    got, status = fe.decode_jpeg_batch(streams, rows=int(rows), cols=int(cols), n_threads=3)
    assert list(status) == [0, 0, -6, 0, -5, 0, 0, -1, -1]      # corrupt, unsupported, wrong size, NULL stream
    good = [0, 1, 3, 5, 6]
    for i, n in zip(good, names):
        ref = vectors["bgr_" + n]
        assert np.array_equal(got[i], ref), n
    for i in (2, 4, 7, 8):
        assert not got[i].any()
    # without a status array a failed frame is an error of the call
    from lane_slam_amd import LanefrontError
    import ctypes
    bufs = [np.frombuffer(s, np.uint8) for s in streams[:3]]
    ptrs = (ctypes.c_void_p * 3)(*[b.ctypes.data for b in bufs])
    sizes = (ctypes.c_size_t * 3)(*[b.size for b in bufs])
    out = np.empty((3, int(rows), int(cols), 3), np.uint8)
    rc = fe.lib.lf_jpeg_decode_batch(fe.h, ptrs, sizes, 3, int(rows), int(cols), out.ctypes.data_as(ctypes.c_void_p), 0, 1, None)
    assert rc == -6 and b"could not be decoded" in fe.lib.lf_last_error(fe.h)
    with pytest.raises(ValueError):
        from lane_slam_amd.jpg import image_cv_from_jpg
        image_cv_from_jpg(b"definitely not a jpeg")


def test_camera_sized_batch_against_oracle_and_pillow():
    """640x480 synthetic lane frames encoded at run time (Pillow, if present) in the three chroma layouts:
    the batch decode must equal the pinned oracle and Pillow's libjpeg-turbo, and feeding the decoded
    device buffer straight into the front end must give the same segments as feeding the pixels."""
    Image = pytest.importorskip("PIL.Image")
    from oracle.oracle import jpeg_decode
    n = 12
    cfg = default_config("parity")
    fe = FrontEnd(cfg, max_frames=n, max_lines_per_color=256)
    streams = []
    for i in range(n):
        b = io.BytesIO()
        Image.fromarray(synth.make_frame(40 + i)[..., ::-1].copy()).save(b, "JPEG", quality=(60, 75, 90)[i % 3], subsampling=i % 3)
        streams.append(b.getvalue())
    frames, status = fe.decode_jpeg_batch(streams, n_threads=4)
    assert not status.any() and frames.shape == (n, 480, 640, 3)
    for i in range(n):
        assert np.array_equal(frames[i], jpeg_decode(streams[i])), i
        pil = np.asarray(Image.open(io.BytesIO(streams[i])).convert("RGB"))[..., ::-1]
        assert np.array_equal(frames[i], pil), i
    # decode into the handle's own staging buffer and process from there
    dev, nbytes = fe.frames_buffer()
    assert nbytes >= frames.nbytes
    st = fe.decode_jpeg_batch(streams, n_threads=4, device_ptr=dev)
    assert not st.any()
    a = fe.process_batch(frames)
    fe.decode_jpeg_batch(streams, n_threads=4, device_ptr=dev)
    b = fe.process_batch(dev, n_frames=n)
    assert b.n == a.n and a.n > 0
    for k in ("frame_offset", "lines", "normals", "color", "ground", "keep", "desc", "code"):
        assert np.array_equal(getattr(a, k), getattr(b, k)), k


def test_decode_into_the_handle_buffer_is_bounds_checked(vectors):
    from lane_slam_amd import LanefrontError
    fe2 = FrontEnd(default_config("parity"), max_frames=2, max_lines_per_color=16)
    dev, nbytes = fe2.frames_buffer()
    assert nbytes == 2 * 480 * 640 * 3
    data = bytes(vectors["jpeg_full_640x480_420"])
    assert not fe2.decode_jpeg_batch([data, data], device_ptr=dev).any()
    with pytest.raises(LanefrontError):
        fe2.decode_jpeg_batch([data, data, data], device_ptr=dev)       # three frames into a two-frame buffer
