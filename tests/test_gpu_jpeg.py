"""GPU parity of the JPEG ingest stage: lf_jpeg_decode_batch through the C ABI against
the committed libjpeg-turbo vectors (bit exact) and against the pinned oracle."""
import hashlib
import io
import os

import numpy as np
import pytest
import torch  # noqa: F401  (before the HIP library is loaded: torch brings its own HIP runtime, which has to initialise first)

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


def test_device_entropy_decoder_equals_the_host_decoder(fe, vectors):
    """lf_jpeg_decode_batch_gpu (Huffman decoding by self-synchronising subsequences on the GPU, k_jhuff.hip) against
    lf_jpeg_decode_batch (host Huffman threads): every golden stream, one batch with all of them, and several hundred
    damaged streams -- the same pixels and the same per-frame status (a corrupt frame is refused by both, never decoded
    differently)."""
    names = [str(n) for n in vectors["names"]]
    by_size = {}
    for name in names:
        data = bytes(vectors["jpeg_" + name])
        from lane_slam_amd.jpg import jpg_info
        by_size.setdefault(tuple(jpg_info(data)[:2]), []).append((name, data))
    for (rows, cols), items in by_size.items():
        streams = [d for _, d in items]
        g, gs = fe.decode_jpeg_batch(streams, rows=rows, cols=cols, entropy="gpu")
        hh, hs = fe.decode_jpeg_batch(streams, rows=rows, cols=cols, entropy="host", n_threads=2)
        assert np.array_equal(gs, hs) and not gs.any(), [n for n, _ in items]
        assert np.array_equal(g, hh), (rows, cols)
        for i, (name, _) in enumerate(items):
            if "bgr_" + name in vectors:
                assert np.array_equal(g[i], vectors["bgr_" + name]), name
    # damaged entropy data: bit flips, truncation, bytes turned into markers, a stuffed zero removed
    rng = np.random.default_rng(77)
    base = [bytes(vectors["jpeg_" + n]) for n in ("lane_q75_420", "lane_rst_420", "lane_q30_422", "gray")]
    rows, cols = (int(v) for v in vectors["shape_lane_q75_420"][:2])
    bad = []
    for i in range(400):
        d = bytearray(base[i % len(base)])
        sos = d.rfind(b"\xff\xda")
        kind = i % 5
        if kind == 0:
            for _ in range(int(rng.integers(1, 4))):
                p = int(rng.integers(sos + 14, len(d) - 2)); d[p] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            d = d[:int(rng.integers(sos + 20, len(d)))]
        elif kind == 2:
            p = int(rng.integers(sos + 14, len(d) - 2)); d[p] = 0xFF; d[p + 1] = int(rng.choice([0xD0, 0xD3, 0xD9, 0x00, 0xFF, 0xC4]))
        elif kind == 3:
            p = d.find(b"\xff\x00", sos)
            if p > 0:
                del d[p + 1]
        else:
            p = int(rng.integers(sos + 14, len(d) - 2)); d[p:p] = bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8))
        bad.append(bytes(d))
    for k in range(0, len(bad), 32):
        chunk = bad[k:k + 32]
        g, gs = fe.decode_jpeg_batch(chunk, rows=rows, cols=cols, entropy="gpu")
        hh, hs = fe.decode_jpeg_batch(chunk, rows=rows, cols=cols, entropy="host", n_threads=4)
        assert np.array_equal(gs, hs), (k, gs, hs)
        assert np.array_equal(g, hh), k
    assert True


def test_device_entropy_decoder_on_camera_frames():
    Image = pytest.importorskip("PIL.Image")
    n = 24
    fe2 = FrontEnd(default_config("parity"), max_frames=n, max_lines_per_color=64)
    streams = []
    for i in range(n):
        b = io.BytesIO()
        kw = {"optimize": True} if i % 4 == 3 else {}
        Image.fromarray(synth.make_frame(70 + i)[..., ::-1].copy()).save(b, "JPEG", quality=(35, 80, 95)[i % 3], subsampling=(i // 3) % 3, **kw)
        streams.append(b.getvalue())
    g, gs = fe2.decode_jpeg_batch(streams, entropy="gpu")
    assert not gs.any()
    for i in range(n):
        pil = np.asarray(Image.open(io.BytesIO(streams[i])).convert("RGB"))[..., ::-1]
        assert np.array_equal(g[i], pil), i
    fe2.close()


def test_device_entropy_decoder_on_large_frames_and_restart_intervals():
    """1920x1080 streams: scans of several hundred KB, beyond what the decode kernel stages in LDS (the global-memory form of
    the same code), thousands of subsequences per frame, with and without restart intervals; against Pillow and the host
    entropy path."""
    Image = pytest.importorskip("PIL.Image")
    cfg = default_config("fullres", in_size=(1080, 1920))
    fe2 = FrontEnd(cfg, max_frames=4, max_lines_per_color=64)
    rng = np.random.default_rng(5)
    streams = []
    for i in range(4):
        img = synth.make_batch(1, 40 + i, rows=1080, cols=1920)[0]
        if i % 2:
            img = np.clip(img.astype(np.int16) + rng.integers(-25, 26, img.shape, dtype=np.int16), 0, 255).astype(np.uint8)    # noisy: a long scan
        b = io.BytesIO()
        kw = {"restart_marker_blocks": 37} if i >= 2 else {}
        try:
            Image.fromarray(img[..., ::-1].copy()).save(b, "JPEG", quality=(92, 97)[i % 2], subsampling=(2, 0)[i % 2], **kw)
        except TypeError:                                   # an older Pillow without restart_marker_blocks
            b = io.BytesIO()
            Image.fromarray(img[..., ::-1].copy()).save(b, "JPEG", quality=(92, 97)[i % 2], subsampling=(2, 0)[i % 2])
        streams.append(b.getvalue())
    assert max(len(s) for s in streams) > 200 * 1024          # really beyond the 96 KB LDS stage
    g, gs = fe2.decode_jpeg_batch(streams, entropy="gpu")
    h, hs = fe2.decode_jpeg_batch(streams, entropy="host", n_threads=4)
    assert not gs.any() and not hs.any()
    for i in range(4):
        pil = np.asarray(Image.open(io.BytesIO(streams[i])).convert("RGB"))[..., ::-1]
        assert np.array_equal(g[i], pil), i
        assert np.array_equal(g[i], h[i]), i
    fe2.close()


def test_queued_decode_equals_the_waiting_call(fe, vectors):
    """lf_jpeg_decode_batch_gpu_async + lf_jpeg_status: the same frames and the same per-frame status as the call that waits, a batch
    with streams the reference would drop included; then the queued batch straight into the front end, nothing waited for in between."""
    names = ["lane_q75_444", "lane_q30_422", "lane_q95_420", "gray", "lane_rst_420"]
    streams = [bytes(vectors["jpeg_" + n]) for n in names]
    streams.insert(2, bytes(vectors["jpeg_truncated"]))
    streams.insert(4, bytes(vectors["jpeg_progressive"]))
    streams.append(b"")
    rows, cols = (int(v) for v in vectors["shape_lane_q75_444"][:2])
    want, want_status = fe.decode_jpeg_batch(streams, rows=rows, cols=cols, n_threads=3)
    d = torch.full((len(streams), rows, cols, 3), 7, dtype=torch.uint8, device="cuda")
    assert fe.decode_jpeg_batch_async(streams, device_ptr=d.data_ptr(), rows=rows, cols=cols, n_threads=3) == d.data_ptr()
    status = fe.jpeg_status()
    assert list(status) == list(want_status) == [0, 0, -6, 0, -5, 0, 0, -1]
    assert np.array_equal(d.cpu().numpy(), want)
    # a status asked for without a queued batch, or for another number of frames, is an error
    from lane_slam_amd import LanefrontError
    import ctypes
    st = (ctypes.c_int * 3)()
    assert fe.lib.lf_jpeg_status(fe.h, st, 3, None) != 0
    fe2 = FrontEnd(default_config("parity"), max_frames=8)
    assert fe2.lib.lf_jpeg_status(fe2.h, st, 3, None) != 0
    # decode -> detect on the handle's own buffer, queued back to back
    B = 6
    host = synth.make_batch(B, seed0=40, threads=2)
    js = []
    from PIL import Image
    for i in range(B):
        b = io.BytesIO()
        Image.fromarray(host[i][..., ::-1].copy()).save(b, "JPEG", quality=85, subsampling=2)
        js.append(b.getvalue())
    frames, st_ = fe2.decode_jpeg_batch(js)
    assert not st_.any()
    ref = fe2.process_batch(frames)
    buf = fe2.decode_jpeg_batch_async(js)
    got = fe2.process_batch(buf, n_frames=B)
    assert not fe2.jpeg_status().any()
    assert got.n == ref.n and ref.n > 0
    for k in ("frame_offset", "lines", "ground", "keep", "code"):
        assert np.array_equal(getattr(got, k), getattr(ref, k)), k
    # ... and the form that decodes only the rows the front end reads (from the crop line on): the same segments; the rows it produced
    # are the whole decode's, the rows above it are left as they were
    import torch as _t
    rows, cols = fe2.cfg["in_size"]
    nb = B * rows * cols * 3
    filler = _t.full((nb,), 9, dtype=_t.uint8, device="cuda")
    import ctypes as ct
    hip = ct.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(ct.c_void_p(buf), ct.c_void_p(filler.data_ptr()), ct.c_size_t(nb), 3) == 0      # device to device
    assert fe2.decode_jpeg_batch_async(js, for_detect=True) == buf
    got2 = fe2.process_batch(buf, n_frames=B)
    assert not fe2.jpeg_status().any()
    for k in ("frame_offset", "lines", "ground", "keep", "code"):
        assert np.array_equal(getattr(got2, k), getattr(ref, k)), k
    back = _t.empty(nb, dtype=_t.uint8, device="cuda")
    assert hip.hipMemcpy(ct.c_void_p(back.data_ptr()), ct.c_void_p(buf), ct.c_size_t(nb), 3) == 0
    back = back.cpu().numpy().reshape(B, rows, cols, 3)
    cut = fe2.cfg["top_cutoff"] * rows // fe2.cfg["img_size"][0]
    assert np.array_equal(back[:, cut:], frames[:, cut:])
    assert (back[:, :cut - 16] == 9).all()
    with pytest.raises(ValueError):
        fe2.decode_jpeg_batch_async(js, device_ptr=filler.data_ptr(), for_detect=True)
    fe2.close()
