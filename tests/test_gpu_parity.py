"""GPU parity tests: every stage of the HIP path (through the C ABI) against the CPU oracle
on identical seeded inputs.  Integer/byte/index work and endpoints must be bit-exact;
descriptor floats are checked both bit-exact (same deterministic arithmetic) and within the
north-star tolerance 1e-4."""
import ctypes

import os

import numpy as np
import pytest

from lane_slam_amd import FrontEnd, LanefrontError, LineDetectorHIP, default_config, synth
from lane_slam_amd import _lib
from lane_slam_amd.config import DEFAULT_DETECTOR_CONFIGURATION

pytestmark = pytest.mark.gpu


def _frames(n, seed0=0, with_noise=True):
    fr = synth.make_batch(n, seed0)
    if with_noise:
        rng = np.random.default_rng(99)
        fr[-1] = rng.integers(0, 256, fr[-1].shape, dtype=np.uint8)       # worst case: pure noise
    return fr


@pytest.fixture(scope="module", params=["parity", "fullres"])
def setup(request):
    from oracle.oracle import Oracle
    geo = request.param
    cfg = default_config(geo)
    n = 6 if geo == "parity" else 3
    fe = FrontEnd(cfg, max_frames=n, max_lines_per_color=4096)
    frames = _frames(n)
    seg = fe.process_batch(frames, describe=True)
    yield geo, cfg, fe, Oracle(cfg), frames, seg
    fe.close()


def test_detmath_device_matches_oracle():
    from oracle.oracle import detmath_lib
    fe = FrontEnd(default_config("parity"))
    L = detmath_lib()
    rng = np.random.default_rng(5)
    vp = ctypes.c_void_p

    def dev(which, a, b=None):
        a = np.ascontiguousarray(a, np.float64)
        y = np.empty_like(a)
        bp = None if b is None else np.ascontiguousarray(b, np.float64).ctypes.data_as(vp)
        fe._check(fe.lib.lf_debug_detmath(fe.h, which, a.ctypes.data_as(vp), bp, y.ctypes.data_as(vp), a.size))
        return y

    def host1(which, a):
        a = np.ascontiguousarray(a, np.float64)
        y = np.empty_like(a)
        L.lfo_vec_unary(which, a.ctypes.data_as(vp), y.ctypes.data_as(vp), a.size)
        return y

    def host2(which, a, b):
        a = np.ascontiguousarray(a, np.float64)
        b = np.ascontiguousarray(b, np.float64)
        y = np.empty_like(a)
        L.lfo_vec_binary(which, a.ctypes.data_as(vp), b.ctypes.data_as(vp), y.ctypes.data_as(vp), a.size)
        return y

    n = 200000
    for which, lo, hi in [(0, -50, 50), (1, 1e-9, 1e9), (2, -30, 30), (3, -30, 30), (4, -100, 100), (5, -1, 1),
                          (6, 1e-9, 1e9), (7, -0.2, 0.2)]:
        x = rng.uniform(lo, hi, n)
        assert np.array_equal(dev(which, x), host1(which, x)), "detmath fn %d differs on device" % which
    a, b = rng.uniform(-5, 5, n), rng.uniform(-5, 5, n)
    assert np.array_equal(dev(8, a, b), host2(0, a, b))
    a = rng.uniform(0.01, 30, n)
    b = np.where(rng.random(n) < 0.5, rng.integers(0, 40, n).astype(float), rng.uniform(-3, 3, n))
    assert np.array_equal(dev(9, a, b), host2(1, a, b))
    # IEEE basics the whole contract rests on: sqrt and division, double and float
    x = np.concatenate([rng.uniform(0, 1e6, n), rng.uniform(0, 1e-300, 1000)])
    assert np.array_equal(dev(10, x), np.sqrt(x))
    a, b = rng.uniform(-1e3, 1e3, n), rng.uniform(-1e3, 1e3, n)
    assert np.array_equal(dev(11, a, b), a / b)
    xf = rng.uniform(0, 1e6, n).astype(np.float32)
    assert np.array_equal(dev(13, xf.astype(np.float64)).astype(np.float32), np.sqrt(xf))
    af, bf = rng.uniform(-1e3, 1e3, n).astype(np.float32), rng.uniform(-1e3, 1e3, n).astype(np.float32)
    assert np.array_equal(dev(14, af.astype(np.float64), bf.astype(np.float64)).astype(np.float32), af / bf)
    yf, xf2 = rng.uniform(-10, 10, n).astype(np.float32), rng.uniform(-10, 10, n).astype(np.float32)
    ref = np.empty(n, np.float32)
    L.lfo_vec_fast_atan2(yf.ctypes.data_as(vp), xf2.ctypes.data_as(vp), ref.ctypes.data_as(vp), n)
    assert np.array_equal(dev(12, yf.astype(np.float64), xf2.astype(np.float64)).astype(np.float32), ref)
    fe.close()


def test_pre_masks_canny(setup):
    geo, cfg, fe, o, frames, seg = setup
    n = frames.shape[0]
    bgr = fe.fetch(_lib.LF_BUF_BGR, n)
    masks = fe.fetch(_lib.LF_BUF_MASKS, n)
    edges = fe.fetch(_lib.LF_BUF_EDGES, n)
    for f in range(n):
        ob = o.preprocess(frames[f])
        assert np.array_equal(bgr[f], ob)
        bw = o.color_masks(o.bgr2hsv(ob))
        for c in range(3):
            assert np.array_equal(masks[f, c], o.dilate(bw[c])), (f, c)
        oe = o.canny(ob)
        assert np.array_equal(edges[f], oe), "canny frame %d: %d px differ" % (f, (edges[f] != oe).sum())
    assert edges.any() and masks.any()


def test_lsd_gradient_and_order(setup):
    geo, cfg, fe, o, frames, seg = setup
    n = frames.shape[0]
    ang = fe.fetch(_lib.LF_BUF_LSD_ANGLE, n)
    mod = fe.fetch(_lib.LF_BUF_LSD_MODGRAD, n)
    order = fe.fetch(_lib.LF_BUF_LSD_ORDER, n)
    norder = fe.fetch(_lib.LF_BUF_LSD_NORDER, n)
    masks = fe.fetch(_lib.LF_BUF_MASKS, n)
    edges = fe.fetch(_lib.LF_BUF_EDGES, n)
    for f in range(n):
        for c in range(3):
            ec = masks[f, c] & edges[f]
            scaled = o.lsd_scaled_image(ec)
            oang, omod, oorder = o.lsd_ll_angle(scaled)
            defined = oang != -1024.0
            g_def = ang[f, c] != np.float32(-1024.0)
            assert np.array_equal(defined, g_def), (f, c)
            rad = ang[f, c].astype(np.float64) * (np.pi / 180)
            assert np.array_equal(rad[defined], oang[defined])
            assert np.array_equal(mod[f, c][defined], omod[defined])       # the pipeline keeps magnitudes of defined pixels only
            assert not mod[f, c][~defined].any()
            k = int(norder[f, c])
            assert k == int(defined.sum())
            got = (order[f, c, :k] & 0xFFFFF).astype(np.int64)      # compact index = raster rank among defined pixels
            want_addr = oorder[defined.ravel()[oorder]]             # the oracle lists every pixel; keep defined ones
            rank = np.cumsum(defined.ravel()) - 1
            assert np.array_equal(got, rank[want_addr]), (f, c)


def test_segments_match_oracle(setup):
    geo, cfg, fe, o, frames, seg = setup
    n = frames.shape[0]
    total = 0
    for f in range(n):
        r = o.process_frame(frames[f], cap=3 * 4096)
        s = seg.frame(f)
        assert s.n == r["n"], "frame %d: %d segments vs oracle %d" % (f, s.n, r["n"])
        total += s.n
        assert np.array_equal(s.lines, r["lines"])                       # endpoints: bit exact
        assert np.array_equal(s.normals, r["normals"])                   # normals: bit exact
        assert np.array_equal(s.color, r["color"])                       # colour labels
        assert np.array_equal(s.pixels_normalized, r["pixels_normalized"])
        assert np.array_equal(s.ground, r["ground"])
        assert np.array_equal(s.keep, r["keep"])
        np.testing.assert_allclose(s.desc, r["desc"], rtol=0, atol=1e-4, equal_nan=True)   # north-star tolerance
        assert np.array_equal(s.desc, r["desc"], equal_nan=True)          # and in fact the same bits
        assert np.array_equal(s.code, r["code"])
    assert total > 20
    assert seg.n == total


def test_plugin_matches_reference_interface():
    from oracle.oracle import Oracle
    cfg = default_config("parity")
    o = Oracle(cfg)
    det = LineDetectorHIP(dict(DEFAULT_DETECTOR_CONFIGURATION))
    for seed in (0, 4, 7):
        work = o.preprocess(synth.make_frame(seed))          # what the node hands to setImage
        det.setImage(work)
        assert np.array_equal(det.getImage(), work)
        bw = o.color_masks(o.bgr2hsv(work))
        edges = o.canny(work)
        for ci, color in enumerate(("white", "yellow", "red")):
            d = det.detectLines(color)
            area = o.dilate(bw[ci])
            lines = o.lsd(area & edges)
            assert np.array_equal(d.area, area)
            if len(lines) == 0:
                assert isinstance(d.lines, list) and len(d.lines) == 0      # line_detector_lsd.py:68-71
                continue
            ol, on, oc = o.find_normals(area, lines)
            assert d.lines.dtype == np.float32 and d.normals.dtype == np.float64 and d.centers.dtype == np.float32
            assert np.array_equal(d.lines, ol) and np.array_equal(d.normals, on) and np.array_equal(d.centers, oc)
    with pytest.raises(Exception):
        det.detectLines("blue")
    with pytest.raises(ValueError):
        LineDetectorHIP({"hsv_white1": [0, 0, 0]})
    # empty image -> [] for every colour, like cv2 returning None
    det.setImage(np.zeros((80, 160, 3), np.uint8))
    for color in ("white", "yellow", "red"):
        assert det.detectLines(color).lines == []


def _associate_both_rules(fe, o, q, m):
    """lf_associate under both tie rules against the oracle's two statements: the reference's first-discovered rule (the
    default, lfo_match_mih) and the lowest index (lfo_match).  Returns the lowest-index result."""
    fe.set_tie_rule("mihasher")
    gi, gd = fe.associate(q, m)
    wi, wd, _ = o.match_mih(q, m)
    assert np.array_equal(gd, wd) and np.array_equal(gi, wi), ("mihasher", int((gi != wi).sum()))
    fe.set_tie_rule("lowest")
    idx, dist = fe.associate(q, m)
    oi, od = o.match(q, m)
    assert np.array_equal(dist, od) and np.array_equal(idx, oi), ("lowest", int((idx != oi).sum()))
    assert np.array_equal(gd, dist)                # the distance never depends on the rule
    return idx, dist


def test_associate_matches_matcher_semantics():
    from oracle.oracle import Oracle
    o = Oracle(default_config("parity"))
    fe = FrontEnd(default_config("parity"))
    rng = np.random.default_rng(17)
    nm, nq = 3000, 700
    m = rng.integers(0, 256, (nm, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    # planted neighbours at distances 0..128, duplicates (ties), and far queries
    for i in range(0, 300):
        src = m[rng.integers(0, nm)].copy()
        bits = rng.choice(256, size=i % 129, replace=False)
        for b in bits:
            src[b >> 3] ^= np.uint8(1 << (b & 7))
        q[i] = src
    m[10] = m[5]
    q[300] = m[5]
    idx, dist = _associate_both_rules(fe, o, q, m)
    assert idx[300] == 5 and dist[300] == 0
    assert (idx >= 0).all()       # 3000 random codes: every random query has a neighbour within 128 bits
    # farther than D = 128 -> "no match" (binary_descriptor_matcher.cpp:721)
    far_q = np.concatenate([~m[:1], m[:1]])
    i_far, d_far = _associate_both_rules(fe, o, far_q, m[:1])
    assert i_far.tolist() == [-1, 0] and d_far.tolist() == [-1.0, 0.0]
    # ragged sizes and the empty map
    for a, b in [(1, 1), (5, 63), (129, 64), (33, 65), (128, 1000)]:
        _associate_both_rules(fe, o, q[:a], m[:b])
    for rule in ("mihasher", "lowest"):
        fe.set_tie_rule(rule)
        i3, d3 = fe.associate(q[:7], m[:0])
        assert (i3 == -1).all() and (d3 == -1).all()
    # float LBD distances within the north-star tolerance
    qd = rng.random((200, 72)).astype(np.float32)
    md = rng.random((900, 72)).astype(np.float32)
    qd /= np.linalg.norm(qd, axis=1, keepdims=True)
    md /= np.linalg.norm(md, axis=1, keepdims=True)
    fi, fd = fe.associate_float(qd, md)
    gi, gd = o.match_float(qd, md)
    np.testing.assert_allclose(fd, gd, rtol=0, atol=1e-4)
    true_d = np.linalg.norm(qd.astype(np.float64) - md[fi].astype(np.float64), axis=1)
    np.testing.assert_allclose(true_d, gd, rtol=0, atol=1e-4)     # the chosen neighbour is (near-)optimal
    fe.close()


def test_full_size_properties():
    """BASELINE config 2 size (256 x 640x480, full-res geometry): size-independent properties."""
    cfg = default_config("fullres")
    n = 256
    fe = FrontEnd(cfg, max_frames=n, max_lines_per_color=1024)
    base = synth.make_batch(8, 100)
    frames = np.ascontiguousarray(np.tile(base, (n // 8, 1, 1, 1)))
    seg = fe.process_batch(frames, describe=True)
    # identical frames give identical segment lists wherever they sit in the batch
    first = [seg.frame(f) for f in range(8)]
    for f in range(8, n):
        a, b = first[f % 8], seg.frame(f)
        assert a.n == b.n
        assert np.array_equal(a.lines, b.lines) and np.array_equal(a.code, b.code) and np.array_equal(a.keep, b.keep)
    assert np.all(np.diff(seg.frame_offset) >= 0) and seg.frame_offset[-1] == seg.n
    assert np.isin(seg.color, (0, 1, 2)).all()
    # colour order inside every frame: white, yellow, red
    for f in range(8):
        assert np.all(np.diff(seg.frame(f).color.astype(int)) >= 0)
    # a frame's descriptors match themselves at distance 0
    idx, dist = fe.associate(seg.code, seg.code)
    assert (dist == 0).all()
    fe.close()


def test_anti_instagram_transform_on_device(golden_dir):
    """a-1 with a non-identity AntiInstagram transform (the fast identity path is skipped):
    float32 scale/shift + convertScaleAbs rounding must match the oracle, which itself is pinned
    to the reference's scaleandshift2 by tests/golden/scaleandshift.npz."""
    import os
    from oracle.oracle import Oracle
    g = np.load(os.path.join(golden_dir, "scaleandshift.npz"))
    frames = synth.make_batch(2, 50)
    for i in range(1, g["scales"].shape[0]):
        for geo in ("parity", "fullres"):
            cfg = default_config(geo)
            cfg["ai_scale"] = list(g["scales"][i])
            cfg["ai_shift"] = list(g["shifts"][i])
            fe = FrontEnd(cfg, max_frames=2, max_lines_per_color=4096)
            seg = fe.process_batch(frames, describe=False)
            o = Oracle(cfg)
            bgr = fe.fetch(_lib.LF_BUF_BGR, 2)
            for f in range(2):
                assert np.array_equal(bgr[f], o.preprocess(frames[f])), (i, geo, f)
                r = o.process_frame(frames[f], cap=3 * 4096, describe=False)
                assert np.array_equal(seg.frame(f).lines, r["lines"])
            fe.close()


def test_edge_cases_and_errors():
    from oracle.oracle import Oracle
    from lane_slam_amd import LanefrontError
    cfg = default_config("parity")
    fe = FrontEnd(cfg, max_frames=3, max_lines_per_color=2048)
    # all-black and all-white frames: no segments, offsets stay zero (cv2 LSD returning None -> [])
    blank = np.zeros((3, 480, 640, 3), np.uint8)
    blank[1] = 255
    seg = fe.process_batch(blank)
    assert seg.n == 0 and seg.frame_offset.tolist() == [0, 0, 0, 0]
    # a single frame, then a ragged batch mixing empty and busy frames
    frames = synth.make_batch(3, 20)
    frames[1] = 0
    o = Oracle(cfg)
    seg = fe.process_batch(frames)
    for f in range(3):
        r = o.process_frame(frames[f])
        assert seg.frame(f).n == r["n"] and np.array_equal(seg.frame(f).lines, r["lines"])
    assert seg.frame(1).n == 0
    one = fe.process_batch(frames[2])
    assert one.n == seg.frame(2).n and np.array_equal(one.code, seg.frame(2).code)
    # too many frames / too few line slots -> LF_ERR_CAPACITY, never silent truncation
    with pytest.raises(LanefrontError) as e:
        fe.process_batch(np.zeros((4, 480, 640, 3), np.uint8))
    assert e.value.code == -2
    fe.close()
    small = FrontEnd(cfg, max_frames=1, max_lines_per_color=2)
    with pytest.raises(LanefrontError) as e:
        small.process_batch(frames[0])
    assert e.value.code == -2 and "max_lines_per_color" in str(e.value)
    small.close()
    with pytest.raises(ValueError):
        FrontEnd(cfg).process_batch(np.zeros((1, 100, 100, 3), np.uint8))
    # a different camera size feeding the same working geometry (nearest-neighbour resize 2x)
    cfg2 = default_config("parity")
    cfg2["in_size"] = [240, 320]
    fe2 = FrontEnd(cfg2, max_frames=2, max_lines_per_color=2048)
    fr2 = np.ascontiguousarray(synth.make_batch(2, 30)[:, ::2, ::2])
    s2 = fe2.process_batch(fr2)
    o2 = Oracle(cfg2)
    for f in range(2):
        r = o2.process_frame(fr2[f])
        assert np.array_equal(s2.frame(f).lines, r["lines"]) and np.array_equal(s2.frame(f).ground, r["ground"])
    fe2.close()


def test_plugin_handles_changing_image_size():
    from oracle.oracle import Oracle
    det = LineDetectorHIP(dict(DEFAULT_DETECTOR_CONFIGURATION))
    for geo in ("parity", "fullres", "parity"):
        cfg = default_config(geo)
        o = Oracle(cfg)
        work = o.preprocess(synth.make_frame(3))
        det.setImage(work)
        d = det.detectLines("yellow")
        bw = o.color_masks(o.bgr2hsv(work))
        area = o.dilate(bw[1])
        ol, on, oc = o.find_normals(area, o.lsd(area & o.canny(work)))
        assert np.array_equal(d.lines, ol) and np.array_equal(d.normals, on)


@pytest.mark.parametrize("img_size,top_cutoff", [((75, 96), 12), ((200, 224), 31), ((131, 32), 3)])
def test_odd_working_geometries_match_oracle(img_size, top_cutoff):
    """Working images whose sides are no multiple of the kernels' strip / band / tile sizes (k_canny_nms: 64-column
    strips x 62-row bands, one wave each; k_pre 128x22, k_lbd_grad 64x64, k_lsd_grad 32x32 scaled tiles): a last band of
    ONE row (63 = 62 + 1), half strips, an image narrower than one strip.  Edge planes, masks and the final segments
    against the oracle, bit for bit."""
    from oracle.oracle import Oracle
    cfg = default_config("parity")
    cfg["img_size"] = list(img_size)
    cfg["top_cutoff"] = top_cutoff
    n = 3
    fe = FrontEnd(cfg, max_frames=n, max_lines_per_color=2048)
    o = Oracle(cfg)
    assert (fe.rows, fe.cols) == (img_size[0] - top_cutoff, img_size[1])
    frames = synth.make_batch(n, 40)
    seg = fe.process_batch(frames)
    edges = fe.fetch(_lib.LF_BUF_EDGES, n)
    masks = fe.fetch(_lib.LF_BUF_MASKS, n)
    dx = fe.fetch(_lib.LF_BUF_LBD_DX, n)
    for f in range(n):
        work = o.preprocess(frames[f])
        assert np.array_equal(edges[f], o.canny(work)), "canny frame %d" % f
        bw = o.color_masks(o.bgr2hsv(work))
        for c in range(3):
            assert np.array_equal(masks[f, c], o.dilate(bw[c])), (f, c)
        r = o.process_frame(frames[f])
        s_ = seg.frame(f)
        assert s_.n == r["n"]
        for k in ("lines", "normals", "color", "pixels_normalized", "ground", "keep", "code"):
            assert np.array_equal(getattr(s_, k), r[k]), (f, k)
    assert edges.any() and dx.any()
    fe.close()


def test_set_image_on_an_upscaling_handle():
    """lf_set_image takes the WORKING image (img_size - top_cutoff rows); on a handle whose working image is larger
    than its input frames (img_size > in_size, nearest-neighbour upscaling) the staging buffer must still hold it
    (it used to be sized from the input frame only and the copy ran past its end)."""
    from oracle.oracle import Oracle
    import ctypes
    cfg = default_config("parity")
    cfg["in_size"] = [120, 160]
    cfg["img_size"] = [480, 640]
    cfg["top_cutoff"] = 160
    fe = FrontEnd(cfg, max_frames=1, max_lines_per_color=2048)
    assert (fe.rows, fe.cols) == (320, 640)
    o = Oracle(default_config("fullres"))
    work = o.preprocess(synth.make_frame(11))                   # a 320 x 640 working image
    fe._check(fe.lib.lf_set_image(fe.h, work.ctypes.data_as(ctypes.c_void_p), 320, 640, work.strides[0]))
    lines = np.empty((2048, 4), np.float32)
    n = ctypes.c_int()
    area = np.empty((320, 640), np.uint8)
    fe._check(fe.lib.lf_detect_lines(fe.h, 0, lines.ctypes.data_as(ctypes.c_void_p), None, None, area.ctypes.data_as(ctypes.c_void_p),
                                     2048, ctypes.byref(n)))
    bw = o.dilate(o.color_masks(o.bgr2hsv(work))[0])
    ol, on, oc = o.find_normals(bw, o.lsd(bw & o.canny(work)))
    assert np.array_equal(area, bw) and np.array_equal(lines[: n.value], ol) and n.value > 3
    # and the batch path of the same handle upscales its 120 x 160 frames (nearest neighbour) like the oracle
    small = np.ascontiguousarray(synth.make_frame(12)[::4, ::4])
    seg = fe.process_batch(small[None])
    r = Oracle(cfg).process_frame(small)
    assert seg.n == r["n"] and np.array_equal(seg.lines, r["lines"]) and np.array_equal(seg.code, r["code"])
    fe.close()


def test_1080p_geometry_matches_oracle():
    """BASELINE config 5 geometry: 1920x1080 frames, img_size [1080,1920], top_cutoff 360 -> 1920x720
    working image (bit planes exceed one CU's LDS: HBM-resident hysteresis, whole-CU LSD problems)."""
    from oracle.oracle import Oracle
    cfg = default_config("fullres", in_size=(1080, 1920))
    frames = synth.make_batch(2, 70, rows=1080, cols=1920)
    rng = np.random.default_rng(7)
    frames[1, 400:700, 300:900] = rng.integers(0, 256, (300, 600, 3), dtype=np.uint8)     # a noisy patch
    fe = FrontEnd(cfg, max_frames=2, max_lines_per_color=8192)
    seg = fe.process_batch(frames)
    o = Oracle(cfg)
    edges = fe.fetch(_lib.LF_BUF_EDGES, 2)
    for f in range(2):
        assert np.array_equal(edges[f], o.canny(o.preprocess(frames[f])))
        r = o.process_frame(frames[f], cap=3 * 8192)
        s = seg.frame(f)
        assert s.n == r["n"] and s.n > 10
        assert np.array_equal(s.lines, r["lines"]) and np.array_equal(s.code, r["code"])
        assert np.array_equal(s.ground, r["ground"]) and np.array_equal(s.keep, r["keep"])
    fe.close()


def _clutter(rng, rows, cols, n_strokes, noise):
    img = np.zeros((rows, cols), np.uint8)
    for _ in range(n_strokes):
        y, x = rng.integers(0, rows), rng.integers(0, cols)
        dy, dx = rng.integers(-rows // 8, rows // 8 + 1), rng.integers(-cols // 5, cols // 5 + 1)
        for s in np.linspace(0, 1, 4 * max(abs(dy), abs(dx), 1)):
            yy, xx = int(y + s * dy + rng.normal(0, 0.6)), int(x + s * dx + rng.normal(0, 0.6))
            if 0 <= yy < rows and 0 <= xx < cols:
                img[yy, xx] = 255
    img[rng.random(img.shape) < noise] = 255
    return img


@pytest.mark.parametrize("geo", ["parity", "fullres"])
def test_lsd_on_clutter_matches_oracle(geo):
    """Jittered strokes, blobs and salt noise drive regions through refine, reduce_region_radius
    and every rect_improve stage on the 64-lane code paths (most regions end up rejected)."""
    from oracle.oracle import Oracle
    cfg = default_config(geo)
    o = Oracle(cfg)
    fe = FrontEnd(cfg, max_frames=1, max_lines_per_color=8192)
    rng = np.random.default_rng(2026)
    total = 0
    for t in range(6):
        img = _clutter(rng, fe.rows, fe.cols, 30 + 10 * t, 0.01 * (t % 3 + 1))
        if t == 5:
            img[:] = (rng.random(img.shape) < 0.3) * 255            # dense noise: huge regions, region-list spill
        ref = o.lsd(img, cap=8192)
        got = fe.lsd_binary(img)
        assert got.shape == ref.shape, (geo, t, got.shape, ref.shape)
        assert np.array_equal(got, ref), (geo, t)
        total += len(ref)
    assert total > 10
    fe.close()


def test_lsd_on_many_components(tmp_path):
    """k_lsd_label / k_lsd_grow corner cases on images made of many separate strokes (one connected component each,
    some too small to hold a region): lines must come back in the sequential detector's order.  A second process
    repeats it with the component list cut to 5 entries (LF_DIAG_COMP_CAP), which forces the "more eligible
    components than the list holds" fallback -- the whole problem grown as one component."""
    import os
    import subprocess
    import sys
    from oracle.oracle import Oracle
    cfg = default_config("fullres")
    o = Oracle(cfg)
    rng = np.random.default_rng(77)
    imgs, refs = [], []
    for t, (n_strokes, length) in enumerate([(90, 40), (160, 22), (60, 70)]):
        img = np.zeros((320, 640), np.uint8)
        for _ in range(n_strokes):
            y, x = int(rng.integers(6, 314)), int(rng.integers(6, 634 - length))
            slope = rng.uniform(-0.6, 0.6)
            for k in range(length):
                yy = int(round(y + slope * k))
                if 0 <= yy < 320:
                    img[yy, x + k] = 255
        imgs.append(img)
        refs.append(o.lsd(img, cap=8192))
    assert sum(len(r) for r in refs) > 100
    np.savez(tmp_path / "cases.npz", imgs=np.stack(imgs), **{"ref%d" % i: r for i, r in enumerate(refs)})
    script = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "from lane_slam_amd import FrontEnd, default_config\n"
        "d = np.load(%r)\n"
        "fe = FrontEnd(default_config('fullres'), max_frames=1, max_lines_per_color=8192)\n"
        "for i, img in enumerate(d['imgs']):\n"
        "    got = fe.lsd_binary(img)\n"
        "    ref = d['ref%%d' %% i]\n"
        "    assert got.shape == ref.shape and np.array_equal(got, ref), i\n"
        "print('ok')\n" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / "cases.npz")))
    for cap in (None, "5"):
        env = dict(os.environ)
        if cap:
            env["LF_DIAG_COMP_CAP"] = cap
        p = subprocess.run([sys.executable, "-c", script], capture_output=True, env=env)
        assert p.returncode == 0 and b"ok" in p.stdout, (cap, p.stderr.decode()[-800:])


def test_lbd_gradient_planes(setup):
    """gray -> 5x5 fixed-point Gaussian -> Sobel (the LBD inputs) must be integer-exact everywhere.
    (This test caught hipcc's v_ashr_pk_u8_i32 fusion producing a wrong byte on gfx950.)"""
    geo, cfg, fe, o, frames, seg = setup
    n = frames.shape[0]
    dx = fe.fetch(_lib.LF_BUF_LBD_DX, n)
    dy = fe.fetch(_lib.LF_BUF_LBD_DY, n)
    for f in range(n):
        odx, ody = o.sobel3(o.gaussian5(o.bgr2gray(o.preprocess(frames[f]))))
        assert np.array_equal(dx[f], odx) and np.array_equal(dy[f], ody), f


def test_larger_dilation_kernel():
    """dilation_kernel_size 5: the generic MORPH_ELLIPSE path of k_pre (the default 3x3 cross has its own)."""
    from oracle.oracle import Oracle
    cfg = default_config("parity")
    cfg["detector"]["dilation_kernel_size"] = 5
    fe = FrontEnd(cfg, max_frames=2, max_lines_per_color=4096)
    frames = synth.make_batch(2, 40)
    seg = fe.process_batch(frames)
    o = Oracle(cfg)
    masks = fe.fetch(_lib.LF_BUF_MASKS, 2)
    for f in range(2):
        bw = o.color_masks(o.bgr2hsv(o.preprocess(frames[f])))
        for c in range(3):
            assert np.array_equal(masks[f, c], o.dilate(bw[c]))
        r = o.process_frame(frames[f])
        assert np.array_equal(seg.frame(f).lines, r["lines"]) and np.array_equal(seg.frame(f).code, r["code"])
    fe.close()


@pytest.mark.parametrize("variant", ["scale1", "scale06", "refine0", "refine1", "bins256_ang15", "canny_swapped_hsv"])
def test_configuration_variants(variant):
    """Non-default detector / LSD parameters take the generic code paths (no Gaussian at scale 1, an
    11-tap... 9-tap row filter without the lookup table at scale 0.6, other refine levels, bins, tolerances)."""
    from oracle.oracle import Oracle
    cfg = default_config("parity")
    if variant == "scale1":
        cfg["lsd"]["scale"] = 1.0
    elif variant == "scale06":
        cfg["lsd"]["scale"] = 0.6
    elif variant == "refine0":
        cfg["lsd"]["refine"] = 0
    elif variant == "refine1":
        cfg["lsd"]["refine"] = 1
    elif variant == "bins256_ang15":
        cfg["lsd"]["n_bins"] = 256
        cfg["lsd"]["ang_th"] = 15.0
        cfg["lsd"]["density_th"] = 0.8
        cfg["lsd"]["log_eps"] = 1.0
    else:
        cfg["detector"]["canny_thresholds"] = [150, 60]            # OpenCV swaps them
        cfg["detector"]["hsv_white1"] = [0, 0, 120]
        cfg["detector"]["hsv_yellow1"] = [20, 100, 80]
        cfg["detector"]["hsv_red4"] = [179, 255, 250]
    fe = FrontEnd(cfg, max_frames=3, max_lines_per_color=4096)
    o = Oracle(cfg)
    frames = synth.make_batch(3, 60)
    seg = fe.process_batch(frames)
    total = 0
    for f in range(3):
        r = o.process_frame(frames[f])
        s = seg.frame(f)
        assert s.n == r["n"], (variant, f, s.n, r["n"])
        assert np.array_equal(s.lines, r["lines"]) and np.array_equal(s.normals, r["normals"])
        assert np.array_equal(s.ground, r["ground"]) and np.array_equal(s.keep, r["keep"])
        assert np.array_equal(s.code, r["code"])
        total += s.n
    assert total > 5
    fe.close()


@pytest.mark.gpu
def test_associate_large_map_and_many_ties():
    """Maps beyond 4096 column blocks per workgroup chunk (the in-accumulator block counter has 12 bits, so the
    launcher must split them) and heavy duplication: ties must resolve to the lowest map index everywhere."""
    from oracle.oracle import Oracle
    o = Oracle(default_config("parity"))
    fe = FrontEnd(default_config("parity"))
    rng = np.random.default_rng(23)
    nm, nq = 150001, 260
    base = rng.integers(0, 256, (500, 32), dtype=np.uint8)
    m = base[rng.integers(0, 500, nm)]                      # every code occurs ~300 times
    q = base[rng.integers(0, 500, nq)].copy()
    q[::3, 5] ^= 0x11                                        # a third of the queries two bits away
    idx, dist = _associate_both_rules(fe, o, q, m)
    assert (dist <= 2).all() and (np.diff(np.sort(idx)) >= 0).all()


@pytest.mark.gpu
@pytest.mark.parametrize("geo", ["fullres", "parity"])
def test_real_camera_frames_match_oracle(geo, golden_dir):
    """Three of the reference's own Duckiebot camera frames (night, day, ice rink lighting; decoded pixels in
    tests/golden/real_frames.npz, generator make_golden.py: golden_real_frames): every output field of the front end
    bit-identical to the oracle on camera images as well, and the anti-instagram clustering of their bottom strips."""
    from oracle.oracle import Oracle, kmeans as oracle_kmeans
    z = np.load(os.path.join(golden_dir, "real_frames.npz"))
    frames = np.stack([z["frame%d" % k] for k in range(3)])
    cfg = default_config(geo)
    fe = FrontEnd(cfg, max_frames=3, max_lines_per_color=1024)
    o = Oracle(cfg)
    seg = fe.process_batch(frames, describe=True)
    total = 0
    for f in range(3):
        r = o.process_frame(frames[f], cap=3 * 4096)
        s = seg.frame(f)
        assert s.n == r["n"], (f, s.n, r["n"])
        total += s.n
        for k in ("lines", "normals", "color", "pixels_normalized", "ground", "keep", "code"):
            assert np.array_equal(getattr(s, k), r[k]), (f, k)
        assert np.array_equal(s.desc, r["desc"], equal_nan=True)
        pts = np.ascontiguousarray(frames[f][-100:].reshape(-1, 3))
        for init in ([[60, 60, 60], [50, 240, 240], [240, 240, 240]], [[60, 60, 60], [60, 60, 240], [50, 240, 240], [240, 240, 240]]):
            got, want = fe.kmeans(pts, init), oracle_kmeans(pts, init)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2] == want[2]
    assert total > 30


@pytest.mark.gpu
def test_associate_odd_sizes_match_matcher_semantics():
    """Query and map counts around every granule of the associator (32-row blocks, 64-row tiles, 256-query workgroups,
    16 384-row chunks) and codes at distance 0 / 128 / 129 / 256: exact nearest neighbour, lowest index on ties, nothing
    beyond 128 -- on the FP4 kernel (ungated, the default) through lf_associate."""
    from oracle.oracle import Oracle
    o = Oracle(default_config("parity"))
    fe = FrontEnd(default_config("parity"))
    rng = np.random.default_rng(77)
    for nq, nm in ((1, 1), (31, 63), (255, 64), (256, 65), (257, 1000), (1000, 127), (300, 16385), (64, 40000), (513, 33000)):
        m = rng.integers(0, 256, (nm, 32), dtype=np.uint8)
        q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
        # some queries are copies of map rows with 0, 128, 129 or 256 bits flipped
        for i in range(0, nq, 5):
            src = m[rng.integers(0, nm)].copy()
            nflip = (0, 128, 129, 256)[(i // 5) % 4]
            bits = rng.permutation(256)[:nflip]
            for b in bits:
                src[b >> 3] ^= 1 << (b & 7)
            q[i] = src
        _associate_both_rules(fe, o, q, m)


@pytest.mark.gpu
def test_stream_handle_and_async_errors():
    fe = FrontEnd(default_config("parity"), max_frames=2, max_lines_per_color=64)
    assert fe.stream_ptr() != 0
    other = FrontEnd(default_config("parity"), max_frames=2, max_lines_per_color=64)
    assert other.stream_ptr() != fe.stream_ptr()            # one stream per handle
    assert fe.wait() == 0                                    # nothing queued: not an error
    with pytest.raises(LanefrontError):
        fe.submit_device(0, 1, {}, 0)                        # null frames


@pytest.mark.gpu
def test_hysteresis_strips_follow_a_serpentine_weak_chain():
    """1920x1080: a low-contrast zig-zag bar (weak edges only) that climbs and descends the whole working image
    several times; only its first few pixels have high contrast (the only strong edges).  Hysteresis has to carry the
    strong label along the chain across every LDS strip, in both directions, repeatedly."""
    from oracle.oracle import Oracle
    cfg = default_config("fullres", in_size=(1080, 1920))
    o = Oracle(cfg)
    fe = FrontEnd(cfg, max_frames=1, max_lines_per_color=4096)
    frame = np.full((1080, 1920, 3), 70, np.uint8)
    top, bot = 390, 1050
    mid, amp = (top + bot) / 2.0, (bot - top) / 2.0
    for t in np.linspace(0.0, 1.0, 40000):                    # a sine wave, five periods across the frame
        cx = 60 + t * 1760
        cy = mid + amp * np.cos(2 * np.pi * 5 * t)
        # +26 grey levels: weak edges only -- except the first 18 px of the band, the only strong edges of the frame
        frame[int(round(cy)) - 2:int(round(cy)) + 3, int(round(cx)) - 2:int(round(cx)) + 3] = 230 if t < 0.01 else 96
    fe.process_batch(frame[None])
    edges = fe.fetch(_lib.LF_BUF_EDGES, 1)[0]
    work = o.preprocess(frame)
    ref = o.canny(work)
    assert np.array_equal(edges, ref)
    # the chain really is promoted end to end (not only near the seed), i.e. hysteresis did the work
    cols = np.nonzero(ref.any(axis=0))[0]
    assert cols.min() < 100 and cols.max() > 1700 and ref.sum() // 255 > 10000
    head = frame.copy()
    head[:, 100:] = 70                                        # the strong head alone: a few hundred edge pixels
    assert o.canny(o.preprocess(head)).sum() // 255 < 400


def test_knn_and_radius_match_follow_the_matcher_semantics():
    """knnMatch / radiusMatch (binary_descriptor_matcher.cpp:258-335, 428-504): the nearest map codes within 128 bits,
    nearest first, index order among equals; against the oracle's brute force and its own 1-NN."""
    from oracle.oracle import Oracle
    o = Oracle(default_config("parity"))
    fe = FrontEnd(default_config("parity"))
    rng = np.random.default_rng(23)
    nm, nq = 2100, 530                                     # crosses the 256-code tiles and the 256-query workgroups
    m = rng.integers(0, 256, (nm, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    for i in range(300):                                   # planted neighbours at 0..128 bits, in groups (ties)
        src = q[i % 90].copy()
        for b in rng.choice(256, size=int(rng.integers(0, 129)) if i % 3 else (i % 40), replace=False):
            src[b >> 3] ^= np.uint8(1 << (b & 7))
        m[int(rng.integers(0, nm))] = src
    m[5] = m[900] = m[1700] = q[7]                         # exact duplicates: index order
    # the reference's order among equally near codes (the default), then the index order
    for k in (1, 3, 16):
        gi, gd = fe.knn_match(q, m, k)
        oi, od = o.knn_match(q, m, k, tie_rule="mihasher")
        assert np.array_equal(gi, oi) and np.array_equal(gd, od), k
    wi, wd, _ = o.match_mih(q, m)
    assert np.array_equal(gi[:, 0], wi) and np.array_equal(gd[:, 0], wd)       # k-NN's first column = match()
    for r in (0.0, 17.5, 64.0, 128.0):
        go, gi2, gd2 = fe.radius_match(q, m, r)
        oo, oi2, od2 = o.radius_match(q, m, r, tie_rule="mihasher")
        assert np.array_equal(go, oo) and np.array_equal(gi2, oi2) and np.array_equal(gd2, od2), r
    fe.set_tie_rule("lowest")
    for k in (1, 3, 16):
        gi, gd = fe.knn_match(q, m, k)
        oi, od = o.knn_match(q, m, k)
        assert np.array_equal(gi, oi) and np.array_equal(gd, od), k
    i1, d1 = o.match(q, m)
    assert np.array_equal(gi[:, 0], i1) and np.array_equal(gd[:, 0], d1)       # k-NN's first column = match()
    assert list(gi[7, :3]) == [5, 900, 1700]
    far = (m[:6] ^ np.uint8(0xff))                          # 6 codes, some of them beyond 128 bits of most queries
    gi6, gd6 = fe.knn_match(q, np.concatenate([m[:3], far[3:]]), 16)
    oi6, od6 = o.knn_match(q, np.concatenate([m[:3], far[3:]]), 16)
    assert np.array_equal(gi6, oi6) and np.array_equal(gd6, od6) and (gi6 == -1).any() and (gi6[:, 6:] == -1).all()
    for r in (0.0, 17.5, 64.0, 128.0, 300.0):
        go, gi, gd = fe.radius_match(q, m, r)
        oo, oi, od = o.radius_match(q, m, r)
        assert np.array_equal(go, oo) and np.array_equal(gi, oi) and np.array_equal(gd, od), r
    assert go[-1] > nq                                     # radius >= 128 is cut at D = 128
    with pytest.raises(LanefrontError):
        fe.knn_match(q, m, 17)
    e_i, e_d = fe.knn_match(q[:3], np.zeros((0, 32), np.uint8), 2)
    assert (e_i == -1).all() and (e_d == -1).all()
    n_i, n_d = fe.knn_match(np.zeros((0, 32), np.uint8), m, 4)                 # no queries: empty results, no error
    assert n_i.shape == (0, 4) and n_d.shape == (0, 4)
    z_o, z_i, z_d = fe.radius_match(np.zeros((0, 32), np.uint8), m, 30.0)
    assert list(z_o) == [0] and z_i.size == 0 and z_d.size == 0
    z_o, z_i, z_d = fe.radius_match(q[:5], np.zeros((0, 32), np.uint8), 30.0)
    assert list(z_o) == [0] * 6 and z_i.size == 0
    fe.close()


@pytest.mark.parametrize("form", ["bit plane", "row lists", "bit plane, 2048 USED bits", "bit plane, 8192 USED bits"])
@pytest.mark.parametrize("level", ["0", "1", "2"])
def test_region_growing_slice_size_does_not_change_results(level, form):
    """k_lsd_grow's LDS slice (13 / 28 / 40 KB, normally chosen from the share of problems that overflowed it in the previous
    batch) and its form -- the defined pixels as a bit plane with running counts (k_lsd_grow_bm, the default since round 4; problems
    beyond its USED bits go to the bounded row-list code behind it) or as row lists (k_lsd_grow<MODE>) -- decide which code a
    problem takes: a matter of speed only.  Every combination, forced, on frames whose problems straddle the sizes (lane frames,
    clutter, noise), against the oracle."""
    from oracle.oracle import Oracle
    cfg = default_config("fullres")
    o = Oracle(cfg)
    frames = _frames(3, seed0=321)
    rng = np.random.default_rng(8)
    busy = frames[0].copy()
    for _ in range(260):                                      # many strokes in lane colours: problems with 6 - 12 k defined pixels
        y, x = int(rng.integers(170, 470)), int(rng.integers(10, 620))
        busy[y:y + int(rng.integers(1, 4)), x:x + int(rng.integers(6, 60))] = ((235, 235, 235), (40, 220, 235), (40, 40, 220))[int(rng.integers(0, 3))]
    frames = np.concatenate([frames, busy[None]])
    os.environ["LF_GROW_LDS_LEVEL"] = level
    os.environ["LF_GROW_BITMAP"] = "1" if form == "bit plane" else "0" if form == "row lists" else form.split()[-3]
    try:
        fe = FrontEnd(cfg, max_frames=4, max_lines_per_color=4096)
        seg = fe.process_batch(frames)
        seg2 = fe.process_batch(frames)                       # and again: nothing left over from the first batch
    finally:
        del os.environ["LF_GROW_LDS_LEVEL"]
        del os.environ["LF_GROW_BITMAP"]
    for f in range(4):
        r = o.process_frame(frames[f], cap=3 * 4096)
        for sg in (seg, seg2):
            s = sg.frame(f)
            assert s.n == r["n"], (level, form, f, s.n, r["n"])
            assert np.array_equal(s.lines, r["lines"]) and np.array_equal(s.color, r["color"]) and np.array_equal(s.keep, r["keep"])
    fe.close()


def test_region_growing_slice_follows_the_workload():
    """Without the override the handle moves between the slice sizes by what the last batch needed -- and gives the same
    segments whichever it is on: busy frames, then lane frames, then busy frames again."""
    from oracle.oracle import Oracle
    cfg = default_config("fullres")
    o = Oracle(cfg)
    lane = synth.make_batch(2, seed0=11)
    rng = np.random.default_rng(9)
    busy = lane.copy()
    for img in busy:
        for _ in range(300):
            y, x = int(rng.integers(170, 470)), int(rng.integers(10, 620))
            img[y:y + int(rng.integers(1, 4)), x:x + int(rng.integers(6, 60))] = ((235, 235, 235), (40, 220, 235), (40, 40, 220))[int(rng.integers(0, 3))]
    fe = FrontEnd(cfg, max_frames=2, max_lines_per_color=4096)
    want = {id(lane): [o.process_frame(f, cap=3 * 4096) for f in lane], id(busy): [o.process_frame(f, cap=3 * 4096) for f in busy]}
    for batch in (busy, busy, lane, lane, busy):
        seg = fe.process_batch(batch)
        for f in range(2):
            s, r = seg.frame(f), want[id(batch)][f]
            assert s.n == r["n"] and np.array_equal(s.lines, r["lines"]) and np.array_equal(s.keep, r["keep"])
    fe.close()


@pytest.mark.parametrize("env", ["LF_ORDER_HBM", "LF_ORDER_SORT", "LF_HYST_JACOBI"])
def test_other_forms_of_the_round4_kernels(env):
    """Round 4 replaced three kernels of the batch path by forms that need a workgroup no larger than k_lsd_grow's (ordering by rank
    in a bit plane, hysteresis by column sweeps); the forms they replaced stay in the library behind process-wide switches --
    LF_ORDER_SORT (the sorting k_lsd_order), LF_HYST_JACOBI (one row per sweep) -- and k_lsd_order_bm has an HBM branch for problems
    of more than 65 535 records that no frame reaches (LF_ORDER_HBM sends every problem down it).  Each, in a process of its own,
    against the oracle: lane frames, a busy frame, 160x120 and 640x480."""
    import subprocess, sys, textwrap
    code = textwrap.dedent('''
        import numpy as np
        from lane_slam_amd import FrontEnd, default_config, synth
        from oracle.oracle import Oracle
        rng = np.random.default_rng(8)
        for geometry in ("fullres", "parity"):
            cfg = default_config(geometry)
            o = Oracle(cfg)
            frames = synth.make_batch(3, seed0=321)
            busy = frames[0].copy()
            for _ in range(260):
                y, x = int(rng.integers(170, 470)), int(rng.integers(10, 620))
                busy[y:y + int(rng.integers(1, 4)), x:x + int(rng.integers(6, 60))] = ((235, 235, 235), (40, 220, 235), (40, 40, 220))[int(rng.integers(0, 3))]
            frames = np.concatenate([frames, busy[None]])
            fe = FrontEnd(cfg, max_frames=4, max_lines_per_color=4096)
            for _ in range(2):
                seg = fe.process_batch(frames, describe=True)
                for f in range(4):
                    r = o.process_frame(frames[f], cap=3 * 4096)
                    s = seg.frame(f)
                    assert s.n == r["n"], (geometry, f, s.n, r["n"])
                    assert np.array_equal(s.lines, r["lines"]) and np.array_equal(s.color, r["color"]) and np.array_equal(s.code, r["code"])
            fe.close()
        print("same")
    ''')
    envd = dict(os.environ)
    envd[env] = "1"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=envd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("same"), out.stdout[-2000:] + out.stderr[-4000:]
