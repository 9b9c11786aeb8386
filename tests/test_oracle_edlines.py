"""Known-answer tests of the oracle's EDLines / multi-octave KeyLine restatement (oracle/lf_oracle_edlines.c; SURVEY
8f-4).  The reference's C++ (binary_descriptor_custom.cpp:689-1024, 1374-2751) cannot be built here (OpenCV headers),
so these are analytic checks that catch a wrong restatement -- PARITY UNPINNED, as for LSD and LBD."""
import numpy as np
import pytest

from oracle import oracle as O


def _half_plane(rows, cols, fn, lo=40, hi=200):
    """4 x 4 supersampled shape: a border that cuts a pixel in half gives it the middle value, so the gradient has ONE
    peak column / row (a hard step has a two-pixel plateau, which the anchor test -- a maximum by at least 8 over both
    neighbours, on odd coordinates only -- never accepts)."""
    yy, xx = np.mgrid[0:4 * rows, 0:4 * cols]
    cover = fn((xx + 0.5) / 4.0, (yy + 0.5) / 4.0).reshape(rows, 4, cols, 4).mean(axis=(1, 3))
    return np.round(lo + (hi - lo) * cover).astype(np.uint8)


def test_gaussian_taps_and_blur():
    assert list(O.gaussian_taps_q8(5, 1.0)) == [14, 63, 103, 63, 14]            # what lfo_gaussian5_u8 hard-codes
    for s in (2 ** 0.5, 2.0, 2.8284271):
        t = O.gaussian_taps_q8(5, s)
        assert list(t) == list(t[::-1]) and abs(int(t.sum()) - 256) <= 3 and t[2] == t.max()
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    from lane_slam_amd import default_config
    o = O.Oracle(default_config("parity"))
    assert np.array_equal(O.gaussian_blur_u8(img, 5, 1.0), o.gaussian5(img))     # the general form contains the fixed one
    flat = np.full((20, 30), 100, np.uint8)
    for s in (1.0, 2 ** 0.5, 2.0):
        g = int(O.gaussian_taps_q8(5, s).sum())
        assert (O.gaussian_blur_u8(flat, 5, s) == ((100 * g * g + (1 << 15)) >> 16)).all()


def test_resize_and_pyrdown():
    inv = float(np.float64(np.float32(1.0)) / np.sqrt(2.0))
    flat = np.full((320, 640), 77, np.uint8)
    r = O.resize_linear_u8(flat, inv)
    assert r.shape == (226, 453) and (r == 77).all()
    ramp = np.tile(np.arange(200, dtype=np.uint8), (40, 1))
    r = O.resize_linear_u8(ramp, inv)
    assert r.shape == (28, 141) and (np.diff(r.astype(int), axis=1) >= 0).all() and np.abs(r.astype(int) - r[0]).max() <= 1   # the two >> 16 truncate
    # bilinear sample positions: destination column d reads source (d + 0.5) * sqrt(2) - 0.5
    want = (np.arange(141) + 0.5) * np.sqrt(2.0) - 0.5
    assert np.abs(r[5].astype(float) - np.clip(want, 0, 199)).max() <= 1.0
    p = O.pyrdown_u8(flat)
    assert p.shape == (160, 320) and (p == 77).all()
    p = O.pyrdown_u8(ramp)
    assert p.shape == (20, 100) and np.abs(p[3].astype(int)[2:-2] - 2 * np.arange(100)[2:-2]).max() <= 1


def test_nfa_against_the_binomial_tail():
    from scipy.stats import binom
    log_nt = 2.0 * (np.log10(640) + np.log10(320))
    for n, k in [(20, 10), (20, 20), (50, 7), (100, 30), (300, 60), (15, 0), (64, 63)]:
        got = O.ed_nfa(n, k, 0.125, log_nt)
        tail = binom.sf(k - 1, n, 0.125) if k > 0 else 1.0
        want = -np.log10(tail) - log_nt
        assert abs(got - want) <= 0.1 * abs(want) + 1e-9, (n, k, got, want)      # the series stops at 10 % error
    assert O.ed_nfa(16, 16, 0.125, log_nt) > 0 > O.ed_nfa(16, 3, 0.125, log_nt)


def test_gradient_planes_and_the_byte_quirks():
    rng = np.random.default_rng(3)
    img = O.gaussian_blur_u8(rng.integers(0, 256, (40, 60), dtype=np.uint8), 5, 1.0)
    e = O.edlines(img)
    dx, dy = e["dx"].astype(int), e["dy"].astype(int)
    s = np.abs(dx) + np.abs(dy)
    half_even = lambda v: np.where(v % 4 == 2, v // 4 + ((v // 4) & 1), (v + 2) // 4)
    assert np.array_equal(e["gwo"], half_even(s))
    assert np.array_equal(e["g"], half_even(np.where(s > 81, s, 0)))
    assert np.array_equal(e["dir"], np.where(np.abs(dx) < np.abs(dy), 255, 0))
    # anchors: column-major order, odd coordinates only
    a = np.stack([e["ax"], e["ay"]], 1).astype(int)
    assert (a % 2 == 1).all() and (np.diff(a[:, 0] * 10000 + a[:, 1]) > 0).all()


def _chains(e):
    return [np.stack([e["xcors"][s:t], e["ycors"][s:t]], 1).astype(int) for s, t in zip(e["sid"][:-1], e["sid"][1:])]


def test_chain_invariants_on_noise_and_lanes():
    from lane_slam_amd import default_config, synth
    o = O.Oracle(default_config("fullres"))
    rng = np.random.default_rng(7)
    noisy = O.gaussian_blur_u8((rng.random((90, 120)) < 0.5).astype(np.uint8) * 220, 5, 1.0)
    lane = O.gaussian_blur_u8(o.bgr2gray(o.preprocess(synth.make_batch(1, 3)[0])), 5, 1.0)
    for img in (noisy, lane):
        e = O.edlines(img)
        assert e is not None and e["n_edges"] > 0
        seen = set()
        for ch in _chains(e):
            assert len(ch) >= 15                                                 # minLineLen + 1 recorded, the anchor once
            assert (np.abs(np.diff(ch, axis=0)).max(axis=1) <= 1).all()          # a walk moves to a neighbour
            assert (e["g"][ch[:, 1], ch[:, 0]] > 0).all() and (e["edge"][ch[:, 1], ch[:, 0]] == 1).all()
            keys = set(map(tuple, ch))
            assert len(keys) == len(ch) and not (keys & seen)                    # every edge pixel belongs to one chain, once
            seen |= keys
        # the pixels of a line are a run of its chain, lines do not overlap
        allpx = np.stack([e["xcors"], e["ycors"]], 1).astype(int)
        pos = {tuple(p): i for i, p in enumerate(allpx)}
        last = -1
        for s, t in zip(e["lsid"][:-1], e["lsid"][1:]):
            idx = [pos[(int(x), int(y))] for x, y in zip(e["lx"][s:t], e["ly"][s:t])]
            assert len(idx) >= 15 and (np.diff(idx) == 1).all() and idx[0] > last
            last = idx[-1]
        # endpoints are the projections of the first / last pixel onto the fitted line
        for i in range(e["n_lines"]):
            a, b, c = e["equations"][i]
            assert abs(a * a + b * b - 1) < 1e-12
            for (px, py), (qx, qy) in (((e["lx"][e["lsid"][i]], e["ly"][e["lsid"][i]]), e["endpoints"][i][:2]),
                                       ((e["lx"][e["lsid"][i + 1] - 1], e["ly"][e["lsid"][i + 1] - 1]), e["endpoints"][i][2:])):
                assert abs(a * qx + b * qy + c) < 1e-3 and np.hypot(qx - px, qy - py) < 4.0


@pytest.mark.parametrize("kind", ["vertical", "horizontal", "diagonal", "square", "flat", "border"])
def test_edlines_on_analytic_images(kind):
    rows, cols = 120, 160
    if kind == "vertical":
        img = _half_plane(rows, cols, lambda x, y: x >= 71.5)
    elif kind == "horizontal":
        img = _half_plane(rows, cols, lambda x, y: y >= 51.5)
    elif kind == "diagonal":
        img = _half_plane(rows, cols, lambda x, y: x - y >= 20)
    elif kind == "square":
        img = _half_plane(rows, cols, lambda x, y: (x >= 41.5) & (x < 121.5) & (y >= 31.5) & (y < 91.5))
    elif kind == "flat":
        img = np.full((rows, cols), 90, np.uint8)
    else:
        img = _half_plane(rows, cols, lambda x, y: y >= 5)                       # an edge hugging the upper border
    e = O.edlines(O.gaussian_blur_u8(img, 5, 1.0))
    ep, d = e["endpoints"], e["direction"]
    if kind in ("flat", "border"):
        assert e["n_lines"] == 0                                                 # nothing / "we don't keep the border line"
        return
    if kind == "vertical":
        assert e["n_lines"] == 1 and np.abs(ep[0, [0, 2]] - 71).max() < 1.0 and abs(ep[0, 1] - ep[0, 3]) > 100
        assert abs(d[0] + np.pi / 2) < 0.02               # dark on the left of the direction: bright right -> pointing up
    elif kind == "horizontal":
        assert e["n_lines"] == 1 and np.abs(ep[0, [1, 3]] - 51).max() < 1.0 and abs(ep[0, 0] - ep[0, 2]) > 140
        assert abs(d[0]) < 0.02                           # bright below -> pointing right (+x), dark (above) on its left
    elif kind == "diagonal":
        assert e["n_lines"] >= 1
        i = int(np.argmax(np.hypot(ep[:, 0] - ep[:, 2], ep[:, 1] - ep[:, 3])))
        assert np.abs((ep[i, [0, 2]] - ep[i, [1, 3]]) - 20).max() < 1.5 and min(abs(abs(d[i]) - np.pi / 4), abs(abs(d[i]) - 3 * np.pi / 4)) < 0.05
    else:
        assert e["n_lines"] == 4
        xs, ys = np.sort(ep[:, [0, 2]].mean(1)), np.sort(ep[:, [1, 3]].mean(1))
        assert abs(xs[0] - 41) < 1.5 and abs(xs[-1] - 121) < 1.5 and abs(ys[0] - 31) < 1.5 and abs(ys[-1] - 91) < 1.5
        assert sorted(np.round(d / (np.pi / 2)).astype(int).tolist()) in ([-2, -1, 0, 1], [-1, 0, 1, 2])    # four directions, one turn


def test_octave_keylines_structure_and_descriptors():
    img = _half_plane(240, 320, lambda x, y: (x >= 61.5) & (x < 251.5) & (y >= 51.5) & (y < 201.5))
    k = O.octave_keylines(img, 3)
    assert k["octave_size"] == [(240, 320), (170, 226), (120, 160)] and k["n"] == sum(k["octave_lines"]) > 8
    # detectImpl's order: class id, then octave; a class holds one line per octave at most
    key = k["class_id"].astype(int) * 8 + k["octave"]
    assert (np.diff(key) > 0).all()
    # the four sides are found in every octave and grouped: 4 classes with 3 members
    cls, cnt = np.unique(k["class_id"], return_counts=True)
    assert (cnt == 3).sum() >= 4
    scale = np.array([1.0, np.float32(np.sqrt(2.0)), np.float32(np.sqrt(2.0) * np.float32(np.sqrt(2.0)))], np.float32)
    assert np.array_equal(k["start_end"], k["in_octave"] * scale[k["octave"]][:, None])
    for c in cls[cnt == 3]:
        m = k["class_id"] == c
        mid = (k["start_end"][m][:, :2] + k["start_end"][m][:, 2:]) / 2
        assert np.abs(mid - mid[0]).max() < 6 and np.ptp(k["angle"][m]) < 0.1        # the same side of the square
    n = np.linalg.norm(k["desc"], axis=1)
    assert np.abs(n - 1).max() < 1e-5 and (k["desc"] >= 0).all()
    assert (k["num_pixels"] >= 15).all()
    assert np.allclose(k["response"], k["line_length"] / np.array([320, 226, 160])[k["octave"]])
    # one octave = the first octave of three
    k1 = O.octave_keylines(img, 1)
    m = k["octave"] == 0
    assert k1["n"] == m.sum() and np.array_equal(k1["in_octave"], k["in_octave"][m]) and np.array_equal(k1["code"], k["code"][m])
    # compute-only path on the pyrDown pyramid: octave 0 shares blur (sigma 1) and Sobel with the detection data
    d, c = O.describe_keylines(img, k["in_octave"], k["angle"], k["num_pixels"], k["octave"])
    assert np.array_equal(d[m], k["desc"][m]) and np.array_equal(c[m], k["code"][m])
    # (the detector's octaves shrink by sqrt 2, the compute-only pyramid by reductionRatio = 2: KeyLines of octave >= 1
    # fall outside the smaller images -- the reference's two pyramids do not fit each other; not asserted)
    # KeyLines that DO live on the pyrDown pyramid (as LSDDetector_custom.cpp:130-215 makes them, scale 2): the octave-0
    # lines at half and quarter size
    io = np.concatenate([k["in_octave"][m] / (1 << o) for o in range(3)])
    ang = np.tile(k["angle"][m], 3)
    npx = np.concatenate([np.maximum(k["num_pixels"][m] >> o, 1) for o in range(3)])
    octv = np.repeat(np.arange(3, dtype=np.int32), int(m.sum()))
    d, c = O.describe_keylines(img, io, ang, npx, octv)
    assert np.abs(np.linalg.norm(d, axis=1) - 1).max() < 1e-5 and np.array_equal(d[:int(m.sum())], k["desc"][m])
    # the same side of the square looks alike in every octave
    nm = int(m.sum())
    bits = np.unpackbits(c[:nm] ^ c[nm:2 * nm], axis=1).sum(1)
    assert np.median(bits) < 64


def test_real_camera_frames_are_plausible(golden_dir):
    import os
    from lane_slam_amd import default_config
    real = np.load(os.path.join(golden_dir, "real_frames.npz"))
    o = O.Oracle(default_config("fullres"))
    for name in real.files:
        if real[name].ndim != 3:
            continue
        gray = o.bgr2gray(o.preprocess(real[name]))
        k = O.octave_keylines(gray, 2)
        assert k is not None and 3 <= k["n"] <= 400, (name, k and k["n"])
        assert (k["octave"] == 1).any() and np.abs(np.linalg.norm(k["desc"], axis=1) - 1).max() < 1e-5
