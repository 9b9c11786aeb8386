"""Known-answer tests for the oracle stages that cannot be pinned against the reference itself
(OpenCV / unbuilt C++; DESIGN.md section 2 "parity unpinned"): analytic truths the published
algorithms must satisfy.  SURVEY.md section 8(c) lists them: HSV / inRange truth tables against a float
reference, Canny on step / ramp images, LSD on images with analytically known segments, the
homography round trip and the undistort <-> distort self check, brute-force Hamming for the matcher.
They do not replace a comparison with a real OpenCV 3.x, but they catch a restatement that is wrong
rather than merely rounded differently.  CPU only."""
import copy

import numpy as np

from lane_slam_amd.config import default_config


# ------------------------------------------------------------------ a-2 BGR -> HSV (8 bit)
def _hsv_float_reference(bgr):
    """OpenCV's documented 8-bit formula in float64: V = max, S = 255 (V - min) / V, H = 30 * sector angle
    (H in [0, 180)), each rounded to nearest."""
    b, g, r = (bgr[..., i].astype(np.float64) for i in range(3))
    v = np.maximum(np.maximum(b, g), r)
    mn = np.minimum(np.minimum(b, g), r)
    d = v - mn
    s = np.where(v > 0, 255.0 * d / np.where(v > 0, v, 1), 0.0)
    dd = np.where(d > 0, d, 1)
    h = np.where(v == r, (g - b) / dd, np.where(v == g, 2.0 + (b - r) / dd, 4.0 + (r - g) / dd))
    h = np.where(d > 0, 30.0 * h, 0.0)
    h = np.where(h < 0, h + 180.0, h)
    return h, s, v


def test_hsv_primary_colours(oracle_parity):
    o = oracle_parity
    table = {  # BGR -> HSV, OpenCV documentation values
        (0, 0, 255): (0, 255, 255), (0, 255, 0): (60, 255, 255), (255, 0, 0): (120, 255, 255),
        (0, 255, 255): (30, 255, 255), (255, 255, 0): (90, 255, 255), (255, 0, 255): (150, 255, 255),
        (255, 255, 255): (0, 0, 255), (0, 0, 0): (0, 0, 0), (128, 128, 128): (0, 0, 128),
    }
    bgr = np.array(list(table.keys()), np.uint8).reshape(1, -1, 3)
    hsv = o.bgr2hsv(bgr).reshape(-1, 3)
    assert [tuple(int(x) for x in row) for row in hsv] == list(table.values())


def test_hsv_truth_table_all_triples(oracle_parity):
    """All 256^3 BGR triples: V exact, S and H within one count of the float formula (OpenCV's 8-bit path
    uses 12-bit fixed-point reciprocals, so +-1 is its documented accuracy), H always < 180."""
    o = oracle_parity
    g, r = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    worst_s = worst_h = 0
    for b in range(0, 256):
        bgr = np.stack([np.full_like(g, b), g, r], axis=-1)
        hsv = o.bgr2hsv(bgr).astype(np.int32)
        h, s, v = _hsv_float_reference(bgr)
        assert np.array_equal(hsv[..., 2], v.astype(np.int32))
        assert hsv[..., 0].max() < 180
        worst_s = max(worst_s, int(np.abs(hsv[..., 1] - s).max() > 1.0))
        dh = np.abs(hsv[..., 0] - h)
        dh = np.minimum(dh, 180.0 - dh)
        worst_h = max(worst_h, int(dh.max() > 1.0))
    assert worst_s == 0 and worst_h == 0


# ------------------------------------------------------------------ a-3 inRange / dilate
def test_inrange_is_inclusive_on_all_channels(oracle_parity):
    o = oracle_parity
    det = o.cfg["detector"]
    lo, hi = np.array(det["hsv_white1"]), np.array(det["hsv_white2"])
    rows = []
    for ch in range(3):
        for val, inside in ((lo[ch], True), (hi[ch], True), (lo[ch] - 1, False), (hi[ch] + 1, False)):
            if 0 <= val <= 255:
                px = ((lo + hi) // 2).copy()
                px[ch] = val
                rows.append((px, inside))
    hsv = np.array([p for p, _ in rows], np.uint8).reshape(1, -1, 3)
    white = o.color_masks(hsv)[0].reshape(-1)
    assert [bool(x) for x in white] == [ins for _, ins in rows]
    assert set(np.unique(white)) <= {0, 255}


def test_red_mask_is_union_of_two_ranges(oracle_parity):
    o = oracle_parity
    det = o.cfg["detector"]
    mid1 = (np.array(det["hsv_red1"]) + np.array(det["hsv_red2"])) // 2
    mid2 = (np.array(det["hsv_red3"]) + np.array(det["hsv_red4"])) // 2
    between = mid1.copy()
    between[0] = 90                                    # hue far from both red ranges
    hsv = np.array([mid1, mid2, between], np.uint8).reshape(1, 3, 3)
    assert list(o.color_masks(hsv)[2].reshape(-1)) == [255, 255, 0]


def test_dilate_3x3_ellipse_is_the_cross(oracle_parity):
    o = oracle_parity
    img = np.zeros((9, 9), np.uint8)
    img[4, 4] = 255
    out = o.dilate(img, 3)
    expect = np.zeros_like(img)
    expect[4, 3:6] = 255
    expect[3:6, 4] = 255
    assert np.array_equal(out, expect)
    corner = np.zeros((5, 5), np.uint8)
    corner[0, 0] = 255                                 # out-of-image neighbours are ignored
    out = o.dilate(corner, 3)
    assert out[0, 0] == out[0, 1] == out[1, 0] == 255 and out.sum() == 3 * 255
    assert np.array_equal(o.dilate(img, 1), img)


# ------------------------------------------------------------------ a-2 Canny
def test_canny_step_ramp_and_flat(oracle_parity):
    o = oracle_parity
    flat = np.full((40, 60, 3), 90, np.uint8)
    assert o.canny(flat).sum() == 0
    step = flat.copy()
    step[:, 30:] = 200                                  # vertical step, 110 counts: Sobel L1 = 440 > 200
    e = o.canny(step)
    cols = np.unique(np.nonzero(e)[1])
    assert set(np.unique(e)) == {0, 255}
    assert len(cols) == 1 and cols[0] in (29, 30)       # one pixel wide, at the step
    assert (e[:, cols[0]] == 255).all()                 # hysteresis links the whole column
    ramp = np.tile(np.arange(60, dtype=np.uint8)[None, :, None] * 2 + 40, (40, 1, 3))
    assert o.canny(ramp).sum() == 0                     # slope 2/px: |dx| = 16 < low threshold 80
    # the channel with the largest gradient decides: a step in one channel only is still an edge
    one = flat.copy()
    one[:, 30:, 1] = 200
    assert np.array_equal(o.canny(one), e)
    # weak edge (between the thresholds) survives only when connected to a strong one
    weak = flat.copy()
    weak[:, 30:] = 90 + 30                              # Sobel L1 = 120: weak only
    assert o.canny(weak).sum() == 0


# ------------------------------------------------------------------ a-4 LSD on known geometry
def _seg_len(l):
    return np.hypot(l[:, 2] - l[:, 0], l[:, 3] - l[:, 1])


def test_lsd_empty_and_constant(oracle_parity):
    o = oracle_parity
    assert len(o.lsd(np.zeros((80, 160), np.uint8))) == 0
    assert len(o.lsd(np.full((80, 160), 255, np.uint8))) == 0


def test_lsd_half_plane_gives_one_segment_on_the_edge(oracle_parity):
    o = oracle_parity
    img = np.zeros((80, 160), np.uint8)
    img[:, 70:] = 255                                   # vertical step edge at x = 69.5 .. 70
    lines = o.lsd(img)
    assert len(lines) == 1
    x1, y1, x2, y2 = lines[0]
    assert abs(x1 - 70) <= 1.0 and abs(x2 - 70) <= 1.0
    assert abs(y1 - y2) > 70                            # spans (almost) the full height
    img = np.zeros((80, 160), np.uint8)
    img[40:, :] = 255                                   # horizontal
    lines = o.lsd(img)
    assert len(lines) == 1
    assert abs(lines[0][1] - 40) <= 1.0 and abs(lines[0][3] - 40) <= 1.0 and abs(lines[0][0] - lines[0][2]) > 150


def test_lsd_diagonal_edge(oracle_parity):
    o = oracle_parity
    yy, xx = np.mgrid[0:80, 0:160]
    img = np.where(xx - yy > 40, 255, 0).astype(np.uint8)          # 45 degree edge x - y = 40.5
    lines = o.lsd(img)
    assert len(lines) >= 1
    long_ = lines[np.argmax(_seg_len(lines))]
    ang = np.degrees(np.arctan2(long_[3] - long_[1], long_[2] - long_[0])) % 180
    assert abs(ang - 45) < 2.0
    for (x, y) in ((long_[0], long_[1]), (long_[2], long_[3])):
        assert abs((x - y) - 40.5) / np.sqrt(2) <= 1.0              # both endpoints within a pixel of the edge
    assert _seg_len(lines).max() > 90


def test_lsd_thin_bar_gives_two_antiparallel_segments(oracle_parity):
    """A 3 px bright bar has two step edges 3 px apart with opposite gradient: LSD must keep them as two
    segments (level-line angles differ by 180 degrees), one on each side."""
    o = oracle_parity
    img = np.zeros((80, 160), np.uint8)
    img[20:60, 80:83] = 255
    lines = o.lsd(img)
    vert = lines[np.abs(lines[:, 0] - lines[:, 2]) < 1.5]
    assert len(vert) == 2
    xs = np.sort((vert[:, 0] + vert[:, 2]) / 2)
    assert abs(xs[0] - 80) <= 1.0 and abs(xs[1] - 83) <= 1.0
    d0 = np.sign(vert[0][3] - vert[0][1])
    d1 = np.sign(vert[1][3] - vert[1][1])
    assert d0 == -d1                                    # dark side convention flips the direction


# ------------------------------------------------------------------ a-6 / a-7 projection
def _oracle_with(cfg_edit):
    from oracle.oracle import Oracle
    cfg = copy.deepcopy(default_config("parity"))
    cfg_edit(cfg)
    return Oracle(cfg)


def test_normalize_lines_formula(oracle_parity):
    o = oracle_parity
    lines = np.array([[0, 0, 159, 79], [12.25, 3.5, 100.75, 60.125]], np.float32)
    cut, (h, w) = o.cfg["top_cutoff"], o.cfg["img_size"]
    expect = ((lines.astype(np.float64) + np.array([0, cut, 0, cut])) *
              np.array([1.0 / w, 1.0 / h, 1.0 / w, 1.0 / h])).astype(np.float32)
    assert np.array_equal(o.normalize_lines(lines), expect)


def test_homography_without_distortion_matches_numpy():
    """With D = 0, R = I, P = [K | 0] rectification is the identity, so ground = H [u v 1]."""
    def edit(cfg):
        cfg["D"] = [0.0] * 5
        k = cfg["K"]
        cfg["P"] = [k[0], k[1], k[2], 0.0, k[3], k[4], k[5], 0.0, 0.0, 0.0, 1.0, 0.0]
    o = _oracle_with(edit)
    rng = np.random.default_rng(5)
    pn = rng.uniform(0.05, 0.95, (64, 4)).astype(np.float32)
    got = o.ground_project(pn)
    hm = np.array(o.cfg["H"]).reshape(3, 3)
    ch, cw = o.cfg["cam_size"]
    for j in range(2):
        u = cw * pn[:, 2 * j].astype(np.float64)
        v = ch * pn[:, 2 * j + 1].astype(np.float64)
        g = hm @ np.stack([u, v, np.ones_like(u)])
        assert np.allclose(got[:, 2 * j], g[0] / g[2], rtol=0, atol=1e-9)
        assert np.allclose(got[:, 2 * j + 1], g[1] / g[2], rtol=0, atol=1e-9)


def test_undistort_inverts_the_plumb_bob_model():
    """H = I exposes the rectified pixel; pushing it back through P^-1, R^-1, the forward distortion
    model and K must land on the input pixel (5 fixed-point iterations: a few hundredths of a pixel)."""
    def edit(cfg):
        cfg["H"] = [1.0, 0, 0, 0, 1.0, 0, 0, 0, 1.0]
    o = _oracle_with(edit)
    cfg = o.cfg
    K = np.array(cfg["K"]).reshape(3, 3)
    P = np.array(cfg["P"]).reshape(3, 4)
    k1, k2, p1, p2, k3 = cfg["D"]
    ch, cw = cfg["cam_size"]

    def reprojection_error(pn):
        got = o.ground_project(pn)
        err = []
        for j in range(2):
            u0 = cw * pn[:, 2 * j].astype(np.float64)
            v0 = ch * pn[:, 2 * j + 1].astype(np.float64)
            x = (got[:, 2 * j] - P[0, 2]) / P[0, 0]
            y = (got[:, 2 * j + 1] - P[1, 2]) / P[1, 1]
            r2 = x * x + y * y
            rad = 1 + k1 * r2 + k2 * r2 ** 2 + k3 * r2 ** 3
            xd = x * rad + 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
            yd = y * rad + p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
            err.append(np.hypot(K[0, 0] * xd + K[0, 2] - u0, K[1, 1] * yd + K[1, 2] - v0))
        return np.concatenate(err)

    rng = np.random.default_rng(6)
    # centre of the image: the iteration has converged
    assert reprojection_error(rng.uniform(0.3, 0.7, (200, 4)).astype(np.float32)).max() < 0.01
    # whole image: OpenCV 3.x stops after 5 iterations whatever the residual, which leaves up to a few
    # pixels at the periphery of this wide-angle lens (k1 = -0.26); the restatement must do the same,
    # not iterate to convergence
    full = reprojection_error(rng.uniform(0.02, 0.98, (400, 4)).astype(np.float32))
    assert 0.05 < full.max() < 6.0


def test_pixel_clamp_quirk(oracle_parity):
    """vector2pixel clamps u to [0, cw-1], v < 0 to 0 and v > ch-1 to ZERO (GroundProjection.py:44-47)."""
    o = oracle_parity
    a = o.ground_project(np.array([[0.5, 1.5, 0.5, 0.0]], np.float32))[0]
    assert np.array_equal(a[0:2], a[2:4])               # v beyond the last row lands on row 0
    b = o.ground_project(np.array([[-0.2, 0.5, 0.0, 0.5]], np.float32))[0]
    assert np.array_equal(b[0:2], b[2:4])               # u below 0 clamps to 0


# ------------------------------------------------------------------ a-9 LBD
def test_lbd_descriptor_invariants(oracle_parity):
    o = oracle_parity
    rng = np.random.default_rng(11)
    gray = (rng.random((80, 160)) * 255).astype(np.uint8)
    gray[:, 80:] //= 3
    dx, dy = o.sobel3(o.gaussian5(gray))
    lines = np.array([[20, 10, 120, 60], [80, 5, 80, 75], [10, 40, 150, 40], [100, 70, 30, 20]], np.float32)
    ext, ang, npx = o.keylines(lines, 80, 160)
    assert np.allclose(ang, np.arctan2(lines[:, 3] - lines[:, 1], lines[:, 2] - lines[:, 0]), atol=1e-6)
    assert np.array_equal(npx, np.maximum(np.abs(np.rint(lines[:, 2]) - np.rint(lines[:, 0])),
                                          np.abs(np.rint(lines[:, 3]) - np.rint(lines[:, 1]))).astype(np.int32) + 1)
    desc, code = o.lbd(dx, dy, ext, ang, npx)
    assert np.allclose(np.linalg.norm(desc.astype(np.float64), axis=1), 1.0, atol=1e-5)   # final renormalisation
    assert (desc >= 0).all()                              # means of |.| sums and standard deviations
    assert code.shape == (4, 32) and len({bytes(c) for c in code}) == 4
    # reversing a segment turns the support region by 180 degrees: a different descriptor, same length
    ext2, ang2, npx2 = o.keylines(lines[:, [2, 3, 0, 1]], 80, 160)
    assert np.array_equal(npx, npx2)
    desc2, _ = o.lbd(dx, dy, ext2, ang2, npx2)
    assert np.abs(desc2 - desc).max() > 1e-3


def test_gaussian_and_sobel_on_constant_and_step(oracle_parity):
    o = oracle_parity
    flat = np.full((20, 30), 77, np.uint8)
    # OpenCV 3.0-3.3 runs the 8-bit separable filter in fixed point: cvRound(256 k) = {14, 63, 103, 63, 14},
    # which sums to 257, two passes, one rounding shift by 16 -- a constant c comes out as
    # (c 257^2 + 2^15) >> 16 (77 -> 78), the restated quirk listed in DESIGN.md section 2
    assert np.array_equal(o.gaussian5(flat), np.full_like(flat, (77 * 257 * 257 + 32768) >> 16))
    assert np.array_equal(o.gaussian5(np.zeros((8, 8), np.uint8)), np.zeros((8, 8), np.uint8))
    assert o.gaussian5(np.full((8, 8), 255, np.uint8)).min() == 255      # saturates
    dx, dy = o.sobel3(flat)
    assert not dx.any() and not dy.any()                  # reflect-101 border: no gradient on a constant
    step = flat.copy()
    step[:, 15:] = 177
    dx, dy = o.sobel3(step)
    assert not dy.any()
    assert dx[:, 14].tolist() == [400] * 20 and dx[:, 15].tolist() == [400] * 20 and not dx[:, :14].any()
    assert o.bgr2gray(np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255]]], np.uint8)).tolist() \
        == [[255, 0, 29, 150, 76]]                        # 0.114 B + 0.587 G + 0.299 R


# ------------------------------------------------------------------ a-10 matcher
def _hamming_matrix(q, t):
    x = q[:, None, :] ^ t[None, :, :]
    return np.unpackbits(x, axis=2).sum(axis=2)


def test_match_is_exact_hamming_nearest_neighbour(oracle_parity):
    o = oracle_parity
    rng = np.random.default_rng(3)
    t = rng.integers(0, 256, (400, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (60, 32), dtype=np.uint8)
    for i in range(40):                                   # planted neighbours at distance 0 .. 39
        q[i] = t[(7 * i) % 400]
        flip = rng.choice(256, size=i, replace=False)
        for f in flip:
            q[i, f // 8] ^= np.uint8(1 << (f % 8))
    idx, dist = o.match(q, t)
    hm = _hamming_matrix(q, t)
    best = hm.min(axis=1)
    found = best <= 128
    assert np.array_equal(dist[found], best[found].astype(np.float32))
    assert (hm[np.arange(60)[found], idx[found]] == best[found]).all()
    unique = found & ((hm == best[:, None]).sum(axis=1) == 1)
    assert np.array_equal(idx[unique], hm.argmin(axis=1)[unique])
    assert np.array_equal(idx[:40][unique[:40]], (7 * np.arange(40) % 400)[unique[:40]])
    assert (idx[~found] == -1).all()


def test_match_beyond_128_is_no_match(oracle_parity):
    o = oracle_parity
    t = np.zeros((3, 32), np.uint8)
    q = np.zeros((2, 32), np.uint8)
    q[0, :17] = 0xFF                                      # distance 136 to every train code
    q[1, :16] = 0xFF                                      # distance exactly 128: still returned
    idx, dist = o.match(q, t)
    assert idx[0] == -1
    assert idx[1] >= 0 and dist[1] == 128.0


# ------------------------------------------------------------------ realism (build container only)
REF_IMAGES = "/root/reference/src/anti_instagram/annotation-tool/images"


import os                                                       # noqa: E402
import pytest                                                   # noqa: E402


@pytest.mark.skipif(not os.path.isdir(REF_IMAGES), reason="the reference's real camera frames exist only in the build container")
def test_real_duckiebot_frames_give_plausible_segments(oracle_parity):
    """Not parity -- plausibility: on the reference's real 640x480 camera frames, with the reference's default
    thresholds and geometry, the restated pipeline must find lane markings of all three colours in the range
    SURVEY 8(a) expects (5-60 segments per frame at 160x120) and line sanity must keep a sensible share."""
    import glob
    from oracle.oracle import jpeg_decode
    o = oracle_parity
    files = sorted(glob.glob(os.path.join(REF_IMAGES, "*.jpg")))[::6]
    n, kept, colours = [], [], np.zeros(3, int)
    for f in files:
        r = o.process_frame(jpeg_decode(open(f, "rb").read()), cap=4096)
        n.append(r["n"])
        kept.append(int(r["keep"].sum()))
        for c in range(3):
            colours[c] += int((r["color"] == c).sum())
    assert 10 <= np.mean(n) <= 80 and min(n) >= 1
    assert (colours > 0).all() and colours[0] > colours[2]        # white dominates, red (stop lines) is rare
    assert 0.2 <= np.sum(kept) / np.sum(n) <= 0.8
