"""Seeded synthetic lane frames.

The reference's rosbag is not in its tree (README.md:53 links to a download), so
benchmarks and parity tests use frames rendered from a ground-plane lane model
through the inverse of the reference's default homography
(src/duckietown/include/calibrations/camera_extrinsic/default.yaml:1): asphalt
N(70,8), a dashed yellow centre line, solid white edge lines, an optional red stop
line, Gaussian pixel noise.  Marking colours sit inside the HSV boxes of
default.yaml:16-23; widths follow line_sanity_node.py:17-19.
"""
import numpy as np

from .config import DEFAULT_HOMOGRAPHY

YELLOW_BGR = (40, 220, 235)
WHITE_BGR = (235, 235, 235)
RED_BGR = (40, 40, 220)


def _ground_coords(rows, cols, H):
    v, u = np.mgrid[0:rows, 0:cols].astype(np.float64)
    # the homography is calibrated for 640x480 pixels; scale other sizes onto it
    su, sv = 640.0 / cols, 480.0 / rows
    uu, vv = u * su, v * sv
    g0 = H[0] * uu + H[1] * vv + H[2]
    g1 = H[3] * uu + H[4] * vv + H[5]
    g2 = H[6] * uu + H[7] * vv + H[8]
    with np.errstate(divide="ignore", invalid="ignore"):
        X = g0 / g2
        Y = g1 / g2
    return X, Y


_cache = {}


def make_frame(seed, rows=480, cols=640, H=None):
    """One BGR uint8 frame, deterministic in `seed`."""
    H = DEFAULT_HOMOGRAPHY if H is None else H
    key = (rows, cols, tuple(H))
    if key not in _cache:
        _cache[key] = _ground_coords(rows, cols, H)
    X, Y = _cache[key]
    rng = np.random.default_rng(seed)
    img = rng.normal(70.0, 8.0, size=(rows, cols, 3))
    empty = rng.random() < 0.05
    d = rng.uniform(-0.10, 0.10)
    phi = rng.uniform(-0.4, 0.4)
    has_red = rng.random() < 0.3
    s_red = rng.uniform(0.25, 0.6)
    dash_phase = rng.uniform(0.0, 0.06)
    if not empty:
        ground = (X > 0.08) & (X < 1.5) & np.isfinite(X) & np.isfinite(Y)
        c, s = np.cos(phi), np.sin(phi)
        S = c * X + s * Y
        T = -s * X + c * Y + d
        W = 0.23
        yellow = ground & (T >= W / 2) & (T <= W / 2 + 0.025) & (((S + dash_phase) % 0.06) < 0.035)
        white_r = ground & (T <= -W / 2) & (T >= -W / 2 - 0.05)
        white_l = ground & (T >= W / 2 + 0.025 + W) & (T <= W / 2 + 0.025 + W + 0.05)
        img[yellow] = YELLOW_BGR
        img[white_r | white_l] = WHITE_BGR
        if has_red:
            red = ground & (S >= s_red) & (S <= s_red + 0.05) & (T > -W / 2) & (T < W / 2)
            img[red] = RED_BGR
        marked = yellow | white_r | white_l
        if has_red:
            marked |= red
        noise = rng.normal(0.0, 3.0, size=(rows, cols, 3))
        img = np.where(marked[..., None], img + noise, img)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def make_batch(n, seed0=0, rows=480, cols=640, threads=1):
    """(n, rows, cols, 3) uint8, seeds seed0 .. seed0+n-1 (frame i depends on its seed only; threads > 1 just
    renders several frames at once)."""
    out = np.empty((n, rows, cols, 3), dtype=np.uint8)
    if n and threads > 1:
        make_frame(seed0, rows, cols)          # fill the ground-coordinate cache before the threads start
        from concurrent.futures import ThreadPoolExecutor

        def work(i):
            out[i] = make_frame(seed0 + i, rows, cols)
        with ThreadPoolExecutor(threads) as ex:
            list(ex.map(work, range(n)))
        return out
    for i in range(n):
        out[i] = make_frame(seed0 + i, rows, cols)
    return out


def random_codes(n, seed=1234):
    """n random 256-bit codes (config 5's live map), uint8 (n, 32)."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
