"""CPU check of the product's LSD grow logic: lane_slam_amd/csrc/lsd_grow.h compiled for the
host with one lane (tests/hostsim) must reproduce the oracle's segments bit for bit.  This
exercises the control flow of the HIP kernel without a GPU; the wave-parallel forms of the
same routines are covered by the -m gpu tests."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from lane_slam_amd import default_config as _default_config, synth


def default_config(*a, **k):
    """The harness hands the growing code its seeds in raster order inside a bin (the OpenCV 3.0 order: the growing logic under test
    does not depend on which order it is given) -- so the oracle it is compared with runs that order too."""
    cfg = _default_config(*a, **k)
    cfg["lsd"]["seed_order"] = "opencv30"
    return cfg


@pytest.fixture(scope="module")
def oracle_parity():
    from oracle.oracle import Oracle
    return Oracle(default_config("parity"))


@pytest.fixture(scope="module")
def oracle_fullres():
    from oracle.oracle import Oracle
    return Oracle(default_config("fullres"))

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hostsim():
    so = os.path.join(HERE, "hostsim", "_build", "libhostsim.so")
    src = os.path.join(HERE, "hostsim", "hostsim.cpp")
    hdr = os.path.join(HERE, "..", "lane_slam_amd", "csrc", "lsd_grow.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", so, src])
    lib = ctypes.CDLL(so)
    lib.hs_lsd_detect.restype = ctypes.c_int
    lib.hs_lsd_detect_ex.restype = ctypes.c_int
    return lib


def _run(hs, o, ec, reg_lds, by_components=False, n_components=None):
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    scaled = o.lsd_scaled_image(ec)
    H, W = scaled.shape
    rho, prec, p, lognt = (ctypes.c_double() for _ in range(4))
    mrs = ctypes.c_int()
    lsd = o.cfg["lsd"]
    hs.hs_lsd_params(ctypes.c_double(lsd["ang_th"]), ctypes.c_double(lsd["quant"]), H, W, ctypes.byref(rho),
                     ctypes.byref(prec), ctypes.byref(p), ctypes.byref(lognt), ctypes.byref(mrs))
    lines = np.zeros((4096, 4), np.float32)
    nc = ctypes.c_int()
    n = hs.hs_lsd_detect_ex(P(scaled), H, W, rho, prec, p, lognt, ctypes.c_double(lsd["log_eps"]),
                            ctypes.c_double(lsd["density_th"]), ctypes.c_double(lsd["scale"]), mrs, lsd["refine"],
                            lsd["n_bins"], P(lines), 4096, reg_lds, int(by_components), ctypes.byref(nc))
    if n_components is not None:
        n_components.append(nc.value)
    return lines[:n]


def test_product_lsd_logic_matches_oracle(hostsim, oracle_parity):
    o = oracle_parity
    total = 0
    for seed in range(10):             # seed 7 holds a thin rectangle that stops at rect_improve's last-stage width guard
        bgr = o.preprocess(synth.make_frame(seed))
        bw = o.color_masks(o.bgr2hsv(bgr))
        edges = o.canny(bgr)
        for c in range(3):
            ec = o.dilate(bw[c]) & edges
            ref = o.lsd(ec)
            for reg_lds in (0, 37, 1 << 20):       # region list fully global / split / fully "LDS"
                got = _run(hostsim, o, ec, reg_lds)
                assert got.shape == ref.shape and np.array_equal(got, ref), (seed, c, reg_lds)
            total += len(ref)
    assert total > 50


def test_connected_components_are_independent_subproblems(hostsim, oracle_parity, oracle_fullres):
    """What k_lsd_label / k_lsd_grow rely on: growing every connected component of the defined pixels on its own
    (largest first, its seeds in global order) and merging the lines by seed position reproduces the sequential
    detector bit for bit -- on lane frames (a few components each) and on clutter (dozens, most of them too small to
    matter)."""
    ncs = []
    for o, seeds in ((oracle_parity, range(6)), (oracle_fullres, range(2))):
        for seed in seeds:
            bgr = o.preprocess(synth.make_frame(seed))
            bw = o.color_masks(o.bgr2hsv(bgr))
            edges = o.canny(bgr)
            for c in range(3):
                ec = o.dilate(bw[c]) & edges
                ref = o.lsd(ec)
                got = _run(hostsim, o, ec, 64, by_components=True, n_components=ncs)
                assert got.shape == ref.shape and np.array_equal(got, ref), (seed, c)
    rng = np.random.default_rng(5)
    o = oracle_parity
    for t in range(4):
        img = np.zeros((80, 160), np.uint8)
        for _ in range(25):
            y, x = rng.integers(0, 80), rng.integers(0, 160)
            dy, dx = rng.integers(-6, 7), rng.integers(-25, 26)
            for s in np.linspace(0, 1, 60):
                yy, xx = int(y + s * dy + rng.normal(0, 0.6)), int(x + s * dx + rng.normal(0, 0.6))
                if 0 <= yy < 80 and 0 <= xx < 160:
                    img[yy, xx] = 255
        img[rng.random(img.shape) < 0.02] = 255
        ref = o.lsd(img)
        got = _run(hostsim, o, img, 64, by_components=True, n_components=ncs)
        assert got.shape == ref.shape and np.array_equal(got, ref), t
    assert max(ncs) >= 3 and len(set(ncs)) > 2


def test_product_lsd_logic_on_clutter(hostsim, oracle_parity):
    """Random blobs and noise push regions through refine / reduce_region_radius / every
    rect_improve stage (most regions are rejected by the NFA test)."""
    o = oracle_parity
    rng = np.random.default_rng(42)
    hit = 0
    for t in range(6):
        img = np.zeros((80, 160), np.uint8)
        for _ in range(30):
            y, x = rng.integers(0, 80), rng.integers(0, 160)
            dy, dx = rng.integers(-6, 7), rng.integers(-25, 26)
            for s in np.linspace(0, 1, 60):
                yy, xx = int(y + s * dy + rng.normal(0, 0.6)), int(x + s * dx + rng.normal(0, 0.6))
                if 0 <= yy < 80 and 0 <= xx < 160:
                    img[yy, xx] = 255
        img[rng.random(img.shape) < 0.02] = 255
        ref = o.lsd(img)
        got = _run(hostsim, o, img, 64)
        assert got.shape == ref.shape and np.array_equal(got, ref), t
        hit += len(ref)
    assert hit > 10
