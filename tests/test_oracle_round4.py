"""CPU checks of the oracle pieces added in round 4 (test infrastructure for the GPU tests of the same round): the live map's
tie-rule switch against the matcher restatement, the std::sort helpers behind the OpenCV >= 3.2 seed order, the EDLines
detector composition of the batched path, the LSDDetectorC composition."""
import ctypes

import numpy as np

from lane_slam_amd import default_config, synth
from oracle import oracle as O


def test_map_tie_rule_switch_equals_the_matcher_restatement():
    rng = np.random.default_rng(2)
    m = rng.integers(0, 256, (1500, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (120, 32), dtype=np.uint8)
    for i in range(0, 120, 2):                               # planted ties
        d = int(rng.integers(0, 50))
        for _ in range(3):
            c = q[i].copy()
            for b in rng.choice(256, size=d, replace=False):
                c[b >> 3] ^= np.uint8(1 << (b & 7))
            m[int(rng.integers(0, 1500))] = c
    o = O.Oracle(default_config("parity"))
    wi, wd, ties = o.match_mih(q, m)
    li, ld = o.match(q, m)
    a = O.OracleMap(capacity=2048, kept_only=False, tie_rule="mihasher")
    b = O.OracleMap(capacity=2048, kept_only=False, tie_rule="lowest")
    a.seed(m); b.seed(m)
    gi, gd = a.associate(q)
    assert np.array_equal(gi, wi) and np.array_equal(gd, wd)
    bi, bd = b.associate(q)
    assert np.array_equal(bi, li) and np.array_equal(bd, ld) and (wi != li).any() and (ties > 1).sum() >= 40


def test_std_sort_helpers():
    O.build()
    lib = ctypes.CDLL(O._SO)
    lib.lfo_std_sort_keys.restype = ctypes.c_longlong
    rng = np.random.default_rng(4)

    def srt(keys):
        keys = np.ascontiguousarray(keys, np.int32)
        order = np.empty(len(keys), np.int32)
        c = lib.lfo_std_sort_keys(keys.ctypes.data_as(ctypes.c_void_p), len(keys), order.ctypes.data_as(ctypes.c_void_p))
        return order, c
    for n in (1, 16, 17, 1000):
        k = rng.integers(0, 8, n).astype(np.int32)
        o, _ = srt(k)
        assert sorted(o.tolist()) == list(range(n)) and (np.diff(k[o]) <= 0).all()          # a permutation, descending by key
        if n <= 16:                                                                          # below the threshold: plain insertion sort = stable
            assert np.array_equal(o, np.argsort(-k, kind="stable"))
    # the ordered_points path of the oracle's LSD is the same call: (H - 1) x (W - 1) raster, addresses y * W + x
    H, W = 9, 14
    bins = rng.integers(0, 5, (H - 1) * (W - 1)).astype(np.int32)
    order = np.empty((H - 1) * (W - 1), np.int32)
    lib.lfo_std_sort_seed_order(bins.ctypes.data_as(ctypes.c_void_p), H, W, order.ctypes.data_as(ctypes.c_void_p))
    o, _ = srt(bins)
    assert np.array_equal(order, (o // (W - 1)) * W + o % (W - 1))
    # the killer input drives std::sort past its depth limit (heap sort): far more comparisons than a random permutation
    for n in (200, 1024):
        k = np.empty(n, np.int32)
        lib.lfo_antiqsort_keys(n, k.ctypes.data_as(ctypes.c_void_p))
        assert sorted(k.tolist()) == list(range(n))
        _, ck = srt(k)
        _, cr = srt(rng.permutation(n))
        assert ck > 2 * cr


def test_edlines_detector_composition_and_lsd_keylines():
    cfg = default_config("fullres")
    o = O.Oracle(cfg)
    frame = synth.make_frame(3)
    r = o.process_frame_edlines(frame)
    assert r["n"] > 5 and (np.diff(r["color"].astype(int)) >= 0).all() and r["code"].shape == (r["n"], 32)
    assert r["keep"].sum() > 0 and np.isfinite(r["ground"]).all()
    assert o.process_frame_edlines(np.zeros_like(frame))["n"] == 0
    gray = o.bgr2gray(o.preprocess(frame))
    k = O.lsd_octave_keylines(gray, 3)
    per = np.bincount(k["octave"], minlength=3)
    assert k["n"] == per.sum() and per[0] > per[2] > 0
    assert np.array_equal(k["class_id"], np.arange(k["n"]))                                   # class ids count through the octaves
    assert np.allclose(k["start_end"], k["in_octave"] * (2.0 ** k["octave"])[:, None])
    assert (k["num_pixels"] >= 1).all() and np.allclose(np.linalg.norm(k["desc"], axis=1), 1, atol=1e-5)
    k32 = O.lsd_octave_keylines(gray, 1, seed_order="opencv32")
    assert k32["n"] > 0
