"""The LSD seed order of the OpenCV the reference actually runs on (VERDICT r3 #1b): cv2's LSD behind
/root/reference/src/line_detector/include/line_detector/line_detector_lsd.py:64-72 on ROS Kinetic = OpenCV 3.3.1 orders the seeds
with std::sort(ordered_points, compare_norm) over EVERY pixel of the gradient image; inside a bin the order is what libstdc++'s
introsort leaves.  `cfg["lsd"]["seed_order"] = "opencv32"` (lf_config.lsd_seed_order = LF_LSD_SEED_OPENCV32) reproduces that
order on the device (k_lsd_seed32.hip); checked (1) as a sort, against the REAL std::sort of this image's libstdc++
(oracle/lf_oracle_sort.cpp) on key arrays of every shape including a killer input that drives std::sort into its heap-sort
fallback, and (2) end to end against the oracle with the same switch: every segment, bit for bit."""
import ctypes

import numpy as np
import pytest

from lane_slam_amd import FrontEnd, LanefrontError, default_config, synth

pytestmark = pytest.mark.gpu


def _oracle_lib():
    from oracle import oracle as O
    O.build()
    lib = ctypes.CDLL(O._SO)
    lib.lfo_std_sort_keys.restype = ctypes.c_longlong
    return lib


def _std_sort(lib, keys):
    keys = np.ascontiguousarray(keys, np.int32)
    order = np.empty(keys.shape[0], np.int32)
    ncmp = lib.lfo_std_sort_keys(keys.ctypes.data_as(ctypes.c_void_p), keys.shape[0], order.ctypes.data_as(ctypes.c_void_p))
    return order, ncmp


def _gpu_sort(fe, keys):
    keys = np.ascontiguousarray(keys, np.int32)
    order = np.empty(keys.shape[0], np.int32)
    fe._check(fe.lib.lf_debug_std_sort(fe.h, keys.ctypes.data, keys.shape[0], order.ctypes.data))
    return order


def _killer(lib, n):
    k = np.empty(n, np.int32)
    lib.lfo_antiqsort_keys(n, k.ctypes.data_as(ctypes.c_void_p))
    return k


def test_sort_emulation_equals_libstdcxx_std_sort():
    lib = _oracle_lib()
    fe = FrontEnd(default_config("parity"))
    rng = np.random.default_rng(5)
    cases = []
    for n in (1, 2, 3, 15, 16, 17, 18, 33, 64, 65, 100, 1000, 2047, 2048, 2049, 2050, 4097, 8001, 20000):
        cases.append(("uniform", rng.integers(0, 1024, n)))
        cases.append(("few", rng.integers(0, 4, n)))
        z = np.zeros(n, np.int64)
        m = rng.random(n) < 0.03
        z[m] = rng.integers(1, 1024, int(m.sum()))
        cases.append(("mostly zero", z))
    for n in (17, 500, 5000):
        cases.append(("equal", np.full(n, 7)))
        cases.append(("ascending", np.sort(rng.integers(0, 1024, n))))
        cases.append(("descending", np.sort(rng.integers(0, 1024, n))[::-1]))
        cases.append(("organ pipe", np.minimum(np.arange(n), n - 1 - np.arange(n)) % 1024))
    # what the detector really sorts: a 511 x 255 gradient image that is flat almost everywhere
    img = np.zeros((255, 511), np.int64)
    for _ in range(40):
        y, x = int(rng.integers(0, 250)), int(rng.integers(0, 480))
        img[y:y + 3, x:x + int(rng.integers(5, 30))] = rng.integers(1, 1024, 1)[0]
    img[rng.random(img.shape) < 0.01] = 3
    cases.append(("gradient image", img.reshape(-1)))
    cases.append(("131071 uniform", rng.integers(0, 1024, 131071)))
    # killer inputs (McIlroy's adversary played against this libstdc++): std::sort uses up its depth limit and heap sorts
    for n in (200, 1000, 1024):
        k = _killer(lib, n)
        _, ncmp = _std_sort(lib, k)
        _, nrand = _std_sort(lib, rng.permutation(n))
        assert ncmp > 2 * nrand, (n, ncmp, nrand)             # the input does bite
        cases.append(("killer %d" % n, k))
        cases.append(("killer %d + tail" % n, np.concatenate([k, rng.integers(0, 1024, 300)])))
        cases.append(("killer x 3", np.concatenate([k, k, k])))
    for name, keys in cases:
        keys = np.asarray(keys, np.int64)
        # (a) as the detector sorts: elements with key 0 are its flat pixels -- never seeds, anonymous on the device (the sparse
        # form keeps only the non-zero keys) -- so the order of the OTHERS is what is compared; (b) every key raised by one: no
        # element is anonymous, the whole order is compared
        want, _ = _std_sort(lib, keys)
        got = _gpu_sort(fe, keys)
        nz = int((keys != 0).sum())
        assert np.array_equal(got[:nz], want[keys[want] != 0]), (name, len(keys), int((got[:nz] != want[keys[want] != 0]).sum()))
        assert np.array_equal(np.sort(got[nz:]), np.flatnonzero(keys == 0)), name
        up = np.minimum(keys + 1, 1023)
        want, _ = _std_sort(lib, up)
        got = _gpu_sort(fe, up)
        assert np.array_equal(got, want), (name + " + 1", len(keys), int((got != want).sum()))
    with pytest.raises(LanefrontError):
        _gpu_sort(fe, np.array([5, 1024], np.int32))
    fe.close()


@pytest.mark.parametrize("geometry,n", [("parity", 30), ("fullres", 6)])
def test_segments_under_the_opencv32_seed_order(geometry, n):
    """The 36 frames of the census in tests/test_parity_deviations.py (12 of them change with the seed order) + clutter: the
    device with lsd.seed_order = opencv32 against the oracle running the real std::sort; and the two orders do differ."""
    from oracle.oracle import Oracle
    cfg = default_config(geometry)
    cfg["lsd"]["seed_order"] = "opencv32"
    cfg30 = default_config(geometry)
    cfg30["lsd"]["seed_order"] = "opencv30"                  # the A/B option (the default is opencv32 since round 5)
    frames = synth.make_batch(n, seed0=700)
    rng = np.random.default_rng(77)
    busy = frames[0].copy()
    for _ in range(200):                                      # strokes in lane colours: many bins, crowded seeds
        y, x = int(rng.integers(170, 470)), int(rng.integers(10, 620))
        busy[y:y + int(rng.integers(1, 4)), x:x + int(rng.integers(6, 60))] = ((235, 235, 235), (40, 220, 235), (40, 40, 220))[int(rng.integers(0, 3))]
    frames = np.concatenate([frames, busy[None]])
    B = frames.shape[0]
    o, o30 = Oracle(cfg), Oracle(cfg30)
    fe, fe30 = FrontEnd(cfg, max_frames=B, max_lines_per_color=2048), FrontEnd(cfg30, max_frames=B, max_lines_per_color=2048)
    seg, seg30 = fe.process_batch(frames, describe=True), fe30.process_batch(frames, describe=True)
    seg_again = fe.process_batch(frames, describe=True)
    differ = 0
    for f in range(B):
        r, r30 = o.process_frame(frames[f], cap=3 * 2048), o30.process_frame(frames[f], cap=3 * 2048)
        for sg in (seg, seg_again):
            s = sg.frame(f)
            assert s.n == r["n"], (f, s.n, r["n"])
            assert np.array_equal(s.lines, r["lines"]) and np.array_equal(s.normals, r["normals"]) and np.array_equal(s.color, r["color"])
            assert np.array_equal(s.keep, r["keep"]) and np.array_equal(s.ground, r["ground"]) and np.array_equal(s.code, r["code"])
        s30 = seg30.frame(f)
        assert s30.n == r30["n"] and np.array_equal(s30.lines, r30["lines"])              # the other setting is untouched
        differ += int(r["n"] != r30["n"] or not np.array_equal(r["lines"], r30["lines"]))
    print("\n%s: %d of %d frames have a different SegmentList under the two seed orders" % (geometry, differ, B))
    assert differ >= 1
    fe.close(); fe30.close()


def test_plugin_and_bad_values():
    """The one-frame plugin path (lf_set_image / lf_detect_lines) takes the same configuration; unknown values are refused."""
    from oracle.oracle import Oracle
    from lane_slam_amd import LineDetectorHIP
    cfg = default_config("parity")
    cfg["lsd"]["seed_order"] = "opencv32"
    frame = synth.make_batch(1, seed0=705)[0]
    o = Oracle(cfg)
    r = o.process_frame(frame, cap=3 * 512)
    fe = FrontEnd(cfg, max_frames=1, max_lines_per_color=512)
    s = fe.process_batch(frame[None]).frame(0)
    assert s.n == r["n"] and np.array_equal(s.lines, r["lines"])
    fe.close()
    bad = default_config("parity")
    bad["lsd"]["seed_order"] = "opencv45"
    with pytest.raises(ValueError):
        FrontEnd(bad)
    # no image-size limit (the reference's cv2 LSD has none): 1080p frames, 1536 x 576 LSD pixels -- no LDS bit plane of the
    # gradient image there, the list is ordered by counting passes and merged
    hd = default_config("fullres", in_size=(1080, 1920))
    hd["lsd"]["seed_order"] = "opencv32"
    big = synth.make_batch(2, 70, rows=1080, cols=1920)
    big[1, 400:700, 300:900] = np.random.default_rng(7).integers(0, 256, (300, 600, 3), dtype=np.uint8)     # a noisy patch
    ohd = Oracle(hd)
    fe = FrontEnd(hd, max_frames=2, max_lines_per_color=8192)
    seg = fe.process_batch(big)
    for f in range(2):
        r = ohd.process_frame(big[f], cap=3 * 8192, describe=False)
        s = seg.frame(f)
        assert s.n == r["n"] and s.n > 0 and np.array_equal(s.lines, r["lines"]), (f, s.n, r["n"])
    fe.close()
    # the plugin classes take it as a keyword beside the reference's 13 configuration keys
    from lane_slam_amd import DEFAULT_DETECTOR_CONFIGURATION
    cfg = default_config("parity")
    cfg["lsd"]["seed_order"] = "opencv32"
    o = Oracle(cfg)
    det = LineDetectorHIP(dict(DEFAULT_DETECTOR_CONFIGURATION), lsd_seed_order="opencv32")
    work = o.preprocess(frame)
    det.setImage(work)
    r = o.process_frame(frame, cap=3 * 512, describe=False)
    got = [det.detectLines(c) for c in ("white", "yellow", "red")]
    n = sum(len(d.lines) for d in got)
    assert n == r["n"] and n > 0
    cat = np.concatenate([np.asarray(d.lines, np.float32).reshape(-1, 4) for d in got])
    assert np.array_equal(cat, r["lines"])
