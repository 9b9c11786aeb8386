"""Anti-instagram colour clustering (k-means part): the oracle against the reference's own runKMeans
(tests/golden/kmeans.npz: kmeans.py:22-47 executed here with scikit-learn 1.7.2, tests/golden/make_golden.py), the GPU
kernel against the oracle bit for bit, and the Python mirror of the reference interface."""
import os

import numpy as np
import pytest

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
NAMES = ("random", "lane_a", "lane_b", "lane_c", "noisy")


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "kmeans.npz"))


def _points(img):
    x, y, p = img[-100:].shape
    return np.ascontiguousarray(np.transpose(np.reshape(img[-100:].transpose(), [p, x * y])))


def test_point_order_is_the_references(golden):
    assert np.array_equal(_points(golden["img_random"]), golden["pts_random"])


@pytest.mark.parametrize("name", NAMES)
@pytest.mark.parametrize("k", (3, 4))
def test_oracle_matches_reference_runKMeans(golden, name, k):
    """centres and score to 1e-9 relative (scikit-learn sums float64 samples in thread chunks, the oracle sums integers),
    label counts exactly -- including the frames that lack one of the init colours (empty cluster re-seeded) and the
    random image, which has exact distance ties in its first iteration and does not converge within max_iter."""
    pts = _points(golden["img_" + name])
    centers, counts, inertia, n_iter = O.kmeans(pts, golden["inits%d" % k])
    ref_c, ref_n, ref_s = golden["centers%d_%s" % (k, name)], golden["counts%d_%s" % (k, name)], float(golden["score%d_%s" % (k, name)])
    assert np.array_equal(counts, ref_n)
    assert np.max(np.abs(centers - ref_c) / np.maximum(1.0, np.abs(ref_c))) < 1e-9
    assert abs(-inertia - ref_s) <= 1e-9 * abs(ref_s)
    assert 1 <= n_iter <= 25


def test_oracle_known_answers():
    # two tight blobs: one iteration moves the centres onto the blob means, the second leaves the labels unchanged
    pts = np.array([[10, 10, 10]] * 5 + [[12, 10, 10]] * 5 + [[200, 200, 200]] * 4 + [[204, 200, 200]] * 4, np.uint8)
    c, n, inertia, it = O.kmeans(pts, [[0, 0, 0], [255, 255, 255]])
    assert np.allclose(c, [[11, 10, 10], [202, 200, 200]]) and n.tolist() == [10, 8] and it == 2
    assert inertia == pytest.approx(10 * 1.0 + 8 * 4.0)
    # an init colour nobody is near: the farthest sample (lowest index among equals) re-seeds it
    pts = np.array([[0, 0, 0]] * 6 + [[9, 0, 0]] * 2, np.uint8)
    c, n, inertia, it = O.kmeans(pts, [[1, 0, 0], [250, 250, 250]])
    assert sorted(n.tolist()) == [2, 6] and inertia == pytest.approx(0.0)
    with pytest.raises(ValueError):
        O.kmeans(np.zeros((2, 3), np.uint8), [[0, 0, 0], [1, 1, 1], [2, 2, 2]])        # two samples, three clusters: one stays empty


@pytest.mark.gpu
@pytest.mark.parametrize("k", (3, 4))
def test_gpu_kmeans_is_bit_identical_to_oracle(golden, k):
    from lane_slam_amd import FrontEnd, default_config, synth
    fe = FrontEnd(default_config("parity"), max_frames=1, max_lines_per_color=64)
    imgs = [golden["img_" + n] for n in NAMES] + [synth.make_frame(31), synth.make_frame(8)[::4, ::4]]
    rng = np.random.default_rng(9)
    imgs.append(rng.integers(0, 256, (100, 64, 3), dtype=np.uint8))
    imgs.append(np.clip(rng.normal(128, 30, (100, 1920, 3)), 0, 255).astype(np.uint8))           # 1080p-wide strip
    for img in imgs:
        pts = _points(img)
        want = O.kmeans(pts, golden["inits%d" % k])
        got = fe.kmeans(pts, golden["inits%d" % k])
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2] == want[2] and got[3] == want[3]
    with pytest.raises(Exception):
        fe.kmeans(np.zeros((2, 3), np.uint8), [[0, 0, 0], [1, 1, 1], [2, 2, 2]])
    if k == 4:
        # more clusters than the register-accumulation path holds (k > 4: one LDS atomic per sample), and the 16-cluster limit
        rng = np.random.default_rng(12)
        pts = rng.integers(0, 256, (30000, 3), dtype=np.uint8)
        for kk in (6, 16):
            init = rng.integers(0, 256, (kk, 3)).astype(np.float64)
            want, got = O.kmeans(pts, init), fe.kmeans(pts, init)
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2] == want[2] and got[3] == want[3]


@pytest.mark.gpu
def test_runKMeans_mirrors_the_reference_interface(golden):
    from lane_slam_amd import anti_instagram as ai
    assert np.array_equal(ai.CENTERS, golden["inits3"]) and np.array_equal(ai.CENTERS2, golden["inits4"])
    for name in ("lane_a", "noisy"):
        centers, labelcount, score = ai.runKMeans(golden["img_" + name], 3, ai.CENTERS)
        assert np.max(np.abs(centers - golden["centers3_" + name])) < 1e-7
        assert [labelcount[i] for i in range(3)] == golden["counts3_" + name].tolist()
        assert abs(score - float(golden["score3_" + name])) <= 1e-9 * abs(score)
