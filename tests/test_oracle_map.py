"""Known-answer tests of the live map + associator contract as the oracle states it (oracle/lf_oracle_map.c).
The contract is the build's own (the reference's line_associator is a stub), so these tests pin its wording:
matching = BinaryDescriptorMatcher::match semantics (== lfo_match when ungated), colour gating, append / merge,
ring wrap, overflow, the pose transform."""
import numpy as np
import pytest

from oracle.oracle import Oracle, OracleMap
from lane_slam_amd.config import default_config


def _codes(rng, n):
    return rng.integers(0, 256, (n, 32), dtype=np.uint8)


def _flip(code, bits):
    c = code.copy()
    for b in bits:
        c[b >> 3] ^= np.uint8(1 << (b & 7))
    return c


def test_ungated_association_is_the_matcher():
    rng = np.random.default_rng(1)
    o = Oracle(default_config("parity"))
    m = OracleMap(capacity=512)
    codes = _codes(rng, 300)
    m.seed(codes)
    q = _codes(rng, 80)
    q[:20] = [_flip(codes[i], rng.choice(256, size=i, replace=False)) for i in range(20)]
    i1, d1 = m.associate(q)
    i2, d2, _ = o.match_mih(q, codes)                        # the map's default tie rule is the matcher's (Mihasher::query's first-found)
    assert np.array_equal(i1, i2) and np.array_equal(d1, d2)
    low = OracleMap(capacity=512, tie_rule="lowest")
    low.seed(codes)
    i3, d3 = low.associate(q)
    i4, d4 = o.match(q, codes)
    assert np.array_equal(i3, i4) and np.array_equal(d3, d4)
    assert m.state()["size"] == 300 and m.state()["head"] == 300


def test_gating_max_distance_and_wildcards():
    rng = np.random.default_rng(2)
    base = _codes(rng, 1)[0]
    m = OracleMap(capacity=64, color_gating=True, max_distance=40)
    # entry 0: white, exact copy; entry 1: yellow at 3 bits; entry 2: wildcard at 10 bits
    m.seed(np.stack([base, _flip(base, [1, 2, 3]), _flip(base, range(10))]), colors=np.array([0, 1, 255], np.uint8))
    q = np.stack([base] * 4)
    idx, dist = m.associate(q, np.array([0, 1, 2, 255], np.uint8))
    assert idx.tolist() == [0, 1, 2, 0] and dist.tolist() == [0.0, 3.0, 10.0, 0.0]     # red only sees the wildcard
    far = _flip(base, range(100, 141))
    idx, dist = m.associate(far[None], np.array([0], np.uint8))
    assert idx.tolist() == [-1] and dist.tolist() == [-1.0]                              # 41 bits > max_distance
    m2 = OracleMap(capacity=64, color_gating=False)
    m2.seed(np.stack([base, _flip(base, [1, 2, 3])]), colors=np.array([0, 1], np.uint8))
    idx, _ = m2.associate(q[:1], np.array([1], np.uint8))
    assert idx.tolist() == [0]                                                            # ungated: colour is ignored


def test_append_policy_kept_only_and_ring():
    rng = np.random.default_rng(3)
    m = OracleMap(capacity=8, policy="append", kept_only=True)
    c = _codes(rng, 6)
    keep = np.array([1, 0, 1, 1, 0, 1], np.uint8)
    g = rng.normal(size=(6, 4))
    idx, dist = m.step(c, np.zeros(6, np.uint8), keep, g, step=0)
    assert (idx == -1).all()                                   # empty map
    st = m.state()
    assert st["size"] == 4 and st["head"] == 4 and st["total_appended"] == 4
    f = m.fetch()
    assert np.array_equal(f["code"][:4], c[keep != 0]) and np.array_equal(f["ground"][:4], g[keep != 0])
    assert f["hits"][:4].tolist() == [1] * 4 and f["last_seen"][:4].tolist() == [0] * 4
    c2 = _codes(rng, 6)
    m.step(c2, np.zeros(6, np.uint8), np.ones(6, np.uint8), g, step=1)     # 4 + 6 = 10 > 8: wraps
    st = m.state()
    assert st["size"] == 8 and st["head"] == 2 and st["total_appended"] == 10
    f = m.fetch()
    assert np.array_equal(f["code"][4:8], c2[:4]) and np.array_equal(f["code"][:2], c2[4:])
    assert f["last_seen"].tolist() == [1, 1, 0, 0, 1, 1, 1, 1]
    # more appends than capacity in one update: the last `capacity` survive
    c3 = _codes(rng, 19)
    m.step(c3, np.zeros(19, np.uint8), np.ones(19, np.uint8), None, step=2)
    f = m.fetch()
    head = m.state()["head"]
    assert head == (2 + 19) % 8
    order = [(head + k) % 8 for k in range(8)]               # oldest -> newest
    assert np.array_equal(f["code"][order], c3[-8:])


def test_merge_policy_refreshes_matched_entries():
    rng = np.random.default_rng(4)
    m = OracleMap(capacity=32, policy="merge", merge_distance=8, kept_only=False, color_gating=True)
    base = _codes(rng, 4)
    m.seed(base, colors=np.array([0, 0, 1, 1], np.uint8), ground=np.zeros((4, 4)))
    q = np.stack([_flip(base[0], [5]), _flip(base[0], [5, 6]), _flip(base[2], range(20)), _flip(base[3], [0, 1]),
                  _codes(rng, 1)[0]])
    col = np.array([0, 0, 1, 1, 2], np.uint8)
    g = np.arange(20, dtype=np.float64).reshape(5, 4)
    idx, dist = m.step(q, col, None, g, step=7)
    assert idx[:4].tolist() == [0, 0, 2, 3] and dist[:4].tolist() == [1.0, 2.0, 20.0, 2.0]
    st = m.state()
    # q0 and q1 refresh entry 0 (q1, the later one, leaves its data), q3 refreshes entry 3;
    # q2 (20 bits > merge_distance) and q4 (no red entry) are appended
    assert st["total_refreshed"] == 3 and st["total_appended"] == 4 + 2 and st["size"] == 6
    f = m.fetch()
    assert np.array_equal(f["code"][0], q[1]) and np.array_equal(f["ground"][0], g[1]) and f["hits"][0] == 3
    assert np.array_equal(f["code"][3], q[3]) and f["hits"][3] == 2 and f["last_seen"][3] == 7
    assert np.array_equal(f["code"][1], base[1]) and f["hits"][1] == 1 and f["last_seen"][1] == -1
    assert np.array_equal(f["code"][4], q[2]) and np.array_equal(f["code"][5], q[4]) and f["color"][5] == 2


def test_full_error_mode_drops_and_flags():
    rng = np.random.default_rng(5)
    m = OracleMap(capacity=4, when_full="error", kept_only=False)
    c = _codes(rng, 6)
    m.step(c, None, None, None, step=0)
    st = m.state()
    assert st["size"] == 4 and st["overflow"] == 1 and st["total_appended"] == 6
    assert np.array_equal(m.fetch()["code"], c[:4])


def test_pose_transform():
    m = OracleMap(capacity=8)
    g = np.array([[1.0, 0.0, 0.0, 2.0], [0.5, -0.25, 3.0, 1.0], [1.0, 1.0, 2.0, 2.0]])
    fo = np.array([0, 2, 2, 3], np.int32)                      # frame 1 is empty
    poses = np.array([[10.0, -4.0, np.pi / 2], [0.0, 0.0, 0.3], [1.0, 2.0, 0.0]])
    out = m.to_map_frame(g, fo, poses)
    exp = g.copy()
    for f in range(3):
        x, y, th = poses[f]
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        for s in range(fo[f], fo[f + 1]):
            exp[s, :2] = R @ g[s, :2] + (x, y)
            exp[s, 2:] = R @ g[s, 2:] + (x, y)
    np.testing.assert_allclose(out, exp, rtol=0, atol=1e-12)
    assert np.array_equal(out[2], g[2] + [1.0, 2.0, 1.0, 2.0])                 # theta = 0: a pure translation
