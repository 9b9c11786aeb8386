"""The C ABI from C: tests/c_abi/abi_client.c (gcc, C11, only include/lanefront.h) linked against
liblanefront.so.  CPU part: it compiles and links against every symbol it uses.  GPU part: it runs the batch path,
the associator and the SegmentList serialiser with host pointers, and its output equals the Python binding's and
the oracle's."""
import os
import struct
import subprocess

import numpy as np
import pytest

from lane_slam_amd import _lib, default_config, synth
from lane_slam_amd.config import LfConfig, fill_struct

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def build_client():
    exe = os.path.join(HERE, "hostsim", "_build", "abi_client")
    src = os.path.join(HERE, "c_abi", "abi_client.c")
    so = os.path.join(ROOT, "lane_slam_amd", "liblanefront.so")
    deps = [src, os.path.join(ROOT, "include", "lanefront.h"), so]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-Werror", "-o", exe, src,
                               "-L" + os.path.dirname(so), "-l:liblanefront.so", "-Wl,-rpath," + os.path.dirname(so),
                               "-Wl,--allow-shlib-undefined"])
    return exe


def test_c_client_compiles_and_links():
    exe = build_client()
    assert os.access(exe, os.X_OK)
    # usage error path needs no GPU
    p = subprocess.run([exe], capture_output=True)
    assert p.returncode == 2 and b"usage" in p.stderr


@pytest.mark.gpu
def test_c_client_matches_python_binding_and_oracle(tmp_path):
    from lane_slam_amd import FrontEnd
    from lane_slam_amd import segment_msgs as sm
    from oracle.oracle import Oracle
    exe = build_client()
    cfg = default_config("parity")
    n = 6
    frames = synth.make_batch(n, 500)
    c = LfConfig()
    fill_struct(c, cfg)
    (tmp_path / "cfg.bin").write_bytes(bytes(c))
    (tmp_path / "frames.bin").write_bytes(frames.tobytes())
    p = subprocess.run([exe, str(tmp_path / "cfg.bin"), str(tmp_path / "frames.bin"), str(n), str(tmp_path / "out.bin")],
                       capture_output=True)
    assert p.returncode == 0, p.stderr.decode()
    raw = (tmp_path / "out.bin").read_bytes()
    total, rc_small, body_bytes, n_stages = struct.unpack_from("<4i", raw, 0)
    assert rc_small == -2 and n_stages == _lib.LF_N_STAGES          # LF_ERR_CAPACITY
    map_size1, map_size2, matched0, appended, refreshed = struct.unpack_from("<5i", raw, 16)
    pos = 36

    def take(dtype, count):
        nonlocal pos
        a = np.frombuffer(raw, dtype, count=count, offset=pos)
        pos += a.nbytes
        return a

    fo = take(np.int32, n + 1)
    lines = take(np.float32, total * 4).reshape(-1, 4)
    ground = take(np.float64, total * 4).reshape(-1, 4)
    keep = take(np.uint8, total)
    code = take(np.uint8, total * 32).reshape(-1, 32)
    idx = take(np.int32, total)
    dist = take(np.float32, total)
    boff = take(np.int64, n + 1)
    body = take(np.uint8, body_bytes)
    n_kl = int(take(np.int32, 1)[0])
    kfo = take(np.int32, n + 1)
    kio = take(np.float32, n_kl * 4).reshape(-1, 4)
    koct = take(np.int32, n_kl)
    kcls = take(np.int32, n_kl)
    kcode = take(np.uint8, n_kl * 32).reshape(-1, 32)
    assert pos == len(raw) and total > 20
    fe = FrontEnd(cfg, max_frames=n, max_lines_per_color=512)
    seg = fe.process_batch(frames)
    assert seg.n == total and np.array_equal(fo, seg.frame_offset)
    assert np.array_equal(lines, seg.lines) and np.array_equal(ground, seg.ground)
    assert np.array_equal(keep, seg.keep) and np.array_equal(code, seg.code)
    o = Oracle(cfg)
    oi, od, _ = o.match_mih(code, code[::-1].copy())          # lf_associate's default tie rule: the reference's
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    ref_body, ref_off = sm.serialize_segments(fe, seg, sm.FILTERED)
    assert np.array_equal(boff, ref_off) and np.array_equal(body, ref_body)
    # the live map from plain C (lf_map_step_host): the same two steps through the oracle's statement of the contract
    from oracle.oracle import OracleMap
    om = OracleMap(capacity=4096, max_distance=128, policy="merge", kept_only=True, merge_distance=0)
    om.step(code, seg.color, keep, ground, 0)
    assert om.state()["size"] == map_size1
    i2, d2 = om.step(code, seg.color, keep, ground, 1)
    st = om.state()
    assert (st["size"], st["total_appended"], st["total_refreshed"]) == (map_size2, appended, refreshed)
    assert matched0 == int(((keep != 0) & (i2 >= 0) & (d2 == 0)).sum()) and matched0 >= int(keep.sum()) > 0
    # and through the Python binding's host-array step
    from lane_slam_amd import LineAssociator
    la = LineAssociator(capacity=4096, max_distance=128, policy="merge", kept_only=True, merge_distance=0)
    la.step(seg, step=0)
    gi, gd = la.step(seg, step=1)
    assert np.array_equal(gi, i2) and np.array_equal(gd, d2) and la.state()["size"] == map_size2
    la.close()
    # the EDLines / KeyLines call from plain C against the oracle
    from oracle import oracle as O
    assert n_kl > 0 and kfo[-1] == n_kl
    for f in range(n):
        r = O.octave_keylines(o.bgr2gray(o.preprocess(frames[f])), 2)
        a, b = int(kfo[f]), int(kfo[f + 1])
        assert b - a == r["n"] and np.array_equal(kio[a:b], r["in_octave"]) and np.array_equal(koct[a:b], r["octave"])
        assert np.array_equal(kcls[a:b], r["class_id"]) and np.array_equal(kcode[a:b], r["code"])
    r0 = o.process_frame(frames[0])
    assert np.array_equal(lines[fo[0]:fo[1]], r0["lines"]) and np.array_equal(keep[fo[0]:fo[1]], r0["keep"])
