"""SegmentList glue.  CPU part: the wire helpers against an independent struct.pack restatement
of the ROS 1 serialisation of the reference's .msg definitions (Segment.msg:1-8, Vector2D.msg:1-2, Header,
geometry_msgs/Point).  GPU part (-m gpu): lf_serialize_segments / lf_deserialize_segments through the C ABI
against the same restatement, for the three publishing nodes' stages."""
import struct

import numpy as np
import pytest

from lane_slam_amd import segment_msgs as sm


def reference_body(seg, f, stage):
    """One frame's `Segment[] segments` bytes, field by field in .msg order, as rospy's serialiser writes them
    for what each node fills (line_detector_node.py:251-265, ground_projection_node.py:55-65, line_sanity_node.py:48-72)."""
    a, b = int(seg.frame_offset[f]), int(seg.frame_offset[f + 1])
    recs = []
    for i in range(a, b):
        if stage == sm.FILTERED and not seg.keep[i]:
            continue
        pn = [0.0] * 4 if stage != sm.DETECTOR else [float(v) for v in seg.pixels_normalized[i]]
        nm = [0.0] * 2 if stage != sm.DETECTOR else [float(v) for v in seg.normals[i]]
        g = [0.0] * 4 if stage == sm.DETECTOR else [float(v) for v in seg.ground[i]]
        recs.append(struct.pack("<B", int(seg.color[i])) + struct.pack("<4f", *pn) + struct.pack("<2f", *nm) +
                    struct.pack("<6d", g[0], g[1], 0.0, g[2], g[3], 0.0))
    return struct.pack("<I", len(recs)) + b"".join(recs)


def test_record_dtype_and_message_helpers():
    assert sm.SEGMENT_DTYPE.itemsize == 73
    assert [sm.SEGMENT_DTYPE.fields[k][1] for k in ("color", "pixels_normalized", "normal", "points")] == [0, 1, 17, 25]
    rec = np.zeros(2, sm.SEGMENT_DTYPE)
    rec["color"] = [1, 2]
    rec["pixels_normalized"][0] = [[0.25, 0.5], [0.75, 1.0]]
    rec["normal"][1] = [-1.0, 0.5]
    rec["points"][1] = [[1.5, -0.25, 0.0], [2.5, 0.125, 0.0]]
    body = struct.pack("<I", 2) + rec.tobytes()
    hdr = sm.header_bytes(7, 1500000000, 123456789, "duckiebot/camera")
    assert hdr == struct.pack("<III", 7, 1500000000, 123456789) + struct.pack("<I", 16) + b"duckiebot/camera"
    msg = sm.segment_list_message(hdr, body)
    seq, secs, nsecs, fid, body2, view = sm.split_segment_list(msg)
    assert (seq, secs, nsecs, fid) == (7, 1500000000, 123456789, "duckiebot/camera") and body2 == body
    assert view["color"].tolist() == [1, 2] and view["points"][1, 1, 0] == 2.5 and view["normal"][1, 0] == -1.0
    with pytest.raises(ValueError):
        sm.split_segment_list(msg[:-3])


@pytest.mark.gpu
def test_gpu_bodies_match_the_wire_format_and_round_trip():
    from lane_slam_amd import FrontEnd, default_config, synth
    cfg = default_config("parity")
    n = 9
    fe = FrontEnd(cfg, max_frames=n, max_lines_per_color=256)
    frames = synth.make_batch(n, 100)
    frames[4] = 70                                     # a frame without segments: count 0, 4-byte body
    seg = fe.process_batch(frames)
    assert seg.n > 50 and seg.frame_offset[5] == seg.frame_offset[4]
    for stage in (sm.DETECTOR, sm.GROUND, sm.FILTERED):
        bodies, off = sm.serialize_segments(fe, seg, stage)
        assert off[0] == 0 and off[-1] == bodies.size
        for f in range(n):
            assert bodies[off[f]:off[f + 1]].tobytes() == reference_body(seg, f, stage), (stage, f)
        fo, color, pn, nm, gr = sm.deserialize_segments(fe, bodies, off)
        if stage == sm.FILTERED:
            k = seg.keep.astype(bool)
            assert np.array_equal(color, seg.color[k]) and np.array_equal(gr, seg.ground[k])
            assert fo[-1] == int(k.sum())
        else:
            assert np.array_equal(fo, seg.frame_offset) and np.array_equal(color, seg.color)
            if stage == sm.DETECTOR:
                assert np.array_equal(pn, seg.pixels_normalized) and np.array_equal(nm, seg.normals) and not gr.any()
            else:
                assert np.array_equal(gr, seg.ground) and not pn.any() and not nm.any()
    # the reference's own nodes' messages (tests/golden/node_pipeline.npz: processImage_ / lineseglist_cb run from the
    # reference, serialised with the field order of its .msg files): lf_serialize_segments must write those bytes
    import os
    from lane_slam_amd.frontend import Segments
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "node_pipeline.npz"))
    hdr_len = len(sm.header_bytes(7, 1234, 5678, "cam"))
    for ci in range(int(g["n_cases"])):
        k = int(g["det_color%d" % ci].size)
        s2 = Segments()
        s2.n = k
        s2.frame_offset = np.array([0, k], np.int32)
        s2.color = g["det_color%d" % ci]
        s2.pixels_normalized = g["det_pn64_%d" % ci].astype(np.float32).reshape(k, 4)
        s2.normals = g["det_normal64_%d" % ci].astype(np.float32).reshape(k, 2)
        s2.ground = np.ascontiguousarray(g["gp_points%d" % ci].reshape(k, 6)[:, [0, 1, 3, 4]])
        s2.keep = np.ones(k, np.uint8)
        body, off = sm.serialize_segments(fe, s2, sm.DETECTOR)
        assert body.tobytes() == g["det_wire%d" % ci].tobytes()[hdr_len:], ci
        body, off = sm.serialize_segments(fe, s2, sm.GROUND)
        assert body.tobytes() == g["gp_wire%d" % ci].tobytes()[hdr_len:], ci
    # a whole message, read back the way a subscriber would
    bodies, off = sm.serialize_segments(fe, seg, sm.GROUND)
    msg = sm.segment_list_message(sm.header_bytes(3, 10, 20, "cam"), bodies[off[2]:off[3]])
    view = sm.split_segment_list(msg)[5]
    a, b = seg.frame_offset[2], seg.frame_offset[3]
    assert np.array_equal(view["points"][:, 0, :2], seg.ground[a:b, :2]) and np.array_equal(view["color"], seg.color[a:b])
    # error path: a corrupt count
    from lane_slam_amd import LanefrontError
    with pytest.raises(LanefrontError):
        bad = bodies.copy()
        bad[off[1]] ^= 1                                # count no longer matches the length
        sm.deserialize_segments(fe, bad, off)
