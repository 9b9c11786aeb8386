"""SegmentList wire helpers: what rospy's generated serialisers would produce for
duckietown_msgs/SegmentList, built from the struct-of-arrays block without a Python loop over segments.

Message layout (ref: src/duckietown_msgs/msg/SegmentList.msg:1-2, Segment.msg:1-8, Vector2D.msg:1-2,
std_msgs/Header, geometry_msgs/Point; ROS 1 serialisation, little endian):
    Header   u32 seq | u32 stamp.secs | u32 stamp.nsecs | u32 len + frame_id bytes
    body     u32 count | count x 73-byte Segment records          <- lf_serialize_segments writes these
The record as a packed numpy dtype is SEGMENT_DTYPE, so a subscriber can read a body with np.frombuffer."""
import ctypes
import struct

import numpy as np

from . import _lib

SEGMENT_DTYPE = np.dtype([("color", "u1"), ("pixels_normalized", "<f4", (2, 2)), ("normal", "<f4", (2,)),
                          ("points", "<f8", (2, 3))])
assert SEGMENT_DTYPE.itemsize == 73

DETECTOR, GROUND, FILTERED = _lib.LF_MSG_DETECTOR, _lib.LF_MSG_GROUND, _lib.LF_MSG_FILTERED


def header_bytes(seq, secs, nsecs, frame_id=""):
    fid = frame_id.encode() if isinstance(frame_id, str) else bytes(frame_id)
    return struct.pack("<IIII", seq, secs, nsecs, len(fid)) + fid


def segment_list_message(header, body):
    """A complete serialised SegmentList: header_bytes(...) + one body from serialize_segments."""
    return bytes(header) + bytes(body)


def split_segment_list(msg):
    """(seq, secs, nsecs, frame_id, body bytes, records view) of a serialised SegmentList."""
    msg = bytes(msg)
    seq, secs, nsecs, n = struct.unpack_from("<IIII", msg, 0)
    frame_id = msg[16:16 + n].decode()
    body = msg[16 + n:]
    (count,) = struct.unpack_from("<I", body, 0)
    if len(body) != 4 + 73 * count:
        raise ValueError("SegmentList body: %d bytes for %d segments" % (len(body), count))
    return seq, secs, nsecs, frame_id, body, np.frombuffer(body, SEGMENT_DTYPE, count=count, offset=4)


def _fill(s, seg, names):
    keep_alive = []
    for k in names:
        a = np.ascontiguousarray(getattr(seg, k))
        keep_alive.append(a)
        setattr(s, k, a.ctypes.data)
    return keep_alive


def serialize_segments(fe, seg, stage):
    """seg: a host `Segments` (FrontEnd.process_batch).  Returns (bodies uint8 array, offsets int64[n_frames+1]):
    frame f's body is bodies[offsets[f]:offsets[f+1]]."""
    n_frames = len(seg.frame_offset) - 1
    s = _lib.LfSegments()
    s.capacity = int(seg.n)
    names = ["frame_offset", "color"] + (["pixels_normalized", "normals"] if stage == DETECTOR else ["ground"] + (["keep"] if stage == FILTERED else []))
    alive = _fill(s, seg, names)
    cap = 4 * n_frames + 73 * int(seg.n)
    out = np.empty(cap, np.uint8)
    off = np.zeros(n_frames + 1, np.int64)
    fe._check(fe.lib.lf_serialize_segments(fe.h, ctypes.byref(s), 0, n_frames, int(stage), out.ctypes.data_as(ctypes.c_void_p), cap, 0,
                                           off.ctypes.data_as(ctypes.POINTER(ctypes.c_int64))))
    del alive
    return out[: off[-1]], off


def deserialize_segments(fe, bodies, offsets):
    """Inverse of serialize_segments: returns (frame_offset, color, pixels_normalized (n,4), normals (n,2), ground (n,4))."""
    bodies = np.ascontiguousarray(np.frombuffer(bytes(bodies), np.uint8) if not isinstance(bodies, np.ndarray) else bodies, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n_frames = len(offsets) - 1
    cap = max(1, int(bodies.size // 73) + 1)
    fo = np.zeros(n_frames + 1, np.int32)
    color = np.empty(cap, np.uint8)
    pn = np.empty((cap, 4), np.float32)
    nm = np.empty((cap, 2), np.float32)
    gr = np.empty((cap, 4), np.float64)
    s = _lib.LfSegments()
    s.capacity = cap
    s.frame_offset, s.color, s.pixels_normalized, s.normals, s.ground = (a.ctypes.data for a in (fo, color, pn, nm, gr))
    total = ctypes.c_int()
    fe._check(fe.lib.lf_deserialize_segments(fe.h, bodies.ctypes.data_as(ctypes.c_void_p), 0,
                                             offsets.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), n_frames, ctypes.byref(s), 0,
                                             ctypes.byref(total)))
    n = total.value
    return fo, color[:n], pn[:n], nm[:n], gr[:n]
