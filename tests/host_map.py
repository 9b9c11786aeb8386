"""TEST INFRASTRUCTURE: a host stand-in for lane_slam_amd.LineAssociator's device interface, built on the oracle's
live map (oracle/lf_oracle_map.c), so that lane_slam_amd.distributed.ShardedAssociator -- the per-step protocol
bench.py runs on MI355Xs -- can be exercised on CPU tensors with gloo.  "Device" pointers are host addresses here.
Blocks are packed / parsed with the numpy dtypes of lane_slam_amd.distributed (an independent statement of the
layout include/lanefront.h gives; tests/test_gpu_map.py holds the HIP packer to the same bytes)."""
import ctypes

import numpy as np

from lane_slam_amd.distributed import BLOCK_MAGIC, BLOCK_ROW_BYTES, HEADER_DTYPE, ROW_DTYPE
from lane_slam_amd.frontend import LanefrontError
from oracle.oracle import OracleMap


def _view(ptr, dtype, count):
    if count == 0:
        return np.empty(0, dtype)
    buf = (ctypes.c_uint8 * (np.dtype(dtype).itemsize * count)).from_address(int(ptr))
    return np.frombuffer(buf, dtype=dtype, count=count)


class HostMap(object):
    def __init__(self, **kw):
        self.o = OracleMap(**kw)
        self.gating = bool(kw.get("color_gating", False))

    def stream_ptr(self):
        return 0

    def associate_device(self, fe, code_ptr, color_ptr, n, idx_ptr, dist_ptr):
        codes = _view(code_ptr, np.uint8, n * 32).reshape(n, 32)
        colors = _view(color_ptr, np.uint8, n) if color_ptr else None
        idx, dist = self.o.associate(codes, colors)
        _view(idx_ptr, np.int32, n)[:] = idx
        _view(dist_ptr, np.float32, n)[:] = dist

    def pack_block_device(self, fe, ptrs, n, n_frames, idx_ptr, dist_ptr, poses, step, block_ptr, block_rows):
        blk = _view(block_ptr, np.uint8, block_rows * BLOCK_ROW_BYTES)
        hdr = np.zeros(1, HEADER_DTYPE)
        hdr["magic"], hdr["count"], hdr["step"], hdr["n_frames"] = BLOCK_MAGIC, n, step, n_frames
        if n + 1 > block_rows:
            # as lf_map_pack_block: a header with the overflow marker, then LF_ERR_CAPACITY
            hdr["count"], hdr["overflow"] = 0, n
            blk[:BLOCK_ROW_BYTES] = np.frombuffer(hdr.tobytes(), np.uint8)
            raise LanefrontError(-2, "%d segments do not fit a block of %d rows" % (n, block_rows))
        blk[:BLOCK_ROW_BYTES] = np.frombuffer(hdr.tobytes(), np.uint8)
        rows = np.zeros(n, ROW_DTYPE)
        rows["code"] = _view(ptrs["code"], np.uint8, n * 32).reshape(n, 32)
        g = _view(ptrs["ground"], np.float64, n * 4).reshape(n, 4)
        if poses is not None:
            fo = _view(ptrs["frame_offset"], np.int32, n_frames + 1)
            g = self.o.to_map_frame(g, fo, poses)
        rows["ground"] = g
        rows["idx"] = _view(idx_ptr, np.int32, n)
        rows["dist"] = _view(dist_ptr, np.float32, n)
        rows["color"] = _view(ptrs["color"], np.uint8, n)
        rows["keep"] = _view(ptrs["keep"], np.uint8, n)
        blk[BLOCK_ROW_BYTES:(n + 1) * BLOCK_ROW_BYTES] = np.frombuffer(rows.tobytes(), np.uint8)

    def update_device(self, blocks_ptr, n_blocks, block_rows):
        raw = _view(blocks_ptr, np.uint8, n_blocks * block_rows * BLOCK_ROW_BYTES).reshape(n_blocks, block_rows * BLOCK_ROW_BYTES)
        parts, step = [], 0
        hdrs = [np.frombuffer(raw[b, :BLOCK_ROW_BYTES].tobytes(), HEADER_DTYPE)[0] for b in range(n_blocks)]
        if any(int(h["overflow"]) for h in hdrs):
            # as lf_map_update: nothing is applied on any replica; the device map reports it at its next synchronising
            # call, this stand-in at once
            raise LanefrontError(-2, "a rank's segments did not fit its block: the step was applied on no replica")
        for b in range(n_blocks):
            hdr = hdrs[b]
            assert int(hdr["magic"]) == BLOCK_MAGIC
            if b == 0:
                step = int(hdr["step"])
            parts.append(np.frombuffer(raw[b].tobytes(), ROW_DTYPE, count=int(hdr["count"]), offset=BLOCK_ROW_BYTES))
        rows = np.concatenate(parts)
        self.o.update(rows["code"], rows["color"], rows["keep"], rows["ground"], rows["idx"], rows["dist"], step)
