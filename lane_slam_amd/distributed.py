"""Multi-GPU association against a replicated live map.

Frames shard across ranks as independent batches: detect -> describe -> project -> sanity need no collective, and
neither does association (the map is replicated, every rank matches only its own frames' descriptors).  The one
exchange per step is ONE all-gather of fixed-capacity segment BLOCKS (include/lanefront.h, "live map": header row
with the count, then 80 bytes per segment: code, ground endpoints in the map frame, association result, colour,
keep), after which every rank applies the same blocks in rank order to its replica, so the replicas -- and with
frames dealt to ranks in contiguous chunks, the single-GPU map -- stay identical, entry for entry.

`ShardedAssociator.step` is the whole per-step protocol; bench.py and the world-size-2 gloo test both call it.  The
map behind it is anything with LineAssociator's device interface (associate_device / pack_block_device /
update_device / stream_ptr): lane_slam_amd.LineAssociator on an MI355X, or the oracle-backed stand-in the CPU test
uses.  A single rank skips the collective and applies its own block through the same calls.
"""
import numpy as np
import torch
import torch.distributed as dist

BLOCK_ROW_BYTES = 80
BLOCK_MAGIC = 0x4B42464C          # "LFBK"

# numpy view of a block's segment rows / header row (an independent statement of the layout in lanefront.h)
ROW_DTYPE = np.dtype({"names": ["code", "ground", "idx", "dist", "color", "keep", "pad"],
                      "formats": [("u1", 32), ("<f8", 4), "<i4", "<f4", "u1", "u1", ("u1", 6)],
                      "offsets": [0, 32, 64, 68, 72, 73, 74], "itemsize": BLOCK_ROW_BYTES})
HEADER_DTYPE = np.dtype({"names": ["magic", "count", "step", "n_frames", "overflow"], "formats": ["<u4", "<u4", "<i4", "<u4", "<u4"],
                         "offsets": [0, 4, 8, 12, 16], "itemsize": BLOCK_ROW_BYTES})
LF_ERR_CAPACITY = -2


def block_header(block_bytes):
    """(count, step, n_frames) of a block given as a uint8 numpy array."""
    h = np.frombuffer(block_bytes[:BLOCK_ROW_BYTES].tobytes(), HEADER_DTYPE)[0]
    if int(h["magic"]) != BLOCK_MAGIC:
        raise ValueError("not a segment block (bad magic)")
    return int(h["count"]), int(h["step"]), int(h["n_frames"])


def block_rows(block_bytes):
    """Structured view (ROW_DTYPE) of a block's valid segment rows."""
    count = block_header(block_bytes)[0]
    return np.frombuffer(block_bytes.tobytes(), ROW_DTYPE, count=count, offset=BLOCK_ROW_BYTES)


class ShardedAssociator(object):
    """One rank's replica of the live map plus the per-step exchange.

    amap            LineAssociator (or an object with its device interface)
    block_segments  capacity of one rank's block; a step with more segments raises (never truncates) -- on EVERY rank:
                    the overflowing rank still takes part in the all-gather with a header-only block that carries the
                    overflow marker, every replica skips that step's update, the overflowing rank raises at once and
                    the others at their next call that synchronises with the map (lanefront.h, "failing updates")
    device          torch device of the segment arrays and blocks
    backend         "nccl" (= RCCL over xGMI, device buffers), "gloo" (CPU tensors; with a CUDA device the blocks are
                    staged through the host -- dry runs only)
    """

    def __init__(self, amap, block_segments, device="cpu", backend="nccl", force_collective=False):
        self.map = amap
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.collective = self.world > 1 or (force_collective and dist.is_initialized())
        self.rows = int(block_segments) + 1
        self.device = torch.device(device)
        self.backend = backend
        nbytes = self.rows * BLOCK_ROW_BYTES
        self.block = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        self.gathered = torch.zeros(self.world * nbytes, dtype=torch.uint8, device=self.device) if self.collective else self.block
        self.cuda = self.device.type == "cuda"
        self.ext = torch.cuda.ExternalStream(amap.stream_ptr(), device=self.device) if self.cuda else None

    def step(self, fe, out, n, n_frames, idx, dist_out, poses=None, step=0):
        """Associate the n segments of this rank's batch, exchange blocks, update the replica.

        fe       the FrontEnd whose stream produced `out` (None: the caller has ordered that)
        out      dict of tensors: frame_offset, code, color, keep, ground (device resident)
        idx, dist_out   int32 / float32 tensors [>= n] receiving this rank's association results
        poses    (n_frames, 3) map -> duck pose per frame, or None
        """
        m = self.map
        ptrs = {k: out[k].data_ptr() for k in ("frame_offset", "code", "color", "keep", "ground")}
        if n > 0:
            m.associate_device(fe, ptrs["code"], ptrs["color"], n, idx.data_ptr(), dist_out.data_ptr())
        # LF_ERR_CAPACITY when n does not fit the block: fail loudly, never truncate -- but only AFTER the collective,
        # which the other ranks have entered or will enter (the block is then a header with the overflow marker)
        overflow = None
        try:
            m.pack_block_device(fe, ptrs, n, n_frames, idx.data_ptr(), dist_out.data_ptr(), poses, step, self.block.data_ptr(),
                                self.rows)
        except Exception as e:
            if getattr(e, "code", None) != LF_ERR_CAPACITY or not self.collective:
                raise
            overflow = e
        if self.collective:
            if self.cuda:
                torch.cuda.current_stream(self.device).wait_stream(self.ext)      # the block is complete
            if self.cuda and self.backend == "gloo":
                hb = self.block.cpu()
                hg = torch.empty(self.gathered.shape, dtype=torch.uint8)
                dist.all_gather_into_tensor(hg, hb)
                self.gathered.copy_(hg)
            else:
                dist.all_gather_into_tensor(self.gathered, self.block)             # the step's ONE collective
            if self.cuda:
                self.ext.wait_stream(torch.cuda.current_stream(self.device))
        try:
            m.update_device(self.gathered.data_ptr(), self.world if self.collective else 1, self.rows)
        finally:
            if overflow is not None:
                raise overflow
