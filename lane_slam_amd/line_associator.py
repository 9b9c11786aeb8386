"""LineAssociator: the package's counterpart of the reference's line_associator node + show_map's segment store.

The reference node is an unfinished stub (ref: src/line_associator/src/line_associator_node.py:12-86) and show_map
keeps every received segment in an append-only list (ref: src/show_map/src/show_map.py:28-42), with the map -> duck
pose published separately by odometry (ref: src/odometry/src/odometry.py:110-120).  What is built here is therefore
this package's own contract, stated in include/lanefront.h ("live map"): a device-resident live map, matched with
BinaryDescriptorMatcher::match semantics (ref: src/line_descriptor/src/binary_descriptor_matcher.cpp:197-254) on the
matrix cores, optionally gated by Segment.color, and updated from the association results (append, or refresh the
matched entry).  All of it runs in liblanefront.so (lf_map_*); this class only marshals arguments.
"""
import ctypes

import numpy as np

from . import _lib

BLOCK_ROW_BYTES = 80
_POLICY = {"append": 0, "merge": 1}
_WHEN_FULL = {"ring": 0, "error": 1}


class LineAssociator(object):
    def __init__(self, capacity=65536, color_gating=False, max_distance=128, policy="append", kept_only=True,
                 merge_distance=0, when_full="ring", device=0, tie_rule=None):
        self.lib = _lib.load()
        # tie_rule=None: the library's default -- the reference's rule ("mihasher"), or "lowest" with the library's one warning under
        # LF_ASSOC_INT8 (the int8 A/B kernels have no tie pass; an EXPLICIT "mihasher" there raises, as lf_map_set_tie_rule does)
        if tie_rule is not None and tie_rule not in _lib.TIE_RULES:
            raise ValueError("tie_rule must be one of %r" % (sorted(_lib.TIE_RULES),))
        if policy not in _POLICY or when_full not in _WHEN_FULL:
            raise ValueError("policy must be 'append' or 'merge', when_full 'ring' or 'error'")
        self.capacity = int(capacity)
        self.color_gating = bool(color_gating)
        c = _lib.LfMapConfig(self.capacity, int(self.color_gating), int(max_distance), _POLICY[policy], int(bool(kept_only)),
                             int(merge_distance), _WHEN_FULL[when_full])
        self.m = ctypes.c_void_p()
        rc = self.lib.lf_map_create(int(device), ctypes.byref(c), ctypes.byref(self.m))
        if rc != 0:
            msg = self.lib.lf_map_last_error(None).decode()
            self.m = None
            from .frontend import LanefrontError
            raise LanefrontError(rc, msg)
        if tie_rule is not None:
            self._check(self.lib.lf_map_set_tie_rule(self.m, _lib.TIE_RULES[tie_rule]))

    def close(self):
        if getattr(self, "m", None):
            self.lib.lf_map_destroy(self.m)
            self.m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            from .frontend import LanefrontError
            raise LanefrontError(rc, self.lib.lf_map_last_error(self.m).decode())

    def stream_ptr(self):
        p = ctypes.c_void_p()
        self._check(self.lib.lf_map_get_stream(self.m, ctypes.byref(p)))
        return p.value or 0

    def synchronize(self):
        self._check(self.lib.lf_map_synchronize(self.m))

    def set_profiling(self, on):
        self._check(self.lib.lf_map_set_profiling(self.m, int(bool(on))))

    def timing(self):
        """{stage name: (ms, launches)} since the previous call (HIP events on the map's stream); resets."""
        ms = np.zeros(_lib.LF_MAP_N_STAGES, np.float64)
        ln = np.zeros(_lib.LF_MAP_N_STAGES, np.int32)
        self._check(self.lib.lf_map_get_timing(self.m, ms.ctypes.data, ln.ctypes.data, _lib.LF_MAP_N_STAGES))
        return {self.lib.lf_map_stage_name(i).decode(): (float(ms[i]), int(ln[i])) for i in range(_lib.LF_MAP_N_STAGES)}

    # ------------------------------------------------------------------ host arrays
    def seed(self, codes, colors=None, ground=None):
        codes = np.ascontiguousarray(codes, np.uint8).reshape(-1, 32)
        colors = None if colors is None else np.ascontiguousarray(colors, np.uint8)
        ground = None if ground is None else np.ascontiguousarray(ground, np.float64).reshape(-1, 4)
        self._check(self.lib.lf_map_seed(self.m, codes.ctypes.data, None if colors is None else colors.ctypes.data,
                                         None if ground is None else ground.ctypes.data, codes.shape[0], 0))

    def seed_device(self, code_ptr, n, color_ptr=None, ground_ptr=None):
        self._check(self.lib.lf_map_seed(self.m, int(code_ptr), color_ptr and int(color_ptr), ground_ptr and int(ground_ptr), int(n), 1))

    def state(self):
        """{'size', 'head', 'total_appended', 'total_refreshed'} (waits for the map's stream)."""
        size, head = ctypes.c_int(), ctypes.c_int()
        ta, tr = ctypes.c_int64(), ctypes.c_int64()
        self._check(self.lib.lf_map_size(self.m, ctypes.byref(size), ctypes.byref(head), ctypes.byref(ta), ctypes.byref(tr)))
        return {"size": size.value, "head": head.value, "total_appended": ta.value, "total_refreshed": tr.value}

    def associate(self, codes, colors=None):
        """(idx int32, dist float32) of each code's nearest map entry; idx == -1: none within max_distance."""
        codes = np.ascontiguousarray(codes, np.uint8).reshape(-1, 32)
        colors = None if colors is None else np.ascontiguousarray(colors, np.uint8)
        n = codes.shape[0]
        idx, dist = np.empty(n, np.int32), np.empty(n, np.float32)
        self._check(self.lib.lf_map_associate(self.m, None, codes.ctypes.data, None if colors is None else colors.ctypes.data, n,
                                              idx.ctypes.data, dist.ctypes.data, 0))
        return idx, dist

    def step(self, seg, poses=None, step=0):
        """Associate the segments of a host `Segments` block (FrontEnd.process_batch) against the map, then update the
        map with them; returns (idx, dist).  poses: (n_frames, 3) map -> duck (x, y, theta) per frame, or None."""
        n, n_frames = int(seg.n), len(seg.frame_offset) - 1
        keep_alive, pp = self._poses(poses, n_frames)
        s = _lib.LfSegments()
        s.capacity = n
        alive = []
        for k in ("frame_offset", "code", "color", "keep", "ground"):
            v = getattr(seg, k)
            if v is not None:
                a = np.ascontiguousarray(v)
                alive.append(a)
                setattr(s, k, a.ctypes.data)
        idx, dist = np.empty(n, np.int32), np.empty(n, np.float32)
        self._check(self.lib.lf_map_step_host(self.m, ctypes.byref(s), n, n_frames, pp, int(step), idx.ctypes.data, dist.ctypes.data))
        return idx, dist

    def fetch(self, first=0, n=None):
        n = self.capacity - first if n is None else n
        out = {"code": np.empty((n, 32), np.uint8), "color": np.empty(n, np.uint8), "ground": np.empty((n, 4), np.float64),
               "hits": np.empty(n, np.int32), "last_seen": np.empty(n, np.int32)}
        self._check(self.lib.lf_map_fetch(self.m, int(first), int(n), out["code"].ctypes.data, out["color"].ctypes.data,
                                          out["ground"].ctypes.data, out["hits"].ctypes.data, out["last_seen"].ctypes.data))
        return out

    # ------------------------------------------------------------------ device resident
    @staticmethod
    def _segs(out_ptrs):
        s = _lib.LfSegments()
        s.capacity = 0
        for k, v in out_ptrs.items():
            setattr(s, k, int(v))
        return s

    @staticmethod
    def _poses(poses, n_frames):
        if poses is None:
            return None, None
        a = np.ascontiguousarray(poses, np.float64).reshape(-1, 3)
        if a.shape[0] != n_frames:
            raise ValueError("poses must be (n_frames, 3) = x, y, theta per frame")
        return a, a.ctypes.data

    def associate_device(self, fe, code_ptr, color_ptr, n, idx_ptr, dist_ptr):
        """Queue the association of n device-resident codes on the map's stream (fe: the FrontEnd whose stream
        produced them, or None)."""
        self._check(self.lib.lf_map_associate(self.m, fe.h if fe is not None else None, int(code_ptr), color_ptr and int(color_ptr),
                                              int(n), int(idx_ptr), int(dist_ptr), 1))

    def pack_block_device(self, fe, out_ptrs, n, n_frames, idx_ptr, dist_ptr, poses, step, block_ptr, block_rows):
        keep_alive, pp = self._poses(poses, n_frames)
        s = self._segs(out_ptrs)
        self._check(self.lib.lf_map_pack_block(self.m, fe.h if fe is not None else None, ctypes.byref(s), int(n), int(n_frames),
                                               int(idx_ptr), int(dist_ptr), pp, int(step), int(block_ptr), int(block_rows)))

    def update_device(self, blocks_ptr, n_blocks, block_rows):
        self._check(self.lib.lf_map_update(self.m, int(blocks_ptr), int(n_blocks), int(block_rows)))

    def step_device(self, fe, out_ptrs, n, n_frames, idx_ptr, dist_ptr, poses=None, step=0):
        """associate + update for the n segments of a batch that is resident on the device (out_ptrs: the dict given to
        FrontEnd.submit_device; frame_offset, code, color, keep, ground are read)."""
        keep_alive, pp = self._poses(poses, n_frames)
        s = self._segs(out_ptrs)
        self._check(self.lib.lf_map_step(self.m, fe.h if fe is not None else None, ctypes.byref(s), int(n), int(n_frames), pp,
                                         int(step), int(idx_ptr), int(dist_ptr)))
