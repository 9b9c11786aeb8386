"""Multi-GPU merge of per-rank segment lists.

Frames shard across ranks as independent batches (no data-path collective for detect ->
describe -> project -> sanity).  The only exchange is ONE all-gather per step of fixed-capacity
segment blocks, so that every rank can append the same segments, in the same order, to its
replica of the live map before the next association (the reference's map is an append-only
list, src/show_map/src/show_map.py:28-42).  Association itself needs no collective: the map
is replicated and each rank matches only its own frames' descriptors.

Block layout (uint8, one row per segment, padded to `capacity` rows):
    bytes 0..31  binary LBD code      byte 32  keep flag      byte 33  colour
The segment count travels in a separate tiny all-gather.  Works on any torch.distributed
backend: "nccl" (= RCCL over xGMI on MI355X) for device tensors, "gloo" for the CPU tests.
"""
import torch
import torch.distributed as dist

BLOCK_COLS = 34


def pack_block(block, code, keep, color, n):
    """Write the first n segments into the padded block (in place); rows >= n are left as they are."""
    n = min(int(n), block.shape[0])
    block[:n, :32] = code[:n]
    block[:n, 32] = keep[:n]
    block[:n, 33] = color[:n]
    return n


def all_gather_blocks(block, n, gathered=None, counts=None):
    """All-gather one padded block per rank.  Returns (gathered [world, capacity, 34], counts [world])."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    cap = block.shape[0]
    if gathered is None:
        gathered = torch.empty(world * cap, BLOCK_COLS, dtype=torch.uint8, device=block.device)
    if counts is None:
        counts = torch.empty(world, dtype=torch.int32, device=block.device)
    mine = torch.tensor([n], dtype=torch.int32, device=block.device)
    if world == 1:
        gathered.view(1, cap, BLOCK_COLS)[0].copy_(block)
        counts[0] = n
    else:
        dist.all_gather_into_tensor(counts, mine)
        dist.all_gather_into_tensor(gathered, block)
    return gathered.view(world, cap, BLOCK_COLS), counts


def merged_codes(gathered, counts, kept_only=True):
    """Concatenate the valid rows rank-major / segment-minor: identical on every rank, and equal to
    the single-GPU order when frames were dealt to ranks in contiguous chunks."""
    out = []
    for r in range(gathered.shape[0]):
        n = int(counts[r])
        rows = gathered[r, :n]
        if kept_only:
            rows = rows[rows[:, 32] != 0]
        out.append(rows[:, :32])
    return torch.cat(out, dim=0) if out else gathered.new_zeros((0, 32))


class LiveMap(object):
    """Append-only code map with fixed capacity (oldest entries are overwritten once full)."""

    def __init__(self, capacity, device="cpu", initial=None):
        self.codes = torch.zeros(capacity, 32, dtype=torch.uint8, device=device)
        self.size = 0
        self.head = 0
        if initial is not None:
            self.append(initial)

    def append(self, codes):
        cap = self.codes.shape[0]
        n = codes.shape[0]
        if n >= cap:
            self.codes.copy_(codes[-cap:])
            self.size, self.head = cap, 0
            return
        end = self.head + n
        if end <= cap:
            self.codes[self.head:end] = codes
        else:
            k = cap - self.head
            self.codes[self.head:] = codes[:k]
            self.codes[: end - cap] = codes[k:]
        self.head = end % cap
        self.size = min(cap, self.size + n)

    def view(self):
        return self.codes[: self.size]
