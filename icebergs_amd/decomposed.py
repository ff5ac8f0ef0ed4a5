"""Domain-decomposed runs: one tile of the ocean grid and one handle per rank, bergs migrating between ranks.

This is the reference's own parallel layout (send_bergs_to_other_pes, icebergs_framework.F90:2997-3247) with
`torch.distributed` point-to-point messages where the reference has mpp_send / mpp_recv: per exchange pass the two counts,
then the records in the reference's wire format (pack_berg_into_buffer2, FW:3250-3301).  East/west first, then north/south
(FW:3022-3230), so a berg that leaves through a corner hops twice.  Inside one node `distributed.py` replicates the grid and
shards the bergs by index instead (no exchange at all); this module is for grids that should not be replicated, or for
several nodes.

The tile only needs `pack_pair(axis)` -> (records for the east/north neighbour, records for the west/south neighbour) and
`unpack_pair(from_west_or_south, from_east_or_north)`: `HipTile` wraps an `Icebergs` handle; the CPU tests plug the oracle
in the same way.
"""
import numpy as np
import torch

WIDTH = 34  # buffer_width without bonds, FW:21


class HipTile:
    """the migration calls of one handle (kid_pack_emigrants_pair / kid_unpack_immigrants_pair)"""

    def __init__(self, icebergs):
        self.ib = icebergs

    def pack_pair(self, axis):
        return self.ib.pack_emigrants_pair(axis)

    def unpack_pair(self, from_lo, from_hi):
        self.ib.unpack_immigrants_pair(from_lo, from_hi)


class TileExchange:
    """ranks laid out as an ntx x nty array of tiles, rank = ty * ntx + tx (x fastest, like the reference's domain layout);
    a tile on the edge of the box has no neighbour there (NULL_PE: its leavers are packed and dropped, FW:3050) unless
    cyclic_x closes the zonal direction"""

    def __init__(self, ntx, nty, dist, rank=None, cyclic_x=False, device="cpu"):
        self.ntx, self.nty, self.dist, self.device = int(ntx), int(nty), dist, device
        self.rank = dist.get_rank() if rank is None else int(rank)
        assert dist.get_world_size() == self.ntx * self.nty, "one rank per tile"
        self.tx, self.ty = self.rank % self.ntx, self.rank // self.ntx
        self.cyclic_x = bool(cyclic_x)
        self.sent = 0          # records handed to neighbours so far
        self.received = 0

    def neighbour(self, dx, dy):
        tx, ty = self.tx + dx, self.ty + dy
        if self.cyclic_x and self.ntx > 1:
            tx %= self.ntx
        if tx < 0 or tx >= self.ntx or ty < 0 or ty >= self.nty:
            return None
        return ty * self.ntx + tx

    def _swap(self, to_hi, hi, to_lo, lo):
        """send `to_hi` to rank `hi` and `to_lo` to rank `lo`; returns (from_lo, from_hi).  Counts first, then the records
        (FW:3050-3097: mpp_send of nbergs_to_send, then of the buffer)"""
        dist = self.dist
        if hi is not None and hi == lo:        # two tiles across a cyclic direction: both neighbours are the same rank
            return self._swap_same_peer(to_hi, to_lo, hi)
        out = {hi: to_hi, lo: to_lo}
        counts_in = {}
        ops, keep = [], []
        for peer, recs in ((hi, to_hi), (lo, to_lo)):
            if peer is None:
                continue
            c = torch.tensor([len(recs)], dtype=torch.int64, device=self.device)
            r = torch.zeros(1, dtype=torch.int64, device=self.device)
            keep.append(c)
            counts_in[peer] = r
            ops += [dist.P2POp(dist.isend, c, peer), dist.P2POp(dist.irecv, r, peer)]
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        ops, got = [], {}
        for peer in (hi, lo):
            if peer is None:
                continue
            if len(out[peer]):
                t = torch.from_numpy(np.ascontiguousarray(out[peer])).to(self.device)
                keep.append(t)
                ops.append(dist.P2POp(dist.isend, t, peer))
            n_in = int(counts_in[peer].item())
            if n_in:
                got[peer] = torch.empty((n_in, WIDTH), dtype=torch.float64, device=self.device)
                ops.append(dist.P2POp(dist.irecv, got[peer], peer))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        empty = np.empty((0, WIDTH))
        from_lo = got[lo].cpu().numpy() if lo in got else empty
        from_hi = got[hi].cpu().numpy() if hi in got else empty
        self.sent += (len(to_hi) if hi is not None else 0) + (len(to_lo) if lo is not None else 0)
        self.received += len(from_lo) + len(from_hi)
        return from_lo, from_hi

    def _swap_same_peer(self, to_hi, to_lo, peer):
        """ntx == 2 with cyclic_x: the east and the west neighbour are the same rank.  One message each way carrying both
        groups, the first row saying how many belong to the first group."""
        dist = self.dist
        mine = np.concatenate([np.full((1, WIDTH), float(len(to_hi))), to_hi, to_lo]) if len(to_hi) + len(to_lo) else np.full((1, WIDTH), 0.0)
        c = torch.tensor([len(mine)], dtype=torch.int64, device=self.device)
        r = torch.zeros(1, dtype=torch.int64, device=self.device)
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, c, peer), dist.P2POp(dist.irecv, r, peer)]):
            w.wait()
        t = torch.from_numpy(np.ascontiguousarray(mine)).to(self.device)
        g = torch.empty((int(r.item()), WIDTH), dtype=torch.float64, device=self.device)
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, t, peer), dist.P2POp(dist.irecv, g, peer)]):
            w.wait()
        theirs = g.cpu().numpy()
        k = int(theirs[0, 0])
        # what the peer sent east arrives here from the west, and the other way round
        from_lo, from_hi = theirs[1:1 + k], theirs[1 + k:]
        self.sent += len(to_hi) + len(to_lo)
        self.received += len(from_lo) + len(from_hi)
        return from_lo, from_hi

    def exchange(self, tile):
        """send_bergs_to_other_pes for this rank's tile: call it between evolve_icebergs and thermodynamics (IB:5447)"""
        for axis, (dx, dy) in enumerate(((1, 0), (0, 1))):
            to_hi, to_lo = tile.pack_pair(axis)
            hi, lo = self.neighbour(dx, dy), self.neighbour(-dx, -dy)
            from_lo, from_hi = self._swap(to_hi, hi, to_lo, lo)
            if len(from_lo) or len(from_hi):
                tile.unpack_pair(from_lo, from_hi)            # from the west / south first, FW:3064, 3160
