"""Multi-GPU decomposition of the frame (host-side plumbing, torch.distributed only).

The reference shards work by scanline across threads (core/tracer/tracer.cpp:1144-1145,
5385-5386) and culls by 8-row screen tiles (core/engine/engine.h:38-39).  Across GPUs we shard
by contiguous blocks of 8-row TILE ROWS:

  * a step renders `world` frames ("frames in flight": consecutive animation frames in
    production, the same frozen snapshot in bench.py);
  * frame f is cut into `world` blocks of tile rows; block b of frame f is rendered by rank
    (b - f) mod world, i.e. rank r renders block (r + f) mod world of every frame, so each rank
    touches every screen region once per step (sky/geometry imbalance cancels);
  * ONE exchange per step (grouped point-to-point sends = an all-to-all over RCCL/xGMI, one
    message per peer link) moves the blocks so that rank f ends the step owning the complete
    frame f.  No reduction is needed: pixels are written exactly once.

Per-GPU work is one frame's worth of rays per step whatever `world` is (weak scaling).
With world == 1 nothing is exchanged.
"""
import torch
import torch.distributed as dist

TILE_H = 8


def block_rows(height, world):
    """Row boundaries of the `world` tile-row blocks: block b = rows [lo[b], lo[b+1])."""
    groups = (height + TILE_H - 1) // TILE_H
    return [min(height, ((groups * b) // world) * TILE_H) for b in range(world + 1)]


def block_of(rank, frame, world):
    """Block of frame `frame` that rank `rank` renders."""
    return (rank + frame) % world


class FrameExchange:
    """Assembles, on rank f, the complete frame f from the blocks every rank rendered."""

    def __init__(self, height, width, world, rank):
        self.h, self.w, self.world, self.rank = height, width, world, rank
        self.lo = block_rows(height, world)

    def my_rows(self, frame):
        b = block_of(self.rank, frame, self.world)
        return self.lo[b], self.lo[b + 1]

    def exchange(self, frames, final, group=None):
        """frames: list of `world` [h,w] tensors (rank's blocks rendered in place);
        final: [h,w] tensor receiving frame `rank`.  Returns after the transfers completed
        (on the current stream for NCCL)."""
        self.exchange_many([(frames, final)], group)

    def exchange_many(self, steps, group=None):
        """The exchanges of several steps as ONE grouped call: `steps` = [(frames, final), ...] in the same
        order on every rank.  Fewer, larger collectives: a grouped send/recv has a fixed cost of tens of
        microseconds on RCCL, comparable to a whole step of demo-scene rendering."""
        world, rank, lo = self.world, self.rank, self.lo
        if world == 1:
            for frames, final in steps:
                final.copy_(frames[0])
            return
        # gloo has no device-memory send/recv: stage through host memory (CPU tests, one-GPU rehearsals)
        staged = steps[0][1].is_cuda and dist.get_backend(group) == "gloo"
        if staged:
            torch.cuda.current_stream().synchronize()
        recv_host = []
        ops, keep = [], []
        for frames, final in steps:
            for peer in range(world):
                sb = block_of(rank, peer, world)            # my block of frame `peer` goes to rank `peer`
                rb = block_of(peer, rank, world)            # peer's block of frame `rank` comes to me
                src = frames[peer][lo[sb]:lo[sb + 1]]
                dst = final[lo[rb]:lo[rb + 1]]
                if peer == rank:
                    dst.copy_(src)
                    continue
                if staged:
                    src = src.cpu()
                    recv_host.append((torch.empty(dst.shape, dtype=dst.dtype), dst))
                    dst = recv_host[-1][0]
                if src.numel():
                    ops.append(dist.P2POp(dist.isend, src, peer, group))
                if dst.numel():
                    ops.append(dist.P2POp(dist.irecv, dst, peer, group))
                keep += [src, dst]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for host, dev in recv_host:
            dev.copy_(host)

    def gather_many(self, steps, root=0, group=None):
        """Gather mode ("final image gathered", BASELINE.json north_star): every frame of every step ends complete
        on rank `root` instead of frame f on rank f -- what a display or a file writer on one rank needs.
        `steps` = [(frames, finals)], finals = list of `world` [h,w] tensors on the root (ignored elsewhere).
        One grouped send/recv for all steps; the root receives (world - 1) / world of every frame, so this
        mode is bound by the root's links (7 x 153 GB/s xGMI in an 8-GPU node) where the all-to-all of
        exchange_many spreads the traffic over all links."""
        world, rank, lo = self.world, self.rank, self.lo
        if world == 1:
            for frames, finals in steps:
                finals[0].copy_(frames[0])
            return
        staged = steps[0][0][0].is_cuda and dist.get_backend(group) == "gloo"
        if staged:
            torch.cuda.current_stream().synchronize()
        recv_host, ops, keep = [], [], []
        for frames, finals in steps:
            for f in range(world):
                if rank == root:
                    for peer in range(world):
                        b = block_of(peer, f, world)
                        dst = finals[f][lo[b]:lo[b + 1]]
                        if peer == root:
                            dst.copy_(frames[f][lo[b]:lo[b + 1]])
                            continue
                        if staged:
                            recv_host.append((torch.empty(dst.shape, dtype=dst.dtype), dst))
                            dst = recv_host[-1][0]
                        if dst.numel():
                            ops.append(dist.P2POp(dist.irecv, dst, peer, group))
                        keep.append(dst)
                else:
                    b = block_of(rank, f, world)
                    src = frames[f][lo[b]:lo[b + 1]]
                    if staged:
                        src = src.cpu()
                    if src.numel():
                        ops.append(dist.P2POp(dist.isend, src, root, group))
                    keep.append(src)
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for host, dev in recv_host:
            dev.copy_(host)


class SplitFrame:
    """ONE frame over `world` GPUs (BASELINE.json north_star: "shards framebuffer tiles across the 8 GPUs of one node with the
    final image gathered"): strong scaling.  The 8-row tile rows of the frame are dealt round-robin -- tile row g belongs to
    rank g mod world, the reference's own row interleave across threads (tracer.cpp:1144-1145) at tile-row granularity, so sky
    and geometry spread evenly -- every rank renders its tile rows into a full-size frame buffer (qr_scene_set_tile_rows(rank,
    world)), compacts them, and one grouped send per rank brings them to the root, which scatters them into the final frame.
    Frame buffers are allocated with the height rounded up to whole tile rows (alloc_rows) so that they view as
    [tile row, 8, width]."""

    def __init__(self, height, width, world, rank):
        self.h, self.w, self.world, self.rank = height, width, world, rank
        self.groups = (height + TILE_H - 1) // TILE_H
        self.alloc_rows = self.groups * TILE_H

    def my_groups(self, rank=None):
        return range(self.rank if rank is None else rank, self.groups, self.world)

    def _view(self, frame):
        return frame[: self.alloc_rows].view(self.groups, TILE_H, self.w)

    def gather(self, steps, root=0, group=None):
        """steps = [(frame, final)]: `frame` this rank's rendered buffer (alloc_rows x w), `final` the root's assembled frame
        (alloc_rows x w; ignored elsewhere).  One grouped send/recv for all steps."""
        world, rank = self.world, self.rank
        if world == 1:
            for frame, final in steps:
                final.copy_(frame)
            return
        staged = steps[0][0].is_cuda and dist.get_backend(group) == "gloo"
        if staged:
            torch.cuda.current_stream().synchronize()
        ops, keep, scatter = [], [], []
        for frame, final in steps:
            if rank == root:
                fv = self._view(final)
                fv[rank::world].copy_(self._view(frame)[rank::world])
                for peer in range(world):
                    if peer == root:
                        continue
                    n = len(self.my_groups(peer))
                    if n == 0:
                        continue
                    buf = torch.empty((n, TILE_H, self.w), dtype=final.dtype, device="cpu" if staged else final.device)
                    ops.append(dist.P2POp(dist.irecv, buf, peer, group))
                    scatter.append((buf, fv, peer))
            else:
                src = self._view(frame)[rank::world].contiguous()
                if staged:
                    src = src.cpu()
                if src.numel():
                    ops.append(dist.P2POp(dist.isend, src, root, group))
                keep.append(src)
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for buf, fv, peer in scatter:
            fv[peer::world].copy_(buf)
