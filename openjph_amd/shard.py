"""Tile sharding across the GPUs of one node.

Tiles are fully independent in JPEG 2000 -- own DWT, quantisation, code-blocks, packets and
tile-part (reference: ojph_codestream_local.cpp:113-180, ojph_tile.cpp:584-774) -- so the path
shards by contiguous runs of tiles with NO data-path collective: every rank (one process per GPU)
runs convert -> DWT -> HT block coder over its own tiles.  The only exchange is the final
codestream gather: the per-tile Psot lengths (tiny) and the variable-length tile-part byte ranges
travel to rank 0, which prepends the main header (with the TLM marker when requested) and appends
EOC.  Decoding mirrors it: every rank parses the (small) headers, decodes its tiles into its rows
of the frame; a gather of the image is only needed if one process wants the whole frame.

torch.distributed is plumbing (backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the
CPU tests); the codec work is behind the C ABI.
"""
import numpy as np


def tile_range(num_tiles: int, rank: int, world: int):
    """Contiguous, balanced run of tiles for `rank`: (first, count).  The first num_tiles % world
    ranks take one extra tile, so that a rank's output is a contiguous run of tile-parts."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(num_tiles, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def assemble(main_header: bytes, tile_parts) -> bytes:
    """main header | tile-parts in tile order | EOC"""
    return b"".join([main_header] + list(tile_parts) + [b"\xff\xd9"])


def gather_bytes(payload, group=None, dst=0, device=None, as_tensors=False):
    """Variable-length gather (gatherv) of one byte string per rank to `dst`: -> (list of per-rank byte strings
    on `dst`, None elsewhere; the sizes, on every rank).

    payload: bytes, or a uint8 torch tensor (on the device for nccl / RCCL: the tile-parts as
    Encoder.finish_tiles_device leaves them, so that they travel GPU -> GPU over xGMI without a host round trip).
    Two steps, as the path needs: an all-gather of the LENGTHS (every rank learns the prefix-sum offsets -- 8 bytes
    per rank), then only the actual bytes travel, and only to `dst`: point-to-point sends matched by receives of
    the exact sizes (gatherv to rank 0, not an all-gather of padded buffers).
    as_tensors: `dst` gets the uint8 tensors as they arrived (on `device`) instead of host byte strings."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = device if device is not None else "cpu"
    if isinstance(payload, (bytes, bytearray, memoryview)):
        t = torch.frombuffer(bytearray(payload), dtype=torch.uint8) if len(payload) else torch.zeros(0, dtype=torch.uint8)
    else:
        t = payload
    t = t.to(dev).contiguous()
    n = torch.tensor([t.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    if rank != dst:
        if sizes[rank]:
            dist.send(t, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
        return None, sizes
    bufs = [t if r == rank else torch.empty(sizes[r], dtype=torch.uint8, device=dev) for r in range(world)]
    reqs = [dist.irecv(bufs[r], src=dist.get_global_rank(group, r) if group is not None else r, group=group)
            for r in range(world) if r != rank and sizes[r]]
    for q in reqs:
        q.wait()
    if as_tensors:
        return bufs, sizes
    return [bytes(v.numpy().tobytes()) for v in to_host(bufs)], sizes


_PINNED = {}


def to_host(bufs):
    """uint8 tensors (on a device, or already on the host) -> host views, in order, inside ONE pinned buffer that is kept
    from call to call: the copies run at link speed (a pageable destination, what Tensor.cpu() gives, went at 3 GB/s for
    the 450 MB of a 16K frame -- 150 ms against the 3 ms the frame takes to code)"""
    import torch
    total = sum(int(b.numel()) for b in bufs)
    if not any(b.is_cuda for b in bufs):
        return [b for b in bufs]
    host = _PINNED.get("buf")
    if host is None or host.numel() < total:
        host = torch.empty(max(total, 1 << 20), dtype=torch.uint8).pin_memory()
        _PINNED["buf"] = host
    out, at = [], 0
    for b in bufs:
        n = int(b.numel())
        v = host[at:at + n]
        if n:
            v.copy_(b, non_blocking=True)
        out.append(v); at += n
    torch.cuda.synchronize()
    return out


class HostGather:
    """The final gather of a node WITHOUT a receiving GPU: one shared-memory segment for the codestream, every rank copies its
    tile-parts from its own GPU over its own link straight to their place in it (the offsets are a prefix sum of the lengths
    every rank already exchanges for the main header).  Where the RCCL gatherv funnels the whole codestream through rank 0's
    GPU and then through rank 0's one link to the host, here N links work at once and nothing crosses xGMI.  Ranks of ONE node
    only (POSIX shared memory).  Collective: every rank of the group constructs it, calls place() per frame, and close().

    capacity: bytes of the segment (a bound of the codestream's size).  register: make the segment known to the HIP runtime
    (pinned, link-speed copies) -- off for host-only runs (the gloo tests)."""

    def __init__(self, capacity: int, group=None, register=True):
        import torch.distributed as dist
        from multiprocessing import shared_memory
        self._dist = dist.is_available() and dist.is_initialized()       # (a single process: the same calls, nobody to meet)
        self.rank, self.world = (dist.get_rank(group), dist.get_world_size(group)) if self._dist else (0, 1)
        self.group = group
        name = [None]
        if self.rank == 0:
            self._shm = shared_memory.SharedMemory(create=True, size=int(capacity))
            name[0] = self._shm.name
        if self._dist:
            dist.broadcast_object_list(name, src=0, group=group)
        if self.rank != 0:
            self._shm = shared_memory.SharedMemory(name=name[0])
            try:                                              # (Python < 3.13 lets every attaching process's tracker unlink the
                from multiprocessing import resource_tracker  #  segment at exit: it is rank 0's to remove)
                resource_tracker.unregister(self._shm._name, "shared_memory")
            except Exception:
                pass
        self.capacity = int(capacity)
        self.view = np.frombuffer(self._shm.buf, dtype=np.uint8, count=self.capacity)
        self._registered = False
        if register:
            from . import capi
            rc = capi.lib().ojphgpu_host_register(self.view.ctypes.data, self.capacity)
            self._registered = rc == 0
        if self._dist:
            dist.barrier(group=group)

    def place(self, part, lengths_of_all_ranks, header: bytes = None):
        """part: this rank's tile-parts (uint8 tensor on its device, or on the host, or bytes); lengths_of_all_ranks: the
        byte counts of every rank's part, in rank order; header: the main header (rank 0 passes it).  Returns the total
        length once every rank's bytes are in place (a barrier closes the call); rank 0 appends EOC."""
        import torch
        import torch.distributed as dist
        hlen = torch.tensor([len(header) if (self.rank == 0 and header is not None) else 0], dtype=torch.int64)
        if self._dist:
            if torch.cuda.is_available() and dist.get_backend(self.group) == "nccl":
                hlen = hlen.cuda(); dist.broadcast(hlen, src=0, group=self.group); hlen = hlen.cpu()
            else:
                dist.broadcast(hlen, src=0, group=self.group)
        at = int(hlen.item()) + int(sum(int(x) for x in lengths_of_all_ranks[:self.rank]))
        total = int(hlen.item()) + int(sum(int(x) for x in lengths_of_all_ranks)) + 2
        if total > self.capacity:
            raise ValueError("codestream of %d bytes does not fit the segment (%d)" % (total, self.capacity))
        n = int(lengths_of_all_ranks[self.rank])
        dst = torch.from_numpy(self.view[at:at + n])
        if hasattr(part, "is_cuda"):
            if n:
                dst.copy_(part[:n], non_blocking=True)
            if part.is_cuda:
                torch.cuda.synchronize(part.device)
        elif n:
            self.view[at:at + n] = np.frombuffer(part, dtype=np.uint8, count=n)
        if self.rank == 0:
            if header is not None:
                self.view[:len(header)] = np.frombuffer(header, dtype=np.uint8)
            self.view[total - 2] = 0xFF; self.view[total - 1] = 0xD9
        if self._dist:
            dist.barrier(group=self.group)
        return total

    def close(self):
        import torch.distributed as dist
        if self._dist:
            dist.barrier(group=self.group)
        if self._registered:
            from . import capi
            capi.lib().ojphgpu_host_unregister(self.view.ctypes.data)
            self._registered = False
        self.view = None
        self._shm.close()
        if self.rank == 0:
            self._shm.unlink()


def gather_tile_lengths(lens: np.ndarray, num_tiles: int, first: int, group=None, device=None, parts_per_tile=1):
    """All ranks learn Psot of every tile-part (needed by the TLM marker and for file offsets);
    lens holds parts_per_tile entries per tile of this rank's run."""
    import torch
    import torch.distributed as dist
    dev = device if device is not None else "cpu"
    full = torch.zeros(num_tiles * parts_per_tile, dtype=torch.int64, device=dev)
    if len(lens):
        full[first * parts_per_tile:first * parts_per_tile + len(lens)] = torch.from_numpy(np.asarray(lens, dtype=np.int64)).to(dev)
    dist.all_reduce(full, op=dist.ReduceOp.SUM, group=group)
    return full.cpu().numpy().astype(np.uint32)


_NODE = {}          # per process group: do its ranks share this node, and the node's HostGather segment


def ranks_share_a_node(group=None) -> bool:
    """every rank of the group runs on this host (asked once per group)"""
    import socket
    import torch.distributed as dist
    key = ("same", id(group))
    if key not in _NODE:
        names = [None] * dist.get_world_size(group)
        dist.all_gather_object(names, socket.gethostname(), group=group)
        _NODE[key] = len(set(names)) == 1
    return _NODE[key]


def gather_mode(group=None, gather=None) -> str:
    """How the tile-parts of the ranks meet: "host" -- the node's shared segment (HostGather: every rank copies its own
    tile-parts over its own link, nothing passes through rank 0's GPU; SURVEY.md section 8(e): "simpler and equally fast since
    the data must reach host memory for the file write anyway") -- is the DEFAULT whenever all ranks share a node; "rccl" --
    the gatherv to rank 0 over the process group (RCCL send / recv over xGMI on the GPU box, gloo in the CPU tests) -- when
    they do not, or when asked for: gather="rccl" / OJPHGPU_GATHER=rccl."""
    import os
    mode = gather or os.environ.get("OJPHGPU_GATHER", "host")
    if mode not in ("host", "rccl"):
        raise ValueError("gather must be 'host' or 'rccl'")
    return mode if mode == "rccl" or ranks_share_a_node(group) else "rccl"


def node_segment(need: int, group=None):
    """the group's HostGather segment, grown (collectively: every rank sees the same `need`) when a codestream outgrows it"""
    import torch
    key = ("seg", id(group))
    hg = _NODE.get(key)
    if hg is None or hg.capacity < need:
        if hg is not None:
            hg.close()
        hg = HostGather(max(int(need * 1.5), 1 << 20), group, register=torch.cuda.is_available())
        _NODE[key] = hg
    return hg


def close_node_segment(group=None):
    """collective: releases the group's HostGather segment (before the process group goes away)"""
    hg = _NODE.pop(("seg", id(group)), None)
    if hg is not None:
        hg.close()
    _NODE.pop(("same", id(group)), None)


def encode_sharded(encode_tiles, plan, group=None, device=None, gather=None):
    """encode_tiles(first, count) -> (tile-part bytes or uint8 tensor, Psot array) for this rank's tile run (the GPU
    encoder's finish_tiles / finish_tiles_device, or the oracle pipeline in the CPU tests).  Returns the whole
    codestream on rank 0, None elsewhere.  The tile-parts meet as gather_mode() says: in the node's shared host segment by
    default, by the gatherv to rank 0 when asked for (or when the ranks span nodes)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    first, count = tile_range(plan.num_tiles, rank, world)
    part, lens = encode_tiles(first, count) if count else (b"", np.zeros(0, np.uint32))
    all_lens = gather_tile_lengths(lens, plan.num_tiles, first, group, device, plan.parts_per_tile)
    if gather_mode(group, gather) == "host":
        ppt = plan.parts_per_tile
        runs = [tile_range(plan.num_tiles, r, world) for r in range(world)]
        sizes = [int(np.asarray(all_lens[f * ppt:(f + c) * ppt], dtype=np.uint64).sum()) for f, c in runs]
        header = plan.t2_main_header(all_lens)               # (every rank: the segment's size follows from its length)
        hg = node_segment(len(header) + sum(sizes) + 2, group)
        n = hg.place(part, sizes, header if rank == 0 else None)
        return bytes(hg.view[:n]) if rank == 0 else None
    parts, _ = gather_bytes(part, group, 0, device)
    if rank != 0:
        return None
    return assemble(plan.t2_main_header(all_lens), parts)


def encode_sharded_to_host(encode_tiles, plan, gather: HostGather, group=None, device=None):
    """encode_sharded with the node's HostGather instead of the gatherv to rank 0: every rank places its tile-parts itself.
    encode_tiles(first, count) -> (tile-part bytes / uint8 tensor on the rank's device, Psot array).  Returns the length of
    the codestream (it lies at the start of gather.view on every rank)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    first, count = tile_range(plan.num_tiles, rank, world)
    part, lens = encode_tiles(first, count) if count else (b"", np.zeros(0, np.uint32))
    all_lens = gather_tile_lengths(lens, plan.num_tiles, first, group, device, plan.parts_per_tile)
    ppt = plan.parts_per_tile
    sizes = [int(np.asarray(all_lens[tile_range(plan.num_tiles, r, world)[0] * ppt:
                                     (tile_range(plan.num_tiles, r, world)[0] + tile_range(plan.num_tiles, r, world)[1]) * ppt], dtype=np.uint64).sum())
             for r in range(world)]
    header = plan.t2_main_header(all_lens) if rank == 0 else None
    return gather.place(part, sizes, header)
