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


def encode_sharded(encode_tiles, plan, group=None, device=None):
    """encode_tiles(first, count) -> (tile-part bytes, Psot array) for this rank's tile run (the GPU
    encoder's finish_tiles, or the oracle pipeline in the CPU tests).  Returns the whole
    codestream on rank 0, None elsewhere."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    first, count = tile_range(plan.num_tiles, rank, world)
    part, lens = encode_tiles(first, count) if count else (b"", np.zeros(0, np.uint32))
    all_lens = gather_tile_lengths(lens, plan.num_tiles, first, group, device, plan.parts_per_tile)
    parts, _ = gather_bytes(part, group, 0, device)
    if rank != 0:
        return None
    return assemble(plan.t2_main_header(all_lens), parts)
