"""Multi-GPU path: tiles shard across ranks with no data-path collective;
one gather of variable-length tile-parts at the end.  The N>1 logic is covered here with
world_size-2/3 `gloo` process groups on CPU (device stages replaced by the oracle pipeline), and on
the GPU box with several tile-range encoders/decoders sharing the one GPU."""
import os
import socket

import numpy as np
import pytest

from tests.synth import synth_image


def test_tile_range_is_a_partition():
    from openjph_amd.shard import tile_range
    for nt in (1, 2, 7, 12, 256):
        for world in (1, 2, 3, 8):
            runs = [tile_range(nt, r, world) for r in range(world)]
            pos = 0
            for first, count in runs:
                assert first == pos and count >= 0
                pos += count
            assert pos == nt
            counts = [c for _, c in runs]
            assert max(counts) - min(counts) <= 1
    with pytest.raises(ValueError):
        tile_range(4, 2, 2)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, case, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from openjph_amd import shard
        from openjph_amd.plan import Plan, make_params
        from tests import cpu_pipeline as cp
        img = synth_image(case["nc"], case["h"], case["w"], case["bd"], seed=3)
        kw = dict(case["kw"], bit_depth=case["bd"])
        plan = Plan(make_params(case["w"], case["h"], case["nc"], **kw))
        assert shard.gather_mode() == "host"                 # the ranks share this node: the shared segment is the default
        cs = shard.encode_sharded(lambda first, count: cp.encode_tiles(plan, img, first, count), plan)
        # ... and the gatherv to rank 0 over the process group, behind its switch: the same bytes
        cs2 = shard.encode_sharded(lambda first, count: cp.encode_tiles(plan, img, first, count), plan, gather="rccl")
        if rank == 0 and cs2 != cs:
            cs = b"the two gathers differ"
        shard.close_node_segment()
        # the same frame with a caller-owned HostGather (every rank places its own tile-parts; no receiving rank)
        cap = [len(cs) + 4096 if rank == 0 else 0]
        dist.broadcast_object_list(cap, src=0)
        hg = shard.HostGather(cap[0], register=False)
        n = shard.encode_sharded_to_host(lambda first, count: cp.encode_tiles(plan, img, first, count), plan, hg)
        same = bytes(hg.view[:n]) == (cs if rank == 0 else bytes(hg.view[:n]))
        if rank == 0:
            q.put(cs if same and n == len(cs) else b"host gather differs")
        hg.close()
        dist.barrier()
    finally:
        dist.destroy_process_group()


SHARD_CASES = [
    dict(nc=1, h=300, w=500, bd=16, kw=dict(tile=(128, 128))),                       # 12 tiles, C4-like
    dict(nc=1, h=256, w=256, bd=8, kw=dict(tile=(64, 64), tlm=True)),                # TLM needs every Psot
    dict(nc=3, h=100, w=260, bd=8, kw=dict(tile=(128, 128), color_transform=True)),  # 3 tiles on 2 ranks
    dict(nc=1, h=64, w=64, bd=8, kw=dict()),                                         # 1 tile: rank 1 idle
    dict(nc=3, h=200, w=200, bd=8, kw=dict(tile=(64, 64), tlm=True, prog_order="CPRL", tileparts="C")),   # 3 tile-parts per tile in the TLM
]


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("ci", range(len(SHARD_CASES)))
def test_sharded_encode_gloo_matches_single_process(ci, world):
    import torch.multiprocessing as mp
    from tests import cpu_pipeline as cp
    case = SHARD_CASES[ci]
    if world == 3 and ci not in (0, 1, 4):
        pytest.skip("covered at world_size 2")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, case, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    img = synth_image(case["nc"], case["h"], case["w"], case["bd"], seed=3)
    want, *_ = cp.encode(img, bit_depth=case["bd"], **case["kw"])
    assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(len(SHARD_CASES)))
def test_tile_range_encoders_and_decoders_on_gpu(ci):
    """Several tile-range encoders (what the ranks of a node run) == the whole-frame encoder; several
    tile-range decoders writing one frame buffer == the whole-frame decoder."""
    import torch
    from openjph_amd import codec, shard
    from openjph_amd.plan import Plan, make_params
    case = SHARD_CASES[ci]
    img = synth_image(case["nc"], case["h"], case["w"], case["bd"], seed=3)
    kw = dict(case["kw"], bit_depth=case["bd"])
    plan = Plan(make_params(case["w"], case["h"], case["nc"], **kw))
    whole = codec.Encoder(plan=plan).encode(img)
    d_img = torch.from_numpy(img).cuda()
    for world in (2, 3):
        parts, lens = [], []
        for r in range(world):
            first, count = shard.tile_range(plan.num_tiles, r, world)
            if count == 0:
                parts.append(b""); continue
            enc = codec.Encoder(plan=plan, tiles=(first, count))
            enc.run_device(d_img)
            b, l = enc.finish_tiles()
            parts.append(b); lens.append(l)
        got = shard.assemble(plan.t2_main_header(np.concatenate(lens)), parts)
        assert got == whole
        out = torch.zeros_like(d_img)
        for r in range(world):
            first, count = shard.tile_range(plan.num_tiles, r, world)
            if count:
                dec = codec.Decoder(whole, tiles=(first, count))
                dec.run_device(out)
                assert dec.failed_blocks() == 0
        assert np.array_equal(out.cpu().numpy(), codec.decode(whole))


@pytest.mark.gpu
def test_tile_parts_assembled_on_the_device_and_gathered_over_rccl():
    """the RCCL branch on the one GPU of the box: a world_size-1 "nccl" group (RCCL refuses two ranks on one
    device) runs the length all-gather on the device; the tile-parts assembled in HBM
    (ojphgpu_encoder_finish_tiles_device) equal the host-assembled ones byte for byte"""
    import torch
    import torch.distributed as dist
    from openjph_amd import codec, shard
    from openjph_amd.plan import Plan, make_params
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    img = synth_image(1, 300, 500, 16, seed=3)
    plan = Plan(make_params(500, 300, 1, bit_depth=16, tile=(128, 128), tlm=True))
    want = codec.Encoder(plan=plan).encode(img)
    d_img = torch.from_numpy(img).cuda()
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        dev = torch.device("cuda", 0)
        parts, lens = [], []
        for r in range(3):                                     # three "ranks" share the GPU, one after the other
            first, count = shard.tile_range(plan.num_tiles, r, 3)
            e = codec.Encoder(plan=plan, tiles=(first, count))
            e.run_device(d_img)
            dpart, ln = e.finish_tiles_device()
            hpart, ln2 = e.finish_tiles()
            assert dpart.is_cuda and bytes(dpart.cpu().numpy().tobytes()) == hpart and np.array_equal(ln, ln2)
            got, sizes = shard.gather_bytes(dpart, device=dev)  # all-gather of lengths over RCCL; no peer to send to
            assert sizes == [len(hpart)] and got == [hpart]
            all_l = shard.gather_tile_lengths(ln, plan.num_tiles, first, device=dev, parts_per_tile=plan.parts_per_tile)
            parts.append(hpart); lens.append(all_l)
        all_lens = np.sum(np.stack(lens), axis=0).astype(np.uint32)
        assert shard.assemble(plan.t2_main_header(all_lens), parts) == want
    finally:
        dist.destroy_process_group()


def _selfcheck_worker(rank, world, port, local_rank, mode, q):
    """bench.start_process_group on CPU (gloo, no device): the self-check N > 1 runs begin with"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), OJPH_BENCH_PG_TIMEOUT_S="8")
    os.environ.pop("OJPH_BENCH_ONE_GPU", None)
    import torch
    import torch.distributed as dist
    import bench
    if mode == "absent" and rank == 1:
        return                                                # a rank that never starts: the others must not wait for ever
    info = bench.start_process_group("gloo", rank, world, local_rank, None, torch, dist)
    if rank == 0:
        q.put(info)
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["ok", "same_device", "absent"])
def test_bench_multi_rank_self_check(mode):
    """`bench.py --gpus N` checks its process group before it times anything: N ranks on N distinct devices, the first
    collectives answered -- and a rank whose peers never come says so and exits (status 3) instead of hanging"""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_selfcheck_worker, args=(r, 2, port, 0 if mode == "same_device" else r, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert not p.is_alive(), "a rank hangs"
    codes = [p.exitcode for p in procs]
    if mode == "ok":
        assert codes == [0, 0]
        info = q.get(timeout=10)
        assert info["distinct_devices"] == 2 and len(info["rank_devices"]) == 2 and "ok" in info["self_check"]
    elif mode == "same_device":
        assert codes == [4, 4], codes                         # two ranks on one device: refused, loudly
    else:
        assert codes[0] == 3 and codes[1] == 0, codes         # rank 0 gave up on the rendezvous with a diagnostic
