#!/usr/bin/env python3
"""bench.py -- HTJ2K hot path on MI355X: Msamples/s for encode+decode of BASELINE.json's headline
configuration (8K 7680x4320, 12-bit, 4:4:4, irreversible 9/7, Qstep 0.001, 64x64 blocks, 5 levels).

A "step" is one pass of the hot path over one synthetic frame whose samples are already resident
in HBM: encode (sample convert -> 5 DWT levels -> HT block encode) followed by decode (HT block
decode -> 5 inverse DWT levels -> convert) of that frame's code-block bytes, also resident in HBM.
Host Tier-2 (packet headers) and PCIe are outside the timed region; DESIGN.md quotes them.

One process per GPU.  The path shards with no data-path collective: single-tile frames (the default
C3 workload) are replicas -- with N > 1 every rank codes its own independent frame, weak scaling,
value = samples of all ranks / time; tiled frames (--workload c4_...) shard by contiguous runs of
tiles of ONE frame, strong scaling, with the final tile-part gather over RCCL outside the timed
region (it is host Tier-2 + PCIe work, like the single-GPU finish()).

`python bench.py --gpus N` with N > 1 and no launcher around it starts its N ranks itself (self_launch: the driver's own
torch.distributed.run line on 127.0.0.1) and fails loudly when the node has fewer GPUs; under a launcher --gpus must equal
WORLD_SIZE.  Every default run (any N) also carries `strong_scaling_c4`: the 16K x 16K frame in 256 tiles sharded over
the ranks, with the RCCL gatherv of the tile-parts timed on its own -- one 1/2/4/8 sweep answers both the replica
(weak) and the tile-sharded (strong) question of north_star.

Prints ONE JSON line (rank 0).  Extra objects on that line:
  roofline     -- the dominant kernel of the step against the HBM roofline, from live HIP-event
                  timings on the codec's own stream (algorithmic bytes: SURVEY.md section 8(d))
  cpu_baseline -- the reference library itself (oracle/_ref, built from /root/reference) timed
                  on this box's host CPU, single thread (the library is single-threaded)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured copy ceiling

WORKLOADS = {
    # name: (width, height, comps, bit_depth, reversible, color_transform, qstep, tile)
    "c3_8k_444_12b_irv97": (7680, 4320, 3, 12, False, False, 0.001, (0, 0)),
    "c2_4k_rgb_8b_rev53": (3840, 2160, 3, 8, True, True, -1.0, (0, 0)),
    "c4_16k_gray_16b_rev53_tiled": (16384, 16384, 1, 16, True, False, -1.0, (1024, 1024)),
    "c1_256_gray_8b_rev53": (256, 256, 1, 8, True, False, -1.0, (0, 0)),
    # BASELINE config #5: independent 4K 10-bit frames coded as a batch (--frames B per step and per GPU)
    "c5_4k_444_10b_irv97_batch": (3840, 2160, 3, 10, False, False, -1.0, (0, 0)),
    # SURVEY.md section 8(f) N4 -- the paths only deep or Part-2 codestreams take, so that they have a number too:
    # 32-bit samples (more than 32 bits of precision: the reference's 64-bit sample path, int64 planes, ojph_encode_codeblock64 /
    # ojph_decode_codeblock64, the general lifting kernels)
    "c6_4k_gray_32b_rev53": (3840, 2160, 1, 32, True, False, -1.0, (0, 0)),
    # the 9/7 written as an ATK marker segment (four float lifting steps + K): the same arithmetic as c3 / c5 through the
    # general lifting kernels -- decodes to the plain 9/7's samples bit for bit (tests/test_gpu_part2.py)
    "c7_4k_444_12b_atk97": (3840, 2160, 3, 12, False, False, 0.001, (0, 0)),
}
ATK97 = {2: dict(steps=[0.443506852043971, 0.882911075530934, -0.052980118572961, -1.586134342059924], K=1.230174104914001)}
WORKLOAD_EXTRA = {          # what a workload needs beyond the tuple: make_params keywords, the sample container it exists in
    "c6_4k_gray_32b_rev53": dict(container=32),
    "c7_4k_444_12b_atk97": dict(params=dict(atk=ATK97, wavelet=2)),
}


def workload_image(name, seed_offset=0, frame=0):
    """the workload's input exactly as SURVEY.md section 8(d) specifies it (tests/synth.py survey_*: C2 and
    C3 reproduce the survey's known answers KA-3 / KA-4 with seed_offset 0); [C, H, W] int32"""
    from tests import synth
    if name.startswith("c3"):
        return synth.survey_c3(seed=1234 + seed_offset)
    if name.startswith("c2"):
        return synth.survey_c2(seed=1234 + seed_offset)
    if name.startswith("c4"):
        return synth.survey_c4(seed=1234 + seed_offset)
    if name.startswith("c5"):
        return synth.survey_c5(frame=frame, seed=1234 + seed_offset)
    if name.startswith("c6"):
        return synth.survey_c6(seed=1234 + seed_offset)
    if name.startswith("c7"):
        return synth.survey_c7(seed=1234 + seed_offset)
    return synth.c1_image()


def dwt_alg_bytes(nsamples, levels, container=32, elem=4):
    """each level reads its input once and writes its four sub-bands once, `elem`-byte elements (4; 8 on the 64-bit sample
    path); the image side of the top level moves container / 8 bytes per sample"""
    return 2.0 * elem * nsamples * sum(4.0 ** -l for l in range(levels)) - (elem - container / 8.0) * nsamples * (1 if levels else 0)


def plan_is_tiled(tile):
    return tile[0] > 0


def self_launch(n):
    """`python bench.py --gpus N` with no launcher: re-execute this script as N ranks (one process per GPU, the driver's own
    launch line: torch.distributed.run, rendezvous on 127.0.0.1).  Rank 0's JSON line is the only thing on stdout.
    Fails loudly when the node has fewer than N GPUs -- a smaller run must not pass for an N-GPU one."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < n and not os.environ.get("OJPH_BENCH_ONE_GPU"):
        sys.stderr.write("bench.py: --gpus %d asked for, %d GPU(s) visible on this node\n" % (n, have))
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="ranks = GPUs of this node (one process per GPU).  Under a launcher (WORLD_SIZE in the environment) it must "
                         "equal WORLD_SIZE; without one, N > 1 makes bench.py start its N ranks itself")
    ap.add_argument("--steps", type=int, default=1500,
                    help="timed steps (default 1500: about 2 s of GPU time on the 8K frame, long enough for an external sampler to see)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="c3_8k_444_12b_irv97", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-reps", type=int, default=2)
    ap.add_argument("--frames", type=int, default=0,
                    help="independent frames coded per step as one batch (default 1; 8 for the c5 batch workload)")
    ap.add_argument("--container", type=int, default=16, choices=(8, 16, 32),
                    help="bits of the sample containers of the frames in HBM: 16 = int16 / uint16 planes (how 9..16-bit frames "
                         "exist in files and capture buffers; the default), 8 = int8 / uint8 planes (8-bit frames), 32 = int32 planes "
                         "(what the reference's line_buf exchanges line by line)")
    ap.add_argument("--streams", type=int, default=1, choices=(1, 2),
                    help="2: the frame being encoded and the frame being decoded are issued on two HIP streams")
    ap.add_argument("--e2e-frames", type=int, default=48,
                    help="frames per frame-pipeline measurement (host memory -> codestream in host memory and back); 0 = skip")
    ap.add_argument("--calibrate", action="store_true",
                    help="also launch one elementwise kernel of known traffic (PMC unit calibration)")
    ap.add_argument("--plain", action="store_true",
                    help="only the timed steps and their per-launch event timings (no two-stream / no-overlap objects, pipelines, "
                         "strong-scaling part): what the rocprofv3 passes run, so that a trace holds ONE schedule of every launch")
    ap.add_argument("--no-strong", action="store_true",
                    help="skip the strong-scaling sub-measurement (the 16K x 16K frame in 256 tiles sharded over the ranks)")
    ap.add_argument("--strong-steps", type=int, default=100)
    args = ap.parse_args()
    if args.plain:
        args.no_strong = True; args.e2e_frames = 0

    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None:
        if args.gpus is not None and args.gpus > 1:
            sys.exit(self_launch(args.gpus))     # no launcher around us: become one (N ranks of this script, one per GPU)
        args.gpus = 1
    elif args.gpus is None:
        args.gpus = int(env_world)
    elif args.gpus != int(env_world):
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks" % (args.gpus, env_world))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # the all-core CPU figure forks worker processes: done first, before this process owns a GPU context
    cpu_all = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:          # (the CPU figures belong to the N = 1 line)
        w_, h_, nc_, bd_, rev_, ct_, qstep_, _tile = WORKLOADS[args.workload]
        try:
            cpu_all = cpu_baseline_all_cores(workload_image(args.workload), bd_, rev_, ct_, qstep_, _tile)
        except Exception as e:                     # the single-thread figure stands on its own
            cpu_all = {"value": None, "error": str(e)[:200]}
    import torch
    import torch.distributed as dist
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # OJPH_BENCH_BACKEND=gloo + OJPH_BENCH_ONE_GPU=1: smoke-test the N > 1 code path on a 1-GPU box
    # (all ranks share cuda:0, control traffic over gloo).  The driver's runs use the defaults.
    backend = os.environ.get("OJPH_BENCH_BACKEND", "nccl")
    if os.environ.get("OJPH_BENCH_ONE_GPU"):
        local_rank = 0
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist_info = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist_info = start_process_group(backend, rank, world, local_rank, dev, torch, dist)

    from openjph_amd import codec
    from openjph_amd.plan import make_params

    w, h, nc, bd, rev, ct, qstep, tile = WORKLOADS[args.workload]
    extra = WORKLOAD_EXTRA.get(args.workload, {})
    if "container" in extra:
        args.container = extra["container"]
    frames = args.frames if args.frames > 0 else (8 if "batch" in args.workload else 1)
    nsamples = w * h * nc * frames
    if frames > 1:
        img = np.stack([workload_image(args.workload, 0, rank * frames + f) for f in range(frames)])
    else:
        img = workload_image(args.workload, 0 if plan_is_tiled(tile) else rank)
    def to_dev(a):                               # the frame as it sits in HBM
        return torch.from_numpy(a.astype({8: np.int8, 16: np.int16}[args.container]) if args.container != 32 else a).to(dev)
    d_img = to_dev(img)
    params = make_params(w, h, nc, bit_depth=bd, reversible=rev, color_transform=ct, qstep=qstep, tile=tile, **extra.get("params", {}))
    from openjph_amd.plan import Plan
    from openjph_amd import shard
    plan = Plan(params)
    # single-tile frames do not shard (replicas only): every rank codes its own frame (weak scaling).
    # tiled frames shard by contiguous runs of tiles: all ranks share one frame (strong scaling).
    tiled = plan.num_tiles > 1 and world > 1 and frames == 1
    if tiled:
        d_img = to_dev(img)                      # all ranks share the one frame
        my_tiles = shard.tile_range(plan.num_tiles, rank, world)
        assert my_tiles[1] > 0, "more ranks than tiles"
    else:
        my_tiles = (0, plan.num_tiles)
    my_share = my_tiles[1] / plan.num_tiles
    # The encode job and the decode job of a step are independent (a transcoder decodes frame n-1's
    # codestream while frame n is being encoded), so they are issued on two HIP streams and the GPU
    # overlaps the latency-bound block-decoder chains with the bandwidth-bound transforms.
    s_enc = torch.cuda.Stream(dev) if args.streams == 2 else torch.cuda.current_stream(dev)
    s_dec = torch.cuda.Stream(dev) if args.streams == 2 else torch.cuda.current_stream(dev)
    torch.cuda.synchronize(dev)
    with torch.cuda.stream(s_enc):
        enc = codec.Encoder(plan=plan, device=local_rank, tiles=my_tiles if frames == 1 else None, frames=frames)
    t0 = time.perf_counter()
    if tiled:
        enc.run_device(d_img)
        cdev = dev if backend == "nccl" else None                # device tensors over RCCL, host tensors over gloo
        # RCCL: the tile-parts are assembled in HBM and travel GPU -> GPU; gloo (CPU smoke runs): through the host
        part, lens = enc.finish_tiles_device() if backend == "nccl" else enc.finish_tiles()
        all_lens = shard.gather_tile_lengths(lens, plan.num_tiles, my_tiles[0], device=cdev, parts_per_tile=plan.parts_per_tile)
        parts, _ = shard.gather_bytes(part, device=cdev)         # RCCL: the final codestream gather
        cs = shard.assemble(plan.t2_main_header(all_lens), parts) if rank == 0 else None
        box = [cs]
        dist.broadcast_object_list(box, src=0)                   # every rank decodes from the same stream
        cs = box[0]
    else:
        cs = enc.encode(d_img)                   # also serves as the first warm-up + produces the decoder's input
    t_e2e_enc = time.perf_counter() - t0
    torch.cuda.synchronize(dev)
    with torch.cuda.stream(s_dec):
        dec = codec.Decoder(cs, device=local_rank, tiles=my_tiles if frames == 1 else None)
    d_out = torch.zeros_like(d_img) if tiled else torch.empty_like(d_img)
    d_out.zero_()                                # (also loads torch's fill kernel NOW: its first use stalls the host for tens of
    torch.cuda.synchronize(dev)                  #  milliseconds, and a chip left idle that long starts the next steps at low clocks)
    t0 = time.perf_counter()
    dec.run_device(d_out)
    torch.cuda.synchronize(dev)
    failed = dec.failed_blocks()
    assert failed == 0 or os.environ.get("OJPH_BENCH_NOCHECK"), "decode failed for %d code-blocks" % failed
    mask = None
    if tiled:                                    # compare only this rank's tile rows / columns
        mask = torch.zeros_like(d_img, dtype=torch.bool)
        for t in range(my_tiles[0], my_tiles[0] + my_tiles[1]):
            _, _, (x0, y0, tw, th) = plan.comp_plane(t, 0)
            mask[:, y0:y0 + th, x0:x0 + tw] = True

    def roundtrip_err():
        if mask is not None:
            return int(((d_out.int() - d_img.int()).abs() * mask).max().item())
        return int((d_out.int() - d_img.int()).abs().max().item())
    err = roundtrip_err()
    if rev:
        assert err == 0, "reversible round trip is not lossless"
    coded_bytes = enc.coded_bytes()
    nsamples_rank = nsamples * my_share          # samples this rank codes per step
    c_rate = coded_bytes / nsamples_rank
    levels = int(params.num_decomps)

    def step():
        enc.run_device(d_img)
        dec.run_device(d_out)

    enc.set_timing(False); dec.set_timing(False)     # the timed region carries no per-launch event pairs
    # Out of idle first: after 50 ms or more without work (the set-up above: parsing, uploads, checks on the host) the chip's
    # first eight to ten steps run 5-10 % slower while its clocks come up (tools/step_times.py, profiles/r06_c_step_times.txt)
    # -- more than the W warm-up steps a short run asks for (the driver's --steps 20 --warmup 5 measured 0.993 ms per step where
    # 200 or 1500 steps measure 0.968).  These untimed steps are the same steps; `settle_steps` in the line says how many.
    settle = int(os.environ.get("OJPH_BENCH_SETTLE_STEPS", "50"))
    for _ in range(settle):
        step()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    d_out.zero_()                                # the timed steps must produce the frame again (checked after the loop)
    _, epoch0 = dec.giveup_epoch()               # (synchronises)
    torch.cuda.synchronize(dev)
    step_events = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)] if os.environ.get("OJPH_BENCH_STEP_EVENTS") else None
    t0 = time.perf_counter()
    if step_events:                              # (diagnosis only: an event after every step -- how the timed region's steps differ)
        step_events[0].record()
    for i in range(args.steps):
        step()
        if step_events:
            step_events[i + 1].record()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if step_events:
        sys.stderr.write("timed region %.3f ms, enqueued after %.3f ms; per step (ms): %s\n" % (elapsed * 1e3, t_enq * 1e3, " ".join(
            "%.3f" % step_events[i].elapsed_time(step_events[i + 1]) for i in range(min(args.steps, 40)))))
    # What the timed steps produced, checked AFTER them: the last step's decode is collected (block verdicts; a one-launch
    # block decoder whose wait ran out would be repeated here and counted), no timed run gave up un-noticed (the give-up epoch
    # of the decoder object, bracketing the loop), the samples of the LAST timed step against the input, and the coded size.
    giveup1, epoch1 = dec.giveup_epoch()
    failed_after = dec.failed_blocks()
    retries_after = dec.fused_retries()
    err_after = roundtrip_err()
    assert failed_after == 0 or os.environ.get("OJPH_BENCH_NOCHECK"), "timed region: decode failed for %d code-blocks" % failed_after
    assert err_after == err or os.environ.get("OJPH_BENCH_NOCHECK"), "timed region: round-trip error %d, was %d before it" % (err_after, err)
    assert enc.coded_bytes() == coded_bytes, "timed region: the encoder's coded size changed"
    # ... and the codestream (tiled: this rank's tile-parts) the LAST timed encode produced: the bytes of the first one,
    # which the decoder above decodes and which the parity tests compare with the reference's
    import hashlib
    if tiled:
        last_part, _ = enc.finish_tiles()
        first_part = part.cpu().numpy().tobytes() if hasattr(part, "cpu") else bytes(part)
        cs_same = bytes(last_part) == first_part
        cs_digest = hashlib.sha256(bytes(last_part)).hexdigest()
    else:
        last_cs = enc.finish() if frames == 1 else [enc.finish(f) for f in range(frames)]
        cs_same = last_cs == cs
        cs_digest = hashlib.sha256(last_cs if frames == 1 else b"".join(last_cs)).hexdigest()
    assert cs_same, "timed region: the last timed encode's codestream differs from the first encode's"
    # rank 0's frame of C2 / C3 is the survey's known-answer frame: the reference's own codestream and decoded samples
    # (digests in tests/golden/survey_ka.json, made by tests/golden/make_survey_ka.py from the reference built here)
    ref_cs = ref_dec = None
    if rank == 0 and frames == 1 and not tiled and args.workload[:2] in ("c2", "c3"):
        gold = json.load(open(os.path.join(ROOT, "tests", "golden", "survey_ka.json")))[args.workload[:2]]
        gold = gold.get("generic", gold)
        ref_cs = bool(len(last_cs) == gold["bytes"] and cs_digest == gold["sha256"])
        if "decoded_sha256" in gold:
            ref_dec = bool(hashlib.sha256(np.ascontiguousarray(d_out.cpu().numpy().astype(np.int32)).tobytes()).hexdigest() == gold["decoded_sha256"])
        assert ref_cs and ref_dec is not False, "timed region: codestream / decoded samples are not the reference's (tests/golden/survey_ka.json)"
    verified = {"verified_after_timing": True, "failed_blocks": int(failed_after), "fused_retries": int(retries_after),
                "fused_runs_timed": int(epoch1 - epoch0), "fused_giveups_in_timed_region": bool(giveup1 > epoch0),
                "roundtrip_max_abs_err_last_step": int(err_after), "coded_bytes_last_step": int(coded_bytes),
                "codestream_last_step_equals_first": bool(cs_same), "codestream_last_step_sha256": cs_digest,
                "codestream_last_step_equals_reference_digest": ref_cs, "decoded_last_step_equals_reference_digest": ref_dec}
    per_rank_ms = [round(elapsed * 1e3 / args.steps, 4)]
    if world > 1:                                # every rank's own time travels: a straggler shows; value uses the maximum
        t = torch.zeros(world, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        t[rank] = elapsed
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        per_rank_ms = [round(float(x) * 1e3 / args.steps, 4) for x in t.tolist()]
        elapsed = float(t.max().item())
    ms_per_step = elapsed * 1e3 / args.steps
    if args.calibrate:                           # known traffic: reads 4 B/elem, writes 4 B/elem, 16 B per lane
        a = torch.zeros(64 * 1024 * 1024, dtype=torch.float32, device=dev)
        b = torch.empty_like(a)
        for _ in range(3):
            torch.add(a, 1.0, out=b)
        torch.cuda.synchronize(dev)
        del a, b

    # per-kernel timings: HIP events on the codec's own stream, averaged over a second run of the
    # same steps (reading the events synchronises, so it is kept out of the timed region above)
    reps = max(1, min(args.steps, 10))
    enc.set_timing(True); dec.set_timing(True)
    te, td = None, None
    for _ in range(reps):
        step()
        a, b = enc.timing(), dec.timing()
        if te is None:
            te, td = a, b
        else:
            for acc, cur in ((te, a), (td, b)):
                for k, v in cur.items():
                    acc[k] = [x + y for x, y in zip(acc[k], v)] if isinstance(v, list) else acc[k] + v
    for acc in (te, td):
        for k, v in acc.items():
            acc[k] = [x / reps for x in v] if isinstance(v, list) else v / reps
    ns = nsamples_rank
    # bytes of a working sample: 4 (int32 / float planes); 8 on the reference's 64-bit sample path (components of more than
    # 32 bits of precision: int64 planes) -- the algorithmic bytes of 8(d) with that element size
    elem = 8.0 if any(plan.comp_style(c)["wide"] for c in range(nc)) else 4.0
    kernels = {
        "dwt_forward(all levels)": (dwt_alg_bytes(ns, levels, args.container, elem), te["dwt_ms"]),
        "dwt_forward(level 1)": ((elem + args.container / 8.0) * ns, te["dwt_levels_ms"][0] if te["dwt_levels_ms"] else 0.0),
        "dwt_inverse(all levels)": (dwt_alg_bytes(ns, levels, args.container, elem), td["dwt_ms"]),
        "dwt_inverse(level 1)": ((elem + args.container / 8.0) * ns, td["dwt_levels_ms"][-1] if td["dwt_levels_ms"] else 0.0),
        "ht_encode": ((elem + c_rate) * ns, te["ht_ms"]),           # all launches of the block encoder (sum)
        # block decoder, SURVEY.md section 8(d): it reads the c N coded bytes and writes the 4 N coefficient bytes -- (c + 4) N
        # for the whole decoder.  Split by launch: step 1 reads the MEL / VLC share of the coded bytes (about a fifth), step 2
        # the MagSgn share and writes the coefficients.  The per-quad records step 1 hands to step 2 (4 bytes per quad = 1 byte
        # per sample, written once and read once) and the flat strings of the optional prep launch are INTERMEDIATE bytes:
        # real traffic, not algorithmic -- reported next to the algorithmic figure, never inside it.
        "ht_dec_prep": (0.0, td["ht_prep_ms"]),
        "ht_dec_step1": (0.2 * c_rate * ns, td["ht_step1_ms"]),
        "ht_dec_step2": ((0.8 * c_rate + elem) * ns, td["ht_step2_ms"]),
    }
    intermediate = {"ht_dec_prep": 0.8 * c_rate * ns, "ht_dec_step1": 1.0 * ns, "ht_dec_step2": 1.0 * ns}
    if td["ht_step1_ms"] == 0.0 and td["ht_prep_ms"] < 0.02 and td["ht_step2_ms"] > 0:
        # step 1 and step 2 ran as ONE launch (chains first, step-2 workers behind them slice by slice): one entry with the
        # decoder's (c + 4) N; the records pass through memory once in each direction inside the launch (2 N intermediate)
        b1, _ = kernels.pop("ht_dec_step1"); b2, _ = kernels.pop("ht_dec_step2"); kernels.pop("ht_dec_prep")
        if elem == 8.0:
            # 64-bit samples: ht_decode64_launch's three launches (flat strings, step 1 on them, 64-bit step 2) under one span;
            # the flat strings and the records are written once and read once
            kernels["ht_dec64(prep + step 1 + step 2, three launches)"] = (b1 + b2, td["ht_step2_ms"])
            intermediate = {"ht_dec64(prep + step 1 + step 2, three launches)": 2.0 * ns + 2.0 * c_rate * ns}
        else:
            kernels["ht_dec_fused(step 1 + step 2)"] = (b1 + b2, td["ht_step2_ms"])
            intermediate = {"ht_dec_fused(step 1 + step 2)": 2.0 * ns}
    elif td["ht_prep_ms"] < 0.02:
        # separate launches without the prep launch (step 1's partner wavefronts read the MEL / VLC bytes as they are):
        # the prep span is empty
        kernels.pop("ht_dec_prep"); intermediate.pop("ht_dec_prep")
    if len(te["ht_launches_ms"]) == 2:
        # the encoder codes the top resolution's blocks on a side stream, concurrently with the lower
        # DWT levels and followed by the rest: two launches of the same kernel per frame, listed one
        # by one with their own algorithmic bytes (rocprofv3's per-kernel average is their mean)
        from openjph_amd.plan import parse_codestream
        first = cs if not isinstance(cs, (list, tuple)) else cs[0]
        pl = parse_codestream(first)
        cb = pl.coded_blocks()
        top = np.array([int(pl.bands[int(b["band"])]["res"]) == levels for b in pl.blocks])
        # (the side stream's launch takes the first enc.top_blocks() of the top resolution's blocks in plan order -- the encoder
        # cuts the two branches so that the side one ends first, ojphgpu_codec.cpp -- the rest of them join the other launch)
        side = np.zeros_like(top)
        side[np.flatnonzero(top)[:enc.top_blocks()]] = True
        top = side
        area = np.array([int(b["w"]) * int(b["h"]) for b in pl.blocks], dtype=np.float64)
        coded = (cb["len1"].astype(np.float64) + cb["len2"])
        del kernels["ht_encode"]
        kernels["ht_encode[top resolution, side stream]"] = (elem * area[top].sum() + coded[top].sum(), te["ht_launches_ms"][0])
        kernels["ht_encode[lower resolutions]"] = (elem * area[~top].sum() + coded[~top].sum(), te["ht_launches_ms"][1])
    if te["convert_ms"] > 0 or td["convert_ms"] > 0:   # otherwise the conversion (and the colour transform) is fused into the top DWT level
        kernels["convert_forward"] = ((elem + args.container / 8.0) * ns, te["convert_ms"])
        kernels["convert_inverse"] = ((elem + args.container / 8.0) * ns, td["convert_ms"])
    kinfo = {}
    for k, (b, ms) in kernels.items():
        kinfo[k] = {"ms": round(ms, 4), "alg_GB": round(b / 1e9, 4), "GBps": round(b / 1e6 / ms, 1) if ms > 0 else None}
        if k in intermediate:
            kinfo[k]["intermediate_GB"] = round(intermediate[k] / 1e9, 4)
    # the dominant LAUNCH: the "(all levels)" entries are families of launches (roofline_dwt reports them)
    dom = max((k for k in kernels if "(all levels)" not in k), key=lambda k: kernels[k][1])
    dom_b, dom_ms = kernels[dom]
    achieved = dom_b / 1e6 / dom_ms if dom_ms > 0 else 0.0
    traffic = None
    pmc, pmc_state = committed_counters("pmc_traffic.json")       # HBM bytes per launch from the committed PMC pass ...
    if pmc_state == "current":                                     # ... of exactly these kernel sources, or nothing
        traffic = pmc.get(args.workload, {}).get(dom.split("[")[0]) if "[" not in dom else pmc.get(args.workload, {}).get(dom)

    # The same steps with the two independent jobs of a step -- encode this frame, decode that codestream
    # -- on two HIP streams (what a transcoder or a capture + playback node runs): reported next to
    # `value`, which stays the one-stream figure its per-kernel numbers belong to.
    # (HIP multiplexes streams onto 4 hardware queues: the one-stream codec objects and their side streams
    # are released first, or the two new streams would share queues and run one after the other)
    two_stream_ms = None
    if args.streams == 1 and not tiled and world == 1 and not args.plain:
        import gc
        del enc, dec
        gc.collect()
        torch.cuda.synchronize(dev)
        s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        with torch.cuda.stream(s1):
            enc2 = codec.Encoder(plan=plan, device=local_rank, frames=frames)
        with torch.cuda.stream(s2):
            dec2 = codec.Decoder(cs, device=local_rank)
        d_out2 = torch.empty_like(d_img)
        enc2.set_timing(False); dec2.set_timing(False)
        torch.cuda.synchronize(dev)

        def step2():
            with torch.cuda.stream(s1):
                enc2.run_device(d_img)
            with torch.cuda.stream(s2):
                dec2.run_device(d_out2)
        for _ in range(args.warmup):
            step2()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step2()
        torch.cuda.synchronize(dev)
        two_stream_ms = (time.perf_counter() - t0) * 1e3 / args.steps
        assert torch.equal(d_out2, d_out) or os.environ.get("OJPH_BENCH_NOCHECK"), "two-stream decode differs"
        del enc2, dec2, d_out2

    # The DWT launches with nothing beside them: codec objects made with OJPHGPU_NO_OVERLAP=1 issue every launch on
    # one stream, so the small lower levels are not stretched by the block coder that normally runs next to them
    # (their durations in `kernels` are; the schedule that makes the step fastest makes those spans longest)
    dwt_alone = None
    if world == 1 and not tiled and not args.plain:
        try:
            os.environ["OJPHGPU_NO_OVERLAP"] = "1"
            enc3 = codec.Encoder(plan=plan, device=local_rank, frames=frames)
            dec3 = codec.Decoder(cs, device=local_rank)
            d_out3 = torch.empty_like(d_img)
            fa = ia = 0.0
            for i in range(8):
                enc3.run_device(d_img); dec3.run_device(d_out3)
                if i >= 3:
                    fa += enc3.timing()["dwt_ms"] / 5; ia += dec3.timing()["dwt_ms"] / 5
            dwt_alone = (fa, ia)
            del enc3, dec3, d_out3
        except Exception:
            dwt_alone = None
        finally:
            os.environ.pop("OJPHGPU_NO_OVERLAP", None)

    # Frame pipelines: host memory -> codestream in host memory (and back) with PCIe and host Tier-2 inside the
    # timed region, steady state over --e2e-frames frames: what a capture / playback process gets.
    e2e = None
    if args.e2e_frames > 0 and world == 1 and not tiled:
        try:
            import gc
            gc.collect(); torch.cuda.synchronize(dev)
            # (a batch workload goes through the pipes frame by frame: a pipe's slots are its batch)
            e2e = e2e_pipelines(plan, img if frames == 1 else img[0], cs, args.e2e_frames, args.container, torch)
        except Exception as e:                   # reported, never fatal for the headline figure
            e2e = {"error": str(e)[:300]}

    # the strong-scaling question of north_star (one 16K frame, 256 tiles over the ranks) inside the same run
    strong = None
    strong_hung = False
    if not args.no_strong and args.workload.startswith("c3") and frames == 1:
        import gc
        try:
            del enc, dec
        except NameError:
            pass
        gc.collect(); torch.cuda.synchronize(dev)
        if world == 1:
            try:
                strong = strong_scaling_c4(args, rank, world, local_rank, dev, backend, torch, dist)
            except Exception as e:
                strong = {"error": str(e)[:300]}
        else:
            # N > 1: this part holds the one exchange the path has -- the RCCL gatherv of the tile-parts, point-to-point sends and
            # receives of device tensors -- which no run of this repository has ever made with two devices.  It must not be able to
            # take the headline line with it: it runs on a thread of its own with a deadline (OJPH_BENCH_STRONG_TIMEOUT_S, 300);
            # a rank whose part has not come back by then reports that instead, the line is printed, and the process leaves
            # without waiting for the process group (os._exit below: a collective that hangs cannot be cancelled).
            import threading
            box = {}

            def strong_part():
                try:
                    torch.cuda.set_device(local_rank)            # (the current device is a per-thread setting)
                    box["r"] = strong_scaling_c4(args, rank, world, local_rank, dev, backend, torch, dist)
                except Exception as e:                           # noqa: BLE001 -- reported in the line
                    box["r"] = {"error": "rank %d: %s" % (rank, str(e)[:300])}
            th = threading.Thread(target=strong_part, daemon=True)
            th.start()
            th.join(float(os.environ.get("OJPH_BENCH_STRONG_TIMEOUT_S", "300")))
            if th.is_alive():
                strong_hung = True
                strong = {"error": "rank %d: the tile-sharded part (RCCL gatherv of the tile-parts) did not come back in time; set NCCL_DEBUG=INFO, "
                                   "or OJPH_BENCH_BACKEND=gloo to take RCCL out of the picture" % rank}
                sys.stderr.write("bench.py rank %d/%d: strong_scaling_c4 did not come back; the line is printed without it\n" % (rank, world))
            else:
                strong = box.get("r")

    result = {
        "metric": "Msamples/s encode+decode, 8K 12-bit 4:4:4; achieved HBM GB/s vs roofline",
        "value": round(nsamples * (1 if tiled else world) / (ms_per_step * 1e-3) / 1e6, 2),
        "unit": "Msamples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_steps": settle,
        "ms_per_step": round(ms_per_step, 4),
        "per_rank_ms_per_step": per_rank_ms,
        "higher_is_better": True, "scaling": "strong" if tiled else "weak", "vs_baseline": None,
        "dtype": ("i64" if elem == 8 else "i32") if rev else "f32", "data": "synthetic",
        "value_covers": "device-resident step: samples in HBM -> convert + DWT + quantise + HT block encode -> coded block bytes in HBM, "
                        "then those bytes -> HT block decode + inverse DWT + convert -> samples in HBM; PCIe and host Tier-2 are "
                        "outside (e2e_steady_Msamples_s has them inside)",
        "e2e_steady_Msamples_s": ({k: v["Msamples_s"] for k, v in e2e.items() if isinstance(v, dict) and "Msamples_s" in v} if e2e and "error" not in e2e else None),
        "config": {"workload": args.workload, "width": w, "height": h, "components": nc, "bit_depth": bd,
                   "wavelet": ("5/3 reversible" if rev else "9/7 irreversible") + (" as an ATK marker segment (general lifting kernels)" if "params" in extra else "")
                              + (", 64-bit sample path" if elem == 8 else ""), "qstep": qstep if not rev else None,
                   "decomps": levels, "block": [int(params.block_w), int(params.block_h)],
                   "tile": list(tile), "frames_per_step": frames * (1 if tiled else world),
                   "sharding": ("%d tiles per GPU of one frame" % my_tiles[1]) if tiled else "one frame per GPU (replicas)",
                   "hip_streams": args.streams, "sample_container_bits": args.container,
                   "coded_bytes_per_sample": round(c_rate, 4),
                   "encode_ms": round(te["total_ms"], 4), "decode_ms": round(td["total_ms"], 4),
                   "encode_Msamples_s": round(ns / te["total_ms"] / 1e3, 1),
                   "decode_Msamples_s": round(ns / td["total_ms"] / 1e3, 1),
                   "e2e_first_encode_s_incl_pcie_tier2": round(t_e2e_enc, 3),
                   "roundtrip_max_abs_err": int(err),
                   "two_streams_ms_per_step": round(two_stream_ms, 4) if two_stream_ms else None,
                   "two_streams_Msamples_s": round(nsamples / two_stream_ms / 1e3, 2) if two_stream_ms else None},
        # achieved = SURVEY.md section 8(d)'s algorithmic bytes of the launch / its measured duration; what the launch moves
        # beyond them (records handed from step 1 to step 2, halo re-reads) shows in `traffic` and `traffic_ratio`
        "roofline": {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "algorithmic_bytes": int(dom_b), "intermediate_bytes": int(intermediate.get(dom, 0)),
                     "traffic_ratio": round(traffic / dom_b, 3) if traffic and dom_b else None,
                     "traffic_source": "profiles/pmc_traffic.json (%s)" % pmc_state},
        "kernels": kinfo,
    }
    result.update(verified)
    if e2e:
        result["e2e"] = e2e
    if strong:
        result["strong_scaling_c4"] = strong
    result["dist"] = {"backend": ("rccl" if backend == "nccl" else backend) if world > 1 else None,
                      "world_size": dist.get_world_size() if world > 1 else 1,
                      "devices": torch.cuda.device_count(), "one_gpu_smoke": bool(os.environ.get("OJPH_BENCH_ONE_GPU"))}
    if dist_info:
        result["dist"].update(dist_info)
    rv = roofline_valu(args.workload, kinfo)
    if rv:
        result["roofline_valu"] = rv
        # which roof the dominant launch is nearer to: the HBM fraction above, or the VALU-issue fraction of the same launch
        kd = (rv.get("kernels") or {}).get(dom, {})
        result["roofline"].update(limiter_fields(result["roofline"]["frac"], kd))
    if "dwt_forward(level 1)" in kernels and kernels["dwt_forward(level 1)"][1] > 0:
        # the HBM-bound kernel family of the path (north_star sets its roofline target on it); the
        # block coder launches above are bound by integer VALU issue, not by HBM.  Level 1 -- the two
        # launches that move 69 % of the DWT's bytes -- runs alone on the GPU; the small launches of
        # the lower levels share it with the block coder of the other stream (encoder: side stream,
        # decoder: step 2 of the top resolution), which stretches their durations: both figures are given.
        (bf, mf), (bi_, mi) = kernels["dwt_forward(level 1)"], kernels["dwt_inverse(level 1)"]
        (af, amf), (ai, ami) = kernels["dwt_forward(all levels)"], kernels["dwt_inverse(all levels)"]
        ach = (bf + bi_) / 1e6 / (mf + mi)
        ach_all = (af + ai) / 1e6 / (amf + ami)
        tr = None
        if pmc_state == "current":
            w_ = pmc.get(args.workload, {})
            tr = w_.get("dwt_forward(level 1)", 0) + w_.get("dwt_inverse(level 1)", 0) or None
        result["roofline_dwt"] = {"kernel": "dwt_forward + dwt_inverse (level 1, 2 launches)", "bound": "hbm",
                                  "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": tr,
                                  "all_levels": {"launches": 2 * levels, "achieved": round(ach_all, 1),
                                                 "frac": round(ach_all / HBM_PEAK_GBS, 4)}}
        if dwt_alone and dwt_alone[0] > 0 and dwt_alone[1] > 0:      # the same launches with nothing running beside them
            a_alone = (af + ai) / 1e6 / (dwt_alone[0] + dwt_alone[1])
            result["roofline_dwt"]["all_levels_alone"] = {
                "forward_ms": round(dwt_alone[0], 4), "inverse_ms": round(dwt_alone[1], 4), "achieved": round(a_alone, 1),
                "frac": round(a_alone / HBM_PEAK_GBS, 4),
                "forward_frac": round(af / 1e6 / dwt_alone[0] / HBM_PEAK_GBS, 4), "inverse_frac": round(ai / 1e6 / dwt_alone[1] / HBM_PEAK_GBS, 4)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(img if frames == 1 else img[0], bd, rev, ct, qstep, tile, args.cpu_reps)
        result["cpu_baseline"]["all_cores"] = cpu_all
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        if strong_hung:                              # a hung collective cannot be torn down: leave, the line is out
            sys.stderr.flush()
            os._exit(0)
        dist.destroy_process_group()


def start_process_group(backend, rank, world, local_rank, dev, torch, dist):
    """N > 1: the process group, and a self-check before anything is timed -- so that the first run on an 8-GPU node either
    works or says at once what is wrong instead of hanging in a collective.  (The RCCL path has never met a second device in
    this repository's own runs: RCCL refuses two ranks on one GPU, and the development boxes have one.)
      * a watchdog around the group's first collective: no answer within OJPH_BENCH_PG_TIMEOUT_S (default 120) seconds -> the
        rank prints who it is, what it was waiting for and the environment that matters, and exits with status 3;
      * every rank's (host, device index, device UUID / PCI bus id) is gathered: the N ranks must sit on N distinct devices
        (OJPH_BENCH_ONE_GPU smoke runs excepted) -- a launcher that gave two ranks the same LOCAL_RANK fails here, loudly;
      * the RCCL version, the rendezvous address and the devices go into the line's `dist` object."""
    import datetime
    import socket
    import threading
    limit = float(os.environ.get("OJPH_BENCH_PG_TIMEOUT_S", "120"))

    def watchdog(what):
        def fire():
            sys.stderr.write("bench.py rank %d/%d (local %d, %s): no answer from %s within %.0f s -- backend %s, MASTER_ADDR=%s MASTER_PORT=%s "
                             "HSA_ENABLE_IPC_MODE_LEGACY=%s NCCL_DEBUG=%s, %d device(s) visible.  Set NCCL_DEBUG=INFO and run again; every rank must be "
                             "started (torch.distributed.run --nproc-per-node N) and reach this point.\n"
                             % (rank, world, local_rank, socket.gethostname(), what, limit, backend, os.environ.get("MASTER_ADDR"),
                                os.environ.get("MASTER_PORT"), os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), os.environ.get("NCCL_DEBUG"),
                                torch.cuda.device_count()))
            sys.stderr.flush()
            os._exit(3)
        t = threading.Timer(limit, fire)
        t.daemon = True
        t.start()
        return t
    w = watchdog("init_process_group (rendezvous)")
    dist.init_process_group(backend, timeout=datetime.timedelta(seconds=max(limit * 4, 600)))
    w.cancel()
    if dev is not None:
        props = torch.cuda.get_device_properties(dev)
        ident = str(getattr(props, "uuid", "") or "") or "%s/%s" % (getattr(props, "pci_bus_id", "?"), getattr(props, "pci_device_id", "?"))
        mine = (socket.gethostname(), int(local_rank), ident, props.name)
    else:                                                # (the CPU tests of this function: no device to name)
        mine = (socket.gethostname(), int(local_rank), "no-device-%d" % local_rank, "cpu")
    w = watchdog("the first collective (all_gather of the ranks' devices)")
    everyone = [None] * world
    dist.all_gather_object(everyone, mine)
    t = torch.ones(1, device=dev if backend == "nccl" and dev is not None else "cpu")
    dist.all_reduce(t)                                   # (a data-path collective as well: device tensors over RCCL)
    if backend == "nccl" and dev is not None:
        torch.cuda.synchronize(dev)
    w.cancel()
    assert int(t.item()) == world, "all_reduce over %d ranks returned %s" % (world, t.item())
    distinct = len(set((h, u) for h, _, u, _ in everyone))
    if not os.environ.get("OJPH_BENCH_ONE_GPU") and distinct != world:
        sys.stderr.write("bench.py: %d ranks sit on %d distinct device(s): %s -- one process per GPU is the contract (LOCAL_RANK = device index)\n"
                         % (world, distinct, everyone))
        sys.stderr.flush()
        dist.destroy_process_group()
        sys.exit(4)
    try:
        ver = ".".join(str(x) for x in torch.cuda.nccl.version())
    except Exception:
        ver = None
    return {"rccl_version": ver if backend == "nccl" else None, "distinct_devices": distinct,
            "rank_devices": ["%s:%d %s" % (h, i, n) for h, i, _, n in everyone], "self_check": "rendezvous, all_gather_object, all_reduce: ok",
            "master": "%s:%s" % (os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT"))}


def strong_scaling_c4(args, rank, world, local_rank, dev, backend, torch, dist):
    """north_star's second scaling question, answered inside the same run as the replica figure: ONE 16384 x 16384 16-bit
    frame in 256 tiles of 1024 x 1024 (BASELINE config 4), reversible 5/3, its tiles sharded over the ranks in contiguous
    runs (strong scaling: the frame is fixed, a rank codes 256 / N tiles).  The timed step is encode + decode of the
    rank's tiles, device resident, like `value`; the final codestream gather -- tile-parts assembled in HBM, an all-reduce
    of the Psot lengths, a gatherv of the tile-part bytes to rank 0 over RCCL (point-to-point sends / receives of the exact
    sizes) -- is timed separately.  Rank 0 checks the assembled codestream against the digest of the reference's own
    (tests/golden/survey_ka.json, made by tests/golden/make_survey_ka.py); every rank checks its tiles decode losslessly.
    Tile independence: ojph_codestream_local.cpp:113-180, ojph_tile.cpp:584-610."""
    import hashlib
    from openjph_amd import codec, shard
    from openjph_amd.plan import Plan, make_params
    from tests import synth
    name = "c4_16k_gray_16b_rev53_tiled"
    w, h, nc, bd, rev, ct, qstep, tile = WORKLOADS[name]
    size = int(os.environ.get("OJPH_BENCH_STRONG_SIZE", "0")) or w            # (smaller frame: CPU-side smoke runs of this code path)
    img = synth.survey_c4(size=size)
    w = h = size
    d_img = torch.from_numpy(img.astype(np.int16) if args.container == 16 else img).to(dev)
    plan = Plan(make_params(w, h, nc, bit_depth=bd, reversible=rev, color_transform=ct, qstep=qstep, tile=tile))
    first, count = shard.tile_range(plan.num_tiles, rank, world)
    assert count > 0, "more ranks than tiles"
    enc = codec.Encoder(plan=plan, device=local_rank, tiles=(first, count))
    enc.run_device(d_img)
    torch.cuda.synchronize(dev)
    cdev = dev if backend == "nccl" else None                 # device tensors over RCCL, host tensors over gloo

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    best = None
    for _ in range(3):
        sync()
        t0 = time.perf_counter()
        part, lens = enc.finish_tiles_device() if backend == "nccl" else enc.finish_tiles()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        if world > 1:
            all_lens = shard.gather_tile_lengths(lens, plan.num_tiles, first, device=cdev, parts_per_tile=plan.parts_per_tile)
            bufs, sizes = shard.gather_bytes(part, device=cdev, as_tensors=True)
        else:
            all_lens, bufs, sizes = np.asarray(lens, np.uint32), [part if hasattr(part, "cpu") else torch.frombuffer(bytearray(part), dtype=torch.uint8)], [len(part)]
        sync()
        t2 = time.perf_counter()
        host = shard.to_host(bufs) if rank == 0 else None       # device -> ONE pinned host buffer (kept from call to call)
        t3 = time.perf_counter()
        parts = [v.numpy().tobytes() for v in host] if rank == 0 else None
        cur = (t2 - t1, t1 - t0, t3 - t2)
        best = cur if best is None or cur[0] < best[0] else best
    cs = shard.assemble(plan.t2_main_header(all_lens), parts) if rank == 0 else None
    digest_ok = None
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "survey_ka.json")))["c4"]
    if rank == 0 and size == WORKLOADS[name][0]:
        digest_ok = bool(len(cs) == gold["bytes"] and hashlib.sha256(cs).hexdigest() == gold["sha256"])
    if world > 1:                                             # every rank decodes its tiles from the same codestream
        n = torch.tensor([len(cs) if rank == 0 else 0], dtype=torch.int64, device=cdev or "cpu")
        dist.broadcast(n, src=0)
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=cdev or "cpu")
        if rank == 0:
            buf.copy_(torch.frombuffer(bytearray(cs), dtype=torch.uint8))
        dist.broadcast(buf, src=0)
        cs = cs if rank == 0 else buf.cpu().numpy().tobytes()
        del buf
    dec = codec.Decoder(cs, device=local_rank, tiles=(first, count))
    d_out = torch.zeros_like(d_img)
    dec.run_device(d_out)
    torch.cuda.synchronize(dev)
    assert dec.failed_blocks() == 0, "strong-scaling decode failed"
    lossless = True
    for t in range(first, first + count):
        _, _, (x0, y0, tw, th) = plan.comp_plane(t, 0)
        lossless = lossless and bool(torch.equal(d_out[:, y0:y0 + th, x0:x0 + tw], d_img[:, y0:y0 + th, x0:x0 + tw]))
    assert lossless, "rank %d: its tiles of the reversible frame did not come back exactly" % rank
    enc.set_timing(False); dec.set_timing(False)
    steps = max(1, min(args.strong_steps, args.steps))
    for _ in range(min(3, args.warmup)):
        enc.run_device(d_img); dec.run_device(d_out)
    sync(); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        enc.run_device(d_img); dec.run_device(d_out)
    torch.cuda.synchronize(dev)
    mine = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
    per_rank = [mine]
    if world > 1:
        t = torch.zeros(world, dtype=torch.float64, device=cdev or "cpu")
        t[rank] = mine
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        per_rank = [float(x) for x in t.tolist()]
    ms = max(per_rank) * 1e3 / steps
    moved = int(sum(sizes)) - int(sizes[0])
    # END TO END: the frame in (pinned) host memory -> the .j2c bytes in (pinned) host memory on rank 0, everything inside the
    # timed region: every rank uploads the rows its tiles lie in, codes them, lays its tile-parts out in HBM; the Psot lengths
    # are all-reduced, the tile-parts travel to rank 0 (RCCL gatherv), rank 0 copies them behind the main header into ONE pinned
    # buffer and appends EOC.  Best of 5; max over ranks (a barrier closes the region).
    h_img = torch.from_numpy(img.astype(np.int16) if args.container == 16 else img).pin_memory()
    ntx = (w + tile[0] - 1) // tile[0]
    r0, r1 = (first // ntx) * tile[1], min(h, ((first + count - 1) // ntx + 1) * tile[1])      # image rows of this rank's run of tiles
    host_cs = torch.empty(int(len(cs)) + (1 << 20), dtype=torch.uint8).pin_memory() if rank == 0 else None
    e2e_best, e2e_ok = None, None
    for _ in range(5):
        sync()
        t0 = time.perf_counter()
        d_img[:, r0:r1].copy_(h_img[:, r0:r1], non_blocking=True)
        enc.run_device(d_img)
        part, lens = enc.finish_tiles_device() if backend == "nccl" else enc.finish_tiles()
        if world > 1:
            all_lens = shard.gather_tile_lengths(lens, plan.num_tiles, first, device=cdev, parts_per_tile=plan.parts_per_tile)
            bufs, _ = shard.gather_bytes(part, device=cdev, as_tensors=True)
        else:
            all_lens, bufs = np.asarray(lens, np.uint32), [part if hasattr(part, "is_cuda") else torch.frombuffer(bytearray(part), dtype=torch.uint8)]
        n_out = 0
        if rank == 0:
            hdr = plan.t2_main_header(all_lens)
            host_cs[:len(hdr)] = torch.frombuffer(bytearray(hdr), dtype=torch.uint8)
            at = len(hdr)
            for b in bufs:
                host_cs[at:at + b.numel()].copy_(b, non_blocking=True); at += int(b.numel())
            torch.cuda.synchronize(dev)
            host_cs[at] = 0xFF; host_cs[at + 1] = 0xD9
            n_out = at + 2
        sync()
        dt = time.perf_counter() - t0
        if rank == 0 and e2e_ok is None and size == WORKLOADS[name][0]:
            e2e_ok = bool(n_out == gold["bytes"] and hashlib.sha256(host_cs[:n_out].numpy().tobytes()).hexdigest() == gold["sha256"])
        e2e_best = dt if e2e_best is None or dt < e2e_best else e2e_best
    if world > 1:
        t = torch.tensor([e2e_best], dtype=torch.float64, device=cdev or "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        e2e_best = float(t.item())
    # ... and with the node's shared-memory gather instead of the gatherv (shard.HostGather): every rank copies its tile-parts over
    # its OWN link straight to their place in one host segment; no rank receives anything, nothing crosses xGMI
    host_gather = None
    try:
        hg = shard.HostGather(int(len(cs)) + (1 << 20) if rank == 0 else 0, register=True) if world == 1 else None
        if world > 1:
            cap = torch.tensor([int(len(cs)) + (1 << 20) if rank == 0 else 0], dtype=torch.int64, device=cdev or "cpu")
            dist.broadcast(cap, src=0)
            hg = shard.HostGather(int(cap.item()), register=True)
        ppt = plan.parts_per_tile
        hg_best, hg_ok = None, None
        for _ in range(5):
            sync()
            t0 = time.perf_counter()
            d_img[:, r0:r1].copy_(h_img[:, r0:r1], non_blocking=True)
            enc.run_device(d_img)
            part, lens = enc.finish_tiles_device() if (backend == "nccl" or world == 1) else enc.finish_tiles()
            all_lens = (shard.gather_tile_lengths(lens, plan.num_tiles, first, device=cdev, parts_per_tile=ppt) if world > 1
                        else np.asarray(lens, np.uint32))
            sizes = [int(np.asarray(all_lens[shard.tile_range(plan.num_tiles, r, world)[0] * ppt:
                                             sum(shard.tile_range(plan.num_tiles, r, world)) * ppt], dtype=np.uint64).sum()) for r in range(world)]
            n_out = hg.place(part, sizes, plan.t2_main_header(all_lens) if rank == 0 else None)
            sync()
            dt = time.perf_counter() - t0
            if rank == 0 and hg_ok is None and size == WORKLOADS[name][0]:
                hg_ok = bool(n_out == gold["bytes"] and hashlib.sha256(bytes(hg.view[:n_out])).hexdigest() == gold["sha256"])
            hg_best = dt if hg_best is None or dt < hg_best else hg_best
        if world > 1:
            t = torch.tensor([hg_best], dtype=torch.float64, device=cdev or "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            hg_best = float(t.item())
        host_gather = {"ms": round(hg_best * 1e3, 3), "Msamples_s": round(w * h * nc / hg_best / 1e6, 1), "codestream_equals_reference_digest": hg_ok,
                       "registered_with_hip": bool(hg._registered),
                       "covers": "the same job with the tile-parts placed by every rank itself in ONE shared host segment (POSIX shared memory, "
                                 "hipHostRegister): N links at once, no receiving GPU"}
        hg.close()
    except Exception as e:                                    # (the figure above stands; this one says why it is missing)
        host_gather = {"error": str(e)[:300]}
    # the same end-to-end job through the ONE-process form of the sharding (C ABI section 8: a host thread + encoder per device,
    # tile-parts copied from every GPU straight to their place in one pinned buffer) -- measured on rank 0's GPU at N = 1
    inproc = None
    if world == 1:
        try:
            del enc
            me = codec.MultiEncoder(plan=plan, devices=[local_rank])
            h32 = torch.from_numpy(img.astype(np.int16) if args.container == 16 else img.astype(np.int32)).pin_memory()
            bt = None
            for _ in range(4):
                t0 = time.perf_counter()
                out = me.encode(h32, copy=False)
                dt = time.perf_counter() - t0
                bt = dt if bt is None or dt < bt else bt
            inproc = {"ms": round(bt * 1e3, 3), "Msamples_s": round(w * h * nc / bt / 1e6, 1), "devices": 1,
                      "bytes_equal_reference": bool(size != WORKLOADS[name][0] or hashlib.sha256(out.tobytes()).hexdigest() == gold["sha256"]),
                      "covers": "ojphgpu_multi_encode_container: the frame in %d-bit containers in pinned host memory -> codestream in pinned host "
                                "memory (uploads, kernels, Tier-2 layout, placement in HBM, download)" % (16 if args.container == 16 else 32)}
            del me, h32
            enc = None
        except Exception as e:
            inproc = {"error": str(e)[:200]}
    del enc, dec, d_out, d_img
    return {"workload": name, "scaling": "strong", "n_gpus": world, "width": w, "height": h, "tiles": int(plan.num_tiles),
            "tiles_per_rank": [shard.tile_range(plan.num_tiles, r, world)[1] for r in range(world)],
            "steps": steps, "ms_per_step": round(ms, 4), "per_rank_ms_per_step": [round(x * 1e3 / steps, 4) for x in per_rank],
            "value": round(w * h * nc / ms / 1e3, 2), "unit": "Msamples/s",
            "value_covers": "encode + decode of every rank's tiles, device resident (max over ranks); the gather below is outside "
                            "(e2e_encode has it inside)",
            "e2e_encode": {"ms": round(e2e_best * 1e3, 3), "Msamples_s": round(w * h * nc / e2e_best / 1e6, 1),
                           "codestream_equals_reference_digest": e2e_ok,
                           "covers": "frame in pinned host memory -> .j2c bytes in pinned host memory on rank 0: every rank's upload of its "
                                     "tiles' rows, kernels, tile-parts laid out in HBM, all-reduce of the Psot lengths, gatherv of the "
                                     "tile-parts to rank 0, rank 0's copies behind the main header; max over ranks, best of 5",
                           "shared_host_segment": host_gather,
                           "one_process_multi_device": inproc},
            "codestream_bytes": len(cs), "codestream_equals_reference_digest": digest_ok, "tiles_lossless_on_every_rank": True,
            "gather": {"backend": "rccl" if backend == "nccl" else backend,
                       "tile_parts_assembled_ms": round(best[1] * 1e3, 3),
                       "lengths_allreduce_and_gatherv_ms": round(best[0] * 1e3, 3),
                       "rank0_device_to_host_ms": round(best[2] * 1e3, 3),
                       "bytes_received_by_rank0": moved,
                       "gatherv_GBps": round(moved / best[0] / 1e9, 2) if moved and best[0] > 0 else None}}


def limiter_fields(frac_hbm, kd):
    """`roofline.bound` is the contract's field and stays the roof `achieved / peak / frac` are quoted against ("hbm": SURVEY
    8(d)'s algorithmic bytes over the HBM peak).  What actually LIMITS the launch is a separate field, `limiter`, from the
    committed SQ counters of exactly this build (kd = roofline_valu's entry of the launch; empty when they are stale):
    "latency" when its wavefronts are parked at s_waitcnt for more than half of their cycles (wait_share), otherwise the roof
    it sits nearer to -- "valu-issue" (frac_valu_issue > frac) or "hbm".  -> the fields to merge into `roofline`."""
    fv = (kd or {}).get("frac")
    if fv is None:
        return {"limiter": None}
    ws = kd.get("wait_share")
    lim = "latency" if (ws or 0) > 0.5 else ("valu-issue" if fv > frac_hbm else "hbm")
    return {"frac_valu_issue": fv, "frac_valu_issue_at_2p4_cycles": kd.get("frac_at_2p4_cycles"), "wait_share": ws, "limiter": lim,
            "limiter_note": "`limiter`: `latency` when the launch's wavefronts are parked at s_waitcnt for more than half of their cycles "
                            "(wait_share); otherwise the roof it sits closer to.  bound / achieved / peak / frac stay the HBM figures (SURVEY 8(d) "
                            "bytes over the HBM peak); frac_valu_issue is the same launch against the 4.2-cycle VALU roof, ..._at_2p4_cycles "
                            "against the 2.4-cycle one"}


def committed_counters(name):
    """profiles/<name> -- counter passes are separate rocprofv3 runs (tools/evidence.sh: FETCH_SIZE / WRITE_SIZE passes, tools/sq_round.sh), their
    results are committed.  They only describe THIS build if they were taken on the same kernel sources: the files carry
    the digest of those sources (openjph_amd/build.py kernel_sources_digest) and a file with another digest, or none,
    is refused: -> ({}, "stale: ...").  -> (contents, "current") otherwise."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return {}, "missing"
    from openjph_amd.build import kernel_sources_digest
    have, want = d.get("_kernels_sha256"), kernel_sources_digest()
    if have != want:
        return {}, "stale: taken on kernel sources %s, this build is %s" % ((have or "unstamped")[:12], want[:12])
    return d, "current"


def roofline_valu(workload, kinfo):
    """The block coder is bound by instruction issue, not by HBM: its launches against THAT roof.  Wavefront
    instructions come from the committed SQ counter pass (profiles/sq_counters.json, tools/sq_round.sh: SQ_INSTS_VALU
    / SQ_INSTS_SALU summed over the launch, stamped with the digest of the kernel sources and refused when stale); a SIMD
    issues one wave64 VALU instruction of the coders' mix per ~4 cycles (tools/micro/valu_issue.hip), the chip has 1024 SIMDs
    at up to 2.4 GHz; the time comes from this run."""
    sq, state = committed_counters("sq_counters.json")
    if state != "current":
        return {"limiter": None, "kernels": {}, "source": "profiles/sq_counters.json (%s)" % state}
    sq = sq.get(workload, {})
    waits = sq.get("_waits", {})
    out = {}
    for k, v in kinfo.items():
        base = k.split("(")[0].split("[")[0]
        c = sq.get(k) or sq.get(base)
        if not c or not v["ms"]:
            continue
        # two roofs, from the probe's two instruction classes: every VALU instruction of the cheap class (2.4 cycles per wave64
        # instruction per SIMD) / of the dear one (4.2); the coders' mix lies between, nearer the dear one.  `frac` is the
        # dear-class figure (the old 4-cycle roof was about that); frac_at_2p4_cycles the guide's optimistic one.
        ms_at = lambda cyc: c["valu_insts"] * cyc / (1024 * 2.4e9) * 1e3
        w = waits.get(k) or waits.get(base) or {}
        out[k] = {"valu_wave_insts": c["valu_insts"], "salu_wave_insts": c.get("salu_insts"), "valu_issue_ms_at_4p2_cycles": round(ms_at(4.2), 4),
                  "valu_issue_ms_at_2p4_cycles": round(ms_at(2.4), 4), "measured_ms": v["ms"],
                  "frac": round(ms_at(4.2) / v["ms"], 3), "frac_at_2p4_cycles": round(ms_at(2.4) / v["ms"], 3),
                  "wait_share": w.get("wait_share"), "issue_stall_share": w.get("issue_stall_share"),
                  "limiter": "latency" if (w.get("wait_share") or 0) > 0.5 else "valu-issue"}
    if not out:
        return None
    return {"limiter": "valu-issue or latency, per kernel: `latency` where the kernel's wavefronts spend more than half of their cycles parked at s_waitcnt "
                     "(wait_share = SQ_WAIT_ANY / SQ_WAVE_CYCLES of the committed SQ pass)",
            "peak": "1024 SIMDs x 1 wave64 VALU instruction / {2.4, 4.2} cycles x 2.4 GHz (tools/micro/valu_issue.hip, profiles/r03_valu_issue_probe.txt: "
                    "4.2 cycles per instruction per SIMD for shifts-left / bit-field / compare / select / cross-lane / 3-operand forms, "
                    "2.4 for add / and / or / xor / shift-right / fp32 add-mul-fma; the coders' mix is mostly the former)", "kernels": out,
            "source": "profiles/sq_counters.json (current: same kernel sources, %s)" % json.load(open(os.path.join(ROOT, "profiles", "sq_counters.json")))["_kernels_sha256"][:12],
            "issue_rate_probe": "tools/micro/valu_issue.hip, output in profiles/"}


def pcie_bandwidth(torch, nbytes=256 << 20, reps=4):
    """hipMemcpyAsync between hipHostMalloc memory (what the pipes' slots are made of) and HBM, one direction at a time -- a
    PROBE of the link as a plain copy sees it, not a roof: the encoder pipe's uploads reach the same or more, because several
    copies are in flight.  (Round 3 timed torch's pin_memory() buffers here and got 33 GB/s host-to-device under a pipe that
    moved 51 GB/s: that allocation is registered pageable memory, placed wherever the first touch put it.)"""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hptr, dptr = ctypes.c_void_p(), ctypes.c_void_p()
    if hip.hipHostMalloc(ctypes.byref(hptr), ctypes.c_size_t(nbytes), ctypes.c_uint(0)) != 0:
        return {"error": "hipHostMalloc failed"}
    out = {"probe": "hipMemcpyAsync, hipHostMalloc <-> HBM, %d MiB x %d, one direction at a time (not a roof)" % (nbytes >> 20, reps)}
    try:
        if hip.hipMalloc(ctypes.byref(dptr), ctypes.c_size_t(nbytes)) != 0:
            return {"error": "hipMalloc failed"}
        ctypes.memset(hptr, 1, nbytes)                   # first touch by this thread
        for name, (dst, src, kind) in dict(h2d=(dptr, hptr, 1), d2h=(hptr, dptr, 2)).items():
            hip.hipMemcpyAsync(dst, src, ctypes.c_size_t(nbytes), ctypes.c_int(kind), None); hip.hipDeviceSynchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                hip.hipMemcpyAsync(dst, src, ctypes.c_size_t(nbytes), ctypes.c_int(kind), None)
            hip.hipDeviceSynchronize()
            out[name] = round(nbytes * reps / (time.perf_counter() - t0) / 1e9, 1)
    finally:
        if dptr: hip.hipFree(dptr)
        hip.hipHostFree(hptr)
    return out


def run_encoder_pipe(plan, img, n, depth=4, threads=2, container=16, want=None, packed=None, start=None):
    """n frames through an encoder pipe in steady state; every slot is filled once (the frame a capture device would
    have written there), later submissions send the slot again: each frame pays its H2D, kernels, Tier-2, D2H"""
    from openjph_amd.pipeline import EncoderPipe
    pipe = EncoderPipe(plan=plan, depth=depth, container=container, host_threads=threads, packed=packed)
    k = 0
    filled = None
    if packed:
        from openjph_amd.pipeline import pack_bits
        filled = pack_bits(img, packed)                # what a capture device delivering packed samples would have written
    while k < depth:
        buf = pipe.acquire()
        if buf is None:
            break
        buf[:] = filled if packed else img.astype(buf.dtype)
        pipe.submit(); k += 1
    first = None
    while pipe.in_flight:
        c = pipe.collect()
        first = c if first is None else first
    if want is not None:
        assert first == want, "pipeline codestream differs from the one-frame encoder's"
    if start is not None:
        start.wait(300)                                # (both pipes at once: the timed regions begin together, after both have filled their slots)
    t0 = time.perf_counter()
    sub = col = 0
    nbytes = 0
    while col < n:
        while sub < n and pipe.acquire() is not None:
            pipe.submit(); sub += 1
        nbytes += len(pipe.collect(copy=False)); col += 1
    dt = time.perf_counter() - t0
    st = pipe.stats()
    pipe.close()
    assert nbytes == n * len(first)
    return dt, st


def run_decoder_pipe(cs, n, depth=4, threads=2, container=16, want=None, packed=None, start=None):
    from openjph_amd.pipeline import DecoderPipe
    pipe = DecoderPipe(cs, depth=depth, container=container, host_threads=threads, packed=packed)
    k = 0
    while k < depth:
        buf = pipe.acquire(len(cs))
        if buf is None:
            break
        buf[:] = np.frombuffer(cs, np.uint8)
        pipe.submit(); k += 1
    first = None
    while pipe.in_flight:
        f = pipe.collect()
        first = f if first is None else first
    if want is not None:
        if packed:
            from openjph_amd.pipeline import unpack_bits
            first = unpack_bits(first, packed, want.size).reshape(want.shape)
            want = np.clip(want.astype(np.int64), 0, (1 << packed) - 1)      # packing clamps (a 9/7 decode can leave 2^B)
        assert np.array_equal(first.astype(np.int64), want.astype(np.int64)), "pipeline frame differs from the one-frame decoder's"
    if start is not None:
        start.wait(300)
    t0 = time.perf_counter()
    sub = col = 0
    while col < n:
        while sub < n and pipe.acquire(len(cs)) is not None:      # the slot still holds the codestream
            pipe.submit(); sub += 1
        pipe.collect(copy=False); col += 1
    dt = time.perf_counter() - t0
    st = pipe.stats()
    pipe.close()
    return dt, st


def e2e_pipelines(plan, img, cs, n, container, torch):
    """steady-state Msamples/s of the encoder pipe, the decoder pipe, and both at once (a transcoder: every step one
    frame goes in and one comes out in each direction)"""
    import threading
    nsamp = img.size
    depth, threads = 6, 4                   # slots per pipe; host threads per pipe (finishers / parse workers)
    out = {"frames": n, "depth": depth, "host_threads": threads, "sample_container_bits": container, "memcpy_probe_GBps": pcie_bandwidth(torch)}
    if isinstance(cs, (list, tuple)):
        cs = cs[0]
    dt, st = run_encoder_pipe(plan, img, n, depth, threads, container=container, want=cs)
    out["encode"] = {"Msamples_s": round(nsamp * n / dt / 1e6, 1), "ms_per_frame": round(dt * 1e3 / n, 3),
                     "host_tier2_ms": round(st["host_tier2_ms"], 3), "latency_ms": round(st["latency_ms"], 2), "tier2_threads": st["tier2_threads"]}
    dt, st = run_decoder_pipe(cs, n, depth, threads, container=container)
    out["decode"] = {"Msamples_s": round(nsamp * n / dt / 1e6, 1), "ms_per_frame": round(dt * 1e3 / n, 3),
                     "host_parse_ms": round(st["host_parse_ms"], 3), "latency_ms": round(st["latency_ms"], 2)}
    res = {}
    gate = threading.Barrier(2)                            # filling an encoder pipe's slots takes longer than a decoder pipe's whole timed run
    te = threading.Thread(target=lambda: res.__setitem__("e", run_encoder_pipe(plan, img, n, depth, threads, container=container, start=gate)))
    td = threading.Thread(target=lambda: res.__setitem__("d", run_decoder_pipe(cs, n, depth, threads, container=container, start=gate)))
    t0 = time.perf_counter()
    te.start(); td.start(); te.join(); td.join()
    if "e" in res and "d" in res:
        wall = max(res["e"][0], res["d"][0])               # both pipes code n frames; the slower one sets the step rate
        out["encode+decode"] = {"Msamples_s": round(nsamp * n / wall / 1e6, 1), "ms_per_step": round(wall * 1e3 / n, 3),
                                "encode_ms_per_frame": round(res["e"][0] * 1e3 / n, 3), "decode_ms_per_frame": round(res["d"][0] * 1e3 / n, 3)}
    # the same pipes with the frames crossing PCIe as bit-packed planes (1.5 bytes per 12-bit sample instead of 2): the link
    # is what bounds the figures above
    try:
        fmts = [plan.comp_format(c) for c in range(img.shape[0])]
        bd = 0 if any(sg for _, sg in fmts) else max(b for b, _ in fmts)
    except Exception:
        bd = 0
    if container == 16 and bd in (10, 12, 14) and img.ndim == 3:
        try:
            dt, _ = run_encoder_pipe(plan, img, n, depth, threads, container=container, packed=bd)
            pe = round(nsamp * n / dt / 1e6, 1)
            dt, _ = run_decoder_pipe(cs, n, depth, threads, container=container, packed=bd)
            out["bit_packed"] = {"bits_per_sample_on_pcie": bd, "encode_Msamples_s": pe, "decode_Msamples_s": round(nsamp * n / dt / 1e6, 1)}
            # ... and both directions at once (the transcoder case): the two directions share ~70 GB/s of link payload
            res = {}
            gate = threading.Barrier(2)
            te = threading.Thread(target=lambda: res.__setitem__("e", run_encoder_pipe(plan, img, n, depth, threads, container=container, packed=bd, start=gate)))
            td = threading.Thread(target=lambda: res.__setitem__("d", run_decoder_pipe(cs, n, depth, threads, container=container, packed=bd, start=gate)))
            te.start(); td.start(); te.join(); td.join()
            if "e" in res and "d" in res:
                wall = max(res["e"][0], res["d"][0])
                out["bit_packed"]["encode+decode_Msamples_s"] = round(nsamp * n / wall / 1e6, 1)
                out["bit_packed"]["encode+decode_ms_per_step"] = round(wall * 1e3 / n, 3)
        except Exception as e:
            out["bit_packed"] = {"error": str(e)[:200]}
    return out


def cpu_baseline(img, bd, rev, ct, qstep, tile, reps):
    """The reference library (its own SIMD dispatch) on this host, one thread, same frame."""
    from oracle import refbind
    if not refbind.available():
        return {"value": None, "unit": "Msamples/s", "cores": 0, "kind": "reference",
                "sample": "oracle/_ref/libojph_ref.so missing"}
    r = refbind.Ref()
    best_e = best_d = 1e30
    cs = None
    for _ in range(max(1, reps)):
        t0 = time.perf_counter()
        cs = r.encode(img, bd, reversible=rev, color_transform=ct, qstep=qstep, tile=tile)
        t1 = time.perf_counter()
        r.decode(cs)
        t2 = time.perf_counter()
        best_e = min(best_e, t1 - t0); best_d = min(best_d, t2 - t1)
    n = img.size
    out = {"value": round(n / (best_e + best_d) / 1e6, 2), "unit": "Msamples/s", "cores": 1, "kind": "reference",
           "note": "ATK workloads: the reference has no ATK writer; timed is its plain 9/7 on the same frame -- every wavelet of the "
                   "reference runs through the one general lifting path (ojph_transform.cpp:209-852), so this IS the arithmetic an ATK "
                   "codestream of these steps costs it",
            "covers": "the whole library call: sample conversion + DWT + block coder + its Tier-2 and memory-file I/O "
                      "(compare with e2e_steady_Msamples_s; `value` is the device-resident step without Tier-2 / PCIe)",
            "sample": "the full %dx%dx%d frame, best of %d (encode %.3f s, decode %.3f s; simd level %d; host has %d cpus)"
                      % (img.shape[2], img.shape[1], img.shape[0], reps, best_e, best_d, r.simd_level(), os.cpu_count()),
            "encode_Msamples_s": round(n / best_e / 1e6, 2), "decode_Msamples_s": round(n / best_d / 1e6, 2)}
    return out


def _cpu_worker(args):
    """one process = one reference codestream object coding WHOLE frames (the library is single-threaded;
    independent frames are how it scales on a host)"""
    bd, rev, ct, qstep, tile, rounds = args
    from oracle import refbind
    r = refbind.Ref()
    img = _CPU_FRAME                                  # inherited through fork (copy-on-write: never written)
    t0 = time.perf_counter()
    for _ in range(rounds):
        cs = r.encode(img, bd, reversible=rev, color_transform=ct, qstep=qstep, tile=tile)
        r.decode(cs)
    return time.perf_counter() - t0


_CPU_FRAME = None


def cpu_baseline_all_cores(img, bd, rev, ct, qstep, tile=(0, 0), rounds=1):
    """aggregate encode+decode rate of P independent processes, each coding the whole frame `rounds` times.  P = all
    of os.cpu_count(), lowered only if the host's free memory would not hold P decoded frames (a process needs the
    decoded int32 frame + codestream + the library's line buffers: about 6 bytes per sample)"""
    import multiprocessing as mp
    global _CPU_FRAME
    cpus = os.cpu_count() or 1
    per_proc = img.size * 6 + (64 << 20)
    try:
        avail = [int(l.split()[1]) * 1024 for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0]
    except Exception:
        avail = 64 << 30
    procs = max(1, min(cpus, int(avail * 0.4 // per_proc)))
    _CPU_FRAME = img
    ctx = mp.get_context("fork")
    small = (bd, rev, ct, qstep, tile, 0)
    with ctx.Pool(procs) as pool:
        pool.map(_cpu_worker, [small] * procs, chunksize=1)          # start the workers, load the library
        t0 = time.perf_counter()
        pool.map(_cpu_worker, [(bd, rev, ct, qstep, tile, rounds)] * procs, chunksize=1)
        wall = time.perf_counter() - t0
    _CPU_FRAME = None
    n = img.size * rounds * procs
    return {"value": round(n / wall / 1e6, 2), "unit": "Msamples/s", "cores": procs, "host_cpus": cpus,
            "sample": "%d processes (host has %d cpus), each encode+decode of the whole %dx%dx%d frame x %d (%.2f s wall); "
                      "includes the reference's Tier-2 and memory-file I/O, like the single-thread figure"
                      % (procs, cpus, img.shape[2], img.shape[1], img.shape[0], rounds, wall)}


if __name__ == "__main__":
    main()
