"""The live halves of the bench contract (tests/test_bench_contract.py holds check_line and the CPU half).  This file sorts
LAST on purpose: under `pytest -m gpu -x` every parity test runs before a formatting assertion of the bench line can stop
the run (round 5: one stale label here hid 761 parity tests from the driver)."""
import json
import os
import subprocess
import sys

import pytest

from tests.test_bench_contract import ROOT, check_line


@pytest.mark.gpu
def test_live_bench_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--e2e-frames", "8"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1                                          # ONE JSON line on stdout
    d = json.loads(lines[0])
    check_line(d, want_cpu_baseline=False)
    assert d["steps"] == 3 and d["warmup"] == 1 and d["config"]["roundtrip_max_abs_err"] <= 8
    assert d["e2e"]["frames"] == 8 and d["e2e_steady_Msamples_s"]["encode"] > 0 and d["e2e_steady_Msamples_s"]["encode+decode"] > 0


@pytest.mark.gpu
def test_two_ranks_self_launched_on_one_gpu():
    """the N > 1 path of bench.py end to end on the one GPU of the test box: `--gpus 2` starts two ranks itself (both on
    cuda:0, control and gather traffic over gloo), the line says n_gpus 2, and the tile-sharded 16K frame the two ranks
    assemble is byte-identical to the reference's codestream (digest in tests/golden/survey_ka.json)"""
    env = dict(os.environ, OJPH_BENCH_BACKEND="gloo", OJPH_BENCH_ONE_GPU="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    check_line(d, want_cpu_baseline=False)
    assert d["n_gpus"] == 2 and len(d["per_rank_ms_per_step"]) == 2 and d["dist"]["world_size"] == 2
    assert d["config"]["frames_per_step"] == 2 and d["scaling"] == "weak"
    s = d["strong_scaling_c4"]
    assert s["n_gpus"] == 2 and s["tiles_per_rank"] == [128, 128] and len(s["per_rank_ms_per_step"]) == 2
    assert s["codestream_equals_reference_digest"] is True and s["tiles_lossless_on_every_rank"] is True
    assert s["gather"]["bytes_received_by_rank0"] > 100e6 and s["value"] > 0
    # the two end-to-end forms of the gather: tile-parts sent to rank 0 (gatherv), and every rank placing its own in ONE shared
    # host segment (shard.HostGather, the default of shard.encode_sharded on one node) -- both must be the reference's bytes
    e = s["e2e_encode"]
    assert e["codestream_equals_reference_digest"] is True and e["ms"] > 0
    hs = e["shared_host_segment"]
    assert "error" not in hs, hs
    assert hs["codestream_equals_reference_digest"] is True and hs["ms"] > 0 and hs["Msamples_s"] > 0
