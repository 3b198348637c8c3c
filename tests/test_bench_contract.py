"""The one JSON line bench.py prints (driver contract): checked on the last committed bench line of
profiles/ (CPU) and on a live short run (GPU)."""
import glob
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric": str, "value": (int, float), "unit": str, "n_gpus": int, "steps": int, "warmup": int,
            "ms_per_step": (int, float), "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict,
            "roofline": dict}


def check_line(d, want_cpu_baseline):
    for k, t in REQUIRED.items():
        assert k in d and isinstance(d[k], t), k
    assert "vs_baseline" in d and d["vs_baseline"] is None          # BASELINE.md holds no number for this metric
    assert d["higher_is_better"] is True and d["scaling"] in ("weak", "strong") and d["data"] == "synthetic"
    assert d["unit"] == "Msamples/s" and d["value"] > 0 and d["ms_per_step"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    for r in (d["roofline"], d.get("roofline_dwt", d["roofline"])):
        # `bound` is the driver contract's field: the roof achieved / peak / frac are quoted against, "hbm" | "mfma" and nothing
        # else.  What limits the launch according to the committed SQ counters of the build is a field of its own, `limiter`
        # (bench.limiter_fields): None when those counters are stale, else one of the three below, consistent with its inputs
        assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] > 0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        assert r["traffic"] is None or r["traffic"] > 0
        lim = r.get("limiter")
        assert lim in (None, "hbm", "valu-issue", "latency")
        if lim == "latency":
            assert r["wait_share"] > 0.5 and "limiter_note" in r
        elif lim == "valu-issue":
            assert r["frac_valu_issue"] > r["frac"] and (r["wait_share"] or 0) <= 0.5 and "limiter_note" in r
        elif lim == "hbm":
            assert r["frac_valu_issue"] <= r["frac"] and (r["wait_share"] or 0) <= 0.5
    # the timed region is verified after it ran: the last timed step's block verdicts, samples and codestream
    assert d["verified_after_timing"] is True and d["failed_blocks"] == 0 and d["fused_retries"] >= 0
    assert d["fused_giveups_in_timed_region"] is False and d["codestream_last_step_equals_first"] is True
    assert d["roundtrip_max_abs_err_last_step"] == d["config"]["roundtrip_max_abs_err"]
    if d["config"]["workload"][:2] in ("c2", "c3") and d["config"]["frames_per_step"] == d["n_gpus"]:
        assert d["codestream_last_step_equals_reference_digest"] is True
    # value is what the step time says: samples of all frames of the step / time
    c = d["config"]
    samples = c["width"] * c["height"] * c["components"] * c["frames_per_step"]
    assert abs(d["value"] - samples / d["ms_per_step"] / 1e3) / d["value"] < 1e-3
    if want_cpu_baseline:
        b = d["cpu_baseline"]
        assert b["kind"] in ("reference", "port") and b["cores"] >= 1 and b["unit"] == "Msamples/s" and b["sample"]


def test_committed_bench_line_keeps_the_contract():
    # the NEWEST committed headline line (profiles/rNN_x_bench.json; rNN_x_bench_<other workload>.json are not it): a change of
    # bench.py's line that the contract does not know fails HERE, on the CPU, before it can fail on the driver's GPU box
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*_bench.json")))
    assert files
    d = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    assert int(os.path.basename(files[-1])[1:3]) >= 6, files[-1]
    check_line(d, want_cpu_baseline=True)
    assert d["config"]["workload"] == "c3_8k_444_12b_irv97" and d["n_gpus"] == 1
    # the PMC traffic file carries the kernels the bench line names, and the digest of the kernel sources it was taken on
    pmc_all = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    assert d["roofline"]["kernel"] in pmc_all[d["config"]["workload"]] and len(pmc_all["_kernels_sha256"]) == 64
    assert len(json.load(open(os.path.join(ROOT, "profiles", "sq_counters.json")))["_kernels_sha256"]) == 64
    # a timed region an external sampler can see, the pipelines' end-to-end figures, every host cpu in the all-core
    # baseline, per-rank times, the strong-scaling part (16K frame, 256 tiles) with the reference's codestream digest
    assert d["steps"] >= 1000 and d["steps"] * d["ms_per_step"] >= 1000.0
    e = d["e2e_steady_Msamples_s"]
    assert e["encode"] > 10000 and e["decode"] > 10000 and d["e2e"]["frames"] >= 32
    assert d["cpu_baseline"]["all_cores"]["cores"] == d["cpu_baseline"]["all_cores"]["host_cpus"]
    assert len(d["per_rank_ms_per_step"]) == d["n_gpus"] and "value_covers" in d
    s = d["strong_scaling_c4"]
    assert s["scaling"] == "strong" and s["tiles"] == 256 and s["codestream_equals_reference_digest"] is True and s["value"] > 0
    assert d["dist"]["world_size"] == 1


def test_stale_counter_files_are_refused(tmp_path, monkeypatch):
    """bench.py only quotes committed PMC / SQ counters taken on the kernel sources it runs"""
    import bench
    from openjph_amd.build import kernel_sources_digest
    prof = tmp_path / "profiles"; prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (prof / "x.json").write_text(json.dumps({"_kernels_sha256": "0" * 64, "w": {"k": 1}}))
    got, state = bench.committed_counters("x.json")
    assert got == {} and state.startswith("stale")
    (prof / "x.json").write_text(json.dumps({"_kernels_sha256": kernel_sources_digest(), "w": {"k": 1}}))
    got, state = bench.committed_counters("x.json")
    assert got["w"]["k"] == 1 and state == "current"
    assert bench.committed_counters("missing.json") == ({}, "missing")




@pytest.mark.parametrize("frac_hbm,kd,want", [
    (0.19, {"frac": 0.66, "frac_at_2p4_cycles": 0.38, "wait_share": 0.60}, "latency"),
    (0.19, {"frac": 0.66, "frac_at_2p4_cycles": 0.38, "wait_share": 0.31}, "valu-issue"),
    (0.19, {"frac": 0.66, "frac_at_2p4_cycles": 0.38, "wait_share": None}, "valu-issue"),
    (0.52, {"frac": 0.20, "frac_at_2p4_cycles": 0.11, "wait_share": 0.45}, "hbm"),
    (0.52, {"frac": 0.20, "frac_at_2p4_cycles": 0.11, "wait_share": 0.51}, "latency"),
    (0.52, {}, None), (0.52, None, None),
])
def test_every_limiter_label_bench_can_print_passes_the_contract(frac_hbm, kd, want):
    """bench.limiter_fields on synthetic counters, one case per branch -- and a whole line carrying each label goes through
    check_line (round 5 ended with the driver's GPU suite red because bench.py printed a label this file did not know)"""
    import bench
    f = bench.limiter_fields(frac_hbm, kd)
    assert f["limiter"] == want
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*_bench.json")))
    d = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    for k in ("frac_valu_issue", "frac_valu_issue_at_2p4_cycles", "wait_share", "limiter", "limiter_note"):
        d["roofline"].pop(k, None)
    d["roofline"]["frac"] = frac_hbm; d["roofline"]["achieved"] = frac_hbm * d["roofline"]["peak"]
    d["roofline"].update(f)
    check_line(d, want_cpu_baseline=True)


def test_roofline_valu_labels_follow_the_committed_counters(tmp_path, monkeypatch):
    """bench.roofline_valu end to end on a synthetic profiles/sq_counters.json stamped with this build's digest"""
    import bench
    from openjph_amd.build import kernel_sources_digest
    prof = tmp_path / "profiles"; prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    sq = {"_kernels_sha256": kernel_sources_digest(),
          "w": {"ht_dec_fused": {"valu_insts": 3.0e7, "salu_insts": 1.0e7}, "ht_encode": {"valu_insts": 9.0e7, "salu_insts": 7.0e7},
                "_waits": {"ht_dec_fused": {"wait_share": 0.6, "issue_stall_share": 0.1}, "ht_encode": {"wait_share": 0.2}}}}
    (prof / "sq_counters.json").write_text(json.dumps(sq))
    kinfo = {"ht_dec_fused(step 1 + step 2)": {"ms": 0.30}, "ht_encode[top resolution, side stream]": {"ms": 0.24}, "dwt_forward(level 1)": {"ms": 0.1}}
    rv = bench.roofline_valu("w", kinfo)
    ks = rv["kernels"]
    assert ks["ht_dec_fused(step 1 + step 2)"]["limiter"] == "latency" and ks["ht_encode[top resolution, side stream]"]["limiter"] == "valu-issue"
    assert "dwt_forward(level 1)" not in ks
    assert bench.limiter_fields(0.19, ks["ht_dec_fused(step 1 + step 2)"])["limiter"] == "latency"
    sq["_kernels_sha256"] = "0" * 64
    (prof / "sq_counters.json").write_text(json.dumps(sq))
    rv = bench.roofline_valu("w", kinfo)
    assert rv["kernels"] == {} and rv["limiter"] is None and "stale" in rv["source"]


def test_gpus_flag_without_enough_devices_fails_loudly():
    """`python bench.py --gpus 2` with no launcher starts its own ranks -- and must refuse, not run a silent 1-GPU bench,
    when the node does not have that many GPUs (this container has none)"""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "OJPH_BENCH_ONE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and not r.stdout.strip()
    assert b"--gpus 2" in r.stderr and b"visible" in r.stderr
    # under a launcher the flag must agree with WORLD_SIZE
    env["WORLD_SIZE"] = "4"; env["RANK"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, cwd=ROOT)
    assert r.returncode != 0 and b"WORLD_SIZE=4" in r.stderr
