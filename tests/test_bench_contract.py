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
        assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] > 0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
        assert r["traffic"] is None or r["traffic"] > 0
    # value is what the step time says: samples of all frames of the step / time
    c = d["config"]
    samples = c["width"] * c["height"] * c["components"] * c["frames_per_step"]
    assert abs(d["value"] - samples / d["ms_per_step"] / 1e3) / d["value"] < 1e-3
    if want_cpu_baseline:
        b = d["cpu_baseline"]
        assert b["kind"] in ("reference", "port") and b["cores"] >= 1 and b["unit"] == "Msamples/s" and b["sample"]


def test_committed_bench_line_keeps_the_contract():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r02_*_bench.json")))
    assert files
    d = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    check_line(d, want_cpu_baseline=True)
    assert d["config"]["workload"] == "c3_8k_444_12b_irv97" and d["n_gpus"] == 1
    # the PMC traffic file carries the kernels the bench line names
    pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))[d["config"]["workload"]]
    assert d["roofline"]["kernel"] in pmc
    # round 2: a timed region an external sampler can see, the pipelines' end-to-end figures, every host cpu in the
    # all-core baseline, the VALU roof of the block coder
    assert d["steps"] >= 1000 and d["steps"] * d["ms_per_step"] >= 1500.0
    e = d["e2e_steady_Msamples_s"]
    assert e["encode"] > 10000 and e["decode"] > 10000 and d["e2e"]["frames"] >= 32
    assert d["cpu_baseline"]["all_cores"]["cores"] == d["cpu_baseline"]["all_cores"]["host_cpus"]
    assert "ht_encode[top resolution, side stream]" in d["roofline_valu"]["kernels"]
    assert len(d["per_rank_ms_per_step"]) == d["n_gpus"] and "value_covers" in d


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
