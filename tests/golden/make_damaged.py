#!/usr/bin/env python3
"""tests/golden/damaged.json: what the LIVE reference (generic build, oracle/_ref) makes of the damaged codestreams of
tests/damaged_cases.py, read with and without resilience -- "raises", or the digest of its picture; cases on which it does not
come back within a few seconds are left out.  Run where /root/reference exists:      python tests/golden/make_damaged.py"""
import json, multiprocessing, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import refbind                       # noqa: E402
from tests.damaged_cases import cases, digest    # noqa: E402

_R = {}


def job(part, resilient):
    if "r" not in _R:
        _R["r"] = refbind.Ref(generic=True)
    try:
        return digest(_R["r"].decode(part, resilient=resilient, max_samples=1 << 22)[0])
    except refbind.TooLarge:
        return "large"
    except RuntimeError:
        return "raises"


def main():
    ctx = multiprocessing.get_context("fork")
    pool = ctx.Pool(1)
    out = {}
    for name, part in cases():
        for resilient in (False, True):
            try:
                v = pool.apply_async(job, (part, resilient)).get(8)
            except multiprocessing.TimeoutError:
                pool.terminate(); pool.join(); pool = ctx.Pool(1)
                v = "spins"
            if v not in ("spins", "large"):
                out["%s_%d" % (name, int(resilient))] = v
    pool.terminate()
    json.dump({"reference": "aous72/OpenJPH 0.31.0, generic build (oracle/_ref/libojph_refgen.so)", "cases": out},
              open(os.path.join(HERE, "damaged.json"), "w"), indent=0, sort_keys=True)
    print(len(out), "verdicts;", sum(v == "raises" for v in out.values()), "raise")


if __name__ == "__main__":
    main()
