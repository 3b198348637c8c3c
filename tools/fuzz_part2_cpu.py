#!/usr/bin/env python3
"""Random Part-2 configurations (ATK kernels, DFS level kinds, tiles, offsets, block sizes, component styles) written by this
repository's plan / Tier-2 and oracle stages, read back by the LIVE reference (oracle/_ref): the reference must decode every
codestream to exactly the oracle pipeline's samples, at full and at reduced resolution.  CPU only (this container).
    python tools/fuzz_part2_cpu.py [seconds] [seed]"""
import os, sys, time, traceback
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import cpu_pipeline as cp
from tests.synth import synth_image
from oracle import refbind


def rand_case(rng):
    nc = int(rng.integers(1, 4))
    h, w = int(rng.integers(1, 140)), int(rng.integers(1, 160))
    bd = int(rng.integers(2, 15))
    L = int(rng.integers(0, 5))
    kw = dict(bit_depth=bd, num_decomps=L)
    if rng.random() < 0.4:
        kw["tile"] = (int(rng.integers(17, 90)), int(rng.integers(17, 90)))
    if rng.random() < 0.3:
        kw["image_offset"] = (int(rng.integers(0, 9)), int(rng.integers(0, 9)))
    if rng.random() < 0.4:
        kw["block"] = [(64, 64), (32, 32), (16, 64), (128, 32), (8, 8), (4, 256)][int(rng.integers(0, 6))]
    atk, dfs, coc = {}, {}, {}
    rev_frame = rng.random() < 0.6
    kw["reversible"] = rev_frame
    if not rev_frame:
        kw["qstep"] = float(rng.choice([0.05, 0.01, 0.003]))

    def new_atk(rev):
        idx = int(rng.integers(2, 255))
        while idx in atk:
            idx = int(rng.integers(2, 255))
        n = int(rng.choice([2, 2, 2, 4, 3, 6]))
        if rev:
            steps = []
            gmax = float(os.environ.get("FUZZ_GAIN", "1.0"))   # a step adds at most gmax x its neighbours' sum / 2 ... the reference's int32
            for _ in range(n):                                # lifting must not overflow: what it does then is undefined behaviour, nothing to match
                e = int(rng.integers(0, 5))
                amax = max(1, int(gmax * (1 << e) / 2))
                a = int(rng.integers(-amax, amax + 1)) or 1
                steps.append((a, int(rng.integers(0, 1 << e)) if e else 0, e))
            atk[idx] = dict(steps=steps)
        else:
            g = float(os.environ.get("FUZZ_GAIN", "1.0"))     # (large coefficients over many steps and levels take the quantised magnitudes past 31 bits too)
            atk[idx] = dict(steps=[float(np.round(rng.uniform(-1.6 * g, 0.9 * g), 6)) or 0.25 for _ in range(n)], K=float(np.round(rng.uniform(0.8, 1.4), 6)))
        return idx

    def new_dfs(levels):
        idx = int(rng.integers(0, 16))
        while idx in dfs:
            idx = int(rng.integers(0, 16))
        dfs[idx] = [int(rng.choice([1, 1, 2, 3, 0])) for _ in range(levels)]
        return idx

    if rng.random() < 0.5:
        kw["wavelet"] = new_atk(rev_frame)
    for c in range(nc):
        if rng.random() < 0.5:
            rev = bool(rng.random() < 0.5)
            Lc = int(rng.integers(0, 5))
            st = dict(reversible=rev, num_decomps=Lc)
            if rng.random() < 0.5: st["wavelet"] = new_atk(rev)
            if Lc and rng.random() < 0.6: st["dfs"] = new_dfs(Lc)
            if rng.random() < 0.3: st["block"] = [(32, 32), (64, 16), (16, 16)][int(rng.integers(0, 3))]
            coc[c] = st
    if atk: kw["atk"] = atk
    if dfs: kw["dfs"] = dfs
    if coc: kw["coc"] = coc
    if any(not s.get("reversible", rev_frame) for s in coc.values()) and "qstep" not in kw:
        kw["qstep"] = 0.01
    return nc, h, w, bd, kw


def main(seconds=None, seed=None):
    t_end = time.time() + (seconds if seconds is not None else float(sys.argv[1]) if len(sys.argv) > 1 else 60)
    rng = np.random.default_rng(seed if seed is not None else int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    refs = {True: refbind.Ref(generic=False), False: refbind.Ref(generic=True)}
    n = refused = bad = overflowed = 0
    import signal
    def on_alarm(sig, frm): raise TimeoutError("a case took more than 30 s")
    signal.signal(signal.SIGALRM, on_alarm)
    while time.time() < t_end:
        nc, h, w, bd, kw = rand_case(rng)
        signal.alarm(30)
        if os.environ.get("FUZZ_TRACE"):
            open(os.environ["FUZZ_TRACE"], "w").write(repr((nc, h, w, bd, kw)) + "\n")
        img = synth_image(nc, h, w, bd, seed=int(rng.integers(0, 1000)))
        try:
            cs, plan, *_ = cp.encode(img, **kw)
        except TimeoutError as e:
            bad += 1; print("TIMEOUT in encode", nc, h, w, bd, kw, flush=True); continue
        except Exception as e:                               # a configuration the plan refuses (loudly) is not a parity case
            refused += 1
            continue
        n += 1
        try:
            dec, _ = cp.decode(cs)
            rev_all = all(plan.comp_style(i)["reversible"] for i in range(nc))
            want, _ = refs[rev_all].decode(cs)
            ok = np.array_equal(dec, want)
            L0 = min(plan.comp_style(c)["num_decomps"] for c in range(nc))
            # (reduced resolution: not on frames a few samples across -- the reference's own pull loop does not come back on
            #  e.g. a 1 x 14 frame whose component has DFS levels [horz, both, both, both] when one resolution is skipped)
            if ok and L0 >= 1 and min(h, w) >= 8:
                try:
                    d1, _ = cp.decode(cs, skip=(1, 1))
                except Exception as e1:                       # refused here (the COD has fewer decompositions than are skipped): the reference must refuse too
                    try:
                        refs[rev_all].decode(cs, skip=(1, 1)); ok = False
                        print("ONLY WE REFUSE the reduced resolution:", str(e1)[:80], flush=True)
                    except Exception:
                        ok = True
                    d1 = None
                if d1 is not None:
                    w1, _ = refs[rev_all].decode(cs, skip=(1, 1))
                    ok = all(np.array_equal(a, b) for a, b in zip(d1, w1)) if isinstance(w1, list) else np.array_equal(d1, w1)
            if not ok and max(int(np.abs(np.asarray(dec, dtype=np.int64)).max()), int(np.abs(np.asarray(want, dtype=np.int64)).max())) >= (1 << 30):
                overflowed += 1                               # the kernel's gain took the samples past 31 bits: the reference's int32 lifting wraps
                continue                                      # (undefined behaviour in C), the oracle's does not -- nothing to be identical to
            if not ok and bd > 2:                             # ... or an intermediate did: the same configuration on 2-bit samples must agree
                kw2 = dict(kw, bit_depth=2)
                cs2, plan2, *_ = cp.encode(synth_image(nc, h, w, 2, seed=1), **kw2)
                d2, _ = cp.decode(cs2); w2, _ = refs[rev_all].decode(cs2)
                if np.array_equal(d2, w2) and int(np.abs(np.asarray(d2, dtype=np.int64)).max()) < (1 << 30):
                    overflowed += 1
                    continue
            if not ok:
                bad += 1
                print("MISMATCH", nc, h, w, bd, kw, flush=True)
        except Exception as e:
            bad += 1
            print("ERROR %s: %s" % (type(e).__name__, str(e)[:200]), nc, h, w, bd, kw, flush=True)
    signal.alarm(0)
    print("%d random Part-2 codestreams decoded by the live reference like the oracle pipeline (full and half resolution), %d mismatches; "
          "%d configurations refused by the plan; %d set aside: lifting gains that overflow 32 bits" % (n, bad, refused, overflowed))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
