#!/usr/bin/env python3
"""Block level: the oracle's HT block decoders (32- and 64-bit sample paths) against the LIVE reference's
ojph_decode_codeblock32 / 64 (generic C++) on DAMAGED cleanup segments -- changed bytes, changed Scup, cut and
grown segments, random bytes -- of every block shape (odd widths and heights included), with any missing_msbs,
and with SigProp / MagRef segments of random bytes behind them: the same verdict (decoded / refused) and, when
decoded, the same samples.  CPU only.      python tools/fuzz_blocks_cpu.py [seconds] [first seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import refbind, oraclebind as ob
from tests.synth import random_block


def damaged_blocks(seed):
    """the damaged blocks of one seed: (wide, w, h, kmax, mmsb, bytes, len2, passes, causal) -- shared with tools/fuzz_blocks_gpu.py"""
    rng = np.random.default_rng(seed)
    wide = rng.random() < 0.3
    w = int(rng.integers(1, 65)) if rng.random() < 0.7 else int(rng.choice([1, 2, 3, 4, 5, 31, 32, 33, 63, 64, 127, 128, 255, 256, 512, 1024]))
    hmax = max(1, min(64, 4096 // w))
    h = int(rng.integers(1, hmax + 1))
    kmax = int(rng.integers(2, 57 if wide else 30))
    if wide:
        mag = rng.integers(0, 1 << min(kmax, 62), (h, w), dtype=np.uint64) >> rng.integers(0, kmax, (h, w)).astype(np.uint64)
        mag[rng.random((h, w)) > rng.uniform(0.05, 0.9)] = 0
        sm = (mag << np.uint64(63 - kmax)) | (rng.integers(0, 2, (h, w), dtype=np.uint64) << np.uint64(63))
        sm[mag == 0] = 0
        good = ob.ht_encode64(sm, w, h, w, kmax - 1)
    else:
        sm, _ = random_block(rng, w, h, w, kmax, float(rng.uniform(0.05, 0.9)), int(rng.integers(1, 1 << min(kmax, 20))))
        good = ob.ht_encode(sm, w, h, w, kmax - 1, 0)
    if len(good) < 2:
        return
    for trial in range(24):
        b = bytearray(good)
        style = int(rng.integers(0, 7))
        if style == 0:
            for _ in range(int(rng.integers(1, 5))):
                b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        elif style == 1:                                        # Scup nibble / byte
            b[-1 - int(rng.integers(0, 2))] = int(rng.integers(0, 256))
        elif style == 2:                                        # the MEL / VLC tail
            for _ in range(int(rng.integers(1, 4))):
                b[len(b) - 1 - int(rng.integers(0, min(64, len(b))))] = int(rng.choice([0xFF, 0x7F, 0x8F, 0x90, 0, int(rng.integers(0, 256))]))
        elif style == 3:
            b = b[:int(rng.integers(2, len(b) + 1))]
        elif style == 4:
            b = bytearray(rng.integers(0, 256, int(rng.integers(2, 200)), dtype=np.uint8).tobytes())
        elif style == 5:                                        # runs of 0xFF / 0x7F (the stuffing rules)
            at = int(rng.integers(0, len(b))); ln = int(rng.integers(1, 6))
            for i in range(at, min(len(b), at + ln)):
                b[i] = int(rng.choice([0xFF, 0x7F, 0xFF, 0x8F]))
        else:
            b += bytes(rng.integers(0, 256, int(rng.integers(1, 4)), dtype=np.uint8).tolist())
        # 64-bit function: its MagSgn window promises 57 bits (frwd_fetch64, ojph_block_decoder64.cpp:741-746), a sample of up
        # to missing_msbs + 2 bits is taken from it: beyond 55 the reference reads what it has not fetched (nothing to match)
        top = 55 if wide else 30
        u = rng.random()
        mmsb = kmax - 1 if u < 0.6 else (min(top, kmax - 1 + int(rng.integers(0, 4))) if u < 0.85 else int(rng.integers(0, top + 1)))
        passes, tail = 1, b""
        if rng.random() < 0.3:
            passes = int(rng.integers(2, 5))
            tail = rng.integers(0, 256, int(rng.integers(0, 40)), dtype=np.uint8).tobytes()
        causal = bool(rng.random() < 0.3)
        if wide and passes > 1 and mmsb < 30:       # SigProp of the 64-bit function below that: a 32-bit shift by 32 or more
            mmsb = int(rng.integers(30, 56))        # (ojph_block_decoder64.cpp:1549, undefined)
        yield wide, w, h, kmax, mmsb, bytes(b) + tail, len(tail), passes, causal


def main(seconds=None, seed=None):
    t_end = time.time() + (seconds if seconds is not None else float(sys.argv[1]) if len(sys.argv) > 1 else 60)
    seed = seed if seed is not None else int(sys.argv[2]) if len(sys.argv) > 2 else 700000
    r = refbind.Ref(generic=True)
    if hasattr(r.lib, "ref_set_verbose"):
        r.lib.ref_set_verbose(0)
    n = bad = refused = shown = 0
    while time.time() < t_end:
        seed += 1
        for trial, (wide, w, h, kmax, mmsb, data, len2, passes, causal) in enumerate(damaged_blocks(seed - 1)):
            # the reference's code-block buffers have a stride of the NOMINAL width rounded up to 8 (ojph_codeblock.cpp:63,82):
            # what MagRef does to a flagged sample beyond an odd width lands in that padding, not in the next row
            st = (w + 8) & ~7
            if wide:
                ok, d = ob.ht_decode64(data, w, h, st, mmsb, len2=len2, num_passes=passes, stripe_causal=causal)
                ok2, d2 = r.decode_block64(data, mmsb, w, h, st, len2=len2, num_passes=passes, stripe_causal=causal)
            else:
                ok, d = ob.ht_decode(data, w, h, st, mmsb, len2=len2, num_passes=passes, stripe_causal=causal)
                ok2, d2 = r.decode_block(data, mmsb, w, h, st, len2=len2, num_passes=passes, variant=0, stripe_causal=causal)
            n += 1; refused += not ok2
            if bool(ok) != bool(ok2) or (ok and not np.array_equal(d[:, :w], d2[:, :w])):
                bad += 1
                if shown < int(os.environ.get("FUZZ_SHOW", "12")):
                    shown += 1
                    print("DIFFERS: seed %d trial %d wide=%s %dx%d kmax %d mmsb %d passes %d len2 %d causal %s: oracle %s, reference %s  data %s" %
                          (seed - 1, trial, wide, w, h, kmax, mmsb, passes, len2, causal, ok, ok2,
                           data.hex() if len(data) <= 48 or os.environ.get("FUZZ_FULL") else "(%d bytes)" % len(data)), flush=True)
    print("%d damaged blocks (the reference refused %d): %d decoded differently by the oracle" % (n, refused, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
