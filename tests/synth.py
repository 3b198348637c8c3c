"""Deterministic synthetic images (SURVEY.md section 8(d) family): smooth + texture + noise, so
that the bit-rate lands near 0.6-0.8 B/sample like natural content."""
import numpy as np


def synth_image(nc, h, w, bit_depth, seed=1234, signed=False):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    half = float(1 << (bit_depth - 1))
    planes = []
    for c in range(nc):
        base = half + 0.44 * half * np.sin(x / (97.0 + 7 * c)) + 0.39 * half * np.cos(y / (61.0 + 5 * c)) \
            + 0.15 * half * np.sin((x + y) / 13.0)
        base = base + rng.normal(0.0, half * 0.047, size=(h, w)).astype(np.float32)
        planes.append(base)
    img = np.clip(np.rint(np.stack(planes)), 0, (1 << bit_depth) - 1).astype(np.int32)
    if signed:
        img -= int(half)
    return img


# ---------------------------------------------------------------------------------------------
# The BASELINE workloads exactly as SURVEY.md section 8(d) specifies them.  The survey's known
# answers (appendix B: KA-3 = 16 674 994 bytes for C2, KA-4 = 72 601 187 bytes / MSE 1.81186 /
# PAE 8 for C3 with the generic reference build) are reproduced by ONE numpy Generator,
# default_rng(1234), that draws C2's noise first -- N(0, 6^2), shape (3, 2160, 3840) -- and then, from
# the continued stream, C3's -- N(0, 40^2), shape (3, 4320, 7680); float64 arithmetic, clip, then
# truncation to the integer container (found by search against the reference built here; pinned by
# tests/test_survey_ka.py and tests/golden/survey_ka.json).
# ---------------------------------------------------------------------------------------------
def _family(h, w, mid, a, b, c, px, py, pxy):
    """mid + a sin(x/px) + b cos(y/py) + c sin((x+y)/pxy), evaluated in float64 in this order (the
    terms depend on x, y and x + y only, so they come from 1-D tables: same values as an mgrid)"""
    x = np.arange(w, dtype=np.float64)
    y = np.arange(h, dtype=np.float64)
    s = np.arange(w + h, dtype=np.float64)
    base = (mid + a * np.sin(x / px))[None, :] + (b * np.cos(y / py))[:, None]
    t = c * np.sin(s / pxy)
    idx = np.arange(h)[:, None] + np.arange(w)[None, :]
    return base + t[idx]


def _noisy(base, noise, maxval):
    return np.clip(base[None] + noise, 0, maxval).astype(np.uint16).astype(np.int32)


def survey_c2(seed=1234):
    """C2: 3840x2160x3 8-bit (KA-3 with the default seed)."""
    rng = np.random.default_rng(seed)
    return _noisy(_family(2160, 3840, 128, 60, 50, 20, 97, 61, 13), rng.normal(0, 6, (3, 2160, 3840)), 255)


def survey_c3(seed=1234, rows=4320):
    """C3: 7680x4320x3 12-bit planar (KA-4 with the default seed).  rows < 4320 gives the top rows of
    the same image (the noise of a (3, rows, 7680) prefix is NOT a prefix of the full draw, so the
    full noise is drawn and cut)."""
    rng = np.random.default_rng(seed)
    rng.normal(0, 6, (3, 2160, 3840))                  # C2's draw comes first in the survey's stream
    noise = rng.normal(0, 40, (3, 4320, 7680))
    img = _noisy(_family(4320, 7680, 2048, 900, 800, 300, 197, 161, 23), noise, 4095)
    return img if rows >= 4320 else np.ascontiguousarray(img[:, :rows])


def survey_c4(seed=1234, size=16384):
    """C4: 16384x16384x1 16-bit, 'the same family scaled to 16 bits' (C2's formula x 256).  Built slab by slab (the
    float64 base and noise of the whole image would take 4 GB); the (x + y) term of a slab is a sliding window over its
    1-D table -- the same values _family gives."""
    rng = np.random.default_rng(seed)
    x = np.arange(size, dtype=np.float64)
    y = np.arange(size, dtype=np.float64)
    s = np.arange(2 * size, dtype=np.float64)
    row = 32768 + 15360 * np.sin(x / 97)
    col = 12800 * np.cos(y / 61)
    win = np.lib.stride_tricks.sliding_window_view(5120 * np.sin(s / 13), size)       # win[i, j] = t[i + j]
    out = np.empty((1, size, size), dtype=np.int32)
    step = 2048
    for r0 in range(0, size, step):
        r1 = min(r0 + step, size)
        base = (row[None, :] + col[r0:r1, None]) + win[r0:r1]
        n = rng.normal(0, 1536, (r1 - r0, size))
        out[0, r0:r1] = np.clip(base + n, 0, 65535).astype(np.uint16)
    return out


def survey_c5(frame=0, seed=1234):
    """C5: one 3840x2160x3 10-bit frame of the batch, 'the C2 family scaled to 10 bits, seed = 1234 + frame'."""
    rng = np.random.default_rng(seed + frame)
    return _noisy(_family(2160, 3840, 512, 240, 200, 80, 97, 61, 13), rng.normal(0, 24, (3, 2160, 3840)), 1023)


def survey_c6(seed=1234):
    """c6 (bench.py): 3840x2160x1, 32-bit unsigned samples -- the C2 family scaled to 32 bits -- in the si32 container the
    reference's line_buf exchanges (values above 2^31 wrap, as they do there)."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:2160, 0:3840].astype(np.float64)
    v = 2.0 ** 31 + 0.47 * 2.0 ** 31 * np.sin(x / 97) + 0.39 * 2.0 ** 31 * np.cos(y / 61) + 0.12 * 2.0 ** 31 * np.sin((x + y) / 13)
    v = v + rng.normal(0, 2.0 ** 26, v.shape)
    v = np.clip(np.rint(v), 0, 2.0 ** 32 - 1).astype(np.int64)
    return v.astype(np.uint64).astype(np.uint32).astype(np.int32)[None]


def survey_c7(seed=1234):
    """c7 (bench.py): 3840x2160x3 12-bit, the C3 family at 4K."""
    rng = np.random.default_rng(seed)
    return _noisy(_family(2160, 3840, 2048, 900, 800, 300, 197, 161, 23), rng.normal(0, 40, (3, 2160, 3840)), 4095)


def c1_image():
    """BASELINE config #1 / reference tests/test_truncated_decode.cpp:111."""
    y, x = np.mgrid[0:256, 0:256]
    return ((x * 7 + y * 13 + ((x * y) >> 3)) & 0xFF).astype(np.int32)[None]


def random_block(rng, w, h, stride, kmax, density, amp):
    v = (rng.integers(-amp, amp + 1, size=(h, stride)) * (rng.random((h, stride)) < density)).astype(np.int64)
    buf = ((v < 0).astype(np.uint32) << np.uint32(31)) | (np.abs(v).astype(np.uint32) << np.uint32(31 - kmax))
    return buf.astype(np.uint32), v


def ka2_block():
    """64x64 block, K_max 10, LCG-driven sparse large + dense small values."""
    s = 12345
    K = 10
    buf = np.zeros((64, 64), np.uint32)
    for y in range(64):
        for x in range(64):
            s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
            v = (((s >> 16) & 0x3FF) - 512) * (1 if (x + y) % 7 == 0 else 0) + (((s >> 8) & 7) - 3)
            buf[y, x] = ((1 << 31) if v < 0 else 0) | (abs(v) << (31 - K))
    return buf
