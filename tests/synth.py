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
