"""The host-side helpers that build / read the bit-packed sample strings of the pipes (openjph_amd/pipeline.py)."""
import numpy as np
import pytest


@pytest.mark.parametrize("bits", [10, 12, 14])
@pytest.mark.parametrize("n", [1, 31, 32, 33, 1000, 4097])
def test_pack_unpack_roundtrip(bits, n):
    from openjph_amd.pipeline import pack_bits, unpack_bits
    rng = np.random.default_rng(bits * 1000 + n)
    a = rng.integers(0, 1 << bits, n)
    p = pack_bits(a, bits)
    assert p.dtype == np.uint8 and p.size == (n + 31) // 32 * 4 * bits          # whole groups of 32 samples
    assert np.array_equal(unpack_bits(p, bits, n), a)
    # the layout the device kernels read: sample i occupies bits [i * bits, (i + 1) * bits) of a little-endian bit string
    big = int.from_bytes(p.tobytes(), "little")
    for i in (0, n // 2, n - 1):
        assert (big >> (i * bits)) & ((1 << bits) - 1) == int(a[i])


def test_the_12_bit_fast_path_equals_the_generic_one():
    from openjph_amd import pipeline
    a = np.random.default_rng(1).integers(0, 4096, 999)
    fast = pipeline.pack_bits(a, 12)
    v = np.concatenate([a.astype(np.uint64), np.zeros((-a.size) % 32, np.uint64)])
    b = ((v[:, None] >> np.arange(12, dtype=np.uint64)[None, :]) & 1).astype(np.uint8)
    assert np.array_equal(fast, np.packbits(b.reshape(-1), bitorder="little"))
