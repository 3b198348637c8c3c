"""Constants / small formulas shared with the C++ side (kept in sync with ojph_plan.cpp)."""


def block_scratch_bytes(w, h, k_max):
    """ojphgpu::block_scratch_bytes (openjph_amd/csrc/ojph_plan.cpp): per-block scratch slot."""
    bits = w * h * (k_max + 2)
    ms = (bits + 6) // 7 + 16
    total = ms + 3072 + 64
    return (total + 63) & ~63
