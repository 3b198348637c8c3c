"""Damaged codestreams, case by case: what tools/fuzz_flip_cpu.py finds statistically, spelt out.  Every case is
read by the LIVE reference (generic build) and by the host parser + oracle pipeline, with and without resilience: the same
verdict, the same picture."""
import numpy as np
import pytest

from openjph_amd import capi
from openjph_amd.plan import parse_codestream
from tests import cpu_pipeline as cp
from tests.synth import synth_image


def _both(refgen, part):
    out = []
    for resilient in (False, True):
        try:
            want, _ = refgen.decode(part, resilient=resilient)
        except RuntimeError:
            want = None
        try:
            pl = parse_codestream(part, resilient=resilient)
            got = cp.inverse_stages(pl, cp.decode_blocks(pl, part, resilient=resilient))
        except (capi.OjphError, RuntimeError):
            got = None
        assert (want is None) == (got is None), "resilient=%s: reference %s, here %s" % (
            resilient, "raises" if want is None else "decodes", "raises" if got is None else "decodes")
        if want is not None:
            assert np.array_equal(np.asarray(got), np.asarray(want)), "resilient=%s: pictures differ" % resilient
        out.append(want)
    return out


@pytest.fixture(scope="module")
def stream(refgen):
    img = synth_image(1, 72, 88, 8, seed=5)
    cs = refgen.encode(img, 8, num_decomps=3, block=(16, 16), prog_order="LRCP", tileparts="R")
    sots = [i for i in range(len(cs) - 12) if cs[i:i + 4] == b"\xff\x90\x00\x0a"]
    assert len(sots) == 4                       # one tile-part per resolution
    clean, _ = refgen.decode(cs)
    return cs, sots, np.asarray(clean)


def test_bytes_between_tile_parts_are_passed_over(refgen, stream):
    """find_marker (ojph_codestream_local.cpp:706-730): whatever lies between a tile-part's end and the next 0xFF90 is skipped"""
    cs, sots, clean = stream
    part = cs[:sots[2]] + b"\x01\x02\x03\x04\x05" + cs[sots[2]:]
    strict, resilient = _both(refgen, part)
    assert np.array_equal(np.asarray(strict), clean) and np.array_equal(np.asarray(resilient), clean)


def test_a_damaged_sod_is_searched_for(refgen, stream):
    """the SOD of a later tile-part overwritten: the search runs into the packet bytes (and finds an SOD there, or the next SOT)"""
    cs, sots, clean = stream
    b = bytearray(cs); b[sots[2] + 12] = 0x2C
    _both(refgen, bytes(b))


def test_unknown_progression_order_reads_no_packet(refgen, stream):
    """tile::parse_tile_header has no branch for a progression byte above 4 (ojph_tile.cpp:900-901): every block stays empty"""
    cs, sots, clean = stream
    cod = cs.find(b"\xff\x52")
    b = bytearray(cs); b[cod + 5] = 7
    strict, resilient = _both(refgen, bytes(b))
    assert strict is not None and len(np.unique(np.asarray(strict))) == 1


def test_a_precinct_whose_header_threw_is_read_again_from_the_next_tile_part(refgen, stream):
    """the first packet header of tile-part 1 claims missing MSBs beyond K_max: an error -- or, resilient, the rest of the tile-part
    is skipped and the SAME precinct is parsed from tile-part 2's bytes (resolution::parse_one_precinct moves on only after a
    parse has returned), with whatever the failed attempt left in its blocks' headers"""
    cs, sots, clean = stream
    hits = 0
    for value in (0x80, 0x81, 0xC0, 0xE0, 0xF0, 0xFF, 0xAA, 0x00):
        b = bytearray(cs); b[sots[1] + 14] = value          # first byte of the tile-part's first packet header
        strict, resilient = _both(refgen, bytes(b))
        hits += strict is None and resilient is not None
    assert hits >= 1


def test_a_tile_part_shorter_than_its_packets_pads_the_last_block(refgen, stream):
    """Psot of the last tile-part made 1 .. 40 bytes smaller while the file still holds the bytes: bb_read_chunk hands the block
    decoder what the tile-part has left plus zeros (ojph_bitbuffer_read.h:134-150) -- refused (Scup reads 0: an error, or a zero
    block when resilient) or, when only refinement bytes are missing, decoded"""
    cs, sots, clean = stream
    raised = 0
    for less in (1, 2, 3, 7, 20, 40):
        b = bytearray(cs)
        psot = int.from_bytes(cs[sots[3] + 6:sots[3] + 10], "big") - less
        b[sots[3] + 6:sots[3] + 10] = psot.to_bytes(4, "big")
        strict, resilient = _both(refgen, bytes(b))
        raised += strict is None
        assert resilient is not None
    assert raised >= 1


def test_damaged_sot_fields(refgen, stream):
    """Lsot, Isot, TPsot, TNsot and Psot of a later tile-part, one at a time (param_sot::read, ojph_params.cpp:2390-2461; the tile index
    and tile-part index checks of codestream::read / tile::parse_tile_header)"""
    cs, sots, clean = stream
    at = sots[2]
    for off, value in ((3, 9), (4, 0xFF), (5, 0xFF), (5, 1), (10, 7), (11, 1), (6, 0xFF), (9, 0), (7, 0x40)):
        b = bytearray(cs); b[at + off] = value
        _both(refgen, bytes(b))


def test_main_header_garbage_and_unsupported_segments(refgen, stream):
    """bytes in front of SOC, an unknown marker segment and RGN / POC / PPM segments in the main header: passed over or skipped
    with a warning by read_headers (ojph_codestream_local.cpp:768-881)"""
    cs, sots, clean = stream
    qcd = cs.find(b"\xff\x5c")
    for extra in (b"\xff\x5e\x00\x05\x00\x00\x03", b"\xff\x5f\x00\x09\x00\x00\x00\x01\x01\x01\x00", b"\xff\x60\x00\x04\x00\x00", b"\xff\x30", b"\x00\x11\x22"):
        strict, resilient = _both(refgen, cs[:qcd] + extra + cs[qcd:])
        assert np.array_equal(np.asarray(strict), clean)
    strict, _ = _both(refgen, b"\x00\x00\x00\x0cjP  \r\n\x87\n" + cs)
    assert np.array_equal(np.asarray(strict), clean)


def test_cap_and_quantisation_segments_are_checked_like_the_reference(refgen, stream):
    cs, sots, clean = stream
    cap, qcd, cod = cs.find(b"\xff\x50"), cs.find(b"\xff\x5c"), cs.find(b"\xff\x52")
    for at, value in ((cap + 5, 0x03), (cap + 5, 0x00), (cap + 3, 10), (qcd + 3, 0xFF), (qcd + 4, 0x41), (qcd + 4, 0x43), (cod + 10, 9), (cod + 12, 0x41),
                      (cod + 8, 0x90), (cod + 9, 0x21), (cod + 13, 0)):
        b = bytearray(cs); b[at] = value
        _both(refgen, bytes(b))


def _with_sop_eph(cs, sop, eph):
    """a single-tile, single-component LRCP codestream with one precinct per resolution, rewritten with SOP marker segments in
    front of its packets and / or EPH markers behind its packet headers (the reference's encoder writes neither)"""
    pl = parse_codestream(cs)
    coded = pl.coded_blocks()
    by_res = {}
    for k in range(pl.num_blocks):
        if coded[k]["len1"]:
            res = int(pl.bands[int(pl.blocks[k]["band"])]["res"])
            o = int(coded[k]["offset"]); n = int(coded[k]["len1"]) + int(coded[k]["len2"])
            lo, hi = by_res.get(res, (1 << 60, 0))
            by_res[res] = (min(lo, o), max(hi, o + n))
    sot = cs.find(b"\xff\x90\x00\x0a")
    out = bytearray(cs[:sot + 14])
    pos = sot + 14
    for seq, res in enumerate(sorted(by_res)):
        body, end = by_res[res]
        if sop:
            out += b"\xff\x91\x00\x04" + seq.to_bytes(2, "big")
        out += cs[pos:body]                                   # the packet header
        if eph:
            out += b"\xff\x92"
        out += cs[body:end]
        pos = end
    assert cs[pos:pos + 2] == b"\xff\xd9"
    out += cs[pos:]
    out[sot + 6:sot + 10] = (len(out) - 2 - sot).to_bytes(4, "big")       # Psot
    cod = bytes(out).find(b"\xff\x52")
    out[cod + 4] |= (2 if sop else 0) | (4 if eph else 0)                   # Scod
    return bytes(out)


def test_sop_and_eph_markers(refgen):
    """bb_skip_sop / bb_skip_eph (ojph_bitbuffer_read.h:153-221): read when the COD announces them; an SOP of the wrong length or
    a missing EPH is a throw"""
    img = synth_image(1, 64, 64, 8, seed=11)
    cs = refgen.encode(img, 8, num_decomps=2, block=(16, 16), prog_order="LRCP")
    clean = np.asarray(refgen.decode(cs)[0])
    for sop, eph in ((True, True), (True, False), (False, True)):
        part = _with_sop_eph(cs, sop, eph)
        strict, resilient = _both(refgen, part)
        assert np.array_equal(np.asarray(strict), clean), (sop, eph)
        if sop:                                                 # an SOP length of 5, an SOP that is not there
            at = part.find(b"\xff\x91\x00\x04", part.find(b"\xff\x93") + 8)
            b = bytearray(part); b[at + 3] = 5
            _both(refgen, bytes(b))
            b = bytearray(part); b[at + 1] = 0x55
            _both(refgen, bytes(b))
        if eph:                                                 # an EPH overwritten
            at = part.find(b"\xff\x92", part.find(b"\xff\x93") + 2)
            b = bytearray(part); b[at + 1] = 0x00
            strict, resilient = _both(refgen, bytes(b))
            assert strict is None


def test_psot_zero_on_the_last_tile_part(refgen, stream):
    """Psot = 0 ("until the EOC") is legal for the last tile-part: param_sot::get_payload_length answers 0, data_left wraps to nearly
    2^32, the packets are limited by the file alone, and the final seek goes BACK to the start of the tile-part, from where the
    EOC is searched for (ojph_tile.cpp:790-796, :934)"""
    cs, sots, clean = stream
    b = bytearray(cs); b[sots[3] + 6:sots[3] + 10] = b"\x00\x00\x00\x00"
    strict, resilient = _both(refgen, bytes(b))
    assert np.array_equal(np.asarray(strict), clean)
    single = refgen.encode(synth_image(1, 40, 56, 8, seed=3), 8, num_decomps=2)
    at = single.find(b"\xff\x90\x00\x0a")
    b = bytearray(single); b[at + 6:at + 10] = b"\x00\x00\x00\x00"
    strict, resilient = _both(refgen, bytes(b))
    assert np.array_equal(np.asarray(strict), np.asarray(refgen.decode(single)[0]))


def test_padded_blocks_are_listed(refgen, stream):
    """ojphgpu_plan_padded_blocks: one byte short of its cleanup segment with a plausible Scup, the last block is decoded by the
    reference from its bytes plus a zero -- listed with what the packet header said, not coded in ojphgpu_plan_coded_blocks"""
    cs, sots, clean = stream
    b = bytearray(cs)
    psot = int.from_bytes(cs[sots[3] + 6:sots[3] + 10], "big") - 1
    b[sots[3] + 6:sots[3] + 10] = psot.to_bytes(4, "big")
    pl = parse_codestream(bytes(b), resilient=True)
    listed = pl.padded_blocks()
    assert len(listed) == 1
    q = listed[0]
    whole = parse_codestream(cs).coded_blocks()[int(q["block"])]
    assert int(q["got"]) == int(q["len1"]) + int(q["len2"]) - 1 and int(q["offset"]) == int(whole["offset"]) and int(q["len1"]) == int(whole["len1"])
    assert int(pl.coded_blocks()[int(q["block"])]["len1"]) == 0
    assert len(parse_codestream(cs).padded_blocks()) == 0


def test_marker_segments_in_tile_part_headers(refgen, stream):
    """PLT / COM segments in a tile-part header are skipped; PPT / POC / COD / QCD there are "not supported yet" -- a warning, and
    skipped all the same (ojph_codestream_local.cpp:952-1093); with Psot adjusted the picture is the clean one, without it the
    tile-part is that many bytes short"""
    cs, sots, clean = stream
    for seg in (b"\xff\x58\x00\x05\x00\x01\x02", b"\xff\x64\x00\x06\x00\x01hi", b"\xff\x61\x00\x04\x00\x07", b"\xff\x5f\x00\x09\x00\x00\x00\x01\x01\x01\x00",
                b"\xff\x5c\x00\x05\x40\x40\x48"):
        for which in (0, 2):
            at = sots[which]
            first_only = seg[1] == 0x5C and which != 0             # (a QCD is looked for in a tile's FIRST tile-part only)
            b = bytearray(cs[:at + 12] + seg + cs[at + 12:])
            psot = int.from_bytes(cs[at + 6:at + 10], "big") + len(seg)
            b[at + 6:at + 10] = psot.to_bytes(4, "big")
            strict, resilient = _both(refgen, bytes(b))
            if not first_only:
                assert np.array_equal(np.asarray(strict), clean), (seg[:2].hex(), which)
            _both(refgen, cs[:at + 12] + seg + cs[at + 12:])           # Psot left as it was


def test_damaged_codestreams_against_the_committed_reference_verdicts():
    """tests/golden/damaged.json (made by tests/golden/make_damaged.py with the live reference): 240 damaged codestreams, rebuilt here
    from the oracle pipeline's own encoder, each read with and without resilience -- the reference's "raises" or the digest of
    its picture.  Holds where /root/reference and oracle/_ref do not exist.  No deviations: the two cases that used to be refused
    (a SIZ byte gives the colour-transformed components different sample formats; the reference's reader converts component by
    component, ojph_tile.cpp:439-518) are read since round 5."""
    import json, os
    from tests.damaged_cases import cases, digest
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "damaged.json")))["cases"]
    checked = 0
    for name, part in cases():
        for resilient in (False, True):
            key = "%s_%d" % (name, int(resilient))
            if key not in gold:
                continue
            try:
                pl = parse_codestream(part, resilient=resilient)
                got = digest(cp.inverse_stages(pl, cp.decode_blocks(pl, part, resilient=resilient)))
            except (capi.OjphError, RuntimeError):
                got = "raises"
            checked += 1
            assert got == gold[key], "%s: reference %s, here %s" % (key, gold[key][:12], got[:12])
    assert checked == 480


def test_restricted_reading_follows_the_reference_order(refgen, stream):
    """ojphgpu_t2_parse_restricted = read_headers, restrict_input_resolution, read: on a whole codestream the same as restricting
    the parsed plan; on cut and damaged ones the reference's restricted reading (LRCP stops at the highest resolution wanted, the
    bytes of unwanted packets are stepped over) -- same verdict, same reduced picture"""
    cs, sots, clean = stream
    whole = parse_codestream(cs); whole.restrict_resolution(1, 1)
    early = parse_codestream(cs, skip=(1, 1))
    a = cp.inverse_stages(whole, cp.decode_blocks(whole, cs)); b = cp.inverse_stages(early, cp.decode_blocks(early, cs))
    assert np.array_equal(np.asarray(a), np.asarray(b)) and np.array_equal(np.asarray(a), np.asarray(refgen.decode(cs, skip=(1, 1))[0]))
    parts = [cs[:k] for k in (sots[1] + 5, sots[2] + 13, sots[3] + 40, len(cs) - 30, len(cs) - 3)]
    dmg = bytearray(cs); dmg[sots[3] + 14] = 0xFF; parts.append(bytes(dmg))          # the unwanted resolution's packet header
    dmg = bytearray(cs); dmg[sots[1] + 14] = 0xF0; parts.append(bytes(dmg))
    for part in parts:
        for skip in ((1, 1), (2, 1), (3, 3)):
            for resilient in (False, True):
                try:
                    want = np.asarray(refgen.decode(part, resilient=resilient, skip=skip)[0])
                except RuntimeError:
                    want = None
                try:
                    pl = parse_codestream(part, resilient=resilient, skip=skip)
                    got = np.asarray(cp.inverse_stages(pl, cp.decode_blocks(pl, part, resilient=resilient)))
                except (capi.OjphError, RuntimeError):
                    got = None
                assert (want is None) == (got is None), (len(part), skip, resilient)
                assert want is None or np.array_equal(got, want), (len(part), skip, resilient)


def test_back_to_back_sot_segments_are_read_in_linear_time():
    """A crafted file: tens of thousands of SOT segments with Psot = 12 (a tile-part that ends where its payload begins, so
    the position moves BACK behind every search for an SOD) in front of one SOD far away.  The reference's reader -- and this
    parser when it followed it to the letter -- takes quadratic time on it (a 10 MB file holds a host thread
    for hours).  The searches are bounded now: strict reading refuses the file, resilient reading stops where the budget ends."""
    import time
    img = synth_image(1, 32, 32, 8, seed=1)
    cs = bytes(cp.encode(img, bit_depth=8, num_decomps=1)[0])
    sot = cs.find(b"\xff\x90\x00\x0a")
    head, body = cs[:sot], cs[sot + 14:]                       # (main header; the tile-part behind its SOT segment and SOD)
    n = 150000
    seg = b"\xff\x90\x00\x0a\x00\x00" + (12).to_bytes(4, "big") + b"\x00\x01"
    crafted = head + seg * n + b"\x00" * 200000 + b"\xff\x93" + body
    t0 = time.time()
    with pytest.raises(capi.OjphError):
        parse_codestream(crafted, resilient=False)
    pl = parse_codestream(crafted, resilient=True)
    assert time.time() - t0 < 20.0                             # (unbounded: n searches over ~1 MB each)
    assert pl.num_blocks > 0
