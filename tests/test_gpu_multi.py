"""One frame over several GPUs from one process (include/ojphgpu.h section 8, ojphgpu_multi.cpp): contiguous runs of tiles,
a host thread + codec object per device, the tile-parts copied from every device to their place in ONE host buffer.  The
test box has one GPU: the same device is listed two and three times, which runs the same code -- threads, tile runs, prefix
offsets, 2-D copies -- and must give the single encoder's codestream byte for byte (tile independence:
ojph_codestream_local.cpp:113-180, ojph_tile.cpp:584-610)."""
import numpy as np
import pytest

from tests.synth import synth_image

pytestmark = pytest.mark.gpu

CASES = [
    dict(nc=1, h=300, w=500, bd=16, tile=(128, 128)),
    dict(nc=3, h=260, w=390, bd=8, tile=(100, 70), color_transform=True, tlm=True),
    dict(nc=3, h=200, w=300, bd=10, tile=(64, 64), reversible=False, qstep=0.01, tileparts="R"),
    dict(nc=1, h=200, w=200, bd=8),                          # one tile: does not shard, one worker
    dict(nc=2, h=150, w=170, bd=12, tile=(64, 32), downsampling=[(1, 1), (2, 2)], image_offset=(5, 3)),
]


@pytest.mark.parametrize("devices", [(0,), (0, 0), (0, 0, 0)], ids=["1", "2", "3"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "-".join("%s%s" % (k, v) for k, v in c.items()))
def test_multi_device_codec_equals_the_single_device_one(case, devices):
    from openjph_amd import codec
    from openjph_amd.plan import Plan, make_params
    c = dict(case)
    nc, h, w, bd = c.pop("nc"), c.pop("h"), c.pop("w"), c.pop("bd")
    plan = Plan(make_params(w, h, nc, bit_depth=bd, **c))
    if "downsampling" in c:
        planes = [synth_image(1, plan.comp_info(k)["h"], plan.comp_info(k)["w"], bd, seed=40 + k)[0] for k in range(nc)]
        img = plan.pack_frame(planes)
    else:
        img = synth_image(nc, h, w, bd, seed=40)
    want = codec.Encoder(plan=plan).encode(img)
    me = codec.MultiEncoder(plan=plan, devices=devices)
    assert sum(me.tiles_per_worker) == plan.num_tiles and len(me.tiles_per_worker) == min(len(devices), plan.num_tiles)
    for _ in range(2):                                       # the objects are reused frame after frame
        assert me.encode(img) == want
    want_dec = codec.decode(want)
    md = codec.MultiDecoder(want, devices=devices)
    for _ in range(2):
        got = md.decode()
        assert np.array_equal(got.reshape(-1), np.asarray(want_dec).reshape(-1))
    if bd <= 16 and "downsampling" not in c:                # the same frame in 16-bit (8-bit) containers: same bytes, same samples
        dt = (np.uint8 if bd <= 8 else np.uint16)
        assert me.encode(np.asarray(img).astype(dt)) == want
        got = md.decode(dtype=dt)
        assert np.array_equal(got.reshape(-1).astype(np.int64), np.asarray(want_dec).reshape(-1).astype(np.int64))
    if plan.num_tiles > 1 and "downsampling" not in c:      # reduced resolution through the same path
        md = codec.MultiDecoder(want, devices=devices, skip_res=1)
        assert np.array_equal(md.decode().reshape(-1), np.asarray(codec.Decoder(want, skip_res=1).decode()).reshape(-1))


def test_multi_device_rejects_bad_arguments():
    from openjph_amd import capi, codec
    from openjph_amd.plan import Plan, make_params
    plan = Plan(make_params(64, 64, 1, tile=(32, 32)))
    with pytest.raises(capi.OjphError):
        codec.MultiEncoder(plan=plan, devices=())
    with pytest.raises(capi.OjphError):
        codec.MultiDecoder(b"\xff\x4f\xff\x51", devices=(0,))
