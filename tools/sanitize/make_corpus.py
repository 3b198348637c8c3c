import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests import cpu_pipeline as cp
from tests.random_cases import random_case
from tests.part2_cases import CASES, split, image
from oracle import refbind
out = sys.argv[1]; os.makedirs(out, exist_ok=True)
r = refbind.Ref(generic=True)
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 77)
n = 0
srcs = []
for seed in range(900000 + 1000 * (int(sys.argv[3]) if len(sys.argv) > 3 else 0), 900000 + 1000 * (int(sys.argv[3]) if len(sys.argv) > 3 else 0) + int(sys.argv[2])):
    planes, kw, size = random_case(seed)
    if any(q.size == 0 for q in planes) or sum(q.size for q in planes) > 20000: continue
    k2 = dict(kw); bd, sg = k2.pop("bit_depth"), k2.pop("is_signed")
    try: srcs.append(r.encode(planes, bd, is_signed=sg, size=size, **k2))
    except RuntimeError: pass
for i in range(0, len(CASES), 2):
    nc, h, w, bd, kw = split(CASES[i])
    if bd > 16: continue
    srcs.append(bytes(cp.encode(image(nc, h, w, bd), **kw)[0]))
for cs in srcs:
    sot = cs.find(b"\xff\x90\x00\x0a")
    for t in range(40):
        b = bytearray(cs)
        mode = t % 4
        if mode == 0:
            for _ in range(int(rng.integers(1, 4))): b[int(rng.integers(0, sot + 2))] = int(rng.choice([0xFF, 0, 1, 0x52, 0x90, int(rng.integers(0, 256))]))
        elif mode == 1:
            for _ in range(int(rng.integers(1, 4))): b[int(rng.integers(sot, len(b)))] = int(rng.choice([0xFF, 0, 0x90, 0x7F, int(rng.integers(0, 256))]))
        elif mode == 2:
            b = b[:int(rng.integers(2, len(b)))]
        else:
            for _ in range(int(rng.integers(4, 12))): b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        open(os.path.join(out, "%06d.j2c" % n), "wb").write(bytes(b)); n += 1
print(n, "files from", len(srcs), "sources")
