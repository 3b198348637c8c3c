#!/usr/bin/env python3
"""Timeline of ONE fused block-decoder launch on the 8K bench frame: when the chain wavefronts end, when the workers finish
each slice, how long they wait, and whether sharing a CU with a step-1 workgroup matters.
    python tools/build_variant.py tl kernels_ht_dec.hip -DFUSED_TIMELINE && python tools/fused_timeline.py [workload]"""
import ctypes as C, os, shutil, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "openjph_amd", "libojphgpu.so")
shutil.copy(LIB, "/tmp/lib_tl_orig.so")
shutil.copy(os.path.join(ROOT, "openjph_amd", "variants", "lib_%s.so" % os.environ.get("TL_VARIANT", "tl")), LIB)
try:
    from bench import workload_image, WORKLOADS
    from openjph_amd import codec
    from openjph_amd.plan import Plan, make_params
    name = sys.argv[1] if len(sys.argv) > 1 else "c3_8k_444_12b_irv97"
    w, h, nc, bd, rev, ct, qstep, tile = WORKLOADS[name]
    img = workload_image(name)
    d = torch.from_numpy(img.astype(np.int16 if bd <= 15 else np.int32)).cuda()
    enc = codec.Encoder(plan=Plan(make_params(w, h, nc, bit_depth=bd, reversible=rev, qstep=qstep)))
    cs = enc.encode(d)
    dec = codec.Decoder(cs)
    out = torch.empty_like(d)
    L = C.CDLL(LIB)
    for _ in range(3):
        dec.run_device(out)
    torch.cuda.synchronize()
    buf = (C.c_uint32 * (8192 * 12 + 8192))()
    assert L.ojphgpu_debug_fused_timeline(None, 0, 1) == 0
    dec.run_device(out); torch.cuda.synchronize()
    assert L.ojphgpu_debug_fused_timeline(buf, 8192 * 12, 0) == 0
    pub = np.frombuffer(buf, dtype=np.uint32)[8192 * 12:].reshape(1024, 8).astype(np.int64)
    a = np.frombuffer(buf, dtype=np.uint32)[:8192 * 12].reshape(8192, 12).astype(np.int64)
    a = np.concatenate([a, np.arange(8192).reshape(-1, 1)], axis=1)          # column 12: the wavefront's slot = role number * 12 + wavefront
    a = a[a[:, 0] != 0]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.save(os.path.join(ROOT, "gpurun_out", "timeline_%s.npy" % name), a); np.save(os.path.join(ROOT, "gpurun_out", "timeline_pub_%s.npy" % name), pub)
    t0 = a[:, 2].min()
    us = lambda t: (t - t0) / 100.0
    chains, workers, partners = a[a[:, 0] == 1], a[a[:, 0] == 2], a[a[:, 0] == 3]
    cu_of = lambda r: (r[:, 1] >> 16) * 256 + ((r[:, 1] >> 8) & 0xFF)
    print("%s: %d chain, %d partner, %d worker wavefronts; launch from first start to last end %.1f us" %
          (name, len(chains), len(partners), len(workers), us(a[:, 3].max())))
    q = lambda v: "min %.1f  median %.1f  p90 %.1f  max %.1f" % (np.min(v), np.median(v), np.percentile(v, 90), np.max(v))
    print("chains   start: " + q(us(chains[:, 2])))
    print("chains   end:   " + q(us(chains[:, 3])))
    print("chains   length:" + q((chains[:, 3] - chains[:, 2]) / 100.0))
    order = np.argsort(-chains[:, 3])[:12]
    print("slowest chains (step-1 workgroup, wavefront, xcc, cu key, end us): " +
          ", ".join("(%d,%d,%d,%02x,%.0f)" % (r[12] // 12, r[12] % 12, r[1] >> 16, (r[1] >> 8) & 0xFF, us(r[3])) for r in chains[order]))
    print("chain end histogram (10 us bins from 200): " + str(np.histogram(us(chains[:, 3]), bins=[0, 200, 210, 220, 230, 240, 250, 260, 270, 280, 300, 400])[0].tolist()))
    pub = pub[pub[:, 0] != 0]
    for k in range(3):
        print("chains   rows %2d published: %s   (waited for their stores: median %.2f max %.2f us)" % (8 * k + 8, q(us(pub[:, k])), np.median(pub[:, k] - pub[:, 4 + k]) / 100.0, np.max(pub[:, k] - pub[:, 4 + k]) / 100.0))
    slow = np.argsort(-pub[:, 7])[:6]
    print("the six chain wavefronts that ended last, their publications (us): " + "; ".join(" ".join("%.0f" % us(pub[i, k]) for k in (0, 1, 2, 7)) for i in slow))
    # the step-1 workgroups that ended last against the others: SIMDs of their chain wavefronts, who shares their CU
    wg_end = {}
    for r in chains: wg_end.setdefault(int(r[12]) // 12, []).append(r)
    ends = sorted(wg_end.items(), key=lambda kv: -max(x[3] for x in kv[1]))
    everyone = a
    def describe(wg, rows):
        key = cu_of(np.array(rows))[0]
        mates = everyone[cu_of(everyone) == key]
        simd = lambda rr: "".join(str(int((x[1] >> 4) & 3)) for x in rr)
        part = [x for x in partners if x[12] // 12 == wg]
        return "wg %d cu %d/%02x end %.0f: chain simds %s partner simds %s; wavefronts on the CU: %d chain %d partner %d worker (worker simds %s, worker start %.0f)" % (
            wg, key >> 8, key & 0xFF, us(max(x[3] for x in rows)), simd(rows), simd(part), int((mates[:, 0] == 1).sum()), int((mates[:, 0] == 3).sum()),
            int((mates[:, 0] == 2).sum()), simd(mates[mates[:, 0] == 2]), us(np.min(mates[mates[:, 0] == 2][:, 2])) if (mates[:, 0] == 2).any() else -1)
    stat = {}
    for wg, rows in ends:
        key = cu_of(np.array(rows))[0]
        mates = everyone[(cu_of(everyone) == key) & (everyone[:, 0] == 2)]
        rows_sorted = sorted(rows, key=lambda x: x[12])
        k2 = (int((rows_sorted[0][1] >> 4) & 3), int((mates[np.argmin(mates[:, 12])][1] >> 4) & 3) if len(mates) else -1, int(key >> 8) )
        stat.setdefault(k2[:2], []).append(us(max(x[3] for x in rows)))
    print("end of a step-1 workgroup by (SIMD of its first chain wavefront, SIMD of the first wavefront of the worker workgroup on its CU): " +
          "; ".join("%s n=%d median %.0f max %.0f slow(>280) %d" % (k, len(v), np.median(v), np.max(v), sum(1 for x in v if x > 280)) for k, v in sorted(stat.items())))
    for wg, rows in ends[:5]: print("LAST  " + describe(wg, rows))
    for wg, rows in ends[40:43]: print("USUAL " + describe(wg, rows))
    ccu = cu_of(chains)
    per_cu = np.unique(ccu, return_counts=True)[1]
    print("chain wavefronts per CU that has any: " + str(dict(zip(*np.unique(per_cu, return_counts=True)))))
    shared = np.isin(cu_of(workers), ccu)
    print("workers  start: " + q(us(workers[:, 2])))
    for lab, sel in (("on a CU with chains", shared), ("on a CU without", ~shared)):
        ww = workers[sel]
        if len(ww) == 0: continue
        print("workers %s (%d): end %s" % (lab, len(ww), q(us(ww[:, 3]))))
        print("    waited for chains: " + q(ww[:, 11] / 100.0) + " us")
        for sl in range(7):
            if ww[:, 4 + sl].max() > 0: print("    slice %d done: %s" % (sl, q(us(ww[:, 4 + sl]))))
    n1 = len(chains) // 4
    wno = workers[:, 12] - n1 * 12
    order = np.argsort(wno)
    ws = workers[order]
    print("workers by wavefront number (= position of their 5 blocks in the block order), deciles: end us / busy us (end - start - waits)")
    for dct in np.array_split(np.arange(len(ws)), 10):
        r = ws[dct]
        print("   waves %5d..%5d: end median %.0f max %.0f   busy median %.0f max %.0f" % (wno[order][dct[0]], wno[order][dct[-1]], np.median(us(r[:, 3])), np.max(us(r[:, 3])),
              np.median((r[:, 3] - r[:, 2] - r[:, 11]) / 100.0), np.max((r[:, 3] - r[:, 2] - r[:, 11]) / 100.0)))
    busy = (workers[:, 3] - workers[:, 2] - workers[:, 11]) / 100.0
    xcc = workers[:, 1] >> 16
    print("workers by XCD: " + "; ".join("xcc %d n=%d ticket median %d busy median %.0f end median %.0f" % (x, (xcc == x).sum(), np.median(wno[xcc == x]), np.median(busy[xcc == x]), np.median(us(workers[xcc == x][:, 3]))) for x in range(8)))
    se = (workers[:, 1] >> 13) & 7
    print("workers by shader engine: " + "; ".join("se %d n=%d busy median %.0f" % (x, (se == x).sum(), np.median(busy[se == x])) for x in range(8) if (se == x).any()))
    sys.stdout.flush(); os._exit(0)
finally:
    shutil.copy("/tmp/lib_tl_orig.so", LIB)
