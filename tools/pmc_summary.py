#!/usr/bin/env python3
"""Per-kernel FETCH_SIZE / WRITE_SIZE from two rocprofv3 --pmc csv passes.
Usage: pmc_summary.py <dir_fetch> <dir_write> [out.json].  Table values are the raw counter units
(KiB); the optional JSON holds HBM bytes per launch for bench.py's `roofline.traffic`:
(2 * FETCH_SIZE + WRITE_SIZE) * 1024 -- FETCH_SIZE counts 64 B per 128-B request on gfx950
(MI355X_MICROARCH.md, confirmed by the calibration kernel in the same passes)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d):
    acc = defaultdict(float)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name") or row.get("Kernel Name") or ""
            val = float(row.get("Counter_Value") or row.get("Counter Value") or 0)
            did = int(row.get("Dispatch_Id") or row.get("Dispatch Id") or 0)
            acc[(name, did)] += val          # a counter may be split over several rows (per XCC)
    per = defaultdict(list)
    for (name, did), v in sorted(acc.items(), key=lambda kv: kv[0][1]):
        per[name].append(v)
    return per


def main():
    fe, wr = load(sys.argv[1]), load(sys.argv[2])
    names = sorted(set(fe) | set(wr), key=lambda n: -sum(fe.get(n, [0])))
    print("| kernel | launches | FETCH_SIZE avg KiB | WRITE_SIZE avg KiB |")
    print("|---|---|---|---|")
    for n in names:
        f, w = fe.get(n, []), wr.get(n, [])
        short = n if len(n) < 100 else n[:97] + "..."
        print("| `%s` | %d | %.1f | %.1f |" % (short, max(len(f), len(w)), sum(f) / max(len(f), 1), sum(w) / max(len(w), 1)))
    if len(sys.argv) > 3:
        def avg(v):
            return sum(v) / max(len(v), 1)

        def pick(sub, tmpl=None):
            ks = [k for k in names if sub in k and (tmpl is None or tmpl in k)]
            return ks[0] if ks else None

        def clusters(v):
            """launch sizes of one kernel: values grouped where they differ by less than 15 %"""
            out_ = []
            for x in sorted(v):
                if out_ and x <= 1.15 * (sum(out_[-1]) / len(out_[-1])) + 1.0:
                    out_[-1].append(x)
                else:
                    out_.append([x])
            return out_

        def two_sizes(v):
            """a stage that runs as two launches per frame (lower resolutions / top resolution): the two smallest launch
            sizes; the bench's one-launch-over-all-blocks objects (the largest size, when there are three) are left out"""
            c = clusters(v)
            return (c[0], c[1]) if len(c) >= 2 else (c[0], c[0]) if c else ([], [])

        def traffic(k, sel=None):
            if k is None:
                return None
            f, w = fe.get(k, []), wr.get(k, [])
            if sel is not None:
                f = two_sizes(f)[1 if sel == "big" else 0]
                w = two_sizes(w)[1 if sel == "big" else 0]
            return int((2 * avg(f) + avg(w)) * 1024)

        enc = pick("ht_encode_kernel")
        two = enc is not None and len(clusters(fe.get(enc, []))) >= 2
        out = {
            "_note": "HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from the PMC passes summarised next to this "
                     "file; x2 = gfx950 FETCH_SIZE correction (calibrated on an elementwise kernel of known traffic in "
                     "the same passes); multi-launch stages are per frame",
            "ht_dec_prep": traffic(pick("ht_dec_prep")), "ht_dec_step1": traffic(pick("ht_dec_step1_")),
        }
        if pick("ht_dec_fused_kernel"):                      # step 1 + step 2 as one launch
            out["ht_dec_fused(step 1 + step 2)"] = traffic(pick("ht_dec_fused_kernel"))
        s2 = pick("ht_dec_step2")                            # two launches per frame when the decoder overlaps the lower
        if s2 is not None:                                   # synthesis levels with the top resolution's blocks
            if len(clusters(fe.get(s2, []))) >= 2:
                out["ht_dec_step2"] = traffic(s2, "big") + traffic(s2, "small")
            else:
                out["ht_dec_step2"] = traffic(s2)
        if two:
            out["ht_encode[top resolution, side stream]"] = traffic(enc, "big")
            out["ht_encode[lower resolutions]"] = traffic(enc, "small")
        else:
            out["ht_encode"] = traffic(enc)
        # template arguments: <wavelet policy (kernels_dwt.hip: Wv<reversible>), image container bits (0 = arena planes only), planes per wavefront>
        for d in ("forward", "inverse"):
            top = None
            for rev in ("false", "true"):                    # <reversible, container bits, planes per wavefront>
                for bits in ("16", "32", "8"):
                    for nc in ("1", "3"):
                        top = top or pick("dwt_%s_kernel<(anonymous namespace)::Wv<%s>, %s, %s>" % (d, rev, bits, nc))
            low = pick("dwt_%s_kernel<(anonymous namespace)::Wv<false>, 0, 1>" % d) or pick("dwt_%s_kernel<(anonymous namespace)::Wv<true>, 0, 1>" % d)
            if top and low:
                nl = len(fe.get(low, [])) // max(len(fe.get(top, [])), 1)
                out["dwt_%s(all levels)" % d] = traffic(top) + nl * traffic(low)
                out["dwt_%s(level 1)" % d] = traffic(top)
        # keyed by the bench workload (bench.py looks its kernels up under that name); stamped with the digest of the
        # kernel sources the pass ran on (bench.py refuses a file whose stamp is not the digest of the sources it runs)
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from openjph_amd.build import kernel_sources_digest
        json.dump({"_kernels_sha256": kernel_sources_digest(), sys.argv[4] if len(sys.argv) > 4 else "c3_8k_444_12b_irv97": out},
                  open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
