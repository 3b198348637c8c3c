#!/usr/bin/env python3
"""The codestreams a fuzzer run has dumped (FUZZ_DUMP=dir python tools/fuzz_flip_cpu.py .. header) sorted by WHY this library's
parser refused them (OJPHGPU_T2_DEBUG: the line of ojph_t2.cpp or the plan's message) beside what the live reference said.
python tools/classify_dump.py dir [how many]"""
import os, sys, subprocess, collections, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = r'''
import sys
sys.path.insert(0, %r)
from oracle import refbind
r = refbind.Ref(generic=True)
try:
    r.decode(open(sys.argv[1], "rb").read(), resilient=False, max_samples=1 << 22)
    print("REFRESULT decodes")
except Exception as e:
    print("REFRESULT raises")
''' % ROOT
OURS = r'''
import sys
sys.path.insert(0, %r)
from openjph_amd.plan import parse_codestream
from tests import cpu_pipeline as cp
part = open(sys.argv[1], "rb").read()
try:
    pl = parse_codestream(part, resilient=False)
    cp.inverse_stages(pl, cp.decode_blocks(pl, part, resilient=False))
    print("OURRESULT decodes")
except Exception as e:
    print("OURRESULT raises", str(e)[-50:])
''' % ROOT
cls = collections.Counter(); ex = {}
for fn in sorted(glob.glob(os.path.join(sys.argv[1], "*_0.j2c")))[:int(sys.argv[2]) if len(sys.argv) > 2 else 1000]:
    try:
        o = subprocess.run([sys.executable, "-c", REF, fn], capture_output=True, text=True, env=dict(os.environ, REF_SHIM_VERBOSE="1"), timeout=20)
        msgs = [l for l in (o.stdout + o.stderr).splitlines() if l.startswith("ojph error")]
        rr = [l for l in o.stdout.splitlines() if l.startswith("REFRESULT")]
        refm = "reference: " + (msgs[-1].split(": ", 1)[-1][:60] if msgs else (rr[0][10:] if rr else "crashes"))
    except subprocess.TimeoutExpired:
        refm = "reference: spins"
    o2 = subprocess.run([sys.executable, "-c", OURS, fn], capture_output=True, text=True, timeout=120, env=dict(os.environ, OJPHGPU_T2_DEBUG="1"))
    m = [l[18:110] for l in o2.stderr.splitlines() if l.startswith("ojphgpu_t2_parse")]
    ourm = [l[10:] for l in o2.stdout.splitlines() if l.startswith("OURRESULT")]
    key = refm + "  ||  here: " + (m[-1] if m else (ourm[0] if ourm else "crashes"))
    cls[key] += 1; ex.setdefault(key, os.path.basename(fn))
for k, v in cls.most_common():
    print("%5d  %s     (e.g. %s)" % (v, k, ex[k]))
