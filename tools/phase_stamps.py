#!/usr/bin/env python3
"""Diagnostic: where does a window's time go?  Needs `make stamps` (libbiopileup_stamps.so,
in-kernel s_memtime stamps) and a GPU.  Prints mean cycles per phase over all windows."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bio_amd import _abi, snp, synth  # noqa: E402

_abi.LIB_PATH = os.environ.get("STAMPS_LIB") or os.path.join(ROOT, "bio_amd", "libbiopileup_stamps.so")
L = int(float(sys.argv[1])) if len(sys.argv) > 1 else 40_000_000
STITCH = bool(os.environ.get("STITCH"))
MINQ = int(os.environ.get("MINQ", "0"))
GEN = dict(frac_softclip=0, frac_del=0, frac_ins=0, frac_hardclip=0) if os.environ.get("SIMPLE") else {}
sb = synth.generate([(0, 0, L - 1000, 30.0)], with_names=STITCH, **GEN)
rb = sb.to_record_batch()
job = snp.PileupJob(snp.device_opts(snp.Opts(Stitch=STITCH, MinBaseQual=MINQ), "basestrand-tsv"), [L])
job.set_bed({0: [[0, L]]})
job.submit(rb)
job.finish()
job.compute()
tm = job.timing()
nw = int(tm.n_windows)
lib = _abi.load()
lib.bp_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
st = np.zeros((nw, 8, 8), np.uint64)
assert lib.bp_debug_stamps(job.h, st.ctypes.data_as(C.c_void_p), st.size) == 0
st = st.astype(np.int64)
names = ["zero+barrier", "phase1", "barrier1", "phase2", "barrier2", "flush"]
print("kernel %.3f ms, %d windows" % (tm.pileup_ms, nw))
w0 = st[:, 0, :]
tot = (w0[:, 6] - w0[:, 0])
print("wave0 total cycles/window: mean %.0f  p50 %.0f  p95 %.0f" % (tot.mean(), np.median(tot), np.percentile(tot, 95)))
for i, n in enumerate(names):
    d = st[:, :, i + 1] - st[:, :, i]
    print("  %-14s mean %.0f  (all waves)   wave0 mean %.0f  max-wave mean %.0f" % (
        n, d.mean(), d[:, 0].mean(), d.max(axis=1).mean()))
if os.environ.get("P2"):
    p2 = np.zeros((nw, 8, 4), np.uint64)
    buf = np.zeros(nw * 64 + nw * 32, np.uint64)
    assert lib.bp_debug_stamps(job.h, buf.ctypes.data_as(C.c_void_p), buf.size) == 0
    p2 = buf[nw * 64:].reshape(nw, 8, 4).astype(np.int64)
    n = np.maximum(p2[:, :, 3], 1)
    print("phase-2 per unit (serialised, depth 1): issue(LDS) %.0f  wait(vmem) %.0f  process %.0f cycles; units/thread %.1f" % (
        (p2[:, :, 0] / n).mean(), (p2[:, :, 1] / n).mean(), (p2[:, :, 2] / n).mean(), p2[:, :, 3].mean()))
span = st[:, :, 6].max() - st[:, :, 0].min()
print("first start -> last end: %d ticks; sum(window totals)/span = %.1f concurrent windows" % (span, tot.sum() / span))
