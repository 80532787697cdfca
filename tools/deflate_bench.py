#!/usr/bin/env python3
"""The device's BGZF writer by itself: basestrand-tsv-like text of N MB through bp_deflate_bgzf
(host buffer -> device -> members -> host), repeated; prints sizes, the ratio, wall time per call
and, for comparison, zlib levels 1 and 6 on one core over a 16 MB sample.  Run it under
`rocprofv3 --kernel-trace --stats` for the kernels' own durations.  Usage: deflate_bench.py [MB=256] [reps=5]"""
import gzip
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bio_amd import bgzf  # noqa: E402

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rng = np.random.default_rng(1)
n_lines = mb * (1 << 20) // 30
# "chr1\t<pos>\t<REF>\t" + 8 counts of 30x coverage split over strands, most of them 0
pos = 1000000 + np.arange(n_lines)
ref = rng.integers(0, 4, n_lines)
counts = np.zeros((n_lines, 8), np.int64)
depth = rng.poisson(15.0, (n_lines, 2))
counts[np.arange(n_lines), 2 * ref] = depth[:, 0]
counts[np.arange(n_lines), 2 * ref + 1] = depth[:, 1]
err = rng.random(n_lines) < 0.02
counts[err, (2 * ref[err] + 2) % 8] = 1
t0 = time.time()
lines = ["chr1\t%d\t%s\t%s\n" % (p, "ACGT"[r], "\t".join(map(str, c))) for p, r, c in zip(pos[:200000], ref[:200000], counts[:200000])]
unit = "".join(lines).encode()
data = (unit * (mb * (1 << 20) // len(unit) + 1))[: mb << 20]
# (the repeats are 6 MB apart: far outside DEFLATE's 32 KiB window)
print(f"text: {len(data) / 1e6:.0f} MB, {len(unit) / 200000:.1f} bytes a line (made in {time.time() - t0:.1f} s)", flush=True)
arr = np.frombuffer(data, np.uint8)
with bgzf.Inflater() as f:
    out = f.deflate(arr)  # warm-up: buffers, code load
    ts = []
    for _ in range(reps):
        t0 = time.time()
        out = f.deflate(arr)
        ts.append(time.time() - t0)
    t = min(ts)
    print(f"bp_deflate_bgzf: {len(data)} -> {len(out)} bytes ({len(data) / len(out):.2f}x), best of {reps}: {t * 1e3:.1f} ms "
          f"= {len(data) / t / 1e9:.2f} GB/s of text, host buffer to host buffer", flush=True)
    chk = gzip.decompress(bytes(out[: 1 << 26])) if len(out) > (1 << 26) else gzip.decompress(bytes(out))
    # (a cut inside a member would fail: check whole members only when the output is small)
    assert chk == data[: len(chk)], "round trip"
sample = data[: 16 << 20]
for lvl in (1, 6):
    t0 = time.time()
    z = zlib.compress(sample, lvl)
    dt = time.time() - t0
    print(f"zlib -{lvl}, one core, 16 MB sample: {len(sample) / len(z):.2f}x at {len(sample) / dt / 1e6:.0f} MB/s", flush=True)
