#!/usr/bin/env python3
"""Device BGZF inflate rate on a synthetic 30x BAM: kernel time per batch of blocks (HIP events
inside bp_inflate_blocks) and wall time with/without the copy back to the host.
usage: inflate_bench.py [megabases=60] [batch_blocks=4096,...]"""
import os
import sys
import tempfile
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bio_amd import bgzf, host, synth  # noqa: E402

mb = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
batches = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1024,4096,16384").split(",")]
L = int(mb * 1e6)
sb = synth.generate([(0, 0, L - 1000, 30.0)], with_names=True, threads=16)
rb = sb.to_record_batch()
d = tempfile.mkdtemp(prefix="infl_", dir=os.environ.get("E2E_TMP", "/tmp"))
p = os.path.join(d, "x.bam")
host.bam_write(p, ["chr1"], [L], rb, level=1, threads=16)
data = np.fromfile(p, np.uint8)
os.remove(p)
os.rmdir(d)
blocks, end, total = bgzf.scan_blocks(data)
print(f"{len(data) / 1e9:.2f} GB BAM, {len(blocks)} blocks, {total / 1e9:.2f} GB inflated", flush=True)
t0 = time.perf_counter()
k = min(len(blocks), 2000)
for r in blocks[:k]:
    zlib.decompress(data[r["in_off"]:r["in_off"] + r["in_len"]].tobytes(), -15)
zs = time.perf_counter() - t0
print(f"zlib, one thread: {blocks['out_len'][:k].sum() / zs / 1e9:.3f} GB/s of output", flush=True)
with bgzf.Inflater() as inf:
    for nb in batches:
        for to_host in (False, True):
            kms = cms = 0.0
            t0 = time.perf_counter()
            for i in range(0, len(blocks), nb):
                b = blocks[i:i + nb].copy()
                lo, hi = int(b["in_off"][0]), int(b["in_off"][-1] + b["in_len"][-1])
                b["in_off"] -= lo
                o0 = int(b["out_off"][0])
                b["out_off"] -= o0
                inf.inflate(data[lo:hi], b, to_host=to_host)
                a, c = inf.timing()
                kms += a
                cms += c
            wall = time.perf_counter() - t0
            print(f"batch {nb:6d} blocks, to_host={to_host}: inflate kernel {kms:.1f} ms = {total / kms / 1e6:.1f} GB/s, "
                  f"crc kernel {cms:.1f} ms = {total / cms / 1e6:.1f} GB/s, wall {wall:.2f} s = {total / wall / 1e9:.1f} GB/s", flush=True)

    # The stages behind the inflate, one at a time on one stream (so that a kernel trace of this
    # script shows each kernel running alone): the whole file inflated in one call, records found and
    # their fields read on the device (bp_bam_scan), packed in place (bp_submit_bam_device).
    import struct
    from bio_amd import snp
    from bio_amd.snp import Opts
    nb = min(len(blocks), 16384)
    b = blocks[:nb].copy()
    part = data[: int(b["in_off"][-1] + b["in_len"][-1])]
    n_out = int(b["out_off"][-1] + b["out_len"][-1])
    head = bytearray()
    for r in b[:4]:
        head += zlib.decompress(data[r["in_off"]:r["in_off"] + r["in_len"]].tobytes(), -15)
    l_text = struct.unpack_from("<I", head, 4)[0]
    q = 12 + l_text
    for _ in range(struct.unpack_from("<I", head, 8 + l_text)[0]):
        q += 8 + struct.unpack_from("<I", head, q)[0]
    inf.inflate(part, b, to_host=False)
    t0 = time.perf_counter()
    rc, cols, ex, stopped, cstruct = inf.scan(q, n_out, 1)
    scan_s = time.perf_counter() - t0
    n_rec = len(cols["off"])
    print(f"scan: rc {rc}, {n_rec} records of {ex - q} stream bytes in {scan_s * 1e3:.1f} ms wall (kernels + 13 column copies)", flush=True)
    for mq in (0, 20):
        with snp.PileupJob(snp.device_opts(Opts(MinBaseQual=mq)), [L]) as job:
            job.set_bed({0: [[0, L]]})
            t0 = time.perf_counter()
            job.submit_bam_device(inf.device_output, n_out, cstruct, 0, n_rec)
            sub_s = time.perf_counter() - t0
            job.finish()
            print(f"submit_bam_device (min_base_qual {mq}): {n_rec} records, {ex - q} bytes in {sub_s * 1e3:.1f} ms wall "
                  f"(host scan of the columns + pack kernel); depth sum {job.column_sums()[0]}", flush=True)
