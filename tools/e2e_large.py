#!/usr/bin/env python3
"""End-to-end run at whole-contig scale: synthetic 30x reads over L Mb -> BAM (native writer) +
FASTA -> bio-pileup -> files; prints the driver's BIO_PILEUP_STATS line and wall times.
Secondary measurement, not a bench line.  Usage: e2e_large.py [Mb=60] [format=basestrand-tsv]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bio_amd import bamio, host, synth  # noqa: E402

mb = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
fmt = sys.argv[2] if len(sys.argv) > 2 else "basestrand-tsv"
extra = sys.argv[3:]
L = int(mb * 1e6)
d = tempfile.mkdtemp(dir=os.environ.get("E2E_TMP", "/tmp"))
t0 = time.time()
sb = synth.generate([(0, 0, L - 1000, 30.0)], with_names=True, threads=16)
rb = sb.to_record_batch()
bam = os.path.join(d, "in.bam")
host.bam_write(bam, ["chr1"], [L], rb, level=1, threads=16)
rng = np.random.default_rng(1)
fa = os.path.join(d, "ref.fa")
seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)]
with open(fa, "wb") as f:
    f.write(b">chr1\n")
    lines = seq[: L // 60 * 60].reshape(-1, 60)
    out = np.empty((lines.shape[0], 61), np.uint8)
    out[:, :60] = lines
    out[:, 60] = 10
    f.write(out.tobytes())
    if L % 60:
        f.write(seq[L // 60 * 60:].tobytes() + b"\n")
bed = os.path.join(d, "all.bed")
bamio.write_bed(bed, [(0, 0, L)], ["chr1"])
print(f"inputs: {rb.n} records, BAM {os.path.getsize(bam) / 1e9:.2f} GB, FASTA {L / 1e6:.0f} Mb "
      f"(made in {time.time() - t0:.0f} s)", flush=True)
n = rb.n
sb.close()
del rb
exe = os.path.join(ROOT, "bio_amd", "bio-pileup")
import hashlib
rc_all, sums = 0, {}
# E2E_MODES: "device" (the driver's default: BGZF inflate and record scan on the GPU), "scan", "host", "rocprof"
for mode in os.environ.get("E2E_MODES", "device,host").split(","):
    env = dict(os.environ, BIO_PILEUP_STATS="1", BP_PROFILE="1")
    if "@" in mode:  # "device@2": that many pwrite threads per piece of text
        mode, wt = mode.split("@")
        env["BIO_PILEUP_WRITE_THREADS"] = wt
    if mode == "host":
        env["BIO_PILEUP_HOST_INFLATE"] = "1"
    if mode == "scan":  # device inflate, records found by the host's walk
        env["BIO_PILEUP_HOST_SCAN"] = "1"
    cmd = [exe, "-bed", bed, "-format", fmt, *extra, "-out", os.path.join(d, "o"), bam, fa]
    if mode == "rocprof":  # the default mode under rocprofv3's kernel trace (E2E_ROCPROF_DIR holds the csv files)
        env.pop("BP_PROFILE")
        cmd = ["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d",
               os.environ.get("E2E_ROCPROF_DIR", os.path.join(ROOT, "gpurun_out", "e2e_rocprof")), "--"] + cmd
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, env=env)
    wall = time.time() - t0
    prof = [ln for ln in r.stderr.splitlines() if ln.startswith("bp profile")]
    if prof:  # BP_PROFILE=1: one line per job handle; their sums
        import re
        cols = np.array([[float(x) for x in re.findall(r" ([0-9.]+)", ln.split("s):")[1].split(";")[0])] for ln in prof])
        print(f"bp profile, sums over {len(prof)} handles (extract layout scan-parallel scan-serial items ensure copy-wait total):",
              cols.sum(axis=0).round(3).tolist(), flush=True)
    print("\n".join(ln for ln in r.stderr.strip().splitlines() if not ln.startswith("bp profile"))[-2600:], flush=True)
    print(f"spawned at {t0:.3f}, returned at {t0 + wall:.3f} (epoch seconds)", flush=True)
    outs = sorted(p for p in os.listdir(d) if p.startswith("o."))
    size = sum(os.path.getsize(os.path.join(d, p)) for p in outs)
    h = hashlib.md5()
    for p in outs:
        with open(os.path.join(d, p), "rb") as f:
            while True:
                blk = f.read(1 << 24)
                if not blk:
                    break
                h.update(blk)
    sums[mode] = h.hexdigest()
    print(f"[{mode} inflate] bio-pileup -bed all -format {fmt} {' '.join(extra)}: rc {r.returncode}, {wall:.2f} s wall, "
          f"{n / wall / 1e6:.2f} M records/s, output {size / 1e9:.2f} GB in {outs}, md5 {sums[mode]}", flush=True)
    rc_all |= r.returncode
    for p in outs:
        os.remove(os.path.join(d, p))
if len(set(sums.values())) > 1:
    print("OUTPUTS DIFFER between inflate modes", sums, flush=True)
    rc_all |= 1
for p in [bam, fa, bed]:
    os.remove(p)
os.rmdir(d)
sys.exit(rc_all)
