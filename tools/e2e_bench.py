#!/usr/bin/env python3
"""End-to-end timing of BASELINE config 1: a 30x BAM over a 1 Mb region + FASTA -> bio-pileup
-> TSV files (BGZF inflate, BAM decode, packing, H2D, kernels, D2H, text rendering), next to the
CPU oracle's records-in-memory time for the same region.  Secondary measurement, not a bench line."""
import os
import random
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from bio_amd import bamio, synth  # noqa: E402

L = int(float(sys.argv[1])) if len(sys.argv) > 1 else 3_000_000
S, E = 1_000_000, 2_000_000
d = tempfile.mkdtemp()
sb = synth.generate([(0, S - 600, E + 600, 30.0)], with_names=True)
rb = sb.to_record_batch()
t0 = time.time()
bam = os.path.join(d, "c1.bam")
bamio.write_bam(bam, ["chr22"], [L], rb, level=6)
rng = random.Random(1)
fa = os.path.join(d, "ref.fa")
bamio.write_fasta(fa, ["chr22"], ["".join(rng.choice("ACGT") for _ in range(L))])
bed = os.path.join(d, "r.bed")
bamio.write_bed(bed, [(0, S, E)], ["chr22"])
print(f"inputs: {rb.n} records, BAM {os.path.getsize(bam) / 1e6:.1f} MB (written in {time.time() - t0:.0f} s)", flush=True)
exe = os.path.join(ROOT, "bio_amd", "bio-pileup")
for args in (["-bed", bed, "-format", "basestrand-tsv"], ["-region", "chr22:1000001-2000000"],
             ["-bed", bed, "-stitch", "-min-base-qual", "20", "-format", "basestrand-tsv"]):
    best = 1e9
    for _ in range(3):
        t0 = time.time()
        subprocess.run([exe, *args, "-out", os.path.join(d, "o"), bam, fa], check=True)
        best = min(best, time.time() - t0)
    print(f"bio-pileup {' '.join(a for a in args if not a.startswith('/'))}: {best:.2f} s wall (best of 3, process start to files closed)", flush=True)
opts = O.make_opts()
t0 = time.time()
wr, wp, wv = O.pileup(opts, [L], {0: [[S, E]]}, rb)
print(f"CPU oracle (records in memory -> rows, 1 thread): {time.time() - t0:.2f} s, depth sum {int(wv[:, 0].sum())}")
