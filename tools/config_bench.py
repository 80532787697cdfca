#!/usr/bin/env python3
"""Secondary measurements (not bench lines): kernel time of the other BASELINE.json configs'
shapes on one GPU, each checked against the CPU oracle on a sample.
  c4   500 targets x 200 bp, ~10,000x, 2x150 overlapping mates, -stitch -min-base-qual 20
  c1q  30x, 20 Mb, -min-base-qual 20 -clip 5 (the plain path with the quality rule)
  c1s  30x, 20 Mb, -stitch -min-base-qual 20
"""
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from bio_amd import snp, synth  # noqa: E402


def run(name, regions, bed, ref_len, opts, check_rows, **gen):
    sb = synth.generate(regions, with_names=bool(opts.stitch), threads=16, **gen)
    rb = sb.to_record_batch()
    with snp.PileupJob(opts, ref_len) as job:
        job.set_bed(bed)
        t0 = time.time()
        job.submit(rb)
        job.finish()
        sub = time.time() - t0
        ms = []
        for _ in range(5):
            job.compute()
            ms.append(job.timing().pileup_ms)
        sums = job.column_sums()
        ref, pos, vals = job.rows(0, min(check_rows, job.num_rows))
        ab = job.algorithmic_bytes()
    # oracle on the leading rows' worth of BED
    sub_bed, left = {}, len(pos)
    for r in sorted(bed):
        for s, e in bed[r]:
            if left <= 0:
                break
            take = min(left, e - s)
            sub_bed.setdefault(r, []).append([s, s + take])
            left -= take
    wr, wp, wv = O.pileup(opts, ref_len, sub_bed, rb)
    ok = bool((wv == vals[:len(wp)]).all())
    k = float(np.median(ms))
    print(f"{name}: records {rb.n}  rows {sums and len(pos)}+  depth sum {sums[0]}  kernel {k:.3f} ms  "
          f"{sums[0] / k / 1e6:.1f} Gdepth/s  algorithmic {ab / k / 1e6:.0f} GB/s ({ab / k / 1e6 / 8000:.1%} of HBM peak)  "
          f"submit {sub:.2f} s  parity on {len(wp)} rows: {'ok' if ok else 'MISMATCH'}", flush=True)
    sb.close()
    return ok


def main():
    which = sys.argv[1:] or ["c4", "c1q", "c1s"]
    ok = True
    if "c4" in which:
        rng = random.Random(4)
        targets, p = [], 10_000
        for _ in range(500):
            targets.append([p, p + 200])
            p += 200 + rng.randint(300, 3000)
        bed = {0: targets}
        regions = [(0, s, e, 10000.0) for s, e in targets]
        opts = O.make_opts(stitch=1, min_base_qual=20, mapq=60)
        ok &= run("c4 deep panel, stitched", regions, bed, [p + 10_000], opts, 2000,
                  insert_mean=170.0, insert_sd=20.0, insert_min=100, insert_max=300)
    if "c4p" in which:
        rng = random.Random(4)
        targets, p = [], 10_000
        for _ in range(500):
            targets.append([p, p + 200])
            p += 200 + rng.randint(300, 3000)
        bed = {0: targets}
        regions = [(0, s, e, 10000.0) for s, e in targets]
        for nm, o in (("minq0", O.make_opts(mapq=60)), ("minq20", O.make_opts(min_base_qual=20, mapq=60))):
            ok &= run("c4 deep panel, not stitched, " + nm, regions, bed, [p + 10_000], o, 2000,
                      insert_mean=170.0, insert_sd=20.0, insert_min=100, insert_max=300)
    L = 20_000_000
    if "c1q" in which:
        ok &= run("30x 20 Mb, min-base-qual 20 clip 5", [(0, 0, L - 1000, 30.0)], {0: [[0, L]]}, [L],
                  O.make_opts(min_base_qual=20, clip=5), 2_000_000)
    if "c1s" in which:
        ok &= run("30x 20 Mb, stitched min-base-qual 20", [(0, 0, L - 1000, 30.0)], {0: [[0, L]]}, [L],
                  O.make_opts(stitch=1, min_base_qual=20), 2_000_000)
    LF = 248_956_422  # the bench workload's contig with the other option sets
    if "c2q" in which:
        ok &= run("30x chr1, min-base-qual 20 clip 5", [(0, 0, LF - 1000, 30.0)], {0: [[0, LF]]}, [LF],
                  O.make_opts(min_base_qual=20, clip=5), 1_000_000)
    if "c2m" in which:
        ok &= run("30x chr1, min-base-qual 20", [(0, 0, LF - 1000, 30.0)], {0: [[0, LF]]}, [LF],
                  O.make_opts(min_base_qual=20), 1_000_000)
    if "c2s" in which:
        ok &= run("30x chr1, stitched min-base-qual 20", [(0, 0, LF - 1000, 30.0)], {0: [[0, LF]]}, [LF],
                  O.make_opts(stitch=1, min_base_qual=20), 1_000_000)
    if "c5" in which:
        ok &= run_per_read("c5 60x 20 Mb, per-read lists (-cols dpref,quals)", L, 60.0,
                           O.make_opts(min_base_qual=13), 200_000)
    if "c5full" in which:  # BASELINE config 5's size on one contig: chr1 at 60x (~120 GB of HBM)
        ok &= run_per_read("c5 60x chr1, per-read lists (-cols dpref,quals)", 248_956_422, 60.0,
                           O.make_opts(min_base_qual=13), 200_000)
    sys.exit(0 if ok else 1)


def run_per_read(name, L, cov, opts, check_rows):
    """BASELINE config 5's shape (60x, ordered per-read quals): time of bp_per_read_build after
    the rows exist; the leading rows' lists are compared with the oracle's entry for entry."""
    sb = synth.generate([(0, 0, L - 1000, cov)], threads=16)
    rb = sb.to_record_batch()
    with snp.PileupJob(opts, [L]) as job:
        job.set_bed({0: [[0, L]]})
        job.submit(rb)
        job.finish()
        k_ms = job.timing().pileup_ms
        ts, ks, ss = [], [], []
        for _ in range(3):
            job.compute()  # invalidates the lists
            t0 = time.perf_counter()
            job._check(job.lib.bp_per_read_build(job.h))
            ts.append((time.perf_counter() - t0) * 1e3)
            ks.append(job.timing().per_read_ms)
            ss.append(job.timing().per_read_scan_ms)
        n_feat = int(job.lib.bp_per_read_total(job.h))
        off, feats = job.per_read(0, check_rows)
        ab = job.algorithmic_bytes()
    hi = int(np.searchsorted(rb.pos, check_rows + 600, "left"))
    _, _, _, woff, wfeat = O.pileup_per_read(opts, [L], {0: [[0, check_rows]]}, rb.slice(0, hi))
    ok = bool((woff == off).all() and (wfeat == feats).all())
    t, k, sc = float(np.median(ts)), float(np.median(ks)), float(np.median(ss))
    bytes_ = ab + 6 * n_feat + 32 * L  # + the lists written once and 4 u64 offsets per row
    print(f"{name}: records {rb.n}  entries {n_feat}  rows kernel {k_ms:.3f} ms  lists: gather kernel {k:.2f} ms "
          f"({n_feat / k / 1e6:.1f} Gentry/s, {bytes_ / k / 1e6:.0f} GB/s algorithmic = {bytes_ / k / 1e6 / 8000:.1%} of HBM peak), "
          f"scan {sc:.2f} ms, all of bp_per_read_build {t:.2f} ms  "
          f"parity on {check_rows} rows: {'ok' if ok else 'MISMATCH'}", flush=True)
    sb.close()
    return ok


if __name__ == "__main__":
    main()
