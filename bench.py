#!/usr/bin/env python3
"""bench.py -- aligned bases piled up per second on MI355X (BASELINE.json metric).

A step = one pass of the pileup hot path (window planning + count/flush kernels, i.e.
bp_compute; with per-read lists also their scan + gather, bp_per_read_build) over one resident
batch of synthetic coordinate-sorted records.

--config selects the workload (BASELINE.json configs):
  c2   (default, BASELINE configs[1]) synthetic 30x chr1 (248,956,422 bp), 2x150 FR pairs with
       the section 8(d) CIGAR/flag/MAPQ mix, BED = whole contig, basestrand (per-strand counts in
       one pass), DefaultOpts
  c2q  the same records, -min-base-qual 20            (quality-rule kernel variant)
  c2c  the same records, -min-base-qual 20 -clip 5    (clip variant)
  c2s  the same records, -stitch -min-base-qual 20    (stitched variant)
  c4   (configs[3]) 500 targets x 200 bp at ~10,000x, 2x150 mates with insert ~N(170,20),
       -stitch -min-base-qual 20 -mapq 60
  c5   (configs[4] on one GPU) chr1 at 60x with the ordered per-read lists of -cols dpref,quals
  c3   (configs[2]) 24 contigs with hg38's lengths (3.1 Gb) at 30x cut into N position ranges, one
       per rank, reads partitioned by start, boundary rows merged over RCCL; needs --gpus >= 4
       (or BENCH_EMULATE_RANK=g/N to run one rank's share on a one-GPU box)
With --gpus N (c2 only besides c3) every rank owns a chr1-sized range of ONE contig of N x chr1
positions: weak scaling, reads partitioned by start, boundary merge per step.

Prints ONE JSON line (rank 0).
"""
import argparse
import hashlib
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHR1_LEN = 248_956_422
# hg38 primary assembly, chr1..chr22, chrX, chrY (the contig lengths of BASELINE config 3)
HG38_LENS = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
             138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
             83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# secondary ceiling: the kernel issues one ds_add_u32 per aligned base;
# a wave's ds_add_u32 (64 lanes x address + data) occupies a CU's LDS for 4 cycles = 16 lane
# atomics per cycle and CU, x 256 CUs x 2.4 GHz
LDS_ATOMIC_PEAK_G = 16 * 256 * 2.4

CONFIGS = {
    "c2": dict(opts={}, coverage=30.0, what="DefaultOpts"),
    "c2q": dict(opts=dict(MinBaseQual=20), coverage=30.0, what="-min-base-qual 20"),
    "c2c": dict(opts=dict(MinBaseQual=20, Clip=5), coverage=30.0, what="-min-base-qual 20 -clip 5"),
    "c2s": dict(opts=dict(MinBaseQual=20, Stitch=True), coverage=30.0, what="-stitch -min-base-qual 20"),
    "c4": dict(opts=dict(MinBaseQual=20, Stitch=True, Mapq=60), coverage=10000.0, what="-stitch -min-base-qual 20 -mapq 60"),
    "c5": dict(opts=dict(MinBaseQual=13), coverage=60.0, per_read=True, what="-min-base-qual 13 -cols dpref,quals"),
    "c3": dict(opts={}, coverage=30.0, what="DefaultOpts"),
}


def keep_mask(rb, flag_exclude=0xF00, mapq=60):
    """Records passing the read-level filters of processShard (pileup.go:691, :717-729) in a
    basestrand run (StrandNone dropped, pileup.go:724-729)."""
    flags = rb.flags.astype(np.uint32)
    keep = ((flags & flag_exclude) == 0) & (rb.mapq >= mapq) & (rb.n_cigar > 0)
    f = flags & 0xF0
    same = rb.ref_id == rb.mate_ref_id
    fwd = same & ((f == 0x60) | (f == 0x90) | (((flags & 8) != 0) & ((f & 0x30) == 0)))
    rev = same & ((f == 0x50) | (f == 0xA0) | (((flags & 8) != 0) & ((f & 0x30) == 0x30)))
    return keep & (fwd | rev)


def kept_aligned_bases(rb, opts_kw, bed=None):
    """B of SURVEY.md section 8(d), counted on the host from the generated records: CIGAR-M
    bases of records passing the read-level filters that lie on a BED position (bed = sorted
    [[start, end), ...] of the records' one contig; None: every position is a BED position)."""
    keep = keep_mask(rb, opts_kw["flag_exclude"], opts_kw["mapq"])
    off = rb.cigar_off.astype(np.int64)
    nc = rb.n_cigar.astype(np.int64)
    # the ops of the batch's records, flattened (a sliced batch still points into its parent's arrays)
    first = np.cumsum(nc) - nc
    read_of_op = np.repeat(np.arange(rb.n), nc)
    cig = rb.cigar[np.repeat(off - first, nc) + np.arange(int(nc.sum()))]
    ops = cig & 15
    lens = (cig >> 4).astype(np.int64)
    if bed is None:
        mlen = np.where(ops == 0, lens, 0)
        csum = np.concatenate([[0], np.cumsum(mlen)])
        per_read = csum[first + nc] - csum[first]
        return int(per_read[keep].sum())
    # reference position of every op: pos of its read + reference bases consumed before it
    rcons = np.where((ops == 0) | (ops == 2) | (ops == 3), lens, 0)
    rsum = np.concatenate([[0], np.cumsum(rcons)])
    op_start = rb.pos.astype(np.int64)[read_of_op] + (rsum[:-1] - rsum[first][read_of_op])
    sel = (ops == 0) & keep[read_of_op]
    s, e = op_start[sel], op_start[sel] + lens[sel]
    ep = np.asarray(bed, np.int64).reshape(-1)
    cum = np.concatenate([[0], np.cumsum(ep[1::2] - ep[0::2])])  # BED positions before interval k

    def cov(x):  # BED positions < x
        k = np.searchsorted(ep, x, "right")
        inside = (k & 1) == 1
        return cum[k // 2] + np.where(inside, x - ep[np.maximum(k - 1, 0)], 0)
    return int((cov(e) - cov(s)).sum())


def c4_panel():
    """BASELINE.json configs[3]: 500 targets x 200 bp on one contig (seeded)."""
    rng = random.Random(4)
    targets, p = [], 10_000
    for _ in range(500):
        targets.append([p, p + 200])
        p += 200 + rng.randint(300, 3000)
    return targets, p + 10_000


def kernel_source_hash():
    """What a committed traffic measurement is valid for: the kernels' source text."""
    h = hashlib.sha256()
    for f in ("pileup_kernels.hip", "pileup_device.h"):
        h.update(open(os.path.join(ROOT, "bio_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def committed_traffic(cfg, kernel):
    """HBM traffic is not measurable inside this process (PMC counters need their own rocprofv3
    passes): report the committed measurement of the same config and kernel if it was taken on
    these very kernel sources, else null."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", f"traffic_{cfg}.json")))
        if tj["kernel_source_sha256"] == kernel_source_hash() and kernel in tj["kernel"]:
            return tj["traffic_bytes_per_launch"], f"profiles/traffic_{cfg}.json ({tj['source']})"
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def e2e_block(rb, L, B, o, opts_kw, job, cpu_sample_mb, with_cpu):
    """The whole run the reference's users see, for the same records: BAM (BGZF) + FASTA + BED ->
    bio-pileup -> basestrand TSV on disk, process start to files closed, as aligned bases per
    second; and beside it the CPU side of the same thing on a bounded sample: BAM -> records
    (oracle/bam_decode.c, zlib, one thread) -> rows (oracle/pileup_oracle.c), without rendering."""
    import re
    import shutil
    import subprocess
    import tempfile
    from bio_amd import bamio, host
    base = os.environ.get("E2E_TMP", "/tmp")
    need = int(rb.n * 330 * 0.3 + 36 * L + 1.1 * L) + (1 << 30)  # BAM + text + FASTA, roughly
    if shutil.disk_usage(base).free < need:
        return {"skipped": f"less than {need / 1e9:.0f} GB free under {base} (set E2E_TMP)"}
    d = tempfile.mkdtemp(prefix="bp_e2e_", dir=base)
    try:
        t0 = time.time()
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
        del out, lines
        bedp = os.path.join(d, "all.bed")
        bamio.write_bed(bedp, [(0, 0, L)], ["chr1"])
        make_s = time.time() - t0
        flags = ["-bed", bedp, "-format", "basestrand-tsv"]
        if o.MinBaseQual:
            flags += ["-min-base-qual", str(o.MinBaseQual)]
        if o.Clip:
            flags += ["-clip", str(o.Clip)]
        if o.Stitch:
            flags += ["-stitch"]
        exe = os.path.join(ROOT, "bio_amd", "bio-pileup")
        t0 = time.time()
        r = subprocess.run([exe, *flags, "-out", os.path.join(d, "o"), bam, fa], capture_output=True, text=True,
                           env=dict(os.environ, BIO_PILEUP_STATS="1"))
        wall = time.time() - t0
        if r.returncode != 0:
            raise SystemExit("e2e: bio-pileup failed: " + r.stderr[-500:])
        outp = os.path.join(d, "o.basestrand.tsv")
        out_bytes = os.path.getsize(outp)
        # the file says what the device rows say (first 20,000 positions)
        n_chk = min(20000, L)
        _, _, dv = job.rows(0, n_chk)
        with open(outp, "rb") as f:
            head = f.read(64 * n_chk).split(b"\n")[1:n_chk + 1]
        for i, ln in enumerate(head):
            fld = ln.split(b"\t")
            if int(fld[1]) != i + 1 or [int(x) for x in fld[3:11]] != dv[i][1:9].tolist() or fld[2] != bytes(seq[i:i + 1]):
                raise SystemExit(f"e2e: line {i + 2} of the output differs from the device rows")
        m = re.search(r"read_s=([0-9.]+).*waiting for inflated blocks ([0-9.]+).*device_inflate_s=([0-9.]+).*"
                      r"resident=(\d) \(host walks (\d+)\).*total_s=([0-9.]+)", r.stderr)
        resident = bool(m and m.group(4) == "1")
        how = ("the inflated stream stays on the device: records found and their fields read there (bp_bam_scan), "
               "packed in place (bp_submit_bam_device)" if resident else
               "records found on the host, packed on the device (bp_submit_bam)")
        blk = {"value": round(B / wall / 1e9, 3), "unit": "Gbase/s", "wall_s": round(wall, 2),
               "what": f"bio-pileup {' '.join(x for x in flags if not x.startswith('/'))}: BAM ({os.path.getsize(bam) / 1e9:.2f} GB, "
                       f"BGZF level 1, {rb.n} records) + FASTA + BED -> basestrand.tsv ({out_bytes / 1e9:.2f} GB) on local disk, "
                       f"process start to files closed; BGZF inflated on the device (bp_inflate_blocks), {how}, "
                       f"text rendered on the device",
               "driver_stats": {"read_s": float(m.group(1)), "waiting_for_inflate_s": float(m.group(2)),
                                "device_inflate_kernels_s": float(m.group(3)), "resident": resident,
                                "host_walks": int(m.group(5)), "total_s": float(m.group(6))} if m else None,
               "inputs_made_in_s": round(make_s, 1)}
        if not (o.MinBaseQual or o.Clip or o.Stitch):
            # the same run with -format basestrand-tsv-bgz: the lines leave the device as BGZF members
            # (deflate_kernels.hip); the file's first members and its last one are inflated here (zlib)
            # and compared with the plain file
            import struct
            import zlib
            flags_z = ["-bed", bedp, "-format", "basestrand-tsv-bgz"]
            t0 = time.time()
            r = subprocess.run([exe, *flags_z, "-out", os.path.join(d, "z"), bam, fa], capture_output=True, text=True)
            wall_z = time.time() - t0
            if r.returncode != 0:
                raise SystemExit("e2e: bio-pileup -format basestrand-tsv-bgz failed: " + r.stderr[-500:])
            zp = os.path.join(d, "z.basestrand.tsv.gz")
            z_bytes = os.path.getsize(zp)
            with open(zp, "rb") as f, open(outp, "rb") as g:
                got, p = b"", 0
                buf = f.read(64 << 16)
                for _ in range(48):
                    total = struct.unpack_from("<H", buf, p + 16)[0] + 1
                    got += zlib.decompressobj(-15).decompress(buf[p + 18:p + total - 8])
                    p += total
                if got != g.read(len(got)) or len(got) < 40 * 0xff00:  # (the header line is a short member of its own)
                    raise SystemExit("e2e: the first BGZF members do not inflate to the head of the plain file")
                f.seek(-28, 2)
                if f.read() != bamio._EOF:
                    raise SystemExit("e2e: no BGZF EOF member at the end of the -bgz file")
            blk["bgz"] = {"value": round(B / wall_z / 1e9, 3), "unit": "Gbase/s", "wall_s": round(wall_z, 2),
                          "what": f"the same run with -format basestrand-tsv-bgz: the lines compressed on the device "
                                  f"(bp_render_fetch_bgzf), {z_bytes / 1e9:.2f} GB written ({out_bytes / z_bytes:.2f}x); "
                                  f"first 48 members inflate (zlib) to the head of the plain file"}
            os.remove(zp)
        for pth in (outp, bam):
            os.remove(pth)
        if with_cpu:
            import oracle_lib as O
            sample_bp = int(min(cpu_sample_mb * 1e6, L))
            hi = int(np.searchsorted(rb.pos, sample_bp + (1200 if o.Stitch else 0), "left"))
            sbam = os.path.join(d, "sample.bam")
            host.bam_write(sbam, ["chr1"], [L], rb.slice(0, hi), level=1, threads=16)
            oopts = O.make_opts(clip=o.Clip, mapq=o.Mapq, min_base_qual=o.MinBaseQual, stitch=int(o.Stitch))
            tc = time.perf_counter()
            srb, _ = O.bam_decode(sbam)
            dec_s = time.perf_counter() - tc
            _, _, wv = O.pileup(oopts, [L], {0: [[0, sample_bp]]}, srb, parallelism=1)
            cpu_s = time.perf_counter() - tc
            n_cmp = sample_bp - 1200
            _, _, dv = job.rows(0, n_cmp)
            if not (dv == wv[:n_cmp]).all():
                raise SystemExit("e2e: device rows differ from the CPU oracle's rows from the decoded sample BAM")
            cpu_B = kept_aligned_bases(srb, opts_kw, [[0, sample_bp]])
            blk["cpu"] = {"value": round(cpu_B / cpu_s / 1e9, 4), "unit": "Gbase/s", "cores": 1, "kind": "port",
                          "cpu_model": cpu_model(), "nproc": os.cpu_count(),
                          "sample": f"first {sample_bp / 1e6:.0f} Mb ({hi} records, {os.path.getsize(sbam) / 1e9:.2f} GB BAM): "
                                    f"BGZF inflate + record decode (oracle/bam_decode.c, zlib) {dec_s:.1f} s, then rows "
                                    f"(oracle/pileup_oracle.c) - {cpu_s:.1f} s in all, 1 thread, no text output"}
            # ... and on all host cores of the box's share: the whole contig as T position ranges,
            # one BAM and one thread each (independent jobs with a read halo, like the reference's,
            # pileup.go:862-932), every thread inflating + decoding its file and piling up its range
            if not o.Stitch:
                from concurrent.futures import ThreadPoolExecutor
                T = min(16, os.cpu_count() or 1)
                edges = [L * i // T for i in range(T + 1)]
                parts = []
                for i in range(T):
                    a = int(np.searchsorted(rb.pos, max(0, edges[i] - 1024), "left"))
                    b = int(np.searchsorted(rb.pos, edges[i + 1], "left"))
                    pth = os.path.join(d, f"part{i}.bam")
                    host.bam_write(pth, ["chr1"], [L], rb.slice(a, b), level=1, threads=16)
                    parts.append(pth)

                def one(i):
                    prb, _ = O.bam_decode(parts[i])
                    _, _, v = O.pileup(oopts, [L], {0: [[edges[i], edges[i + 1]]]}, prb, parallelism=1)
                    return v.sum(axis=0, dtype=np.uint64)
                tc = time.perf_counter()
                with ThreadPoolExecutor(T) as ex:
                    tot = sum(ex.map(one, range(T)))
                mt_s = time.perf_counter() - tc
                if [int(x) for x in tot] != job.column_sums():
                    raise SystemExit("e2e: column sums of the multi-threaded CPU run differ from the device's")
                blk["cpu"]["all_cores"] = {"value": round(B / mt_s / 1e9, 4), "unit": "Gbase/s", "cores": T,
                                           "sample": f"the whole contig as {T} position ranges, one BAM and one thread each "
                                                     f"(BGZF inflate + record decode + rows), in {mt_s:.1f} s; all 11 column sums "
                                                     f"equal the device's"}
        return blk
    finally:
        shutil.rmtree(d, ignore_errors=True)


def self_launch(n):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks as fresh child
    processes (torch.distributed.run, one rank per GPU) BEFORE this process makes any GPU call,
    pass their output through and leave with their exit status."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    sys.exit(subprocess.run(cmd).returncode)


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--length", type=int, default=CHR1_LEN, help="contig length of the c2*/c5 configs (default chr1)")
    ap.add_argument("--coverage", type=float, default=None)
    ap.add_argument("--cpu-sample-mb", type=float, default=None,
                    help="genomic megabases of the workload timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true",
                    help="skip the end-to-end block (BAM + FASTA -> files through bio-pileup; c2* configs at N=1)")
    ap.add_argument("--e2e-cpu-sample-mb", type=float, default=20.0)
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    cname = args.config
    coverage = args.coverage if args.coverage is not None else cfg["coverage"]

    emulate = os.environ.get("BENCH_EMULATE_RANK")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not emulate:
        self_launch(args.gpus)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not (emulate and world == 1):
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's world size and --gpus disagree "
                         f"(leave WORLD_SIZE unset and bench.py starts its own ranks; BENCH_EMULATE_RANK=g/N runs one "
                         f"rank's share alone)")
    if world > 1 and cname not in ("c2", "c3"):
        raise SystemExit(f"--config {cname} is a one-GPU workload")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the pileup path has no CPU fallback")
    # one rank per GPU; BENCH_DIST_BACKEND=gloo lets several ranks share one GPU to rehearse the
    # multi-rank flow on a single-GPU box
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)

    from bio_amd import partition, snp, synth
    sys.path.insert(0, os.path.join(ROOT, "tests"))

    L = args.length
    o = snp.Opts(**cfg["opts"])
    opts_kw = dict(flag_exclude=o.FlagExclude, mapq=o.Mapq)
    bopts = snp.device_opts(o, "basestrand-tsv", device=dev_index)
    per_read = bool(cfg.get("per_read"))

    t0 = time.time()
    threads = min(16, os.cpu_count() or 1)
    prank, pworld = rank, world
    if world == 1 and emulate:
        prank, pworld = (int(x) for x in emulate.split("/"))
    tail = 1024 if pworld > 1 else 0
    bed_list = None  # BED intervals when they are not the whole range (c4)
    gen_kw = {}
    if cname == "c4":
        targets, total_len = c4_panel()
        regions = [(0, s, e, coverage) for s, e in targets]
        bed = {0: targets}
        bed_list = targets
        ref_lens, ref_names = [total_len], ["chr1"]
        gen_kw = dict(insert_mean=170.0, insert_sd=20.0, insert_min=100, insert_max=300)
        own_rows, tail_rows = 500 * 200, 0
        workload = (f"synthetic deep panel: 500 targets x 200 bp at ~{coverage:g}x, 2x150 mates with insert ~N(170,20) "
                    f"(every pair overlaps), {cfg['what']}, basestrand counts")
    elif cname == "c3":
        # The genome's positions are cut into pworld runs; a run may span contigs.  Reads are
        # partitioned by start: behind the END of a rank's last range its reads spill into `tail`
        # rows that belong to the next rank (when that rank continues the same contig).
        if pworld < 4:
            raise SystemExit("--config c3 (3.1 Gb at 30x) needs --gpus >= 4: a rank's share must fit one GPU and one job's 32 GiB of payload")
        share = partition.boundary_share({r: [[0, l]] for r, l in enumerate(HG38_LENS)}, pworld, prank, tail)
        last = share["ranges"][-1]
        # reads are generated up to the end of a range that the next rank continues, and stop
        # 1,000 positions short of a contig's end
        regions = [(r, s, (e if (r, s, e) == last and share["spill"] else e - 1000), coverage) for r, s, e in share["ranges"]]
        bed = share["bed"]
        tail_rows = share["tail_rows"]
        own_rows = share["own_rows"]
        ref_lens = list(HG38_LENS)
        ref_names = [f"chr{i + 1}" for i in range(22)] + ["chrX", "chrY"]
        total_len = sum(HG38_LENS)
        workload = (f"synthetic {coverage:g}x whole genome: 24 contigs with hg38's lengths ({total_len} bp) cut into "
                    f"{pworld} position ranges, 2x150 FR pairs, basestrand counts, DefaultOpts")
    else:
        # N > 1: ONE contig of N x L positions cut into N position ranges (weak scaling: every
        # rank owns L positions at the same coverage).  Reads are partitioned, not replicated:
        # rank g gets the reads that start in its range, so what they contribute behind its end
        # lands in `tail` extra rows that go to rank g+1 after the kernels (the boundary merge of
        # BASELINE config 3; the only data-path communication, one small all-gather per step).
        total_len = L * pworld
        start, end, bed_end = partition.boundary_ranges(total_len, pworld, prank, tail)
        gen_end = end if prank + 1 < pworld else end - 1000
        regions = [(0, start, gen_end, coverage)]
        bed = {0: [[start, bed_end]]}
        ref_lens, ref_names = [total_len], ["chr1"]
        own_rows, tail_rows = end - start, bed_end - end
        workload = (f"synthetic {coverage:g}x chr1-sized contig ({L} bp) per GPU, 2x150 FR pairs, BED = whole contig, "
                    f"basestrand (per-strand) counts, {cfg['what']}")
    sb = synth.generate(regions, seed=synth.SEED + prank, threads=threads, with_names=bool(o.Stitch), **gen_kw)
    rb = sb.to_record_batch()
    if cname in ("c2", "c3") and pworld > 1:
        # the generator also starts pairs a little before a region: those belong to the left neighbour
        r0, s0 = regions[0][0], regions[0][1]
        if s0 > 0:
            key = rb.ref_id.astype(np.int64) << 32 | rb.pos.astype(np.int64)
            rb = rb.slice(int(np.searchsorted(key, (r0 << 32) | s0, "left")), rb.n)
    gen_s = time.time() - t0
    B = kept_aligned_bases(rb, opts_kw, bed_list)

    job = snp.PileupJob(bopts, ref_lens, ref_names)
    job.set_bed(bed)
    job.reserve(rb.n, len(rb.cigar) * 4 + len(rb.seq) + len(rb.qual))
    t0 = time.time()
    job.submit(rb)
    torch.cuda.synchronize()
    submit_s = time.time() - t0
    host_bytes = rb.algorithmic_input_bytes()

    wire = "cuda" if backend == "nccl" else "cpu"

    def step():
        job.compute()
        if per_read:
            job._check(job.lib.bp_per_read_build(job.h))
        if dist is not None:
            partition.exchange_boundaries(dist, torch, job, own_rows, tail_rows, tail, rank, world, device=wire)

    t0 = time.time()
    job.finish()
    finish_s = time.time() - t0  # stitched runs: includes the upload of the mate table
    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    kernel_ms, plan_ms, pr_ms, scan_ms = [], [], [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        t = job.timing()
        kernel_ms.append(t.pileup_ms)
        plan_ms.append(t.plan_ms)
        pr_ms.append(t.per_read_ms)
        scan_ms.append(t.per_read_scan_ms)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        # outside the data path: MAX of elapsed, SUM of bases
        elapsed, total_B = partition.reduce_timing(dist, elapsed, B, device=wire)
    else:
        total_B = B

    # sanity outside the timed region: device depth sum against the host-counted aligned bases
    sums = job.column_sums()
    if dist is not None:
        # after the merge: depth over the rows each rank OWNS, summed over ranks, is every
        # aligned base of every rank (a tail left rank g and arrived at rank g+1)
        t_sum = 0
        if tail_rows:
            tt = torch.zeros(tail_rows * 11, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()  # the job copies on its own stream: the fill must be done first
            job.rows_export(own_rows, tail_rows, tt.data_ptr())
            t_sum = int(tt.view(-1, 11)[:, 0].to(torch.int64).sum().item())
        both = torch.tensor([sums[0] - t_sum, B], dtype=torch.int64, device=wire)
        dist.all_reduce(both)
        if int(both[0]) != int(both[1]):
            raise SystemExit(f"merged depth over owned rows {int(both[0])} != aligned bases of all ranks {int(both[1])}")
    elif os.environ.get("BENCH_TUNING_NO_CHECK"):  # tuning builds may be wrong on purpose
        pass
    elif o.Stitch:
        # positions both mates cover count once (pileup.go:409-430)
        if not (B // 2 <= sums[0] <= B):
            raise SystemExit(f"stitched device depth sum {sums[0]} outside [{B // 2}, {B}]")
    elif sums[0] != B:
        raise SystemExit(f"device depth sum {sums[0]} != host aligned bases {B}")
    algo_bytes = job.algorithmic_bytes()
    n_rows = job.num_rows
    tm = job.timing()
    n_feat = int(job.lib.bp_per_read_total(job.h)) if per_read else 0
    if per_read and n_feat != sum(sums[1:9]):
        raise SystemExit(f"{n_feat} list entries != {sum(sums[1:9])} counted bases")

    cpu = None
    if rank == 0 and pworld == 1 and not args.no_cpu_baseline:
        import oracle_lib as O
        oopts = O.make_opts(clip=o.Clip, mapq=o.Mapq, min_base_qual=o.MinBaseQual, stitch=int(o.Stitch))
        if cname == "c4":
            sample_bed = {0: targets}
            hi = rb.n
            what = "all 500 targets"
            n_cmp = 200 * len(targets)
        else:
            mb = args.cpu_sample_mb if args.cpu_sample_mb is not None else (20.0 if per_read else 100.0 if o.Stitch else 200.0)
            sample_bp = int(min(mb * 1e6, L))
            sample_bed = {0: [[0, sample_bp]]}
            hi = int(np.searchsorted(rb.pos, sample_bp + (1200 if o.Stitch else 0), "left"))
            what = f"first {sample_bp / 1e6:.0f} Mb"
            n_cmp = sample_bp - 1200
        sample = rb.slice(0, hi)
        tc = time.perf_counter()
        if per_read:
            wr, wp, wv, woff, wfeat = O.pileup_per_read(oopts, ref_lens, sample_bed, sample, parallelism=1)
        else:
            wr, wp, wv = O.pileup(oopts, ref_lens, sample_bed, sample, parallelism=1)
        cpu_s = time.perf_counter() - tc
        cpu_B = kept_aligned_bases(sample, opts_kw, sample_bed[0])
        # and it is a parity check too: same rows from the device
        dr, dp, dv = job.rows(0, n_cmp)
        if not (dv == wv[:n_cmp]).all():
            raise SystemExit("device rows differ from the CPU oracle on the baseline sample")
        if per_read:
            doff, dfeat = job.per_read(0, n_cmp)
            if not ((doff == woff[:4 * n_cmp + 1]).all() and dfeat.tobytes() == wfeat[:int(woff[4 * n_cmp])].tobytes()):
                raise SystemExit("device per-read lists differ from the CPU oracle on the baseline sample")
        cpu = {"value": round(cpu_B / cpu_s / 1e9, 4), "unit": "Gbase/s", "cores": 1, "kind": "port",
               "cpu_model": cpu_model(), "nproc": os.cpu_count(),
               "sample": f"{what} of the same workload ({hi} records, {cpu_B} aligned bases on the sample's BED) in "
                         f"{cpu_s:.1f} s, records in memory -> rows{' + per-read lists' if per_read else ''}, same options, "
                         f"C restatement of the reference algorithm (oracle/pileup_oracle.c), 1 thread"}
        del wr, wp, wv, dr, dp, dv
        if cname == "c2":
            # the same restatement on all host cores of the box's share: independent position
            # ranges with a read halo, like the reference's jobs (pileup.go:862-932)
            T = min(16, os.cpu_count() or 1)
            mt_sums, mt_s = O.whole_contig_column_sums(oopts, L, rb, T)
            if mt_sums != sums:
                raise SystemExit(f"column sums of the device rows {sums} != those of the CPU oracle's rows over the whole contig {mt_sums}")
            mt_B = mt_sums[0]
            cpu["all_cores"] = {"value": round(mt_B / mt_s / 1e9, 4), "unit": "Gbase/s", "cores": T,
                                "sample": f"the whole workload as {T} position ranges, one thread each, in {mt_s:.1f} s; "
                                          f"all 11 column sums of its rows equal the device's"}

    e2e = None
    if rank == 0 and pworld == 1 and not args.no_e2e and cname in ("c2", "c2q", "c2c", "c2s") and not emulate:
        # (a secondary block: whatever goes wrong in it - no room for 12 GB of files under E2E_TMP, say -
        # must not take the bench line with it)
        try:
            e2e = e2e_block(rb, L, B, o, opts_kw, job, args.e2e_cpu_sample_mb, not args.no_cpu_baseline)
        except (Exception, SystemExit) as ex:
            e2e = {"error": str(ex)[:300]}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        k_ms = float(np.mean(kernel_ms))
        variant = ("stitched" if o.Stitch else "quality rule" if o.MinBaseQual else "every base passes") + (" + clip" if o.Clip else "")
        kernels = [{"kernel": "bp::pileup_kernel", "variant": variant, "ms_avg": round(k_ms, 4),
                    "algorithmic_bytes_per_launch": algo_bytes,
                    "achieved_GBps": round(algo_bytes / (k_ms * 1e-3) / 1e9, 1),
                    "frac": round(algo_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}]
        dom, dom_ms, dom_bytes = "bp::pileup_kernel", k_ms, algo_bytes
        if per_read:
            g_ms = float(np.mean(pr_ms))
            # the records once more + 44 B per row read + 6 B per entry written + 16 B of list offsets per row
            g_bytes = algo_bytes + 6 * n_feat + 16 * n_rows
            kernels.append({"kernel": "bp::per_read_kernel", "ms_avg": round(g_ms, 4),
                            "algorithmic_bytes_per_launch": g_bytes, "list_entries": n_feat,
                            "achieved_GBps": round(g_bytes / (g_ms * 1e-3) / 1e9, 1),
                            "frac": round(g_bytes / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                            "scan_ms_avg": round(float(np.mean(scan_ms)), 4)})
            if g_ms > k_ms:
                dom, dom_ms, dom_bytes = "bp::per_read_kernel", g_ms, g_bytes
        if o.Stitch:
            # once per job, in the bp_finish before the timed steps: the pair rule worked out for every
            # pair and left in the packed records, where the counting kernel of every step finds it
            kernels.append({"kernel": "bp::stitch_resolve_kernel", "ms": round(float(tm.stitch_resolve_ms), 4),
                            "runs": "once per job (first compute after the last submit), not in the timed steps"})
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        traffic, traffic_src = committed_traffic(cname, dom)
        out = {
            "metric": "aligned bases piled up/sec",
            "value": round(total_B / (elapsed / args.steps) / 1e9, 3),
            "unit": "Gbase/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong" if cname == "c3" else "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": workload, "name": cname,
                "records_per_gpu": rb.n, "aligned_bases_per_gpu": B, "bed_positions_per_gpu": n_rows,
                "windows": int(tm.n_windows), "gpus_arg": args.gpus, "world_size": world,
                "partitioning": "one job, one GPU" if pworld == 1 else
                                f"{total_len} bp cut into {pworld} position ranges, reads partitioned by start, "
                                f"{tail}-row boundary merge per step (all-gather of {tail * 44} B per rank, stream-ordered)",
            },
            "roofline": {
                "bound": "hbm", "kernel": dom,
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": dom_bytes,
                "kernel_ms_avg": round(dom_ms, 4), "plan_ms_avg": round(float(np.mean(plan_ms)), 4),
                "kernels": kernels,
                "lds_atomics": {"achieved": round(B / (k_ms * 1e-3) / 1e9, 1), "peak": LDS_ATOMIC_PEAK_G,
                                "unit": "G ds_add_u32 lanes/s", "frac": round(B / (k_ms * 1e-3) / 1e9 / LDS_ATOMIC_PEAK_G, 4)},
            },
            "cpu_baseline": cpu,
            "e2e": e2e,
            "setup": {"generate_s": round(gen_s, 1), "pack_and_h2d_s": round(submit_s, 2), "finish_s": round(finish_s, 3),
                      # inside bp_submit: the serial pass in record order (order check, errors, mate pairing
                      # in stitch mode) and the passes on all packing threads (spans, filters, hashes, packing)
                      "submit_serial_scan_s": round(tm.submit_scan_ms / 1e3, 3),
                      "submit_parallel_pack_s": round(tm.submit_pack_ms / 1e3, 3),
                      "h2d_inclusive_Gbase_per_s": round(B / (submit_s + elapsed / args.steps) / 1e9, 3),
                      "stitch_resolve_ms": round(float(tm.stitch_resolve_ms), 4),
                      # one step with the once-per-job pair-rule kernel counted in (stitched runs; else = value)
                      "resolve_inclusive_Gbase_per_s": round(B / (elapsed / args.steps + tm.stitch_resolve_ms * 1e-3) / 1e9, 3),
                      "host_input_bytes": host_bytes},
        }
        if emulate and world == 1:
            out["config"]["emulated_rank"] = emulate
        print(json.dumps(out))
    job.close()
    sb.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
