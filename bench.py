#!/usr/bin/env python3
"""bench.py -- aligned bases piled up per second on MI355X (BASELINE.json metric).

A step = one pass of the pileup hot path (window planning + count/flush kernels, i.e.
bp_compute) over one resident batch of synthetic coordinate-sorted records.

Workload at N=1 (BASELINE.json configs[1]): synthetic 30x chr1 (248,956,422 bp), 2x150 FR
pairs with the SURVEY.md section 8(d) CIGAR/flag/MAPQ mix, BED = whole contig, basestrand
semantics (per-strand counts in one pass).  With --gpus N every rank processes its own
chr1-sized contig (independent genomic partitions, no data-path collective): weak scaling.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CHR1_LEN = 248_956_422
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# secondary ceiling: the kernel issues one ds_add_u32 per aligned base;
# a wave's ds_add_u32 (64 lanes x address + data) occupies a CU's LDS for 4 cycles = 16 lane
# atomics per cycle and CU, x 256 CUs x 2.4 GHz
LDS_ATOMIC_PEAK_G = 16 * 256 * 2.4


def kept_aligned_bases(rb, opts_kw):
    """B of SURVEY.md section 8(d), counted on the host from the generated records:
    CIGAR-M bases of records passing the read-level filters (all of them lie on the BED)."""
    flags = rb.flags.astype(np.uint32)
    keep = ((flags & opts_kw["flag_exclude"]) == 0) & (rb.mapq >= opts_kw["mapq"]) & (rb.n_cigar > 0)
    f = flags & 0xF0
    same = rb.ref_id == rb.mate_ref_id
    fwd = same & ((f == 0x60) | (f == 0x90) | (((flags & 8) != 0) & ((f & 0x30) == 0)))
    rev = same & ((f == 0x50) | (f == 0xA0) | (((flags & 8) != 0) & ((f & 0x30) == 0x30)))
    keep &= fwd | rev  # basestrand formats drop StrandNone (pileup.go:724-729)
    ops = rb.cigar & 15
    mlen = np.where(ops == 0, rb.cigar >> 4, 0).astype(np.int64)
    csum = np.concatenate([[0], np.cumsum(mlen)])
    off = rb.cigar_off.astype(np.int64)
    per_read = csum[off + rb.n_cigar.astype(np.int64)] - csum[off]
    return int(per_read[keep].sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--length", type=int, default=CHR1_LEN, help="contig length (default chr1)")
    ap.add_argument("--coverage", type=float, default=30.0)
    ap.add_argument("--cpu-sample-mb", type=float, default=200.0,
                    help="genomic megabases of the workload timed on the CPU oracle (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the pileup path has no CPU fallback")
    # one rank per GPU; BENCH_DIST_BACKEND=gloo lets several ranks share one GPU to rehearse the
    # multi-rank flow on a single-GPU box (the ranks' only traffic is two scalar reductions)
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

    from bio_amd import _abi, snp, synth
    sys.path.insert(0, os.path.join(ROOT, "tests"))

    L = args.length
    opts_kw = dict(flag_exclude=0xF00, mapq=60)
    o = snp.Opts()  # DefaultOpts
    bopts = snp.device_opts(o, "basestrand-tsv", device=dev_index)

    t0 = time.time()
    threads = min(16, os.cpu_count() or 1)
    from bio_amd import partition
    # N > 1: ONE contig of N x L positions cut into N position ranges (weak scaling: every
    # rank owns L positions at the same coverage).  Reads are partitioned, not replicated:
    # rank g gets the reads that start in its range, so what they contribute behind its end
    # lands in `tail` extra rows that go to rank g+1 after the kernels (the boundary merge of
    # BASELINE config 3; the only data-path communication, one small all-gather per step).
    # (BENCH_EMULATE_RANK=g/N: this single process takes the place of rank g of N, without the
    # exchange, to check one shard of a large run on a one-GPU box)
    prank, pworld = rank, world
    if world == 1 and os.environ.get("BENCH_EMULATE_RANK"):
        prank, pworld = (int(x) for x in os.environ["BENCH_EMULATE_RANK"].split("/"))
    tail = 1024 if pworld > 1 else 0
    total_len = L * pworld
    start, end, bed_end = partition.boundary_ranges(total_len, pworld, prank, tail)
    gen_end = end if prank + 1 < pworld else end - 1000
    sb = synth.generate([(0, start, gen_end, args.coverage)], seed=synth.SEED + prank, threads=threads)
    rb = sb.to_record_batch()
    if start > 0:  # the generator also starts pairs a little before its region: not ours
        rb = rb.slice(int(np.searchsorted(rb.pos, start, "left")), rb.n)
    gen_s = time.time() - t0
    B = kept_aligned_bases(rb, opts_kw)
    own_rows, tail_rows = end - start, bed_end - end

    job = snp.PileupJob(bopts, [total_len], ["chr1"])
    job.set_bed({0: [[start, bed_end]]})
    job.reserve(rb.n, len(rb.cigar) * 4 + len(rb.seq) + len(rb.qual))
    t0 = time.time()
    job.submit(rb)
    torch.cuda.synchronize()
    submit_s = time.time() - t0
    host_bytes = rb.algorithmic_input_bytes()

    wire = "cuda" if backend == "nccl" else "cpu"

    def step():
        job.compute()
        if dist is not None:
            partition.exchange_boundaries(dist, torch, job, own_rows, tail_rows, tail, rank, world, device=wire)

    for _ in range(args.warmup):
        step()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    kernel_ms = []
    plan_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        t = job.timing()
        kernel_ms.append(t.pileup_ms)
        plan_ms.append(t.plan_ms)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        # outside the data path: MAX of elapsed, SUM of bases
        elapsed, total_B = partition.reduce_timing(dist, elapsed, B, device="cuda" if backend == "nccl" else "cpu")
    else:
        total_B = B

    # sanity outside the timed region: device depth sum == host-counted aligned bases
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
    elif sums[0] != B and not os.environ.get("BENCH_TUNING_NO_CHECK"):  # tuning builds may be wrong on purpose
        raise SystemExit(f"device depth sum {sums[0]} != host aligned bases {B}")
    algo_bytes = job.algorithmic_bytes()
    n_rows = job.num_rows
    tm = job.timing()

    cpu = None
    if rank == 0 and pworld == 1 and not args.no_cpu_baseline:
        import oracle_lib as O
        sample_bp = int(min(args.cpu_sample_mb * 1e6, L))
        hi = int(np.searchsorted(rb.pos, sample_bp, "left"))
        sample = rb.slice(0, hi)
        oopts = O.make_opts()
        tc = time.perf_counter()
        wr, wp, wv = O.pileup(oopts, [L], {0: [[0, sample_bp]]}, sample, parallelism=1)
        cpu_s = time.perf_counter() - tc
        cpu_B = int(wv[:, 0].sum())
        # and it is a parity check too: same rows from the device
        dr, dp, dv = job.rows(0, sample_bp - 600)
        if not (dv == wv[:sample_bp - 600]).all():
            raise SystemExit("device rows differ from the CPU oracle on the baseline sample")
        cpu = {"value": round(cpu_B / cpu_s / 1e9, 4), "unit": "Gbase/s", "cores": 1, "kind": "port",
               "sample": f"first {sample_bp / 1e6:.0f} Mb of the same workload ({hi} records, "
                         f"{cpu_B} aligned bases) in {cpu_s:.1f} s, records in memory -> rows, "
                         f"C restatement of the reference algorithm (oracle/pileup_oracle.c), 1 thread"}
        del wr, wp, wv, dr, dp, dv
        # the same restatement on all host cores of the box's share: independent position
        # ranges with a read halo, like the reference's jobs (pileup.go:862-932)
        from concurrent.futures import ThreadPoolExecutor
        T = min(16, os.cpu_count() or 1)
        edges = [L * i // T for i in range(T + 1)]

        def one(i):
            s, e = edges[i], edges[i + 1]
            a = int(np.searchsorted(rb.pos, max(0, s - 1024), "left"))
            b = int(np.searchsorted(rb.pos, e, "left"))
            _, _, v = O.pileup(oopts, [L], {0: [[s, e]]}, rb.slice(a, b), parallelism=1)
            return int(v[:, 0].sum())
        tc = time.perf_counter()
        with ThreadPoolExecutor(T) as ex:
            mt_B = sum(ex.map(one, range(T)))
        mt_s = time.perf_counter() - tc
        if mt_B != B:
            raise SystemExit(f"multi-threaded CPU run counted {mt_B} bases, expected {B}")
        cpu["all_cores"] = {"value": round(mt_B / mt_s / 1e9, 4), "unit": "Gbase/s", "cores": T,
                            "sample": f"the whole workload as {T} position ranges, one thread each, in {mt_s:.1f} s"}

    # HBM traffic is not measurable inside this process (PMC counters need their own
    # rocprofv3 passes): report the committed measurement of the same workload, if any
    traffic = None
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
        if tj["workload"] == {"length": L, "coverage": args.coverage}:
            traffic = tj["traffic_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        k_ms = float(np.mean(kernel_ms))
        achieved = algo_bytes / (k_ms * 1e-3) / 1e9
        out = {
            "metric": "aligned bases piled up/sec",
            "value": round(total_B / (elapsed / args.steps) / 1e9, 3),
            "unit": "Gbase/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": f"synthetic {args.coverage:g}x chr1-sized contig ({L} bp) per GPU, 2x150 FR pairs, "
                            "BED = whole contig, basestrand (per-strand) counts, DefaultOpts",
                "records_per_gpu": rb.n, "aligned_bases_per_gpu": B, "bed_positions_per_gpu": n_rows,
                "windows": int(tm.n_windows),
                "partitioning": "one contig, one GPU" if world == 1 else
                                f"one contig of {total_len} bp cut into {world} position ranges, reads partitioned by "
                                f"start, {tail}-row boundary merge per step (all-gather of {tail * 44} B per rank)",
            },
            "roofline": {
                "bound": "hbm", "kernel": "bp::pileup_kernel",
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)"
                                  if traffic else None,
                "algorithmic_bytes_per_launch": algo_bytes,
                "kernel_ms_avg": round(k_ms, 4), "plan_ms_avg": round(float(np.mean(plan_ms)), 4),
                "lds_atomics": {"achieved": round(B / (k_ms * 1e-3) / 1e9, 1), "peak": LDS_ATOMIC_PEAK_G,
                                "unit": "G ds_add_u32 lanes/s", "frac": round(B / (k_ms * 1e-3) / 1e9 / LDS_ATOMIC_PEAK_G, 4)},
            },
            "cpu_baseline": cpu,
            "setup": {"generate_s": round(gen_s, 1), "pack_and_h2d_s": round(submit_s, 2),
                      "h2d_inclusive_Gbase_per_s": round(B / (submit_s + elapsed / args.steps) / 1e9, 3),
                      "host_input_bytes": host_bytes},
        }
        print(json.dumps(out))
    job.close()
    sb.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
