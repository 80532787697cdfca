"""Multi-GPU path on CPU: world_size-2 gloo processes, each owning one genomic partition.
The partition/padding logic and bench.py's only communication (MAX time, SUM units) are
exercised for real; the per-partition counts come from the CPU oracle here (the checker
standing in for the device), and the concatenated rows must equal the one-job result bit
for bit."""
import os
import random
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O
from bio_amd import partition as P, records as R
from test_oracle_vs_naive import random_bed, random_records

REF_LENS = [6000, 2500, 4000]
SPAN = 120


def _inputs(seed):
    rng = random.Random(seed)
    recs = []
    for r, L in enumerate(REF_LENS):
        part = random_records(rng, 300, ref_len=L, read_len=40)
        for x in part:
            x["name"] = f"c{r}{x['name']}"
            x["mate_ref_id"] = r if x["mate_ref_id"] == x["ref_id"] else (r + 1) % len(REF_LENS)
            x["ref_id"] = r
        recs += part
    recs.sort(key=lambda x: (x["ref_id"], x["pos"]))
    bed = {r: random_bed(rng, L, dense=True)[0] for r, L in enumerate(REF_LENS) if r != 1}
    bed[1] = [[10, 20], [2000, 2400]]
    return R.from_records(recs), bed


def _partition_rows(batch, bed, ranges, stitch):
    opts = O.make_opts(max_read_len=100, max_read_span=SPAN, stitch=stitch, min_base_qual=20)
    pl, pr, pb = P.padding(SPAN, stitch)
    part = batch.take_spans(P.record_spans(batch, ranges, pl, pr))
    sub = P.subset_bed(bed, ranges, pb)
    if not sub:
        z = np.zeros(0, np.int32)
        return z, z, np.zeros((0, 11), np.uint32)
    # one unpadded shard per contig range = "the job sees exactly what it was handed"
    rows = O.pileup(opts, REF_LENS, sub, part, parallelism=1)
    return P.own_rows(rows, ranges)


def _worker(rank, world, port, seed, stitch, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        batch, bed = _inputs(seed)
        ranges = P.plan_partitions(bed, world)[rank]
        ref, pos, vals = _partition_rows(batch, bed, ranges, stitch)
        dist.barrier()
        elapsed, units = P.reduce_timing(dist, 0.25 * (rank + 1), int(vals[:, 0].sum()))
        gathered = [None] * world
        dist.all_gather_object(gathered, (ref, pos, vals))
        if rank == 0:
            q.put((elapsed, units, gathered))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("stitch", [0, 1])
def test_two_ranks_equal_one_job(stitch):
    seed, world = 40 + stitch, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, seed, stitch, q)) for r in range(world)]
    for p in procs:
        p.start()
    elapsed, units, gathered = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    batch, bed = _inputs(seed)
    opts = O.make_opts(max_read_len=100, max_read_span=SPAN, stitch=stitch, min_base_qual=20)
    wr, wp, wv = O.pileup(opts, REF_LENS, bed, batch, parallelism=1)
    gr = np.concatenate([g[0] for g in gathered])
    gp = np.concatenate([g[1] for g in gathered])
    gv = np.concatenate([g[2] for g in gathered])
    assert (gr == wr).all() and (gp == wp).all()
    assert (gv == wv).all()
    assert elapsed == 0.5 and units == int(wv[:, 0].sum())
    assert all(len(g[1]) > 0 for g in gathered)


# ---- read-disjoint partition of one contig with a boundary merge (bench.py at N > 1) ----------
class _HostJob:
    """Stands in for snp.PileupJob: rows in host memory, same export / accumulate calls."""

    def __init__(self, vals):
        self.vals = np.ascontiguousarray(vals, np.uint32)

    def rows_export(self, first, n, ptr):
        import ctypes
        ctypes.memmove(ptr, self.vals[first:first + n].ctypes.data, n * 44)

    def rows_accumulate(self, first, n, ptr):
        import ctypes
        add = np.empty((n, 11), np.uint32)
        ctypes.memmove(add.ctypes.data, ptr, n * 44)
        self.vals[first:first + n] += add


MERGE_LEN, MERGE_TAIL, MERGE_SPAN = 9000, 200, 120


def _merge_inputs(seed):
    rng = random.Random(seed)
    recs = random_records(rng, 900, ref_len=MERGE_LEN, read_len=40)
    return R.from_records(recs)


def _merge_worker(rank, world, port, seed, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        batch = _merge_inputs(seed)
        start, end, bed_end = P.boundary_ranges(MERGE_LEN, world, rank, MERGE_TAIL)
        # every read goes to exactly one rank: the one that owns its start
        lo, hi = int(np.searchsorted(batch.pos, start, "left")), int(np.searchsorted(batch.pos, end, "left"))
        opts = O.make_opts(max_read_len=100, max_read_span=MERGE_SPAN, min_base_qual=20)
        shard, n = O.region_shard(0, start, bed_end)  # "the job sees exactly what it was handed"
        _, pos, vals = O.pileup(opts, [MERGE_LEN], {0: [[start, bed_end]]}, batch.slice(lo, hi), shards=shard, n_shards=n)
        job = _HostJob(vals)
        P.exchange_boundaries(dist, torch, job, end - start, bed_end - end, MERGE_TAIL, rank, world,
                              device="cpu", stage="cpu")
        gathered = [None] * world
        dist.all_gather_object(gathered, (pos[:end - start], job.vals[:end - start]))
        if rank == 0:
            q.put(gathered)
    finally:
        dist.destroy_process_group()


def test_boundary_merge_three_ranks_equal_one_job():
    seed, world = 77, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_merge_worker, args=(r, world, port, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    batch = _merge_inputs(seed)
    opts = O.make_opts(max_read_len=100, max_read_span=MERGE_SPAN, min_base_qual=20)
    wr, wp, wv = O.pileup(opts, [MERGE_LEN], {0: [[0, MERGE_LEN]]}, batch, parallelism=1)
    gp = np.concatenate([g[0] for g in gathered])
    gv = np.concatenate([g[1] for g in gathered])
    assert (gp == wp).all() and (gv == wv).all()
    # the merge mattered: reads do straddle the cuts
    cut = MERGE_LEN // world
    assert ((batch.pos < cut) & (batch.pos + 30 > cut)).any()
    with pytest.raises(ValueError, match="shorter than"):
        P.exchange_boundaries(None, torch, None, 10, 200, 200, 0, 2, device="cpu", stage="cpu")


# ---- BASELINE config 3's shape: 24 contigs cut into N runs, boundary merge where a run ends inside
# a contig (partition.boundary_share, the code bench.py --config c3 runs) --------------------------
HG38_LENS = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
             138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
             83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]
C3_TAIL, C3_SPAN = 200, 120


def _c3_lens(case, world=3):
    """hg38's lengths scaled down 40,000 times, the last contig's length adjusted so that the first
    cut (total // world) falls `case` positions behind the end of a contig: 0 = exactly at a
    contig boundary, -90 = just before it (a short tail, clipped by the contig's end), +50 = just
    after it (rank 0 owns the first 50 positions of the next contig)."""
    lens = [max(600, x // 40000) for x in HG38_LENS]
    cum = np.cumsum(lens)
    k = int(np.argmin(np.abs(cum - cum[-1] / world)))
    want_total = world * (int(cum[k]) + case)
    lens[-1] += want_total - int(cum[-1])
    assert lens[-1] >= 600 and sum(lens) == want_total
    return lens, k


def _c3_inputs(case, seed):
    lens, k = _c3_lens(case)
    rng = random.Random(seed)
    recs = []
    for r, L in enumerate(lens):
        part = random_records(rng, max(60, L // 12), ref_len=L, read_len=40)
        for x in part:
            x["name"] = f"c{r}{x['name']}"
            x["mate_ref_id"] = r if x["mate_ref_id"] == x["ref_id"] else (r + 1) % len(lens)
            x["ref_id"] = r
        recs += part
    recs.sort(key=lambda x: (x["ref_id"], x["pos"]))
    return R.from_records(recs), lens, {r: [[0, L]] for r, L in enumerate(lens)}, k


def _c3_worker(rank, world, port, case, seed, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        batch, lens, bed, _ = _c3_inputs(case, seed)
        share = P.boundary_share(bed, world, rank, C3_TAIL)
        lo, hi = P.owned_record_span(batch, share)
        opts = O.make_opts(max_read_len=100, max_read_span=C3_SPAN, min_base_qual=20)
        ref, pos, vals = O.pileup(opts, lens, share["bed"], batch.slice(lo, hi), parallelism=1)
        assert len(pos) == share["own_rows"] + share["tail_rows"]
        job = _HostJob(vals)
        P.exchange_boundaries(dist, torch, job, share["own_rows"], share["tail_rows"], C3_TAIL, rank, world,
                              device="cpu", stage="cpu")
        n = share["own_rows"]
        gathered = [None] * world
        dist.all_gather_object(gathered, (ref[:n], pos[:n], job.vals[:n], share, (lo, hi)))
        if rank == 0:
            q.put(gathered)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", [0, -90, 50])
def test_c3_shares_three_ranks_24_contigs_equal_one_job(case):
    seed, world = 300 + case, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_c3_worker, args=(r, world, port, case, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    batch, lens, bed, k = _c3_inputs(case, seed)
    opts = O.make_opts(max_read_len=100, max_read_span=C3_SPAN, min_base_qual=20)
    wr, wp, wv = O.pileup(opts, lens, bed, batch, parallelism=1)
    assert (np.concatenate([g[0] for g in gathered]) == wr).all()
    assert (np.concatenate([g[1] for g in gathered]) == wp).all()
    assert (np.concatenate([g[2] for g in gathered]) == wv).all()
    # every read went to exactly one rank
    spans = [g[4] for g in gathered]
    assert spans[0][0] == 0 and spans[-1][1] == batch.n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    # and the first cut is the case the test is named for
    s0, s1 = gathered[0][3], gathered[1][3]
    last = s0["ranges"][-1]
    if case == 0:
        assert last == (k, 0, lens[k]) and not s0["spill"] and s0["tail_rows"] == 0 and s1["ranges"][0][:2] == (k + 1, 0)
    elif case < 0:
        assert last == (k, 0, lens[k] + case) and s0["spill"] and s0["tail_rows"] == -case  # clipped by the contig's end
        # reads of rank 0 do reach over the cut
        cut_reads = (batch.ref_id == k) & (batch.pos < last[2]) & (batch.pos + 30 > last[2])
        assert cut_reads.any()
    else:
        assert last == (k + 1, 0, case) and s0["spill"] and s0["tail_rows"] == C3_TAIL and s1["ranges"][0][:2] == (k + 1, case)
    assert gathered[1][3]["spill"] and gathered[1][3]["tail_rows"] == C3_TAIL  # the second cut lies inside a contig
    assert not gathered[2][3]["spill"]


def test_boundary_share_properties():
    bed = {0: [[0, 1000]], 1: [[100, 400], [450, 500], [900, 2000]], 2: [[0, 50]]}
    for world in (1, 2, 3, 5):
        shares = [P.boundary_share(bed, world, g, 100) for g in range(world)]
        assert sum(s["own_rows"] for s in shares) == P.bed_positions(bed)
        assert shares[0]["rec_lo"] is None and shares[-1]["rec_hi"] is None and shares[-1]["tail_rows"] == 0
        for a, b in zip(shares, shares[1:]):
            assert a["rec_hi"] == b["rec_lo"] == b["ranges"][0][:2]
            assert a["spill"] == (a["ranges"][-1][0] == b["ranges"][0][0])
            # the tail rows are the first rows of the next rank
            if a["spill"]:
                nref, nstart = b["ranges"][0][:2]
                first = [(r, x) for r, s, e in b["ranges"] for x in range(s, e)][:a["tail_rows"]]
                tail = [(r, x) for r in sorted(a["bed"]) for s, e in a["bed"][r] for x in range(s, e)][a["own_rows"]:]
                assert tail == first and all(r == nref and nstart <= x < nstart + 100 for r, x in tail)
    with pytest.raises(ValueError, match="shorter than"):
        P.boundary_share({0: [[0, 300]]}, 3, 0, 150)
    with pytest.raises(ValueError, match="fewer BED positions"):
        P.boundary_share({0: [[0, 2]]}, 3, 0, 150)


def test_plan_partitions_properties():
    rng = random.Random(7)
    for world in (1, 2, 3, 8, 50):
        bed = {r: random_bed(rng, 3000)[0] for r in range(3)}
        parts = P.plan_partitions(bed, world)
        assert len(parts) == world
        sizes = [sum(e - s for _, s, e in p) for p in parts]
        assert sum(sizes) == P.bed_positions(bed)
        assert max(sizes) - min(sizes) <= 1
        # ascending, disjoint, and exactly the BED positions
        flat = [x for p in parts for x in p]
        assert flat == sorted(flat)
        seen = {(r, x) for r, s, e in flat for x in range(s, e)}
        want = {(r, x) for r, iv in bed.items() for s, e in iv for x in range(s, e)}
        assert seen == want and len(seen) == sum(sizes)
    assert P.plan_partitions({0: [[5, 7]]}, 4) in ([[(0, 5, 6)], [(0, 6, 7)], [], []],
                                                    [[], [(0, 5, 6)], [], [(0, 6, 7)]],
                                                    [[], [(0, 5, 6)], [(0, 6, 7)], []],
                                                    [[(0, 5, 6)], [], [(0, 6, 7)], []],
                                                    [[], [], [(0, 5, 6)], [(0, 6, 7)]],
                                                    [[], [(0, 5, 6)], [], [(0, 6, 7)]])


def test_take_spans_keeps_names_aligned():
    batch, _ = _inputs(3)
    sub = batch.take_spans([(5, 9), (20, 21), (30, 30), (40, 44)])
    idx = list(range(5, 9)) + [20] + list(range(40, 44))
    assert sub.n == len(idx)
    for k, i in enumerate(idx):
        a = bytes(sub.name[int(sub.name_off[k]):int(sub.name_off[k + 1])])
        b = bytes(batch.name[int(batch.name_off[i]):int(batch.name_off[i + 1])])
        assert a == b and sub.pos[k] == batch.pos[i]
