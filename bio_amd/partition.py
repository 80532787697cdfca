"""Genomic partitioning for multi-GPU runs: rank g owns a contiguous coordinate range and
receives the records of its padded range -- the reference's own job structure
(pileupSNPMain, pileup/snp/pileup.go:862-909: shard slice per job, BED Subset of exactly
its range, maxReadSpan padding), with no data-path collective.  Host logic only; the counts
of a partition come from the device path (bio_amd.snp)."""
import numpy as np


def bed_positions(bed):
    return sum(e - s for iv in bed.values() for s, e in iv)


def plan_partitions(bed, world_size):
    """Cut the BED positions (ascending (ref,pos) order) into world_size contiguous runs of
    near-equal size.  Returns, per rank, a list of (ref_id, start, end) ranges (a run may
    span contigs).  Never splits a position; ranks can be empty when there are fewer
    positions than ranks."""
    total = bed_positions(bed)
    cuts = [(total * g) // world_size for g in range(world_size + 1)]
    parts = [[] for _ in range(world_size)]
    g, done = 0, 0
    for ref in sorted(bed):
        for s, e in bed[ref]:
            while s < e:
                while g < world_size - 1 and done >= cuts[g + 1]:
                    g += 1
                take = e - s if g == world_size - 1 else min(e - s, cuts[g + 1] - done)
                if take <= 0:
                    g += 1
                    continue
                if parts[g] and parts[g][-1][0] == ref and parts[g][-1][2] == s:
                    parts[g][-1] = (ref, parts[g][-1][1], s + take)
                else:
                    parts[g].append((ref, s, s + take))
                s += take
                done += take
    return parts


def subset_bed(bed, ranges, pad=0):
    """BED restricted to the (padded) ranges: {ref: [[s,e],...]}, sorted and disjoint."""
    out = {}
    for ref, s, e in ranges:
        lo, hi = max(0, s - pad), e + pad
        for a, b in bed.get(ref, []):
            x, y = max(a, lo), min(b, hi)
            if x < y:
                iv = out.setdefault(ref, [])
                if iv and iv[-1][1] >= x:
                    iv[-1][1] = max(iv[-1][1], y)
                else:
                    iv.append([x, y])
    return out


def padding(max_read_span, stitch):
    """(records left, records right, BED) padding of a partition.  Unstitched, a position
    only sees reads that start up to maxReadSpan before it (gbam.Shard.PaddedStart).
    Stitched, a read's mate and the BED positions that decide whether the mate is kept lie
    up to two spans away on either side (see bio_amd/csrc/host/pileup.cpp step 6)."""
    if stitch:
        return 2 * max_read_span, 2 * max_read_span, 2 * max_read_span
    return max_read_span, 0, 0


def record_spans(batch, ranges, pad_left, pad_right):
    """Index spans [(lo,hi),...] (ascending, disjoint) of the coordinate-sorted records that
    start inside any padded range; feed them to RecordBatch.take_spans / slice."""
    key = batch.ref_id.astype(np.int64) << 32 | batch.pos.astype(np.int64)
    spans = []
    for ref, s, e in ranges:
        lo = int(np.searchsorted(key, (ref << 32) | max(0, s - pad_left), "left"))
        hi = int(np.searchsorted(key, (ref << 32) | (e + pad_right), "left"))
        if spans and lo <= spans[-1][1]:
            spans[-1][1] = max(spans[-1][1], hi)
        elif hi > lo:
            spans.append([lo, hi])
    return [(a, b) for a, b in spans]


def run_partition(bopts, ref_len, bed, batch, ranges, max_read_span, stitch, ref_names=None):
    """Rows of one partition from the device path: padded records and BED in, own rows out."""
    from . import snp
    pl, pr, pb = padding(max_read_span, stitch)
    part = batch.take_spans(record_spans(batch, ranges, pl, pr))
    rows = snp.run_job(bopts, ref_len, subset_bed(bed, ranges, pb), [part], ref_names)
    return own_rows(rows, ranges)


def own_rows(rows, ranges):
    """Drop the rows of the padding: keep (ref,pos) inside the rank's own ranges."""
    ref, pos, vals = rows
    keep = np.zeros(len(pos), bool)
    for r, s, e in ranges:
        keep |= (ref == r) & (pos >= s) & (pos < e)
    return ref[keep], pos[keep], vals[keep]


def reduce_timing(dist, elapsed_s, units, device="cpu"):
    """bench.py's only communication: MAX of the ranks' elapsed time, SUM of their units."""
    import torch
    te = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    tu = torch.tensor([units], dtype=torch.int64, device=device)
    dist.all_reduce(te, op=dist.ReduceOp.MAX)
    dist.all_reduce(tu, op=dist.ReduceOp.SUM)
    return float(te.item()), int(tu.item())


# ---- read-disjoint partition of ONE contig with a boundary merge (SURVEY 8e scheme B) --------
#
# Rank g of G owns positions [start_g, end_g) of the contig and is given every read it is
# told to own, each read to exactly one rank (bench.py: the fragments that start in its range).
# Its job also covers the `tail` positions behind end_g, where those reads spill over; after
# the kernels, the tail rows travel to rank g+1 and are added to its first rows.  That is the
# only data-path communication: one all-gather of tail x 11 uint32 per rank (22-45 KB).

def boundary_ranges(total_len, world_size, rank, tail):
    """-> (start, end, bed_end): rank's own positions [start, end) and the end of the range its
    job computes, end + tail clipped to the contig."""
    start = total_len * rank // world_size
    end = total_len * (rank + 1) // world_size
    return start, end, min(end + tail, total_len)


def boundary_share(bed, world_size, rank, tail):
    """The read-disjoint scheme over a whole genome (BASELINE config 3): the BED positions of all
    contigs are cut into world_size runs of equal size (plan_partitions); a run may span contigs
    and may end inside one.  Every read goes to exactly one rank, by its start: rank g owns the
    reads from the start of its first range up to the start of rank g+1's first range (rank 0 from
    the beginning, the last rank to the end).  When rank g+1 starts on the contig rank g ends on
    (`spill`), g's reads reach into positions g+1 owns - at most `tail` behind g+1's start, which
    the caller guarantees by tail >= max_read_span - so g's job also covers the BED positions in
    [next_start, next_start + tail): those tail_rows rows come last in the job's row order and are
    what travels to rank g+1 (exchange_boundaries), whose first tail_rows rows they are.  A rank
    whose successor starts a new contig has no spill: nothing its reads cover belongs to anyone
    else.  Raises when a tail would reach two ranks far or a rank is empty.
    -> dict(ranges=[(ref,start,end)...], bed={ref: [[s,e]...]} for the job, own_rows, tail_rows,
    spill, rec_lo / rec_hi = (ref,pos) keys of the reads the rank is given (None = open end))."""
    parts = plan_partitions(bed, world_size)
    if any(not p for p in parts):
        raise ValueError(f"fewer BED positions than ranks: {world_size} ranks")
    ranges = parts[rank]
    job_bed = {}
    for r, s, e in ranges:
        job_bed.setdefault(r, []).append([s, e])
    own = sum(e - s for _, s, e in ranges)
    spill, tail_rows = False, 0
    rec_lo = None if rank == 0 else (ranges[0][0], ranges[0][1])
    rec_hi = None
    if rank + 1 < world_size:
        nref, nstart, _ = parts[rank + 1][0]
        rec_hi = (nref, nstart)
        if nref == ranges[-1][0]:
            spill = True
            lim = nstart + tail
            ext = [[max(s, nstart), min(e, lim)] for r, s, e in parts[rank + 1] if r == nref and s < lim]
            tail_rows = sum(e - s for s, e in ext)
            if tail_rows != sum(min(e, lim) - max(s, nstart) for s, e in bed[nref] if s < lim and e > nstart):
                raise ValueError(f"rank {rank + 1}'s share is shorter than the {tail}-position boundary tail")
            for s, e in ext:
                if job_bed[nref][-1][1] == s:
                    job_bed[nref][-1][1] = e
                else:
                    job_bed[nref].append([s, e])
    return dict(ranges=ranges, bed=job_bed, own_rows=own, tail_rows=tail_rows, spill=spill,
                rec_lo=rec_lo, rec_hi=rec_hi)


def owned_record_span(batch, share):
    """(lo, hi): the coordinate-sorted records a rank of the read-disjoint scheme is given."""
    key = batch.ref_id.astype(np.int64) << 32 | batch.pos.astype(np.int64)
    lo = 0 if share["rec_lo"] is None else int(np.searchsorted(key, (share["rec_lo"][0] << 32) | share["rec_lo"][1], "left"))
    hi = batch.n if share["rec_hi"] is None else int(np.searchsorted(key, (share["rec_hi"][0] << 32) | share["rec_hi"][1], "left"))
    return lo, hi


def exchange_boundaries(dist, torch, job, own_rows, tail_rows, tail, rank, world_size, device="cuda",
                        stage="cuda"):
    """After job.compute(): all-gather the tail rows and add the left neighbour's onto this
    rank's first rows.  own_rows / tail_rows = this job's row counts inside and behind its own
    range (tail_rows < tail only for the last rank, which has no right neighbour); `tail` = the
    fixed row count every rank contributes; own_rows >= tail (a tail reaches one rank far).
    `device` = where the collective's tensors live ("cuda" for RCCL, "cpu" for a gloo
    rehearsal); `stage` = where job.rows_export / rows_accumulate read and write ("cuda": the
    rows stay in HBM; "cpu" only for the host-side stand-in job of the CPU tests)."""
    if world_size > 1 and own_rows < tail:
        raise ValueError(f"partition of {own_rows} positions is shorter than the {tail}-row boundary tail")
    send = torch.zeros(tail * 11, dtype=torch.int32, device=stage)
    if stage == "cuda":
        # The job works on its own HIP stream and torch (and RCCL under it) on theirs.  The
        # hand-overs are ordered on the device, by events, with no host wait: the export runs
        # behind the zero fill, the collective behind the export, the add behind the collective
        # (a collective called with async_op=False has torch's current stream wait for it).
        js = torch.cuda.ExternalStream(job.stream)
        cur = torch.cuda.current_stream()
        js.wait_stream(cur)
        if tail_rows:
            job.rows_export_async(own_rows, tail_rows, send.data_ptr())
        send.record_stream(js)  # the caching allocator must not hand the buffer out again before js is past it
        cur.wait_stream(js)
        wire = send if device == stage else send.to(device)
        got = [torch.empty_like(wire) for _ in range(world_size)]
        dist.all_gather(got, wire)
        if rank > 0:
            left = got[rank - 1] if device == stage else got[rank - 1].to(stage)
            js.wait_stream(cur)
            job.rows_accumulate_async(0, tail, left.data_ptr())
            left.record_stream(js)
        return
    # host-side stand-in job (the CPU tests): plain calls, nothing is asynchronous
    if tail_rows:
        job.rows_export(own_rows, tail_rows, send.data_ptr())
    wire = send if device == stage else send.to(device)
    got = [torch.empty_like(wire) for _ in range(world_size)]
    dist.all_gather(got, wire)
    if rank > 0:
        left = got[rank - 1] if device == stage else got[rank - 1].to(stage)
        job.rows_accumulate(0, tail, left.data_ptr())
