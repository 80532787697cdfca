"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle (bit-exact)."""
import ctypes as C
import json
import os
import random

import numpy as np
import pytest

import oracle_lib as O
from bio_amd import _abi, records as R, snp, synth
from test_oracle_vs_naive import CASES, random_bed, random_records

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def device_rows(opts, ref_len, bed, batches):
    return snp.run_job(opts, ref_len, bed, batches)


def assert_same(got, want, what=""):
    gr, gp, gv = got
    wr, wp, wv = want
    assert len(gp) == len(wp), (what, len(gp), len(wp))
    assert (gr == wr).all() and (gp == wp).all(), what
    bad = np.nonzero((gv != wv).any(axis=1))[0]
    assert len(bad) == 0, (what, len(bad), [(int(wp[i]), gv[i].tolist(), wv[i].tolist()) for i in bad[:5]])


def test_reference_kat_on_device():
    """pileup/snp/pileup_test.go:56-315 through the device path."""
    kat = json.load(open(os.path.join(GOLDEN, "pileup_kat.json")))
    o = kat["opts"]
    for case in kat["cases"]:
        opts = O.make_opts(clip=o["clip"], flag_exclude=o["flag_exclude"], mapq=o["mapq"],
                           max_read_len=o["max_read_len"], max_read_span=o["max_read_span"],
                           min_base_qual=o["min_base_qual"], stitch=o["stitch"], ignore_strand=0)
        batch = R.from_records([dict(r, ref_id=0, mate_ref_id=0) for r in case["reads"]])
        ref, pos, vals = device_rows(opts, [kat["ref"]["len"]], {0: kat["bed"]}, [batch])
        assert [int(p) for p in pos] == [w["pos"] for w in case["want"]]
        for i, w in enumerate(case["want"]):
            assert vals[i, 1:9].reshape(4, 2).tolist() == w["counts"], (case["name"], w["pos"])


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_random_small_vs_oracle(seed, case):
    rng = random.Random(1000 * case + seed)
    kw = dict(CASES[case])
    ref_len = 3000
    batch = R.from_records(random_records(rng, 150, ref_len=ref_len, n_ref=2))
    bed = random_bed(rng, ref_len, n_ref=2, dense=(seed % 2 == 0))
    opts = O.make_opts(max_read_len=100, max_read_span=120, **kw)
    want = O.pileup(opts, [ref_len, ref_len], bed, batch, parallelism=1)
    got = device_rows(opts, [ref_len, ref_len], bed, [batch])
    assert_same(got, want, f"case {case} seed {seed}")


def test_split_submits_and_empty_inputs():
    rng = random.Random(9)
    ref_len = 5000
    recs = random_records(rng, 400, ref_len=ref_len)
    batch = R.from_records(recs)
    bed = random_bed(rng, ref_len, dense=True)
    for stitch in (0, 1):
        opts = O.make_opts(max_read_len=100, max_read_span=120, stitch=stitch, min_base_qual=20)
        want = O.pileup(opts, [ref_len], bed, batch)
        parts = [batch.slice(0, 100), batch.slice(100, 101), batch.slice(101, 101),
                 batch.slice(101, batch.n)]
        assert_same(device_rows(opts, [ref_len], bed, parts), want, f"split stitch={stitch}")
        # no records at all: one zero row per BED position
        ref, pos, vals = device_rows(opts, [ref_len], bed, [])
        assert len(pos) == O.n_bed_positions(bed) and vals.sum() == 0
    # empty BED: no rows
    ref, pos, vals = device_rows(O.make_opts(), [ref_len], {}, [batch])
    assert len(pos) == 0


@pytest.mark.parametrize("clip", [0, 3])
def test_stitched_pairs_completed_across_submits_and_computes(clip):
    """Stitch mode with a compute after every submit: a read counted alone by one compute is half of
    a pair by the next (its mate came with a later submit), so the pair rule (stitch_resolve_kernel)
    runs several times over a growing job, each pair worked once; every compute's rows are the
    oracle's rows for the records submitted so far, and a compute repeated changes nothing."""
    sb = synth.generate([(0, 0, 60_000, 40.0)], seed=5, insert_mean=200.0, insert_sd=40.0, insert_min=60,
                        with_names=True)
    rb = sb.to_record_batch()
    ref_len, bed = [61_000], {0: [[100, 30_000], [30_500, 59_000]]}
    opts = O.make_opts(stitch=1, min_base_qual=20, clip=clip)
    cuts = [0, rb.n // 5, rb.n // 5 + 1, rb.n // 2, rb.n]
    with snp.PileupJob(opts, ref_len) as job:
        job.set_bed(bed)
        for a, b in zip(cuts[:-1], cuts[1:]):
            job.submit(rb.slice(a, b))
            job.compute()
            assert_same(job.rows(), O.pileup(opts, ref_len, bed, rb.slice(0, b), parallelism=1), f"after {b} records")
            job.compute()
            assert_same(job.rows(), O.pileup(opts, ref_len, bed, rb.slice(0, b), parallelism=1), f"again after {b} records")
        job.finish()
        assert_same(job.rows(), O.pileup(opts, ref_len, bed, rb, parallelism=1), "finished")
        assert job.timing().stitch_resolve_ms > 0
    sb.close()


def _synth_case(regions, bed, ref_len, opts, region_filter=None, **gen):
    sb = synth.generate(regions, **gen)
    rb = sb.to_record_batch()
    if region_filter is not None:
        ref, s, e = region_filter
        lo = int(np.searchsorted(rb.pos, s, "left"))
        hi = int(np.searchsorted(rb.pos, e, "left"))
        shards, n = O.region_shard(ref, s, e)
        want = O.pileup(opts, ref_len, bed, rb, shards=shards, n_shards=n)
        got = device_rows(opts, ref_len, bed, [rb.slice(lo, hi)])
    else:
        want = O.pileup(opts, ref_len, bed, rb, parallelism=1)
        got = device_rows(opts, ref_len, bed, [rb])
    return got, want, sb


@pytest.mark.parametrize("mode", ["basestrand", "tsv", "fwd", "rev", "stitch", "minq_clip"])
def test_c1_shape_1mb_30x(mode):
    """BASELINE config 1 shape: 1 Mb region at 30x, 2x150 reads, CIGAR/flag/MAPQ mix."""
    kw = dict(basestrand=dict(), tsv=dict(ignore_strand=1),
              fwd=dict(ignore_strand=1, strand_req=1), rev=dict(ignore_strand=1, strand_req=2),
              stitch=dict(stitch=1, min_base_qual=20), minq_clip=dict(min_base_qual=13, clip=5))[mode]
    opts = O.make_opts(**kw)
    ref_len = [3_000_000]
    bed = {0: [[1_000_000, 2_000_000]]}
    got, want, sb = _synth_case([(0, 1_000_000, 2_000_000, 30.0)], bed, ref_len, opts,
                                with_names=True)
    assert_same(got, want, mode)
    assert want[2][:, 0].sum() > 10e6  # the case is not vacuous


@pytest.mark.parametrize("mode", ["basestrand", "stitch"])
@pytest.mark.parametrize("n_windows", [2 * 2048 + 5, 5 * 2048 + 3])
def test_windows_per_workgroup_in_between(n_windows, mode):
    """A workgroup takes n_windows / 2048 consecutive windows, capped at 24 (launch_pileup): the
    small cases above run with 1, the full-size ones with 24; here 2 and 5, with a last workgroup
    that gets fewer windows than the others.  Whole rows against the oracle."""
    kw = dict(basestrand=dict(), stitch=dict(stitch=1, min_base_qual=20))[mode]
    opts = O.make_opts(**kw)
    L = n_windows * 1024 - 300
    bed = {0: [[0, L]]}
    got, want, sb = _synth_case([(0, 0, L - 1000, 4.0)], bed, [L], opts, with_names=True)
    assert len(want[1]) == L
    assert_same(got, want, f"{n_windows} windows {mode}")
    assert want[2][:, 0].sum() > 10e6
    sb.close()


def test_c1_region_mode_no_halo():
    """-region semantics: a single unpadded shard (pileup.go:1091-1097)."""
    opts = O.make_opts()
    bed = {0: [[1_200_000, 1_300_000]]}
    got, want, _ = _synth_case([(0, 1_000_000, 1_500_000, 30.0)], bed, [3_000_000], opts,
                               region_filter=(0, 1_200_000, 1_300_000))
    assert_same(got, want, "region")


@pytest.mark.parametrize("mode", ["stitch", "plain", "minq_clip"])
def test_c4_shape_deep_panel(mode):
    """BASELINE config 4 shape (scaled): 200-bp targets, deep, overlapping 2x150 mates.  Each
    window holds ~4,700 records: more than two slices, so the deep-window path (extra work
    items, atomic flush) is what is checked here."""
    rng = random.Random(4)
    targets, p = [], 10_000
    for _ in range(12):
        targets.append([p, p + 200])
        p += 200 + rng.randint(50, 3000)
    bed = {0: targets}
    regions = [(0, s, e, 2000.0) for s, e in targets]
    opts = O.make_opts(mapq=60, **dict(stitch=dict(stitch=1, min_base_qual=20), plain=dict(),
                                       minq_clip=dict(min_base_qual=25, clip=7))[mode])
    got, want, sb = _synth_case(regions, bed, [p + 10_000], opts, with_names=True,
                                insert_mean=170.0, insert_sd=20.0, insert_min=100, insert_max=300)
    assert_same(got, want, "panel")
    assert want[2][:, 0].max() > 1000


@pytest.mark.parametrize("clip", [0, 4])
def test_stitched_mates_with_indels(clip):
    """Every pair overlaps and most mates carry an indel or a soft clip: the segments of a
    record are cut at the mate's M ops (pileup.go:366-465 merges the two aligned-base lists)."""
    opts = O.make_opts(stitch=1, min_base_qual=25, clip=clip)
    ref_len = [400_000]
    bed = {0: [[50_000, 120_000], [120_500, 121_000], [130_000, 300_000]]}
    got, want, sb = _synth_case([(0, 40_000, 320_000, 40.0)], bed, ref_len, opts, with_names=True,
                                insert_mean=190.0, insert_sd=40.0, insert_min=60, insert_max=400,
                                frac_softclip=0.25, frac_del=0.25, frac_ins=0.25, frac_hardclip=0.05,
                                sub_rate=0.05, frac_low_mapq=0.0)
    assert_same(got, want, "indel mates")
    assert want[2][:, 9:11].sum() > 10_000  # disagreeing overlaps (counted as X) occur


def test_snp_panel_many_tiny_intervals():
    """1-bp targets like cmd/bio-pileup/testdata/common_chr2_subset.bed."""
    rng = random.Random(11)
    iv, p = [], 100_000
    while p < 400_000:
        iv.append([p, p + 1])
        p += rng.randint(2, 120)
    bed = {0: iv}
    opts = O.make_opts(ignore_strand=1)
    got, want, _ = _synth_case([(0, 100_000, 400_000, 20.0)], bed, [1_000_000], opts)
    assert_same(got, want, "snp panel")


def test_multi_reference_and_empty_contigs():
    """Rows for contigs without reads and after the last read (pileup_support.go:112-123)."""
    opts = O.make_opts()
    ref_len = [500_000, 300_000, 400_000, 100_000]
    bed = {0: [[100_000, 150_000]], 1: [[0, 1000]], 2: [[50_000, 60_000], [70_000, 70_500]],
           3: [[99_000, 100_000]]}
    got, want, _ = _synth_case([(0, 90_000, 160_000, 15.0), (2, 40_000, 80_000, 15.0)], bed, ref_len,
                               opts)
    assert_same(got, want, "multi-ref")


def test_errors_match_reference_wording():
    base = dict(name="x", ref_id=0, pos=100, mapq=60, flags=0x63, mate_pos=150)
    opts = O.make_opts(max_read_len=20, max_read_span=30)
    bed = {0: [[0, 1000]]}

    def run(rec, bed=bed, opts=opts):
        return device_rows(opts, [1000], bed, [R.from_records([rec])])

    long_read = dict(base, cigar=[("M", 21)], seq="A" * 21, qual=[30] * 21)
    with pytest.raises(snp.PileupError, match="maxReadLen is 20, but read x at"):
        run(long_read)
    wide = dict(base, cigar=[("M", 5), ("N", 26), ("M", 5)], seq="A" * 10, qual=[30] * 10)
    with pytest.raises(snp.PileupError, match="maxReadSpan is 30"):
        run(wide)
    badop = dict(base, cigar=[("M", 5), ("=", 5)], seq="A" * 10, qual=[30] * 10)
    with pytest.raises(snp.PileupError, match="unexpected CIGAR code"):
        run(badop)
    ref, pos, vals = run(badop, bed={0: [[100, 105]]})
    assert vals[:, 0].tolist() == [1] * 5
    ref, pos, vals = run(dict(long_read, mapq=3))
    assert vals.sum() == 0
    with pytest.raises(snp.PileupError, match="invalid clip"):
        device_rows(O.make_opts(clip=250), [1000], bed, [])
    with pytest.raises(snp.PileupError, match="max-read-len"):
        device_rows(O.make_opts(max_read_len=600), [1000], bed, [])
    with pytest.raises(snp.PileupError, match="minBaseQual too large"):
        device_rows(O.make_opts(min_base_qual=96), [1000], bed, [])
    # unsorted input
    a = dict(base, cigar=[("M", 5)], seq="AAAAA", qual=[30] * 5)
    with pytest.raises(snp.PileupError, match="out of coordinate order"):
        device_rows(opts, [1000], bed, [R.from_records([a, dict(a, pos=50)])])
    # a failed submit sticks: the job's bookkeeping is half advanced, it cannot go on
    with snp.PileupJob(opts, [1000]) as job:
        job.set_bed(bed)
        job.submit(R.from_records([a]))
        with pytest.raises(snp.PileupError, match="maxReadLen"):
            job.submit(R.from_records([dict(a, pos=200), dict(long_read, pos=300)]))
        for call in (lambda: job.submit(R.from_records([dict(a, pos=400)])), job.finish, job.compute):
            with pytest.raises(snp.PileupError, match="earlier bp_submit of this job failed .*maxReadLen"):
                call()
    # stitched overlap with an out-of-range quality: Go panics, we return BP_EDATA
    q = [30] * 5
    p1 = dict(name="p", ref_id=0, pos=100, mapq=60, flags=0x63, mate_pos=102, cigar=[("M", 5)],
              seq="AAAAA", qual=[30, 30, 30, 99, 30])
    p2 = dict(name="p", ref_id=0, pos=102, mapq=60, flags=0x93, mate_pos=100, cigar=[("M", 5)],
              seq="AAAAA", qual=q)
    with pytest.raises(snp.PileupError, match="index out of range"):
        device_rows(O.make_opts(stitch=1), [1000], bed, [R.from_records([p1, p2])])
    with pytest.raises(O.OracleError, match="index out of range"):
        O.pileup(O.make_opts(stitch=1), [1000], bed, R.from_records([p1, p2]))


def test_checksum_properties_at_scale():
    """Size-independent properties on a larger case (20 Mb at 30x): column sums from the
    device equal sums over rows, depth equals the kept aligned bases, strand split adds up."""
    L = 20_000_000
    sb = synth.generate([(0, 0, L, 30.0)], frac_softclip=0, frac_del=0, frac_ins=0, frac_hardclip=0,
                        frac_low_mapq=0, frac_dup=0, frac_secondary=0, frac_mate_other_ref=0)
    rb = sb.to_record_batch()
    bed = {0: [[0, L + 1000]]}
    opts = O.make_opts(ignore_strand=1)
    with snp.PileupJob(opts, [L + 1000]) as job:
        job.set_bed(bed)
        job.submit(rb)
        job.finish()
        sums = job.column_sums()
        # every read is plain 150M, unfiltered, inside the BED: depth = all aligned bases
        assert sums[0] == sb.aligned_bases
        assert sum(sums[1:]) == sums[0]  # min_base_qual 0: every base lands in a count slot
        assert sums[2] == sums[4] == sums[6] == sums[8] == sums[10] == 0  # tsv: slot 0 only
        ref, pos, vals = job.rows(5_000_000, 100_000)
        assert (pos == np.arange(5_000_000, 5_100_000)).all()
        # a window of rows against the oracle run on just the overlapping reads
        lo = int(np.searchsorted(rb.pos, 5_000_000 - 600))
        hi = int(np.searchsorted(rb.pos, 5_100_000))
        wr, wp, wv = O.pileup(opts, [L + 1000], {0: [[5_000_000, 5_100_000]]}, rb.slice(lo, hi))
        assert (vals == wv).all()
    opts2 = O.make_opts(ignore_strand=0)
    with snp.PileupJob(opts2, [L + 1000]) as job:
        job.set_bed(bed)
        job.submit(rb)
        job.finish()
        s2 = job.column_sums()
        assert s2[0] == sums[0]
        assert [s2[1 + 2 * b] + s2[2 + 2 * b] for b in range(5)] == [sums[1 + 2 * b] for b in range(5)]


@pytest.mark.parametrize("stitch", [0, 1])
def test_logical_partitions_on_one_device(stitch):
    """G genomic partitions (what G ranks would own) computed one after
    the other on this device must concatenate to the one-job rows bit for bit."""
    from bio_amd import partition as P
    rng = random.Random(77 + stitch)
    ref_lens = [6000, 3000]
    recs = []
    for r, L in enumerate(ref_lens):
        part = random_records(rng, 500, ref_len=L, read_len=60)
        for x in part:
            x["name"] = f"c{r}{x['name']}"
            x["mate_ref_id"] = r if x["mate_ref_id"] == x["ref_id"] else 1 - r
            x["ref_id"] = r
        recs += part
    recs.sort(key=lambda x: (x["ref_id"], x["pos"]))
    batch = R.from_records(recs)
    bed = {r: random_bed(rng, L, dense=True)[0] for r, L in enumerate(ref_lens)}
    opts = O.make_opts(max_read_len=100, max_read_span=150, stitch=stitch, min_base_qual=20)
    want = device_rows(opts, ref_lens, bed, [batch])
    assert_same(want, O.pileup(opts, ref_lens, bed, batch), "one job")
    for world in (2, 3, 7):
        parts = [P.run_partition(opts, ref_lens, bed, batch, rg, 150, bool(stitch))
                 for rg in P.plan_partitions(bed, world)]
        got = tuple(np.concatenate([p[i] for p in parts]) for i in range(3))
        assert_same(got, want, f"{world} partitions stitch={stitch}")


def test_boundary_merge_on_device():
    """bench.py's N > 1 layout on one device: reads partitioned by start over 3 position ranges,
    each job covering its range plus a tail; bp_rows_export / bp_rows_accumulate move the tails
    to the right neighbour.  The merged rows are those of one job (and of the oracle)."""
    import torch
    from bio_amd import partition as P
    L, G, tail = 600_000, 3, 1024
    opts = O.make_opts(min_base_qual=13)
    sb = synth.generate([(0, 0, L - 1000, 25.0)])
    rb = sb.to_record_batch()
    want = device_rows(opts, [L], {0: [[0, L]]}, [rb])
    assert_same(want, O.pileup(opts, [L], {0: [[0, L]]}, rb), "one job")
    jobs, spans = [], []
    for g in range(G):
        start, end, bed_end = P.boundary_ranges(L, G, g, tail)
        lo, hi = int(np.searchsorted(rb.pos, start, "left")), int(np.searchsorted(rb.pos, end, "left"))
        job = snp.PileupJob(opts, [L])
        job.set_bed({0: [[start, bed_end]]})
        job.submit(rb.slice(lo, hi))
        job.finish()
        jobs.append(job)
        spans.append((end - start, bed_end - end))
    # what exchange_boundaries does, with the neighbour in the same process
    for g in range(G - 1):
        own, tl = spans[g]
        buf = torch.zeros(tail * 11, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()  # the jobs work on their own streams
        jobs[g].rows_export(own, tl, buf.data_ptr())
        jobs[g + 1].rows_accumulate(0, tail, buf.data_ptr())
    got_pos = np.concatenate([jobs[g].rows(0, spans[g][0])[1] for g in range(G)])
    got_val = np.concatenate([jobs[g].rows(0, spans[g][0])[2] for g in range(G)])
    assert (got_pos == want[1]).all() and (got_val == want[2]).all()
    # and bench.py's own exchange (partition.exchange_boundaries: hand-overs ordered by stream
    # events, no host wait) with a stand-in for the collective that copies on torch's stream as
    # RCCL would; the ranks take their turns in ascending order, so the left neighbour's tail
    # is there when a rank gathers.  Fresh rows first.
    class _Gather:
        sent = {}

        def __init__(self, rank):
            self.rank = rank

        def all_gather(self, got, wire):
            _Gather.sent[self.rank] = wire.clone()
            for r, t in enumerate(got):
                if r in _Gather.sent:
                    t.copy_(_Gather.sent[r], non_blocking=True)
    for j in jobs:
        j.compute()
    for g in range(G):
        P.exchange_boundaries(_Gather(g), torch, jobs[g], spans[g][0], spans[g][1], tail, g, G)
    got_val2 = np.concatenate([jobs[g].rows(0, spans[g][0])[2] for g in range(G)])
    assert (got_val2 == want[2]).all()
    with pytest.raises(snp.PileupError, match="out of range"):
        jobs[0].rows_export(spans[0][0], tail + 1, buf.data_ptr())
    with pytest.raises(snp.PileupError, match="null argument"):
        jobs[0].rows_export(spans[0][0], tail, 0)
    for j in jobs:
        j.close()
    sb.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_per_read_lists_vs_oracle(seed):
    """appendBase (pileup.go:235-250): ordered per-read feature lists from the device against
    the oracle's, entry for entry."""
    rng = random.Random(600 + seed)
    ref_lens = [4000, 2500]
    recs = []
    for r, L in enumerate(ref_lens):
        part = random_records(rng, 300, ref_len=L, read_len=60)
        for x in part:
            x["mate_ref_id"] = r if x["mate_ref_id"] == x["ref_id"] else 1 - r
            x["ref_id"] = r
        recs += part
    recs.sort(key=lambda x: (x["ref_id"], x["pos"]))
    batch = R.from_records(recs)
    bed = {r: random_bed(rng, L, dense=(seed != 2))[0] for r, L in enumerate(ref_lens)}
    kw = [dict(min_base_qual=20, mapq=0), dict(clip=4, min_base_qual=10, ignore_strand=1),
          dict(strand_req=2, ignore_strand=1, mapq=20)][seed - 1]
    opts = O.make_opts(max_read_len=100, max_read_span=150, **kw)
    wr, wp, wv, woff, wf = O.pileup_per_read(opts, ref_lens, bed, batch)
    with snp.PileupJob(opts, ref_lens) as job:
        job.set_bed(bed)
        job.submit(batch)
        job.finish()
        assert_same(job.rows(), (wr, wp, wv), "rows")
        off, feats = job.per_read()
        assert (off == woff).all()
        assert feats.tobytes() == wf.tobytes()
        assert len(feats) > 2000
        # a sub-range of rows
        o2, f2 = job.per_read(100, 50)
        lo, hi = int(woff[400]), int(woff[600])
        assert (o2 == woff[400:601] - woff[400]).all() and f2.tobytes() == wf[lo:hi].tobytes()
    sopts = O.make_opts(max_read_len=100, max_read_span=150, stitch=1)
    with snp.PileupJob(sopts, ref_lens) as job:
        job.set_bed(bed)
        job.submit(batch)
        job.finish()
        with pytest.raises(snp.PileupError, match="stitched per-read features not yet supported"):
            job.per_read()


@pytest.mark.parametrize("case", ["deep_spot", "many_ops", "dense"])
def test_per_read_lists_deep_and_fragmented(case):
    """The per-read gather beyond its common case: a tile whose lists exceed the LDS staging
    area (entries stored directly), more (record, M op) pieces than one round's list holds,
    and uniformly deep tiles that fill the staging area; each against the oracle."""
    rng = random.Random({"deep_spot": 71, "many_ops": 72, "dense": 73}[case])
    L = 60_000 if case == "deep_spot" else 3_000
    recs = []

    def rec(k, pos, cg, lq):
        seq = "".join(rng.choice("ACGTACGTACGTACGTN") for _ in range(lq))
        return dict(name="r%d" % k, ref_id=0, pos=pos, mapq=60, flags=0x1 | 0x2 | 0x20 | 0x40, cigar=cg, seq=seq,
                    qual=[rng.choice([2, 12, 23, 37, 41]) for _ in range(lq)], mate_ref_id=0, mate_pos=pos + 100)
    if case == "deep_spot":
        # a thin background and 900 reads piled on ~150 positions: the staging area is sized
        # for the average tile, so the spot's tiles take the direct path
        for k in range(300):
            recs.append(rec(k, rng.randint(0, L - 200), [("M", 80)], 80))
        for k in range(900):
            a = rng.randint(10, 60)
            recs.append(rec(1000 + k, 30_000 + rng.randint(0, 60), [("S", 3), ("M", a), ("D", 2), ("M", 70 - a)], 73))
    elif case == "many_ops":
        # 1M1D / 2M1I ladders: up to 30 M ops under one 32-position tile, 200 such records on
        # the same stretch -> several rounds of the piece list per batch of 64 records
        for k in range(260):
            cg, lq = [], 0
            for _ in range(rng.randint(20, 40)):
                m = rng.randint(1, 3)
                cg.append(("M", m))
                lq += m
                if rng.random() < 0.5:
                    cg.append(("D", rng.randint(1, 2)))
                else:
                    i = rng.randint(1, 2)
                    cg.append(("I", i))
                    lq += i
            cg.append(("M", 5))
            lq += 5
            recs.append(rec(k, 500 + rng.randint(0, 120), cg, lq))
    else:
        for k in range(4000):
            a = rng.randint(5, 90)
            cg = [("M", 100)] if k % 3 else [("M", a), ("I", 2), ("M", 98 - a)]
            recs.append(rec(k, rng.randint(0, L - 150), cg, 100))
    recs.sort(key=lambda x: x["pos"])
    batch = R.from_records(recs)
    bed = {0: [[0, L]]} if case != "many_ops" else {0: [[400, 431], [433, 700], [701, 702], [800, 900]]}
    opts = O.make_opts(max_read_len=200, max_read_span=300, min_base_qual=13, clip=2 if case == "dense" else 0)
    wr, wp, wv, woff, wf = O.pileup_per_read(opts, [L], bed, batch)
    with snp.PileupJob(opts, [L]) as job:
        job.set_bed(bed)
        job.submit(batch)
        job.finish()
        assert_same(job.rows(), (wr, wp, wv), "rows")
        off, feats = job.per_read()
        assert (off == woff).all()
        assert feats.tobytes() == wf.tobytes()
        assert len(feats) > 5_000


@pytest.fixture(scope="module")
def chr1_30x():
    """BASELINE.json configs[1]: synthetic 30x chr1 (what bench.py times), generated once for
    the full-size tests of this module (names included: the stitched run keys on them)."""
    from bench import CHR1_LEN, kept_aligned_bases
    sb = synth.generate([(0, 0, CHR1_LEN - 1000, 30.0)], threads=16, with_names=True)
    rb = sb.to_record_batch()
    B = kept_aligned_bases(rb, dict(flag_exclude=0xF00, mapq=60))
    yield CHR1_LEN, rb, B
    sb.close()


FULL_WINDOWS = (1_000, 123_456_789, 248_956_422 - 150_000)


def _oracle_windows(job, opts, L, rb, pad=600):
    """Three windows of 100 kb far apart in the contig against the oracle run on just the reads
    that can reach them (with stitching: and their mates, two spans away)."""
    for w0 in FULL_WINDOWS:
        lo = int(np.searchsorted(rb.pos, w0 - pad))
        hi = int(np.searchsorted(rb.pos, w0 + 100_000 + pad))
        _, wp, wv = O.pileup(opts, [L], {0: [[w0, w0 + 100_000]]}, rb.slice(lo, hi))
        _, pos, vals = job.rows(w0, 100_000)
        assert (pos == wp).all() and (vals == wv).all(), w0


def test_full_size_chr1_30x_properties(chr1_30x):
    """BASELINE.json configs[1] at its full size (chr1, 248,956,422 positions, 30x, per-strand
    counts, DefaultOpts; what bench.py times), through properties that do not need the oracle
    at that size: depth = the aligned bases counted on the host, every base in a count slot,
    a second compute gives the same sums, linearity (the rows of two halves of the records add up
    to the rows of all of them, row by row), and three windows of rows far into the contig
    against the oracle run on just the reads that reach them."""
    L, rb, B = chr1_30x
    bed = {0: [[0, L]]}
    opts = O.make_opts()  # DefaultOpts, basestrand semantics (StrandNone dropped)
    with snp.PileupJob(opts, [L]) as job:
        job.set_bed(bed)
        job.submit(rb)
        job.finish()
        assert job.num_rows == L
        sums = job.column_sums()
        assert sums[0] == B
        assert sum(sums[1:]) == sums[0]
        job.compute()
        assert job.column_sums() == sums
        # all 11 column sums of the oracle's rows over the WHOLE contig (16 position ranges)
        want_sums, _ = O.whole_contig_column_sums(opts, L, rb, 16)
        assert want_sums == sums
        _oracle_windows(job, opts, L, rb)
        # linearity: halves of the records, in two jobs, merged on the device
        half = rb.n // 2
        with snp.PileupJob(opts, [L]) as ja, snp.PileupJob(opts, [L]) as jb:
            for j, part in ((ja, rb.slice(0, half)), (jb, rb.slice(half, rb.n))):
                j.set_bed(bed)
                j.submit(part)
                j.finish()
            ptr = C.c_void_p()
            jb._check(jb.lib.bp_rows_device(jb.h, C.byref(ptr)))
            ja.rows_accumulate(0, L, ptr.value)
            assert ja.column_sums() == sums
            step = 20_000_000
            for r0 in range(0, L, step):
                n = min(step, L - r0)
                assert (ja.rows(r0, n)[2] == job.rows(r0, n)[2]).all(), r0


def test_full_size_c3_one_ranks_share_rank3_of_8():
    """BASELINE.json configs[2] (3.1 Gb at 30x over 8 GPUs), ONE rank's share at its full size on
    the one GPU there is: rank 3 of 8 owns 386 M positions from the middle of chr6 to the first
    7.6 Mb of chr9 - three contig boundaries inside the share, a cut inside a contig at either
    end - and is given the ~77 M reads that start there (partition.boundary_share, the code
    bench.py --config c3 runs).  Depth over own + tail rows = the aligned bases counted on the
    host; the 1,024 tail rows that would travel to rank 4 come out of bp_rows_export as bp_rows
    has them, hold what the rank's reads put behind its end, and equal the oracle's; oracle rows
    on 100 kb windows either side of every contig boundary and at both ends of the share."""
    import torch
    from bench import HG38_LENS, kept_aligned_bases
    from bio_amd import partition as P
    world, rank, tail = 8, 3, 1024
    gbed = {r: [[0, l]] for r, l in enumerate(HG38_LENS)}
    share = P.boundary_share(gbed, world, rank, tail)
    ranges = share["ranges"]
    assert [r for r, _, _ in ranges] == [5, 6, 7, 8] and ranges[0][1] > 0 and ranges[-1][2] < HG38_LENS[8]
    assert share["spill"] and share["tail_rows"] == tail and share["own_rows"] == sum(HG38_LENS) * 4 // 8 - sum(HG38_LENS) * 3 // 8
    last = ranges[-1]
    regions = [(r, s, (e if (r, s, e) == last else e - 1000), 30.0) for r, s, e in ranges]
    sb = synth.generate(regions, seed=synth.SEED + rank, threads=16)
    rb = sb.to_record_batch()
    key = rb.ref_id.astype(np.int64) << 32 | rb.pos.astype(np.int64)
    rb = rb.slice(int(np.searchsorted(key, (ranges[0][0] << 32) | ranges[0][1], "left")), rb.n)
    key = rb.ref_id.astype(np.int64) << 32 | rb.pos.astype(np.int64)
    assert 70_000_000 < rb.n < 85_000_000
    B = kept_aligned_bases(rb, dict(flag_exclude=0xF00, mapq=60))
    opts = O.make_opts()
    row0 = {}  # first row of each range
    acc = 0
    for r, s, e in ranges:
        row0[r] = (acc, s)
        acc += e - s
    with snp.PileupJob(opts, HG38_LENS) as job:
        job.set_bed(share["bed"])
        job.reserve(rb.n, len(rb.cigar) * 4 + len(rb.seq) + len(rb.qual))
        job.submit(rb)
        job.finish()
        own, n_tail = share["own_rows"], share["tail_rows"]
        assert job.num_rows == own + n_tail
        sums = job.column_sums()
        assert sums[0] == B and sum(sums[1:]) == B

        def check(ref, w0, w1):
            lo = int(np.searchsorted(key, (ref << 32) | max(0, w0 - 600), "left"))
            hi = int(np.searchsorted(key, (ref << 32) | w1, "left"))
            wr, wp, wv = O.pileup(opts, HG38_LENS, {ref: [[w0, w1]]}, rb.slice(lo, hi))
            base, s = row0[ref]
            gr, gp, gv = job.rows(base + w0 - s, w1 - w0)
            assert (gr == wr).all() and (gp == wp).all() and (gv == wv).all(), (ref, w0)
            return gv
        # both ends of the share (the first rows lack what rank 2's reads would add: so does the oracle here)
        first = check(ranges[0][0], ranges[0][1], ranges[0][1] + 100_000)
        assert first[:100, 0].mean() < first[1000:, 0].mean()
        check(last[0], last[2] - 100_000, last[2])
        # either side of the three contig boundaries
        for r, s, e in ranges[:-1]:
            v = check(r, e - 100_000, e)
            assert v[-450:].sum() == 0 and v[:90_000, 0].min() > 0  # pairs start 1,000 or more short of the contig's end
        for r, s, e in ranges[1:]:
            assert s == 0
            check(r, 0, 100_000)
        # the tail: rows [own, own + 1024) = chr9 positions behind the share's end
        buf = torch.zeros(tail * 11, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        job.rows_export(own, n_tail, buf.data_ptr())
        exported = buf.cpu().numpy().view(np.uint32).reshape(tail, 11)
        tr, tp, tv = job.rows(own, n_tail)
        assert (exported == tv).all() and (tr == last[0]).all() and tp[0] == last[2] and tp[-1] == last[2] + tail - 1
        lo = int(np.searchsorted(key, (last[0] << 32) | (last[2] - 600), "left"))
        _, _, wv = O.pileup(opts, HG38_LENS, {last[0]: [[last[2], last[2] + tail]]}, rb.slice(lo, rb.n))
        assert (tv == wv).all()
        assert tv[:100, 0].mean() > 5 and tv[700:].sum() == 0  # the reads reach < 1,024 behind the end
    sb.close()


@pytest.mark.parametrize("mode", ["minq20", "minq20_clip5", "stitch_minq20"])
def test_full_size_chr1_30x_other_kernel_variants(chr1_30x, mode):
    """The same contig through the other instantiations of the pileup kernel (quality rule,
    quality rule + clip, stitched), at full size: Depth counts every aligned base whatever its
    quality (pileup.go:223-231), so the depth sum is the host's count (stitched: between half of it
    and all of it, overlaps count once); the count slots can only lose bases against DefaultOpts,
    column by column; a second compute reproduces the sums; three far-apart windows equal the
    oracle's rows.  Stitched also holds the reference's own end-to-end property
    DP_stitched <= DP_unstitched <= 2 DP_stitched, row by row (cmd/bio-pileup/pileup_snp_test.go:85-93)."""
    L, rb, B = chr1_30x
    bed = {0: [[0, L]]}
    kw = dict(minq20=dict(min_base_qual=20), minq20_clip5=dict(min_base_qual=20, clip=5),
              stitch_minq20=dict(stitch=1, min_base_qual=20))[mode]
    opts = O.make_opts(**kw)
    with snp.PileupJob(O.make_opts(), [L]) as base, snp.PileupJob(opts, [L]) as job:
        for j in (base, job):
            j.set_bed(bed)
            j.submit(rb)
            j.finish()
        s0, s1 = base.column_sums(), job.column_sums()
        assert s0[0] == B
        if kw.get("stitch"):
            assert B // 2 <= s1[0] < B
        else:
            assert s1[0] == B
        assert sum(s1[1:]) < s1[0]          # some bases fail the quality rule: Depth only
        assert all(a <= b for a, b in zip(s1[1:9], s0[1:9]))
        job.compute()
        assert job.column_sums() == s1
        _oracle_windows(job, opts, L, rb, pad=1200 if kw.get("stitch") else 600)
        if kw.get("stitch"):
            step = 20_000_000
            for r0 in range(0, L, step):
                n = min(step, L - r0)
                d1, d0 = job.rows(r0, n)[2][:, 0], base.rows(r0, n)[2][:, 0]
                assert (d1 <= d0).all() and (d0 <= 2 * d1).all(), r0


def _c4_panel():
    """BASELINE.json configs[3] (C4): 500 targets x 200 bp on one contig."""
    rng = random.Random(4)
    targets, p = [], 10_000
    for _ in range(500):
        targets.append([p, p + 200])
        p += 200 + rng.randint(300, 3000)
    return targets, p + 10_000


def test_full_size_c4_deep_panel_stitched():
    """BASELINE config 4 at its stated size: 500 targets x 200 bp at ~10,000x, 2x150 mates with
    insert ~N(170,20) so that every pair overlaps, -stitch -min-base-qual 20 -mapq 60 (16.7 M
    records, every window deep: the sliced second launch with its plane copies does all the
    work).  Properties over the whole panel (depth between half and all of the kept aligned bases,
    a second compute reproduces the sums, stitched depth against the unstitched rows as in
    pileup_snp_test.go:85-93) and the oracle's rows on 12 whole targets (first, middle, last 4)."""
    from bench import kept_aligned_bases
    targets, L = _c4_panel()
    bed = {0: targets}
    sb = synth.generate([(0, s, e, 10000.0) for s, e in targets], with_names=True, threads=16,
                        insert_mean=170.0, insert_sd=20.0, insert_min=100, insert_max=300)
    rb = sb.to_record_batch()
    assert rb.n > 15_000_000
    opts = O.make_opts(stitch=1, min_base_qual=20, mapq=60)
    with snp.PileupJob(O.make_opts(mapq=60), [L]) as base, snp.PileupJob(opts, [L]) as job:
        for j in (base, job):
            j.set_bed(bed)
            j.submit(rb)
            j.finish()
        assert job.num_rows == 500 * 200
        s0, s1 = base.column_sums(), job.column_sums()
        assert s0[0] > 500 * 200 * 5000
        assert s0[0] // 2 <= s1[0] < s0[0]
        job.compute()
        assert job.column_sums() == s1
        _, pos, v1 = job.rows()
        v0 = base.rows()[2]
        assert (v1[:, 0] <= v0[:, 0]).all() and (v0[:, 0] <= 2 * v1[:, 0]).all()
        assert v1[:, 0].min() > 2000
        for t0 in (0, 248, 496):
            sub = targets[t0:t0 + 4]
            lo = int(np.searchsorted(rb.pos, sub[0][0] - 1200))
            hi = int(np.searchsorted(rb.pos, sub[-1][1] + 1200))
            _, wp, wv = O.pileup(opts, [L], {0: sub}, rb.slice(lo, hi))
            r0 = 200 * t0
            assert (pos[r0:r0 + 800] == wp).all() and (v1[r0:r0 + 800] == wv).all(), t0
    sb.close()


def test_full_size_c5_60x_chr1_per_read_lists():
    """BASELINE config 5 at its size on one GPU: chr1 at 60x with the ordered per-read lists of
    -cols dpref,quals (appendBase, pileup.go:235-250; 99.6 M records, ~12 G list entries, ~120 GB
    resident).  Over everything: the list offsets end at the sum of the A,C,G,T counts (a list's
    length is its count), and the build checks on the device that every list ends where the scan
    put it.  Lists of three 100 kb windows against the oracle's, entry for entry.  (PAM input, the
    other half of config 5, is not built: INTEGRATION.md.)"""
    from bench import CHR1_LEN
    L = CHR1_LEN
    sb = synth.generate([(0, 0, L - 1000, 60.0)], threads=16)
    rb = sb.to_record_batch()
    opts = O.make_opts(min_base_qual=13)
    with snp.PileupJob(opts, [L]) as job:
        job.set_bed({0: [[0, L]]})
        job.submit(rb)
        job.finish()
        sums = job.column_sums()
        job._check(job.lib.bp_per_read_build(job.h))
        n_feat = int(job.lib.bp_per_read_total(job.h))
        assert n_feat == sum(sums[1:9]) and n_feat > 10_000_000_000
        off_end = np.zeros(5, np.uint64)
        job._check(job.lib.bp_per_read_offsets(job.h, L - 1, 1, off_end.ctypes.data_as(C.POINTER(C.c_uint64))))
        assert int(off_end[-1]) == n_feat
        for w0 in FULL_WINDOWS:
            lo = int(np.searchsorted(rb.pos, w0 - 600))
            hi = int(np.searchsorted(rb.pos, w0 + 100_000))
            _, wp, wv, woff, wfeat = O.pileup_per_read(opts, [L], {0: [[w0, w0 + 100_000]]}, rb.slice(lo, hi))
            _, pos, vals = job.rows(w0, 100_000)
            assert (pos == wp).all() and (vals == wv).all(), w0
            off, feats = job.per_read(w0, 100_000)
            assert (off == woff).all() and feats.tobytes() == wfeat.tobytes(), w0
    sb.close()


def test_rows_to_text_on_device():
    """bp_render_text against the same lines formatted here from bp_rows: basestrand-tsv
    (ConvertPileupRowsToBasestrandTSV, output.go:376-453: chrom, 1-based pos, REF from
    Seq8ToASCIITable, the eight per-strand counts) and the .ref.tsv / .alt.tsv files of the tsv
    format (output.go:149-276) under several -cols sets, over several BED intervals of two
    references, in sub-ranges that straddle the 1,024-row tiles of the length scan."""
    rng = random.Random(77)
    ref_lens = [6000, 3500]
    names = ["chr2_subset", "chrUn_KI270442v1"]
    recs = []
    for r, L in enumerate(ref_lens):
        part = random_records(rng, 500, ref_len=L, read_len=80)
        for x in part:
            x["mate_ref_id"] = r if x["mate_ref_id"] == x["ref_id"] else 1 - r
            x["ref_id"] = r
        recs += part
    recs.sort(key=lambda x: (x["ref_id"], x["pos"]))
    batch = R.from_records(recs)
    bed = {0: [[10, 2500], [2600, 2601], [3000, 5990]], 1: [[0, 1200], [1500, 3400]]}
    opts = O.make_opts(max_read_len=100, max_read_span=150, min_base_qual=12)
    ascii_of, enum_of = "=ACMGRSVTWYHKDBN", {1: 0, 2: 1, 4: 2, 8: 3}

    def lines(kind, cols, name, p, b8, v):
        head = "%s\t%d\t%s" % (name, p + 1, ascii_of[b8])
        if kind == 0:
            return [head + "\t" + "\t".join(str(int(c)) for c in v[1:9]) + "\n"]
        rb = enum_of.get(b8, 4)
        out = []
        for alt in ([rb] if kind == 1 else [a for a in range(5) if a != rb]):
            c = int(v[1 + 2 * alt]) + int(v[2 + 2 * alt])
            if kind == 2 and c == 0:
                continue
            f = [head] + (["ACGTN"[alt]] if kind == 2 else [])
            if cols & (1 if kind == 1 else 2):
                f.append(str(int(v[0])))
            if cols & 64:
                f.append(str(c))
            if cols & 128:
                f.append("0")
            out.append("\t".join(f) + "\n")
        return out

    with snp.PileupJob(opts, ref_lens, names) as job:
        job.set_bed(bed)
        job.submit(batch)
        job.finish()
        ref, pos, vals = job.rows()
        assert vals[:, 0].max() > 10
        n0 = int((ref == 0).sum())
        for r, (lo, hi) in enumerate([(0, n0), (n0, len(pos))]):
            seq = np.array([rng.choice([1, 2, 4, 8, 15, 5, 0]) for _ in range(ref_lens[r])], np.uint8)
            start = int(pos[lo])
            job.set_ref_seq8(r, start, seq[start:int(pos[hi - 1]) + 1])
            for kind, cols in [(0, 0), (1, 1 | 64 | 128), (2, 2 | 64 | 128), (1, 0), (2, 0), (1, 64), (2, 2), (2, 1 | 128)]:
                per_row = [lines(kind, cols, names[r], int(pos[i]), int(seq[pos[i]]), vals[i]) for i in range(lo, hi)]
                want = "".join("".join(x) for x in per_row).encode()
                assert job.render_text(lo, hi - lo, kind, cols) == want, (r, kind, cols)
                # the same lines as BGZF members from the device (bp_render_fetch_bgzf), both ways of calling it
                import gzip
                for how in ("one", "two"):
                    z = job.render_text(lo, hi - lo, kind, cols, bgzf=how)
                    assert (gzip.decompress(z) if want else z) == want, (r, kind, cols, how)
                    assert len(z) < len(want) or len(want) < 200
                # a sub-range that starts and ends inside tiles of the scan
                a, b = lo + 700, min(hi, lo + 2900)
                assert job.render_text(a, b - a, kind, cols) == "".join("".join(x) for x in per_row[a - lo:b - lo]).encode()
            assert job.render_text(lo, 0) == b""
            # rows of the other reference have no base to show
            other = (n0, 5) if r == 0 else (0, 5)
            with pytest.raises(snp.PileupError, match="no reference base"):
                job.render_text(*other)
        with pytest.raises(snp.PileupError, match="per-read columns"):
            job.render_text(n0, 5, 1, 8)


def _qual_pair_batch():
    """96 x 96 two-read pairs, one per (i, j): 4M reads at P and P+3 that agree on the one position
    they share, where the stored read has quality i and the read that finds it quality j."""
    n_pair = 96 * 96
    n = 2 * n_pair
    k = np.arange(n_pair)
    P = 16 * k + 8
    first = np.arange(n) % 2 == 0
    pos = np.where(first, np.repeat(P, 2), np.repeat(P, 2) + 3).astype(np.int32)
    mate_pos = np.where(first, np.repeat(P, 2) + 3, np.repeat(P, 2)).astype(np.int32)
    qual = np.full((n, 4), 30, np.uint8)
    qual[0::2, 3] = k // 96   # i: last base of the stored read
    qual[1::2, 0] = k % 96    # j: first base of its mate
    names = np.frombuffer("".join("p%05d" % (x // 2) for x in range(n)).encode(), np.uint8)
    return R.RecordBatch(
        ref_id=np.zeros(n, np.int32), pos=pos, mate_ref_id=np.zeros(n, np.int32), mate_pos=mate_pos,
        flags=np.where(first, 0x63, 0x93).astype(np.uint16), mapq=np.full(n, 60, np.uint8),
        n_cigar=np.ones(n, np.uint32), l_seq=np.full(n, 4, np.uint32), cigar_off=np.arange(n, dtype=np.uint64),
        cigar=np.full(n, (4 << 4) | 0, np.uint32), seq_off=2 * np.arange(n, dtype=np.uint64),
        seq=np.full(2 * n, 0x11, np.uint8), qual_off=4 * np.arange(n, dtype=np.uint64), qual=qual.reshape(-1),
        name_off=6 * np.arange(n + 1, dtype=np.uint64), name=names), P


def test_qual_sum_table_exhaustive_through_the_device():
    """qualSumTable / qualPassTable (pileup/snp/qual.go:35-81) through the stitched device path,
    every (minBaseQual, q0, q1): the kernel works from a per-row threshold the library derives
    with its own copy of the formula, so a wrong entry anywhere would show here.  The oracle's
    table is the float64 restatement pinned by qual_internal_test.go:46-54."""
    batch, P = _qual_pair_batch()
    L = int(P[-1]) + 32
    bed = {0: [[0, L]]}
    n_pass_prev = None
    for minq in range(1, 96):
        opts = O.make_opts(stitch=1, min_base_qual=minq)
        want = O.pileup(opts, [L], bed, batch)
        got = device_rows(opts, [L], bed, [batch])
        assert_same(got, want, f"minBaseQual {minq}")
        # the shared position of pair (i, j) holds A+ = 1 iff qualSum[i][j] >= minBaseQual
        a_plus = got[2][P + 3, 1]
        depth = got[2][P + 3, 0]
        assert (depth == 1).all()
        lib = O.load()
        tab = np.array([[lib.orc_qual_pass(minq, i, j) for j in range(96)] for i in range(96)], np.uint32) \
            if minq in (1, 20, 43, 95) else None
        if tab is not None:
            assert (a_plus.reshape(96, 96) == tab).all(), minq
        n_pass = int(a_plus.sum())
        assert n_pass_prev is None or n_pass <= n_pass_prev  # raising the bar never passes more pairs
        n_pass_prev = n_pass
    assert 0 < n_pass_prev < 96 * 96
