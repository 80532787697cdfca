"""Cross-checks the C oracle against the independent naive Python statement of the same
semantics on seeded random inputs (the oracle must not be
self-certifying), and checks the oracle's shard/job invariance."""
import os
import random
import sys

import numpy as np
import pytest

import oracle_lib as O
from bio_amd import records as R

sys.path.insert(0, os.path.join(O.ROOT, "oracle"))
import naive_pileup as N  # noqa: E402


def random_records(rng, n_pairs, ref_len=3000, read_len=40, weird=True, n_ref=1):
    recs = []
    for k in range(n_pairs):
        ref = rng.randrange(n_ref)
        L = rng.randint(5, read_len)
        s = rng.randint(0, ref_len - 3 * read_len)
        ins = rng.randint(L, 2 * read_len + 10)
        first_r1 = rng.random() < 0.5

        def mk(pos, is_first):
            r = rng.random()
            if not weird or r < 0.6:
                cg, lq = [("M", L)], L
            elif r < 0.7:
                a = rng.randint(1, L - 2)
                cg, lq = [("S", a), ("M", L - a)], L
            elif r < 0.8:
                a = rng.randint(1, L - 2)
                cg, lq = [("M", a), ("D", rng.randint(1, 8)), ("M", L - a)], L
            elif r < 0.9:
                a = rng.randint(1, max(1, L - 3))
                i = rng.randint(1, max(1, L - a - 1))
                if L - a - i < 1:
                    cg, lq = [("M", L)], L
                else:
                    cg, lq = [("M", a), ("I", i), ("M", L - a - i)], L
            elif r < 0.95:
                a = rng.randint(1, L - 2)
                cg, lq = [("H", 3), ("M", a), ("N", rng.randint(1, 20)), ("M", L - a), ("S", 0)], L
            else:
                cg, lq = [("H", 2), ("S", 1), ("M", L - 1), ("H", 4)] if L > 1 else [("M", L)], L
            fl = 0x1 | 0x2
            if is_first:
                fl |= 0x20 | (0x40 if first_r1 else 0x80)
            else:
                fl |= 0x10 | (0x80 if first_r1 else 0x40)
            x = rng.random()
            if x < 0.03:
                fl |= 0x400
            elif x < 0.05:
                fl |= 0x800
            elif x < 0.08:
                fl ^= 0x30  # malformed orientation -> other strand / none
            elif x < 0.10:
                fl = (fl | 0x8) & ~0xC0
            seq = "".join(rng.choice("ACGTACGTACGTN") for _ in range(lq))
            if rng.random() < 0.05:
                seq = seq[:-1] + "R"  # ambiguity code -> X
            return dict(name="q%d" % k, ref_id=ref, pos=pos, mapq=rng.choice([60, 60, 60, 60, 20]),
                        flags=fl, cigar=cg, seq=seq,
                        qual=[rng.choice([2, 12, 23, 37, 41, 95]) for _ in range(lq)],
                        mate_ref_id=ref if rng.random() > 0.03 else ref + 1)

        a = mk(s, True)
        b = mk(min(s + max(0, ins - L), ref_len - read_len - 30), False)
        if rng.random() < 0.05:
            b["pos"] = a["pos"]
        a["mate_pos"], b["mate_pos"] = b["pos"], a["pos"]
        keep = rng.random()
        if keep < 0.9:
            recs += [a, b]
        elif keep < 0.95:
            recs.append(a)
        else:
            recs.append(b)
    recs.sort(key=lambda r: (r["ref_id"], r["pos"]))
    return recs


def random_bed(rng, ref_len, n_ref=1, dense=False):
    bed = {}
    for r in range(n_ref):
        iv, p = [], rng.randint(0, 50)
        while p < ref_len - 10:
            ln = rng.randint(1, 400 if dense else 60)
            e = min(p + ln, ref_len)
            iv.append([p, e])
            p = e + rng.randint(1, 30 if dense else 200)
        bed[r] = iv
    return bed


def rows_to_dict(ref, pos, vals):
    return {(int(r), int(p)): [int(x) for x in v] for r, p, v in zip(ref, pos, vals)}


CASES = [
    dict(stitch=0, ignore_strand=0, min_base_qual=0, clip=0),
    dict(stitch=0, ignore_strand=1, min_base_qual=20, clip=3),
    dict(stitch=0, ignore_strand=1, strand_req=1, min_base_qual=13, clip=0),
    dict(stitch=0, ignore_strand=1, strand_req=2, min_base_qual=13, clip=0),
    dict(stitch=1, ignore_strand=0, min_base_qual=43, clip=0),
    dict(stitch=1, ignore_strand=1, min_base_qual=30, clip=2),
    dict(stitch=1, ignore_strand=0, min_base_qual=0, clip=0, mapq=0, flag_exclude=0),
]


@pytest.mark.parametrize("seed", [1, 2, 3])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_oracle_matches_naive(seed, case):
    rng = random.Random(1000 * case + seed)
    kw = dict(CASES[case])
    ref_len = 3000
    recs = random_records(rng, 150, ref_len=ref_len, n_ref=2)
    batch = R.from_records(recs)
    bed = random_bed(rng, ref_len, n_ref=2, dense=(seed % 2 == 0))
    opts = O.make_opts(max_read_len=100, max_read_span=120, **kw)
    ref, pos, vals = O.pileup(opts, [ref_len, ref_len], bed, batch, parallelism=1)
    got = rows_to_dict(ref, pos, vals)
    want = N.pileup(opts, bed, batch)
    assert len(got) == len(want) == O.n_bed_positions(bed)
    assert list(zip(ref, pos)) == sorted(want.keys())
    bad = [(k, got[k], want[k]) for k in want if got[k] != want[k]]
    assert not bad, bad[:5]


@pytest.mark.parametrize("stitch", [0, 1])
def test_oracle_job_invariance(stitch):
    """-bed mode: results do not depend on how the 100-kb shards are grouped into jobs
    (halo = max_read_span, exact BED Subset per job; pileup.go:896-909, :1073-1084)."""
    rng = random.Random(77 + stitch)
    ref_len = 6000
    recs = random_records(rng, 600, ref_len=ref_len, weird=True)
    # well-formed strands only: a pair whose reads report different strands is the one
    # documented job-boundary dependence of the reference
    batch = R.from_records(recs)
    bed = random_bed(rng, ref_len, dense=True)
    opts = O.make_opts(max_read_len=100, max_read_span=120, stitch=stitch, min_base_qual=20)
    base = None
    for shard_size, par in [(100000, 1), (1000, 1), (1000, 3), (700, 6), (128, 47)]:
        shards, n = O.position_shards([ref_len], shard_size, opts.max_read_span)
        ref, pos, vals = O.pileup(opts, [ref_len], bed, batch, shards=shards, n_shards=n,
                                  parallelism=par)
        if base is None:
            base = vals.copy()
            base_pos = pos.copy()
        else:
            assert (pos == base_pos).all()
            diff = np.nonzero((vals != base).any(axis=1))[0]
            if stitch == 0:
                assert len(diff) == 0, (shard_size, par, pos[diff][:5])
            else:
                # strand slot may move for malformed-flag pairs straddling a job boundary;
                # depth and per-base totals never do
                assert (vals[:, 0] == base[:, 0]).all()
                tot = vals[:, 1:].reshape(-1, 5, 2).sum(axis=2)
                btot = base[:, 1:].reshape(-1, 5, 2).sum(axis=2)
                assert (tot == btot).all()


def test_region_mode_quirk():
    """-region: one unpadded shard (pileup.go:1091-1097): reads starting before the region
    start are never seen even when they overlap it."""
    rng = random.Random(5)
    ref_len = 3000
    recs = random_records(rng, 300, ref_len=ref_len, weird=False)
    batch = R.from_records(recs)
    bed = {0: [[1000, 2000]]}
    for stitch in (0, 1):
        opts = O.make_opts(max_read_len=100, max_read_span=120, stitch=stitch)
        shards, n = O.region_shard(0, 1000, 2000)
        ref, pos, vals = O.pileup(opts, [ref_len], bed, batch, shards=shards, n_shards=n)
        want = N.pileup(opts, bed, batch, region_filter=(0, 1000, 2000))
        assert rows_to_dict(ref, pos, vals) == want
        # and it differs from -bed mode on the same interval near the region start
        r2, p2, v2 = O.pileup(opts, [ref_len], bed, batch)
        assert v2[:30, 0].sum() > vals[:30, 0].sum()


def test_oracle_errors():
    """Data errors abort the job with the reference's wording (pileup.go:766-772, :359)."""
    base = dict(name="x", ref_id=0, pos=100, mapq=60, flags=0x63, mate_pos=150)
    opts = O.make_opts(max_read_len=20, max_read_span=30)
    bed = {0: [[0, 1000]]}
    long_read = dict(base, cigar=[("M", 21)], seq="A" * 21, qual=[30] * 21)
    with pytest.raises(O.OracleError, match="maxReadLen is 20"):
        O.pileup(opts, [1000], bed, R.from_records([long_read]))
    wide = dict(base, cigar=[("M", 5), ("N", 26), ("M", 5)], seq="A" * 10, qual=[30] * 10)
    with pytest.raises(O.OracleError, match="maxReadSpan is 30"):
        O.pileup(opts, [1000], bed, R.from_records([wide]))
    badop = dict(base, cigar=[("M", 5), ("=", 5)], seq="A" * 10, qual=[30] * 10)
    with pytest.raises(O.OracleError, match="unexpected CIGAR code"):
        O.pileup(opts, [1000], bed, R.from_records([badop]))
    # a bad op that starts at/after the end of the last overlapping BED interval is never reached
    ref, pos, vals = O.pileup(opts, [1000], {0: [[100, 105]]}, R.from_records([badop]))
    assert vals[:, 0].tolist() == [1] * 5
    # filtered reads never raise
    ref, pos, vals = O.pileup(opts, [1000], bed, R.from_records([dict(long_read, mapq=3)]))
    assert vals.sum() == 0
    with pytest.raises(O.OracleError, match="invalid clip"):
        O.pileup(O.make_opts(clip=250), [1000], bed, R.from_records([long_read]))
    with pytest.raises(O.OracleError, match="max-read-len"):
        O.pileup(O.make_opts(max_read_len=600), [1000], bed, R.from_records([long_read]))


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_oracle_per_read_lists_match_naive(seed):
    """appendBase (pileup.go:235-250): ordered per-read feature lists, oracle vs naive."""
    rng = random.Random(500 + seed)
    ref_len = 2500
    batch = R.from_records(random_records(rng, 120, ref_len=ref_len))
    bed = random_bed(rng, ref_len, dense=True)
    opts = O.make_opts(max_read_len=100, max_read_span=120, min_base_qual=20, clip=seed - 1, mapq=0)
    ref, pos, vals, off, feats = O.pileup_per_read(opts, [ref_len], bed, batch)
    # counts are unchanged by perReadNeeded
    r2, p2, v2 = O.pileup(opts, [ref_len], bed, batch)
    assert (pos == p2).all() and (vals == v2).all()
    want = {}
    N.pileup(opts, bed, batch, per_read=want)
    n_lists = 0
    for i in range(len(pos)):
        lists = want.get((int(ref[i]), int(pos[i])), [[], [], [], []])
        for b in range(4):
            got = feats[int(off[4 * i + b]):int(off[4 * i + b + 1])]
            assert [tuple(int(x) for x in f) for f in got] == lists[b], (int(pos[i]), b)
            # list length = passing non-X count (pileup.go:241-248)
            assert len(got) == int(vals[i, 1 + 2 * b]) + int(vals[i, 2 + 2 * b])
            n_lists += len(got) > 0
    assert n_lists > 500
    # stitched per-read features: the reference panics at the first merged pair
    sopts = O.make_opts(max_read_len=100, max_read_span=120, stitch=1, mapq=0)
    with pytest.raises(O.OracleError, match="stitched per-read features not yet supported"):
        O.pileup_per_read(sopts, [ref_len], bed, batch)
