"""The reference's mate-table tests restated as properties of whole pileups.

pileup/snp/firstread_internal_test.go:26-160 (TestFirstreadSNPTable) feeds firstreadSNPTable
10,000 pairs closer than maxReadSpan, 500 singletons and 5,000 pairs further apart and
requires addOrRemove to pair exactly the close ones; pileup_internal_test.go:165-297
(TestAddOrphanReads) requires a stored read whose mate never comes to be handled as an orphan.
Both test unexported Go functions with Go's math/rand; what is restated here is what those
decisions mean for the rows: with -stitch, the positions a paired read and its mate both cover
get Depth + 1, everything else (singletons, far pairs, reads whose mate was filtered out)
counts as single reads.  Same shapes as the reference's tests: 2,048 start positions,
CIGAR 160M95D (span 255 = max-read-span), flags 0x61 / 0x91, singletons with no mate
reference.  The expected depth comes from interval arithmetic, not from the oracle.
"""
import random

import numpy as np
import pytest

import oracle_lib as O
from bio_amd import records as R

N_POS, READ_LEN, SPAN = 2048, 160, 255
REF_LEN = N_POS + SPAN + 10


def build_case(seed, n_pair=10_000, n_singleton=500, n_far=5_000, n_orphan=1_000):
    rng = random.Random(seed)
    seq, qual = "A" * READ_LEN, [40] * READ_LEN
    cigar = [("M", READ_LEN), ("D", SPAN - READ_LEN)]
    recs = []
    depth = np.zeros(REF_LEN + 1, np.int64)

    def rec(name, pos, flags, mate_pos, mate_ref=0, mapq=60):
        return dict(name=name, ref_id=0, pos=pos, mapq=mapq, flags=flags, cigar=cigar, seq=seq, qual=qual,
                    mate_ref_id=mate_ref, mate_pos=mate_pos)

    def cover(*starts):  # Depth + 1 on the union of the reads' M bases
        iv = sorted((s, s + READ_LEN) for s in starts)
        end = -1
        for s, e in iv:
            s = max(s, end)
            if e > s:
                depth[s] += 1
                depth[e] -= 1
                end = e

    for i in range(n_pair):  # firstread_internal_test.go:45-76
        p1 = rng.randrange(N_POS)
        p2 = min(max(p1 + rng.randint(-(SPAN - 1), SPAN - 1), 0), N_POS - 1)
        recs += [rec("pair%d" % i, p1, 0x61, p2), rec("pair%d" % i, p2, 0x91, p1)]
        cover(p1, p2)
    for i in range(n_singleton):  # :78-90 (MateRef nil)
        p = rng.randrange(N_POS)
        recs.append(rec("singleton%d" % i, p, 0x60, -1, mate_ref=-1))
        cover(p)
    for i in range(n_far):  # :91-121
        p1 = rng.randrange(N_POS)
        gap = SPAN + rng.randrange(N_POS // 2 - SPAN)
        p2 = p1 - gap if p1 >= N_POS // 2 else p1 + gap
        recs += [rec("farpair%d" % i, p1, 0x61, p2), rec("farpair%d" % i, p2, 0x91, p1)]
        cover(p1)
        cover(p2)
    for i in range(n_orphan):  # a close pair whose mate is filtered out (MAPQ) or absent: an orphan
        p1 = rng.randrange(N_POS)
        p2 = min(max(p1 + rng.randint(-(SPAN - 1), SPAN - 1), 0), N_POS - 1)
        recs.append(rec("orphan%d" % i, p1, 0x61, p2))
        if i % 2:
            recs.append(rec("orphan%d" % i, p2, 0x91, p1, mapq=3))
        cover(p1)
    order = sorted(range(len(recs)), key=lambda k: recs[k]["pos"])  # stable: ties keep arrival order
    return R.from_records([recs[k] for k in order]), np.cumsum(depth)[:REF_LEN]


def _opts(stitch):
    # ignore_strand (the tsv formats) keeps the singletons, whose strand is StrandNone (common.go:95)
    return O.make_opts(stitch=stitch, ignore_strand=1, mapq=60, max_read_len=READ_LEN, max_read_span=SPAN,
                       min_base_qual=10)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_oracle_pairs_exactly_the_close_mates(seed):
    batch, want_depth = build_case(seed)
    bed = {0: [[0, REF_LEN]]}
    ref, pos, vals = O.pileup(_opts(1), [REF_LEN], bed, batch)
    assert (vals[:, 0] == want_depth).all()
    # agreeing overlaps of q40 bases pass the quality-sum rule: every depth is an A count
    assert (vals[:, 1] == want_depth).all() and vals[:, 2:].sum() == 0
    # and without -stitch every read counts by itself
    _, _, v0 = O.pileup(_opts(0), [REF_LEN], bed, batch)
    assert (v0[:, 0] >= vals[:, 0]).all() and (v0[:, 0] <= 2 * vals[:, 0]).all()  # cmd/bio-pileup/pileup_snp_test.go:85-93


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2])
def test_device_pairs_exactly_the_close_mates(seed):
    from bio_amd import snp
    batch, want_depth = build_case(seed)
    bed = {0: [[0, REF_LEN]]}
    ref, pos, vals = snp.run_job(_opts(1), [REF_LEN], bed, [batch])
    assert (vals[:, 0] == want_depth).all()
    assert (vals[:, 1] == want_depth).all() and vals[:, 2:].sum() == 0
    wr, wp, wv = O.pileup(_opts(1), [REF_LEN], bed, batch)
    assert (vals == wv).all()
