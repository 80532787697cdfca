"""BamReader::NextChunk against BamReader::Next (host code, no device): the chunk walk cuts a
buffer into stretches, guesses a record start in each and checks every guess against the walk
before it, so it must always come out with exactly the records of the one-at-a-time walk -
also when the bytes are full of things that look like records, when records are longer than a
stretch, and when the file ends mid-record."""
import os
import random
import struct

import numpy as np
import pytest

from bio_amd import bamio, host, records as R


def _records(rng, n, l_lo, l_hi, decoys=False):
    recs, pos = [], 100
    for i in range(n):
        L = rng.randint(l_lo, l_hi)
        pos += rng.randint(0, 5)
        q = [rng.randint(2, 40) for _ in range(L)]
        if decoys and L >= 140:
            # the quality string holds what reads as the fixed block of a small record, so that
            # a scan for record starts finds it: block_size 40, ref 0, pos 7, l_read_name 2, ...
            # (three in a row: a guess is only taken when the two records behind it look right too)
            fake = (struct.pack("<IiiBBHHHIiii", 40, 0, 7, 2, 0, 0, 0, 0, 0, 0, 3, 0) + b"x\0" + b"\0" * 6) * 3
            at = rng.randint(0, L - len(fake))
            q[at:at + len(fake)] = list(fake)
        recs.append(dict(name="r%d" % i, ref_id=0, pos=pos, mapq=60, flags=0x63, cigar=[("M", L)], mate_ref_id=0,
                         mate_pos=pos + 10, seq="".join(rng.choice("ACGT") for _ in range(L)), qual=q))
    return recs


@pytest.mark.parametrize("case", ["plain", "decoys", "long_records", "tiny_chunks"])
def test_chunk_walk_is_the_record_walk(tmp_path, case, monkeypatch):
    rng = random.Random(hash(case) & 0xffff)
    if case == "long_records":
        # records far longer than a 256 KiB stretch (and one longer than the reader's 1 MiB head room)
        recs = _records(rng, 40, 300_000, 420_000)
        recs += _records(rng, 1, 1_300_000, 1_300_001)
        recs.sort(key=lambda r: r["pos"])
    else:
        recs = _records(rng, 40_000 if case == "decoys" else 12_000, 60, 250, decoys=(case == "decoys"))
    if case == "tiny_chunks":
        monkeypatch.setenv("BIO_PILEUP_BATCH_BLOCKS", "1")
    rb = R.from_records(recs)
    path = str(tmp_path / "x.bam")
    host.bam_write(path, ["chr1"], [1 << 28], rb, level=1, threads=2)
    n, nc, h1, h2, srt, ok, rewalked = host.bam_scan_chunks(path)
    assert n == rb.n and h1 == h2 and srt and ok
    assert nc >= (20 if case == "tiny_chunks" else 1)
    if case == "decoys":
        assert rewalked > 0  # some guesses did fall for a decoy, and were caught
    if case == "plain":
        assert rewalked == 0


def test_unsorted_and_broken_records_are_reported_as_such(tmp_path):
    rng = random.Random(3)
    recs = _records(rng, 9000, 100, 200)
    recs[4000]["pos"] = 5  # out of order
    rb = R.from_records(recs)
    path = str(tmp_path / "u.bam")
    host.bam_write(path, ["chr1"], [1 << 28], rb, level=1, threads=2)
    n, nc, h1, h2, srt, ok, _ = host.bam_scan_chunks(path)
    assert n == rb.n and h1 == h2 and not srt and ok
    # a file cut off in the middle of a record: the same error either way (the record-by-record pass
    # of bio_bam_scan_chunks meets it first)
    st, off = bamio.record_stream(rb)
    whole = str(tmp_path / "w.bam")
    bamio.write_bam(whole, ["chr1"], [1 << 28], R.from_records(recs[:50]))
    import gzip
    body = gzip.open(whole, "rb").read()
    cut = str(tmp_path / "c.bam")
    w = bamio.BgzfWriter(cut)
    w.write(body[:-37])
    w.close()
    with pytest.raises(ValueError, match="truncated record"):
        host.bam_scan_chunks(cut)
    with pytest.raises(ValueError, match="truncated record"):
        host.bam_scan(cut)
