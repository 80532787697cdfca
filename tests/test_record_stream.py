"""bamio.record_stream (test support for bp_submit_bam) writes what bamio.write_bam writes:
the inflated body of the file, record for record."""
import gzip
import random
import struct

import numpy as np

from bio_amd import bamio, records as R
from test_oracle_vs_naive import random_records


def test_record_stream_is_the_body_of_the_bam_file(tmp_path):
    rng = random.Random(5)
    rb = R.from_records(random_records(rng, 200, n_ref=2))
    path = str(tmp_path / "x.bam")
    bamio.write_bam(path, ["a", "b"], [3000, 2500], rb)
    data = gzip.open(path, "rb").read()  # BGZF is a multi-member gzip file
    l_text = struct.unpack_from("<i", data, 4)[0]
    p = 8 + l_text
    n_ref = struct.unpack_from("<i", data, p)[0]
    p += 4
    for _ in range(n_ref):
        p += 8 + struct.unpack_from("<i", data, p)[0]
    st, off = bamio.record_stream(rb)
    # (the bin field is not part of what the pileup reads: write_bam computes it, record_stream does not)
    a = np.frombuffer(data[p:], np.uint8).copy()
    assert a.size == st.size
    for o in off:
        a[int(o) + 14:int(o) + 16] = st[int(o) + 14:int(o) + 16]
    assert (a == st).all()
    # offsets chain through the block sizes
    at = 0
    for o in off:
        assert at == int(o)
        at += 4 + struct.unpack_from("<I", st, at)[0]
    assert at == st.size
    # with aux bytes every record grows by that much and stays well formed
    st2, off2 = bamio.record_stream(rb, aux_bytes=7)
    assert st2.size == st.size + 7 * rb.n
    assert bytes(st2[int(off2[1]) - 7:int(off2[1])]) == b"XXZaaa\0"


def test_oracle_bam_decoder_reads_back_what_was_written(tmp_path):
    """oracle/bam_decode.c (the CPU baseline's BAM -> records step) against the records a BAM was
    written from, for the Python writer and the product's writer"""
    import oracle_lib as O
    from bio_amd import host
    rng = random.Random(9)
    rb = R.from_records(random_records(rng, 400, n_ref=2))
    for how in ("python", "product"):
        path = str(tmp_path / f"{how}.bam")
        if how == "python":
            bamio.write_bam(path, ["a", "b"], [3000, 2500], rb)
        else:
            host.bam_write(path, ["a", "b"], [3000, 2500], rb, level=1, threads=2)
        got, n_ref = O.bam_decode(path)
        assert n_ref == 2 and got.n == rb.n
        for k in ("ref_id", "pos", "mate_ref_id", "mate_pos", "flags", "mapq", "n_cigar", "l_seq", "cigar", "seq", "qual"):
            assert (getattr(got, k) == getattr(rb, k)).all(), (how, k)
        assert bytes(got.name) == bytes(rb.name) and (got.name_off == rb.name_off).all()
