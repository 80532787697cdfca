"""CPU: the product's output writers (bio_amd/csrc/host/output.cpp, reached through
bio_pileup_write_rows) against the output files the reference's own tests hold
(tests/golden/ref_text/SOURCES.json): every body line is parsed back into a row (+ per-read
lists from the CSV columns), the rows go through the writers, and the files must come back byte
for byte, header included.  Pins pileup/snp/output.go:42-300 (tsv) and :376-453
(basestrand-tsv).  The reference's BAM/FASTA inputs of these files are git-LFS stubs, so
BAM -> rows on wgs_chr2_subset stays parity unpinned; this pins rows -> bytes.

Also the record-level wire formats (row.go:113-227, basestrand.go:77-135) on the literal
vectors of pileup/snp/basestrand_test.go."""
import struct

import numpy as np
import pytest

import ref_text as T
from bio_amd import host as H
from bio_amd.snp import PER_READ_DTYPE, PileupError


def _write(tmp_path, fmt, cols, rows, per_read=False):
    pre = str(tmp_path / "out")
    H.write_rows(fmt, pre, cols, rows["names"], rows["seqs"], rows["ref_id"], rows["pos"], rows["vals"],
                 rows["off"] if per_read else None, rows["feats"] if per_read else None)
    return pre


def test_generic_tsv_pair(tmp_path):
    """cmd/bio-pileup/pileup_snp_test.go:66-71: -cols=-lowq -> DP + *_depth_tier1."""
    rows = T.parse_tsv_pair("generic.ref.tsv.expected", "generic.alt.tsv.no_n.expected", per_read=False)
    assert len(rows["pos"]) == 25
    pre = _write(tmp_path, "tsv", "-lowq", rows)
    assert open(pre + ".ref.tsv", "rb").read() == T.read("generic.ref.tsv.expected")
    # the expected file is the .alt.tsv without its ALT=N lines; the rows carry no N counts
    assert open(pre + ".alt.tsv", "rb").read() == T.read("generic.alt.tsv.no_n.expected")
    # an N count adds exactly one line per position, after the bases (output.go:219-275)
    rows["vals"][1, 9] = 3
    pre = _write(tmp_path, "tsv", "-lowq", rows)
    alt = open(pre + ".alt.tsv", "rb").read().decode().split("\n")
    assert alt[1:3] == ["chr2_subset\t114145\tG\tA\t1", "chr2_subset\t114145\tG\tN\t3"]


def test_ordered2_tsv_pair_with_per_read_lists(tmp_path):
    """pileup_snp_test.go:97-103: -cols=enddists,strands: CSV lists, 3P = fraglen-1-5P, '+'/'-'
    strands, '.' for an empty REF list and for ALT=N."""
    rows = T.parse_tsv_pair("ordered2.ref.tsv.expected", "ordered2.alt.tsv.expected", per_read=True)
    assert len(rows["feats"]) > 90
    pre = _write(tmp_path, "tsv", "enddists,strands", rows, per_read=True)
    assert open(pre + ".ref.tsv", "rb").read() == T.read("ordered2.ref.tsv.expected")
    assert open(pre + ".alt.tsv", "rb").read() == T.read("ordered2.alt.tsv.expected")
    with pytest.raises(PileupError, match="without feature lists"):
        _write(tmp_path, "tsv", "enddists,strands", rows, per_read=False)


@pytest.mark.parametrize("name,n", [("sample1.basestrand.tsv", 2), ("test_pileup_snp.tsv", 999)])
def test_basestrand_tsv(tmp_path, name, n):
    rows = T.parse_basestrand(name)
    assert len(rows["pos"]) == n
    for fmt in ("basestrand-tsv", "basestrand-tsv-bgz"):
        pre = _write(tmp_path, fmt, "", rows)
        if fmt.endswith("bgz"):
            import gzip
            got = gzip.open(pre + ".basestrand.tsv.gz", "rb").read()
        else:
            got = open(pre + ".basestrand.tsv", "rb").read()
        assert got == T.read(name)


def test_write_rows_errors(tmp_path):
    rows = T.parse_basestrand("sample1.basestrand.tsv")
    with pytest.raises(PileupError, match="unrecognized format"):
        _write(tmp_path, "vcf", "", rows)
    with pytest.raises(PileupError, match="no reference base"):
        H.write_rows("basestrand-tsv", str(tmp_path / "x"), "", rows["names"], [None], rows["ref_id"], rows["pos"],
                     rows["vals"])


# ---------------------------------------------------------------- wire formats
def test_basestrand_record_and_trailer():
    """marshalBaseStrand / trailer (basestrand.go:77-135) on the piles of
    TestReadWriteBaseStrandsRio (basestrand_test.go:373-390) and the text of
    TestWriteBaseStrandToTSV / TestWriteBaseStrandTsv (:309-327, :348-352)."""
    piles = [(0, 123, [0, 1, 2, 3, 4, 5, 6, 7]), (1, 456, [0, 1, 2, 2, 4, 4, 8, 8]), (1, 777, [9, 8, 7, 6, 5, 4, 3, 2])]
    ref_id = [p[0] for p in piles]
    pos = [p[1] for p in piles]
    vals = np.zeros((3, 11), np.uint32)
    for i, p in enumerate(piles):
        vals[i, 1:9] = p[2]
        vals[i, 0] = 99   # Depth and the X counts are not part of a BaseStrandPile
        vals[i, 9:] = 7
    blob = H.marshal_basestrand(ref_id, pos, vals)
    assert len(blob) == 120
    for i, (r, p, c) in enumerate(piles):
        assert blob[40 * i:40 * i + 40] == struct.pack("<10I", r, p, *c)
    r2, p2, c2 = H.unmarshal_basestrand(blob)
    assert r2.tolist() == ref_id and p2.tolist() == pos and c2.tolist() == [p[2] for p in piles]
    tr = H.basestrand_rio_trailer(3)
    assert tr == struct.pack("<qq", 1, 3)
    assert H.parse_basestrand_rio_trailer(tr) == 3
    with pytest.raises(ValueError, match="unrecognized trailer version: got 2, want 1"):
        H.parse_basestrand_rio_trailer(struct.pack("<qq", 2, 3))
    text = H.basestrand_piles_to_tsv([0, 1], [123, 456], [[1, 2, 3, 4, 5, 6, 7, 8], [1, 1, 2, 2, 3, 3, 4, 4]],
                                     ["chr1", "chr2"])
    assert text == (b"CHROM\tPOS\tA+\tA-\tC+\tC-\tG+\tG-\tT+\tT-\n"
                    b"chr1\t123\t1\t2\t3\t4\t5\t6\t7\t8\n"
                    b"chr2\t456\t1\t1\t2\t2\t3\t3\t4\t4\n")
    text = H.basestrand_piles_to_tsv([0], [1], [[1, 2, 3, 4, 5, 6, 7, 8]], ["chr1"], ref_bases=["A"])
    assert text == b"#CHROM\tPOS\tREF\tA+\tA-\tC+\tC-\tG+\tG-\tT+\tT-\nchr1\t1\tA\t1\t2\t3\t4\t5\t6\t7\t8\n"


def test_pileup_row_wire_format():
    """MarshalPileupRow / unmarshalPileupRow (row.go:92-227): [0,4) fieldsPresent, [4,8) refID,
    [8,12) pos, [12,16) depth, 40 bytes of counts if present, then per present list a u32 length and
    6-byte entries; FieldsPresent as flushToInternal sets it (pileup.go:518-570)."""
    vals = np.array([9, 1, 2, 3, 4, 5, 6, 7, 8, 0, 1], np.uint32)
    la = np.array([(3, 150, 37, 1), (149, 150, 12, 2)], PER_READ_DTYPE)
    lg = np.array([(77, 151, 23, 1)], PER_READ_DTYPE)
    assert H.row_fields(np.zeros(11, np.uint32), [0, 0, 0, 0], True) == 0
    assert H.row_fields(vals, [2, 0, 1, 0], False) == 1
    f = H.row_fields(vals, [2, 0, 1, 0], True)
    assert f == 1 | 2 | 8
    rec = H.marshal_pileup_row(f, 5, 100006, vals, [la, None, lg, None])
    want = struct.pack("<4I", f, 5, 100006, 9) + struct.pack("<10I", *vals[1:].tolist())
    want += struct.pack("<I", 2) + struct.pack("<HHBB", 3, 150, 37, 1) + struct.pack("<HHBB", 149, 150, 12, 2)
    want += struct.pack("<I", 1) + struct.pack("<HHBB", 77, 151, 23, 1)
    assert rec == want
    f2, r2, p2, v2, lists = H.unmarshal_pileup_row(rec)
    assert (f2, r2, p2) == (f, 5, 100006) and (v2 == vals).all()
    assert lists[0].tobytes() == la.tobytes() and lists[2].tobytes() == lg.tobytes()
    assert len(lists[1]) == 0 and len(lists[3]) == 0
    # zero-depth row: 16 bytes (pileup.go:521-527)
    z = H.marshal_pileup_row(0, 1, 7, np.zeros(11, np.uint32))
    assert z == struct.pack("<4I", 0, 1, 7, 0)
    assert H.unmarshal_pileup_row(z)[3].sum() == 0
    # counts only: 56 bytes
    c = H.marshal_pileup_row(1, 0, 1, vals)
    assert len(c) == 56 and c[16:] == struct.pack("<10I", *vals[1:].tolist())
    with pytest.raises(ValueError, match="truncated"):
        H.unmarshal_pileup_row(rec[:-1])
