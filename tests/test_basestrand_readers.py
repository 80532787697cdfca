"""The exported readers of basestrand files against the reference's literal
vectors: pileup/snp/basestrand_test.go:34-72 (sample1.basestrand.tsv), :75-158
(TestReadBaseStrandTsv), :160-239 (TestReadSingleStrandBaseStrandTsv), :250-275
(test_pileup_snp.tsv: row 30, 999 rows) and :277-300 (TestChrId).  Host code in
libbiopileup.so (bio_amd/csrc/host/wire.cpp); no GPU."""
import os

import pytest

from bio_amd import host

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_text")
KEYS = ("FwdA", "RevA", "FwdC", "RevC", "FwdG", "RevG", "FwdT", "RevT")


def tsv(rows):  # makeTestTsv, basestrand_test.go:242-248
    return "".join("\t".join(r) + "\n" for r in rows).encode()


def row(chr_, pos, ref, *c):
    return dict(Chr=chr_, Pos=pos, Ref=ref, **dict(zip(KEYS, c)))


def test_read_basestrand_tsv_from_file():  # basestrand_test.go:34-72
    rows = host.read_basestrand_tsv(open(os.path.join(GOLD, "sample1.basestrand.tsv"), "rb").read())
    assert rows == [row("chr8", 50824362, "C", 0, 0, 0, 2, 0, 0, 0, 0),
                    row("chr8", 50824381, "T", 0, 0, 0, 0, 0, 0, 0, 7)]


def test_read_basestrand_tsv_literals():  # basestrand_test.go:75-158
    one = [["chr1", "569378", "C", "0", "0", "0", "23", "0", "0", "0", "0"]]
    assert host.read_basestrand_tsv(tsv(one)) == [row("chr1", 569378, "C", 0, 0, 0, 23, 0, 0, 0, 0)]
    two = one + [["chr1", "569381", "A", "0", "23", "0", "0", "0", "0", "0", "0"]]
    assert host.read_basestrand_tsv(tsv(two)) == [row("chr1", 569378, "C", 0, 0, 0, 23, 0, 0, 0, 0),
                                                  row("chr1", 569381, "A", 0, 23, 0, 0, 0, 0, 0, 0)]
    with pytest.raises(ValueError, match="invalid syntax"):
        host.read_basestrand_tsv(tsv([["chr1", "569378", "C", "A", "0", "0", "23", "0", "0", "0", "0"]]))
    assert host.read_basestrand_tsv(b"") == []
    # the '#' header is a comment, CRLF and a missing final newline are tolerated
    assert host.read_basestrand_tsv(b"#CHROM\tPOS\r\nchr2\t7\tG\t1\t2\t3\t4\t5\t6\t7\t8") == [row("chr2", 7, "G", 1, 2, 3, 4, 5, 6, 7, 8)]
    with pytest.raises(ValueError, match="wrong number of fields"):
        host.read_basestrand_tsv(b"chr2\t7\tG\t1\n")
    with pytest.raises(ValueError, match="out of range"):
        host.read_basestrand_tsv(tsv([["chr1", "9223372036854775808", "C"] + ["0"] * 8]))
    assert host.read_basestrand_tsv(tsv([["chr1", "-9223372036854775808", "C"] + ["0"] * 8]))[0]["Pos"] == -2 ** 63


def test_read_single_strand_basestrand_tsv():  # basestrand_test.go:160-239
    hdr = ["CHROM", "POS", "DEPTH", "REF", "A", "C", "G", "T", "N", "INS", "DEL"]
    assert host.read_single_strand_basestrand_tsv(tsv([hdr]), tsv([hdr])) == []
    fwd = [hdr,
           ["chr1", "569378", "5", "A", "5", "0", "0", "0", "0", "0", "0"],
           ["chr1", "569381", "8", "G", "0", "0", "7", "1", "0", "0", "0"],
           ["chr1", "569384", "7", "G", "0", "0", "7", "0", "0", "0", "0"]]
    rev = [hdr,
           ["chr1", "569378", "8", "A", "8", "0", "0", "0", "0", "0", "0"],
           ["chr1", "569381", "7", "G", "0", "0", "6", "1", "0", "0", "0"],
           ["chr1", "569384", "7", "G", "0", "0", "7", "0", "0", "0", "0"]]
    want = [row("chr1", 569378, "A", 5, 8, 0, 0, 0, 0, 0, 0),
            row("chr1", 569381, "G", 0, 0, 0, 0, 7, 6, 1, 1),
            row("chr1", 569384, "G", 0, 0, 0, 0, 7, 7, 0, 0)]
    assert host.read_single_strand_basestrand_tsv(tsv(fwd), tsv(rev)) == want
    # columns are found by name: another order, an extra column
    perm = [10, 3, 0, 2, 1, 7, 6, 5, 4, 9, 8]
    fwd2 = [[r[i] for i in perm] + ["x"] for r in fwd]
    assert host.read_single_strand_basestrand_tsv(tsv(fwd2), tsv(rev)) == want
    # the shorter file ends the read (basestrand.go:323-338)
    assert host.read_single_strand_basestrand_tsv(tsv(fwd[:3]), tsv(rev)) == want[:2]
    bad = [list(r) for r in rev]
    bad[2][1] = "569382"
    with pytest.raises(ValueError, match="expected equal chromosome position but got fwd: chr1 569381 and rev: chr1 569382"):
        host.read_single_strand_basestrand_tsv(tsv(fwd), tsv(bad))
    bad = [list(r) for r in rev]
    bad[1][3] = "C"
    with pytest.raises(ValueError, match="expected equal reference alleles but got fwd: A and rev: C"):
        host.read_single_strand_basestrand_tsv(tsv(fwd), tsv(bad))
    with pytest.raises(ValueError, match="DEL does not appear"):
        host.read_single_strand_basestrand_tsv(tsv([r[:10] for r in fwd]), tsv(rev))
    with pytest.raises(ValueError, match="invalid syntax"):
        host.read_single_strand_basestrand_tsv(tsv(fwd), tsv([hdr, ["chr1", "569378", "8", "A", "8", "0", "0", "0", "0", "q", "0"]]))


def test_test_pileup_snp_tsv():  # basestrand_test.go:250-275
    rows = host.read_basestrand_tsv(open(os.path.join(GOLD, "test_pileup_snp.tsv"), "rb").read())
    assert len(rows) == 999
    assert rows[30] == dict(Chr="chr1", Pos=713277, Ref="C", FwdA=0, FwdC=0, FwdG=0, FwdT=8, RevA=0, RevC=12, RevG=0, RevT=0)
    # and the product's writer turns them back into the file, byte for byte (WriteBaseStrandTsv)
    import numpy as np
    names = sorted({r["Chr"] for r in rows})
    text = host.basestrand_piles_to_tsv([names.index(r["Chr"]) for r in rows], [r["Pos"] for r in rows],
                                        np.array([[r[k] for k in KEYS] for r in rows], np.uint32), names,
                                        [r["Ref"] for r in rows])
    assert text == open(os.path.join(GOLD, "test_pileup_snp.tsv"), "rb").read()


def test_chr_id():  # basestrand_test.go:277-300, basestrand.go:401-422
    assert host.chr_id("chr12") == 12
    assert host.chr_id("chrX") == 23
    assert [host.chr_id(f"chr{i}") for i in range(1, 23)] == list(range(1, 23))
    assert (host.chr_id("chrY"), host.chr_id("chrM")) == (24, 25)
    for bad, why in (("chr0", "0 is not a valid chromosome id"), ("chr23", "23 is not a valid chromosome id"),
                     ("chrUn", "invalid syntax"), ("chr", "invalid syntax"), ("c", "shorter")):
        with pytest.raises(ValueError, match=why):
            host.chr_id(bad)
