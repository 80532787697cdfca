"""Parses the output files the reference's own tests hold (tests/golden/ref_text/, sources in
SOURCES.json there) back into rows: (ref names, per-reference Seq8, ref_id, pos, vals[n,11],
off[4n+1], feats).  Test infrastructure for test_reference_text.py / test_gpu_reference_text.py:
the product's writers must turn these rows back into the files byte for byte."""
import os

import numpy as np

from bio_amd.snp import PER_READ_DTYPE

DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_text")
SEQ8_ASCII = "=ACMGRSVTWYHKDBN"   # pileup.Seq8ToASCIITable (pileup/common.go:72)
ENUM = {"A": 0, "C": 1, "G": 2, "T": 3, "N": 4}   # pileup.EnumToASCIITable (common.go:69)
STRAND = {".": 0, "+": 1, "-": 2}                 # pileup.StrandTypeToASCIITable (common.go:91)


def read(name):
    return open(os.path.join(DIR, name), "rb").read()


def _lines(name):
    text = read(name).decode()
    assert text.endswith("\n")
    rows = [ln.split("\t") for ln in text[:-1].split("\n")]
    return rows[0], rows[1:]


class Rows:
    def __init__(self):
        self.names = []
        self.key = {}       # (ref, pos) -> row index
        self.ref_id, self.pos, self.ref_char = [], [], []
        self.vals = []
        self.lists = []     # per row: 4 lists of (dist5p, fraglen, qual, strand)

    def row(self, chrom, pos1, ref_char):
        if chrom not in self.names:
            self.names.append(chrom)
        r = self.names.index(chrom)
        k = (r, int(pos1) - 1)
        if k not in self.key:
            self.key[k] = len(self.pos)
            self.ref_id.append(r)
            self.pos.append(int(pos1) - 1)
            self.ref_char.append(ref_char)
            self.vals.append([0] * 11)
            self.lists.append([[], [], [], []])
        i = self.key[k]
        assert self.ref_char[i] == ref_char
        return i

    def finish(self):
        n = len(self.pos)
        order = sorted(range(n), key=lambda i: (self.ref_id[i], self.pos[i]))
        assert order == list(range(n)), "the reference's files are in (ref, pos) order"
        seqs = []
        for r in range(len(self.names)):
            top = max(p for q, p in zip(self.ref_id, self.pos) if q == r)
            s = np.full(top + 1, 15, np.uint8)
            for q, p, c in zip(self.ref_id, self.pos, self.ref_char):
                if q == r:
                    s[p] = SEQ8_ASCII.index(c)
            seqs.append(s)
        off = [0]
        feats = []
        for ls in self.lists:
            for b in range(4):
                feats += ls[b]
                off.append(len(feats))
        f = np.zeros(len(feats), PER_READ_DTYPE)
        for i, (d5, fl, q, s) in enumerate(feats):
            f[i] = (d5, fl, q, s)
        return dict(names=self.names, seqs=seqs, ref_id=np.array(self.ref_id, np.int32),
                    pos=np.array(self.pos, np.int32), vals=np.array(self.vals, np.uint32).reshape(n, 11),
                    off=np.array(off, np.uint64), feats=f)


def _csv(field):
    return [] if field == "." else field.split(",")


def parse_tsv_pair(ref_name, alt_name, per_read):
    """A .ref.tsv / .alt.tsv pair of the tsv format (output.go:149-276).  per_read: the files
    carry 5P_DISTS, 3P_DISTS, STRANDS (-cols=enddists,strands); else DP + *_depth_tier1
    (-cols=-lowq).  tsv-format counts all sit in strand slot 0 (pileup.go:799-801)."""
    R = Rows()
    hdr, body = _lines(ref_name)
    for f in body:
        i = R.row(f[0], f[1], f[2])
        rb = ENUM.get(f[2], 4)
        if per_read:
            assert hdr[3:] == ["5P_DISTS", "3P_DISTS", "STRANDS"]
            d5, d3, st = _csv(f[3]), _csv(f[4]), _csv(f[5])
            assert len(d5) == len(d3) == len(st)
            R.lists[i][rb] = [(int(a), int(a) + int(b) + 1, 0, STRAND[c]) for a, b, c in zip(d5, d3, st)]
            R.vals[i][1 + 2 * rb] = len(d5)
        else:
            assert hdr[3:] == ["DP", "ref_depth_tier1"]
            R.vals[i][0] = int(f[3])
            R.vals[i][1 + 2 * rb] = int(f[4])
    hdr, body = _lines(alt_name)
    for f in body:
        i = R.row(f[0], f[1], f[2])
        ab = ENUM[f[3]]
        if per_read:
            assert hdr[4:] == ["5P_DISTS", "3P_DISTS", "STRANDS"]
            if ab == 4:
                assert f[4:] == [".", ".", "."]
                R.vals[i][1 + 2 * ab] = 1  # reported because its count is nonzero; the value is not shown
            else:
                d5, d3, st = _csv(f[4]), _csv(f[5]), _csv(f[6])
                R.lists[i][ab] = [(int(a), int(a) + int(b) + 1, 0, STRAND[c]) for a, b, c in zip(d5, d3, st)]
                R.vals[i][1 + 2 * ab] = len(d5)
        else:
            assert hdr[4:] == ["alt_depth_tier1"]
            R.vals[i][1 + 2 * ab] = int(f[4])
    if per_read:
        for v in R.vals:
            v[0] = sum(v[1:])
    return R.finish()


def parse_basestrand(name):
    """A .basestrand.tsv file (output.go:400, :437-453): the eight per-strand counts."""
    R = Rows()
    hdr, body = _lines(name)
    assert hdr == ["#CHROM", "POS", "REF", "A+", "A-", "C+", "C-", "G+", "G-", "T+", "T-"]
    for f in body:
        i = R.row(f[0], f[1], f[2])
        R.vals[i][1:9] = [int(x) for x in f[3:11]]
        R.vals[i][0] = sum(R.vals[i][1:9])
    return R.finish()
