"""CPU tests of the C++ host layers (include/bio_pileup_host.h): parsers, loaders, option
validation and the command line.  No counts are computed here; the end-to-end run is in
test_gpu_host.py."""
import gzip
import json
import os
import random
import subprocess

import numpy as np
import pytest

from bio_amd import bamio, host
from bio_amd.records import from_records
from bio_amd.snp import Opts, PileupError

import oracle_lib
from test_oracle_vs_naive import random_records

GOLD = os.path.join(os.path.dirname(__file__), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ interval
def test_parse_region_reference_vectors():
    kat = json.load(open(os.path.join(GOLD, "misc_kat.json")))["region_strings"]["cases"]
    for region, name, start0, end in kat:
        assert host.parse_region(region) == (name, start0, end)


@pytest.mark.parametrize("region,msg", [
    ("", "empty region string"),
    (":5", "empty contig ID"),
    ("chr1:0", "position 0 in region string out of range"),
    ("chr1:0-5", "position 0 in region string out of range"),
    ("chr1:5-5", "invalid range string 5-5"),
    ("chr1:5-2147483647", "invalid range string 5-2147483647"),
    ("chr1:a-5", 'parsing "a": invalid syntax'),
    ("chr1:5-b", 'parsing "b": invalid syntax'),
])
def test_parse_region_errors(region, msg):
    with pytest.raises(ValueError, match=msg):
        host.parse_region(region)


def test_bed_loader_reference_vectors():
    kat = json.load(open(os.path.join(GOLD, "interval_kat.json")))
    bed = host.Bed(os.path.join(GOLD, "interval_test1.bed"), ["chr1", "chr2"])
    assert bed.endpoints(0).tolist() == kat["test1_endpoints"]
    assert bed.endpoints(1) is None  # never mentioned
    for start, limit, want in kat["test1_intersects"]:
        assert bed.intersects(0, start, limit) == want, (start, limit)
    assert not bed.intersects(1, 0, 1 << 30)
    bed2 = host.Bed(os.path.join(GOLD, "interval_test2.bed"), ["chr2", "chrX", "chr1"])
    assert bed2.endpoints(2).tolist() == kat["test2_zero_based"]["chr1"]
    assert bed2.endpoints(0).tolist() == kat["test2_zero_based"]["chr2"]
    assert bed2.endpoints(1) is None


def test_bed_loader_merges_gzip_and_empties(tmp_path):
    text = ("# not a comment line in the reference either: 3 tokens are needed\n"
            "\n"
            "c1 10 20 name\n"
            "c1\t20\t30\n"       # touching -> merged
            "c1\t25\t28\n"       # contained
            "c1\t40\t40\n"       # empty -> dropped
            "c1\t50\t60\n"
            "c2\t7\t7\n"         # mentioned, but empty
            "c3 5 6\n"
            "c4 9 9\n"          # empty, first line of its contig, last line of the file
            )
    p = tmp_path / "a.bed.gz"
    with gzip.open(p, "wb") as f:
        f.write(text.encode())
    with pytest.raises(ValueError, match="invalid syntax"):
        host.Bed(str(p), ["c1"])  # first line: token 2 "a" is not a number
    text = text.split("\n", 1)[1]
    with gzip.open(p, "wb") as f:
        f.write(text.encode())
    bed = host.Bed(str(p), ["c3", "c2", "c1", "c0", "c4"])
    assert bed.endpoints(2).tolist() == [10, 30, 50, 60]
    assert bed.endpoints(1).tolist() == []  # mentioned: present but empty (bedunion.go:473-479)
    # ...while the end-of-input flush keeps the (-1,-1) placeholder (bedunion.go:532-538)
    assert bed.endpoints(4).tolist() == [-1, -1]
    assert bed.endpoints(0).tolist() == [5, 6]
    assert bed.endpoints(3) is None
    plain = tmp_path / "a.bed"
    plain.write_text(text)
    assert host.Bed(str(plain), ["c1"]).endpoints(0).tolist() == [10, 30, 50, 60]


@pytest.mark.parametrize("text,msg", [
    ("c1 10 20\nc1 5 8\n", "unsorted input"),
    ("c1 10 20\nc2 5 8\nc1 30 40\n", "split reference"),
    ("c1 10\n", "line 1 has fewer tokens than expected"),
    ("c1 -3 5\n", "negative start coordinate"),
    ("c1 9 5\n", "invalid coordinate pair on line 1"),
    ("c1 9 2147483647\n", "invalid coordinate pair on line 1"),
])
def test_bed_loader_errors(tmp_path, text, msg):
    p = tmp_path / "bad.bed"
    p.write_text(text)
    with pytest.raises(ValueError, match=msg):
        host.Bed(str(p), ["c1", "c2"])


def test_bed_subset_matches_oracle_positions(tmp_path):
    """Subset keeps exactly the positions of [start, limit) (bedunion.go:753-852); compared as
    position sets with the oracle's restatement, which also reproduces the reference's
    left-over empty interval."""
    rng = np.random.default_rng(5)
    eps = np.sort(rng.choice(np.arange(1, 4000), size=60, replace=False)).astype(np.int32)
    p = tmp_path / "s.bed"
    p.write_text("".join(f"c\t{eps[i]}\t{eps[i + 1]}\n" for i in range(0, 60, 2)))
    full = host.Bed(str(p), ["c"]).endpoints(0)
    assert full.tolist() == eps.tolist()

    def positions(e):
        return [x for k in range(0, len(e), 2) for x in range(e[k], e[k + 1])]

    for _ in range(200):
        a, b = sorted(int(x) for x in rng.integers(0, 4100, 2))
        if a == b:
            continue
        got = host.Bed(str(p), ["c"], subset=(0, a, 0, b)).endpoints(0)
        want = oracle_lib.bed_subset1(eps, a, b)
        assert positions(got.tolist()) == positions(list(want)), (a, b)
        assert positions(got.tolist()) == [x for x in positions(eps.tolist()) if a <= x < b]


# ------------------------------------------------------------------ cols
def test_parse_cols():
    D = host.COLS_DEFAULT
    assert host.parse_cols("") == D
    assert host.parse_cols("dpref,highq") == host.COL_DPREF | host.COL_HIGHQ  # generic.ref.tsv.expected
    assert host.parse_cols("enddists,strands") == host.COL_ENDDISTS | host.COL_STRANDS  # ordered2.*.expected
    assert host.parse_cols("+dpalt,-lowq") == (D | host.COL_DPALT) & ~host.COL_LOWQ
    with pytest.raises(ValueError, match="either all terms"):
        host.parse_cols("+dpalt,lowq")
    with pytest.raises(ValueError, match="either all terms"):
        host.parse_cols("dpalt,-lowq")
    with pytest.raises(ValueError, match="parseCols: depth not found"):
        host.parse_cols("depth")
    with pytest.raises(ValueError, match="parseCols: x not found"):
        host.parse_cols("+x")


# ------------------------------------------------------------------ fasta
def test_fasta_parallel_loader_equals_line_loader(tmp_path):
    """A plain file that starts with a header is loaded on all threads from a mapping of the file
    (fasta.cpp: LoadPlainParallel); the same bytes gzip-compressed take the line-by-line loader.
    Both must store the same sequences: random line lengths, CRLF and bare CR, empty lines, a
    header without sequence lines, names cut at the first space, a last line with and without its
    newline, a sequence of several 4 MiB blocks with line ends on the block edges."""
    import random
    rng = random.Random(5)
    for trial in range(6):
        parts, names = [], []
        for k in range(rng.randrange(1, 6)):
            name = f"c{trial}_{k}"
            names.append(name)
            parts.append(f">{name} some description\r\n" if rng.random() < 0.3 else f">{name}\n")
            if rng.random() < 0.2:
                continue  # a header without sequence lines
            n_lines = rng.randrange(1, 40) if trial else 80000
            for _ in range(n_lines):
                ln = "".join(rng.choice("ACGTacgtNnRY\r") for _ in range(rng.choice([0, 1, 7, 60, 61, 120])))
                parts.append(ln + rng.choice(["\n", "\n", "\r\n", "\r\r\n"]))
        text = "".join(parts)
        if trial % 2:
            text = text.rstrip("\n") + rng.choice(["", "\r", "A"])
        if trial == 0:  # line ends exactly at and around the first 4 MiB edge of the first sequence
            head = text[: text.index("\n") + 1]
            body = ("ACGT" * 15 + "\r\n") * ((4 << 20) // 62 + 2)
            text = head + body + text[len(head):]
        plain, gz = tmp_path / f"t{trial}.fa", tmp_path / f"t{trial}.fa.gz"
        plain.write_bytes(text.encode())
        gz.write_bytes(gzip.compress(text.encode(), 1))
        for name in names:
            try:
                want = host.fasta_seq8(str(gz), name)
            except ValueError as ex:
                with pytest.raises(ValueError, match=str(ex)[:20]):
                    host.fasta_seq8(str(plain), name)
                continue
            got = host.fasta_seq8(str(plain), name)
            assert got.tobytes() == want.tobytes(), (trial, name, len(got), len(want))


def test_fasta_seq8(tmp_path):
    p = tmp_path / "r.fa"
    p.write_bytes(b">c1 first contig\nACGTacgt\nNnRY-\n\n>c2\r\nGG\r\nTT\r\n>c3 x y\nA\n")
    assert host.fasta_seq8(str(p), "c1").tolist() == [1, 2, 4, 8, 1, 2, 4, 8, 15, 15, 15, 15, 15]
    assert host.fasta_seq8(str(p), "c2").tolist() == [4, 4, 8, 8]
    assert host.fasta_seq8(str(p), "c3").tolist() == [1]
    with pytest.raises(ValueError, match="sequence not found: c4"):
        host.fasta_seq8(str(p), "c4")
    gz = tmp_path / "r.fa.gz"
    with gzip.open(gz, "wb") as f:
        f.write(p.read_bytes())
    assert host.fasta_seq8(str(gz), "c2").tolist() == [4, 4, 8, 8]


# ------------------------------------------------------------------ bam
def _small_bam(tmp_path, n=3000, seed=3, ref_len=20000, **kw):
    recs = random_records(random.Random(seed), n, ref_len=ref_len, read_len=150, **kw)
    batch = from_records(recs)
    path = str(tmp_path / "t.bam")
    bamio.write_bam(path, ["chrT"], [ref_len], batch)
    return path, batch


def test_bam_reader_decodes_every_record(tmp_path):
    path, batch = _small_bam(tmp_path)
    n, sl, sc, n_ref = host.bam_scan(path)
    assert (n, n_ref) == (batch.n, 1)
    assert sl == int(batch.l_seq.sum()) and sc == int(batch.n_cigar.sum())
    assert batch.n * 300 > 2 * 0xFF00  # spans several BGZF blocks


def test_native_bam_writer_round_trip(tmp_path):
    path, batch = _small_bam(tmp_path)
    native = str(tmp_path / "n.bam")
    host.bam_write(native, ["chrT"], [20000], batch, level=1, threads=3)
    assert host.bam_scan(native) == host.bam_scan(path)
    # same inflated bytes as the pure-Python writer (header and records)
    import zlib

    def inflate(p):
        raw, out, o = open(p, "rb").read(), b"", 0
        while o < len(raw):
            bsize = int.from_bytes(raw[o + 16:o + 18], "little") + 1
            out += zlib.decompress(raw[o + 18:o + bsize - 8], -15)
            o += bsize
        return out
    assert inflate(native) == inflate(path)


def test_bam_reader_rejects_corruption(tmp_path):
    path, _ = _small_bam(tmp_path, n=400)
    raw = bytearray(open(path, "rb").read())
    raw[len(raw) // 2] ^= 0x55
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(raw)
    with pytest.raises(ValueError, match="bgzf|bam"):
        host.bam_scan(bad)
    open(bad, "wb").write(bytes(raw[:len(raw) // 2]))
    with pytest.raises(ValueError, match="bgzf|bam"):
        host.bam_scan(bad)
    with pytest.raises(ValueError, match="no such file"):
        host.bam_scan(str(tmp_path / "missing.bam"))


def _indexed_bam(tmp_path, kind):
    """3 contigs, the middle one without reads; a gap on the last one"""
    rng = random.Random(11)
    recs = random_records(rng, 2500, ref_len=200_000, read_len=100, n_ref=1)
    more = [dict(r, ref_id=2, mate_ref_id=2) for r in random_records(rng, 2500, ref_len=90_000, read_len=100)]
    far = [dict(r, ref_id=2, mate_ref_id=2, pos=r["pos"] + 400_000, mate_pos=r["mate_pos"] + 400_000)
           for r in random_records(rng, 800, ref_len=50_000, read_len=100)]
    batch = from_records(sorted(recs + more + far, key=lambda r: (r["ref_id"], r["pos"])))
    path = str(tmp_path / f"i_{kind}.bam")
    bamio.write_bam(path, ["c0", "c1", "c2"], [200_000, 50_000, 500_000], batch, index=kind)
    return path, path + "." + kind, batch


@pytest.mark.parametrize("kind", ["bai", "gbai"])
def test_bam_index_seek_reaches_every_wanted_record(tmp_path, kind):
    """bamIterator.reset + Scan (bamprovider.go:286-348, 417-435): the offset an index gives
    is at or before the first record at or after the target, and close to it."""
    path, ipath, b = _indexed_bam(tmp_path, kind)
    total = b.n
    key = b.ref_id.astype(np.int64) * (1 << 32) + b.pos
    for ref, pos in [(0, 0), (0, 1), (0, 99_999), (0, 150_000), (0, 199_999), (1, 0), (1, 40_000), (2, 0),
                     (2, 45_000), (2, 95_000), (2, 250_000), (2, 400_000), (2, 430_000)]:
        want = int(total - np.searchsorted(key, ref * (1 << 32) + pos, "left"))
        found, fr, fp, skipped, after = host.bam_seek_scan(path, ipath, ref, pos)
        assert found and after == want, (kind, ref, pos, after, want)
        # the seek saves work: far fewer records are skipped than lie before the target
        assert skipped <= min(total - want, 700), (kind, ref, pos, skipped)
    # past every record: the .bai shows there is nothing; a .gbai lands on its last entry
    found, fr, fp, skipped, after = host.bam_seek_scan(path, ipath, 2, 499_000)
    assert after == 0 and (not found or skipped < 700)


def test_bam_index_errors(tmp_path):
    path, ipath, _ = _indexed_bam(tmp_path, "bai")
    with pytest.raises(ValueError, match="no such file"):
        host.bam_seek_scan(path, str(tmp_path / "nope.bai"), 0, 0)
    bad = str(tmp_path / "bad.bai")
    open(bad, "wb").write(b"BAM\x01" + open(ipath, "rb").read()[4:])
    with pytest.raises(ValueError, match="bad magic"):
        host.bam_seek_scan(path, bad, 0, 0)
    open(bad, "wb").write(open(ipath, "rb").read()[:-5])
    with pytest.raises(ValueError, match="truncated"):
        host.bam_seek_scan(path, bad, 0, 0)
    badg = str(tmp_path / "bad.gbai")
    with gzip.open(badg, "wb") as f:
        f.write(b"GBAI" + bytes(12) + bytes(20))
    with pytest.raises(ValueError, match="bad magic"):
        host.bam_seek_scan(path, badg, 0, 0)


# ------------------------------------------------------------------ snp.Pileup validation
def _inputs(tmp_path, ref_len=20000, fa_len=None):
    path, _ = _small_bam(tmp_path, n=50, ref_len=ref_len)
    fa = str(tmp_path / "t.fa")
    bamio.write_fasta(fa, ["chrT"], ["A" * (fa_len or ref_len)])
    bed = str(tmp_path / "t.bed")
    bamio.write_bed(bed, [(0, 100, 200)], ["chrT"])
    return path, fa, bed


@pytest.mark.parametrize("kw,fmt,msg", [
    (dict(Clip=-1, Region="chrT"), "tsv", "Pileup: invalid clip= argument"),
    (dict(Clip=250, Region="chrT"), "tsv", "Pileup: invalid clip= argument"),
    (dict(MaxReadLen=600, Region="chrT"), "tsv", "max-read-len= argument cannot be larger than max-read-span="),
    (dict(Region="chrT"), "vcf", "Pileup: unrecognized format= argument"),
    (dict(Region="chrT", Cols="dpref"), "basestrand-rio", "-cols cannot be used with basestrand-rio output"),
    (dict(Region="chrT", Cols="nope"), "tsv", "parseCols: nope not found"),
    (dict(), "tsv", "Pileup: either -bed or -region is currently required"),
    (dict(Region="chrT", BedPath="BED"), "tsv", "-region and -bed flags can't be used together yet"),
    (dict(Region="chrU:5-9"), "tsv", "Pileup: region= contig not in BAM/PAM"),
    (dict(Region="chrT:0-9"), "tsv", "position 0 in region string out of range"),
    (dict(Region="chrT", PerStrand=True), "basestrand-tsv", "single-strand mode not supported with basestrand output"),
    (dict(Region="chrT", MinBaseQual=96), "tsv", "newQualPassTable: minBaseQual too large"),
    (dict(Region="chrT"), "basestrand-rio", "basestrand-rio output is not supported by this build"),
    (dict(Region="chrT", Cols="quals", Stitch=True), "tsv", "stitched per-read features not yet supported"),
])
def test_pileup_option_errors(tmp_path, kw, fmt, msg):
    bam, fa, bed = _inputs(tmp_path)
    if kw.get("BedPath") == "BED":
        kw = dict(kw, BedPath=bed)
    with pytest.raises(PileupError, match=msg):
        host.pileup(bam, fa, fmt, str(tmp_path / "out"), Opts(**kw))


def test_pileup_fasta_length_mismatch(tmp_path):
    bam, fa, bed = _inputs(tmp_path, fa_len=19999)
    with pytest.raises(PileupError, match=r"inconsistent lengths for contig chrT \(20000 in BAM/PAM header, 19999 in .fa\)"):
        host.pileup(bam, fa, "tsv", str(tmp_path / "out"), Opts(BedPath=bed))


def test_pileup_needs_a_gpu(tmp_path):
    """Valid inputs on a box without a GPU: the driver must refuse, not count on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    bam, fa, bed = _inputs(tmp_path)
    with pytest.raises(PileupError, match="no CPU fallback"):
        host.pileup(bam, fa, "tsv", str(tmp_path / "out"), Opts(BedPath=bed))


# ------------------------------------------------------------------ command line
def _cli(*args):
    exe = os.path.join(ROOT, "bio_amd", "bio-pileup")
    return subprocess.run([exe, *args], capture_output=True, text=True, timeout=120)


def test_cli_flag_syntax(tmp_path):
    bam, fa, bed = _inputs(tmp_path)
    r = _cli("-bed", bed, bam)
    assert r.returncode == 1 and "Missing positional arguments" in r.stderr
    r = _cli("-bed", bed, bam, fa, "extra")
    assert r.returncode == 1 and "Too many positional arguments" in r.stderr
    r = _cli("-nope", bam, fa)
    assert r.returncode == 2 and "flag provided but not defined: -nope" in r.stderr
    r = _cli("-mapq", "sixty", bam, fa)
    assert r.returncode == 2 and 'invalid value "sixty" for flag -mapq' in r.stderr
    r = _cli("-mapq")
    assert r.returncode == 2 and "flag needs an argument: -mapq" in r.stderr
    # Go flag forms: --name=value, bool without value, hex ints; errors come from snp.Pileup
    r = _cli("--format=tsv", "-stitch", "-flag-exclude", "0xf00", "-clip=300", "-region", "chrT", bam, fa)
    assert r.returncode == 2 and "panic: Pileup: invalid clip= argument" in r.stderr
    r = _cli("-per-strand=false", "-stitch=true", bam, fa)
    assert r.returncode == 2 and "either -bed or -region is currently required" in r.stderr
    r = _cli("-h")
    assert r.returncode == 0 and "Usage:" in r.stdout and "-max-read-span" in r.stdout
