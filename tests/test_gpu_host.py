"""GPU end-to-end: BAM + FASTA + BED files -> bio::Pileup (C++ driver over the HIP path) ->
TSV files, compared byte for byte with text rendered from the CPU oracle's rows by the
rules of pileup/snp/output.go (restated below, test-side)."""
import gzip
import os
import random
import subprocess

import numpy as np
import pytest

import oracle_lib as O
from bio_amd import bamio, host, records as R
from bio_amd.snp import Opts, PileupError
from test_oracle_vs_naive import random_bed, random_records

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SEQ8_ASCII = "=ACMGRSVTWYHKDBN"          # pileup/common.go:72
SEQ8_ENUM = [4, 0, 1, 4, 2, 4, 4, 4, 3, 4, 4, 4, 4, 4, 4, 4]  # common.go:66
ASCII_SEQ8 = {c: v for v, cs in ((1, "Aa"), (2, "Cc"), (4, "Gg"), (8, "Tt")) for c in cs}  # biosimd_amd64.go:471-487


def render_tsv(rows, ref_names, ref_ascii, cols):
    """output.go:83-131 header, :149-275 body -> (ref text, alt text)"""
    ref, pos, vals = rows
    hr, ha = ["#CHROM", "POS", "REF"], ["#CHROM", "POS", "REF", "ALT"]
    if cols & host.COL_DPREF:
        hr.append("DP")
    if cols & host.COL_DPALT:
        ha.append("DP")
    if cols & host.COL_HIGHQ:
        hr.append("ref_depth_tier1"); ha.append("alt_depth_tier1")
    if cols & host.COL_LOWQ:
        hr.append("ref_depth_tier2"); ha.append("alt_depth_tier2")
    out_r, out_a = ["\t".join(hr)], ["\t".join(ha)]
    for i in range(len(pos)):
        name = ref_names[ref[i]]
        b8 = ASCII_SEQ8.get(ref_ascii[ref[i]][pos[i]], 15)
        ch, rb = SEQ8_ASCII[b8], SEQ8_ENUM[b8]
        v = [int(x) for x in vals[i]]
        cnt = [v[1 + 2 * b] + v[2 + 2 * b] for b in range(5)]
        f = [name, str(pos[i] + 1), ch]
        if cols & host.COL_DPREF:
            f.append(str(v[0]))
        if cols & host.COL_HIGHQ:
            f.append(str(cnt[rb]))
        if cols & host.COL_LOWQ:
            f.append("0")
        out_r.append("\t".join(f))
        for alt in range(5):
            if alt == rb or cnt[alt] == 0:
                continue
            f = [name, str(pos[i] + 1), ch, "ACGTN"[alt]]
            if cols & host.COL_DPALT:
                f.append(str(v[0]))
            if cols & host.COL_HIGHQ:
                f.append(str(cnt[alt]))
            if cols & host.COL_LOWQ:
                f.append("0")
            out_a.append("\t".join(f))
    return "\n".join(out_r) + "\n", "\n".join(out_a) + "\n"


def render_basestrand(rows, ref_names, ref_ascii):
    """output.go:400, :445-453"""
    ref, pos, vals = rows
    out = ["#CHROM\tPOS\tREF\tA+\tA-\tC+\tC-\tG+\tG-\tT+\tT-"]
    for i in range(len(pos)):
        b8 = ASCII_SEQ8.get(ref_ascii[ref[i]][pos[i]], 15)
        out.append("\t".join([ref_names[ref[i]], str(pos[i] + 1), SEQ8_ASCII[b8]] +
                             [str(int(x)) for x in vals[i][1:9]]))
    return "\n".join(out) + "\n"


def _lists_cols(cols, f):
    """per-read columns of one feature list (output.go:173-200)"""
    out = []
    if cols & host.COL_ENDDISTS:
        out.append(",".join(str(int(x["dist5p"])) for x in f))
        out.append(",".join(str((int(x["fraglen"]) - 1 - int(x["dist5p"])) & 0xFFFF) for x in f))
    if cols & host.COL_QUALS:
        out.append(",".join(str(int(x["qual"])) for x in f))
    if cols & host.COL_FRAGLENS:
        out.append(",".join(str(int(x["fraglen"])) for x in f))
    if cols & host.COL_STRANDS:
        out.append(",".join(".+-"[int(x["strand"])] for x in f))
    return out


def _n_per_read_cols(cols):
    return (2 * bool(cols & host.COL_ENDDISTS) + bool(cols & host.COL_QUALS) + bool(cols & host.COL_FRAGLENS) +
            bool(cols & host.COL_STRANDS))


def render_tsv_per_read(res, ref_names, ref_ascii, cols):
    """output.go:83-131 header, :149-275 body with per-read column sets -> (ref text, alt text)"""
    ref, pos, vals, off, feats = res
    hr, ha = ["#CHROM", "POS", "REF"], ["#CHROM", "POS", "REF", "ALT"]
    if cols & host.COL_DPREF:
        hr.append("DP")
    if cols & host.COL_DPALT:
        ha.append("DP")
    for bit, names in ((host.COL_ENDDISTS, ["5P_DISTS", "3P_DISTS"]), (host.COL_QUALS, ["QUALS"]),
                       (host.COL_FRAGLENS, ["FRAGLENS"]), (host.COL_STRANDS, ["STRANDS"])):
        if cols & bit:
            hr += names
            ha += names
    if cols & host.COL_HIGHQ:
        hr.append("ref_depth_tier1"); ha.append("alt_depth_tier1")
    if cols & host.COL_LOWQ:
        hr.append("ref_depth_tier2"); ha.append("alt_depth_tier2")
    empty = ["."] * _n_per_read_cols(cols)
    out_r, out_a = ["\t".join(hr)], ["\t".join(ha)]
    for i in range(len(pos)):
        name = ref_names[ref[i]]
        b8 = ASCII_SEQ8.get(ref_ascii[ref[i]][pos[i]], 15)
        ch, rb = SEQ8_ASCII[b8], SEQ8_ENUM[b8]
        v = [int(x) for x in vals[i]]
        cnt = [v[1 + 2 * b] + v[2 + 2 * b] for b in range(5)]
        lists = [feats[int(off[4 * i + b]):int(off[4 * i + b + 1])] for b in range(4)]
        f = [name, str(pos[i] + 1), ch]
        if cols & host.COL_DPREF:
            f.append(str(v[0]))
        f += empty if rb == 4 or len(lists[rb]) == 0 else _lists_cols(cols, lists[rb])
        if cols & host.COL_HIGHQ:
            f.append(str(cnt[rb]))
        if cols & host.COL_LOWQ:
            f.append("0")
        out_r.append("\t".join(f))
        for alt in range(5):
            if alt == rb or cnt[alt] == 0:
                continue
            f = [name, str(pos[i] + 1), ch, "ACGTN"[alt]]
            if cols & host.COL_DPALT:
                f.append(str(v[0]))
            f += empty if alt == 4 else _lists_cols(cols, lists[alt])
            if cols & host.COL_HIGHQ:
                f.append(str(cnt[alt]))
            if cols & host.COL_LOWQ:
                f.append("0")
            out_a.append("\t".join(f))
    return "\n".join(out_r) + "\n", "\n".join(out_a) + "\n"


def render_basestrand_per_read(res, ref_names, ref_ascii, cols):
    """output.go:400-421 header (each per-read piece starts with a tab and gets another from
    WriteString), :445-553 body"""
    ref, pos, vals, off, feats = res
    bs = ["A+", "A-", "C+", "C-", "G+", "G-", "T+", "T-"]
    hdr = "#CHROM\tPOS\tREF\t" + "\t".join(bs)
    groups = []
    for bit, names in ((host.COL_ENDDISTS, ["5P_DISTS", "3P_DISTS"]), (host.COL_QUALS, ["QUALS"]),
                       (host.COL_FRAGLENS, ["FRAGLENS"]), (host.COL_STRANDS, ["STRANDS"])):
        if cols & bit:
            hdr += "\t" + "".join("\t%s_%s" % (n, b) for n in names for b in bs)
            groups += names
    out = [hdr]
    val = {"5P_DISTS": lambda x: str(int(x["dist5p"])),
           "3P_DISTS": lambda x: str((int(x["fraglen"]) - 1 - int(x["dist5p"])) & 0xFFFF),
           "QUALS": lambda x: str(int(x["qual"])), "FRAGLENS": lambda x: str(int(x["fraglen"])),
           "STRANDS": lambda x: "+" if int(x["strand"]) == 1 else "-"}
    for i in range(len(pos)):
        b8 = ASCII_SEQ8.get(ref_ascii[ref[i]][pos[i]], 15)
        f = [ref_names[ref[i]], str(pos[i] + 1), SEQ8_ASCII[b8]] + [str(int(x)) for x in vals[i][1:9]]
        if int(vals[i][0]) == 0:
            f += ["."] * (8 * len(groups))
        else:
            for g in groups:
                for b in range(4):
                    lst = feats[int(off[4 * i + b]):int(off[4 * i + b + 1])]
                    plus = [val[g](x) for x in lst if int(x["strand"]) == 1]
                    minus = [val[g](x) for x in lst if int(x["strand"]) != 1]
                    f += [",".join(plus) or ".", ",".join(minus) or "."] if len(lst) else [".", "."]
        out.append("\t".join(f))
    return "\n".join(out) + "\n"


class Case:
    def __init__(self, tmp_path, seed=11, n_pairs=2500, ref_lens=(9000, 4000, 6000, 3000), read_len=100,
                 bed_refs=(0, 1, 3), aux=False):
        rng = random.Random(seed)
        self.tmp = tmp_path
        self.ref_lens = list(ref_lens)
        self.ref_names = [f"chr{k + 1}" for k in range(len(ref_lens))]
        # reads on contigs 0, 2 and 3: contig 1 has BED positions but no reads (flushEmptyContigs)
        recs = []
        for r in (0, 2, 3):
            part = random_records(rng, n_pairs // 3, ref_len=ref_lens[r], read_len=read_len)
            for x in part:
                x["name"] = f"c{r}{x['name']}"
                x["mate_ref_id"] = r if x["mate_ref_id"] == x["ref_id"] else (r + 1) % len(ref_lens)
                x["ref_id"] = r
            recs += part
        recs.sort(key=lambda x: (x["ref_id"], x["pos"]))
        self.recs = recs
        self.batch = R.from_records(recs)
        self.ref_ascii = ["".join(rng.choice("ACGTacgtNR") if rng.random() < 0.1 else rng.choice("ACGT")
                                  for _ in range(l)) for l in ref_lens]
        full = {r: iv for r in bed_refs for iv in [random_bed(rng, ref_lens[r], dense=True)[0]]}
        self.bed = full
        self.bam = str(tmp_path / "in.bam")
        self.aux = None
        if aux:
            # DS as type C, DL as type i; one read without tags is placed where filters drop it
            self.ds = [rng.randint(0, 5) for _ in recs]
            self.dl = [rng.randint(0, 3) for _ in recs]
            self.aux = [b"XAZhello\0" + b"DSC" + bytes([a]) + b"DLi" + int(b).to_bytes(4, "little")
                        for a, b in zip(self.ds, self.dl)]
        bamio.write_bam(self.bam, self.ref_names, self.ref_lens, self.batch, aux=self.aux)
        # contig 2 (index 2) is missing from the FASTA: tolerated, it has no BED positions
        self.fa = str(tmp_path / "ref.fa")
        keep = [k for k in range(len(ref_lens)) if k != 2 or 2 in bed_refs]
        bamio.write_fasta(self.fa, [self.ref_names[k] for k in keep], [self.ref_ascii[k] for k in keep])
        self.bed_path = str(tmp_path / "targets.bed")
        bamio.write_bed(self.bed_path, [(r, s, e) for r in sorted(full) for s, e in full[r]], self.ref_names)

    def oracle(self, **kw):
        opts = O.make_opts(max_read_len=kw.pop("max_read_len", 500), max_read_span=kw.pop("max_read_span", 511), **kw)
        return O.pileup(opts, self.ref_lens, self.bed, self.batch)

    def read(self, name):
        p = os.path.join(str(self.tmp), name)
        if name.endswith(".gz"):
            return gzip.open(p, "rb").read().decode()
        return open(p).read()


def test_bed_mode_tsv_and_basestrand(tmp_path):
    c = Case(tmp_path)
    out = str(tmp_path / "o")
    host.pileup(c.bam, c.fa, "tsv", out, Opts(BedPath=c.bed_path, Mapq=30))
    want = c.oracle(mapq=30, ignore_strand=1)
    wr, wa = render_tsv(want, c.ref_names, c.ref_ascii, host.COLS_DEFAULT)
    assert c.read("o.ref.tsv") == wr
    assert c.read("o.alt.tsv") == wa
    assert wa.count("\n") > 50 and int(want[2][:, 0].sum()) > 50000

    host.pileup(c.bam, c.fa, "basestrand-tsv", out, Opts(BedPath=c.bed_path, Mapq=30, MinBaseQual=20))
    want = c.oracle(mapq=30, min_base_qual=20, ignore_strand=0)
    assert c.read("o.basestrand.tsv") == render_basestrand(want, c.ref_names, c.ref_ascii)


def test_cols_and_bgz(tmp_path):
    c = Case(tmp_path, seed=12)
    out = str(tmp_path / "o")
    host.pileup(c.bam, c.fa, "tsv-bgz", out, Opts(BedPath=c.bed_path, Cols="dpref,dpalt,highq", Clip=3))
    want = c.oracle(clip=3, ignore_strand=1)
    cols = host.COL_DPREF | host.COL_DPALT | host.COL_HIGHQ
    wr, wa = render_tsv(want, c.ref_names, c.ref_ascii, cols)
    assert c.read("o.ref.tsv.gz") == wr and c.read("o.alt.tsv.gz") == wa
    # header of the reference's own expected file for -cols dpref,highq
    # (cmd/bio-pileup/testdata/generic.ref.tsv.expected:1)
    host.pileup(c.bam, c.fa, "tsv", out, Opts(BedPath=c.bed_path, Cols="dpref,highq"))
    assert c.read("o.ref.tsv").split("\n")[0] == "#CHROM\tPOS\tREF\tDP\tref_depth_tier1"
    assert c.read("o.alt.tsv").split("\n")[0] == "#CHROM\tPOS\tREF\tALT\talt_depth_tier1"
    raw = open(out + ".ref.tsv.gz", "rb").read()
    assert raw[:4] == b"\x1f\x8b\x08\x04" and raw[12:14] == b"BC" and raw.endswith(bamio._EOF)
    host.pileup(c.bam, c.fa, "basestrand-tsv-bgz", out, Opts(BedPath=c.bed_path))
    assert c.read("o.basestrand.tsv.gz") == render_basestrand(c.oracle(), c.ref_names, c.ref_ascii)


def test_bgz_by_the_device_and_by_the_host_inflate_to_the_plain_files(tmp_path, monkeypatch):
    """-bgz: the device compresses the lines it renders (bp_render_fetch_bgzf); BIO_PILEUP_HOST_DEFLATE=1
    keeps the host's deflate.  Both files are BGZF with the EOF member and inflate to the plain format's
    bytes; small jobs make many short runs of members."""
    import gzip
    c = Case(tmp_path, seed=31)
    cases = (("basestrand-tsv", ["basestrand.tsv"], Opts(BedPath=c.bed_path)),
             ("tsv", ["ref.tsv", "alt.tsv"], Opts(BedPath=c.bed_path, Cols="dpref,dpalt")),
             # -per-strand in one pass: two files from one handle, each its strand's share of the rows
             ("tsv", ["strand.fwd.ref.tsv", "strand.fwd.alt.tsv", "strand.rev.ref.tsv", "strand.rev.alt.tsv"],
              Opts(BedPath=c.bed_path, PerStrand=True)))
    for fmt, names, o in cases:
        host.pileup(c.bam, c.fa, fmt, str(tmp_path / "p"), o)
        monkeypatch.setenv("BIO_PILEUP_JOB_SPAN", "900")
        host.pileup(c.bam, c.fa, fmt + "-bgz", str(tmp_path / "d"), o)
        monkeypatch.delenv("BIO_PILEUP_JOB_SPAN")
        monkeypatch.setenv("BIO_PILEUP_HOST_DEFLATE", "1")
        host.pileup(c.bam, c.fa, fmt + "-bgz", str(tmp_path / "h"), o)
        monkeypatch.delenv("BIO_PILEUP_HOST_DEFLATE")
        for nm in names:
            plain = open(tmp_path / ("p." + nm), "rb").read()
            assert len(plain) > 1000
            for k in "dh":
                raw = open(tmp_path / (k + "." + nm + ".gz"), "rb").read()
                assert raw.endswith(bamio._EOF) and raw[:4] == b"\x1f\x8b\x08\x04" and raw[12:14] == b"BC"
                assert gzip.decompress(raw) == plain, (fmt, nm, k)


@pytest.mark.parametrize("stitch", [False, True])
def test_job_cuts_with_halo_change_nothing(tmp_path, stitch, monkeypatch):
    """Cutting contigs into many small device jobs (halo = max-read-span, like the reference's
    padded shards) gives the same files as one job per contig."""
    c = Case(tmp_path, seed=13 + stitch)
    o = Opts(BedPath=c.bed_path, Stitch=stitch, MinBaseQual=25, MaxReadLen=150, MaxReadSpan=300)
    want = render_basestrand(c.oracle(stitch=int(stitch), min_base_qual=25, max_read_len=150, max_read_span=300),
                             c.ref_names, c.ref_ascii)
    host.pileup(c.bam, c.fa, "basestrand-tsv", str(tmp_path / "one"), o)
    assert c.read("one.basestrand.tsv") == want
    for span in (301, 700, 2500):
        monkeypatch.setenv("BIO_PILEUP_JOB_SPAN", str(span))
        host.pileup(c.bam, c.fa, "basestrand-tsv", str(tmp_path / "cut"), o)
        assert c.read("cut.basestrand.tsv") == want, span
    # and many small chunks of the stream (each handed to the submitting thread while reading goes on)
    monkeypatch.setenv("BIO_PILEUP_BATCH_BLOCKS", "1")
    for span in (700, 1 << 26):
        monkeypatch.setenv("BIO_PILEUP_JOB_SPAN", str(span))
        host.pileup(c.bam, c.fa, "basestrand-tsv", str(tmp_path / "sub"), o)
        assert c.read("sub.basestrand.tsv") == want, span


@pytest.mark.parametrize("stitch", [False, True])
def test_reader_modes_write_the_same_files(tmp_path, monkeypatch, capfd, stitch):
    """The three ways the BAM reaches the pack kernel - inflated stream resident on the device with
    records found there (the default; stitched runs get the read names from the scan too), device inflate with the host's record walk
    (BIO_PILEUP_HOST_SCAN=1), zlib on host threads (BIO_PILEUP_HOST_INFLATE=1) - write the same
    bytes, also with one-block batches (records straddle every batch edge) and small jobs; the
    stats line says which one ran."""
    c = Case(tmp_path, seed=77)
    o = Opts(BedPath=c.bed_path, MinBaseQual=20, MaxReadLen=150, MaxReadSpan=300, Stitch=stitch)
    want = render_basestrand(c.oracle(min_base_qual=20, max_read_len=150, max_read_span=300, stitch=int(stitch)),
                             c.ref_names, c.ref_ascii)
    monkeypatch.setenv("BIO_PILEUP_STATS", "1")
    for blocks, span in ((0, 0), (1, 0), (2, 700)):
        if blocks:
            monkeypatch.setenv("BIO_PILEUP_BATCH_BLOCKS", str(blocks))
        if span:
            monkeypatch.setenv("BIO_PILEUP_JOB_SPAN", str(span))
        for env, resident in ((None, 1), ("BIO_PILEUP_HOST_SCAN", 0), ("BIO_PILEUP_HOST_INFLATE", 0)):
            if env:
                monkeypatch.setenv(env, "1")
            capfd.readouterr()
            host.pileup(c.bam, c.fa, "basestrand-tsv", str(tmp_path / "m"), o)
            err = capfd.readouterr().err
            if env:
                monkeypatch.delenv(env)
            assert c.read("m.basestrand.tsv") == want, (blocks, span, env)
            assert "resident=%d" % resident in err, err
            if resident:
                assert "(host walks 0)" in err, err
            if env == "BIO_PILEUP_HOST_INFLATE":
                assert "device_inflate_s=0.000" in err, err


@pytest.mark.parametrize("fmt_cols", [("basestrand-tsv", ""), ("tsv", "dpref,quals,strands")])
def test_jobs_spread_over_devices_change_nothing(tmp_path, monkeypatch, fmt_cols):
    """-devices N: job k runs on logical device k mod N (mapped round-robin onto the GPUs present:
    on a one-GPU box all of them land on that GPU, each with its own closer thread), jobs are
    computed concurrently and written in job order.  The files are those of a run on one device,
    byte for byte - device-rendered text and host-rendered per-read columns alike - and the
    command line takes the flag."""
    import subprocess
    fmt, cols = fmt_cols
    c = Case(tmp_path, seed=21)
    o = Opts(BedPath=c.bed_path, Cols=cols, MinBaseQual=13)
    monkeypatch.setenv("BIO_PILEUP_JOB_SPAN", "900")
    names = ["basestrand.tsv"] if fmt == "basestrand-tsv" else ["ref.tsv", "alt.tsv"]
    host.pileup(c.bam, c.fa, fmt, str(tmp_path / "one"), o)
    want = [c.read("one." + n) for n in names]
    assert all(w.count("\n") > 100 for w in want[:1])
    for devices in (1, 3, 5, -1):
        host.pileup(c.bam, c.fa, fmt, str(tmp_path / "many"), o, devices=devices)
        assert [c.read("many." + n) for n in names] == want, devices
    exe = os.path.join(os.path.dirname(host.__file__), "bio-pileup")
    args = [exe, "-bed", c.bed_path, "-format", fmt, "-min-base-qual", "13", "-devices", "4", "-out", str(tmp_path / "cli")]
    if cols:
        args += ["-cols", cols]
    subprocess.run(args + [c.bam, c.fa], check=True, timeout=120)
    assert [c.read("cli." + n) for n in names] == want
    r = subprocess.run([exe, "-bed", c.bed_path, "-devices", "0", c.bam, c.fa], capture_output=True, timeout=60)
    assert r.returncode == 2 and b"-devices" in r.stderr


def test_per_strand(tmp_path):
    c = Case(tmp_path, seed=15)
    out = str(tmp_path / "ps")
    host.pileup(c.bam, c.fa, "tsv", out, Opts(BedPath=c.bed_path, PerStrand=True, Stitch=True))
    for sfx, req in ((".strand.fwd", 1), (".strand.rev", 2)):
        want = c.oracle(stitch=1, ignore_strand=1, strand_req=req)
        wr, wa = render_tsv(want, c.ref_names, c.ref_ascii, host.COLS_DEFAULT)
        assert c.read("ps" + sfx + ".ref.tsv") == wr
        assert c.read("ps" + sfx + ".alt.tsv") == wa


def test_per_strand_in_one_pass(tmp_path, monkeypatch):
    """-per-strand without -stitch and -min-base-qual: ONE split-strand pass yields both files
    (each strand's share of the rows).  Both equal the oracle's two passes (pileup.go:1118-1125),
    through the device's text and the host's, and the two-handle path (BIO_PILEUP_TWO_PASS=1)
    writes the same bytes.  Mapq 0 keeps the reads with odd flags in."""
    c = Case(tmp_path, seed=19)
    for cols, mask in (("", host.COLS_DEFAULT), ("dpref,dpalt,highq", host.COL_DPREF | host.COL_DPALT | host.COL_HIGHQ)):
        got = {}
        for mode in ("device", "host_text", "two_pass"):
            monkeypatch.delenv("BIO_PILEUP_HOST_TEXT", raising=False)
            monkeypatch.delenv("BIO_PILEUP_TWO_PASS", raising=False)
            if mode == "host_text":
                monkeypatch.setenv("BIO_PILEUP_HOST_TEXT", "1")
            if mode == "two_pass":
                monkeypatch.setenv("BIO_PILEUP_TWO_PASS", "1")
            out = str(tmp_path / f"p1_{mode}")
            host.pileup(c.bam, c.fa, "tsv", out, Opts(BedPath=c.bed_path, PerStrand=True, Mapq=0, Cols=cols))
            got[mode] = [c.read(f"p1_{mode}{sfx}.{f}.tsv") for sfx in (".strand.fwd", ".strand.rev") for f in ("ref", "alt")]
        want = []
        for req in (1, 2):
            rows = c.oracle(mapq=0, ignore_strand=1, strand_req=req)
            want += list(render_tsv(rows, c.ref_names, c.ref_ascii, mask))
        assert got["device"] == want
        assert got["host_text"] == want
        assert got["two_pass"] == want
    assert want[0] != want[2]  # the strands do differ


def test_region_mode(tmp_path):
    """-region: one unpadded shard (pileup.go:1091-1096) - reads starting before the region
    are never seen; a contig-only region is clamped to the contig (documented divergence)."""
    c = Case(tmp_path, seed=16, bed_refs=(0,))
    out = str(tmp_path / "r")
    host.pileup(c.bam, c.fa, "basestrand-tsv", out, Opts(Region="chr1:2001-3500", Mapq=0))
    opts = O.make_opts(mapq=0)
    shard, n = O.region_shard(0, 2000, 3500)
    want = O.pileup(opts, c.ref_lens, {0: [[2000, 3500]]}, c.batch, shards=shard, n_shards=n)
    assert c.read("r.basestrand.tsv") == render_basestrand(want, c.ref_names, c.ref_ascii)
    padded = O.pileup(opts, c.ref_lens, {0: [[2000, 3500]]}, c.batch)
    assert (padded[2] != want[2]).any()  # the quirk is visible on this input

    host.pileup(c.bam, c.fa, "basestrand-tsv", out, Opts(Region="chr4"))
    shard, n = O.region_shard(3, 0, c.ref_lens[3])
    want = O.pileup(O.make_opts(), c.ref_lens, {3: [[0, c.ref_lens[3]]]}, c.batch, shards=shard, n_shards=n)
    assert c.read("r.basestrand.tsv") == render_basestrand(want, c.ref_names, c.ref_ascii)
    host.pileup(c.bam, c.fa, "basestrand-tsv", out, Opts(Region="chr2:77"))
    assert c.read("r.basestrand.tsv").split("\n")[1].split("\t")[:2] == ["chr2", "77"]


def test_index_seeks_change_nothing(tmp_path):
    """With a .bai or .gbai the driver skips the stretches of the BAM that no BED interval
    needs (bamIterator.reset, bamprovider.go:286-348); the files are those of a front-to-back
    read, and far fewer records are decoded."""
    import re
    import shutil
    rng = random.Random(23)
    ref_lens, ref_names = [700_000, 50_000, 400_000], ["chr1", "chr2", "chr3"]
    recs = []
    for r in (0, 2):
        part = random_records(rng, 9000, ref_len=ref_lens[r], read_len=100)
        for x in part:
            x["name"] = f"c{r}{x['name']}"
            x["mate_ref_id"] = r
            x["ref_id"] = r
        recs += part
    recs.sort(key=lambda x: (x["ref_id"], x["pos"]))
    batch = R.from_records(recs)
    ascii_ = ["".join(rng.choice("ACGT") for _ in range(l)) for l in ref_lens]
    bed = {0: [[1000, 1500], [350_000, 350_400], [640_000, 640_100]], 1: [[10, 20]], 2: [[250_000, 250_300]]}
    fa, bedp = str(tmp_path / "r.fa"), str(tmp_path / "t.bed")
    bamio.write_fasta(fa, ref_names, ascii_)
    bamio.write_bed(bedp, [(r, s, e) for r in sorted(bed) for s, e in bed[r]], ref_names)
    plain = str(tmp_path / "plain.bam")
    bamio.write_bam(plain, ref_names, ref_lens, batch)
    exe = os.path.join(ROOT, "bio_amd", "bio-pileup")
    # (the whole file is less than one 16 MiB chunk of the reader: small chunks, so that there is something to skip)
    env = dict(os.environ, BIO_PILEUP_STATS="1", BIO_PILEUP_BATCH_BLOCKS="2")

    def run(bam, tag, *extra):
        r = subprocess.run([exe, *extra, "-format", "basestrand-tsv", "-mapq", "0", "-out", str(tmp_path / tag), bam, fa],
                           capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stderr
        m = re.search(r"records_read=(\d+) records_kept=(\d+) seeks=(\d+)", r.stderr)
        return open(str(tmp_path / tag) + ".basestrand.tsv").read(), [int(x) for x in m.groups()]

    want = render_basestrand(O.pileup(O.make_opts(mapq=0), ref_lens, bed, batch), ref_names, ascii_)
    got, (n_read, kept_plain, seeks) = run(plain, "p", "-bed", bedp)
    assert got == want and seeks == 0 and batch.n // 2 < n_read <= batch.n  # (it stops behind the last interval)
    shard, n = O.region_shard(2, 249_000, 251_000)
    want_region = render_basestrand(O.pileup(O.make_opts(mapq=0), ref_lens, {2: [[249_000, 251_000]]}, batch,
                                             shards=shard, n_shards=n), ref_names, ascii_)
    for kind in ("bai", "gbai"):
        bam = str(tmp_path / f"{kind}.bam")
        bamio.write_bam(bam, ref_names, ref_lens, batch, index=kind)
        got, (n_read, kept, seeks) = run(bam, kind, "-bed", bedp)
        assert got == want, kind
        assert seeks >= 3 and n_read < batch.n // 2, (kind, seeks, n_read)
        # index or not, the device is fed the same records: those of the 100 kb position shards
        # that meet the padded BED (what the reference scans)
        assert kept == kept_plain < batch.n // 2, (kind, kept, kept_plain)
        got, (n_read, _, seeks) = run(bam, kind + "r", "-region", "chr3:249001-251000")
        assert got == want_region and seeks == 1 and n_read < batch.n // 8, (kind, seeks, n_read)
        # -index names the file explicitly
        moved = str(tmp_path / f"elsewhere.{kind}")
        shutil.move(bam + "." + kind, moved)
        got, (n_read, _, seeks) = run(bam, kind + "x", "-bed", bedp, "-index", moved)
        assert got == want and seeks >= 3
    # ... so an over-long read far from every interval is an error neither way, and one near an
    # interval is an error both ways (pileup.go:766-768 is checked before the BED test)
    def with_long_read(pos):
        x = dict(recs[0], name="toolong", ref_id=0, mate_ref_id=0, pos=pos, mate_pos=pos, cigar=[("M", 600)],
                 seq="A" * 600, qual=[30] * 600, mapq=60, flags=0x63)
        rr = sorted(recs + [x], key=lambda y: (y["ref_id"], y["pos"]))
        return R.from_records(rr)
    for pos, fails in ((450_000, False), (349_000, True)):
        b2 = with_long_read(pos)
        for kind in (None, "bai"):
            bam = str(tmp_path / f"long{pos}{kind}.bam")
            bamio.write_bam(bam, ref_names, ref_lens, b2, index=kind)
            r = subprocess.run([exe, "-bed", bedp, "-format", "basestrand-tsv", "-mapq", "0", "-out", str(tmp_path / "lg"), bam, fa],
                               capture_output=True, text=True, timeout=300)
            if fails:
                assert r.returncode != 0 and "maxReadLen is 500, but read toolong" in r.stderr, (pos, kind, r.stderr)
            else:
                assert r.returncode == 0, (pos, kind, r.stderr)
                assert open(str(tmp_path / "lg") + ".basestrand.tsv").read() == want
    r = subprocess.run([exe, "-bed", bedp, "-index", str(tmp_path / "nope.bai"), "-out", str(tmp_path / "z"), plain, fa],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "no such file" in r.stderr


def test_bag_filters(tmp_path):
    """-min-bag-depth / -remove-sq read DS / DL aux tags on the host (pileup.go:695-716).
    The hts accessors are not in the reference tree: absent-tag behaviour is parity unpinned."""
    c = Case(tmp_path, seed=17, aux=True)
    out = str(tmp_path / "b")
    host.pileup(c.bam, c.fa, "basestrand-tsv", out, Opts(BedPath=c.bed_path, MinBagDepth=3, RemoveSq=True))
    keep = np.array([a >= 3 and b >= 2 for a, b in zip(c.ds, c.dl)])
    idx = np.nonzero(keep)[0]
    want = O.pileup(O.make_opts(), c.ref_lens, c.bed, R.from_records([c.recs[i] for i in idx]))
    assert c.read("b.basestrand.tsv") == render_basestrand(want, c.ref_names, c.ref_ascii)
    assert 0 < len(idx) < c.batch.n


def test_data_errors_surface_with_reference_wording(tmp_path):
    c = Case(tmp_path, seed=18)
    with pytest.raises(PileupError, match=r"pileupMutable.processShard: maxReadLen is 30, but read \S+ at chr\d:\d+ has length \d+"):
        host.pileup(c.bam, c.fa, "tsv", str(tmp_path / "e"), Opts(BedPath=c.bed_path, MaxReadLen=30, MaxReadSpan=511))
    # reads are at most 100 long; some carry a D or N op that pushes the span past 100
    with pytest.raises(PileupError, match=r"maxReadSpan is 100, but read \S+ at chr\d:\d+ has span 1\d\d"):
        host.pileup(c.bam, c.fa, "tsv", str(tmp_path / "e"), Opts(BedPath=c.bed_path, MaxReadLen=100, MaxReadSpan=100, Mapq=0))


def test_cli_binary(tmp_path):
    c = Case(tmp_path, seed=19)
    exe = os.path.join(ROOT, "bio_amd", "bio-pileup")
    r = subprocess.run([exe, "-bed", c.bed_path, "-format=basestrand-tsv", "-stitch", "-min-base-qual", "20",
                        "--out", str(tmp_path / "cli"), c.bam, c.fa], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    want = c.oracle(stitch=1, min_base_qual=20)
    assert c.read("cli.basestrand.tsv") == render_basestrand(want, c.ref_names, c.ref_ascii)


def test_per_read_column_sets(tmp_path, monkeypatch):
    """-cols with enddists / quals / fraglens / strands (appendBase, pileup.go:235-250; rendering
    output.go:165-203, :230-264, :454-553).  The reference's own expected files use
    enddists,strands (ordered2.*.expected) and dpref,quals,fraglens."""
    c = Case(tmp_path, seed=21)
    out = str(tmp_path / "pr")
    kw = dict(max_read_len=500, max_read_span=511)
    for cols_arg, cols, okw, pkw in (
            ("enddists,strands", host.COL_ENDDISTS | host.COL_STRANDS, dict(Mapq=60), dict(mapq=60)),
            ("dpref,quals,fraglens", host.COL_DPREF | host.COL_QUALS | host.COL_FRAGLENS,
             dict(Mapq=0, MinBaseQual=20, Clip=5), dict(mapq=0, min_base_qual=20, clip=5)),
            ("+dpalt,+strands", host.COLS_DEFAULT | host.COL_DPALT | host.COL_STRANDS, dict(Mapq=30), dict(mapq=30))):
        host.pileup(c.bam, c.fa, "tsv", out, Opts(BedPath=c.bed_path, Cols=cols_arg, **okw))
        want = O.pileup_per_read(O.make_opts(ignore_strand=1, **kw, **pkw), c.ref_lens, c.bed, c.batch)
        wr, wa = render_tsv_per_read(want, c.ref_names, c.ref_ascii, cols)
        assert c.read("pr.ref.tsv") == wr, cols_arg
        assert c.read("pr.alt.tsv") == wa, cols_arg
        assert wr.count(",") > 1000
    # basestrand-tsv with every per-read set, whole contigs and cut into small jobs
    allc = host.COL_ENDDISTS | host.COL_QUALS | host.COL_FRAGLENS | host.COL_STRANDS
    want = render_basestrand_per_read(O.pileup_per_read(O.make_opts(mapq=20, **kw), c.ref_lens, c.bed, c.batch),
                                      c.ref_names, c.ref_ascii, allc)
    for span in (None, 600):
        if span:
            monkeypatch.setenv("BIO_PILEUP_JOB_SPAN", str(span))
        host.pileup(c.bam, c.fa, "basestrand-tsv", out,
                    Opts(BedPath=c.bed_path, Cols="enddists,quals,fraglens,strands", Mapq=20))
        assert c.read("pr.basestrand.tsv") == want, span
    monkeypatch.delenv("BIO_PILEUP_JOB_SPAN")
    with pytest.raises(PileupError, match="stitched per-read features not yet supported"):
        host.pileup(c.bam, c.fa, "tsv", out, Opts(BedPath=c.bed_path, Cols="quals", Stitch=True))


def test_host_renderer_writes_the_same_files(tmp_path, monkeypatch):
    """Without per-read columns the lines come from the device (bp_render_text); the host
    renderer (kept for the per-read column sets, forced here with BIO_PILEUP_HOST_TEXT) must
    write the same bytes."""
    c = Case(tmp_path, seed=21)
    o = Opts(BedPath=c.bed_path, Mapq=20, MinBaseQual=15, Cols="dpref,dpalt,highq,lowq")
    for fmt, files in (("tsv", ["ref.tsv", "alt.tsv"]), ("basestrand-tsv", ["basestrand.tsv"])):
        monkeypatch.delenv("BIO_PILEUP_HOST_TEXT", raising=False)
        host.pileup(c.bam, c.fa, fmt, str(tmp_path / "dev"), Opts(BedPath=c.bed_path, Mapq=20, MinBaseQual=15,
                                                                  Cols=o.Cols if fmt == "tsv" else ""))
        monkeypatch.setenv("BIO_PILEUP_HOST_TEXT", "1")
        host.pileup(c.bam, c.fa, fmt, str(tmp_path / "cpu"), Opts(BedPath=c.bed_path, Mapq=20, MinBaseQual=15,
                                                                  Cols=o.Cols if fmt == "tsv" else ""))
        for f in files:
            a, b = c.read("dev." + f), c.read("cpu." + f)
            assert a == b and a.count("\n") > 100, f


@pytest.mark.parametrize("stitch", [False, True])
def test_many_chunks_and_jobs_at_12mb(tmp_path, monkeypatch, stitch):
    """The reader's chunk pipeline at a size where it matters: 12 Mb at 30x (2.4 M records, 0.65 GB
    inflated = 40 chunks of 16 MiB), cut into 6 jobs so that chunks cross job boundaries and the
    padding of the next job (handed-over-whole chunks, record-by-record chunks and stashed pieces
    all occur), against the oracle's rows for every position; once more with small chunks and a
    quality rule with clip."""
    import pandas as pd
    from bio_amd import synth
    L = 12_000_000
    sb = synth.generate([(0, 0, L - 1000, 30.0)], with_names=True, threads=8)
    rb = sb.to_record_batch()
    bam, fa, bedp = str(tmp_path / "in.bam"), str(tmp_path / "ref.fa"), str(tmp_path / "all.bed")
    host.bam_write(bam, ["chr1"], [L], rb, level=1, threads=8)
    rng = np.random.default_rng(3)
    seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)]
    with open(fa, "wb") as f:
        f.write(b">chr1\n" + seq.tobytes() + b"\n")
    bamio.write_bed(bedp, [(0, 0, L)], ["chr1"])
    monkeypatch.setenv("BIO_PILEUP_JOB_SPAN", str(1 << 21))

    def check(out, **okw):
        t = pd.read_csv(out + ".basestrand.tsv", sep="\t")
        want = O.pileup(O.make_opts(**okw), [L], {0: [[0, L]]}, rb)
        assert len(t) == L and (t["POS"].to_numpy() == np.arange(1, L + 1)).all()
        got = t[["A+", "A-", "C+", "C-", "G+", "G-", "T+", "T-"]].to_numpy(np.uint32)
        bad = np.nonzero((got != want[2][:, 1:9]).any(axis=1))[0]
        assert len(bad) == 0, (len(bad), bad[:5])
        assert (t["REF"].to_numpy().astype("S1").view(np.uint8) == seq).all()
        assert int(want[2][:, 0].sum()) > 2.5e8

    out = str(tmp_path / "o")
    host.pileup(bam, fa, "basestrand-tsv", out, Opts(BedPath=bedp, Stitch=stitch, MinBaseQual=20 if stitch else 0))
    check(out, stitch=int(stitch), min_base_qual=20 if stitch else 0)
    if not stitch:
        monkeypatch.setenv("BIO_PILEUP_BATCH_BLOCKS", "7")
        host.pileup(bam, fa, "basestrand-tsv", out, Opts(BedPath=bedp, MinBaseQual=13, Clip=4))
        check(out, min_base_qual=13, clip=4)
    sb.close()
