"""Device BGZF inflate (bp_inflate_blocks, inflate_kernels.hip) against zlib, the byte-exact
checker: every block of BAM files written at level 1, level 6 and level 0 (stored blocks), hand
made blocks for the corners of RFC 1951 (fixed codes, several deflate blocks per BGZF block,
empty stored blocks from sync flushes, maximal matches, distance 1 runs, 64 KiB blocks, one-byte
blocks, long codes behind the root tables), and corrupt or truncated blocks, which must be refused
as the reference's reader refuses them (CRC32 / ISIZE checks of bgzf.Reader under
encoding/bamprovider/bamprovider.go:417-435)."""
import os
import random
import struct
import zlib

import numpy as np
import pytest

from bio_amd import bgzf, host, snp, synth

pytestmark = pytest.mark.gpu


def bgzf_block(payload, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, flush_at=(), raw=None, extra=b""):
    """One BGZF block around `payload` (or around the given raw deflate bytes)."""
    if raw is None:
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
        raw, last = b"", 0
        for cut, mode in flush_at:
            raw += co.compress(payload[last:cut]) + co.flush(mode)
            last = cut
        raw += co.compress(payload[last:]) + co.flush()
    xlen = 6 + len(extra)
    total = 12 + xlen + len(raw) + 8
    assert total <= 65536, total
    hdr = struct.pack("<BBBBIBBH", 31, 139, 8, 4, 0, 0, 255, xlen) + extra + struct.pack("<BBHH", 66, 67, 2, total - 1)
    return hdr + raw + struct.pack("<II", zlib.crc32(payload) & 0xFFFFFFFF, len(payload))


def corner_payloads():
    rng = random.Random(5)
    nr = np.random.default_rng(5)
    text = b"".join(b"read%06d\tchr1\t%d\t60\t150M\t=\t%d\n" % (i, 1000 + 7 * i, 1300 + 7 * i) for i in range(1400))
    quals = bytes(nr.choice(np.frombuffer(b"#,7F", np.uint8), 60000, p=[.02, .05, .13, .8]))
    skew = bytes(nr.choice(np.arange(256, dtype=np.uint8), 50000, p=np.r_[[0.5, 0.25, 0.125], np.full(253, 0.125 / 253)]))
    return {
        "one_byte": b"x",
        "three_bytes": b"abc",
        "four_bytes": b"abcd",
        "five_bytes": b"abcde",
        "zeros_64k": bytes(65536),
        "run_of_two": b"ab" * 30000,
        "run_of_three": b"abc" * 20000,
        "run_of_257": bytes(range(256)) * 200 + b"q" * 257 * 3,
        "random_20k": rng.randbytes(20000),
        "random_64k": rng.randbytes(65000),
        "text": text[:65000],
        "quals": quals,
        "skewed_alphabet": skew,           # many codes longer than the 10-bit root table
        "far_matches": rng.randbytes(32768) * 2,  # distance 32768
        "mixed": (text[:9000] + bytes(3000) + rng.randbytes(5000) + quals[:20000] + text[:9000])[:65000],
    }


def try_block(make):
    try:
        return make()
    except AssertionError:
        return None


def test_corner_blocks_inflate_to_what_zlib_says():
    cases = []
    for name, pl in corner_payloads().items():
        makers = [(f"l{lvl}", (lambda pl=pl, lvl=lvl: bgzf_block(pl, lvl))) for lvl in (1, 6, 9, 0)]
        makers += [("fixed", lambda pl=pl: bgzf_block(pl, 6, zlib.Z_FIXED)),
                   ("huffman_only", lambda pl=pl: bgzf_block(pl, 6, zlib.Z_HUFFMAN_ONLY)),
                   ("rle", lambda pl=pl: bgzf_block(pl, 6, zlib.Z_RLE)),
                   ("filtered", lambda pl=pl: bgzf_block(pl, 6, zlib.Z_FILTERED))]
        if len(pl) > 100:
            c1, c2 = len(pl) // 3, 2 * len(pl) // 3
            makers.append(("flushes", lambda pl=pl, c1=c1, c2=c2: bgzf_block(
                pl, 6, flush_at=((c1, zlib.Z_SYNC_FLUSH), (c1, zlib.Z_SYNC_FLUSH), (c2, zlib.Z_FULL_FLUSH)))))
        for tag, mk in makers:
            b = try_block(mk)
            if b is not None:
                cases.append((f"{name}/{tag}", pl, b))
    assert len(cases) > 100
    # an extra subfield in front of BC, and the EOF marker block (ISIZE 0) in the middle of the file
    cases.append(("extra_subfield", b"hello world" * 10, bgzf_block(b"hello world" * 10, 6, extra=struct.pack("<BBH4s", 88, 89, 4, b"abcd"))))
    eof = bgzf_block(b"", 6)
    file_bytes = b"".join(b for _, _, b in cases[:7]) + eof + b"".join(b for _, _, b in cases[7:]) + eof
    blocks, end, total = bgzf.scan_blocks(file_bytes)
    assert end == len(file_bytes) and len(blocks) == len(cases) and total == sum(len(p) for _, p, _ in cases)
    want = b"".join(p for _, p, _ in cases)
    # zlib agrees with the payloads (the checker checked)
    for (name, pl, _), r in zip(cases, blocks):
        assert zlib.decompress(file_bytes[r["in_off"]:r["in_off"] + r["in_len"]], -15) == pl, name
    with bgzf.Inflater() as inf:
        got = inf.inflate(np.frombuffer(file_bytes, np.uint8), blocks)
        if got.tobytes() != want:
            for (name, pl, _), r in zip(cases, blocks):
                g = got[r["out_off"]:r["out_off"] + r["out_len"]].tobytes()
                assert g == pl, (name, next(i for i in range(len(pl)) if g[i] != pl[i]))
        # the same blocks in another order and at other output alignments, without the CRC pass
        order = list(range(len(cases)))
        random.Random(1).shuffle(order)
        b2 = blocks[order].copy()
        pad = np.array([(3 * k) % 7 for k in range(len(order))], np.int64)
        b2["out_off"] = np.concatenate([[0], np.cumsum(b2["out_len"].astype(np.int64) + pad)[:-1]]) + 1
        got2 = inf.inflate(np.frombuffer(file_bytes, np.uint8), b2, check_crc=False)
        for k, r in zip(order, b2):
            assert got2[r["out_off"]:r["out_off"] + r["out_len"]].tobytes() == cases[k][1], cases[k][0]
        # ... and with it (the CRC kernel at every alignment of a block's first byte)
        got3 = inf.inflate(np.frombuffer(file_bytes, np.uint8), b2, check_crc=True)
        assert (got3[b2["out_off"][5]:][:10] == got2[b2["out_off"][5]:][:10]).all()


def _bam(tmp_path, level, n_mb=2.0, coverage=30.0):
    L = int(n_mb * 1e6)
    sb = synth.generate([(0, 0, L - 1000, coverage)], with_names=True, threads=8)
    rb = sb.to_record_batch()
    p = str(tmp_path / f"l{level}.bam")
    host.bam_write(p, ["chr1"], [L], rb, level=level, threads=8)
    sb.close()
    return p


@pytest.mark.parametrize("level", [1, 6, 0])
def test_every_block_of_a_bam_inflates_to_what_zlib_says(tmp_path, level):
    data = open(_bam(tmp_path, level), "rb").read()
    blocks, end, total = bgzf.scan_blocks(data)
    assert end == len(data) and len(blocks) > 300
    want = b"".join(zlib.decompress(data[r["in_off"]:r["in_off"] + r["in_len"]], -15) for r in blocks)
    assert len(want) == total and want[:4] == b"BAM\x01"
    with bgzf.Inflater() as inf:
        got = inf.inflate(np.frombuffer(data, np.uint8), blocks)
        assert got.tobytes() == want
        ms, crc_ms = inf.timing()
        assert ms > 0 and crc_ms > 0
        # a second, smaller call on the same inflater (its buffers keep the first call's bytes)
        few = blocks[10:40].copy()
        few["out_off"] -= few["out_off"][0]
        got = inf.inflate(np.frombuffer(data, np.uint8), few)
        lo = int(blocks["out_off"][10])
        assert got.tobytes() == want[lo:lo + len(got)]


def test_corrupt_and_truncated_blocks_are_refused(tmp_path):
    data = bytearray(open(_bam(tmp_path, 1, n_mb=0.5), "rb").read())
    blocks, _, _ = bgzf.scan_blocks(bytes(data))
    assert len(blocks) > 20
    clean = np.frombuffer(bytes(data), np.uint8)
    with bgzf.Inflater() as inf:
        inf.inflate(clean, blocks)

        def refused(comp, blk, what, idx):
            with pytest.raises(snp.PileupError, match=what) as ei:
                inf.inflate(comp, blk)
            assert inf.bad_block == idx, (inf.bad_block, idx)
            return ei.value.code
        # a flipped bit in the payload: the structure may survive, the checksum does not
        rng = random.Random(9)
        for _ in range(12):
            k = rng.randrange(len(blocks))
            bad = clean.copy()
            bad[int(blocks["in_off"][k]) + rng.randrange(20, int(blocks["in_len"][k]))] ^= 1 << rng.randrange(8)
            assert refused(bad, blocks, "bgzf: corrupt block", k) == -4
        # a wrong CRC in the trailer
        b = blocks.copy()
        b["crc32"][7] ^= 0x10
        refused(clean, b, r"corrupt block 7 \(CRC-32 mismatch\)", 7)
        # ... is only seen when the CRC is checked
        inf.inflate(clean, b, check_crc=False)
        # ISIZE too small / too large
        b = blocks.copy()
        b["out_len"][3] -= 1
        refused(clean, b, r"corrupt block 3 \(more data than ISIZE\)", 3)
        b = blocks.copy()
        b["out_len"][-1] += 1
        refused(clean, b, r"less data than ISIZE", len(blocks) - 1)
        # a block cut short (a truncated file): the stream runs out
        b = blocks.copy()
        b["in_len"][5] //= 2
        refused(clean, b, r"corrupt block 5", 5)
        # garbage
        junk = np.frombuffer(random.Random(3).randbytes(len(clean)), np.uint8)
        with pytest.raises(snp.PileupError, match="bgzf: corrupt block 0"):
            inf.inflate(junk, blocks)
        # a table that does not fit the bytes given is refused on the host
        b = blocks.copy()
        b["in_off"][2] = len(clean)
        with pytest.raises(snp.PileupError, match="block 2 does not fit"):
            inf.inflate(clean, b)
        # and the inflater still works
        assert inf.inflate(clean, blocks).size == int(blocks["out_len"].sum())


# ---- the record scan on the device (bp_bam_scan) and bp_submit_bam_device ----------------------
def _host_columns(stream, first):
    """record offsets and fixed fields of an inflated BAM stream, walked one by one in Python"""
    out, p, n = [], first, len(stream)
    while p + 4 <= n:
        bs = struct.unpack_from("<I", stream, p)[0]
        if bs < 32 or p + 4 + bs > n:
            break
        ref, pos, w3, w4, lseq, mref, mpos = struct.unpack_from("<iiIIIii", stream, p + 4)
        l_name, ncig = w3 & 0xFF, w4 & 0xFFFF
        span, bad, ops = 0, False, []
        for k in range(ncig):
            v = struct.unpack_from("<I", stream, p + 36 + l_name + 4 * k)[0]
            ops.append(v & 15)
            if (v & 15) in (0, 2, 3, 7, 8):
                span += v >> 4
            bad |= (v & 15) > 5
        facts = (1 if ops == [0] else 0) | (2 if bad else 0)
        out.append((p, ref, pos, mref, mpos, w4 >> 16, (w3 >> 8) & 0xFF, l_name, ncig, lseq, 4 + bs, span, facts))
        p += 4 + bs
    return out, p


def test_device_record_scan_and_submit(tmp_path):
    """bp_bam_scan over a stream inflated on the device: the same records, fields and spans as a
    walk on the host, for the whole stream and from a start in the middle, with a stream cut
    mid-record; and bp_submit_bam_device of those columns gives bp_submit_bam's rows."""
    L = 400_000
    sb = synth.generate([(0, 0, L - 1000, 30.0)], with_names=True, threads=8)
    rb = sb.to_record_batch()
    p = str(tmp_path / "s.bam")
    host.bam_write(p, ["chr1", "chr2"], [L, 1000], rb, level=1, threads=8)
    data = np.fromfile(p, np.uint8)
    blocks, _, total = bgzf.scan_blocks(data)
    stream = b"".join(zlib.decompress(data[r["in_off"]:r["in_off"] + r["in_len"]].tobytes(), -15) for r in blocks)
    # the header: magic, l_text, text, n_ref, references
    l_text = struct.unpack_from("<I", stream, 4)[0]
    q = 12 + l_text
    for _ in range(struct.unpack_from("<I", stream, 8 + l_text)[0]):
        q += 8 + struct.unpack_from("<I", stream, q)[0]
    want, want_exit = _host_columns(stream, q)
    assert len(want) == rb.n and want_exit == len(stream)
    keys = [k for k, _ in bgzf.Inflater.COLS]
    with bgzf.Inflater() as inf:
        inf.inflate(data, blocks, to_host=False)
        rc, cols, ex, stopped, cstruct = inf.scan(q, total, 2)
        assert rc == 0, inf.lib.bp_inflate_last_error(inf.h).decode()
        assert ex == want_exit and stopped == 0 and len(cols["off"]) == rb.n
        for j, k in enumerate(keys):
            assert (cols[k].astype(np.int64) == np.array([w[j] for w in want], np.int64)).all(), k
        # a stream that ends mid-record, from a record in the middle
        mid = want[len(want) // 3][0]
        cut = want[-7][0] + 50
        rc, c2, ex2, st2, _ = inf.scan(mid, cut, 2)
        assert rc == 0 and st2 == 1 and ex2 == want[-7][0]
        assert (c2["off"] == cols["off"][len(want) // 3:len(want) - 7]).all()
        # a start that is no record start: the chain leads nowhere sensible, or a guess fails - never a silent success with other records
        rc, c3, ex3, st3, _ = inf.scan(mid + 1, total, 2)
        assert rc == 1 or st3 == 1 or len(c3["off"]) != len(want) - len(want) // 3
        # the columns handed to a job, against the same records handed over as host bytes
        import oracle_lib as O
        opts = O.make_opts(min_base_qual=13, clip=2)
        bed = {0: [[1000, L - 5000]]}
        rc, cols, ex, stopped, cstruct = inf.scan(q, total, 2)
        with snp.PileupJob(opts, [L, 1000]) as a, snp.PileupJob(opts, [L, 1000]) as b:
            a.set_bed(bed)
            b.set_bed(bed)
            a.submit_bam(np.frombuffer(stream, np.uint8), cols["off"])
            half = rb.n // 2
            b.submit_bam_device(inf.device_output, total, cstruct, 0, half)
            b.submit_bam_device(inf.device_output, total, cstruct, half, rb.n - half)
            a.finish()
            b.finish()
            ra, rbw = a.rows(), b.rows()
            assert (ra[2] == rbw[2]).all() and int(ra[2][:, 0].sum()) > 1_000_000
            wv = O.pileup(opts, [L, 1000], bed, rb)[2]
            assert (rbw[2] == wv).all()
        with snp.PileupJob(O.make_opts(stitch=1), [L, 1000]) as j:
            j.set_bed(bed)
            with pytest.raises(snp.PileupError, match="stitch mode needs the read names"):
                j.submit_bam_device(inf.device_output, total, cstruct, 0, 10)
        # ... and with the names gathered by the scan a stitched job pairs its reads from the device
        # stream as from host bytes: the names are the stream's, the rows the oracle's
        rc, ncols, ex, stopped, nstruct = inf.scan(q, total, 2, names=True)
        assert rc == 0 and len(ncols["off"]) == rb.n and len(ncols["name_off"]) == rb.n + 1
        for i in list(range(0, rb.n, 997)) + [rb.n - 1]:
            o, ln = int(ncols["off"][i]), int(ncols["l_name"][i])
            a, b = int(ncols["name_off"][i]), int(ncols["name_off"][i + 1])
            assert b - a == ln - 1 and bytes(ncols["names"][a:b]) == stream[o + 36:o + 36 + ln - 1], i
        sopts = O.make_opts(stitch=1, min_base_qual=20, clip=1)
        with snp.PileupJob(sopts, [L, 1000]) as a, snp.PileupJob(sopts, [L, 1000]) as b:
            a.set_bed(bed)
            b.set_bed(bed)
            a.submit_bam(np.frombuffer(stream, np.uint8), ncols["off"])
            third = rb.n // 3
            for lo, hi in ((0, third), (third, third + 1), (third + 1, rb.n)):  # (pairs split over calls)
                b.submit_bam_device(inf.device_output, total, nstruct, lo, hi - lo)
            a.finish()
            b.finish()
            ra, rbw = a.rows(), b.rows()
            assert (ra[2] == rbw[2]).all()
            assert (rbw[2] == O.pileup(sopts, [L, 1000], bed, rb)[2]).all()
    sb.close()
