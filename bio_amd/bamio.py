"""Minimal BAM / FASTA / BED writers for tests and end-to-end runs (public SAM-spec layout;
the record block is the one gbam.Unmarshal decodes, encoding/bam/unmarshal.go:126-224).
Writers only: reading is the native library's job (bio_amd/csrc/host/bam.cpp)."""
import gzip
import struct
import zlib

import numpy as np

_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
_BLOCK = 0xFF00


def _bgzf_block(data, level):
    co = zlib.compressobj(level, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    bsize = 18 + len(comp) + 8 - 1
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + comp +
            struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


class BgzfWriter:
    def __init__(self, path, level=1):
        self.f = open(path, "wb")
        self.buf = bytearray()
        self.level = level
        self.fpos = 0

    def tell(self):
        """BGZF virtual offset of the next byte written."""
        return (self.fpos << 16) | len(self.buf)

    def _emit(self, data):
        blk = _bgzf_block(data, self.level)
        self.f.write(blk)
        self.fpos += len(blk)

    def write(self, b):
        self.buf += b
        while len(self.buf) >= _BLOCK:
            self._emit(bytes(self.buf[:_BLOCK]))
            del self.buf[:_BLOCK]

    def close(self):
        if self.buf:
            self._emit(bytes(self.buf))
            self.buf = bytearray()
        self.f.write(_EOF)
        self.f.close()


def _reg2bin(beg, end):
    end -= 1
    for shift, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return off + (beg >> shift)
    return 0


GBAI_MAGIC = bytes([0x47, 0x42, 0x41, 0x49, 0x01, 0xf1, 0x78, 0x5c, 0x7b, 0xcb, 0xc1, 0xba, 0x08, 0x23, 0xb1, 0x19])


def write_bai(path, n_ref, recs):
    """recs: (ref_id, pos, end, voffset_begin, voffset_end) per record, in file order.  SAM
    specification section 5.2: bins with chunks, then the 16 kb linear index."""
    bins = [dict() for _ in range(n_ref)]
    lin = [[] for _ in range(n_ref)]
    for ref, pos, end, vb, ve in recs:
        if ref < 0:
            continue
        end = max(end, pos + 1)
        ch = bins[ref].setdefault(_reg2bin(pos, end), [])
        if ch and ch[-1][1] == vb:
            ch[-1][1] = ve
        else:
            ch.append([vb, ve])
        lo, hi = pos >> 14, (end - 1) >> 14
        v = lin[ref]
        if len(v) <= hi:
            v.extend([0] * (hi + 1 - len(v)))
        for w in range(lo, hi + 1):
            if v[w] == 0 or vb < v[w]:
                v[w] = vb
    out = bytearray(b"BAI\x01" + struct.pack("<i", n_ref))
    for r in range(n_ref):
        out += struct.pack("<i", len(bins[r]))
        for b in sorted(bins[r]):
            out += struct.pack("<Ii", b, len(bins[r][b]))
            for vb, ve in bins[r][b]:
                out += struct.pack("<QQ", vb, ve)
        out += struct.pack("<i", len(lin[r]))
        for x in lin[r]:
            out += struct.pack("<Q", x)
    with open(path, "wb") as f:
        f.write(out)


def write_gbai(path, recs, every=64):
    """.gbai (encoding/bam/gindex.go:45-73): gzip of the magic and (ref, pos, seq, voffset)
    entries: the first record of every reference, then every `every`-th record."""
    out = bytearray(GBAI_MAGIC)
    last_ref, last_key, seq = None, None, 0
    for i, (ref, pos, end, vb, ve) in enumerate(recs):
        seq = seq + 1 if (ref, pos) == last_key else 0
        last_key = (ref, pos)
        if ref != last_ref or i % every == 0:
            out += struct.pack("<iiIQ", ref, pos, seq, vb)
        last_ref = ref
    with gzip.open(path, "wb") as f:
        f.write(bytes(out))


def write_bam(path, ref_names, ref_lens, batch, names=None, aux=None, level=1, index=None):
    """batch: bio_amd.records.RecordBatch (coordinate-sorted).  names: list of bytes/str per
    record (default: batch names, else r<i>).  aux: optional list of raw aux byte strings.
    index: "bai" / "gbai" also writes <path>.bai / <path>.gbai."""
    w = BgzfWriter(path, level)
    irecs = []
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(
        f"@SQ\tSN:{n}\tLN:{l}\n" for n, l in zip(ref_names, ref_lens))
    hdr = b"BAM\x01" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(ref_names))
    for n, l in zip(ref_names, ref_lens):
        nb = n.encode() + b"\0"
        hdr += struct.pack("<i", len(nb)) + nb + struct.pack("<i", int(l))
    w.write(hdr)
    b = batch
    have_names = names is None and b.name_off is not None and b.name is not None
    for i in range(b.n):
        if names is not None:
            nm = names[i] if isinstance(names[i], bytes) else str(names[i]).encode()
        elif have_names:
            nm = bytes(b.name[int(b.name_off[i]):int(b.name_off[i + 1])])
        else:
            nm = b"r%d" % i
        nm += b"\0"
        nc, ls = int(b.n_cigar[i]), int(b.l_seq[i])
        cig = b.cigar[int(b.cigar_off[i]):int(b.cigar_off[i]) + nc]
        span = int(sum(int(c) >> 4 for c in cig if (int(c) & 15) in (0, 2, 3, 7, 8)))
        pos = int(b.pos[i])
        body = struct.pack("<iiBBHHHIiii", int(b.ref_id[i]), pos, len(nm), int(b.mapq[i]),
                           _reg2bin(pos, pos + max(span, 1)), nc, int(b.flags[i]), ls,
                           int(b.mate_ref_id[i]), int(b.mate_pos[i]), 0)
        body += nm + np.asarray(cig, np.uint32).tobytes()
        body += bytes(b.seq[int(b.seq_off[i]):int(b.seq_off[i]) + (ls + 1) // 2])
        body += bytes(b.qual[int(b.qual_off[i]):int(b.qual_off[i]) + ls])
        if aux is not None:
            body += aux[i]
        vb = w.tell()
        w.write(struct.pack("<i", len(body)) + body)
        if index:
            irecs.append((int(b.ref_id[i]), pos, pos + max(span, 1), vb, w.tell()))
    w.close()
    if index == "bai":
        write_bai(path + ".bai", len(ref_names), irecs)
    elif index == "gbai":
        write_gbai(path + ".gbai", irecs)


def write_fasta(path, names, seqs, width=60, gz=False):
    """seqs: ASCII strings/bytes."""
    op = gzip.open if gz else open
    with op(path, "wb") as f:
        for n, s in zip(names, seqs):
            if isinstance(s, str):
                s = s.encode()
            f.write(b">" + n.encode() + b" test contig\n")
            for o in range(0, len(s), width):
                f.write(s[o:o + width] + b"\n")


def write_bed(path, intervals, ref_names, gz=False):
    """intervals: list of (ref_id, start, end)"""
    op = gzip.open if gz else open
    with op(path, "wb") as f:
        for r, s, e in intervals:
            f.write(f"{ref_names[r]}\t{s}\t{e}\ttarget\n".encode())


def record_stream(batch, aux_bytes=0):
    """The records of `batch` as they lie in an inflated BAM stream (SAM spec 4.2), for
    bp_submit_bam: (uint8 array, uint32 offsets of each record's block_size field).  Names are
    the batch's (else r<i>); aux_bytes > 0 appends that many bytes of one `XX:Z:` tag to every
    record so that the fields do not end where the record does."""
    b = batch
    n = int(b.n)
    have_names = b.name_off is not None and b.name is not None
    if have_names:
        noff = np.asarray(b.name_off, np.int64)
        nlen = noff[1:n + 1] - noff[:n]
        names = np.asarray(b.name, np.uint8)
    else:
        parts = [b"r%d" % i for i in range(n)]
        nlen = np.array([len(x) for x in parts], np.int64)
        noff = np.concatenate([[0], np.cumsum(nlen)])
        names = np.frombuffer(b"".join(parts), np.uint8)
    if n and nlen.max() > 254:
        raise ValueError("read name longer than 254 bytes")
    nc = np.asarray(b.n_cigar, np.int64)[:n]
    ls = np.asarray(b.l_seq, np.int64)[:n]
    sb = (ls + 1) // 2
    if aux_bytes and aux_bytes < 4:
        raise ValueError("aux_bytes: 0 or at least 4 (tag, type, NUL)")
    body = 32 + nlen + 1 + 4 * nc + sb + ls + aux_bytes
    off = np.concatenate([[0], np.cumsum(4 + body)])
    out = np.zeros(int(off[-1]), np.uint8)
    fixed = np.zeros(n, dtype=np.dtype([("bs", "<u4"), ("ref", "<i4"), ("pos", "<i4"), ("ln", "u1"), ("mapq", "u1"),
                                        ("bin", "<u2"), ("nc", "<u2"), ("flag", "<u2"), ("ls", "<u4"),
                                        ("mref", "<i4"), ("mpos", "<i4"), ("tlen", "<i4")]))
    fixed["bs"] = body
    fixed["ref"] = np.asarray(b.ref_id)[:n]
    fixed["pos"] = np.asarray(b.pos)[:n]
    fixed["ln"] = nlen + 1
    fixed["mapq"] = np.asarray(b.mapq)[:n]
    fixed["bin"] = 4680
    fixed["nc"] = nc
    fixed["flag"] = np.asarray(b.flags)[:n]
    fixed["ls"] = ls
    fixed["mref"] = np.asarray(b.mate_ref_id)[:n]
    fixed["mpos"] = np.asarray(b.mate_pos)[:n]
    out[(off[:n, None] + np.arange(36)[None, :]).reshape(-1)] = fixed.view(np.uint8).reshape(-1)

    def scatter(dst0, src, src0, lens):
        lens = np.asarray(lens, np.int64)
        tot = int(lens.sum())
        if tot == 0:
            return
        first = np.cumsum(lens) - lens
        ar = np.arange(tot)
        out[np.repeat(dst0 - first, lens) + ar] = src[np.repeat(np.asarray(src0, np.int64) - first, lens) + ar]
    at = off[:n] + 36
    scatter(at, names, noff[:n], nlen)
    at = at + nlen + 1
    scatter(at, np.asarray(b.cigar, np.uint32).view(np.uint8), 4 * np.asarray(b.cigar_off, np.int64)[:n], 4 * nc)
    at = at + 4 * nc
    scatter(at, np.asarray(b.seq, np.uint8), np.asarray(b.seq_off, np.int64)[:n], sb)
    at = at + sb
    scatter(at, np.asarray(b.qual, np.uint8), np.asarray(b.qual_off, np.int64)[:n], ls)
    at = at + ls
    if aux_bytes:
        tag = np.frombuffer(b"XXZ" + b"a" * (aux_bytes - 4) + b"\0", np.uint8)
        out[(at[:, None] + np.arange(aux_bytes)[None, :]).reshape(-1)] = np.tile(tag, n)
    return out, off[:n].astype(np.uint32)
