"""Python face of the C++ host layers in libbiopileup.so (include/bio_pileup_host.h): the
snp.Pileup driver (pileup/snp/pileup.go:970-1130), the bio-pileup command line
(cmd/bio-pileup/main.go) and the parsers they are built from.  Everything here calls the
native library; nothing is computed in Python."""
import ctypes as C

import numpy as np

from . import _abi
from .snp import Opts, PileupError

# column-set bits (pileup.go:134-157)
COL_DPREF, COL_DPALT, COL_ENDDISTS, COL_QUALS, COL_FRAGLENS, COL_STRANDS, COL_HIGHQ, COL_LOWQ = (
    1, 2, 4, 8, 16, 32, 64, 128)
COLS_DEFAULT = COL_DPREF | COL_HIGHQ | COL_LOWQ


def _err():
    return C.create_string_buffer(1024)


def _b(s):
    return None if s is None else s.encode()


def pileup(xampath, fapath, fmt="tsv", out_prefix="bio-pileup", opts=None, device=0, devices=0):
    """snp.Pileup (pileup.go:970).  Raises PileupError with the reference's wording."""
    o = opts or Opts()
    lib = _abi.load()
    co = _abi.bio_pileup_opts()
    lib.bio_pileup_default_opts(C.byref(co))
    co.bed_path, co.region, co.bam_index_path = _b(o.BedPath), _b(o.Region), _b(o.BamIndexPath)
    co.clip, co.cols, co.flag_exclude, co.mapq = o.Clip, _b(o.Cols), o.FlagExclude, o.Mapq
    co.max_read_len, co.max_read_span = o.MaxReadLen, o.MaxReadSpan
    co.min_bag_depth, co.min_base_qual, co.parallelism = o.MinBagDepth, o.MinBaseQual, o.Parallelism
    co.per_strand, co.remove_sq, co.stitch = int(o.PerStrand), int(o.RemoveSq), int(o.Stitch)
    co.temp_dir, co.device, co.devices = _b(o.TempDir), device, devices
    e = _err()
    if lib.bio_pileup_run(_b(xampath), _b(fapath), _b(fmt), _b(out_prefix), C.byref(co), e, len(e)):
        raise PileupError(-1, e.value.decode())


def main(argv):
    """cmd/bio-pileup main(); argv[0] is the program name.  Returns the exit status."""
    lib = _abi.load()
    arr = (C.c_char_p * (len(argv) + 1))(*[a.encode() for a in argv], None)
    return lib.bio_pileup_main(len(argv), arr)


def parse_region(region):
    """interval.ParseRegionString -> (ref_name, start0, end)"""
    lib = _abi.load()
    name = C.create_string_buffer(4096)
    s, t, e = C.c_int32(), C.c_int32(), _err()
    if lib.bio_parse_region(region.encode(), name, len(name), C.byref(s), C.byref(t), e, len(e)):
        raise ValueError(e.value.decode())
    return name.value.decode(), s.value, t.value


def parse_cols(cols, default=COLS_DEFAULT):
    lib = _abi.load()
    out, e = C.c_int32(), _err()
    if lib.bio_parse_cols(cols.encode(), default, C.byref(out), e, len(e)):
        raise ValueError(e.value.decode())
    return out.value


class Bed:
    """interval.BEDUnion loaded from a path against a list of reference names."""

    def __init__(self, path, ref_names, subset=None):
        self._h = None
        lib = _abi.load()
        self._lib = lib
        names = (C.c_char_p * len(ref_names))(*[n.encode() for n in ref_names])
        sub = (C.c_int32 * 4)(*subset) if subset is not None else None
        h, e = C.c_void_p(), _err()
        if lib.bio_bed_load(path.encode(), len(ref_names), names, sub, C.byref(h), e, len(e)):
            raise ValueError(e.value.decode())
        self._h = h

    def endpoints(self, ref_id):
        """EndpointsByID; None for a reference the BED never mentions."""
        n = self._lib.bio_bed_endpoints(self._h, ref_id, None, 0)
        if n < 0:
            return None
        out = np.zeros(n, np.int32)
        self._lib.bio_bed_endpoints(self._h, ref_id, out.ctypes.data_as(C.POINTER(C.c_int32)), n)
        return out

    def intersects(self, ref_id, start, limit):
        return bool(self._lib.bio_bed_intersects(self._h, ref_id, start, limit))

    def close(self):
        if self._h:
            self._lib.bio_bed_free(self._h)
            self._h = None

    __del__ = close


def fasta_seq8(path, name):
    lib = _abi.load()
    e = _err()
    n = lib.bio_fasta_seq8(path.encode(), name.encode(), None, 0, e, len(e))
    if n < 0:
        raise ValueError(e.value.decode())
    out = np.zeros(n, np.uint8)
    lib.bio_fasta_seq8(path.encode(), name.encode(), out.ctypes.data_as(C.POINTER(C.c_uint8)), n, e, len(e))
    return out


def bam_scan(path):
    """-> (n_records, sum l_seq, sum n_cigar, n_ref) as the driver's BAM reader decodes them"""
    lib = _abi.load()
    a, b, c, d, e = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int32(), _err()
    if lib.bio_bam_scan(path.encode(), C.byref(a), C.byref(b), C.byref(c), C.byref(d), e, len(e)):
        raise ValueError(e.value.decode())
    return a.value, b.value, c.value, d.value


def bam_scan_chunks(path):
    """-> (n_records, n_chunks, hash record by record, hash chunk by chunk, sorted, fields_ok,
    n_rewalked): the driver's reader run both ways over one file"""
    lib = _abi.load()
    n, nc, h1, h2, srt, ok, e = C.c_int64(), C.c_int64(), C.c_uint64(), C.c_uint64(), C.c_int32(), C.c_int32(), _err()
    rw = C.c_int64()
    if lib.bio_bam_scan_chunks(path.encode(), C.byref(n), C.byref(nc), C.byref(h1), C.byref(h2), C.byref(srt),
                               C.byref(ok), C.byref(rw), e, len(e)):
        raise ValueError(e.value.decode())
    return n.value, nc.value, h1.value, h2.value, bool(srt.value), bool(ok.value), rw.value


def bam_seek_scan(bam, index, ref_id, pos):
    """-> (found, first_ref, first_pos, n_skipped, n_at_or_after): the driver's index lookup and
    seek, then a read to the end of the file"""
    lib = _abi.load()
    f, fr, fp, sk, af, e = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int64(), C.c_int64(), _err()
    if lib.bio_bam_seek_scan(bam.encode(), index.encode(), ref_id, pos, C.byref(f), C.byref(fr), C.byref(fp),
                             C.byref(sk), C.byref(af), e, len(e)):
        raise ValueError(e.value.decode())
    return bool(f.value), fr.value, fp.value, sk.value, af.value


def bam_write(path, ref_names, ref_lens, batch, level=1, threads=8):
    """Native BAM writer (fast path for tools; bamio.write_bam is the pure-Python one)."""
    lib = _abi.load()
    names = (C.c_char_p * len(ref_names))(*[n.encode() for n in ref_names])
    lens = (C.c_int32 * len(ref_lens))(*[int(x) for x in ref_lens])
    cb = batch.as_c()
    e = _err()
    if lib.bio_bam_write(path.encode(), len(ref_names), names, lens, C.byref(cb), level, threads, e, len(e)):
        raise ValueError(e.value.decode())


def write_rows(fmt, out_prefix, cols, ref_names, ref_seq8, ref_id, pos, vals, off=None, feats=None):
    """ConvertPileupRowsToTSV / ConvertPileupRowsToBasestrandTSV (output.go:42-300, :376-575) over
    rows in memory, with the product's writers.  ref_seq8: one uint8 array (or None) per reference."""
    lib = _abi.load()
    names = (C.c_char_p * len(ref_names))(*[n.encode() for n in ref_names])
    seqs = [None if s is None else np.ascontiguousarray(s, np.uint8) for s in ref_seq8]
    sp = (C.c_void_p * len(seqs))(*[None if s is None else s.ctypes.data for s in seqs])
    sl = (C.c_int64 * len(seqs))(*[0 if s is None else len(s) for s in seqs])
    ref_id = np.ascontiguousarray(ref_id, np.int32)
    pos = np.ascontiguousarray(pos, np.int32)
    vals = np.ascontiguousarray(vals, np.uint32)
    i32p = C.POINTER(C.c_int32)
    offp = featp = None
    if off is not None:
        off = np.ascontiguousarray(off, np.uint64)
        feats = np.ascontiguousarray(feats)
        offp, featp = off.ctypes.data_as(C.POINTER(C.c_uint64)), feats.ctypes.data
    e = _err()
    if lib.bio_pileup_write_rows(fmt.encode(), out_prefix.encode(), _b(cols), len(ref_names), names, sp, sl, len(pos),
                                 ref_id.ctypes.data_as(i32p), pos.ctypes.data_as(i32p),
                                 vals.ctypes.data_as(C.POINTER(C.c_uint32)), offp, featp, e, len(e)):
        raise PileupError(-1, e.value.decode())


def row_fields(vals, list_len, per_read_needed):
    """PileupRow.FieldsPresent as flushToInternal sets it (pileup.go:518-570)."""
    lib = _abi.load()
    v = np.ascontiguousarray(vals, np.uint32)
    ll = np.ascontiguousarray(list_len, np.uint32)
    u32p = C.POINTER(C.c_uint32)
    return int(lib.bio_pileup_row_fields(v.ctypes.data_as(u32p), ll.ctypes.data_as(u32p), int(per_read_needed)))


def marshal_pileup_row(fields, ref_id, pos, vals, lists=None):
    """MarshalPileupRow (row.go:113-175) -> bytes.  lists: four arrays of snp.PER_READ_DTYPE."""
    from .snp import PER_READ_DTYPE
    lib = _abi.load()
    v = np.ascontiguousarray(vals, np.uint32)
    ls = [np.ascontiguousarray(x if x is not None else [], PER_READ_DTYPE) for x in (lists or [None] * 4)]
    ll = np.array([len(x) for x in ls], np.uint32)
    lp = (C.c_void_p * 4)(*[x.ctypes.data if len(x) else None for x in ls])
    u32p = C.POINTER(C.c_uint32)
    n = lib.bio_marshal_pileup_row(fields, ref_id, pos, v.ctypes.data_as(u32p), ll.ctypes.data_as(u32p), lp, None, 0)
    if n < 0:
        raise ValueError("bio_marshal_pileup_row: bad argument")
    out = C.create_string_buffer(int(n))
    lib.bio_marshal_pileup_row(fields, ref_id, pos, v.ctypes.data_as(u32p), ll.ctypes.data_as(u32p), lp, out, n)
    return out.raw


def unmarshal_pileup_row(data):
    """unmarshalPileupRow (row.go:179-227) -> (fields, ref_id, pos, vals[11], [4 lists])"""
    from .snp import PER_READ_DTYPE
    lib = _abi.load()
    f, r, p = C.c_uint32(), C.c_uint32(), C.c_uint32()
    vals = np.zeros(_abi.BP_ROW_U32, np.uint32)
    ll = np.zeros(4, np.uint32)
    u32p = C.POINTER(C.c_uint32)
    buf = C.create_string_buffer(bytes(data), len(data))
    feats = np.zeros(max(len(data) // 6, 1), PER_READ_DTYPE)
    n = lib.bio_unmarshal_pileup_row(buf, len(data), C.byref(f), C.byref(r), C.byref(p), vals.ctypes.data_as(u32p),
                                     ll.ctypes.data_as(u32p), feats.ctypes.data, len(feats))
    if n < 0:
        raise ValueError("unmarshalPileupRow: truncated record")
    lists, o = [], 0
    for b in range(4):
        lists.append(feats[o:o + int(ll[b])].copy())
        o += int(ll[b])
    return f.value, r.value, p.value, vals, lists


def marshal_basestrand(ref_id, pos, vals):
    """marshalBaseStrand (basestrand.go:103-135) over n rows -> bytes (40 per row)"""
    lib = _abi.load()
    ref_id = np.ascontiguousarray(ref_id, np.int32)
    pos = np.ascontiguousarray(pos, np.int32)
    vals = np.ascontiguousarray(vals, np.uint32)
    out = C.create_string_buffer(40 * len(pos) + 1)
    i32p = C.POINTER(C.c_int32)
    lib.bio_marshal_basestrand(len(pos), ref_id.ctypes.data_as(i32p), pos.ctypes.data_as(i32p),
                               vals.ctypes.data_as(C.POINTER(C.c_uint32)), out)
    return out.raw[:40 * len(pos)]


def unmarshal_basestrand(data):
    lib = _abi.load()
    n = len(data) // 40
    ref_id, pos, c8 = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros((n, 8), np.uint32)
    buf = C.create_string_buffer(bytes(data), len(data))
    i32p = C.POINTER(C.c_int32)
    lib.bio_unmarshal_basestrand(n, buf, ref_id.ctypes.data_as(i32p), pos.ctypes.data_as(i32p),
                                 c8.ctypes.data_as(C.POINTER(C.c_uint32)))
    return ref_id, pos, c8


def basestrand_rio_trailer(num_piles):
    lib = _abi.load()
    out = C.create_string_buffer(16)
    lib.bio_basestrand_rio_trailer(num_piles, out)
    return out.raw


def parse_basestrand_rio_trailer(data):
    lib = _abi.load()
    n, e = C.c_int64(), _err()
    buf = C.create_string_buffer(bytes(data), len(data))
    if lib.bio_parse_basestrand_rio_trailer(buf, len(data), C.byref(n), e, len(e)):
        raise ValueError(e.value.decode())
    return n.value


def basestrand_piles_to_tsv(ref_id, pos, counts8, ref_names, ref_bases=None):
    """WriteBaseStrandToTSV (ref_bases None) / WriteBaseStrandTsv (basestrand.go:204-231, :282-290)"""
    lib = _abi.load()
    ref_id = np.ascontiguousarray(ref_id, np.int32)
    pos = np.ascontiguousarray(pos, np.int32)
    c8 = np.ascontiguousarray(counts8, np.uint32)
    names = (C.c_char_p * len(ref_names))(*[n.encode() for n in ref_names])
    i32p = C.POINTER(C.c_int32)
    args = (len(pos), ref_id.ctypes.data_as(i32p), pos.ctypes.data_as(i32p), c8.ctypes.data_as(C.POINTER(C.c_uint32)),
            len(ref_names), names, None if ref_bases is None else "".join(ref_bases).encode())
    n = lib.bio_basestrand_piles_to_tsv(*args, None, 0)
    if n < 0:
        raise ValueError("bio_basestrand_piles_to_tsv: reference id out of range")
    out = C.create_string_buffer(int(n) + 1)
    lib.bio_basestrand_piles_to_tsv(*args, out, n)
    return out.raw[:n]


def _tsv_rows(call, *texts):
    lib = _abi.load()
    e = _err()
    args = []
    for t in texts:
        args += [t, len(t)]
    n = call(lib)(*args, None, 0, e, len(e))
    if n < 0:
        raise ValueError(e.value.decode())
    rows = (_abi.bio_basestrand_tsv_row * max(int(n), 1))()
    call(lib)(*args, rows, n, e, len(e))
    keys = ("FwdA", "RevA", "FwdC", "RevC", "FwdG", "RevG", "FwdT", "RevT")
    return [dict(Chr=r.chr.decode(), Pos=r.pos, Ref=r.ref.decode(), **dict(zip(keys, r.counts))) for r in rows[:n]]


def read_basestrand_tsv(text):
    """ReadBaseStrandTsv (basestrand.go:262-279) -> list of BaseStrandTsvRow dicts"""
    return _tsv_rows(lambda lib: lib.bio_read_basestrand_tsv, bytes(text))


def read_single_strand_basestrand_tsv(fwd, rev):
    """ReadSingleStrandBaseStrandTsv (basestrand.go:309-367)"""
    return _tsv_rows(lambda lib: lib.bio_read_single_strand_basestrand_tsv, bytes(fwd), bytes(rev))


def chr_id(name):
    """ChrId (basestrand.go:401-422)"""
    lib = _abi.load()
    v, e = C.c_int32(), _err()
    if lib.bio_chr_id(name.encode(), C.byref(v), e, len(e)):
        raise ValueError(e.value.decode())
    return v.value
