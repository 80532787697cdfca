"""ctypes wrapper around oracle/_build/liboracle.so (the CPU checker; test infrastructure).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from bio_amd._abi import BP_ROW_U32, bp_opts, bp_record_batch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "_build", "liboracle.so")


class orc_shard(C.Structure):
    _fields_ = [("start_ref", C.c_int32), ("end_ref", C.c_int32), ("start", C.c_int32),
                ("end", C.c_int32), ("padding", C.c_int32)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        build()
    lib = C.CDLL(LIB)
    i32p, u32p, u64p = C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    lib.orc_pileup.argtypes = [C.POINTER(bp_opts), C.c_int32, i32p, u64p, i32p,
                               C.POINTER(bp_record_batch), C.c_int32, C.POINTER(orc_shard),
                               C.c_int32, C.c_int64, i32p, i32p, u32p, C.POINTER(C.c_int64),
                               C.c_char_p, C.c_int32]
    lib.orc_pileup.restype = C.c_int
    lib.orc_position_shards.argtypes = [C.c_int32, i32p, C.c_int32, C.c_int32,
                                        C.POINTER(orc_shard), C.c_int32]
    lib.orc_position_shards.restype = C.c_int32
    lib.orc_align_relevant_bases.argtypes = [i32p, C.c_int32, C.c_int32, u32p, C.c_int32,
                                             C.c_int32, i32p, i32p]
    lib.orc_align_relevant_bases.restype = C.c_int
    lib.orc_qual_pass.argtypes = [C.c_int, C.c_int, C.c_int]
    lib.orc_qual_pass.restype = C.c_int
    lib.orc_qual_sum.argtypes = [C.c_int, C.c_int]
    lib.orc_qual_sum.restype = C.c_int
    lib.orc_qual_round_margin.restype = C.c_double
    lib.orc_get_strand.argtypes = [C.c_int32, C.c_int32, C.c_uint16]
    lib.orc_get_strand.restype = C.c_int
    lib.orc_next_exp2.argtypes = [C.c_int32]
    lib.orc_next_exp2.restype = C.c_int32
    lib.orc_unpack_seq.argtypes = [C.POINTER(C.c_uint8), C.POINTER(C.c_uint8), C.c_uint32]
    lib.orc_unpack_seq.restype = None
    lib.orc_murmur3_32.argtypes = [C.c_char_p, C.c_size_t]
    lib.orc_murmur3_32.restype = C.c_uint32
    lib.orc_shard_pad_start.argtypes = [C.c_int32, C.c_int32]
    lib.orc_shard_pad_start.restype = C.c_int32
    lib.orc_shard_pad_end.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.orc_shard_pad_end.restype = C.c_int32
    lib.orc_bed_subset1.argtypes = [i32p, C.c_int32, C.c_int32, C.c_int32, i32p, C.c_int32]
    lib.orc_bed_subset1.restype = C.c_int32
    lib.orc_bed_intersects1.argtypes = [i32p, C.c_int32, C.c_int32, C.c_int32]
    lib.orc_bed_intersects1.restype = C.c_int
    _lib = lib
    return lib


def make_opts(**kw):
    """bp_opts with the reference's DefaultOpts (pileup/snp/pileup.go:56-68)."""
    o = bp_opts()
    o.clip = 0
    o.flag_exclude = 0xF00
    o.mapq = 60
    o.max_read_len = 500
    o.max_read_span = 511
    o.min_base_qual = 0
    o.stitch = 0
    o.ignore_strand = 0
    o.strand_req = 0
    o.device = 0
    for k, v in kw.items():
        if not hasattr(o, k):
            raise KeyError(k)
        setattr(o, k, int(v))
    return o


def bed_arrays(n_ref, bed):
    """bed: {ref_id: [[start,end],...] sorted disjoint} -> (bed_off u64[n_ref+1], bed_ep i32)."""
    off = np.zeros(n_ref + 1, np.uint64)
    eps = []
    for r in range(n_ref):
        iv = bed.get(r, [])
        for s, e in iv:
            eps += [s, e]
        off[r + 1] = len(eps)
    return off, np.array(eps, np.int32) if eps else np.zeros(0, np.int32)


def n_bed_positions(bed):
    return sum(e - s for iv in bed.values() for s, e in iv)


def position_shards(ref_len, shard_size=100000, padding=511):
    lib = load()
    rl = np.ascontiguousarray(ref_len, np.int32)
    n = lib.orc_position_shards(len(rl), rl.ctypes.data_as(C.POINTER(C.c_int32)), shard_size,
                                padding, None, 0)
    arr = (orc_shard * max(n, 1))()
    lib.orc_position_shards(len(rl), rl.ctypes.data_as(C.POINTER(C.c_int32)), shard_size, padding,
                            arr, n)
    return arr, n


def region_shard(ref_id, start0, end):
    """The single unpadded shard of -region mode (pileup.go:1091-1097)."""
    arr = (orc_shard * 1)()
    arr[0] = orc_shard(ref_id, ref_id, start0, end, 0)
    return arr, 1


def bed_subset1(endpoints, start, limit):
    """BEDUnion.Subset on one contig (bedunion.go:753-852) -> list of endpoints."""
    lib = load()
    ep = np.ascontiguousarray(endpoints, np.int32)
    i32p = C.POINTER(C.c_int32)
    out = np.zeros(len(ep) + 4, np.int32)
    n = lib.orc_bed_subset1(ep.ctypes.data_as(i32p), len(ep), start, limit, out.ctypes.data_as(i32p), len(out))
    return out[:n].tolist()


class OracleError(RuntimeError):
    pass


# PerReadFeatures (pileup/snp/row.go:22-31) as the oracle and the C ABI lay them out
FEAT_DTYPE = np.dtype([("dist5p", "<u2"), ("fraglen", "<u2"), ("qual", "u1"), ("strand", "u1")])


def pileup(opts, ref_len, bed, batch, shards=None, n_shards=None, parallelism=1):
    """Run the oracle.  bed = {ref_id: [[s,e],...]}.  shards default to the -bed mode
    100-kb position shards with Padding = max_read_span (pileup.go:1073-1084).
    Returns (ref_id[P], pos[P], vals[P,11])."""
    lib = load()
    rl = np.ascontiguousarray(ref_len, np.int32)
    n_ref = len(rl)
    if shards is None:
        shards, n_shards = position_shards(rl, 100000, opts.max_read_span)
    off, ep = bed_arrays(n_ref, bed)
    cap = n_bed_positions(bed) + 1
    out_ref = np.zeros(cap, np.int32)
    out_pos = np.zeros(cap, np.int32)
    out_vals = np.zeros((cap, BP_ROW_U32), np.uint32)
    n_rows = C.c_int64(0)
    err = C.create_string_buffer(512)
    cb = batch.as_c()
    i32p, u32p, u64p = C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    rc = lib.orc_pileup(C.byref(opts), n_ref, rl.ctypes.data_as(i32p), off.ctypes.data_as(u64p),
                        ep.ctypes.data_as(i32p), C.byref(cb), n_shards, shards, parallelism, cap,
                        out_ref.ctypes.data_as(i32p), out_pos.ctypes.data_as(i32p),
                        out_vals.ctypes.data_as(u32p), C.byref(n_rows), err, 512)
    if rc != 0:
        raise OracleError(err.value.decode())
    n = n_rows.value
    return out_ref[:n], out_pos[:n], out_vals[:n]


def whole_contig_column_sums(opts, L, batch, threads=16):
    """The 11 column sums (Depth, A+ .. X-) of the oracle's rows over ALL of one contig, worked as
    `threads` position ranges on one thread each - independent jobs with a read halo, like the
    reference's own (pileup.go:862-932); with stitching the BED is padded by two spans as well and
    the padding's rows are dropped, which gives the one-job rows for any cut.
    -> (sums[11] as Python ints, seconds)."""
    import time
    from concurrent.futures import ThreadPoolExecutor
    span = int(opts.max_read_span)
    pad_r, pad_b = (2 * span + 2, 2 * span + 2) if opts.stitch else (span + 1, 0)
    edges = [L * i // threads for i in range(threads + 1)]

    def one(i):
        s, e = edges[i], edges[i + 1]
        a = int(np.searchsorted(batch.pos, max(0, s - pad_r), "left"))
        b = int(np.searchsorted(batch.pos, e + (pad_r if opts.stitch else 0), "left"))
        bs, be = max(0, s - pad_b), min(L, e + pad_b)
        _, _, v = pileup(opts, [L], {0: [[bs, be]]}, batch.slice(a, b), parallelism=1)
        return v[s - bs:s - bs + (e - s)].sum(axis=0, dtype=np.uint64)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(threads) as ex:
        tot = sum(ex.map(one, range(threads)))
    return [int(x) for x in tot], time.perf_counter() - t0


def pileup_per_read(opts, ref_len, bed, batch, shards=None, n_shards=None, parallelism=1, feat_cap=None):
    """The oracle with perReadNeeded set: (ref, pos, vals, off[4P+1], feats[FEAT_DTYPE])."""
    lib = load()
    rl = np.ascontiguousarray(ref_len, np.int32)
    n_ref = len(rl)
    if shards is None:
        shards, n_shards = position_shards(rl, 100000, opts.max_read_span)
    off, ep = bed_arrays(n_ref, bed)
    cap = n_bed_positions(bed) + 1
    if feat_cap is None:
        feat_cap = int(batch.l_seq.astype(np.int64).sum()) + 1
    out_ref = np.zeros(cap, np.int32)
    out_pos = np.zeros(cap, np.int32)
    out_vals = np.zeros((cap, BP_ROW_U32), np.uint32)
    out_off = np.zeros(4 * cap + 1, np.uint64)
    out_feat = np.zeros(feat_cap, FEAT_DTYPE)
    assert FEAT_DTYPE.itemsize == 6
    n_rows, n_feat = C.c_int64(0), C.c_int64(0)
    err = C.create_string_buffer(512)
    cb = batch.as_c()
    i32p, u32p, u64p = C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
    lib.orc_pileup_per_read.restype = C.c_int
    rc = lib.orc_pileup_per_read(C.byref(opts), n_ref, rl.ctypes.data_as(i32p), off.ctypes.data_as(u64p),
                                 ep.ctypes.data_as(i32p), C.byref(cb), C.c_int32(n_shards), shards,
                                 C.c_int32(parallelism), C.c_int64(cap), out_ref.ctypes.data_as(i32p),
                                 out_pos.ctypes.data_as(i32p), out_vals.ctypes.data_as(u32p), C.byref(n_rows),
                                 out_off.ctypes.data_as(u64p), out_feat.ctypes.data_as(C.c_void_p),
                                 C.c_int64(feat_cap), C.byref(n_feat), err, C.c_int32(512))
    if rc != 0:
        raise OracleError(err.value.decode())
    n = n_rows.value
    return out_ref[:n], out_pos[:n], out_vals[:n], out_off[:4 * n + 1], out_feat[:n_feat.value]


class orc_bam(C.Structure):
    """oracle/bam_decode.c"""
    _fields_ = [("n", C.c_int64), ("n_ref", C.c_int32)] + \
               [(k, C.POINTER(C.c_int32)) for k in ("ref_id", "pos", "mate_ref_id", "mate_pos")] + \
               [("flags", C.POINTER(C.c_uint16)), ("mapq", C.POINTER(C.c_uint8)),
                ("n_cigar", C.POINTER(C.c_uint32)), ("l_seq", C.POINTER(C.c_uint32))] + \
               [(k, C.POINTER(C.c_uint64)) for k in ("cigar_off", "seq_off", "qual_off", "name_off")] + \
               [("cigar", C.POINTER(C.c_uint32)), ("seq", C.POINTER(C.c_uint8)), ("qual", C.POINTER(C.c_uint8)),
                ("name", C.POINTER(C.c_uint8))] + \
               [(k, C.c_uint64) for k in ("n_cigar_ops", "n_seq_bytes", "n_qual_bytes", "n_name_bytes")]


def bam_decode(path):
    """BAM file -> RecordBatch with the oracle's own single-threaded BGZF + record decoder
    (oracle/bam_decode.c); returns (batch, n_ref)."""
    from bio_amd import records as R
    lib_path = os.path.join(ORACLE_DIR, "_build", "libbamdecode.so")
    if not os.path.exists(lib_path):
        build()
    lib = C.CDLL(lib_path)
    lib.orc_bam_decode.argtypes = [C.c_char_p, C.POINTER(orc_bam), C.c_char_p, C.c_int]
    lib.orc_bam_decode.restype = C.c_int
    lib.orc_bam_free.argtypes = [C.POINTER(orc_bam)]
    lib.orc_bam_free.restype = None
    b = orc_bam()
    err = C.create_string_buffer(256)
    if lib.orc_bam_decode(path.encode(), C.byref(b), err, 256) != 0:
        raise OracleError(err.value.decode())
    n = int(b.n)

    def arr(p, cnt):
        return np.ctypeslib.as_array(p, shape=(max(int(cnt), 1),))[:int(cnt)].copy()
    kw = {k: arr(getattr(b, k), n) for k in ("ref_id", "pos", "mate_ref_id", "mate_pos", "flags", "mapq", "n_cigar",
                                              "l_seq", "cigar_off", "seq_off", "qual_off")}
    kw["name_off"] = arr(b.name_off, n + 1)
    kw["cigar"] = arr(b.cigar, b.n_cigar_ops)
    kw["seq"] = arr(b.seq, b.n_seq_bytes)
    kw["qual"] = arr(b.qual, b.n_qual_bytes)
    kw["name"] = arr(b.name, b.n_name_bytes)
    n_ref = int(b.n_ref)
    lib.orc_bam_free(C.byref(b))
    return R.RecordBatch(**kw), n_ref
