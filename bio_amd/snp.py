"""Host-side mirror of the reference's pileup/snp interface for the device path.

Opts / DEFAULT_OPTS mirror snp.Opts / snp.DefaultOpts (pileup/snp/pileup.go:36-68);
PileupJob is one job of pileupSNPMain (pileup.go:862-939) driven through the C ABI
(include/bio_pileup.h).  Plumbing only: every count is computed by libbiopileup.so.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _abi


@dataclass
class Opts:
    """snp.Opts (pileup/snp/pileup.go:36-54)."""
    BedPath: str = ""
    Region: str = ""
    BamIndexPath: str = ""
    Clip: int = 0
    Cols: str = ""
    FlagExclude: int = 0xF00
    Mapq: int = 60
    MaxReadLen: int = 500
    MaxReadSpan: int = 511
    MinBagDepth: int = 0
    MinBaseQual: int = 0
    Parallelism: int = 0
    PerStrand: bool = False
    RemoveSq: bool = False
    Stitch: bool = False
    TempDir: str = ""


DEFAULT_OPTS = Opts()

FORMATS = ("basestrand-rio", "basestrand-tsv", "basestrand-tsv-bgz", "tsv", "tsv-bgz")


# bp_per_read = PerReadFeatures (pileup/snp/row.go:22-31)
PER_READ_DTYPE = np.dtype([("dist5p", "<u2"), ("fraglen", "<u2"), ("qual", "u1"), ("strand", "u1")])


class PileupError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


def device_opts(opts, fmt="basestrand-tsv", strand_req=_abi.STRAND_NONE, device=0):
    """snp.Opts + output format -> bp_opts, as pileupSNPMain derives pileupContext
    (pileup.go:885-892): ignoreStrand iff the format is tsv/tsv-bgz."""
    if fmt not in FORMATS:
        raise PileupError(_abi.BP_EINVAL, "Pileup: unrecognized format= argument")
    o = _abi.bp_opts()
    o.clip = opts.Clip
    o.flag_exclude = opts.FlagExclude
    o.mapq = opts.Mapq
    o.max_read_len = opts.MaxReadLen
    o.max_read_span = opts.MaxReadSpan
    o.min_base_qual = opts.MinBaseQual
    o.stitch = int(opts.Stitch)
    o.ignore_strand = int(fmt in ("tsv", "tsv-bgz"))
    o.strand_req = strand_req
    o.device = device
    return o


class PileupJob:
    """One device job: create -> set_bed -> submit* -> finish -> rows."""

    def __init__(self, bopts, ref_len, ref_names=None):
        self.lib = _abi.load()
        self.h = C.c_void_p()
        rl = np.ascontiguousarray(ref_len, np.int32)
        self.n_ref = len(rl)
        names = None
        if ref_names is not None:
            names = (C.c_char_p * len(ref_names))(*[n.encode() for n in ref_names])
        rc = self.lib.bp_create(C.byref(bopts), len(rl), rl.ctypes.data_as(C.POINTER(C.c_int32)),
                                names, C.byref(self.h))
        if rc != 0:
            raise PileupError(rc, self.lib.bp_last_error(None).decode())

    def _check(self, rc):
        if rc != 0:
            raise PileupError(rc, self.lib.bp_last_error(self.h).decode())

    def reserve(self, n_records, payload_bytes):
        self._check(self.lib.bp_reserve(self.h, n_records, payload_bytes))

    def set_bed(self, bed):
        """bed: {ref_id: [[start,end],...]} or {ref_id: flat endpoint array}."""
        for ref_id, iv in bed.items():
            ep = np.ascontiguousarray(np.asarray(iv, np.int32).reshape(-1))
            self._check(self.lib.bp_set_bed(self.h, ref_id, ep.ctypes.data_as(C.POINTER(C.c_int32)),
                                            len(ep)))

    def submit(self, batch):
        cb = batch.as_c()
        self._check(self.lib.bp_submit(self.h, C.byref(cb)))

    def submit_bam(self, stream, rec_off):
        """records as they lie in an inflated BAM stream (uint8 array) + the offset of each one's
        block_size field (uint32 array): packed on the device (bp_submit_bam)"""
        stream = np.ascontiguousarray(stream, np.uint8)
        rec_off = np.ascontiguousarray(rec_off, np.uint32)
        self._check(self.lib.bp_submit_bam(self.h, stream.ctypes.data, stream.size,
                                           rec_off.ctypes.data_as(C.POINTER(C.c_uint32)), rec_off.size))

    def submit_bam_device(self, stream_dev, n_bytes, cols, first, n):
        """records [first, first+n) of a device-side scan (bgzf.Inflater.scan's bp_bam_columns) of
        the stream at device address stream_dev: nothing of them is read on the host"""
        self._check(self.lib.bp_submit_bam_device(self.h, stream_dev, n_bytes, C.byref(cols), first, n))

    def finish(self):
        self._check(self.lib.bp_finish(self.h))

    def compute(self):
        self._check(self.lib.bp_compute(self.h))

    def compute_static(self):
        """compute() with the kernels' scratch in the library's static device array: allocates nothing"""
        self._check(self.lib.bp_compute_static(self.h))

    @property
    def num_rows(self):
        return int(self.lib.bp_num_rows(self.h))

    def rows(self, first=0, n=None):
        if n is None:
            n = self.num_rows - first
        ref = np.zeros(n, np.int32)
        pos = np.zeros(n, np.int32)
        vals = np.zeros((n, _abi.BP_ROW_U32), np.uint32)
        i32p = C.POINTER(C.c_int32)
        self._check(self.lib.bp_rows(self.h, first, n, ref.ctypes.data_as(i32p),
                                     pos.ctypes.data_as(i32p),
                                     vals.ctypes.data_as(C.POINTER(C.c_uint32))))
        return ref, pos, vals

    def per_read(self, first=0, n=None):
        """Per-read feature lists of rows [first, first+n) (bp_per_read_*): (off[4n+1] relative
        to the returned array, feats[PER_READ_DTYPE])."""
        if n is None:
            n = self.num_rows - first
        self._check(self.lib.bp_per_read_build(self.h))
        off = np.zeros(4 * n + 1, np.uint64)
        self._check(self.lib.bp_per_read_offsets(self.h, first, n, off.ctypes.data_as(C.POINTER(C.c_uint64))))
        lo, hi = int(off[0]), int(off[-1])
        feats = np.zeros(hi - lo, PER_READ_DTYPE)
        self._check(self.lib.bp_per_read_features(self.h, lo, hi - lo, feats.ctypes.data_as(C.c_void_p)))
        return off - off[0], feats

    def set_ref_seq8(self, ref_id, start, seq8):
        """Seq8 bases (numpy uint8) of positions [start, start+len) of one reference, for render_text"""
        a = np.ascontiguousarray(seq8, np.uint8)
        self._check(self.lib.bp_set_ref_seq8(self.h, ref_id, start, a.ctypes.data, len(a)))

    def render_text(self, first=0, n=None, kind=0, cols=0, bgzf=None):
        """Output lines of rows [first, first+n), rendered on the device (bytes).  kind: 0
        basestrand-tsv, 1 .ref.tsv, 2 .alt.tsv; cols: the -cols bits (1 dpref, 2 dpalt, 64 highq,
        128 lowq).  bgzf: the lines as BGZF members compressed on the device instead
        (bp_render_fetch_bgzf) - "one" in one call with a buffer of bp_render_bgzf_bound bytes, "two"
        asking for the length first."""
        if n is None:
            n = self.num_rows - first
        if bgzf:
            ln = C.c_size_t(0)
            self._check(self.lib.bp_render_text(self.h, kind, cols, first, n, None, 0, C.byref(ln)))
            clen = C.c_size_t(0)
            if bgzf == "two":
                self._check(self.lib.bp_render_fetch_bgzf(self.h, None, 0, C.byref(clen)))
                cap = clen.value
            else:
                cap = int(self.lib.bp_render_bgzf_bound(self.h))
            buf = C.create_string_buffer(max(cap, 1))
            self._check(self.lib.bp_render_fetch_bgzf(self.h, buf, cap, C.byref(clen)))
            return buf.raw[:clen.value]
        cap = int(self.lib.bp_render_text_bound(self.h, kind, n))
        buf = C.create_string_buffer(max(cap, 1))
        ln = C.c_size_t(0)
        self._check(self.lib.bp_render_text(self.h, kind, cols, first, n, buf, cap, C.byref(ln)))
        return buf.raw[:ln.value]

    def rows_export(self, first, n, dst_device_ptr):
        """rows [first, first+n) -> n*11 uint32 at a device address (boundary exchange)"""
        self._check(self.lib.bp_rows_export(self.h, first, n, dst_device_ptr))

    def rows_accumulate(self, first, n, src_device_ptr):
        """rows [first, first+n) += n*11 uint32 at a device address (boundary exchange)"""
        self._check(self.lib.bp_rows_accumulate(self.h, first, n, src_device_ptr))

    def rows_export_async(self, first, n, dst_device_ptr):
        """rows_export, only enqueued on the job's stream"""
        self._check(self.lib.bp_rows_export_async(self.h, first, n, dst_device_ptr))

    def rows_accumulate_async(self, first, n, src_device_ptr):
        """rows_accumulate, only enqueued on the job's stream"""
        self._check(self.lib.bp_rows_accumulate_async(self.h, first, n, src_device_ptr))

    @property
    def stream(self):
        """the job's hipStream_t as an integer (torch.cuda.ExternalStream takes it)"""
        p = C.c_void_p()
        self._check(self.lib.bp_stream(self.h, C.byref(p)))
        return p.value or 0

    def column_sums(self):
        s = (C.c_uint64 * _abi.BP_ROW_U32)()
        self._check(self.lib.bp_column_sums(self.h, s))
        return [int(x) for x in s]

    def timing(self):
        t = _abi.bp_timing()
        self.lib.bp_get_timing(self.h, C.byref(t))
        return t

    def algorithmic_bytes(self):
        return int(self.lib.bp_algorithmic_bytes(self.h))

    def close(self):
        if self.h:
            self.lib.bp_destroy(self.h)
            self.h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run_job(bopts, ref_len, bed, batches, ref_names=None):
    """Convenience: one job over a list of batches; returns (ref_id, pos, vals)."""
    with PileupJob(bopts, ref_len, ref_names) as job:
        job.set_bed(bed)
        for b in batches:
            job.submit(b)
        job.finish()
        return job.rows()
