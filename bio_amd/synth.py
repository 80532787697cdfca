"""ctypes wrapper for libbpsynth.so, the seeded synthetic read generator (bench/test support)."""
import ctypes as C
import os

import numpy as np

from . import _abi
from .records import RecordBatch

SEED = 0x62696F70696C6575  # "biopileu", SURVEY.md section 8(d)


class bps_region(C.Structure):
    _fields_ = [("ref_id", C.c_int32), ("start", C.c_int32), ("end", C.c_int32),
                ("coverage", C.c_double)]


class bps_params(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64), ("read_len", C.c_int32),
        ("insert_mean", C.c_double), ("insert_sd", C.c_double),
        ("insert_min", C.c_int32), ("insert_max", C.c_int32),
        ("frac_softclip", C.c_double), ("frac_del", C.c_double), ("frac_ins", C.c_double),
        ("frac_hardclip", C.c_double), ("sub_rate", C.c_double),
        ("frac_low_mapq", C.c_double), ("frac_dup", C.c_double), ("frac_secondary", C.c_double),
        ("frac_mate_other_ref", C.c_double),
        ("with_names", C.c_int32), ("threads", C.c_int32),
        ("n_regions", C.c_int32), ("regions", C.POINTER(bps_region)),
    ]


_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_abi.SYNTH_LIB_PATH):
            raise RuntimeError(f"{_abi.SYNTH_LIB_PATH} not built (run `make synth`)")
        lib = C.CDLL(_abi.SYNTH_LIB_PATH)
        lib.bps_generate.argtypes = [C.POINTER(bps_params), C.POINTER(C.c_void_p)]
        lib.bps_generate.restype = C.c_int
        lib.bps_view.argtypes = [C.c_void_p, C.POINTER(_abi.bp_record_batch)]
        lib.bps_view.restype = None
        lib.bps_aligned_bases.argtypes = [C.c_void_p]
        lib.bps_aligned_bases.restype = C.c_int64
        lib.bps_payload_sizes.argtypes = [C.c_void_p] + [C.POINTER(C.c_int64)] * 4
        lib.bps_payload_sizes.restype = C.c_int64
        lib.bps_ref_nibble.argtypes = [C.c_uint64, C.c_int32, C.c_int64]
        lib.bps_ref_nibble.restype = C.c_uint8
        lib.bps_free.argtypes = [C.c_void_p]
        lib.bps_free.restype = None
        _lib = lib
    return _lib


class SynthBatch:
    """A generated batch; memory is owned by the C++ side until close()."""

    def __init__(self, handle):
        self._h = handle
        lib = load()
        self.view = _abi.bp_record_batch()
        lib.bps_view(handle, C.byref(self.view))
        self.n = int(self.view.n)
        self.aligned_bases = int(lib.bps_aligned_bases(handle))
        nc, ns, nq, nn = (C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64())
        lib.bps_payload_sizes(handle, C.byref(nc), C.byref(ns), C.byref(nq), C.byref(nn))
        self._sizes = (nc.value, ns.value, nq.value, nn.value)

    def as_c(self):
        return self.view

    def to_record_batch(self):
        """Zero-copy numpy view (valid until close())."""
        v, n = self.view, self.n
        nc, ns, nq, nn = self._sizes

        def arr(p, cnt, dt):
            if cnt == 0:
                return np.zeros(0, dt)
            return np.ctypeslib.as_array(p, shape=(cnt,)).view(dt)

        kw = dict(
            ref_id=arr(v.ref_id, n, np.int32), pos=arr(v.pos, n, np.int32),
            mate_ref_id=arr(v.mate_ref_id, n, np.int32), mate_pos=arr(v.mate_pos, n, np.int32),
            flags=arr(v.flags, n, np.uint16), mapq=arr(v.mapq, n, np.uint8),
            n_cigar=arr(v.n_cigar, n, np.uint32), l_seq=arr(v.l_seq, n, np.uint32),
            cigar_off=arr(v.cigar_off, n, np.uint64), cigar=arr(v.cigar, nc, np.uint32),
            seq_off=arr(v.seq_off, n, np.uint64), seq=arr(v.seq, ns, np.uint8),
            qual_off=arr(v.qual_off, n, np.uint64), qual=arr(v.qual, nq, np.uint8))
        if nn:
            kw["name_off"] = arr(v.name_off, n + 1, np.uint64)
            kw["name"] = arr(v.name, nn, np.uint8)
        return RecordBatch(**kw)

    def close(self):
        if self._h:
            load().bps_free(self._h)
            self._h = None

    def __del__(self):
        self.close()


def generate(regions, seed=SEED, read_len=150, insert_mean=350.0, insert_sd=50.0, insert_min=150,
             insert_max=510, frac_softclip=0.04, frac_del=0.03, frac_ins=0.02, frac_hardclip=0.01,
             sub_rate=0.005, frac_low_mapq=0.10, frac_dup=0.01, frac_secondary=0.005,
             frac_mate_other_ref=0.002, with_names=False, threads=None):
    """regions: list of (ref_id, start, end, coverage).  Defaults = SURVEY.md section 8(d) C1/C2."""
    lib = load()
    regs = (bps_region * len(regions))(*[bps_region(*r) for r in regions])
    if threads is None:
        threads = min(16, os.cpu_count() or 1)
    p = bps_params(seed, read_len, insert_mean, insert_sd, insert_min, insert_max, frac_softclip,
                   frac_del, frac_ins, frac_hardclip, sub_rate, frac_low_mapq, frac_dup,
                   frac_secondary, frac_mate_other_ref, int(with_names), threads, len(regions), regs)
    h = C.c_void_p()
    rc = lib.bps_generate(C.byref(p), C.byref(h))
    if rc != 0:
        raise RuntimeError("bps_generate failed")
    return SynthBatch(h)
