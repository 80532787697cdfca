"""In-memory record batches (SoA numpy arrays) in the layout of bp_record_batch.

This is the analogue of the reference's bamprovider.NewFakeProvider
(encoding/bamprovider/fakeprovider.go:27-140): literal records served to the pileup path
without a BAM file.
"""
import ctypes as C

import numpy as np

from ._abi import bp_record_batch

# sam.Flags (github.com/grailbio/hts/sam): names used by the reference's tests.
FLAG = {
    "Paired": 0x1, "ProperPair": 0x2, "Unmapped": 0x4, "MateUnmapped": 0x8, "Reverse": 0x10,
    "MateReverse": 0x20, "Read1": 0x40, "Read2": 0x80, "Secondary": 0x100, "QCFail": 0x200,
    "Duplicate": 0x400, "Supplementary": 0x800,
}
CIGAR_OPS = "MIDNSHP=XB"
# BAM 4-bit base codes "=ACMGRSVTWYHKDBN" (SAM spec 4.2.3)
_SEQ_CODE = {c: i for i, c in enumerate("=ACMGRSVTWYHKDBN")}


def pack_seq(seq):
    """ASCII bases -> BAM packed 4-bit (first base in the high nibble)."""
    codes = [_SEQ_CODE.get(c.upper(), 15) for c in seq]
    if len(codes) & 1:
        codes.append(0)
    return bytes((codes[i] << 4) | codes[i + 1] for i in range(0, len(codes), 2))


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


class RecordBatch:
    """Owns the numpy arrays behind one bp_record_batch."""

    FIELDS = ("ref_id", "pos", "mate_ref_id", "mate_pos", "flags", "mapq", "n_cigar", "l_seq",
              "cigar_off", "cigar", "seq_off", "seq", "qual_off", "qual", "name_off", "name")

    def __init__(self, **arrays):
        self.ref_id = np.ascontiguousarray(arrays["ref_id"], dtype=np.int32)
        self.pos = np.ascontiguousarray(arrays["pos"], dtype=np.int32)
        self.mate_ref_id = np.ascontiguousarray(arrays["mate_ref_id"], dtype=np.int32)
        self.mate_pos = np.ascontiguousarray(arrays["mate_pos"], dtype=np.int32)
        self.flags = np.ascontiguousarray(arrays["flags"], dtype=np.uint16)
        self.mapq = np.ascontiguousarray(arrays["mapq"], dtype=np.uint8)
        self.n_cigar = np.ascontiguousarray(arrays["n_cigar"], dtype=np.uint32)
        self.l_seq = np.ascontiguousarray(arrays["l_seq"], dtype=np.uint32)
        self.cigar_off = np.ascontiguousarray(arrays["cigar_off"], dtype=np.uint64)
        self.cigar = np.ascontiguousarray(arrays["cigar"], dtype=np.uint32)
        self.seq_off = np.ascontiguousarray(arrays["seq_off"], dtype=np.uint64)
        self.seq = np.ascontiguousarray(arrays["seq"], dtype=np.uint8)
        self.qual_off = np.ascontiguousarray(arrays["qual_off"], dtype=np.uint64)
        self.qual = np.ascontiguousarray(arrays["qual"], dtype=np.uint8)
        name_off = arrays.get("name_off")
        name = arrays.get("name")
        self.name_off = None if name_off is None else np.ascontiguousarray(name_off, dtype=np.uint64)
        self.name = None if name is None else np.ascontiguousarray(name, dtype=np.uint8)
        self.n = int(self.pos.shape[0])

    def as_c(self):
        b = bp_record_batch()
        b.n = self.n
        b.ref_id = _ptr(self.ref_id, C.c_int32)
        b.pos = _ptr(self.pos, C.c_int32)
        b.mate_ref_id = _ptr(self.mate_ref_id, C.c_int32)
        b.mate_pos = _ptr(self.mate_pos, C.c_int32)
        b.flags = _ptr(self.flags, C.c_uint16)
        b.mapq = _ptr(self.mapq, C.c_uint8)
        b.n_cigar = _ptr(self.n_cigar, C.c_uint32)
        b.l_seq = _ptr(self.l_seq, C.c_uint32)
        b.cigar_off = _ptr(self.cigar_off, C.c_uint64)
        b.cigar = _ptr(self.cigar, C.c_uint32)
        b.seq_off = _ptr(self.seq_off, C.c_uint64)
        b.seq = _ptr(self.seq, C.c_uint8)
        b.qual_off = _ptr(self.qual_off, C.c_uint64)
        b.qual = _ptr(self.qual, C.c_uint8)
        if self.name is not None:
            b.name_off = _ptr(self.name_off, C.c_uint64)
            b.name = _ptr(self.name, C.c_uint8)
        return b

    def slice(self, lo, hi):
        """Records [lo, hi) as a new batch sharing the payload arrays (offsets are absolute)."""
        kw = {k: getattr(self, k) for k in ("cigar", "seq", "qual", "name")}
        for k in ("ref_id", "pos", "mate_ref_id", "mate_pos", "flags", "mapq", "n_cigar", "l_seq",
                  "cigar_off", "seq_off", "qual_off"):
            kw[k] = getattr(self, k)[lo:hi]
        kw["name_off"] = None if self.name_off is None else self.name_off[lo:hi + 1]
        return RecordBatch(**kw)

    def take_spans(self, spans):
        """Records of the index spans [(lo,hi),...] (ascending, disjoint) as one batch; payload
        arrays are shared, names are re-gathered so name_off stays n+1 contiguous entries."""
        spans = [(int(a), int(b)) for a, b in spans if b > a]
        kw = {k: getattr(self, k) for k in ("cigar", "seq", "qual")}
        for k in ("ref_id", "pos", "mate_ref_id", "mate_pos", "flags", "mapq", "n_cigar", "l_seq",
                  "cigar_off", "seq_off", "qual_off"):
            a = getattr(self, k)
            kw[k] = np.concatenate([a[lo:hi] for lo, hi in spans]) if spans else a[:0]
        kw["name_off"], kw["name"] = None, None
        if self.name_off is not None:
            offs, blobs, run = [], [], 0
            for lo, hi in spans:
                b0, b1 = int(self.name_off[lo]), int(self.name_off[hi])
                offs.append(self.name_off[lo:hi].astype(np.int64) - b0 + run)
                blobs.append(self.name[b0:b1])
                run += b1 - b0
            offs.append(np.array([run], np.int64))
            kw["name_off"] = np.concatenate(offs).astype(np.uint64)
            kw["name"] = np.concatenate(blobs) if blobs else np.zeros(0, np.uint8)
        return RecordBatch(**kw)

    def aligned_bases(self):
        """Sum of CIGAR-M lengths (the unit of BASELINE.json's metric, before filters)."""
        ops = self.cigar & 15
        lens = (self.cigar >> 4).astype(np.int64)
        return int(lens[ops == 0].sum())

    def algorithmic_input_bytes(self):
        """sum(32 + 4*n_cigar + ceil(l/2) + l)  (SURVEY.md section 8d)."""
        l = self.l_seq.astype(np.int64)
        return int((32 + 4 * self.n_cigar.astype(np.int64) + (l + 1) // 2 + l).sum())


def from_records(records):
    """Build a RecordBatch from a list of dicts:
    {name, ref_id, pos, mapq, flags (int or list of names), cigar [(op,len)], mate_ref_id,
     mate_pos, seq (ASCII) or seq_codes (list of nibbles), qual (list of ints)}."""
    n = len(records)
    ref_id = np.zeros(n, np.int32)
    pos = np.zeros(n, np.int32)
    mate_ref_id = np.zeros(n, np.int32)
    mate_pos = np.zeros(n, np.int32)
    flags = np.zeros(n, np.uint16)
    mapq = np.zeros(n, np.uint8)
    n_cigar = np.zeros(n, np.uint32)
    l_seq = np.zeros(n, np.uint32)
    cigar_off = np.zeros(n, np.uint64)
    seq_off = np.zeros(n, np.uint64)
    qual_off = np.zeros(n, np.uint64)
    name_off = np.zeros(n + 1, np.uint64)
    cigar, seq, qual, name = [], bytearray(), bytearray(), bytearray()
    for i, r in enumerate(records):
        ref_id[i] = r.get("ref_id", 0)
        pos[i] = r["pos"]
        mate_ref_id[i] = r.get("mate_ref_id", ref_id[i])
        mate_pos[i] = r.get("mate_pos", -1)
        f = r.get("flags", 0)
        if not isinstance(f, int):
            f = sum(FLAG[x] for x in f)
        flags[i] = f
        mapq[i] = r.get("mapq", 60)
        cg = r["cigar"]
        n_cigar[i] = len(cg)
        cigar_off[i] = len(cigar)
        for op, ln in cg:
            code = CIGAR_OPS.index(op) if isinstance(op, str) else int(op)
            cigar.append((int(ln) << 4) | code)
        q = r["qual"]
        l_seq[i] = len(q)
        seq_off[i] = len(seq)
        if "seq_codes" in r:
            codes = list(r["seq_codes"])
            if len(codes) & 1:
                codes.append(0)
            seq += bytes((codes[k] << 4) | codes[k + 1] for k in range(0, len(codes), 2))
        else:
            assert len(r["seq"]) == len(q)
            seq += pack_seq(r["seq"])
        qual_off[i] = len(qual)
        qual += bytes(q)
        name_off[i] = len(name)
        name += r.get("name", "r%d" % i).encode()
    name_off[n] = len(name)
    return RecordBatch(
        ref_id=ref_id, pos=pos, mate_ref_id=mate_ref_id, mate_pos=mate_pos, flags=flags, mapq=mapq,
        n_cigar=n_cigar, l_seq=l_seq, cigar_off=cigar_off,
        cigar=np.array(cigar if cigar else [0], np.uint32)[:len(cigar)] if cigar else np.zeros(0, np.uint32),
        seq_off=seq_off, seq=np.frombuffer(bytes(seq) or b"\0", np.uint8)[:len(seq)].copy(),
        qual_off=qual_off, qual=np.frombuffer(bytes(qual) or b"\0", np.uint8)[:len(qual)].copy(),
        name_off=name_off, name=np.frombuffer(bytes(name) or b"\0", np.uint8)[:len(name)].copy())
