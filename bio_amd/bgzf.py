"""BGZF on the device: a ctypes face of bp_inflate_* (include/bio_pileup.h) for tests, tools and
bench.py.  The block-header walk here is Python plumbing for those callers (the product's own
walk is BamReader's, bio_amd/csrc/host/bam.cpp); every byte of output comes from the device."""
import ctypes as C
import struct

import numpy as np

from . import _abi
from .snp import PileupError

BLOCK_DTYPE = np.dtype([("in_off", "<u4"), ("in_len", "<u4"), ("out_off", "<u4"), ("out_len", "<u4"), ("crc32", "<u4")])


def scan_blocks(data, start=0, max_blocks=None):
    """BGZF block headers and trailers of `data` (bytes-like) from `start` on (SAM spec 4.1) ->
    (blocks[BLOCK_DTYPE] with ISIZE > 0, offset behind the last whole block, total ISIZE)."""
    mv = memoryview(data)
    n = len(mv)
    p, out, rows = start, 0, []
    while p + 18 <= n and (max_blocks is None or len(rows) < max_blocks):
        if mv[p] != 31 or mv[p + 1] != 139 or mv[p + 2] != 8 or not (mv[p + 3] & 4):
            raise ValueError(f"bgzf: invalid block header at {p}")
        xlen = struct.unpack_from("<H", mv, p + 10)[0]
        if p + 12 + xlen > n:
            break
        bsize, q = None, p + 12
        while q + 4 <= p + 12 + xlen:
            si1, si2, slen = mv[q], mv[q + 1], struct.unpack_from("<H", mv, q + 2)[0]
            if si1 == 66 and si2 == 67 and slen == 2:
                bsize = struct.unpack_from("<H", mv, q + 4)[0]
            q += 4 + slen
        if bsize is None:
            raise ValueError(f"bgzf: missing BC subfield at {p}")
        total = bsize + 1
        if p + total > n:
            break
        crc, isize = struct.unpack_from("<II", mv, p + total - 8)
        if isize:
            rows.append((p + 12 + xlen, total - 12 - xlen - 8, out, isize, crc))
            out += isize
        p += total
    return np.array(rows, BLOCK_DTYPE), p, out


class Inflater:
    """One bp_inflater."""

    def __init__(self, device=0):
        self.lib = _abi.load()
        self.h = C.c_void_p()
        rc = self.lib.bp_inflate_create(device, C.byref(self.h))
        if rc != 0:
            raise PileupError(rc, self.lib.bp_inflate_last_error(None).decode())

    def inflate(self, comp, blocks, out_bytes=None, check_crc=True, to_host=True):
        """comp: uint8 array of compressed bytes; blocks: BLOCK_DTYPE array.  -> uint8 array of
        out_bytes (None when to_host is False: the output stays on the device)."""
        comp = np.ascontiguousarray(comp, np.uint8)
        blocks = np.ascontiguousarray(blocks, BLOCK_DTYPE)
        if out_bytes is None:
            out_bytes = int((blocks["out_off"].astype(np.int64) + blocks["out_len"]).max()) if len(blocks) else 0
        out = np.empty(out_bytes, np.uint8) if to_host else None
        bad = C.c_int64(-1)
        rc = self.lib.bp_inflate_blocks(self.h, comp.ctypes.data, comp.size, blocks.ctypes.data, len(blocks),
                                        out.ctypes.data if to_host and out_bytes else None, out_bytes, int(check_crc),
                                        C.byref(bad))
        self.bad_block = bad.value
        if rc != 0:
            raise PileupError(rc, self.lib.bp_inflate_last_error(self.h).decode())
        return out

    COLS = (("off", np.uint32), ("ref_id", np.int32), ("pos", np.int32), ("mate_ref_id", np.int32), ("mate_pos", np.int32),
            ("flags", np.uint16), ("mapq", np.uint8), ("l_name", np.uint8), ("n_cigar", np.uint32), ("l_seq", np.uint32),
            ("bytes", np.uint32), ("span", np.int32), ("facts", np.uint8))

    def scan(self, first, end, n_ref, cap=None, stream=None, names=False):
        """bp_bam_scan over the last inflate's device output (or `stream`, a device address): ->
        (rc, cols dict of numpy arrays cut to n, exit, stopped, the bp_bam_columns struct).  rc 1:
        the host has to walk this stream.  names=True: the read names come too (cols["name_off"],
        n + 1 offsets into cols["names"])."""
        if cap is None:
            cap = (end - first) // 36 + 1
        arrs = {k: np.zeros(cap, t) for k, t in self.COLS}
        c = _abi.bp_bam_columns()
        c.cap = cap
        for k, _ in self.COLS:
            setattr(c, k, arrs[k].ctypes.data)
        if names:
            arrs["name_off"] = np.zeros(cap + 1, np.uint32)
            arrs["names"] = np.zeros(max(end - first, 1), np.uint8)
            c.name_off = arrs["name_off"].ctypes.data
            c.names = arrs["names"].ctypes.data
            c.names_cap = len(arrs["names"])
        n, ex, st = C.c_int64(), C.c_uint64(), C.c_int32()
        rc = self.lib.bp_bam_scan(self.h, stream if stream is not None else self.device_output, first, end, n_ref,
                                  C.byref(c), C.byref(n), C.byref(ex), C.byref(st))
        if rc < 0:
            raise PileupError(rc, self.lib.bp_inflate_last_error(self.h).decode())
        self._keep = arrs
        out = {k: v[:n.value] for k, v in arrs.items() if k not in ("name_off", "names")}
        if names:
            out["name_off"] = arrs["name_off"][:n.value + 1]
            out["names"] = arrs["names"][:int(arrs["name_off"][n.value])]
        return rc, out, ex.value, st.value, c

    def deflate(self, data):
        """bp_deflate_bgzf: bytes-like -> uint8 array of complete BGZF members (no EOF member)."""
        data = np.ascontiguousarray(np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else data, np.uint8)
        cap = int(self.lib.bp_deflate_bgzf_bound(data.size))
        out = np.empty(max(cap, 1), np.uint8)
        n = C.c_size_t()
        rc = self.lib.bp_deflate_bgzf(self.h, data.ctypes.data if data.size else None, data.size, out.ctypes.data, cap,
                                      C.byref(n))
        if rc != 0:
            raise PileupError(rc, self.lib.bp_inflate_last_error(self.h).decode())
        return out[:n.value]

    def timing(self):
        a, b = C.c_float(), C.c_float()
        self.lib.bp_inflate_timing(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    @property
    def device_output(self):
        return self.lib.bp_inflate_device_output(self.h)

    def close(self):
        if self.h:
            self.lib.bp_inflate_destroy(self.h)
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
