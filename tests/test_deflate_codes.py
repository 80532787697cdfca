"""Host half of the device's BGZF writer (bio_amd/csrc/deflate_api.hip, bp_deflate_codes): Huffman
code lengths limited to 15 bits, canonical codes, the dynamic block header.  No GPU: a token stream
is encoded here in Python with the codes the library built and zlib inflates it.  RFC 1951 is the
specification and zlib the checker; the reference's bgzf.Writer (output.go:70-71, :391) lives in
biogo/hts, which the tree does not hold."""
import ctypes as C
import zlib

import numpy as np
import pytest

from bio_amd import _abi

LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEN_EXTRA = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
DIST_BASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073,
             4097, 6145, 8193, 12289, 16385, 24577]
DIST_EXTRA = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]


def len_symbol(length):
    k = max(i for i in range(29) if LEN_BASE[i] <= length and (i < 28 or length == 258))
    if length == 258:
        k = 28
    return 257 + k, LEN_EXTRA[k], length - LEN_BASE[k]


def dist_symbol(dist):
    k = max(i for i in range(30) if DIST_BASE[i] <= dist)
    return k, DIST_EXTRA[k], dist - DIST_BASE[k]


class Bits:
    def __init__(self):
        self.v, self.n = 0, 0

    def put(self, val, nb):
        self.v |= (val & ((1 << nb) - 1)) << self.n
        self.n += nb

    def bytes(self):
        return self.v.to_bytes((self.n + 7) // 8, "little")


def codes_for(hist, n_members=1):
    lib = _abi.load()
    h = np.ascontiguousarray(hist, np.uint32)
    out = np.zeros(288 + 32 + 1 + 128, np.uint32)
    rc = lib.bp_deflate_codes(h.ctypes.data_as(C.c_void_p), n_members, out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return out[:288], out[288:320], int(out[320]), out[321:]


def encode(tokens):
    """tokens: ints (literals) or (length, distance).  Returns (raw deflate bytes, code lengths)."""
    hist = np.zeros(320, np.uint32)
    for t in tokens:
        if isinstance(t, tuple):
            hist[len_symbol(t[0])[0]] += 1
            hist[288 + dist_symbol(t[1])[0]] += 1
        else:
            hist[t] += 1
    ll, dc, hbits, hdr = codes_for(hist)
    b = Bits()
    for i in range((hbits + 31) // 32):
        b.put(int(hdr[i]), min(32, hbits - 32 * i))
    for t in tokens:
        if isinstance(t, tuple):
            s, nb, ext = len_symbol(t[0])
            assert ll[s] >> 16, "length symbol without a code"
            b.put(int(ll[s]) & 0xffff, int(ll[s]) >> 16)
            b.put(ext, nb)
            s, nb, ext = dist_symbol(t[1])
            assert dc[s] >> 16, "distance symbol without a code"
            b.put(int(dc[s]) & 0xffff, int(dc[s]) >> 16)
            b.put(ext, nb)
        else:
            assert ll[t] >> 16, "literal without a code"
            b.put(int(ll[t]) & 0xffff, int(ll[t]) >> 16)
    b.put(int(ll[256]) & 0xffff, int(ll[256]) >> 16)
    return b.bytes(), (ll >> 16, dc >> 16)


def expand(tokens):
    out = bytearray()
    for t in tokens:
        if isinstance(t, tuple):
            for _ in range(t[0]):
                out.append(out[-t[1]])
        else:
            out.append(t)
    return bytes(out)


def check(tokens):
    raw, lens = encode(tokens)
    assert zlib.decompressobj(-15).decompress(raw) == expand(tokens)
    return lens


def test_symbol_tables_cover_every_length_and_distance():
    assert [len_symbol(x)[0] for x in (3, 10, 11, 12, 13, 257, 258)] == [257, 264, 265, 265, 266, 284, 285]
    assert [dist_symbol(x)[0] for x in (1, 4, 5, 7, 9, 32768)] == [0, 3, 4, 5, 6, 29]


def test_text_like_stream_inflates():
    rng = np.random.default_rng(3)
    alphabet = np.frombuffer(b"0123456789\t\nACGTchr", np.uint8)
    tokens = []
    for _ in range(4000):
        if len(tokens) > 40 and rng.random() < 0.4:
            tokens.append((int(rng.integers(4, 40)), int(rng.integers(1, 40))))
        else:
            tokens.append(int(alphabet[rng.integers(0, len(alphabet))]))
    check(tokens)


def test_single_literal_and_no_matches():
    # one literal symbol + end of block: the distance code still has to be complete
    lens = check([65] * 10)
    assert (lens[1] > 0).sum() == 2
    check([7])


def test_every_symbol_once():
    tokens = list(range(256))
    for length in LEN_BASE:
        tokens.append((length, 1))
    for d in DIST_BASE:
        if d <= len(expand(tokens)):
            tokens.append((3, d))
    check(tokens)


def test_skewed_counts_are_cut_to_15_bits():
    # Fibonacci-like counts make Huffman's tree a chain far deeper than 15
    hist = np.zeros(320, np.uint32)
    a, b = 1, 1
    for s in range(40):
        hist[s] = a
        a, b = b, a + b
    hist[288] = 1
    ll, dc, hbits, hdr = codes_for(hist)
    lens = (ll >> 16).astype(int)
    assert lens[:40].min() >= 1 and lens.max() == 15 and lens[256] > 0
    # complete: Kraft sum exactly 1
    assert sum(2 ** (15 - x) for x in lens if x) == 2 ** 15
    # and zlib accepts the header and the codes: every symbol with a count once
    b = Bits()
    for i in range((hbits + 31) // 32):
        b.put(int(hdr[i]), min(32, hbits - 32 * i))
    for s in range(40):
        b.put(int(ll[s]) & 0xffff, int(ll[s]) >> 16)
    b.put(int(ll[256]) & 0xffff, int(ll[256]) >> 16)
    assert zlib.decompressobj(-15).decompress(b.bytes()) == bytes(range(40))


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_random_histograms_give_complete_codes(seed):
    rng = np.random.default_rng(seed)
    hist = np.zeros(320, np.uint32)
    k = int(rng.integers(1, 286))
    idx = rng.choice(286, k, replace=False)
    hist[idx] = (rng.pareto(0.6, k) * 10 + 1).astype(np.uint32)
    kd = int(rng.integers(0, 30))
    hist[288 + rng.choice(30, kd, replace=False)] = (rng.pareto(0.8, kd) * 5 + 1).astype(np.uint32)
    ll, dc, hbits, hdr = codes_for(hist, n_members=3)
    for lens in ((ll >> 16).astype(int), (dc >> 16).astype(int)):
        assert lens.max() <= 15
        assert sum(2 ** (15 - x) for x in lens if x) == 2 ** 15
    assert all((ll[s] >> 16) > 0 for s in np.nonzero(hist[:286])[0])
    assert hbits <= 128 * 32
    # the header alone, followed by end of block, is a valid empty block
    b = Bits()
    for i in range((hbits + 31) // 32):
        b.put(int(hdr[i]), min(32, hbits - 32 * i))
    b.put(int(ll[256]) & 0xffff, int(ll[256]) >> 16)
    assert zlib.decompressobj(-15).decompress(b.bytes()) == b""
