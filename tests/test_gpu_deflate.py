"""The device's BGZF writer (deflate_kernels.hip behind bp_deflate_bgzf / bp_render_fetch_bgzf):
every member is a well-formed BGZF block (SAM spec 4.1) and the run inflates, with zlib as the
checker (it verifies CRC-32 and ISIZE of every member), to exactly the input.  The bytes themselves
are this compressor's own: the reference's writer is biogo/hts' bgzf over compress/flate, absent
from the tree, and no two DEFLATE implementations agree on bytes."""
import gzip
import struct
import zlib

import numpy as np
import pytest

from bio_amd import bgzf

pytestmark = pytest.mark.gpu


def members(buf):
    """-> [(total, isize)] after checking every header field BGZF fixes."""
    out, p = [], 0
    buf = bytes(buf)
    while p < len(buf):
        assert buf[p:p + 4] == b"\x1f\x8b\x08\x04" and buf[p + 10:p + 12] == b"\x06\x00", p
        assert buf[p + 12:p + 16] == b"BC\x02\x00", p
        total = struct.unpack_from("<H", buf, p + 16)[0] + 1
        crc, isize = struct.unpack_from("<II", buf, p + total - 8)
        body = zlib.decompressobj(-15).decompress(buf[p + 18:p + total - 8])
        assert len(body) == isize <= 0xff00 and zlib.crc32(body) == crc, p
        out.append((total, isize))
        p += total
    assert p == len(buf)
    return out


def text_like(n_lines, seed):
    rng = np.random.default_rng(seed)
    pos = 1000000 + np.arange(n_lines)
    counts = rng.poisson(3.0, (n_lines, 8)) * (rng.random((n_lines, 8)) < 0.3)
    ref = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n_lines)]
    return "".join(f"chr1\t{p}\t{chr(r)}\t" + "\t".join(map(str, c)) + "\n" for p, r, c in zip(pos, ref, counts)).encode()


@pytest.fixture(scope="module")
def inf():
    with bgzf.Inflater() as f:
        yield f


@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 63, 64, 65, 2039, 2040, 2041, 4080, 8159, 8160, 8161, 0xff00 - 1, 0xff00, 0xff00 + 1, 3 * 0xff00 + 77])
def test_sizes_around_every_boundary(inf, n):
    # (windows of 64 positions, segments of 2,040 bytes - a wave each -, members of 0xff00)
    data = text_like(n // 20 + 2, n)[:n]
    out = inf.deflate(data)
    ms = members(out)
    assert sum(i for _, i in ms) == n and len(ms) == (n + 0xff00 - 1) // 0xff00
    assert gzip.decompress(bytes(out)) == data if n else len(out) == 0


def test_text_compresses_and_inflates(inf):
    data = text_like(300000, 1)
    out = inf.deflate(data)
    assert gzip.decompress(bytes(out)) == data
    ratio = len(data) / len(out)
    print(f"{len(data)} -> {len(out)} bytes: {ratio:.2f}x (zlib -6: {len(data) / len(zlib.compress(data, 6)):.2f}x, -1: {len(data) / len(zlib.compress(data, 1)):.2f}x)")
    assert ratio > 2.5


def test_random_bytes_fall_back_to_stored(inf):
    rng = np.random.default_rng(5)
    data = rng.integers(0, 256, 200000, dtype=np.uint8).tobytes()
    out = inf.deflate(data)
    assert gzip.decompress(bytes(out)) == data
    assert len(out) <= inf.lib.bp_deflate_bgzf_bound(len(data))
    assert all(t == i + 31 for t, i in members(out))


def test_long_runs_and_far_matches(inf):
    rng = np.random.default_rng(6)
    block = rng.integers(0, 256, 5000, dtype=np.uint8).tobytes()
    data = b"\x00" * 100000 + block + b"A" * 300 + block + b"\t0" * 40000 + block[:1000] * 30 + bytes(range(256)) * 50
    out = inf.deflate(data)
    assert gzip.decompress(bytes(out)) == data
    assert len(out) < len(data) / 4


def test_mixed_members_one_stored_one_not(inf):
    rng = np.random.default_rng(7)
    data = text_like(3000, 2)[:0xff00] + rng.integers(0, 256, 0xff00, dtype=np.uint8).tobytes() + text_like(3000, 3)
    out = inf.deflate(data)
    assert gzip.decompress(bytes(out)) == data
    ms = members(out)
    assert ms[1][0] == 0xff00 + 31 and ms[0][0] < 0xff00 // 2


def test_repeated_calls_reuse_the_buffers(inf):
    for seed, n in ((1, 50000), (2, 500), (3, 120000)):
        data = text_like(n, seed)
        assert gzip.decompress(bytes(inf.deflate(data))) == data


def test_skewed_symbol_counts_use_length_limited_codes(inf):
    """Fibonacci-like byte counts: Huffman's tree would be 30 deep, the codes are cut to 15 bits
    (deflate_api.hip); shuffled, so that the literals carry the stream."""
    rng = np.random.default_rng(8)
    counts, a, b = [], 1, 1
    for _ in range(31):
        counts.append(a)
        a, b = b, a + b
    data = np.repeat(np.arange(31, dtype=np.uint8) + 48, counts)
    rng.shuffle(data)
    data = data.tobytes()
    out = inf.deflate(data)
    assert gzip.decompress(bytes(out)) == data
    assert len(out) < len(data) / 2.5  # ~2.6 bits of entropy a byte
