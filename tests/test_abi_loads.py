"""CPU: the C-ABI library loads and exports every symbol include/bio_pileup.h declares.
No compute calls (there is no GPU in the CPU test tier)."""
import ctypes as C
import os
import re

from bio_amd import _abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "bio_pileup.h")).read()
    declared = set(re.findall(r"\b(bp_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_abi.EXPORTS), declared ^ set(_abi.EXPORTS)
    lib = _abi.load()
    for name in declared:
        assert hasattr(lib, name), name


def test_host_header_symbols_exported():
    hdr = open(os.path.join(ROOT, "include", "bio_pileup_host.h")).read()
    declared = set(re.findall(r"\b(bio_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_abi.HOST_EXPORTS), declared ^ set(_abi.HOST_EXPORTS)
    lib = _abi.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert C.sizeof(_abi.bio_pileup_opts) == 96


def test_struct_sizes_match_header():
    # bp_opts: 16 int32; bp_record_batch: int64 + 16 pointers; bp_timing: 3 float + pad + 2 int64
    assert C.sizeof(_abi.bp_opts) == 64
    assert C.sizeof(_abi.bp_record_batch) == 8 + 16 * 8
    assert C.sizeof(_abi.bp_timing) == 56


def test_no_cpu_fallback_without_device():
    """On a box without a GPU the library must fail loudly, not compute on the host."""
    import torch
    if torch.cuda.is_available():
        return
    lib = _abi.load()
    assert lib.bp_init(0) == _abi.BP_EDEVICE
    assert b"no CPU fallback" in lib.bp_last_error(None)
