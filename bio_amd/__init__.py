"""bio_amd: MI355X (gfx950) implementation of the grailbio/bio pileup/snp hot path.

The product is libbiopileup.so (HIP kernels + C++ host code behind the C ABI declared in
include/bio_pileup.h).  This Python package is plumbing: a ctypes mirror of that ABI and a
host-side mirror of the reference's snp.Pileup interface for tests and bench.py.
"""
from . import _abi  # noqa: F401

__all__ = ["_abi", "records", "snp"]
