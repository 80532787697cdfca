"""bp_compute_static, the entry point that allocates nothing (include/bio_pileup.h), against
bp_compute / bp_finish through the C ABI: same rows, bit for bit, on two calls in a row, the second
on a smaller input that leaves most of what the first call wrote into the shared static scratch
stale (record ranges of more windows, a longer work-item queue of deep windows).  Collected last
(the file name sorts behind the other tests)."""
import numpy as np
import pytest

import oracle_lib as O
from bio_amd import snp, synth

pytestmark = pytest.mark.gpu


def _inputs(big):
    """A 30x stretch plus a few deep targets (their windows go through the work-item queue)."""
    if big:
        regions = [(0, 1_000, 401_000, 30.0)] + [(0, 500_000 + 3_000 * k, 500_200 + 3_000 * k, 4000.0) for k in range(6)]
        L = 600_000
    else:
        regions = [(0, 2_000, 52_000, 30.0)] + [(0, 80_000 + 5_000 * k, 80_200 + 5_000 * k, 3000.0) for k in range(2)]
        L = 100_000
    bed = {0: [[r[1], r[2]] for r in regions]}
    sb = synth.generate(regions, with_names=True, threads=8, seed=synth.SEED + (1 if big else 2),
                        insert_mean=250.0, insert_sd=40.0, insert_min=100, insert_max=500)
    return L, bed, sb


def _doubled(job):
    """rows += rows on the device, so that a compute that wrote nothing would be seen"""
    import ctypes as C
    p = C.c_void_p()
    job._check(job.lib.bp_rows_device(job.h, C.byref(p)))
    job.rows_accumulate(0, job.num_rows, p)


@pytest.mark.parametrize("kw", [dict(), dict(stitch=1, min_base_qual=20)], ids=["default", "stitched"])
def test_static_compute_equals_compute_twice_in_a_row(kw):
    opts = O.make_opts(**kw)
    want, jobs, keep = [], [], []
    try:
        for big in (True, False):
            L, bed, sb = _inputs(big)
            rb = sb.to_record_batch()
            keep.append(sb)
            job = snp.PileupJob(opts, [L])
            jobs.append(job)
            job.set_bed(bed)
            with pytest.raises(snp.PileupError, match="call bp_finish first"):
                job.compute_static()
            job.submit(rb)
            job.finish()                      # the existing entry point
            ref = job.rows()
            oracle = O.pileup(opts, [L], bed, rb)
            assert (ref[1] == oracle[1]).all() and (ref[2] == oracle[2]).all()
            assert ref[2][:, 0].max() > 2000  # deep windows: the queue was used
            want.append(ref)
        assert jobs[0].num_rows > jobs[1].num_rows
        # two calls in a row on the shared scratch, the second with the smaller input
        for job, ref in zip(jobs, want):
            _doubled(job)
            assert (job.rows()[2] == 2 * ref[2]).all()
            # the entry point's defining property: the device's free memory is the same before
            # and after (hipMemGetInfo through torch; the kernels' scratch here is megabytes, so a
            # hipMalloc slipped into the path would show), while the allocating path, asked for a
            # buffer it does not have yet, does move it
            import torch
            torch.cuda.synchronize()
            free0 = torch.cuda.mem_get_info(0)[0]
            job.compute_static()
            assert torch.cuda.mem_get_info(0)[0] == free0
            got = job.rows()
            assert (got[0] == ref[0]).all() and (got[1] == ref[1]).all() and (got[2] == ref[2]).all()
            assert job.timing().pileup_ms > 0
        # and back to the large one, then the allocating path again on the same handles
        jobs[0].compute_static()
        assert (jobs[0].rows()[2] == want[0][2]).all()
        for job, ref in zip(jobs, want):
            job.compute()
            assert (job.rows()[2] == ref[2]).all()
    finally:
        for j in jobs:
            j.close()
        for sb in keep:
            sb.close()


def test_static_compute_refuses_what_its_scratch_cannot_hold():
    """more windows than the static array has room for: a clear error, nothing launched"""
    n_iv = (1 << 19) + 1
    starts = np.arange(n_iv, dtype=np.int64) * 2048
    ep = np.stack([starts, starts + 1], axis=1).astype(np.int32)
    L = int(starts[-1]) + 10
    with snp.PileupJob(O.make_opts(), [L]) as job:
        job.set_bed({0: ep})
        job.finish()
        assert job.num_rows == n_iv
        with pytest.raises(snp.PileupError, match=r"524289 windows .* static scratch holds 524288"):
            job.compute_static()
        job.compute()
        assert job.column_sums()[0] == 0
