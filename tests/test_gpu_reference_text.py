"""GPU twin of test_reference_text.py: the count-only output files the reference's tests hold
go through the DEVICE renderer (bp_render_text, kinds basestrand / .ref.tsv / .alt.tsv).  The
rows of each file are put on the device as they are (a job over the file's positions with no
records computes zero rows; bp_rows_accumulate adds the file's values), rendered there, and the
text must equal the file's body byte for byte (pileup/snp/output.go:149-276, :437-453)."""
import numpy as np
import pytest

import oracle_lib as O
import ref_text as T
from bio_amd import snp

pytestmark = pytest.mark.gpu


def _render(rows, kind, cols):
    import torch
    names = rows["names"]
    ref_len = [len(s) + 1 for s in rows["seqs"]]
    bed = {}
    for r, p in zip(rows["ref_id"].tolist(), rows["pos"].tolist()):
        iv = bed.setdefault(r, [])
        if iv and iv[-1][1] == p:
            iv[-1][1] = p + 1
        else:
            iv.append([p, p + 1])
    out = b""
    opts = O.make_opts(ignore_strand=1 if kind else 0)
    with snp.PileupJob(opts, ref_len, names) as job:
        job.set_bed(bed)
        job.finish()
        assert job.num_rows == len(rows["pos"])
        ref, pos, vals = job.rows()
        assert (ref == rows["ref_id"]).all() and (pos == rows["pos"]).all() and vals.sum() == 0
        dev = torch.from_numpy(rows["vals"].astype(np.int64)).to(torch.int32).cuda().contiguous()
        torch.cuda.synchronize()
        job.rows_accumulate(0, len(pos), dev.data_ptr())
        assert (job.rows()[2] == rows["vals"]).all()
        for r in range(len(names)):
            idx = np.nonzero(rows["ref_id"] == r)[0]
            lo, hi = int(idx[0]), int(idx[-1]) + 1
            start = int(rows["pos"][lo])
            job.set_ref_seq8(r, start, rows["seqs"][r][start:int(rows["pos"][hi - 1]) + 1])
            out += job.render_text(lo, hi - lo, kind, cols)
    return out


def _body(name):
    text = T.read(name)
    return text[text.index(b"\n") + 1:]


@pytest.mark.parametrize("name", ["sample1.basestrand.tsv", "test_pileup_snp.tsv"])
def test_basestrand_files_through_the_device(name):
    assert _render(T.parse_basestrand(name), 0, 0) == _body(name)


def test_generic_tsv_pair_through_the_device():
    rows = T.parse_tsv_pair("generic.ref.tsv.expected", "generic.alt.tsv.no_n.expected", per_read=False)
    # -cols=-lowq: dpref, highq (pileup_snp_test.go:66)
    assert _render(rows, 1, 1 | 64) == _body("generic.ref.tsv.expected")
    assert _render(rows, 2, 1 | 64) == _body("generic.alt.tsv.no_n.expected")
