"""The oracle's P-SLICE path against golden vectors produced by the REFERENCE'S OWN inter search (TEncSearch.cpp,
TComDataCU.cpp, TComPrediction.cpp, TComLoopFilter.cpp compiled in place; oracle/ref/make_golden_inter.py): every merge
candidate (with / without residual), inter 2Nx2N / Nx2N / 2NxN and intra candidate of every CU the encoder visits in the P
pictures of four short lowdelay_P clips (incl. three pictures of 416x240 and SearchRange 64), and every deblocked P picture.
Compared per candidate: distortion, bits, cost (f64, exact), skip / merge / motion of the first and last partition, CRC-32
of all motion and mode arrays, TU tree, coefficients, reconstruction, CABAC state.  CPU only."""
import glob
import importlib.util
import os
import sys

import numpy as np
import pytest

import hmo_py
import search_trace as st

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "inter_*.npz")))


def _gen():
    spec = importlib.util.spec_from_file_location("make_golden_inter", os.path.join(ROOT, "oracle", "ref", "make_golden_inter.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(f)[6:-4] for f in FIXTURES])
def test_every_p_picture_candidate_matches_the_reference(path, built):
    assert FIXTURES, "no inter fixtures committed"
    g = np.load(path)
    case = os.path.basename(path)[6:-4]
    m = _gen()
    assert list(g["ifields"]) == st.IFIELDS and list(g["fields"]) == st.FIELDS
    gen, w, h, qp, seed, n_pic, sr = m.CASES[case]
    assert (w, h, qp, seed, n_pic, sr) == tuple(int(g[k]) for k in ("width", "height", "qp", "seed", "pictures", "search_range"))
    dbk = {}

    def on_picture(poc, enc, ref, bad):
        if poc:
            planes = [a.copy() for a in enc.rec]
            hmo_py.deblock_pic(np.frombuffer(b"".join(bytes(enc.ctu(a)) for a in range(enc.n_ctu)), np.uint8), w, h, planes)
            dbk[poc] = [st.crc(p) for p in planes]

    recs = m.run_case(case, None, on_picture)                   # the oracle alone (no reference library involved)
    for poc in range(1, n_pic):
        want, got = g[f"inter_{poc}"], recs[poc][0]
        assert got.shape == want.shape, (poc, got.shape, want.shape)
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert bad.size == 0, "picture %d candidate %d:\n  reference %s\n  oracle    %s" % (poc, bad[0], st.ifmt(want[bad[0]]), st.ifmt(got[bad[0]]))
        wi, gi = g[f"intra_{poc}"], recs[poc][1]
        assert gi.shape == wi.shape and np.array_equal(gi, wi), (poc, "intra candidates inside the P picture")
        assert dbk[poc] == [int(v) for v in g[f"deblock_{poc}"][:3]], (poc, "deblocked P picture")


def test_ldp_slice_parameters():
    """QP offsets / lambdas of HM's lowdelay_P GOP table (TEncSlice::initEncSlice): depth 0 at POC % 4 == 0."""
    t, qp, lam = hmo_py.ldp_slice(0, 32)
    assert t == hmo_py.SLICE_I and qp == 32 and abs(lam - 0.57 * 0.85 * 2 ** (20 / 3)) < 1e-9
    assert [hmo_py.ldp_slice(p, 32)[1] for p in range(1, 9)] == [35, 34, 35, 33, 35, 34, 35, 33]
    assert abs(hmo_py.ldp_slice(4, 32)[2] - 0.578 * 2 ** (21 / 3)) < 1e-9            # key picture: no depth factor
    assert abs(hmo_py.ldp_slice(2, 32)[2] - 0.4624 * 2 ** (22 / 3) * (22 / 6)) < 1e-9
