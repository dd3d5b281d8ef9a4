"""CU level of the oracle against golden vectors produced by the REFERENCE'S OWN TEncCu functions -- xCheckRDCostMerge2Nx2N
(whole function, FastDecisionForMerge early-outs included), xCheckRDCostInter, xCheckRDCostIntra, xCheckBestMode,
deriveTestModeAMP -- compiled in place from the reference's TEncCu.cpp without the body of xCompressCU
(oracle/ref/build_ref.sh, oracle/ref/make_golden_cu.py).  For every CU inside the picture, at every depth, of two I pictures
and the P pictures of four lowdelay_P clips (full / TZ search, TMVP, AMP): the CU that survives all candidates -- prediction
mode, partition size, skip / merge, distortion, bits, cost (f64, exact), CRC-32 of motion, modes + TU tree, coefficients,
reconstruction and of the coder handed to the next CU ([depth][CI_NEXT_BEST]).  CPU only."""
import glob
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "cu_*.npz")))


def _gen():
    spec = importlib.util.spec_from_file_location("make_golden_cu", os.path.join(ROOT, "oracle", "ref", "make_golden_cu.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(f)[3:-4] for f in FIXTURES])
def test_surviving_cu_matches_the_reference(path, built):
    assert FIXTURES, "no CU-level fixtures committed"
    g = np.load(path)
    case = os.path.basename(path)[3:-4]
    m = _gen()
    assert list(g["fields"]) == m.FIELDS and list(g["case"]) == list(map(str, m.CASES[case]))
    recs, _ = m.run_case(case, False)                            # the oracle alone (no reference library involved)
    n = 0
    for poc, got in enumerate(recs):
        want = g[f"cu_{poc}"]
        assert got.shape == want.shape, (poc, got.shape, want.shape)
        bad = np.nonzero((got != want).any(axis=1))[0]
        assert bad.size == 0, "picture %d CU %d:\n  reference %s\n  oracle    %s" % (poc, bad[0], dict(zip(m.FIELDS, want[bad[0]].tolist())), dict(zip(m.FIELDS, got[bad[0]].tolist())))
        n += len(want)
    assert n >= 170


def test_fixtures_cover_the_cu_level_cases():
    """the fixtures hold what they are there for: intra and inter winners, skipped CUs, asymmetric partitions, intra CUs in P pictures"""
    assert len(FIXTURES) >= 6
    allrec = np.concatenate([np.load(f)[k] for f in FIXTURES for k in np.load(f).files if k.startswith("cu_") and len(np.load(f)[k])])
    pred, part, skip, merge = allrec[:, 4], allrec[:, 5], allrec[:, 6], allrec[:, 7]
    assert (pred == 1).sum() > 300 and (pred == 0).sum() > 1500 and skip.sum() > 1000
    assert ((pred == 0) & (part >= 4)).sum() > 0, "no asymmetric partition survived anywhere"
    assert ((pred == 0) & (part == 1)).sum() > 0 and ((pred == 0) & (part == 2)).sum() > 0
    assert ((pred == 0) & (merge == 1) & (skip == 0)).sum() > 0, "no merge CU with a residual"
    assert ((pred == 1) & (part == 3)).sum() > 0, "no intra NxN winner"
