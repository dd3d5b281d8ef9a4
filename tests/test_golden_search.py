"""The oracle's SEARCH LOOPS against golden vectors produced by the REFERENCE'S OWN TEncSearch.cpp (compiled in place,
oracle/ref/build_ref.sh): estIntraPredLumaQT + estIntraPredChromaQT with everything below them, on EVERY CU candidate
the encoder visits on four pictures (incl. BASELINE configs[0]'s 416x240 frame): distortion (luma, total), bits, bins,
cost (f64, exact), prediction modes, CRC-32 of TU tree / cbf / transform-skip, of all coefficients, of the reconstruction
and of the CABAC state after the CU (tests/golden/search_*.npz, oracle/ref/make_golden_search.py).
CPU only; needs neither /root/reference nor a GPU."""
import glob
import os

import numpy as np
import pytest

import hmo_py
import search_trace as st

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIXTURES = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "search_*.npz")))


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(f)[7:-4] for f in FIXTURES])
def test_every_cu_candidate_matches_the_reference_search(path, built, pkg):
    assert FIXTURES, "no search fixtures committed"
    g = np.load(path)
    assert list(g["fields"]) == st.FIELDS
    w, h, qp = int(g["width"]), int(g["height"]), int(g["qp"])
    Y, U, V = getattr(pkg.synth, str(g["generator"]))(w, h, seed=int(g["seed"]))
    enc = hmo_py.Encoder(Y, U, V, qp)
    want, got = g["rec"], []

    def on_event(ev, depth, arg):
        if ev == hmo_py.EV_INTRA_END:
            got.append(st.record_from_oracle(enc, depth, arg))

    enc.set_trace(on_event)
    enc.compress_frame()
    got = np.stack(got)
    assert got.shape == want.shape, (got.shape, want.shape)          # same candidates in the same order
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, "first differing candidate %d:\n  reference %s\n  oracle    %s" % (bad[0], st.fmt(want[bad[0]]), st.fmt(got[bad[0]]))
    # the luma-only artefact of BASELINE configs[1] (per-PU best mode + luma distortion) is part of every record
    assert (want[:, st.FIELDS.index("dist_luma")] <= want[:, st.FIELDS.index("dist")]).all()


def test_reference_and_oracle_agree_live(built, pkg):
    """Where oracle/_ref was built from /root/reference: a fresh picture (not a committed fixture) through both."""
    lib = os.path.join(ROOT, "oracle", "_ref", "libhmleaf.so")
    if not os.path.exists(lib):
        pytest.skip("oracle/_ref not built (no /root/reference here)")
    import ctypes as C
    if not hasattr(C.CDLL(lib), "ref_intra_cu"):
        pytest.skip("oracle/_ref/libhmleaf.so predates the search driver")
    w, h, qp = 128, 64, 29
    Y, U, V = pkg.synth.mixed(w, h, seed=99)
    enc = hmo_py.Encoder(Y, U, V, qp)
    ref = st.RefSearch(w, h, qp, (Y, U, V))
    n, bad = [0], []

    def on_event(ev, depth, arg):
        if ev == hmo_py.EV_INTRA_BEGIN:
            ref.load_state(enc, depth)
        elif ev == hmo_py.EV_INTRA_END:
            cu = enc.test_cu(depth)
            r = st.record_from_ref(ref.intra_cu(enc.cur_ctu(), cu.zidx, depth, arg), ref, depth, enc.cur_ctu(), cu.zidx, arg)
            m = st.record_from_oracle(enc, depth, arg)
            n[0] += 1
            if not np.array_equal(r, m):
                bad.append("reference %s\noracle    %s" % (st.fmt(r), st.fmt(m)))

    enc.set_trace(on_event)
    enc.compress_frame()
    assert n[0] > 100 and not bad, "\n".join(bad[:3])
