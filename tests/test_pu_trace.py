"""BASELINE configs[1] -- intra-luma RDO (TEncSearch::estIntraPredLumaQT over the 35 modes at every depth): the per-PU
artefact SURVEY.md 8(d) names for it (RMD survivors with their SATD costs, candidate list after the MPM additions, winning
luma mode, luma distortion, RD cost) as an optional side output of the ordinary decision, for all 341 PUs of the five PU
layers of a CTU.  Engine source on the CPU emulator == oracle here; the HIP engine == oracle on a whole 1920x1080 picture
at QP 32 in the gpu-marked test.  (The winning mode and luma distortion of every PU are pinned by the reference's own
estIntraPredLumaQT through tests/test_golden_search.py; the SATD list rests on the leaf-pinned SATD and mode-bit costs.)"""
import numpy as np
import pytest

import emu_py
import hmo_py


def _check_structure(tr, w, h):
    """every PU inside the picture searched exactly once per layer; the records are self-consistent"""
    wc = (w + 63) // 64
    for a in range(tr.shape[0]):
        x0, y0 = (a % wc) * 64, (a // wc) * 64
        if x0 + 64 <= w and y0 + 64 <= h:
            assert int(tr[a]["valid"].sum()) == hmo_py.PUS_PER_CTU, a
    v = tr[tr["valid"] == 1]
    assert len(v) and (v["n_rd"] >= v["n_rmd"]).all() and (v["n_rd"] <= 11).all() and (v["best_mode"] < 35).all()
    for r in v[:: max(1, len(v) // 500)]:
        assert r["best_mode"] in r["rd_mode"][:r["n_rd"]]
        c = r["rmd_cost"][:r["n_rmd"]]
        assert (np.diff(c) >= 0).all()                          # CandCostList is kept sorted (xUpdateCandList)


@pytest.mark.parametrize("gen,w,h,qp", [("mixed", 128, 64, 32), ("textured", 136, 72, 27)])
def test_emulated_engine_pu_trace_equals_oracle(built, pkg, gen, w, h, qp):
    f = getattr(pkg.synth, gen)(w, h, 3)
    o, e = hmo_py.Encoder(*f, qp), emu_py.EmuEncoder(*f, qp)
    to, te = o.enable_pu_trace(), e.enable_pu_trace()
    o.compress_frame()
    e.compress_frame()
    assert to.tobytes() == te.tobytes()
    _check_structure(to, w, h)
    assert pkg.engine.PU_TRACE_DTYPE == hmo_py.PU_TRACE_DTYPE and pkg.engine.pu_index(3, 1, 37) == 85 + 37 and pkg.engine.pu_index(2, 0, 48) == 5 + 3


@pytest.mark.gpu
def test_1080p_luma_rdo_artefact_on_the_gpu(pkg):
    """configs[1]: 1920x1080 all-intra QP 32, every PU of every CTU (510 CTUs x up to 341 PUs), one chain per CTU row"""
    w, h, qp = 1920, 1080, 32
    f = pkg.synth.textured(w, h, 7)
    eng = pkg.CuEngine(w, h, max_chains=17)
    n_sl, rec, out = eng.init_slice_chains(0, f, qp, 30)
    trace = eng.enable_pu_trace(0)
    for k in range(1, n_sl):
        eng.enable_pu_trace(k, trace)
    eng.compress_chains(0, n_sl, 30)
    eng.sync()
    got = eng.pu_trace_array(trace)
    o = hmo_py.Encoder(*f, qp, slice_ctus=30)
    want = o.enable_pu_trace()
    o.compress_frame()
    assert got.tobytes() == want.tobytes()
    _check_structure(got, w, h)
    eng.destroy()
