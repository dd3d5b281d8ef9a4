"""Whole-CTU syntax against the REFERENCE'S OWN entropy coder (tests/golden/syntax_*.npz, generator
oracle/ref/make_golden_syntax.py: TEncEntropy / TEncSbac / TEncBinCABACCounter compiled in place, driven CU by CU in
xEncodeCU's call order over pictures the oracle decided).  After every CTU the Q15 bit counter and all 160 context states
of the coder that carries the contexts from CTU to CTU (m_pppcRDSbacCoder[0][CI_CURR_BEST]) must equal the reference's:
this pins split flags, part size, prediction-mode coding with MPMs from the real neighbourhood, the transform tree with
its cbf / subdivision flags, coefficient coding in final order and the terminating bits -- for the oracle, for the
engine source on the CPU emulator and (gpu-marked) for the HIP engine through the C ABI."""
import glob
import os

import numpy as np
import pytest

import hmo_py
from test_golden_leaf import HM2O

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "syntax_*.npz")))
IDS = [os.path.basename(p)[7:-4] for p in GOLD]


def _frame(pkg, g):
    return getattr(pkg.synth, str(g["generator"]))(int(g["width"]), int(g["height"]), seed=int(g["seed"]))


def _check_ctu(g, a, ctx, frac, arrays, who):
    # same decisions as when the fixture was made (the fixture's inputs) ...
    for name in ("depth", "part_size", "tr_idx", "intra_dir", "cbf", "tskip"):
        assert np.array_equal(arrays[name], g[name][a]), (who, a, name)
    # ... and the reference's coder state after coding them
    assert frac == int(g["frac"][a]), (who, a, "Q15 bit counter", frac, int(g["frac"][a]))
    st = g["states"][a]
    for o, hm in HM2O:
        assert ctx[o] == st[hm], (who, a, "context", o)


def test_golden_set_is_complete():
    assert len(GOLD) == 4


@pytest.mark.parametrize("path", GOLD, ids=IDS)
def test_oracle_ctu_syntax_matches_reference_coder(built, pkg, path):
    g = np.load(path)
    o = hmo_py.Encoder(*_frame(pkg, g), int(g["qp"]))
    for a in range(o.n_ctu):
        o.compress_ctu(a)
        ctx, frac = o.cabac()
        _check_ctu(g, a, ctx, frac, o.ctu_arrays(a), "oracle")
        assert o.replay_bits(a) == int(g["bits"][a])


@pytest.mark.parametrize("path", GOLD, ids=IDS)
def test_emulated_engine_ctu_syntax_matches_reference_coder(built, pkg, path):
    import emu_py
    g = np.load(path)
    e = emu_py.EmuEncoder(*_frame(pkg, g), int(g["qp"]))
    for a in range(e.n_ctu):
        e.compress_ctu(a)
        ctx, frac = e.cabac()
        _check_ctu(g, a, ctx, frac, e.ctu_arrays(a), "engine source (emulator)")


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLD, ids=IDS)
def test_gpu_ctu_syntax_matches_reference_coder(pkg, path):
    g = np.load(path)
    w, h = int(g["width"]), int(g["height"])
    eng = pkg.CuEngine(w, h, max_chains=1)
    eng.init_chain(0, _frame(pkg, g), qp=int(g["qp"]))
    for a in range(eng.n_ctu):
        arrays = eng.compress_ctu(0, a)
        ctx, frac = eng.ctx_state(0)
        _check_ctu(g, a, ctx, frac, arrays, "HIP engine")
    eng.destroy()
