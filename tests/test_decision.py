"""The fork's fast CU-size decision (Naive model on the N_OBF feature, the reference's default control): the
Verifying / Testing states of xCompressCU in the oracle and in the engine source (CPU wave emulator), plus the host
arithmetic of libfcu.so (fcu_decision_switch, fcu_frame_state), which needs no GPU.

Parity status: UNPINNED above the leaves -- the hooks live in TEncCu.cpp / tools_YS.cpp, which cannot be built here
(OpenCV, libsvm); the oracle restates them line by line and these tests hold the engine to the oracle and both to the
properties the rule implies."""
import numpy as np
import pytest

import hmo_py

TRAINING, VERIFYING, TESTING = hmo_py.TRAINING, hmo_py.VERIFYING, hmo_py.TESTING
ARRAYS = ["depth", "part_size", "pred_mode", "tr_idx", "tskip", "cbf", "intra_dir", "coeff_y", "coeff_cb", "coeff_cr"]


def _same(a, b):
    for k, v in a.items():
        if isinstance(v, np.ndarray):
            if not np.array_equal(v, b[k]):
                return False
        elif v != b[k]:
            return False
    return True


def _frame(enc):
    enc.compress_frame()
    return [enc.ctu_arrays(a) for a in range(enc.n_ctu)]


def check_rule(ctus, obf, w, h, sw_skip, sw_term, depth_exception=0):
    """What the Testing state implies for the published quadtree: walking every CTU from depth 0, a CU inside the
    picture that was reached at depth d with its terminate switch on and no outlier block is a leaf (2Nx2N at depth
    3); with its skip switch on and an outlier block it is divided (NxN at depth 3)."""
    zs = hmo_py.load().hmo_zscan_to_raster
    zs.restype = np.ctypeslib.ndpointer(np.uint8, shape=(256,))
    z2r = np.asarray(zs()).astype(int)
    r2z = np.argsort(z2r)
    w_ctu = (w + 63) // 64
    n_checked = 0
    for a, c in enumerate(ctus):
        cx, cy = (a % w_ctu) * 64, (a // w_ctu) * 64

        def walk(x, y, d):
            nonlocal n_checked
            s = 64 >> d
            if x >= w or y >= h:
                return
            part = r2z[((y - cy) // 4) * 16 + (x - cx) // 4]
            inside = x + s <= w and y + s <= h
            split = c["depth"][part] > d
            if inside:
                n = int((obf[y // 4:(y + s) // 4, x // 4:(x + s) // 4] > 0).sum())
                exc = d == 3 and n > 0 and depth_exception
                divided = split or (d == 3 and c["part_size"][part] == 3)
                if sw_term[d] and n == 0 and not exc:
                    assert not divided, (a, x, y, d, "terminate label but divided")
                    n_checked += 1
                if sw_skip[d] and n > 0 and not exc:
                    assert divided, (a, x, y, d, "skip-2Nx2N label but not divided")
                    n_checked += 1
            if split and d < 3:
                for i in range(4):
                    walk(x + (i & 1) * (s // 2), y + (i >> 1) * (s // 2), d + 1)

        walk(cx, cy, 0)
    return n_checked


@pytest.fixture(scope="module")
def small(pkg, built):
    w, h, qp = 192, 128, 32
    Y, U, V = pkg.synth.smooth(w, h, seed=7)
    obf, _ = hmo_py.obf_prepass(Y)
    base = _frame(hmo_py.Encoder(Y, U, V, qp))
    return dict(w=w, h=h, qp=qp, yuv=(Y, U, V), obf=obf, base=base)


def test_verifying_is_exhaustive_and_counts_every_cu(small):
    Y, U, V = small["yuv"]
    v = hmo_py.Encoder(Y, U, V, small["qp"])
    v.set_decision(VERIFYING, small["obf"])
    got = _frame(v)
    assert all(_same(a, b) for a, b in zip(small["base"], got))          # no pruning in the Verifying state
    ver = v.verify_counts()
    n_ctu = len(got)
    # every CU inside the picture is visited once per depth and lands in exactly one of TP / FP / TN / FN
    assert ver[:, :4].sum(axis=1).tolist() == [n_ctu * 4 ** d for d in range(4)]
    # the label is the OBF predicate: "skip" (TP + FP) counts the CUs that hold an outlier block
    obf = small["obf"]
    for d in range(4):
        q = 16 >> d
        blocks = (obf > 0).reshape(obf.shape[0] // q, q, obf.shape[1] // q, q).any(axis=(1, 3))
        assert ver[d, 0] + ver[d, 1] == blocks.sum()
    # TP + FN at depth d = CUs whose RDO outcome was "divide": from the published tree only where the walk got there
    assert ver[0, 0] + ver[0, 3] == sum(int(c["depth"][0] > 0) for c in got)
    assert (ver[:, 4:] >= 0).all()


def test_testing_with_switches_off_is_training(small):
    Y, U, V = small["yuv"]
    t = hmo_py.Encoder(Y, U, V, small["qp"])
    t.set_decision(TESTING, small["obf"])
    assert all(_same(a, b) for a, b in zip(small["base"], _frame(t)))


@pytest.mark.parametrize("sw_skip,sw_term,dex", [((1, 1, 1, 1), (1, 1, 1, 1), 0), ((0, 0, 0, 0), (1, 1, 1, 1), 0),
                                                  ((1, 1, 1, 1), (0, 0, 0, 0), 1), ((1, 0, 1, 0), (0, 1, 0, 1), 0)])
def test_testing_obeys_the_rule_and_never_beats_rdo(small, sw_skip, sw_term, dex):
    Y, U, V = small["yuv"]
    t = hmo_py.Encoder(Y, U, V, small["qp"])
    t.set_decision(TESTING, small["obf"], sw_skip, sw_term, depth_exception=dex)
    got = _frame(t)
    assert check_rule(got, small["obf"], small["w"], small["h"], sw_skip, sw_term, dex) > 0
    # pruning only removes candidates: with the same entering state the first CTU cannot get cheaper
    assert got[0]["total_cost"] >= small["base"][0]["total_cost"]
    assert all(np.isfinite(c["total_cost"]) and c["total_cost"] < 1e300 for c in got)


CASES = [("smooth", 192, 128, 32), ("mixed", 136, 72, 27), ("textured", 128, 64, 37)]


@pytest.mark.parametrize("gen,w,h,qp", CASES)
def test_emulated_engine_matches_oracle_in_all_states(built, pkg, gen, w, h, qp):
    import emu_py
    Y, U, V = getattr(pkg.synth, gen)(w, h, seed=5)
    obf, _ = hmo_py.obf_prepass(Y)

    def both(state, *sw, dex=0):
        o, e = hmo_py.Encoder(Y, U, V, qp), emu_py.EmuEncoder(Y, U, V, qp)
        o.set_decision(state, obf, *sw, depth_exception=dex)
        e.set_decision(state, obf, *sw, depth_exception=dex)
        for a in range(o.n_ctu):
            o.compress_ctu(a)
            e.compress_ctu(a)
            assert _same(o.ctu_arrays(a), e.ctu_arrays(a)), (state, sw, a)
            ca, fa = o.cabac()
            cb, fb = e.cabac()
            assert np.array_equal(ca, cb) and fa == fb
        for p, q in zip(o.rec, e.rec):
            assert np.array_equal(p, q)
        return o, e

    o, e = both(VERIFYING)
    ver = o.verify_counts()
    assert np.array_equal(ver, e.verify_counts())                         # counts and the f64 loss sums
    measured = hmo_py.decision_switch(ver)
    for sw in (measured, ((1, 1, 1, 1), (1, 1, 1, 1)), ((0, 1, 0, 1), (1, 0, 1, 0))):
        both(TESTING, *sw)
    both(TESTING, (1, 1, 1, 1), (1, 1, 1, 1), dex=1)


def test_decision_switch_and_frame_state_host_arithmetic(built, pkg):
    """fcu_decision_switch / fcu_frame_state of libfcu.so (no device involved) against the oracle's restatement of
    SetDecisionSwitch and against hand-computed cases."""
    eng = pkg.engine
    rng = np.random.default_rng(3)
    for _ in range(300):
        v = rng.integers(0, 30, (4, 6)).astype(np.float64)
        v[rng.integers(0, 4)] = 0
        th = rng.choice([0.0, 0.5, 0.8, 0.95], 4)
        a, b = eng.decision_switch(v, th, th[::-1]), hmo_py.decision_switch(v, th, th[::-1])
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    v = np.zeros((4, 6))
    v[0, :4] = [9, 1, 4, 1]        # skip precision 0.9 > 0.8 on; terminate precision 0.8, not > 0.8: off
    v[1, :4] = [8, 2, 5, 0]        # 0.8 off; 1.0 on
    v[2, :4] = [0, 0, 0, 0]        # no samples: both off (precision defined as 0)
    v[3, :4] = [1, 0, 0, 3]        # 1.0 on; 0.0 off
    sk, te = eng.decision_switch(v)
    assert sk.tolist() == [1, 0, 0, 1] and te.tolist() == [0, 1, 0, 0]
    # g_iP 60, g_iT 2, g_iV 1 (tools_YS.cpp:41-43): two Training pictures, one Verifying, 57 Testing, repeat
    states = [eng.frame_state(p) for p in range(123)]
    assert states[:4] == [0, 0, 1, 2] and states[59:64] == [2, 0, 0, 1, 2] and states.count(1) == 3
    assert eng.frame_state(5, period=4, n_training=1, n_verifying=2) == 1
