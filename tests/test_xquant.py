"""The quantiser without RDOQ (TComTrQuant::xQuant's plain branch + signBitHidingHDQ, reached with
RDOQ 0 / RDOQTS 0): the oracle against tests/golden/xquant.npz (the reference's own transformNxN / invTransformNxN on
558 random blocks, oracle/ref/make_golden_xquant.py), and the engine source on the CPU emulator against the oracle on
whole pictures decided with RDOQ off."""
import ctypes as C
import os

import numpy as np
import pytest

import emu_py
import hmo_py

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p = lambda a: a.ctypes.data_as(C.c_void_p)


def test_oracle_plain_quantiser_matches_the_reference(built, pkg):
    g = np.load(os.path.join(ROOT, "tests", "golden", "xquant.npz"))
    lib = hmo_py.load()
    lib.hmo_test_tq.argtypes = [C.c_void_p] + [C.c_int] * 6 + [C.c_void_p] * 3
    lib.hmo_test_begin.argtypes = [C.c_void_p, C.c_int]
    Y, U, V = pkg.synth.mixed(128, 128, 1)
    encs, off, n_sbh = {}, 0, 0
    for qp, is_p, rdoq, rdoq_ts, comp, l, ldir, cdir, trd, ts, a in g["meta"]:
        key = (int(qp), int(is_p), int(rdoq), int(rdoq_ts))
        if key not in encs:
            encs[key] = hmo_py.Encoder(Y, U, V, key[0], slice_type=key[1], rdoq=key[2], rdoq_ts=key[3])
            lib.hmo_test_begin(C.c_void_p(encs[key].h), 0)
        n2 = 1 << (2 * int(l))
        resi = np.ascontiguousarray(g["resi"][off:off + n2])
        coef, rout = np.zeros(n2, np.int32), np.zeros(n2, np.int16)
        got = lib.hmo_test_tq(C.c_void_p(encs[key].h), int(comp), int(l), int(ldir), int(cdir), int(trd), int(ts), p(resi), p(coef), p(rout))
        assert got == a, (off, "uiAbsSum")
        assert np.array_equal(coef, g["coef"][off:off + n2]), (key, comp, l, ts, "levels")
        assert np.array_equal(rout, g["rout"][off:off + n2]), (key, comp, l, ts, "residual")
        n_sbh += int(np.abs(coef).sum() != a)
        off += n2
    assert off == len(g["resi"]) and n_sbh > 20                 # sign hiding changed a level in many of the cases


@pytest.mark.parametrize("gen,w,h,qp,rdoq,rdoq_ts", [("mixed", 128, 64, 32, 0, 0), ("textured", 72, 72, 27, 0, 1), ("mixed", 64, 64, 37, 1, 0)])
def test_emulated_engine_without_rdoq_equals_oracle(built, pkg, gen, w, h, qp, rdoq, rdoq_ts):
    f = getattr(pkg.synth, gen)(w, h, 5)
    o, e = hmo_py.Encoder(*f, qp, rdoq=rdoq, rdoq_ts=rdoq_ts), emu_py.EmuEncoder(*f, qp, rdoq=rdoq, rdoq_ts=rdoq_ts)
    for a in range(o.n_ctu):
        o.compress_ctu(a)
        e.compress_ctu(a)
        A, B = o.ctu_arrays(a), e.ctu_arrays(a)
        for k, v in A.items():
            assert (np.array_equal(v, B[k]) if isinstance(v, np.ndarray) else v == B[k]), (a, k)
        (ca, fa), (cb, fb) = o.cabac(), e.cabac()
        assert fa == fb and np.array_equal(ca, cb), (a, "coder state")
    for x, y in zip(o.rec, e.rec):
        assert np.array_equal(x, y)
    if not rdoq:                                                  # and the switch does something
        ref = hmo_py.Encoder(*f, qp)
        ref.compress_frame()
        assert any(not np.array_equal(x, y) for x, y in zip(o.rec, ref.rec))


@pytest.mark.gpu
@pytest.mark.parametrize("gen,w,h,qp,rdoq,rdoq_ts,p_pic", [("mixed", 192, 128, 32, 0, 0, 0), ("textured", 136, 72, 27, 0, 1, 0), ("mixed", 128, 128, 30, 0, 0, 1)])
def test_gpu_without_rdoq_equals_oracle(pkg, gen, w, h, qp, rdoq, rdoq_ts, p_pic):
    """HIP engine through the C ABI with RDOQ off (I picture, and a P picture on its deblocked predecessor)"""
    import search_trace as st
    eng = pkg.CuEngine(w, h, max_chains=1)
    prev = pad = None
    for poc in range(2 if p_pic else 1):
        f = st.moving_frame(pkg.synth, gen, w, h, 3, poc)
        fp = pkg.engine.ldp_slice(qp, poc)
        fp.rdoq, fp.rdoq_ts, fp.search_range, fp.fast_search = rdoq, rdoq_ts, 16, 1
        _, q, lam = hmo_py.ldp_slice(poc, qp)
        eng.init_chain(0, f, fp.qp, params=fp, ref=pad)
        o = hmo_py.Encoder(*f, q, lambda_override=lam, rdoq=rdoq, rdoq_ts=rdoq_ts) if poc == 0 else \
            hmo_py.Encoder(*f, q, ref=prev, lambda_override=lam, search_range=16, fast_search=1, rdoq=rdoq, rdoq_ts=rdoq_ts)
        for a in range(eng.n_ctu):
            got = eng.compress_ctu(0, a)
            o.compress_ctu(a)
            for k, v in o.ctu_arrays(a).items():
                assert (np.array_equal(v, got[k]) if isinstance(v, np.ndarray) else v == got[k]), (poc, a, k)
        for x, y in zip(eng.rec_planes(0), o.rec):
            assert np.array_equal(x, y), poc
        eng.deblock(0)
        eng.sync()
        o.deblock()
        prev = [x.copy() for x in o.rec]
        pad = eng.pad_reference(eng._keep[0][1])
    eng.destroy()
