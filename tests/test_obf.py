"""Fork pre-pass (outlier-block-flag map, TEncSlice::getOutlierWithDCT): the whole pre-pass against the OBF maps the
REFERENCE'S OWN getOutlierWithDCT produced (tests/golden/obf.npz, oracle/ref/make_golden_obf.py: TEncSlice.cpp:878-1173
compiled in place and run through oracle/ref/ref_driver.cpp:ref_obf) -- oracle on the CPU, the HIP kernels on the GPU --,
the threshold fit alone against the reference's TCMprocessOneSequence (tests/golden/tcm.npz), oracle self-consistency."""
import ctypes as C
import os

import numpy as np
import pytest

import hmo_py


def _dct_blocks(Y):
    """All 4x4 forward DCTs of a plane through the leaf-pinned oracle transform: [h/4, w/4, 16] int32."""
    lib = hmo_py.load()
    h, w = Y.shape
    out = np.zeros((h // 4, w // 4, 16), np.int32)
    blk = np.zeros(16, np.int16)
    coef = np.zeros(16, np.int32)
    for by in range(h // 4):
        for bx in range(w // 4):
            blk[:] = Y[by * 4:by * 4 + 4, bx * 4:bx * 4 + 4].astype(np.int16).ravel()
            lib.hmo_fwd_transform(C.c_void_p(blk.ctypes.data), 4, C.c_void_p(coef.ctypes.data), 2, 0)
            out[by, bx] = coef
    return out


def test_obf_counts_follow_the_thresholds(pkg):
    Y, _, _ = pkg.synth.mixed(128, 64, seed=4)
    obf, yc = hmo_py.obf_prepass(Y)
    assert obf.shape == (16, 32) and yc[0] == 0 and np.all(yc[1:] >= 0) and np.all(yc == np.floor(yc))
    coef = _dct_blocks(Y)
    want = np.zeros_like(obf)
    for x in range(1, 16):
        c = coef[:, :, x]
        want += ((c != 0) & (np.abs(c) >= yc[x] * 8)).astype(np.int16)
    assert np.array_equal(obf, want)


def test_flat_plane_has_no_outliers(pkg):
    Y = np.full((64, 64), 117, np.uint8)
    obf, yc = hmo_py.obf_prepass(Y)
    assert not obf.any() and not yc.any()


def test_threshold_of_a_laplacian_with_outliers():
    """A Laplacian bulk plus a uniform tail: the fitted boundary separates them (sanity of the restated fit)."""
    rng = np.random.default_rng(3)
    bulk = np.abs(np.round(rng.laplace(0, 6, 40000))).astype(int)
    tail = rng.integers(120, 400, 400)
    amp = np.concatenate([bulk, tail])
    hist = np.bincount(amp).astype(np.int32)
    err = C.c_int(0)
    yc = hmo_py.load().hmo_tcm_threshold(C.c_void_p(hist.ctypes.data), int(amp.max()), int(amp.size), C.cast(C.byref(err), C.c_void_p))
    assert err.value == 0 and bulk.max() * 0.5 < yc < 150


GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tcm.npz")


def test_tcm_fit_matches_the_reference(built, pkg):
    """Oracle restatement and the engine's host-side fit (libfcu.so, host arithmetic) == the reference's Yc on every case
    where the reference's result is defined (peak >= 3, or all-zero input)."""
    g = np.load(GOLD)
    lo, le = hmo_py.load(), pkg.load_lib()
    n_def = 0
    for i in range(len(g["len"])):
        if not g["defined"][i]:
            continue                                            # FindStartPoint reads uninitialised buckets there (peak 1, 2)
        n_def += 1
        peak, n = int(g["peak"][i]), int(g["len"][i])
        hist = np.ascontiguousarray(g["hist"][i][:max(peak, 0) + 1], np.int32)
        err = C.c_int(0)
        yo = lo.hmo_tcm_threshold(C.c_void_p(hist.ctypes.data), peak, n, C.cast(C.byref(err), C.c_void_p))
        assert err.value == 0 and yo == g["yc"][i], (i, peak, n, yo, g["yc"][i])
        hu = hist.astype(np.uint32)
        ye = le.fcu_tcm_threshold(hu.ctypes.data, hu.size, n)
        assert ye == g["yc"][i], (i, peak, n, ye, g["yc"][i])
    assert n_def >= 20


OBF_GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "obf.npz")


def _obf_cases(pkg):
    g = np.load(OBF_GOLD)
    for i, (gen, (w, h), seed) in enumerate(zip(g["gen"], g["size"], g["seed"])):
        yield str(gen), int(w), int(h), getattr(pkg.synth, str(gen))(int(w), int(h), seed=int(seed))[0], g["obf%d" % i]


def test_oracle_prepass_matches_the_reference(pkg):
    """hmo_obf.c == the reference's getOutlierWithDCT on every fixture picture (sizes on and off the 16-sample grid)."""
    n = 0
    for gen, w, h, Y, want in _obf_cases(pkg):
        obf, _ = hmo_py.obf_prepass(Y)
        assert np.array_equal(obf, want), (gen, w, h, int((obf != want).sum()))
        n += 1
    assert n >= 6


@pytest.mark.gpu
def test_gpu_prepass_matches_the_reference(pkg):
    """fcu_obf_prepass (obf_hist -> host fit -> obf_count) == the reference's own OBF maps."""
    for gen, w, h, Y, want in _obf_cases(pkg):
        eng = pkg.CuEngine(w, h, max_chains=1)
        obf, yc, ms = eng.obf_prepass(Y[None])
        assert np.array_equal(obf.cpu().numpy()[0], want), (gen, w, h)
        eng.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("gen,w,h", [("mixed", 256, 128), ("smooth", 416, 240), ("mixed", 136, 72), ("textured", 3840, 2160)])
def test_gpu_prepass_matches_oracle(pkg, gen, w, h):
    frames = [getattr(pkg.synth, gen)(w, h, seed=s)[0] for s in (7, 8)]
    eng = pkg.CuEngine(w, h, max_chains=1)
    obf, yc, ms = eng.obf_prepass(np.stack(frames))
    obf = obf.cpu().numpy()
    for i, Y in enumerate(frames):
        o_ref, yc_ref = hmo_py.obf_prepass(Y)
        assert np.array_equal(yc[i], yc_ref), (yc[i], yc_ref)
        assert np.array_equal(obf[i], o_ref)
    assert ms[0] > 0 and ms[1] > 0
    eng.destroy()
