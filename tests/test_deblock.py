"""In-loop deblocking (TComLoopFilter::loopFilterPic): the oracle against the golden vectors produced by the reference's
own TComLoopFilter (tests/golden/deblock_*.npz, generator oracle/ref/make_golden_deblock.py) -- PINNED.  The GPU side
(`fcu_deblock`) is compared with the oracle in tests/test_gpu_parity.py."""
import ctypes as C
import glob
import os

import numpy as np
import pytest

import hmo_py

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "deblock_*.npz")))


def ctus_from_golden(g):
    """hmo Ctu array (TComDataCU layout) filled with the fields deblocking reads."""
    n = g["depth"].shape[0]
    arr = (hmo_py.Ctu * n)()
    for a in range(n):
        for name in ("depth", "part_size", "pred_mode", "tr_idx", "qp"):
            np.ctypeslib.as_array(getattr(arr[a], name))[:] = g[name][a]
    return arr


def test_golden_set_is_complete():
    assert len(GOLD) == 6


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[8:-4] for p in GOLD])
def test_oracle_deblocking_matches_reference(built, path):
    g = np.load(path)
    w, h = int(g["width"]), int(g["height"])
    rec = [np.ascontiguousarray(g["rec_" + c]).copy() for c in "yuv"]
    arr = ctus_from_golden(g)
    hmo_py.deblock_pic(bytes(arr), w, h, rec, int(g["beta_offset_div2"]), int(g["tc_offset_div2"]))
    for k, c in enumerate("yuv"):
        want = g["out_" + c]
        assert np.array_equal(rec[k], want), f"plane {c}: {np.argwhere(rec[k] != want)[:4].tolist()}"
    assert any(not np.array_equal(g["rec_" + c], g["out_" + c]) for c in "yuv")      # the filter did something


def test_encoder_deblock_is_the_picture_function(built, pkg):
    """hmo_deblock on an encoder = hmo_deblock_pic on its CTU array; deblocking is idempotent in neither direction, so
    only equality of the two entry points and the untouched picture border columns are checked."""
    Y, U, V = pkg.synth.smooth(192, 128, seed=3)
    e = hmo_py.Encoder(Y, U, V, 32)
    e.compress_frame()
    before = [r.copy() for r in e.rec]
    n = e.n_ctu
    arr = (hmo_py.Ctu * n)()
    for a in range(n):
        C.memmove(C.addressof(arr[a]), C.addressof(e.ctu(a)), C.sizeof(hmo_py.Ctu))
    rec2 = [r.copy() for r in before]
    hmo_py.deblock_pic(bytes(arr), 192, 128, rec2)
    e.deblock()
    for p, q in zip(e.rec, rec2):
        assert np.array_equal(p, q)
    assert not np.array_equal(e.rec[0], before[0])
    # nothing further than 3 samples from an 8-sample grid line can change
    diff = np.argwhere(e.rec[0] != before[0])
    assert all(min(x % 8, 8 - x % 8) <= 3 or min(y % 8, 8 - y % 8) <= 3 for y, x in diff)


@pytest.mark.parametrize("path", GOLD, ids=[os.path.basename(p)[8:-4] for p in GOLD])
def test_kernel_source_on_cpu_matches_reference(built, path):
    """csrc/fcu_deblock.h compiled for the CPU (tests/emu/dbk_emu.cpp: grid as a loop) against the reference's output:
    the GPU-less check of the kernels' indexing and arithmetic."""
    lib = C.CDLL(os.path.join(os.path.dirname(__file__), "emu", "libdbk_emu.so"))
    lib.dbk_emu.argtypes = [C.c_void_p] * 4 + [C.c_int] * 4
    g = np.load(path)
    w, h = int(g["width"]), int(g["height"])
    rec = [np.ascontiguousarray(g["rec_" + c]).copy() for c in "yuv"]
    arr = ctus_from_golden(g)
    lib.dbk_emu(C.addressof(arr), rec[0].ctypes.data, rec[1].ctypes.data, rec[2].ctypes.data, w, h,
                int(g["beta_offset_div2"]), int(g["tc_offset_div2"]))
    for k, c in enumerate("yuv"):
        want = g["out_" + c]
        assert np.array_equal(rec[k], want), f"plane {c}: {np.argwhere(rec[k] != want)[:4].tolist()}"
