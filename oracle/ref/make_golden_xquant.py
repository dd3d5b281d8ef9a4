#!/usr/bin/env python3
"""Generates tests/golden/xquant.npz: the REFERENCE's own quantiser without RDOQ -- TComTrQuant::transformNxN -> xQuant's
plain branch (TComTrQuant.cpp:1160-1240) with signBitHidingHDQ (:991-1124) -> invTransformNxN, compiled in place into
oracle/_ref/libhmleaf.so -- on random residual blocks: 4x4 .. 32x32, luma and chroma, transform skip, QP 22 / 32 / 37, I and P
slices (rounding offset 171 vs 85), the three RDOQ / RDOQTS combinations that reach the plain branch.  Data only.

Run in the build container only:  python oracle/ref/make_golden_xquant.py
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def main():
    L = C.CDLL(os.path.join(HERE, "..", "_ref", "libhmleaf.so"))
    rng = np.random.default_rng(77)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    path_arr = lambda path: (C.c_int * max(1, len(path)))(*path)
    cfgs = [(1, [], 0), (2, [], 0), (3, [], 0), (3, [0], 0), (2, [1], 0), (1, [1, 2], 0), (1, [], 1), (2, [], 2), (3, [], 1), (2, [3], 1)]
    meta, resis, coefs, routs = [], [], [], []
    for qp in (22, 32, 37):
        assert L.ref_setup(128, 128, qp) > 0
        for is_p in (0, 1):
            L.ref_set_slice_type(is_p)
            for rdoq, rdoq_ts in ((0, 0), (0, 1), (1, 0)):
                L.ref_set_rdoq(rdoq, rdoq_ts)
                for it in range(40):
                    d, path, comp = cfgs[rng.integers(0, len(cfgs))]
                    size = (64 >> (d + len(path))) >> (1 if comp else 0)
                    l = int(np.log2(size))
                    ts = int(l == 2 and rng.random() < 0.5)
                    if (rdoq_ts if ts else rdoq):
                        ts = 1 - ts if l == 2 else ts           # keep to blocks that take the plain branch
                        if (rdoq_ts if ts else rdoq):
                            continue
                    ldir, cdir = int(rng.choice([0, 1, 10, 26, 34, 2, 18])), int(rng.choice([0, 1, 10, 26, 36]))
                    part_size = 3 if (d == 3 and len(path) == 1 and comp == 0) else 0
                    amp = [3, 10, 30, 90, 200][rng.integers(0, 5)]
                    base = rng.normal(0, amp, (size, size))
                    if rng.random() < 0.5:
                        k = np.outer(np.hanning(size + 2)[1:-1], np.hanning(size + 2)[1:-1]); base = base * 0.3 + amp * k * rng.normal()
                    resi = np.clip(np.rint(base), -255, 255).astype(np.int16)
                    coef = np.zeros(size * size, np.int32); rout = np.zeros(size * size, np.int16)
                    zidx = [0, 64, 16, 4][d] if d else 0
                    a = L.ref_tq(0, zidx, d, len(path), path_arr(path), comp, part_size, ldir, cdir, ts, p(resi), p(coef), p(rout))
                    assert a >= 0
                    meta.append([qp, is_p, rdoq, rdoq_ts, comp, l, ldir, cdir, len(path), ts, a])
                    resis.append(resi.ravel()); coefs.append(coef); routs.append(rout)
    out = os.path.join(ROOT, "tests", "golden", "xquant.npz")
    np.savez_compressed(out, meta=np.array(meta, np.int32), resi=np.concatenate(resis), coef=np.concatenate(coefs), rout=np.concatenate(routs))
    m = np.array(meta)
    print("wrote", out, len(meta), "cases;", int((m[:, 10] > 0).sum()), "with coded coefficients;", "levels changed by sign hiding are inside the fixture")


if __name__ == "__main__":
    main()
