#!/usr/bin/env python3
"""Generates tests/golden/tcm.npz by RUNNING THE REFERENCE'S OWN threshold fit: TCMprocessOneSequence with
FindStartPoint / ComputeLikelyhood / ComputeLambdaGivenYc (Lib/TLibEncoder/TEncSlice.cpp:193-392), compiled in place into
oracle/_ref/libhmleaf.so (oracle/ref/build_ref.sh keeps exactly these four free functions of TEncSlice.cpp).

Cases: Laplacian amplitude sequences of several scales and lengths, Laplacian bulk + uniform outlier tail (the situation
the fork's model is made for), all-zero input, one non-zero sample, a
single occupied bucket, and sequences whose peak is 1 or 2.  For peak < 3 the reference's FindStartPoint reads
buck[peak+1..3].count, which TCMprocessOneSequence never initialises (TEncSlice.cpp:233-244 vs :360-365): such cases are
recorded with `defined` = 0 -- the fixture then holds what one run happened to return -- and the tests only require
agreement where the reference's result is defined.

Inputs are stored as amplitude histograms (the fit only looks at |C[k]|).  Data only.
Run in the build container only:  python oracle/ref/make_golden_tcm.py
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def main():
    L = C.CDLL(os.path.join(HERE, "..", "_ref", "libhmleaf.so"))
    tcm = getattr(L, "_Z21TCMprocessOneSequencePiiS_PdS0_S0_")          # TCMprocessOneSequence(int*, int, int*, double*, double*, double*)
    tcm.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    tcm.restype = None
    rng = np.random.default_rng(2024)
    seqs = []
    for scale, n in ((1.5, 6240), (3.0, 6240), (6.0, 32400), (12.0, 32400), (25.0, 129600), (60.0, 129600), (4.0, 518400)):
        seqs.append(np.round(rng.laplace(0, scale, n)).astype(np.int32))
    for scale, n, lo, hi, k in ((6.0, 40000, 120, 400, 400), (3.0, 32400, 40, 90, 900), (10.0, 129600, 300, 2000, 2000),
                                (2.0, 6240, 30, 60, 30), (8.0, 518400, 200, 4000, 5000), (1.0, 32400, 10, 20, 3000)):
        a = np.round(rng.laplace(0, scale, n)).astype(np.int32)
        a[rng.choice(n, k, replace=False)] = rng.integers(lo, hi, k) * rng.choice([-1, 1], k)
        seqs.append(a)
    # (uniformly distributed input is left out: the reference's fixed-point iteration ComputeLambdaGivenYc then creeps towards
    #  lambda ~ 1e8 in steps of ~0.2 -- about 1e9 iterations, the same in the reference, the oracle and the engine's host fit)
    for scale, n, lo, hi, k in ((5.0, 20000, 60, 200, 150), (1.2, 6240, 8, 30, 200)):
        a = np.round(rng.laplace(0, scale, n)).astype(np.int32)
        a[rng.choice(n, k, replace=False)] = rng.integers(lo, hi, k) * rng.choice([-1, 1], k)
        seqs.append(a)
    seqs.append(np.zeros(6240, np.int32))                                 # all zero
    z = np.zeros(6240, np.int32); z[17] = 9; seqs.append(z)               # one sample
    seqs.append(np.full(6240, 7, np.int32))                               # a single occupied bucket
    z = np.zeros(32400, np.int32); z[::3] = 1; seqs.append(z)             # peak 1
    z = np.zeros(32400, np.int32); z[::3] = 1; z[::7] = -2; seqs.append(z)   # peak 2
    z = np.zeros(32400, np.int32); z[::2] = 1; z[::5] = 2; z[::9] = -3; seqs.append(z)   # peak 3 (first defined one)
    for scale, n in ((0.6, 32400), (0.9, 129600)):                        # mostly zero, short tails
        seqs.append(np.round(rng.laplace(0, scale, n)).astype(np.int32))
    hists, lens, peaks, probs, lambdas, ycs, defined = [], [], [], [], [], [], []
    for a in seqs:
        a = np.ascontiguousarray(a)
        pk, pr, la, yc = C.c_int(0), C.c_double(0), C.c_double(0), C.c_double(0)
        tcm(a.ctypes.data, a.size, C.byref(pk), C.byref(pr), C.byref(la), C.byref(yc))
        h = np.bincount(np.abs(a)).astype(np.int32)
        hists.append(h); lens.append(a.size); peaks.append(pk.value); probs.append(pr.value); lambdas.append(la.value); ycs.append(yc.value)
        defined.append(int(pk.value == 0 or pk.value >= 3))
        print("len %7d peak %5d -> Yc %6.1f prob %.6f lambda %.6f%s" % (a.size, pk.value, yc.value, pr.value, la.value, "" if defined[-1] else "   (reads uninitialised buckets)"))
    m = max(len(h) for h in hists)
    H = np.zeros((len(hists), m), np.int32)
    for i, h in enumerate(hists):
        H[i, :len(h)] = h
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "tcm.npz"), hist=H, len=np.array(lens), peak=np.array(peaks),
                        prob=np.array(probs), lam=np.array(lambdas), yc=np.array(ycs), defined=np.array(defined))
    print(len(hists), "cases")


if __name__ == "__main__":
    main()
