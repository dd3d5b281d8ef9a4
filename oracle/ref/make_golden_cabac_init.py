#!/usr/bin/env python3
"""Generates tests/golden/cabac_init.npz by RUNNING THE REFERENCE'S OWN TEncSbac::resetEntropy (TEncSbac.cpp:106-156) for a P
slice with both values of the encoder's table choice (TComSlice::getEncCABACTableIdx() = P_SLICE / B_SLICE with
cabac_init_present_flag -- the choice TEncSbac::determineCabacInitIdx makes after the previous slice) and for an I slice, at
every QP 0..51: the context states of all contexts the oracle / engine model, in their order.
Run in the build container only:  python oracle/ref/make_golden_cabac_init.py"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def one(qp):
    import hmo_py
    import search_trace as st
    Y = np.full((64, 64), 128, np.uint8); U = np.full((32, 32), 128, np.uint8)
    r = st.RefSearch(64, 64, qp, (Y, U, U), search_range=8)
    out = {}
    s = np.zeros(512, np.uint8)
    r.L.ref_cabac_reset(); r.L.ref_cabac_states(s.ctypes.data_as(C.c_void_p)); out["I"] = r.from_hm(s[:r.n_hm])
    r.setup_p((Y, U, U), 10.0)
    for name, b in (("P", 0), ("B", 1)):
        r.L.ref_set_cabac_table(b)
        r.L.ref_cabac_reset(); r.L.ref_cabac_states(s.ctypes.data_as(C.c_void_p)); out[name] = r.from_hm(s[:r.n_hm])
    np.save("/tmp/_cabac_init_%d.npy" % qp, np.stack([out["I"], out["P"], out["B"]]))


def main():
    if len(sys.argv) > 1:
        return one(int(sys.argv[1]))
    rows = []
    for qp in range(52):
        subprocess.check_call([sys.executable, os.path.abspath(__file__), str(qp)])
        rows.append(np.load("/tmp/_cabac_init_%d.npy" % qp)); os.remove("/tmp/_cabac_init_%d.npy" % qp)
    a = np.stack(rows)                                           # [qp][I, P, B][context]
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "cabac_init.npz"), states=a, order=np.array(["I", "P", "B"]))
    print("contexts that differ between the P and the B table at QP 32:", int((a[32, 1] != a[32, 2]).sum()))


if __name__ == "__main__":
    main()
