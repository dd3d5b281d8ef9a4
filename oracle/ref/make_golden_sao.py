#!/usr/bin/env python3
"""Generates tests/golden/sao.npz: the REFERENCE's own TEncSampleAdaptiveOffset / TComSampleAdaptiveOffset (built in place
into oracle/_ref/libhmleaf.so, driver entry ref_sao) run on deblocked pictures.  Inputs are reproducible without the
reference: synthetic frames (package synth), decided and deblocked by the oracle.  Stored per case: the signalled
parameters of every CTU, CRC32s of the statistics and of the three output planes, the slice-level switches and the
disabled-rate the picture leaves behind (m_saoDisabledRate).

Run here (needs /root/reference):  python oracle/ref/make_golden_sao.py
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import hmo_py                                               # noqa: E402
import search_trace as st                                   # noqa: E402
import __graft_entry__ as g                                 # noqa: E402

# name: (generator, width, height, qp, seed, slice_ctus, slice_type, layer, disabledRate of layer-1)
CASES = {
    "survey416_qp32": ("survey_frame", 416, 240, 32, 1234, 0, 0, 0, (0, 0, 0)),
    "textured_qp32": ("textured", 192, 128, 32, 3, 0, 0, 0, (0, 0, 0)),
    "mixed_qp27": ("mixed", 136, 72, 27, 3, 0, 0, 0, (0, 0, 0)),
    "textured_qp22_slices": ("textured", 256, 192, 22, 3, 3, 0, 0, (0, 0, 0)),
    "smooth_qp37": ("smooth", 200, 136, 37, 3, 0, 0, 0, (0, 0, 0)),
    "mixed_qp27_slices": ("mixed", 320, 200, 27, 3, 2, 0, 0, (0, 0, 0)),
    "mixed_qp30_P_layer2": ("mixed", 192, 136, 30, 5, 0, 1, 2, (0.2, 0.4, 0.6)),       # chroma of Cr switched off by the rate of layer 1
    "textured_qp35_P_layer1_lumaoff": ("textured", 128, 128, 35, 5, 0, 1, 1, (0.8, 0.1, 0.1)),
}


def canon(params):
    """the fields of a SAOOffset that are signalled: mode; merge direction; type, band position and offsets of a NEW mode"""
    out = np.zeros_like(params)
    for a in range(params.shape[0]):
        for c in range(3):
            p = params[a, c]
            out[a, c, 0] = p[0]
            if p[0] == 2:
                out[a, c, 1] = p[1]
            elif p[0] == 1:
                out[a, c] = p
                if p[1] != 4:
                    out[a, c, 2] = 0
    return out


def decided_picture(pkg, case):
    gen, w, h, qp, seed, slice_ctus, slice_type, layer, dis = case
    f = getattr(pkg.synth, gen)(w, h, seed)
    enc = hmo_py.Encoder(*f, qp, slice_ctus=slice_ctus)
    enc.compress_frame()
    enc.deblock()
    return f, [p.copy() for p in enc.rec], enc.n_ctu


def ref_sao(case, f, rec, n):
    gen, w, h, qp, seed, slice_ctus, slice_type, layer, dis = case
    R = st.RefSearch(w, h, qp, f)
    L = R.L
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    for c in range(3):
        L.ref_set_rec(c, vp(rec[c]))
    lam = 0.57 * 2.0 ** ((qp - 12) / 3.0)
    rp, rs = np.zeros((n, 3, 35), np.int32), np.zeros((n, 3, 5, 2, 32), np.int64)
    misc, rate = np.zeros(3, np.int32), np.zeros(3, np.float64)
    lams, d = np.array(hmo_py.slice_lambdas(qp, lam), np.float64), np.array(dis, np.float64)
    L.ref_sao.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5
    L.ref_sao(slice_type, qp, vp(lams), slice_ctus, layer, vp(d), vp(rp), vp(rs), vp(misc), vp(rate))
    out = [np.zeros_like(p) for p in rec]
    for c in range(3):
        L.ref_get_rec(c, vp(out[c]))
    return rp, rs, misc, rate, out


def main():
    pkg = g.load_package()
    data = {}
    for name, case in CASES.items():
        f, rec, n = decided_picture(pkg, case)
        rp, rs, misc, rate, out = ref_sao(case, f, rec, n)
        data[name + "/params"] = canon(rp)
        data[name + "/stats_crc"] = np.array([st.crc(rs[:, :, :, 0]), st.crc(rs[:, :, :, 1])], np.int64)       # diff, count
        data[name + "/planes_crc"] = np.array([st.crc(p) for p in out], np.int64)
        data[name + "/in_crc"] = np.array([st.crc(p) for p in rec], np.int64)
        data[name + "/enabled"] = misc
        data[name + "/rate"] = rate
        print(name, "modes", np.bincount(rp[:, :, 0].ravel(), minlength=3), "enabled", misc, "rate", rate,
              "changed", [int((a != b).sum()) for a, b in zip(out, rec)])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "sao.npz"), **data)


if __name__ == "__main__":
    main()
