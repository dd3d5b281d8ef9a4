#!/usr/bin/env python3
"""Generates tests/golden/leaf_*.npz by RUNNING THE REFERENCE'S OWN leaf code
(oracle/_ref/libhmleaf.so, built in place from /root/reference by build_ref.sh).

The fixtures are data only -- seeded random inputs and the outputs the reference produced for them:
tables, forward/inverse transforms, SATD/SSE, RD cost, reference samples + the 35 intra predictors
with real z-scan availability, MPM lists / split contexts, transform+RDOQ+sign-hiding+reconstruction
residuals, and CABAC coefficient bit counts with evolving context states.

Run in the build container only (needs /root/reference):  python oracle/ref/make_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(ROOT, "tests", "golden")
W, H = 192, 128          # 3x2 CTUs


def load():
    L = C.CDLL(os.path.join(HERE, "..", "_ref", "libhmleaf.so"))
    L.ref_rd_cost.restype = C.c_double
    L.ref_lambda.restype = C.c_double
    L.ref_cabac_frac.restype = C.c_ulonglong
    L.ref_satd.restype = C.c_uint
    L.ref_sse.restype = C.c_uint
    return L


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def path_arr(path):
    return (C.c_int * max(1, len(path)))(*path)


def main():
    qp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    L = load()
    n_ctu = L.ref_setup(W, H, qp)
    assert n_ctu == 6
    rng = np.random.default_rng(1000 + qp)
    os.makedirs(OUT, exist_ok=True)
    G = {"qp": np.array(qp), "width": np.array(W), "height": np.array(H)}

    # ---- tables
    z2r = np.zeros(256, np.int32); L.ref_zscan_to_raster(p(z2r)); G["z2r"] = z2r
    for t in range(3):
        for l in range(2, 6):
            s = np.zeros(1 << (2 * l), np.int32); L.ref_scan(1, t, l, p(s)); G[f"scan_{t}_{l}"] = s
        for l in range(0, 4):
            s = np.zeros(1 << (2 * l), np.int32); L.ref_scan(0, t, l, p(s)); G[f"scancg_{t}_{l}"] = s
    for l in range(2, 6):
        m = np.zeros(1 << (2 * l), np.int32); L.ref_dct(l, p(m)); G[f"dct_{l}"] = m
    G["chroma_qp"] = np.array([L.ref_chroma_qp(q) for q in range(52)], np.int32)
    G["entropy_bits"] = np.array([L.ref_entropy_bits(i) for i in range(128)], np.int32)
    G["next_state"] = np.array([[L.ref_next_state(s, b) for b in range(2)] for s in range(128)], np.int32)
    G["lambda"] = np.array([L.ref_lambda(i) for i in range(3)], np.float64)
    st = np.zeros(512, np.uint8); n = L.ref_cabac_states(p(st)); G["init_states"] = st[:n].copy()

    # ---- transforms
    for l in range(2, 6):
        n_ = 1 << l
        for dst in ((0, 1) if l == 2 else (0,)):
            xs, cs, ci, ro = [], [], [], []
            for k in range(12):
                amp = [255, 40, 8, 255][k % 4]
                x = rng.integers(-amp, amp + 1, (n_, n_)).astype(np.int16)
                if k % 4 == 3:
                    x[:] = 0; x[rng.integers(0, n_), rng.integers(0, n_)] = 255
                c = np.zeros((n_, n_), np.int32); L.ref_fwd(p(x), l, dst, p(c))
                xs.append(x); cs.append(c)
                q = (rng.integers(-3000, 3001, (n_, n_)) * (rng.random((n_, n_)) < 0.3)).astype(np.int32)
                if k == 0:
                    q[:] = 32767
                if k == 1:
                    q[:] = -32768
                r = np.zeros((n_, n_), np.int16); L.ref_inv(p(q), l, dst, p(r))
                ci.append(q); ro.append(r)
            G[f"fwd_in_{l}_{dst}"] = np.array(xs); G[f"fwd_out_{l}_{dst}"] = np.array(cs)
            G[f"inv_in_{l}_{dst}"] = np.array(ci); G[f"inv_out_{l}_{dst}"] = np.array(ro)

    # ---- SATD / SSE / RD cost
    for n_ in (4, 8, 16, 32, 64):
        a = rng.integers(0, 256, (6, n_, n_)).astype(np.uint8)
        b = np.clip(a.astype(int) + rng.integers(-40, 41, a.shape), 0, 255).astype(np.uint8)
        G[f"dist_a_{n_}"] = a; G[f"dist_b_{n_}"] = b
        G[f"satd_{n_}"] = np.array([L.ref_satd(p(a[i]), p(b[i]), n_, n_) for i in range(6)], np.uint32)
        G[f"sse_{n_}"] = np.array([[L.ref_sse(p(a[i]), p(b[i]), n_, n_, c) for c in range(3)] for i in range(6)], np.uint32)
    bits = rng.integers(0, 200000, 64).astype(np.uint32); dist = rng.integers(0, 1 << 24, 64).astype(np.uint32)
    G["rd_bits"] = bits; G["rd_dist"] = dist
    G["rd_cost"] = np.array([L.ref_rd_cost(int(b), int(d)) for b, d in zip(bits, dist)], np.float64)

    # ---- intra prediction with real availability
    rec = [rng.integers(0, 256, (H, W)).astype(np.uint8), rng.integers(0, 256, (H // 2, W // 2)).astype(np.uint8),
           rng.integers(0, 256, (H // 2, W // 2)).astype(np.uint8)]
    # smooth gradient region so that strong intra smoothing triggers for some 32x32 blocks
    yy, xx = np.mgrid[0:H, 0:W]
    rec[0][:, 96:] = np.clip(40 + xx[:, 96:] // 4 + yy[:, 96:] // 3, 0, 255).astype(np.uint8)
    for c in range(3):
        L.ref_set_rec(c, p(rec[c]))
        G[f"rec_{c}"] = rec[c]
    blocks = []   # (ctu, zidx, depth, path) -> TU
    lite = qp != 32        # prediction does not depend on QP: other QPs carry a small subset
    for ctu in ((4,) if lite else (0, 1, 2, 4, 5)):
        blocks += [(ctu, 0, 0, []), (ctu, 0, 1, []), (ctu, 64, 1, []), (ctu, 128, 1, []), (ctu, 192, 1, []),
                   (ctu, 16, 2, []), (ctu, 48, 2, []), (ctu, 96, 2, []), (ctu, 208, 2, []), (ctu, 240, 2, []),
                   (ctu, 4, 3, []), (ctu, 28, 3, []), (ctu, 84, 3, []), (ctu, 252, 3, []), (ctu, 168, 3, []),
                   (ctu, 0, 3, [0]), (ctu, 20, 3, [1]), (ctu, 44, 3, [2]), (ctu, 252, 3, [3]), (ctu, 100, 3, [3]),
                   (ctu, 64, 1, [1]), (ctu, 128, 1, [2, 3]), (ctu, 192, 2, [0]), (ctu, 0, 0, [3])]
    recs = []
    for (ctu, z, d, path) in blocks:
        for comp in range(3):
            modes = range(35) if comp == 0 else (0, 1, 10, 26, 34, 2, 18)
            size = (64 >> (d + len(path))) >> (1 if comp else 0)
            if size < 4 or (comp == 0 and size > 64):
                continue
            for mode in modes:
                pred = np.zeros((size, size), np.uint8); ru = np.zeros(4 * size + 1, np.int16); rf = np.zeros(4 * size + 1, np.int16)
                r = L.ref_intra(ctu, z, d, len(path), path_arr(path), comp, mode, p(pred), p(ru), p(rf))
                if r == 0:
                    continue
                zz = z2r[z]
                x0, y0 = (ctu % 3) * 64 + (zz % 16) * 4, (ctu // 3) * 64 + (zz // 16) * 4
                s = 64 >> d
                for ch in path:
                    s >>= 1
                    x0 += (ch & 1) * s; y0 += (ch >> 1) * s
                recs.append((comp, x0 >> (1 if comp else 0), y0 >> (1 if comp else 0), int(np.log2(size)), mode, r, pred, ru, rf))
    G["intra_meta"] = np.array([[a[0], a[1], a[2], a[3], a[4], a[5]] for a in recs], np.int32)
    G["intra_pred"] = np.concatenate([a[6].ravel() for a in recs])
    G["intra_ru"] = np.concatenate([a[7] for a in recs])
    G["intra_rf"] = np.concatenate([a[8] for a in recs])

    # ---- MPM / split context from random neighbour decisions
    fields = {}
    for ctu in range(6):
        depth = np.zeros(256, np.uint8)
        for i in range(0, 256, 64):          # random quadtree per 32x32
            d1 = rng.integers(1, 4)
            if d1 == 1:
                depth[i:i + 64] = 1
            else:
                for j in range(i, i + 64, 16):
                    d2 = rng.integers(2, 4)
                    depth[j:j + 16] = d2
        ldir = rng.integers(0, 35, 256).astype(np.uint8)
        pm = np.where(rng.random(256) < 0.9, 1, 0).astype(np.uint8)   # a few "inter" partitions -> DC in the MPM rule
        for f, v in ((0, depth), (3, ldir), (2, pm)):
            L.ref_set_ctu_field(ctu, f, p(v)); fields[(ctu, f)] = v
        G[f"nb_depth_{ctu}"] = depth; G[f"nb_ldir_{ctu}"] = ldir; G[f"nb_pm_{ctu}"] = pm
    mpm = np.zeros((6, 256, 4), np.int32); sctx = np.zeros((6, 256, 3), np.int32)
    for ctu in range(6):
        for part in range(256):
            pr = (C.c_int * 3)()
            mpm[ctu, part, 3] = L.ref_mpm(ctu, part, pr)
            mpm[ctu, part, :3] = list(pr)
            for d in range(3):
                sctx[ctu, part, d] = L.ref_ctx_split(ctu, part, d)
    G["mpm"] = mpm; G["split_ctx"] = sctx
    # restore intra everywhere for the TQ part
    ones = np.ones(256, np.uint8)
    for ctu in range(6):
        L.ref_set_ctu_field(ctu, 2, p(ones))

    # ---- transform + RDOQ + sign hiding + inverse, and coefficient coding with evolving contexts
    L.ref_cabac_reset()
    tq = []
    cfgs = [  # (depth, path, comp)
        (1, [], 0), (2, [], 0), (3, [], 0), (3, [0], 0), (3, [2], 0), (2, [1], 0), (1, [3], 0), (1, [1, 2], 0),
        (1, [], 1), (2, [], 2), (3, [], 1), (1, [2], 2), (2, [3], 1), (0, [1], 1),
    ]
    for it in range(220):
        d, path, comp = cfgs[rng.integers(0, len(cfgs))]
        size = (64 >> (d + len(path))) >> (1 if comp else 0)
        l = int(np.log2(size))
        ldir = int(rng.choice([0, 1, 10, 26, 34, 2, 8, 12, 22, 30, 6, 14, 18]))
        cdir = int(rng.choice([0, 1, 10, 26, 34, 36]))
        part_size = 3 if (d == 3 and len(path) == 1 and comp == 0) else 0
        ts = int(l == 2 and rng.random() < 0.35)
        amp = [3, 10, 30, 90, 200][rng.integers(0, 5)]
        base = rng.normal(0, amp, (size, size))
        if rng.random() < 0.5:   # low-pass structure so that big TUs have few coefficients
            k = np.outer(np.hanning(size + 2)[1:-1], np.hanning(size + 2)[1:-1]); base = base * 0.3 + amp * k * rng.normal()
        resi = np.clip(np.rint(base), -255, 255).astype(np.int16)
        coef = np.zeros(size * size, np.int32); rout = np.zeros(size * size, np.int16)
        ctu = 4
        zidx = [0, 64, 16, 4][d] if d else 0
        a = L.ref_tq(ctu, zidx, d, len(path), path_arr(path), comp, part_size, ldir, cdir, ts, p(resi), p(coef), p(rout))
        assert a >= 0
        if a > 0:                  # advance the contexts like xGetIntraBitsQT would
            L.ref_code_coeff(ctu, zidx, d, len(path), path_arr(path), comp, part_size, ldir, cdir, ts, p(coef))
        st = np.zeros(512, np.uint8); L.ref_cabac_states(p(st))
        tq.append(dict(comp=comp, log2=l, ldir=ldir, cdir=cdir, trd=len(path), ts=ts, abs=a, resi=resi.ravel(), coef=coef, rout=rout,
                       frac=int(L.ref_cabac_frac()), states=st[:185].copy()))
        if it % 37 == 36:
            L.ref_cabac_reset_bits()
            tq[-1]["reset_after"] = 1
    G["tq_meta"] = np.array([[t["comp"], t["log2"], t["ldir"], t["cdir"], t["trd"], t["ts"], t["abs"], t.get("reset_after", 0)] for t in tq], np.int32)
    G["tq_resi"] = np.concatenate([t["resi"] for t in tq]); G["tq_coef"] = np.concatenate([t["coef"] for t in tq])
    G["tq_rout"] = np.concatenate([t["rout"] for t in tq])
    G["tq_frac"] = np.array([t["frac"] for t in tq], np.uint64)
    G["tq_states"] = np.array([t["states"] for t in tq], np.uint8)
    out = os.path.join(OUT, f"leaf_qp{qp}.npz")
    np.savez_compressed(out, **G)
    print("wrote", out, os.path.getsize(out), "bytes;", len(recs), "intra cases,", len(tq), "TQ cases")


if __name__ == "__main__":
    main()
