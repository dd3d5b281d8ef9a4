#!/usr/bin/env python3
"""Randomised GPU-vs-oracle sweep of the lowdelay_P pipeline (a measurement script, not part of the test suite): short clips
of random size (off the CTU grid), base QP, content and motion; per picture random slices, search range, TZ / full search,
RDOQ / RDOQTS, TMVP, tool flags, one to four reference pictures (the last n decided pictures), P pictures started from the P or
the B context tables; decide -> deblock (random offsets) -> optional SAO -> padded reference of the next picture.
Every fcu_ctu_out field, the reconstruction, the coder state, the deblocked and SAO-filtered planes and the signalled SAO
parameters must be identical to the oracle's.  One line per clip and a summary; exit code 1 on any mismatch."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]


def canon(params):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clips", type=int, default=30)
    ap.add_argument("--seed", type=int, default=2027)
    args = ap.parse_args()
    import torch
    import __graft_entry__ as g
    import hmo_py
    pkg = g.load_package()
    rng = np.random.default_rng(args.seed)
    bad = 0
    for clip in range(args.clips):
        w, h = int(rng.integers(8, 33)) * 8, int(rng.integers(8, 21)) * 8
        base_qp = int(rng.integers(10, 45))
        gen = ["smooth", "mixed", "textured"][int(rng.integers(0, 3))]
        nref = int(rng.choice([1, 1, 2, 3, 4]))                 # reference pictures in list 0: the last nref decided pictures
        n_pic = int(rng.integers(2, 5)) + (nref > 1) * int(rng.integers(1, 3))
        btab = int(rng.random() < 0.3)                          # P pictures start from the B-slice context tables (cabac_init_flag)
        dxy = (int(rng.integers(-5, 6)), int(rng.integers(-3, 4)))
        seed = int(rng.integers(0, 10000))
        w_ctu, n_ctu = (w + 63) // 64, ((w + 63) // 64) * ((h + 63) // 64)
        sl = int(rng.choice([0, 0, 1, w_ctu]))
        sr = int(rng.choice([4, 8, 16, 32, 64]))
        fast = int(rng.integers(0, 2))
        rdoq, rdoq_ts = (1, 1) if rng.random() < 0.7 else (int(rng.integers(0, 2)), int(rng.integers(0, 2)))
        tools = dict(transform_skip=int(rng.random() < 0.8), transform_skip_fast=int(rng.integers(0, 2)), sign_hiding=int(rng.random() < 0.8),
                     strong_intra_smoothing=int(rng.integers(0, 2)))
        sao = int(rng.random() < 0.6)
        tmvp = int(rng.random() < 0.5)
        amp = int(rng.random() < 0.5)
        shear = int(rng.random() < 0.5)                      # bands of the picture move the other way: motion boundaries at CU quarters
        boff, toff = int(rng.integers(-2, 3)), int(rng.integers(-2, 3))
        big = getattr(pkg.synth, gen)(w + 64, h + 64, seed=seed)
        eng = pkg.CuEngine(w, h, max_chains=1)
        rate_e, rate_o = pkg.engine.SaoRate(), hmo_py.SaoState()
        prev = pad = prev_out = prev_ctus = None
        dpb = []                                                # (poc, filtered planes, padded device planes, the POCs its list 0 named)
        diffs = []
        for poc in range(n_pic):
            ox, oy = 32 + dxy[0] * poc, 32 + dxy[1] * poc
            nz = np.random.default_rng(seed * 31 + poc).integers(-2, 3, (h, w))
            Y = np.clip(big[0][oy:oy + h, ox:ox + w].astype(np.int16) + nz, 0, 255).astype(np.uint8)
            U = np.ascontiguousarray(big[1][oy // 2:oy // 2 + h // 2, ox // 2:ox // 2 + w // 2])
            V = np.ascontiguousarray(big[2][oy // 2:oy // 2 + h // 2, ox // 2:ox // 2 + w // 2])
            if shear:
                ox2, oy2 = 32 - dxy[1] * poc, 32 + dxy[0] * poc
                yy, xx = np.mgrid[0:h, 0:w]
                m = np.where(xx < w // 2, (yy % 64) >= 48, (xx % 64) < 16)
                mc = m[::2, ::2]
                Y = np.where(m, np.clip(big[0][oy2:oy2 + h, ox2:ox2 + w].astype(np.int16) + nz, 0, 255).astype(np.uint8), Y)
                U = np.where(mc, big[1][oy2 // 2:oy2 // 2 + h // 2, ox2 // 2:ox2 // 2 + w // 2], U)
                V = np.where(mc, big[2][oy2 // 2:oy2 // 2 + h // 2, ox2 // 2:ox2 // 2 + w // 2], V)
            f = (np.ascontiguousarray(Y), np.ascontiguousarray(U), np.ascontiguousarray(V))
            fp = pkg.engine.ldp_slice(base_qp, poc)
            stype, qp, lam = hmo_py.ldp_slice(poc, base_qp)
            if not (0 <= qp <= 51):
                break
            fp.search_range, fp.fast_search, fp.rdoq, fp.rdoq_ts = sr, fast, rdoq, rdoq_ts
            fp.tmvp = 1 if (tmvp and poc) else 0
            fp.amp = amp
            fp.cabac_b_table = 1 if (btab and poc) else 0
            for k, v in tools.items():
                setattr(fp, k, v)
            rl = dpb[-nref:][::-1]
            pocs = [r[0] for r in rl]
            crp = (rl[0][3] or [rl[0][0] - 1]) if rl else None
            if poc and nref > 1:
                eng.init_chain(0, f, fp.qp, slice_ctus=sl, params=fp, refs=[r[2] for r in rl], ref_pocs=pocs, poc=poc, col_ref_pocs=crp, col=prev_out if fp.tmvp else None)
            else:
                eng.init_chain(0, f, fp.qp, slice_ctus=sl, params=fp, ref=pad, col=prev_out if fp.tmvp else None)
            eng.compress_chains(0, 1, n_ctu)
            eng.sync()
            ot = dict(tools)
            ot["strong_smoothing"] = ot.pop("strong_intra_smoothing")
            kw = dict(slice_ctus=sl, lambda_override=lam, rdoq=rdoq, rdoq_ts=rdoq_ts, **ot)
            if poc == 0:
                o = hmo_py.Encoder(*f, qp, **kw)
            elif nref > 1:
                o = hmo_py.Encoder(*f, qp, refs=[r[1] for r in rl], ref_pocs=pocs, poc=poc, col_ref_pocs=crp, col=prev_ctus if tmvp else None, search_range=sr,
                                   fast_search=fast, amp=amp, cabac_b_table=btab, **kw)
            else:
                o = hmo_py.Encoder(*f, qp, ref=prev, col=prev_ctus if tmvp else None, search_range=sr, fast_search=fast, amp=amp, cabac_b_table=btab, **kw)
            o.compress_frame()
            for a in range(n_ctu):
                got, want = eng.ctu_out(0, a), o.ctu_arrays(a)
                for k, v in want.items():
                    if not (np.array_equal(v, got[k]) if isinstance(v, np.ndarray) else v == got[k]):
                        diffs.append(f"poc{poc}.ctu{a}.{k}")
            if any(not np.array_equal(p, q) for p, q in zip(eng.rec_planes(0), o.rec)):
                diffs.append(f"poc{poc}.rec")
            (ce, fe), (co, fo) = eng.ctx_state(0, full=True), o.cabac(full=True)
            if fe != fo or not np.array_equal(ce[:175], co[:175]):
                diffs.append(f"poc{poc}.cabac")
            prev_out, prev_ctus = eng._keep[0][2], o.all_ctus_bytes()
            eng.deblock(0, boff, toff)
            eng.sync()
            o.deblock(boff, toff)
            if any(not np.array_equal(p, q) for p, q in zip(eng.rec_planes(0), o.rec)):
                diffs.append(f"poc{poc}.deblock")
            rec = [p.copy() for p in o.rec]
            if sao:
                layer = hmo_py.ldp_layer(poc)
                en_o, en_e = rate_o.enabled(layer), rate_e.enabled(layer)
                if en_o != en_e:
                    diffs.append(f"poc{poc}.sao_enabled")
                params, off, _ = hmo_py.sao_picture(f, rec, qp, stype, lam, enabled=en_o, slice_ctus=sl)
                coded, off_e, _ = eng.sao([{"org": eng._keep[0][0], "rec": eng._keep[0][1], "qp": fp.qp, "lambda_": fp.lambda_, "slice_type": fp.slice_type,
                                            "slice_ctus": sl, "enabled": en_e}])
                rate_o.update(layer, off, n_ctu)
                rate_e.update(layer, off_e[0], n_ctu)
                if list(off_e[0]) != off or not np.array_equal(canon(pkg.engine.sao_coded_to_array(coded[0])), canon(params)):
                    diffs.append(f"poc{poc}.sao_params")
                if any(not np.array_equal(p, q) for p, q in zip(eng.rec_planes(0), rec)):
                    diffs.append(f"poc{poc}.sao_planes")
            prev = rec
            pad = eng.pad_reference(eng._keep[0][1])
            dpb.append((poc, rec, pad, pocs if poc else []))
        eng.destroy()
        bad += bool(diffs)
        print(f"clip {clip:3d} {gen:8s} {w}x{h} qp{base_qp:2d} pics {n_pic} motion {dxy} slice_ctus {sl} sr {sr} fast {fast} rdoq {rdoq}/{rdoq_ts} "
              f"tools {list(tools.values())} sao {sao} tmvp {tmvp} amp {amp} shear {shear} refs {nref} btab {btab} dbk {boff}/{toff}: {'OK' if not diffs else 'MISMATCH ' + ','.join(diffs[:6])}", flush=True)
    print(f"{args.clips - bad} of {args.clips} clips identical")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
