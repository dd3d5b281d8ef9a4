#!/usr/bin/env python3
"""Decide the CU quadtrees of an all-intra 8-bit 4:2:0 .yuv clip on an MI355X -- the part of `TAppEncoder -c
encoder_intra_main.cfg` that this repository replaces (no bitstream is written).

  python tools/fcu_decide.py -i clip_1920x1080.yuv -w 1920 -h 1080 -q 32 -f 10 --fast --rec rec.yuv --depth depth.npy

--fast runs the fork's Training / Verifying / Testing cycle (period / training / verifying pictures as in the reference:
60 / 2 / 1); without it every picture gets the exhaustive HM search.  --rec receives the deblocked reconstruction,
--depth an array [pictures, CTUs, 256] of CU depths per 4x4 partition in z-order (TComDataCU::getDepth).

Slices: by default a picture is ONE slice, as in the reference's configuration (SliceMode 0) -- pictures side by side
(--in-flight) are then the source of parallelism.  --slice-ctus N / --row-slices select HM's SliceMode 1 with N CTUs (one
CTU row) per slice: a different encoder configuration (neighbourhood cut and CABAC reset at every slice start), whose
slices are decided concurrently.  The slice mode is echoed on every picture line.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser(add_help=False)
    ap.add_argument("--help", action="help")
    ap.add_argument("-i", "--input", required=True)
    ap.add_argument("-w", "--width", type=int, required=True)
    ap.add_argument("-h", "--height", type=int, required=True)
    ap.add_argument("-q", "--qp", type=int, default=32)
    ap.add_argument("-f", "--frames", type=int, default=1 << 30)
    ap.add_argument("--fast", action="store_true")
    ap.add_argument("--period", type=int, default=60)
    ap.add_argument("--training", type=int, default=2)
    ap.add_argument("--verifying", type=int, default=1)
    ap.add_argument("--no-deblock", action="store_true")
    ap.add_argument("--in-flight", type=int, default=16, help="pictures decided side by side when the schedule allows it")
    ap.add_argument("--slice-ctus", type=int, default=0, help="SliceMode 1: CTUs per slice (default: one slice per picture)")
    ap.add_argument("--row-slices", action="store_true", help="SliceMode 1 with one CTU row per slice")
    ap.add_argument("--rec")
    ap.add_argument("--depth")
    args = ap.parse_args()
    import __graft_entry__ as g
    pkg = g.load_package()
    seq = pkg.sequence
    slice_ctus = (args.width + 63) // 64 if args.row_slices else (args.slice_ctus or None)
    dec = seq.SequenceDecider(args.width, args.height, args.qp, slice_ctus=slice_ctus, fast=args.fast, deblock=not args.no_deblock, in_flight=args.in_flight,
                              schedule=seq.FastDecisionSchedule(args.period, args.training, args.verifying))
    names = {seq.TRAINING: "training", seq.VERIFYING: "verifying", seq.TESTING: "testing"}
    rec_f = open(args.rec, "wb") if args.rec else None
    depths = []
    n = 0
    t_all = time.perf_counter()
    while n < args.frames:
        group = []
        for i in range(min(dec.group_size(), args.frames - n)):
            yuv = seq.read_yuv420(args.input, args.width, args.height, n + i)
            if yuv is None:
                break
            group.append(yuv)
        if not group:
            break
        t0 = time.perf_counter()
        res = dec.decide_group(group)
        dt = time.perf_counter() - t0
        for r, yuv in zip(res, group):
            # PSNR of the (deblocked) reconstruction as TEncGOP::xCalculateAddPSNR reports it (TEncGOP.cpp): 10 log10(255^2 N / SSD)
            psnr = []
            for p, o in zip(r["rec"], yuv):
                d = p.cpu().numpy().astype(np.int64) - o.astype(np.int64)
                ssd = float((d * d).sum())
                psnr.append(999.99 if ssd == 0 else 10.0 * np.log10(255.0 * 255.0 * d.size / ssd))
            hist = np.bincount(r["depth"].ravel(), minlength=4)
            print(f"POC {r['poc']:4d} {names[r['state']]:9s} [{dec.slice_mode}] skip2Nx2N={r['sw_skip'].tolist()} terminate={r['sw_term'].tolist()} "
                  f"partitions at depth 0..3 = {hist.tolist()}  TU trials {r['tu_trials']}  "
                  f"PSNR Y {psnr[0]:.4f} U {psnr[1]:.4f} V {psnr[2]:.4f} dB", flush=True)
            if rec_f:
                seq.write_yuv420(rec_f, [p.cpu().numpy() for p in r["rec"]])
            if args.depth:
                depths.append(r["depth"])
        print(f"  {len(res)} picture(s) side by side: {dt * 1e3:.1f} ms", flush=True)
        n += len(res)
    dt_all = time.perf_counter() - t_all
    if rec_f:
        rec_f.close()
    if args.depth:
        np.save(args.depth, np.stack(depths) if depths else np.zeros((0, 0, 256), np.uint8))
    dec.close()
    print(f"{n} pictures decided in {dt_all:.2f} s")


if __name__ == "__main__":
    main()
