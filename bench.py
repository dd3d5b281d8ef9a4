#!/usr/bin/env python3
"""bench.py -- CTUs/sec of the CU depth/mode RDO decision (TEncCu::compressCtu equivalent).

Default workload (BASELINE.json configs[2]): 3840x2160 all-intra, QP {22,27,32,37}, full depth-0..3 quadtree + chroma RDO,
synthetic "textured" YUV, resident in HBM before the timed region.  A chain = one slice of one (frame, QP)
= the unit HM decides strictly serially.  Slices are HM's SliceMode 1 with `--slice-ctus` CTUs (default 120 = two CTU rows:
17 slices per 4K frame, the last one holding the partial bottom row), so the timed region walks CTUs without above
neighbours (first row of a slice), CTUs with above / above-right neighbours (second row) and the 48-sample-high bottom
row, and every chain is walked to the end of its slice.  A *step* advances every chain by `--ctus-per-step` CTUs
(compressCtu + encodeCtu replay per CTU) in one launch; chains are ordered longest (lowest QP) first.

`--config ldp` (BASELINE.json configs[4]): 3840x2160 lowdelay_P, QP 32: picture 0 (intra) of every clip is decided,
deblocked and padded untimed; the timed steps decide P-picture CTUs (merge / AMVP / integer ME +-SearchRange -- TZ search as in
the reference cfg, `--fast-search 0` for the full search -- with sub-sample refinement / inter RQT / intra fallback) against
that reference.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...; every rank owns its own frames /
   clips -- no collective on the data path; weak scaling.  `--shard slices`: ONE set of frames whose slice chains are split
   over the ranks (sharding.slices_for_rank), outputs gathered on rank 0 after the timed region; strong scaling.)

Prints ONE JSON line on rank 0 (metric/value/... + "roofline" + "cpu_baseline" + transfer / chain-count / split-match detail).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_CTU = 55300          # src 12288 + neighbours 1024 + rec 12288 + coeff 24576 + meta 5120
# lowdelay_P adds the reference samples a CTU's search can touch, read once: the luma window (64 + 2 * SearchRange)^2 and the
# two chroma blocks with their interpolation margin 2 x (32 + 8)^2 -- SURVEY.md 8(d) gives no figure for configs[4]; DESIGN.md 3e
algo_bytes_ldp = lambda sr: ALGO_BYTES_PER_CTU + (64 + 2 * sr) ** 2 + 2 * (32 + 8) ** 2
HBM_PEAK_GBPS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s
PMC_SUMMARY = {"intra": os.path.join(ROOT, "profiles", "r03_pmc_summary.json"), "ldp": os.path.join(ROOT, "profiles", "r03_pmc_summary_ldp.json")}
# wave instructions the chip's SIMDs can issue per second at the spec clock: 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64
# VALU instruction (MI355X_MICROARCH.md, cycle constants: v_fma_f32 wave64 = 2 cycles on a SIMD-32)
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2


def measured_counters(config, chains):
    """Per-CTU figures of the engine kernel from the committed rocprofv3 PMC passes of this bench (profiles/r03_pmc_summary*.json,
    folded by tools/pmc_summary.py: FETCH_SIZE and WRITE_SIZE in separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for gfx950; SQ_WAIT_ANY / SQ_WAVE_CYCLES; SQ_INSTS_VALU) and the commit the profiled library was built from.
    The counters scale with the CTUs a launch decides, so they are kept per CTU and apply to any number of CTUs per step;
    they are only attached when the profiled run had the same number of chains in flight (waits depend on it)."""
    try:
        with open(PMC_SUMMARY[config]) as f:
            d = json.load(f)
        if d.get("config") != config or d.get("chains_per_launch") != chains:
            return None
        per_launch = d["chains_per_launch"] * d["ctus_per_chain_per_launch"]
        c = d.get("counters_per_launch", {})
        return {"traffic_bytes_per_ctu": float(d["traffic_bytes_per_ctu"]), "wait_share": d.get("wait_share_of_wave_cycles"),
                "valu_per_ctu": (c["SQ_INSTS_VALU"] / per_launch) if "SQ_INSTS_VALU" in c else None,
                "salu_per_ctu": (c["SQ_INSTS_SALU"] / per_launch) if "SQ_INSTS_SALU" in c else None,
                "l2_hit_rate": d.get("l2_hit_rate"), "commit": d.get("commit")}
    except (OSError, ValueError, KeyError, ZeroDivisionError):
        return None


def gen_textured_gpu(torch, dev, w, h, seed, shift=(0, 0)):
    """The SURVEY 8d 'textured' generator evaluated on the device (torch RNG); `shift` moves the deterministic part
    (lowdelay_P clips: global motion of (3, 1) samples per picture) while the noise is drawn afresh per call."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    y = torch.arange(h, device=dev, dtype=torch.float32)[:, None] + shift[1]
    x = torch.arange(w, device=dev, dtype=torch.float32)[None, :] + shift[0]
    noise = torch.randn((h, w), generator=g, device=dev) * 18 * (1 + ((x // 64 + y // 64) % 3)) / 2
    Y = 128 + 50 * torch.sin(x / 9 + y / 31) + 35 * torch.cos(y / 7) * torch.sin(x / 53) + noise
    cy = torch.arange(h // 2, device=dev, dtype=torch.float32)[:, None] + shift[1] / 2
    cx = torch.arange(w // 2, device=dev, dtype=torch.float32)[None, :] + shift[0] / 2
    U = 128 + 25 * torch.sin(cx / 23) + torch.randn((h // 2, w // 2), generator=g, device=dev) * 5 + 0 * cy
    V = 128 + 25 * torch.cos(cy / 19) + torch.randn((h // 2, w // 2), generator=g, device=dev) * 5 + 0 * cx
    f = lambda a: a.round().clamp(0, 255).to(torch.uint8).contiguous()
    return f(Y), f(U), f(V)


def gen_moving_gpu(torch, dev, w, h, seed, poc, shear=0, vec=(3, 1)):
    """picture `poc` of a lowdelay_P clip: one smooth-noise texture (seeded per clip) moving by (3, 1) samples per picture
    plus fresh +-2 noise -- the same recipe as tests/search_trace.py:moving_frame, on the device.  shear: bands of the
    picture (rows 48..63 of every 64 in the left half, columns 0..15 of every 64 in the right half) show a second texture
    moving by (-2, 3): motion boundaries inside the CUs, so the partition shapes of a CU find different vectors."""
    if shear:
        a = gen_moving_gpu(torch, dev, w, h, seed, poc)
        b = gen_moving_gpu(torch, dev, w, h, seed + 50000, poc, vec=(-2, 3))
        yy = torch.arange(h, device=dev)[:, None]; xx = torch.arange(w, device=dev)[None, :]
        m = torch.where(xx < w // 2, (yy % 64) >= 48, (xx % 64) < 16)
        mc = m[::2, ::2]
        return torch.where(m, b[0], a[0]).contiguous(), torch.where(mc, b[1], a[1]).contiguous(), torch.where(mc, b[2], a[2]).contiguous()
    Y, U, V = gen_textured_gpu(torch, dev, w + 32, h + 32, seed)
    dx, dy = (vec[0] * poc) % 32, (vec[1] * poc) % 32
    g = torch.Generator(device=dev)
    g.manual_seed(100000 + 17 * seed + poc)
    n = torch.randint(-2, 3, (h, w), generator=g, device=dev, dtype=torch.int16)
    Yc = (Y[dy:dy + h, dx:dx + w].to(torch.int16) + n).clamp(0, 255).to(torch.uint8).contiguous()
    return Yc, U[dy // 2:dy // 2 + h // 2, dx // 2:dx // 2 + w // 2].contiguous(), V[dy // 2:dy // 2 + h // 2, dx // 2:dx // 2 + w // 2].contiguous()


def split_match(want, got):
    return int(((want["depth"] == got["depth"]) & (want["part_size"] == got["part_size"]) & (want["pred_mode"] == got["pred_mode"])).sum()), want["depth"].size


def cpu_leg(make_encoder, first_ctu, gpu_lookup, budget_s, label):
    """The oracle (plain-C restatement of the reference loop, pinned candidate by candidate by the reference's own search
    code -- oracle/README.md) on this box's host cores, 1 thread, on the first CTUs of a chain of the same workload.  A
    reported baseline, not the target.  As the checker it compares the CU split decisions (depth, part size and prediction
    mode per 4x4 partition) the GPU published for the same CTUs -- the second half of BASELINE's metric."""
    enc = make_encoder()
    t0 = time.time()
    n = same = total = 0
    while first_ctu + n < enc.n_ctu and (time.time() - t0) < budget_s:
        a = first_ctu + n
        enc.compress_ctu(a)
        got = gpu_lookup(a)
        if got is not None:
            s, t = split_match(enc.ctu_arrays(a), got)
            same += s
            total += t
        n += 1
    dt = time.time() - t0
    return n, dt, same, total, label


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=["intra", "ldp"], default="intra")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--frames", type=int, default=0, help="frames (intra) / clips (ldp) per GPU; 0 = 120 (intra: x4 QPs x17 slices = 8160 chains) / 240 (ldp: x17 slices = 4080 chains)")
    ap.add_argument("--slice-ctus", type=int, default=120, help="HM SliceMode 1 / SliceArgument: CTUs per slice = per chain (0: one slice per frame)")
    ap.add_argument("--ctus-per-step", type=int, default=0, help="0 = walk the whole slice over warmup + steps launches; lowdelay_P with the full search: 2")
    ap.add_argument("--qps", default="22,27,32,37")
    ap.add_argument("--search-range", type=int, default=64)
    ap.add_argument("--fast-search", type=int, default=1, help="lowdelay_P integer motion search: 1 = TZ search (FastSearch 1, the reference cfg's setting), 0 = full search")
    ap.add_argument("--amp", type=int, default=0, help="lowdelay_P: 1 = asymmetric motion partitions (AMP 1 of the reference cfg)")
    ap.add_argument("--refs", type=int, default=1, help="lowdelay_P: reference pictures in list 0 (1..4; the reference cfg's num_ref_idx_active is 4): pictures 0..refs-1 "
                    "are decided untimed, picture `refs` is timed with the `refs` pictures before it as RefPicList0, most recent first")
    ap.add_argument("--shear", type=int, default=0, help="lowdelay_P: 1 = clips with motion boundaries inside the CUs (bands moving with a second vector) instead of one global motion")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true", help="skip the chain-count sweep and the transfer measurement")
    ap.add_argument("--shard", choices=["frames", "slices"], default="frames")
    ap.add_argument("--state", choices=["training", "testing"], default="training",
                    help="fork state of the timed steps (intra).  training (default, the headline metric): exhaustive RDO.  testing: "
                         "the fork's Naive pruning; OBF maps from the device pre-pass, the first warm-up step runs in the Verifying "
                         "state and its counters set the per-depth switches (SetDecisionSwitch)")
    ap.add_argument("--no-ldp-leg", action="store_true", help="default intra run on one GPU: skip the short lowdelay_P measurement reported under \"ldp\" in the same JSON line")
    return ap.parse_args(argv)


def run(args, emit=True):
    """One measurement (args.config) -> the result dict on rank 0 (None elsewhere); printed as the JSON line when `emit`."""
    import numpy as np
    import torch
    import __graft_entry__ as g
    pkg = g.load_package()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # FCU_BENCH_REHEARSAL=1 (never set by the driver): rehearse the N>1 code path on a one-GPU box -- all ranks share
    # cuda:0 and rendezvous over gloo, since RCCL refuses two ranks on one device.  Not a scaling measurement; says so.
    rehearsal = os.environ.get("FCU_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    ldp = args.config == "ldp"
    W, H = args.width, args.height
    n_ctu = ((W + 63) // 64) * ((H + 63) // 64)
    sl = args.slice_ctus if args.slice_ctus > 0 else n_ctu
    n_sl = (n_ctu + sl - 1) // sl
    qps = [32] if ldp else [int(q) for q in args.qps.split(",")]
    n_frames = args.frames or (240 if ldp else 120)
    # whole slices are walked over warmup + steps launches (interior CTUs, second CTU row, partial bottom row in the timed
    # region); only the full search of lowdelay_P, 5x slower, keeps to 2 CTUs per step
    cps = args.ctus_per_step or (2 if (ldp and not args.fast_search) else max(1, sl // (args.warmup + args.steps)))
    walked = (args.warmup + args.steps) * cps
    assert walked <= sl, "bench walks past the end of the slice"
    strong = args.shard == "slices" and world > 1
    # chains: (frame seed, qp, slice).  weak scaling: every rank owns its own frames; --shard slices: all ranks see the same
    # frames and each owns whole slices of them (no slice is shared, no collective on the data path)
    seeds = [s for s, _ in pkg.sharding.chains_for_rank(n_frames, [0], 0 if strong else rank)]
    owner_of = lambda k: 0
    if strong:
        owned = {first // sl: r for r in range(world) for first, _ in pkg.sharding.slices_for_rank(n_ctu, sl, world, r)}
        owner_of = lambda k: owned[k]
    my_slices = [k for k in range(n_sl) if not strong or owner_of(k) == rank]
    chain_list = sorted(((seed, qp, k) for seed in seeds for qp in qps for k in my_slices), key=lambda c: (c[1], c[0], c[2]))
    n_chains = len(chain_list)
    eng = pkg.CuEngine(W, H, max_chains=n_chains, device=local)
    nb = pkg.engine.CTU_OUT_BYTES

    frames, recs, outs, refs, fps = {}, {}, {}, {}, {}
    for seed in seeds:
        frames[seed] = gen_moving_gpu(torch, dev, W, H, seed, args.refs if ldp else 0, shear=args.shear) if ldp else gen_textured_gpu(torch, dev, W, H, seed)
    for seed in seeds:
        for qp in qps:
            recs[(seed, qp)] = [torch.zeros_like(p) for p in frames[seed]]
            outs[(seed, qp)] = torch.zeros(nb * n_ctu, dtype=torch.uint8, device=dev)

    if ldp:
        # untimed: pictures 0 .. refs-1 of every clip (picture 0 intra, the others P on the pictures before them; same slice
        # structure), loop filter, padding -> the reference pictures of the timed picture `refs`, most recent first
        assert 1 <= args.refs <= 4
        pads = {seed: [] for seed in seeds}
        for poc in range(args.refs):
            fpp = pkg.engine.ldp_slice(32, poc)
            fpp.search_range, fpp.fast_search, fpp.amp = args.search_range, args.fast_search, args.amp
            pics = {seed: gen_moving_gpu(torch, dev, W, H, seed, poc, shear=args.shear) for seed in seeds}
            for ci, (seed, qp, k) in enumerate(chain_list):
                kw = {} if poc == 0 else (dict(ref=pads[seed][0]) if poc == 1 else dict(refs=list(pads[seed]), ref_pocs=list(range(poc - 1, -1, -1)), poc=poc))
                eng.init_chain(ci, pics[seed], fpp.qp, slice_ctus=sl if n_sl > 1 else 0, rec=recs[(seed, qp)], out=outs[(seed, qp)], params=fpp, **kw)
                if n_sl > 1:
                    eng.set_range(ci, k * sl, min(sl, n_ctu - k * sl))
            eng.compress_chains(0, n_chains, sl)
            for seed in seeds:
                eng.deblock(out=outs[(seed, 32)], rec=recs[(seed, 32)])
                pads[seed].insert(0, eng.pad_reference(recs[(seed, 32)]))
                recs[(seed, 32)] = [torch.zeros_like(p) for p in frames[seed]]
            eng.sync()
            del pics
        refs = {seed: pads[seed][0] for seed in seeds}
        ref_pocs = list(range(args.refs - 1, -1, -1))
        fp1 = pkg.engine.ldp_slice(32, args.refs)
        fp1.search_range = args.search_range
        fp1.fast_search = args.fast_search
        fp1.amp = args.amp

    def bind(ci):
        seed, qp, k = chain_list[ci]
        if ldp:
            kw = dict(ref=refs[seed]) if args.refs == 1 else dict(refs=pads[seed], ref_pocs=ref_pocs, poc=args.refs)
            eng.init_chain(ci, frames[seed], fp1.qp, slice_ctus=sl if n_sl > 1 else 0, rec=recs[(seed, qp)], out=outs[(seed, qp)], params=fp1, **kw)
        else:
            eng.init_chain(ci, frames[seed], qp=qp, slice_ctus=sl if n_sl > 1 else 0, rec=recs[(seed, qp)], out=outs[(seed, qp)])
        if n_sl > 1:
            eng.set_range(ci, k * sl, min(sl, n_ctu - k * sl))

    for ci in range(n_chains):
        bind(ci)
    torch.cuda.synchronize()

    def step():
        eng.compress_chains(0, n_chains, cps)

    switches = None
    if args.state == "testing" and not ldp:
        assert args.warmup >= 1, "--state testing uses the first warm-up step as the Verifying step"
        obf = {seed: eng.obf_prepass(frames[seed][0])[0][0].contiguous() for seed in seeds}
        for ci, (seed, qp, k) in enumerate(chain_list):
            eng.set_decision(ci, pkg.engine.VERIFYING, obf[seed])
        step()
        ver = eng.verify_counts(0, n_chains)
        switches = pkg.engine.decision_switch(ver)
        for ci, (seed, qp, k) in enumerate(chain_list):
            eng.set_decision(ci, pkg.engine.TESTING, obf[seed], *switches)
    for _ in range(args.warmup - (1 if switches is not None else 0)):
        step()
    eng.sync()
    eng.kernel_ms()                       # drop warm-up launches from the event accumulator
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    eng.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = pkg.sharding.reduce_step_time(dist, time.perf_counter() - t0, None if rehearsal else dev)
    kernel_ms, launches = eng.kernel_ms()

    # chains at the end of a frame have fewer CTUs left than cps: count what was really decided in the timed steps
    timed_ctus_gpu = 0
    for seed, qp, k in chain_list:
        n_in_slice = min(sl, n_ctu - k * sl)
        timed_ctus_gpu += max(0, min(n_in_slice, walked) - min(n_in_slice, args.warmup * cps))
    if dist is not None:
        t = torch.tensor([timed_ctus_gpu], dtype=torch.float64, device=None if rehearsal else dev)
        dist.all_reduce(t)
        total = int(t.item())
    else:
        total = timed_ctus_gpu
    value = total / dt

    if strong:                             # after the timed region: the other ranks' slices of every frame reach rank 0
        for key in sorted(outs):
            v = outs[key].view(n_ctu, nb)
            for k in range(n_sl):
                owner = owner_of(k)
                a, b = k * sl, min(n_ctu, (k + 1) * sl)
                buf = v[a:b].contiguous() if not rehearsal else v[a:b].cpu().contiguous()
                dist.broadcast(buf, src=owner)
                if rank == 0 and owner != 0:
                    v[a:b] = buf.to(dev)

    if rank == 0:
        per_launch = timed_ctus_gpu / max(1, args.steps)
        algo = algo_bytes_ldp(args.search_range) if ldp else ALGO_BYTES_PER_CTU
        achieved = (algo * per_launch) / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        pmc = measured_counters(args.config, n_chains) if (switches is None and not args.amp and not args.shear and args.refs == 1 and world == 1) else None
        traffic = pmc["traffic_bytes_per_ctu"] * per_launch if pmc else None
        kernel_ctus_per_s = per_launch / (kernel_ms * 1e-3) if kernel_ms > 0 else 0.0
        what = (f"{W}x{H} lowdelay_P QP32 (BASELINE configs[4]): P pictures referencing the deblocked, padded picture{'s' if args.refs > 1 else ' 0'} before them in their clip; "
                f"merge + AMVP + {'TZ search (FastSearch 1)' if args.fast_search else 'full search (FastSearch 0)'} +-{args.search_range} (FEN) + half/quarter refinement (HadamardME) + inter RQT + intra fallback; "
                f"{'one reference picture' if args.refs == 1 else str(args.refs) + ' reference pictures (the ' + str(args.refs) + ' pictures before it, decided untimed)'}, TMVP off, AMP {'on' if args.amp else 'off'}{'; sheared motion (--shear 1)' if args.shear else ''}" if ldp else
                f"{W}x{H} all-intra QP{{{args.qps}}}, full depth-0..3 quadtree + chroma RDO (BASELINE configs[2])")
        res = {
            "metric": "CTUs/sec (RDO decision only) at 4K " + ("lowdelay_P" if ldp else "all-intra"), "value": value, "unit": "CTUs/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "int32+f64",
            "data": "synthetic",
            "config": {"workload": what + f"; {len(seeds)} {'clips' if ldp else 'frames'} x {len(qps)} QP x {len(my_slices)} slices of {sl} CTUs (HM SliceMode 1) = {n_chains} chains/GPU, "
                                          f"{cps} CTUs/chain/step, lowest QP first",
                       "chains_per_gpu": n_chains, "ctus_per_step": int(per_launch) * (1 if strong else world),
                       "timed_ctu_range": f"CTUs {args.warmup * cps}..{walked - 1} of every {sl}-CTU slice: first-row CTUs (no above neighbour), second-row CTUs "
                                          f"(above / above-right neighbours) and, in the last slice of a frame, the partial bottom CTU row",
                       **({"rehearsal": "all ranks on one GPU over gloo: not a scaling measurement"} if rehearsal else {}),
                       **({"shard": "slices of the same frames over the ranks; outputs gathered on rank 0 after the timed region"} if strong else {}),
                       "state": args.state if switches is None else
                       f"testing (Naive switches skip2Nx2N={switches[0].tolist()} terminate={switches[1].tolist()} from a Verifying step)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_bytes_per_ctu": pmc["traffic_bytes_per_ctu"] if pmc else None,
                         "traffic_GBps": (traffic / (kernel_ms * 1e-3) / 1e9) if (traffic and kernel_ms > 0) else None,
                         "traffic_profile_commit": pmc["commit"] if pmc else None,
                         "kernel": "fcu_ctu_engine", "kernel_ms": kernel_ms, "launches": launches, "algorithmic_bytes_per_ctu": algo,
                         # what actually limits the kernel (`bound` above is the nominal roofline BASELINE asks for)
                         "limit": "latency: instruction issue of one serial stream per chain (RDOQ / CABAC bit counting), each wave waiting on memory for about wait_share of its cycles",
                         "wait_share": pmc["wait_share"] if pmc else None,
                         "valu_instructions_per_ctu": pmc["valu_per_ctu"] if pmc else None,
                         "valu_issue_frac": (pmc["valu_per_ctu"] * kernel_ctus_per_s / VALU_ISSUE_PEAK) if (pmc and pmc["valu_per_ctu"]) else None,
                         "note": "`achieved` counts algorithmic bytes only; `traffic` = HBM-side bytes per launch (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE per CTU from the committed PMC "
                                 "passes of this bench, times the CTUs of a launch); `valu_issue_frac` = VALU wave instructions retired per second over 256 CUs x 4 SIMDs x 2.4 GHz / 2: DESIGN.md 3"},
        }
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        if world == 1 and not args.no_cpu_baseline:
            import hmo_py
            qp_list = [32] if ldp else qps
            legs = []
            for qp in qp_list:
                ci = chain_list.index((seeds[0], qp, 0))
                fr = [p.cpu().numpy() for p in frames[seeds[0]]]
                if ldp:
                    _, q1, lam = hmo_py.ldp_slice(args.refs, 32)
                    host_refs = [[p.cpu().numpy() for p in _unpad(t, W, H)] for t in pads[seeds[0]]]
                    rk = dict(ref=host_refs[0]) if args.refs == 1 else dict(refs=host_refs, ref_pocs=ref_pocs, poc=args.refs)
                    mk = lambda fr=fr: hmo_py.Encoder(*fr, q1, slice_ctus=sl if n_sl > 1 else 0, lambda_override=lam, search_range=args.search_range,
                                                      fast_search=args.fast_search, amp=args.amp, **rk)
                else:
                    mk = lambda fr=fr, qp=qp: hmo_py.Encoder(*fr, qp, slice_ctus=sl if n_sl > 1 else 0)
                look = lambda a, ci=ci: eng.ctu_out(ci, a) if a < walked and switches is None else None
                legs.append(cpu_leg(mk, 0, look, 15.0 if qp == 32 else 1.5, f"QP{qp}"))
            n32, dt32 = [(n, d) for n, d, _, _, lab in legs if lab == "QP32"][0]
            res["cpu_baseline"] = {"value": n32 / dt32, "unit": "CTUs/sec", "cores": 1, "kind": "port",
                                   "sample": f"first {n32} CTUs of slice 0 of {'clip' if ldp else 'frame'} 0 ({W}x{H}, QP32{', P picture' if ldp else ''}), oracle/libhmo.so single thread, {dt32:.1f}s"}
            same, tot = sum(l[2] for l in legs), sum(l[3] for l in legs)
            res["split_flag_match"] = {"partitions_equal": same, "partitions": tot, "frac": (same / tot) if tot else None,
                                       "chains": [f"{lab}: {n} CTUs" for n, _, _, _, lab in legs],
                                       "what": "depth + part size + prediction mode per 4x4 partition, one GPU chain per QP vs the in-repo oracle on the same CTUs; "
                                               "the oracle's PU/TU search loops are pinned by the reference's own TEncSearch.cpp, its CU-level glue (xCompressCU) is restated"}
        else:
            res["cpu_baseline"] = None
        if world == 1 and not args.no_sweep:
            # (b) transfers the metric's definition puts next to the decision: source planes in, fcu_ctu_out out (pinned host memory)
            nf = min(8, len(seeds))
            host = [torch.empty((nf,) + tuple(p.shape), dtype=torch.uint8).pin_memory() for p in frames[seeds[0]]]
            devb = [torch.empty((nf,) + tuple(p.shape), dtype=torch.uint8, device=dev) for p in frames[seeds[0]]]
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for hb, db in zip(host, devb):
                db.copy_(hb, non_blocking=True)
            torch.cuda.synchronize()
            h2d_s = time.perf_counter() - t1
            src_bytes = sum(hb.numel() for hb in host)
            ob = outs[(seeds[0], qps[0])]
            hout = torch.empty(ob.shape, dtype=torch.uint8).pin_memory()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            hout.copy_(ob, non_blocking=True)
            torch.cuda.synchronize()
            d2h_s = time.perf_counter() - t1
            h2d_gbps, d2h_gbps = src_bytes / h2d_s / 1e9, ob.numel() / d2h_s / 1e9
            # per decided CTU: its share of the source planes (once per frame, shared by the QP chains) + its fcu_ctu_out
            per_ctu_in = (W * H * 3 // 2) / n_ctu / (1 if ldp else len(qps))
            xfer_s = total * (per_ctu_in / (h2d_gbps * 1e9) + nb / (d2h_gbps * 1e9))
            res["transfers"] = {"h2d_GBps": h2d_gbps, "d2h_GBps": d2h_gbps, "source_bytes_per_ctu": per_ctu_in, "ctu_out_bytes": nb,
                                "seconds_for_timed_ctus": xfer_s, "value_incl_transfers": total / (dt + xfer_s),
                                "note": "pinned host memory, measured after the timed region; `value` is with inputs resident in HBM"}
            # (c) throughput against the number of chains in flight (SURVEY 8d config 3 asks for >= 16 frames: 64 chains = 16 frames x 4 QPs)
            sweep = {}
            k = 2 if ldp else 4
            for n in (64, 256, 1024, 4096, n_chains):
                if n > n_chains or str(n) in sweep:
                    continue
                for ci in range(n):
                    bind(ci)
                eng.sync()
                t1 = time.perf_counter()
                eng.compress_chains(0, n, k)
                eng.sync()
                sweep[str(n)] = n * k / (time.perf_counter() - t1)
            res["chains_sweep"] = {"ctus_per_sec": sweep, "ctus_per_chain": k,
                                   "note": "first CTUs of the first n chains (lowest QP first), one launch each; one chain = one wavefront"}
            if not ldp and n_sl > 1:
                # (d) the reference's own slice configuration (SliceMode 0: one slice per picture, so one chain per (frame, QP)):
                # what a stream that cannot be cut into slices sees -- reported next to the SliceMode-1 headline, never as it
                n1 = min(len(seeds) * len(qps), 480)
                pairs = [(seed, qp) for qp in qps for seed in seeds][:n1]
                for ci, (seed, qp) in enumerate(pairs):
                    eng.init_chain(ci, frames[seed], qp=qp, slice_ctus=0, rec=recs[(seed, qp)], out=outs[(seed, qp)])
                eng.sync()
                t1 = time.perf_counter()
                eng.compress_chains(0, n1, k)
                eng.sync()
                res["slice_mode_0"] = {"chains": n1, "ctus_per_sec": n1 * k / (time.perf_counter() - t1), "ctus_per_chain": k,
                                       "note": "one slice per picture (the reference cfg's SliceMode 0): one chain per (frame, QP); a chain advances about "
                                               "5 CTUs/s, so throughput is the number of independent pictures in flight times that"}
    eng.destroy()
    del frames, recs, outs, refs
    torch.cuda.empty_cache()
    if rank == 0 and emit:
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    return res if rank == 0 else None


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.config != "intra" or world > 1 or args.no_ldp_leg or args.state != "training":
        run(args)
        return
    # the default command (the driver's): the intra measurement is the JSON line; a short lowdelay_P measurement (BASELINE
    # configs[4], the shape of `--config ldp`: 240 clips x 17 slices, TZ search, AMP off, fewer steps) rides under "ldp"
    res = run(args, emit=False)
    a2 = parse_args(["--config", "ldp", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-sweep"])
    ldp = run(a2, emit=False)
    res["ldp"] = {k: ldp[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step")}
    res["ldp"]["workload"] = ldp["config"]["workload"]
    res["ldp"]["timed_ctu_range"] = ldp["config"]["timed_ctu_range"]
    res["ldp"]["roofline"] = ldp["roofline"]
    print(json.dumps(res), flush=True)


def _unpad(padded, w, h):
    """padded reference planes (engine.pad_reference) -> the picture planes"""
    out = []
    for k, t in enumerate(padded):
        m = 80 >> (1 if k else 0)
        pw, ph = (w >> (1 if k else 0)) + 2 * m, (h >> (1 if k else 0)) + 2 * m
        out.append(t.view(ph, pw)[m:ph - m, m:pw - m])
    return out


if __name__ == "__main__":
    main()
