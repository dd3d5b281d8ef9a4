"""N>1 path on CPU: two gloo ranks each decide their own shard of chains (with the emulated engine) and
the union equals the single-process result; the step time is the MAX over ranks."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, zlib
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests")); sys.path.insert(0, os.path.join(%(root)r, "oracle"))
import numpy as np, torch, torch.distributed as dist
import __graft_entry__ as g
pkg = g.load_package()
import emu_py
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
sums = []
for seed, qp in pkg.sharding.chains_for_rank(2, [27, 37], rank):
    Y, U, V = pkg.synth.mixed(64, 64, seed=seed)
    e = emu_py.EmuEncoder(Y, U, V, qp)
    e.compress_frame()
    a = e.ctu_arrays(0)
    sums.append(zlib.crc32(a["depth"].tobytes() + a["intra_dir"].tobytes() + a["coeff_y"].tobytes() + e.rec[0].tobytes()))
t = torch.tensor(sums, dtype=torch.int64)
out = [torch.zeros_like(t) for _ in range(world)]
dist.all_gather(out, t)
step = pkg.sharding.reduce_step_time(dist, 1.0 + rank)
if rank == 0:
    import json
    print("RESULT " + json.dumps([[o.tolist() for o in out], step]))
dist.destroy_process_group()
'''


def test_two_rank_gloo_sharding(built, pkg, tmp_path):
    import zlib
    import emu_py
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0]
    import json
    got = json.loads(line[len("RESULT "):])
    sums, step = got
    assert step == 2.0                                   # MAX over ranks of (1.0, 2.0)
    seen = set()
    for rank in range(2):
        want = []
        for seed, qp in pkg.sharding.chains_for_rank(2, [27, 37], rank):
            assert (seed, qp) not in seen                # shards are disjoint
            seen.add((seed, qp))
            Y, U, V = pkg.synth.mixed(64, 64, seed=seed)
            e = emu_py.EmuEncoder(Y, U, V, qp)
            e.compress_frame()
            a = e.ctu_arrays(0)
            want.append(zlib.crc32(a["depth"].tobytes() + a["intra_dir"].tobytes() + a["coeff_y"].tobytes() + e.rec[0].tobytes()))
        assert sums[rank] == want


SLICE_WORKER = r'''
import os, sys, zlib, json, ctypes as C
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests")); sys.path.insert(0, os.path.join(%(root)r, "oracle"))
import numpy as np, torch, torch.distributed as dist
import __graft_entry__ as g
pkg = g.load_package()
import emu_py, hmo_py, search_trace as st
from test_inter_emu import dbk_emu
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
w, h, base_qp, sl, sr, n_pic = 192, 128, 30, 2, 8, 3          # 6 CTUs, 3 slices of 2 CTUs: rank 0 owns slices 0-1, rank 1 slice 2
n_ctu = 6
mine = pkg.sharding.slices_for_rank(n_ctu, sl, world, rank)
owner = pkg.sharding.slice_owner(n_ctu, sl, world)
assert [owner[f // sl] for f, _ in mine] == [rank] * len(mine)
prev, crcs = None, []
for poc in range(n_pic):
    f = st.moving_frame(pkg.synth, "mixed", w, h, 21, poc)
    _, qp, lam = hmo_py.ldp_slice(poc, base_qp)
    e = emu_py.EmuEncoder(*f, qp, slice_ctus=sl, lam=lam) if poc == 0 else emu_py.EmuEncoder(*f, qp, slice_ctus=sl, ref=prev, lam=lam, search_range=sr)
    for first, n in mine:                                      # this rank's slices only
        for a in range(first, first + n):
            e.compress_ctu(a)
    planes = [torch.from_numpy(p) for p in e.rec]              # share memory with the emulator's planes / output array
    out = torch.from_numpy(np.frombuffer(e.out, dtype=np.uint8))
    planes = [p.view(h >> (1 if k else 0), w >> (1 if k else 0)) for k, p in enumerate(planes)]
    nb = C.sizeof(hmo_py.Ctu)
    pkg.sharding.merge_picture(dist, planes, out, sl, nb)      # the one exchange of the picture: own samples + decision heads
    dbk_emu(e.out, e.rec, w, h)                                # every rank filters the whole picture itself
    prev = [p.copy() for p in e.rec]
    v = np.frombuffer(e.out, dtype=np.uint8).reshape(n_ctu, nb)
    own = np.zeros(n_ctu, bool)
    for first, n in mine:
        own[first:first + n] = True
    # planes and decision heads of the whole picture; whole records (coefficients, totals) of this rank's own CTUs
    crcs.append([zlib.crc32(p.tobytes()) for p in e.rec] + [zlib.crc32(v[:, :pkg.sharding.CTU_HEAD_BYTES].tobytes())] +
                [zlib.crc32(v[a].tobytes()) if own[a] else -1 for a in range(n_ctu)])
t = torch.tensor(crcs, dtype=torch.int64)
got = [torch.zeros_like(t) for _ in range(world)]
dist.all_gather(got, t)
if rank == 0:
    print("RESULT " + json.dumps([g_.tolist() for g_ in got]))
dist.destroy_process_group()
'''


def test_slices_of_a_lowdelay_clip_over_two_ranks(built, pkg, tmp_path):
    """bench.py --shard slices / the inter reference hand-off: slices of every picture of a lowdelay_P clip decided on two
    ranks (engine source on the CPU), one merge_picture exchange per picture (each rank broadcasts the samples and the
    decision heads of its own CTUs), loop filter replicated -- every rank ends every picture with the planes and decisions of
    the single-process run; the coefficient arrays stay with the rank that produced them."""
    import json
    import zlib
    import emu_py
    import hmo_py
    import search_trace as st
    from test_inter_emu import dbk_emu
    script = tmp_path / "slice_worker.py"
    script.write_text(SLICE_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29543")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29543", str(script)],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    got = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT")][0][len("RESULT "):])
    head_n = 4                                                 # three planes + the decision heads: the same on both ranks
    assert [r[:head_n] for r in got[0]] == [r[:head_n] for r in got[1]]
    w, h, base_qp, sl, sr, n_pic = 192, 128, 30, 2, 8, 3
    prev = None
    for poc in range(n_pic):
        f = st.moving_frame(pkg.synth, "mixed", w, h, 21, poc)
        _, qp, lam = hmo_py.ldp_slice(poc, base_qp)
        e = emu_py.EmuEncoder(*f, qp, slice_ctus=sl, lam=lam) if poc == 0 else emu_py.EmuEncoder(*f, qp, slice_ctus=sl, ref=prev, lam=lam, search_range=sr)
        e.compress_frame()
        dbk_emu(e.out, e.rec, w, h)
        prev = [p.copy() for p in e.rec]
        v = np.frombuffer(e.out, dtype=np.uint8).reshape(-1, C.sizeof(hmo_py.Ctu))
        assert got[0][poc][:head_n] == [zlib.crc32(p.tobytes()) for p in e.rec] + [zlib.crc32(v[:, :pkg.sharding.CTU_HEAD_BYTES].tobytes())], poc
        whole = [zlib.crc32(v[a].tobytes()) for a in range(len(v))]
        for a in range(len(v)):                                  # every CTU's whole record (levels, totals) lives on exactly its owner
            owners = [r for r in range(2) if got[r][poc][head_n + a] != -1]
            assert len(owners) == 1 and got[owners[0]][poc][head_n + a] == whole[a], (poc, a)
