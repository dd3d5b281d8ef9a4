"""The C-ABI library loads and exports every symbol include/fcu.h declares; without a GPU it
fails loudly instead of falling back to a CPU path."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "fcu.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fcu_[a-z_0-9]+)\s*\(", txt)))


def test_header_symbols_are_exported(built, pkg):
    lib = C.CDLL(pkg.lib_path())
    syms = declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), f"libfcu.so does not export {s}"
    assert sorted(pkg.engine.EXPORTS) == syms


def test_build_line_carries_the_exec_mask_workaround(built, pkg):
    """The shipped engine is only correct when built with `-mllvm -amdgpu-remove-redundant-endcf=false` (DESIGN.md 2:
    without it a VGPR reload lands under a partial exec mask -> stale lanes -> GPU memory fault).  The library
    reports its own build line; a build that lost the flag, or whose compiler is not the recorded one, fails here
    on the CPU before anything reaches a GPU."""
    import __graft_entry__ as g
    lib = C.CDLL(pkg.lib_path())
    lib.fcu_build_info.restype = C.c_char_p
    info = lib.fcu_build_info().decode()
    assert g.REQUIRED_FLAG in info and "-mllvm" in info, info
    assert "-ffp-contract=off" in info and "--offload-arch=gfx950" in info, info
    assert "clang" in info and "HIP" in info, info


def test_unknown_frame_parameter_is_rejected(pkg):
    """init_chain must not drop a misspelt tool flag (the oracle says strong_smoothing, the ABI strong_intra_smoothing)."""
    known = {n for n, _ in pkg.engine.FrameParams._fields_}
    assert "strong_intra_smoothing" in known and "strong_smoothing" not in known
    src = open(os.path.join(ROOT, "fast-cu-decision-hevc_amd", "engine.py")).read()
    assert "unknown frame parameter" in src


def test_struct_layouts_match_between_binding_and_oracle(built, pkg):
    import hmo_py
    assert C.sizeof(pkg.engine.CtuOut) == C.sizeof(hmo_py.Ctu)
    for (n1, _), (n2, _) in zip(pkg.engine.CtuOut._fields_, hmo_py.Ctu._fields_):
        assert n1 == n2
        assert getattr(pkg.engine.CtuOut, n1).offset == getattr(hmo_py.Ctu, n2).offset


def test_binding_layouts_match_the_library(built, pkg):
    """The ctypes structures of engine.py against sizeof() inside libfcu.so (fcu_abi_sizeof): a field added on one side only
    (the round-2 `amp` parameter was the occasion) must fail here, not corrupt memory on the GPU box.  Defaults of the frame
    parameters are checked on the way (host arithmetic, no GPU)."""
    e = pkg.engine
    lib = C.CDLL(pkg.lib_path())
    want = {0: C.sizeof(e.CtuOut), 1: C.sizeof(e.SeqParams), 2: C.sizeof(e.FrameParams), 3: C.sizeof(e.DecisionParams),
            4: C.sizeof(e.VerifyCounts), 5: C.sizeof(e.SaoCtu), 6: C.sizeof(e.SaoParams), 7: e.PU_TRACE_DTYPE.itemsize}
    for which, size in want.items():
        assert lib.fcu_abi_sizeof(which) == size, (which, lib.fcu_abi_sizeof(which), size)
    assert lib.fcu_abi_sizeof(99) == -1
    fp = e.FrameParams()
    lib.fcu_default_frame_params(C.byref(fp), 32)
    assert (fp.qp, fp.rdoq, fp.rdoq_ts, fp.amp, fp.tmvp, fp.fast_search, fp.max_merge_cand, fp.search_range) == (32, 1, 1, 0, 0, 0, 5, 64)


def test_no_gpu_means_loud_failure(built, pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = pkg.load_lib()
    sp = pkg.engine.SeqParams(128, 64, 1, 0)
    h = C.c_void_p()
    r = lib.fcu_create(C.byref(sp), C.byref(h))
    assert r == -1 and not h.value                      # FCU_ERR_NO_DEVICE
    assert b"no CPU fallback" in lib.fcu_last_error()
    with pytest.raises(pkg.FcuError):
        pkg.CuEngine(128, 64)


def test_bad_arguments_are_rejected(built, pkg):
    lib = pkg.load_lib()
    h = C.c_void_p()
    sp = pkg.engine.SeqParams(130, 64, 1, 0)            # width not a multiple of the minimum CU size
    assert lib.fcu_create(C.byref(sp), C.byref(h)) == -2
    assert lib.fcu_chain_position(None, 0) == -1


def test_slices_partition_a_frame(pkg):
    """4K frame, one CTU row per slice, 8 ranks: every CTU belongs to exactly one chain range."""
    n_ctu, sl = 2040, 60
    seen = []
    for rank in range(8):
        for first, n in pkg.sharding.slices_for_rank(n_ctu, sl, 8, rank):
            assert first % sl == 0 and n > 0
            seen += list(range(first, first + n))
    assert seen == list(range(n_ctu))


def test_ranks_never_share_a_frame(pkg):
    """bench.py's weak-scaling shards: every (frame seed, QP) chain belongs to exactly one rank, also when a rank owns
    more frames than the default seed stride."""
    for frames in (2, 1024, 2048, 5000):
        seen = set()
        for rank in range(8):
            mine = pkg.sharding.chains_for_rank(frames, [22, 27, 32, 37], rank)
            assert len(mine) == frames * 4 and len(set(mine)) == len(mine)
            assert not (seen & set(mine))
            seen |= set(mine)


def test_exchange_head_is_the_decision_part_of_a_ctu_record(pkg):
    """sharding.merge_picture sends the head of every fcu_ctu_out record: exactly the bytes in front of the coefficient arrays"""
    assert pkg.engine.CtuOut.coeff_y.offset == pkg.sharding.CTU_HEAD_BYTES
    assert pkg.engine.CtuOut.mvd.offset + pkg.engine.CtuOut.mvd.size == pkg.sharding.CTU_HEAD_BYTES
