"""Host side of the engine: ctypes binding of libfcu.so and a `TEncCu`-shaped class.

The reference boundary is the C++ class `TEncCu` (Lib/TLibEncoder/TEncCu.h:104-118):
create / init / compressCtu / encodeCtu / destroy, called once per CTU by
`TEncSlice::compressSlice` (TEncSlice.cpp:1380-1551).  `CuEngine` keeps those names and
their meaning; the batched form (`compress_chains`) is what a throughput-oriented caller uses.

torch is plumbing only (device buffers, streams); the arithmetic is in the HIP kernels.
The module raises if libfcu.so is missing or no GPU is present: there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
NPART = 256


class FcuError(RuntimeError):
    pass


class FrameParams(C.Structure):
    """fcu_frame_params (include/fcu.h)."""
    _fields_ = [("qp", C.c_int), ("slice_ctus", C.c_int), ("transform_skip", C.c_int),
                ("transform_skip_fast", C.c_int), ("sign_hiding", C.c_int), ("strong_intra_smoothing", C.c_int),
                ("lambda_", C.c_double), ("sqrt_lambda", C.c_double), ("chroma_weight", C.c_double),
                ("rdoq_lambda", C.c_double * 3),
                ("slice_type", C.c_int), ("search_range", C.c_int), ("fast_enc", C.c_int), ("hadamard_me", C.c_int),
                ("fast_merge_decision", C.c_int), ("max_merge_cand", C.c_int), ("fast_search", C.c_int), ("tmvp", C.c_int), ("rdoq", C.c_int), ("rdoq_ts", C.c_int), ("amp", C.c_int), ("cabac_b_table", C.c_int)]


class SeqParams(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("max_chains", C.c_int), ("device", C.c_int)]


class CtuOut(C.Structure):
    """fcu_ctu_out: the TComDataCU arrays published by copyToPic (TComDataCU.h:72-164)."""
    _fields_ = [("depth", C.c_uint8 * NPART), ("width", C.c_uint8 * NPART), ("height", C.c_uint8 * NPART),
                ("skip", C.c_uint8 * NPART), ("part_size", C.c_int8 * NPART), ("pred_mode", C.c_int8 * NPART),
                ("tq_bypass", C.c_uint8 * NPART), ("qp", C.c_int8 * NPART), ("chroma_qp_adj", C.c_uint8 * NPART),
                ("tr_idx", C.c_uint8 * NPART), ("tskip", (C.c_uint8 * NPART) * 3), ("cbf", (C.c_uint8 * NPART) * 3),
                ("intra_dir", (C.c_uint8 * NPART) * 2), ("ipcm", C.c_uint8 * NPART),
                ("merge_flag", C.c_uint8 * NPART), ("merge_idx", C.c_uint8 * NPART), ("inter_dir", C.c_uint8 * NPART),
                ("mvp_idx", C.c_int8 * NPART), ("ref_idx", C.c_int8 * NPART),
                ("mv", (C.c_int16 * 2) * NPART), ("mvd", (C.c_int16 * 2) * NPART),
                ("coeff_y", C.c_int32 * 4096), ("coeff_cb", C.c_int32 * 1024), ("coeff_cr", C.c_int32 * 1024),
                ("total_cost", C.c_double), ("total_dist", C.c_uint32), ("total_bits", C.c_uint32),
                ("total_bins", C.c_uint32)]


CTU_OUT_BYTES = C.sizeof(CtuOut)

TRAINING, VERIFYING, TESTING = 0, 1, 2          # CurrentState (getCurrentState, tools_YS.cpp:1237-1242)


class SaoOffset(C.Structure):
    _fields_ = [("mode", C.c_int8), ("type", C.c_int8), ("band", C.c_int8), ("pad", C.c_int8), ("offset", C.c_int8 * 32)]


class SaoCtu(C.Structure):
    _fields_ = [("c", SaoOffset * 3)]


class SaoParams(C.Structure):
    _fields_ = [("slice_type", C.c_int32), ("qp", C.c_int32), ("slice_ctus", C.c_int32), ("enabled", C.c_int32 * 3), ("lambda_", C.c_double * 3)]


SAO_CTU_BYTES = C.sizeof(SaoCtu)


class DecisionParams(C.Structure):
    """fcu_decision_params (include/fcu.h): frame state, Naive switches per depth, the frame's OBF map."""
    _fields_ = [("state", C.c_int), ("depth_exception", C.c_int), ("sw_skip2nx2n", C.c_uint8 * 4),
                ("sw_terminate", C.c_uint8 * 4), ("dev_obf", C.c_void_p)]


class VerifyCounts(C.Structure):
    """fcu_verify_counts: g_iVerResult[depth][TP, FP, TN, FN, FPLoss, FNLoss]."""
    _fields_ = [("n", (C.c_double * 6) * 4)]

EXPORTS = ["fcu_default_frame_params", "fcu_create", "fcu_destroy", "fcu_num_ctus", "fcu_chain_begin",
           "fcu_compress_chains", "fcu_compress_ctu", "fcu_get_ctx_state", "fcu_chain_position", "fcu_sync",
           "fcu_kernel_ms", "fcu_last_error", "fcu_debug_counters", "fcu_chain_set_range", "fcu_obf_prepass", "fcu_chains_per_cu",
           "fcu_chain_set_decision", "fcu_get_verify_counts", "fcu_decision_switch", "fcu_frame_state", "fcu_deblock",
           "fcu_build_info", "fcu_abi_sizeof", "fcu_tcm_threshold", "fcu_chain_set_reference", "fcu_pad_reference", "fcu_pad_sizes", "fcu_ldp_slice", "fcu_get_ctx_state_full",
           "fcu_sao", "fcu_sao_enabled", "fcu_sao_update_rate", "fcu_ldp_layer", "fcu_chain_set_pu_trace", "fcu_pu_index", "fcu_chain_set_collocated",
           "fcu_chain_set_references", "fcu_chain_set_collocated_pocs", "fcu_chain_get_search_state", "fcu_chain_set_search_state"]
MAX_REF = 4                                                # FCU_MAX_REF: reference pictures in list 0

SLICE_I, SLICE_P = 0, 1
PUS_PER_CTU = 341
PU_TRACE_DTYPE = np.dtype([("valid", np.uint8), ("best_mode", np.uint8), ("n_rmd", np.uint8), ("n_rd", np.uint8), ("rd_mode", np.uint8, (12,)),
                           ("best_dist", np.uint32), ("pad", np.uint32), ("best_cost", np.float64), ("rmd_cost", np.float64, (8,))])   # fcu_pu_trace


def pu_index(depth, nxn, zidx):
    """fcu_pu_index: position of a PU inside its CTU's 341 records"""
    return load_lib().fcu_pu_index(depth, nxn, zidx)

REF_MARGIN = 80


def lib_path():
    # FCU_LIB: another build of the same library next to it (diagnostic variants: profiling timers, register budgets)
    return os.path.join(_HERE, os.environ.get("FCU_LIB", "libfcu.so"))


_lib = None


def load_lib():
    """Loads libfcu.so (the HIP extension).  Fails loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    p = lib_path()
    if not os.path.exists(p):
        raise FcuError(f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    lib = C.CDLL(p)
    lib.fcu_create.argtypes = [C.POINTER(SeqParams), C.POINTER(C.c_void_p)]
    lib.fcu_destroy.argtypes = [C.c_void_p]
    lib.fcu_num_ctus.argtypes = [C.c_void_p]
    lib.fcu_default_frame_params.argtypes = [C.POINTER(FrameParams), C.c_int]
    lib.fcu_chain_begin.argtypes = [C.c_void_p, C.c_int, C.POINTER(FrameParams)] + [C.c_void_p] * 7
    lib.fcu_compress_chains.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.fcu_chain_set_range.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.fcu_obf_prepass.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_void_p]
    lib.fcu_compress_ctu.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.POINTER(CtuOut)]
    lib.fcu_get_ctx_state.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_uint64)]
    lib.fcu_get_ctx_state_full.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_uint64)]
    lib.fcu_chain_position.argtypes = [C.c_void_p, C.c_int]
    lib.fcu_sync.argtypes = [C.c_void_p]
    lib.fcu_kernel_ms.restype = C.c_double
    lib.fcu_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    lib.fcu_debug_counters.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_ulonglong)]
    lib.fcu_last_error.restype = C.c_char_p
    lib.fcu_build_info.restype = C.c_char_p
    lib.fcu_tcm_threshold.restype = C.c_double
    lib.fcu_tcm_threshold.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.fcu_chain_set_reference.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.fcu_pad_reference.argtypes = [C.c_void_p] * 8
    lib.fcu_pad_sizes.restype = None
    lib.fcu_pad_sizes.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    lib.fcu_ldp_slice.restype = None
    lib.fcu_ldp_slice.argtypes = [C.POINTER(FrameParams), C.c_int, C.c_int]
    lib.fcu_chain_set_decision.argtypes = [C.c_void_p, C.c_int, C.POINTER(DecisionParams)]
    lib.fcu_get_verify_counts.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(VerifyCounts)]
    lib.fcu_decision_switch.restype = None
    lib.fcu_decision_switch.argtypes = [C.POINTER(VerifyCounts), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.fcu_frame_state.argtypes = [C.c_int] * 4
    lib.fcu_deblock.argtypes = [C.c_void_p] * 5 + [C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
    lib.fcu_sao.argtypes = [C.c_void_p, C.c_int, C.POINTER(SaoParams), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.c_void_p]
    lib.fcu_sao_enabled.restype = None
    lib.fcu_sao_enabled.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.fcu_sao_update_rate.restype = None
    lib.fcu_sao_update_rate.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    lib.fcu_ldp_layer.argtypes = [C.c_int]
    lib.fcu_chain_set_pu_trace.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.fcu_chain_set_collocated.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.fcu_chain_set_references.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.c_int]
    lib.fcu_chain_set_collocated_pocs.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]
    lib.fcu_chain_get_search_state.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int32)]
    lib.fcu_chain_set_search_state.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int32)]
    _lib = lib
    return lib


def decision_switch(ver, th_skip=(0, 0, 0, 0), th_term=(0, 0, 0, 0)):
    """SetDecisionSwitch (tools_YS.cpp:1123-1154) on summed verification counters [4, 6]: (sw_skip[4], sw_term[4]).
    Host arithmetic of libfcu.so; needs no GPU."""
    lib = load_lib()
    v = VerifyCounts()
    a = np.ascontiguousarray(ver, np.float64).reshape(4, 6)
    for d in range(4):
        for k in range(6):
            v.n[d][k] = a[d, k]
    ts, tt = np.ascontiguousarray(th_skip, np.float64), np.ascontiguousarray(th_term, np.float64)
    sk, te = np.zeros(4, np.uint8), np.zeros(4, np.uint8)
    lib.fcu_decision_switch(C.byref(v), ts.ctypes.data, tt.ctypes.data, sk.ctypes.data, te.ctypes.data)
    return sk, te


def sao_coded_to_array(coded):
    """uint8 [.., n_ctu, SAO_CTU_BYTES] (device tensor or array) -> int32 [.., n_ctu, 3, 35]: mode, type, band, offset[32]"""
    a = coded.cpu().numpy() if hasattr(coded, "cpu") else np.asarray(coded)
    a = a.view(np.int8).reshape(a.shape[:-1] + (3, 36)).astype(np.int32)
    return np.concatenate([a[..., :3], a[..., 4:]], axis=-1)


class SaoRate:
    """m_saoDisabledRate across the pictures of one sequence (fcu_sao_enabled / fcu_sao_update_rate)"""

    def __init__(self):
        self.rate = np.zeros((3, 8), np.float64)

    def enabled(self, layer):
        e = np.zeros(3, np.int32)
        load_lib().fcu_sao_enabled(self.rate.ctypes.data, layer, e.ctypes.data)
        return [int(v) for v in e]

    def update(self, layer, off_count, n_ctu):
        oc = np.ascontiguousarray(off_count, np.int32)
        load_lib().fcu_sao_update_rate(self.rate.ctypes.data, layer, oc.ctypes.data, n_ctu)


def ldp_layer(poc):
    return load_lib().fcu_ldp_layer(poc)


def ldp_slice(base_qp, poc):
    """FrameParams of picture `poc` under HM's lowdelay_P GOP table (fcu_ldp_slice): slice type, QP, lambda, inter defaults."""
    fp = FrameParams()
    load_lib().fcu_ldp_slice(C.byref(fp), base_qp, poc)
    return fp


def frame_state(poc, period=60, n_training=2, n_verifying=1):
    """getCurrentState (tools_YS.cpp:1237-1242) with the reference's defaults g_iP / g_iT / g_iV (tools_YS.cpp:41-43)."""
    return load_lib().fcu_frame_state(poc, period, n_training, n_verifying)


def ctu_to_dict(c):
    out = {}
    for name, _ in CtuOut._fields_:
        v = getattr(c, name)
        out[name] = np.ctypeslib.as_array(v).copy() if hasattr(v, "_length_") else v
    return out


class CuEngine:
    """`TEncCu` stand-in for a set of independent chains (frames / slices) on one GPU.

    create()        <- TEncCu::create  : allocates the per-chain device scratch
    init_chain()    <- TEncCu::init + TEncSlice::setUpLambda : binds planes, QP, lambda
    compress_ctu()  <- TEncCu::compressCtu + encodeCtu : one CTU of one chain, result to host
    compress_chains(): the batched form, many chains x k CTUs per launch
    destroy()       <- TEncCu::destroy
    """

    def __init__(self, width, height, max_chains=1, device=0):
        import torch
        if not torch.cuda.is_available():
            raise FcuError("no GPU visible: the CU engine has no CPU fallback")
        self.torch = torch
        self.lib = load_lib()
        self.width, self.height, self.max_chains, self.device = width, height, max_chains, device
        self.h = C.c_void_p()
        self._keep = {}
        self._keep_obf = {}
        self._keep_ref = {}
        self.create()

    # -- TEncCu::create
    def create(self):
        sp = SeqParams(self.width, self.height, self.max_chains, self.device)
        r = self.lib.fcu_create(C.byref(sp), C.byref(self.h))
        if r != 0:
            raise FcuError(f"fcu_create failed ({r}): {self.lib.fcu_last_error().decode()}")
        self.n_ctu = self.lib.fcu_num_ctus(self.h)

    def _chk(self, r, what):
        if r != 0:
            raise FcuError(f"{what} failed ({r}): {self.lib.fcu_last_error().decode()}")

    # -- TEncCu::init + slice parameters
    def init_chain(self, chain, org, qp, slice_ctus=0, rec=None, out=None, ref=None, params=None, col=None,
                   refs=None, ref_pocs=None, poc=None, col_ref_pocs=None, **flags):
        """org: (Y,U,V) uint8 torch tensors on this device (or numpy arrays, uploaded once).
        params: a FrameParams to start from (e.g. ldp_slice(base_qp, poc)) instead of the I-slice defaults for `qp`;
        ref: padded reference planes from pad_reference() -- required for a P slice;
        col: the reference picture's fcu_ctu_out array (TMVP, with params.tmvp = 1);
        refs / ref_pocs / poc: several reference pictures instead of `ref` -- RefPicList0 as a list of pad_reference() triplets
        with their POCs and this picture's POC; col_ref_pocs: the POCs the list 0 of refs[0] (the collocated picture) named."""
        if refs is not None:
            assert ref is None and 1 <= len(refs) <= MAX_REF and len(ref_pocs) == len(refs) and poc is not None
        torch = self.torch
        dev = torch.device("cuda", self.device)
        planes = []
        for a in org:
            t = torch.as_tensor(a) if not torch.is_tensor(a) else a
            planes.append(t.to(device=dev, dtype=torch.uint8).contiguous())
        if rec is None:
            rec = [torch.zeros_like(p) for p in planes]
        if out is None:
            out = torch.zeros(self.n_ctu * CTU_OUT_BYTES, dtype=torch.uint8, device=dev)
        fp = FrameParams()
        if params is not None:
            C.memmove(C.byref(fp), C.byref(params), C.sizeof(FrameParams))
        else:
            self.lib.fcu_default_frame_params(C.byref(fp), qp)
        fp.slice_ctus = slice_ctus
        known = {n for n, _ in FrameParams._fields_}
        for k, v in flags.items():
            if k not in known:                                  # a misspelt tool flag must not be dropped silently
                raise TypeError(f"init_chain: unknown frame parameter {k!r} (fcu_frame_params has {sorted(known)})")
            setattr(fp, k, v)
        self._chk(self.lib.fcu_chain_begin(self.h, chain, C.byref(fp), *[p.data_ptr() for p in planes],
                                           *[p.data_ptr() for p in rec], out.data_ptr()), "fcu_chain_begin")
        self._keep[chain] = (planes, rec, out)
        if ref is not None:
            self._chk(self.lib.fcu_chain_set_reference(self.h, chain, *[p.data_ptr() for p in ref]), "fcu_chain_set_reference")
            self._keep_ref[chain] = ref
        if refs is not None:
            ptrs = (C.c_void_p * (3 * len(refs)))(*[p.data_ptr() for r in refs for p in r])
            pocs = (C.c_int * len(refs))(*[int(v) for v in ref_pocs])
            self._chk(self.lib.fcu_chain_set_references(self.h, chain, len(refs), ptrs, pocs, int(poc)), "fcu_chain_set_references")
            self._keep_ref[chain] = refs
            if col_ref_pocs:
                crp = (C.c_int * len(col_ref_pocs))(*[int(v) for v in col_ref_pocs])
                self._chk(self.lib.fcu_chain_set_collocated_pocs(self.h, chain, int(ref_pocs[0]), crp, len(col_ref_pocs)), "fcu_chain_set_collocated_pocs")
        if col is not None:                                   # TMVP: the reference picture's fcu_ctu_out array (uint8 device tensor)
            assert col.is_cuda and col.numel() >= self.n_ctu * CTU_OUT_BYTES
            self._chk(self.lib.fcu_chain_set_collocated(self.h, chain, col.data_ptr()), "fcu_chain_set_collocated")
            self._keep_ref[("col", chain)] = col
        return rec, out

    def search_state(self, chain):
        """m_integerMv2Nx2N of the chain: [(x, y)] per reference index (fcu_chain_get_search_state)"""
        xy = (C.c_int32 * (2 * MAX_REF))()
        self._chk(self.lib.fcu_chain_get_search_state(self.h, chain, xy), "fcu_chain_get_search_state")
        return [(int(xy[2 * r]), int(xy[2 * r + 1])) for r in range(MAX_REF)]

    def set_search_state(self, chain, state):
        """after init_chain: the state the encoder's previous searches left (fcu_chain_set_search_state)"""
        xy = (C.c_int32 * (2 * MAX_REF))(*[int(v) for p in (list(state) + [(0, 0)] * MAX_REF)[:MAX_REF] for v in p])
        self._chk(self.lib.fcu_chain_set_search_state(self.h, chain, xy), "fcu_chain_set_search_state")

    def pad_reference(self, planes, stream=None):
        """(Y, U, V) device planes of a reconstructed (loop-filtered) picture -> padded planes for P chains
        (TComPicYuv::extendPicBorder).  Returns three uint8 tensors."""
        torch = self.torch
        sz = (C.c_size_t * 3)()
        self.lib.fcu_pad_sizes(self.h, sz)
        dev = torch.device("cuda", self.device)
        src = [torch.as_tensor(p).to(device=dev, dtype=torch.uint8).contiguous() for p in planes]
        out = [torch.empty(int(sz[k]), dtype=torch.uint8, device=dev) for k in range(3)]
        s = C.c_void_p(stream.cuda_stream) if stream is not None else None
        self._chk(self.lib.fcu_pad_reference(self.h, *[p.data_ptr() for p in src], *[p.data_ptr() for p in out], s), "fcu_pad_reference")
        return out

    def init_slice_chains(self, first_chain, org, qp, slice_ctus, **flags):
        """One frame as ceil(n_ctu / slice_ctus) chains, one per slice (SliceMode 1): they share the frame's
        source / reconstruction planes and fcu_ctu_out array.  Returns (n_slices, rec, out)."""
        n_sl = (self.n_ctu + slice_ctus - 1) // slice_ctus
        rec, out = self.init_chain(first_chain, org, qp, slice_ctus=slice_ctus, **flags)
        planes = self._keep[first_chain][0]
        for k in range(n_sl):
            if k:
                self.init_chain(first_chain + k, planes, qp, slice_ctus=slice_ctus, rec=rec, out=out, **flags)
            first = k * slice_ctus
            self.set_range(first_chain + k, first, min(slice_ctus, self.n_ctu - first))
        return n_sl, rec, out

    def set_range(self, chain, first_ctu, n_ctus):
        self._chk(self.lib.fcu_chain_set_range(self.h, chain, first_ctu, n_ctus), "fcu_chain_set_range")

    # -- TEncCu::compressCtu (+ encodeCtu replay)
    def compress_ctu(self, chain, ctu_rs_addr):
        c = CtuOut()
        self._chk(self.lib.fcu_compress_ctu(self.h, chain, ctu_rs_addr, C.byref(c)), "fcu_compress_ctu")
        return ctu_to_dict(c)

    def compress_chains(self, first, n, ctus, stream=None):
        s = C.c_void_p(stream.cuda_stream) if stream is not None else None
        self._chk(self.lib.fcu_compress_chains(self.h, first, n, ctus, s), "fcu_compress_chains")

    def sync(self):
        self._chk(self.lib.fcu_sync(self.h), "fcu_sync")

    def kernel_ms(self):
        n = C.c_int(0)
        ms = self.lib.fcu_kernel_ms(self.h, C.byref(n))
        return ms, n.value

    def debug_counters(self, chain):
        buf = (C.c_ulonglong * 17)()
        self._chk(self.lib.fcu_debug_counters(self.h, chain, buf), "fcu_debug_counters")
        return list(buf)

    def position(self, chain):
        return self.lib.fcu_chain_position(self.h, chain)

    def ctx_state(self, chain, full=False):
        """context states after the chain's last CTU: the 160 of an I slice, or all 176 (full=True), + the Q15 counter"""
        ctx = np.zeros(176 if full else 160, np.uint8)
        frac = C.c_uint64(0)
        if full:
            self._chk(self.lib.fcu_get_ctx_state_full(self.h, chain, ctx.ctypes.data, C.byref(frac)), "fcu_get_ctx_state_full")
        else:
            self._chk(self.lib.fcu_get_ctx_state(self.h, chain, ctx.ctypes.data, C.byref(frac)), "fcu_get_ctx_state")
        return ctx, int(frac.value)

    def rec_planes(self, chain):
        return [p.cpu().numpy() for p in self._keep[chain][1]]

    def ctu_out(self, chain, ctu_rs_addr):
        buf = self._keep[chain][2][ctu_rs_addr * CTU_OUT_BYTES:(ctu_rs_addr + 1) * CTU_OUT_BYTES].cpu().numpy()
        return ctu_to_dict(CtuOut.from_buffer_copy(buf.tobytes()))

    # -- fork pre-pass (TEncSlice::getOutlierWithDCT)
    def obf_prepass(self, luma):
        """luma: uint8 tensor [n, height, width] on this device (or a numpy array).  Returns (obf int16 tensor
        [n, height/4, width/4], yc float64 array [n, 16], (hist_ms, count_ms))."""
        torch = self.torch
        dev = torch.device("cuda", self.device)
        t = torch.as_tensor(luma) if not torch.is_tensor(luma) else luma
        t = t.to(device=dev, dtype=torch.uint8).contiguous()
        if t.dim() == 2:
            t = t[None]
        n = t.shape[0]
        assert tuple(t.shape[1:]) == (self.height, self.width)
        obf = torch.zeros((n, self.height // 4, self.width // 4), dtype=torch.int16, device=dev)
        yc = np.zeros((n, 16), np.float64)
        ms = (C.c_float * 2)()
        self._chk(self.lib.fcu_obf_prepass(self.h, n, t.data_ptr(), obf.data_ptr(), yc.ctypes.data, ms, None), "fcu_obf_prepass")
        return obf, yc, (ms[0], ms[1])

    # -- fork decision hooks of xCompressCU
    def set_decision(self, chain, state, obf=None, sw_skip=(0, 0, 0, 0), sw_term=(0, 0, 0, 0), depth_exception=0):
        """obf: this frame's map from obf_prepass (int16 tensor [height/4, width/4] on this device)."""
        dp = DecisionParams()
        dp.state, dp.depth_exception = state, depth_exception
        for d in range(4):
            dp.sw_skip2nx2n[d], dp.sw_terminate[d] = int(sw_skip[d]), int(sw_term[d])
        if obf is not None:
            assert obf.is_contiguous() and obf.dtype == self.torch.int16 and tuple(obf.shape) == (self.height // 4, self.width // 4)
            dp.dev_obf = obf.data_ptr()
            self._keep_obf[chain] = obf
        self._chk(self.lib.fcu_chain_set_decision(self.h, chain, C.byref(dp)), "fcu_chain_set_decision")

    def verify_counts(self, first, n=1):
        v = VerifyCounts()
        self._chk(self.lib.fcu_get_verify_counts(self.h, first, n, C.byref(v)), "fcu_get_verify_counts")
        return np.array([[v.n[d][k] for k in range(6)] for d in range(4)], np.float64)

    # -- per-PU record of the luma search (BASELINE configs[1])
    def enable_pu_trace(self, chain, trace=None):
        """Binds a device array of n_ctu x 341 fcu_pu_trace records to the chain (several slice chains of a picture may share
        one).  Returns the uint8 tensor; pu_trace_array() turns it into a structured numpy array."""
        torch = self.torch
        if trace is None:
            trace = torch.zeros(self.n_ctu * PUS_PER_CTU * PU_TRACE_DTYPE.itemsize, dtype=torch.uint8, device=torch.device("cuda", self.device))
        self._chk(self.lib.fcu_chain_set_pu_trace(self.h, chain, trace.data_ptr()), "fcu_chain_set_pu_trace")
        self._keep_obf[("pu_trace", chain)] = trace
        return trace

    def pu_trace_array(self, trace):
        return trace.cpu().numpy().view(PU_TRACE_DTYPE).reshape(self.n_ctu, PUS_PER_CTU)

    # -- TComLoopFilter::loopFilterPic
    def deblock(self, chain=None, beta_offset_div2=0, tc_offset_div2=0, timed=False, stream=None, out=None, rec=None):
        """Deblocks, in place, the reconstruction planes bound to `chain` (all slice chains of a picture share them)
        once every CTU of the picture has been decided -- or explicit device tensors: `out` = the picture's
        fcu_ctu_out array as uint8, `rec` = (Y, U, V).  Returns (ms vertical pass, ms horizontal pass) if timed."""
        if chain is not None:
            _, rec, out = self._keep[chain]
        assert out.numel() >= self.n_ctu * CTU_OUT_BYTES and rec[0].numel() == self.width * self.height
        ms = (C.c_float * 2)() if timed else None
        s = C.c_void_p(stream.cuda_stream) if stream is not None else None
        self._chk(self.lib.fcu_deblock(self.h, out.data_ptr(), rec[0].data_ptr(), rec[1].data_ptr(), rec[2].data_ptr(),
                                       beta_offset_div2, tc_offset_div2, ms, s), "fcu_deblock")
        return (ms[0], ms[1]) if timed else None

    # -- TEncSampleAdaptiveOffset::SAOProcess
    def sao(self, pictures, timed=False, stream=None):
        """SAO of completely decided, deblocked pictures, in place on their reconstruction planes.  pictures: list of dicts
        {org: (Y,U,V) device tensors, rec: (Y,U,V) device tensors, qp, lambda_ (the slice's luma lambda), slice_type,
        slice_ctus, enabled (3 ints, default all on), chroma_weight (default: from the QP, chroma QP offset 0)}.  Returns (coded uint8 tensor
        [n, n_ctu, SAO_CTU_BYTES] on the device, off_count int32 array [n, 3], kernel ms x4 or None)."""
        torch = self.torch
        n = len(pictures)
        dev = torch.device("cuda", self.device)
        prm = (SaoParams * n)()
        org, rec = (C.c_void_p * (3 * n))(), (C.c_void_p * (3 * n))()
        for i, p in enumerate(pictures):
            w = p.get("chroma_weight")
            prm[i].slice_type, prm[i].qp, prm[i].slice_ctus = p.get("slice_type", SLICE_I), p["qp"], p.get("slice_ctus", 0)
            lam = p["lambda_"]
            for k in range(3):
                prm[i].enabled[k] = int(p.get("enabled", (1, 1, 1))[k])
                prm[i].lambda_[k] = lam if k == 0 else (lam / w if w else 0.0)
                for t in (p["org"][k], p["rec"][k]):
                    assert t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous() and t.numel() == (self.width >> (1 if k else 0)) * (self.height >> (1 if k else 0))
                org[3 * i + k], rec[3 * i + k] = p["org"][k].data_ptr(), p["rec"][k].data_ptr()
        coded = torch.zeros((n, self.n_ctu, SAO_CTU_BYTES), dtype=torch.uint8, device=dev)
        off = np.zeros((n, 3), np.int32)
        ms = (C.c_float * 4)() if timed else None
        s = C.c_void_p(stream.cuda_stream) if stream is not None else None
        self._chk(self.lib.fcu_sao(self.h, n, prm, org, rec, coded.data_ptr(), off.ctypes.data, ms, s), "fcu_sao")
        return coded, off, (list(ms) if timed else None)

    # -- TEncCu::destroy
    def destroy(self):
        if self.h:
            self.lib.fcu_destroy(self.h)
            self.h = C.c_void_p()
        self._keep = {}
        self._keep_obf = {}

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
