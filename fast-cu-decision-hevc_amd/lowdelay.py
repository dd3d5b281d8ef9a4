"""Host driver of the P-slice path (BASELINE configs[4], lowdelay_P): what TEncGOP::compressGOP does around the CU
decision for a clip whose pictures reference the previous reconstructed picture (TEncGOP.cpp:1096-1160 per picture:
slice parameters -> compressSlice -> loop filter -> reference for the next picture).

Pictures of ONE clip are strictly sequential (every P picture needs the filtered reconstruction of its predecessor), so
the parallelism of a launch comes from the slices of a picture and from independent clips side by side: clip s, slice k
is chain s * n_slices + k.  Per picture: fcu_ldp_slice (QP / lambda of HM's lowdelay_P GOP table) -> fcu_chain_begin
(+ fcu_chain_set_reference for P) -> one fcu_compress_chains launch over all chains -> fcu_deblock -> fcu_sao (SAO 1, the
reference's lowdelay configuration; one batched call for all clips) -> fcu_pad_reference.
Across GPUs the reference picture is the only data a rank would need from another one (one copy per picture);
with whole clips per rank there is none.

TZ search state: HM's encoder carries the integer vector of its last 2Nx2N search (TEncSearch::m_integerMv2Nx2N) across slices
and pictures.  The slice chains of this driver run side by side and start it from zero; the difference shows only for a slice
whose first CTU is too small for a 64x64 CU (fcu_chain_get / set_search_state for callers that need HM's value).

Context-table choice of a P picture (cabac_init_flag): HM initialises a P slice from the B-slice tables when
TEncSbac::determineCabacInitIdx picked them after the previous slice (TEncSlice.cpp:1750-1753).  That choice is made on the
BITSTREAM coder's final state, which includes the SAO syntax and a per-context "bins were coded" flag the RD path does not
carry; this driver therefore runs the configuration `CabacInitPresent 0` (every P picture starts from the P tables,
cabac_b_table = 0) unless the caller names the table per picture (`cabac_b_table` below) -- the TAppEncoder adapter does,
from TComSlice::getEncCABACTableIdx (adapter/TEncCuFcu.cpp).
"""
import numpy as np

from . import engine as _engine

HM_LDP_RPS = {1: (-1, -5, -9, -13), 2: (-1, -2, -6, -10), 3: (-1, -3, -7, -11), 0: (-1, -4, -8, -12)}     # reference pictures of Frame1..Frame4 in encoder_lowdelay_P_main.cfg


def ref_pocs(poc, n_refs, rps="hm"):
    """RefPicList0 of picture `poc` (POCs, nearest first).  rps "hm": the reference picture sets of HM's lowdelay_P cfg --
    the previous picture and the last GOP-boundary pictures -- with the pictures that do not exist yet dropped as the encoder
    does at the start of a sequence (TEncGOP::selectReferencePictureSet + TComSlice::checkThatAllRefPicsAreAvailable);
    rps "recent": the last n_refs pictures."""
    if poc == 0:
        return []
    d = HM_LDP_RPS[poc % 4] if rps == "hm" else tuple(-k for k in range(1, n_refs + 1))
    return [poc + k for k in d if poc + k >= 0][:n_refs]


class LowDelayPDecider:
    """`n_clips` clips of width x height decided picture by picture on one GPU.
    slice_ctus: CTUs per slice (HM SliceMode 1); None = one slice per picture (the reference configuration)."""

    def __init__(self, width, height, base_qp, n_clips=1, search_range=64, slice_ctus=None, deblock=True, sao=False, tmvp=False, fast_search=1, amp=False, device=0,
                 n_refs=1, rps="hm"):
        """n_refs: reference pictures in list 0 (the reference cfg's num_ref_idx_active is 4; 1 = the previous picture only);
        rps: which pictures those are (ref_pocs above)."""
        self.width, self.height, self.base_qp, self.n_clips, self.search_range = width, height, base_qp, n_clips, search_range
        n_ctu = ((width + 63) // 64) * ((height + 63) // 64)
        self.slice_ctus = slice_ctus if slice_ctus else n_ctu
        self.n_slices = (n_ctu + self.slice_ctus - 1) // self.slice_ctus
        self.eng = _engine.CuEngine(width, height, max_chains=n_clips * self.n_slices, device=device)
        self.do_deblock = deblock
        self.tmvp, self.fast_search = tmvp, fast_search    # TMVPMode / FastSearch of the reference cfg (TZ search by default)
        self.amp = amp                                   # AMP of the reference cfg (asymmetric motion partitions)
        self.col = [None] * n_clips                      # fcu_ctu_out array of each clip's previous picture (TMVP motion field)
        self.do_sao = sao                                # SAO 1 of the reference's cfg; off by default: the loop-filter goldens stop at deblocking
        self.sao_rate = [_engine.SaoRate() for _ in range(n_clips)]      # m_saoDisabledRate per clip
        self.poc = 0
        self.ref = [None] * n_clips                      # padded reference planes per clip
        assert 1 <= n_refs <= _engine.MAX_REF and rps in ("hm", "recent")
        self.n_refs, self.rps = n_refs, rps
        self.dpb = [dict() for _ in range(n_clips)]      # per clip: poc -> (padded planes, the POCs its list 0 named); pictures a later one may reference

    def frame_params(self, poc, cabac_b_table=0):
        fp = _engine.ldp_slice(self.base_qp, poc)
        fp.cabac_b_table = 1 if (cabac_b_table and poc > 0) else 0
        fp.search_range = self.search_range
        fp.fast_search = self.fast_search
        fp.tmvp = 1 if (self.tmvp and poc > 0) else 0
        fp.amp = 1 if self.amp else 0
        return fp

    def decide_picture(self, frames, cabac_b_table=0):
        """frames: one (Y, U, V) per clip for picture self.poc; cabac_b_table: 1 = this P picture starts from the B-slice
        context tables (module docstring).  Returns per clip a dict: poc, slice_type, qp, `out` (the
        fcu_ctu_out array as a uint8 device tensor), `rec` (device planes after the loop filters that are enabled), `rec_unfiltered`
        (a copy before the loop filters, for parity checks), `sao` (the picture's fcu_sao_ctu array when SAO is on)."""
        eng, poc = self.eng, self.poc
        assert len(frames) == self.n_clips
        fp = self.frame_params(poc, cabac_b_table)
        res = []
        rl = ref_pocs(poc, self.n_refs, self.rps)
        for s, f in enumerate(frames):
            first = s * self.n_slices
            ref = self.ref[s] if fp.slice_type == _engine.SLICE_P else None
            col = self.col[s] if fp.tmvp else None
            kw = dict(ref=ref)
            if ref is not None and self.n_refs > 1:
                kw = dict(refs=[self.dpb[s][q][0] for q in rl], ref_pocs=rl, poc=poc, col_ref_pocs=self.dpb[s][rl[0]][1])
            rec, out = eng.init_chain(first, f, fp.qp, slice_ctus=self.slice_ctus if self.n_slices > 1 else 0, params=fp, col=col, **kw)
            planes = eng._keep[first][0]
            for k in range(self.n_slices):
                if k:
                    eng.init_chain(first + k, planes, fp.qp, slice_ctus=self.slice_ctus, rec=rec, out=out, params=fp, col=col, **kw)
                if self.n_slices > 1:
                    a = k * self.slice_ctus
                    eng.set_range(first + k, a, min(self.slice_ctus, eng.n_ctu - a))
            res.append({"poc": poc, "slice_type": fp.slice_type, "qp": fp.qp, "lambda": fp.lambda_, "out": out, "rec": rec, "first": first})
        eng.compress_chains(0, self.n_clips * self.n_slices, self.slice_ctus)
        for s, r in enumerate(res):
            r["rec_unfiltered"] = [p.clone() for p in r["rec"]]
            if self.do_deblock:
                eng.deblock(r["first"])
        if self.do_sao:
            layer = _engine.ldp_layer(poc)
            pics = [{"org": eng._keep[r["first"]][0], "rec": r["rec"], "qp": fp.qp, "lambda_": fp.lambda_, "slice_type": fp.slice_type,
                     "slice_ctus": self.slice_ctus if self.n_slices > 1 else 0, "enabled": self.sao_rate[s].enabled(layer)} for s, r in enumerate(res)]
            coded, off, _ = eng.sao(pics)
            for s, r in enumerate(res):
                r["sao"], r["sao_enabled"] = coded[s], pics[s]["enabled"]
                self.sao_rate[s].update(layer, off[s], eng.n_ctu)
        for s, r in enumerate(res):
            self.col[s] = r["out"]                           # stays in HBM: the next picture's collocated motion field
            self.ref[s] = eng.pad_reference(r["rec"])          # reference of the next picture of this clip
            if self.n_refs > 1:
                r["ref_pocs"] = rl
                self.dpb[s][poc] = (self.ref[s], rl)
                for q in [q for q in self.dpb[s] if not any(q in ref_pocs(t, self.n_refs, self.rps) for t in range(poc + 1, poc + 17))]:
                    del self.dpb[s][q]                           # no later picture names it: its planes go back to the allocator
        eng.sync()
        self.poc += 1
        return res

    def close(self):
        self.eng.destroy()
