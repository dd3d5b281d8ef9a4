"""Picture-level host driver around the engine: what TEncTop / TEncGOP / TEncSlice do around `TEncCu` for an
all-intra sequence, reduced to the calls of include/fcu.h.

  FastDecisionSchedule   the fork's per-picture control of its fast CU decision: state of the picture
                         (getCurrentState, tools_YS.cpp:1237-1242), reset at the start of a period
                         (TEncTop.cpp:393-398 -> resetYSGlobal, tools_YS.cpp:89-110), switches from the Verifying
                         pictures' counters after the last of them (TEncTop.cpp:548-552 -> ReportPerformance ->
                         SetDecisionSwitch, tools_YS.cpp:1123-1154).  Pure host logic, engine-agnostic.
  SequenceDecider        per picture: OBF pre-pass (TEncGOP.cpp:1096), one chain per slice (TEncGOP.cpp:1102-1138,
                         SliceMode 1), the decisions, the in-loop deblocking (TEncGOP.cpp:1155-1160).
  read_yuv420 / write_yuv420   planar 8-bit 4:2:0 files as TVideoIOYuv reads / writes them for fileBitDepth 8
                         (TLibVideoIO/TVideoIOYuv.cpp:247-400 readPlane, :402-560 writePlane, :680, :767).

torch is plumbing (device buffers); there is no CPU fallback for the decisions.
"""
import numpy as np

from . import engine as _engine

TRAINING, VERIFYING, TESTING = _engine.TRAINING, _engine.VERIFYING, _engine.TESTING


class FastDecisionSchedule:
    """The reference keeps this state in globals (g_iPOC, g_iP/g_iT/g_iV, g_iVerResult, g_bDecisionSwitch); here the
    caller owns it.  `decision_switch` / `frame_state` default to the host functions of libfcu.so."""

    def __init__(self, period=60, n_training=2, n_verifying=1, th_skip=(0, 0, 0, 0), th_term=(0, 0, 0, 0),
                 decision_switch=None, frame_state=None):
        self.period, self.n_training, self.n_verifying = period, n_training, n_verifying
        self.th_skip, self.th_term = th_skip, th_term
        self._switch = decision_switch or _engine.decision_switch
        self._state = frame_state or _engine.frame_state
        self.sw_skip = np.zeros(4, np.uint8)
        self.sw_term = np.zeros(4, np.uint8)
        self.ver = np.zeros((4, 6), np.float64)

    def begin_picture(self, poc):
        """State and switches picture `poc` is decided with."""
        if poc % self.period == 0:                        # resetYSGlobal: counters and switches start again
            self.sw_skip[:] = 0
            self.sw_term[:] = 0
            self.ver[:] = 0
        return self._state(poc, self.period, self.n_training, self.n_verifying), self.sw_skip.copy(), self.sw_term.copy()

    def end_picture(self, poc, verify_counts=None):
        """After the picture is decided: Verifying pictures add their counters; the last one sets the switches."""
        r = poc % self.period
        if self.n_training <= r < self.n_training + self.n_verifying:
            if verify_counts is None:
                raise ValueError("a Verifying picture must report its counters")
            self.ver += np.asarray(verify_counts, np.float64).reshape(4, 6)
        if r == self.n_training + self.n_verifying - 1:
            sk, te = self._switch(self.ver, self.th_skip, self.th_term)
            self.sw_skip[:] = sk
            self.sw_term[:] = te


def read_yuv420(path, width, height, frame):
    """Frame `frame` of a planar 8-bit 4:2:0 file -> (Y, U, V) uint8 arrays; None past the end of the file."""
    ysz, csz = width * height, (width // 2) * (height // 2)
    with open(path, "rb") as f:
        f.seek(frame * (ysz + 2 * csz))
        buf = f.read(ysz + 2 * csz)
    if len(buf) < ysz + 2 * csz:
        return None
    a = np.frombuffer(buf, np.uint8)
    return (a[:ysz].reshape(height, width).copy(), a[ysz:ysz + csz].reshape(height // 2, width // 2).copy(),
            a[ysz + csz:].reshape(height // 2, width // 2).copy())


def write_yuv420(f, planes):
    """Appends one frame (Y, U, V uint8 arrays) to the binary file object `f`."""
    for p in planes:
        f.write(np.ascontiguousarray(p, np.uint8).tobytes())


class SequenceDecider:
    """All-intra sequence, picture by picture, on one GPU.  `fast=False` keeps every picture in the Training state
    (plain HM RDO); `fast=True` runs the fork's Training / Verifying / Testing cycle with its default (Naive) control."""

    def __init__(self, width, height, qp, slice_ctus=None, fast=True, deblock=True, device=0, schedule=None, in_flight=1, **flags):
        """slice_ctus: CTUs per slice (HM's SliceMode 1 / SliceArgument).  None = one slice per picture, which is the
        reference's default (SliceMode 0, TAppEncCfg.cpp:838) and what `encoder_intra_main.cfg` encodes; a smaller value
        (e.g. the picture width in CTUs for one slice per CTU row) is a DIFFERENT encoder configuration -- the slices then
        run as concurrent chains, but every slice start cuts the intra neighbourhood and resets CABAC."""
        self.width, self.height, self.qp, self.fast, self.do_deblock, self.flags = width, height, qp, fast, deblock, flags
        self.in_flight = max(1, in_flight)
        w_ctu = (width + 63) // 64
        n_ctu = w_ctu * ((height + 63) // 64)
        self.slice_ctus = slice_ctus if slice_ctus else n_ctu
        self.slice_mode = "SliceMode 0 (one slice per picture)" if self.slice_ctus >= n_ctu else f"SliceMode 1, SliceArgument {self.slice_ctus}"
        self.n_slices = (n_ctu + self.slice_ctus - 1) // self.slice_ctus
        self.eng = _engine.CuEngine(width, height, max_chains=self.n_slices * self.in_flight, device=device)
        self.schedule = schedule or FastDecisionSchedule()
        self.poc = 0

    def decide(self, yuv):
        """Decides one picture.  Returns a dict: poc, state, switches, `out` (the fcu_ctu_out array as a uint8 device
        tensor), `rec` (device planes, deblocked when enabled), `depth` ([n_ctu, 256] numpy, z-order), verify counters."""
        return self.decide_group([yuv])[0]

    def group_size(self):
        """How many of the next pictures may be decided side by side: they share their state, and no switch that one
        of them reads is set by another (the switches change only after the last Verifying picture of a period)."""
        if not self.fast:
            return self.in_flight
        sc, r = self.schedule, self.poc % self.schedule.period
        end = sc.n_training if r < sc.n_training else (sc.n_training + sc.n_verifying if r < sc.n_training + sc.n_verifying else sc.period)
        return max(1, min(self.in_flight, end - r))

    def decide_group(self, yuvs):
        """Decides len(yuvs) <= group_size() consecutive pictures in one launch: picture i owns the chains
        [i * n_slices, (i + 1) * n_slices).  Returns one dict per picture (see decide)."""
        eng = self.eng
        assert 1 <= len(yuvs) <= self.group_size()
        pics = []
        for i, yuv in enumerate(yuvs):
            poc = self.poc + i
            state, sk, te = self.schedule.begin_picture(poc) if self.fast else (TRAINING, np.zeros(4, np.uint8), np.zeros(4, np.uint8))
            first = i * self.n_slices
            n_sl, rec, out = eng.init_slice_chains(first, yuv, self.qp, self.slice_ctus, **self.flags)
            if state != TRAINING:
                obf = eng.obf_prepass(eng._keep[first][0][0])[0][0].contiguous()
                for k in range(n_sl):
                    eng.set_decision(first + k, state, obf, sk, te)
            pics.append({"poc": poc, "state": state, "sw_skip": sk, "sw_term": te, "out": out, "rec": rec, "first": first})
        eng.compress_chains(0, len(yuvs) * self.n_slices, self.slice_ctus)
        nb = _engine.CTU_OUT_BYTES
        for p in pics:
            p["verify"] = eng.verify_counts(p["first"], self.n_slices) if p["state"] == VERIFYING else None
            if self.fast:
                self.schedule.end_picture(p["poc"], p["verify"])
            if self.do_deblock:
                eng.deblock(p["first"])
        eng.sync()
        for p in pics:
            p["depth"] = p["out"].view(eng.n_ctu, nb)[:, :256].cpu().numpy().copy()      # fcu_ctu_out.depth leads the struct
            p["tu_trials"] = sum(eng.debug_counters(p["first"] + k)[16] for k in range(self.n_slices))
        self.poc += len(yuvs)
        return pics

    def close(self):
        self.eng.destroy()
