/*
 * fcu_engine.h -- the CTU decision engine: one 64-lane wavefront per chain.
 *
 * A chain is the unit that HM processes strictly sequentially (a slice of an I picture:
 * reconstructed neighbours, CABAC contexts and neighbour decisions all flow CTU to CTU,
 * SURVEY.md 7.1).  Chains are independent, so the GPU runs thousands of them side by
 * side, one wavefront each; inside a chain the wave
 *   - spreads pixel work (35-mode prediction, Hadamard SATD, DCT/DST, dequant, recon, SSE)
 *     across its 64 lanes,
 *   - evaluates the independent RDO candidates of a PU (each restarts from the same CABAC
 *     snapshot, TEncSearch.cpp:2457) on different lanes: RDOQ and bit counting are
 *     lane-private serial code,
 *   - keeps CABAC snapshots, reference samples and the SATD staging buffer in LDS.
 *
 * Code style: "phases".  FCU_FOR_LANES { ... } is per-lane code followed by a workgroup
 * barrier; everything outside such a block is wave-uniform, reads memory only, and never
 * stores.  All cross-lane communication goes through memory (LDS / the chain's global
 * scratch).  The same source therefore also builds as a plain C++ wave emulator
 * (-DFCU_EMU, tests only) used to check the engine's logic where no GPU is present.
 *
 * Every routine cites the reference code it replaces (paths relative to the reference tree).
 */
#pragma once
#include <stdint.h>
#include <string.h>
#include <type_traits>
#include "../../include/fcu.h"

#ifdef FCU_EMU
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#define FCU_DEV static inline
#define FCU_MEMBER inline
#define FCU_NOINLINE
#define FCU_INLINE
#define FCU_TABLE static const
/* The emulator runs the 64 lanes of a phase one after the other.  FCU_UNI(x) claims "x is the same in every lane":
 * on the device it is a readfirstlane that silently takes lane 0's value, so a false claim corrupts the other lanes.
 * The emulator checks the claim: inside a phase every lane that reaches the same FCU_UNI site must bring the value the
 * first lane brought (tests/emu; the device build cannot check this). */
static int g_emu_phase = 0, g_emu_in_phase = 0, g_emu_lane = 0, g_emu_relax = 0;
/* a lane-private serial walk whose calls are data dependent per lane (the lanes that do call are in lockstep on the
 * device, but the emulator cannot line their calls up): the check is suspended inside */
#define FCU_EMU_RELAX(on) (g_emu_relax = (on))
static inline int fcu_emu_phase_begin() { g_emu_phase++; g_emu_in_phase = 1; g_emu_lane = 0; return 0; }
static inline int fcu_emu_phase_next(int lane) { if (lane + 1 >= 64) g_emu_in_phase = 0; g_emu_lane = lane + 1; return lane + 1; }
/* the k-th visit of a site by a lane must bring what the k-th visit by the first visiting lane brought (lanes run in
 * lockstep on the device; a lane may visit a site more often than the first one did: different trip counts) */
template <class T> static inline T fcu_emu_uni(T v, int site, int line)
{
  enum { SITES = 2048, K = 32, B = 16 };
  struct Seen { int phase, first, cur, idx, n; unsigned char vals[K][B]; };
  static Seen seen[SITES];
  if (g_emu_in_phase && !g_emu_relax && sizeof(T) <= B && site < SITES) {
    Seen &s = seen[site];
    if (s.phase != g_emu_phase) { s.phase = g_emu_phase; s.first = s.cur = g_emu_lane; s.idx = 0; s.n = 0; }
    if (s.cur != g_emu_lane) { s.cur = g_emu_lane; s.idx = 0; }
    if (g_emu_lane == s.first) { if (s.n < K) memcpy(s.vals[s.n++], &v, sizeof(T)); }
    else if (s.idx < s.n && memcmp(s.vals[s.idx], &v, sizeof(T)) != 0) { fprintf(stderr, "FCU_UNI at line %d: value differs between lanes of one phase\n", line); abort(); }
    s.idx++;
  }
  return v;
}
#define FCU_FOR_LANES for (int lane = fcu_emu_phase_begin(); lane < 64; lane = fcu_emu_phase_next(lane))
#define FCU_ATOMIC_ADD(p, v) (*(p) += (v))
#define FCU_ATOMIC_MAX(p, v) do { if (*(p) < (v)) *(p) = (v); } while (0)
#define FCU_ATOMIC_OR(p, v) (*(p) |= (v))
/* wave reductions of per-lane partial results into one LDS word (all 64 lanes call them, outside divergent code) */
#define FCU_WAVE_ADD(p, v) (*(p) += (v))
#define FCU_WAVE_MIN64(p, v) do { if ((v) < *(p)) *(p) = (v); } while (0)
#define FCU_IN_LDS(p) do { } while (0)
#define FCU_UNI(x) fcu_emu_uni((x), __COUNTER__, __LINE__)
#define FCU_HBM
#define FCU_FLOOR(x) floor(x)
#define FCU_CHECK(c) do { if (!(c)) { fprintf(stderr, "FCU_CHECK failed: %s (line %d)\n", #c, __LINE__); abort(); } } while (0)
#else
#define FCU_DEV __device__ static
#define FCU_MEMBER __device__ inline
#define FCU_NOINLINE __noinline__
#define FCU_INLINE __attribute__((always_inline))
#define FCU_TABLE __device__ static const
#define FCU_FOR_LANES for (int lane = (int)threadIdx.x, fcu_once_ = 1; fcu_once_; fcu_once_ = 0, __syncthreads())
#define FCU_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#define FCU_ATOMIC_MAX(p, v) atomicMax((p), (v))
#define FCU_ATOMIC_OR(p, v) atomicOr((p), (v))
/* wave reductions in registers, then ONE lane touches LDS -- the SAD / SSE / Hadamard partial sums (north star: "wavefront
 * reductions for per-PU costs").  The sum runs on the DPP path of the VALU (six v_add_u32_dpp: two quad permutes, two row
 * rotations, row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3; lane 63 then holds the total) instead of
 * six ds_bpermute round trips through the LDS crossbar (__shfl_xor).  All 64 lanes must be active. */
__device__ inline uint32_t fcu_wave_sum(uint32_t v)
{
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xb1, 0xf, 0xf, false);    /* quad_perm:[1,0,3,2] */
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4e, 0xf, 0xf, false);    /* quad_perm:[2,3,0,1] */
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x124, 0xf, 0xf, false);   /* row_ror:4 */
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xf, 0xf, false);   /* row_ror:8: every lane of a row holds the row's sum */
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   /* row_bcast:15 -> rows 1, 3 */
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   /* row_bcast:31 -> rows 2, 3 */
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ inline unsigned long long fcu_wave_min64(unsigned long long v)
{
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, o), hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), o);
    const unsigned long long w = ((unsigned long long)hi << 32) | lo;
    v = w < v ? w : v;
  }
  return v;
}
#define FCU_WAVE_ADD(p, v) do { const uint32_t s_ = fcu_wave_sum(v); if (threadIdx.x == 0) *(p) += s_; } while (0)
#define FCU_WAVE_MIN64(p, v) do { const unsigned long long s_ = fcu_wave_min64(v); if (threadIdx.x == 0 && s_ < *(p)) *(p) = s_; } while (0)
/* address-space fact for a pointer that crossed a call boundary (lets the compiler emit ds_ instead of flat_ accesses) */
/* wave-uniform value -> scalar registers.  Everything the orchestration code passes around (chain/scratch/CU
 * pointers, TU descriptors, depths) is the same in all lanes; saying so keeps it in SGPRs, turns the control
 * flow into scalar branches and keeps such values out of VGPR spill code around calls. */
template <class T> __device__ inline T fcu_uni(T v)
{
  static_assert(sizeof(T) % 4 == 0, "dword-sized objects only");
  int w[sizeof(T) / 4];
  __builtin_memcpy(w, &v, sizeof(T));
  for (unsigned i = 0; i < sizeof(T) / 4; i++) w[i] = __builtin_amdgcn_readfirstlane(w[i]);
  __builtin_memcpy(&v, w, sizeof(T));
  return v;
}
#define FCU_UNI(x) fcu_uni(x)
#define FCU_EMU_RELAX(on) do { } while (0)
/* pointer-to-HBM type qualifier: accesses become global_ instead of flat_ instructions, which (unlike flat) may stay
 * in flight across a partial s_waitcnt -- needed for the one-ahead loads of the serial coefficient loops */
#if defined(__HIP_DEVICE_COMPILE__)
#define FCU_HBM __attribute__((address_space(1)))
#else
#define FCU_HBM
#endif
#define FCU_GENERIC_(p) ((const __attribute__((address_space(0))) void *)(p))
#define FCU_IN_LDS(p) __builtin_assume(__builtin_amdgcn_is_shared(FCU_GENERIC_(p)))
#define FCU_FLOOR(x) floor(x)
#define FCU_CHECK(c) do { } while (0)
#endif
/* Squared error of one sample into the distortion word of its candidate `v` (samples i of candidates of n2 samples each, dealt
 * 64 per iteration).  Blocks of 64 samples or more: the 64 lanes of an iteration share the candidate, so each lane sums in a
 * register and the wave reduces once per candidate; 4x4 blocks (four candidates per iteration) keep the LDS atomic. */
#define FCU_DIST_ADD(acc, v, e, i, n2) do { \
    if ((n2) >= 64) { (acc) += (uint32_t)((e) * (e)); if (((((i) - lane) + 64) & ((n2) - 1)) == 0) { FCU_WAVE_ADD(&g_S.vc_dist[v], acc); (acc) = 0; } } \
    else FCU_ATOMIC_ADD(&g_S.vc_dist[v], (uint32_t)((e) * (e))); } while (0)
#define FCU_SERIAL FCU_FOR_LANES if (lane == 0)
/* section timers (diagnostic build only: -DFCU_PROFILE; shader-clock ticks summed per chain by lane 0) */
#if defined(FCU_PROFILE) && !defined(FCU_EMU)
#define FCU_TIC(v) const long long v = clock64()
#ifdef FCU_PROFILE_RQT   /* inter residual quadtree variant: slots 0..8 belong to the stages of inter_tu_trials / est_inter_residual_qt (FCU_QTOC); 10 stays the CTU total */
#define FCU_QTIC(v) long long v = clock64()
#define FCU_QTOC(E_, v, idx) do { if (threadIdx.x == 0) (E_).C->prof[idx] += (unsigned long long)(clock64() - v); v = clock64(); } while (0)
#define FCU_TOC(E_, v, idx) do { if ((idx) == 10 && threadIdx.x == 0) (E_).C->prof[idx] += (unsigned long long)(clock64() - v); } while (0)
#define FCU_COUNT(E_, idx, n) do { } while (0)
#elif defined(FCU_PROFILE_INTER)  /* P-path variant: slots 0..9 and 11..14 belong to the sections of compress_cu's P branch / pred_inter_search (FCU_ITOC); 10 stays the CTU total */
#define FCU_ITOC(E_, v, idx) do { if (threadIdx.x == 0) (E_).C->prof[idx] += (unsigned long long)(clock64() - v); } while (0)
#define FCU_TOC(E_, v, idx) do { if ((idx) == 10 && threadIdx.x == 0) (E_).C->prof[idx] += (unsigned long long)(clock64() - v); } while (0)
#define FCU_COUNT(E_, idx, n) do { } while (0)
#elif defined(FCU_PROFILE_RDOQ) || defined(FCU_PROFILE_DEPTH)   /* slots 11..15 belong to the sub-timers inside the serial RDOQ (FCU_PROFILE_DEPTH: 11..14 to the time spent at CU depth 0..3 without its sub-CUs) in these variants */
#define FCU_TOC(E_, v, idx) do { if ((idx) < 11 && threadIdx.x == 0) (E_).C->prof[idx] += (unsigned long long)(clock64() - v); } while (0)
#define FCU_COUNT(E_, idx, n) do { } while (0)
#else
#define FCU_TOC(E_, v, idx) do { if (threadIdx.x == 0) (E_).C->prof[idx] += (unsigned long long)(clock64() - v); } while (0)
#define FCU_COUNT(E_, idx, n) do { (E_).C->prof[idx] += (unsigned long long)(n); } while (0)
#endif
#else
#define FCU_TIC(v) do { } while (0)
#define FCU_TOC(E_, v, idx) do { } while (0)
#define FCU_COUNT(E_, idx, n) do { } while (0)
#endif
#ifndef FCU_QTOC
#define FCU_QTIC(v) do { } while (0)
#define FCU_QTOC(E_, v, idx) do { } while (0)
#endif
#ifndef FCU_ITOC
#define FCU_ITOC(E_, v, idx) do { } while (0)
#endif

#include "fcu_tables.h"

/* hot tables mirrored in LDS (the per-bin cost/transition table is read once per context-coded bin and several
 * times per RDOQ coefficient; on-chip reads keep the vector-memory pipeline for the scratch traffic) */
struct HotTables { uint32_t bin[256]; uint16_t scan[3][16]; uint8_t scan_cg8[3][4]; uint8_t ctx_ind_map4x4[16]; uint8_t group_idx[32];
                   uint64_t scan4_nib[3], map4_nib; uint32_t cnt_bits[4]; };   /* 4x4 scans / 4x4 sig-ctx map as nibbles, sig-ctx counts per pattern as 2-bit fields */
#ifdef FCU_EMU
static HotTables g_hot;
#else
__shared__ HotTables g_hot;
#endif

#if defined(FCU_PROFILE_RQT) && !defined(FCU_EMU)     /* lane 0 of the lane-private RDOQ (the luma variant of an inter TU): slots 11..15 */
#define FCU_RTIC(v) long long v = clock64()
#define FCU_RTOC(P_, v, idx) do { if (!SER && threadIdx.x == 0) { ((Chain *)((char *)&(P_) - __builtin_offsetof(Chain, p)))->prof[idx] += (unsigned long long)(clock64() - v); v = clock64(); } } while (0)
#elif defined(FCU_PROFILE_RDOQ) && !defined(FCU_EMU)
#define FCU_RTIC(v) long long v = clock64()
#define FCU_RTOC(P_, v, idx) do { if (SER) { ((Chain *)((char *)&(P_) - __builtin_offsetof(Chain, p)))->prof[idx] += (unsigned long long)(clock64() - v); v = clock64(); } } while (0)
#else
#define FCU_RTIC(v) do { } while (0)
#define FCU_RTOC(P_, v, idx) do { } while (0)
#endif
namespace fcu {

/* ---- constants ------------------------------------------------------------------------ */
enum { CTU = 64, MAXDEPTH = 3, NPART = 256, LOG2_MAXTU = 5, LOG2_MINTU = 2, TU_MAXDEPTH_INTRA = 3 };
enum { SIZE_2Nx2N = 0, SIZE_2NxN = 1, SIZE_Nx2N = 2, SIZE_NxN = 3, SIZE_2NxnU = 4, SIZE_2NxnD = 5, SIZE_nLx2N = 6, SIZE_nRx2N = 7, SIZE_NONE = 8, MODE_INTER = 0, MODE_INTRA = 1, MODE_NONE = 2 };
enum { SLICE_I = 0, SLICE_P = 1 };
enum { PLANAR = 0, DC = 1, HOR = 10, VER = 26, DM_CHROMA = 36 };
/* context layout (counts: TLibCommon/ContextTables.h:49-161) */
enum { CTX_SPLIT = 0, CTX_PARTSIZE = 3, CTX_INTRA_LUMA = 4, CTX_CHROMA_PRED = 5, CTX_CBF_LUMA = 6, CTX_CBF_CHROMA = 11,
       CTX_SUBDIV = 16, CTX_SIGCG = 19, CTX_SIG = 23, CTX_LASTX = 67, CTX_LASTY = 97, CTX_ONE = 127, CTX_ABS = 151,
       CTX_TSKIP = 157, NCTX_INTRA = 160,
       /* inter syntax of P slices, appended: skip flag (3), merge flag, merge index, pred mode, part-size contexts 1..3,
        * mvd (2), ref idx (2), mvp idx, rqt root cbf, pad */
       CTX_SKIP = 160, CTX_MERGE_FLAG = 163, CTX_MERGE_IDX = 164, CTX_PRED_MODE = 165, CTX_PARTSIZE1 = 166, CTX_MVD = 169,
       CTX_REF = 171, CTX_MVP_IDX = 173, CTX_ROOT_CBF = 174, NCTX = 176 };
enum { CI_CURR_BEST = 0, CI_NEXT_BEST, CI_TEMP_BEST, CI_CHROMA_INTRA, CI_QT_TRAFO_TEST, CI_QT_TRAFO_ROOT, CI_NUM };
enum { MEMO_K = 6, MEMO_POOL = 6144 + 1536 + 384 + 96 };    /* slots per depth; levels (or residual samples) of one slot of every depth: 1.5 * (64 >> d)^2 */
enum { MAXVC = 20, MAXLC = 16, POOL = 5120 };   /* MAXVC candidate variants of a batch, MAXLC lane-private coders (the bit count runs in rounds) */
/* <= 8 RMD survivors + 2 MPMs (iMode, TEncSearch.cpp:2407-2428), x2 transform-skip variants; 5 x 32x32 */
#define FCU_MAX_DOUBLE 1.7e+308

/* coder state copied by TEncSbac::load/store (TEncSbac.cpp:397-426) */
struct Cabac { uint8_t ctx[NCTX]; uint64_t frac; uint32_t bins; uint32_t pad_; };

struct Params {
  int width, height, qp, qp_c, slice_ctus, slice_type;
  int transform_skip, ts_fast, sign_hiding, strong_smoothing;
  double lambda, sqrt_lambda, chroma_weight, rdoq_lambda[3];
  double err_scale[2][4];     /* [luma/chroma][log2-2]   setErrScaleCoeff, TComTrQuant.cpp:3018-3040 */
  long long rd_factor[2];     /* sign-hiding rdFactor,   TComTrQuant.cpp:2444-2447 */
  /* P slices (BASELINE configs[4]) */
  int search_range, fast_enc, had_me, fdm, max_merge_cand, fast_search;
  int tmvp;                    /* TMVPMode: temporal merge / AMVP candidate from Chain::col */
  int amp;                     /* asymmetric motion partitions (AMP with AMP_ENC_SPEEDUP + AMP_MRG, TEncCu.cpp:381-450,836-943) */
  int rdoq, rdoq_ts;           /* RDOQ / RDOQTS: 0 = the plain quantiser of xQuant (quant_plain) for blocks without / with transform skip */
  uint32_t lambda_motion_sad;  /* m_uiLambdaMotionSAD = floor(65536 * sqrt(lambda)), TComRdCost.cpp:194-219 */
  int cabac_b_table;           /* P slice initialised from the B-slice context tables (encCABACTableIdx == B_SLICE, TEncSbac.cpp:111-115) */
};

/* per-chain descriptor in HBM */
struct Chain {
  const uint8_t *org[3];
  uint8_t *rec[3];
  int stride[3];
  /* P slice: reference picture (list 0, index 0), padded planes: ref[c] points at sample (0,0), the border is
   * replicated FCU_REF_MARGIN (>> 1 for chroma) samples to every side (TComPicYuv::extendPicBorder) */
  const uint8_t *ref[3];
  int ref_stride[3];
  fcu_ctu_out *out;
  Params p;
  int w_ctu, h_ctu, n_ctu;
  int next_ctu, end_ctu;       /* the chain decides CTUs [next_ctu, end_ctu): the whole frame, or whole slices of it */
  Cabac state;                 /* m_pppcRDSbacCoder[0][CI_CURR_BEST] between CTUs */
  unsigned long long n_tu_trials;
  unsigned long long prof[16];
  /* fork decision state (tools_YS.cpp): frame state, per-depth switches of the Naive model, OBF count map, g_iVerResult */
  int dec_state, depth_exception, obf_stride;
  uint8_t sw_skip[4], sw_term[4];
  const int16_t *obf;
  double ver[4][6];
  const fcu_ctu_out *col;      /* TMVP: the reference picture's fcu_ctu_out array (its motion field), or null */
  int int_mv[2];               /* (kept for layout; the TZ start vectors live in int_mv_r below) */
  fcu_pu_trace *pu_trace;      /* optional [n_ctu][FCU_PUS_PER_CTU] record of the luma search (fcu_chain_set_pu_trace) */
  /* reference picture list 0 of a P slice: refs[r] = padded planes of RefPicList0[r] (refs[0] == ref; all share ref_stride),
   * ref_poc[r] its POC, poc the current picture's, n_ref = num_ref_idx_l0_active (1..FCU_MAX_REF).  col_poc / col_ref_poc: the
   * collocated picture (= RefPicList0[0]) and the POCs its own list 0 named, for the TMVP scaling (xGetColMVP).
   * int_mv_r[r] = m_integerMv2Nx2N[list 0][r]: integer vector of the chain's last 2Nx2N motion search on that reference. */
  const uint8_t *refs[FCU_MAX_REF][3];
  int n_ref, poc, ref_poc[FCU_MAX_REF], col_poc, col_ref_poc[FCU_MAX_REF];
  int int_mv_r[FCU_MAX_REF][2];
};
enum { DEC_TRAINING = 0, DEC_VERIFYING = 1, DEC_TESTING = 2 };

/* per-depth working CU (TComDataCU best/temp objects, TEncCu.cpp:163-198) */
struct CuObj {
  double cost; uint32_t dist, bits, bins;
  int depth_cu, x, y, zidx, nparts;
  uint8_t depth[NPART]; int8_t part_size[NPART], pred_mode[NPART]; uint8_t tr_idx[NPART];
  uint8_t tskip[3][NPART], cbf[3][NPART], intra_dir[2][NPART];
  uint8_t skip[NPART], merge_flag[NPART], merge_idx[NPART], inter_dir[NPART]; int8_t mvp_idx[NPART], ref_idx[NPART];
  int16_t mv[NPART][2], mvd[NPART][2];
  alignas(16) int16_t coef[3][CTU * CTU];
};
struct Yuv16 { int16_t y[64 * 64], u[32 * 32], v[32 * 32]; };
struct Yuv { uint8_t y[64 * 64], u[32 * 32], v[32 * 32]; };
enum { FCU_REF_MARGIN = 80 };                     /* luma border of a reference picture: g_uiMaxCUWidth + 16 (TComPic::create) */
/* state of one chroma-mode trial: the mode's own reconstruction (overlay of PicYuvRec inside the CU), levels, flags */
struct ChromaModeBuf { uint8_t u[32 * 32], v[32 * 32]; alignas(16) int16_t coef[2][1024]; uint8_t cbf[2][NPART], tskip[2][NPART]; };

/* RDOQ's per-coefficient locals as one record (TComTrQuant.cpp:2082-2095) */
struct RdoqRec { double cc, cs, c0; int32_t up, dn, sd, du; };   /* pdCostCoeff, pdCostSig, pdCostCoeff0, rateIncUp/Down, sigRateDelta, deltaU */

/* per-chain scratch in HBM (L2 resident working set) */
struct alignas(16) Scratch {
  CuObj cu[4][2];
  Yuv org[4], predt[4], reco[4][2];
  Yuv qt_rec[4];                                   /* m_pcQTTempTComYuv[layer] */
  alignas(16) int16_t qt_coef[3][4][CTU * CTU];                /* m_ppcQTTempCoeff[comp][layer] */
  int16_t ts_coef[3][1024]; Yuv ts_rec; uint8_t shared_pred[3][1024];
  uint8_t tmp_tr_idx[NPART], tmp_cbf[NPART], tmp_tskip[NPART];
  ChromaModeBuf cm[5];
  Cabac slots[MAXDEPTH + 2][CI_NUM];               /* the colder snapshots (NEXT/TEMP_BEST, QT_TRAFO_*) live in L2 */
  /* candidate pools: slot v occupies [v*N*N, (v+1)*N*N) */
  uint8_t p_pred[POOL]; int16_t p_resi[POOL]; int32_t p_tmp[POOL]; int32_t p_tcoef[POOL]; uint8_t p_rec[POOL];
  /* scan-order domain of a batch: element (scan position sp, slot v) at [sp * nslots + v] */
  int32_t p_lscan[POOL]; int16_t p_qscan[POOL];
  /* RDOQ locals (TComTrQuant.cpp:2082-2095), same interleaving */
  RdoqRec r_rec[POOL]; double r_cg[MAXVC * 64];
  /* 64x64 first pass: per-candidate reconstruction and levels of the whole PU (candidates advance side by side) */
  uint8_t c64_rec[5][CTU * CTU]; alignas(16) int16_t c64_coef[5][CTU * CTU];
  Cabac c64_state[5][4]; uint32_t c64_distk[5][4];   /* ... and its coder / distortion after each of the four TUs */
  /* inter: residual of the CU, chosen residual, residual per RQT layer (m_pcQTTempTComYuv holds residuals here), m_tmpYuvPred,
   * the nine interpolated blocks of a fractional refinement round */
  Yuv16 resi_cu, resi_best, qt_resi[4];
  Yuv tmp_pred;
  uint8_t me_pred[9][CTU * CTU];
  int16_t me_h[3][72 * 64];                        /* sub-sample refinement: the three horizontally filtered planes of a round (14-bit intermediates, xExtDIFUpSamplingH/Q) */
  int par_ps[4];                                   /* eParentPartSize of the CU being compressed at each depth (SIZE_NONE: intra / none); read only with AMP on */
  /* Residual-coding memo of the inter candidates of ONE CU (encode_res_and_calc_rd_inter_cu): the inter RQT is a function of
   * the CU's prediction (= its per-partition motion field), the source block and the coder snapshot [depth][CI_CURR_BEST],
   * none of which change between the candidates of a CU except the motion.  Candidates whose motion field equals an earlier
   * one's (merge candidates with equal vectors, Nx2N / 2NxN / AMP shapes that found the 2Nx2N vector twice) reuse its transform
   * tree, levels and residual instead of searching them again.  Slots are per depth, FIFO, cleared when a CU starts. */
  int memo_valid[4][MEMO_K], memo_next[4], memo_zero[4][MEMO_K];
  int16_t memo_mv[4][MEMO_K][NPART][2]; int8_t memo_ref[4][MEMO_K][NPART];
  uint8_t memo_tr_idx[4][MEMO_K][NPART], memo_cbf[4][MEMO_K][3][NPART], memo_tskip[4][MEMO_K][3][NPART];
  alignas(16) int16_t memo_coef[MEMO_K * MEMO_POOL], memo_resi[MEMO_K * MEMO_POOL];
};

struct Env { Chain *C; Scratch *G; int cur_ctu, slice_start; };
/* transform unit descriptor (TComTU / TComTURecurse, TLibCommon/TComTU.cpp:47-207) */
struct TU { int log2, tr_depth, part, nparts, x, y, off_y, cw, cwo, cx, cy, c_tr_depth, c_code_all, off_c, cu_depth, plast; };

/* per-chain LDS */
struct Shared {
  /* hot coders, one LDS array so that coder ids index it directly:
   * [CAB_GOON] go-on coder, [CAB_CUR0+d] = [depth][CI_CURR_BEST], [CAB_LANE0+k] lane-private trial coders */
  Cabac cab[1 + (MAXDEPTH + 1) + MAXLC];
  uint8_t ref5[5][68]; int dc5[5]; uint32_t cm_dist[5];     /* chroma: per-mode reference samples (N <= 16) */
  /* Three tenants that are never live together (8 KB of LDS per chain = 20 chains per CU, five waves per SIMD):
   *  - reference samples of the block being predicted (+ the availability flags that build them): dead once the
   *    predictions of the batch are in the candidate pools, i.e. before any quantisation;
   *  - the records of the coefficient group in flight in the serial RDOQ;
   *  - the |level| lists of the lane-private bit counters. */
  union {
    int16_t lane_abs[MAXLC][32];                    /* per lane: |level| list of the coefficient group being coded [0..15], the group's levels [16..31] */
    RdoqRec rq_rec[16];                             /* serial RDOQ: records of the coefficient group in flight */
    struct {
      int32_t colsum[128];                          /* availability flags of build_ref / chroma_leaf_refs5 */
      union {                                       /* luma reference samples / scratch copy of the chroma sets */
        struct { uint8_t ref[264], reff[264]; };
        uint8_t ref5b[5][68];
      };
    };
  };
  union {
    int16_t rq_lv[16];                              /* serial RDOQ: levels of the coefficient group in flight */
    uint32_t sad[36];                               /* RMD SATD per mode; [35] SSE accumulator of a TU trial */
  };
  int dc;
  int best_idx[4], reco_best_idx[4];               /* which of cu[d][0/1] / reco[d][0/1] is "best" */
  /* PU / TU mailbox written by serial blocks */
  union {
    struct { int rd_mode[12]; int n_rd; int preds[3]; int n_mpm; };   /* luma PU */
    int uni[8];                                                       /* chroma leaf: transform-skip choice per mode */
  };
/* est: the estBit table of the coder RDOQ prices against, bits[ctx][bin] (TEncSbac.cpp:1722-1956), lives in the upper
   * lane coders (FCU_EST below): it is built and read before the bit count that loads them; [159] (the pad context) carries rqt_root_cbf for inter luma */
  union {                                           /* intra candidate batch / inter mailboxes: never live at the same time */
    struct {
      uint8_t vc_slot[MAXVC];                       /* lane coder that holds a variant's state after the bit count */
      int vc_abs[MAXVC], vc_lsp[MAXVC], vc_last[MAXVC]; uint32_t vc_dist[MAXVC]; uint32_t vc_bits[MAXVC]; double vc_cost[MAXVC];   /* chroma uses [10..14] of vc_dist */
    };
    struct {
      uint32_t acc[16];                             /* wave-reduced sums (SAD / SSE / Hadamard per candidate or variant) */
      unsigned long long me_best;                   /* (cost << 32) | raster index of the best integer position */
      int mrg_mv[5][2], mrg_ref[5], amvp[2][2];     /* merge candidates, AMVP candidates of the PU being searched */
      int iv_abs[6], iv_lsp[6], iv_top[6]; uint32_t iv_dist[6], iv_bits[6];   /* inter TU variants: Y, Y-ts, Cb, Cb-ts, Cr, Cr-ts */
      int it_abs[3], it_ts[3]; uint32_t it_dist[3]; /* chosen variant per component */
      double iq_cost[5]; uint32_t iq_bits[5], iq_dist[5], iq_zero;   /* xEstimateInterResidualQT accumulators per recursion level */
      int16_t tz_x[16], tz_y[16]; uint8_t tz_pt[16], tz_d[16]; int tz_n;          /* TZ search: the positions of the round in flight (test order) */
      uint32_t tz_best; int tz_bx, tz_by, tz_dist, tz_round, tz_point;               /* IntTZSearchStruct */
      int mrg_buf[5], best_is_skip, me_out[4];      /* xCheckRDCostMerge2Nx2N bookkeeping; motion_estimation results (mvx, mvy) */
    };
  };
  int pu_best_vc, pu_best_mode, pu_nvc; uint32_t pu_best_dist; double pu_best_cost;
  /* sequential TU trial mailbox */
  int t_abs, t_lsp, t_last; uint32_t t_dist;
  int rw_abs, rw_lsp;                               /* rdoq_wave's result */
  /* RQT recursion results */
  double q_cost[4]; uint32_t q_dist[4];
  /* chroma search */
  int c_best_mode; uint32_t c_best_dist; double c_best_cost; int c_modes[5];
  uint32_t c_dist;
  Env env;
  /* explicit stacks of the serial tree walkers (a private array indexed by the stack pointer would live in scratch memory) */
  TU wk_st[4]; int wk_ci[4]; int wk_part[4], wk_child[4];
  double cand_cost[12];                             /* RMD candidate costs (CandCostList, TEncSearch.cpp:2289) */
  uint32_t c64_dist[5]; uint8_t c64_cbf[5][4]; int c64_valid;      /* 64x64 first pass: per-candidate distortion / cbf of its four TUs */
  uint64_t q_frac[5], t_frac;                       /* exact (Q15) bit counts of the chosen RQT subtrees per recursion level / of the last walk */
  double dec_j0, dec_j1; int dec_cnt, dec_flip;     /* fork hooks: J0 / J1 / Num_OBF / bPartition_True of the CU being closed */
};

enum { CAB_GOON = 0, CAB_CUR0 = 1, CAB_LANE0 = 1 + (MAXDEPTH + 1) };
/* RDOQ's rate table (NCTX_INTRA x 2 words) borrows lane coders [MAXLC-7, MAXLC): est_build() fills it before the RDOQ of a
 * batch, the bit count that follows is the first to load those coders, and nothing reads the table afterwards. */
enum { EST_CODERS = 7 };
static_assert(EST_CODERS * sizeof(Cabac) >= NCTX_INTRA * 2 * sizeof(uint32_t) && EST_CODERS <= MAXLC, "est table does not fit its lane coders");
#define FCU_EST ((uint32_t *)&g_S.cab[CAB_LANE0 + MAXLC - EST_CODERS])
#ifdef FCU_EMU
static Shared g_S;
#else
__shared__ Shared g_S;
#endif
/* the wave's environment lives in LDS (g_S.env): non-inlined functions fetch it (to scalar registers) instead of
 * receiving it as six vector-register arguments that would be spilled around every call */
FCU_DEV Env env_get() { return FCU_UNI(g_S.env); }
FCU_DEV Cabac *slot_ptr(const Env E, int d, int ci) { return ci == CI_CURR_BEST ? &g_S.cab[CAB_CUR0 + d] : &E.G->slots[d][ci]; }


/* ======================================================================================== */
/* small helpers                                                                            */
/* ======================================================================================== */
FCU_DEV int part_x(int z) { return (k_z2r[z] & 15) << 2; }
FCU_DEV int part_y(int z) { return (k_z2r[z] >> 4) << 2; }
FCU_DEV int zidx_of(int lx, int ly) { return k_r2z[((ly & 63) >> 2) * 16 + ((lx & 63) >> 2)]; }
FCU_DEV int ilog2(int n) { int l = 0; while ((1 << l) < n) l++; return l; }
FCU_DEV int iabs(int v) { return v < 0 ? -v : v; }
FCU_DEV int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
FCU_DEV int clip3i(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
FCU_DEV uint8_t *yuv_plane(Yuv *b, int comp) { return comp == 0 ? b->y : (comp == 1 ? b->u : b->v); }
FCU_DEV double rd_cost(const Params &p, uint32_t bits, uint32_t dist)          /* TComRdCost::calcRdCost, TComRdCost.cpp:56-123 */
{ return FCU_FLOOR((double)dist + ((double)bits * p.lambda) + 0.5); }

FCU_DEV void tu_root(TU &t, int depth)
{
  t.log2 = 6 - depth; t.tr_depth = 0; t.part = 0; t.nparts = NPART >> (2 * depth); t.x = t.y = 0; t.off_y = 0;
  t.cw = t.cwo = (CTU >> depth) >> 1; t.cx = t.cy = 0; t.c_tr_depth = 0; t.c_code_all = 1; t.off_c = 0;
  t.cu_depth = depth; t.plast = 0;
}
FCU_DEV void tu_child(TU &c, const TU &p, int i, int processLast)
{
  const int s = 1 << (p.log2 - 1);
  c.log2 = p.log2 - 1; c.tr_depth = p.tr_depth + 1; c.cu_depth = p.cu_depth; c.plast = processLast;
  c.nparts = p.nparts >> 2; if (c.nparts < 1) c.nparts = 1;
  c.part = p.part + i * c.nparts;
  c.x = p.x + (i & 1) * s; c.y = p.y + (i >> 1) * s; c.off_y = p.off_y + i * s * s;
  const int pw = p.cwo;
  if ((pw >> 1) >= 4) {
    const int cs = pw >> 1;
    c.cw = c.cwo = cs; c.c_code_all = 1; c.c_tr_depth = p.c_tr_depth + 1;
    c.cx = p.cx + (i & 1) * cs; c.cy = p.cy + (i >> 1) * cs; c.off_c = p.off_c + i * cs * cs;
  } else {
    c.cwo = pw; c.c_code_all = 0; c.c_tr_depth = p.c_tr_depth; c.cx = p.cx; c.cy = p.cy; c.off_c = p.off_c;
    c.cw = (processLast ? (i == 3) : (i == 0)) ? pw : 0;
  }
}
/* A TU is a function of (CU depth, transform depth, first partition, processLast of the last split): non-inlined
 * functions receive this 13-bit key in one register and rebuild the descriptor with scalar arithmetic, instead of a
 * 64-byte struct passed through the stack (a scratch-memory round trip per call and lane). */
FCU_DEV uint32_t tu_key(const TU &t) { return (uint32_t)(t.cu_depth | (t.tr_depth << 2) | (t.part << 4) | (t.plast << 12)); }
FCU_DEV TU tu_of_key(uint32_t k)
{
  TU t; tu_root(t, (int)(k & 3));
  const int trd = (int)((k >> 2) & 3), part = (int)((k >> 4) & 255), pl = (int)((k >> 12) & 1);
  for (int l = 0; l < trd; l++) {
    TU c; int np = t.nparts >> 2; if (np < 1) np = 1;
    tu_child(c, t, ((part - t.part) / np) & 3, pl);
    t = c;
  }
  return t;
}
FCU_DEV int tu_part_c(const TU &t) { return t.c_code_all ? t.part : (t.part & ~3); }
FCU_DEV int tu_nparts_c(const TU &t) { return t.c_code_all ? t.nparts : t.nparts * 4; }

/* ======================================================================================== */
/* CABAC bit counter -- per-lane callable (TEncBinCoderCABACCounter.cpp:59-136)              */
/* ======================================================================================== */
FCU_DEV void cab_init(Cabac *c, int qp, int sliceType, int bTable)   /* ContextModel::init, ContextModel.cpp:56-65; TEncSbac::resetEntropy :106-156 */
{
  if (qp < 0) qp = 0; if (qp > 51) qp = 51;
  for (int i = 0; i < NCTX; i++) {
    int iv = sliceType == SLICE_P ? (bTable ? k_ctx_init_B[i] : k_ctx_init_P[i]) : k_ctx_init_I[i], slope = (iv >> 4) * 5 - 45, offset = ((iv & 15) << 3) - 16;
    int st = ((slope * qp) >> 4) + offset; st = st < 1 ? 1 : (st > 126 ? 126 : st);
    int mps = st >= 64;
    c->ctx[i] = (uint8_t)(((mps ? (st - 64) : (63 - st)) << 1) + mps);
  }
  c->frac = 0; c->bins = 0; c->pad_ = 0;
}
FCU_DEV void cab_copy(Cabac *d, const Cabac *s, int lane)     /* cooperative copy: 44 dwords */
{ const uint32_t *a = (const uint32_t *)s; uint32_t *b = (uint32_t *)d; if (lane < (int)(sizeof(Cabac) / 4)) b[lane] = a[lane]; }
FCU_DEV void cab_copy1(Cabac *d, const Cabac *s)              /* single-lane copy */
{ const uint32_t *a = (const uint32_t *)s; uint32_t *b = (uint32_t *)d; for (int i = 0; i < (int)(sizeof(Cabac) / 4); i++) b[i] = a[i]; }
/* the hot primitives take the coder id (index into g_S.cab) so that every access is a plain LDS access */
#define FCU_CB g_S.cab[cid]
FCU_DEV void cab_bin(int cid, int bin, int ctx)
{
  const uint32_t e = g_hot.bin[FCU_CB.ctx[ctx] * 2 + bin];      /* (bits << 8) | next state */
  FCU_CB.bins++;
  FCU_CB.frac += (uint64_t)(e >> 8);
  FCU_CB.ctx[ctx] = (uint8_t)e;
}
FCU_DEV void cab_ep(int cid, int n) { FCU_CB.bins += (uint32_t)n; FCU_CB.frac += (uint64_t)32768 * (uint64_t)n; }
FCU_DEV void cab_trm(int cid, int bin) { FCU_CB.bins++; FCU_CB.frac += (uint64_t)k_entropy_bits[126 ^ bin]; }
FCU_DEV void cab_reset_bits(int cid) { FCU_CB.frac &= 32767; FCU_CB.bins = 0; }
FCU_DEV uint32_t cab_bits(int cid) { return (uint32_t)(FCU_CB.frac >> 15); }
FCU_DEV int ctx_bits(int cid, int ctx, int bin) { return (int)(g_hot.bin[FCU_CB.ctx[ctx] * 2 + bin] >> 8); }
/* TEncSbac::estBit: the costs of both bins of every context of coder `cid`, one LDS word each; called by all lanes
 * in the phase before RDOQ (the contexts are frozen while RDOQ runs) */
enum { EST_ROOT_CBF = NCTX_INTRA - 1 };
FCU_DEV void est_build(int cid, int lane) { for (int i = lane; i < NCTX_INTRA * 2; i += 64) { const int cx = (i >> 1) == EST_ROOT_CBF ? CTX_ROOT_CBF : (i >> 1); FCU_EST[i] = g_hot.bin[FCU_CB.ctx[cx] * 2 + (i & 1)] >> 8; } }

/* TComDataCU::getCoefScanIdx, TComDataCU.cpp:3356-3411 */
FCU_DEV int coef_scan_idx(int dir, int log2, int comp)
{
  if (log2 > (comp ? 2 : 3)) return 0;
  if (iabs(dir - VER) <= 4) return 1;
  if (iabs(dir - HOR) <= 4) return 2;
  return 0;
}
/* coefficient-group flags: bit (cgy*wg+cgx) of a 64-bit mask (<= 8x8 groups) */
FCU_DEV int pattern_sig_ctx(uint64_t cg, int cgx, int cgy, int wg)             /* TComTrQuant.cpp:2584-2609 */
{
  if (wg <= 1) return 0;
  int r = 0, l = 0;
  if (cgx < wg - 1) r = (int)((cg >> (cgy * wg + cgx + 1)) & 1);
  if (cgy < wg - 1) l = (int)((cg >> ((cgy + 1) * wg + cgx)) & 1);
  return r + (l << 1);
}
FCU_DEV int sig_cg_ctx(uint64_t cg, int cgx, int cgy, int wg)                  /* TComTrQuant.cpp:2949-2969 */
{
  int r = 0, l = 0;
  if (cgx < wg - 1) r = (int)((cg >> (cgy * wg + cgx + 1)) & 1);
  if (cgy < wg - 1) l = (int)((cg >> ((cgy + 1) * wg + cgx)) & 1);
  return (r + l) != 0;
}
FCU_DEV int first_sig_ctx(int log2, int scan, int ch)                          /* TComChromaFormat.cpp:129-155 */
{ if (log2 == 2) return 0; if (log2 == 3) return 9 + ((scan != 0 && !ch) ? 6 : 0); return ch ? 12 : 21; }
FCU_DEV int sig_ctx_inc(int pattern, int first, int pos, int log2, int ch)     /* TComTrQuant.cpp:2619-2718 */
{
  const int py = pos >> log2, px = pos - (py << log2);
  if (px + py == 0) return 0;
  int offset;
  if (log2 == 2) offset = g_hot.ctx_ind_map4x4[4 * py + px];
  else {
    int cnt; const int xs = px & 3, ys = py & 3;
    if (pattern == 0) { int t = xs + ys; cnt = (t >= 3) ? 0 : ((t >= 1) ? 1 : 2); }
    else if (pattern == 1) cnt = (ys >= 2) ? 0 : ((ys >= 1) ? 1 : 2);
    else if (pattern == 2) cnt = (xs >= 2) ? 0 : ((xs >= 1) ? 1 : 2);
    else cnt = 2;
    offset = ((((px >> 2) + (py >> 2)) > 0) ? (ch ? 0 : 3) : 0) + cnt;
  }
  return first + offset;
}
FCU_DEV int coef_remain_bins(uint32_t symbol, uint32_t rparam)             /* bypass bins of xWriteCoefRemainExGolomb, TEncSbac.cpp:338-391 */
{
  int code = (int)symbol;
  if (code < (3 << rparam)) return (int)(((uint32_t)code >> rparam) + 1) + (int)rparam;
  uint32_t length = rparam;
  code -= 3 << rparam;
  while (code >= (1 << length)) code -= 1 << (length++);
  return (int)(3 + length + 1 - rparam) + (int)length;
}
/* TEncSbac::codeCoeffNxN (+codeTransformSkipFlags, codeLastSignificantXY), TEncSbac.cpp:997-1535.
 * Inside the engine the levels of a TU are kept in SCAN ORDER (coef[sp * st] = level at scan position sp;
 * st = 1 in CU objects, st = number of slots in an interleaved candidate batch), so the coder walks memory
 * linearly.  `last` = scan position of the last non-zero level, or -1: find it here (returns at once on an
 * all-zero TU).  The significant-group flags the reference gathers in its first loop (:1255-1275) are
 * derived group by group on the way down (a group's right/below neighbours come earlier in reverse scan). */
/* SER = 1: called by one lane (serial sections): every argument is wave-uniform */
template <int SER>
FCU_DEV FCU_INLINE void code_coeff_body(int c, const int16_t *coef, int st, int last, int log2, int comp, int scanType, int tsFlag, const Params &P_, int16_t *absCoeff)
{
  if (SER) { c = FCU_UNI(c); coef = FCU_UNI(coef); last = FCU_UNI(last); scanType = FCU_UNI(scanType); tsFlag = FCU_UNI(tsFlag); absCoeff = FCU_UNI(absCoeff); }
  const Params &P = *FCU_UNI(&P_);
  st = FCU_UNI(st); log2 = FCU_UNI(log2); comp = FCU_UNI(comp);
  FCU_IN_LDS(absCoeff);
  const FCU_HBM int16_t *coefg = (const FCU_HBM int16_t *)coef;
  const int ch = comp ? 1 : 0, N = 1 << log2; int n2 = N * N;
  if (last < 0 && st == 1 && n2 >= 64) {                       /* contiguous levels: skip empty runs of four groups (128 bytes) with wide loads */
    int q = (n2 >> 6) - 1;
    for (; q >= 0; q--) {
      uint32_t w[32];
      __builtin_memcpy(w, __builtin_assume_aligned((const void *)(coef + q * 64), 16), 128);   /* TU-contiguous levels start on 32-byte boundaries */
      uint32_t any = 0;
#pragma unroll
      for (int k = 0; k < 32; k++) any |= w[k];
      if (any) break;
    }
    if (q < 0) return;
    n2 = (q + 1) * 64;                                         /* the group-wise search below starts in the first non-empty run */
  }
  if (last < 0) {                                             /* find the last non-zero level, a group's sixteen loads at a time */
    for (int cg = (n2 >> 4) - 1; cg >= 0 && last < 0; cg--) {
      uint32_t nz = 0;
#pragma unroll
      for (int k = 0; k < 16; k++) nz |= (uint32_t)(coefg[(cg * 16 + k) * st] != 0) << k;
      if (nz) last = cg * 16 + 31 - __builtin_clz(nz);
    }
    if (last < 0) return;
  }
  /* bit/bin totals are kept in registers and added to the coder once at the end (sums commute) */
  uint64_t fr = 0; uint32_t nb = 0;
#define FCU_BIN(bin_, ctx_) do { const int cx_ = (ctx_); const uint32_t e_ = g_hot.bin[g_S.cab[c].ctx[cx_] * 2 + (bin_)]; nb++; fr += (uint64_t)(e_ >> 8); g_S.cab[c].ctx[cx_] = (uint8_t)e_; } while (0)
#define FCU_EP(n_) do { const uint32_t n__ = (uint32_t)(n_); nb += n__; fr += (uint64_t)32768 * (uint64_t)n__; } while (0)
  if (P.transform_skip && log2 == 2) FCU_BIN(tsFlag, CTX_TSKIP + ch);
  const uint16_t *scan = log2 == 2 ? &g_hot.scan[scanType][0] : k_scan + k_scan_off[scanType * 4 + log2 - 2];
  const uint8_t *scanCG = log2 <= 3 ? g_hot.scan_cg8[log2 == 2 ? 0 : scanType] : k_scan_cg + k_scan_cg_off[scanType * 4 + log2 - 2];
  const int wg = N >> 2, firstSig = first_sig_ctx(log2, scanType, ch);
  uint64_t cgflag = 0;
  const uint64_t scan4 = g_hot.scan4_nib[scanType], map4 = g_hot.map4_nib;
  const int scanPosLast = last, posLast = scan[last], lastVal = coefg[last * st];
  {
    int py = posLast >> log2, px = posLast - (py << log2);
    if (scanType == 2) { int t = px; px = py; py = t; }
    const int gx = g_hot.group_idx[px], gy = g_hot.group_idx[py], cc = log2 - 2;
    const int off = ch ? 0 : (cc * 3 + ((cc + 1) >> 2)), sh = ch ? cc : ((cc + 3) >> 2);
    const int bx = CTX_LASTX + (ch ? 15 : 0) + off, by = CTX_LASTY + (ch ? 15 : 0) + off, gmax = g_hot.group_idx[N - 1];
    int k;
    for (k = 0; k < gx; k++) FCU_BIN(1, bx + (k >> sh));
    if (gx < gmax) FCU_BIN(0, bx + (k >> sh));
    for (k = 0; k < gy; k++) FCU_BIN(1, by + (k >> sh));
    if (gy < gmax) FCU_BIN(0, by + (k >> sh));
    if (gx > 3) FCU_EP((gx - 2) >> 1);
    if (gy > 3) FCU_EP((gy - 2) >> 1);
  }
  const int baseCG = CTX_SIGCG + (ch ? 2 : 0), baseSig = CTX_SIG + (ch ? 28 : 0), lastSet = scanPosLast >> 4;
  uint32_t c1 = 1, goRice;
  int scanPosSig = scanPosLast;
  for (int sub = lastSet; sub >= 0; sub--) {
    int numNonZero = 0; const int subPos = sub << 4;
    goRice = 0;
    int lastNZ = -1, firstNZ = 16, escape = 0;
    if (scanPosSig == scanPosLast) { absCoeff[0] = (int16_t)iabs(lastVal); numNonZero = 1; lastNZ = scanPosSig; firstNZ = scanPosSig; scanPosSig--; }
    const int cgpos = scanCG[sub], cgy = cgpos / wg, cgx = cgpos - cgy * wg;
    /* the group's sixteen levels: one round of loads, parked in LDS for the serial walk below */
    int16_t *stage = absCoeff + 16;
    int any = 0;
    {
      int lv[16];
#pragma unroll
      for (int k = 0; k < 16; k++) lv[k] = coefg[(subPos + k) * st];
#pragma unroll
      for (int k = 0; k < 16; k++) { any |= lv[k]; stage[k] = (int16_t)lv[k]; }
    }
    if (sub == lastSet || sub == 0) cgflag |= 1ull << cgpos;
    else {
      if (any) cgflag |= 1ull << cgpos;
      FCU_BIN(any != 0, baseCG + sig_cg_ctx(cgflag, cgx, cgy, wg));
    }
    if ((cgflag >> cgpos) & 1) {
      const int pattern = pattern_sig_ctx(cgflag, cgx, cgy, wg);
      const uint32_t cntBits = g_hot.cnt_bits[pattern];
      const int sigBase = baseSig + firstSig + ((!ch && (cgx + cgy) > 0) ? 3 : 0);
      /* a group of a TU > 4x4 uses three significance contexts (+ DC): their states stay in registers for the group */
      uint32_t s0 = 0, s1 = 0, s2 = 0;
      if (log2 > 2) { s0 = g_S.cab[c].ctx[sigBase]; s1 = g_S.cab[c].ctx[sigBase + 1]; s2 = g_S.cab[c].ctx[sigBase + 2]; }
      for (; scanPosSig >= subPos; scanPosSig--) {
        const int v = stage[scanPosSig - subPos], sig = v != 0;
        const int p4 = (int)((scan4 >> (4 * (scanPosSig - subPos))) & 15);     /* getSigCtxInc on the packed 4x4 scan (see rdoq) */
        if (scanPosSig > subPos || sub == 0 || numNonZero) {
          if (log2 == 2) FCU_BIN(sig, baseSig + (p4 ? (int)((map4 >> (4 * p4)) & 15) : 0));
          else if (scanPosSig == 0) FCU_BIN(sig, baseSig);
          else {
            const int cnt = (int)((cntBits >> (2 * p4)) & 3);
            const uint32_t e = g_hot.bin[(cnt == 0 ? s0 : (cnt == 1 ? s1 : s2)) * 2 + sig], ns = e & 255;
            nb++; fr += (uint64_t)(e >> 8);
            if (cnt == 0) s0 = ns; else if (cnt == 1) s1 = ns; else s2 = ns;
          }
        }
        if (sig) { absCoeff[numNonZero++] = (int16_t)iabs(v); if (lastNZ == -1) lastNZ = scanPosSig; firstNZ = scanPosSig; }
      }
      if (log2 > 2) { g_S.cab[c].ctx[sigBase] = (uint8_t)s0; g_S.cab[c].ctx[sigBase + 1] = (uint8_t)s1; g_S.cab[c].ctx[sigBase + 2] = (uint8_t)s2; }
    } else scanPosSig = subPos - 1;
    if (numNonZero > 0) {
      const int signHidden = (lastNZ - firstNZ >= 4);
      const int ctxSet = (ch ? 4 : 0) + ((!ch && sub > 0) ? 2 : 0) + (c1 == 0);
      c1 = 1;
      const int baseOne = CTX_ONE + 4 * ctxSet, numC1 = numNonZero < 8 ? numNonZero : 8;
      int firstC2 = -1;
      for (int idx = 0; idx < numC1; idx++) {
        const int sym = absCoeff[idx] > 1;
        FCU_BIN(sym, baseOne + (int)c1);
        if (sym) { c1 = 0; if (firstC2 == -1) firstC2 = idx; else escape = 1; }
        else if (c1 < 3 && c1 > 0) c1++;
      }
      if (c1 == 0 && firstC2 != -1) { const int sym = absCoeff[firstC2] > 2; FCU_BIN(sym, CTX_ABS + ctxSet); if (sym) escape = 1; }
      escape = escape || (numNonZero > 8);
      FCU_EP((P.sign_hiding && signHidden) ? numNonZero - 1 : numNonZero);
      int firstCoeff2 = 1;
      if (escape)
        for (int idx = 0; idx < numNonZero; idx++) {
          const int base = (idx < 8) ? (2 + firstCoeff2) : 1;
          if (absCoeff[idx] >= base) {
            FCU_EP(coef_remain_bins((uint32_t)(absCoeff[idx] - base), goRice));
            if (absCoeff[idx] > (3 << goRice)) goRice = goRice + 1 < 4 ? goRice + 1 : 4;
          }
          if (absCoeff[idx] >= 2) firstCoeff2 = 0;
        }
    }
  }
  g_S.cab[c].frac += fr; g_S.cab[c].bins += nb;
#undef FCU_BIN
#undef FCU_EP
}

template <int SER>
FCU_DEV FCU_NOINLINE void code_coeff_nxn(int c, const int16_t *coef, int st, int last, int log2, int comp, int scanType, int tsFlag, const Params &P, int16_t *absCoeff)
{ code_coeff_body<SER>(c, coef, st, last, log2, comp, scanType, tsFlag, P, absCoeff); }

/* ======================================================================================== */
/* RDOQ -- per-lane callable (TComTrQuant::xRateDistOptQuant, TComTrQuant.cpp:2033-2573)     */
/* Works in the scan-order domain: the producer (forward transform phase) hands over          */
/* sign * lLevelDouble per scan position, the levels come back per scan position; element     */
/* stride `st` interleaves the candidates of a batch so that one lane-private iteration of    */
/* all lanes touches consecutive bytes.  The seven per-coefficient arrays of the reference     */
/* (:2082-2095) are one record per scan position.                                              */
/* ======================================================================================== */

/* sign(coef) * min(|coef| * quantScale, MAX_INT - (1 << (qbits-1)))   (TComTrQuant.cpp:2117-2128) */
FCU_DEV int32_t level_double(int32_t coef, int qcoef, int qbits)
{
  const long long t = (long long)iabs(coef) * qcoef, lim = 2147483647LL - (1LL << (qbits - 1));
  const int32_t l = (int32_t)(t < lim ? t : lim);
  return coef < 0 ? -l : l;
}
FCU_DEV int rdoq_qbits(int log2, int qp) { return 14 + qp / 6 + (15 - 8 - log2); }
/* uiMaxAbsLevel > 0 for this level_double() value (:2130) */
FCU_DEV int level_nonzero(int32_t ld, int qbits) { return iabs(ld) >= ((int32_t)1 << (qbits - 1)); }

struct LevelBits { int g10, g11, g20, g21; };              /* greater1 / greater2 flag costs of the current contexts */
FCU_DEV int ic_rate(const LevelBits &b, uint32_t absLevel, uint32_t goRice, uint32_t c1Idx, uint32_t c2Idx)
{                                                          /* xGetICRate, TComTrQuant.cpp:2807-2881 */
  int rate = 32768;
  const uint32_t baseLevel = (c1Idx < 8) ? (2 + (c2Idx < 1)) : 1;
  if (absLevel >= baseLevel) {
    uint32_t symbol = absLevel - baseLevel, length;
    if (symbol < (3u << goRice)) { length = symbol >> goRice; rate += (int)((length + 1 + goRice) << 15); }
    else {
      length = goRice; symbol = symbol - (3u << goRice);
      while (symbol >= (1u << length)) symbol -= (1u << (length++));
      rate += (int)((3 + length + 1 - goRice + length) << 15);
    }
    if (c1Idx < 8) { rate += b.g11; if (c2Idx < 1) rate += b.g21; }
  } else if (absLevel == 1) rate += b.g10;
  else if (absLevel == 2) { rate += b.g11; rate += b.g20; }
  else rate = 0;
  return rate;
}
/* rateHi / rateLo: xGetICRate of the two levels tried, maxAbsLevel and maxAbsLevel - 1 (rateLo only when that level is >= 1).
 * The caller needs the rates of the chosen level and of its two neighbours afterwards (rateIncUp / rateIncDown,
 * TComTrQuant.cpp:2263-2270): two of those three are among the values computed here. */
template <class CB>
FCU_DEV uint32_t coded_level(CB cb, double lambda, double *codedCost, double *codedCost0, double *codedCostSig,
                             int32_t levelDouble, uint32_t maxAbsLevel, int ctxSig, const LevelBits &lb,
                             uint32_t goRice, uint32_t c1Idx, uint32_t c2Idx, int qbits, double errScale, int last, int *rateHi, int *rateLo)
{                                                          /* xGetCodedLevel, TComTrQuant.cpp:2738-2794 */
  double currCostSig = 0; uint32_t bestAbs = 0;
  if (!last && maxAbsLevel < 3) {
    *codedCostSig = lambda * (double)cb(ctxSig, 0);
    *codedCost = *codedCost0 + *codedCostSig;
    if (maxAbsLevel == 0) return bestAbs;
  } else *codedCost = FCU_MAX_DOUBLE;
  if (!last) currCostSig = lambda * (double)cb(ctxSig, 1);
  const uint32_t minAbs = maxAbsLevel > 1 ? maxAbsLevel - 1 : 1;
  for (int a = (int)maxAbsLevel; a >= (int)minAbs; a--) {
    double err = (double)(levelDouble - ((int32_t)a << qbits));
    const int rate = ic_rate(lb, (uint32_t)a, goRice, c1Idx, c2Idx);
    if (a == (int)maxAbsLevel) *rateHi = rate; else *rateLo = rate;
    double cost = err * err * errScale + lambda * (double)rate;
    cost += currCostSig;
    if (cost < *codedCost) { bestAbs = (uint32_t)a; *codedCost = cost; *codedCostSig = currCostSig; }
  }
  return bestAbs;
}
/* `c` is the coder whose contexts estBit() would snapshot (TEncSbac.cpp:1722-1956); cbfCtx
 * is the QT-CBF context of this TU (getCtxQtCbf + getCBFContextOffset).
 * src[sp*st] = level_double() of the coefficient at scan position sp; dst[sp*st] receives its level.
 * topNZ = highest scan position whose uiMaxAbsLevel is > 0 (found by the producer phase; -1: none).
 * Levels are written for the scan positions of coefficient groups <= topNZ/16 only; everything above is
 * zero by construction and the consumers know topNZ.  Returns uiAbsSum and the scan position of the last
 * non-zero level (-1: none). */
struct RdoqOut { int abs_sum, last; };
/* EST = 1: the caller has built g_S.est for coder `c` (est_build); EST = 0: costs are looked up in the coder itself */
/* The quantiser without RDOQ: xQuant's plain branch (TComTrQuant.cpp:1160-1240) + signBitHidingHDQ (:991-1124) in the
 * engine's scan-order domain.  src = sign * |coefficient| * quantiser scale per scan position (stride st), dst = levels;
 * positions above the coefficient group of topNZ can only quantise to 0 and are not touched (callers treat them as 0).
 * abs_sum is the sum before sign hiding, as the reference returns it; last = highest non-zero scan position afterwards. */
FCU_DEV FCU_NOINLINE RdoqOut quant_plain(const int32_t *src, int16_t *dst, int st, int topNZ, int log2, int comp, const Params &P)
{
  RdoqOut o = { 0, -1 };
  if (topNZ < 0) return o;
  const FCU_HBM int32_t *srcg = (const FCU_HBM int32_t *)src; FCU_HBM int16_t *dstg = (FCU_HBM int16_t *)dst;
  const int qp = comp ? P.qp_c : P.qp, qbits = rdoq_qbits(log2, qp), qbits8 = qbits - 8;
  const long long add = (long long)(P.slice_type == SLICE_I ? 171 : 85) << (qbits - 9);
  const int cgTop = topNZ >> 4;
  int absSum = 0;
  for (int sp = cgTop * 16 + 15; sp >= 0; sp--) {
    const int32_t ld = srcg[sp * st];
    const int q = (int)(((long long)iabs(ld) + add) >> qbits);
    absSum += q;
    dstg[sp * st] = (int16_t)clip3i(-32768, 32767, ld < 0 ? -q : q);
  }
  o.abs_sum = absSum;
  if (P.sign_hiding && absSum >= 2) {
    int lastCG = -1;
    for (int cg = cgTop; cg >= 0; cg--) {
      const int sub = cg << 4;
      int lv[16], du[16];
      int first = 16, last = -1, sum = 0;
      for (int n = 0; n < 16; n++) {
        const int32_t ld = srcg[(sub + n) * st]; lv[n] = dstg[(sub + n) * st];
        du[n] = (int)(((long long)iabs(ld) - ((long long)iabs(lv[n]) << qbits)) >> qbits8);
        if (lv[n]) { if (first == 16) first = n; last = n; }
      }
      for (int n = first; n <= last; n++) sum += lv[n];
      if (last >= 0 && lastCG == -1) lastCG = 1;
      if (last - first >= 4) {
        const unsigned signbit = lv[first] > 0 ? 0 : 1;
        if (signbit != (unsigned)(sum & 1)) {
          int curCost = 0x7fffffff, minCostInc = 0x7fffffff, minPos = -1, finalChange = 0, curChange = 0;
          for (int n = (lastCG == 1 ? last : 15); n >= 0; --n) {
            if (lv[n] != 0) {
              if (du[n] > 0) { curCost = -du[n]; curChange = 1; }
              else if (n == first && iabs(lv[n]) == 1) curCost = 0x7fffffff;
              else { curCost = du[n]; curChange = -1; }
            } else if (n < first) {
              const unsigned thisSign = srcg[(sub + n) * st] >= 0 ? 0 : 1;
              if (thisSign != signbit) curCost = 0x7fffffff; else { curCost = -du[n]; curChange = 1; }
            } else { curCost = -du[n]; curChange = 1; }
            if (curCost < minCostInc) { minCostInc = curCost; finalChange = curChange; minPos = n; }
          }
          if (lv[minPos] == 32767 || lv[minPos] == -32768) finalChange = -1;
          const int nv = srcg[(sub + minPos) * st] >= 0 ? lv[minPos] + finalChange : lv[minPos] - finalChange;
          dstg[(sub + minPos) * st] = (int16_t)nv;
        }
      }
      if (lastCG == 1) lastCG = 0;
    }
  }
  for (int sp = cgTop * 16 + 15; sp >= 0; sp--) if (dstg[sp * st]) { o.last = sp; break; }
  return o;
}

/* The passes of xRateDistOptQuant after its first loop, shared by rdoq() and rdoq_wave(): the last significant position
 * (:2360-2436), the signs, sign bit hiding (:2442-2572). */
template <class CB>
FCU_DEV FCU_INLINE int rdoq_last_pos(CB cb, const FCU_HBM int16_t *dstg, const FCU_HBM RdoqRec *recg, const FCU_HBM double *cgg,
                                     int st, int log2, int ch, int scanType, int cbfCtx, const uint16_t *scan, const uint8_t *scanCG,
                                     uint64_t cgflag, int cgLastScanPos, int lastScanPos, double baseCost, double blockUncodedCost, double lambda)
{
  const int N = 1 << log2;
  double bestCost; int bestLastIdxP1 = 0;
  bestCost = blockUncodedCost + lambda * (double)cb(cbfCtx, 0);
  baseCost += lambda * (double)cb(cbfCtx, 1);
  /* estLastSignificantPositionBit (TEncSbac.cpp:1863-1923) evaluated on demand */
  const int lcc = log2 - 2, loff = ch ? 0 : (lcc * 3 + ((lcc + 1) >> 2)), lsh = ch ? lcc : ((lcc + 3) >> 2);
  const int lbx = CTX_LASTX + (ch ? 15 : 0) + loff, lby = CTX_LASTY + (ch ? 15 : 0) + loff, lgmax = g_hot.group_idx[N - 1];
  int foundLast = 0;
  for (int cgScanPos = cgLastScanPos; cgScanPos >= 0; cgScanPos--) {
    const int cgBlk = scanCG[cgScanPos];
    baseCost -= cgg[cgScanPos * st];
    if ((cgflag >> cgBlk) & 1) {
      const int top = (cgScanPos * 16 + 15 > lastScanPos) ? lastScanPos - cgScanPos * 16 : 15;   /* positions above the last one are skipped */
      int aLv = dstg[(cgScanPos * 16 + top) * st]; double aCs = recg[(cgScanPos * 16 + top) * st].cs;    /* one ahead */
      for (int posInCG = top; posInCG >= 0; posInCG--) {
        const int scanPos = cgScanPos * 16 + posInCG;
        const int lv = aLv; const double curCs = aCs;
        if (posInCG > 0) { aLv = dstg[(scanPos - 1) * st]; aCs = recg[(scanPos - 1) * st].cs; }
        if (lv) {
          const int blk = scan[scanPos];
          const int py = blk >> log2, px = blk - (py << log2);
          const int ax = scanType == 2 ? py : px, ay = scanType == 2 ? px : py;
          const int gx = g_hot.group_idx[ax], gy = g_hot.group_idx[ay];
          int bxs = 0, bys = 0;
          for (int k = 0; k < gx; k++) bxs += cb(lbx + (k >> lsh), 1);
          if (gx < lgmax) bxs += cb(lbx + (gx >> lsh), 0);
          for (int k = 0; k < gy; k++) bys += cb(lby + (k >> lsh), 1);
          if (gy < lgmax) bys += cb(lby + (gy >> lsh), 0);
          double r = (double)(bxs + bys);                                  /* xGetRateLast, TComTrQuant.cpp:2898-2916 */
          if (gx > 3) r += 32768.0 * (double)((gx - 2) >> 1);
          if (gy > 3) r += 32768.0 * (double)((gy - 2) >> 1);
          const double costLast = lambda * r;
          const double totalCost = baseCost + costLast - curCs;
          if (totalCost < bestCost) { bestLastIdxP1 = scanPos + 1; bestCost = totalCost; }
          if (lv > 1) { foundLast = 1; break; }
          baseCost -= recg[scanPos * st].cc; baseCost += recg[scanPos * st].c0;
        } else baseCost -= curCs;
      }
      if (foundLast) break;
    }
  }
  return bestLastIdxP1;
}
/* signs back on the kept levels, zeros above the chosen last position; returns uiAbsSum */
FCU_DEV FCU_INLINE int rdoq_signs(const FCU_HBM int32_t *srcg, FCU_HBM int16_t *dstg, int st, int bestLastIdxP1, int lastScanPos)
{
  int absSum = 0;
  for (int sp0 = 0; sp0 < bestLastIdxP1; sp0 += 16) {           /* signs back on the kept levels, sixteen positions per round of loads */
    int lv[16]; int32_t sv[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { const int ok = sp0 + k < bestLastIdxP1; lv[k] = ok ? dstg[(sp0 + k) * st] : 0; sv[k] = ok ? srcg[(sp0 + k) * st] : 0; }
#pragma unroll
    for (int k = 0; k < 16; k++) if (sp0 + k < bestLastIdxP1) { absSum += lv[k]; dstg[(sp0 + k) * st] = (int16_t)((sv[k] < 0) ? -lv[k] : lv[k]); }
  }
  for (int sp = bestLastIdxP1; sp <= lastScanPos; sp++) dstg[sp * st] = 0;

  return absSum;
}
FCU_DEV FCU_INLINE void rdoq_sign_hiding(const Params &P, const FCU_HBM int32_t *srcg, FCU_HBM int16_t *dstg, const FCU_HBM RdoqRec *recg, int st, int ch, int bestLastIdxP1, int absSum)
{
  if (P.sign_hiding && absSum >= 2) {                        /* TComTrQuant.cpp:2442-2572 */
    const long long rdFactor = P.rd_factor[ch];
    int lastCG = -1;
    for (int subSet = (bestLastIdxP1 - 1) >> 4; subSet >= 0; subSet--) {   /* groups above hold no level any more */
      const int subPos = subSet << 4; int firstNZ = 16, lastNZ = -1, sum = 0, n;
      uint32_t nzMask = 0, negMask = 0;                          /* the group's sixteen levels in one go */
      {
        int lv16[16];
#pragma unroll
        for (int k = 0; k < 16; k++) lv16[k] = dstg[(subPos + k) * st];
#pragma unroll
        for (int k = 0; k < 16; k++) { sum += lv16[k]; nzMask |= (uint32_t)(lv16[k] != 0) << k; negMask |= (uint32_t)(lv16[k] < 0) << k; }
      }
      if (nzMask) { lastNZ = 31 - __builtin_clz(nzMask); firstNZ = __builtin_ctz(nzMask); }
      if (lastNZ >= 0 && lastCG == -1) lastCG = 1;
      if (lastNZ - firstNZ >= 4) {
        const uint32_t signbit = (negMask >> firstNZ) & 1;
        if (signbit != (uint32_t)(sum & 1)) {
          long long minCostInc = 0x7fffffffffffffffLL, curCost = 0x7fffffffffffffffLL;
          int minPos = -1, finalChange = 0, curChange = 0;
          for (n = (lastCG == 1 ? lastNZ : 15); n >= 0; --n) {
            const int sp = n + subPos; const int lv = dstg[sp * st]; const RdoqRec q = recg[sp * st];
            if (lv != 0) {
              const long long costUp = rdFactor * (-q.du) + q.up;
              long long costDown = rdFactor * (q.du) + q.dn - ((iabs(lv) == 1) ? q.sd : 0);
              if (lastCG == 1 && lastNZ == n && iabs(lv) == 1) costDown -= (4 << 15);
              if (costUp < costDown) { curCost = costUp; curChange = 1; }
              else { curChange = -1; if (n == firstNZ && iabs(lv) == 1) curCost = 0x7fffffffffffffffLL; else curCost = costDown; }
            } else {
              curCost = rdFactor * (-(long long)(iabs(q.du))) + (1 << 15) + q.up + q.sd;
              curChange = 1;
              if (n < firstNZ) { const uint32_t thissign = srcg[sp * st] >= 0 ? 0 : 1; if (thissign != signbit) curCost = 0x7fffffffffffffffLL; }
            }
            if (curCost < minCostInc) { minCostInc = curCost; finalChange = curChange; minPos = sp; }
          }
          const int old = dstg[minPos * st];
          if (old == 32767 || old == -32768) finalChange = -1;
          const int nv = srcg[minPos * st] >= 0 ? old + finalChange : old - finalChange;
          dstg[minPos * st] = (int16_t)nv;
        }
      }
      if (lastCG == 1) lastCG = 0;
    }
  }
}
template <int SER, class CB>
FCU_DEV FCU_INLINE RdoqOut rdoq_finish(CB cb, const Params &P, const FCU_HBM int32_t *srcg, FCU_HBM int16_t *dstg, FCU_HBM RdoqRec *recg, FCU_HBM double *cgg,
                                       int st, int log2, int ch, int scanType, int cbfCtx, const uint16_t *scan, const uint8_t *scanCG,
                                       uint64_t cgflag, int cgLastScanPos, int lastScanPos, double baseCost, double blockUncodedCost, double lambda)
{
  FCU_RTIC(rt_);
  const int bestLastIdxP1 = rdoq_last_pos(cb, dstg, recg, cgg, st, log2, ch, scanType, cbfCtx, scan, scanCG, cgflag, cgLastScanPos, lastScanPos, baseCost, blockUncodedCost, lambda);
  FCU_RTOC(P, rt_, 13);                                       /* last-position search */
  const int absSum = rdoq_signs(srcg, dstg, st, bestLastIdxP1, lastScanPos);
  FCU_RTOC(P, rt_, 14);                                       /* signs */
  rdoq_sign_hiding(P, srcg, dstg, recg, st, ch, bestLastIdxP1, absSum);
  FCU_RTOC(P, rt_, 15);                                       /* sign hiding */
  int last = bestLastIdxP1 - 1;                              /* sign hiding may have cleared the last level */
  while (last >= 0 && dstg[last * st] == 0) last--;
  RdoqOut o = { absSum, last };
  return o;
}


template <int SER, int EST>
FCU_DEV FCU_NOINLINE RdoqOut rdoq(int c, const int32_t *src, int16_t *dst, int st, int topNZ, int log2, int comp, int scanType, int cbfCtx, const Params &P_, RdoqRec *rec, double *costCGSig)
{
  const Params &P = *FCU_UNI(&P_);
  st = FCU_UNI(st); log2 = FCU_UNI(log2); comp = FCU_UNI(comp); cbfCtx = FCU_UNI(cbfCtx);
  if (SER) { c = FCU_UNI(c); src = FCU_UNI(src); dst = FCU_UNI(dst); topNZ = FCU_UNI(topNZ); scanType = FCU_UNI(scanType); rec = FCU_UNI(rec); costCGSig = FCU_UNI(costCGSig); }
  FCU_RTIC(rt_);
  if (topNZ < 0) { RdoqOut z = { 0, -1 }; return z; }            /* every level is 0: the reference leaves with uiAbsSum 0 (:2330) */
  const FCU_HBM int32_t *srcg = (const FCU_HBM int32_t *)src; FCU_HBM int16_t *dstg = (FCU_HBM int16_t *)dst;
  FCU_HBM RdoqRec *recg = (FCU_HBM RdoqRec *)rec; FCU_HBM double *cgg = (FCU_HBM double *)costCGSig;
  auto cb = [&](int ctx, int bin) -> int { return EST ? (int)FCU_EST[ctx * 2 + bin] : ctx_bits(c, ctx, bin); };
  const int ch = comp ? 1 : 0, N = 1 << log2, n2 = N * N;
  const int qp = comp ? P.qp_c : P.qp;
  const int qbits = rdoq_qbits(log2, qp);
  const double lambda = P.rdoq_lambda[comp];
  const double errScale = P.err_scale[ch][log2 - 2];
  /* the reference clears all seven arrays; only entries at scan positions <= the last significant one are
   * ever read back (last-position search, group zero-out, sign hiding), and each of those is written below */
  const uint16_t *scan = log2 == 2 ? &g_hot.scan[scanType][0] : k_scan + k_scan_off[scanType * 4 + log2 - 2];
  const uint8_t *scanCG = log2 <= 3 ? g_hot.scan_cg8[log2 == 2 ? 0 : scanType] : k_scan_cg + k_scan_cg_off[scanType * 4 + log2 - 2];
  const int wg = N >> 2, firstSig = first_sig_ctx(log2, scanType, ch);
  const int sigOff = CTX_SIG + (ch ? 28 : 0), cgBase = CTX_SIGCG + (ch ? 2 : 0);
  uint64_t cgflag = 0;
  int cgLastScanPos = -1; uint32_t ctxSet = 0; int c1 = 1, c2 = 0;
  double baseCost = 0, blockUncodedCost = 0;
  int lastScanPos = -1; uint32_t c1Idx = 0, c2Idx = 0, goRice = 0;
  int absSum = 0;

  const uint64_t scan4 = g_hot.scan4_nib[scanType], map4 = g_hot.map4_nib;
  int g10 = 0, g10Ctx = -1;
  /* coefficient groups above the last candidate level: only the uncoded cost accumulates (in scan order) */
  const int cgTop = topNZ >> 4;
  for (int cg = (n2 >> 4) - 1; cg > cgTop; cg--) {              /* a group's sixteen loads first, then the adds in scan order */
    int32_t v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = srcg[(cg * 16 + k) * st];
#pragma unroll
    for (int k = 15; k >= 0; k--) { const double err = (double)iabs(v[k]); blockUncodedCost += err * err * errScale; }
  }
  baseCost = blockUncodedCost;
  FCU_RTOC(P, rt_, 11);                                       /* set-up + uncoded tail */
  int32_t ahead = srcg[(cgTop * 16 + 15) * st];
  for (int cgScanPos = cgTop; cgScanPos >= 0; cgScanPos--) {
    const int cgBlk = scanCG[cgScanPos], cgy = cgBlk / wg, cgx = cgBlk - cgy * wg;
    double rdSigCost = 0, rdSigCost0 = 0, rdCodedLevelandDist = 0, rdUncodedDist = 0; int nnzBeforePos0 = 0;
    const int pattern = pattern_sig_ctx(cgflag, cgx, cgy, wg);
    cgg[cgScanPos * st] = 0;
    /* The contexts are frozen during RDOQ (estBit snapshot, TEncSbac.cpp:1722-1956), so the significance costs of a
     * group are loop invariants: in a TU > 4x4 only the three contexts sigBase + cnt occur besides DC (sig_ctx_inc). */
    const int sigBase = sigOff + firstSig + ((!ch && (cgx + cgy) > 0) ? 3 : 0);
    int b00 = 0, b01 = 0, b02 = 0, b10 = 0, b11 = 0, b12 = 0;
    if (log2 > 2) {
      b00 = cb(sigBase, 0); b01 = cb(sigBase + 1, 0); b02 = cb(sigBase + 2, 0);
      b10 = cb(sigBase, 1); b11 = cb(sigBase + 1, 1); b12 = cb(sigBase + 2, 1);
    }
    const uint32_t cntBits = g_hot.cnt_bits[pattern];
    for (int posInCG = 15; posInCG >= 0; posInCG--) {
      const int scanPos = cgScanPos * 16 + posInCG;
      const int32_t levelDouble = iabs(ahead);
      if (scanPos > 0) ahead = srcg[(scanPos - 1) * st];      /* one ahead: its latency overlaps this coefficient's work */
      uint32_t maxAbsLevel = (uint32_t)((levelDouble + ((int32_t)1 << (qbits - 1))) >> qbits);
      if (maxAbsLevel > 32767u) maxAbsLevel = 32767u;
      const double err = (double)levelDouble;
      double c0 = err * err * errScale, cc = 0, cs = 0;
      blockUncodedCost += c0;
      uint32_t level = maxAbsLevel;
      if (maxAbsLevel > 0 && lastScanPos < 0) {
        lastScanPos = scanPos;
        ctxSet = (uint32_t)((ch ? 4 : 0) + ((!ch && (scanPos >> 4) > 0) ? 2 : 0));
        cgLastScanPos = cgScanPos;
      }
      if (lastScanPos >= 0) {
        const int oneCtx = CTX_ONE + 4 * (int)ctxSet + c1;
        if (oneCtx != g10Ctx) { g10 = cb(oneCtx, 0); g10Ctx = oneCtx; }
        /* position inside the 4x4 group (raster) from the packed 4x4 scan; its significance context */
        const int p4 = (int)((scan4 >> (4 * posInCG)) & 15);
        int ctxSig, sbit0, sbit1;
        if (log2 == 2) { ctxSig = sigOff + (p4 ? (int)((map4 >> (4 * p4)) & 15) : 0); sbit0 = cb(ctxSig, 0); sbit1 = cb(ctxSig, 1); }
        else if (scanPos == 0) { ctxSig = sigOff; sbit0 = cb(ctxSig, 0); sbit1 = cb(ctxSig, 1); }
        else { const int cnt = (int)((cntBits >> (2 * p4)) & 3); ctxSig = sigBase + cnt; sbit0 = cnt == 0 ? b00 : (cnt == 1 ? b01 : b02); sbit1 = cnt == 0 ? b10 : (cnt == 1 ? b11 : b12); }
        int sdel = 0, rup, rdn = 0;
        if (maxAbsLevel == 0) {                              /* xGetCodedLevel's early exit (:2752-2760); never the last position */
          /* short path: level 0 leaves c1/c2/Rice state alone; everything the long path does for it, and on to the next */
          cs = lambda * (double)sbit0; cc = c0 + cs;
          RdoqRec r; r.cc = cc; r.cs = cs; r.c0 = c0; r.up = g10; r.dn = 0; r.sd = sbit1 - sbit0;
          r.du = (int32_t)(levelDouble >> (qbits - 8));
          if (SER) g_S.rq_rec[posInCG] = r; else recg[scanPos * st] = r;
          baseCost += cc;
          if (SER) g_S.rq_lv[posInCG] = 0; else dstg[scanPos * st] = 0;
          rdSigCost += cs;
          if (posInCG == 0) rdSigCost0 = cs;
          continue;
        } else {
          const int absCtx = CTX_ABS + (int)ctxSet + c2;
          LevelBits lb; lb.g10 = g10; lb.g11 = cb(oneCtx, 1); lb.g20 = cb(absCtx, 0); lb.g21 = cb(absCtx, 1);
          int rateHi = 0, rateLo = 0;                          /* xGetICRate(maxAbsLevel), (maxAbsLevel - 1) from the level choice */
          if (scanPos == lastScanPos)
            level = coded_level(cb, lambda, &cc, &c0, &cs, levelDouble, maxAbsLevel,
                                sigOff, lb, goRice, c1Idx, c2Idx, qbits, errScale, 1, &rateHi, &rateLo);
          else {
            level = coded_level(cb, lambda, &cc, &c0, &cs, levelDouble, maxAbsLevel,
                                ctxSig, lb, goRice, c1Idx, c2Idx, qbits, errScale, 0, &rateHi, &rateLo);
            sdel = sbit1 - sbit0;
          }
          if (level > 0) {                                     /* the chosen level is maxAbsLevel or maxAbsLevel - 1: one new rate, not three */
            if (level == maxAbsLevel) {
              rup = ic_rate(lb, level + 1, goRice, c1Idx, c2Idx) - rateHi;
              rdn = (maxAbsLevel > 1 ? rateLo : ic_rate(lb, 0, goRice, c1Idx, c2Idx)) - rateHi;
            } else {
              rup = rateHi - rateLo;
              rdn = ic_rate(lb, level - 1, goRice, c1Idx, c2Idx) - rateLo;
            }
          } else rup = lb.g10;
        }
        RdoqRec r; r.cc = cc; r.cs = cs; r.c0 = c0; r.up = rup; r.dn = rdn; r.sd = sdel;
        r.du = (int32_t)((levelDouble - ((int32_t)level << qbits)) >> (qbits - 8));
        if (SER) g_S.rq_rec[posInCG] = r; else recg[scanPos * st] = r;
        baseCost += cc;
        const uint32_t baseLevel = (c1Idx < 8) ? (2 + (c2Idx < 1)) : 1;
        if (level >= baseLevel) { if (level > 3u * (1u << goRice)) goRice = goRice + 1 < 4 ? goRice + 1 : 4; }
        if (level >= 1) c1Idx++;
        if (level > 1) { c1 = 0; c2 += (c2 < 2); c2Idx++; }
        else if (c1 < 3 && c1 > 0 && level) c1++;
      } else baseCost += c0;
      if (SER) g_S.rq_lv[posInCG] = (int16_t)level; else dstg[scanPos * st] = (int16_t)level;
      rdSigCost += cs;
      if (posInCG == 0) rdSigCost0 = cs;
      if (level) {
        cgflag |= 1ull << cgBlk;
        rdCodedLevelandDist += cc - cs;
        rdUncodedDist += c0;
        if (posInCG != 0) nnzBeforePos0++;
      }
    }
    if (lastScanPos >= 0 && cgScanPos > 0) {                 /* context set of the next group (:2306-2316), once the last position is known */
      ctxSet = (uint32_t)((ch ? 4 : 0) + ((!ch && (cgScanPos - 1) > 0) ? 2 : 0) + (c1 == 0));
      c1 = 1; c2 = 0; c1Idx = 0; c2Idx = 0; goRice = 0;
    }
    if (cgLastScanPos >= 0) {
      if (cgScanPos) {
        if (((cgflag >> cgBlk) & 1) == 0) {
          const int ctxSig = cgBase + sig_cg_ctx(cgflag, cgx, cgy, wg);
          baseCost += lambda * (double)cb(ctxSig, 0) - rdSigCost;
          cgg[cgScanPos * st] = lambda * (double)cb(ctxSig, 0);
        } else if (cgScanPos < cgLastScanPos) {
          if (nnzBeforePos0 == 0) { baseCost -= rdSigCost0; rdSigCost -= rdSigCost0; }
          double costZeroCG = baseCost;
          const int ctxSig = cgBase + sig_cg_ctx(cgflag, cgx, cgy, wg);
          baseCost += lambda * (double)cb(ctxSig, 1);
          costZeroCG += lambda * (double)cb(ctxSig, 0);
          cgg[cgScanPos * st] = lambda * (double)cb(ctxSig, 1);
          costZeroCG += rdUncodedDist; costZeroCG -= rdCodedLevelandDist; costZeroCG -= rdSigCost;
          if (costZeroCG < baseCost) {
            cgflag &= ~(1ull << cgBlk); baseCost = costZeroCG;
            cgg[cgScanPos * st] = lambda * (double)cb(ctxSig, 0);
            for (int posInCG = 15; posInCG >= 0; posInCG--) {
              const int scanPos = cgScanPos * 16 + posInCG;
              if (SER) { if (g_S.rq_lv[posInCG]) { g_S.rq_lv[posInCG] = 0; g_S.rq_rec[posInCG].cc = g_S.rq_rec[posInCG].c0; g_S.rq_rec[posInCG].cs = 0; } }
              else if (dstg[scanPos * st]) { dstg[scanPos * st] = 0; recg[scanPos * st].cc = recg[scanPos * st].c0; recg[scanPos * st].cs = 0; }
            }
          }
        }
      } else cgflag |= 1ull << cgBlk;
    }
    if (SER) {                                               /* the serial variant kept the group in LDS: one burst of stores per group */
#pragma unroll
      for (int k = 0; k < 16; k++) { recg[(cgScanPos * 16 + k) * st] = g_S.rq_rec[k]; dstg[(cgScanPos * 16 + k) * st] = g_S.rq_lv[k]; }
    }
  }
  FCU_RTOC(P, rt_, 12);                                       /* main loop */
  if (lastScanPos < 0) { RdoqOut z = { 0, -1 }; return z; }
  return rdoq_finish<SER>(cb, P, srcg, dstg, recg, cgg, st, log2, ch, scanType, cbfCtx, scan, scanCG, cgflag, cgLastScanPos, lastScanPos, baseCost, blockUncodedCost, lambda);
}

/* A value every lane holds one element of, readable and writable by element from wave-uniform code: a register plus
 * v_readlane / v_writelane on the GPU, an array on the CPU emulator (whose lanes run one after the other). */
#ifdef FCU_EMU
template <class T> struct LaneVar {
  T v[64];
  LaneVar() { for (int i = 0; i < 64; i++) v[i] = T(); }
  void set(int lane, T x) { v[lane] = x; }                   /* inside a lane loop: this lane's element */
  T own(int lane) const { return v[lane]; }
  T get(int k) const { return v[k]; }                        /* wave-uniform code: element k */
  void put(int k, T x) { v[k] = x; }
};
FCU_DEV uint32_t lane_mask16(const LaneVar<int> &f) { uint32_t m = 0; for (int k = 0; k < 16; k++) m |= (uint32_t)(f.v[k] != 0) << k; return m; }
#else
#if defined(__HIP_DEVICE_COMPILE__)
#define FCU_READLANE(v, k) __builtin_amdgcn_readlane((v), (k))
#define FCU_WRITELANE(val_, k_, old_) (((int)threadIdx.x == (k_)) ? (val_) : (old_))     /* (this compiler has no v_writelane builtin: compare + select) */
#define FCU_BALLOT(p) __builtin_amdgcn_ballot_w64(p)
#else                                                         /* host pass over the device code: parsed, never run */
#define FCU_READLANE(v, k) (v)
#define FCU_WRITELANE(val_, k_, old_) (val_)
#define FCU_BALLOT(p) ((unsigned long long)(p))
#endif
template <class T> struct LaneVar;
template <> struct LaneVar<int> {
  int r = 0;
  __device__ void set(int, int x) { r = x; }
  __device__ int own(int) const { return r; }
  __device__ int get(int k) const { return FCU_READLANE(r, k); }
  __device__ void put(int k, int x) { r = FCU_WRITELANE(x, k, r); }
};
template <> struct LaneVar<double> {
  double r = 0;
  __device__ void set(int, double x) { r = x; }
  __device__ double own(int) const { return r; }
  __device__ double get(int k) const
  { union { double d; int i[2]; } u; u.d = r; u.i[0] = FCU_READLANE(u.i[0], k); u.i[1] = FCU_READLANE(u.i[1], k); return u.d; }
  __device__ void put(int k, double x)
  { union { double d; int i[2]; } u, w; u.d = r; w.d = x; u.i[0] = FCU_WRITELANE(w.i[0], k, u.i[0]); u.i[1] = FCU_WRITELANE(w.i[1], k, u.i[1]); r = u.d; }
};
__device__ inline uint32_t lane_mask16(const LaneVar<int> &f) { return (uint32_t)FCU_BALLOT(f.r != 0) & 0xffffu; }
#endif

#ifndef FCU_WAVE_RDOQ_MIN_LOG2
#define FCU_WAVE_RDOQ_MIN_LOG2 2
#endif
/* The same quantiser for ONE block with the whole wave (called from wave-uniform code, not from inside a lane loop): the
 * un-split trial of a transform unit has a single candidate, and then a lane-private rdoq() leaves 63 lanes idle while one
 * lane walks every coefficient.  What is serial in xRateDistOptQuant's first loop is the order of the floating-point sums and
 * the greater1 / greater2 / Rice state, which only a non-zero level moves; everything else about a coefficient -- |level_double|,
 * uiMaxAbsLevel, its uncoded cost, its significance context and rates, and for a coefficient that can only quantise to 0 the
 * whole record -- depends on its position and on the group flags of the groups already walked.  So per coefficient group:
 *   lanes 0..15   one coefficient each, results kept in the lane's registers (LaneVar);
 *   the wave      the walk in scan order as wave-uniform code: a coefficient's prepared numbers come over v_readlane (no memory
 *                 access), three additions for one whose uiMaxAbsLevel is 0, the level choice (coded_level) for the others,
 *                 results back into the owning lane (v_writelane); then the group decisions;
 *   lanes 0..15   each stores its coefficient's record and level to the pools.
 * Every floating-point expression and every comparison is the one rdoq() evaluates, in the same order; the passes after the
 * first loop are rdoq_finish() on lane 0.  st = 1, EST = 1 (the caller has built g_S.est for coder c).
 * Result: g_S.rw_abs / g_S.rw_lsp. */
FCU_DEV FCU_NOINLINE void rdoq_wave(int c, const int32_t *src, int16_t *dst, int topNZ, int log2, int comp, int scanType, int cbfCtx, const Params &P_, RdoqRec *rec, double *costCGSig)
{
  const Params &P = *FCU_UNI(&P_);
  c = FCU_UNI(c); src = FCU_UNI(src); dst = FCU_UNI(dst); topNZ = FCU_UNI(topNZ); log2 = FCU_UNI(log2); comp = FCU_UNI(comp);
  scanType = FCU_UNI(scanType); cbfCtx = FCU_UNI(cbfCtx); rec = FCU_UNI(rec); costCGSig = FCU_UNI(costCGSig);
  if (topNZ < 0) { FCU_SERIAL { g_S.rw_abs = 0; g_S.rw_lsp = -1; } return; }   /* every level is 0: the reference leaves with uiAbsSum 0 (:2330) */
  const FCU_HBM int32_t *srcg = (const FCU_HBM int32_t *)src; FCU_HBM int16_t *dstg = (FCU_HBM int16_t *)dst;
  FCU_HBM RdoqRec *recg = (FCU_HBM RdoqRec *)rec; FCU_HBM double *cgg = (FCU_HBM double *)costCGSig;
  auto cb = [&](int ctx, int bin) -> int { return (int)FCU_EST[ctx * 2 + bin]; };
  const int ch = comp ? 1 : 0, N = 1 << log2, n2 = N * N;
  const int qp = comp ? P.qp_c : P.qp;
  const int qbits = rdoq_qbits(log2, qp);
  const double lambda = P.rdoq_lambda[comp];
  const double errScale = P.err_scale[ch][log2 - 2];
  const uint16_t *scan = log2 == 2 ? &g_hot.scan[scanType][0] : k_scan + k_scan_off[scanType * 4 + log2 - 2];
  const uint8_t *scanCG = log2 <= 3 ? g_hot.scan_cg8[log2 == 2 ? 0 : scanType] : k_scan_cg + k_scan_cg_off[scanType * 4 + log2 - 2];
  const int wg = N >> 2, firstSig = first_sig_ctx(log2, scanType, ch);
  const int sigOff = CTX_SIG + (ch ? 28 : 0), cgBase = CTX_SIGCG + (ch ? 2 : 0);
  const uint64_t scan4 = g_hot.scan4_nib[scanType], map4 = g_hot.map4_nib;
  const int cgTop = topNZ >> 4;
  /* the walk's state: wave-uniform values (the same in every lane) */
  uint64_t cgflag = 0;
  int cgLastScanPos = -1; uint32_t ctxSet = 0; int c1 = 1, c2 = 0;
  double baseCost = 0, blockUncodedCost = 0;
  int lastScanPos = -1; uint32_t c1Idx = 0, c2Idx = 0, goRice = 0;
  int g10 = 0, g10Ctx = -1;
  /* coefficient groups above the last candidate level: only the uncoded cost accumulates (in scan order); lanes 0..15 load and
   * square, the wave adds in order */
  for (int cg = (n2 >> 4) - 1; cg > cgTop; cg -= 4) {           /* four groups per round of loads: lanes 16 j .. 16 j + 15 take group cg - j */
    LaneVar<double> t0;
    FCU_FOR_LANES { const int g = cg - (lane >> 4); if (g > cgTop) { const double err = (double)iabs(srcg[g * 16 + (lane & 15)]); t0.set(lane, err * err * errScale); } }
    for (int j = 0; j < 4 && cg - j > cgTop; j++)
      for (int k = 15; k >= 0; k--) blockUncodedCost += t0.get(j * 16 + k);
  }
  baseCost = blockUncodedCost;
  /* memory latency off the walk: lane k holds the block position of group k for the whole call, and a group's coefficients are
   * fetched while the group before it is walked */
  LaneVar<int> vCgBlk, vNext;
  LaneVar<double> vCgSig;                                      /* costCGSig of group k in lane k (also stored to the pool) */
  FCU_FOR_LANES { if (lane <= cgTop) vCgBlk.set(lane, scanCG[lane]); if (lane < 16) vNext.set(lane, srcg[cgTop * 16 + lane]); }
  for (int cgScanPos = cgTop; cgScanPos >= 0; cgScanPos--) {
    const int cgBlk = vCgBlk.get(cgScanPos), cgy = cgBlk / wg, cgx = cgBlk - cgy * wg;
    const int sigBase = sigOff + firstSig + ((!ch && (cgx + cgy) > 0) ? 3 : 0);
    const uint32_t cntBits = g_hot.cnt_bits[pattern_sig_ctx(cgflag, cgx, cgy, wg)];
    LaneVar<int> vLd, vMax, vSb0, vSb1, vUp, vDn, vSd, vDu, vLv, vCand;
    LaneVar<double> vC0, vCs, vCc;
    FCU_FOR_LANES {
      if (lane < 16) {                                         /* one coefficient of this group per lane */
        const int posInCG = lane, scanPos = cgScanPos * 16 + posInCG;
        const int32_t levelDouble = iabs(vNext.own(lane));
        if (cgScanPos > 0) vNext.set(lane, srcg[scanPos - 16]);
        uint32_t maxAbsLevel = (uint32_t)((levelDouble + ((int32_t)1 << (qbits - 1))) >> qbits);
        if (maxAbsLevel > 32767u) maxAbsLevel = 32767u;
        const double err = (double)levelDouble;
        const double c0 = err * err * errScale;
        const int p4 = (int)((scan4 >> (4 * posInCG)) & 15);
        int ctxSig;
        if (log2 == 2) ctxSig = sigOff + (p4 ? (int)((map4 >> (4 * p4)) & 15) : 0);
        else if (scanPos == 0) ctxSig = sigOff;
        else ctxSig = sigBase + (int)((cntBits >> (2 * p4)) & 3);
        const int sbit0 = cb(ctxSig, 0), sbit1 = cb(ctxSig, 1);
        vLd.set(lane, levelDouble); vMax.set(lane, (int)maxAbsLevel); vSb0.set(lane, sbit0); vSb1.set(lane, sbit1);
        vC0.set(lane, c0); vCand.set(lane, maxAbsLevel != 0);
        /* xGetCodedLevel's early exit (:2752-2760) for uiMaxAbsLevel 0: the whole record but rateIncUp, which follows the greater1 state */
        const double cs = lambda * (double)sbit0;
        vCs.set(lane, cs); vCc.set(lane, c0 + cs); vSd.set(lane, sbit1 - sbit0); vDu.set(lane, (int32_t)(levelDouble >> (qbits - 8)));
        vUp.set(lane, 0); vDn.set(lane, 0); vLv.set(lane, 0);
      }
    }
    const uint32_t cand = lane_mask16(vCand);
    double rdSigCost = 0, rdSigCost0 = 0, rdCodedLevelandDist = 0, rdUncodedDist = 0; int nnzBeforePos0 = 0;
    double cgSig = 0;                                          /* costCGSig of this group */
    for (int posInCG = 15; posInCG >= 0; posInCG--) {
      const int scanPos = cgScanPos * 16 + posInCG;
      const int isCand = (int)((cand >> posInCG) & 1);
      double c0 = vC0.get(posInCG), cc = 0, cs = 0;
      blockUncodedCost += c0;
      uint32_t level = 0;
      if (isCand && lastScanPos < 0) {
        lastScanPos = scanPos;
        ctxSet = (uint32_t)((ch ? 4 : 0) + ((!ch && (scanPos >> 4) > 0) ? 2 : 0));
        cgLastScanPos = cgScanPos;
      }
      if (lastScanPos >= 0) {
        const int oneCtx = CTX_ONE + 4 * (int)ctxSet + c1;
        if (oneCtx != g10Ctx) { g10 = cb(oneCtx, 0); g10Ctx = oneCtx; }
        if (!isCand) {                                         /* level 0 leaves c1/c2/Rice state alone */
          cs = vCs.get(posInCG); cc = vCc.get(posInCG);
          vUp.put(posInCG, g10);
          baseCost += cc;
          rdSigCost += cs;
          if (posInCG == 0) rdSigCost0 = cs;
          continue;
        }
        const int32_t levelDouble = vLd.get(posInCG); const uint32_t maxAbsLevel = (uint32_t)vMax.get(posInCG);
        const int sbit0 = vSb0.get(posInCG), sbit1 = vSb1.get(posInCG);
        auto cbs = [&](int, int bin) -> int { return bin ? sbit1 : sbit0; };   /* the significance rates of this coefficient, looked up by its lane */
        int sdel = 0, rup, rdn = 0;
        const int absCtx = CTX_ABS + (int)ctxSet + c2;
        LevelBits lb; lb.g10 = g10; lb.g11 = cb(oneCtx, 1); lb.g20 = cb(absCtx, 0); lb.g21 = cb(absCtx, 1);
        int rateHi = 0, rateLo = 0;                            /* xGetICRate(maxAbsLevel), (maxAbsLevel - 1) from the level choice */
        if (scanPos == lastScanPos)
          level = coded_level(cbs, lambda, &cc, &c0, &cs, levelDouble, maxAbsLevel,
                              0, lb, goRice, c1Idx, c2Idx, qbits, errScale, 1, &rateHi, &rateLo);
        else {
          level = coded_level(cbs, lambda, &cc, &c0, &cs, levelDouble, maxAbsLevel,
                              0, lb, goRice, c1Idx, c2Idx, qbits, errScale, 0, &rateHi, &rateLo);
          sdel = sbit1 - sbit0;
        }
        if (level > 0) {                                       /* the chosen level is maxAbsLevel or maxAbsLevel - 1: one new rate, not three */
          if (level == maxAbsLevel) {
            rup = ic_rate(lb, level + 1, goRice, c1Idx, c2Idx) - rateHi;
            rdn = (maxAbsLevel > 1 ? rateLo : ic_rate(lb, 0, goRice, c1Idx, c2Idx)) - rateHi;
          } else {
            rup = rateHi - rateLo;
            rdn = ic_rate(lb, level - 1, goRice, c1Idx, c2Idx) - rateLo;
          }
        } else rup = lb.g10;
        vCc.put(posInCG, cc); vCs.put(posInCG, cs); vC0.put(posInCG, c0); vUp.put(posInCG, rup); vDn.put(posInCG, rdn); vSd.put(posInCG, sdel);
        vDu.put(posInCG, (int32_t)((levelDouble - ((int32_t)level << qbits)) >> (qbits - 8)));
        baseCost += cc;
        const uint32_t baseLevel = (c1Idx < 8) ? (2 + (c2Idx < 1)) : 1;
        if (level >= baseLevel) { if (level > 3u * (1u << goRice)) goRice = goRice + 1 < 4 ? goRice + 1 : 4; }
        if (level >= 1) c1Idx++;
        if (level > 1) { c1 = 0; c2 += (c2 < 2); c2Idx++; }
        else if (c1 < 3 && c1 > 0 && level) c1++;
      } else baseCost += c0;
      vLv.put(posInCG, (int)level);
      rdSigCost += cs;
      if (posInCG == 0) rdSigCost0 = cs;
      if (level) {
        cgflag |= 1ull << cgBlk;
        rdCodedLevelandDist += cc - cs;
        rdUncodedDist += c0;
        if (posInCG != 0) nnzBeforePos0++;
      }
    }
    if (lastScanPos >= 0 && cgScanPos > 0) {                   /* context set of the next group (:2306-2316), once the last position is known */
      ctxSet = (uint32_t)((ch ? 4 : 0) + ((!ch && (cgScanPos - 1) > 0) ? 2 : 0) + (c1 == 0));
      c1 = 1; c2 = 0; c1Idx = 0; c2Idx = 0; goRice = 0;
    }
    int zeroed = 0;
    if (cgLastScanPos >= 0) {
      if (cgScanPos) {
        if (((cgflag >> cgBlk) & 1) == 0) {
          const int ctxSig = cgBase + sig_cg_ctx(cgflag, cgx, cgy, wg);
          baseCost += lambda * (double)cb(ctxSig, 0) - rdSigCost;
          cgSig = lambda * (double)cb(ctxSig, 0);
        } else if (cgScanPos < cgLastScanPos) {
          if (nnzBeforePos0 == 0) { baseCost -= rdSigCost0; rdSigCost -= rdSigCost0; }
          double costZeroCG = baseCost;
          const int ctxSig = cgBase + sig_cg_ctx(cgflag, cgx, cgy, wg);
          baseCost += lambda * (double)cb(ctxSig, 1);
          costZeroCG += lambda * (double)cb(ctxSig, 0);
          cgSig = lambda * (double)cb(ctxSig, 1);
          costZeroCG += rdUncodedDist; costZeroCG -= rdCodedLevelandDist; costZeroCG -= rdSigCost;
          if (costZeroCG < baseCost) {
            cgflag &= ~(1ull << cgBlk); baseCost = costZeroCG;
            cgSig = lambda * (double)cb(ctxSig, 0);
            zeroed = 1;                                        /* the group's levels go to 0 (applied by their lanes below) */
          }
        }
      } else cgflag |= 1ull << cgBlk;
    }
    /* records and levels of the group to the pools.  The passes that follow read the records of groups that kept a level only
     * (last position: flagged groups; sign hiding: groups with two levels four positions apart): a group without one stores
     * its sixteen zero levels and nothing else */
    const int keepsLevel = (int)((cgflag >> cgBlk) & 1);
    FCU_FOR_LANES {
      if (lane < 16) {
        const int sp = cgScanPos * 16 + lane;
        if (keepsLevel) {
          RdoqRec r; r.cc = vCc.own(lane); r.cs = vCs.own(lane); r.c0 = vC0.own(lane); r.up = vUp.own(lane); r.dn = vDn.own(lane); r.sd = vSd.own(lane); r.du = vDu.own(lane);
          recg[sp] = r;
        }
        dstg[sp] = (int16_t)(zeroed ? 0 : vLv.own(lane));
      }
      if (lane == 0) cgg[cgScanPos] = cgSig;
    }
    vCgSig.put(cgScanPos, cgSig);
  }
  if (lastScanPos < 0) { FCU_SERIAL { g_S.rw_abs = 0; g_S.rw_lsp = -1; } return; }
  /* rdoq_last_pos as a wave: per group that kept a level, lanes fetch level and record of a position each and price it as the
   * last one (xGetRateLast); the walk over the positions -- the running cost without the coefficients above -- is wave-uniform
   * code on those numbers, same expressions in the same order */
  int bestLastIdxP1 = 0;
  {
    double bestCost = blockUncodedCost + lambda * (double)cb(cbfCtx, 0);
    baseCost += lambda * (double)cb(cbfCtx, 1);
    const int lcc = log2 - 2, loff = ch ? 0 : (lcc * 3 + ((lcc + 1) >> 2)), lsh = ch ? lcc : ((lcc + 3) >> 2);
    const int lbx = CTX_LASTX + (ch ? 15 : 0) + loff, lby = CTX_LASTY + (ch ? 15 : 0) + loff, lgmax = g_hot.group_idx[N - 1];
    int foundLast = 0;
    for (int cgScanPos = cgLastScanPos; cgScanPos >= 0 && !foundLast; cgScanPos--) {
      const int cgBlk = vCgBlk.get(cgScanPos);
      baseCost -= vCgSig.get(cgScanPos);
      if (!((cgflag >> cgBlk) & 1)) continue;
      const int top = (cgScanPos * 16 + 15 > lastScanPos) ? lastScanPos - cgScanPos * 16 : 15;   /* positions above the last one are skipped */
      LaneVar<int> wLv; LaneVar<double> wCs, wCc, wC0, wCl;
      FCU_FOR_LANES {
        if (lane <= top) {
          const int scanPos = cgScanPos * 16 + lane;
          const int lv = dstg[scanPos];
          const RdoqRec q = recg[scanPos];
          wLv.set(lane, lv); wCs.set(lane, q.cs); wCc.set(lane, q.cc); wC0.set(lane, q.c0);
          double costLast = 0;
          if (lv) {
            const int blk = scan[scanPos];
            const int py = blk >> log2, px = blk - (py << log2);
            const int ax = scanType == 2 ? py : px, ay = scanType == 2 ? px : py;
            const int gx = g_hot.group_idx[ax], gy = g_hot.group_idx[ay];
            int bxs = 0, bys = 0;
            for (int k = 0; k < gx; k++) bxs += cb(lbx + (k >> lsh), 1);
            if (gx < lgmax) bxs += cb(lbx + (gx >> lsh), 0);
            for (int k = 0; k < gy; k++) bys += cb(lby + (k >> lsh), 1);
            if (gy < lgmax) bys += cb(lby + (gy >> lsh), 0);
            double r = (double)(bxs + bys);                                  /* xGetRateLast, TComTrQuant.cpp:2898-2916 */
            if (gx > 3) r += 32768.0 * (double)((gx - 2) >> 1);
            if (gy > 3) r += 32768.0 * (double)((gy - 2) >> 1);
            costLast = lambda * r;
          }
          wCl.set(lane, costLast);
        }
      }
      for (int posInCG = top; posInCG >= 0; posInCG--) {
        const int lv = wLv.get(posInCG); const double curCs = wCs.get(posInCG);
        if (lv) {
          const double totalCost = baseCost + wCl.get(posInCG) - curCs;
          if (totalCost < bestCost) { bestLastIdxP1 = cgScanPos * 16 + posInCG + 1; bestCost = totalCost; }
          if (lv > 1) { foundLast = 1; break; }
          baseCost -= wCc.get(posInCG); baseCost += wC0.get(posInCG);
        } else baseCost -= curCs;
      }
    }
  }
  FCU_SERIAL { g_S.rw_abs = 0; }
  FCU_FOR_LANES {                                              /* rdoq_signs with a position per lane */
    uint32_t acc = 0;
    for (int sp = lane; sp <= lastScanPos; sp += 64) {
      if (sp < bestLastIdxP1) { const int lv = dstg[sp]; acc += (uint32_t)lv; dstg[sp] = (int16_t)((srcg[sp] < 0) ? -lv : lv); }
      else dstg[sp] = 0;
    }
    FCU_WAVE_ADD((uint32_t *)&g_S.rw_abs, acc);
  }
  FCU_SERIAL {
    const int absSum = g_S.rw_abs;
    rdoq_sign_hiding(P, srcg, dstg, recg, 1, ch, bestLastIdxP1, absSum);
    int last = bestLastIdxP1 - 1;                              /* sign hiding may have cleared the last level */
    while (last >= 0 && dstg[last] == 0) last--;
    g_S.rw_lsp = last;
  }
}

/* ======================================================================================== */
/* intra prediction: closed form per pixel (TComPrediction.cpp:183-496,755-841)              */
/* ======================================================================================== */
FCU_DEV int use_filtered_ref(int mode, int log2, int isLuma)               /* TComPattern.cpp:523-548 */
{
  if (!isLuma || mode == DC) return 0;
  const int d1 = iabs(mode - HOR), d2 = iabs(mode - VER);
  return (d1 < d2 ? d1 : d2) > k_filter_thr[log2 - 2];
}
/* ref[0..4N]: bottom-left ... corner(2N) ... top-right.  corner[+i] = top(i-1), corner[-i] = left(i-1) */
FCU_DEV int pred_pixel(const uint8_t *ref, int log2, int mode, int isLuma, int dc, int x, int y)
{
  const int N = 1 << log2;
  const uint8_t *corner = ref + 2 * N;
  if (mode == PLANAR)
    return ((N - 1 - x) * corner[-1 - y] + (x + 1) * corner[1 + N] + (N - 1 - y) * corner[1 + x] + (y + 1) * corner[-1 - N] + N) >> (log2 + 1);
  if (mode == DC) {
    int v = dc;
    if (isLuma && N <= 16) {
      if (x == 0 && y == 0) v = (corner[1] + corner[-1] + 2 * dc + 2) >> 2;
      else if (y == 0) v = (corner[1 + x] + 3 * dc + 2) >> 2;
      else if (x == 0) v = (corner[-1 - y] + 3 * dc + 2) >> 2;
    }
    return v;
  }
  const int isVer = mode >= 18;
  const int angMode = isVer ? mode - VER : -(mode - HOR);
  /* intraPredAngle / invAngle (TComPrediction.cpp:287-288) packed into constants: 9 x 6 bits, 8 x 13 bits (no table loads per pixel) */
  const int absAng = iabs(angMode);
  const int angAbs = (int)((0x2069544d245080ull >> (6 * absAng)) & 63);
  const int invAngle = absAng == 0 ? 0 : (int)(((absAng <= 4 ? 0x13b0e38ccd000ull : 0x8004ec30c1e2ull) >> (13 * ((absAng - 1) & 3))) & 8191);
  const int angle = angMode < 0 ? -angAbs : angAbs;
  const int px = isVer ? x : y, py = isVer ? y : x, ms = isVer ? 1 : -1;   /* main array direction from the corner */
  if (angle == 0) {
    int v = corner[ms * (px + 1)];
    if (isLuma && N <= 16 && px == 0) v = clip8(v + ((corner[-ms * (py + 1)] - corner[0]) >> 1));
    return v;
  }
  const int deltaPos = (py + 1) * angle, di = deltaPos >> 5, df = deltaPos & 31;
  const int i0 = px + di + 1;
  int r0, r1;
  { int i = i0; r0 = (i >= 0) ? corner[ms * i] : corner[-ms * ((128 + (-i) * invAngle) >> 8)]; }
  if (!df) return r0;
  { int i = i0 + 1; r1 = (i >= 0) ? corner[ms * i] : corner[-ms * ((128 + (-i) * invAngle) >> 8)]; }
  return ((32 - df) * r0 + df * r1 + 16) >> 5;
}

/* z-scan availability (H.265 6.4.1 == getPULeft/Above/.../BelowLeftAdi, TComDataCU.cpp:1071-1390) */
FCU_DEV int unit_available(const Env E, int lx, int ly, int cx, int cy)
{
  const Params &P = E.C->p;
  if (lx < 0 || ly < 0 || lx >= P.width || ly >= P.height) return 0;
  const int ctuN = (ly >> 6) * E.C->w_ctu + (lx >> 6), ctuC = (cy >> 6) * E.C->w_ctu + (cx >> 6);
  if (ctuN < E.slice_start) return 0;
  if (ctuN < ctuC) return 1;
  if (ctuN > ctuC) return 0;
  return zidx_of(lx, ly) < zidx_of(cx, cy);
}

/* Reference samples of a block (initAdiPatternChType + fillReferenceSamples + smoothing,
 * TComPattern.cpp:104-521) -> g_S.ref (unfiltered), g_S.reff (filtered, luma only), g_S.dc. */
FCU_DEV FCU_NOINLINE void build_ref(int comp, int px, int py, int log2, int wantFilt)
{
  const Env E = env_get(); comp = FCU_UNI(comp); px = FCU_UNI(px); py = FCU_UNI(py); log2 = FCU_UNI(log2); wantFilt = FCU_UNI(wantFilt);
  const int N = 1 << log2, sh = comp ? 1 : 0, unit = 4 >> sh, total = 4 * N + 1;
  const int lx0 = px << sh, ly0 = py << sh;
  const uint8_t *rec = E.C->rec[comp]; const int stride = E.C->stride[comp];
  uint8_t *ref = g_S.ref, *avail = (uint8_t *)g_S.colsum;    /* availability flags staged in colsum (>= 257 bytes) */
  FCU_FOR_LANES {
    for (int i = lane; i < total; i += 64) {
      int a, v = 0;
      if (i < 2 * N) { const int y = 2 * N - 1 - i; a = unit_available(E, lx0 - 4, ((py + y) / unit * unit) << sh, lx0, ly0); if (a) v = rec[(py + y) * stride + px - 1]; }
      else if (i == 2 * N) { a = unit_available(E, lx0 - 4, ly0 - 4, lx0, ly0); if (a) v = rec[(py - 1) * stride + px - 1]; }
      else { const int x = i - 2 * N - 1; a = unit_available(E, ((px + x) / unit * unit) << sh, ly0 - 4, lx0, ly0); if (a) v = rec[(py - 1) * stride + px + x]; }
      avail[i] = (uint8_t)a; ref[i] = (uint8_t)v;
    }
  }
  /* substitution: every unavailable sample takes the nearest available one below it in walk
   * order, or (for the leading run) the first available one */
  FCU_FOR_LANES {
    for (int i = lane; i < total; i += 64) {
      if (!avail[i]) {
        int j = i - 1;
        while (j >= 0 && !avail[j]) j--;
        if (j < 0) { j = i + 1; while (j < total && !avail[j]) j++; }
        g_S.reff[i] = (j < total) ? ref[j] : 128;            /* staged in reff, merged below */
      }
    }
  }
  FCU_FOR_LANES { for (int i = lane; i < total; i += 64) if (!avail[i]) ref[i] = g_S.reff[i]; }
  FCU_FOR_LANES {
    if (lane == 0) { int sum = 0; for (int i = 0; i < N; i++) sum += ref[2 * N + 1 + i] + ref[2 * N - 1 - i]; g_S.dc = (sum + N) >> (log2 + 1); }
    if (wantFilt) {
      int strong = 0;
      if (comp == 0 && E.C->p.strong_smoothing && N >= 32) {
        const int bl = ref[0], tl = ref[2 * N], tr = ref[4 * N];
        strong = (iabs(bl + tl - 2 * ref[N]) < 8) && (iabs(tl + tr - 2 * ref[3 * N]) < 8);
      }
      for (int i = lane; i < total; i += 64) {
        int v;
        if (i == 0 || i == 4 * N) v = ref[i];
        else if (strong) {
          const int bl = ref[0], tl = ref[2 * N], tr = ref[4 * N];
          if (i < 2 * N) v = ((2 * N - i) * bl + i * tl + N) >> (log2 + 1);
          else if (i == 2 * N) v = tl;
          else { const int k = i - 2 * N; v = ((2 * N - k) * tl + k * tr + N) >> (log2 + 1); }
        } else v = (ref[i - 1] + 2 * ref[i] + ref[i + 1] + 2) >> 2;
        g_S.reff[i] = (uint8_t)v;
      }
    }
  }
}

/* ======================================================================================== */
/* transforms: one output coefficient per call (xTrMxN / xITrMxN, TComTrQuant.cpp:860-987;    */
/* the partial butterflies are exact factorizations of these products)                       */
/* ======================================================================================== */
/* Every stage is a dot product of one basis row with one CONTIGUOUS data row, unrolled per size so that the loads
 * of a row are issued together (wide loads) instead of one dependent load per multiply-add:
 *   fwd1: tmp[k*N + y]   = sum_n T[k][n]  * resi[y*N + n]          fwd2: coef[k2*N + k1] = sum_y T[k2][y] * tmp[k1*N + y]
 *   inv1: tmp2[y*N + kh] = sum_kv Tt[y][kv] * deqT[kh*N + kv]      inv2: resi[y*N + x]   = sum_kh Tt[x][kh] * tmp2[y*N + kh]
 * (Tt = transposed basis; deqT = de-quantised coefficients stored transposed by the caller, see tr_index). */
template <int N, class T>
FCU_DEV int32_t dot_row(const int8_t *m, const T *a)
{
  m = (const int8_t *)__builtin_assume_aligned(m, N >= 16 ? 16 : N);
  a = (const T *)__builtin_assume_aligned(a, sizeof(T) * N >= 16 ? 16 : 8);
  uint32_t mw[N / 4];                                         /* four basis values per word: N/4 (wide) loads */
  __builtin_memcpy(mw, m, N);
  int32_t s = 0;
#pragma unroll
  for (int n = 0; n < N; n++) s += (int32_t)(int8_t)(mw[n >> 2] >> (8 * (n & 3))) * (int32_t)a[n];
  return s;
}
/* The input line of a lane's dot products, held in registers across the iterations of an element loop.  Every caller walks
 * its outputs with a stride of 64 lanes, and 64 is a multiple of N: inside one block a lane keeps meeting the same input
 * line (N of them per block, every one feeding N outputs).  Re-reading the line per output made each pass stream its whole
 * N x N tile N * N / 64 times through L1 / L2 -- 16 x for 32x32 -- and with thousands of resident chains that stream lands
 * in HBM.  A lane (re)loads its line only when the line's address changes (next candidate block).  The input arrays are
 * never written during the phase that reads them, so the copy cannot go stale; a cache lives for one phase only. */
template <int N, class T> struct RowCache {
  T v[N]; const T *src;
  FCU_MEMBER RowCache() : src(nullptr) {}
  FCU_MEMBER const T *get(const T *p)
  {
    if (p != src) { src = p; __builtin_memcpy(v, (const T *)__builtin_assume_aligned(p, sizeof(T) * N >= 16 ? 16 : 8), sizeof(v)); }
    return v;
  }
};
template <int N, class T>
FCU_DEV int32_t dot_regs(const int8_t *m, const T *a)         /* a: a RowCache line (registers) */
{
  m = (const int8_t *)__builtin_assume_aligned(m, N >= 16 ? 16 : N);
  uint32_t mw[N / 4];
  __builtin_memcpy(mw, m, N);
  int32_t s = 0;
#pragma unroll
  for (int n = 0; n < N; n++) s += (int32_t)(int8_t)(mw[n >> 2] >> (8 * (n & 3))) * (int32_t)a[n];
  return s;
}
/* size dispatch OUTSIDE the element loops: f(std::integral_constant<int, log2>) */
template <class F>
FCU_DEV void by_log2(int log2, F f)
{
  switch (log2) {
    case 2: f(std::integral_constant<int, 2>()); break;
    case 3: f(std::integral_constant<int, 3>()); break;
    case 4: f(std::integral_constant<int, 4>()); break;
    default: f(std::integral_constant<int, 5>()); break;
  }
}
template <int LOG2> FCU_DEV const int8_t *basis_row(int useDst, int k) { return (useDst && LOG2 == 2) ? k_dst4 + k * 4 : k_dct + k_dct_off[LOG2 - 2] + (k << LOG2); }
template <int LOG2> FCU_DEV const int8_t *basis_col(int useDst, int n) { return (useDst && LOG2 == 2) ? k_dst4_t + n * 4 : k_dct_t + k_dct_off[LOG2 - 2] + (n << LOG2); }
FCU_DEV int tr_index(int i, int log2) { const int N = 1 << log2; return ((i & (N - 1)) << log2) + (i >> log2); }   /* raster index of the transposed block */
template <int LOG2> FCU_DEV int32_t fwd1(const int16_t *resi, int useDst, int idx)
{ constexpr int N = 1 << LOG2, s1 = LOG2 - 1; const int k = idx >> LOG2, y = idx & (N - 1); return (dot_row<N>(basis_row<LOG2>(useDst, k), resi + (y << LOG2)) + (1 << (s1 - 1))) >> s1; }
template <int LOG2> FCU_DEV int32_t fwd2(const int32_t *tmp, int useDst, int idx)
{ constexpr int N = 1 << LOG2, s2 = LOG2 + 6; const int k2 = idx >> LOG2, k1 = idx & (N - 1); return (dot_row<N>(basis_row<LOG2>(useDst, k2), tmp + (k1 << LOG2)) + (1 << (s2 - 1))) >> s2; }
template <int LOG2> FCU_DEV int32_t inv1(const int32_t *deqT, int useDst, int idx)
{ constexpr int N = 1 << LOG2; const int y = idx >> LOG2, kh = idx & (N - 1); return clip3i(-32768, 32767, (dot_row<N>(basis_col<LOG2>(useDst, y), deqT + (kh << LOG2)) + 64) >> 7); }
template <int LOG2> FCU_DEV int32_t inv2(const int32_t *tmp2, int useDst, int idx)
{ constexpr int N = 1 << LOG2; const int y = idx >> LOG2, x = idx & (N - 1); return clip3i(-32768, 32767, (dot_row<N>(basis_col<LOG2>(useDst, x), tmp2 + (y << LOG2)) + 2048) >> 12); }
/* three of the four passes with the lane's input line cached (declare one RowCache per element loop) */
template <int LOG2> FCU_DEV int32_t fwd1(RowCache<(1 << LOG2), int16_t> &rc, const int16_t *resi, int useDst, int idx)
{ constexpr int N = 1 << LOG2, s1 = LOG2 - 1; const int k = idx >> LOG2, y = idx & (N - 1); return (dot_regs<N>(basis_row<LOG2>(useDst, k), rc.get(resi + (y << LOG2))) + (1 << (s1 - 1))) >> s1; }
template <int LOG2> FCU_DEV int32_t fwd2(RowCache<(1 << LOG2), int32_t> &rc, const int32_t *tmp, int useDst, int idx)
{ constexpr int N = 1 << LOG2, s2 = LOG2 + 6; const int k2 = idx >> LOG2, k1 = idx & (N - 1); return (dot_regs<N>(basis_row<LOG2>(useDst, k2), rc.get(tmp + (k1 << LOG2))) + (1 << (s2 - 1))) >> s2; }
template <int LOG2> FCU_DEV int32_t inv1(RowCache<(1 << LOG2), int32_t> &rc, const int32_t *deqT, int useDst, int idx)
{ constexpr int N = 1 << LOG2; const int y = idx >> LOG2, kh = idx & (N - 1); return clip3i(-32768, 32767, (dot_regs<N>(basis_col<LOG2>(useDst, y), rc.get(deqT + (kh << LOG2))) + 64) >> 7); }
/* (inv2's input line follows the high index bits: neighbouring lanes share it, one broadcast read per iteration) */
struct DeqParams { int scale, rs, lo, hi; };                /* xDeQuant flat, TComTrQuant.cpp:1242-1352: loop invariants of one TU */
FCU_DEV DeqParams deq_params(int log2, int qp)
{
  DeqParams d;
  const int per = qp / 6, rem = qp % 6;
  d.rs = 6 - ((15 - 8 - log2) + per); d.scale = k_inv_quant_scales[rem];
  int tbd = 32 + d.rs - 7; if (tbd > 16) tbd = 16;
  d.lo = -(1 << (tbd - 1)); d.hi = (1 << (tbd - 1)) - 1;
  return d;
}
FCU_DEV int32_t dequant1(int q, const DeqParams &d)
{
  const int c = clip3i(d.lo, d.hi, q);
  if (d.rs > 0) return clip3i(-32768, 32767, (c * d.scale + (1 << (d.rs - 1))) >> d.rs);
  return clip3i(-32768, 32767, (c * d.scale) << (-d.rs));
}

/* ======================================================================================== */
/* neighbour decisions (MPM, split context)                                                  */
/* ======================================================================================== */
FCU_DEV int inside_cu(const CuObj *cu, int lx, int ly) { const int s = CTU >> cu->depth_cu; return lx >= cu->x && lx < cu->x + s && ly >= cu->y && ly < cu->y + s; }
FCU_DEV int nb_depth(const Env E, const CuObj *cu, int lx, int ly)
{ return inside_cu(cu, lx, ly) ? cu->depth[zidx_of(lx, ly) - cu->zidx] : E.C->out[(ly >> 6) * E.C->w_ctu + (lx >> 6)].depth[zidx_of(lx, ly)]; }
FCU_DEV int nb_luma_dir(const Env E, const CuObj *cu, int lx, int ly)
{
  if (inside_cu(cu, lx, ly)) { const int p = zidx_of(lx, ly) - cu->zidx; return cu->pred_mode[p] == MODE_INTRA ? cu->intra_dir[0][p] : DC; }
  const fcu_ctu_out *c = &E.C->out[(ly >> 6) * E.C->w_ctu + (lx >> 6)]; const int p = zidx_of(lx, ly);
  return c->pred_mode[p] == MODE_INTRA ? c->intra_dir[0][p] : DC;
}
FCU_DEV int left_ctu_ok(const Env E, int lx, int ly) { if (lx == 0) return 0; if (lx & 63) return 1; return (ly >> 6) * E.C->w_ctu + (lx >> 6) - 1 >= E.slice_start; }
FCU_DEV int above_ctu_ok(const Env E, int lx, int ly) { if (ly == 0) return 0; if (ly & 63) return 1; return (ly >> 6) * E.C->w_ctu + (lx >> 6) - E.C->w_ctu >= E.slice_start; }
/* getIntraDirPredictor, TComDataCU.cpp:1542-1624 */
FCU_DEV int intra_dir_predictor(const Env E, const CuObj *cu, int part, int *preds)
{
  const int z = cu->zidx + part, lx = (cu->x & ~63) + part_x(z), ly = (cu->y & ~63) + part_y(z);
  const int left = left_ctu_ok(E, lx, ly) ? nb_luma_dir(E, cu, lx - 1, ly) : DC;
  const int above = ((ly & 63) != 0) ? nb_luma_dir(E, cu, lx, ly - 1) : DC;
  if (left == above) {
    if (left > 1) { preds[0] = left; preds[1] = ((left + 29) % 32) + 2; preds[2] = ((left - 1) % 32) + 2; }
    else { preds[0] = PLANAR; preds[1] = DC; preds[2] = VER; }
    return 1;
  }
  preds[0] = left; preds[1] = above;
  preds[2] = (left && above) ? PLANAR : ((left + above) < 2 ? VER : DC);
  return 2;
}
FCU_DEV int min_tu_log2_in_cu(int depth, int partSize)      /* getQuadtreeTULog2MinSizeInCU, TComDataCU.cpp:1658-1686 */
{
  const int log2Cb = 6 - depth, split = partSize == SIZE_NxN;
  if (log2Cb < LOG2_MINTU + TU_MAXDEPTH_INTRA - 1 + split) return LOG2_MINTU;
  const int m = log2Cb - (TU_MAXDEPTH_INTRA - 1 + split);
  return m > LOG2_MAXTU ? LOG2_MAXTU : m;
}

#include "fcu_inter.h"

/* ---- syntax element coders (serial, any Cabac) ------------------------------------------ */
FCU_DEV void code_split_flag(const Env E, int c, const CuObj *cu, int part, int depth)   /* TEncSbac.cpp:613-628 */
{
  if (depth == MAXDEPTH) return;
  const int z = cu->zidx + part, lx = (cu->x & ~63) + part_x(z), ly = (cu->y & ~63) + part_y(z);
  int ctx = 0;
  if (left_ctu_ok(E, lx, ly)) ctx += nb_depth(E, cu, lx - 1, ly) > depth;
  if (above_ctu_ok(E, lx, ly)) ctx += nb_depth(E, cu, lx, ly - 1) > depth;
  cab_bin(c, cu->depth[part] > depth, CTX_SPLIT + ctx);
}
/* codeIntraDirLumaAng for one PU with known MPM list (TEncSbac.cpp:643-696) */
FCU_DEV void code_luma_dir_bits(int c, int dir, const int *preds)
{
  int predIdx = -1;
  for (int i = 0; i < 3; i++) if (dir == preds[i]) predIdx = i;
  cab_bin(c, predIdx != -1, CTX_INTRA_LUMA);
  cab_ep(c, predIdx != -1 ? (predIdx ? 2 : 1) : 5);
}
FCU_DEV FCU_NOINLINE void code_intra_dir_luma(int c, const CuObj *cu, int part, int multiple)
{
  const Env E = env_get(); c = FCU_UNI(c); cu = FCU_UNI(cu); part = FCU_UNI(part); multiple = FCU_UNI(multiple);
  int preds[4][3], predIdx[4];
  const int partNum = multiple ? (cu->part_size[part] == SIZE_NxN ? 4 : 1) : 1;
  const int partOffset = (NPART >> (cu->depth[part] << 1)) >> 2;
  for (int j = 0; j < partNum; j++) {
    const int dir = cu->intra_dir[0][part + partOffset * j];
    intra_dir_predictor(E, cu, part + partOffset * j, preds[j]);
    predIdx[j] = -1;
    for (int i = 0; i < 3; i++) if (dir == preds[j][i]) predIdx[j] = i;
    cab_bin(c, predIdx[j] != -1, CTX_INTRA_LUMA);
  }
  for (int j = 0; j < partNum; j++) cab_ep(c, predIdx[j] != -1 ? (predIdx[j] ? 2 : 1) : 5);
}
FCU_DEV void code_intra_dir_chroma(int c, int dir)       /* TEncSbac.cpp:698-725 */
{ if (dir == DM_CHROMA) cab_bin(c, 0, CTX_CHROMA_PRED); else { cab_bin(c, 1, CTX_CHROMA_PRED); cab_ep(c, 2); } }
FCU_DEV int chroma_final_mode(const CuObj *cu, int part) { int m = cu->intra_dir[1][part]; return m == DM_CHROMA ? cu->intra_dir[0][part & ~3] : m; }

/* search-time tree walkers: xEncSubdivCbfQT / xEncCoeffQT / xGetIntraBitsQT, TEncSearch.cpp:866-1090 */
FCU_DEV FCU_NOINLINE void enc_subdiv_cbf_qt(int c, const CuObj *cu, uint32_t root_k, int bLuma, int bChroma)
{
  const Env E = env_get(); c = FCU_UNI(c); cu = FCU_UNI(cu); const TU root = tu_of_key(FCU_UNI(root_k)); bLuma = FCU_UNI(bLuma); bChroma = FCU_UNI(bChroma);
  TU *st = g_S.wk_st; int *ci = g_S.wk_ci; int sp = 0;
  st[0] = root; ci[0] = -1;
  while (sp >= 0) {
    const TU tu = st[sp];
    if (ci[sp] < 0) {
      const int subdiv = cu->tr_idx[tu.part] > tu.tr_depth;
      if (cu->part_size[0] == SIZE_NxN && tu.tr_depth == 0) { }
      else if (tu.log2 > LOG2_MAXTU) { }
      else if (tu.log2 == LOG2_MINTU) { }
      else if (tu.log2 == min_tu_log2_in_cu(cu->depth[tu.part], cu->part_size[tu.part])) { }
      else if (bLuma) cab_bin(c, subdiv, CTX_SUBDIV + 5 - tu.log2);
      if (bChroma)
        for (int comp = 1; comp < 3; comp++)
          if (tu.c_code_all && (tu.tr_depth == 0 || ((cu->cbf[comp][tu.part] >> (tu.tr_depth - 1)) & 1))) {
            const int lowest = tu.tr_depth + ((subdiv && !(tu.cw >= 8)) ? 1 : 0);
            cab_bin(c, (cu->cbf[comp][tu_part_c(tu)] >> lowest) & 1, CTX_CBF_CHROMA + tu.tr_depth);
          }
      if (!subdiv) { if (bLuma) cab_bin(c, (cu->cbf[0][tu.part] >> tu.tr_depth) & 1, CTX_CBF_LUMA + (tu.tr_depth == 0 ? 1 : 0)); sp--; continue; }
      ci[sp] = 0;
    }
    if (ci[sp] >= 4) { sp--; continue; }
    { TU ch; tu_child(ch, st[sp], ci[sp], 0); ci[sp]++; sp++; st[sp] = ch; ci[sp] = -1; }
  }
  (void)E;
}
FCU_DEV FCU_NOINLINE void enc_coeff_qt(int c, const CuObj *cu, uint32_t root_k, int comp, int realCoeff)
{
  const Env E = env_get(); c = FCU_UNI(c); cu = FCU_UNI(cu); const TU root = tu_of_key(FCU_UNI(root_k)); comp = FCU_UNI(comp); realCoeff = FCU_UNI(realCoeff);
  TU *st = g_S.wk_st; int *ci = g_S.wk_ci; int sp = 0;
  st[0] = root; ci[0] = -1;
  while (sp >= 0) {
    const TU tu = st[sp];
    if (ci[sp] < 0) {
      if (!(cu->tr_idx[tu.part] > tu.tr_depth)) {
        if (!(comp && tu.cw == 0) && ((cu->cbf[comp][tu.part] >> tu.tr_depth) & 1)) {
          const int layer = LOG2_MAXTU - tu.log2;
          const int16_t *buf = realCoeff ? cu->coef[comp] : E.G->qt_coef[comp][layer];
          const int log2 = comp ? ilog2(tu.cw) : tu.log2, pc = comp ? tu_part_c(tu) : tu.part;
          const int dir = comp ? chroma_final_mode(cu, pc) : cu->intra_dir[0][pc];
          code_coeff_nxn<1>(c, buf + (comp ? tu.off_c : tu.off_y), 1, -1, log2, comp, coef_scan_idx(dir, log2, comp), cu->tskip[comp][pc], E.C->p, g_S.lane_abs[0]);
        }
        sp--; continue;
      }
      ci[sp] = 0;
    }
    if (ci[sp] >= 4) { sp--; continue; }
    { TU ch; tu_child(ch, st[sp], ci[sp], 0); ci[sp]++; sp++; st[sp] = ch; ci[sp] = -1; }
  }
}
FCU_DEV void enc_intra_header(const Env E, int c, const CuObj *cu, int trDepth, int part, int bLuma, int bChroma)
{
  if (bLuma) {
    if (part == 0 && E.C->p.slice_type != SLICE_I) { code_skip_flag(E, c, cu, 0); code_pred_mode(c, cu, 0); }   /* TEncSearch.cpp:990-1000 */
    if (part == 0 && cu->depth[0] == MAXDEPTH) cab_bin(c, cu->part_size[0] == SIZE_2Nx2N, CTX_PARTSIZE);
    if (cu->part_size[0] == SIZE_2Nx2N) { if (part == 0) code_intra_dir_luma(c, cu, 0, 0); }
    else { const int q = cu->nparts >> 2; if (trDepth > 0 && (part & (q - 1)) == 0) code_intra_dir_luma(c, cu, part, 0); }
  }
  if (bChroma && part == 0) code_intra_dir_chroma(c, cu->intra_dir[1][part]);
}
FCU_DEV FCU_NOINLINE uint32_t intra_bits_qt(int c, const CuObj *cu, uint32_t tu_k, int bLuma, int bChroma)
{
  const Env E = env_get(); c = FCU_UNI(c); cu = FCU_UNI(cu); const TU tu = tu_of_key(FCU_UNI(tu_k)); bLuma = FCU_UNI(bLuma); bChroma = FCU_UNI(bChroma);
  cab_reset_bits(c);
  enc_intra_header(E, c, cu, tu.tr_depth, tu.part, bLuma, bChroma);
  enc_subdiv_cbf_qt(c, cu, tu_key(tu), bLuma, bChroma);
  if (bLuma) enc_coeff_qt(c, cu, tu_key(tu), 0, 0);
  if (bChroma) { enc_coeff_qt(c, cu, tu_key(tu), 1, 0); enc_coeff_qt(c, cu, tu_key(tu), 2, 0); }
  return cab_bits(c);
}

/* xGetIntraBitsQT (luma only) for the un-split TU that tu_trial() has just coded: the same bins as
 * intra_bits_qt() on a leaf, with the PU's MPM list from the RMD (g_S.preds, same intra_dir_predictor call)
 * and the levels still in scan order in the trial buffers */
FCU_DEV FCU_NOINLINE uint32_t leaf_luma_bits(int c, const CuObj *cu, uint32_t tu_k)
{
  const Env E = env_get(); c = FCU_UNI(c); cu = FCU_UNI(cu); const TU tu = tu_of_key(FCU_UNI(tu_k));
  const int part = tu.part, partSize = cu->part_size[0], log2 = tu.log2;
  cab_reset_bits(c);
  if (part == 0 && E.C->p.slice_type != SLICE_I) { code_skip_flag(E, c, cu, 0); code_pred_mode(c, cu, 0); }
  if (part == 0 && cu->depth[0] == MAXDEPTH) cab_bin(c, partSize == SIZE_2Nx2N, CTX_PARTSIZE);
  if (partSize == SIZE_2Nx2N ? (part == 0) : (tu.tr_depth > 0 && (part & ((cu->nparts >> 2) - 1)) == 0))
    code_luma_dir_bits(c, cu->intra_dir[0][part], g_S.preds);
  if (!(partSize == SIZE_NxN && tu.tr_depth == 0) && log2 != LOG2_MINTU && log2 != min_tu_log2_in_cu(cu->depth[part], cu->part_size[part]))
    cab_bin(c, 0, CTX_SUBDIV + 5 - log2);
  const int cbf = g_S.t_abs > 0;
  cab_bin(c, cbf, CTX_CBF_LUMA + (tu.tr_depth == 0 ? 1 : 0));
  if (cbf) code_coeff_body<1>(c, E.G->p_qscan, 1, g_S.t_lsp, log2, 0, coef_scan_idx(cu->intra_dir[0][part], log2, 0), cu->tskip[0][part], E.C->p, g_S.lane_abs[0]);
  return cab_bits(c);
}

/* final-order CU syntax: encodeCoeff / xEncodeTransform, TEncEntropy.cpp:201-400 */
FCU_DEV FCU_NOINLINE void encode_transform(int c, const CuObj *cu, int cuPart, uint32_t root_k)
{
  const Env E = env_get(); c = FCU_UNI(c); cu = FCU_UNI(cu); cuPart = FCU_UNI(cuPart); const TU root = tu_of_key(FCU_UNI(root_k));
  TU *st = g_S.wk_st; int *ci = g_S.wk_ci; int sp = 0;
  st[0] = root; ci[0] = -1;
  while (sp >= 0) {
    const TU tu = st[sp];
    if (ci[sp] < 0) {
      const int part = cuPart + tu.part, trIdx = tu.tr_depth, subdiv = cu->tr_idx[part] > trIdx;
      if (cu->part_size[part] == SIZE_NxN && trIdx == 0) { }
      else if (tu.log2 > LOG2_MAXTU) { }
      else if (tu.log2 == LOG2_MINTU) { }
      else if (tu.log2 == min_tu_log2_in_cu(cu->depth[part], cu->part_size[part])) { }
      else cab_bin(c, subdiv, CTX_SUBDIV + 5 - tu.log2);
      const int first = trIdx == 0;
      for (int comp = 1; comp < 3; comp++)
        if (first || tu.c_code_all)
          if (first || ((cu->cbf[comp][part] >> (trIdx - 1)) & 1)) {
            const int lowest = trIdx + ((subdiv && !(tu.cwo >= 8)) ? 1 : 0);
            cab_bin(c, (cu->cbf[comp][cuPart + tu_part_c(tu)] >> lowest) & 1, CTX_CBF_CHROMA + trIdx);
          }
      if (!subdiv) {
        cab_bin(c, (cu->cbf[0][part] >> trIdx) & 1, CTX_CBF_LUMA + (trIdx == 0 ? 1 : 0));
        for (int comp = 0; comp < 3; comp++) {
          if (comp && tu.cw == 0) continue;
          if (!((cu->cbf[comp][part] >> trIdx) & 1)) continue;
          const int log2 = comp ? ilog2(tu.cw) : tu.log2, pc = cuPart + (comp ? tu_part_c(tu) : tu.part);
          const int dir = comp ? chroma_final_mode(cu, pc) : cu->intra_dir[0][pc];
          const int16_t *coef = cu->coef[comp] + (comp ? (cuPart * 4 + tu.off_c) : (cuPart * 16 + tu.off_y));
          code_coeff_nxn<1>(c, coef, 1, -1, log2, comp, coef_scan_idx(dir, log2, comp), cu->tskip[comp][pc], E.C->p, g_S.lane_abs[0]);
        }
        sp--; continue;
      }
      ci[sp] = 0;
    }
    if (ci[sp] >= 4) { sp--; continue; }
    { TU ch; tu_child(ch, st[sp], ci[sp], 1); ci[sp]++; sp++; st[sp] = ch; ci[sp] = -1; }
  }
}
FCU_DEV void encode_cu_syntax(const Env E, int c, const CuObj *cu, int cuPart, int depth)   /* TEncCu.cpp:2117-2141 / 1753-1778 */
{
  if (E.C->p.slice_type != SLICE_I) {                        /* encodeSkipFlag / encodePredMode are no-ops in I slices */
    if (cu->pred_mode[cuPart] == MODE_INTER) { encode_cu_syntax_inter(E, c, cu, cuPart, depth); return; }
    code_skip_flag(E, c, cu, cuPart); code_pred_mode(c, cu, cuPart);
  }
  if (depth == MAXDEPTH) cab_bin(c, cu->part_size[cuPart] == SIZE_2Nx2N, CTX_PARTSIZE);
  code_intra_dir_luma(c, cu, cuPart, 1);
  code_intra_dir_chroma(c, cu->intra_dir[1][cuPart]);
  TU root; tu_root(root, depth);
  encode_transform(c, cu, cuPart, tu_key(root));
}

/* ======================================================================================== */
/* CU object helpers (cooperative)                                                           */
/* ======================================================================================== */
FCU_DEV FCU_NOINLINE void cu_init(CuObj *cu, int depth, int x, int y, int zidx)       /* initEstData / initSubCU */
{
  const Env E = env_get(); cu = FCU_UNI(cu); depth = FCU_UNI(depth); x = FCU_UNI(x); y = FCU_UNI(y); zidx = FCU_UNI(zidx);
  const int n = NPART >> (2 * depth), s = CTU >> depth;
  FCU_FOR_LANES {
    if (lane == 0) { cu->cost = FCU_MAX_DOUBLE; cu->dist = 0; cu->bits = 0; cu->bins = 0; cu->depth_cu = depth; cu->x = x; cu->y = y; cu->zidx = zidx; cu->nparts = n; }
    for (int i = lane; i < n; i += 64) {
      cu->depth[i] = (uint8_t)depth; cu->part_size[i] = SIZE_NONE; cu->pred_mode[i] = MODE_NONE; cu->tr_idx[i] = 0;
      cu->tskip[0][i] = cu->tskip[1][i] = cu->tskip[2][i] = 0; cu->cbf[0][i] = cu->cbf[1][i] = cu->cbf[2][i] = 0;
      cu->intra_dir[0][i] = DC; cu->intra_dir[1][i] = 0;
      cu->skip[i] = 0; cu->merge_flag[i] = 0; cu->merge_idx[i] = 0; cu->inter_dir[i] = 0; cu->mvp_idx[i] = -1; cu->ref_idx[i] = -1;
      cu->mv[i][0] = cu->mv[i][1] = 0; cu->mvd[i][0] = cu->mvd[i][1] = 0;
    }
    for (int i = lane; i < s * s; i += 64) cu->coef[0][i] = 0;
    for (int i = lane; i < s * s / 4; i += 64) { cu->coef[1][i] = 0; cu->coef[2][i] = 0; }
  }
  (void)E;
}
FCU_DEV FCU_NOINLINE void cu_copy_part_from(CuObj *dst, const CuObj *src, int partUnitIdx)   /* copyPartFrom */
{
  const Env E = env_get(); dst = FCU_UNI(dst); src = FCU_UNI(src); partUnitIdx = FCU_UNI(partUnitIdx);
  const int n = src->nparts, off = partUnitIdx * n;
  FCU_FOR_LANES {
    if (lane == 0) { dst->dist += src->dist; dst->bits += src->bits; dst->bins += src->bins; }
    for (int i = lane; i < n; i += 64) {
      dst->depth[off + i] = src->depth[i]; dst->part_size[off + i] = src->part_size[i]; dst->pred_mode[off + i] = src->pred_mode[i];
      dst->tr_idx[off + i] = src->tr_idx[i];
      for (int c = 0; c < 3; c++) { dst->tskip[c][off + i] = src->tskip[c][i]; dst->cbf[c][off + i] = src->cbf[c][i]; }
      dst->intra_dir[0][off + i] = src->intra_dir[0][i]; dst->intra_dir[1][off + i] = src->intra_dir[1][i];
      dst->skip[off + i] = src->skip[i]; dst->merge_flag[off + i] = src->merge_flag[i]; dst->merge_idx[off + i] = src->merge_idx[i];
      dst->inter_dir[off + i] = src->inter_dir[i]; dst->mvp_idx[off + i] = src->mvp_idx[i]; dst->ref_idx[off + i] = src->ref_idx[i];
      dst->mv[off + i][0] = src->mv[i][0]; dst->mv[off + i][1] = src->mv[i][1]; dst->mvd[off + i][0] = src->mvd[i][0]; dst->mvd[off + i][1] = src->mvd[i][1];
    }
    for (int i = lane; i < n * 16; i += 64) dst->coef[0][off * 16 + i] = src->coef[0][i];
    for (int i = lane; i < n * 4; i += 64) { dst->coef[1][off * 4 + i] = src->coef[1][i]; dst->coef[2][off * 4 + i] = src->coef[2][i]; }
  }
  (void)E;
}
FCU_DEV FCU_NOINLINE void cu_copy_to_pic(const CuObj *cu)                             /* copyToPic */
{
  const Env E = env_get(); cu = FCU_UNI(cu);
  fcu_ctu_out *p = &E.C->out[E.cur_ctu];
  const int n = cu->nparts, off = cu->zidx, qp = E.C->p.qp;
  FCU_FOR_LANES {
    if (lane == 0) { p->total_cost = cu->cost; p->total_dist = cu->dist; p->total_bits = cu->bits; p->total_bins = cu->bins; }
    for (int i = lane; i < n; i += 64) {
      p->depth[off + i] = cu->depth[i]; p->width[off + i] = p->height[off + i] = (uint8_t)(CTU >> cu->depth[i]);   /* the sub-CUs' own sizes */ p->skip[off + i] = cu->skip[i];
      p->merge_flag[off + i] = cu->merge_flag[i]; p->merge_idx[off + i] = cu->merge_idx[i]; p->inter_dir[off + i] = cu->inter_dir[i];
      p->mvp_idx[off + i] = cu->mvp_idx[i]; p->ref_idx[off + i] = cu->ref_idx[i];
      p->mv[off + i][0] = cu->mv[i][0]; p->mv[off + i][1] = cu->mv[i][1]; p->mvd[off + i][0] = cu->mvd[i][0]; p->mvd[off + i][1] = cu->mvd[i][1];
      p->part_size[off + i] = cu->part_size[i]; p->pred_mode[off + i] = cu->pred_mode[i]; p->qp[off + i] = (int8_t)qp;
      p->tr_idx[off + i] = cu->tr_idx[i];
      for (int c = 0; c < 3; c++) { p->tskip[c][off + i] = cu->tskip[c][i]; p->cbf[c][off + i] = cu->cbf[c][i]; }
      p->intra_dir[0][off + i] = cu->intra_dir[0][i]; p->intra_dir[1][off + i] = cu->intra_dir[1][i];
    }
    /* m_pcTrCoeff is raster order inside each TU; the engine keeps scan order, converted here.  Only the
     * depth-0 call matters for the coefficients (it rewrites the whole CTU and nothing reads them before). */
    if (cu->depth_cu == 0) {
      for (int i = lane; i < 4096; i += 64) {
        const int part = i >> 4; int v = 0;
        if (cu->pred_mode[part] != MODE_NONE) {
          const int log2 = 6 - cu->depth[part] - cu->tr_idx[part], np = 1 << (2 * (log2 - 2)), tp = part & ~(np - 1);
          const int st = cu->pred_mode[part] == MODE_INTRA ? coef_scan_idx(cu->intra_dir[0][tp], log2, 0) : 0;   /* inter: SCAN_DIAG */
          v = cu->coef[0][tp * 16 + k_iscan[k_scan_off[st * 4 + log2 - 2] + (i - tp * 16)]];
          if (cu->pred_mode[part] == MODE_INTER && !cu->cbf[0][part]) v = 0;   /* dropped residual: the levels of an earlier candidate are still there */
        }
        p->coeff_y[i] = v;
      }
      for (int i = lane; i < 1024; i += 64) {
        const int part = i >> 2; int vb = 0, vr = 0;
        if (cu->pred_mode[part] != MODE_NONE) {
          const int ll = 6 - cu->depth[part] - cu->tr_idx[part];            /* luma TU; chroma is half of it, 4x4 covers four luma 4x4 */
          const int log2 = ll > 2 ? ll - 1 : 2, np = ll > 2 ? 1 << (2 * (ll - 2)) : 4, tp = part & ~(np - 1);
          const int mode = chroma_final_mode(cu, tp), o = k_scan_off[(cu->pred_mode[part] == MODE_INTRA ? coef_scan_idx(mode, log2, 1) : 0) * 4 + log2 - 2];
          const int sp = k_iscan[o + (i - tp * 4)];
          vb = cu->coef[1][tp * 4 + sp]; vr = cu->coef[2][tp * 4 + sp];
          if (cu->pred_mode[part] == MODE_INTER) { if (!cu->cbf[1][part]) vb = 0; if (!cu->cbf[2][part]) vr = 0; }
        }
        p->coeff_cb[i] = vb; p->coeff_cr[i] = vr;
      }
    }
  }
}
FCU_DEV FCU_NOINLINE void copy_reco_to_pic(const Yuv *r, int x, int y, int s)
{
  const Env E = env_get(); r = FCU_UNI(r); x = FCU_UNI(x); y = FCU_UNI(y); s = FCU_UNI(s);
  FCU_FOR_LANES {
    for (int c = 0; c < 3; c++) {
      const int sh = c ? 1 : 0, bs = c ? 32 : 64, w = E.C->p.width >> sh, h = E.C->p.height >> sh;
      const uint8_t *src = c == 0 ? r->y : (c == 1 ? r->u : r->v);
      const int px = x >> sh, py = y >> sh, n = s >> sh;
      for (int i = lane; i < n * n; i += 64) { const int yy = i / n, xx = i % n; if (py + yy < h && px + xx < w) E.C->rec[c][(py + yy) * E.C->stride[c] + px + xx] = src[yy * bs + xx]; }
    }
  }
}

/* ======================================================================================== */
/* generic (sequential-candidate) TU trial: xIntraCodingTUBlock, TEncSearch.cpp:1092-1387    */
/* pixel phases use all lanes, RDOQ runs on lane 0 against coder `*cab`                       */
/* ======================================================================================== */
FCU_DEV FCU_NOINLINE void tu_trial(CuObj *cu, uint32_t tu_k, int comp, int cab, int save1load2)
{
  const Env E = env_get(); cu = FCU_UNI(cu); const TU tu = tu_of_key(FCU_UNI(tu_k)); comp = FCU_UNI(comp); cab = FCU_UNI(cab); save1load2 = FCU_UNI(save1load2);
  Scratch *G = E.G; const Params &P = E.C->p;
  if (comp && tu.cw == 0) { FCU_SERIAL { g_S.t_dist = 0; g_S.t_abs = 0; g_S.t_lsp = -1; } return; }
  FCU_TIC(t11_);
  const int d = cu->depth_cu, N = comp ? tu.cw : (1 << tu.log2), log2 = ilog2(N), n2 = N * N;
  const int bx = comp ? tu.cx : tu.x, by = comp ? tu.cy : tu.y, bs = comp ? 32 : 64, sh = comp ? 1 : 0;
  const int part = tu.part, layer = LOG2_MAXTU - tu.log2;
  uint8_t *org = yuv_plane(&G->org[d], comp) + by * bs + bx;
  uint8_t *pred = yuv_plane(&G->predt[d], comp) + by * bs + bx;
  uint8_t *recqt = yuv_plane(&G->qt_rec[layer], comp) + by * bs + bx;
  const int px = (cu->x >> sh) + bx, py = (cu->y >> sh) + by;
  uint8_t *recpic = E.C->rec[comp] + py * E.C->stride[comp] + px; const int rs = E.C->stride[comp];
  int16_t *coef = G->qt_coef[comp][layer] + (comp ? tu.off_c : tu.off_y);
  const int useTS = cu->tskip[comp][part];
  const int mode = comp ? chroma_final_mode(cu, part) : cu->intra_dir[0][part];
  const int useDst = comp == 0 && log2 == 2;
  const int qp = comp ? P.qp_c : P.qp;

  if (save1load2 != 2) {
    const int filt = use_filtered_ref(mode, log2, comp == 0);
    build_ref(comp, px, py, log2, filt);
    FCU_FOR_LANES {
      const uint8_t *r = filt ? g_S.reff : g_S.ref; const int dc = g_S.dc;
      if (lane == 0) g_S.t_last = -1;
      for (int i = lane; i < n2; i += 64) {
        const int y = i >> log2, x = i & (N - 1);
        const int v = pred_pixel(r, log2, mode, comp == 0, dc, x, y);
        pred[y * bs + x] = (uint8_t)v;
        if (save1load2 == 1) G->shared_pred[comp][i] = (uint8_t)v;
        G->p_resi[i] = (int16_t)(org[y * bs + x] - v);
      }
    }
  } else {
    FCU_FOR_LANES {
      if (lane == 0) g_S.t_last = -1;
      for (int i = lane; i < n2; i += 64) {
        const int y = i >> log2, x = i & (N - 1); const int v = G->shared_pred[comp][i];
        pred[y * bs + x] = (uint8_t)v; G->p_resi[i] = (int16_t)(org[y * bs + x] - v);
      }
    }
  }
  /* forward transform; the coefficients go to RDOQ as sign * lLevelDouble in scan order */
  const int scanType = coef_scan_idx(mode, log2, comp), qbits = rdoq_qbits(log2, qp), qscale = k_quant_scales[qp % 6];
  const uint16_t *iscan = k_iscan + k_scan_off[scanType * 4 + log2 - 2];
  if (useTS) { FCU_FOR_LANES { est_build(cab, lane); for (int i = lane; i < n2; i += 64) { const int sp = iscan[i]; const int32_t ld = level_double((int32_t)G->p_resi[i] << (15 - 8 - log2), qscale, qbits); G->p_lscan[sp] = ld; if (level_nonzero(ld, qbits)) FCU_ATOMIC_MAX(&g_S.t_last, sp); } } }
  else {
    FCU_FOR_LANES { by_log2(log2, [&](auto L) { constexpr int LG = decltype(L)::value; RowCache<(1 << LG), int16_t> rc; for (int i = lane; i < n2; i += 64) G->p_tmp[i] = fwd1<LG>(rc, G->p_resi, useDst, i); }); }
    FCU_FOR_LANES { est_build(cab, lane); by_log2(log2, [&](auto L) { constexpr int LG = decltype(L)::value; RowCache<(1 << LG), int32_t> rc; for (int i = lane; i < n2; i += 64) { const int sp = iscan[i]; const int32_t ld = level_double(fwd2<LG>(rc, G->p_tmp, useDst, i), qscale, qbits); G->p_lscan[sp] = ld; if (level_nonzero(ld, qbits)) FCU_ATOMIC_MAX(&g_S.t_last, sp); } }); }
  }
  {
    FCU_TIC(t8_);
    const int cbfCtx = comp ? (CTX_CBF_CHROMA + tu.tr_depth) : (CTX_CBF_LUMA + (tu.tr_depth == 0 ? 1 : 0));
    const int useRdoq = FCU_UNI((int)(useTS ? P.rdoq_ts : P.rdoq));
    const int onWave = useRdoq && log2 >= FCU_WAVE_RDOQ_MIN_LOG2;      /* a 4x4 block is one group: the phases of rdoq_wave cost what its sixteen serial iterations do */
    if (onWave) rdoq_wave(cab, G->p_lscan, G->p_qscan, FCU_UNI(g_S.t_last), log2, comp, scanType, cbfCtx, P, G->r_rec, G->r_cg);
    FCU_FOR_LANES {
      if (comp == 0) for (int i = lane; i < tu.nparts; i += 64) cu->tr_idx[part + i] = (uint8_t)tu.tr_depth;   /* setTrIdxSubParts */
      if (lane == 0) {
        if (onWave) { g_S.t_abs = g_S.rw_abs; g_S.t_lsp = g_S.rw_lsp; }
        else if (useRdoq) { const RdoqOut o = rdoq<1, 1>(cab, G->p_lscan, G->p_qscan, 1, g_S.t_last, log2, comp, scanType, cbfCtx, P, G->r_rec, G->r_cg); g_S.t_abs = o.abs_sum; g_S.t_lsp = o.last; }
        else { const RdoqOut o = quant_plain(G->p_lscan, G->p_qscan, 1, g_S.t_last, log2, comp, P); g_S.t_abs = o.abs_sum; g_S.t_lsp = o.last; }
        E.C->n_tu_trials++;
        FCU_COUNT(E, 15, (1ull << 40) + (unsigned long long)(g_S.t_last >= 0 ? ((g_S.t_last >> 4) + 1) * 16 : 0));   /* calls : coefficient iterations */
      }
    }
    FCU_TOC(E, t8_, 8);
  }
  const int absSum = g_S.t_abs;
  FCU_FOR_LANES {                                            /* setCbfPartRange + coefficient store */
    const int np = comp ? tu_nparts_c(tu) : tu.nparts;
    for (int i = lane; i < np; i += 64) cu->cbf[comp][part + i] = (uint8_t)((absSum > 0 ? 1 : 0) << tu.tr_depth);
    const int cgEnd = ((g_S.t_last >> 4) + 1) << 4;                /* RDOQ wrote the levels of scan positions < cgEnd */
    const DeqParams dq = deq_params(log2, qp);
    for (int i = lane; i < n2; i += 64) { const int sp = iscan[i]; const int q = (absSum > 0 && sp < cgEnd) ? G->p_qscan[sp] : 0; coef[sp] = (int16_t)q; G->p_tmp[useTS ? i : tr_index(i, log2)] = dequant1(q, dq); }
  }
  if (absSum > 0) {
    if (useTS) { FCU_FOR_LANES { const int s = 15 - 8 - log2; for (int i = lane; i < n2; i += 64) G->p_resi[i] = (int16_t)((G->p_tmp[i] + (1 << (s - 1))) >> s); } }
    else {
      FCU_FOR_LANES { by_log2(log2, [&](auto L) { constexpr int LG = decltype(L)::value; RowCache<(1 << LG), int32_t> rc; for (int i = lane; i < n2; i += 64) G->p_tcoef[i] = inv1<LG>(rc, G->p_tmp, useDst, i); }); }
      FCU_FOR_LANES { by_log2(log2, [&](auto L) { for (int i = lane; i < n2; i += 64) G->p_resi[i] = (int16_t)inv2<decltype(L)::value>(G->p_tcoef, useDst, i); }); }
    }
  }
  FCU_FOR_LANES {
    if (lane == 0) g_S.sad[35] = 0;
  }
  FCU_FOR_LANES {
    uint32_t sse = 0;
    for (int i = lane; i < n2; i += 64) {
      const int y = i >> log2, x = i & (N - 1);
      const int r = clip8(pred[y * bs + x] + (absSum > 0 ? G->p_resi[i] : 0));
      pred[y * bs + x] = (uint8_t)r; recqt[y * bs + x] = (uint8_t)r; recpic[y * rs + x] = (uint8_t)r;
      const int e = org[y * bs + x] - r; sse += (uint32_t)(e * e);
    }
    FCU_WAVE_ADD(&g_S.sad[35], sse);
  }
  FCU_SERIAL { const uint32_t sse = g_S.sad[35]; g_S.t_dist = comp ? (uint32_t)(P.chroma_weight * (double)sse) : sse; }
  FCU_TOC(E, t11_, 11);
}

/* xStoreIntraResultQT / xLoadIntraResultQT, TEncSearch.cpp:1760-1850 */
FCU_DEV FCU_NOINLINE void store_intra_result_qt(uint32_t tu_k, int comp)
{
  const Env E = env_get(); const TU tu = tu_of_key(FCU_UNI(tu_k)); comp = FCU_UNI(comp);
  Scratch *G = E.G;
  if (comp && tu.cw == 0) return;
  const int N = comp ? tu.cw : (1 << tu.log2), layer = LOG2_MAXTU - tu.log2, bs = comp ? 32 : 64, bx = comp ? tu.cx : tu.x, by = comp ? tu.cy : tu.y;
  const int16_t *src = G->qt_coef[comp][layer] + (comp ? tu.off_c : tu.off_y);
  const uint8_t *s = yuv_plane(&G->qt_rec[layer], comp) + by * bs + bx; uint8_t *t = yuv_plane(&G->ts_rec, comp) + by * bs + bx;
  FCU_FOR_LANES { for (int i = lane; i < N * N; i += 64) { G->ts_coef[comp][i] = src[i]; t[(i / N) * bs + (i % N)] = s[(i / N) * bs + (i % N)]; } }
}
FCU_DEV FCU_NOINLINE void load_intra_result_qt(const CuObj *cu, uint32_t tu_k, int comp)
{
  const Env E = env_get(); cu = FCU_UNI(cu); const TU tu = tu_of_key(FCU_UNI(tu_k)); comp = FCU_UNI(comp);
  Scratch *G = E.G;
  if (comp && tu.cw == 0) return;
  const int N = comp ? tu.cw : (1 << tu.log2), layer = LOG2_MAXTU - tu.log2, bs = comp ? 32 : 64, sh = comp ? 1 : 0, bx = comp ? tu.cx : tu.x, by = comp ? tu.cy : tu.y;
  int16_t *dst = G->qt_coef[comp][layer] + (comp ? tu.off_c : tu.off_y);
  uint8_t *t = yuv_plane(&G->qt_rec[layer], comp) + by * bs + bx; const uint8_t *s = yuv_plane(&G->ts_rec, comp) + by * bs + bx;
  uint8_t *pic = E.C->rec[comp] + ((cu->y >> sh) + by) * E.C->stride[comp] + (cu->x >> sh) + bx; const int rs = E.C->stride[comp];
  FCU_FOR_LANES { for (int i = lane; i < N * N; i += 64) { dst[i] = G->ts_coef[comp][i]; const uint8_t v = s[(i / N) * bs + (i % N)]; t[(i / N) * bs + (i % N)] = v; pic[(i / N) * rs + (i % N)] = v; } }
}

/* ======================================================================================== */
/* xRecurIntraCodingLumaQT (sequential path), TEncSearch.cpp:1393-1713                        */
/* LEVEL = recursion level (compile-time unrolled, <= 3).  Adds to g_S.q_dist/q_cost[LEVEL].   */
/* ======================================================================================== */
template <int LEVEL>
FCU_DEV FCU_NOINLINE void recur_luma_qt(CuObj *cu, uint32_t tu_k, int checkFirst, int reuseVc = -1)
{
  const Env E = env_get(); cu = FCU_UNI(cu); const TU tu = tu_of_key(FCU_UNI(tu_k)); checkFirst = FCU_UNI(checkFirst); reuseVc = FCU_UNI(reuseVc);
  Scratch *G = E.G; const Params &P = E.C->p;
  const int d = FCU_UNI((int)cu->depth_cu), part = tu.part, trDepth = tu.tr_depth, fullDepth = d + trDepth, log2 = tu.log2;
  const int partSize = FCU_UNI((int)cu->part_size[part]);
  const int checkFull = log2 <= LOG2_MAXTU;
  int checkSplit = log2 > min_tu_log2_in_cu(d, partSize);
  if (checkFirst && checkFull) checkSplit = 0;
  int checkTS = P.transform_skip && log2 == 2;
  if (P.ts_fast) checkTS = checkTS && (partSize == SIZE_NxN);
  double singleCost = FCU_MAX_DOUBLE; uint32_t singleDist = 0, singleCbf = 0; int bestModeId = 0; uint64_t singleFrac = 0;

  if (checkFull) {
    if (checkTS) {
      FCU_FOR_LANES cab_copy(slot_ptr(E, fullDepth, CI_QT_TRAFO_ROOT), &g_S.cab[CAB_GOON], lane);
      for (int modeId = 0; modeId < 2; modeId++) {
        FCU_FOR_LANES { for (int i = lane; i < tu.nparts; i += 64) cu->tskip[0][part + i] = (uint8_t)modeId; }
        tu_trial(cu, tu_key(tu), 0, (CAB_GOON), modeId == 0 ? 1 : 2);
        const uint32_t tmpDist = FCU_UNI(g_S.t_dist), tmpCbf = FCU_UNI((uint32_t)((cu->cbf[0][part] >> trDepth) & 1));
        double tmpCost;
        if (modeId == 1 && tmpCbf == 0) tmpCost = FCU_MAX_DOUBLE;
        else {
          { FCU_TIC(t12_); FCU_SERIAL { const uint64_t f0 = g_S.cab[CAB_GOON].frac & 32767; g_S.vc_bits[0] = leaf_luma_bits(CAB_GOON, cu, tu_key(tu)); g_S.t_frac = g_S.cab[CAB_GOON].frac - f0; } FCU_TOC(E, t12_, 12); }
          tmpCost = FCU_UNI(rd_cost(P, g_S.vc_bits[0], tmpDist));
        }
        if (tmpCost < singleCost) {
          singleCost = tmpCost; singleDist = tmpDist; singleCbf = tmpCbf; bestModeId = modeId; singleFrac = FCU_UNI(g_S.t_frac);
          if (bestModeId == 0) { store_intra_result_qt(tu_key(tu), 0); FCU_FOR_LANES cab_copy(slot_ptr(E, fullDepth, CI_TEMP_BEST), &g_S.cab[CAB_GOON], lane); }
        }
        if (modeId == 0) FCU_FOR_LANES cab_copy(&g_S.cab[CAB_GOON], slot_ptr(E, fullDepth, CI_QT_TRAFO_ROOT), lane);
      }
      FCU_FOR_LANES { for (int i = lane; i < tu.nparts; i += 64) cu->tskip[0][part + i] = (uint8_t)bestModeId; }
      if (bestModeId == 0) {
        load_intra_result_qt(cu, tu_key(tu), 0);
        FCU_FOR_LANES { for (int i = lane; i < tu.nparts; i += 64) cu->cbf[0][part + i] = (uint8_t)(singleCbf << trDepth); cab_copy(&g_S.cab[CAB_GOON], slot_ptr(E, fullDepth, CI_TEMP_BEST), lane); }
      }
    } else {
      if (checkSplit) FCU_FOR_LANES cab_copy(slot_ptr(E, fullDepth, CI_QT_TRAFO_ROOT), &g_S.cab[CAB_GOON], lane);
      FCU_FOR_LANES { for (int i = lane; i < tu.nparts; i += 64) cu->tskip[0][part + i] = 0; }
      if (reuseVc >= 0) {
        /* the un-split trial of the re-run (TEncSearch.cpp:2518-2586) repeats the first-pass trial of the same mode
         * from the same snapshot: take its levels, reconstruction, distortion, bits and coder state instead of
         * recomputing them (the candidate pools still hold them) */
        const int bv = reuseVc, N = 1 << log2, n2 = N * N, layer = LOG2_MAXTU - log2, cbf = FCU_UNI((int)(g_S.vc_abs[bv] > 0));
        FCU_SERIAL { g_S.t_frac = g_S.cab[CAB_LANE0 + g_S.vc_slot[bv]].frac - (g_S.cab[CAB_GOON].frac & 32767); }   /* the lane coder started from this snapshot */
        singleFrac = FCU_UNI(g_S.t_frac);
        FCU_FOR_LANES {
          for (int i = lane; i < n2; i += 64) {
            G->qt_coef[0][layer][tu.off_y + i] = (cbf && (i >> 4) <= (g_S.vc_last[bv] >> 4)) ? G->p_qscan[i * g_S.pu_nvc + bv] : (int16_t)0;
            G->qt_rec[layer].y[(tu.y + (i >> log2)) * 64 + tu.x + (i & (N - 1))] = G->p_rec[bv * n2 + i];
          }
          for (int i = lane; i < tu.nparts; i += 64) { cu->tr_idx[part + i] = (uint8_t)trDepth; cu->cbf[0][part + i] = (uint8_t)(cbf << trDepth); }
          cab_copy(&g_S.cab[CAB_GOON], &g_S.cab[CAB_LANE0 + g_S.vc_slot[bv]], lane);
        }
        singleDist = FCU_UNI(g_S.vc_dist[bv]); singleCbf = (uint32_t)cbf; singleCost = FCU_UNI(g_S.vc_cost[bv]);
      } else if (reuseVc <= -2) {
        /* 32x32 TU k of a 64x64 PU whose earlier TUs all kept the first pass's un-split result: coder and neighbourhood
         * are what they were in the first pass (pu_first_pass_64), so this trial would reproduce that pass's TU k of the
         * winning candidate -- levels, reconstruction, distortion, bits and coder state are taken from there */
        const int k = -2 - reuseVc, bv = FCU_UNI(g_S.pu_best_vc), N = 1 << log2, n2 = N * N, layer = LOG2_MAXTU - log2, cbf = FCU_UNI((int)g_S.c64_cbf[bv][k]);
        FCU_SERIAL {
          const uint64_t fprev = k ? G->c64_state[bv][k - 1].frac : (slot_ptr(E, d, CI_CURR_BEST)->frac & 32767);
          g_S.t_frac = G->c64_state[bv][k].frac - fprev;
          g_S.vc_bits[0] = (uint32_t)(((fprev & 32767) + g_S.t_frac) >> 15);
          g_S.t_dist = G->c64_distk[bv][k];
        }
        singleFrac = FCU_UNI(g_S.t_frac);
        FCU_FOR_LANES {
          for (int i = lane; i < n2; i += 64) {
            G->qt_coef[0][layer][tu.off_y + i] = G->c64_coef[bv][tu.off_y + i];
            const int o = (tu.y + (i >> log2)) * 64 + tu.x + (i & (N - 1));
            G->qt_rec[layer].y[o] = G->c64_rec[bv][o];
          }
          for (int i = lane; i < tu.nparts; i += 64) { cu->tr_idx[part + i] = (uint8_t)trDepth; cu->cbf[0][part + i] = (uint8_t)(cbf << trDepth); }
          cab_copy(&g_S.cab[CAB_GOON], &G->c64_state[bv][k], lane);
        }
        FCU_SERIAL { g_S.cab[CAB_GOON].frac = (g_S.cab[CAB_GOON].frac & 32767) + ((uint64_t)g_S.vc_bits[0] << 15); }
        singleDist = FCU_UNI(g_S.t_dist); singleCbf = (uint32_t)cbf;
        singleCost = FCU_UNI(rd_cost(P, g_S.vc_bits[0], singleDist));
      } else {
        tu_trial(cu, tu_key(tu), 0, (CAB_GOON), 0);
        singleDist = FCU_UNI(g_S.t_dist);
        if (checkSplit) singleCbf = FCU_UNI((uint32_t)((cu->cbf[0][part] >> trDepth) & 1));
        { FCU_TIC(t12_); FCU_SERIAL { const uint64_t f0 = g_S.cab[CAB_GOON].frac & 32767; g_S.vc_bits[0] = leaf_luma_bits(CAB_GOON, cu, tu_key(tu)); g_S.t_frac = g_S.cab[CAB_GOON].frac - f0; } FCU_TOC(E, t12_, 12); }
        singleFrac = FCU_UNI(g_S.t_frac);
        singleCost = FCU_UNI(rd_cost(P, g_S.vc_bits[0], singleDist));
      }
    }
  }
  if (checkSplit) {
    if constexpr (LEVEL < 3) {
      if (checkFull) { FCU_FOR_LANES { cab_copy(slot_ptr(E, fullDepth, CI_QT_TRAFO_TEST), &g_S.cab[CAB_GOON], lane); } FCU_FOR_LANES { cab_copy(&g_S.cab[CAB_GOON], slot_ptr(E, fullDepth, CI_QT_TRAFO_ROOT), lane); } }
      else FCU_FOR_LANES cab_copy(slot_ptr(E, fullDepth, CI_QT_TRAFO_ROOT), &g_S.cab[CAB_GOON], lane);
      FCU_SERIAL { g_S.q_dist[LEVEL + 1] = 0; g_S.q_cost[LEVEL + 1] = 0; g_S.q_frac[LEVEL + 1] = 0; }
      uint32_t splitCbf = 0;
      int chainOk = LEVEL == 0 && log2 == 6 && !checkFirst && FCU_UNI(g_S.c64_valid);   /* see the reuse branch above */
      for (int i = 0; i < 4; i++) {
        TU c; tu_child(c, tu, i, 0);
        recur_luma_qt<LEVEL + 1>(cu, tu_key(c), checkFirst, chainOk ? -2 - i : -1);
        splitCbf |= FCU_UNI((uint32_t)((cu->cbf[0][c.part] >> c.tr_depth) & 1));
        chainOk = chainOk && FCU_UNI((int)cu->tr_idx[c.part]) == c.tr_depth;
      }
      const uint32_t splitDist = FCU_UNI(g_S.q_dist[LEVEL + 1]);
      /* uiSplitBits = xGetIntraBitsQT of the whole split subtree from QT_TRAFO_ROOT (TEncSearch.cpp:1600-1606).  Every bin
       * of that walk has already been counted, from the same state of its context, by the walks of the children's chosen
       * encodings (child 0 starts from this node's root state and, when this node begins the CU, also carries the
       * part-size / luma-mode bins; the subdivision and cbf contexts are indexed by TU size / depth), so the exact Q15
       * count is this node's own subdivision flag plus the children's counts, and the coder after the last child is
       * the coder after the walk.  Not for the NxN root, whose walk leaves out the luma modes its children count. */
      const int sumBits = !(partSize == SIZE_NxN && trDepth == 0);
      FCU_FOR_LANES {
        if (splitCbf) for (int o = lane; o < tu.nparts; o += 64) cu->cbf[0][part + o] |= (uint8_t)(1 << trDepth);
        if (!sumBits) cab_copy(&g_S.cab[CAB_GOON], slot_ptr(E, fullDepth, CI_QT_TRAFO_ROOT), lane);
      }
      {
        FCU_TIC(t12_);
        FCU_SERIAL {
          const uint64_t f0 = slot_ptr(E, fullDepth, CI_QT_TRAFO_ROOT)->frac & 32767;
          if (sumBits) {
            g_S.cab[CAB_GOON].frac = f0; g_S.cab[CAB_GOON].bins = 0;
            if (log2 <= LOG2_MAXTU) cab_bin((CAB_GOON), 1, CTX_SUBDIV + 5 - log2);
            g_S.cab[CAB_GOON].frac += g_S.q_frac[LEVEL + 1];
            g_S.vc_bits[0] = (uint32_t)(g_S.cab[CAB_GOON].frac >> 15);
          } else g_S.vc_bits[0] = intra_bits_qt((CAB_GOON), cu, tu_key(tu), 1, 0);
          g_S.t_frac = g_S.cab[CAB_GOON].frac - f0;
        }
        FCU_TOC(E, t12_, 12);
      }
      const double splitCost = FCU_UNI(rd_cost(P, g_S.vc_bits[0], splitDist));
      if (splitCost < singleCost) { FCU_SERIAL { g_S.q_dist[LEVEL] += splitDist; g_S.q_cost[LEVEL] += splitCost; g_S.q_frac[LEVEL] += g_S.t_frac; } return; }
      FCU_FOR_LANES {
        cab_copy(&g_S.cab[CAB_GOON], slot_ptr(E, fullDepth, CI_QT_TRAFO_TEST), lane);
        for (int i = lane; i < tu.nparts; i += 64) { cu->tr_idx[part + i] = (uint8_t)trDepth; cu->cbf[0][part + i] = (uint8_t)(singleCbf << trDepth); cu->tskip[0][part + i] = (uint8_t)bestModeId; }
        const int N = 1 << log2, layer = LOG2_MAXTU - log2;
        const uint8_t *s = G->qt_rec[layer].y + tu.y * 64 + tu.x;
        uint8_t *p = E.C->rec[0] + (cu->y + tu.y) * E.C->stride[0] + cu->x + tu.x; const int rs = E.C->stride[0];
        for (int i = lane; i < N * N; i += 64) p[(i >> log2) * rs + (i & (N - 1))] = s[(i >> log2) * 64 + (i & (N - 1))];
      }
    }
  }
  FCU_SERIAL { g_S.q_dist[LEVEL] += singleDist; g_S.q_cost[LEVEL] += singleCost; g_S.q_frac[LEVEL] += singleFrac; }
}

/* xSetIntraResultLumaQT, TEncSearch.cpp:1717-1757 (iterative) */
FCU_DEV FCU_NOINLINE void set_intra_result_luma_qt(CuObj *cu, uint32_t root_k, Yuv *reco)
{
  const Env E = env_get(); cu = FCU_UNI(cu); const TU root = tu_of_key(FCU_UNI(root_k)); reco = FCU_UNI(reco);
  Scratch *G = E.G;
  TU *st = g_S.wk_st; int *ci = g_S.wk_ci; int sp = 0;
  st[0] = root; ci[0] = -1;
  while (sp >= 0) {
    const TU tu = st[sp];
    if (ci[sp] < 0) {
      if (cu->tr_idx[tu.part] == tu.tr_depth) {
        const int N = 1 << tu.log2, layer = LOG2_MAXTU - tu.log2;
        FCU_FOR_LANES {
          for (int i = lane; i < N * N; i += 64) {
            cu->coef[0][tu.off_y + i] = G->qt_coef[0][layer][tu.off_y + i];
            const int yy = tu.y + (i >> tu.log2), xx = tu.x + (i & (N - 1));
            reco->y[yy * 64 + xx] = G->qt_rec[layer].y[yy * 64 + xx];
          }
        }
        sp--; continue;
      }
      ci[sp] = 0;
    }
    if (ci[sp] >= 4) { sp--; continue; }
    { TU ch; tu_child(ch, st[sp], ci[sp], 0); ci[sp]++; sp++; st[sp] = ch; ci[sp] = -1; }
  }
}

/* ======================================================================================== */
/* RMD: 35 predictions + Hadamard SATD staged through LDS (TEncSearch.cpp:2300-2361,          */
/* TComRdCost.cpp:1343-1604) and the sorted candidate list (xUpdateCandList :5345-5370)       */
/* ======================================================================================== */
/* SATD of one USZ x USZ block of (source - prediction) for one mode (xCalcHADs8x8 / xCalcHADs4x4, TComRdCost.cpp:1343-1534).
 * Row by row: predict, row butterfly, then add the row into the USZ*USZ column-transform accumulators with the sign of the
 * Sylvester matrix (-1)^popcount(k & y); the sum of magnitudes does not depend on the order of the Hadamard outputs. */
template <int USZ>
FCU_DEV uint32_t satd_unit(const uint8_t *org, int log2, int mode, int dc, int bx, int by)
{
  const uint8_t *r = use_filtered_ref(mode, log2, 1) ? g_S.reff : g_S.ref;
  int acc[USZ * USZ];
#pragma unroll
  for (int i = 0; i < USZ * USZ; i++) acc[i] = 0;
  for (int y = 0; y < USZ; y++) {
    uint8_t o[USZ]; int row[USZ];
    __builtin_memcpy(o, org + (by + y) * 64 + bx, USZ);
#pragma unroll
    for (int x = 0; x < USZ; x++) row[x] = (int)o[x] - pred_pixel(r, log2, mode, 1, dc, bx + x, by + y);
#pragma unroll
    for (int len = 1; len < USZ; len <<= 1)
#pragma unroll
      for (int i = 0; i < USZ; i += 2 * len)
#pragma unroll
        for (int j = i; j < i + len; j++) { const int a = row[j], b = row[j + len]; row[j] = a + b; row[j + len] = a - b; }
#pragma unroll
    for (int k = 0; k < USZ; k++) {
      const int neg = -(__builtin_popcount((unsigned)(k & y)) & 1);          /* 0 or -1 */
#pragma unroll
      for (int j = 0; j < USZ; j++) acc[k * USZ + j] += (row[j] ^ neg) - neg;
    }
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < USZ * USZ; i++) s += iabs(acc[i]);
  return (uint32_t)(USZ == 8 ? ((s + 2) >> 2) : ((s + 1) >> 1));
}

FCU_DEV FCU_NOINLINE void rmd(CuObj *cu, uint32_t tu_k)
{
  const Env E = env_get(); cu = FCU_UNI(cu); const TU tu = tu_of_key(FCU_UNI(tu_k));
  Scratch *G = E.G; const Params &P = E.C->p;
  const int d = cu->depth_cu, N = 1 << tu.log2, log2 = tu.log2;
  build_ref(0, cu->x + tu.x, cu->y + tu.y, log2, 1);
  const uint8_t *org = G->org[d].y + tu.y * 64 + tu.x;
  /* one (mode, Hadamard block) unit per lane, entirely in registers: no staging, no barrier between the stages */
  FCU_FOR_LANES { if (lane < 36) g_S.sad[lane] = 0; }
  FCU_FOR_LANES {
    const int dc = g_S.dc;
    if (N >= 8) {
      const int bpr = N >> 3, nblk = bpr * bpr;
      for (int u = lane; u < 35 * nblk; u += 64) { const int mode = u / nblk, blk = u - mode * nblk; FCU_ATOMIC_ADD(&g_S.sad[mode], satd_unit<8>(org, log2, mode, dc, (blk % bpr) * 8, (blk / bpr) * 8)); }
    } else {
      for (int u = lane; u < 35; u += 64) FCU_ATOMIC_ADD(&g_S.sad[u], satd_unit<4>(org, log2, u, dc, 0, 0));
    }
  }
  /* mode bits + sorted insert (serial) */
  FCU_SERIAL {
    int preds[3];
    const int nm = intra_dir_predictor(E, cu, tu.part, preds);
    g_S.preds[0] = preds[0]; g_S.preds[1] = preds[1]; g_S.preds[2] = preds[2]; g_S.n_mpm = nm;
    const Cabac *cb = slot_ptr(E, d, CI_CURR_BEST);
    const uint64_t carry = cb->frac & 32767;               /* loadIntraDirMode + resetBits, TEncSearch.cpp:5313-5340 */
    int numFull = k_rd_mode_num[log2 - 2];
    double *candCost = g_S.cand_cost;                    /* in LDS: the sorted insert indexes it with run-time subscripts */
    for (int i = 0; i < numFull; i++) candCost[i] = FCU_MAX_DOUBLE;
    for (int mode = 0; mode < 35; mode++) {
      int predIdx = -1;
      for (int i = 0; i < 3; i++) if (mode == preds[i]) predIdx = i;
      const uint64_t fr = carry + (uint64_t)ctx_bits(CAB_CUR0 + d, CTX_INTRA_LUMA, predIdx != -1) + (uint64_t)32768 * (uint64_t)(predIdx != -1 ? (predIdx ? 2 : 1) : 5);
      const uint32_t modeBits = (uint32_t)(fr >> 15);
      const double cost = (double)g_S.sad[mode] + (double)modeBits * P.sqrt_lambda;
      int shift = 0;
      while (shift < numFull && cost < candCost[numFull - 1 - shift]) shift++;
      if (shift != 0) {
        for (int i = 1; i < shift; i++) { g_S.rd_mode[numFull - i] = g_S.rd_mode[numFull - 1 - i]; candCost[numFull - i] = candCost[numFull - 1 - i]; }
        g_S.rd_mode[numFull - shift] = mode; candCost[numFull - shift] = cost;
      }
    }
    for (int j = 0; j < nm; j++) {
      int inc = 0;
      for (int i = 0; i < numFull; i++) inc |= (preds[j] == g_S.rd_mode[i]);
      if (!inc) g_S.rd_mode[numFull++] = preds[j];
    }
    g_S.n_rd = numFull;
  }
}

/* ======================================================================================== */
/* first-pass RDO of one PU whose TU is not split: all candidates (x transform-skip variants) */
/* side by side -- pixel phases on all lanes, RDOQ + bit counting one candidate per lane.     */
/* Restates the loop TEncSearch.cpp:2447-2516 for the bCheckFirst case (:1428-1444).          */
/* ======================================================================================== */
FCU_DEV FCU_NOINLINE void pu_first_pass_batched(CuObj *cu, uint32_t tu_k)
{
  const Env E = env_get(); cu = FCU_UNI(cu); const TU tu = tu_of_key(FCU_UNI(tu_k));
  Scratch *G = E.G; const Params &P = E.C->p;
  const int d = cu->depth_cu, N = 1 << tu.log2, log2 = tu.log2, n2 = N * N, part = tu.part;
  const int partSize = cu->part_size[part];
  int checkTS = P.transform_skip && log2 == 2;
  if (P.ts_fast) checkTS = checkTS && (partSize == SIZE_NxN);
  const int nc = g_S.n_rd, tsv = checkTS ? 2 : 1, tss = tsv - 1, nvc = nc * tsv;   /* v / tsv == v >> tss, v % tsv == v & tss */
  FCU_CHECK(nvc <= MAXVC && nvc * n2 <= POOL && (MAXLC % 2) == 0);
  const int qbits = rdoq_qbits(log2, P.qp), qscale = k_quant_scales[P.qp % 6];
  const int useDst = log2 == 2;
  const uint8_t *org = G->org[d].y + tu.y * 64 + tu.x;
  /* reference samples are shared by all candidates: the TU is the whole PU */
  build_ref(0, cu->x + tu.x, cu->y + tu.y, log2, 1);
  FCU_FOR_LANES {                                            /* prediction + residual per candidate */
    const int dc = g_S.dc;
    for (int i = lane; i < nc * n2; i += 64) {
      const int cnd = i / n2, p = i - cnd * n2, y = p >> log2, x = p & (N - 1), mode = g_S.rd_mode[cnd];
      const uint8_t *r = use_filtered_ref(mode, log2, 1) ? g_S.reff : g_S.ref;
      const int v = pred_pixel(r, log2, mode, 1, dc, x, y);
      G->p_pred[i] = (uint8_t)v; G->p_resi[i] = (int16_t)(org[y * 64 + x] - v);
    }
  }
  FCU_FOR_LANES { if (lane < nvc) g_S.vc_last[lane] = -1; by_log2(log2, [&](auto L) { constexpr int LG = decltype(L)::value; RowCache<(1 << LG), int16_t> rc; for (int i = lane; i < nc * n2; i += 64) { const int cnd = i / n2; G->p_tmp[i] = fwd1<LG>(rc, G->p_resi + cnd * n2, useDst, i - cnd * n2); } }); }
  FCU_FOR_LANES {                                            /* slot v = cand*tsv + ts */
    est_build(CAB_CUR0 + d, lane);
    by_log2(log2, [&](auto L) { constexpr int LG = decltype(L)::value; RowCache<(1 << LG), int32_t> rc;
      for (int i = lane; i < nvc * n2; i += 64) {
        const int v = i / n2, p = i - v * n2, cnd = (v >> tss), ts = (v & tss);
        const uint16_t *iscan = k_iscan + k_scan_off[coef_scan_idx(g_S.rd_mode[cnd], log2, 0) * 4 + log2 - 2];
        const int32_t t = ts ? ((int32_t)G->p_resi[cnd * n2 + p] << (15 - 8 - log2)) : fwd2<LG>(rc, G->p_tmp + cnd * n2, useDst, p);
        const int sp = iscan[p]; const int32_t ld = level_double(t, qscale, qbits);
        G->p_lscan[sp * nvc + v] = ld;
        if (level_nonzero(ld, qbits)) FCU_ATOMIC_MAX(&g_S.vc_last[v], sp);
      }
    });
  }
  FCU_TIC(t2_);
  FCU_FOR_LANES {                                            /* RDOQ: one virtual candidate per lane */
    if (lane < nvc) {
      const int mode = g_S.rd_mode[(lane >> tss)];
      RdoqRec *rrec = G->r_rec + lane; double *rcg = G->r_cg + lane;
      const int cbfCtx = CTX_CBF_LUMA + (tu.tr_depth == 0 ? 1 : 0);
      const RdoqOut o = ((lane & tss) ? P.rdoq_ts : P.rdoq) ? rdoq<0, 1>(CAB_CUR0 + d, G->p_lscan + lane, G->p_qscan + lane, nvc, g_S.vc_last[lane], log2, 0, coef_scan_idx(mode, log2, 0), cbfCtx, P, rrec, rcg)
                                                            : quant_plain(G->p_lscan + lane, G->p_qscan + lane, nvc, g_S.vc_last[lane], log2, 0, P);
      g_S.vc_abs[lane] = o.abs_sum; g_S.vc_lsp[lane] = o.last;
      g_S.vc_dist[lane] = 0;
    }
    if (lane == 0) E.C->n_tu_trials += (unsigned long long)nvc;
  }
  FCU_TOC(E, t2_, 2);
  FCU_FOR_LANES {                                            /* levels back to raster order + dequantisation */
    const DeqParams dq = deq_params(log2, P.qp);
    for (int i = lane; i < nvc * n2; i += 64) {
      const int v = i / n2, p = i - v * n2;
      const uint16_t *iscan = k_iscan + k_scan_off[coef_scan_idx(g_S.rd_mode[(v >> tss)], log2, 0) * 4 + log2 - 2];
      const int sp = iscan[p];
      const int q = (g_S.vc_abs[v] > 0 && (sp >> 4) <= (g_S.vc_last[v] >> 4)) ? G->p_qscan[sp * nvc + v] : 0;
      G->p_tmp[((v & tss)) ? i : v * n2 + tr_index(p, log2)] = dequant1(q, dq);
    }
  }
  FCU_FOR_LANES {
    by_log2(log2, [&](auto L) { constexpr int LG = decltype(L)::value; RowCache<(1 << LG), int32_t> rc;
      for (int i = lane; i < nvc * n2; i += 64) {
        const int v = i / n2, p = i - v * n2, ts = (v & tss);
        if (ts) { const int s = 15 - 8 - log2; G->p_tcoef[i] = (G->p_tmp[i] + (1 << (s - 1))) >> s; }
        else G->p_tcoef[i] = inv1<LG>(rc, G->p_tmp + v * n2, useDst, p);
      }
    });
  }
  FCU_FOR_LANES {
    by_log2(log2, [&](auto L) {
      uint32_t acc = 0;
      for (int i = lane; i < nvc * n2; i += 64) {
        const int v = i / n2, p = i - v * n2, cnd = (v >> tss), ts = (v & tss), y = p >> log2, x = p & (N - 1);
        int res = 0;
        if (g_S.vc_abs[v] > 0) res = ts ? (int16_t)G->p_tcoef[i] : inv2<decltype(L)::value>(G->p_tcoef + v * n2, useDst, p);
        const int r = clip8(G->p_pred[cnd * n2 + p] + res);
        G->p_rec[i] = (uint8_t)r;
        const int e = org[y * 64 + x] - r;
        FCU_DIST_ADD(acc, v, e, i, n2);
      }
    });
  }
  FCU_TIC(t3_);
  /* Bits of (header, subdiv, cbf, coefficients): xGetIntraBitsQT, one variant per lane on MAXLC lane-private coders.
   * More than MAXLC variants (a 4x4 PU with 9-10 candidates x 2) take a second round; it reuses coder slots, but
   * never the one of the best variant so far: the re-run of the winner (recur_luma_qt) takes its coder state from there. */
  for (int vbase = 0; vbase < nvc; vbase += MAXLC) {
    const int slotBase = vbase ? (g_S.vc_slot[g_S.pu_best_vc] + 1) % MAXLC : 0;
#if defined(FCU_EMU) && defined(FCU_EMU_TRACE_ROUNDS)
    if (vbase) fprintf(stderr, "second bit-count round: %d variants\n", nvc);
#endif
    FCU_FOR_LANES {
      const int vc = vbase + lane;
      if (lane < MAXLC && vc < nvc) {
        const int cnd = (vc >> tss), ts = (vc & tss), mode = g_S.rd_mode[cnd], cbf = g_S.vc_abs[vc] > 0;
        const int slot = (slotBase + lane) % MAXLC;
        double cost;
        g_S.vc_slot[vc] = (uint8_t)slot;
        if (ts && !cbf) cost = FCU_MAX_DOUBLE;               /* TS with CBF 0 is forbidden, TEncSearch.cpp:1503-1507 */
        else {
          const int c = CAB_LANE0 + slot;
          cab_copy1(&g_S.cab[c], slot_ptr(E, d, CI_CURR_BEST));
          cab_reset_bits(c);
          if (part == 0 && P.slice_type != SLICE_I) { code_skip_flag(E, c, cu, 0); code_pred_mode(c, cu, 0); }
          if (part == 0 && d == MAXDEPTH) cab_bin(c, partSize == SIZE_2Nx2N, CTX_PARTSIZE);
          code_luma_dir_bits(c, mode, g_S.preds);
          if (!(partSize == SIZE_NxN && tu.tr_depth == 0) && log2 <= LOG2_MAXTU && log2 != LOG2_MINTU && log2 != min_tu_log2_in_cu(d, partSize))
            cab_bin(c, 0, CTX_SUBDIV + 5 - log2);
          cab_bin(c, cbf, CTX_CBF_LUMA + (tu.tr_depth == 0 ? 1 : 0));
          if (cbf) code_coeff_nxn<0>(c, G->p_qscan + vc, nvc, g_S.vc_lsp[vc], log2, 0, coef_scan_idx(mode, log2, 0), ts, P, g_S.lane_abs[slot]);
          g_S.vc_bits[vc] = cab_bits(c);
          cost = rd_cost(P, g_S.vc_bits[vc], g_S.vc_dist[vc]);
        }
        g_S.vc_cost[vc] = cost;
      }
    }
    FCU_SERIAL {                                             /* strict '<', earlier candidate wins ties; candidates coded so far */
      const int ncDone = ((vbase + MAXLC < nvc) ? vbase + MAXLC : nvc) / tsv;
      double best = FCU_MAX_DOUBLE; int bv = 0;
      for (int cnd = 0; cnd < ncDone; cnd++) {
        int v = cnd * tsv; double c = g_S.vc_cost[v];
        if (tsv == 2 && g_S.vc_cost[v + 1] < c) { v = v + 1; c = g_S.vc_cost[v]; }
        if (c < best) { best = c; bv = v; }
      }
      g_S.pu_nvc = nvc; g_S.pu_best_vc = bv; g_S.pu_best_cost = best; g_S.pu_best_dist = g_S.vc_dist[bv]; g_S.pu_best_mode = g_S.rd_mode[(bv >> tss)];
    }
  }
  FCU_TOC(E, t3_, 3);
  {                                                          /* xSetIntraResultLumaQT + decision snapshot */
    const int bv = g_S.pu_best_vc, ts = (bv & tss), cbf = g_S.vc_abs[bv] > 0;
    Yuv *reco = &G->reco[d][1 - g_S.reco_best_idx[d]];
    FCU_FOR_LANES {
      for (int i = lane; i < n2; i += 64) {
        cu->coef[0][tu.off_y + i] = (cbf && (i >> 4) <= (g_S.vc_last[bv] >> 4)) ? G->p_qscan[i * nvc + bv] : (int16_t)0;
        reco->y[(tu.y + (i >> log2)) * 64 + tu.x + (i & (N - 1))] = G->p_rec[bv * n2 + i];
      }
      for (int i = lane; i < tu.nparts; i += 64) { G->tmp_tr_idx[i] = (uint8_t)tu.tr_depth; G->tmp_cbf[i] = (uint8_t)(cbf << tu.tr_depth); G->tmp_tskip[i] = (uint8_t)ts; }
    }
  }
}

/* First pass of a 64x64 PU (TEncSearch.cpp:2440-2516, where xRecurIntraCodingLumaQT can only code it as four 32x32 TUs).
 * The candidates share the snapshot they start from and nothing else, so they advance side by side: for each of the four
 * TUs in turn, the prediction of every candidate from that candidate's own reconstruction of the earlier TUs (swapped
 * into the picture plane for build_ref), then one batched transform, RDOQ with one candidate per lane on its own running
 * coder, reconstruction and bit count.  A candidate's bits are the Q15 sum of its four TU walks (see recur_luma_qt). */
FCU_DEV FCU_NOINLINE void pu_first_pass_64(CuObj *cu, uint32_t root_k)
{
  const Env E = env_get(); cu = FCU_UNI(cu); const TU root = tu_of_key(FCU_UNI(root_k));
  Scratch *G = E.G; const Params &P = E.C->p;
  const int d = cu->depth_cu, nc = g_S.n_rd, log2 = 5, N = 32, n2 = N * N, partSize = cu->part_size[root.part];
  FCU_CHECK(nc <= 5 && nc * n2 <= POOL && root.log2 == 6);
  const int qbits = rdoq_qbits(log2, P.qp), qscale = k_quant_scales[P.qp % 6];
  uint8_t *recpic = E.C->rec[0] + cu->y * E.C->stride[0] + cu->x; const int rs = E.C->stride[0];
  FCU_FOR_LANES {
    if (lane < nc) { const int c = CAB_LANE0 + lane; cab_copy1(&g_S.cab[c], slot_ptr(E, d, CI_CURR_BEST)); cab_reset_bits(c); g_S.c64_dist[lane] = 0; }
  }
  for (int k = 0; k < 4; k++) {
    TU tu; tu_child(tu, root, k, 0);
    const uint8_t *org = G->org[d].y + tu.y * 64 + tu.x;
    for (int cand = 0; cand < nc; cand++) {
      const int mode = g_S.rd_mode[cand], filt = use_filtered_ref(mode, log2, 1);
      if (k) {
        FCU_FOR_LANES {                                      /* this candidate's earlier TUs become the neighbourhood */
          for (int i = lane; i < k * n2; i += 64) {
            const int j = i >> 10, q = i & (n2 - 1), yy = (j >> 1) * N + (q >> log2), xx = (j & 1) * N + (q & (N - 1));
            recpic[yy * rs + xx] = G->c64_rec[cand][yy * 64 + xx];
          }
        }
      }
      build_ref(0, cu->x + tu.x, cu->y + tu.y, log2, filt);
      FCU_FOR_LANES {
        const uint8_t *r = filt ? g_S.reff : g_S.ref; const int dc = g_S.dc;
        for (int i = lane; i < n2; i += 64) {
          const int y = i >> log2, x = i & (N - 1);
          const int v = pred_pixel(r, log2, mode, 1, dc, x, y);
          G->p_pred[cand * n2 + i] = (uint8_t)v; G->p_resi[cand * n2 + i] = (int16_t)(org[y * 64 + x] - v);
        }
      }
    }
    FCU_FOR_LANES { if (lane < nc) g_S.vc_last[lane] = -1; by_log2(log2, [&](auto L) { constexpr int LG = decltype(L)::value; RowCache<(1 << LG), int16_t> rc; for (int i = lane; i < nc * n2; i += 64) { const int cnd = i / n2; G->p_tmp[i] = fwd1<LG>(rc, G->p_resi + cnd * n2, 0, i - cnd * n2); } }); }
    FCU_FOR_LANES {
      by_log2(log2, [&](auto L) { constexpr int LG = decltype(L)::value; RowCache<(1 << LG), int32_t> rc;
        for (int i = lane; i < nc * n2; i += 64) {
          const int v = i / n2, p = i - v * n2;
          const uint16_t *iscan = k_iscan + k_scan_off[coef_scan_idx(g_S.rd_mode[v], log2, 0) * 4 + log2 - 2];
          const int32_t t = fwd2<LG>(rc, G->p_tmp + v * n2, 0, p);
          const int sp = iscan[p]; const int32_t ld = level_double(t, qscale, qbits);
          G->p_lscan[sp * nc + v] = ld;
          if (level_nonzero(ld, qbits)) FCU_ATOMIC_MAX(&g_S.vc_last[v], sp);
        }
      });
    }
    FCU_FOR_LANES {                                          /* RDOQ: one candidate per lane, priced against its own coder */
      if (lane < nc) {
        RdoqRec *rrec = G->r_rec + lane; double *rcg = G->r_cg + lane;
        const RdoqOut o = P.rdoq ? rdoq<0, 0>(CAB_LANE0 + lane, G->p_lscan + lane, G->p_qscan + lane, nc, g_S.vc_last[lane], log2, 0,
                                              coef_scan_idx(g_S.rd_mode[lane], log2, 0), CTX_CBF_LUMA + (tu.tr_depth == 0 ? 1 : 0), P, rrec, rcg)
                                 : quant_plain(G->p_lscan + lane, G->p_qscan + lane, nc, g_S.vc_last[lane], log2, 0, P);
        g_S.vc_abs[lane] = o.abs_sum; g_S.vc_lsp[lane] = o.last; g_S.vc_dist[lane] = 0;
      }
      if (lane == 0) E.C->n_tu_trials += (unsigned long long)nc;
    }
    FCU_FOR_LANES {                                          /* levels kept per candidate (scan order) + dequantisation */
      const DeqParams dq = deq_params(log2, P.qp);
      for (int i = lane; i < nc * n2; i += 64) {
        const int v = i / n2, p = i - v * n2;
        const uint16_t *iscan = k_iscan + k_scan_off[coef_scan_idx(g_S.rd_mode[v], log2, 0) * 4 + log2 - 2];
        const int sp = iscan[p];
        const int q = (g_S.vc_abs[v] > 0 && (sp >> 4) <= (g_S.vc_last[v] >> 4)) ? G->p_qscan[sp * nc + v] : 0;
        G->c64_coef[v][tu.off_y + sp] = (int16_t)q;
        G->p_tmp[v * n2 + tr_index(p, log2)] = dequant1(q, dq);
      }
    }
    FCU_FOR_LANES { by_log2(log2, [&](auto L) { constexpr int LG = decltype(L)::value; RowCache<(1 << LG), int32_t> rc; for (int i = lane; i < nc * n2; i += 64) { const int v = i / n2; G->p_tcoef[i] = inv1<LG>(rc, G->p_tmp + v * n2, 0, i - v * n2); } }); }
    FCU_FOR_LANES {
      by_log2(log2, [&](auto L) {
        uint32_t acc = 0;
        for (int i = lane; i < nc * n2; i += 64) {
          const int v = i / n2, p = i - v * n2, y = p >> log2, x = p & (N - 1);
          const int res = g_S.vc_abs[v] > 0 ? inv2<decltype(L)::value>(G->p_tcoef + v * n2, 0, p) : 0;
          const int r = clip8(G->p_pred[i] + res);
          G->c64_rec[v][(tu.y + y) * 64 + tu.x + x] = (uint8_t)r;
          const int e = org[y * 64 + x] - r;
          FCU_DIST_ADD(acc, v, e, i, n2);
        }
      });
    }
    FCU_FOR_LANES {                                          /* the TU's walk (leaf_luma_bits) on the candidate's running coder */
      if (lane < nc) {
        const int c = CAB_LANE0 + lane, mode = g_S.rd_mode[lane], cbf = g_S.vc_abs[lane] > 0;
        if (tu.part == 0 && P.slice_type != SLICE_I) { code_skip_flag(E, c, cu, 0); code_pred_mode(c, cu, 0); }
        if (tu.part == 0 && d == MAXDEPTH) cab_bin(c, partSize == SIZE_2Nx2N, CTX_PARTSIZE);
        if (tu.part == 0) code_luma_dir_bits(c, mode, g_S.preds);
        if (log2 != LOG2_MINTU && log2 != min_tu_log2_in_cu(d, partSize)) cab_bin(c, 0, CTX_SUBDIV + 5 - log2);
        cab_bin(c, cbf, CTX_CBF_LUMA + (tu.tr_depth == 0 ? 1 : 0));
        if (cbf) code_coeff_nxn<0>(c, G->p_qscan + lane, nc, g_S.vc_lsp[lane], log2, 0, coef_scan_idx(mode, log2, 0), 0, P, g_S.lane_abs[lane]);
        g_S.c64_dist[lane] += g_S.vc_dist[lane]; g_S.c64_cbf[lane][k] = (uint8_t)cbf;
        G->c64_distk[lane][k] = g_S.vc_dist[lane]; cab_copy1(&G->c64_state[lane][k], &g_S.cab[c]);
      }
    }
  }
  FCU_SERIAL {                                               /* strict '<' in candidate order (TEncSearch.cpp:2488-2510) */
    double best = FCU_MAX_DOUBLE; int bv = 0;
    for (int cand = 0; cand < nc; cand++) {
      const double c = rd_cost(P, (uint32_t)(g_S.cab[CAB_LANE0 + cand].frac >> 15), g_S.c64_dist[cand]);
      if (c < best) { best = c; bv = cand; }
    }
    g_S.pu_best_vc = bv; g_S.pu_best_cost = best; g_S.pu_best_dist = g_S.c64_dist[bv]; g_S.pu_best_mode = g_S.rd_mode[bv];
    g_S.c64_valid = 1;
  }
  {                                                          /* xSetIntraResultLumaQT + decision snapshot of the winner */
    const int bv = g_S.pu_best_vc;
    Yuv *reco = &G->reco[d][1 - g_S.reco_best_idx[d]];
    const int any = g_S.c64_cbf[bv][0] | g_S.c64_cbf[bv][1] | g_S.c64_cbf[bv][2] | g_S.c64_cbf[bv][3];
    FCU_FOR_LANES {
      for (int i = lane; i < CTU * CTU; i += 64) { cu->coef[0][i] = G->c64_coef[bv][i]; reco->y[i] = G->c64_rec[bv][i]; }
      for (int i = lane; i < root.nparts; i += 64) {
        G->tmp_tr_idx[i] = 1; G->tmp_cbf[i] = (uint8_t)((g_S.c64_cbf[bv][i / (root.nparts >> 2)] << 1) | any); G->tmp_tskip[i] = 0;
      }
    }
  }
}

/* ======================================================================================== */
/* estIntraPredLumaQT, TEncSearch.cpp:2178-2655                                               */
/* ======================================================================================== */
FCU_DEV int pu_trace_index(int depth, int nxn, int zidx) { return nxn ? 85 + zidx : (depth == 0 ? 0 : depth == 1 ? 1 + (zidx >> 6) : depth == 2 ? 5 + (zidx >> 4) : 21 + (zidx >> 2)); }   /* = fcu_pu_index (include/fcu.h) */
FCU_DEV FCU_NOINLINE void est_intra_pred_luma(CuObj *cu)
{
  const Env E = env_get(); cu = FCU_UNI(cu);
  Scratch *G = E.G;
  const int d = cu->depth_cu, partSize = cu->part_size[0];
  const int initTrDepth = partSize == SIZE_2Nx2N ? 0 : 1, numPU = 1 << (2 * initTrDepth), qNumParts = cu->nparts >> 2;
  uint32_t overallDistY = 0;
  TU root; tu_root(root, d);
  Yuv *recoT = &G->reco[d][1 - g_S.reco_best_idx[d]];
  for (int pu = 0; pu < numPU; pu++) {
    TU tu; if (initTrDepth == 0) tu = root; else tu_child(tu, root, pu, 0);
    const int partOffset = tu.part, N = 1 << tu.log2, log2 = tu.log2;
    FCU_SERIAL g_S.c64_valid = 0;
    { FCU_TIC(t_); rmd(cu, tu_key(tu)); FCU_TOC(E, t_, 0); }
    fcu_pu_trace *ptr = FCU_UNI(E.C->pu_trace ? E.C->pu_trace + (size_t)E.cur_ctu * FCU_PUS_PER_CTU + pu_trace_index(d, initTrDepth, cu->zidx + partOffset) : (fcu_pu_trace *)nullptr);
    if (ptr) FCU_SERIAL {                                    /* candidate list and CandCostList as the RMD leaves them */
      const int nRmd = k_rd_mode_num[log2 - 2];
      ptr->n_rmd = (uint8_t)nRmd; ptr->n_rd = (uint8_t)g_S.n_rd; ptr->pad = 0;
      for (int i = 0; i < 12; i++) ptr->rd_mode[i] = (uint8_t)(i < g_S.n_rd ? g_S.rd_mode[i] : 0);
      for (int i = 0; i < 8; i++) ptr->rmd_cost[i] = i < nRmd ? g_S.cand_cost[i] : 0.0;
    }
    const int singleTU = log2 <= LOG2_MAXTU;                /* first pass never splits such a PU */
    if (singleTU) { FCU_TIC(t_); pu_first_pass_batched(cu, tu_key(tu)); FCU_TOC(E, t_, 1); }
    else if (g_S.n_rd <= 5) { FCU_TIC(t_); pu_first_pass_64(cu, tu_key(tu)); FCU_TOC(E, t_, 1); }   /* 64x64: four 32x32 TUs, candidates side by side */
    else {                                                   /* more candidates than the pools hold: one after the other */
      FCU_SERIAL { g_S.pu_best_cost = FCU_MAX_DOUBLE; g_S.pu_best_mode = 0; g_S.pu_best_dist = 0; }
      const int nc = g_S.n_rd;
      for (int m = 0; m < nc; m++) {
        const int orgMode = g_S.rd_mode[m];
        FCU_FOR_LANES { for (int i = lane; i < tu.nparts; i += 64) cu->intra_dir[0][partOffset + i] = (uint8_t)orgMode; cab_copy(&g_S.cab[CAB_GOON], slot_ptr(E, d, CI_CURR_BEST), lane); if (lane == 0) { g_S.q_dist[0] = 0; g_S.q_cost[0] = 0; } }
        recur_luma_qt<0>(cu, tu_key(tu), 1);
        if (g_S.q_cost[0] < g_S.pu_best_cost) {
          FCU_SERIAL { g_S.pu_best_mode = orgMode; g_S.pu_best_dist = g_S.q_dist[0]; g_S.pu_best_cost = g_S.q_cost[0]; }
          set_intra_result_luma_qt(cu, tu_key(tu), recoT);
          FCU_FOR_LANES { for (int i = lane; i < tu.nparts; i += 64) { G->tmp_tr_idx[i] = cu->tr_idx[partOffset + i]; G->tmp_cbf[i] = cu->cbf[0][partOffset + i]; G->tmp_tskip[i] = cu->tskip[0][partOffset + i]; } }
        }
      }
    }
    /* best mode again with the full RQT (TEncSearch.cpp:2518-2586).  When the TU cannot split the
     * re-run reproduces the first-pass trial exactly (same snapshot, same inputs) and `<` keeps the
     * earlier result, so it is skipped. */
    if (log2 > min_tu_log2_in_cu(d, partSize)) {
      const int orgMode = g_S.pu_best_mode;
      FCU_FOR_LANES { for (int i = lane; i < tu.nparts; i += 64) cu->intra_dir[0][partOffset + i] = (uint8_t)orgMode; cab_copy(&g_S.cab[CAB_GOON], slot_ptr(E, d, CI_CURR_BEST), lane); if (lane == 0) { g_S.q_dist[0] = 0; g_S.q_cost[0] = 0; } }
      { FCU_TIC(t_); recur_luma_qt<0>(cu, tu_key(tu), 0, singleTU ? g_S.pu_best_vc : -1); FCU_TOC(E, t_, 4); }
      if (g_S.q_cost[0] < g_S.pu_best_cost) {
        FCU_SERIAL { g_S.pu_best_dist = g_S.q_dist[0]; g_S.pu_best_cost = g_S.q_cost[0]; }
        set_intra_result_luma_qt(cu, tu_key(tu), recoT);
        FCU_FOR_LANES { for (int i = lane; i < tu.nparts; i += 64) { G->tmp_tr_idx[i] = cu->tr_idx[partOffset + i]; G->tmp_cbf[i] = cu->cbf[0][partOffset + i]; G->tmp_tskip[i] = cu->tskip[0][partOffset + i]; } }
      }
    }
    overallDistY += g_S.pu_best_dist;
    const int bestMode = g_S.pu_best_mode;
    if (ptr) FCU_SERIAL { ptr->best_mode = (uint8_t)g_S.pu_best_mode; ptr->best_dist = g_S.pu_best_dist; ptr->best_cost = g_S.pu_best_cost; ptr->valid = 1; }
    FCU_FOR_LANES {
      for (int i = lane; i < tu.nparts; i += 64) {
        cu->tr_idx[partOffset + i] = G->tmp_tr_idx[i]; cu->cbf[0][partOffset + i] = G->tmp_cbf[i]; cu->tskip[0][partOffset + i] = G->tmp_tskip[i];
        cu->intra_dir[0][partOffset + i] = (uint8_t)bestMode;
      }
      if (pu != numPU - 1) {
        uint8_t *p = E.C->rec[0] + (cu->y + tu.y) * E.C->stride[0] + cu->x + tu.x; const int rs = E.C->stride[0];
        for (int i = lane; i < N * N; i += 64) p[(i >> log2) * rs + (i & (N - 1))] = recoT->y[(tu.y + (i >> log2)) * 64 + tu.x + (i & (N - 1))];
      }
    }
  }
  if (numPU > 1) {
    FCU_SERIAL {
      uint8_t cy = 0, cu1 = 0, cv = 0;
      for (int p = 0, idx = 0; p < 4; p++, idx += qNumParts) { cy |= (cu->cbf[0][idx] >> 1) & 1; cu1 |= (cu->cbf[1][idx] >> 1) & 1; cv |= (cu->cbf[2][idx] >> 1) & 1; }
      for (int o = 0; o < 4 * qNumParts; o++) { cu->cbf[0][o] |= cy; cu->cbf[1][o] |= cu1; cu->cbf[2][o] |= cv; }
    }
  }
  FCU_FOR_LANES { cab_copy(&g_S.cab[CAB_GOON], slot_ptr(E, d, CI_CURR_BEST), lane); if (lane == 0) cu->dist = overallDistY; }
}

/* ======================================================================================== */
/* chroma search, the five modes side by side (estIntraPredChromaQT + xRecurIntraChroma-      */
/* CodingQT, TEncSearch.cpp:1916-2120,2661-2810).  Every mode restarts from [d][CURR_BEST]    */
/* (:2713) and only reads its OWN reconstruction inside the CU, so the modes are independent:  */
/* the luma TU tree is walked once, and at every chroma leaf the wave runs prediction /         */
/* transform / reconstruction for all modes, RDOQ and bit counting one mode (x transform-skip   */
/* variant) per lane on lane-private coders.                                                   */
/* ======================================================================================== */
FCU_DEV void chroma_leaf_refs5(const Env E, const CuObj *cu, int comp, int px, int py, int log2)
{
  /* reference samples of one chroma block for the five modes: outside the CU from the picture, inside from
   * the mode's own overlay (what PicYuvRec would hold during that mode's trial, TEncSearch.cpp:1374) */
  Scratch *G = E.G;
  const int N = 1 << log2, total = 4 * N + 1, lx0 = px << 1, ly0 = py << 1;
  const uint8_t *rec = E.C->rec[comp]; const int stride = E.C->stride[comp];
  const int cx0 = cu->x >> 1, cy0 = cu->y >> 1, cs = (CTU >> cu->depth_cu) >> 1;
  uint8_t *avail = (uint8_t *)g_S.colsum;
  FCU_FOR_LANES {
    for (int i = lane; i < total; i += 64) {
      int a, x, y;
      if (i < 2 * N) { y = py + 2 * N - 1 - i; x = px - 1; a = unit_available(E, lx0 - 4, (y / 2 * 2) << 1, lx0, ly0); }
      else if (i == 2 * N) { y = py - 1; x = px - 1; a = unit_available(E, lx0 - 4, ly0 - 4, lx0, ly0); }
      else { x = px + i - 2 * N - 1; y = py - 1; a = unit_available(E, (x / 2 * 2) << 1, ly0 - 4, lx0, ly0); }
      avail[i] = (uint8_t)a;
      if (a) {
        const int inside = x >= cx0 && x < cx0 + cs && y >= cy0 && y < cy0 + cs;
        for (int m = 0; m < 5; m++) {
          const uint8_t *ov = comp == 1 ? G->cm[m].u : G->cm[m].v;
          g_S.ref5[m][i] = inside ? ov[(y - cy0) * 32 + (x - cx0)] : rec[y * stride + x];
        }
      }
    }
  }
  FCU_FOR_LANES {
    for (int i = lane; i < total; i += 64) {
      if (!avail[i]) {
        int j = i - 1;
        while (j >= 0 && !avail[j]) j--;
        if (j < 0) { j = i + 1; while (j < total && !avail[j]) j++; }
        for (int m = 0; m < 5; m++) g_S.ref5b[m][i] = (j < total) ? g_S.ref5[m][j] : 128;
      }
    }
  }
  FCU_FOR_LANES { for (int i = lane; i < total; i += 64) if (!avail[i]) for (int m = 0; m < 5; m++) g_S.ref5[m][i] = g_S.ref5b[m][i]; }
  FCU_FOR_LANES { if (lane < 5) { int sum = 0; for (int i = 0; i < N; i++) sum += g_S.ref5[lane][2 * N + 1 + i] + g_S.ref5[lane][2 * N - 1 - i]; g_S.dc5[lane] = (sum + N) >> (log2 + 1); } }
}

/* xGetIntraBitsQT(rTu, false, true) for mode slot m from the lane-private coder c (serial, one lane) */
FCU_DEV FCU_NOINLINE uint32_t chroma_tree_bits(int c, const CuObj *cu, int m, int mode, int16_t *absbuf)
{
  const Env E = env_get(); cu = FCU_UNI(cu);
  FCU_IN_LDS(absbuf);
  const ChromaModeBuf *B = &E.G->cm[m];
  FCU_EMU_RELAX(1);
  cab_reset_bits(c);
  code_intra_dir_chroma(c, mode);
  /* the five lanes (modes) walk the same luma TU tree in lockstep: they share the LDS walker stack (same values) */
  TU *st = g_S.wk_st; int *ci = g_S.wk_ci;
  for (int pass = 0; pass < 3; pass++) {                      /* 0: subdiv/cbf walk, 1: Cb coefficients, 2: Cr coefficients */
    int sp = 0; tu_root(st[0], cu->depth_cu); ci[0] = -1;
    while (sp >= 0) {
      const TU tu = st[sp];
      if (ci[sp] < 0) {
        const int subdiv = cu->tr_idx[tu.part] > tu.tr_depth;
        if (pass == 0) {
          for (int k = 0; k < 2; k++)
            if (tu.c_code_all && (tu.tr_depth == 0 || ((B->cbf[k][tu.part] >> (tu.tr_depth - 1)) & 1))) {
              const int lowest = tu.tr_depth + ((subdiv && !(tu.cw >= 8)) ? 1 : 0);
              cab_bin(c, (B->cbf[k][tu_part_c(tu)] >> lowest) & 1, CTX_CBF_CHROMA + tu.tr_depth);
            }
        } else if (!subdiv) {
          const int k = pass - 1;
          if (tu.cw != 0 && ((B->cbf[k][tu.part] >> tu.tr_depth) & 1)) {
            const int log2 = ilog2(tu.cw), pc = tu_part_c(tu);
            const int fmode = mode == DM_CHROMA ? cu->intra_dir[0][pc & ~3] : mode;
            code_coeff_nxn<0>(c, B->coef[k] + tu.off_c, 1, -1, log2, 1 + k, coef_scan_idx(fmode, log2, 1 + k), B->tskip[k][pc], E.C->p, absbuf);
          }
        }
        if (!subdiv) { sp--; continue; }
        ci[sp] = 0;
      }
      if (ci[sp] >= 4) { sp--; continue; }
      { TU ch; tu_child(ch, st[sp], ci[sp], 0); ci[sp]++; sp++; st[sp] = ch; ci[sp] = -1; }
    }
  }
  FCU_EMU_RELAX(0);
  return cab_bits(c);
}

/* one chroma leaf of the TU tree: both components, the five modes (x transform-skip variant) side by side */
FCU_DEV FCU_NOINLINE void chroma_leaf_trials(CuObj *cu, uint32_t tu_k)
{
  const Env E = env_get(); cu = FCU_UNI(cu); const TU tu = tu_of_key(FCU_UNI(tu_k));
  Scratch *G = E.G; const Params &P = E.C->p;
  const int d = cu->depth_cu;
  const int N = tu.cw, log2 = ilog2(N), n2 = N * N, trDepth = tu.tr_depth;
  int checkTS = P.transform_skip && N <= 4;
  if (P.ts_fast) {
    checkTS = checkTS && (tu.log2 == 2);
    if (checkTS) { int nb = 0; const int maxp = tu.part + (tu.c_code_all ? 1 : 4); for (int p = tu.part; p < maxp; p++) nb += cu->tskip[0][p]; checkTS = checkTS && (nb > 0); }
  }
  const int tsv = checkTS ? 2 : 1, tss = tsv - 1;                /* v / tsv == v >> tss, v % tsv == v & tss */
  /* Without a transform-skip trial nothing is coded between the two components (the bits are counted on the whole tree
   * afterwards), so Cb and Cr of a mode are priced against the same coder state and are independent of each other: both
   * go through one round, ten variants side by side.  With the trial the coder moves on after Cb's winner
   * (TEncSearch.cpp:1985-2058) and the components stay in sequence; so do 32x32 blocks, which would not fit the pools. */
  const int perRound = (tsv == 1 && 10 * n2 <= POOL) ? 2 : 1;
  const int nm = 5 * perRound, nvc = nm * tsv;                  /* prediction blocks / variants of a round */
  const int qbits = rdoq_qbits(log2, P.qp_c), qscale = k_quant_scales[P.qp_c % 6];
  const int subPart = tu_part_c(tu), nPartsC = tu_nparts_c(tu);
  const int lumaDir = cu->intra_dir[0][tu.part & ~3];
  for (int comp0 = 1; comp0 < 3; comp0 += perRound) {
    const int px = (cu->x >> 1) + tu.cx, py = (cu->y >> 1) + tu.cy;
    for (int cc = 0; cc < perRound; cc++) {
      const int comp = comp0 + cc;
      const uint8_t *org = yuv_plane(&G->org[d], comp) + tu.cy * 32 + tu.cx;
      chroma_leaf_refs5(E, cu, comp, px, py, log2);
      FCU_FOR_LANES {
        for (int i = lane; i < 5 * n2; i += 64) {
          const int m = i / n2, p = i - m * n2, y = p >> log2, x = p & (N - 1);
          const int mode = g_S.c_modes[m] == DM_CHROMA ? lumaDir : g_S.c_modes[m];
          const int pr = pred_pixel(g_S.ref5[m], log2, mode, 0, g_S.dc5[m], x, y);
          G->p_pred[cc * 5 * n2 + i] = (uint8_t)pr; G->p_resi[cc * 5 * n2 + i] = (int16_t)(org[y * 32 + x] - pr);
        }
      }
    }
    FCU_FOR_LANES { if (lane < nvc) g_S.vc_last[lane] = -1; by_log2(log2, [&](auto L) { constexpr int LG = decltype(L)::value; RowCache<(1 << LG), int16_t> rc; for (int i = lane; i < nm * n2; i += 64) { const int b = i / n2; G->p_tmp[i] = fwd1<LG>(rc, G->p_resi + b * n2, 0, i - b * n2); } }); }
    FCU_FOR_LANES {
      by_log2(log2, [&](auto L) { constexpr int LG = decltype(L)::value; RowCache<(1 << LG), int32_t> rc;
        for (int i = lane; i < nvc * n2; i += 64) {
          const int v = i / n2, p = i - v * n2, b = (v >> tss), ts = (v & tss), m = b % 5, comp = comp0 + b / 5;
          const int mode = g_S.c_modes[m] == DM_CHROMA ? lumaDir : g_S.c_modes[m];
          const uint16_t *iscan = k_iscan + k_scan_off[coef_scan_idx(mode, log2, comp) * 4 + log2 - 2];
          const int32_t t = ts ? ((int32_t)G->p_resi[b * n2 + p] << (15 - 8 - log2)) : fwd2<LG>(rc, G->p_tmp + b * n2, 0, p);
          const int sp = iscan[p]; const int32_t ld = level_double(t, qscale, qbits);
          G->p_lscan[sp * nvc + v] = ld;
          if (level_nonzero(ld, qbits)) FCU_ATOMIC_MAX(&g_S.vc_last[v], sp);
        }
      });
    }
    FCU_TIC(t13_);
    FCU_FOR_LANES {                                      /* RDOQ from the mode's coder state (its QT_TRAFO_ROOT) */
      if (lane < nvc) {
        const int b = (lane >> tss), m = b % 5, comp = comp0 + b / 5;
        const int mode = g_S.c_modes[m] == DM_CHROMA ? lumaDir : g_S.c_modes[m];
        RdoqRec *rrec = G->r_rec + lane; double *rcg = G->r_cg + lane;
        const RdoqOut o = ((lane & tss) ? P.rdoq_ts : P.rdoq)
          ? rdoq<0, 0>(CAB_LANE0 + m, G->p_lscan + lane, G->p_qscan + lane, nvc, g_S.vc_last[lane], log2, 1 /* Cb and Cr share every parameter the call reads; the argument is wave-uniform */, coef_scan_idx(mode, log2, comp), CTX_CBF_CHROMA + trDepth, P, rrec, rcg)
          : quant_plain(G->p_lscan + lane, G->p_qscan + lane, nvc, g_S.vc_last[lane], log2, 1, P);
        g_S.vc_abs[lane] = o.abs_sum; g_S.vc_lsp[lane] = o.last;
        g_S.vc_dist[lane] = 0;
      }
      if (lane == 0) E.C->n_tu_trials += (unsigned long long)nvc;
    }
    FCU_TOC(E, t13_, 13);
    FCU_FOR_LANES {                                      /* levels back to raster order + dequantisation */
      const DeqParams dq = deq_params(log2, P.qp_c);
      for (int i = lane; i < nvc * n2; i += 64) {
        const int v = i / n2, p = i - v * n2, b = (v >> tss), m = b % 5, comp = comp0 + b / 5;
        const int mode = g_S.c_modes[m] == DM_CHROMA ? lumaDir : g_S.c_modes[m];
        const uint16_t *iscan = k_iscan + k_scan_off[coef_scan_idx(mode, log2, comp) * 4 + log2 - 2];
        const int sp = iscan[p];
        const int q = (g_S.vc_abs[v] > 0 && (sp >> 4) <= (g_S.vc_last[v] >> 4)) ? G->p_qscan[sp * nvc + v] : 0;
        G->p_tmp[((v & tss)) ? i : v * n2 + tr_index(p, log2)] = dequant1(q, dq);
      }
    }
    FCU_FOR_LANES {
      by_log2(log2, [&](auto L) { constexpr int LG = decltype(L)::value; RowCache<(1 << LG), int32_t> rc;
        for (int i = lane; i < nvc * n2; i += 64) {
          const int v = i / n2, p = i - v * n2, ts = (v & tss);
          if (ts) { const int sft = 15 - 8 - log2; G->p_tcoef[i] = (G->p_tmp[i] + (1 << (sft - 1))) >> sft; }
          else G->p_tcoef[i] = inv1<LG>(rc, G->p_tmp + v * n2, 0, p);
        }
      });
    }
    FCU_FOR_LANES {
      by_log2(log2, [&](auto L) {
        uint32_t acc = 0;
        for (int i = lane; i < nvc * n2; i += 64) {
          const int v = i / n2, p = i - v * n2, b = (v >> tss), ts = (v & tss), y = p >> log2, x = p & (N - 1);
          const uint8_t *org = yuv_plane(&G->org[d], comp0 + b / 5) + tu.cy * 32 + tu.cx;
          int res = 0;
          if (g_S.vc_abs[v] > 0) res = ts ? (int16_t)G->p_tcoef[i] : inv2<decltype(L)::value>(G->p_tcoef + v * n2, 0, p);
          const int r = clip8(G->p_pred[b * n2 + p] + res);
          G->p_rec[i] = (uint8_t)r;
          const int e = org[y * 32 + x] - r;
          FCU_DIST_ADD(acc, v, e, i, n2);
        }
      });
    }
    FCU_FOR_LANES {                                      /* per mode: transform-skip decision (TEncSearch.cpp:1985-2058) */
      if (lane < 5) {
        const int m = lane; int bestTs = 0;
        const int mode = g_S.c_modes[m] == DM_CHROMA ? lumaDir : g_S.c_modes[m];
        if (tsv == 2) {                                      /* one component per round: variants 2m (coded) and 2m+1 (skipped) */
          const int comp = comp0;
          uint32_t dsel = (uint32_t)(P.chroma_weight * (double)g_S.vc_dist[m * 2]);
          const uint32_t d0 = dsel, d1 = (uint32_t)(P.chroma_weight * (double)g_S.vc_dist[m * 2 + 1]);
          const int c0 = CAB_LANE0 + 5 + m, c1 = CAB_LANE0 + 10 + m;
          cab_copy1(&g_S.cab[c0], &g_S.cab[CAB_LANE0 + m]); cab_reset_bits(c0);
          if (g_S.vc_abs[m * 2] > 0) code_coeff_nxn<0>(c0, G->p_qscan + m * 2, nvc, g_S.vc_lsp[m * 2], log2, comp, coef_scan_idx(mode, log2, comp), 0, P, g_S.lane_abs[lane]);
          const double cost0 = rd_cost(P, cab_bits(c0), d0);
          double cost1 = FCU_MAX_DOUBLE;
          if (g_S.vc_abs[m * 2 + 1] > 0) {
            cab_copy1(&g_S.cab[c1], &g_S.cab[CAB_LANE0 + m]); cab_reset_bits(c1);
            code_coeff_nxn<0>(c1, G->p_qscan + m * 2 + 1, nvc, g_S.vc_lsp[m * 2 + 1], log2, comp, coef_scan_idx(mode, log2, comp), 1, P, g_S.lane_abs[lane]);
            cost1 = rd_cost(P, cab_bits(c1), d1);
          }
          if (cost1 < cost0) { bestTs = 1; dsel = d1; cab_copy1(&g_S.cab[CAB_LANE0 + m], &g_S.cab[c1]); } else cab_copy1(&g_S.cab[CAB_LANE0 + m], &g_S.cab[c0]);
          g_S.cm_dist[m] += dsel;
        } else {
          for (int cc = 0; cc < perRound; cc++) g_S.cm_dist[m] += (uint32_t)(P.chroma_weight * (double)g_S.vc_dist[cc * 5 + m]);
        }
        g_S.uni[m] = bestTs;
      }
    }
    FCU_FOR_LANES {                                      /* publish the chosen variant into the mode's buffers */
      for (int i = lane; i < nm * n2; i += 64) {
        const int b = i / n2, p = i - b * n2, m = b % 5, comp = comp0 + b / 5, v = b * tsv + g_S.uni[m], cbf = g_S.vc_abs[v] > 0;
        G->cm[m].coef[comp - 1][tu.off_c + p] = (cbf && (p >> 4) <= (g_S.vc_last[v] >> 4)) ? G->p_qscan[p * nvc + v] : (int16_t)0;
        uint8_t *ov = comp == 1 ? G->cm[m].u : G->cm[m].v;
        ov[(tu.cy + (p >> log2)) * 32 + tu.cx + (p & (N - 1))] = G->p_rec[v * n2 + p];
      }
      for (int i = lane; i < nm * nPartsC; i += 64) {
        const int b = i / nPartsC, p = i - b * nPartsC, m = b % 5, comp = comp0 + b / 5, v = b * tsv + g_S.uni[m];
        G->cm[m].cbf[comp - 1][subPart + p] = (uint8_t)((g_S.vc_abs[v] > 0 ? 1 : 0) << trDepth);
        G->cm[m].tskip[comp - 1][subPart + p] = (uint8_t)g_S.uni[m];
      }
    }
  }
}

FCU_DEV FCU_NOINLINE void est_intra_pred_chroma(CuObj *cu)
{
  const Env E = env_get(); cu = FCU_UNI(cu);
  Scratch *G = E.G; const Params &P = E.C->p;
  const int d = cu->depth_cu, n = cu->nparts, cs = (CTU >> d) >> 1;
  /* getAllowedChromaDir, TComDataCU.cpp:1509-1533; the list lives in LDS (it is indexed per element below) */
  FCU_FOR_LANES {
    if (lane == 0) {
      int ml[5] = { PLANAR, VER, HOR, DC, DM_CHROMA };
      const int luma = cu->intra_dir[0][0]; for (int i = 0; i < 4; i++) if (luma == ml[i]) { ml[i] = 34; break; }
      for (int i = 0; i < 5; i++) g_S.c_modes[i] = ml[i];
    }
    if (lane < 5) { cab_copy1(&g_S.cab[CAB_LANE0 + lane], slot_ptr(E, d, CI_CURR_BEST)); g_S.cm_dist[lane] = 0; }
    for (int i = lane; i < 5 * n; i += 64) { const int m = i / n, p = i - m * n; G->cm[m].cbf[0][p] = G->cm[m].cbf[1][p] = 0; G->cm[m].tskip[0][p] = G->cm[m].tskip[1][p] = 0; }
  }
  /* ---- walk the luma TU tree; chroma leaves in z-order ---- */
  TU *st = g_S.wk_st; int *ci = g_S.wk_ci; int sp = 0;
  tu_root(st[0], d); ci[0] = -1;
  while (sp >= 0) {
    if (ci[sp] < 0) {
      const TU tu = st[sp];
      const int subdiv = cu->tr_idx[tu.part] > tu.tr_depth;
      if (!subdiv) {
        if (tu.cw != 0) chroma_leaf_trials(cu, tu_key(tu));
        sp--; continue;
      }
      ci[sp] = 0;
    }
    if (ci[sp] >= 4) { sp--; continue; }
    { TU ch; tu_child(ch, st[sp], ci[sp], 0); ci[sp]++; sp++; st[sp] = ch; ci[sp] = -1; }
  }
  /* ---- per mode: CBF propagation up the tree, bits, cost ---- */
  FCU_TIC(t14_);
  FCU_FOR_LANES {
    if (lane < 5) {
      const int m = lane; ChromaModeBuf *B = &G->cm[m];
      for (int t = 2; t >= 0; t--) {                            /* nodes of relative depth t, deepest first */
        const int np = n >> (2 * t);
        if (np < 4) continue;
        for (int part = 0; part < n; part += np) {
          if (!(cu->tr_idx[part] > t)) continue;
          for (int k = 0; k < 2; k++) {
            int any = 0;
            for (int q = 0; q < 4; q++) any |= (B->cbf[k][part + q * (np >> 2)] >> (t + 1)) & 1;
            if (any) for (int o = 0; o < np; o++) B->cbf[k][part + o] |= (uint8_t)(1 << t);
          }
        }
      }
      const int c = CAB_LANE0 + 5 + m;
      cab_copy1(&g_S.cab[c], slot_ptr(E, d, CI_CURR_BEST));
      g_S.vc_bits[m] = chroma_tree_bits(c, cu, m, g_S.c_modes[m], g_S.lane_abs[lane]);
      g_S.vc_cost[m] = rd_cost(P, g_S.vc_bits[m], g_S.cm_dist[m]);
    }
  }
  FCU_TOC(E, t14_, 14);
  FCU_SERIAL {
    double best = FCU_MAX_DOUBLE; int bm = 0;
    for (int m = 0; m < 5; m++) if (g_S.vc_cost[m] < best) { best = g_S.vc_cost[m]; bm = m; }
    g_S.c_best_mode = bm;
  }
  {
    const int bm = g_S.c_best_mode; const ChromaModeBuf *B = &G->cm[bm];
    Yuv *recoT = &G->reco[d][1 - g_S.reco_best_idx[d]];
    FCU_FOR_LANES {
      for (int i = lane; i < cs * cs; i += 64) {
        cu->coef[1][i] = B->coef[0][i]; cu->coef[2][i] = B->coef[1][i];
        const int o = (i / cs) * 32 + (i % cs); recoT->u[o] = B->u[o]; recoT->v[o] = B->v[o];
      }
      for (int i = lane; i < n; i += 64) {
        cu->cbf[1][i] = B->cbf[0][i]; cu->cbf[2][i] = B->cbf[1][i]; cu->tskip[1][i] = B->tskip[0][i]; cu->tskip[2][i] = B->tskip[1][i];
        cu->intra_dir[1][i] = (uint8_t)g_S.c_modes[bm];
      }
      if (lane == 0) cu->dist += g_S.cm_dist[bm];
      cab_copy(&g_S.cab[CAB_GOON], slot_ptr(E, d, CI_CURR_BEST), lane);
    }
  }
}

/* ======================================================================================== */
/* CU level: xCheckRDCostIntra / xCheckBestMode / xCompressCU (TEncCu.cpp:460-1616,2064-2255)  */
/* ======================================================================================== */
FCU_DEV CuObj *cu_best(const Env E, int d) { return &E.G->cu[d][g_S.best_idx[d]]; }
FCU_DEV CuObj *cu_temp(const Env E, int d) { return &E.G->cu[d][1 - g_S.best_idx[d]]; }

FCU_DEV FCU_NOINLINE void check_best_mode(int d)
{
  const Env E = env_get(); d = FCU_UNI(d);
  const int change = cu_temp(E, d)->cost < cu_best(E, d)->cost;
  FCU_FOR_LANES {
    if (change) { cab_copy(slot_ptr(E, d, CI_NEXT_BEST), slot_ptr(E, d, CI_TEMP_BEST), lane); if (lane == 0) { g_S.best_idx[d] = 1 - g_S.best_idx[d]; g_S.reco_best_idx[d] = 1 - g_S.reco_best_idx[d]; } }
  }
}
FCU_DEV FCU_NOINLINE void check_rd_cost_intra(int d, int partSize)
{
  const Env E = env_get(); d = FCU_UNI(d); partSize = FCU_UNI(partSize);
  Scratch *G = E.G; const Params &P = E.C->p;
  CuObj *cu = cu_temp(E, d);
  const int n = cu->nparts, s = CTU >> d;
  FCU_FOR_LANES { for (int i = lane; i < n; i += 64) { cu->part_size[i] = (int8_t)partSize; cu->pred_mode[i] = MODE_INTRA; } }
  est_intra_pred_luma(cu);
  {
    const Yuv *recoT = &G->reco[d][1 - g_S.reco_best_idx[d]];
    FCU_FOR_LANES { uint8_t *p = E.C->rec[0] + cu->y * E.C->stride[0] + cu->x; const int rs = E.C->stride[0]; for (int i = lane; i < s * s; i += 64) p[(i / s) * rs + (i % s)] = recoT->y[(i / s) * 64 + (i % s)]; }
  }
  { FCU_TIC(t_); est_intra_pred_chroma(cu); FCU_TOC(E, t_, 6); }
  FCU_TIC(t7_);
  FCU_FOR_LANES {
    if (lane == 0) {
      cab_reset_bits((CAB_GOON));
      encode_cu_syntax(E, (CAB_GOON), cu, 0, d);
      cu->bits = cab_bits((CAB_GOON)); cu->bins = g_S.cab[CAB_GOON].bins;
      cu->cost = rd_cost(P, cu->bits, cu->dist);
    }
  }
  FCU_FOR_LANES cab_copy(slot_ptr(E, d, CI_TEMP_BEST), &g_S.cab[CAB_GOON], lane);
  FCU_TOC(E, t7_, 7);
  check_best_mode(d);
}

/* xCheckRDCostInter, TEncCu.cpp:2025-2062 */
FCU_DEV FCU_NOINLINE void check_rd_cost_inter(int d, int partSize, int useMrg)
{
  const Env E = env_get(); d = FCU_UNI(d); partSize = FCU_UNI(partSize); useMrg = FCU_UNI(useMrg);
  CuObj *cu = cu_temp(E, d);
  const int n = cu->nparts;
  FCU_FOR_LANES { for (int i = lane; i < n; i += 64) { cu->part_size[i] = (int8_t)partSize; cu->pred_mode[i] = MODE_INTER; } }
  pred_inter_search(cu, partSize, useMrg);
  { FCU_TIC(p_); encode_res_and_calc_rd_inter_cu(cu, 0); FCU_ITOC(E, p_, 9); }
  check_best_mode(d);
}
/* xCheckRDCostMerge2Nx2N, TEncCu.cpp:1900-2018 (early skip detection off) */
FCU_DEV FCU_NOINLINE void check_rd_cost_merge_2nx2n(int d)
{
  const Env E = env_get(); d = FCU_UNI(d);
  Scratch *G = E.G; const Params &P = E.C->p;
  const CuObj *t0 = cu_temp(E, d);
  const int x = t0->x, y = t0->y, zidx = t0->zidx, n = t0->nparts;
  FCU_FOR_LANES { CuObj *cu = cu_temp(E, d); for (int i = lane; i < n; i += 64) cu->part_size[i] = SIZE_2Nx2N; if (lane < 5) g_S.mrg_buf[lane] = 0; if (lane == 0) g_S.best_is_skip = 0; }
  FCU_SERIAL merge_candidates(cu_temp(E, d), SIZE_2Nx2N, 0);
  int cmv[5][2], cref[5];
  for (int c = 0; c < 5; c++) { cmv[c][0] = FCU_UNI(g_S.mrg_mv[c][0]); cmv[c][1] = FCU_UNI(g_S.mrg_mv[c][1]); cref[c] = FCU_UNI(g_S.mrg_ref[c]); }
  const int nc = P.max_merge_cand;
  for (int noRes = 0; noRes < 2; noRes++)
    for (int c = 0; c < nc; c++) {
      if (noRes == 1 && FCU_UNI(g_S.mrg_buf[c]) == 1) continue;
      if (FCU_UNI(g_S.best_is_skip) && noRes == 0) continue;
      CuObj *cu = cu_temp(E, d);
      FCU_FOR_LANES {
        for (int i = lane; i < n; i += 64) { cu->pred_mode[i] = MODE_INTER; cu->part_size[i] = SIZE_2Nx2N; }
        pu_set_motion(cu, SIZE_2Nx2N, 0, lane, cmv[c][0], cmv[c][1], cref[c]); pu_set_info(cu, SIZE_2Nx2N, 0, lane, 1, c, 0, 0, -1);
      }
      { FCU_TIC(p_); mc_pu(cu, SIZE_2Nx2N, 0, &G->predt[d], 0); FCU_ITOC(E, p_, 8); }
      { FCU_TIC(p_); encode_res_and_calc_rd_inter_cu(cu, noRes); FCU_ITOC(E, p_, noRes ? 12 : 11); }
      FCU_SERIAL { if (noRes == 0 && !qt_root_cbf(cu, 0)) g_S.mrg_buf[c] = 1; }
      check_best_mode(d);
      cu_init(cu_temp(E, d), d, x, y, zidx);
      FCU_SERIAL { if (P.fdm && !g_S.best_is_skip) g_S.best_is_skip = !qt_root_cbf(cu_best(E, d), 0); }
    }
}

/* fork hooks of xCompressCU for its default control (YSGlobalControl, tools_YS.cpp:4-58: Naive model on N_OBF).
 * Num_OBF = 4x4 blocks of the CU with a positive OBF count (TEncCu.cpp:585-600); the Naive label is Skip2Nx2N when
 * there is one, TerminateCU when there is none (tools_YS.cpp:686-695, TEncCu.cpp:670-678). */
FCU_DEV FCU_NOINLINE int cu_num_obf(int x, int y, int s)
{
  const Env E = env_get(); x = FCU_UNI(x); y = FCU_UNI(y); s = FCU_UNI(s);
  FCU_SERIAL g_S.dec_cnt = 0;
  FCU_FOR_LANES {
    const int q = s >> 2; int n = 0;
    for (int i = lane; i < q * q; i += 64) n += E.C->obf[((y >> 2) + i / q) * E.C->obf_stride + (x >> 2) + i % q] > 0;
    FCU_WAVE_ADD((uint32_t *)&g_S.dec_cnt, (uint32_t)n);
  }
  return FCU_UNI(g_S.dec_cnt);
}
/* countTFPN + countRDLoss (tools_YS.cpp:968-986; ResultType order TP,FP,TN,FN,FPLoss,FNLoss, globals_YS.h:81-89) */
FCU_DEV void count_verify(Chain *C, int d, int predictSkip, int partitionTrue, double j0, double j1)
{
  const double loss = fabs(j0 - j1);
  if (predictSkip) { C->ver[d][partitionTrue ? 0 : 1] += 1.0; if (!partitionTrue) C->ver[d][4] += loss; }
  else { C->ver[d][partitionTrue ? 3 : 2] += 1.0; if (partitionTrue) C->ver[d][5] += loss; }
}

/* -DFCU_PROFILE -DFCU_PROFILE_DEPTH: prof[11 + D] = ticks spent at CU depth D without its sub-CUs (what a wave-per-depth
 * frontier could overlap; measured on MI355X: 13 / 17 / 21 / 48 % for depths 0..3) */
#if defined(FCU_PROFILE) && defined(FCU_PROFILE_DEPTH) && !defined(FCU_EMU)
#define FCU_DTIC(v) long long v = clock64()
#define FCU_DADD(E_, v, idx, sign) do { if (threadIdx.x == 0) (E_).C->prof[idx] += (unsigned long long)((sign) * (clock64() - v)); } while (0)
#else
#define FCU_DTIC(v) do { } while (0)
#define FCU_DADD(E_, v, idx, sign) do { } while (0)
#endif
template <int D>
FCU_DEV FCU_NOINLINE void compress_cu()
{
  const Env E = env_get();
  FCU_DTIC(dt_);
  Scratch *G = E.G; const Params &P = E.C->p;
  const CuObj *b0 = cu_best(E, D);
  const int x = b0->x, y = b0->y, zidx = b0->zidx, s = CTU >> D;
  const int boundary = !((x + s - 1 < P.width) && (y + s - 1 < P.height));
  const int state = E.C->dec_state;
  int skip2Nx2N = 0, earlyTerminate = 0, predictSkip = 0;
  if (!boundary && state != DEC_TRAINING) {
    const int nobf = cu_num_obf(x, y, s);
    predictSkip = nobf > 0;
    if (state == DEC_TESTING) {                              /* TEncCu.cpp:951-996 */
      if (E.C->sw_term[D]) earlyTerminate = !predictSkip;
      if (E.C->sw_skip[D]) skip2Nx2N = predictSkip;
      if (D == 3 && nobf > 0 && E.C->depth_exception) skip2Nx2N = earlyTerminate = 0;
    }
  }
  if (!boundary) {
    FCU_FOR_LANES {                                          /* source block -> L2-resident CU buffer */
      for (int i = lane; i < s * s; i += 64) G->org[D].y[(i / s) * 64 + (i % s)] = E.C->org[0][(y + i / s) * E.C->stride[0] + x + (i % s)];
      const int h = s / 2;
      for (int i = lane; i < h * h; i += 64) {
        G->org[D].u[(i / h) * 32 + (i % h)] = E.C->org[1][(y / 2 + i / h) * E.C->stride[1] + x / 2 + (i % h)];
        G->org[D].v[(i / h) * 32 + (i % h)] = E.C->org[2][(y / 2 + i / h) * E.C->stride[2] + x / 2 + (i % h)];
      }
    }
    cu_init(cu_temp(E, D), D, x, y, zidx);
    int tryIntra = 1;
    if (P.slice_type == SLICE_P) {                           /* inter candidates first (TEncCu.cpp:753-943; ESD / CFM off) */
      FCU_FOR_LANES { if (lane < MEMO_K) G->memo_valid[D][lane] = 0; if (lane == MEMO_K) G->memo_next[D] = 0; }
      { FCU_TIC(p_); check_rd_cost_merge_2nx2n(D); FCU_ITOC(E, p_, 0); }
      cu_init(cu_temp(E, D), D, x, y, zidx);
      { FCU_TIC(p_); check_rd_cost_inter(D, SIZE_2Nx2N, 0); FCU_ITOC(E, p_, 1); } cu_init(cu_temp(E, D), D, x, y, zidx);
      { FCU_TIC(p_); check_rd_cost_inter(D, SIZE_Nx2N, 0); FCU_ITOC(E, p_, 2); } cu_init(cu_temp(E, D), D, x, y, zidx);
      { FCU_TIC(p_); check_rd_cost_inter(D, SIZE_2NxN, 0); FCU_ITOC(E, p_, 3); } cu_init(cu_temp(E, D), D, x, y, zidx);
      if (P.amp && D < MAXDEPTH) {                             /* AMP with AMP_ENC_SPEEDUP + AMP_MRG (:836-943); deriveTestModeAMP (:381-430) */
        const CuObj *bb = cu_best(E, D);
        const int bps = FCU_UNI((int)bb->part_size[0]), bmrg = FCU_UNI((int)bb->merge_flag[0]), bskip = FCU_UNI((int)bb->skip[0]), par = FCU_UNI(G->par_ps[D]);
        int hor = 0, ver = 0, mhor = 0, mver = 0;
        if (bps == SIZE_2NxN) hor = 1;
        else if (bps == SIZE_Nx2N) ver = 1;
        else if (bps == SIZE_2Nx2N && !bmrg && !bskip) hor = ver = 1;
        if (par >= SIZE_2NxnU && par <= SIZE_nRx2N) mhor = mver = 1;
        if (par == SIZE_NONE) { if (bps == SIZE_2NxN) mhor = 1; else if (bps == SIZE_Nx2N) mver = 1; }
        if (bps == SIZE_2Nx2N && !bskip) mhor = mver = 1;
        if (s == 64) hor = ver = 0;
        if (hor || mhor) {
          check_rd_cost_inter(D, SIZE_2NxnU, !hor); cu_init(cu_temp(E, D), D, x, y, zidx);
          check_rd_cost_inter(D, SIZE_2NxnD, !hor); cu_init(cu_temp(E, D), D, x, y, zidx);
        }
        if (ver || mver) {
          check_rd_cost_inter(D, SIZE_nLx2N, !ver); cu_init(cu_temp(E, D), D, x, y, zidx);
          check_rd_cost_inter(D, SIZE_nRx2N, !ver); cu_init(cu_temp(E, D), D, x, y, zidx);
        }
      }
      const CuObj *b = cu_best(E, D);                          /* intra only when the best inter candidate has a residual (:1033-1036) */
      tryIntra = FCU_UNI((int)(b->cbf[0][0] | b->cbf[1][0] | b->cbf[2][0])) != 0;
    }
    { FCU_TIC(p_); if (!skip2Nx2N && tryIntra) check_rd_cost_intra(D, SIZE_2Nx2N);      /* :1040; skipped => the best cost stays MAX_DOUBLE (:1077) */
      FCU_ITOC(E, p_, 4); }
    cu_init(cu_temp(E, D), D, x, y, zidx);
    if (D == MAXDEPTH && !earlyTerminate && tryIntra) {      /* :1141-1143 */
      FCU_SERIAL { g_S.dec_j0 = cu_best(E, D)->cost; g_S.dec_flip = g_S.best_idx[D]; }
      check_rd_cost_intra(D, SIZE_NxN);
      FCU_SERIAL { g_S.dec_flip ^= g_S.best_idx[D]; g_S.dec_j1 = g_S.dec_flip ? cu_best(E, D)->cost : cu_temp(E, D)->cost; }   /* :1175-1183 */
      cu_init(cu_temp(E, D), D, x, y, zidx);
    }
    FCU_SERIAL {
      CuObj *best = cu_best(E, D);
      if (best->cost != FCU_MAX_DOUBLE) {                   /* fork: TEncCu.cpp:1224 */
        cab_reset_bits((CAB_GOON));
        code_split_flag(E, (CAB_GOON), best, 0, D);
        best->bits += cab_bits((CAB_GOON)); best->bins += g_S.cab[CAB_GOON].bins;
        best->cost = rd_cost(P, best->bits, best->dist);
      }
    }
  }
  cu_init(cu_temp(E, D), D, x, y, zidx);
  if constexpr (D < MAXDEPTH) {
   if (!earlyTerminate) {                                    /* bSubBranch = false, TEncCu.cpp:1257-1260 */
    const int nd = D + 1, hs = s >> 1, qn = NPART >> (2 * nd);
    if (P.amp) FCU_SERIAL { const CuObj *b = cu_best(E, D); G->par_ps[nd] = (boundary || b->pred_mode[0] != MODE_INTER) ? SIZE_NONE : b->part_size[0]; }   /* eParentPartSize, :1355-1363 */
    for (int i = 0; i < 4; i++) {
      const int sx = x + (i & 1) * hs, sy = y + (i >> 1) * hs;
      cu_init(&G->cu[nd][0], nd, sx, sy, zidx + i * qn);
      cu_init(&G->cu[nd][1], nd, sx, sy, zidx + i * qn);
      if (sx < P.width && sy < P.height) {
        FCU_FOR_LANES cab_copy(slot_ptr(E, nd, CI_CURR_BEST), i == 0 ? slot_ptr(E, D, CI_CURR_BEST) : slot_ptr(E, nd, CI_NEXT_BEST), lane);
        { FCU_DTIC(dc_); compress_cu<D + 1>(); FCU_DADD(E, dc_, 11 + D, -1); }   /* the child's time is not this depth's */
        cu_copy_part_from(cu_temp(E, D), cu_best(E, nd), i);
        {                                                    /* xCopyYuv2Tmp */
          const Yuv *src = &G->reco[nd][g_S.reco_best_idx[nd]]; Yuv *dst = &G->reco[D][1 - g_S.reco_best_idx[D]];
          FCU_FOR_LANES {
            for (int k = lane; k < hs * hs; k += 64) dst->y[((i >> 1) * hs + k / hs) * 64 + (i & 1) * hs + k % hs] = src->y[(k / hs) * 64 + k % hs];
            const int h = hs / 2;
            for (int k = lane; k < h * h; k += 64) { const int o = ((i >> 1) * h + k / h) * 32 + (i & 1) * h + k % h; dst->u[o] = src->u[(k / h) * 32 + k % h]; dst->v[o] = src->v[(k / h) * 32 + k % h]; }
          }
        }
      } else {
        cu_copy_to_pic(cu_best(E, nd));
        cu_copy_part_from(cu_temp(E, D), cu_best(E, nd), i);
      }
    }
    FCU_SERIAL {
      CuObj *t = cu_temp(E, D);
      if (!boundary) { cab_reset_bits((CAB_GOON)); code_split_flag(E, (CAB_GOON), t, 0, D); t->bits += cab_bits((CAB_GOON)); t->bins += g_S.cab[CAB_GOON].bins; }
      t->cost = rd_cost(P, t->bits, t->dist);
      if (skip2Nx2N) cu_best(E, D)->cost = FCU_MAX_DOUBLE;   /* :1446-1449 */
      g_S.dec_j0 = cu_best(E, D)->cost; g_S.dec_j1 = t->cost; g_S.dec_flip = g_S.best_idx[D];   /* J0, J1 :1450-1451 */
    }
    FCU_FOR_LANES cab_copy(slot_ptr(E, D, CI_TEMP_BEST), slot_ptr(E, nd, CI_NEXT_BEST), lane);
    check_best_mode(D);
    FCU_SERIAL g_S.dec_flip ^= g_S.best_idx[D];              /* bPartition_True = xCheckBestMode(...) */
   }
  }
  if (!boundary && state == DEC_VERIFYING) {                 /* TEncCu.cpp:1489-1497 */
    FCU_SERIAL count_verify(E.C, D, predictSkip, g_S.dec_flip, g_S.dec_j0, g_S.dec_j1);
  }
  cu_copy_to_pic(cu_best(E, D));
  copy_reco_to_pic(&G->reco[D][g_S.reco_best_idx[D]], x, y, s);
  FCU_DADD(E, dt_, 11 + D, 1);
}

/* ---- encodeCtu replay: xEncodeCU, TEncCu.cpp:1679-1778 (serial, iterative) --------------- */
FCU_DEV FCU_NOINLINE void encode_ctu(int c, const CuObj *ctu, int lastCtuOfSlice)
{
  const Env E = env_get(); c = FCU_UNI(c); ctu = FCU_UNI(ctu); lastCtuOfSlice = FCU_UNI(lastCtuOfSlice);
  const Params &P = E.C->p;
  int *stPart = g_S.wk_part, *stChild = g_S.wk_child; int sp = 0;
  stPart[0] = 0; stChild[0] = -1;
  while (sp >= 0) {
    const int depth = sp, part = stPart[sp];
    if (stChild[sp] < 0) {
      const int cx = ctu->x + part_x(part), cy = ctu->y + part_y(part), s = CTU >> depth;
      int boundary = 0;
      if (cx + s - 1 < P.width && cy + s - 1 < P.height) code_split_flag(E, c, ctu, part, depth); else boundary = 1;
      if (!((depth < ctu->depth[part] && depth < MAXDEPTH) || boundary)) {
        encode_cu_syntax(E, c, ctu, part, depth);
        const int rx = cx + s, ry = cy + s;                  /* finishCU / isLastSubCUOfCtu, TEncCu.cpp:1629-1645 */
        if (((rx % CTU) == 0 || rx == P.width) && ((ry % CTU) == 0 || ry == P.height) && !lastCtuOfSlice) cab_trm(c, 0);
        sp--; continue;
      }
      stChild[sp] = 0;
    }
    if (stChild[sp] >= 4) { sp--; continue; }
    {
      const int qn = (NPART >> (2 * depth)) >> 2, p = part + stChild[sp] * qn;
      stChild[sp]++;
      if (ctu->x + part_x(p) < P.width && ctu->y + part_y(p) < P.height) { sp++; stPart[sp] = p; stChild[sp] = -1; }
    }
  }
}

/* cooperative fill of the LDS table mirror; called once per kernel launch */
FCU_DEV void load_hot_tables()
{
  FCU_FOR_LANES {
    for (int i = lane; i < 256; i += 64) g_hot.bin[i] = k_bin[i];
    for (int i = lane; i < 3 * 16; i += 64) { const int t = i / 16, j = i % 16; g_hot.scan[t][j] = k_scan[k_scan_off[t * 4 + 0] + j]; }
    if (lane < 12) g_hot.scan_cg8[lane / 4][lane % 4] = k_scan_cg[k_scan_cg_off[(lane / 4) * 4 + 1] + lane % 4];
    if (lane < 16) g_hot.ctx_ind_map4x4[lane] = k_ctx_ind_map4x4[lane];
    if (lane < 32) g_hot.group_idx[lane] = k_group_idx[lane];
    if (lane < 3) { uint64_t v = 0; for (int j = 0; j < 16; j++) v |= (uint64_t)k_scan[k_scan_off[lane * 4 + 0] + j] << (4 * j); g_hot.scan4_nib[lane] = v; }
    if (lane == 3) { uint64_t v = 0; for (int j = 0; j < 16; j++) v |= (uint64_t)k_ctx_ind_map4x4[j] << (4 * j); g_hot.map4_nib = v; }
    if (lane >= 4 && lane < 8) {                               /* getSigCtxInc's cnt per pattern and 4x4 raster position (TComTrQuant.cpp:2650-2690) */
      const int pattern = lane - 4; uint32_t v = 0;
      for (int j = 0; j < 16; j++) {
        const int xs = j & 3, ys = j >> 2; int cnt;
        if (pattern == 0) { const int t = xs + ys; cnt = (t >= 3) ? 0 : ((t >= 1) ? 1 : 2); }
        else if (pattern == 1) cnt = (ys >= 2) ? 0 : ((ys >= 1) ? 1 : 2);
        else if (pattern == 2) cnt = (xs >= 2) ? 0 : ((xs >= 1) ? 1 : 2);
        else cnt = 2;
        v |= (uint32_t)cnt << (2 * j);
      }
      g_hot.cnt_bits[pattern] = v;
    }
  }
}

/* ---- one CTU of one chain: the loop body of TEncSlice::compressSlice, TEncSlice.cpp:1380-1551 */
FCU_DEV FCU_NOINLINE void compress_ctu(Chain *C, Scratch *G, int ctuRsAddr)
{
  C = FCU_UNI(C); G = FCU_UNI(G); ctuRsAddr = FCU_UNI(ctuRsAddr);
  Env E; E.C = C; E.G = G;
  FCU_TIC(t10_);
  const Params &P = C->p;
  const int sliceLen = P.slice_ctus > 0 ? P.slice_ctus : C->n_ctu;
  const int sliceStart = (ctuRsAddr / sliceLen) * sliceLen;
  int sliceEnd = sliceStart + sliceLen; if (sliceEnd > C->n_ctu) sliceEnd = C->n_ctu;
  E.cur_ctu = ctuRsAddr; E.slice_start = sliceStart;
  fcu_ctu_out *out = &C->out[ctuRsAddr];
  const int x = (ctuRsAddr % C->w_ctu) * CTU, y = (ctuRsAddr / C->w_ctu) * CTU;
  FCU_SERIAL { g_S.env = E; if (ctuRsAddr == sliceStart) cab_init(slot_ptr(E, 0, CI_CURR_BEST), P.qp, P.slice_type, P.cabac_b_table); else cab_copy1(slot_ptr(E, 0, CI_CURR_BEST), &C->state); }
  FCU_FOR_LANES {                                            /* TComDataCU::initCtu defaults, TComDataCU.cpp:474-560 */
    for (int i = lane; i < NPART; i += 64) {
      out->depth[i] = 0; out->width[i] = CTU; out->height[i] = CTU; out->skip[i] = 0; out->part_size[i] = SIZE_NONE; out->pred_mode[i] = MODE_NONE;
      out->tq_bypass[i] = 0; out->qp[i] = (int8_t)P.qp; out->chroma_qp_adj[i] = 0; out->tr_idx[i] = 0; out->ipcm[i] = 0;
      out->merge_flag[i] = 0; out->merge_idx[i] = 0; out->inter_dir[i] = 0; out->mvp_idx[i] = -1; out->ref_idx[i] = -1;   /* clearMvField: NOT_VALID */
      out->mv[i][0] = out->mv[i][1] = 0; out->mvd[i][0] = out->mvd[i][1] = 0;
      for (int c = 0; c < 3; c++) { out->tskip[c][i] = 0; out->cbf[c][i] = 0; }
      out->intra_dir[0][i] = DC; out->intra_dir[1][i] = 0;
    }
    for (int i = lane; i < 4096; i += 64) out->coeff_y[i] = 0;
    for (int i = lane; i < 1024; i += 64) { out->coeff_cb[i] = 0; out->coeff_cr[i] = 0; }
    cab_copy(&g_S.cab[CAB_GOON], slot_ptr(E, 0, CI_CURR_BEST), lane);
    if (lane == 0) { for (int d = 0; d < 4; d++) { g_S.best_idx[d] = 0; g_S.reco_best_idx[d] = 0; } G->par_ps[0] = SIZE_NONE; out->total_cost = FCU_MAX_DOUBLE; out->total_dist = out->total_bits = out->total_bins = 0; }
  }
  cu_init(&G->cu[0][0], 0, x, y, 0);
  cu_init(&G->cu[0][1], 0, x, y, 0);
  compress_cu<0>();
  /* encodeCtu on [0][CI_CURR_BEST] (TEncSlice.cpp:1474-1487): replay the winner to advance the contexts */
  {
    const CuObj *view = cu_best(E, 0);                     /* what copyToPic has just published (whole CTU) */
    FCU_FOR_LANES {
      cab_copy(&g_S.cab[CAB_GOON], slot_ptr(E, 0, CI_CURR_BEST), lane);
    }
    FCU_TIC(t9_);
    FCU_SERIAL {
      cab_reset_bits((CAB_GOON));
      encode_ctu((CAB_GOON), view, ctuRsAddr == sliceEnd - 1);
      cab_copy1(&C->state, &g_S.cab[CAB_GOON]);
    }
    FCU_TOC(E, t9_, 9);
  }
  FCU_TOC(E, t10_, 10);
}

} // namespace fcu
