/*
 * fcu.h -- C ABI of the MI355X CU-decision engine (libfcu.so).
 *
 * Drop-in boundary for HM's `TEncCu` (reference: Lib/TLibEncoder/TEncCu.h:104-118):
 *
 *   reference entry point                         replacement
 *   -------------------------------------------   ------------------------------------------
 *   TEncCu::create  (TEncCu.cpp:163)              fcu_create
 *   TEncCu::destroy (TEncCu.cpp:214)              fcu_destroy
 *   TEncCu::init + TEncSlice::setUpLambda         fcu_chain_begin   (per slice-chain parameters,
 *     (TEncCu.cpp:306, TEncSlice.cpp:496-524)                        lambda as f64 bit patterns)
 *   slice loop of TEncGOP::compressGOP             fcu_chain_set_range (one chain per slice of a frame)
 *     (TEncGOP.cpp:1102-1138, SliceMode 1)
 *   TEncCu::compressCtu (TEncCu.cpp:329)          fcu_compress_ctu  (one CTU of one chain) /
 *     + TEncCu::encodeCtu context replay          fcu_compress_chains (batched, many chains)
 *     (TEncCu.cpp:359, TEncSlice.cpp:1468-1487)
 *   TEncSlice::getOutlierWithDCT (fork pre-pass)   fcu_obf_prepass
 *     (TEncSlice.cpp:878-1173, TEncGOP.cpp:1096)
 *   fork decision hooks of TEncCu::xCompressCU     fcu_chain_set_decision (frame state + Naive switches + OBF map),
 *     (TEncCu.cpp:504-507,585-603,951-996,           fcu_get_verify_counts (g_iVerResult of the Verifying frame),
 *      1040,1143,1257,1446,1489-1497;                fcu_decision_switch (SetDecisionSwitch), fcu_frame_state
 *      tools_YS.cpp:686-695,968-986,1123-1154,1237)   (getCurrentState)
 *   TEncSearch::estIntraPredLumaQT per-PU results  fcu_chain_set_pu_trace (BASELINE configs[1]: luma RDO artefact)
 *     (TEncSearch.cpp:2178-2655)
 *   TComLoopFilter::loopFilterPic                  fcu_deblock (in-loop deblocking of the decided picture)
 *     (TComLoopFilter.cpp:130, TEncGOP.cpp:1160)
 *   TEncSampleAdaptiveOffset::SAOProcess           fcu_sao (statistics, per-CTU parameter decision, offset pass) +
 *     (TEncSampleAdaptiveOffset.cpp:257,              fcu_sao_enabled / fcu_sao_update_rate (decidePicParams and the
 *      TEncGOP.cpp:1427-1441)                         m_saoDisabledRate bookkeeping across pictures, :363-395,895-917)
 *   m_pppcRDSbacCoder[0][CI_CURR_BEST] state      fcu_get_ctx_state
 *     (TEncSlice.cpp:1417,1477)
 *
 * Plain pointers and sizes only; no torch / HIP types.  All `dev_*` pointers are device
 * (HBM) addresses owned by the caller; planes are 8-bit 4:2:0 (HM's int16 `Pel` planes are
 * narrowed by the adapter).  The library FAILS (returns FCU_ERR_NO_DEVICE)
 * when no HIP device is present: there is no CPU fallback.
 */
#ifndef FCU_H
#define FCU_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define FCU_NPART 256            /* 4x4 partitions per 64x64 CTU, z-order (TComDataCU) */

enum { FCU_OK = 0, FCU_ERR_NO_DEVICE = -1, FCU_ERR_ARG = -2, FCU_ERR_HIP = -3, FCU_ERR_STATE = -4 };

/* Per-CTU result: the TComDataCU arrays that copyToPic publishes (TComDataCU.h:72-164,
 * TComDataCU.cpp:992-1065).  One entry per 4x4 luma partition, z-order.  Coefficients are
 * TU-contiguous at offset absPartIdx*16 (luma) / absPartIdx*4 (chroma) like m_pcTrCoeff. */
typedef struct fcu_ctu_out {
  uint8_t  depth[FCU_NPART], width[FCU_NPART], height[FCU_NPART];   /* m_puhDepth/Width/Height */
  uint8_t  skip[FCU_NPART];                                         /* m_skipFlag              */
  int8_t   part_size[FCU_NPART], pred_mode[FCU_NPART];              /* m_pePartSize/PredMode   */
  uint8_t  tq_bypass[FCU_NPART];                                    /* m_CUTransquantBypass    */
  int8_t   qp[FCU_NPART];                                           /* m_phQP                  */
  uint8_t  chroma_qp_adj[FCU_NPART];                                /* m_ChromaQpAdj           */
  uint8_t  tr_idx[FCU_NPART];                                       /* m_puhTrIdx              */
  uint8_t  tskip[3][FCU_NPART];                                     /* m_puhTransformSkip      */
  uint8_t  cbf[3][FCU_NPART];                                       /* m_puhCbf (bit t = depth t) */
  uint8_t  intra_dir[2][FCU_NPART];                                 /* m_puhIntraDir           */
  uint8_t  ipcm[FCU_NPART];                                         /* m_pbIPCMFlag            */
  /* inter prediction data, reference picture list 0 (P slices; list 1 does not exist in the configurations built) */
  uint8_t  merge_flag[FCU_NPART], merge_idx[FCU_NPART];             /* m_pbMergeFlag, m_puhMergeIndex */
  uint8_t  inter_dir[FCU_NPART];                                    /* m_puhInterDir (1 = list 0) */
  int8_t   mvp_idx[FCU_NPART], ref_idx[FCU_NPART];                  /* m_apiMVPIdx[0], m_acCUMvField[0] refIdx (-1 = none) */
  int16_t  mv[FCU_NPART][2], mvd[FCU_NPART][2];                     /* m_acCUMvField[0] mv / mvd, quarter samples, [hor, ver] */
  int32_t  coeff_y[4096], coeff_cb[1024], coeff_cr[1024];           /* m_pcTrCoeff             */
  double   total_cost;                                              /* m_dTotalCost            */
  uint32_t total_dist, total_bits, total_bins;                      /* m_uiTotal*              */
} fcu_ctu_out;

typedef struct fcu_seq_params {
  int width, height;             /* luma samples, multiples of 8 (SPS)                       */
  int max_chains;                /* chains (frame/slice sequences) decided concurrently      */
  int device;                    /* HIP device ordinal                                       */
} fcu_seq_params;

typedef struct fcu_frame_params {
  int qp;                        /* slice QP                                                 */
  int slice_ctus;                /* SliceMode 1 / SliceArgument (CTUs per slice); 0 = 1 slice */
  int transform_skip, transform_skip_fast, sign_hiding, strong_intra_smoothing;
  /* 0.0 => derive as HM does for an I slice (TEncSlice.cpp:686-706, 496-524); a P slice must give lambda (its GOP
   * entry's QP factor, fcu_ldp_slice), sqrt_lambda / chroma_weight / rdoq_lambda are then derived from it when left 0 */
  double lambda, sqrt_lambda, chroma_weight, rdoq_lambda[3];
  /* P slices (lowdelay_P, BASELINE configs[4]); all 0 in an I slice */
  int slice_type;                /* FCU_SLICE_I / FCU_SLICE_P: a P chain needs fcu_chain_set_reference                   */
  int search_range;              /* SearchRange in integer samples (64)                                                 */
  int fast_enc;                  /* FEN: every other row in the integer-search SAD of blocks taller than 8              */
  int hadamard_me;               /* HadamardME: SATD in the sub-sample refinement and the merge estimation              */
  int fast_merge_decision;       /* FDM                                                                                  */
  int max_merge_cand;            /* MaxNumMergeCand (5)                                                                  */
  int fast_search;               /* FastSearch: 0 = full search (xPatternSearch), 1 = TZ search (xTZSearch, HM's cfg default) */
  int tmvp;                      /* TMVPMode: temporal merge / AMVP candidate from the collocated picture = the reference picture
                                    (collocated_from_l0, collocated_ref_idx 0); needs fcu_chain_set_collocated                */
  int rdoq, rdoq_ts;             /* RDOQ / RDOQTS (fcu_default_frame_params: 1, 1): 0 = TComTrQuant::xQuant's plain quantiser with
                                    signBitHidingHDQ for blocks without / with transform skip */
  int amp;                       /* AMP: asymmetric motion partitions 2NxnU / 2NxnD / nLx2N / nRx2N at CU sizes 64..16, selected as
                                    HM does with AMP_ENC_SPEEDUP + AMP_MRG (TEncCu.cpp:381-450,836-943); part_size 4..7 in fcu_ctu_out */
  int cabac_b_table;             /* P slice: 1 = its contexts start from the B-slice tables (TComSlice::getEncCABACTableIdx() == B_SLICE
                                    with cabac_init_present_flag: TEncSbac::resetEntropy, TEncSbac.cpp:111-115 -- the choice
                                    TEncSbac::determineCabacInitIdx made after the previous slice, TEncSlice.cpp:1750-1753); 0 = the
                                    slice type's own tables */
} fcu_frame_params;
enum { FCU_SLICE_I = 0, FCU_SLICE_P = 1 };
#define FCU_REF_MARGIN_LUMA 80   /* border of a padded reference plane: g_uiMaxCUWidth + 16 (TComPic::create); chroma: 40 */

typedef struct fcu_ctx fcu_ctx;

void fcu_default_frame_params(fcu_frame_params *fp, int qp);
int  fcu_create(const fcu_seq_params *sp, fcu_ctx **out);
void fcu_destroy(fcu_ctx *c);
int  fcu_num_ctus(const fcu_ctx *c);
/* Bind a chain to its planes/output and reset it to CTU 0.  dev_out holds fcu_num_ctus() entries. */
int  fcu_chain_begin(fcu_ctx *c, int chain, const fcu_frame_params *fp,
                     const uint8_t *dev_org_y, const uint8_t *dev_org_u, const uint8_t *dev_org_v,
                     uint8_t *dev_rec_y, uint8_t *dev_rec_u, uint8_t *dev_rec_v,
                     fcu_ctu_out *dev_out);
/* P slice: the reference picture of the chain (list 0, index 0) = the previous picture after the loop filters, as padded
 * planes made by fcu_pad_reference (pointers to the first byte of each padded plane; luma stride = width + 2 * 80).
 * Slice QP and lambda of picture `poc` under HM's lowdelay_P GOP table come from fcu_ldp_slice. */
int  fcu_chain_set_reference(fcu_ctx *c, int chain, const uint8_t *dev_pad_y, const uint8_t *dev_pad_u, const uint8_t *dev_pad_v);
/* Several reference pictures (HM's lowdelay_P cfg lists four): RefPicList0[r] = dev_pad_planes[3r .. 3r+2] (padded Y, U, V of
 * fcu_pad_reference) at POC ref_pocs[r], r < n_ref <= FCU_MAX_REF; cur_poc = the picture being decided.  The search then loops
 * over the reference indices (TEncSearch::predInterSearch, TEncSearch.cpp:3110-3190), codes ref_idx, and scales neighbouring /
 * collocated vectors that point at another picture by the POC distances (TComDataCU::xGetDistScaleFactor, TComDataCU.cpp:3312).
 * The collocated picture of TMVP is RefPicList0[0]; name the POCs ITS list 0 referenced with fcu_chain_set_collocated_pocs
 * (default: one reference at its POC - 1). */
#define FCU_MAX_REF 4
int  fcu_chain_set_references(fcu_ctx *c, int chain, int n_ref, const uint8_t *const *dev_pad_planes, const int *ref_pocs, int cur_poc);
int  fcu_chain_set_collocated_pocs(fcu_ctx *c, int chain, int col_poc, const int *col_ref_pocs, int n);
/* TEncSearch::m_integerMv2Nx2N[REF_PIC_LIST_0][r] (TEncSearch.h:123; read and written at TEncSearch.cpp:3833-3842): the integer
 * vector of the last 2Nx2N TZ search on reference index r, an extra start point of the next search.  In HM it is a member of the
 * encoder that is never reset: it crosses CTUs, slices and pictures.  A chain carries it from CTU to CTU and, when one chain walks
 * several slices, from slice to slice; fcu_chain_begin starts it from zero.  A caller that runs slices or pictures one after the
 * other as HM does (the adapter) reads it with _get after a chain's last CTU and puts it back with _set after the next
 * fcu_chain_begin; slices decided side by side as independent chains cannot know the state the slice before them ends with and
 * start from zero -- which differs from HM only for a slice whose first CTU is too small for a 64x64 CU (otherwise the slice's
 * first search overwrites the state before anything reads it).  xy = { x0, y0, x1, y1, ... } for FCU_MAX_REF indices. */
int  fcu_chain_get_search_state(fcu_ctx *c, int chain, int32_t *xy);
int  fcu_chain_set_search_state(fcu_ctx *c, int chain, const int32_t *xy);
/* TMVP: the motion field of the chain's reference picture = the fcu_ctu_out array that picture was decided into (device
 * pointer, fcu_num_ctus() entries, kept alive by the caller).  What TComPic::compressMotion keeps (the top-left 4x4 partition
 * of every 16x16 block) is read in place; frame_params.tmvp switches the temporal candidates on (TComDataCU.cpp:2528-2563,
 * 2863-2900, xGetColMVP :3175-3242).  NULL: no collocated picture (every temporal candidate unavailable). */
int  fcu_chain_set_collocated(fcu_ctx *c, int chain, const fcu_ctu_out *dev_col_out);
/* Reference picture padding (TComPicYuv::extendPicBorder): copies the width x height planes into planes of
 * (width + 160) x (height + 160) luma / (width/2 + 80) x (height/2 + 80) chroma samples with the border replicated.
 * One kernel on `hip_stream`, asynchronous. */
int  fcu_pad_reference(fcu_ctx *c, const uint8_t *dev_y, const uint8_t *dev_u, const uint8_t *dev_v,
                       uint8_t *dev_pad_y, uint8_t *dev_pad_u, uint8_t *dev_pad_v, void *hip_stream);
/* bytes of the three padded planes of this sequence: out3[0] luma, out3[1] = out3[2] chroma */
void fcu_pad_sizes(const fcu_ctx *c, size_t *out3);
/* Slice type, QP and lambda of picture `poc` under HM's encoder_lowdelay_P_main GOP table (QP offsets 3,2,3,1; QP factors
 * 0.4624 x3, 0.578) as TEncSlice::initEncSlice derives them (TEncSlice.cpp:560-740): fills fp->qp / lambda / slice_type and the
 * configuration defaults (SearchRange 64, FEN, HadamardME, FDM, MaxNumMergeCand 5).  Pure host arithmetic. */
void fcu_ldp_slice(fcu_frame_params *fp, int base_qp, int poc);
/* pic->getSlice(0)->getDepth() of picture `poc` under that table (GOPSize 4; TEncSlice.cpp:236-262): 0, 2, 1, 2 for
 * poc % 4 = 0..3 -- the temporal layer fcu_sao_enabled / fcu_sao_update_rate key on */
int  fcu_ldp_layer(int poc);
/* Restrict a bound chain to the CTUs [first_ctu, first_ctu + n_ctus) of its frame.  Both ends must be slice
 * boundaries (frame_params.slice_ctus), where HM resets the entropy coder (TEncSlice.cpp:1392-1395) and masks the
 * neighbourhood (TComDataCU::getPULeft/Above): the slices of ONE frame then run as independent chains that share
 * the frame's planes and fcu_ctu_out array (disjoint writes) -- the way a single frame is spread over chains / GPUs. */
int  fcu_chain_set_range(fcu_ctx *c, int chain, int first_ctu, int n_ctus);
/* Advance chains [first, first+n) by up to `ctus` CTUs each (raster order; compressCtu +
 * encodeCtu replay per CTU).  Asynchronous on `hip_stream` (hipStream_t or NULL). */
int  fcu_compress_chains(fcu_ctx *c, int first, int n, int ctus, void *hip_stream);
/* HM-shaped call: decide CTU `ctuRsAddr` (must be the chain's next CTU) and copy its
 * TComDataCU arrays to host memory.  Synchronous. */
int  fcu_compress_ctu(fcu_ctx *c, int chain, uint32_t ctuRsAddr, fcu_ctu_out *host_out);
/* context state of m_pppcRDSbacCoder[0][CI_CURR_BEST] after the chain's last CTU:
 * 160 context bytes (engine order, see fcu_engine.h) + the Q15 fractional bit counter */
int  fcu_get_ctx_state(fcu_ctx *c, int chain, uint8_t *ctx160, uint64_t *frac_bits);
/* all 176 context states (160 residual / intra contexts + the inter syntax, engine order) + the Q15 counter */
int  fcu_get_ctx_state_full(fcu_ctx *c, int chain, uint8_t *ctx176, uint64_t *frac_bits);
int  fcu_chain_position(fcu_ctx *c, int chain);          /* next CTU to be decided */
int  fcu_sync(fcu_ctx *c);
/* average duration (ms) of the engine kernel launches recorded with HIP events on the launch
 * stream since the last call; resets the accumulator */
double fcu_kernel_ms(fcu_ctx *c, int *launches);
/* diagnostic counters of a chain: out17[0..15] section timers (shader clocks, -DFCU_PROFILE builds only),
 * out17[16] = TU trials so far */
int  fcu_debug_counters(fcu_ctx *c, int chain, unsigned long long *out17);
/* Fork pre-pass TEncSlice::getOutlierWithDCT (TEncSlice.cpp:878-1173, called per picture at TEncGOP.cpp:1096):
 * outlier-block-flag maps of n_frames luma planes (each width*height bytes, contiguous).  dev_obf receives
 * n_frames * (width/4)*(height/4) int16 counts (what xCompressCU reads at TEncCu.cpp:585-603); host_yc (may be
 * NULL) receives the 16 per-frequency TCM thresholds of every frame; kernel_ms2 (may be NULL) the durations of
 * the histogram and the counting kernel.  Synchronous (the threshold fit runs on the host between the kernels). */
int  fcu_obf_prepass(fcu_ctx *c, int n_frames, const uint8_t *dev_y, int16_t *dev_obf, double *host_yc,
                     float *kernel_ms2, void *hip_stream);
/* The host step of fcu_obf_prepass on its own: TCMprocessOneSequence (TEncSlice.cpp:343-392) on the histogram of
 * |coefficient / 8| of one frequency (hist[a] = samples of amplitude a, n_samples in total); returns the threshold Yc.
 * Pure host arithmetic (doubles + libm, as the reference); needs no GPU. */
double fcu_tcm_threshold(const unsigned *hist, int hist_len, int n_samples);
/* ---- the fork's fast CU-size decision (its default control: Naive model on the N_OBF feature, YSGlobalControl,
 * tools_YS.cpp:4-58).  A chain starts in FCU_TRAINING (exhaustive RDO).  FCU_VERIFYING is exhaustive too and counts,
 * per depth, how the Naive label ("split" when the CU holds an outlier block, "do not split" when it holds none)
 * compares with the RDO outcome.  FCU_TESTING prunes: at a depth whose sw_skip2nx2n is on, a CU labelled "split" skips
 * its 2Nx2N check; at a depth whose sw_terminate is on, a CU labelled "do not split" is not divided further (at depth 3:
 * NxN is not tried).  dev_obf is the frame's map from fcu_obf_prepass ((height/4) x (width/4) int16). */
enum { FCU_TRAINING = 0, FCU_VERIFYING = 1, FCU_TESTING = 2 };      /* CurrentState, getCurrentState tools_YS.cpp:1237 */
typedef struct fcu_decision_params {
  int state;
  int depth_exception;           /* g_bDepthException (tools_YS.cpp:25): no pruning of depth-3 CUs that hold outliers */
  uint8_t sw_skip2nx2n[4];       /* g_bDecisionSwitch[depth][Naive][Skip2Nx2N]   */
  uint8_t sw_terminate[4];       /* g_bDecisionSwitch[depth][Naive][TerminateCU] */
  const int16_t *dev_obf;
} fcu_decision_params;
/* g_iVerResult[depth][Naive][TP, FP, TN, FN, FPLoss, FNLoss] (globals_YS.h:81-89) */
typedef struct fcu_verify_counts { double n[4][6]; } fcu_verify_counts;
/* Set the decision state of a bound chain (any time between launches) and clear its verification counters. */
int  fcu_chain_set_decision(fcu_ctx *c, int chain, const fcu_decision_params *dp);
/* Sum of the verification counters of chains [first, first+n), added up in chain order.  Synchronous. */
int  fcu_get_verify_counts(fcu_ctx *c, int first, int n, fcu_verify_counts *host_sum);
/* SetDecisionSwitch (tools_YS.cpp:1123-1154): a switch turns on when the precision of its label on the Verifying frame
 * exceeds the threshold (0 => the reference's default 0.8).  Pure host arithmetic. */
void fcu_decision_switch(const fcu_verify_counts *v, const double th_skip[4], const double th_term[4],
                         uint8_t sw_skip2nx2n[4], uint8_t sw_terminate[4]);
/* getCurrentState (tools_YS.cpp:1237-1242) for picture `poc` with g_iP = period, g_iT = n_training, g_iV = n_verifying
 * (reference defaults 60 / 2 / 1, tools_YS.cpp:41-43) */
int  fcu_frame_state(int poc, int period, int n_training, int n_verifying);
/* In-loop deblocking of a completely decided all-intra picture, in place on its reconstruction planes:
 * TComLoopFilter::loopFilterPic (TComLoopFilter.cpp:130-155) as TEncGOP::compressGOP runs it after the last slice of the
 * picture (TEncGOP.cpp:1155-1160), with the encoder's default control -- filter enabled, LFCrossSliceBoundaryFlag 1,
 * LFCrossTileBoundaryFlag 1 (TAppEncCfg.cpp:813-818,848-849) -- and the slice's beta / tc offsets (div 2, -6..6).
 * dev_out is the picture's fcu_ctu_out array (depth, part_size, tr_idx and qp are read).  Two kernels on `hip_stream`;
 * asynchronous unless kernel_ms2 is given, which then receives the durations of the vertical- and horizontal-edge pass. */
int  fcu_deblock(fcu_ctx *c, const fcu_ctu_out *dev_out, uint8_t *dev_rec_y, uint8_t *dev_rec_u, uint8_t *dev_rec_v,
                 int beta_offset_div2, int tc_offset_div2, float *kernel_ms2, void *hip_stream);
/* ---- sample adaptive offset --------------------------------------------------------------------------------------
 * SAOOffset / SAOBlkParam of the reference (TypeDef.h:760-800) narrowed to bytes: mode 0 off / 1 new / 2 merge;
 * type: edge class 0..3 or 4 = band offset for a new mode, 0 = left / 1 = above for a merge; band = band position;
 * offset[class]: EO offsets at [0,1,3,4] (class 2 is always 0), BO offsets at [band .. band+3] (mod 32). */
typedef struct { int8_t mode, type, band, pad; int8_t offset[32]; } fcu_sao_offset;
typedef struct { fcu_sao_offset c[3]; } fcu_sao_ctu;                    /* Y, Cb, Cr */
typedef struct {
  int32_t slice_type;        /* FCU_SLICE_I / FCU_SLICE_P: context initialisation of the SAO syntax */
  int32_t qp;                /* slice QP */
  int32_t slice_ctus;        /* SliceArgument (0 = one slice): merge candidates do not cross slices */
  int32_t enabled[3];        /* slice-level switches (fcu_sao_enabled) */
  double  lambda[3];         /* TComSlice::getLambdas(): lambda, lambda / chroma weight (x2); [1], [2] = 0: derived from [0] and the QP (chroma QP offsets 0) */
} fcu_sao_params;
/* TEncSampleAdaptiveOffset::SAOProcess (TEncSampleAdaptiveOffset.cpp:257-287; TEncGOP.cpp:1427-1441, SaoCtuBoundary 0) of
 * n_pics completely decided and deblocked pictures of this context's size, in place on their reconstruction planes.
 * dev_org / dev_rec: host arrays of 3 * n_pics device pointers (Y, U, V of picture 0, then picture 1 ...);
 * dev_coded: device array [n_pics][num_ctus] receiving the parameters as signalled (what the adapter stores into
 * TComPicSym::getSAOBlkParam()); off_count (host, 3 * n_pics, may be NULL): CTUs whose reconstructed mode is off, per
 * component -- the input of fcu_sao_update_rate.  Four kernels on `hip_stream`; the call returns after they have finished
 * (off_count and kernel_ms4 -- statistics, candidates, decision, offset pass -- are read back).  No CPU fallback. */
int  fcu_sao(fcu_ctx *c, int n_pics, const fcu_sao_params *params, const uint8_t *const *dev_org, uint8_t *const *dev_rec,
             fcu_sao_ctu *dev_coded, int32_t *off_count, float *kernel_ms4, void *hip_stream);
/* decidePicParams (:363-395): enabled[comp] = 0 when the picture's temporal layer is > 0 and the share of SAO-off CTUs in
 * layer - 1 exceeded 0.75 (luma) / 0.5 (chroma).  rate = m_saoDisabledRate[3][8], zero-initialised by the caller per sequence. */
void fcu_sao_enabled(const double rate[3][8], int layer, int32_t enabled[3]);
/* the bookkeeping at the end of decideBlkParams (:895-917) */
void fcu_sao_update_rate(double rate[3][8], int layer, const int32_t off_count[3], int num_ctus);

/* ---- per-PU record of the luma search (BASELINE configs[1]: intra-luma RDO, TEncSearch::estIntraPredLumaQT over the 35
 * modes at all depths).  Exhaustive RDO visits every PU of the five layers of a CTU -- 1 + 4 + 16 + 64 PUs of 2Nx2N CUs at
 * depth 0..3 and 256 PUs of NxN CUs at depth 3 = 341 -- and estIntraPredLumaQT (TEncSearch.cpp:2178-2655) leaves per PU: the
 * RMD survivors with their SATD costs (CandCostList, :2289-2336), the candidate list after the MPM additions (:2407-2428),
 * the winning mode, its luma distortion and RD cost after the full-RQT re-run (:2518-2586).  Optional side output of the
 * ordinary decision: bind a device array of fcu_num_ctus() * FCU_PUS_PER_CTU records to a chain and every PU searched
 * from then on is recorded at [ctu][fcu_pu_index]; PUs outside the picture or pruned by the fork's Testing state stay
 * valid = 0.  The decisions themselves do not change. */
#define FCU_PUS_PER_CTU 341
typedef struct fcu_pu_trace {
  uint8_t  valid;            /* 1 once the PU has been searched                                    */
  uint8_t  best_mode;        /* luma intra direction of the PU                                     */
  uint8_t  n_rmd;            /* RMD survivors (g_aucIntraModeNumFast: 8, 8, 3, 3, 3 for 4..64)     */
  uint8_t  n_rd;             /* full-RD candidates after the MPM additions                         */
  uint8_t  rd_mode[12];      /* the candidates in test order                                       */
  uint32_t best_dist;        /* luma SSE of the winner                                             */
  uint32_t pad;
  double   best_cost;        /* its RD cost                                                        */
  double   rmd_cost[8];      /* CandCostList of the survivors: SATD + sqrt(lambda) * mode bits     */
} fcu_pu_trace;
/* layer offsets 0 / 1 / 5 / 21 (2Nx2N CUs at depth 0..3) and 85 (NxN PUs); zidx = z-order index of the PU's first 4x4 partition */
int  fcu_pu_index(int depth, int nxn, int zidx);
int  fcu_chain_set_pu_trace(fcu_ctx *c, int chain, fcu_pu_trace *dev_trace);

/* diagnostic: chains (one-wave workgroups of the engine kernel) the runtime keeps resident per compute unit */
int  fcu_chains_per_cu(void);
/* text of the calling thread's last failure (one buffer per host thread) */
const char *fcu_last_error(void);
/* compiler version and the exact flags libfcu.so was built with.  The engine relies on
 * `-mllvm -amdgpu-remove-redundant-endcf=false`: a build without it is refused by tests/test_cabi.py. */
const char *fcu_build_info(void);
/* sizeof() of the ABI's structures as this library was compiled, by FCU_ABI_* index (-1 for an unknown index): a binding
 * in another language (ctypes, cgo, JNI) checks its own layouts against them before the first call; tests/test_cabi.py does. */
enum { FCU_ABI_CTU_OUT = 0, FCU_ABI_SEQ_PARAMS = 1, FCU_ABI_FRAME_PARAMS = 2, FCU_ABI_DECISION_PARAMS = 3, FCU_ABI_VERIFY_COUNTS = 4,
       FCU_ABI_SAO_CTU = 5, FCU_ABI_SAO_PARAMS = 6, FCU_ABI_PU_TRACE = 7 };
int  fcu_abi_sizeof(int which);

#ifdef __cplusplus
}
#endif
#endif
