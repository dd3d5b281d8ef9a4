/*
 * hmo.h -- ORACLE (test infrastructure, NOT the product path).
 *
 * Plain-C CPU restatement of the HM-16.3 all-intra CU depth/mode RDO decision loop
 * (TEncCu::compressCtu + TEncCu::encodeCtu context replay) as found in
 * Jiraiya812/Fast-CU-Decision-HEVC.  Default state "Training" (exhaustive HM RDO, SURVEY.md section 5 / 7.4
 * item 11); hmo_set_decision selects the fork's Verifying / Testing states with its default Naive model.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this
 * code.  The HIP product path never links or calls it.
 *
 * Parity status: see oracle/README.md ("pinning").
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference).
 */
#ifndef HMO_H
#define HMO_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- fixed coding structure (encoder_intra_main.cfg) ---------------- */
#define HMO_CTU          64     /* MaxCUWidth/Height                                  */
#define HMO_MAXDEPTH     3      /* CU depths 0..3 (64,32,16,8); g_uiMaxCUDepth-g_uiAddCUDepth */
#define HMO_NPART        256    /* 4x4 partitions per CTU (z-order)                   */
#define HMO_LOG2_MAXTU   5      /* QuadtreeTULog2MaxSize                              */
#define HMO_LOG2_MINTU   2      /* QuadtreeTULog2MinSize                              */
#define HMO_TU_MAXDEPTH_INTRA 3 /* QuadtreeTUMaxDepthIntra                            */

/* enums as TypeDef.h:470-489 */
#define HMO_SIZE_2Nx2N   0
#define HMO_SIZE_2NxN    1
#define HMO_SIZE_Nx2N    2
#define HMO_SIZE_NxN     3
#define HMO_SIZE_2NxnU   4      /* asymmetric motion partitions (AMP): 1/4 + 3/4 */
#define HMO_SIZE_2NxnD   5
#define HMO_SIZE_nLx2N   6
#define HMO_SIZE_nRx2N   7
#define HMO_SIZE_NONE    8      /* NUMBER_OF_PART_SIZES */
#define HMO_MODE_INTER   0
#define HMO_MODE_INTRA   1
#define HMO_MODE_NONE    2      /* NUMBER_OF_PREDICTION_MODES */
#define HMO_PLANAR 0
#define HMO_DC     1
#define HMO_HOR    10
#define HMO_VER    26
#define HMO_DM_CHROMA 36

/* CABAC context layout (own numbering; counts from ContextTables.h:49-161) */
enum {
  HMO_CTX_SPLIT      = 0,    /* 3  */
  HMO_CTX_PARTSIZE   = 3,    /* 1  (ctx 0 of 4; intra only uses the first) */
  HMO_CTX_INTRA_LUMA = 4,    /* 1  prev_intra_luma_pred_flag */
  HMO_CTX_CHROMA_PRED= 5,    /* 1  */
  HMO_CTX_CBF_LUMA   = 6,    /* 5  */
  HMO_CTX_CBF_CHROMA = 11,   /* 5  */
  HMO_CTX_SUBDIV     = 16,   /* 3  */
  HMO_CTX_SIGCG      = 19,   /* 2 luma + 2 chroma */
  HMO_CTX_SIG        = 23,   /* 28 luma + 16 chroma */
  HMO_CTX_LASTX      = 67,   /* 15 luma + 15 chroma */
  HMO_CTX_LASTY      = 97,   /* 15 luma + 15 chroma */
  HMO_CTX_ONE        = 127,  /* 16 luma + 8 chroma */
  HMO_CTX_ABS        = 151,  /* 4 luma + 2 chroma */
  HMO_CTX_TSKIP      = 157,  /* 1 luma + 1 chroma */
  HMO_NCTX_INTRA     = 160,  /* the contexts an I slice touches (+1 pad) */
  /* inter syntax (P slices), appended so that the all-intra layout above stays what it was */
  HMO_CTX_SKIP       = 160,  /* 3  cu_skip_flag                        */
  HMO_CTX_MERGE_FLAG = 163,  /* 1                                      */
  HMO_CTX_MERGE_IDX  = 164,  /* 1                                      */
  HMO_CTX_PRED_MODE  = 165,  /* 1                                      */
  HMO_CTX_PARTSIZE1  = 166,  /* 3  part_mode contexts 1..3 (0 = HMO_CTX_PARTSIZE) */
  HMO_CTX_MVD        = 169,  /* 2  abs_mvd_greater0 / greater1         */
  HMO_CTX_REF        = 171,  /* 2  ref_idx                             */
  HMO_CTX_MVP_IDX    = 173,  /* 1                                      */
  HMO_CTX_ROOT_CBF   = 174,  /* 1  rqt_root_cbf                        */
  HMO_NCTX           = 176   /* + 1 pad */
};
#define HMO_SLICE_I 0
#define HMO_SLICE_P 1

/* coder state that HM copies with TEncSbac::load/store (TEncSbac.cpp:397-426,
 * TEncBinCoderCABAC.cpp:148-159): all context states + the fractional bit counter. */
typedef struct {
  uint8_t  ctx[HMO_NCTX];
  uint64_t frac;             /* m_fracBits (Q15) */
} HmoCabac;

/* slots per depth, TypeDef.h:554-563 */
enum { CI_CURR_BEST = 0, CI_NEXT_BEST, CI_TEMP_BEST, CI_CHROMA_INTRA, CI_QT_TRAFO_TEST, CI_QT_TRAFO_ROOT, CI_NUM };

/* Encoder parameters for one chain (= one slice sequence of one frame). */
typedef struct {
  int width, height;            /* luma samples */
  int qp;                       /* slice QP */
  int slice_ctus;               /* SliceMode 1 / SliceArgument; 0 = whole frame is one slice */
  int transform_skip;           /* TransformSkip (1) */
  int transform_skip_fast;      /* TransformSkipFast (1) */
  int sign_hiding;              /* SignHideFlag (1) */
  int strong_smoothing;         /* StrongIntraSmoothing (1) */
  /* derived by hmo_params_finish(): */
  double lambda;                /* TComRdCost::m_dLambda             TEncSlice.cpp:686-706 */
  double sqrt_lambda;           /* TComRdCost::m_sqrtLambda          TComRdCost.cpp:197    */
  double chroma_weight;         /* m_distortionWeight[Cb/Cr]         TEncSlice.cpp:510     */
  double rdoq_lambda[3];        /* TComTrQuant::m_lambdas            TEncSlice.cpp:512     */
  int    qp_c;                  /* chroma QP (g_aucChromaScale)                             */
  /* ---- inter (P slices, BASELINE configs[4]); all 0 for an I slice ---- */
  int slice_type;               /* HMO_SLICE_I / HMO_SLICE_P */
  int search_range;             /* SearchRange (integer samples) */
  int fast_search;              /* FastSearch: 0 = full search (xPatternSearch), 1 = TZ search */
  int fast_enc;                 /* FEN (getUseFastEnc): sub-sampled SAD in the integer search */
  int had_me;                   /* HadamardME: SATD in the fractional search / merge estimation */
  int fdm;                      /* FDM (getUseFastDecisionForMerge) */
  int max_merge_cand;           /* MaxNumMergeCand (5) */
  int amp;                      /* AMP: asymmetric inter partitions at depths 0..2 with HM's AMP_ENC_SPEEDUP / AMP_MRG test selection */
  int tmvp;                     /* TMVPMode: temporal merge / AMVP candidate from the collocated (= reference) picture; needs hmo_set_col */
  int rdoq, rdoq_ts;            /* RDOQ / RDOQTS (1, 1): 0 = the plain quantiser of xQuant with signBitHidingHDQ */
  double lambda_override;       /* > 0: slice lambda given by the caller (P-slice QP factor, TEncSlice.cpp:686-706) */
  unsigned lambda_motion_sad, lambda_motion_sse;   /* m_uiLambdaMotionSAD / SSE, TComRdCost.cpp:194-219 */
  int cabac_b_table;            /* P slice initialised from the B-slice context tables: TComSlice::getEncCABACTableIdx() == B_SLICE with
                                   cabac_init_present_flag (TEncSbac::resetEntropy, TEncSbac.cpp:111-115) */
  int search_state_per_slice;   /* 0 (HM): m_integerMv2Nx2N, the TZ search's extra start point (TEncSearch.h:123, TEncSearch.cpp:3833-3842), is
                                   encoder state that is never reset -- it crosses slices and pictures (hmo_set_int_mv carries it from
                                   picture to picture).  1: it starts from zero with every slice, which is what a slice decided as an
                                   independent chain next to the slices before it can know; the two differ only when a slice begins with a
                                   CTU too small for a 64x64 CU (otherwise the first search of the slice overwrites the state unread) */
} HmoParams;

/* Per-CTU decisions, TComDataCU layout (TComDataCU.h:72-164).  One entry
 * per 4x4 luma partition in z-order. */
typedef struct {
  uint8_t  depth[HMO_NPART], width[HMO_NPART], height[HMO_NPART];
  uint8_t  skip[HMO_NPART];
  int8_t   part_size[HMO_NPART], pred_mode[HMO_NPART];
  uint8_t  tq_bypass[HMO_NPART];
  int8_t   qp[HMO_NPART];
  uint8_t  chroma_qp_adj[HMO_NPART];
  uint8_t  tr_idx[HMO_NPART];
  uint8_t  tskip[3][HMO_NPART];
  uint8_t  cbf[3][HMO_NPART];
  uint8_t  intra_dir[2][HMO_NPART];
  uint8_t  ipcm[HMO_NPART];
  /* inter (list 0 only: P slices) -- m_pbMergeFlag, m_puhMergeIndex, m_puhInterDir, m_apiMVPIdx[0], m_acCUMvField[0] */
  uint8_t  merge_flag[HMO_NPART], merge_idx[HMO_NPART], inter_dir[HMO_NPART];
  int8_t   mvp_idx[HMO_NPART], ref_idx[HMO_NPART];
  int16_t  mv[HMO_NPART][2], mvd[HMO_NPART][2];        /* quarter-sample units, [hor, ver] */
  int32_t  coeff_y[HMO_CTU * HMO_CTU];          /* TU-contiguous, offset = absPartIdx*16 */
  int32_t  coeff_cb[HMO_CTU * HMO_CTU / 4];
  int32_t  coeff_cr[HMO_CTU * HMO_CTU / 4];
  double   total_cost;
  uint32_t total_dist, total_bits, total_bins;
} HmoCtu;

typedef struct HmoEnc HmoEnc;

/* ---- public oracle API ------------------------------------------------------------ */
void    hmo_params_default(HmoParams *p, int width, int height, int qp);
void    hmo_params_finish(HmoParams *p);
HmoEnc *hmo_create(const HmoParams *p);
void    hmo_destroy(HmoEnc *e);
/* planes: 8-bit 4:2:0.  rec planes are written (picture-sized, stride = plane width). */
void    hmo_set_planes(HmoEnc *e, const uint8_t *orgY, const uint8_t *orgU, const uint8_t *orgV,
                       uint8_t *recY, uint8_t *recU, uint8_t *recV);
/* P slice: the reference picture (list 0, index 0), i.e. the previous picture after the loop filters */
void    hmo_set_ref_planes(HmoEnc *e, const uint8_t *refY, const uint8_t *refU, const uint8_t *refV);
/* several reference pictures (RefPicList0[idx], idx < HMO_MAX_REF): planes + POC of each, and the current picture's POC.
 * hmo_set_ref_planes alone = one reference at POC distance 1. */
#define HMO_MAX_REF 4
void    hmo_set_ref_picture(HmoEnc *e, int idx, const uint8_t *refY, const uint8_t *refU, const uint8_t *refV, int poc);
void    hmo_set_poc(HmoEnc *e, int poc, int n_ref);
/* TMVP with several references: POC of the collocated picture and the POCs its own reference list named (by refIdx) */
void    hmo_set_col_pocs(HmoEnc *e, int col_poc, const int *col_ref_poc, int n);
/* compressCtu + encodeCtu replay for CTU ctuRsAddr (must be called in raster order). */
void    hmo_compress_ctu(HmoEnc *e, int ctuRsAddr);
const HmoCtu *hmo_get_ctu(const HmoEnc *e, int ctuRsAddr);
/* CABAC state of m_pppcRDSbacCoder[0][CI_CURR_BEST] after the last encodeCtu. */
const HmoCabac *hmo_get_cabac(const HmoEnc *e);
/* whole frame */
void    hmo_compress_frame(HmoEnc *e);
int     hmo_num_ctus(const HmoEnc *e);
uint32_t hmo_ctu_replay_bits(const HmoEnc *e, int ctuRsAddr);

/* in-loop deblocking of the decided picture (TComLoopFilter::loopFilterPic, TComLoopFilter.cpp:130; TEncGOP.cpp:1160),
 * in place on the reconstruction planes */
void    hmo_deblock(HmoEnc *e, int betaOffsetDiv2, int tcOffsetDiv2);
void    hmo_deblock_pic(const HmoCtu *pic, int width, int height, uint8_t *recY, uint8_t *recU, uint8_t *recV,
                        int betaOffsetDiv2, int tcOffsetDiv2);

/* per-PU record of the luma search (BASELINE configs[1]; estIntraPredLumaQT, TEncSearch.cpp:2178-2655): same layout and
 * indexing as include/fcu.h:fcu_pu_trace / fcu_pu_index.  buf = [num_ctus][341] records, or NULL to stop recording. */
typedef struct {
  uint8_t  valid, best_mode, n_rmd, n_rd;
  uint8_t  rd_mode[12];
  uint32_t best_dist, pad;
  double   best_cost;
  double   rmd_cost[8];
} HmoPuTrace;
void    hmo_set_col(HmoEnc *e, const HmoCtu *col);          /* decided CTUs of the reference picture (its motion field), kept alive by the caller */
void    hmo_set_pu_trace(HmoEnc *e, HmoPuTrace *buf);
void    hmo_set_int_mv(HmoEnc *e, const int *xy);           /* ... as the encoder's previous searches left it (x, y per reference index) */
void    hmo_test_int_mv(const HmoEnc *e, int *xy);          /* m_integerMv2Nx2N as the search holds it now (TZ search state) */

/* sample adaptive offset (TEncSampleAdaptiveOffset::SAOProcess, TEncSampleAdaptiveOffset.cpp:257; TEncGOP.cpp:1434), hmo_sao.c */
typedef struct { int mode, type, aux; int offset[32]; } HmoSaoOffset;      /* mode 0 off / 1 new / 2 merge; type: EO 0..3, BO 4 (merge: 0 left, 1 above); aux: band position */
typedef struct { HmoSaoOffset c[3]; } HmoSaoBlk;
typedef struct { int64_t diff[5][32], count[5][32]; } HmoSaoStat;           /* [type][class]: EO classes at [edgeType + 2], BO bands at [band] */
void    hmo_sao_stats(int width, int height, const uint8_t *const org[3], const uint8_t *const src[3], HmoSaoStat *stats);
void    hmo_sao_picture(int width, int height, int slice_ctus, int qp, int slice_type, const double lambda[3], const int enabled[3],
                        const uint8_t *const org[3], uint8_t *const rec[3], HmoSaoBlk *coded, HmoSaoStat *stats_out, int off_count[3]);

/* fork states (CurrentState, globals_YS.h; getCurrentState, tools_YS.cpp:1237-1242) */
#define HMO_TRAINING  0
#define HMO_VERIFYING 1
#define HMO_TESTING   2
void    hmo_set_decision(HmoEnc *e, int state, const uint8_t *sw_skip, const uint8_t *sw_term, int depth_exception, const int16_t *obf);
void    hmo_get_verify(const HmoEnc *e, double *out24);
void    hmo_decision_switch(const double *ver24, const double *th_skip, const double *th_term, uint8_t *sw_skip, uint8_t *sw_term);

/* ---- leaf functions exported for known-answer tests -------------------------------- */
void    hmo_fwd_transform(const int16_t *resi, int stride, int32_t *coef, int log2, int useDst);
void    hmo_inv_transform(const int32_t *coef, int16_t *resi, int stride, int log2, int useDst);
uint32_t hmo_satd(const uint8_t *org, int so, const uint8_t *pred, int sp, int w, int h);
uint32_t hmo_sse(const uint8_t *org, int so, const uint8_t *rec, int sr, int w, int h);
/* intra prediction from a linear reference array ref[0..4N] (bottom-left .. corner .. top-right) */
void    hmo_intra_pred(const uint8_t *ref, const uint8_t *refFilt, int log2, int mode, int isLuma,
                       uint8_t *dst, int dstStride);
void    hmo_filter_ref(const uint8_t *ref, uint8_t *out, int n, int strong);
int     hmo_use_filtered_ref(int mode, int log2, int isLuma);
void    hmo_cabac_init(HmoCabac *c, int qp);                 /* I-slice tables */
void    hmo_cabac_init_st(HmoCabac *c, int qp, int slice_type);
void    hmo_cabac_init_tab(HmoCabac *c, int qp, int slice_type, int b_table);
const int16_t *hmo_dct_matrix(int log2);   /* N*N, row-major */
const uint16_t *hmo_scan(int scanType, int log2);
const uint8_t *hmo_zscan_to_raster(void);

/* fork pre-pass (TEncSlice::getOutlierWithDCT, TEncSlice.cpp:878-1173): outlier-block-flag map of a luma plane */
int     hmo_obf_prepass(const uint8_t *y, int w, int h, int stride, int16_t *obf, double *yc16);
double  hmo_tcm_threshold(const int *hist, int peak, int len, int *err);
#ifdef __cplusplus
}
#endif
#endif
