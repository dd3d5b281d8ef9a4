/* hmo_int.h -- ORACLE internals (test infrastructure). */
#ifndef HMO_INT_H
#define HMO_INT_H
#include "hmo.h"

#define HMO_MAX_DOUBLE 1.7e+308   /* MAX_DOUBLE, CommonDef.h */

/* tables (hmo_tables.c) */
extern uint8_t  hmo_z2r[HMO_NPART], hmo_r2z[HMO_NPART];
extern int16_t  hmo_T[4][32 * 32];
extern const int16_t hmo_dst4[16];
extern uint16_t hmo_scan_tab[3][4][1024];
extern uint8_t  hmo_scan_cg[3][4][64];
extern const int hmo_quant_scales[6], hmo_inv_quant_scales[6];
extern const uint8_t hmo_chroma_scale[58];
extern const uint8_t hmo_group_idx[32], hmo_min_in_group[10], hmo_ctx_ind_map4x4[16];
extern const int hmo_ang_table[9], hmo_inv_ang_table[9];
extern const uint8_t hmo_intra_filter_thr[5], hmo_rd_mode_num[5];
extern const uint8_t hmo_next_mps[128], hmo_next_lps[128];
extern const int32_t hmo_entropy_bits[128];
extern const uint8_t hmo_ctx_init_I[HMO_NCTX], hmo_ctx_init_P[HMO_NCTX], hmo_ctx_init_B[HMO_NCTX];
void hmo_init_tables(void);

/* Working copy of one CU's decisions (the per-depth TComDataCU best/temp objects,
 * TEncCu.cpp:163-198).  Arrays are indexed by partition index relative to the CU. */
typedef struct {
  int      depth_cu;                 /* CU depth */
  int      x, y;                     /* luma position in the picture */
  int      zidx;                     /* z-order index of the CU inside its CTU (m_absZIdxInCtu) */
  int      nparts;
  double   cost; uint32_t dist, bits, bins;
  uint8_t  depth[HMO_NPART];
  int8_t   part_size[HMO_NPART], pred_mode[HMO_NPART];
  uint8_t  tr_idx[HMO_NPART];
  uint8_t  tskip[3][HMO_NPART], cbf[3][HMO_NPART];
  uint8_t  intra_dir[2][HMO_NPART];
  uint8_t  skip[HMO_NPART], merge_flag[HMO_NPART], merge_idx[HMO_NPART], inter_dir[HMO_NPART];
  int8_t   mvp_idx[HMO_NPART], ref_idx[HMO_NPART];
  int16_t  mv[HMO_NPART][2], mvd[HMO_NPART][2];
  int32_t  coef[3][HMO_CTU * HMO_CTU];   /* chroma uses the first quarter */
} HmoCU;

/* CU-sized sample buffer (TComYuv), strides fixed to the CTU size */
typedef struct { uint8_t y[64 * 64], u[32 * 32], v[32 * 32]; } HmoYuv;
typedef struct { int16_t y[64 * 64], u[32 * 32], v[32 * 32]; } HmoYuv16;

/* transform-unit descriptor (state of a TComTU / TComTURecurse, TComTU.cpp:47-207) */
typedef struct {
  int log2;          /* luma log2 size (GetLog2LumaTrSize) */
  int tr_depth;      /* GetTransformDepthRel() */
  int part;          /* GetRelPartIdxTU() : first partition, relative to the CU */
  int nparts;        /* GetAbsPartIdxNumParts() */
  int x, y;          /* luma rect origin inside the CU */
  int off_y;         /* luma coefficient offset (relative to CU) */
  int cw;            /* chroma rect width, 0 = chroma not processed in this section */
  int cwo;           /* chroma rect width ignoring the processed-section masking (mOrigWidth) */
  int cx, cy;        /* chroma rect origin inside the CU */
  int c_tr_depth;    /* GetTransformDepthRelAdj(chroma) */
  int c_code_all;    /* ProcessingAllQuadrants(chroma) */
  int off_c;         /* chroma coefficient offset (relative to CU) */
  int section;
} HmoTU;

struct HmoEnc {
  HmoParams p;
  int w_ctu, h_ctu, n_ctu;
  const uint8_t *org[3];
  uint8_t *rec[3];
  const uint8_t *ref[3];             /* P slice: reference picture (list 0, index 0) = previous picture after the loop filters */
  /* reference picture list 0 (P slices): refs[r] = planes of RefPicList0[r] (refs[0] == ref), ref_poc[r] its POC, poc the
   * current picture's; n_ref = num_ref_idx_l0_active (1..4).  With one reference and consecutive pictures no vector is ever
   * scaled; with several, AMVP / TMVP scale by POC distance (xGetDistScaleFactor, TComDataCU.cpp:3312). */
  const uint8_t *refs[HMO_MAX_REF][3];
  int n_ref, poc, ref_poc[HMO_MAX_REF];
  int col_poc, col_ref_poc[HMO_MAX_REF];      /* collocated picture (= RefPicList0[0]): its POC and the POCs its own list 0 named */
  int stride[3];
  HmoCtu *pic;                       /* per-CTU committed decisions (TComPic CTU objects) */
  uint32_t *replay_bits;
  /* entropy coders: go-on + [depth][slot]  (TEncTop RD coders) */
  HmoCabac goon; uint32_t goon_bins;
  HmoCabac slot[HMO_MAXDEPTH + 2][CI_NUM];
  /* per-depth CU objects and sample buffers (TEncCu.cpp:163-198) */
  HmoCU  *best[4], *temp[4];
  HmoYuv *org_yuv[4];
  HmoYuv *pred_temp[4];              /* m_ppcPredYuvTemp : prediction, overwritten by recon per TU */
  HmoYuv *reco_best[4], *reco_temp[4];
  /* TEncSearch scratch */
  HmoYuv  qt_rec[4];                 /* m_pcQTTempTComYuv[layer], layer = 5 - log2 */
  int32_t qt_coef[3][4][HMO_CTU * HMO_CTU];   /* m_ppcQTTempCoeff[comp][layer] */
  int32_t ts_coef[3][32 * 32];       /* m_pcQTTempTUCoeff */
  HmoYuv  ts_rec;                    /* m_pcQTTempTransformSkipTComYuv */
  uint8_t shared_pred[3][32 * 32];   /* m_pSharedPredTransformSkip */
  uint8_t tmp_tr_idx[HMO_NPART], tmp_cbf[3][HMO_NPART], tmp_tskip[3][HMO_NPART];
  /* inter: residual of the CU, best residual, residual per RQT layer (m_pcQTTempTComYuv holds residuals on this path), m_tmpYuvPred */
  HmoYuv16 resi_cu, resi_best, qt_resi[4];
  HmoYuv  tmp_pred;
  uint64_t n_sad;                    /* integer-search positions evaluated */
  /* current slice */
  int slice_start;                   /* first CTU (raster) of the slice containing the current CTU */
  int cur_ctu;
  /* RDOQ scratch (locals of xRateDistOptQuant, TComTrQuant.cpp:2082-2095) */
  double  rq_cost_coeff[1024], rq_cost_sig[1024], rq_cost_coeff0[1024];
  int     rq_rate_up[1024], rq_rate_down[1024], rq_sig_delta[1024];
  int32_t rq_delta_u[1024];
  /* fork decision state (tools_YS.cpp) */
  int dec_state, depth_exception, obf_stride;
  uint8_t sw_skip[4], sw_term[4];
  const int16_t *obf;
  double ver[4][6];                  /* g_iVerResult[depth][Naive][ResultType] */
  /* statistics for tests */
  uint64_t n_tu_trials, n_rmd;
  uint32_t last_luma_dist;
  /* test hook: called around every CU candidate so that a test can show the same state to the reference's own
   * search code (oracle/ref/make_golden_search.py) or compare with what it returned (tests/test_golden_search.py) */
  void (*trace)(void *user, int event, int depth, int arg);
  void *trace_user;
  struct { int x, y; } int_mv_2nx2n[HMO_MAX_REF];   /* m_integerMv2Nx2N[list 0][ref]: integer vector of the last 2Nx2N motion search on that reference (TZ search start point) */
  const HmoCtu *col;            /* motion field of the collocated picture (TMVP) */
  HmoPuTrace *pu_trace;         /* optional per-PU record of the luma search (hmo_set_pu_trace) */
};
/* trace events: candidate about to be searched / searched (its results sit in temp[depth], reco_temp[depth], slot[depth][CI_TEMP_BEST]) */
enum { HMO_EV_INTRA_BEGIN = 0, HMO_EV_INTRA_END = 1, HMO_EV_INTER_BEGIN = 2, HMO_EV_INTER_END = 3, HMO_EV_MERGE_BEGIN = 4, HMO_EV_MERGE_END = 5,
       /* a CU that lies inside the picture: before its first candidate (arg = eParentPartSize) / after its last one, before the
        * split flag is priced: best[depth], reco_best[depth] and slot[depth][CI_NEXT_BEST] hold the surviving candidate */
       HMO_EV_CU_BEGIN = 6, HMO_EV_CU_DONE = 7 };

/* ---- hmo_cabac.c */
static inline void hmo_cabac_copy(HmoCabac *d, const HmoCabac *s) { *d = *s; }
void hmo_enc_bin(HmoEnc *e, int bin, int ctx);
void hmo_enc_bins_ep(HmoEnc *e, int nbins);
void hmo_enc_bin_trm(HmoEnc *e, int bin);
static inline void hmo_reset_bits(HmoEnc *e) { e->goon.frac &= 32767; e->goon_bins = 0; }
static inline uint32_t hmo_bits(const HmoEnc *e) { return (uint32_t)(e->goon.frac >> 15); }
static inline int hmo_ctx_bits(const HmoCabac *c, int ctx, int bin) { return hmo_entropy_bits[c->ctx[ctx] ^ bin]; }
void hmo_code_coeff_nxn(HmoEnc *e, const HmoCU *cu, const int32_t *coef, int log2, int comp, int part);
int  hmo_coef_scan_idx(const HmoCU *cu, int part, int log2, int comp);
int  hmo_pattern_sig_ctx(const uint8_t *cgflag, int cgx, int cgy, int wg);
int  hmo_sig_cg_ctx(const uint8_t *cgflag, int cgx, int cgy, int wg);
int  hmo_first_sig_ctx(int log2, int scan, int ch);
int  hmo_sig_ctx_inc(int pattern, int first, int pos, int log2, int ch);

/* ---- hmo_pred.c */
void hmo_build_ref(HmoEnc *e, int comp, int px, int py, int log2, int cur_zidx_unused, uint8_t *ref);

/* ---- hmo_trquant.c */
int  hmo_rdoq(HmoEnc *e, const HmoCU *cu, const HmoTU *tu, int comp, const int32_t *src, int32_t *dst, int log2, int part, int tskip);
void hmo_dequant(const int32_t *q, int32_t *c, int n, int log2, int qp);

#endif
