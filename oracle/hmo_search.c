/*
 * hmo_search.c -- ORACLE (test infrastructure).  The CU / PU / TU decision loops:
 * TEncCu::xCompressCU, xCheckRDCostIntra, TEncSearch::estIntraPredLumaQT / ChromaQT,
 * xRecurIntraCodingLumaQT, xRecurIntraChromaCodingQT, and the encodeCtu context replay.
 */
#include "hmo_int.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdio.h>

/* ------------------------------------------------------------------------------------
 * geometry helpers
 * ---------------------------------------------------------------------------------- */
static inline int part_x(int z) { return (hmo_z2r[z] & 15) << 2; }      /* g_auiRasterToPelX[g_auiZscanToRaster[z]] */
static inline int part_y(int z) { return (hmo_z2r[z] >> 4) << 2; }
static inline int cu_size(const HmoCU *cu) { return HMO_CTU >> cu->depth_cu; }
static inline int ctu_of(const HmoEnc *e, int lx, int ly) { return (ly >> 6) * e->w_ctu + (lx >> 6); }
static inline int zidx_of(int lx, int ly) { return hmo_r2z[((ly & 63) >> 2) * 16 + ((lx & 63) >> 2)]; }
static inline int inside_cu(const HmoCU *cu, int lx, int ly)
{ int s = cu_size(cu); return lx >= cu->x && lx < cu->x + s && ly >= cu->y && ly < cu->y + s; }

/* double floor(D + R*lambda + 0.5): TComRdCost::calcRdCost, TComRdCost.cpp:56-123 */
static double calc_rd_cost(const HmoEnc *e, uint32_t bits, uint32_t dist)
{ return floor((double)dist + ((double)bits * e->p.lambda) + 0.5); }

/* ------------------------------------------------------------------------------------
 * TU descriptors (TComTU.cpp:47-207)
 * ---------------------------------------------------------------------------------- */
static void tu_root(HmoTU *t, const HmoCU *cu)
{
  int s = cu_size(cu);
  memset(t, 0, sizeof(*t));
  t->log2 = 6 - cu->depth_cu; t->tr_depth = 0; t->part = 0; t->nparts = cu->nparts;
  t->cw = t->cwo = s >> 1; t->c_tr_depth = 0; t->c_code_all = 1;
}
/* TComTURecurse(parent, bProcessLastOfLevel, QUAD_SPLIT) advanced to section `i` */
static void tu_child(HmoTU *c, const HmoTU *p, int i, int processLast)
{
  int s = 1 << (p->log2 - 1);
  c->log2 = p->log2 - 1; c->tr_depth = p->tr_depth + 1;
  c->nparts = p->nparts >> 2; if (c->nparts < 1) c->nparts = 1;
  c->part = p->part + i * c->nparts;
  c->x = p->x + (i & 1) * s; c->y = p->y + (i >> 1) * s;
  c->off_y = p->off_y + i * s * s;
  c->section = i;
  int pw = p->cwo;                                     /* parent's full chroma rect width */
  if ((pw >> 1) >= 4) {
    int cs = pw >> 1;
    c->cw = c->cwo = cs; c->c_code_all = 1; c->c_tr_depth = p->c_tr_depth + 1;
    c->cx = p->cx + (i & 1) * cs; c->cy = p->cy + (i >> 1) * cs;
    c->off_c = p->off_c + i * cs * cs;
  } else {                                             /* 2x2 would be too small: stay at parent's 4x4 */
    c->cwo = pw; c->c_code_all = 0; c->c_tr_depth = p->c_tr_depth;
    c->cx = p->cx; c->cy = p->cy; c->off_c = p->off_c;
    c->cw = (processLast ? (i == 3) : (i == 0)) ? pw : 0;
  }
}
static inline int tu_part_c(const HmoTU *t) { return t->c_code_all ? t->part : (t->part & ~3); }        /* GetAbsPartIdxTU(chroma) */
static inline int tu_nparts_c(const HmoTU *t) { return t->c_code_all ? t->nparts : t->nparts * 4; }      /* GetAbsPartIdxNumParts(chroma) */

/* ------------------------------------------------------------------------------------
 * CU object helpers (TComDataCU.cpp:592-1065)
 * ---------------------------------------------------------------------------------- */
/* initEstData / initSubCU */
static void cu_init(HmoCU *cu, int depth, int x, int y, int zidx)
{
  cu->depth_cu = depth; cu->x = x; cu->y = y; cu->zidx = zidx; cu->nparts = HMO_NPART >> (2 * depth);
  cu->cost = HMO_MAX_DOUBLE; cu->dist = 0; cu->bits = 0; cu->bins = 0;
  int n = cu->nparts;
  memset(cu->depth, depth, (size_t)n);
  memset(cu->part_size, HMO_SIZE_NONE, (size_t)n);
  memset(cu->pred_mode, HMO_MODE_NONE, (size_t)n);
  memset(cu->tr_idx, 0, (size_t)n);
  for (int c = 0; c < 3; c++) { memset(cu->tskip[c], 0, (size_t)n); memset(cu->cbf[c], 0, (size_t)n); }
  memset(cu->intra_dir[0], HMO_DC, (size_t)n);
  memset(cu->intra_dir[1], 0, (size_t)n);
  memset(cu->skip, 0, (size_t)n); memset(cu->merge_flag, 0, (size_t)n); memset(cu->merge_idx, 0, (size_t)n); memset(cu->inter_dir, 0, (size_t)n);
  memset(cu->mvp_idx, -1, (size_t)n); memset(cu->ref_idx, -1, (size_t)n);           /* clearMvField: NOT_VALID */
  memset(cu->mv, 0, sizeof(cu->mv[0]) * (size_t)n); memset(cu->mvd, 0, sizeof(cu->mvd[0]) * (size_t)n);
  int s = HMO_CTU >> depth;
  memset(cu->coef[0], 0, sizeof(int32_t) * (size_t)(s * s));
  memset(cu->coef[1], 0, sizeof(int32_t) * (size_t)(s * s / 4));
  memset(cu->coef[2], 0, sizeof(int32_t) * (size_t)(s * s / 4));
}
/* copyPartFrom, TComDataCU.cpp:906-990 */
static void cu_copy_part_from(HmoCU *dst, const HmoCU *src, int partUnitIdx)
{
  int n = src->nparts, off = partUnitIdx * n;
  dst->dist += src->dist; dst->bits += src->bits; dst->bins += src->bins;
  memcpy(dst->depth + off, src->depth, (size_t)n);
  memcpy(dst->part_size + off, src->part_size, (size_t)n);
  memcpy(dst->pred_mode + off, src->pred_mode, (size_t)n);
  memcpy(dst->tr_idx + off, src->tr_idx, (size_t)n);
  for (int c = 0; c < 3; c++) { memcpy(dst->tskip[c] + off, src->tskip[c], (size_t)n); memcpy(dst->cbf[c] + off, src->cbf[c], (size_t)n); }
  memcpy(dst->intra_dir[0] + off, src->intra_dir[0], (size_t)n);
  memcpy(dst->intra_dir[1] + off, src->intra_dir[1], (size_t)n);
  memcpy(dst->skip + off, src->skip, (size_t)n); memcpy(dst->merge_flag + off, src->merge_flag, (size_t)n);
  memcpy(dst->merge_idx + off, src->merge_idx, (size_t)n); memcpy(dst->inter_dir + off, src->inter_dir, (size_t)n);
  memcpy(dst->mvp_idx + off, src->mvp_idx, (size_t)n); memcpy(dst->ref_idx + off, src->ref_idx, (size_t)n);
  memcpy(dst->mv + off, src->mv, sizeof(src->mv[0]) * (size_t)n); memcpy(dst->mvd + off, src->mvd, sizeof(src->mvd[0]) * (size_t)n);
  memcpy(dst->coef[0] + off * 16, src->coef[0], sizeof(int32_t) * (size_t)(n * 16));
  memcpy(dst->coef[1] + off * 4, src->coef[1], sizeof(int32_t) * (size_t)(n * 4));
  memcpy(dst->coef[2] + off * 4, src->coef[2], sizeof(int32_t) * (size_t)(n * 4));
}
/* copyToPic, TComDataCU.cpp:992-1065 */
static void cu_copy_to_pic(HmoEnc *e, const HmoCU *cu)
{
  HmoCtu *p = &e->pic[e->cur_ctu];
  int n = cu->nparts, off = cu->zidx;
  p->total_cost = cu->cost; p->total_dist = cu->dist; p->total_bits = cu->bits; p->total_bins = cu->bins;
  memcpy(p->depth + off, cu->depth, (size_t)n);
  for (int i = 0; i < n; i++) p->width[off + i] = p->height[off + i] = (uint8_t)(HMO_CTU >> cu->depth[i]);   /* the sub-CUs' own sizes (m_puhWidth / m_puhHeight are copied per partition) */
  memcpy(p->skip + off, cu->skip, (size_t)n);
  memcpy(p->merge_flag + off, cu->merge_flag, (size_t)n); memcpy(p->merge_idx + off, cu->merge_idx, (size_t)n);
  memcpy(p->inter_dir + off, cu->inter_dir, (size_t)n); memcpy(p->mvp_idx + off, cu->mvp_idx, (size_t)n); memcpy(p->ref_idx + off, cu->ref_idx, (size_t)n);
  memcpy(p->mv + off, cu->mv, sizeof(cu->mv[0]) * (size_t)n); memcpy(p->mvd + off, cu->mvd, sizeof(cu->mvd[0]) * (size_t)n);
  memcpy(p->part_size + off, cu->part_size, (size_t)n);
  memcpy(p->pred_mode + off, cu->pred_mode, (size_t)n);
  memset(p->qp + off, e->p.qp, (size_t)n);
  memcpy(p->tr_idx + off, cu->tr_idx, (size_t)n);
  for (int c = 0; c < 3; c++) { memcpy(p->tskip[c] + off, cu->tskip[c], (size_t)n); memcpy(p->cbf[c] + off, cu->cbf[c], (size_t)n); }
  memcpy(p->intra_dir[0] + off, cu->intra_dir[0], (size_t)n);
  memcpy(p->intra_dir[1] + off, cu->intra_dir[1], (size_t)n);
  memcpy(p->coeff_y + off * 16, cu->coef[0], sizeof(int32_t) * (size_t)(n * 16));
  memcpy(p->coeff_cb + off * 4, cu->coef[1], sizeof(int32_t) * (size_t)(n * 4));
  memcpy(p->coeff_cr + off * 4, cu->coef[2], sizeof(int32_t) * (size_t)(n * 4));
  /* An inter CU whose residual was dropped (skip, or the root-cbf-zero choice of encodeResAndCalcRdInterCU,
   * TEncSearch.cpp:4465-4478) keeps whatever an earlier candidate left in m_pcTrCoeff; HM never reads it (cbf 0).  The
   * published CTU carries zeros there, so that the output does not depend on buffer history. */
  for (int i = 0; i < n; i++) {
    if (cu->pred_mode[i] != HMO_MODE_INTER) continue;
    if (!cu->cbf[0][i]) memset(p->coeff_y + (off + i) * 16, 0, sizeof(int32_t) * 16);
    if (!cu->cbf[1][i]) memset(p->coeff_cb + (off + i) * 4, 0, sizeof(int32_t) * 4);
    if (!cu->cbf[2][i]) memset(p->coeff_cr + (off + i) * 4, 0, sizeof(int32_t) * 4);
  }
}

/* neighbour field access: inside the working CU -> its arrays, else the committed picture
 * data (getPULeft/getPUAbove returning `this` vs. getPic()->getCtu(), TComDataCU.cpp:1071-1140) */
static int nb_depth(const HmoEnc *e, const HmoCU *cu, int lx, int ly)
{ return inside_cu(cu, lx, ly) ? cu->depth[zidx_of(lx, ly) - cu->zidx] : e->pic[ctu_of(e, lx, ly)].depth[zidx_of(lx, ly)]; }
static int nb_luma_dir(const HmoEnc *e, const HmoCU *cu, int lx, int ly)
{
  if (inside_cu(cu, lx, ly)) { int p = zidx_of(lx, ly) - cu->zidx; return cu->pred_mode[p] == HMO_MODE_INTRA ? cu->intra_dir[0][p] : HMO_DC; }
  const HmoCtu *c = &e->pic[ctu_of(e, lx, ly)]; int p = zidx_of(lx, ly);
  return c->pred_mode[p] == HMO_MODE_INTRA ? c->intra_dir[0][p] : HMO_DC;
}
static int left_ctu_ok(const HmoEnc *e, int lx, int ly)     /* slice restriction of getPULeft */
{ if (lx == 0) return 0; if (lx & 63) return 1; return ctu_of(e, lx, ly) - 1 >= e->slice_start; }
static int above_ctu_ok(const HmoEnc *e, int lx, int ly)
{ if (ly == 0) return 0; if (ly & 63) return 1; return ctu_of(e, lx, ly) - e->w_ctu >= e->slice_start; }

/* TComDataCU::getIntraDirPredictor, TComDataCU.cpp:1542-1624 (luma) ; returns *piMode */
static int intra_dir_predictor(const HmoEnc *e, const HmoCU *cu, int part, int preds[3])
{
  int z = cu->zidx + part;
  int lx = (cu->x & ~63) + part_x(z), ly = (cu->y & ~63) + part_y(z);
  int left = left_ctu_ok(e, lx, ly) ? nb_luma_dir(e, cu, lx - 1, ly) : HMO_DC;
  int above = ((ly & 63) != 0) ? nb_luma_dir(e, cu, lx, ly - 1) : HMO_DC;   /* planarAtCtuBoundary */
  if (left == above) {
    if (left > 1) { preds[0] = left; preds[1] = ((left + 29) % 32) + 2; preds[2] = ((left - 1) % 32) + 2; }
    else { preds[0] = HMO_PLANAR; preds[1] = HMO_DC; preds[2] = HMO_VER; }
    return 1;
  }
  preds[0] = left; preds[1] = above;
  if (left && above) preds[2] = HMO_PLANAR;
  else preds[2] = (left + above) < 2 ? HMO_VER : HMO_DC;
  return 2;
}

/* ------------------------------------------------------------------------------------
 * syntax elements (TEncSbac.cpp)
 * ---------------------------------------------------------------------------------- */
static void code_skip_flag(HmoEnc *e, const HmoCU *cu, int part);      /* hmo_inter.h */
static void code_pred_mode(HmoEnc *e, const HmoCU *cu, int part);
static void encode_cu_syntax_inter(HmoEnc *e, const HmoCU *cu, int cuPart, int depth);
/* codeSplitFlag, TEncSbac.cpp:613-628 + getCtxSplitFlag, TComDataCU.cpp:1626-1640 */
static void code_split_flag(HmoEnc *e, const HmoCU *cu, int part, int depth)
{
  if (depth == HMO_MAXDEPTH) return;
  int z = cu->zidx + part;
  int lx = (cu->x & ~63) + part_x(z), ly = (cu->y & ~63) + part_y(z);
  int ctx = 0;
  if (left_ctu_ok(e, lx, ly)) ctx += nb_depth(e, cu, lx - 1, ly) > depth;
  if (above_ctu_ok(e, lx, ly)) ctx += nb_depth(e, cu, lx, ly - 1) > depth;
  hmo_enc_bin(e, cu->depth[part] > depth, HMO_CTX_SPLIT + ctx);
}
/* codePartSize (intra), TEncSbac.cpp:436-448 */
static void code_part_size(HmoEnc *e, const HmoCU *cu, int part, int depth)
{ if (depth == HMO_MAXDEPTH) hmo_enc_bin(e, cu->part_size[part] == HMO_SIZE_2Nx2N, HMO_CTX_PARTSIZE); }
/* codeIntraDirLumaAng, TEncSbac.cpp:643-696 */
static void code_intra_dir_luma(HmoEnc *e, const HmoCU *cu, int part, int multiple)
{
  int dir[4], preds[4][3], predIdx[4];
  int partNum = multiple ? (cu->part_size[part] == HMO_SIZE_NxN ? 4 : 1) : 1;
  int partOffset = (HMO_NPART >> (cu->depth[part] << 1)) >> 2;
  for (int j = 0; j < partNum; j++) {
    dir[j] = cu->intra_dir[0][part + partOffset * j];
    intra_dir_predictor(e, cu, part + partOffset * j, preds[j]);
    predIdx[j] = -1;
    for (int i = 0; i < 3; i++) if (dir[j] == preds[j][i]) predIdx[j] = i;
    hmo_enc_bin(e, predIdx[j] != -1, HMO_CTX_INTRA_LUMA);
  }
  for (int j = 0; j < partNum; j++) {
    if (predIdx[j] != -1) hmo_enc_bins_ep(e, predIdx[j] ? 2 : 1);
    else hmo_enc_bins_ep(e, 5);
  }
}
/* getAllowedChromaDir, TComDataCU.cpp:1509-1533 */
static void allowed_chroma_dir(const HmoCU *cu, int part, int list[5])
{
  list[0] = HMO_PLANAR; list[1] = HMO_VER; list[2] = HMO_HOR; list[3] = HMO_DC; list[4] = HMO_DM_CHROMA;
  int luma = cu->intra_dir[0][part];
  for (int i = 0; i < 4; i++) if (luma == list[i]) { list[i] = 34; break; }
}
/* codeIntraDirChroma, TEncSbac.cpp:698-725 */
static void code_intra_dir_chroma(HmoEnc *e, const HmoCU *cu, int part)
{
  if (cu->intra_dir[1][part] == HMO_DM_CHROMA) hmo_enc_bin(e, 0, HMO_CTX_CHROMA_PRED);
  else { hmo_enc_bin(e, 1, HMO_CTX_CHROMA_PRED); hmo_enc_bins_ep(e, 2); }
}
/* codeQtCbf, TEncSbac.cpp:920-995 (square 4:2:0 blocks only) + getCtxQtCbf TComDataCU.cpp:1642-1656 */
static void code_qt_cbf(HmoEnc *e, const HmoCU *cu, const HmoTU *tu, int comp, int lowestLevel)
{
  if (comp == 0) {
    hmo_enc_bin(e, (cu->cbf[0][tu->part] >> tu->tr_depth) & 1, HMO_CTX_CBF_LUMA + (tu->tr_depth == 0 ? 1 : 0));
  } else {
    int canQuadSplit = tu->cw >= 8;
    int lowestDepth = tu->tr_depth + ((!lowestLevel && !canQuadSplit) ? 1 : 0);
    hmo_enc_bin(e, (cu->cbf[comp][tu_part_c(tu)] >> lowestDepth) & 1, HMO_CTX_CBF_CHROMA + tu->tr_depth);
  }
}
static int min_tu_log2_in_cu(const HmoCU *cu, int part)       /* getQuadtreeTULog2MinSizeInCU, TComDataCU.cpp:1658-1686 */
{
  int log2Cb = 6 - cu->depth[part];
  int split = cu->part_size[part] == HMO_SIZE_NxN;
  if (log2Cb < HMO_LOG2_MINTU + HMO_TU_MAXDEPTH_INTRA - 1 + split) return HMO_LOG2_MINTU;
  int m = log2Cb - (HMO_TU_MAXDEPTH_INTRA - 1 + split);
  return m > HMO_LOG2_MAXTU ? HMO_LOG2_MAXTU : m;
}

/* ------------------------------------------------------------------------------------
 * search-time bit counting (TEncSearch.cpp:866-1090)
 * ---------------------------------------------------------------------------------- */
/* xEncSubdivCbfQT */
static void enc_subdiv_cbf_qt(HmoEnc *e, const HmoCU *cu, const HmoTU *tu, int bLuma, int bChroma)
{
  int subdiv = cu->tr_idx[tu->part] > tu->tr_depth;
  if (cu->part_size[0] == HMO_SIZE_NxN && tu->tr_depth == 0) { /* inferred */ }
  else if (tu->log2 > HMO_LOG2_MAXTU) { }
  else if (tu->log2 == HMO_LOG2_MINTU) { }
  else if (tu->log2 == min_tu_log2_in_cu(cu, tu->part)) { }
  else if (bLuma) hmo_enc_bin(e, subdiv, HMO_CTX_SUBDIV + 5 - tu->log2);
  if (bChroma) {
    for (int comp = 1; comp < 3; comp++)
      if (tu->c_code_all && (tu->tr_depth == 0 || ((cu->cbf[comp][tu->part] >> (tu->tr_depth - 1)) & 1)))
        code_qt_cbf(e, cu, tu, comp, subdiv == 0);
  }
  if (subdiv) {
    for (int i = 0; i < 4; i++) { HmoTU c; tu_child(&c, tu, i, 0); enc_subdiv_cbf_qt(e, cu, &c, bLuma, bChroma); }
  } else if (bLuma) code_qt_cbf(e, cu, tu, 0, 1);
}
/* xEncCoeffQT (+ TEncEntropy::encodeCoeffNxN, TEncEntropy.cpp:660-690) */
static void enc_coeff_qt(HmoEnc *e, const HmoCU *cu, const HmoTU *tu, int comp, int realCoeff)
{
  if (cu->tr_idx[tu->part] > tu->tr_depth) {
    for (int i = 0; i < 4; i++) { HmoTU c; tu_child(&c, tu, i, 0); enc_coeff_qt(e, cu, &c, comp, realCoeff); }
    return;
  }
  if (comp ? tu->cw == 0 : 0) return;
  int layer = HMO_LOG2_MAXTU - tu->log2;
  const int32_t *buf = realCoeff ? cu->coef[comp] : e->qt_coef[comp][layer];
  int off = comp ? tu->off_c : tu->off_y;
  if ((cu->cbf[comp][tu->part] >> tu->tr_depth) & 1) {
    int log2 = comp ? (tu->cw == 4 ? 2 : tu->cw == 8 ? 3 : tu->cw == 16 ? 4 : 5) : tu->log2;
    hmo_code_coeff_nxn(e, cu, buf + off, log2, comp, comp ? tu_part_c(tu) : tu->part);
  }
}
/* xEncIntraHeader */
static void enc_intra_header(HmoEnc *e, const HmoCU *cu, int trDepth, int part, int bLuma, int bChroma)
{
  if (bLuma) {
    if (part == 0 && e->p.slice_type != HMO_SLICE_I) { code_skip_flag(e, cu, 0); code_pred_mode(e, cu, 0); }   /* TEncSearch.cpp:990-1000 */
    if (part == 0) code_part_size(e, cu, 0, cu->depth[0]);
    if (cu->part_size[0] == HMO_SIZE_2Nx2N) { if (part == 0) code_intra_dir_luma(e, cu, 0, 0); }
    else { int q = cu->nparts >> 2; if (trDepth > 0 && (part % q) == 0) code_intra_dir_luma(e, cu, part, 0); }
  }
  if (bChroma && part == 0) code_intra_dir_chroma(e, cu, part);
}
/* xGetIntraBitsQT */
static uint32_t intra_bits_qt(HmoEnc *e, const HmoCU *cu, const HmoTU *tu, int bLuma, int bChroma)
{
  hmo_reset_bits(e);
  enc_intra_header(e, cu, tu->tr_depth, tu->part, bLuma, bChroma);
  enc_subdiv_cbf_qt(e, cu, tu, bLuma, bChroma);
  if (bLuma) enc_coeff_qt(e, cu, tu, 0, 0);
  if (bChroma) { enc_coeff_qt(e, cu, tu, 1, 0); enc_coeff_qt(e, cu, tu, 2, 0); }
  return hmo_bits(e);
}
/* xGetIntraBitsQTChroma */
static uint32_t intra_bits_qt_chroma(HmoEnc *e, const HmoCU *cu, const HmoTU *tu, int comp)
{ hmo_reset_bits(e); enc_coeff_qt(e, cu, tu, comp, 0); return hmo_bits(e); }

/* ------------------------------------------------------------------------------------
 * one TU trial: xIntraCodingTUBlock, TEncSearch.cpp:1092-1387
 * ---------------------------------------------------------------------------------- */
static uint8_t *yuv_plane(HmoYuv *b, int comp) { return comp == 0 ? b->y : (comp == 1 ? b->u : b->v); }

static void intra_coding_tu_block(HmoEnc *e, HmoCU *cu, const HmoTU *tu, int comp, uint32_t *dist, int save1load2)
{
  if (comp && tu->cw == 0) return;
  const int d = cu->depth_cu;
  const int N = comp ? tu->cw : (1 << tu->log2);
  int log2 = 2; while ((1 << log2) < N) log2++;
  const int bx = comp ? tu->cx : tu->x, by = comp ? tu->cy : tu->y;
  const int bs = comp ? 32 : 64;
  const int part = tu->part;                               /* uiAbsPartIdx = GetAbsPartIdxTU() */
  const int layer = HMO_LOG2_MAXTU - tu->log2;
  uint8_t *org = yuv_plane(e->org_yuv[d], comp) + by * bs + bx;
  uint8_t *pred = yuv_plane(e->pred_temp[d], comp) + by * bs + bx;
  uint8_t *recqt = yuv_plane(&e->qt_rec[layer], comp) + by * bs + bx;
  const int sh = comp ? 1 : 0;
  const int px = (cu->x >> sh) + bx, py = (cu->y >> sh) + by;       /* position in the component plane */
  uint8_t *recpic = e->rec[comp] + py * e->stride[comp] + px;
  int32_t *coef = e->qt_coef[comp][layer] + (comp ? tu->off_c : tu->off_y);
  const int useTS = cu->tskip[comp][part];
  int mode = cu->intra_dir[comp ? 1 : 0][part];
  if (comp && mode == HMO_DM_CHROMA) mode = cu->intra_dir[0][part & ~3];
  e->n_tu_trials++;

  if (save1load2 != 2) {
    uint8_t ref[4 * 64 + 1], reff[4 * 64 + 1];
    hmo_build_ref(e, comp, px, py, log2, 0, ref);
    int filt = hmo_use_filtered_ref(mode, log2, comp == 0);
    if (filt) hmo_filter_ref(ref, reff, N, comp == 0 && e->p.strong_smoothing);
    hmo_intra_pred(filt ? reff : ref, 0, log2, mode, comp == 0, pred, bs);
    if (save1load2 == 1) for (int y = 0; y < N; y++) memcpy(e->shared_pred[comp] + y * N, pred + y * bs, (size_t)N);
  } else {
    for (int y = 0; y < N; y++) memcpy(pred + y * bs, e->shared_pred[comp] + y * N, (size_t)N);
  }
  int16_t resi[32 * 32];
  for (int y = 0; y < N; y++) for (int x = 0; x < N; x++) resi[y * N + x] = (int16_t)(org[y * bs + x] - pred[y * bs + x]);

  if (comp == 0) memset(cu->tr_idx + part, tu->tr_depth, (size_t)tu->nparts);     /* setTrIdxSubParts */

  /* transformNxN, TComTrQuant.cpp:1376-1459 */
  int32_t tcoef[32 * 32];
  if (useTS) { for (int i = 0; i < N * N; i++) tcoef[i] = (int32_t)resi[i] << (15 - 8 - log2); }
  else hmo_fwd_transform(resi, N, tcoef, log2, comp == 0 && log2 == 2);
  int absSum = hmo_rdoq(e, cu, tu, comp, tcoef, coef, log2, comp ? tu_part_c(tu) : part, useTS);
  { int np = comp ? tu_nparts_c(tu) : tu->nparts;                                   /* setCbfPartRange */
    memset(cu->cbf[comp] + part, (absSum > 0 ? 1 : 0) << tu->tr_depth, (size_t)np); }

  if (absSum > 0) {
    int32_t dq[32 * 32];
    hmo_dequant(coef, dq, N * N, log2, comp ? e->p.qp_c : e->p.qp);
    if (useTS) { int s = 15 - 8 - log2; for (int i = 0; i < N * N; i++) resi[i] = (int16_t)((dq[i] + (1 << (s - 1))) >> s); }
    else hmo_inv_transform(dq, resi, N, log2, comp == 0 && log2 == 2);
  } else {
    memset(coef, 0, sizeof(int32_t) * (size_t)(N * N));
    memset(resi, 0, sizeof(int16_t) * (size_t)(N * N));
  }
  for (int y = 0; y < N; y++)
    for (int x = 0; x < N; x++) {
      int v = pred[y * bs + x] + resi[y * N + x];
      uint8_t r = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
      pred[y * bs + x] = r; recqt[y * bs + x] = r; recpic[y * e->stride[comp] + x] = r;
    }
  uint32_t sse = hmo_sse(org, bs, pred, bs, N, N);
  if (comp) *dist += (uint32_t)(e->p.chroma_weight * (double)sse);                    /* getDistPart, TComRdCost.cpp:447-450 */
  else *dist += sse;
}

/* xStoreIntraResultQT / xLoadIntraResultQT, TEncSearch.cpp:1760-1850 */
static void store_intra_result_qt(HmoEnc *e, const HmoTU *tu, int comp)
{
  if (comp && tu->cw == 0) return;
  int N = comp ? tu->cw : (1 << tu->log2), layer = HMO_LOG2_MAXTU - tu->log2, bs = comp ? 32 : 64;
  int bx = comp ? tu->cx : tu->x, by = comp ? tu->cy : tu->y;
  memcpy(e->ts_coef[comp], e->qt_coef[comp][layer] + (comp ? tu->off_c : tu->off_y), sizeof(int32_t) * (size_t)(N * N));
  uint8_t *s = yuv_plane(&e->qt_rec[layer], comp) + by * bs + bx, *t = yuv_plane(&e->ts_rec, comp) + by * bs + bx;
  for (int y = 0; y < N; y++) memcpy(t + y * bs, s + y * bs, (size_t)N);
}
static void load_intra_result_qt(HmoEnc *e, const HmoCU *cu, const HmoTU *tu, int comp)
{
  if (comp && tu->cw == 0) return;
  int N = comp ? tu->cw : (1 << tu->log2), layer = HMO_LOG2_MAXTU - tu->log2, bs = comp ? 32 : 64, sh = comp ? 1 : 0;
  int bx = comp ? tu->cx : tu->x, by = comp ? tu->cy : tu->y;
  memcpy(e->qt_coef[comp][layer] + (comp ? tu->off_c : tu->off_y), e->ts_coef[comp], sizeof(int32_t) * (size_t)(N * N));
  uint8_t *t = yuv_plane(&e->qt_rec[layer], comp) + by * bs + bx, *s = yuv_plane(&e->ts_rec, comp) + by * bs + bx;
  uint8_t *pic = e->rec[comp] + ((cu->y >> sh) + by) * e->stride[comp] + (cu->x >> sh) + bx;
  for (int y = 0; y < N; y++) { memcpy(t + y * bs, s + y * bs, (size_t)N); memcpy(pic + y * e->stride[comp], s + y * bs, (size_t)N); }
}

/* ------------------------------------------------------------------------------------
 * xRecurIntraCodingLumaQT, TEncSearch.cpp:1393-1713
 * ---------------------------------------------------------------------------------- */
static void recur_intra_coding_luma_qt(HmoEnc *e, HmoCU *cu, const HmoTU *tu, int checkFirst, uint32_t *distY, double *rdCost)
{
  const int d = cu->depth_cu, part = tu->part, trDepth = tu->tr_depth, fullDepth = d + trDepth, log2 = tu->log2;
  int checkFull = log2 <= HMO_LOG2_MAXTU;
  int checkSplit = log2 > min_tu_log2_in_cu(cu, part);
  if (checkFirst && checkFull) checkSplit = 0;                                   /* HHI_RQT_INTRA_SPEEDUP */
  double singleCost = HMO_MAX_DOUBLE; uint32_t singleDist = 0, singleCbf = 0;
  int checkTS = e->p.transform_skip && log2 == 2;
  if (e->p.transform_skip_fast) checkTS = checkTS && (cu->part_size[part] == HMO_SIZE_NxN);
  int bestModeId = 0;

  if (checkFull) {
    if (checkTS) {
      e->slot[fullDepth][CI_QT_TRAFO_ROOT] = e->goon;
      for (int modeId = 0; modeId < 2; modeId++) {
        uint32_t tmpDist = 0; double tmpCost;
        memset(cu->tskip[0] + part, modeId, (size_t)tu->nparts);
        intra_coding_tu_block(e, cu, tu, 0, &tmpDist, modeId == 0 ? 1 : 2);
        uint32_t tmpCbf = (cu->cbf[0][part] >> trDepth) & 1;
        if (modeId == 1 && tmpCbf == 0) tmpCost = HMO_MAX_DOUBLE;
        else { uint32_t bits = intra_bits_qt(e, cu, tu, 1, 0); tmpCost = calc_rd_cost(e, bits, tmpDist); }
        if (tmpCost < singleCost) {
          singleCost = tmpCost; singleDist = tmpDist; singleCbf = tmpCbf; bestModeId = modeId;
          if (bestModeId == 0) { store_intra_result_qt(e, tu, 0); e->slot[fullDepth][CI_TEMP_BEST] = e->goon; }
        }
        if (modeId == 0) e->goon = e->slot[fullDepth][CI_QT_TRAFO_ROOT];
      }
      memset(cu->tskip[0] + part, bestModeId, (size_t)tu->nparts);
      if (bestModeId == 0) {
        load_intra_result_qt(e, cu, tu, 0);
        memset(cu->cbf[0] + part, (int)(singleCbf << trDepth), (size_t)tu->nparts);
        e->goon = e->slot[fullDepth][CI_TEMP_BEST];
      }
    } else {
      if (checkSplit) e->slot[fullDepth][CI_QT_TRAFO_ROOT] = e->goon;
      memset(cu->tskip[0] + part, 0, (size_t)tu->nparts);
      intra_coding_tu_block(e, cu, tu, 0, &singleDist, 0);
      if (checkSplit) singleCbf = (cu->cbf[0][part] >> trDepth) & 1;
      uint32_t bits = intra_bits_qt(e, cu, tu, 1, 0);
      singleCost = calc_rd_cost(e, bits, singleDist);
    }
  }

  if (checkSplit) {
    if (checkFull) { e->slot[fullDepth][CI_QT_TRAFO_TEST] = e->goon; e->goon = e->slot[fullDepth][CI_QT_TRAFO_ROOT]; }
    else e->slot[fullDepth][CI_QT_TRAFO_ROOT] = e->goon;
    double splitCost = 0.0; uint32_t splitDist = 0, splitCbf = 0;
    for (int i = 0; i < 4; i++) {
      HmoTU c; tu_child(&c, tu, i, 0);
      recur_intra_coding_luma_qt(e, cu, &c, checkFirst, &splitDist, &splitCost);
      splitCbf |= (cu->cbf[0][c.part] >> c.tr_depth) & 1;
    }
    if (splitCbf) for (int o = 0; o < tu->nparts; o++) cu->cbf[0][part + o] |= (uint8_t)(1 << trDepth);
    e->goon = e->slot[fullDepth][CI_QT_TRAFO_ROOT];
    uint32_t splitBits = intra_bits_qt(e, cu, tu, 1, 0);
    splitCost = calc_rd_cost(e, splitBits, splitDist);
    if (splitCost < singleCost) { *distY += splitDist; *rdCost += splitCost; return; }
    e->goon = e->slot[fullDepth][CI_QT_TRAFO_TEST];
    memset(cu->tr_idx + part, trDepth, (size_t)tu->nparts);
    memset(cu->cbf[0] + part, (int)(singleCbf << trDepth), (size_t)tu->nparts);
    memset(cu->tskip[0] + part, bestModeId, (size_t)tu->nparts);
    { int N = 1 << log2, layer = HMO_LOG2_MAXTU - log2;                          /* restore recon for next blocks */
      uint8_t *s = e->qt_rec[layer].y + tu->y * 64 + tu->x;
      uint8_t *p = e->rec[0] + (cu->y + tu->y) * e->stride[0] + cu->x + tu->x;
      for (int y = 0; y < N; y++) memcpy(p + y * e->stride[0], s + y * 64, (size_t)N); }
  }
  *distY += singleDist; *rdCost += singleCost;
}

/* xSetIntraResultLumaQT, TEncSearch.cpp:1717-1757 */
static void set_intra_result_luma_qt(HmoEnc *e, HmoCU *cu, const HmoTU *tu, HmoYuv *reco)
{
  if (cu->tr_idx[tu->part] == tu->tr_depth) {
    int N = 1 << tu->log2, layer = HMO_LOG2_MAXTU - tu->log2;
    memcpy(cu->coef[0] + tu->off_y, e->qt_coef[0][layer] + tu->off_y, sizeof(int32_t) * (size_t)(N * N));
    for (int y = 0; y < N; y++) memcpy(reco->y + (tu->y + y) * 64 + tu->x, e->qt_rec[layer].y + (tu->y + y) * 64 + tu->x, (size_t)N);
  } else for (int i = 0; i < 4; i++) { HmoTU c; tu_child(&c, tu, i, 0); set_intra_result_luma_qt(e, cu, &c, reco); }
}

/* ------------------------------------------------------------------------------------
 * estIntraPredLumaQT, TEncSearch.cpp:2178-2655
 * ---------------------------------------------------------------------------------- */
static void est_intra_pred_luma_qt(HmoEnc *e, HmoCU *cu)
{
  const int d = cu->depth_cu;
  const int initTrDepth = cu->part_size[0] == HMO_SIZE_2Nx2N ? 0 : 1;
  const int numPU = 1 << (2 * initTrDepth);
  const int qNumParts = cu->nparts >> 2;
  uint32_t overallDistY = 0;
  HmoTU root; tu_root(&root, cu);

  for (int pu = 0; pu < numPU; pu++) {
    HmoTU tu;
    if (initTrDepth == 0) tu = root; else tu_child(&tu, &root, pu, 0);
    const int partOffset = tu.part, N = 1 << tu.log2, log2 = tu.log2;
    int numModesForFullRD = hmo_rd_mode_num[log2 - 2];
    int rdModeList[35]; double candCost[35];
    /* ---- RMD over the 35 modes (TEncSearch.cpp:2300-2361) ---- */
    {
      uint8_t ref[4 * 64 + 1], reff[4 * 64 + 1];
      hmo_build_ref(e, 0, cu->x + tu.x, cu->y + tu.y, log2, 0, ref);
      hmo_filter_ref(ref, reff, N, e->p.strong_smoothing);
      for (int i = 0; i < numModesForFullRD; i++) candCost[i] = HMO_MAX_DOUBLE;
      uint8_t *org = e->org_yuv[d]->y + tu.y * 64 + tu.x;
      uint8_t *pred = e->pred_temp[d]->y + tu.y * 64 + tu.x;
      for (int mode = 0; mode < 35; mode++) {
        int filt = hmo_use_filtered_ref(mode, log2, 1);
        hmo_intra_pred(filt ? reff : ref, 0, log2, mode, 1, pred, 64);
        uint32_t sad = hmo_satd(org, 64, pred, 64, N, N);
        e->n_rmd++;
        /* xModeBitsIntra, TEncSearch.cpp:5313-5340: only the intra-dir context and the bit counter are reloaded */
        e->goon.frac = e->slot[d][CI_CURR_BEST].frac;
        e->goon.ctx[HMO_CTX_INTRA_LUMA] = e->slot[d][CI_CURR_BEST].ctx[HMO_CTX_INTRA_LUMA];
        uint8_t orig = cu->intra_dir[0][partOffset];
        cu->intra_dir[0][partOffset] = (uint8_t)mode;
        hmo_reset_bits(e);
        code_intra_dir_luma(e, cu, partOffset, 0);
        cu->intra_dir[0][partOffset] = orig;
        uint32_t modeBits = hmo_bits(e);
        double cost = (double)sad + (double)modeBits * e->p.sqrt_lambda;
        /* xUpdateCandList, TEncSearch.cpp:5345-5370 */
        int shift = 0;
        while (shift < numModesForFullRD && cost < candCost[numModesForFullRD - 1 - shift]) shift++;
        if (shift != 0) {
          for (int i = 1; i < shift; i++) {
            rdModeList[numModesForFullRD - i] = rdModeList[numModesForFullRD - 1 - i];
            candCost[numModesForFullRD - i] = candCost[numModesForFullRD - 1 - i];
          }
          rdModeList[numModesForFullRD - shift] = mode; candCost[numModesForFullRD - shift] = cost;
        }
      }
      int preds[3];
      int numCand = intra_dir_predictor(e, cu, partOffset, preds);
      for (int j = 0; j < numCand; j++) {
        int included = 0;
        for (int i = 0; i < numModesForFullRD; i++) included |= (preds[j] == rdModeList[i]);
        if (!included) rdModeList[numModesForFullRD++] = preds[j];
      }
    }
    HmoPuTrace *ptr = NULL;
    if (e->pu_trace) {                                   /* candidate list and CandCostList as the RMD leaves them */
      const int z = cu->zidx + partOffset, nRmd = hmo_rd_mode_num[log2 - 2];
      ptr = e->pu_trace + (size_t)e->cur_ctu * 341 + (initTrDepth ? 85 + z : (d == 0 ? 0 : d == 1 ? 1 + (z >> 6) : d == 2 ? 5 + (z >> 4) : 21 + (z >> 2)));
      ptr->n_rmd = (uint8_t)nRmd; ptr->n_rd = (uint8_t)numModesForFullRD; ptr->pad = 0;
      for (int i = 0; i < 12; i++) ptr->rd_mode[i] = (uint8_t)(i < numModesForFullRD ? rdModeList[i] : 0);
      for (int i = 0; i < 8; i++) ptr->rmd_cost[i] = i < nRmd ? candCost[i] : 0.0;
    }
    /* ---- RDO over the candidates, no RQT split (TEncSearch.cpp:2447-2516) ---- */
    int bestPUMode = 0; uint32_t bestPUDist = 0; double bestPUCost = HMO_MAX_DOUBLE;
    for (int m = 0; m < numModesForFullRD; m++) {
      int orgMode = rdModeList[m];
      memset(cu->intra_dir[0] + partOffset, orgMode, (size_t)tu.nparts);
      e->goon = e->slot[d][CI_CURR_BEST];
      uint32_t puDist = 0; double puCost = 0.0;
      recur_intra_coding_luma_qt(e, cu, &tu, 1, &puDist, &puCost);
      if (puCost < bestPUCost) {
        bestPUMode = orgMode; bestPUDist = puDist; bestPUCost = puCost;
        set_intra_result_luma_qt(e, cu, &tu, e->reco_temp[d]);
        memcpy(e->tmp_tr_idx, cu->tr_idx + partOffset, (size_t)tu.nparts);
        for (int c = 0; c < 3; c++) { memcpy(e->tmp_cbf[c], cu->cbf[c] + partOffset, (size_t)tu.nparts); memcpy(e->tmp_tskip[c], cu->tskip[c] + partOffset, (size_t)tu.nparts); }
      }
    }
    /* ---- best mode again with the full RQT (TEncSearch.cpp:2518-2586) ---- */
    {
      int orgMode = bestPUMode;
      memset(cu->intra_dir[0] + partOffset, orgMode, (size_t)tu.nparts);
      e->goon = e->slot[d][CI_CURR_BEST];
      uint32_t puDist = 0; double puCost = 0.0;
      recur_intra_coding_luma_qt(e, cu, &tu, 0, &puDist, &puCost);
      if (puCost < bestPUCost) {
        bestPUMode = orgMode; bestPUDist = puDist; bestPUCost = puCost;
        set_intra_result_luma_qt(e, cu, &tu, e->reco_temp[d]);
        memcpy(e->tmp_tr_idx, cu->tr_idx + partOffset, (size_t)tu.nparts);
        for (int c = 0; c < 3; c++) { memcpy(e->tmp_cbf[c], cu->cbf[c] + partOffset, (size_t)tu.nparts); memcpy(e->tmp_tskip[c], cu->tskip[c] + partOffset, (size_t)tu.nparts); }
      }
    }
    overallDistY += bestPUDist;
    if (ptr) { ptr->best_mode = (uint8_t)bestPUMode; ptr->best_dist = bestPUDist; ptr->best_cost = bestPUCost; ptr->valid = 1; }
    memcpy(cu->tr_idx + partOffset, e->tmp_tr_idx, (size_t)tu.nparts);
    for (int c = 0; c < 3; c++) { memcpy(cu->cbf[c] + partOffset, e->tmp_cbf[c], (size_t)tu.nparts); memcpy(cu->tskip[c] + partOffset, e->tmp_tskip[c], (size_t)tu.nparts); }
    if (pu != numPU - 1) {                               /* recon of the best PU for the next PU's prediction */
      uint8_t *s = e->reco_temp[d]->y + tu.y * 64 + tu.x;
      uint8_t *p = e->rec[0] + (cu->y + tu.y) * e->stride[0] + cu->x + tu.x;
      for (int y = 0; y < N; y++) memcpy(p + y * e->stride[0], s + y * 64, (size_t)N);
    }
    memset(cu->intra_dir[0] + partOffset, bestPUMode, (size_t)tu.nparts);
  }
  if (numPU > 1) {
    uint8_t cy = 0, cu_ = 0, cv = 0;
    for (int p = 0, idx = 0; p < 4; p++, idx += qNumParts) {
      cy |= (cu->cbf[0][idx] >> 1) & 1; cu_ |= (cu->cbf[1][idx] >> 1) & 1; cv |= (cu->cbf[2][idx] >> 1) & 1;
    }
    for (int o = 0; o < 4 * qNumParts; o++) { cu->cbf[0][o] |= cy; cu->cbf[1][o] |= cu_; cu->cbf[2][o] |= cv; }
  }
  e->goon = e->slot[d][CI_CURR_BEST];
  cu->dist = overallDistY;
}

/* ------------------------------------------------------------------------------------
 * chroma: xRecurIntraChromaCodingQT, TEncSearch.cpp:1916-2120
 * ---------------------------------------------------------------------------------- */
static void recur_intra_chroma_coding_qt(HmoEnc *e, HmoCU *cu, const HmoTU *tu, uint32_t *dist)
{
  const int part = tu->part, trDepth = tu->tr_depth;
  if (cu->tr_idx[part] == trDepth) {
    if (tu->cw == 0) return;
    const int fullDepth = cu->depth_cu + trDepth;
    int checkTS = e->p.transform_skip && tu->cw <= 4;
    if (e->p.transform_skip_fast) {
      checkTS = checkTS && (tu->log2 == 2);
      if (checkTS) {
        int nb = 0, maxp = part + (tu->c_code_all ? 1 : 4);
        for (int p = part; p < maxp; p++) nb += cu->tskip[0][p];
        checkTS = checkTS && (nb > 0);
      }
    }
    const int subPart = tu_part_c(tu), nPartsC = tu_nparts_c(tu);
    for (int comp = 1; comp < 3; comp++) {
      e->slot[fullDepth][CI_QT_TRAFO_ROOT] = e->goon;
      double singleCost = HMO_MAX_DOUBLE; int bestModeId = 0; uint32_t singleDistC = 0, singleCbfC = 0;
      double tmpCost = 0; int bestTS = 0;
      const int modesToTest = checkTS ? 2 : 1;
      int currModeId = 0;
      for (int tsMode = 0; tsMode < modesToTest; tsMode++) {
        memset(cu->tskip[comp] + subPart, tsMode, (size_t)nPartsC);
        currModeId++;
        const int isOne = modesToTest == 1, isLast = currModeId == modesToTest;
        int s1l2 = isOne ? 0 : (tsMode == 0 ? 1 : 2);
        uint32_t tmpDist = 0;
        intra_coding_tu_block(e, cu, tu, comp, &tmpDist, s1l2);
        uint32_t tmpCbf = (cu->cbf[comp][subPart] >> trDepth) & 1;
        if (tsMode == 1 && tmpCbf == 0) tmpCost = HMO_MAX_DOUBLE;
        else if (!isOne) { uint32_t bits = intra_bits_qt_chroma(e, cu, tu, comp); tmpCost = calc_rd_cost(e, bits, tmpDist); }
        if (tmpCost < singleCost) {
          singleCost = tmpCost; singleDistC = tmpDist; bestTS = tsMode; bestModeId = currModeId; singleCbfC = tmpCbf;
          if (!isOne && !isLast) { store_intra_result_qt(e, tu, comp); e->slot[fullDepth][CI_TEMP_BEST] = e->goon; }
        }
        if (!isOne && !isLast) e->goon = e->slot[fullDepth][CI_QT_TRAFO_ROOT];
      }
      if (bestModeId < modesToTest) {
        load_intra_result_qt(e, cu, tu, comp);
        memset(cu->cbf[comp] + subPart, (int)(singleCbfC << trDepth), (size_t)nPartsC);
        e->goon = e->slot[fullDepth][CI_TEMP_BEST];
      }
      memset(cu->tskip[comp] + subPart, bestTS, (size_t)nPartsC);
      *dist += singleDistC;
    }
  } else {
    uint32_t splitCbf[3] = { 0, 0, 0 };
    for (int i = 0; i < 4; i++) {
      HmoTU c; tu_child(&c, tu, i, 0);
      recur_intra_chroma_coding_qt(e, cu, &c, dist);
      for (int comp = 1; comp < 3; comp++) splitCbf[comp] |= (cu->cbf[comp][c.part] >> c.tr_depth) & 1;
    }
    for (int comp = 1; comp < 3; comp++)
      if (splitCbf[comp]) for (int o = 0; o < tu->nparts; o++) cu->cbf[comp][part + o] |= (uint8_t)(1 << trDepth);
  }
}
/* xSetIntraResultChromaQT, TEncSearch.cpp:2126-2175 */
static void set_intra_result_chroma_qt(HmoEnc *e, HmoCU *cu, const HmoTU *tu, HmoYuv *reco)
{
  if (tu->cw == 0) return;
  if (cu->tr_idx[tu->part] == tu->tr_depth) {
    int N = tu->cw, layer = HMO_LOG2_MAXTU - tu->log2;
    for (int comp = 1; comp < 3; comp++) {
      memcpy(cu->coef[comp] + tu->off_c, e->qt_coef[comp][layer] + tu->off_c, sizeof(int32_t) * (size_t)(N * N));
      uint8_t *s = yuv_plane(&e->qt_rec[layer], comp), *t = yuv_plane(reco, comp);
      for (int y = 0; y < N; y++) memcpy(t + (tu->cy + y) * 32 + tu->cx, s + (tu->cy + y) * 32 + tu->cx, (size_t)N);
    }
  } else for (int i = 0; i < 4; i++) { HmoTU c; tu_child(&c, tu, i, 0); set_intra_result_chroma_qt(e, cu, &c, reco); }
}
/* estIntraPredChromaQT, TEncSearch.cpp:2661-2810 (4:2:0: one chroma PU per CU) */
static void est_intra_pred_chroma_qt(HmoEnc *e, HmoCU *cu)
{
  const int d = cu->depth_cu, n = cu->nparts;
  HmoTU tu; tu_root(&tu, cu);
  int bestMode = 0; uint32_t bestDist = 0; double bestCost = HMO_MAX_DOUBLE;
  int modeList[5];
  allowed_chroma_dir(cu, 0, modeList);
  for (int m = 0; m < 5; m++) {
    e->goon = e->slot[d][CI_CURR_BEST];
    uint32_t dist = 0;
    memset(cu->intra_dir[1], modeList[m], (size_t)n);
    recur_intra_chroma_coding_qt(e, cu, &tu, &dist);
    if (e->p.transform_skip) e->goon = e->slot[d][CI_CURR_BEST];
    uint32_t bits = intra_bits_qt(e, cu, &tu, 0, 1);
    double cost = calc_rd_cost(e, bits, dist);
    if (cost < bestCost) {
      bestCost = cost; bestDist = dist; bestMode = modeList[m];
      set_intra_result_chroma_qt(e, cu, &tu, e->reco_temp[d]);
      for (int c = 1; c < 3; c++) { memcpy(e->tmp_cbf[c], cu->cbf[c], (size_t)n); memcpy(e->tmp_tskip[c], cu->tskip[c], (size_t)n); }
    }
  }
  for (int c = 1; c < 3; c++) { memcpy(cu->cbf[c], e->tmp_cbf[c], (size_t)n); memcpy(cu->tskip[c], e->tmp_tskip[c], (size_t)n); }
  memset(cu->intra_dir[1], bestMode, (size_t)n);
  cu->dist += bestDist;
  e->goon = e->slot[d][CI_CURR_BEST];
}

/* ------------------------------------------------------------------------------------
 * final-order CU syntax (TEncEntropy::encodeCoeff / xEncodeTransform, TEncEntropy.cpp:201-400)
 * ---------------------------------------------------------------------------------- */
static void encode_transform(HmoEnc *e, const HmoCU *cu, int cuPart, const HmoTU *tu)
{
  /* all indices below are relative to `cu`; cuPart = first partition of the coded CU inside `cu` */
  const int part = cuPart + tu->part;
  const int trIdx = tu->tr_depth;
  const int subdiv = cu->tr_idx[part] > trIdx;
  int cbf[3];
  for (int c = 0; c < 3; c++) cbf[c] = (cu->cbf[c][part] >> trIdx) & 1;
  const int cuDepth = cu->depth[part];
  if (cu->part_size[part] == HMO_SIZE_NxN && trIdx == 0) { }
  else if (tu->log2 > HMO_LOG2_MAXTU) { }
  else if (tu->log2 == HMO_LOG2_MINTU) { }
  else if (tu->log2 == min_tu_log2_in_cu(cu, part)) { }
  else hmo_enc_bin(e, subdiv, HMO_CTX_SUBDIV + 5 - tu->log2);
  (void)cuDepth;
  const int first = trIdx == 0;
  for (int comp = 1; comp < 3; comp++) {
    if (first || tu->c_code_all) {
      if (first || ((cu->cbf[comp][part] >> (trIdx - 1)) & 1)) {
        int canQuadSplit = tu->cwo >= 8;
        int lowestDepth = trIdx + ((subdiv && !canQuadSplit) ? 1 : 0);
        int pc = cuPart + tu_part_c(tu);
        hmo_enc_bin(e, (cu->cbf[comp][pc] >> lowestDepth) & 1, HMO_CTX_CBF_CHROMA + trIdx);
      }
    }
  }
  if (subdiv) {
    for (int i = 0; i < 4; i++) { HmoTU c; tu_child(&c, tu, i, 1); encode_transform(e, cu, cuPart, &c); }
    return;
  }
  hmo_enc_bin(e, cbf[0], HMO_CTX_CBF_LUMA + (trIdx == 0 ? 1 : 0));
  /* coefficients: Y, Cb, Cr of this TU.  A view CU positioned at cuPart gives codeCoeffNxN the
   * right relative indices. */
  for (int comp = 0; comp < 3; comp++) {
    if (comp && tu->cw == 0) continue;
    if (!cbf[comp]) continue;
    int N = comp ? tu->cw : (1 << tu->log2), log2 = 2; while ((1 << log2) < N) log2++;
    const int32_t *coef = cu->coef[comp] + (comp ? (cuPart * 4 + tu->off_c) : (cuPart * 16 + tu->off_y));
    /* hmo_code_coeff_nxn reads intra_dir/tskip at `part`; give absolute index inside cu */
    hmo_code_coeff_nxn(e, cu, coef, log2, comp, cuPart + (comp ? tu_part_c(tu) : tu->part));
  }
}
/* CU syntax as coded by xCheckRDCostIntra (TEncCu.cpp:2117-2141) and xEncodeCU (:1753-1778) */
static void encode_cu_syntax(HmoEnc *e, const HmoCU *cu, int cuPart, int depth)
{
  if (e->p.slice_type != HMO_SLICE_I) {                        /* encodeSkipFlag / encodePredMode are no-ops in I slices */
    if (cu->pred_mode[cuPart] == HMO_MODE_INTER) { encode_cu_syntax_inter(e, cu, cuPart, depth); return; }
    code_skip_flag(e, cu, cuPart); code_pred_mode(e, cu, cuPart);
  }
  code_part_size(e, cu, cuPart, depth);
  code_intra_dir_luma(e, cu, cuPart, 1);                       /* encodePredInfo */
  code_intra_dir_chroma(e, cu, cuPart);
  HmoTU root;
  memset(&root, 0, sizeof(root));
  root.log2 = 6 - depth; root.nparts = HMO_NPART >> (2 * depth);
  root.cw = root.cwo = (HMO_CTU >> depth) >> 1; root.c_code_all = 1;
  encode_transform(e, cu, cuPart, &root);
}

#include "hmo_inter.h"

/* ------------------------------------------------------------------------------------
 * xCheckRDCostIntra, TEncCu.cpp:2064-2157 ; xCheckBestMode :2213-2255
 * ---------------------------------------------------------------------------------- */
static int check_best_mode(HmoEnc *e, int d)
{
  if (e->temp[d]->cost < e->best[d]->cost) {
    HmoCU *t = e->best[d]; e->best[d] = e->temp[d]; e->temp[d] = t;
    HmoYuv *y = e->reco_best[d]; e->reco_best[d] = e->reco_temp[d]; e->reco_temp[d] = y;
    e->slot[d][CI_NEXT_BEST] = e->slot[d][CI_TEMP_BEST];
    return 1;
  }
  return 0;
}
static int check_rd_cost_intra(HmoEnc *e, int d, int partSize)
{
  HmoCU *cu = e->temp[d];
  const int n = cu->nparts, s = cu_size(cu);
  memset(cu->part_size, partSize, (size_t)n);
  memset(cu->pred_mode, HMO_MODE_INTRA, (size_t)n);
  if (e->trace) e->trace(e->trace_user, HMO_EV_INTRA_BEGIN, d, partSize);
  est_intra_pred_luma_qt(e, cu);
  e->last_luma_dist = cu->dist;
  for (int y = 0; y < s; y++) memcpy(e->rec[0] + (cu->y + y) * e->stride[0] + cu->x, e->reco_temp[d]->y + y * 64, (size_t)s);
  est_intra_pred_chroma_qt(e, cu);
  hmo_reset_bits(e);
  encode_cu_syntax(e, cu, 0, d);
  e->slot[d][CI_TEMP_BEST] = e->goon;
  cu->bits = hmo_bits(e);
  cu->bins = e->goon_bins;
  cu->cost = calc_rd_cost(e, cu->bits, cu->dist);
  if (e->trace) e->trace(e->trace_user, HMO_EV_INTRA_END, d, partSize);
  return check_best_mode(e, d);
}

static void copy_reco_to_pic(HmoEnc *e, const HmoYuv *r, int x, int y, int s)
{
  for (int c = 0; c < 3; c++) {
    int sh = c ? 1 : 0, bs = c ? 32 : 64, w = (c ? e->p.width / 2 : e->p.width), h = (c ? e->p.height / 2 : e->p.height);
    const uint8_t *src = c == 0 ? r->y : (c == 1 ? r->u : r->v);
    int px = x >> sh, py = y >> sh, n = s >> sh;
    for (int yy = 0; yy < n && py + yy < h; yy++) {
      int cw = n; if (px + cw > w) cw = w - px;
      if (cw > 0) memcpy(e->rec[c] + (py + yy) * e->stride[c] + px, src + yy * bs, (size_t)cw);
    }
  }
}

/* ------------------------------------------------------------------------------------
 * fork decision hooks of xCompressCU with the fork's default control (YSGlobalControl, tools_YS.cpp:4-58):
 * main model Naive on the N_OBF feature, no assistant / multi-model / depth exception unless asked for.
 * ---------------------------------------------------------------------------------- */
/* Num_OBF of a CU: 4x4 blocks of its area whose OBF count is positive (TEncCu.cpp:585-600) */
static int cu_num_obf(const HmoEnc *e, int x, int y, int s)
{
  int n = 0;
  for (int yy = 0; yy < s / 4; yy++)
    for (int xx = 0; xx < s / 4; xx++) n += e->obf[(y / 4 + yy) * e->obf_stride + x / 4 + xx] > 0;
  return n;
}
/* countTFPN + countRDLoss, tools_YS.cpp:968-986 (ResultType order TP,FP,TN,FN,FPLoss,FNLoss, globals_YS.h:81-89) */
static void count_verify(HmoEnc *e, int d, int skipPredicted, int partitionTrue, double j0, double j1)
{
  const double loss = fabs(j0 - j1);
  if (skipPredicted) { e->ver[d][partitionTrue ? 0 : 1] += 1.0; if (!partitionTrue) e->ver[d][4] += loss; }
  else { e->ver[d][partitionTrue ? 3 : 2] += 1.0; if (partitionTrue) e->ver[d][5] += loss; }
}

/* ------------------------------------------------------------------------------------
 * xCompressCU, TEncCu.cpp:460-1616.  Training: exhaustive; Verifying: exhaustive + TP/FP/TN/FN bookkeeping
 * (:1489-1497); Testing: Skip2Nx2N / TerminateCU pruning where the depth's switch is on (:951-996,1040,1143,1257,1446).
 * ---------------------------------------------------------------------------------- */
static void compress_cu(HmoEnc *e, int d, int parentPartSize)
{
  HmoCU *bestInit = e->best[d];
  const int x = bestInit->x, y = bestInit->y, zidx = bestInit->zidx, s = HMO_CTU >> d;
  const int boundary = !((x + s - 1 < e->p.width) && (y + s - 1 < e->p.height));
  int skip2Nx2N = 0, earlyTerminate = 0, predictSkip = 0, partitionTrue = 0;
  double j0 = HMO_MAX_DOUBLE, j1 = 0;
  if (!boundary && e->dec_state != HMO_TRAINING) {
    /* Naive model: label +1 (Skip2Nx2N) when the CU holds an outlier block, -1 (TerminateCU) when it holds none
     * (DoPrediction, tools_YS.cpp:686-695; TEncCu.cpp:670-678).  Testing predicts only where a switch is on (:660). */
    const int nobf = cu_num_obf(e, x, y, s);
    predictSkip = nobf > 0;
    if (e->dec_state == HMO_TESTING) {
      if (e->sw_term[d]) earlyTerminate = !predictSkip;                                  /* :967-970 */
      if (e->sw_skip[d]) skip2Nx2N = predictSkip;                                        /* :971-974 */
      if (d == 3 && nobf > 0 && e->depth_exception) skip2Nx2N = earlyTerminate = 0;      /* :991-995 */
    }
  }
  if (!boundary) {
    for (int yy = 0; yy < s; yy++) memcpy(e->org_yuv[d]->y + yy * 64, e->org[0] + (y + yy) * e->stride[0] + x, (size_t)s);
    for (int yy = 0; yy < s / 2; yy++) {
      memcpy(e->org_yuv[d]->u + yy * 32, e->org[1] + (y / 2 + yy) * e->stride[1] + x / 2, (size_t)(s / 2));
      memcpy(e->org_yuv[d]->v + yy * 32, e->org[2] + (y / 2 + yy) * e->stride[2] + x / 2, (size_t)(s / 2));
    }
    cu_init(e->temp[d], d, x, y, zidx);
    int tryIntra = 1;
    if (e->trace) e->trace(e->trace_user, HMO_EV_CU_BEGIN, d, parentPartSize);
    if (e->p.slice_type == HMO_SLICE_P) {                       /* inter candidates first (TEncCu.cpp:753-943; ESD / CFM / AMP off) */
      check_rd_cost_merge_2nx2n(e, d);                          /* :774 */
      cu_init(e->temp[d], d, x, y, zidx);
      check_rd_cost_inter(e, d, HMO_SIZE_2Nx2N, 0); cu_init(e->temp[d], d, x, y, zidx);      /* :780 */
      check_rd_cost_inter(e, d, HMO_SIZE_Nx2N, 0); cu_init(e->temp[d], d, x, y, zidx);       /* :826 (inter NxN never: 8x8 CUs are excluded, :816) */
      check_rd_cost_inter(e, d, HMO_SIZE_2NxN, 0); cu_init(e->temp[d], d, x, y, zidx);       /* :835 */
      if (e->p.amp && d < HMO_MAXDEPTH) {                       /* AMP with AMP_ENC_SPEEDUP + AMP_MRG (:843-943); deriveTestModeAMP (:393-450) */
        const HmoCU *bb = e->best[d];
        const int bps = bb->part_size[0], par = parentPartSize;
        int hor = 0, ver = 0, mhor = 0, mver = 0;
        if (bps == HMO_SIZE_2NxN) hor = 1;
        else if (bps == HMO_SIZE_Nx2N) ver = 1;
        else if (bps == HMO_SIZE_2Nx2N && !bb->merge_flag[0] && !bb->skip[0]) hor = ver = 1;
        if (par >= HMO_SIZE_2NxnU && par <= HMO_SIZE_nRx2N) mhor = mver = 1;
        if (par == HMO_SIZE_NONE) { if (bps == HMO_SIZE_2NxN) mhor = 1; else if (bps == HMO_SIZE_Nx2N) mver = 1; }
        if (bps == HMO_SIZE_2Nx2N && !bb->skip[0]) mhor = mver = 1;
        if (s == 64) hor = ver = 0;
        if (hor) { check_rd_cost_inter(e, d, HMO_SIZE_2NxnU, 0); cu_init(e->temp[d], d, x, y, zidx); check_rd_cost_inter(e, d, HMO_SIZE_2NxnD, 0); cu_init(e->temp[d], d, x, y, zidx); }
        else if (mhor) { check_rd_cost_inter(e, d, HMO_SIZE_2NxnU, 1); cu_init(e->temp[d], d, x, y, zidx); check_rd_cost_inter(e, d, HMO_SIZE_2NxnD, 1); cu_init(e->temp[d], d, x, y, zidx); }
        if (ver) { check_rd_cost_inter(e, d, HMO_SIZE_nLx2N, 0); cu_init(e->temp[d], d, x, y, zidx); check_rd_cost_inter(e, d, HMO_SIZE_nRx2N, 0); cu_init(e->temp[d], d, x, y, zidx); }
        else if (mver) { check_rd_cost_inter(e, d, HMO_SIZE_nLx2N, 1); cu_init(e->temp[d], d, x, y, zidx); check_rd_cost_inter(e, d, HMO_SIZE_nRx2N, 1); cu_init(e->temp[d], d, x, y, zidx); }
      }
      const HmoCU *b = e->best[d];                              /* intra only when the best inter candidate has a residual (:1033-1036) */
      tryIntra = b->cbf[0][0] != 0 || b->cbf[1][0] != 0 || b->cbf[2][0] != 0;
    }
    if (!skip2Nx2N && tryIntra) check_rd_cost_intra(e, d, HMO_SIZE_2Nx2N);  /* :1040; skipped => best cost stays MAX_DOUBLE (:1077) */
    j0 = e->best[d]->cost;                                      /* :1072,1078 */
    cu_init(e->temp[d], d, x, y, zidx);
    if (d == HMO_MAXDEPTH && !earlyTerminate && tryIntra) {     /* :1141-1143 */
      partitionTrue = check_rd_cost_intra(e, d, HMO_SIZE_NxN);
      j1 = partitionTrue ? e->best[d]->cost : e->temp[d]->cost; /* :1175-1183 */
      cu_init(e->temp[d], d, x, y, zidx);
    }
    if (e->trace) e->trace(e->trace_user, HMO_EV_CU_DONE, d, parentPartSize);
    if (e->best[d]->cost != HMO_MAX_DOUBLE) {               /* fork: TEncCu.cpp:1224 */
      hmo_reset_bits(e);
      code_split_flag(e, e->best[d], 0, d);
      e->best[d]->bits += hmo_bits(e);
      e->best[d]->bins += e->goon_bins;
      e->best[d]->cost = calc_rd_cost(e, e->best[d]->bits, e->best[d]->dist);
    }
    if (d < HMO_MAXDEPTH) j0 = e->best[d]->cost;                /* :1233-1235 */
  }
  cu_init(e->temp[d], d, x, y, zidx);
  if (d < HMO_MAXDEPTH && !earlyTerminate) {                    /* bSubBranch = false, :1257-1260 */
    const int nd = d + 1, hs = s >> 1, qn = (HMO_NPART >> (2 * nd));
    for (int i = 0; i < 4; i++) {
      int sx = x + (i & 1) * hs, sy = y + (i >> 1) * hs;
      cu_init(e->best[nd], nd, sx, sy, zidx + i * qn);
      cu_init(e->temp[nd], nd, sx, sy, zidx + i * qn);
      if (sx < e->p.width && sy < e->p.height) {
        if (i == 0) e->slot[nd][CI_CURR_BEST] = e->slot[d][CI_CURR_BEST];
        else e->slot[nd][CI_CURR_BEST] = e->slot[nd][CI_NEXT_BEST];
        compress_cu(e, nd, (boundary || e->best[d]->pred_mode[0] != HMO_MODE_INTER) ? HMO_SIZE_NONE : e->best[d]->part_size[0]);   /* eParentPartSize, :1341-1351 */
        cu_copy_part_from(e->temp[d], e->best[nd], i);
        /* xCopyYuv2Tmp */
        for (int yy = 0; yy < hs; yy++) memcpy(e->reco_temp[d]->y + ((i >> 1) * hs + yy) * 64 + (i & 1) * hs, e->reco_best[nd]->y + yy * 64, (size_t)hs);
        for (int yy = 0; yy < hs / 2; yy++) {
          memcpy(e->reco_temp[d]->u + ((i >> 1) * hs / 2 + yy) * 32 + (i & 1) * hs / 2, e->reco_best[nd]->u + yy * 32, (size_t)(hs / 2));
          memcpy(e->reco_temp[d]->v + ((i >> 1) * hs / 2 + yy) * 32 + (i & 1) * hs / 2, e->reco_best[nd]->v + yy * 32, (size_t)(hs / 2));
        }
      } else {
        cu_copy_to_pic(e, e->best[nd]);
        cu_copy_part_from(e->temp[d], e->best[nd], i);
      }
    }
    if (!boundary) {
      hmo_reset_bits(e);
      code_split_flag(e, e->temp[d], 0, d);
      e->temp[d]->bits += hmo_bits(e);
      e->temp[d]->bins += e->goon_bins;
    }
    e->temp[d]->cost = calc_rd_cost(e, e->temp[d]->bits, e->temp[d]->dist);
    e->slot[d][CI_TEMP_BEST] = e->slot[nd][CI_NEXT_BEST];
    if (skip2Nx2N) e->best[d]->cost = HMO_MAX_DOUBLE;           /* :1446-1449 */
    j0 = e->best[d]->cost; j1 = e->temp[d]->cost;               /* :1450-1451 */
    partitionTrue = check_best_mode(e, d);
  }
  if (!boundary && e->dec_state == HMO_VERIFYING) count_verify(e, d, predictSkip, partitionTrue, j0, j1);   /* :1489-1497 */
  cu_copy_to_pic(e, e->best[d]);
  copy_reco_to_pic(e, e->reco_best[d], x, y, s);
}

/* ------------------------------------------------------------------------------------
 * encodeCtu replay: TEncCu::xEncodeCU, TEncCu.cpp:1679-1778
 * ---------------------------------------------------------------------------------- */
static void encode_cu(HmoEnc *e, const HmoCU *ctu, int part, int depth, int lastCtuOfSlice)
{
  const int cx = ctu->x + part_x(part), cy = ctu->y + part_y(part), s = HMO_CTU >> depth;
  int boundary = 0;
  if (cx + s - 1 < e->p.width && cy + s - 1 < e->p.height) code_split_flag(e, ctu, part, depth);
  else boundary = 1;
  if ((depth < ctu->depth[part] && depth < HMO_MAXDEPTH) || boundary) {
    int qn = (HMO_NPART >> (2 * depth)) >> 2;
    for (int i = 0; i < 4; i++) {
      int p = part + i * qn;
      if (ctu->x + part_x(p) < e->p.width && ctu->y + part_y(p) < e->p.height) encode_cu(e, ctu, p, depth + 1, lastCtuOfSlice);
    }
    return;
  }
  encode_cu_syntax(e, ctu, part, depth);
  /* finishCU, TEncCu.cpp:1629-1645 ; isLastSubCUOfCtu TComDataCU.cpp */
  {
    int w = e->p.width, h = e->p.height;
    int granW = 8, granH = 8;                                   /* min CU */
    int rx = cx + s, ry = cy + s;
    int lastX = ((rx % HMO_CTU) == 0) || rx == w || rx > w;
    int lastY = ((ry % HMO_CTU) == 0) || ry == h || ry > h;
    (void)granW; (void)granH;
    if (lastX && lastY && !lastCtuOfSlice) hmo_enc_bin_trm(e, 0);
  }
}

/* ------------------------------------------------------------------------------------
 * public API
 * ---------------------------------------------------------------------------------- */
void hmo_params_default(HmoParams *p, int width, int height, int qp)
{
  memset(p, 0, sizeof(*p));
  p->width = width; p->height = height; p->qp = qp; p->slice_ctus = 0;
  p->transform_skip = 1; p->transform_skip_fast = 1; p->sign_hiding = 1; p->strong_smoothing = 1;
  p->slice_type = HMO_SLICE_I; p->search_range = 64; p->fast_search = 0; p->rdoq = 1; p->rdoq_ts = 1; p->fast_enc = 1; p->had_me = 1; p->fdm = 1; p->max_merge_cand = 5;
  hmo_params_finish(p);
}
/* TEncSlice::initEncSlice lambda (TEncSlice.cpp:686-706) + setUpLambda (:496-524) + TComRdCost::setLambda */
void hmo_params_finish(HmoParams *p)
{
  double qp_temp = (double)p->qp - 12;
  double lambda = 0.57 * pow(2.0, qp_temp / 3.0);
  if (p->lambda_override > 0.0) lambda = p->lambda_override;   /* non-I slices: QP factor of the GOP entry etc., computed by the caller */
  p->lambda = lambda;
  p->lambda_motion_sad = (unsigned)floor(65536.0 * sqrt(lambda));   /* TComRdCost::setLambda, TComRdCost.cpp:194-219 */
  p->lambda_motion_sse = (unsigned)floor(65536.0 * lambda);
  p->sqrt_lambda = sqrt(lambda);
  int qpc = p->qp < 0 ? p->qp : hmo_chroma_scale[p->qp > 57 ? 57 : p->qp];
  p->qp_c = qpc;
  double w = pow(2.0, (p->qp - qpc) / 3.0);
  p->chroma_weight = w;
  p->rdoq_lambda[0] = lambda; p->rdoq_lambda[1] = lambda / w; p->rdoq_lambda[2] = lambda / w;
}

HmoEnc *hmo_create(const HmoParams *p)
{
  hmo_init_tables();
  HmoEnc *e = (HmoEnc *)calloc(1, sizeof(HmoEnc));
  e->p = *p;
  e->w_ctu = (p->width + 63) / 64; e->h_ctu = (p->height + 63) / 64; e->n_ctu = e->w_ctu * e->h_ctu;
  e->pic = (HmoCtu *)calloc((size_t)e->n_ctu, sizeof(HmoCtu));
  e->replay_bits = (uint32_t *)calloc((size_t)e->n_ctu, sizeof(uint32_t));
  for (int d = 0; d < 4; d++) {
    e->best[d] = (HmoCU *)calloc(1, sizeof(HmoCU)); e->temp[d] = (HmoCU *)calloc(1, sizeof(HmoCU));
    e->org_yuv[d] = (HmoYuv *)calloc(1, sizeof(HmoYuv)); e->pred_temp[d] = (HmoYuv *)calloc(1, sizeof(HmoYuv));
    e->reco_best[d] = (HmoYuv *)calloc(1, sizeof(HmoYuv)); e->reco_temp[d] = (HmoYuv *)calloc(1, sizeof(HmoYuv));
  }
  return e;
}
void hmo_destroy(HmoEnc *e)
{
  if (!e) return;
  for (int d = 0; d < 4; d++) { free(e->best[d]); free(e->temp[d]); free(e->org_yuv[d]); free(e->pred_temp[d]); free(e->reco_best[d]); free(e->reco_temp[d]); }
  free(e->pic); free(e->replay_bits); free(e);
}
void hmo_set_planes(HmoEnc *e, const uint8_t *orgY, const uint8_t *orgU, const uint8_t *orgV, uint8_t *recY, uint8_t *recU, uint8_t *recV)
{
  e->org[0] = orgY; e->org[1] = orgU; e->org[2] = orgV;
  e->rec[0] = recY; e->rec[1] = recU; e->rec[2] = recV;
  e->stride[0] = e->p.width; e->stride[1] = e->stride[2] = e->p.width / 2;
}
/* reference picture of a P slice (list 0, index 0): planes of the picture size, same strides as the reconstruction */
void hmo_set_ref_planes(HmoEnc *e, const uint8_t *refY, const uint8_t *refU, const uint8_t *refV)
{ e->ref[0] = refY; e->ref[1] = refU; e->ref[2] = refV; e->refs[0][0] = refY; e->refs[0][1] = refU; e->refs[0][2] = refV; e->n_ref = 1; e->poc = 1; e->ref_poc[0] = 0; e->col_poc = 0; e->col_ref_poc[0] = -1; }
void hmo_set_ref_picture(HmoEnc *e, int idx, const uint8_t *refY, const uint8_t *refU, const uint8_t *refV, int poc)
{ if (idx < 0 || idx >= HMO_MAX_REF) return; e->refs[idx][0] = refY; e->refs[idx][1] = refU; e->refs[idx][2] = refV; e->ref_poc[idx] = poc; if (idx == 0) { e->ref[0] = refY; e->ref[1] = refU; e->ref[2] = refV; } }
void hmo_set_poc(HmoEnc *e, int poc, int n_ref) { e->poc = poc; e->n_ref = n_ref < 1 ? 1 : (n_ref > HMO_MAX_REF ? HMO_MAX_REF : n_ref); }
void hmo_set_col_pocs(HmoEnc *e, int col_poc, const int *col_ref_poc, int n) { e->col_poc = col_poc; for (int i = 0; i < HMO_MAX_REF; i++) e->col_ref_poc[i] = i < n ? col_ref_poc[i] : col_poc - 1; }
uint64_t hmo_test_n_sad(const HmoEnc *e) { return e->n_sad; }
int hmo_num_ctus(const HmoEnc *e) { return e->n_ctu; }
/* fork state of the frame (getCurrentState, tools_YS.cpp:1237-1242), the per-depth decision switches of the Naive model
 * (g_bDecisionSwitch[depth][Naive][Skip2Nx2N / TerminateCU]) and the frame's OBF count map ((height/4) x (width/4));
 * clears the verification counters. */
void hmo_set_decision(HmoEnc *e, int state, const uint8_t *sw_skip, const uint8_t *sw_term, int depth_exception, const int16_t *obf)
{
  e->dec_state = state; e->depth_exception = depth_exception; e->obf = obf; e->obf_stride = e->p.width / 4;
  for (int d = 0; d < 4; d++) { e->sw_skip[d] = sw_skip ? sw_skip[d] : 0; e->sw_term[d] = sw_term ? sw_term[d] : 0; }
  memset(e->ver, 0, sizeof(e->ver));
}
/* g_iVerResult[depth][Naive][TP,FP,TN,FN,FPLoss,FNLoss] accumulated over the CTUs compressed since hmo_set_decision */
void hmo_get_verify(const HmoEnc *e, double *out24) { memcpy(out24, e->ver, sizeof(e->ver)); }
/* SetDecisionSwitch, tools_YS.cpp:1123-1154 with getSkipPrecision / getTermPrecision (:1251-1266):
 * a depth's switch turns on when the precision measured on the Verifying frame exceeds its threshold
 * (0 => g_dPrecision_Th_Default = 0.8, tools_YS.cpp:46). */
void hmo_decision_switch(const double *ver24, const double *th_skip, const double *th_term, uint8_t *sw_skip, uint8_t *sw_term)
{
  for (int d = 0; d < 4; d++) {
    const double tp = ver24[d * 6 + 0], fp = ver24[d * 6 + 1], tn = ver24[d * 6 + 2], fn = ver24[d * 6 + 3];
    const double ths = (th_skip && th_skip[d] != 0) ? th_skip[d] : 0.8, tht = (th_term && th_term[d] != 0) ? th_term[d] : 0.8;
    const double ps = (tp + fp == 0) ? 0 : tp / (tp + fp), pt = (tn + fn == 0) ? 0 : tn / (tn + fn);
    sw_skip[d] = ps > ths; sw_term[d] = pt > tht;
  }
}
const HmoCtu *hmo_get_ctu(const HmoEnc *e, int a) { return &e->pic[a]; }
/* test hooks (see hmo_int.h: trace) */
void hmo_set_trace(HmoEnc *e, void (*fn)(void *, int, int, int), void *user) { e->trace = fn; e->trace_user = user; }
void hmo_set_col(HmoEnc *e, const HmoCtu *col) { e->col = col; }
void hmo_set_pu_trace(HmoEnc *e, HmoPuTrace *buf) { e->pu_trace = buf; }
void hmo_set_int_mv(HmoEnc *e, const int *xy) { for (int r = 0; r < HMO_MAX_REF; r++) { e->int_mv_2nx2n[r].x = xy[2 * r]; e->int_mv_2nx2n[r].y = xy[2 * r + 1]; } }
void hmo_test_int_mv(const HmoEnc *e, int *xy) { for (int r = 0; r < HMO_MAX_REF; r++) { xy[2 * r] = e->int_mv_2nx2n[r].x; xy[2 * r + 1] = e->int_mv_2nx2n[r].y; } }
const HmoCU *hmo_test_cu(const HmoEnc *e, int d, int best) { return best ? e->best[d] : e->temp[d]; }
const HmoYuv *hmo_test_reco(const HmoEnc *e, int d, int best) { return best ? e->reco_best[d] : e->reco_temp[d]; }
const HmoCabac *hmo_test_slot(const HmoEnc *e, int d, int ci) { return d < 0 ? &e->goon : &e->slot[d][ci]; }
int hmo_test_cur_ctu(const HmoEnc *e) { return e->cur_ctu; }
uint32_t hmo_test_last_luma_dist(const HmoEnc *e) { return e->last_luma_dist; }
const HmoCabac *hmo_get_cabac(const HmoEnc *e) { return &e->slot[0][CI_CURR_BEST]; }
uint32_t hmo_ctu_replay_bits(const HmoEnc *e, int a) { return e->replay_bits[a]; }

/* TComDataCU::initCtu defaults (TComDataCU.cpp:474-560) */
static void pic_ctu_init(const HmoEnc *e, HmoCtu *c)
{
  memset(c, 0, sizeof(*c));
  memset(c->part_size, HMO_SIZE_NONE, HMO_NPART);
  memset(c->pred_mode, HMO_MODE_NONE, HMO_NPART);
  memset(c->width, HMO_CTU, HMO_NPART); memset(c->height, HMO_CTU, HMO_NPART);
  memset(c->qp, e->p.qp, HMO_NPART);
  memset(c->intra_dir[0], HMO_DC, HMO_NPART);
  memset(c->mvp_idx, -1, HMO_NPART); memset(c->ref_idx, -1, HMO_NPART);
  c->total_cost = HMO_MAX_DOUBLE;
}

/* one iteration of the CTU loop of TEncSlice::compressSlice, TEncSlice.cpp:1380-1551 */
void hmo_compress_ctu(HmoEnc *e, int ctuRsAddr)
{
  const int sliceLen = e->p.slice_ctus > 0 ? e->p.slice_ctus : e->n_ctu;
  const int sliceStart = (ctuRsAddr / sliceLen) * sliceLen;
  int sliceEnd = sliceStart + sliceLen; if (sliceEnd > e->n_ctu) sliceEnd = e->n_ctu;
  e->cur_ctu = ctuRsAddr; e->slice_start = sliceStart;
  if (ctuRsAddr == sliceStart) hmo_cabac_init_tab(&e->slot[0][CI_CURR_BEST], e->p.qp, e->p.slice_type, e->p.cabac_b_table);     /* resetEntropy */
  if (ctuRsAddr == sliceStart && e->p.search_state_per_slice) memset(e->int_mv_2nx2n, 0, sizeof(e->int_mv_2nx2n));
  pic_ctu_init(e, &e->pic[ctuRsAddr]);
  e->goon = e->slot[0][CI_CURR_BEST];
  e->goon_bins = 0;
  /* compressCtu, TEncCu.cpp:329-356 */
  int x = (ctuRsAddr % e->w_ctu) * HMO_CTU, y = (ctuRsAddr / e->w_ctu) * HMO_CTU;
  cu_init(e->best[0], 0, x, y, 0);
  cu_init(e->temp[0], 0, x, y, 0);
  compress_cu(e, 0, HMO_SIZE_NONE);
  /* encodeCtu on [0][CI_CURR_BEST] (TEncSlice.cpp:1474-1487): replay the winner */
  {
    HmoCU *view = e->temp[0];
    const HmoCtu *p = &e->pic[ctuRsAddr];
    view->depth_cu = 0; view->x = x; view->y = y; view->zidx = 0; view->nparts = HMO_NPART;
    memcpy(view->depth, p->depth, HMO_NPART); memcpy(view->part_size, p->part_size, HMO_NPART);
    memcpy(view->pred_mode, p->pred_mode, HMO_NPART); memcpy(view->tr_idx, p->tr_idx, HMO_NPART);
    for (int c = 0; c < 3; c++) { memcpy(view->tskip[c], p->tskip[c], HMO_NPART); memcpy(view->cbf[c], p->cbf[c], HMO_NPART); }
    memcpy(view->intra_dir[0], p->intra_dir[0], HMO_NPART); memcpy(view->intra_dir[1], p->intra_dir[1], HMO_NPART);
    memcpy(view->skip, p->skip, HMO_NPART); memcpy(view->merge_flag, p->merge_flag, HMO_NPART); memcpy(view->merge_idx, p->merge_idx, HMO_NPART);
    memcpy(view->inter_dir, p->inter_dir, HMO_NPART); memcpy(view->mvp_idx, p->mvp_idx, HMO_NPART); memcpy(view->ref_idx, p->ref_idx, HMO_NPART);
    memcpy(view->mv, p->mv, sizeof(p->mv)); memcpy(view->mvd, p->mvd, sizeof(p->mvd));
    memcpy(view->coef[0], p->coeff_y, sizeof(p->coeff_y)); memcpy(view->coef[1], p->coeff_cb, sizeof(p->coeff_cb)); memcpy(view->coef[2], p->coeff_cr, sizeof(p->coeff_cr));
    e->goon = e->slot[0][CI_CURR_BEST];
    hmo_reset_bits(e);
    encode_cu(e, view, 0, 0, ctuRsAddr == sliceEnd - 1);
    e->replay_bits[ctuRsAddr] = hmo_bits(e);
    e->slot[0][CI_CURR_BEST] = e->goon;
  }
}
void hmo_compress_frame(HmoEnc *e)
{
  for (int a = 0; a < e->n_ctu; a++) hmo_compress_ctu(e, a);
}

/* ====================================================================================
 * leaf-level test entry points (compared with the reference's own leaf code through
 * tests/golden/*.npz, see oracle/ref/)
 * ================================================================================== */
void hmo_test_begin(HmoEnc *e, int cur_ctu)
{
  e->cur_ctu = cur_ctu; e->slice_start = 0;
  hmo_cabac_init(&e->goon, e->p.qp); e->goon_bins = 0;
}
/* field ids as oracle/ref/ref_driver.cpp:ref_set_ctu_field */
void hmo_test_set_ctu_field(HmoEnc *e, int ctu, int field, const uint8_t *v)
{
  HmoCtu *c = &e->pic[ctu];
  for (int i = 0; i < HMO_NPART; i++) {
    switch (field) {
      case 0: c->depth[i] = v[i]; break;
      case 1: c->part_size[i] = (int8_t)v[i]; break;
      case 2: c->pred_mode[i] = (int8_t)v[i]; break;
      case 3: c->intra_dir[0][i] = v[i]; break;
      case 4: c->intra_dir[1][i] = v[i]; break;
      case 5: c->tr_idx[i] = v[i]; break;
      default: break;
    }
  }
}
static void view_from_pic(const HmoEnc *e, int ctu, HmoCU *v)
{
  const HmoCtu *p = &e->pic[ctu];
  v->depth_cu = 0; v->x = (ctu % e->w_ctu) * HMO_CTU; v->y = (ctu / e->w_ctu) * HMO_CTU; v->zidx = 0; v->nparts = HMO_NPART;
  memcpy(v->depth, p->depth, HMO_NPART); memcpy(v->part_size, p->part_size, HMO_NPART); memcpy(v->pred_mode, p->pred_mode, HMO_NPART);
  memcpy(v->tr_idx, p->tr_idx, HMO_NPART); memcpy(v->intra_dir[0], p->intra_dir[0], HMO_NPART); memcpy(v->intra_dir[1], p->intra_dir[1], HMO_NPART);
}
int hmo_test_mpm(HmoEnc *e, int ctu, int part, int *preds)
{
  e->cur_ctu = ctu;
  view_from_pic(e, ctu, e->temp[0]);
  return intra_dir_predictor(e, e->temp[0], part, preds);
}
int hmo_test_ctx_split(HmoEnc *e, int ctu, int part, int depth)
{
  e->cur_ctu = ctu;
  HmoCU *cu = e->temp[0];
  view_from_pic(e, ctu, cu);
  int z = part, lx = cu->x + part_x(z), ly = cu->y + part_y(z), ctx = 0;
  if (left_ctu_ok(e, lx, ly)) ctx += nb_depth(e, cu, lx - 1, ly) > depth;
  if (above_ctu_ok(e, lx, ly)) ctx += nb_depth(e, cu, lx, ly - 1) > depth;
  return ctx;
}
/* prediction of the block at component position (px,py), size 1<<log2; returns 2 if the filtered
 * reference was used.  ref_unf/ref_filt: 4N+1 samples in walk order. */
int hmo_test_intra(HmoEnc *e, int comp, int px, int py, int log2, int mode, uint8_t *pred, int16_t *ref_unf, int16_t *ref_filt)
{
  const int N = 1 << log2;
  uint8_t ref[4 * 64 + 1], reff[4 * 64 + 1];
  { int sh = comp ? 1 : 0; e->cur_ctu = ((py << sh) >> 6) * e->w_ctu + ((px << sh) >> 6); }
  hmo_build_ref(e, comp, px, py, log2, 0, ref);
  int filt = hmo_use_filtered_ref(mode, log2, comp == 0);
  memset(reff, 0, sizeof(reff));
  if (filt) hmo_filter_ref(ref, reff, N, comp == 0 && e->p.strong_smoothing);
  hmo_intra_pred(filt ? reff : ref, 0, log2, mode, comp == 0, pred, N);
  for (int i = 0; i <= 4 * N; i++) { ref_unf[i] = ref[i]; ref_filt[i] = filt ? reff[i] : 0; }
  return filt ? 2 : 1;
}
/* forward transform (DST for 4x4 luma, or transform skip) + RDOQ (+ sign hiding) + dequant + inverse,
 * with rate tables taken from the running coder e->goon.  Returns uiAbsSum. */
int hmo_test_tq(HmoEnc *e, int comp, int log2, int lumaDir, int chromaDir, int trDepthRel, int tskip,
                const int16_t *resi, int32_t *coef, int16_t *resi_out)
{
  const int N = 1 << log2;
  HmoCU *cu = e->temp[0];
  memset(cu->pred_mode, HMO_MODE_INTRA, HMO_NPART);
  memset(cu->intra_dir[0], lumaDir, HMO_NPART); memset(cu->intra_dir[1], chromaDir, HMO_NPART);
  HmoTU tu; memset(&tu, 0, sizeof(tu)); tu.tr_depth = trDepthRel; tu.log2 = comp ? log2 + 1 : log2;
  int32_t tcoef[32 * 32];
  if (tskip) for (int i = 0; i < N * N; i++) tcoef[i] = (int32_t)resi[i] << (15 - 8 - log2);
  else hmo_fwd_transform(resi, N, tcoef, log2, comp == 0 && log2 == 2);
  int absSum = hmo_rdoq(e, cu, &tu, comp, tcoef, coef, log2, 0, tskip);
  if (absSum > 0) {
    int32_t dq[32 * 32];
    hmo_dequant(coef, dq, N * N, log2, comp ? e->p.qp_c : e->p.qp);
    if (tskip) { int s = 15 - 8 - log2; for (int i = 0; i < N * N; i++) resi_out[i] = (int16_t)((dq[i] + (1 << (s - 1))) >> s); }
    else hmo_inv_transform(dq, resi_out, N, log2, comp == 0 && log2 == 2);
  } else { memset(coef, 0, sizeof(int32_t) * (size_t)(N * N)); memset(resi_out, 0, sizeof(int16_t) * (size_t)(N * N)); }
  return absSum;
}
void hmo_test_code_coeff(HmoEnc *e, int comp, int log2, int lumaDir, int chromaDir, int tskip, const int32_t *coef)
{
  HmoCU *cu = e->temp[0];
  memset(cu->pred_mode, HMO_MODE_INTRA, HMO_NPART);
  memset(cu->intra_dir[0], lumaDir, HMO_NPART); memset(cu->intra_dir[1], chromaDir, HMO_NPART);
  memset(cu->tskip[comp], tskip, HMO_NPART);
  hmo_code_coeff_nxn(e, cu, coef, log2, comp, 0);
}
void hmo_test_cabac_get(const HmoEnc *e, uint8_t *ctx, uint64_t *frac) { memcpy(ctx, e->goon.ctx, HMO_NCTX_INTRA); *frac = e->goon.frac; }   /* the 160 contexts of the leaf fixtures */
void hmo_test_reset_bits(HmoEnc *e) { hmo_reset_bits(e); }
double hmo_test_rd_cost(const HmoEnc *e, uint32_t bits, uint32_t dist) { return calc_rd_cost(e, bits, dist); }
uint32_t hmo_test_chroma_dist(const HmoEnc *e, uint32_t sse) { return (uint32_t)(e->p.chroma_weight * (double)sse); }
