/*
 * hmo_deblock.c -- ORACLE (test infrastructure, never shipped / never on the product path).
 *
 * CPU restatement of HM's in-loop deblocking of an intra or P picture, TComLoopFilter::loopFilterPic
 * (Lib/TLibCommon/TComLoopFilter.cpp:130-155) with xDeblockCU (:170-236), the edge flags of xSetEdgefilterTU /
 * xSetEdgefilterPU / xSetLoopfilterParam (:270-405), the intra branch of xGetBoundaryStrengthSingle (:436-440),
 * xEdgeFilterLuma (:557-667), xEdgeFilterChroma (:670-791) and the sample filters (:805-946), as configured by
 * default (TAppEncCfg.cpp:813-818,848: filter on, beta / tc offsets from the caller, LFCrossSliceBoundaryFlag 1).
 *
 * Restated per picture position instead of per CU: with 4-sample partitions (g_uiMaxCUWidth >> g_uiMaxCUDepth = 4)
 * an edge is filtered where it lies on the 8-sample grid (PartIdxIncr = 2, :207-213) and the partition on its Q side
 * starts a transform unit or a prediction unit there; Bs = 2 next to an intra CU, else 1 across a transform edge with a
 * coded luma block on either side or across different motion (one reference list: P slices), else 0 (:405-553).
 * All vertical edges of the picture are filtered before the first horizontal one (:133-154).
 *
 * Parity: PINNED -- tests/golden/deblock_*.npz hold the output of the reference's own TComLoopFilter (built in place
 * into oracle/_ref/libhmleaf.so) for the oracle's decisions on several contents / QPs / offsets.
 */
#include <stdlib.h>
#include <string.h>
#include "hmo_int.h"

static const uint8_t k_tc[54] = {      /* sm_tcTable, TComLoopFilter.cpp:59-62 */
  0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,1,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,5,5,6,6,7,8,9,10,11,13,14,16,18,20,22,24 };
static const uint8_t k_beta[52] = {    /* sm_betaTable, :64-67 */
  0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,6,7,8,9,10,11,12,13,14,15,16,17,18,20,22,24,26,28,30,32,34,36,38,40,42,44,46,48,50,52,54,56,58,60,62,64 };
static const uint8_t k_chroma_scale[58] = {   /* g_aucChromaScale[CHROMA_420], TComRom.cpp */
  0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15,16,17,18,19,20,21,22,23,24,25,26,27,28,29,29,30,31,32,33,33,34,34,35,35,36,36,37,37,
  38,39,40,41,42,43,44,45,46,47,48,49,50,51 };

static inline int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
static inline int iabs(int v) { return v < 0 ? -v : v; }

typedef struct { const HmoCtu *pic; int w, h, w_ctu; uint8_t r2z[256]; } Dbk;

/* partition (x4, y4) of the picture in 4-sample units -> its CTU and z-order index */
static inline const HmoCtu *part_at(const Dbk *d, int x4, int y4, int *z)
{
  *z = d->r2z[(y4 & 15) * 16 + (x4 & 15)];
  return &d->pic[(y4 >> 4) * d->w_ctu + (x4 >> 4)];
}
/* m_aapbEdgeFilter for the partition on the Q side: its left (dir 0) / top (dir 1) border starts a transform unit and is
 * not the picture border (bLeftEdge / bTopEdge, :358-403; bInternalEdge :349) */
static int edge_flag(const Dbk *d, int dir, int x4, int y4)
{
  int z; const HmoCtu *c = part_at(d, x4, y4, &z);
  if (c->part_size[z] == HMO_SIZE_NONE) return 0;                     /* outside the picture, :172 */
  const int pos = (dir == 0 ? x4 : y4) * 4;
  if (pos == 0) return 0;
  const int cu = HMO_CTU >> c->depth[z], tu = cu >> c->tr_idx[z];
  if ((pos % tu) == 0) return 1;                                      /* transform-unit (or CU) edge: m_aapucBS starts at 1 */
  /* prediction-unit edge inside the CU (xSetEdgefilterPU, :322-356): not a transform edge, m_aapucBS starts at 0 */
  if (c->part_size[z] == HMO_SIZE_2NxN && dir == 1 && (pos % cu) == cu / 2) return 2;
  if (c->part_size[z] == HMO_SIZE_Nx2N && dir == 0 && (pos % cu) == cu / 2) return 2;
  if (c->part_size[z] == HMO_SIZE_NxN && (pos % cu) == cu / 2) return 2;
  /* asymmetric partitions: the edge at a quarter of the CU (:331-350); only 32x32 and 64x64 CUs put it on the 8-sample grid */
  if (c->part_size[z] == HMO_SIZE_2NxnU && dir == 1 && (pos % cu) == cu / 4) return 2;
  if (c->part_size[z] == HMO_SIZE_2NxnD && dir == 1 && (pos % cu) == cu - cu / 4) return 2;
  if (c->part_size[z] == HMO_SIZE_nLx2N && dir == 0 && (pos % cu) == cu / 4) return 2;
  if (c->part_size[z] == HMO_SIZE_nRx2N && dir == 0 && (pos % cu) == cu - cu / 4) return 2;
  return 0;
}
/* xGetBoundaryStrengthSingle, :405-553 (P slices with one reference picture list): P = left / above partition */
static int boundary_strength(const Dbk *d, int flag, int x4, int y4, int px4, int py4)
{
  int zq, zp; const HmoCtu *q = part_at(d, x4, y4, &zq), *p = part_at(d, px4, py4, &zp);
  if (p->pred_mode[zp] == HMO_MODE_INTRA || q->pred_mode[zq] == HMO_MODE_INTRA) return 2;
  if (flag == 1 && ((((q->cbf[0][zq] >> q->tr_idx[zq]) & 1) != 0) || (((p->cbf[0][zp] >> p->tr_idx[zp]) & 1) != 0))) return 1;
  const int rp = p->ref_idx[zp], rq = q->ref_idx[zq];
  const int mpx = rp < 0 ? 0 : p->mv[zp][0], mpy = rp < 0 ? 0 : p->mv[zp][1], mqx = rq < 0 ? 0 : q->mv[zq][0], mqy = rq < 0 ? 0 : q->mv[zq][1];
  return ((rp < 0) != (rq < 0) || (rp >= 0 && rp != rq) || iabs(mqx - mpx) >= 4 || iabs(mqy - mpy) >= 4) ? 1 : 0;
}
static int qp_at(const Dbk *d, int x4, int y4) { int z; const HmoCtu *c = part_at(d, x4, y4, &z); return c->qp[z]; }

/* xPelFilterLuma, :805-869 (no PCM / lossless parts) */
static void pel_filter_luma(uint8_t *p, int o, int tc, int sw, int thrCut, int filtP, int filtQ)
{
  const int m4 = p[0], m3 = p[-o], m5 = p[o], m2 = p[-2 * o], m6 = p[2 * o], m1 = p[-3 * o], m7 = p[3 * o], m0 = p[-4 * o];
  if (sw) {
    p[-o]     = (uint8_t)clip3(m3 - 2 * tc, m3 + 2 * tc, (m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3);
    p[0]      = (uint8_t)clip3(m4 - 2 * tc, m4 + 2 * tc, (m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3);
    p[-2 * o] = (uint8_t)clip3(m2 - 2 * tc, m2 + 2 * tc, (m1 + m2 + m3 + m4 + 2) >> 2);
    p[o]      = (uint8_t)clip3(m5 - 2 * tc, m5 + 2 * tc, (m3 + m4 + m5 + m6 + 2) >> 2);
    p[-3 * o] = (uint8_t)clip3(m1 - 2 * tc, m1 + 2 * tc, (2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3);
    p[2 * o]  = (uint8_t)clip3(m6 - 2 * tc, m6 + 2 * tc, (m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3);
  } else {
    int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
    if (iabs(delta) < thrCut) {
      delta = clip3(-tc, tc, delta);
      p[-o] = (uint8_t)clip3(0, 255, m3 + delta);
      p[0]  = (uint8_t)clip3(0, 255, m4 - delta);
      const int tc2 = tc >> 1;
      if (filtP) p[-2 * o] = (uint8_t)clip3(0, 255, m2 + clip3(-tc2, tc2, ((((m1 + m3 + 1) >> 1) - m2 + delta) >> 1)));
      if (filtQ) p[o]      = (uint8_t)clip3(0, 255, m5 + clip3(-tc2, tc2, ((((m6 + m4 + 1) >> 1) - m5 - delta) >> 1)));
    }
  }
}
/* xUseStrongFiltering, :921-931 */
static int use_strong(const uint8_t *p, int o, int d, int beta, int tc)
{
  const int m4 = p[0], m3 = p[-o], m7 = p[3 * o], m0 = p[-4 * o];
  return (iabs(m0 - m3) + iabs(m7 - m4) < (beta >> 3)) && (d < (beta >> 2)) && (iabs(m3 - m4) < ((tc * 5 + 1) >> 1));
}
static int calc_dp(const uint8_t *p, int o) { return iabs(p[-3 * o] - 2 * p[-2 * o] + p[-o]); }    /* :934-937 */
static int calc_dq(const uint8_t *p, int o) { return iabs(p[0] - 2 * p[o] + p[2 * o]); }           /* :939-942 */

/* one 4-sample segment of a luma edge, the loop body of xEdgeFilterLuma (:597-665).  p: first sample of the Q side on
 * the segment's first line, o: step across the edge, step: step along it */
static void luma_segment(uint8_t *p, int o, int step, int qp, int betaOff, int tcOff, int bs)
{
  const int tc = k_tc[clip3(0, 53, qp + 2 * (bs - 1) + tcOff * 2)];      /* DEFAULT_INTRA_TC_OFFSET = 2 */
  const int beta = k_beta[clip3(0, 51, qp + betaOff * 2)];
  const int side = (beta + (beta >> 1)) >> 3, thrCut = tc * 10;
  const int dp0 = calc_dp(p, o), dq0 = calc_dq(p, o), dp3 = calc_dp(p + 3 * step, o), dq3 = calc_dq(p + 3 * step, o);
  const int d0 = dp0 + dq0, d3 = dp3 + dq3, dp = dp0 + dp3, dq = dq0 + dq3, d = d0 + d3;
  if (d < beta) {
    const int filtP = dp < side, filtQ = dq < side;
    const int sw = use_strong(p, o, 2 * d0, beta, tc) && use_strong(p + 3 * step, o, 2 * d3, beta, tc);
    for (int i = 0; i < 4; i++) pel_filter_luma(p + i * step, o, tc, sw, thrCut, filtP, filtQ);
  }
}
/* xPelFilterChroma, :881-905 */
static void pel_filter_chroma(uint8_t *p, int o, int tc)
{
  const int m4 = p[0], m3 = p[-o], m5 = p[o], m2 = p[-2 * o];
  const int delta = clip3(-tc, tc, ((((m4 - m3) * 4) + m2 - m5 + 4) >> 3));
  p[-o] = (uint8_t)clip3(0, 255, m3 + delta);
  p[0]  = (uint8_t)clip3(0, 255, m4 - delta);
}

/* In-place deblocking of the picture `rec` (8-bit 4:2:0, strides = plane widths) described by `pic`. */
void hmo_deblock_pic(const HmoCtu *pic, int width, int height, uint8_t *recY, uint8_t *recU, uint8_t *recV,
                     int betaOffsetDiv2, int tcOffsetDiv2)
{
  Dbk d; d.pic = pic; d.w = width; d.h = height; d.w_ctu = (width + 63) / 64;
  const uint8_t *z2r = hmo_zscan_to_raster();
  for (int z = 0; z < 256; z++) d.r2z[z2r[z]] = (uint8_t)z;
  const int w4 = width / 4, h4 = height / 4, cw = width / 2;
  uint8_t *chroma[2] = { recU, recV };
  for (int dir = 0; dir < 2; dir++) {                                  /* EDGE_VER for the whole picture, then EDGE_HOR */
    for (int y4 = 0; y4 < h4; y4++)
      for (int x4 = 0; x4 < w4; x4++) {
        const int pos4 = dir == 0 ? x4 : y4;
        int flag;
        if ((pos4 & 1) || !(flag = edge_flag(&d, dir, x4, y4))) continue;       /* 8-sample grid, :207-213 */
        const int bs = boundary_strength(&d, flag, x4, y4, dir == 0 ? x4 - 1 : x4, dir == 0 ? y4 : y4 - 1);
        if (bs == 0) continue;
        const int qpQ = qp_at(&d, x4, y4), qpP = dir == 0 ? qp_at(&d, x4 - 1, y4) : qp_at(&d, x4, y4 - 1);
        const int qp = (qpP + qpQ + 1) >> 1;
        uint8_t *p = recY + (y4 * 4) * width + x4 * 4;
        if (dir == 0) luma_segment(p, 1, width, qp, betaOffsetDiv2, tcOffsetDiv2, bs);
        else luma_segment(p, width, 1, qp, betaOffsetDiv2, tcOffsetDiv2, bs);
        if ((pos4 & 3) == 0 && bs == 2) {                              /* chroma only across intra edges (:723) */                                         /* chroma: 8-sample chroma grid, :216-221,700-707 */
          int qc = qp;                                                 /* cb / cr QP offsets 0 */
          if (qc >= 58) qc -= 6; else if (qc >= 0) qc = k_chroma_scale[qc];       /* :747-761 */
          const int tc = k_tc[clip3(0, 53, qc + 2 * (2 - 1) + tcOffsetDiv2 * 2)];
          for (int c = 0; c < 2; c++) {
            uint8_t *q = chroma[c] + (y4 * 2) * cw + x4 * 2;
            for (int i = 0; i < 2; i++) {
              if (dir == 0) pel_filter_chroma(q + i * cw, 1, tc);
              else pel_filter_chroma(q + i, cw, tc);
            }
          }
        }
      }
  }
}
/* deblock the reconstruction of an encoder whose CTUs have all been decided */
void hmo_deblock(HmoEnc *e, int betaOffsetDiv2, int tcOffsetDiv2)
{
  hmo_deblock_pic(e->pic, e->p.width, e->p.height, e->rec[0], e->rec[1], e->rec[2], betaOffsetDiv2, tcOffsetDiv2);
}
