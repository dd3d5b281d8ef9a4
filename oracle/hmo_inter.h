/*
 * hmo_inter.h -- ORACLE (test infrastructure).  The P-slice half of the CU decision loop (BASELINE configs[4]), an
 * implementation include of hmo_search.c (it shares that file's static helpers).
 *
 * Restates, for the configuration written down in DESIGN.md 3e (list 0 only with one to four reference pictures, TMVP and AMP
 * optional, full integer search or TZ search, FEN, FDM, HadamardME, no weighted prediction, MaxNumMergeCand 5,
 * Log2ParMrgLevel 2, QuadtreeTUMaxDepthInter 3):
 *   TEncCu::xCheckRDCostMerge2Nx2N (TEncCu.cpp:1900-2018), xCheckRDCostInter (:2025-2062),
 *   TEncSearch::predInterSearch (TEncSearch.cpp:3008-3506), xEstimateMvPredAMVP (:3513-3575), xGetTemplateCost (:3715-3757),
 *   xMotionEstimation (:3764-3860), xSetSearchRange (:3865-3882), xPatternSearch (:3886-3942), xPatternSearchFracDIF
 *   (:4340-4375) + xPatternRefinement (:810-868), xCheckBestMVP (:3660-3712), xMergeEstimation (:2928-2982),
 *   xGetInterPredictionError (:2905-2925), encodeResAndCalcRdInterCU (:4380-4522), xEstimateInterResidualQT (:4525-5162),
 *   xEncodeInterResidualQT (:5166-5241), xSetInterResidualQTData (:5246-5306), xAddSymbolBitsInter (:5382-5421);
 *   TComDataCU::getInterMergeCandidates (TComDataCU.cpp:2340-2688), fillMvpCand (:2781-2913), clipMv (:2930-2942);
 *   TComPrediction::motionCompensation / xPredInterBlk (TComPrediction.cpp:518-705) with TComInterpolationFilter;
 *   TComRdCost SAD / motion-vector cost (TComRdCost.h:163-189, TComRdCost.cpp:278-292,465-964);
 *   inter syntax of TEncSbac (codeSkipFlag :540, codeMergeFlag :560, codeMergeIndex :581, codePredMode :524,
 *   codePartSize :436, codeMvd :780, codeMVPIdx, codeQtRootCbf) and TEncEntropy::encodePUWise / encodeCoeff (:456,:616).
 * Pinned by tests/golden/inter_*.npz (the reference's own TEncSearch.cpp / TComDataCU.cpp / TComPrediction.cpp).
 */

typedef struct { int x, y; } HmoMv;
#define HMO_MAX_UINT 0xffffffffu

/* ------------------------------------------------------------------------------------
 * PU geometry: TComDataCU::getPartIndexAndSize / getPartPosition (TComDataCU.cpp:2165-2240,2706-2772), the partition
 * runs TComCUMvField::setAll / TComDataCU::setSubPart write (TComMotionInfo.cpp:181-327)
 * ---------------------------------------------------------------------------------- */
static int pu_count(int partSize) { return partSize == HMO_SIZE_2Nx2N ? 1 : (partSize == HMO_SIZE_NxN ? 4 : 2); }
static void pu_geom(const HmoCU *cu, int partSize, int pu, int *addr, int *ox, int *oy, int *w, int *h)
{
  const int s = cu_size(cu), n = cu->nparts, q = s >> 2;
  *addr = 0; *ox = 0; *oy = 0; *w = s; *h = s;
  switch (partSize) {
  case HMO_SIZE_2NxN: *h = s >> 1; if (pu) { *addr = n >> 1; *oy = s >> 1; } break;
  case HMO_SIZE_Nx2N: *w = s >> 1; if (pu) { *addr = n >> 2; *ox = s >> 1; } break;
  case HMO_SIZE_NxN: *w = *h = s >> 1; *addr = pu * (n >> 2); *ox = (pu & 1) * (s >> 1); *oy = (pu >> 1) * (s >> 1); break;
  case HMO_SIZE_2NxnU: if (!pu) *h = q; else { *h = s - q; *oy = q; *addr = n >> 3; } break;                 /* getPartIndexAndSize, TComDataCU.cpp:2165-2240 */
  case HMO_SIZE_2NxnD: if (!pu) *h = s - q; else { *h = q; *oy = s - q; *addr = (n >> 1) + (n >> 3); } break;
  case HMO_SIZE_nLx2N: if (!pu) *w = q; else { *w = s - q; *ox = q; *addr = n >> 4; } break;
  case HMO_SIZE_nRx2N: if (!pu) *w = s - q; else { *w = q; *ox = s - q; *addr = (n >> 2) + (n >> 4); } break;
  default: break;
  }
}
/* the 4x4 partitions of a PU (what TComCUMvField::setAll / TComDataCU::setSubPart address for every partition shape,
 * TComMotionInfo.cpp:128-303): all partitions of the CU whose position lies in the PU's rectangle */
static int pu_has(const HmoCU *cu, int ox, int oy, int w, int h, int i)
{ const int z = cu->zidx + i, lx = part_x(z) - (cu->x & 63), ly = part_y(z) - (cu->y & 63); return lx >= ox && lx < ox + w && ly >= oy && ly < oy + h; }
#define PU_FOR(cu, ps, pu, i) for (int a_, ox_, oy_, w_, h_, i = (pu_geom(cu, ps, pu, &a_, &ox_, &oy_, &w_, &h_), 0); i < (cu)->nparts; i++) if (pu_has(cu, ox_, oy_, w_, h_, i))
static void pu_set_motion(HmoCU *cu, int ps, int pu, HmoMv mv, int ref)
{ PU_FOR(cu, ps, pu, i) { cu->mv[i][0] = (int16_t)mv.x; cu->mv[i][1] = (int16_t)mv.y; cu->ref_idx[i] = (int8_t)ref; } }
static void pu_set_mvd(HmoCU *cu, int ps, int pu, HmoMv d) { PU_FOR(cu, ps, pu, i) { cu->mvd[i][0] = (int16_t)d.x; cu->mvd[i][1] = (int16_t)d.y; } }
static void pu_set_merge(HmoCU *cu, int ps, int pu, int flag, int idx) { PU_FOR(cu, ps, pu, i) { cu->merge_flag[i] = (uint8_t)flag; cu->merge_idx[i] = (uint8_t)idx; } }
static void pu_set_dir(HmoCU *cu, int ps, int pu, int dir) { PU_FOR(cu, ps, pu, i) cu->inter_dir[i] = (uint8_t)dir; }
static void pu_set_mvp(HmoCU *cu, int ps, int pu, int idx) { PU_FOR(cu, ps, pu, i) cu->mvp_idx[i] = (int8_t)idx; }

/* ------------------------------------------------------------------------------------
 * neighbour motion: getPULeft / Above / AboveRight / BelowLeft / AboveLeft (TComDataCU.cpp:1071-1390) reduce to
 * "inside the picture, in this slice, earlier in z-scan order than the corner partition the lookup starts from";
 * data comes from the working CU when the neighbour lies inside it (`return this`), else from the committed picture.
 * ---------------------------------------------------------------------------------- */
typedef struct { int avail, inter, skip; HmoMv mv; int ref; } HmoNb;
static HmoNb nb_motion(const HmoEnc *e, const HmoCU *cu, int nx, int ny, int cx, int cy)
{
  HmoNb r; memset(&r, 0, sizeof(r)); r.ref = -1;
  if (nx < 0 || ny < 0 || nx >= e->p.width || ny >= e->p.height) return r;
  const int ctuN = ctu_of(e, nx, ny), ctuC = ctu_of(e, cx, cy);
  if (ctuN < e->slice_start || ctuN > ctuC) return r;
  if (ctuN == ctuC && !(zidx_of(nx, ny) < zidx_of(cx, cy))) return r;
  r.avail = 1;
  if (inside_cu(cu, nx, ny)) {
    const int p = zidx_of(nx, ny) - cu->zidx;
    r.inter = cu->pred_mode[p] == HMO_MODE_INTER; r.skip = cu->skip[p]; r.mv.x = cu->mv[p][0]; r.mv.y = cu->mv[p][1]; r.ref = cu->ref_idx[p];
  } else {
    const HmoCtu *c = &e->pic[ctuN]; const int p = zidx_of(nx, ny);
    r.inter = c->pred_mode[p] == HMO_MODE_INTER; r.skip = c->skip[p]; r.mv.x = c->mv[p][0]; r.mv.y = c->mv[p][1]; r.ref = c->ref_idx[p];
  }
  return r;
}
static int same_motion(const HmoNb *a, const HmoNb *b) { return a->mv.x == b->mv.x && a->mv.y == b->mv.y && a->ref == b->ref; }   /* hasEqualMotion, list 0 */

/* getInterMergeCandidates for a P slice without TMVP: spatial A1, B1, B0, A0, B2, then zero candidates (refIdx 0) */
/* ---- temporal motion vector prediction (TMVP), collocated picture = list 0 index 0 (collocated_from_l0,
 * collocated_ref_idx 0).  xGetColMVP, TComDataCU.cpp:3175-3242: the motion the collocated picture holds at a position after
 * TComPic::compressMotion (the top-left 4x4 partition of every 16x16 block stands for the block); not available where that
 * partition is intra.  Scaled by the two POC distances when they differ (iScale != 4096). */
/* xGetDistScaleFactor, TComDataCU.cpp:3312-3329, and TComMv::scaleMv, TComMv.h:145-150 */
static int dist_scale(int curPoc, int curRefPoc, int colPoc, int colRefPoc)
{
  const int dD = colPoc - colRefPoc, dB = curPoc - curRefPoc;
  if (dD == dB) return 4096;
  const int tdb = dB < -128 ? -128 : (dB > 127 ? 127 : dB), tdd = dD < -128 ? -128 : (dD > 127 ? 127 : dD);
  const int x = (0x4000 + abs(tdd / 2)) / tdd;
  const int sc = (tdb * x + 32) >> 6;
  return sc < -4096 ? -4096 : (sc > 4095 ? 4095 : sc);
}
static HmoMv scale_mv(HmoMv m, int sc)
{
  if (sc == 4096) return m;
  int vx = (sc * m.x + 127 + (sc * m.x < 0)) >> 8, vy = (sc * m.y + 127 + (sc * m.y < 0)) >> 8;
  HmoMv r; r.x = vx < -32768 ? -32768 : (vx > 32767 ? 32767 : vx); r.y = vy < -32768 ? -32768 : (vy > 32767 ? 32767 : vy);
  return r;
}
/* the collocated vector for a PU that references RefPicList0[refIdx]: scaled by the ratio of the two POC distances */
static int col_mvp(const HmoEnc *e, int x, int y, int refIdx, HmoMv *mv)
{
  if (!e->col) return 0;
  const int xc = x & ~15, yc = y & ~15;
  const HmoCtu *c = &e->col[(yc >> 6) * e->w_ctu + (xc >> 6)];
  const int z = zidx_of(xc & 63, yc & 63);
  if (c->pred_mode[z] != HMO_MODE_INTER || c->ref_idx[z] < 0) return 0;
  HmoMv m; m.x = c->mv[z][0]; m.y = c->mv[z][1];
  const int cr = c->ref_idx[z] < HMO_MAX_REF ? c->ref_idx[z] : 0;
  *mv = scale_mv(m, dist_scale(e->poc, e->ref_poc[refIdx], e->col_poc, e->col_ref_poc[cr]));
  return 1;
}
/* the temporal candidate of a PU: bottom-right neighbour (H) when it lies inside the picture and in the same CTU row
 * (TComDataCU.cpp:2528-2563 / :2863-2900), else the centre of the PU (xDeriveCenterIdx) */
static int temporal_candidate(const HmoEnc *e, int xP, int yP, int w, int h, int refIdx, HmoMv *mv)
{
  if (!e->p.tmvp) return 0;
  const int bx = xP + w, by = yP + h;
  if (bx < e->p.width && by < e->p.height && (by & 63) != 0 && col_mvp(e, bx, by, refIdx, mv)) return 1;
  return col_mvp(e, xP + ((w >> 3) << 2), yP + ((h >> 3) << 2), refIdx, mv);
}

static int merge_candidates(const HmoEnc *e, const HmoCU *cu, int partSize, int pu, HmoMv mv[5], int ref[5])
{
  int addr, ox, oy, w, h; pu_geom(cu, partSize, pu, &addr, &ox, &oy, &w, &h);
  const int xP = cu->x + ox, yP = cu->y + oy, maxc = e->p.max_merge_cand;
  const int lbx = xP, lby = yP + h - 1, rtx = xP + w - 1, rty = yP;           /* corner partitions LB / RT; LT = (xP, yP) */
  int n = 0;
  HmoNb a1 = nb_motion(e, cu, xP - 1, yP + h - 1, lbx, lby);
  const int okA1 = a1.avail && !(pu == 1 && (partSize == HMO_SIZE_Nx2N || partSize == HMO_SIZE_nLx2N || partSize == HMO_SIZE_nRx2N)) && a1.inter;
  if (okA1) { mv[n] = a1.mv; ref[n] = a1.ref; n++; }
  if (n == maxc) return n;
  HmoNb b1 = nb_motion(e, cu, xP + w - 1, yP - 1, rtx, rty);
  const int okB1 = b1.avail && !(pu == 1 && (partSize == HMO_SIZE_2NxN || partSize == HMO_SIZE_2NxnU || partSize == HMO_SIZE_2NxnD)) && b1.inter;
  if (okB1 && (!okA1 || !same_motion(&a1, &b1))) { mv[n] = b1.mv; ref[n] = b1.ref; n++; }
  if (n == maxc) return n;
  HmoNb b0 = nb_motion(e, cu, xP + w, yP - 1, rtx, rty);
  const int okB0 = b0.avail && b0.inter;
  if (okB0 && (!okB1 || !same_motion(&b1, &b0))) { mv[n] = b0.mv; ref[n] = b0.ref; n++; }
  if (n == maxc) return n;
  HmoNb a0 = nb_motion(e, cu, xP - 1, yP + h, lbx, lby);
  const int okA0 = a0.avail && a0.inter;
  if (okA0 && (!okA1 || !same_motion(&a1, &a0))) { mv[n] = a0.mv; ref[n] = a0.ref; n++; }
  if (n == maxc) return n;
  if (n < 4) {
    HmoNb b2 = nb_motion(e, cu, xP - 1, yP - 1, xP, yP);
    const int okB2 = b2.avail && b2.inter;
    if (okB2 && (!okA1 || !same_motion(&a1, &b2)) && (!okB1 || !same_motion(&b1, &b2))) { mv[n] = b2.mv; ref[n] = b2.ref; n++; }
  }
  if (n == maxc) return n;
  { HmoMv t; if (temporal_candidate(e, xP, yP, w, h, 0, &t)) { mv[n] = t; ref[n] = 0; n++; } }   /* the temporal merge candidate references index 0 (:2566) */
  if (n == maxc) return n;
  for (int r = 0, refcnt = 0; n < maxc; n++) {                  /* zero candidates walk the reference indices (:2672-2686) */
    mv[n].x = mv[n].y = 0; ref[n] = r;
    if (refcnt == e->n_ref - 1) r = 0; else { ++r; ++refcnt; }
  }
  return n;
}

/* fillMvpCand (TComDataCU.cpp:2781-2925) for RefPicList0[refIdx]: the left pair (A0, A1) and the above triple (B0, B1, B2),
 * each first for a neighbour that references the same picture (xAddMVPCand, :3005-3074), then -- left: if that found nothing;
 * above: only if no left neighbour was inter -- for any inter neighbour with its vector scaled by the ratio of the POC
 * distances (xAddMVPCandOrder, :3084-3215); equal pair pruned, temporal candidate appended, cut / padded to two. */
static int mvp_same(const HmoEnc *e, const HmoNb *nb, int refIdx, HmoMv *cand, int *n)
{ if (nb->avail && nb->ref >= 0 && e->ref_poc[nb->ref] == e->ref_poc[refIdx]) { cand[(*n)++] = nb->mv; return 1; } return 0; }
static int mvp_scaled(const HmoEnc *e, const HmoNb *nb, int refIdx, HmoMv *cand, int *n)
{ if (nb->avail && nb->ref >= 0) { cand[(*n)++] = scale_mv(nb->mv, dist_scale(e->poc, e->ref_poc[refIdx], e->poc, e->ref_poc[nb->ref])); return 1; } return 0; }
static void amvp_candidates(const HmoEnc *e, const HmoCU *cu, int partSize, int pu, int refIdx, HmoMv cand[2])
{
  int addr, ox, oy, w, h; pu_geom(cu, partSize, pu, &addr, &ox, &oy, &w, &h);
  const int xP = cu->x + ox, yP = cu->y + oy;
  const int lbx = xP, lby = yP + h - 1, rtx = xP + w - 1, rty = yP;
  int n = 0; HmoMv c3[4];
  HmoNb a0 = nb_motion(e, cu, xP - 1, yP + h, lbx, lby), a1 = nb_motion(e, cu, xP - 1, yP + h - 1, lbx, lby);
  const int addedSmvp = (a0.avail && a0.inter) || (a1.avail && a1.inter);
  int added = mvp_same(e, &a0, refIdx, c3, &n);
  if (!added) added = mvp_same(e, &a1, refIdx, c3, &n);
  if (!added) { added = mvp_scaled(e, &a0, refIdx, c3, &n); if (!added) mvp_scaled(e, &a1, refIdx, c3, &n); }
  HmoNb b0 = nb_motion(e, cu, xP + w, yP - 1, rtx, rty), b1 = nb_motion(e, cu, xP + w - 1, yP - 1, rtx, rty), b2 = nb_motion(e, cu, xP - 1, yP - 1, xP, yP);
  added = mvp_same(e, &b0, refIdx, c3, &n);
  if (!added) added = mvp_same(e, &b1, refIdx, c3, &n);
  if (!added) mvp_same(e, &b2, refIdx, c3, &n);
  if (!addedSmvp) {
    added = mvp_scaled(e, &b0, refIdx, c3, &n);
    if (!added) added = mvp_scaled(e, &b1, refIdx, c3, &n);
    if (!added) mvp_scaled(e, &b2, refIdx, c3, &n);
  }
  if (n == 2 && c3[0].x == c3[1].x && c3[0].y == c3[1].y) n = 1;
  { HmoMv t; if (n < 3 && temporal_candidate(e, xP, yP, w, h, refIdx, &t)) c3[n++] = t; }
  if (n > 2) n = 2;
  while (n < 2) { c3[n].x = c3[n].y = 0; n++; }
  cand[0] = c3[0]; cand[1] = c3[1];
}

/* clipMv, TComDataCU.cpp:2930-2942 (offsets relative to the CU's position) */
static HmoMv clip_mv(const HmoEnc *e, const HmoCU *cu, HmoMv mv)
{
  const int hmax = (e->p.width + 8 - cu->x - 1) << 2, hmin = (-64 - 8 - cu->x + 1) << 2;
  const int vmax = (e->p.height + 8 - cu->y - 1) << 2, vmin = (-64 - 8 - cu->y + 1) << 2;
  mv.x = mv.x > hmax ? hmax : (mv.x < hmin ? hmin : mv.x);
  mv.y = mv.y > vmax ? vmax : (mv.y < vmin ? vmin : mv.y);
  return mv;
}

/* ------------------------------------------------------------------------------------
 * interpolation: TComInterpolationFilter (luma 8 taps at quarters, chroma 4 taps at eighths) as xPredInterBlk applies it.
 * All three paths of xPredInterBlk (copy / one pass / two passes through a 14-bit intermediate) equal
 * clip8((sum_v sum_h c_v c_h s + 2048) >> 12) with the {64} filter for a zero fraction; reference samples beyond the
 * picture are the replicated border (TComPicYuv::extendPicBorder).
 * ---------------------------------------------------------------------------------- */
static const int8_t k_luma_filter[4][8] = { { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 }, { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
static const int8_t k_chroma_filter[8][4] = { { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 }, { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };
static inline int ref_sample(const HmoEnc *e, int r, int comp, int x, int y)
{
  const int w = comp ? e->p.width >> 1 : e->p.width, h = comp ? e->p.height >> 1 : e->p.height;
  x = x < 0 ? 0 : (x >= w ? w - 1 : x); y = y < 0 ? 0 : (y >= h ? h - 1 : y);
  return e->refs[r][comp][y * e->stride[comp] + x];
}
/* block of size w x h (component samples) whose top-left is at (bx, by) displaced by mv (luma: quarter, chroma: eighth units) */
static void mc_block(const HmoEnc *e, int r, int comp, int bx, int by, int w, int h, int mvx, int mvy, uint8_t *dst, int ds)
{
  const int sh = comp ? 3 : 2, taps = comp ? 4 : 8, half = taps / 2 - 1;
  const int ix = bx + (mvx >> sh), iy = by + (mvy >> sh), fx = mvx & ((1 << sh) - 1), fy = mvy & ((1 << sh) - 1);
  const int8_t *ch = comp ? k_chroma_filter[fx] : k_luma_filter[fx], *cv = comp ? k_chroma_filter[fy] : k_luma_filter[fy];
  int tmp[(64 + 7) * 64];
  for (int y = 0; y < h + taps - 1; y++)
    for (int x = 0; x < w; x++) {
      int s = 0;
      for (int t = 0; t < taps; t++) s += ch[t] * ref_sample(e, r, comp, ix + x + t - half, iy + y - half);
      tmp[y * w + x] = s;
    }
  for (int y = 0; y < h; y++)
    for (int x = 0; x < w; x++) {
      int s = 0;
      for (int t = 0; t < taps; t++) s += cv[t] * tmp[(y + t) * w + x];
      s = (s + 2048) >> 12;
      dst[y * ds + x] = (uint8_t)(s < 0 ? 0 : (s > 255 ? 255 : s));
    }
}
/* motionCompensation of one PU (all three components) into a CU-sized buffer; the vector is clipped first (xPredInterUni) */
static void mc_pu(const HmoEnc *e, const HmoCU *cu, int partSize, int pu, HmoYuv *dst)
{
  int addr, ox, oy, w, h; pu_geom(cu, partSize, pu, &addr, &ox, &oy, &w, &h);
  HmoMv mv; mv.x = cu->mv[addr][0]; mv.y = cu->mv[addr][1];
  const int r = cu->ref_idx[addr] > 0 ? cu->ref_idx[addr] : 0;
  mv = clip_mv(e, cu, mv);
  mc_block(e, r, 0, cu->x + ox, cu->y + oy, w, h, mv.x, mv.y, dst->y + oy * 64 + ox, 64);
  mc_block(e, r, 1, (cu->x + ox) >> 1, (cu->y + oy) >> 1, w >> 1, h >> 1, mv.x, mv.y, dst->u + (oy >> 1) * 32 + (ox >> 1), 32);
  mc_block(e, r, 2, (cu->x + ox) >> 1, (cu->y + oy) >> 1, w >> 1, h >> 1, mv.x, mv.y, dst->v + (oy >> 1) * 32 + (ox >> 1), 32);
}

/* ------------------------------------------------------------------------------------
 * motion-vector cost (TComRdCost.h:163-189, TComRdCost.cpp:278-292): unsigned 32-bit arithmetic as in the reference
 * ---------------------------------------------------------------------------------- */
static uint32_t mv_comp_bits(int v)
{
  uint32_t len = 1, t = (v <= 0) ? (((uint32_t)(-v)) << 1) + 1 : ((uint32_t)v << 1);
  while (t != 1) { t >>= 1; len += 2; }
  return len;
}
static uint32_t mv_bits(int x, int y, HmoMv pred, int scale) { return mv_comp_bits((x << scale) - pred.x) + mv_comp_bits((y << scale) - pred.y); }
static uint32_t motion_cost(const HmoEnc *e, uint32_t bits) { return (e->p.lambda_motion_sad * bits) >> 16; }       /* getCost(b) with m_uiCost = m_uiLambdaMotionSAD */

/* xGetSAD* with iSubShift: rows 0, step, 2*step, ...; sum << shift */
static uint32_t sad_ref(const HmoEnc *e, int ri, const uint8_t *org, int so, int rx, int ry, int w, int h, int subShift)
{
  uint32_t s = 0; const int step = 1 << subShift;
  if (rx >= 0 && ry >= 0 && rx + w <= e->p.width && ry + h <= e->p.height) {
    for (int y = 0; y < h; y += step) { const uint8_t *r = e->refs[ri][0] + (ry + y) * e->stride[0] + rx, *o = org + y * so; for (int x = 0; x < w; x++) s += (uint32_t)abs(o[x] - r[x]); }
  } else {
    for (int y = 0; y < h; y += step) for (int x = 0; x < w; x++) s += (uint32_t)abs(org[y * so + x] - ref_sample(e, ri, 0, rx + x, ry + y));
  }
  return s << subShift;
}
static uint32_t sad_blocks(const uint8_t *a, int sa, const uint8_t *b, int sb, int w, int h)
{ uint32_t s = 0; for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) s += (uint32_t)abs(a[y * sa + x] - b[y * sb + x]); return s; }

/* ------------------------------------------------------------------------------------
 * xEstimateMvPredAMVP + xGetTemplateCost
 * ---------------------------------------------------------------------------------- */
static HmoMv estimate_mvp(HmoEnc *e, HmoCU *cu, int partSize, int pu, int refIdx, HmoMv cand[2], int *bestIdx)
{
  int addr, ox, oy, w, h; pu_geom(cu, partSize, pu, &addr, &ox, &oy, &w, &h);
  amvp_candidates(e, cu, partSize, pu, refIdx, cand);
  const uint8_t *org = e->org_yuv[cu->depth_cu]->y + oy * 64 + ox;
  uint32_t best = HMO_MAX_UINT; int bi = 0;
  for (int i = 0; i < 2; i++) {                                 /* pInfo->iN is always 2 after the zero padding */
    HmoMv c = clip_mv(e, cu, cand[i]);
    uint8_t blk[64 * 64];
    mc_block(e, refIdx, 0, cu->x + ox, cu->y + oy, w, h, c.x, c.y, blk, 64);
    const uint32_t sad = sad_blocks(blk, 64, org, 64, w, h);
    /* calcRdCost(bits = m_auiMVPIdxCost[i][2] = 1, dist, false, DF_SAD), TComRdCost.cpp:100-103 */
    const uint32_t cost = (uint32_t)floor((double)sad + (floor(((double)1 * (double)e->p.lambda_motion_sad) + 0.5) / 65536.0));
    if (best > cost) { best = cost; bi = i; }
  }
  *bestIdx = bi;
  return cand[bi];
}

/* ------------------------------------------------------------------------------------
 * xMotionEstimation: full integer search (xPatternSearch) + half / quarter refinement (xPatternSearchFracDIF)
 * ---------------------------------------------------------------------------------- */
static const int8_t k_refine_h[9][2] = { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 }, { -1, -1 }, { 1, -1 }, { -1, 1 }, { 1, 1 } };
static const int8_t k_refine_q[9][2] = { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, -1 }, { 1, -1 }, { -1, 0 }, { 1, 0 }, { -1, 1 }, { 1, 1 } };

/* ---- xTZSearch (FastSearch 1), TEncSearch.cpp:3981-4180, with TZ_SEARCH_CONFIGURATION (:301-317): zero-vector test on,
 * other predictors off, diamond first search that stops three rounds after the last improvement (FASTME_SMOOTHER_MV),
 * raster search with step 5 when the best distance is larger, star refinement with diamonds, no raster refinement. */
typedef struct { uint32_t best; int bx, by, dist, round, point; const uint8_t *org; int px, py, w, h, sub, ref; HmoMv pred; } HmoTz;
/* xTZSearchHelp, :336-441 (not the selective variant) */
static void tz_help(HmoEnc *e, HmoTz *s, int x, int y, int point, int dist)
{
  uint32_t sad = sad_ref(e, s->ref, s->org, 64, s->px + x, s->py + y, s->w, s->h, s->sub) + motion_cost(e, mv_bits(x, y, s->pred, 2));
  e->n_sad++;
  if (sad < s->best) { s->best = sad; s->bx = x; s->by = y; s->dist = dist; s->round = 0; s->point = point; }
}
/* xTZ2PointSearch, :442-573: the two untested neighbours of the best point of a distance-1 round */
static void tz_two_point(HmoEnc *e, HmoTz *s, HmoMv lt, HmoMv rb)
{
  const int x = s->bx, y = s->by;
  const int L = x - 1 >= lt.x, R = x + 1 <= rb.x, T = y - 1 >= lt.y, B = y + 1 <= rb.y;
  switch (s->point) {
  case 1: if (L) tz_help(e, s, x - 1, y, 0, 2); if (T) tz_help(e, s, x, y - 1, 0, 2); break;
  case 2: if (T) { if (L) tz_help(e, s, x - 1, y - 1, 0, 2); if (R) tz_help(e, s, x + 1, y - 1, 0, 2); } break;
  case 3: if (T) tz_help(e, s, x, y - 1, 0, 2); if (R) tz_help(e, s, x + 1, y, 0, 2); break;
  case 4: if (L) { if (B) tz_help(e, s, x - 1, y + 1, 0, 2); if (T) tz_help(e, s, x - 1, y - 1, 0, 2); } break;
  case 5: if (R) { if (T) tz_help(e, s, x + 1, y - 1, 0, 2); if (B) tz_help(e, s, x + 1, y + 1, 0, 2); } break;
  case 6: if (L) tz_help(e, s, x - 1, y, 0, 2); if (B) tz_help(e, s, x, y + 1, 0, 2); break;
  case 7: if (B) { if (L) tz_help(e, s, x - 1, y + 1, 0, 2); if (R) tz_help(e, s, x + 1, y + 1, 0, 2); } break;
  case 8: if (R) tz_help(e, s, x + 1, y, 0, 2); if (B) tz_help(e, s, x, y + 1, 0, 2); break;
  default: break;
  }
}
/* xTZ8PointDiamondSearch, :625-805 */
static void tz_diamond(HmoEnc *e, HmoTz *s, HmoMv lt, HmoMv rb, int sx, int sy, int d)
{
  const int top = sy - d, bot = sy + d, left = sx - d, right = sx + d;
  s->round += 1;
  if (d == 1) {
    if (top >= lt.y) tz_help(e, s, sx, top, 2, d);
    if (left >= lt.x) tz_help(e, s, left, sy, 4, d);
    if (right <= rb.x) tz_help(e, s, right, sy, 5, d);
    if (bot <= rb.y) tz_help(e, s, sx, bot, 7, d);
  } else if (d <= 8) {
    const int h = d >> 1, top2 = sy - h, bot2 = sy + h, left2 = sx - h, right2 = sx + h;
    if (top >= lt.y && left >= lt.x && right <= rb.x && bot <= rb.y) {
      tz_help(e, s, sx, top, 2, d); tz_help(e, s, left2, top2, 1, h); tz_help(e, s, right2, top2, 3, h);
      tz_help(e, s, left, sy, 4, d); tz_help(e, s, right, sy, 5, d);
      tz_help(e, s, left2, bot2, 6, h); tz_help(e, s, right2, bot2, 8, h); tz_help(e, s, sx, bot, 7, d);
    } else {
      if (top >= lt.y) tz_help(e, s, sx, top, 2, d);
      if (top2 >= lt.y) { if (left2 >= lt.x) tz_help(e, s, left2, top2, 1, h); if (right2 <= rb.x) tz_help(e, s, right2, top2, 3, h); }
      if (left >= lt.x) tz_help(e, s, left, sy, 4, d);
      if (right <= rb.x) tz_help(e, s, right, sy, 5, d);
      if (bot2 <= rb.y) { if (left2 >= lt.x) tz_help(e, s, left2, bot2, 6, h); if (right2 <= rb.x) tz_help(e, s, right2, bot2, 8, h); }
      if (bot <= rb.y) tz_help(e, s, sx, bot, 7, d);
    }
  } else {
    const int q = d >> 2;
    if (top >= lt.y && left >= lt.x && right <= rb.x && bot <= rb.y) {
      tz_help(e, s, sx, top, 0, d); tz_help(e, s, left, sy, 0, d); tz_help(e, s, right, sy, 0, d); tz_help(e, s, sx, bot, 0, d);
      for (int i = 1; i < 4; i++) {
        const int yt = top + q * i, yb = bot - q * i, xl = sx - q * i, xr = sx + q * i;
        tz_help(e, s, xl, yt, 0, d); tz_help(e, s, xr, yt, 0, d); tz_help(e, s, xl, yb, 0, d); tz_help(e, s, xr, yb, 0, d);
      }
    } else {
      if (top >= lt.y) tz_help(e, s, sx, top, 0, d);
      if (left >= lt.x) tz_help(e, s, left, sy, 0, d);
      if (right <= rb.x) tz_help(e, s, right, sy, 0, d);
      if (bot <= rb.y) tz_help(e, s, sx, bot, 0, d);
      for (int i = 1; i < 4; i++) {
        const int yt = top + q * i, yb = bot - q * i, xl = sx - q * i, xr = sx + q * i;
        if (yt >= lt.y) { if (xl >= lt.x) tz_help(e, s, xl, yt, 0, d); if (xr <= rb.x) tz_help(e, s, xr, yt, 0, d); }
        if (yb <= rb.y) { if (xl >= lt.x) tz_help(e, s, xl, yb, 0, d); if (xr <= rb.x) tz_help(e, s, xr, yb, 0, d); }
      }
    }
  }
}
/* xSetSearchRange, :3865-3884 */
static void set_search_range(const HmoEnc *e, const HmoCU *cu, HmoMv pred, int rng, HmoMv *lt, HmoMv *rb)
{
  HmoMv c = clip_mv(e, cu, pred);
  lt->x = c.x - (rng << 2); lt->y = c.y - (rng << 2); rb->x = c.x + (rng << 2); rb->y = c.y + (rng << 2);
  *lt = clip_mv(e, cu, *lt); *rb = clip_mv(e, cu, *rb);
  lt->x >>= 2; lt->y >>= 2; rb->x >>= 2; rb->y >>= 2;
}
static void tz_search(HmoEnc *e, const HmoCU *cu, HmoTz *s, HmoMv lt, HmoMv rb, const HmoMv *intMv2Nx2N, int *outX, int *outY)
{
  const int range = e->p.search_range, raster = 5;
  HmoMv rlt = lt, rrb = rb;                                   /* the raster search may use a window re-centred below */
  HmoMv st = clip_mv(e, cu, s->pred); st.x >>= 2; st.y >>= 2;
  s->best = HMO_MAX_UINT; s->bx = s->by = 0; s->dist = 0; s->round = 0; s->point = 0;
  tz_help(e, s, st.x, st.y, 0, 0);
  tz_help(e, s, 0, 0, 0, 0);                                  /* bTestZeroVector */
  if (intMv2Nx2N) {
    HmoMv m; m.x = intMv2Nx2N->x << 2; m.y = intMv2Nx2N->y << 2; m = clip_mv(e, cu, m); m.x >>= 2; m.y >>= 2;
    tz_help(e, s, m.x, m.y, 0, 0);
    HmoMv cur; cur.x = s->bx << 2; cur.y = s->by << 2;
    set_search_range(e, cu, cur, range, &rlt, &rrb);
  }
  int sx = s->bx, sy = s->by;
  for (int d = 1; d <= range; d *= 2) {                       /* first search */
    tz_diamond(e, s, lt, rb, sx, sy, d);
    if (s->round >= 3) break;                                 /* bFirstSearchStop, uiFirstSearchRounds */
  }
  if (s->dist == 1) { s->dist = 0; tz_two_point(e, s, lt, rb); }
  if (s->dist > raster) {                                     /* raster search */
    s->dist = raster;
    for (int y = rlt.y; y <= rrb.y; y += raster) for (int x = rlt.x; x <= rrb.x; x += raster) tz_help(e, s, x, y, 0, raster);
  }
  while (s->dist > 0) {                                       /* star refinement */
    sx = s->bx; sy = s->by; s->dist = 0; s->point = 0;
    for (int d = 1; d < range + 1; d *= 2) tz_diamond(e, s, lt, rb, sx, sy, d);
    if (s->dist == 1) { s->dist = 0; if (s->point != 0) tz_two_point(e, s, lt, rb); }
  }
  *outX = s->bx; *outY = s->by;
}

static void motion_estimation(HmoEnc *e, HmoCU *cu, int partSize, int pu, int refIdx, HmoMv pred, HmoMv *mvOut, uint32_t *bits, uint32_t *cost)
{
  int addr, ox, oy, w, h; pu_geom(cu, partSize, pu, &addr, &ox, &oy, &w, &h);
  const uint8_t *org = e->org_yuv[cu->depth_cu]->y + oy * 64 + ox;
  const int px = cu->x + ox, py = cu->y + oy, rng = e->p.search_range;
  HmoMv lt, rb;
  set_search_range(e, cu, pred, rng, &lt, &rb);
  const int subShift = (e->p.fast_enc && h > 8) ? 1 : 0;
  int bx = 0, by = 0;
  if (!e->p.fast_search) {                                   /* xPatternSearch, cost scale 2 */
    uint32_t best = HMO_MAX_UINT;
    for (int y = lt.y; y <= rb.y; y++)
      for (int x = lt.x; x <= rb.x; x++) {
        uint32_t s = sad_ref(e, refIdx, org, 64, px + x, py + y, w, h, subShift);
        s += motion_cost(e, mv_bits(x, y, pred, 2));
        if (s < best) { best = s; bx = x; by = y; }
      }
    e->n_sad += (uint64_t)(rb.y - lt.y + 1) * (uint64_t)(rb.x - lt.x + 1);
  } else {                                                   /* xPatternSearchFast -> xTZSearch; m_integerMv2Nx2N, TEncSearch.cpp:3822-3833 */
    HmoTz s; s.org = org; s.px = px; s.py = py; s.w = w; s.h = h; s.sub = subShift; s.pred = pred; s.ref = refIdx;
    const int usePred = partSize != HMO_SIZE_2Nx2N || cu->depth_cu != 0;
    HmoMv imv; imv.x = e->int_mv_2nx2n[refIdx].x; imv.y = e->int_mv_2nx2n[refIdx].y;
    tz_search(e, cu, &s, lt, rb, usePred ? &imv : NULL, &bx, &by);
    if (partSize == HMO_SIZE_2Nx2N) { e->int_mv_2nx2n[refIdx].x = bx; e->int_mv_2nx2n[refIdx].y = by; }
  }
  /* xPatternSearchFracDIF: half positions around (bx, by), cost scale 1; then quarter positions, cost scale 0 */
  uint8_t blk[64 * 64];
  uint32_t bestD = HMO_MAX_UINT; int bh = 0;
  for (int i = 0; i < 9; i++) {
    const int hx = k_refine_h[i][0], hy = k_refine_h[i][1];
    mc_block(e, refIdx, 0, px, py, w, h, (bx << 2) + 2 * hx, (by << 2) + 2 * hy, blk, 64);
    uint32_t d = e->p.had_me ? hmo_satd(org, 64, blk, 64, w, h) : sad_blocks(org, 64, blk, 64, w, h);
    d += motion_cost(e, mv_bits((bx << 1) + hx, (by << 1) + hy, pred, 1));
    if (d < bestD) { bestD = d; bh = i; }
  }
  const int hx = k_refine_h[bh][0], hy = k_refine_h[bh][1];
  bestD = HMO_MAX_UINT; int bq = 0;
  for (int i = 0; i < 9; i++) {
    const int qx = k_refine_q[i][0], qy = k_refine_q[i][1];
    const int mx = (bx << 2) + 2 * hx + qx, my = (by << 2) + 2 * hy + qy;
    mc_block(e, refIdx, 0, px, py, w, h, mx, my, blk, 64);
    uint32_t d = e->p.had_me ? hmo_satd(org, 64, blk, 64, w, h) : sad_blocks(org, 64, blk, 64, w, h);
    d += motion_cost(e, mv_bits(mx, my, pred, 0));
    if (d < bestD) { bestD = d; bq = i; }
  }
  HmoMv mv; mv.x = (bx << 2) + 2 * hx + k_refine_q[bq][0]; mv.y = (by << 2) + 2 * hy + k_refine_q[bq][1];
  const uint32_t mvBits = mv_bits(mv.x, mv.y, pred, 0);
  *bits += mvBits;
  *cost = (uint32_t)(floor(1.0 * ((double)bestD - (double)motion_cost(e, mvBits))) + (double)motion_cost(e, *bits));
  *mvOut = mv;
}

/* xCheckBestMVP */
static void check_best_mvp(const HmoEnc *e, HmoMv mv, const HmoMv cand[2], HmoMv *pred, int *mvpIdx, uint32_t *bits, uint32_t *cost)
{
  const int orgBits = (int)mv_bits(mv.x, mv.y, *pred, 0) + 1;   /* + m_auiMVPIdxCost[idx][2] */
  int bestBits = orgBits, bestIdx = *mvpIdx;
  for (int i = 0; i < 2; i++) {
    if (i == *mvpIdx) continue;
    const int b = (int)mv_bits(mv.x, mv.y, cand[i], 0) + 1;
    if (b < bestBits) { bestBits = b; bestIdx = i; }
  }
  if (bestIdx != *mvpIdx) {
    *pred = cand[bestIdx]; *mvpIdx = bestIdx;
    const uint32_t org = *bits;
    *bits = org - (uint32_t)orgBits + (uint32_t)bestBits;
    *cost = (*cost - motion_cost(e, org)) + motion_cost(e, *bits);
  }
}

/* xGetInterPredictionError: MC of the PU + Hadamard (or SAD) against the source */
static uint32_t inter_pred_error(HmoEnc *e, const HmoCU *cu, int partSize, int pu)
{
  int addr, ox, oy, w, h; pu_geom(cu, partSize, pu, &addr, &ox, &oy, &w, &h);
  mc_pu(e, cu, partSize, pu, &e->tmp_pred);
  const uint8_t *org = e->org_yuv[cu->depth_cu]->y + oy * 64 + ox, *p = e->tmp_pred.y + oy * 64 + ox;
  return e->p.had_me ? hmo_satd(org, 64, p, 64, w, h) : sad_blocks(org, 64, p, 64, w, h);
}

/* predInterSearch for a P slice: loop over the reference indices of list 0 */
static void pred_inter_search(HmoEnc *e, HmoCU *cu, int partSize, int useMrg)
{
  const int d = cu->depth_cu, npu = pu_count(partSize);
  const int normalMC = !(useMrg && cu_size(cu) > 8 && npu == 2);         /* AMP_MRG: merge estimation only (TEncSearch.cpp:3098-3103) */
  memset(e->pred_temp[d], 0, sizeof(HmoYuv));
  for (int pu = 0; pu < npu; pu++) {
    int addr, ox, oy, w, h; pu_geom(cu, partSize, pu, &addr, &ox, &oy, &w, &h);
    const uint32_t mbBits = (partSize == HMO_SIZE_2Nx2N || partSize == HMO_SIZE_NxN) ? 1 : 3;   /* xGetBlkBits, P slice */
    HmoMv pred, mv; int mvpIdx, refBest = 0;
    uint32_t bitsT = mbBits, costT = 0;
    HmoMv zero = { 0, 0 };
    mv = zero; pred = zero; mvpIdx = -1;
    if (normalMC) {
      uint32_t costBest = HMO_MAX_UINT;
      for (int r = 0; r < e->n_ref; r++) {                      /* uni-directional prediction, list 0: every reference index (:3110-3190) */
        HmoMv candR[2], predR, mvR; int idxR;
        uint32_t bitsR = mbBits, costR = 0;
        if (e->n_ref > 1) { bitsR += (uint32_t)r + 1; if (r == e->n_ref - 1) bitsR--; }      /* ref_idx bins (:3114-3121) */
        predR = estimate_mvp(e, cu, partSize, pu, r, candR, &idxR);
        bitsR += 1;                                            /* m_auiMVPIdxCost[idx][AMVP_MAX_NUM_CANDS] */
        motion_estimation(e, cu, partSize, pu, r, predR, &mvR, &bitsR, &costR);
        check_best_mvp(e, mvR, candR, &predR, &idxR, &bitsR, &costR);
        if (costR < costBest) { costBest = costR; bitsT = bitsR; costT = costR; mv = mvR; pred = predR; mvpIdx = idxR; refBest = r; }
      }
      /* motion field of the PU: list 0 wins by construction (:3413-3427) */
      HmoMv mvd; mvd.x = mv.x - pred.x; mvd.y = mv.y - pred.y;
      pu_set_motion(cu, partSize, pu, mv, refBest); pu_set_mvd(cu, partSize, pu, mvd); pu_set_dir(cu, partSize, pu, 1); pu_set_mvp(cu, partSize, pu, mvpIdx);
    } else {                                                   /* the cleared motion field (:3356-3363) */
      pu_set_motion(cu, partSize, pu, zero, -1); pu_set_mvd(cu, partSize, pu, zero); pu_set_mvp(cu, partSize, pu, -1);
    }
    pu_set_merge(cu, partSize, pu, 0, 0);
    if (partSize != HMO_SIZE_2Nx2N) {                          /* merge estimation of the PU (:3448-3498) */
      const uint32_t meCost = normalMC ? inter_pred_error(e, cu, partSize, pu) + motion_cost(e, bitsT) : HMO_MAX_UINT;
      HmoMv mmv[5]; int mref[5];
      const int nc = merge_candidates(e, cu, partSize, pu, mmv, mref);
      uint32_t mrgCost = HMO_MAX_UINT; int mrgIdx = 0;
      for (int c = 0; c < nc; c++) {                           /* xMergeEstimation */
        pu_set_motion(cu, partSize, pu, mmv[c], mref[c]);
        uint32_t cc = inter_pred_error(e, cu, partSize, pu);
        uint32_t b = (uint32_t)c + 1; if (c == e->p.max_merge_cand - 1) b--;
        cc += motion_cost(e, b);
        if (cc < mrgCost) { mrgCost = cc; mrgIdx = c; }
      }
      if (mrgCost < meCost) {
        pu_set_merge(cu, partSize, pu, 1, mrgIdx); pu_set_dir(cu, partSize, pu, 1);
        pu_set_motion(cu, partSize, pu, mmv[mrgIdx], mref[mrgIdx]); pu_set_mvd(cu, partSize, pu, zero); pu_set_mvp(cu, partSize, pu, -1);
      } else {
        pu_set_merge(cu, partSize, pu, 0, 0); pu_set_dir(cu, partSize, pu, 1); pu_set_motion(cu, partSize, pu, mv, refBest);
      }
    }
    mc_pu(e, cu, partSize, pu, e->pred_temp[d]);
  }
}

/* ------------------------------------------------------------------------------------
 * inter syntax
 * ---------------------------------------------------------------------------------- */
static void code_skip_flag(HmoEnc *e, const HmoCU *cu, int part)
{
  const int z = cu->zidx + part, lx = (cu->x & ~63) + part_x(z), ly = (cu->y & ~63) + part_y(z);
  HmoNb l = nb_motion(e, cu, lx - 1, ly, lx, ly), a = nb_motion(e, cu, lx, ly - 1, lx, ly);
  hmo_enc_bin(e, cu->skip[part], HMO_CTX_SKIP + (l.avail ? l.skip : 0) + (a.avail ? a.skip : 0));
}
static void code_pred_mode(HmoEnc *e, const HmoCU *cu, int part) { hmo_enc_bin(e, cu->pred_mode[part] == HMO_MODE_INTRA, HMO_CTX_PRED_MODE); }
static void code_merge_index(HmoEnc *e, const HmoCU *cu, int part)
{
  const int idx = cu->merge_idx[part], n = e->p.max_merge_cand;
  for (int ui = 0; ui < n - 1; ui++) {
    const int sym = ui == idx ? 0 : 1;
    if (ui == 0) hmo_enc_bin(e, sym, HMO_CTX_MERGE_IDX); else hmo_enc_bins_ep(e, 1);
    if (!sym) break;
  }
}
/* codePartSize for an inter CU (AMP off) */
static void code_part_size_inter(HmoEnc *e, const HmoCU *cu, int part, int depth)
{
  const int ps = cu->part_size[part], amp = e->p.amp && depth < HMO_MAXDEPTH;       /* codePartSize, TEncSbac.cpp:436-520 */
  if (ps == HMO_SIZE_2Nx2N) { hmo_enc_bin(e, 1, HMO_CTX_PARTSIZE); return; }
  hmo_enc_bin(e, 0, HMO_CTX_PARTSIZE);
  const int hor = ps == HMO_SIZE_2NxN || ps == HMO_SIZE_2NxnU || ps == HMO_SIZE_2NxnD;
  hmo_enc_bin(e, hor, HMO_CTX_PARTSIZE1);
  if (!hor && depth == HMO_MAXDEPTH && !((HMO_CTU >> depth) == 8)) hmo_enc_bin(e, 1, HMO_CTX_PARTSIZE1 + 1);
  if (amp) {
    if (ps == HMO_SIZE_2NxN || ps == HMO_SIZE_Nx2N) hmo_enc_bin(e, 1, HMO_CTX_PARTSIZE1 + 2);
    else { hmo_enc_bin(e, 0, HMO_CTX_PARTSIZE1 + 2); hmo_enc_bins_ep(e, 1); }      /* 2NxnU / nLx2N: 0, 2NxnD / nRx2N: 1 */
  }
}
static int ep_exgolomb_bins(uint32_t symbol, uint32_t count)                 /* xWriteEpExGolomb, TEncSbac.cpp:300-320 */
{ int n = 0; while (symbol >= (1u << count)) { n++; symbol -= 1u << count; count++; } return n + 1 + (int)count; }
static void code_mvd(HmoEnc *e, int hor, int ver)
{
  hmo_enc_bin(e, hor != 0, HMO_CTX_MVD); hmo_enc_bin(e, ver != 0, HMO_CTX_MVD);
  const uint32_t ah = (uint32_t)abs(hor), av = (uint32_t)abs(ver);
  if (hor) hmo_enc_bin(e, ah > 1, HMO_CTX_MVD + 1);
  if (ver) hmo_enc_bin(e, av > 1, HMO_CTX_MVD + 1);
  if (hor) { if (ah > 1) hmo_enc_bins_ep(e, ep_exgolomb_bins(ah - 2, 1)); hmo_enc_bins_ep(e, 1); }
  if (ver) { if (av > 1) hmo_enc_bins_ep(e, ep_exgolomb_bins(av - 2, 1)); hmo_enc_bins_ep(e, 1); }
}
/* codeRefFrmIdx, TEncSbac.cpp:743-775: first bin on context 0, second on context 1, the rest bypass (truncated unary) */
static void code_ref_idx(HmoEnc *e, int refIdx)
{
  hmo_enc_bin(e, refIdx == 0 ? 0 : 1, HMO_CTX_REF);
  if (refIdx > 0) {
    const int refNum = e->n_ref - 2; refIdx--;
    for (int ui = 0; ui < refNum; ui++) {
      const int sym = ui == refIdx ? 0 : 1;
      if (ui == 0) hmo_enc_bin(e, sym, HMO_CTX_REF + 1); else hmo_enc_bins_ep(e, 1);
      if (!sym) break;
    }
  }
}
/* encodePUWise, TEncEntropy.cpp:456-507 (P slice: no inter_pred_idc; ref_idx only with several reference pictures) */
static void code_pu_wise(HmoEnc *e, const HmoCU *cu, int part)
{
  const int ps = cu->part_size[part], npu = pu_count(ps), n = HMO_NPART >> (2 * cu->depth[part]);
  static const uint8_t k_off16[8] = { 0, 8, 4, 4, 2, 10, 1, 5 };                        /* g_auiPUOffset, TComRom.cpp; TEncEntropy.cpp:339 */
  const int off = (k_off16[ps] * n) >> 4;
  for (int pu = 0, sp = part; pu < npu; pu++, sp += off) {
    hmo_enc_bin(e, cu->merge_flag[sp], HMO_CTX_MERGE_FLAG);
    if (cu->merge_flag[sp]) code_merge_index(e, cu, sp);
    else { if (e->n_ref > 1) code_ref_idx(e, cu->ref_idx[sp]); code_mvd(e, cu->mvd[sp][0], cu->mvd[sp][1]); hmo_enc_bin(e, cu->mvp_idx[sp], HMO_CTX_MVP_IDX); }
  }
}
static int qt_root_cbf(const HmoCU *cu, int part) { return ((cu->cbf[0][part] | cu->cbf[1][part] | cu->cbf[2][part]) & 1); }

/* ------------------------------------------------------------------------------------
 * inter residual quadtree (xEstimateInterResidualQT)
 * ---------------------------------------------------------------------------------- */
static int16_t *yuv16_plane(HmoYuv16 *b, int comp) { return comp == 0 ? b->y : (comp == 1 ? b->u : b->v); }
static uint32_t sse16(const int16_t *a, int sa, const int16_t *b, int sb, int w, int h)
{ uint32_t s = 0; for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) { const int d = a[y * sa + x] - b[y * sb + x]; s += (uint32_t)(d * d); } return s; }
static uint32_t comp_dist(const HmoEnc *e, int comp, uint32_t sse) { return comp ? (uint32_t)(e->p.chroma_weight * (double)sse) : sse; }
static int min_tu_log2_inter(const HmoCU *cu) { int l = 6 - cu->depth_cu; if (l < HMO_LOG2_MINTU + 3 - 1) return HMO_LOG2_MINTU; l -= 2; return l > HMO_LOG2_MAXTU ? HMO_LOG2_MAXTU : l; }   /* QuadtreeTUMaxDepthInter 3 */
static void code_qt_cbf_zero(HmoEnc *e, const HmoTU *tu, int ch)
{ hmo_enc_bin(e, 0, ch ? (HMO_CTX_CBF_CHROMA + tu->tr_depth) : (HMO_CTX_CBF_LUMA + (tu->tr_depth == 0 ? 1 : 0))); }

/* xEncodeInterResidualQT: comp == 3 codes the subdivision / cbf flags, else the coefficients of one component */
static void encode_inter_residual_qt(HmoEnc *e, const HmoCU *cu, const HmoTU *tu, int comp)
{
  const int curTrMode = tu->tr_depth, trMode = cu->tr_idx[tu->part], subdiv = curTrMode != trMode, log2 = tu->log2;
  if (comp == 3) {
    if (log2 <= HMO_LOG2_MAXTU && log2 > min_tu_log2_inter(cu)) hmo_enc_bin(e, subdiv, HMO_CTX_SUBDIV + 5 - log2);
    const int first = curTrMode == 0;
    for (int c = 1; c < 3; c++)
      if (first || tu->c_code_all)
        if (first || ((cu->cbf[c][tu->part] >> (curTrMode - 1)) & 1)) code_qt_cbf(e, cu, tu, c, !subdiv);
    if (!subdiv) code_qt_cbf(e, cu, tu, 0, 1);
  }
  if (!subdiv) {
    if (comp != 3 && !(comp && tu->cw == 0)) {
      if ((cu->cbf[comp][tu->part] >> trMode) & 1) {
        const int layer = HMO_LOG2_MAXTU - log2, N = comp ? tu->cw : (1 << log2); int l2 = 2; while ((1 << l2) < N) l2++;
        hmo_code_coeff_nxn(e, cu, e->qt_coef[comp][layer] + (comp ? tu->off_c : tu->off_y), l2, comp, comp ? tu_part_c(tu) : tu->part);
      }
    }
  } else if (comp == 3 || ((cu->cbf[comp][tu->part] >> curTrMode) & 1)) {
    for (int i = 0; i < 4; i++) { HmoTU c; tu_child(&c, tu, i, 0); encode_inter_residual_qt(e, cu, &c, comp); }
  }
}

static void est_inter_residual_qt(HmoEnc *e, HmoCU *cu, const HmoTU *tu, double *rdCost, uint32_t *rBits, uint32_t *rDist, uint32_t *zeroDist)
{
  const int part = tu->part, trMode = tu->tr_depth, depth = cu->depth_cu + trMode, log2 = tu->log2, layer = HMO_LOG2_MAXTU - log2;
  const int checkFull = log2 <= HMO_LOG2_MAXTU, checkSplit = log2 > min_tu_log2_inter(cu);
  double singleCost = HMO_MAX_DOUBLE; uint32_t singleBits = 0, singleDist = 0, singleDistComp[3] = { 0, 0, 0 };
  int absSum[3] = { 0, 0, 0 }, bestTS[3] = { 0, 0, 0 };
  e->slot[depth][CI_QT_TRAFO_ROOT] = e->goon;
  if (checkFull) {
    memset(cu->tr_idx + part, trMode, (size_t)tu->nparts);
    for (int comp = 0; comp < 3; comp++) {
      if (comp && tu->cw == 0) continue;
      const int N = comp ? tu->cw : (1 << log2), bs = comp ? 32 : 64, bx = comp ? tu->cx : tu->x, by = comp ? tu->cy : tu->y;
      int l2 = 2; while ((1 << l2) < N) l2++;
      const int cpart = comp ? tu_part_c(tu) : part, cnp = comp ? tu_nparts_c(tu) : tu->nparts;
      const int16_t *resi = yuv16_plane(&e->resi_cu, comp) + by * bs + bx;
      int16_t *rq = yuv16_plane(&e->qt_resi[layer], comp) + by * bs + bx;
      int32_t *coef = e->qt_coef[comp][layer] + (comp ? tu->off_c : tu->off_y);
      const int nModes = (e->p.transform_skip && N == 4) ? 2 : 1;
      double minCost = HMO_MAX_DOUBLE;
      int32_t bestCoef[32 * 32]; int16_t bestResi[32 * 32];
      for (int ts = 0; ts < nModes; ts++) {
        const int isFirst = ts == 0;
        memset(cu->tskip[comp] + cpart, ts, (size_t)cnp);
        e->goon = e->slot[depth][CI_QT_TRAFO_ROOT]; hmo_reset_bits(e);
        if (nModes > 1 && !isFirst) { memcpy(bestCoef, coef, sizeof(int32_t) * (size_t)(N * N)); for (int y = 0; y < N; y++) memcpy(bestResi + y * N, rq + y * bs, sizeof(int16_t) * (size_t)N); }
        /* transformNxN */
        int16_t r[32 * 32]; int32_t tcoef[32 * 32];
        for (int y = 0; y < N; y++) for (int x = 0; x < N; x++) r[y * N + x] = resi[y * bs + x];
        if (ts) { for (int i = 0; i < N * N; i++) tcoef[i] = (int32_t)r[i] << (15 - 8 - l2); }
        else hmo_fwd_transform(r, N, tcoef, l2, 0);
        e->n_tu_trials++;
        int curAbs = hmo_rdoq(e, cu, tu, comp, tcoef, coef, l2, cpart, ts);
        memset(cu->cbf[comp] + cpart, (curAbs > 0 ? 1 : 0) << trMode, (size_t)cnp);
        if (curAbs == 0) memset(coef, 0, sizeof(int32_t) * (size_t)(N * N));
        uint32_t nonBits = 0, nonDist = 0; double nonCost = 0;
        if (isFirst || curAbs == 0) {
          uint32_t s = 0; for (int y = 0; y < N; y++) for (int x = 0; x < N; x++) s += (uint32_t)(resi[y * bs + x] * resi[y * bs + x]);
          nonDist = comp_dist(e, comp, s);
          code_qt_cbf_zero(e, tu, comp ? 1 : 0);
          nonBits = hmo_bits(e); nonCost = calc_rd_cost(e, nonBits, nonDist);
        }
        if (zeroDist && isFirst) *zeroDist += nonDist;
        uint32_t curBits = 0, curDist = 0; double curCost = 0;
        if (curAbs > 0) {
          if (isFirst) { e->goon = e->slot[depth][CI_QT_TRAFO_ROOT]; hmo_reset_bits(e); }
          code_qt_cbf(e, cu, tu, comp, 1);
          hmo_code_coeff_nxn(e, cu, coef, l2, comp, cpart);
          curBits = hmo_bits(e);
          int32_t dq[32 * 32]; int16_t rr[32 * 32];
          hmo_dequant(coef, dq, N * N, l2, comp ? e->p.qp_c : e->p.qp);
          if (ts) { const int s = 15 - 8 - l2; for (int i = 0; i < N * N; i++) rr[i] = (int16_t)((dq[i] + (1 << (s - 1))) >> s); }
          else hmo_inv_transform(dq, rr, N, l2, 0);
          for (int y = 0; y < N; y++) memcpy(rq + y * bs, rr + y * N, sizeof(int16_t) * (size_t)N);
          curDist = comp_dist(e, comp, sse16(rq, bs, resi, bs, N, N));
          curCost = calc_rd_cost(e, curBits, curDist);
        } else if (ts == 1) curCost = HMO_MAX_DOUBLE;
        else { curBits = nonBits; curDist = nonDist; curCost = nonCost; }
        if (curCost < minCost || (ts == 1 && curCost == minCost)) {
          if (isFirst && (nonCost < curCost || curAbs == 0)) { memset(coef, 0, sizeof(int32_t) * (size_t)(N * N)); curAbs = 0; curBits = nonBits; curDist = nonDist; curCost = nonCost; }
          absSum[comp] = curAbs; singleDistComp[comp] = curDist; minCost = curCost; bestTS[comp] = ts;
          if (curAbs == 0) for (int y = 0; y < N; y++) memset(rq + y * bs, 0, sizeof(int16_t) * (size_t)N);
        } else {
          memcpy(coef, bestCoef, sizeof(int32_t) * (size_t)(N * N));
          for (int y = 0; y < N; y++) memcpy(rq + y * bs, bestResi + y * N, sizeof(int16_t) * (size_t)N);
        }
      }
      memset(cu->tskip[comp] + cpart, bestTS[comp], (size_t)cnp);
      memset(cu->cbf[comp] + cpart, (absSum[comp] > 0 ? 1 : 0) << trMode, (size_t)cnp);
    }
    e->goon = e->slot[depth][CI_QT_TRAFO_ROOT]; hmo_reset_bits(e);
    if (log2 > min_tu_log2_inter(cu)) hmo_enc_bin(e, 0, HMO_CTX_SUBDIV + 5 - log2);
    for (int k = 0; k < 3; k++) { const int comp = (k + 1) % 3; if (comp && tu->cw == 0) continue; code_qt_cbf(e, cu, tu, comp, 1); }   /* Cb, Cr, Y */
    for (int comp = 0; comp < 3; comp++) {
      if (comp && tu->cw == 0) continue;
      const int N = comp ? tu->cw : (1 << log2); int l2 = 2; while ((1 << l2) < N) l2++;
      if ((cu->cbf[comp][comp ? tu_part_c(tu) : part] >> trMode) & 1)
        hmo_code_coeff_nxn(e, cu, e->qt_coef[comp][layer] + (comp ? tu->off_c : tu->off_y), l2, comp, comp ? tu_part_c(tu) : part);
      singleDist += singleDistComp[comp];
    }
    singleBits = hmo_bits(e);
    singleCost = calc_rd_cost(e, singleBits, singleDist);
  }
  if (checkSplit) {
    if (checkFull) { e->slot[depth][CI_QT_TRAFO_TEST] = e->goon; e->goon = e->slot[depth][CI_QT_TRAFO_ROOT]; }
    uint32_t subDist = 0, subBits = 0; double subCost = 0.0;
    int bestCbf[3];
    for (int comp = 0; comp < 3; comp++) bestCbf[comp] = (cu->cbf[comp][part] >> trMode) & 1;
    for (int i = 0; i < 4; i++) { HmoTU c; tu_child(&c, tu, i, 0); est_inter_residual_qt(e, cu, &c, &subCost, &subBits, &subDist, checkFull ? NULL : zeroDist); }
    int cbfAny = 0; const int q = tu->nparts >> 2;
    for (int comp = 0; comp < 3; comp++) {
      int yuv = 0;
      for (int i = 0; i < 4; i++) yuv |= (cu->cbf[comp][part + i * q] >> (trMode + 1)) & 1;
      for (int o = 0; o < 4 * q; o++) cu->cbf[comp][part + o] |= (uint8_t)(yuv << trMode);
      cbfAny |= yuv;
    }
    e->goon = e->slot[depth][CI_QT_TRAFO_ROOT]; hmo_reset_bits(e);
    encode_inter_residual_qt(e, cu, tu, 3);
    for (int comp = 0; comp < 3; comp++) encode_inter_residual_qt(e, cu, tu, comp);
    subBits = hmo_bits(e);
    subCost = calc_rd_cost(e, subBits, subDist);
    if (!checkFull || (cbfAny && subCost < singleCost)) { *rdCost += subCost; *rBits += subBits; *rDist += subDist; return; }
    *rdCost += singleCost; *rBits += singleBits; *rDist += singleDist;
    memset(cu->tr_idx + part, trMode, (size_t)tu->nparts);
    for (int comp = 0; comp < 3; comp++) {
      if (comp && tu->cw == 0) continue;
      const int cpart = comp ? tu_part_c(tu) : part, cnp = comp ? tu_nparts_c(tu) : tu->nparts;
      memset(cu->cbf[comp] + cpart, bestCbf[comp] << trMode, (size_t)cnp);
      memset(cu->tskip[comp] + cpart, bestTS[comp], (size_t)cnp);
    }
    e->goon = e->slot[depth][CI_QT_TRAFO_TEST];
    return;
  }
  *rdCost += singleCost; *rBits += singleBits; *rDist += singleDist;
}

/* xSetInterResidualQTData: leaves of the chosen tree -> CU coefficients (spatial = 0) or residual samples (spatial = 1) */
static void set_inter_residual_qt_data(HmoEnc *e, HmoCU *cu, const HmoTU *tu, HmoYuv16 *resiOut, int spatial)
{
  if (tu->tr_depth != cu->tr_idx[tu->part]) { for (int i = 0; i < 4; i++) { HmoTU c; tu_child(&c, tu, i, 0); set_inter_residual_qt_data(e, cu, &c, resiOut, spatial); } return; }
  const int layer = HMO_LOG2_MAXTU - tu->log2;
  for (int comp = 0; comp < 3; comp++) {
    if (comp && tu->cw == 0) continue;
    const int N = comp ? tu->cw : (1 << tu->log2), bs = comp ? 32 : 64, bx = comp ? tu->cx : tu->x, by = comp ? tu->cy : tu->y, off = comp ? tu->off_c : tu->off_y;
    if (spatial) { const int16_t *s = yuv16_plane(&e->qt_resi[layer], comp) + by * bs + bx; int16_t *t = yuv16_plane(resiOut, comp) + by * bs + bx; for (int y = 0; y < N; y++) memcpy(t + y * bs, s + y * bs, sizeof(int16_t) * (size_t)N); }
    else memcpy(cu->coef[comp] + off, e->qt_coef[comp][layer] + off, sizeof(int32_t) * (size_t)(N * N));
  }
}

/* final-order transform tree of an inter CU: TEncEntropy::xEncodeTransform (TEncEntropy.cpp:201-400), inter branch */
static void encode_transform_inter(HmoEnc *e, const HmoCU *cu, int cuPart, const HmoTU *tu)
{
  const int part = cuPart + tu->part, trIdx = tu->tr_depth, subdiv = cu->tr_idx[part] > trIdx;
  int cbf[3]; for (int c = 0; c < 3; c++) cbf[c] = (cu->cbf[c][part] >> trIdx) & 1;
  HmoCU view_dummy; (void)view_dummy;
  const int log2Cb = 6 - cu->depth[part]; int minl = log2Cb < 4 ? HMO_LOG2_MINTU : log2Cb - 2; if (minl > HMO_LOG2_MAXTU) minl = HMO_LOG2_MAXTU;
  if (tu->log2 > HMO_LOG2_MAXTU) { }
  else if (tu->log2 == HMO_LOG2_MINTU) { }
  else if (tu->log2 == minl) { }
  else hmo_enc_bin(e, subdiv, HMO_CTX_SUBDIV + 5 - tu->log2);
  const int first = trIdx == 0;
  for (int comp = 1; comp < 3; comp++)
    if (first || tu->c_code_all)
      if (first || ((cu->cbf[comp][part] >> (trIdx - 1)) & 1)) {
        const int lowest = trIdx + ((subdiv && !(tu->cwo >= 8)) ? 1 : 0);
        hmo_enc_bin(e, (cu->cbf[comp][cuPart + tu_part_c(tu)] >> lowest) & 1, HMO_CTX_CBF_CHROMA + trIdx);
      }
  if (subdiv) { for (int i = 0; i < 4; i++) { HmoTU c; tu_child(&c, tu, i, 1); encode_transform_inter(e, cu, cuPart, &c); } return; }
  if (!(trIdx == 0 && !((cu->cbf[1][part] & 1) || (cu->cbf[2][part] & 1)))) hmo_enc_bin(e, cbf[0], HMO_CTX_CBF_LUMA + (trIdx == 0 ? 1 : 0));
  for (int comp = 0; comp < 3; comp++) {
    if (comp && tu->cw == 0) continue;
    if (!cbf[comp]) continue;
    const int N = comp ? tu->cw : (1 << tu->log2); int l2 = 2; while ((1 << l2) < N) l2++;
    hmo_code_coeff_nxn(e, cu, cu->coef[comp] + (comp ? (cuPart * 4 + tu->off_c) : (cuPart * 16 + tu->off_y)), l2, comp, cuPart + (comp ? tu_part_c(tu) : tu->part));
  }
}
/* syntax of one inter CU: xAddSymbolBitsInter (search time) / xEncodeCU (replay) */
static void encode_cu_syntax_inter(HmoEnc *e, const HmoCU *cu, int cuPart, int depth)
{
  code_skip_flag(e, cu, cuPart);
  if (cu->skip[cuPart]) { code_merge_index(e, cu, cuPart); return; }
  code_pred_mode(e, cu, cuPart);
  code_part_size_inter(e, cu, cuPart, depth);
  code_pu_wise(e, cu, cuPart);
  if (!(cu->merge_flag[cuPart] && cu->part_size[cuPart] == HMO_SIZE_2Nx2N)) hmo_enc_bin(e, qt_root_cbf(cu, cuPart), HMO_CTX_ROOT_CBF);
  if (!qt_root_cbf(cu, cuPart)) return;
  HmoTU root; memset(&root, 0, sizeof(root));
  root.log2 = 6 - depth; root.nparts = HMO_NPART >> (2 * depth); root.cw = root.cwo = (HMO_CTU >> depth) >> 1; root.c_code_all = 1;
  encode_transform_inter(e, cu, cuPart, &root);
}

/* encodeResAndCalcRdInterCU: the prediction of the whole CU is in pred_temp[d] */
static void encode_res_and_calc_rd_inter_cu(HmoEnc *e, HmoCU *cu, int skipResidual)
{
  const int d = cu->depth_cu, s = cu_size(cu), n = cu->nparts;
  HmoYuv *org = e->org_yuv[d], *pred = e->pred_temp[d], *rec = e->reco_temp[d];
  if (skipResidual) {
    memset(cu->skip, 1, (size_t)n);
    uint32_t dist = 0;
    for (int comp = 0; comp < 3; comp++) {
      const int bs = comp ? 32 : 64, w = comp ? s >> 1 : s;
      uint8_t *r = yuv_plane(rec, comp), *p = yuv_plane(pred, comp), *o = yuv_plane(org, comp);
      for (int y = 0; y < w; y++) memcpy(r + y * bs, p + y * bs, (size_t)w);
      dist += comp_dist(e, comp, hmo_sse(r, bs, o, bs, w, w));
    }
    e->goon = e->slot[d][CI_CURR_BEST]; hmo_reset_bits(e);
    code_skip_flag(e, cu, 0); code_merge_index(e, cu, 0);
    cu->bits = hmo_bits(e); cu->dist = dist; cu->cost = calc_rd_cost(e, cu->bits, dist);   /* TotalBins is not set on this path */
    e->slot[d][CI_TEMP_BEST] = e->goon;
    return;
  }
  for (int comp = 0; comp < 3; comp++) {
    const int bs = comp ? 32 : 64, w = comp ? s >> 1 : s;
    int16_t *r = yuv16_plane(&e->resi_cu, comp); const uint8_t *p = yuv_plane(pred, comp), *o = yuv_plane(org, comp);
    for (int y = 0; y < w; y++) for (int x = 0; x < w; x++) r[y * bs + x] = (int16_t)(o[y * bs + x] - p[y * bs + x]);
  }
  HmoTU root; tu_root(&root, cu);
  double nonZeroCost = 0; uint32_t nonZeroBits = 0, nonZeroDist = 0, zeroDist = 0;
  e->goon = e->slot[d][CI_CURR_BEST];
  est_inter_residual_qt(e, cu, &root, &nonZeroCost, &nonZeroBits, &nonZeroDist, &zeroDist);
  hmo_reset_bits(e);
  hmo_enc_bin(e, 0, HMO_CTX_ROOT_CBF);                         /* encodeQtRootCbfZero */
  const uint32_t zeroBits = hmo_bits(e);
  const double zeroCost = calc_rd_cost(e, zeroBits, zeroDist);
  if (zeroCost < nonZeroCost || !qt_root_cbf(cu, 0)) {
    memset(cu->tr_idx, 0, (size_t)n);
    for (int c = 0; c < 3; c++) { memset(cu->cbf[c], 0, (size_t)n); memset(cu->tskip[c], 0, (size_t)n); }
  } else set_inter_residual_qt_data(e, cu, &root, NULL, 0);
  e->goon = e->slot[d][CI_CURR_BEST];
  /* xAddSymbolBitsInter */
  if (cu->merge_flag[0] && cu->part_size[0] == HMO_SIZE_2Nx2N && !qt_root_cbf(cu, 0)) memset(cu->skip, 1, (size_t)n);
  hmo_reset_bits(e);
  encode_cu_syntax_inter(e, cu, 0, d);
  const uint32_t finalBits = hmo_bits(e);
  HmoYuv16 *rb = &e->resi_best;
  if (!qt_root_cbf(cu, 0)) memset(rb, 0, sizeof(*rb));
  else set_inter_residual_qt_data(e, cu, &root, rb, 1);
  e->slot[d][CI_TEMP_BEST] = e->goon;
  uint32_t dist = 0;
  for (int comp = 0; comp < 3; comp++) {
    const int bs = comp ? 32 : 64, w = comp ? s >> 1 : s;
    uint8_t *r = yuv_plane(rec, comp); const uint8_t *p = yuv_plane(pred, comp), *o = yuv_plane(org, comp); const int16_t *q = yuv16_plane(rb, comp);
    for (int y = 0; y < w; y++) for (int x = 0; x < w; x++) { const int v = p[y * bs + x] + q[y * bs + x]; r[y * bs + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
    dist += comp_dist(e, comp, hmo_sse(r, bs, o, bs, w, w));
  }
  cu->bits = finalBits; cu->dist = dist; cu->cost = calc_rd_cost(e, finalBits, dist);
}

/* xCheckRDCostInter */
static int check_best_mode(HmoEnc *e, int d);
static void check_rd_cost_inter(HmoEnc *e, int d, int partSize, int useMrg)
{
  HmoCU *cu = e->temp[d];
  memset(cu->part_size, partSize, (size_t)cu->nparts);
  memset(cu->pred_mode, HMO_MODE_INTER, (size_t)cu->nparts);
  if (e->trace) e->trace(e->trace_user, HMO_EV_INTER_BEGIN, d, partSize | (useMrg << 4));
  pred_inter_search(e, cu, partSize, useMrg);
  encode_res_and_calc_rd_inter_cu(e, cu, 0);
  if (e->trace) e->trace(e->trace_user, HMO_EV_INTER_END, d, partSize | (useMrg << 4));
  check_best_mode(e, d);
}
/* xCheckRDCostMerge2Nx2N (early skip detection off) */
static void check_rd_cost_merge_2nx2n(HmoEnc *e, int d)
{
  const int x = e->temp[d]->x, y = e->temp[d]->y, zidx = e->temp[d]->zidx;
  HmoMv mmv[5]; int mref[5], candBuf[5] = { 0, 0, 0, 0, 0 };
  memset(e->temp[d]->part_size, HMO_SIZE_2Nx2N, (size_t)e->temp[d]->nparts);
  const int nc = merge_candidates(e, e->temp[d], HMO_SIZE_2Nx2N, 0, mmv, mref);
  int bestIsSkip = 0;
  for (int noRes = 0; noRes < 2; noRes++)
    for (int c = 0; c < nc; c++) {
      if (noRes == 1 && candBuf[c] == 1) continue;
      if (bestIsSkip && noRes == 0) continue;
      HmoCU *cu = e->temp[d];
      const int n = cu->nparts;
      memset(cu->pred_mode, HMO_MODE_INTER, (size_t)n); memset(cu->part_size, HMO_SIZE_2Nx2N, (size_t)n);
      pu_set_merge(cu, HMO_SIZE_2Nx2N, 0, 1, c); pu_set_dir(cu, HMO_SIZE_2Nx2N, 0, 1); pu_set_motion(cu, HMO_SIZE_2Nx2N, 0, mmv[c], mref[c]);
      if (e->trace) e->trace(e->trace_user, HMO_EV_MERGE_BEGIN, d, c * 2 + noRes);
      mc_pu(e, cu, HMO_SIZE_2Nx2N, 0, e->pred_temp[d]);
      encode_res_and_calc_rd_inter_cu(e, cu, noRes);
      if (e->trace) e->trace(e->trace_user, HMO_EV_MERGE_END, d, c * 2 + noRes);
      if (noRes == 0 && !qt_root_cbf(cu, 0)) candBuf[c] = 1;
      check_best_mode(e, d);
      cu_init(e->temp[d], d, x, y, zidx);
      if (e->p.fdm && !bestIsSkip) bestIsSkip = !qt_root_cbf(e->best[d], 0);
    }
}
