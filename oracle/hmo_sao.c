/*
 * hmo_sao.c -- ORACLE (test infrastructure, never linked into the product).  Sample adaptive offset of one picture:
 * statistics, per-CTU parameter decision (new / merge, rate from the CABAC bit counter) and the offset pass, restating
 *   TEncSampleAdaptiveOffset::SAOProcess     TLibEncoder/TEncSampleAdaptiveOffset.cpp:257-287
 *   getStatistics / getBlkStats              :310-361, :922-1381   (non pre-deblock statistics: SAOLcuBoundary 0)
 *   deriveOffsets / estIterOffset            :441-591
 *   deriveModeNewRDO / deriveModeMergeRDO    :593-788
 *   decideBlkParams                          :790-920
 *   TComSampleAdaptiveOffset::offsetBlock / offsetCTU / getMergeList / reconstructBlkSAOParam
 *                                            TLibCommon/TComSampleAdaptiveOffset.cpp:171-616
 *   TEncSbac::codeSAOOffsetParam / codeSAOBlkParam   TLibEncoder/TEncSbac.cpp:1540-1714
 * for 8-bit 4:2:0, 64x64 CTUs, LFCrossSliceBoundaryFlag 1 and no tiles (HM defaults: neighbour availability is the picture
 * boundary, deriveLoopFilterBoundaryAvailibility TComPicSym.cpp:357-376), merge candidates limited to the CTU's own slice
 * (TComPic::getSAOMergeAvailability, TComPic.cpp:138-143).
 *
 * Pinned by the reference's own classes built in place (oracle/ref/ref_driver.cpp:ref_sao, tests/test_golden_sao.py).
 */
#include "hmo_int.h"
#include <stdlib.h>
#include <string.h>
#include <math.h>

enum { SAO_OFF = 0, SAO_NEW = 1, SAO_MERGE = 2, SAO_BO = 4, SAO_NTYPES = 5, SAO_MAXQ = 7 /* g_saoMaxOffsetQVal, 8 bit */ };

/* ---- bit counter: the two SAO contexts of a TEncSbac + the Q15 fraction (TEncBinCABACCounter) ------------------------- */
typedef struct { uint8_t ctx[2]; uint64_t frac; } SaoCab;            /* ctx[0] sao_merge_flag, ctx[1] sao_type_idx */
static uint8_t ctx_from_init(int iv, int qp)
{
  if (qp < 0) qp = 0;
  if (qp > 51) qp = 51;
  int slope = (iv >> 4) * 5 - 45, offset = ((iv & 15) << 3) - 16, st = ((slope * qp) >> 4) + offset;
  if (st < 1) st = 1;
  if (st > 126) st = 126;
  const int mps = st >= 64;
  return (uint8_t)(((mps ? (st - 64) : (63 - st)) << 1) + mps);
}
static void sc_bin(SaoCab *c, int bin, int k)
{
  const uint8_t s = c->ctx[k];
  c->frac += (uint64_t)hmo_entropy_bits[s ^ bin];
  c->ctx[k] = ((s & 1) == bin) ? hmo_next_mps[s] : hmo_next_lps[s];
}
static void sc_ep(SaoCab *c, int n) { c->frac += (uint64_t)32768 * (uint64_t)n; }
static void sc_reset(SaoCab *c) { c->frac &= 32767; }
static uint32_t sc_bits(const SaoCab *c) { return (uint32_t)(c->frac >> 15); }

/* codeSaoMaxUvlc, TEncSbac.cpp:1545-1572 */
static void code_max_uvlc(SaoCab *c, int code, int maxSymbol)
{
  if (maxSymbol == 0) return;
  if (code == 0) sc_ep(c, 1);
  else sc_ep(c, 1 + (code - 1) + (maxSymbol > code ? 1 : 0));
}
/* codeSAOOffsetParam, TEncSbac.cpp:1602-1677 */
static void code_offset_param(SaoCab *c, int comp, const HmoSaoOffset *p, int enabled)
{
  if (!enabled) return;
  const int first = comp != 2;                                 /* first component of its channel type: Y, Cb */
  if (first) {
    const int sym = p->mode == SAO_OFF ? 0 : (p->type == SAO_BO ? 1 : 2);
    if (sym == 0) sc_bin(c, 0, 1);
    else { sc_bin(c, 1, 1); sc_ep(c, 1); }
  }
  if (p->mode == SAO_NEW) {
    int off[4], k = 0;
    const int n = p->type == SAO_BO ? 4 : 5;
    for (int i = 0; i < n; i++) {
      if (p->type != SAO_BO && i == 2) continue;
      off[k++] = p->offset[p->type == SAO_BO ? (p->aux + i) % 32 : i];
    }
    for (int i = 0; i < 4; i++) code_max_uvlc(c, abs(off[i]), SAO_MAXQ);
    if (p->type == SAO_BO) {
      for (int i = 0; i < 4; i++) if (off[i] != 0) sc_ep(c, 1);
      sc_ep(c, 5);                                             /* sao_band_position */
    } else if (first) sc_ep(c, 2);                             /* sao_eo_class */
  }
}
/* codeSAOBlkParam, TEncSbac.cpp:1679-1714 */
static void code_blk_param(SaoCab *c, const HmoSaoBlk *b, const int *enabled, int leftAvail, int aboveAvail, int onlyMerge)
{
  int isLeft = 0, isAbove = 0;
  if (leftAvail) { isLeft = b->c[0].mode == SAO_MERGE && b->c[0].type == 0; sc_bin(c, isLeft, 0); }
  if (aboveAvail && !isLeft) { isAbove = b->c[0].mode == SAO_MERGE && b->c[0].type == 1; sc_bin(c, isAbove, 0); }
  if (onlyMerge) return;
  if (!isLeft && !isAbove) for (int comp = 0; comp < 3; comp++) code_offset_param(c, comp, &b->c[comp], enabled[comp]);
}

/* ---- statistics --------------------------------------------------------------------------------------------------- */
static int sgn(int v) { return (v > 0) - (v < 0); }
/* getBlkStats of one CTU block of one component, sample by sample: the reference's line-buffered loops visit exactly the
 * samples below (TEncSampleAdaptiveOffset.cpp:955-1370 with isCalculatePreDeblockSamples false).  Availability: L/R/A/B =
 * picture boundary; the right 5 (3) columns and bottom 4 (2) rows of a luma (chroma) block with a right / below
 * neighbour are left out (m_skipLinesR/B, :160-163). */
static void blk_stats(HmoSaoStat *st, const uint8_t *src, const uint8_t *org, int stride, int w, int h, int comp, int L, int R, int A, int B)
{
  memset(st, 0, sizeof(*st));
  const int skipR = comp ? 3 : 5, skipB = comp ? 2 : 4, AL = A && L;
  const int sx = L ? 0 : 1, ex = R ? w - skipR : w - 1, exFull = R ? w - skipR : w;
  const int eyFull = B ? h - skipB : h, ey = B ? h - skipB : h - 1;
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
    const uint8_t *s = src + y * stride + x;
    const int c = s[0], d = org[y * stride + x] - c;
    if (x < exFull && y < eyFull) { st->diff[SAO_BO][c >> 3] += d; st->count[SAO_BO][c >> 3]++; }
    if (x >= sx && x < ex && y < eyFull) { const int k = 2 + sgn(c - s[-1]) + sgn(c - s[1]); st->diff[0][k] += d; st->count[0][k]++; }
    if (x < exFull && y >= (A ? 0 : 1) && y < ey) { const int k = 2 + sgn(c - s[-stride]) + sgn(c - s[stride]); st->diff[1][k] += d; st->count[1][k]++; }
    if (y == 0 ? (A && x >= (AL ? 0 : 1) && x < ex) : (y < ey && x >= sx && x < ex)) {
      const int k = 2 + sgn(c - s[-stride - 1]) + sgn(c - s[stride + 1]); st->diff[2][k] += d; st->count[2][k]++;
    }
    if (y == 0 ? (A && x >= sx && x < ex) : (y < ey && x >= sx && x < ex)) {
      const int k = 2 + sgn(c - s[-stride + 1]) + sgn(c - s[stride - 1]); st->diff[3][k] += d; st->count[3][k]++;
    }
  }
}

/* ---- offsets -------------------------------------------------------------------------------------------------------- */
static int64_t est_dist(int64_t count, int64_t off, int64_t diff) { return count * off * off - diff * off * 2; }
/* estIterOffset, :441-472 */
static int est_iter_offset(int type, double lambda, int offIn, int64_t count, int64_t diff, int64_t *bestDist, double *bestCost)
{
  int it = offIn, out = 0;
  double minCost = lambda;
  while (it != 0) {
    int64_t rate = type == SAO_BO ? abs(it) + 2 : abs(it) + 1;
    if (abs(it) == SAO_MAXQ) rate--;
    const int64_t dist = est_dist(count, it, diff);
    const double cost = (double)dist + lambda * (double)rate;
    if (cost < minCost) { minCost = cost; out = it; *bestDist = dist; *bestCost = cost; }
    it = it > 0 ? it - 1 : it + 1;
  }
  return out;
}
/* deriveOffsets, :474-591 */
static void derive_offsets(int type, const HmoSaoStat *st, double lambda, int *q, int *aux)
{
  memset(q, 0, sizeof(int) * 32);
  const int n = type == SAO_BO ? 32 : 5;
  for (int k = 0; k < n; k++) {
    if (type != SAO_BO && k == 2) continue;
    if (st->count[type][k] == 0) continue;
    const double x = (double)st->diff[type][k] / (double)st->count[type][k];
    int v = x >= 0 ? (int)(x + 0.5) : (int)(x - 0.5);
    q[k] = v < -SAO_MAXQ ? -SAO_MAXQ : (v > SAO_MAXQ ? SAO_MAXQ : v);
  }
  if (type != SAO_BO) {
    int64_t dd; double cc;
    for (int k = 0; k < 5; k++) {
      if (k < 2 && q[k] < 0) q[k] = 0;
      if (k > 2 && q[k] > 0) q[k] = 0;
      if (q[k] != 0) q[k] = est_iter_offset(type, lambda, q[k], st->count[type][k], st->diff[type][k], &dd, &cc);
    }
    *aux = 0;
  } else {
    int64_t dist[32]; double cost[32];
    memset(dist, 0, sizeof(dist));
    for (int k = 0; k < 32; k++) {
      cost[k] = lambda;
      if (q[k] != 0) q[k] = est_iter_offset(type, lambda, q[k], st->count[type][k], st->diff[type][k], &dist[k], &cost[k]);
    }
    double minCost = 1.7e+308;
    for (int band = 0; band < 32 - 4 + 1; band++) {
      double c = cost[band]; c += cost[band + 1]; c += cost[band + 2]; c += cost[band + 3];
      if (c < minCost) { minCost = c; *aux = band; }
    }
    int keep[32]; memset(keep, 0, sizeof(keep));
    for (int i = 0; i < 4; i++) keep[(*aux + i) % 32] = q[(*aux + i) % 32];
    memcpy(q, keep, sizeof(keep));
  }
}
/* getDistortion, :397-433 (offsets already de-quantised: offset step 1 at 8 bit) */
static int64_t get_distortion(int type, int aux, const int *off, const HmoSaoStat *st)
{
  int64_t d = 0;
  if (type != SAO_BO) for (int k = 0; k < 5; k++) d += est_dist(st->count[type][k], off[k], st->diff[type][k]);
  else for (int i = aux; i < aux + 4; i++) { const int b = i % 32; d += est_dist(st->count[type][b], off[b], st->diff[type][b]); }
  return d;
}
/* invertQuantOffsets, TComSampleAdaptiveOffset.cpp:171-193 (offset step 1): keeps the four band offsets / the EO offsets */
static void invert_quant(int type, int aux, int *dst, const int *src)
{
  int tmp[32]; memcpy(tmp, src, sizeof(tmp)); memset(dst, 0, sizeof(tmp));
  if (type == SAO_BO) for (int i = 0; i < 4; i++) dst[(aux + i) % 32] = tmp[(aux + i) % 32];
  else for (int i = 0; i < 5; i++) dst[i] = tmp[i];
}

/* ---- deriveModeNewRDO, :593-734 : coders[0] = BLK_CUR (in), returns the cost; *goon ends as the coder after the CTU ---- */
static double derive_mode_new(const HmoSaoStat *st /*[3]*/, const double *lambda, const int *enabled, int leftAvail, int aboveAvail,
                              const SaoCab *cur, SaoCab *goon, HmoSaoBlk *mode)
{
  SaoCab mid, temp;
  int64_t dist[3], modeDist[3] = { 0, 0, 0 };
  HmoSaoOffset test[3];
  int inv[32];
  memset(test, 0, sizeof(test));
  memset(mode, 0, sizeof(*mode));
  mode->c[0].mode = SAO_OFF;
  *goon = *cur;
  code_blk_param(goon, mode, enabled, leftAvail, aboveAvail, 1);
  mid = *goon;
  {                                                            /* luma */
    mode->c[0].mode = SAO_OFF;
    sc_reset(goon);
    code_offset_param(goon, 0, &mode->c[0], enabled[0]);
    double minCost = lambda[0] * (double)sc_bits(goon);
    temp = *goon;
    if (enabled[0]) for (int type = 0; type < SAO_NTYPES; type++) {
      test[0].mode = SAO_NEW; test[0].type = type;
      derive_offsets(type, &st[0], lambda[0], test[0].offset, &test[0].aux);
      invert_quant(type, test[0].aux, inv, test[0].offset);
      dist[0] = get_distortion(type, test[0].aux, inv, &st[0]);
      *goon = mid; sc_reset(goon);
      code_offset_param(goon, 0, &test[0], enabled[0]);
      const double cost = (double)dist[0] + lambda[0] * (double)(int)sc_bits(goon);
      if (cost < minCost) { minCost = cost; modeDist[0] = dist[0]; mode->c[0] = test[0]; temp = *goon; }
    }
    *goon = temp; mid = *goon;
  }
  {                                                            /* chroma: Cb and Cr share the type */
    double cost = 0; uint32_t prev = 0;
    sc_reset(goon);
    for (int comp = 1; comp < 3; comp++) {
      mode->c[comp].mode = SAO_OFF; modeDist[comp] = 0;
      code_offset_param(goon, comp, &mode->c[comp], enabled[comp]);
      const uint32_t now = sc_bits(goon);
      cost += lambda[comp] * (double)(now - prev); prev = now;
    }
    double minCost = cost;
    for (int type = 0; type < SAO_NTYPES; type++) {
      *goon = mid; sc_reset(goon); prev = 0; cost = 0;
      for (int comp = 1; comp < 3; comp++) {
        if (!enabled[comp]) { test[comp].mode = SAO_OFF; dist[comp] = 0; continue; }
        test[comp].mode = SAO_NEW; test[comp].type = type;
        derive_offsets(type, &st[comp], lambda[comp], test[comp].offset, &test[comp].aux);
        invert_quant(type, test[comp].aux, inv, test[comp].offset);
        dist[comp] = get_distortion(type, test[comp].aux, inv, &st[comp]);
        code_offset_param(goon, comp, &test[comp], enabled[comp]);
        const uint32_t now = sc_bits(goon);
        cost += (double)dist[comp] + lambda[comp] * (double)(now - prev); prev = now;
      }
      if (cost < minCost) { minCost = cost; for (int comp = 1; comp < 3; comp++) { modeDist[comp] = dist[comp]; mode->c[comp] = test[comp]; } }
    }
  }
  double norm = 0;
  for (int comp = 0; comp < 3; comp++) norm += (double)modeDist[comp] / lambda[comp];
  *goon = *cur; sc_reset(goon);
  code_blk_param(goon, mode, enabled, leftAvail, aboveAvail, 0);
  norm += (double)sc_bits(goon);
  return norm;
}

/* deriveModeMergeRDO, :736-788 ; merge[k] = reconstructed parameters of the left (0) / above (1) CTU or NULL */
static double derive_mode_merge(const HmoSaoStat *st, const double *lambda, const int *enabled, const HmoSaoBlk *const *merge,
                                const SaoCab *cur, SaoCab *goon, HmoSaoBlk *mode)
{
  double best = 1.7e+308;
  SaoCab temp = *goon;
  for (int mt = 0; mt < 2; mt++) {
    if (!merge[mt]) continue;
    HmoSaoBlk test = *merge[mt];
    double normDist = 0;
    for (int comp = 0; comp < 3; comp++) {
      test.c[comp].mode = SAO_MERGE; test.c[comp].type = mt;
      const HmoSaoOffset *m = &merge[mt]->c[comp];
      if (m->mode != SAO_OFF) normDist += (double)get_distortion(m->type, m->aux, m->offset, &st[comp]) / lambda[comp];
    }
    *goon = *cur; sc_reset(goon);
    code_blk_param(goon, &test, enabled, merge[0] != NULL, merge[1] != NULL, 0);
    const double cost = normDist + (double)(int)sc_bits(goon);
    if (cost < best) { best = cost; *mode = test; temp = *goon; }
  }
  *goon = temp;
  return best;
}

/* offsetBlock, TComSampleAdaptiveOffset.cpp:317-556, sample by sample (src = the deblocked picture, res = output) */
static void offset_block(int type, const int *offset, const uint8_t *src, uint8_t *res, int stride, int w, int h,
                         int L, int R, int A, int B)
{
  const int AL = A && L, AR = A && R, BL = B && L, BR = B && R;
  const int sx = L ? 0 : 1, ex = R ? w : w - 1;
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
    const uint8_t *s = src + y * stride + x;
    const int c = s[0];
    int k = -1;
    switch (type) {
    case 0: if (x >= sx && x < ex) k = 2 + sgn(c - s[-1]) + sgn(c - s[1]); break;
    case 1: if (y >= (A ? 0 : 1) && y < (B ? h : h - 1)) k = 2 + sgn(c - s[-stride]) + sgn(c - s[stride]); break;
    case 2: {
      int ok;
      if (y == 0) ok = A && x >= (AL ? 0 : 1) && x < ex;
      else if (y == h - 1) ok = x >= (B ? sx : w - 1) && x < (BR ? w : w - 1);
      else ok = x >= sx && x < ex;
      if (ok) k = 2 + sgn(c - s[-stride - 1]) + sgn(c - s[stride + 1]);
      break; }
    case 3: {
      int ok;
      if (y == 0) ok = x >= (A ? sx : w - 1) && x < (AR ? w : w - 1);
      else if (y == h - 1) ok = B && x >= (BL ? 0 : 1) && x < ex;
      else ok = x >= sx && x < ex;
      if (ok) k = 2 + sgn(c - s[-stride + 1]) + sgn(c - s[stride - 1]);
      break; }
    default: k = c >> 3; break;
    }
    if (k >= 0) { const int v = c + offset[k]; res[y * stride + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
  }
}

void hmo_sao_stats(int width, int height, const uint8_t *const org[3], const uint8_t *const src[3], HmoSaoStat *stats /* [n_ctu][3] */)
{
  const int wc = (width + 63) / 64, hc = (height + 63) / 64;
  for (int a = 0; a < wc * hc; a++) {
    const int cx = a % wc, cy = a / wc, x0 = cx * 64, y0 = cy * 64;
    const int bw = x0 + 64 > width ? width - x0 : 64, bh = y0 + 64 > height ? height - y0 : 64;
    for (int comp = 0; comp < 3; comp++) {
      const int sh = comp ? 1 : 0, stride = width >> sh;
      const size_t o = (size_t)(y0 >> sh) * stride + (x0 >> sh);
      blk_stats(&stats[a * 3 + comp], src[comp] + o, org[comp] + o, stride, bw >> sh, bh >> sh, comp,
                cx > 0, x0 + 64 < width, cy > 0, y0 + 64 < height);
    }
  }
}

/* decideBlkParams + offsetCTU for the whole picture.  rec: in = deblocked picture, out = after SAO.  enabled: the slice-level
 * switches of decidePicParams.  coded[n_ctu]: the parameters as signalled; off_count[comp]: CTUs whose reconstructed mode
 * is OFF (feeds m_saoDisabledRate, :895-917). */
void hmo_sao_picture(int width, int height, int slice_ctus, int qp, int slice_type, const double lambda[3], const int enabled[3],
                     const uint8_t *const org[3], uint8_t *const rec[3], HmoSaoBlk *coded, HmoSaoStat *stats_out, int off_count[3])
{
  hmo_init_tables();
  const int wc = (width + 63) / 64, hc = (height + 63) / 64, n = wc * hc;
  uint8_t *src[3];
  for (int comp = 0; comp < 3; comp++) {
    const size_t sz = (size_t)(width >> (comp ? 1 : 0)) * (size_t)(height >> (comp ? 1 : 0));
    src[comp] = (uint8_t *)malloc(sz); memcpy(src[comp], rec[comp], sz);
  }
  HmoSaoStat *stats = stats_out ? stats_out : (HmoSaoStat *)malloc(sizeof(HmoSaoStat) * (size_t)n * 3);
  hmo_sao_stats(width, height, org, (const uint8_t *const *)src, stats);
  HmoSaoBlk *recon = (HmoSaoBlk *)calloc((size_t)n, sizeof(HmoSaoBlk));
  SaoCab goon;                                                 /* initRDOCabacCoder: resetEntropy of the slice, :245-253 */
  static const int init_type[3] = { 160, 185, 200 };           /* INIT_SAO_TYPE_IDX[B, P, I], ContextTables.h:452-458 */
  goon.ctx[0] = ctx_from_init(153, qp);                        /* INIT_SAO_MERGE_FLAG, :444-450 */
  goon.ctx[1] = ctx_from_init(init_type[slice_type == HMO_SLICE_I ? 2 : 1], qp);
  goon.frac = 0;
  const int allOff = !enabled[0] && !enabled[1] && !enabled[2];
  for (int a = 0; a < n; a++) {
    if (allOff) { memset(&coded[a], 0, sizeof(coded[a])); continue; }
    const SaoCab cur = goon;
    SaoCab next = goon;
    const int cx = a % wc, cy = a / wc;
    const int sliceStart = slice_ctus > 0 ? (a / slice_ctus) * slice_ctus : 0;
    const HmoSaoBlk *merge[2] = { NULL, NULL };
    if (cy > 0 && a - wc >= sliceStart) merge[1] = &recon[a - wc];
    if (cx > 0 && a - 1 >= sliceStart) merge[0] = &recon[a - 1];
    double minCost = 1.7e+308;
    HmoSaoBlk mode;
    double cost = derive_mode_new(&stats[a * 3], lambda, enabled, merge[0] != NULL, merge[1] != NULL, &cur, &goon, &mode);
    if (cost < minCost) { minCost = cost; coded[a] = mode; next = goon; }
    cost = derive_mode_merge(&stats[a * 3], lambda, enabled, merge, &cur, &goon, &mode);
    if (cost < minCost) { minCost = cost; coded[a] = mode; next = goon; }
    goon = next;
    recon[a] = coded[a];                                       /* reconstructBlkSAOParam, TComSampleAdaptiveOffset.cpp:252-288 */
    for (int comp = 0; comp < 3; comp++) {
      HmoSaoOffset *p = &recon[a].c[comp];
      if (p->mode == SAO_NEW) invert_quant(p->type, p->aux, p->offset, p->offset);
      else if (p->mode == SAO_MERGE) *p = merge[p->type]->c[comp];
    }
    const int x0 = cx * 64, y0 = cy * 64;
    const int bw = x0 + 64 > width ? width - x0 : 64, bh = y0 + 64 > height ? height - y0 : 64;
    for (int comp = 0; comp < 3; comp++) {
      const HmoSaoOffset *p = &recon[a].c[comp];
      if (p->mode == SAO_OFF) continue;
      const int sh = comp ? 1 : 0, stride = width >> sh;
      const size_t o = (size_t)(y0 >> sh) * stride + (x0 >> sh);
      offset_block(p->type, p->offset, src[comp] + o, rec[comp] + o, stride, bw >> sh, bh >> sh, cx > 0, x0 + 64 < width, cy > 0, y0 + 64 < height);
    }
  }
  if (off_count) for (int comp = 0; comp < 3; comp++) { off_count[comp] = 0; for (int a = 0; a < n; a++) off_count[comp] += recon[a].c[comp].mode == SAO_OFF; }
  free(recon);
  if (!stats_out) free(stats);
  for (int comp = 0; comp < 3; comp++) free(src[comp]);
}
