/*
 * fcu_sao.h -- sample adaptive offset of decided, deblocked pictures on the device
 * (TEncSampleAdaptiveOffset::SAOProcess, Lib/TLibEncoder/TEncSampleAdaptiveOffset.cpp:257-287, called per picture at
 * TEncGOP.cpp:1434; offset pass TComSampleAdaptiveOffset::offsetCTU, Lib/TLibCommon/TComSampleAdaptiveOffset.cpp:558-616).
 * Included by fcu_kernels.hip only (and by the test-only CPU build tests/emu/sao_emu.cpp).
 *
 * Four launches per batch of pictures; everything but the third is data parallel:
 *   sao_stats    one workgroup per (CTU, component, picture): getBlkStats (:922-1381) -- every sample is classified for the
 *                four edge directions and the band table, (org - src) and 1 accumulate in an LDS histogram (integer LDS
 *                atomics: order-free, so the sums are exact whatever the interleaving).  HBM-bound: reads org + src once
 *                (the 3x3 neighbourhood comes from L2), writes 1 280 B per block.
 *   sao_cands    one thread per (CTU, component, type, picture): deriveOffsets / estIterOffset (:441-591) and the
 *                distortion of the derived offsets (:397-433) -- they depend on the statistics and lambda only.
 *   sao_decide   one thread per picture, CTUs in coding order: what is left of decideBlkParams (:790-920) is the rate of
 *                each candidate on the CABAC bit counter (two adaptive contexts carried from CTU to CTU), the new / merge
 *                choice, and the reconstruction of merged parameters.
 *   sao_apply    one workgroup per (CTU, component, picture): offsetBlock (:317-556) from the untouched copy of the
 *                deblocked picture into the reconstruction.  HBM-bound: reads src, writes rec.
 * 8-bit 4:2:0, 64x64 CTUs, LFCrossSliceBoundaryFlag 1, no tiles: neighbour availability is the picture boundary
 * (TComPicSym.cpp:357-376); merge candidates stay inside the CTU's slice (TComPic.cpp:138-143).
 * Algorithmic bytes per picture: stats 2 x 1.5 W H read; apply 1.5 W H read + <= 1.5 W H written.
 */
#pragma once

namespace fcu {

enum { SAO_THREADS = 256, SAO_OFF = 0, SAO_NEW = 1, SAO_MERGE = 2, SAO_BO = 4, SAO_NTYPES = 5, SAO_MAXQ = 7,
       SAO_STAT_INTS = SAO_NTYPES * 2 * 32 };             /* per (CTU, component): [type][diff, count][class] */

struct SaoPic {                                            /* one picture of a batch (device copy) */
  const uint8_t *org[3]; uint8_t *rec[3]; uint8_t *src[3];
  double lambda[3]; int enabled[3]; int slice_type, qp, slice_ctus;
};
struct SaoCand { int32_t aux; int32_t off[5]; long long dist; };   /* EO: off[class]; BO: off[i] of band aux + i */

__device__ static inline int sao_sgn(int v) { return (v > 0) - (v < 0); }

/* ---- statistics: phase 0 clears the histogram, 1 accumulates, 2 writes it out -------------------------------------- */
template <int PHASE>
__device__ static inline void sao_stats_phase(int32_t *hist, const SaoPic *pics, int32_t *stats, int width, int height, int w_ctu, int n_ctu)
{
  const int t = (int)threadIdx.x, a = (int)blockIdx.x, comp = (int)blockIdx.y, pic = (int)blockIdx.z;
  if (PHASE == 0) { for (int i = t; i < SAO_STAT_INTS; i += SAO_THREADS) hist[i] = 0; return; }
  if (PHASE == 2) { int32_t *o = stats + ((size_t)(pic * n_ctu + a) * 3 + comp) * SAO_STAT_INTS; for (int i = t; i < SAO_STAT_INTS; i += SAO_THREADS) o[i] = hist[i]; return; }
  const SaoPic &P = pics[pic];
  const int sh = comp ? 1 : 0, cx = a % w_ctu, cy = a / w_ctu, x0 = cx * 64, y0 = cy * 64;
  const int bw = (x0 + 64 > width ? width - x0 : 64) >> sh, bh = (y0 + 64 > height ? height - y0 : 64) >> sh, stride = width >> sh;
  const int L = cx > 0, R = x0 + 64 < width, A = cy > 0, B = y0 + 64 < height, AL = A && L;
  const int skipR = comp ? 3 : 5, skipB = comp ? 2 : 4;
  const int sx = L ? 0 : 1, ex = R ? bw - skipR : bw - 1, exFull = R ? bw - skipR : bw;
  const int eyFull = B ? bh - skipB : bh, ey = B ? bh - skipB : bh - 1;
  const size_t o0 = (size_t)(y0 >> sh) * stride + (x0 >> sh);
  const uint8_t *src = P.src[comp] + o0, *org = P.org[comp] + o0;
  for (int i = t; i < bw * bh; i += SAO_THREADS) {
    const int y = i / bw, x = i - y * bw;
    const uint8_t *s = src + (size_t)y * stride + x;
    const int c = s[0], d = (int)org[(size_t)y * stride + x] - c;
#define SAO_ADD(type, k) do { atomicAdd(&hist[((type) * 2) * 32 + (k)], d); atomicAdd(&hist[((type) * 2 + 1) * 32 + (k)], 1); } while (0)
    if (x < exFull && y < eyFull) SAO_ADD(SAO_BO, c >> 3);
    if (x >= sx && x < ex && y < eyFull) SAO_ADD(0, 2 + sao_sgn(c - s[-1]) + sao_sgn(c - s[1]));
    if (x < exFull && y >= (A ? 0 : 1) && y < ey) SAO_ADD(1, 2 + sao_sgn(c - s[-stride]) + sao_sgn(c - s[stride]));
    if (y == 0 ? (A && x >= (AL ? 0 : 1) && x < ex) : (y < ey && x >= sx && x < ex)) SAO_ADD(2, 2 + sao_sgn(c - s[-stride - 1]) + sao_sgn(c - s[stride + 1]));
    if (y == 0 ? (A && x >= sx && x < ex) : (y < ey && x >= sx && x < ex)) SAO_ADD(3, 2 + sao_sgn(c - s[-stride + 1]) + sao_sgn(c - s[stride - 1]));
#undef SAO_ADD
  }
}

/* ---- offsets of one (CTU, component, type) ------------------------------------------------------------------------- */
__device__ static inline long long sao_est_dist(long long count, long long off, long long diff) { return count * off * off - diff * off * 2; }
/* estIterOffset, :441-472 */
__device__ static inline int sao_iter_offset(int type, double lambda, int offIn, long long count, long long diff, long long *bestDist, double *bestCost)
{
  int it = offIn, out = 0;
  double minCost = lambda;
  while (it != 0) {
    const int ab = it < 0 ? -it : it;
    long long rate = type == SAO_BO ? ab + 2 : ab + 1;
    if (ab == SAO_MAXQ) rate--;
    const long long dist = sao_est_dist(count, it, diff);
    const double cost = (double)dist + lambda * (double)rate;
    if (cost < minCost) { minCost = cost; out = it; *bestDist = dist; *bestCost = cost; }
    it = it > 0 ? it - 1 : it + 1;
  }
  return out;
}
__device__ static inline int sao_initial_offset(int diff, int count)
{
  if (count == 0) return 0;
  const double x = (double)diff / (double)count;
  const int v = x >= 0 ? (int)(x + 0.5) : (int)(x - 0.5);    /* xRoundIbdi at 8 bit, :54-57 */
  return v < -SAO_MAXQ ? -SAO_MAXQ : (v > SAO_MAXQ ? SAO_MAXQ : v);
}
/* deriveOffsets (:474-591) + getDistortion (:397-433) of one type from its statistics */
__device__ static inline void sao_cand_one(int type, const int32_t *st /* [2][32] of the type */, double lambda, SaoCand *out)
{
  const int32_t *diff = st, *count = st + 32;
  long long dist = 0;
  if (type != SAO_BO) {
    for (int k = 0; k < 5; k++) {
      int q = k == 2 ? 0 : sao_initial_offset(diff[k], count[k]);
      if (k < 2 && q < 0) q = 0;
      if (k > 2 && q > 0) q = 0;
      long long dd; double cc;
      if (q != 0) q = sao_iter_offset(type, lambda, q, count[k], diff[k], &dd, &cc);
      out->off[k] = q;
      dist += sao_est_dist(count[k], q, diff[k]);
    }
    out->aux = 0;
  } else {
    /* the four-band window with the least cost: costs of the 32 bands first (a band's cost is lambda when its offset is 0) */
    double cost[32]; int8_t q[32];
    for (int k = 0; k < 32; k++) {
      cost[k] = lambda;
      int v = sao_initial_offset(diff[k], count[k]);
      long long dd = 0;
      if (v != 0) v = sao_iter_offset(type, lambda, v, count[k], diff[k], &dd, &cost[k]);
      q[k] = (int8_t)v;
    }
    double minCost = 1.7e+308; int band0 = 0;
    for (int band = 0; band < 32 - 4 + 1; band++) {
      double c = cost[band]; c += cost[band + 1]; c += cost[band + 2]; c += cost[band + 3];
      if (c < minCost) { minCost = c; band0 = band; }
    }
    out->aux = band0; out->off[4] = 0;
    for (int i = 0; i < 4; i++) { const int b = band0 + i; out->off[i] = q[b]; dist += sao_est_dist(count[b], q[b], diff[b]); }
  }
  out->dist = dist;
}
__device__ static inline void sao_cands_thread(const SaoPic *pics, const int32_t *stats, SaoCand *cands, int n_ctu, int n_pics)
{
  const long long id = (long long)blockIdx.x * SAO_THREADS + threadIdx.x;
  if (id >= (long long)n_pics * n_ctu * 15) return;
  const int type = (int)(id % 5), comp = (int)((id / 5) % 3);
  const long long blk = id / 15;                            /* pic * n_ctu + ctu */
  const int pic = (int)(blk / n_ctu);
  sao_cand_one(type, stats + ((size_t)blk * 3 + comp) * SAO_STAT_INTS + type * 64, pics[pic].lambda[comp], &cands[id]);
}

/* ---- decision ------------------------------------------------------------------------------------------------------ */
struct SaoCab { uint8_t ctx[2]; unsigned long long frac; };   /* ctx[0] sao_merge_flag, ctx[1] sao_type_idx; the Q15 counter */
__device__ static inline uint8_t sao_ctx_init(int iv, int qp)
{
  qp = qp < 0 ? 0 : (qp > 51 ? 51 : qp);
  const int slope = (iv >> 4) * 5 - 45, offset = ((iv & 15) << 3) - 16;
  int st = ((slope * qp) >> 4) + offset; st = st < 1 ? 1 : (st > 126 ? 126 : st);
  const int mps = st >= 64;
  return (uint8_t)(((mps ? (st - 64) : (63 - st)) << 1) + mps);
}
__device__ static inline void sao_bin(SaoCab &c, int bin, int k) { const uint32_t e = k_bin[c.ctx[k] * 2 + bin]; c.frac += e >> 8; c.ctx[k] = (uint8_t)e; }
__device__ static inline void sao_ep(SaoCab &c, int n) { c.frac += (unsigned long long)32768 * (unsigned long long)n; }
__device__ static inline void sao_reset(SaoCab &c) { c.frac &= 32767; }
__device__ static inline uint32_t sao_bits(const SaoCab &c) { return (uint32_t)(c.frac >> 15); }
/* codeSAOOffsetParam, TEncSbac.cpp:1602-1677 */
__device__ static inline void sao_code_offset(SaoCab &c, int comp, const fcu_sao_offset &p, int enabled)
{
  if (!enabled) return;
  const int first = comp != 2;
  if (first) {
    if (p.mode == SAO_OFF) sao_bin(c, 0, 1);
    else { sao_bin(c, 1, 1); sao_ep(c, 1); }
  }
  if (p.mode == SAO_NEW) {
    int nz = 0;
    for (int i = 0; i < 4; i++) {
      const int v = p.type == SAO_BO ? p.offset[(p.band + i) & 31] : p.offset[i < 2 ? i : i + 1];
      const int ab = v < 0 ? -v : v;
      sao_ep(c, ab == 0 ? 1 : 1 + (ab - 1) + (SAO_MAXQ > ab ? 1 : 0));      /* codeSaoMaxUvlc, :1545-1572 */
      nz += v != 0;
    }
    if (p.type == SAO_BO) sao_ep(c, nz + 5);              /* signs + sao_band_position */
    else if (first) sao_ep(c, 2);                         /* sao_eo_class */
  }
}
/* codeSAOBlkParam, TEncSbac.cpp:1679-1714 */
__device__ static inline void sao_code_blk(SaoCab &c, const fcu_sao_ctu &b, const int *enabled, int leftAvail, int aboveAvail, int onlyMerge)
{
  int isLeft = 0, isAbove = 0;
  if (leftAvail) { isLeft = b.c[0].mode == SAO_MERGE && b.c[0].type == 0; sao_bin(c, isLeft, 0); }
  if (aboveAvail && !isLeft) { isAbove = b.c[0].mode == SAO_MERGE && b.c[0].type == 1; sao_bin(c, isAbove, 0); }
  if (onlyMerge) return;
  if (!isLeft && !isAbove) for (int comp = 0; comp < 3; comp++) sao_code_offset(c, comp, b.c[comp], enabled[comp]);
}
__device__ static inline void sao_offset_from_cand(fcu_sao_offset &o, int type, const SaoCand &cd)
{
  o.mode = SAO_NEW; o.type = (int8_t)type; o.band = (int8_t)cd.aux; o.pad = 0;
  for (int k = 0; k < 32; k++) o.offset[k] = 0;
  if (type == SAO_BO) for (int i = 0; i < 4; i++) o.offset[(cd.aux + i) & 31] = (int8_t)cd.off[i];
  else for (int k = 0; k < 5; k++) o.offset[k] = (int8_t)cd.off[k];
}
__device__ static inline void sao_offset_clear(fcu_sao_offset &o) { o.mode = SAO_OFF; o.type = 0; o.band = 0; o.pad = 0; for (int k = 0; k < 32; k++) o.offset[k] = 0; }
/* getDistortion of already reconstructed offsets against this CTU's statistics (merge candidates, :760-766) */
__device__ static inline long long sao_merge_dist(const fcu_sao_offset &m, const int32_t *st /* [5][2][32] */)
{
  const int32_t *diff = st + m.type * 64, *count = diff + 32;
  long long d = 0;
  if (m.type != SAO_BO) for (int k = 0; k < 5; k++) d += sao_est_dist(count[k], m.offset[k], diff[k]);
  else for (int i = 0; i < 4; i++) { const int b = (m.band + i) & 31; d += sao_est_dist(count[b], m.offset[b], diff[b]); }
  return d;
}

/* decideBlkParams (:790-920) of one picture: coded[] = parameters as signalled, recon[] = after reconstructBlkSAOParam
 * (TComSampleAdaptiveOffset.cpp:252-288), off_count[comp] = CTUs whose reconstructed mode is OFF (-> m_saoDisabledRate) */
__device__ static inline void sao_decide_picture(const SaoPic &P, const int32_t *stats, const SaoCand *cands, fcu_sao_ctu *coded, fcu_sao_ctu *recon,
                                                 int32_t *off_count, int w_ctu, int n_ctu)
{
  SaoCab goon;
  goon.ctx[0] = sao_ctx_init(153, P.qp);                                       /* INIT_SAO_MERGE_FLAG, ContextTables.h:444-450 */
  goon.ctx[1] = sao_ctx_init(P.slice_type == SLICE_I ? 200 : 185, P.qp);       /* INIT_SAO_TYPE_IDX [I] / [P], :452-458 */
  goon.frac = 0;
  const int en[3] = { P.enabled[0], P.enabled[1], P.enabled[2] };
  const int allOff = !en[0] && !en[1] && !en[2];
  int nOff[3] = { 0, 0, 0 };
  for (int a = 0; a < n_ctu; a++) {
    fcu_sao_ctu &out = coded[a];
    if (allOff) { for (int c = 0; c < 3; c++) { sao_offset_clear(out.c[c]); sao_offset_clear(recon[a].c[c]); nOff[c]++; } continue; }
    const SaoCab cur = goon;
    const int cx = a % w_ctu, cy = a / w_ctu;
    const int sliceStart = P.slice_ctus > 0 ? (a / P.slice_ctus) * P.slice_ctus : 0;
    const int aboveAvail = cy > 0 && a - w_ctu >= sliceStart, leftAvail = cx > 0 && a - 1 >= sliceStart;
    const int32_t *st = stats + (size_t)a * 3 * SAO_STAT_INTS;
    const SaoCand *cd = cands + (size_t)a * 15;
    /* ---- deriveModeNewRDO, :593-734 */
    fcu_sao_ctu mode;
    for (int c = 0; c < 3; c++) sao_offset_clear(mode.c[c]);
    long long modeDist[3] = { 0, 0, 0 };
    SaoCab mid = cur, temp;
    sao_code_blk(mid, mode, en, leftAvail, aboveAvail, 1);
    {
      goon = mid; sao_reset(goon);
      sao_code_offset(goon, 0, mode.c[0], en[0]);
      double minCost = P.lambda[0] * (double)sao_bits(goon);
      temp = goon;
      if (en[0]) for (int type = 0; type < SAO_NTYPES; type++) {
        fcu_sao_offset test; sao_offset_from_cand(test, type, cd[type]);
        goon = mid; sao_reset(goon);
        sao_code_offset(goon, 0, test, 1);
        const double cost = (double)cd[type].dist + P.lambda[0] * (double)(int)sao_bits(goon);
        if (cost < minCost) { minCost = cost; modeDist[0] = cd[type].dist; mode.c[0] = test; temp = goon; }
      }
      mid = temp;
    }
    {
      goon = mid; sao_reset(goon);
      double cost = 0; uint32_t prev = 0;
      for (int c = 1; c < 3; c++) { sao_code_offset(goon, c, mode.c[c], en[c]); const uint32_t now = sao_bits(goon); cost += P.lambda[c] * (double)(now - prev); prev = now; }
      double minCost = cost;
      for (int type = 0; type < SAO_NTYPES; type++) {
        goon = mid; sao_reset(goon); prev = 0; cost = 0;
        fcu_sao_offset test[3]; long long dist[3] = { 0, 0, 0 };
        for (int c = 1; c < 3; c++) {
          if (!en[c]) { sao_offset_clear(test[c]); continue; }
          sao_offset_from_cand(test[c], type, cd[c * 5 + type]);
          dist[c] = cd[c * 5 + type].dist;
          sao_code_offset(goon, c, test[c], 1);
          const uint32_t now = sao_bits(goon);
          cost += (double)dist[c] + P.lambda[c] * (double)(now - prev); prev = now;
        }
        if (cost < minCost) { minCost = cost; for (int c = 1; c < 3; c++) { modeDist[c] = dist[c]; mode.c[c] = test[c]; } }
      }
    }
    double normNew = 0;
    for (int c = 0; c < 3; c++) normNew += (double)modeDist[c] / P.lambda[c];
    goon = cur; sao_reset(goon);
    sao_code_blk(goon, mode, en, leftAvail, aboveAvail, 0);
    normNew += (double)sao_bits(goon);
    double minCost = 1.7e+308;
    SaoCab next = cur;
    if (normNew < minCost) { minCost = normNew; out = mode; next = goon; }
    /* ---- deriveModeMergeRDO, :736-788 */
    {
      double best = 1.7e+308; fcu_sao_ctu bestMode = mode; SaoCab bestCab = goon;
      for (int mt = 0; mt < 2; mt++) {
        if (!(mt == 0 ? leftAvail : aboveAvail)) continue;
        const fcu_sao_ctu &m = recon[mt == 0 ? a - 1 : a - w_ctu];
        fcu_sao_ctu test = m;
        double normDist = 0;
        for (int c = 0; c < 3; c++) {
          test.c[c].mode = SAO_MERGE; test.c[c].type = (int8_t)mt;
          if (m.c[c].mode != SAO_OFF) normDist += (double)sao_merge_dist(m.c[c], st + c * SAO_STAT_INTS) / P.lambda[c];
        }
        goon = cur; sao_reset(goon);
        sao_code_blk(goon, test, en, leftAvail, aboveAvail, 0);
        const double cost = normDist + (double)(int)sao_bits(goon);
        if (cost < best) { best = cost; bestMode = test; bestCab = goon; }
      }
      if (best < minCost) { minCost = best; out = bestMode; next = bestCab; }
    }
    goon = next;
    /* ---- reconstructBlkSAOParam */
    fcu_sao_ctu r = out;
    for (int c = 0; c < 3; c++) if (r.c[c].mode == SAO_MERGE) r.c[c] = recon[r.c[c].type == 0 ? a - 1 : a - w_ctu].c[c];
    recon[a] = r;
    for (int c = 0; c < 3; c++) nOff[c] += r.c[c].mode == SAO_OFF;
  }
  for (int c = 0; c < 3; c++) off_count[c] = nOff[c];
}

/* ---- offsetBlock, TComSampleAdaptiveOffset.cpp:317-556 -------------------------------------------------------------- */
__device__ static inline void sao_apply_block(const SaoPic *pics, const fcu_sao_ctu *recon, int width, int height, int w_ctu, int n_ctu)
{
  const int t = (int)threadIdx.x, a = (int)blockIdx.x, comp = (int)blockIdx.y, pic = (int)blockIdx.z;
  const fcu_sao_offset &p = recon[(size_t)pic * n_ctu + a].c[comp];
  if (p.mode == SAO_OFF) return;
  const SaoPic &P = pics[pic];
  const int sh = comp ? 1 : 0, cx = a % w_ctu, cy = a / w_ctu, x0 = cx * 64, y0 = cy * 64;
  const int w = (x0 + 64 > width ? width - x0 : 64) >> sh, h = (y0 + 64 > height ? height - y0 : 64) >> sh, stride = width >> sh;
  const int L = cx > 0, R = x0 + 64 < width, A = cy > 0, B = y0 + 64 < height, AL = A && L, AR = A && R, BL = B && L, BR = B && R;
  const int sx = L ? 0 : 1, ex = R ? w : w - 1, type = p.type;
  const size_t o0 = (size_t)(y0 >> sh) * stride + (x0 >> sh);
  const uint8_t *src = P.src[comp] + o0; uint8_t *res = P.rec[comp] + o0;
  for (int i = t; i < w * h; i += SAO_THREADS) {
    const int y = i / w, x = i - y * w;
    const uint8_t *s = src + (size_t)y * stride + x;
    const int c = s[0];
    int k = -1;
    if (type == 0) { if (x >= sx && x < ex) k = 2 + sao_sgn(c - s[-1]) + sao_sgn(c - s[1]); }
    else if (type == 1) { if (y >= (A ? 0 : 1) && y < (B ? h : h - 1)) k = 2 + sao_sgn(c - s[-stride]) + sao_sgn(c - s[stride]); }
    else if (type == 2) {
      int ok;
      if (y == 0) ok = A && x >= (AL ? 0 : 1) && x < ex;
      else if (y == h - 1) ok = x >= (B ? sx : w - 1) && x < (BR ? w : w - 1);
      else ok = x >= sx && x < ex;
      if (ok) k = 2 + sao_sgn(c - s[-stride - 1]) + sao_sgn(c - s[stride + 1]);
    } else if (type == 3) {
      int ok;
      if (y == 0) ok = x >= (A ? sx : w - 1) && x < (AR ? w : w - 1);
      else if (y == h - 1) ok = B && x >= (BL ? 0 : 1) && x < ex;
      else ok = x >= sx && x < ex;
      if (ok) k = 2 + sao_sgn(c - s[-stride + 1]) + sao_sgn(c - s[stride - 1]);
    } else k = c >> 3;
    if (k >= 0) { const int v = c + p.offset[k]; res[(size_t)y * stride + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
  }
}

#ifndef FCU_EMU
__global__ void __launch_bounds__(SAO_THREADS) sao_stats(const SaoPic *pics, int32_t *stats, int width, int height, int w_ctu, int n_ctu)
{
  __shared__ int32_t hist[SAO_STAT_INTS];
  sao_stats_phase<0>(hist, pics, stats, width, height, w_ctu, n_ctu);
  __syncthreads();
  sao_stats_phase<1>(hist, pics, stats, width, height, w_ctu, n_ctu);
  __syncthreads();
  sao_stats_phase<2>(hist, pics, stats, width, height, w_ctu, n_ctu);
}
__global__ void __launch_bounds__(SAO_THREADS) sao_cands(const SaoPic *pics, const int32_t *stats, SaoCand *cands, int n_ctu, int n_pics)
{ sao_cands_thread(pics, stats, cands, n_ctu, n_pics); }
__global__ void __launch_bounds__(64) sao_decide(const SaoPic *pics, const int32_t *stats, const SaoCand *cands, fcu_sao_ctu *coded, fcu_sao_ctu *recon,
                                                 int32_t *off_count, int w_ctu, int n_ctu, int n_pics)
{
  const int pic = (int)(blockIdx.x * 64 + threadIdx.x);
  if (pic >= n_pics) return;
  sao_decide_picture(pics[pic], stats + (size_t)pic * n_ctu * 3 * SAO_STAT_INTS, cands + (size_t)pic * n_ctu * 15,
                     coded + (size_t)pic * n_ctu, recon + (size_t)pic * n_ctu, off_count + pic * 3, w_ctu, n_ctu);
}
__global__ void __launch_bounds__(SAO_THREADS) sao_apply(const SaoPic *pics, const fcu_sao_ctu *recon, int width, int height, int w_ctu, int n_ctu)
{ sao_apply_block(pics, recon, width, height, w_ctu, n_ctu); }
#endif

} /* namespace fcu */
