/*
 * hmo_trquant.c -- ORACLE (test infrastructure).  Integer transforms, RDOQ, dequantiser.
 * All floating point is IEEE double with the reference's operation order; build with
 * -ffp-contract=off.
 */
#include "hmo_int.h"
#include <stdlib.h>
#include <string.h>
#include <limits.h>


/* xTrMxN + partialButterfly4/8/16/32 + fastForwardDst, TComTrQuant.cpp:388-915.
 * The butterflies are exact integer factorizations of the matrix products below. */
void hmo_fwd_transform(const int16_t *resi, int stride, int32_t *coef, int log2, int useDst)
{
  hmo_init_tables();
  const int N = 1 << log2;
  const int16_t *T = (useDst && log2 == 2) ? hmo_dst4 : hmo_T[log2 - 2];
  const int s1 = log2 + 8 + 6 - 15, s2 = log2 + 6;
  int32_t tmp[32 * 32];
  for (int y = 0; y < N; y++)
    for (int k = 0; k < N; k++) {
      int32_t s = 0;
      for (int n = 0; n < N; n++) s += T[k * N + n] * resi[y * stride + n];
      tmp[k * N + y] = (s + (1 << (s1 - 1))) >> s1;
    }
  for (int k1 = 0; k1 < N; k1++)
    for (int k2 = 0; k2 < N; k2++) {
      int32_t s = 0;
      for (int y = 0; y < N; y++) s += T[k2 * N + y] * tmp[k1 * N + y];
      coef[k2 * N + k1] = (s + (1 << (s2 - 1))) >> s2;
    }
}

static inline int32_t clip3(int32_t lo, int32_t hi, int32_t v) { return v < lo ? lo : (v > hi ? hi : v); }

/* xITrMxN + partialButterflyInverse* + fastInverseDst, TComTrQuant.cpp:440-987 */
void hmo_inv_transform(const int32_t *coef, int16_t *resi, int stride, int log2, int useDst)
{
  hmo_init_tables();
  const int N = 1 << log2;
  const int16_t *T = (useDst && log2 == 2) ? hmo_dst4 : hmo_T[log2 - 2];
  const int s1 = 7, s2 = 12;
  int32_t tmp[32 * 32];
  for (int kh = 0; kh < N; kh++)
    for (int y = 0; y < N; y++) {
      int32_t s = 0;
      for (int kv = 0; kv < N; kv++) s += T[kv * N + y] * coef[kv * N + kh];
      tmp[kh * N + y] = clip3(-32768, 32767, (s + (1 << (s1 - 1))) >> s1);
    }
  for (int y = 0; y < N; y++)
    for (int x = 0; x < N; x++) {
      int32_t s = 0;
      for (int kh = 0; kh < N; kh++) s += T[kh * N + x] * tmp[kh * N + y];
      resi[y * stride + x] = (int16_t)clip3(-32768, 32767, (s + (1 << (s2 - 1))) >> s2);
    }
}

/* xDeQuant (flat scaling), TComTrQuant.cpp:1242-1352 */
void hmo_dequant(const int32_t *q, int32_t *c, int n, int log2, int qp)
{
  const int per = qp / 6, rem = qp % 6;
  const int tshift = 15 - 8 - log2;
  const int rs = 6 - (tshift + per);
  const int scale = hmo_inv_quant_scales[rem];
  int tbd = 32 + rs - 7; if (tbd > 16) tbd = 16;
  const int32_t imin = -(1 << (tbd - 1)), imax = (1 << (tbd - 1)) - 1;
  if (rs > 0) {
    const int32_t add = 1 << (rs - 1);
    for (int i = 0; i < n; i++) c[i] = clip3(-32768, 32767, (clip3(imin, imax, q[i]) * scale + add) >> rs);
  } else {
    for (int i = 0; i < n; i++) c[i] = clip3(-32768, 32767, (clip3(imin, imax, q[i]) * scale) << (-rs));
  }
}

/* xGetICRate, TComTrQuant.cpp:2807-2881 (no extended precision) */
static int ic_rate(const HmoCabac *c, uint32_t absLevel, int ctxOne, int ctxAbs, uint32_t goRice, uint32_t c1Idx, uint32_t c2Idx)
{
  int rate = 32768;
  uint32_t baseLevel = (c1Idx < 8) ? (2 + (c2Idx < 1)) : 1;
  if (absLevel >= baseLevel) {
    uint32_t symbol = absLevel - baseLevel, length;
    if (symbol < (3u << goRice)) {
      length = symbol >> goRice;
      rate += (int)((length + 1 + goRice) << 15);
    } else {
      length = goRice;
      symbol = symbol - (3u << goRice);
      while (symbol >= (1u << length)) symbol -= (1u << (length++));
      rate += (int)((3 + length + 1 - goRice + length) << 15);
    }
    if (c1Idx < 8) {
      rate += hmo_ctx_bits(c, ctxOne, 1);
      if (c2Idx < 1) rate += hmo_ctx_bits(c, ctxAbs, 1);
    }
  } else if (absLevel == 1) {
    rate += hmo_ctx_bits(c, ctxOne, 0);
  } else if (absLevel == 2) {
    rate += hmo_ctx_bits(c, ctxOne, 1);
    rate += hmo_ctx_bits(c, ctxAbs, 0);
  } else rate = 0;
  return rate;
}

/* xGetCodedLevel, TComTrQuant.cpp:2738-2794 */
static uint32_t coded_level(const HmoCabac *c, double lambda, double *codedCost, double *codedCost0, double *codedCostSig,
                            int32_t levelDouble, uint32_t maxAbsLevel, int ctxSig, int ctxOne, int ctxAbs,
                            uint32_t goRice, uint32_t c1Idx, uint32_t c2Idx, int qbits, double errScale, int last)
{
  double currCostSig = 0;
  uint32_t bestAbs = 0;
  if (!last && maxAbsLevel < 3) {
    *codedCostSig = lambda * (double)hmo_ctx_bits(c, ctxSig, 0);
    *codedCost = *codedCost0 + *codedCostSig;
    if (maxAbsLevel == 0) return bestAbs;
  } else *codedCost = HMO_MAX_DOUBLE;
  if (!last) currCostSig = lambda * (double)hmo_ctx_bits(c, ctxSig, 1);
  uint32_t minAbs = maxAbsLevel > 1 ? maxAbsLevel - 1 : 1;
  for (int a = (int)maxAbsLevel; a >= (int)minAbs; a--) {
    double err = (double)(levelDouble - ((int32_t)a << qbits));
    double cost = err * err * errScale + lambda * (double)ic_rate(c, (uint32_t)a, ctxOne, ctxAbs, goRice, c1Idx, c2Idx);
    cost += currCostSig;
    if (cost < *codedCost) { bestAbs = (uint32_t)a; *codedCost = cost; *codedCostSig = currCostSig; }
  }
  return bestAbs;
}

/* xRateDistOptQuant, TComTrQuant.cpp:2033-2573.  Rate tables (estBit, TEncSbac.cpp:1722-1956)
 * are read straight from the go-on coder's contexts, which estBit snapshots right before.
 * Returns uiAbsSum.  `part` = GetAbsPartIdxTU(compID), `trDepthRel` = GetTransformDepthRel(). */
/* The quantiser without RDOQ: xQuant's else branch (TComTrQuant.cpp:1160-1240) + signBitHidingHDQ (:991-1124).  Levels are
 * (|c| * scale + add) >> qbits with add = 171 (I slice) / 85 (else) << (qbits - 9); uiAbsSum is the sum before sign hiding. */
static int quant_plain(HmoEnc *e, const HmoCU *cu, int comp, const int32_t *src, int32_t *dst, int log2, int part)
{
  const int N = 1 << log2, n2 = N * N;
  const int qp = comp ? e->p.qp_c : e->p.qp, per = qp / 6, rem = qp % 6;
  const int qbits = 14 + per + (15 - 8 - log2), qbits8 = qbits - 8;
  const int64_t add = (int64_t)(e->p.slice_type == HMO_SLICE_I ? 171 : 85) << (qbits - 9);
  const int qcoef = hmo_quant_scales[rem];
  int32_t *deltaU = e->rq_delta_u;
  int absSum = 0;
  for (int i = 0; i < n2; i++) {
    const int32_t c = src[i];
    const int64_t t = (int64_t)(c < 0 ? -c : c) * qcoef;
    const int32_t q = (int32_t)((t + add) >> qbits);
    deltaU[i] = (int32_t)((t - ((int64_t)q << qbits)) >> qbits8);
    absSum += q;
    dst[i] = clip3(-32768, 32767, c < 0 ? -q : q);
  }
  if (e->p.sign_hiding && absSum >= 2) {
    const uint16_t *scan = hmo_scan_tab[hmo_coef_scan_idx(cu, part, log2, comp)][log2 - 2];
    int lastCG = -1;
    for (int subSet = (n2 - 1) >> 4; subSet >= 0; subSet--) {
      const int subPos = subSet << 4;
      int first = 16, last = -1, sum = 0, n;
      for (n = 15; n >= 0; --n) if (dst[scan[n + subPos]]) { last = n; break; }
      for (n = 0; n < 16; n++) if (dst[scan[n + subPos]]) { first = n; break; }
      for (n = first; n <= last; n++) sum += dst[scan[n + subPos]];
      if (last >= 0 && lastCG == -1) lastCG = 1;
      if (last - first >= 4) {                               /* SBH_THRESHOLD */
        const uint32_t signbit = dst[scan[subPos + first]] > 0 ? 0 : 1;
        if (signbit != (uint32_t)(sum & 1)) {
          int32_t curCost = 0x7fffffff, minCostInc = 0x7fffffff; int minPos = -1, finalChange = 0, curChange = 0;
          for (n = (lastCG == 1 ? last : 15); n >= 0; --n) {
            const int blk = scan[n + subPos];
            if (dst[blk] != 0) {
              if (deltaU[blk] > 0) { curCost = -deltaU[blk]; curChange = 1; }
              else if (n == first && (dst[blk] == 1 || dst[blk] == -1)) curCost = 0x7fffffff;
              else { curCost = deltaU[blk]; curChange = -1; }
            } else if (n < first) {
              const uint32_t thisSign = src[blk] >= 0 ? 0 : 1;
              if (thisSign != signbit) curCost = 0x7fffffff; else { curCost = -deltaU[blk]; curChange = 1; }
            } else { curCost = -deltaU[blk]; curChange = 1; }
            if (curCost < minCostInc) { minCostInc = curCost; finalChange = curChange; minPos = blk; }
          }
          if (dst[minPos] == 32767 || dst[minPos] == -32768) finalChange = -1;
          if (src[minPos] >= 0) dst[minPos] += finalChange; else dst[minPos] -= finalChange;
        }
      }
      if (lastCG == 1) lastCG = 0;
    }
  }
  return absSum;
}

int hmo_rdoq(HmoEnc *e, const HmoCU *cu, const HmoTU *tu, int comp, const int32_t *src, int32_t *dst, int log2, int part, int tskip)
{
  if (!(tskip ? e->p.rdoq_ts : e->p.rdoq)) return quant_plain(e, cu, comp, src, dst, log2, part);     /* xQuant: useRDOQ = useTransformSkip ? m_useRDOQTS : m_useRDOQ */
  const HmoCabac *c = &e->goon;
  const int ch = comp ? 1 : 0, N = 1 << log2, n2 = N * N;
  const int qp = comp ? e->p.qp_c : e->p.qp;
  const int per = qp / 6, rem = qp % 6;
  const int tshift = 15 - 8 - log2;
  const int qbits = 14 + per + tshift;
  const double lambda = e->p.rdoq_lambda[comp];
  const int qcoef = hmo_quant_scales[rem];
  /* setErrScaleCoeff, TComTrQuant.cpp:3018-3040 */
  double errScale = (double)(1 << 15);
  { double p2 = 1.0; int k = -2 * tshift;          /* pow(2.0, -2*tshift) is exact */
    if (k >= 0) for (int i = 0; i < k; i++) p2 *= 2.0; else for (int i = 0; i < -k; i++) p2 *= 0.5;
    errScale = errScale * p2; }
  errScale = errScale / qcoef / qcoef / 1;

  double *costCoeff = e->rq_cost_coeff, *costSig = e->rq_cost_sig, *costCoeff0 = e->rq_cost_coeff0;
  int *rateIncUp = e->rq_rate_up, *rateIncDown = e->rq_rate_down, *sigRateDelta = e->rq_sig_delta;
  int32_t *deltaU = e->rq_delta_u;
  memset(costCoeff, 0, sizeof(double) * (size_t)n2);
  memset(costSig, 0, sizeof(double) * (size_t)n2);
  memset(rateIncUp, 0, sizeof(int) * (size_t)n2);
  memset(rateIncDown, 0, sizeof(int) * (size_t)n2);
  memset(sigRateDelta, 0, sizeof(int) * (size_t)n2);
  memset(deltaU, 0, sizeof(int32_t) * (size_t)n2);

  const int scanType = hmo_coef_scan_idx(cu, part, log2, comp);
  const uint16_t *scan = hmo_scan_tab[scanType][log2 - 2];
  const uint8_t *scanCG = hmo_scan_cg[scanType][log2 - 2];
  const int wg = N >> 2;
  const int firstSig = hmo_first_sig_ctx(log2, scanType, ch);
  const int sigOff = HMO_CTX_SIG + (ch ? 28 : 0);
  const int cgBase = HMO_CTX_SIGCG + (ch ? 2 : 0);

  double costCGSig[64];
  uint8_t cgflag[64];
  memset(costCGSig, 0, sizeof(costCGSig));
  memset(cgflag, 0, sizeof(cgflag));
  int cgLastScanPos = -1;
  uint32_t ctxSet = 0; int c1 = 1, c2 = 0;
  double baseCost = 0, blockUncodedCost = 0;
  int lastScanPos = -1;
  uint32_t c1Idx = 0, c2Idx = 0, goRice = 0;
  const int cgNum = n2 >> 4;
  int absSum = 0;

  for (int cgScanPos = cgNum - 1; cgScanPos >= 0; cgScanPos--) {
    int cgBlk = scanCG[cgScanPos], cgy = cgBlk / wg, cgx = cgBlk - cgy * wg;
    double rdSigCost = 0, rdSigCost0 = 0, rdCodedLevelandDist = 0, rdUncodedDist = 0; int nnzBeforePos0 = 0;
    const int pattern = hmo_pattern_sig_ctx(cgflag, cgx, cgy, wg);
    for (int posInCG = 15; posInCG >= 0; posInCG--) {
      int scanPos = cgScanPos * 16 + posInCG;
      int blk = scan[scanPos];
      int64_t tmpLevel = (int64_t)abs(src[blk]) * qcoef;
      int64_t lim = (int64_t)INT_MAX - ((int64_t)1 << (qbits - 1));
      int32_t levelDouble = (int32_t)(tmpLevel < lim ? tmpLevel : lim);
      uint32_t maxAbsLevel = (uint32_t)((levelDouble + ((int32_t)1 << (qbits - 1))) >> qbits);
      if (maxAbsLevel > 32767u) maxAbsLevel = 32767u;
      double err = (double)levelDouble;
      costCoeff0[scanPos] = err * err * errScale;
      blockUncodedCost += costCoeff0[scanPos];
      dst[blk] = (int32_t)maxAbsLevel;

      if (maxAbsLevel > 0 && lastScanPos < 0) {
        lastScanPos = scanPos;
        ctxSet = (uint32_t)((ch ? 4 : 0) + ((!ch && (scanPos >> 4) > 0) ? 2 : 0));
        cgLastScanPos = cgScanPos;
      }
      if (lastScanPos >= 0) {
        uint32_t level;
        int oneCtx = HMO_CTX_ONE + 4 * (int)ctxSet + c1;
        int absCtx = HMO_CTX_ABS + (int)ctxSet + c2;
        if (scanPos == lastScanPos) {
          level = coded_level(c, lambda, &costCoeff[scanPos], &costCoeff0[scanPos], &costSig[scanPos], levelDouble, maxAbsLevel,
                              sigOff, oneCtx, absCtx, goRice, c1Idx, c2Idx, qbits, errScale, 1);
        } else {
          int ctxSig = sigOff + hmo_sig_ctx_inc(pattern, firstSig, blk, log2, ch);
          level = coded_level(c, lambda, &costCoeff[scanPos], &costCoeff0[scanPos], &costSig[scanPos], levelDouble, maxAbsLevel,
                              ctxSig, oneCtx, absCtx, goRice, c1Idx, c2Idx, qbits, errScale, 0);
          sigRateDelta[blk] = hmo_ctx_bits(c, ctxSig, 1) - hmo_ctx_bits(c, ctxSig, 0);
        }
        deltaU[blk] = (int32_t)((levelDouble - ((int32_t)level << qbits)) >> (qbits - 8));
        if (level > 0) {
          int rateNow = ic_rate(c, level, oneCtx, absCtx, goRice, c1Idx, c2Idx);
          rateIncUp[blk] = ic_rate(c, level + 1, oneCtx, absCtx, goRice, c1Idx, c2Idx) - rateNow;
          rateIncDown[blk] = ic_rate(c, level - 1, oneCtx, absCtx, goRice, c1Idx, c2Idx) - rateNow;
        } else rateIncUp[blk] = hmo_ctx_bits(c, oneCtx, 0);
        dst[blk] = (int32_t)level;
        baseCost += costCoeff[scanPos];

        uint32_t baseLevel = (c1Idx < 8) ? (2 + (c2Idx < 1)) : 1;
        if (level >= baseLevel) {
          if (level > 3u * (1u << goRice)) goRice = goRice + 1 < 4 ? goRice + 1 : 4;
        }
        if (level >= 1) c1Idx++;
        if (level > 1) { c1 = 0; c2 += (c2 < 2); c2Idx++; }
        else if (c1 < 3 && c1 > 0 && level) c1++;

        if ((scanPos % 16 == 0) && scanPos > 0) {
          ctxSet = (uint32_t)((ch ? 4 : 0) + ((!ch && ((scanPos - 1) >> 4) > 0) ? 2 : 0) + (c1 == 0));
          c1 = 1; c2 = 0; c1Idx = 0; c2Idx = 0; goRice = 0;
        }
      } else baseCost += costCoeff0[scanPos];

      rdSigCost += costSig[scanPos];
      if (posInCG == 0) rdSigCost0 = costSig[scanPos];
      if (dst[blk]) {
        cgflag[cgBlk] = 1;
        rdCodedLevelandDist += costCoeff[scanPos] - costSig[scanPos];
        rdUncodedDist += costCoeff0[scanPos];
        if (posInCG != 0) nnzBeforePos0++;
      }
    }

    if (cgLastScanPos >= 0) {
      if (cgScanPos) {
        if (cgflag[cgBlk] == 0) {
          int ctxSig = cgBase + hmo_sig_cg_ctx(cgflag, cgx, cgy, wg);
          baseCost += lambda * (double)hmo_ctx_bits(c, ctxSig, 0) - rdSigCost;
          costCGSig[cgScanPos] = lambda * (double)hmo_ctx_bits(c, ctxSig, 0);
        } else if (cgScanPos < cgLastScanPos) {
          if (nnzBeforePos0 == 0) { baseCost -= rdSigCost0; rdSigCost -= rdSigCost0; }
          double costZeroCG = baseCost;
          int ctxSig = cgBase + hmo_sig_cg_ctx(cgflag, cgx, cgy, wg);
          baseCost += lambda * (double)hmo_ctx_bits(c, ctxSig, 1);
          costZeroCG += lambda * (double)hmo_ctx_bits(c, ctxSig, 0);
          costCGSig[cgScanPos] = lambda * (double)hmo_ctx_bits(c, ctxSig, 1);
          costZeroCG += rdUncodedDist;
          costZeroCG -= rdCodedLevelandDist;
          costZeroCG -= rdSigCost;
          if (costZeroCG < baseCost) {
            cgflag[cgBlk] = 0;
            baseCost = costZeroCG;
            costCGSig[cgScanPos] = lambda * (double)hmo_ctx_bits(c, ctxSig, 0);
            for (int posInCG = 15; posInCG >= 0; posInCG--) {
              int scanPos = cgScanPos * 16 + posInCG, blk = scan[scanPos];
              if (dst[blk]) { dst[blk] = 0; costCoeff[scanPos] = costCoeff0[scanPos]; costSig[scanPos] = 0; }
            }
          }
        }
      } else cgflag[cgBlk] = 1;
    }
  }

  if (lastScanPos < 0) return 0;

  /* ===== last position ===== */
  double bestCost;
  int bestLastIdxP1 = 0;
  {
    int ctxCbf = ch ? (HMO_CTX_CBF_CHROMA + tu->tr_depth) : (HMO_CTX_CBF_LUMA + (tu->tr_depth == 0 ? 1 : 0));
    if (cu->pred_mode[part] != HMO_MODE_INTRA && !ch && cu->tr_idx[part] == 0) ctxCbf = HMO_CTX_ROOT_CBF;   /* blockRootCbpBits[0], TComTrQuant.cpp:2358-2363 */
    bestCost = blockUncodedCost + lambda * (double)hmo_ctx_bits(c, ctxCbf, 0);
    baseCost += lambda * (double)hmo_ctx_bits(c, ctxCbf, 1);
  }
  /* estLastSignificantPositionBit, TEncSbac.cpp:1863-1923 */
  int lastXBits[10], lastYBits[10];
  {
    int cc = log2 - 2;
    int off = ch ? 0 : (cc * 3 + ((cc + 1) >> 2));
    int sh = ch ? cc : ((cc + 3) >> 2);
    int bx = HMO_CTX_LASTX + (ch ? 15 : 0) + off, by = HMO_CTX_LASTY + (ch ? 15 : 0) + off;
    int gmax = hmo_group_idx[N - 1], k, bitsX = 0, bitsY = 0;
    for (k = 0; k < gmax; k++) { lastXBits[k] = bitsX + hmo_ctx_bits(c, bx + (k >> sh), 0); bitsX += hmo_ctx_bits(c, bx + (k >> sh), 1); }
    lastXBits[k] = bitsX;
    for (k = 0; k < gmax; k++) { lastYBits[k] = bitsY + hmo_ctx_bits(c, by + (k >> sh), 0); bitsY += hmo_ctx_bits(c, by + (k >> sh), 1); }
    lastYBits[k] = bitsY;
  }
  int foundLast = 0;
  for (int cgScanPos = cgLastScanPos; cgScanPos >= 0; cgScanPos--) {
    int cgBlk = scanCG[cgScanPos];
    baseCost -= costCGSig[cgScanPos];
    if (cgflag[cgBlk]) {
      for (int posInCG = 15; posInCG >= 0; posInCG--) {
        int scanPos = cgScanPos * 16 + posInCG;
        if (scanPos > lastScanPos) continue;
        int blk = scan[scanPos];
        if (dst[blk]) {
          int py = blk >> log2, px = blk - (py << log2);
          int ax = px, ay = py;
          if (scanType == 2) { ax = py; ay = px; }    /* xGetRateLast(uiPosY, uiPosX) for SCAN_VER */
          int gx = hmo_group_idx[ax], gy = hmo_group_idx[ay];
          double r = (double)(lastXBits[gx] + lastYBits[gy]);   /* xGetRateLast, TComTrQuant.cpp:2898-2916 */
          if (gx > 3) r += 32768.0 * (double)((gx - 2) >> 1);
          if (gy > 3) r += 32768.0 * (double)((gy - 2) >> 1);
          double costLast = lambda * r;
          double totalCost = baseCost + costLast - costSig[scanPos];
          if (totalCost < bestCost) { bestLastIdxP1 = scanPos + 1; bestCost = totalCost; }
          if (dst[blk] > 1) { foundLast = 1; break; }
          baseCost -= costCoeff[scanPos];
          baseCost += costCoeff0[scanPos];
        } else baseCost -= costSig[scanPos];
      }
      if (foundLast) break;
    }
  }

  for (int sp = 0; sp < bestLastIdxP1; sp++) {
    int blk = scan[sp];
    int32_t level = dst[blk];
    absSum += level;
    dst[blk] = (src[blk] < 0) ? -level : level;
  }
  for (int sp = bestLastIdxP1; sp <= lastScanPos; sp++) dst[scan[sp]] = 0;

  /* ===== sign bit hiding, TComTrQuant.cpp:2442-2572 ===== */
  if (e->p.sign_hiding && absSum >= 2) {
    const double invQ = (double)hmo_inv_quant_scales[rem];
    int64_t rdFactor = (int64_t)(invQ * invQ * (double)(1 << (2 * per)) / lambda / 16 / 1 + 0.5);
    int lastCG = -1;
    for (int subSet = (n2 - 1) >> 4; subSet >= 0; subSet--) {
      int subPos = subSet << 4, firstNZ = 16, lastNZ = -1, sum = 0, n;
      for (n = 15; n >= 0; --n) if (dst[scan[n + subPos]]) { lastNZ = n; break; }
      for (n = 0; n < 16; n++) if (dst[scan[n + subPos]]) { firstNZ = n; break; }
      for (n = firstNZ; n <= lastNZ; n++) sum += dst[scan[n + subPos]];
      if (lastNZ >= 0 && lastCG == -1) lastCG = 1;
      if (lastNZ - firstNZ >= 4) {
        uint32_t signbit = dst[scan[subPos + firstNZ]] > 0 ? 0 : 1;
        if (signbit != (uint32_t)(sum & 1)) {
          int64_t minCostInc = INT64_MAX, curCost = INT64_MAX;
          int minPos = -1, finalChange = 0, curChange = 0;
          for (n = (lastCG == 1 ? lastNZ : 15); n >= 0; --n) {
            int blk = scan[n + subPos];
            if (dst[blk] != 0) {
              int64_t costUp = rdFactor * (-deltaU[blk]) + rateIncUp[blk];
              int64_t costDown = rdFactor * (deltaU[blk]) + rateIncDown[blk] - ((abs(dst[blk]) == 1) ? sigRateDelta[blk] : 0);
              if (lastCG == 1 && lastNZ == n && abs(dst[blk]) == 1) costDown -= (4 << 15);
              if (costUp < costDown) { curCost = costUp; curChange = 1; }
              else {
                curChange = -1;
                if (n == firstNZ && abs(dst[blk]) == 1) curCost = INT64_MAX; else curCost = costDown;
              }
            } else {
              curCost = rdFactor * (-(abs(deltaU[blk]))) + (1 << 15) + rateIncUp[blk] + sigRateDelta[blk];
              curChange = 1;
              if (n < firstNZ) {
                uint32_t thissign = src[blk] >= 0 ? 0 : 1;
                if (thissign != signbit) curCost = INT64_MAX;
              }
            }
            if (curCost < minCostInc) { minCostInc = curCost; finalChange = curChange; minPos = blk; }
          }
          if (dst[minPos] == 32767 || dst[minPos] == -32768) finalChange = -1;
          if (src[minPos] >= 0) dst[minPos] += finalChange; else dst[minPos] -= finalChange;
        }
      }
      if (lastCG == 1) lastCG = 0;
    }
  }
  return absSum;
}
