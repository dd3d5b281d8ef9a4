/*
 * hmo_cabac.c -- ORACLE (test infrastructure).  CABAC bit counter and residual syntax.
 */
#include "hmo_int.h"
#include <stdlib.h>
#include <string.h>

/* ContextModel::init, ContextModel.cpp:56-65 ; TEncSbac::resetEntropy, TEncSbac.cpp:106-156 */
void hmo_cabac_init(HmoCabac *c, int qp) { hmo_cabac_init_st(c, qp, HMO_SLICE_I); }
void hmo_cabac_init_st(HmoCabac *c, int qp, int slice_type) { hmo_cabac_init_tab(c, qp, slice_type, 0); }
void hmo_cabac_init_tab(HmoCabac *c, int qp, int slice_type, int b_table)
{
  hmo_init_tables();
  if (qp < 0) qp = 0;
  if (qp > 51) qp = 51;
  /* eSliceType = encCABACTableIdx for a non-intra slice when cabac_init_present_flag is set (TEncSbac.cpp:111-115) */
  const uint8_t *tab = slice_type == HMO_SLICE_P ? (b_table ? hmo_ctx_init_B : hmo_ctx_init_P) : hmo_ctx_init_I;
  for (int i = 0; i < HMO_NCTX; i++) {
    int iv = tab[i];
    int slope = (iv >> 4) * 5 - 45;
    int offset = ((iv & 15) << 3) - 16;
    int st = ((slope * qp) >> 4) + offset;
    if (st < 1) st = 1;
    if (st > 126) st = 126;
    int mps = st >= 64;
    c->ctx[i] = (uint8_t)(((mps ? (st - 64) : (63 - st)) << 1) + mps);
  }
  c->frac = 0;                                   /* TEncBinCABAC::start, TEncBinCoderCABAC.cpp:69-79 */
}

/* TEncBinCABACCounter::encodeBin, TEncBinCoderCABACCounter.cpp:76-84 */
void hmo_enc_bin(HmoEnc *e, int bin, int ctx)
{
  uint8_t s = e->goon.ctx[ctx];
  e->goon_bins++;
  e->goon.frac += (uint64_t)hmo_entropy_bits[s ^ bin];
  e->goon.ctx[ctx] = ((s & 1) == bin) ? hmo_next_mps[s] : hmo_next_lps[s];
}
/* encodeBinEP / encodeBinsEP, TEncBinCoderCABACCounter.cpp:108-124 */
void hmo_enc_bins_ep(HmoEnc *e, int nbins)
{
  e->goon_bins += (uint32_t)nbins;
  e->goon.frac += (uint64_t)32768 * (uint64_t)nbins;
}
/* encodeBinTrm, TEncBinCoderCABACCounter.cpp:131-135 ; getEntropyBitsTrm ContextModel.h:88 */
void hmo_enc_bin_trm(HmoEnc *e, int bin)
{
  e->goon_bins++;
  e->goon.frac += (uint64_t)hmo_entropy_bits[126 ^ bin];
}

/* TComDataCU::getCoefScanIdx, TComDataCU.cpp:3356-3411 (MDCS) */
int hmo_coef_scan_idx(const HmoCU *cu, int part, int log2, int comp)
{
  int maxlog2 = comp ? 2 : 3;                    /* MDCS_MAXIMUM_WIDTH 8 >> chroma scale */
  if (cu->pred_mode[part] != HMO_MODE_INTRA) return 0;   /* SCAN_DIAG for inter CUs (:3360) */
  if (log2 > maxlog2) return 0;
  int dir = cu->intra_dir[comp ? 1 : 0][part];
  if (dir == HMO_DM_CHROMA) dir = cu->intra_dir[0][part & ~3];
  if (abs(dir - HMO_VER) <= 4) return 1;         /* SCAN_HOR */
  if (abs(dir - HMO_HOR) <= 4) return 2;         /* SCAN_VER */
  return 0;
}

/* TComTrQuant::calcPatternSigCtx, TComTrQuant.cpp:2584-2609 */
int hmo_pattern_sig_ctx(const uint8_t *cgflag, int cgx, int cgy, int wg)
{
  if (wg <= 1) return 0;
  int r = 0, l = 0;
  if (cgx < wg - 1) r = cgflag[cgy * wg + cgx + 1] != 0;
  if (cgy < wg - 1) l = cgflag[(cgy + 1) * wg + cgx] != 0;
  return r + (l << 1);
}
/* TComTrQuant::getSigCoeffGroupCtxInc, TComTrQuant.cpp:2949-2969 */
int hmo_sig_cg_ctx(const uint8_t *cgflag, int cgx, int cgy, int wg)
{
  int r = 0, l = 0;
  if (cgx < wg - 1) r = cgflag[cgy * wg + cgx + 1] != 0;
  if (cgy < wg - 1) l = cgflag[(cgy + 1) * wg + cgx] != 0;
  return (r + l) != 0;
}
/* first significance-map context, getTUEntropyCodingParameters TComChromaFormat.cpp:129-155 */
int hmo_first_sig_ctx(int log2, int scan, int ch)
{
  if (log2 == 2) return 0;
  if (log2 == 3) return 9 + ((scan != 0 && !ch) ? 6 : 0);
  return ch ? 12 : 21;
}
/* TComTrQuant::getSigCtxInc, TComTrQuant.cpp:2619-2718 */
int hmo_sig_ctx_inc(int pattern, int first, int pos, int log2, int ch)
{
  int py = pos >> log2, px = pos - (py << log2);
  if (px + py == 0) return 0;
  int offset;
  if (log2 == 2) offset = hmo_ctx_ind_map4x4[4 * py + px];
  else {
    int cnt, xs = px & 3, ys = py & 3;
    switch (pattern) {
      case 0: { int t = xs + ys; cnt = (t >= 3) ? 0 : ((t >= 1) ? 1 : 2); } break;
      case 1: cnt = (ys >= 2) ? 0 : ((ys >= 1) ? 1 : 2); break;
      case 2: cnt = (xs >= 2) ? 0 : ((xs >= 1) ? 1 : 2); break;
      default: cnt = 2; break;
    }
    int notFirst = ((px >> 2) + (py >> 2)) > 0;
    offset = (notFirst ? (ch ? 0 : 3) : 0) + cnt;
  }
  return first + offset;
}

/* TEncSbac::xWriteCoefRemainExGolomb, TEncSbac.cpp:338-391 (bit count only) */
static void code_coef_remain(HmoEnc *e, uint32_t symbol, uint32_t rparam)
{
  int code = (int)symbol;
  if (code < (3 << rparam)) {
    uint32_t length = (uint32_t)code >> rparam;
    hmo_enc_bins_ep(e, (int)(length + 1));
    hmo_enc_bins_ep(e, (int)rparam);
  } else {
    uint32_t length = rparam;
    code -= 3 << rparam;
    while (code >= (1 << length)) code -= 1 << (length++);
    hmo_enc_bins_ep(e, (int)(3 + length + 1 - rparam));
    hmo_enc_bins_ep(e, (int)length);
  }
}

/* TEncSbac::codeLastSignificantXY, TEncSbac.cpp:1115-1179 */
static void code_last_xy(HmoEnc *e, int px, int py, int log2, int ch, int scan)
{
  if (scan == 2) { int t = px; px = py; py = t; }
  int gx = hmo_group_idx[px], gy = hmo_group_idx[py];
  int c = log2 - 2;
  int off = ch ? 0 : (c * 3 + ((c + 1) >> 2));
  int sh = ch ? c : ((c + 3) >> 2);
  int bx = HMO_CTX_LASTX + (ch ? 15 : 0) + off, by = HMO_CTX_LASTY + (ch ? 15 : 0) + off;
  int gmax = hmo_group_idx[(1 << log2) - 1];
  int k;
  for (k = 0; k < gx; k++) hmo_enc_bin(e, 1, bx + (k >> sh));
  if (gx < gmax) hmo_enc_bin(e, 0, bx + (k >> sh));
  for (k = 0; k < gy; k++) hmo_enc_bin(e, 1, by + (k >> sh));
  if (gy < gmax) hmo_enc_bin(e, 0, by + (k >> sh));
  if (gx > 3) hmo_enc_bins_ep(e, (gx - 2) >> 1);
  if (gy > 3) hmo_enc_bins_ep(e, (gy - 2) >> 1);
}

/* TEncSbac::codeCoeffNxN, TEncSbac.cpp:1181-1535.  `part` = GetAbsPartIdxTU(compID)
 * relative to cu.  Bit counting only: bypass bin values are irrelevant. */
void hmo_code_coeff_nxn(HmoEnc *e, const HmoCU *cu, const int32_t *coef, int log2, int comp, int part)
{
  const int ch = comp ? 1 : 0, N = 1 << log2, n2 = N * N;
  int numSig = 0;
  for (int i = 0; i < n2; i++) numSig += coef[i] != 0;
  if (numSig == 0) abort();                      /* TEncSbac.cpp:1229-1234 */
  int beValid = e->p.sign_hiding;                /* no transquant bypass, no RDPCM */
  if (e->p.transform_skip && log2 == 2)          /* codeTransformSkipFlags, TEncSbac.cpp:997-1028 */
    hmo_enc_bin(e, cu->tskip[comp][part], HMO_CTX_TSKIP + ch);

  const int scanType = hmo_coef_scan_idx(cu, part, log2, comp);
  const uint16_t *scan = hmo_scan_tab[scanType][log2 - 2];
  const uint8_t *scanCG = hmo_scan_cg[scanType][log2 - 2];
  const int wg = N >> 2;
  const int firstSig = hmo_first_sig_ctx(log2, scanType, ch);

  uint8_t cgflag[64];
  memset(cgflag, 0, sizeof(cgflag));
  int scanPosLast = -1, posLast;
  do {
    posLast = scan[++scanPosLast];
    if (coef[posLast] != 0) {
      int py = posLast >> log2, px = posLast - (py << log2);
      cgflag[wg * (py >> 2) + (px >> 2)] = 1;
      numSig--;
    }
  } while (numSig > 0);

  { int py = posLast >> log2, px = posLast - (py << log2); code_last_xy(e, px, py, log2, ch, scanType); }

  const int baseCG = HMO_CTX_SIGCG + (ch ? 2 : 0);
  const int baseSig = HMO_CTX_SIG + (ch ? 28 : 0);
  const int lastSet = scanPosLast >> 4;
  uint32_t c1 = 1, goRice;
  int scanPosSig = scanPosLast;

  for (int sub = lastSet; sub >= 0; sub--) {
    int numNonZero = 0, subPos = sub << 4;
    goRice = 0;
    int absCoeff[16];
    int lastNZ = -1, firstNZ = 16;
    int escape = 0;
    if (scanPosSig == scanPosLast) {
      absCoeff[0] = abs(coef[posLast]);
      numNonZero = 1; lastNZ = scanPosSig; firstNZ = scanPosSig; scanPosSig--;
    }
    int cgpos = scanCG[sub], cgy = cgpos / wg, cgx = cgpos - cgy * wg;
    if (sub == lastSet || sub == 0) cgflag[cgpos] = 1;
    else hmo_enc_bin(e, cgflag[cgpos] != 0, baseCG + hmo_sig_cg_ctx(cgflag, cgx, cgy, wg));

    if (cgflag[cgpos]) {
      int pattern = hmo_pattern_sig_ctx(cgflag, cgx, cgy, wg);
      for (; scanPosSig >= subPos; scanPosSig--) {
        int blk = scan[scanPosSig];
        int sig = coef[blk] != 0;
        if (scanPosSig > subPos || sub == 0 || numNonZero)
          hmo_enc_bin(e, sig, baseSig + hmo_sig_ctx_inc(pattern, firstSig, blk, log2, ch));
        if (sig) {
          absCoeff[numNonZero++] = abs(coef[blk]);
          if (lastNZ == -1) lastNZ = scanPosSig;
          firstNZ = scanPosSig;
        }
      }
    } else scanPosSig = subPos - 1;

    if (numNonZero > 0) {
      int signHidden = (lastNZ - firstNZ >= 4);  /* SBH_THRESHOLD */
      int ctxSet = (ch ? 4 : 0) + ((!ch && sub > 0) ? 2 : 0) + (c1 == 0);
      c1 = 1;
      int baseOne = HMO_CTX_ONE + 4 * ctxSet;
      int numC1 = numNonZero < 8 ? numNonZero : 8;
      int firstC2 = -1;
      for (int idx = 0; idx < numC1; idx++) {
        int sym = absCoeff[idx] > 1;
        hmo_enc_bin(e, sym, baseOne + (int)c1);
        if (sym) {
          c1 = 0;
          if (firstC2 == -1) firstC2 = idx; else escape = 1;
        } else if (c1 < 3 && c1 > 0) c1++;
      }
      if (c1 == 0 && firstC2 != -1) {
        int sym = absCoeff[firstC2] > 2;
        hmo_enc_bin(e, sym, HMO_CTX_ABS + ctxSet);
        if (sym) escape = 1;
      }
      escape = escape || (numNonZero > 8);
      if (beValid && signHidden) hmo_enc_bins_ep(e, numNonZero - 1);
      else hmo_enc_bins_ep(e, numNonZero);
      int firstCoeff2 = 1;
      if (escape) {
        for (int idx = 0; idx < numNonZero; idx++) {
          int base = (idx < 8) ? (2 + firstCoeff2) : 1;
          if (absCoeff[idx] >= base) {
            code_coef_remain(e, (uint32_t)(absCoeff[idx] - base), goRice);
            if (absCoeff[idx] > (3 << goRice)) goRice = goRice + 1 < 4 ? goRice + 1 : 4;
          }
          if (absCoeff[idx] >= 2) firstCoeff2 = 0;
        }
      }
    }
  }
}
