/*
 * ref_driver.cpp -- ORACLE / test infrastructure.  A thin C driver over the REFERENCE's own leaf
 * classes (compiled in place from /root/reference by build_ref.sh).  It contains no restated
 * algorithm: every number it returns is produced by the reference's code.  Used by
 * oracle/ref/make_golden.py to generate tests/golden/*.npz, and (when oracle/_ref exists) by the
 * tests directly.
 */
#include <sstream>
#include <iostream>
#include <fstream>
#include <iomanip>
#include <vector>
#include <list>
#include <map>
#include <set>
#include <deque>
#include <string>
#include <algorithm>
#include <limits>
#include <memory>
#include <functional>
#include <utility>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <cassert>
#include <cstdio>
#include <stdint.h>
#define private public
#define protected public
#include "TLibCommon/CommonDef.h"
#include "TLibCommon/TComRom.h"
#include "TLibCommon/TComPic.h"
#include "TLibCommon/TComSlice.h"
#include "TLibCommon/TComDataCU.h"
#include "TLibCommon/TComTU.h"
#include "TLibCommon/TComPrediction.h"
#include "TLibCommon/TComTrQuant.h"
#include "TLibCommon/TComRdCost.h"
#include "TLibCommon/TComBitCounter.h"
#include "TLibCommon/TComLoopFilter.h"
#include "TLibEncoder/TEncSbac.h"
#include "TLibEncoder/TEncEntropy.h"
#include "TLibEncoder/TEncBinCoderCABACCounter.h"
#include "TLibEncoder/TEncCfg.h"
#include "TLibEncoder/TEncSearch.h"
#include "TLibEncoder/TEncSampleAdaptiveOffset.h"
#include "TLibEncoder/TEncTop.h"
#include "TLibEncoder/TEncCu.h"
#include "TLibEncoder/TEncSlice.h"
/* The reference's own TEncCu.cpp is compiled next to the repository's adapter (which defines TEncCu's public methods over
 * libfcu.so) under the class name TEncCuRef: build_ref.sh passes -DTEncCu=TEncCuRef to that translation unit only, and this
 * driver sees the same class declaration a second time under that name. */
#define TEncCu TEncCuRef
#undef __TENCCU__
#include "TLibEncoder/TEncCu.h"
#undef TEncCu
#undef private
#undef protected
#include <math.h>
#include <string.h>
#include "fcu_marshal.h"

/* Every object of the reference that this driver creates starts from zeroed memory: several reference classes leave
 * members to their owner (TEncTop::init -- not built -- sets e.g. TComRdCost::m_costMode, the TEncCfg fields), and a
 * process whose heap has been used before would hand them arbitrary values.  (The version script keeps these operators
 * local to libhmleaf.so.) */
void *operator new(std::size_t n) { void *p = calloc(1, n ? n : 1); if (!p) throw std::bad_alloc(); return p; }
void *operator new[](std::size_t n) { void *p = calloc(1, n ? n : 1); if (!p) throw std::bad_alloc(); return p; }
void operator delete(void *p) noexcept { free(p); }
void operator delete[](void *p) noexcept { free(p); }
void operator delete(void *p, std::size_t) noexcept { free(p); }
void operator delete[](void *p, std::size_t) noexcept { free(p); }

Void xTrMxN(Int bitDepth, TCoeff *block, TCoeff *coeff, Int iWidth, Int iHeight, Bool useDST, const Int maxTrDynamicRange);
Void xITrMxN(Int bitDepth, TCoeff *coeff, TCoeff *block, Int iWidth, Int iHeight, Bool useDST, const Int maxTrDynamicRange);

/* heap objects that are never destroyed: TComPic::create copies the SPS / PPS (raw pointer members), so static
 * instances would be freed twice by the exit-time destructors */
static TComSPS &g_sps = *new TComSPS; static TComPPS &g_pps = *new TComPPS; static TComPic *g_pic = 0; static TComSlice *g_slice = 0;
static TComPrediction *g_pred = 0; static TComTrQuant *g_trq = 0; static TComRdCost *g_rd = 0;
static TEncSbac *g_sbac = 0; static TEncBinCABACCounter *g_bin = 0; static TEncEntropy *g_ent = 0; static TComBitCounter *g_bits = 0;
static int g_qp = 32;

extern "C" {

static void release_refpics(void);
int ref_setup(int width, int height, int qp)
{
  release_refpics();                                 /* reference pictures of an earlier set-up may have another size */
  g_qp = qp;
  g_uiMaxCUWidth = 64; g_uiMaxCUHeight = 64; g_uiMaxCUDepth = 4; g_uiAddCUDepth = 1;
  g_bitDepth[0] = g_bitDepth[1] = 8; g_maxTrDynamicRange[0] = g_maxTrDynamicRange[1] = 15;
  initROM();
  ContextModel::buildNextStateTable();               /* TEncTop::TEncTop(), TEncTop.cpp */
  UInt *piTmp = &g_auiZscanToRaster[0];
  initZscanToRaster(5, 1, 0, piTmp);           /* m_uhTotalDepth = g_uiMaxCUDepth + 1, TEncCu.cpp:167,206 */
  initRasterToZscan(64, 64, 5);
  initRasterToPelXY(64, 64, 5);
  g_sps.setChromaFormatIdc(CHROMA_420);
  g_sps.setPicWidthInLumaSamples(width); g_sps.setPicHeightInLumaSamples(height);
  g_sps.setMaxCUWidth(64); g_sps.setMaxCUHeight(64); g_sps.setMaxCUDepth(4);
  g_sps.setLog2MinCodingBlockSize(3); g_sps.setLog2DiffMaxMinCodingBlockSize(3);
  g_sps.setQuadtreeTULog2MaxSize(5); g_sps.setQuadtreeTULog2MinSize(2);
  g_sps.setQuadtreeTUMaxDepthInter(3); g_sps.setQuadtreeTUMaxDepthIntra(3);
  g_sps.setMaxTrSize(32); g_sps.setUsePCM(false); g_sps.setUseAMP(false);   /* AMP 0: the inter configuration built; no effect on intra */
  g_sps.setBitDepth(CHANNEL_TYPE_LUMA, 8); g_sps.setBitDepth(CHANNEL_TYPE_CHROMA, 8);
  g_sps.setQpBDOffset(CHANNEL_TYPE_LUMA, 0); g_sps.setQpBDOffset(CHANNEL_TYPE_CHROMA, 0);
  g_sps.setUseStrongIntraSmoothing(true);
  g_pps.setUseTransformSkip(true); g_pps.setTransformSkipLog2MaxSize(2); g_pps.setSignHideFlag(true);
  g_pps.setUseDQP(false); g_pps.setConstrainedIntraPred(false);
  g_pic = new TComPic();
  g_pic->create(g_sps, g_pps, 64, 64, 4, false);
  g_slice = g_pic->getSlice(0);
  g_slice->setSPS(&g_pic->getPicSym()->getSPS()); g_slice->setPPS(&g_pic->getPicSym()->getPPS());
  g_slice->setSliceType(I_SLICE); g_slice->setSliceQp(qp); g_slice->setPic(g_pic);
  g_slice->setSliceCurStartCtuTsAddr(0); g_slice->setSliceCurEndCtuTsAddr(g_pic->getNumberOfCtusInFrame());
  g_slice->setSliceSegmentCurStartCtuTsAddr(0); g_slice->setSliceSegmentCurEndCtuTsAddr(g_pic->getNumberOfCtusInFrame());
  for (UInt a = 0; a < g_pic->getNumberOfCtusInFrame(); a++) {
    TComDataCU *c = g_pic->getCtu(a);
    c->initCtu(g_pic, a);
    c->setPredModeSubParts(MODE_INTRA, 0, 0); c->setPartSizeSubParts(SIZE_2Nx2N, 0, 0);
  }
  g_pred = new TComPrediction(); g_pred->initTempBuff(CHROMA_420);
  g_rd = new TComRdCost(); g_rd->init(); g_rd->setCostMode(COST_STANDARD_LOSSY);   /* TEncTop::init */
  g_trq = new TComTrQuant(); g_trq->init(32, true, true, true, true, false);
  g_trq->setFlatScalingList(CHROMA_420); g_trq->setUseScalingList(false);
  /* slice lambda as TEncSlice::initEncSlice / setUpLambda do for an I slice */
  const double lambda = 0.57 * pow(2.0, ((double)qp - 12) / 3.0);
  const int qpc = (int)g_aucChromaScale[CHROMA_420][qp];
  const double w = pow(2.0, (qp - qpc) / 3.0);
  g_rd->setLambda(lambda);
  g_rd->setDistortionWeight(COMPONENT_Cb, w); g_rd->setDistortionWeight(COMPONENT_Cr, w);
  double lambdas[3] = { lambda, lambda / w, lambda / w };
  g_trq->setLambdas(lambdas);
  g_bits = new TComBitCounter(); g_bin = new TEncBinCABACCounter(); g_sbac = new TEncSbac(); g_ent = new TEncEntropy();
  g_sbac->init(g_bin);
  g_ent->setEntropyCoder(g_sbac, g_slice);
  g_ent->setBitstream(g_bits);
  g_ent->resetEntropy();
  g_bin->setBinCountingEnableFlag(true);
  return (int)g_pic->getNumberOfCtusInFrame();
}

/* ---- tables -------------------------------------------------------------------------- */
void ref_zscan_to_raster(int *out) { for (int i = 0; i < 256; i++) out[i] = (int)g_auiZscanToRaster[i]; }
void ref_scan(int grouped, int type, int log2w, int *out) { const UInt *s = g_scanOrder[grouped ? SCAN_GROUPED_4x4 : SCAN_UNGROUPED][type][log2w][log2w]; for (int i = 0; i < (1 << (2 * log2w)); i++) out[i] = (int)s[i]; }
void ref_dct(int log2, int *out)
{
  const int n = 1 << log2;
  for (int k = 0; k < n; k++) for (int j = 0; j < n; j++)
    out[k * n + j] = log2 == 2 ? g_aiT4[TRANSFORM_FORWARD][k][j] : log2 == 3 ? g_aiT8[TRANSFORM_FORWARD][k][j] : log2 == 4 ? g_aiT16[TRANSFORM_FORWARD][k][j] : g_aiT32[TRANSFORM_FORWARD][k][j];
}
int ref_chroma_qp(int qp) { return g_aucChromaScale[CHROMA_420][qp]; }
int ref_entropy_bits(int stateXorBin) { return ContextModel::m_entropyBits[stateXorBin]; }
int ref_next_state(int state, int bin) { return ContextModel::m_nextState[state][bin]; }

/* ---- transforms / distortion ----------------------------------------------------------- */
void ref_fwd(const short *resi, int log2, int useDst, int *coef)
{
  const int n = 1 << log2; TCoeff blk[1024], out[1024];
  for (int i = 0; i < n * n; i++) blk[i] = resi[i];
  xTrMxN(8, blk, out, n, n, useDst != 0, 15);
  for (int i = 0; i < n * n; i++) coef[i] = out[i];
}
void ref_inv(const int *coef, int log2, int useDst, short *resi)
{
  const int n = 1 << log2; TCoeff in[1024], out[1024];
  for (int i = 0; i < n * n; i++) in[i] = coef[i];
  xITrMxN(8, in, out, n, n, useDst != 0, 15);
  for (int i = 0; i < n * n; i++) resi[i] = (short)out[i];
}
static void to_pel(const unsigned char *s, Pel *d, int n) { for (int i = 0; i < n; i++) d[i] = s[i]; }
unsigned ref_satd(const unsigned char *org, const unsigned char *cur, int w, int h)
{
  static Pel a[4096], b[4096]; to_pel(org, a, w * h); to_pel(cur, b, w * h);
  DistParam dp; g_rd->setDistParam(dp, 8, a, w, b, w, w, h, true); dp.bApplyWeight = false;
  return dp.DistFunc(&dp);
}
unsigned ref_sse(const unsigned char *org, const unsigned char *cur, int w, int h, int comp)
{
  static Pel a[4096], b[4096]; to_pel(org, a, w * h); to_pel(cur, b, w * h);
  return g_rd->getDistPart(8, b, w, a, w, w, h, ComponentID(comp));
}
double ref_rd_cost(unsigned bits, unsigned dist) { return g_rd->calcRdCost(bits, dist); }
double ref_lambda(int which) { return which == 0 ? g_rd->getLambda() : which == 1 ? g_rd->getSqrtLambda() : g_rd->m_distortionWeight[1]; }

/* ---- picture state --------------------------------------------------------------------- */
void ref_set_rec(int comp, const unsigned char *plane)
{
  TComPicYuv *r = g_pic->getPicYuvRec(); const ComponentID c = ComponentID(comp);
  Pel *p = r->getAddr(c); const int s = r->getStride(c), w = r->getWidth(c), h = r->getHeight(c);
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) p[y * s + x] = plane[y * w + x];
}
/* field: 0 depth, 1 part_size, 2 pred_mode, 3 luma dir, 4 chroma dir, 5 tr_idx, 6..8 tskip, 9..11 cbf, 12 width/height from depth */
void ref_set_ctu_field(int ctu, int field, const unsigned char *v)
{
  TComDataCU *c = g_pic->getCtu(ctu);
  for (int i = 0; i < 256; i++) {
    switch (field) {
      case 0: c->getDepth()[i] = v[i]; c->getWidth()[i] = c->getHeight()[i] = (UChar)(64 >> v[i]); break;
      case 1: c->getPartitionSize()[i] = (Char)v[i]; break;
      case 2: c->getPredictionMode()[i] = (Char)v[i]; break;
      case 3: c->getIntraDir(CHANNEL_TYPE_LUMA)[i] = v[i]; break;
      case 4: c->getIntraDir(CHANNEL_TYPE_CHROMA)[i] = v[i]; break;
      case 5: c->getTransformIdx()[i] = v[i]; break;
      case 6: case 7: case 8: c->getTransformSkip(ComponentID(field - 6))[i] = v[i]; break;
      case 9: case 10: case 11: c->getCbf(ComponentID(field - 9))[i] = v[i]; break;
    }
  }
}

void ref_set_ctu_qp(int ctu, const signed char *qp) { TComDataCU *c = g_pic->getCtu(ctu); for (int i = 0; i < 256; i++) c->getQP()[i] = qp[i]; }
void ref_get_rec(int comp, unsigned char *plane)
{
  TComPicYuv *r = g_pic->getPicYuvRec(); const ComponentID c = ComponentID(comp);
  const Pel *p = r->getAddr(c); const int s = r->getStride(c), w = r->getWidth(c), h = r->getHeight(c);
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) plane[y * w + x] = (unsigned char)p[y * s + x];
}
/* the reference's in-loop deblocking of the whole picture (TComLoopFilter::loopFilterPic, called at TEncGOP.cpp:1160
 * with the configuration TEncGOP.cpp:1155-1159 / TEncSlice.cpp:446-470 applies by default: filter enabled, beta / tc
 * offsets 0, LFCrossSliceBoundaryFlag 1, LFCrossTileBoundaryFlag 1) */
void ref_deblock(int betaOffsetDiv2, int tcOffsetDiv2)
{
  g_slice->setDeblockingFilterDisable(false);
  g_slice->setDeblockingFilterBetaOffsetDiv2(betaOffsetDiv2); g_slice->setDeblockingFilterTcOffsetDiv2(tcOffsetDiv2);
  g_slice->setLFCrossSliceBoundaryFlag(true);
  TComLoopFilter lf;
  lf.create(g_uiMaxCUDepth);
  lf.setCfg(true);
  lf.loopFilterPic(g_pic);
  lf.destroy();
}

void ref_set_ctu_coeff(int ctu, int comp, const int *coef, int n)
{ TCoeff *d = g_pic->getCtu(ctu)->getCoeff(ComponentID(comp)); for (int i = 0; i < n; i++) d[i] = coef[i]; }
/* ---- CU syntax through the reference's own entropy coder (TEncEntropy / TEncSbac on the bit counter).  Each wrapper is
 * the call TEncCu::xEncodeCU makes at that point (TEncCu.cpp:1679-1778); which CUs exist is decided by the caller from the
 * depth array. */
void ref_enc_split(int ctu, int part, int depth) { g_ent->encodeSplitFlag(g_pic->getCtu(ctu), part, depth); }      /* :1695 */
void ref_enc_cu(int ctu, int part, int depth)
{
  TComDataCU *cu = g_pic->getCtu(ctu);
  g_ent->encodePredMode(cu, part);                                                                                   /* :1752 */
  g_ent->encodePartSize(cu, part, depth);                                                                            /* :1753 */
  if (cu->isIntra(part) && cu->getPartitionSize(part) == SIZE_2Nx2N) g_ent->encodeIPCMInfo(cu, part);                /* :1755-1765 */
  g_ent->encodePredInfo(cu, part);                                                                                   /* :1768 */
  Bool dqp = false, cqa = false;
  g_ent->encodeCoeff(cu, part, depth, dqp, cqa);                                                                     /* :1773 */
}
/* finishCU, TEncCu.cpp:1618-1640: the 0 terminating bit after the last CU of a CTU that does not end the slice */
void ref_enc_finish(int ctu, int part, int lastCtuOfSlice)
{ if (g_pic->getCtu(ctu)->isLastSubCUOfCtu(part) && !lastCtuOfSlice) g_ent->encodeTerminatingBit(0); }

/* descend a TU tree: root = the CU at (zidx, depth) of CTU `ctu`; path[k] = child index at level k */
struct TuChain { TComTURecurse *lv[5]; int n; };
static TComTU *make_tu(TuChain &t, TComDataCU *cu, int zidx, int depth, int nsplit, const int *path, bool processLast)
{
  t.n = 0;
  t.lv[t.n++] = new TComTURecurse(cu, zidx, depth);
  for (int k = 0; k < nsplit; k++) {
    TComTURecurse *c = new TComTURecurse(*t.lv[t.n - 1], processLast);
    for (int i = 0; i < path[k]; i++) c->nextSection(*t.lv[t.n - 1]);
    t.lv[t.n++] = c;
  }
  return t.lv[t.n - 1];
}
static void free_tu(TuChain &t) { for (int i = t.n - 1; i >= 0; i--) delete t.lv[i]; }

/* intra prediction of one block exactly as xIntraCodingTUBlock does it (TEncSearch.cpp:1166-1171) */
int ref_intra(int ctu, int zidx, int depth, int nsplit, const int *path, int comp, int mode, unsigned char *pred, short *ref_unf, short *ref_filt)
{
  TComDataCU *cu = g_pic->getCtu(ctu); TuChain tc;
  cu->setDepthSubParts(depth, zidx); cu->setSizeSubParts(64 >> depth, 64 >> depth, zidx, depth);
  TComTU *tu = make_tu(tc, cu, zidx, depth, nsplit, path, false);
  const ComponentID c = ComponentID(comp);
  if (!tu->ProcessComponentSection(c)) { free_tu(tc); return 0; }
  const int w = tu->getRect(c).width, h = tu->getRect(c).height;
  Bool above = false, left = false;
  const Bool filt = TComPrediction::filteringIntraReferenceSamples(c, mode, w, h, CHROMA_420, false);
  g_pred->initAdiPatternChType(*tu, above, left, c, filt);
  static Pel dst[64 * 64];
  g_pred->predIntraAng(c, mode, 0, 0, dst, w, *tu, above, left, filt);
  for (int i = 0; i < w * h; i++) pred[i] = (unsigned char)dst[i];
  const int sw = 2 * w + 1;
  const Pel *u = g_pred->getPredictorPtr(c, false), *f = g_pred->getPredictorPtr(c, true);
  /* linear walk: bottom-left ... corner ... top-right */
  for (int i = 0; i < 2 * h; i++) { ref_unf[i] = u[(2 * h - i) * sw]; ref_filt[i] = filt ? f[(2 * h - i) * sw] : 0; }
  for (int i = 0; i <= 2 * w; i++) { ref_unf[2 * h + i] = u[i]; ref_filt[2 * h + i] = filt ? f[i] : 0; }
  free_tu(tc);
  return filt ? 2 : 1;
}

int ref_mpm(int ctu, int part, int *preds)
{
  Int mode = -1; Int p[3] = { -1, -1, -1 };
  g_pic->getCtu(ctu)->getIntraDirPredictor(part, p, COMPONENT_Y, &mode);
  preds[0] = p[0]; preds[1] = p[1]; preds[2] = p[2];
  return mode;
}
int ref_ctx_split(int ctu, int part, int depth) { return g_pic->getCtu(ctu)->getCtxSplitFlag(part, depth); }

/* ---- entropy coder state ----------------------------------------------------------------- */
void ref_cabac_reset(void) { g_ent->resetEntropy(); g_ent->resetBits(); }
void ref_cabac_reset_bits(void) { g_ent->resetBits(); }
unsigned long long ref_cabac_frac(void) { return g_bin->m_fracBits; }
unsigned ref_cabac_bits(void) { return g_ent->getNumberOfWrittenBits(); }
/* HM context order (TEncSbac.cpp:56-96); returns the number of models */
int ref_cabac_states(unsigned char *out) { for (int i = 0; i < g_sbac->m_numContextModels; i++) out[i] = g_sbac->m_contextModels[i].m_ucState; return g_sbac->m_numContextModels; }

/* transformNxN (DCT/DST/TS + RDOQ + sign hiding) then invTransformNxN on one TU; CU fields are set
 * like the search would have set them.  Rate tables come from the current coder state. */
int ref_tq(int ctu, int zidx, int depth, int nsplit, const int *path, int comp, int partSize, int lumaDir, int chromaDir,
           int tskip, const short *resi, int *coef, short *resi_out)
{
  TComDataCU *cu = g_pic->getCtu(ctu); TuChain tc;
  const ComponentID c = ComponentID(comp);
  cu->setPredModeSubParts(MODE_INTRA, zidx, depth); cu->setPartSizeSubParts(PartSize(partSize), zidx, depth);
  cu->setDepthSubParts(depth, zidx); cu->setSizeSubParts(64 >> depth, 64 >> depth, zidx, depth);
  cu->setIntraDirSubParts(CHANNEL_TYPE_LUMA, lumaDir, zidx, depth); cu->setIntraDirSubParts(CHANNEL_TYPE_CHROMA, chromaDir, zidx, depth);
  cu->setQPSubParts(g_qp, zidx, depth);
  TComTU *tu = make_tu(tc, cu, zidx, depth, nsplit, path, false);
  if (!tu->ProcessComponentSection(c)) { free_tu(tc); return -1; }
  const int w = tu->getRect(c).width, h = tu->getRect(c).height;
  cu->setTransformSkipPartRange(tskip, c, tu->GetAbsPartIdxTU(c), tu->GetAbsPartIdxNumParts(c));
  static Pel r[32 * 32]; static TCoeff q[32 * 32], arl[32 * 32];
  for (int i = 0; i < w * h; i++) r[i] = resi[i];
  g_ent->estimateBit(g_trq->m_pcEstBitsSbac, w, h, toChannelType(c));
  const QpParam cQP(*cu, c);
  g_trq->selectLambda(c);
  TCoeff absSum = 0;
  g_trq->transformNxN(*tu, c, r, w, q, arl, absSum, cQP);
  for (int i = 0; i < w * h; i++) coef[i] = q[i];
  if (absSum > 0) g_trq->invTransformNxN(*tu, c, r, w, q, cQP);
  else memset(r, 0, sizeof(Pel) * w * h);
  for (int i = 0; i < w * h; i++) resi_out[i] = r[i];
  free_tu(tc);
  return (int)absSum;
}

/* codeCoeffNxN bit count on the running coder (TEncEntropy::encodeCoeffNxN needs cbf != 0) */
void ref_code_coeff(int ctu, int zidx, int depth, int nsplit, const int *path, int comp, int partSize, int lumaDir, int chromaDir, int tskip, const int *coef)
{
  TComDataCU *cu = g_pic->getCtu(ctu); TuChain tc;
  const ComponentID c = ComponentID(comp);
  cu->setPredModeSubParts(MODE_INTRA, zidx, depth); cu->setPartSizeSubParts(PartSize(partSize), zidx, depth);
  cu->setDepthSubParts(depth, zidx); cu->setSizeSubParts(64 >> depth, 64 >> depth, zidx, depth);
  cu->setIntraDirSubParts(CHANNEL_TYPE_LUMA, lumaDir, zidx, depth); cu->setIntraDirSubParts(CHANNEL_TYPE_CHROMA, chromaDir, zidx, depth);
  TComTU *tu = make_tu(tc, cu, zidx, depth, nsplit, path, false);
  const int w = tu->getRect(c).width, h = tu->getRect(c).height;
  cu->setTransformSkipPartRange(tskip, c, tu->GetAbsPartIdxTU(c), tu->GetAbsPartIdxNumParts(c));
  static TCoeff q[32 * 32];
  for (int i = 0; i < w * h; i++) q[i] = coef[i];
  g_sbac->codeCoeffNxN(*tu, q, c);
  free_tu(tc);
}


/* ==== the reference's PU / TU search loops (TEncSearch.cpp, compiled in place; build_ref.sh) ================================
 * ref_search_setup builds what TEncTop::create / init build around a TEncSearch (TEncTop.cpp:94-231): the RD coder slots
 * [depth][CI_*] on bit counters, the go-on coder, a TEncCfg carrying the search switches, and the per-depth CU / sample
 * buffers of TEncCu::create (TEncCu.cpp:163-198).  The calls below are the calls TEncCu makes (cited per function); no
 * search algorithm is restated here. */
static TEncCfg *g_cfg = 0; static TEncSearch *g_search = 0;
static TEncSbac ***g_rdSbac = 0; static TEncBinCABACCounter ***g_rdBin = 0;
static TEncSbac *g_goOn = 0; static TEncBinCABACCounter *g_goOnBin = 0; static TComBitCounter *g_goOnBits = 0;
static TComDataCU *g_tmpCU[4], *g_bestCU[4];
static TComYuv *g_yOrg[4], *g_yPred[4], *g_yResi[4], *g_yResiBest[4], *g_yReco[4];

int ref_search_setup(int searchRange, int fastSearch, int fastEnc, int hadME, int tsFast)
{
  if (!g_pic) return -1;
  g_cfg = new TEncCfg();
  g_cfg->m_chromaFormatIDC = CHROMA_420; g_cfg->m_uiQuadtreeTULog2MaxSize = 5; g_cfg->m_uiQuadtreeTULog2MinSize = 2;
  g_cfg->m_iFastSearch = fastSearch; g_cfg->m_iSearchRange = searchRange; g_cfg->m_bipredSearchRange = 4;
  g_cfg->m_bUseHADME = hadME != 0; g_cfg->m_bUseFastEnc = fastEnc != 0; g_cfg->m_useRDOQ = true; g_cfg->m_useRDOQTS = true;
  g_cfg->m_rdPenalty = 0; g_cfg->m_reconBasedCrossCPredictionEstimate = false; g_cfg->m_useTransformSkipFast = tsFast != 0;
  g_cfg->m_maxNumMergeCand = 5; g_cfg->m_costMode = COST_STANDARD_LOSSY;
  g_rdSbac = new TEncSbac **[g_uiMaxCUDepth + 1]; g_rdBin = new TEncBinCABACCounter **[g_uiMaxCUDepth + 1];
  for (UInt d = 0; d < g_uiMaxCUDepth + 1; d++) {
    g_rdSbac[d] = new TEncSbac *[CI_NUM]; g_rdBin[d] = new TEncBinCABACCounter *[CI_NUM];
    for (Int ci = 0; ci < CI_NUM; ci++) { g_rdSbac[d][ci] = new TEncSbac; g_rdBin[d][ci] = new TEncBinCABACCounter; g_rdSbac[d][ci]->init(g_rdBin[d][ci]); }
  }
  g_goOn = new TEncSbac; g_goOnBin = new TEncBinCABACCounter; g_goOn->init(g_goOnBin); g_goOnBits = new TComBitCounter;
  g_search = new TEncSearch();
  g_search->init(g_cfg, g_trq, searchRange, 4, fastSearch, g_ent, g_rd, g_rdSbac, g_goOn);
  for (int d = 0; d < 4; d++) {
    const UInt np = 1 << ((4 - d) << 1), w = 64 >> d;
    g_tmpCU[d] = new TComDataCU; g_tmpCU[d]->create(CHROMA_420, np, w, w, false, 4);
    g_bestCU[d] = new TComDataCU; g_bestCU[d]->create(CHROMA_420, np, w, w, false, 4);
    TComYuv **all[5] = { g_yOrg, g_yPred, g_yResi, g_yResiBest, g_yReco };
    for (int k = 0; k < 5; k++) { all[k][d] = new TComYuv; all[k][d]->create(w, w, CHROMA_420); }
  }
  return 0;
}
void ref_set_org(int comp, const unsigned char *plane)
{
  TComPicYuv *r = g_pic->getPicYuvOrg(); const ComponentID c = ComponentID(comp);
  Pel *p = r->getAddr(c); const int s = r->getStride(c), w = r->getWidth(c), h = r->getHeight(c);
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) p[y * s + x] = plane[y * w + x];
}
/* context states in HM's own order (m_contextModels) + the Q15 counter of one RD slot, or of the go-on coder (depth < 0) */
static TEncSbac *coder_of(int depth, int ci) { return depth < 0 ? g_goOn : g_rdSbac[depth][ci]; }
void ref_coder_set(int depth, int ci, const unsigned char *states, unsigned long long frac)
{
  TEncSbac *c = coder_of(depth, ci);
  for (int i = 0; i < c->m_numContextModels; i++) c->m_contextModels[i].m_ucState = states[i];
  ((TEncBinCABACCounter *)c->m_pcBinIf)->m_fracBits = frac;
}
unsigned long long ref_coder_get(int depth, int ci, unsigned char *states)
{
  TEncSbac *c = coder_of(depth, ci);
  for (int i = 0; i < c->m_numContextModels; i++) states[i] = c->m_contextModels[i].m_ucState;
  return ((TEncBinCABACCounter *)c->m_pcBinIf)->m_fracBits;
}
/* the temp CU of depth d positioned at (ctu, zidx): initCtu / initSubCU chain as xCompressCU descends (TEncCu.cpp:332,1337) */
static TComDataCU *position_cu(int ctu, int zidx, int depth)
{
  g_tmpCU[0]->initCtu(g_pic, ctu);
  for (int d = 1; d <= depth; d++) g_tmpCU[d]->initSubCU(g_tmpCU[d - 1], (zidx >> (2 * (4 - d))) & 3, d, g_qp);
  if (depth == 0) g_tmpCU[0]->initEstData(0, g_qp, false);
  return g_tmpCU[depth];
}
/* point the entropy front end at the go-on coder writing into a bit counter, as TEncSlice::compressSlice does before
 * compressCtu (TEncSlice.cpp:1414-1420) */
static void entropy_to_goon(void)
{
  g_ent->setEntropyCoder(g_goOn, g_slice);
  g_ent->setBitstream(g_goOnBits);
  g_goOnBin->setBinCountingEnableFlag(true);
}
struct RefCuOut {                    /* what the caller reads back: the temp CU after the call */
  unsigned dist, bits, bins, dist_luma; double cost;
  unsigned char luma_dir[256], chroma_dir[256], tr_idx[256], cbf[3][256], tskip[3][256];
  unsigned char skip[256], merge_flag[256], merge_idx[256], inter_dir[256], part_size[256], pred_mode[256]; signed char mvp_idx[256], ref_idx[256];
  short mv[256][2], mvd[256][2];
  int coef[3][4096]; unsigned char reco[3][4096]; unsigned char pred[3][4096];
};
static void read_cu(TComDataCU *cu, int depth, RefCuOut *o)
{
  const int np = 256 >> (2 * depth), s = 64 >> depth;
  o->dist = cu->getTotalDistortion(); o->bits = cu->getTotalBits(); o->bins = cu->getTotalBins(); o->cost = cu->getTotalCost();
  for (int i = 0; i < np; i++) {
    o->luma_dir[i] = cu->getIntraDir(CHANNEL_TYPE_LUMA)[i]; o->chroma_dir[i] = cu->getIntraDir(CHANNEL_TYPE_CHROMA)[i];
    o->tr_idx[i] = cu->getTransformIdx()[i];
    for (int c = 0; c < 3; c++) { o->cbf[c][i] = cu->getCbf(ComponentID(c))[i]; o->tskip[c][i] = cu->getTransformSkip(ComponentID(c))[i]; }
    o->skip[i] = cu->getSkipFlag()[i]; o->merge_flag[i] = cu->getMergeFlag()[i]; o->merge_idx[i] = cu->getMergeIndex()[i];
    o->inter_dir[i] = cu->getInterDir()[i]; o->part_size[i] = (unsigned char)cu->getPartitionSize()[i]; o->pred_mode[i] = (unsigned char)cu->getPredictionMode()[i];
    o->mvp_idx[i] = cu->getMVPIdx(REF_PIC_LIST_0)[i]; o->ref_idx[i] = cu->getCUMvField(REF_PIC_LIST_0)->getRefIdx(i);
    const TComMv &m = cu->getCUMvField(REF_PIC_LIST_0)->getMv(i), &dm = cu->getCUMvField(REF_PIC_LIST_0)->getMvd(i);
    o->mv[i][0] = m.getHor(); o->mv[i][1] = m.getVer(); o->mvd[i][0] = dm.getHor(); o->mvd[i][1] = dm.getVer();
  }
  for (int c = 0; c < 3; c++) {
    const int n = c ? (s / 2) * (s / 2) : s * s, w = c ? s / 2 : s;
    const TCoeff *q = cu->getCoeff(ComponentID(c));
    for (int i = 0; i < n; i++) o->coef[c][i] = q[i];
    const Pel *r = g_yReco[depth]->getAddr(ComponentID(c)); const int rs = g_yReco[depth]->getStride(ComponentID(c));
    const Pel *pp = g_yPred[depth]->getAddr(ComponentID(c)); const int ps = g_yPred[depth]->getStride(ComponentID(c));
    for (int y = 0; y < w; y++) for (int x = 0; x < w; x++) { o->reco[c][y * w + x] = (unsigned char)r[y * rs + x]; o->pred[c][y * w + x] = (unsigned char)pp[y * ps + x]; }
  }
}
/* ---- CU level: the reference's own xCheckRDCostMerge2Nx2N / xCheckRDCostInter / xCheckRDCostIntra / xCheckBestMode /
 * deriveTestModeAMP (TEncCu.cpp:381,1900,2025,2064,2213; compiled from the reference's TEncCu.cpp without the body of
 * xCompressCU, whose fork code needs OpenCV) on a TEncCuRef object that shares the driver's search / quantiser / RD-cost /
 * entropy objects and RD coder slots.  ref_cu_run evaluates all candidates of ONE CU from the state the caller loaded
 * (picture, neighbours, coder slot [depth][CI_CURR_BEST]) in xCompressCU's order (TEncCu.cpp:753-1143 -- that order, the
 * cbf gate of the intra test and the AMP call pattern are what this function restates; every call is the reference's) and
 * returns the CU that survived: the best / temp swaps of xCheckBestMode, the TEMP_BEST -> NEXT_BEST coder hand-over and the
 * FastDecisionForMerge early-outs inside xCheckRDCostMerge2Nx2N are the reference's own. */
static TEncCuRef *g_cuRef = 0;
static void cu_ref_setup(void)
{
  if (!g_cuRef) { g_cuRef = new TEncCuRef(); g_cuRef->create(4, 64, 64, CHROMA_420); }
  /* (re)bound on every call: ref_setup / ref_search_setup make new objects for every picture */
  g_cuRef->m_pcEncCfg = g_cfg; g_cuRef->m_pcPredSearch = g_search; g_cuRef->m_pcTrQuant = g_trq; g_cuRef->m_pcRdCost = g_rd;
  g_cuRef->m_pcEntropyCoder = g_ent; g_cuRef->m_pcBinCABAC = 0; g_cuRef->m_pppcRDSbacCoder = g_rdSbac; g_cuRef->m_pcRDGoOnSbacCoder = g_goOn;
  g_cuRef->m_pcRateCtrl = 0;
}
/* best / temp CU of `depth` positioned at (ctu, zidx) the way xCompressCU finds them on entry: initCtu, then initSubCU
 * down the quadtree (TEncCu.cpp:332-334,1337-1339) */
static void cu_ref_position(int ctu, int zidx, int depth)
{
  g_cuRef->m_ppcBestCU[0]->initCtu(g_pic, ctu); g_cuRef->m_ppcTempCU[0]->initCtu(g_pic, ctu);
  for (int d = 1; d <= depth; d++) {
    const int part = (zidx >> (2 * (4 - d))) & 3;
    g_cuRef->m_ppcBestCU[d]->initSubCU(g_cuRef->m_ppcTempCU[d - 1], part, d, g_qp);
    g_cuRef->m_ppcTempCU[d]->initSubCU(g_cuRef->m_ppcTempCU[d - 1], part, d, g_qp);
  }
  g_cuRef->m_ppcOrigYuv[depth]->copyFromPicYuv(g_pic->getPicYuvOrg(), ctu, zidx);       /* TEncCu.cpp:474 */
}
static void read_cu_ref(TComDataCU *cu, int depth, TComYuv *reco, TComYuv *pred, RefCuOut *o)
{
  TComYuv *r0 = g_yReco[depth], *p0 = g_yPred[depth];
  g_yReco[depth] = reco; g_yPred[depth] = pred;
  read_cu(cu, depth, o);
  g_yReco[depth] = r0; g_yPred[depth] = p0;
}
/* One intra CU candidate through the reference.  stage 2: TEncCu::xCheckRDCostIntra itself (TEncCu.cpp:2064-2157, the
 * reference's function on a TEncCuRef object) against a best CU that has not seen a candidate yet, so the candidate always
 * ends up in rpcBestCU after the function's own xCheckBestMode; the caller reads the coder from [depth][CI_TEMP_BEST], where
 * the function stored it.  stage 1 = luma search only (estIntraPredLumaQT, BASELINE configs[1]).  The caller has loaded the
 * picture state the search reads: PicYuvOrg, PicYuvRec (ref_set_org / ref_set_rec), the CTUs' decided arrays
 * (ref_set_ctu_field) and the coder slot [depth][CI_CURR_BEST] (ref_coder_set). */
int ref_intra_cu(int ctu, int zidx, int depth, int partSize, int stage, RefCuOut *out)
{
  if (stage >= 2) {
    cu_ref_setup();
    cu_ref_position(ctu, zidx, depth);
    entropy_to_goon();
    g_goOn->load(g_rdSbac[depth][CI_CURR_BEST]);                                   /* TEncSlice.cpp:1417 / TEncCu.cpp:1343-1347 */
    TComDataCU *&best = g_cuRef->m_ppcBestCU[depth], *&temp = g_cuRef->m_ppcTempCU[depth];
    Double cost = 0;
    g_cuRef->xCheckRDCostIntra(best, temp, cost, PartSize(partSize));
    read_cu_ref(best, depth, g_cuRef->m_ppcRecoYuvBest[depth], g_cuRef->m_ppcPredYuvBest[depth], out);
    {                                                            /* the luma share of the distortion: SSE of the winning luma reconstruction (unweighted, TComRdCost.cpp:433-455) */
      const TComYuv *o = g_cuRef->m_ppcOrigYuv[depth], *r = g_cuRef->m_ppcRecoYuvBest[depth];
      const Pel *po = o->getAddr(COMPONENT_Y), *pr = r->getAddr(COMPONENT_Y); const int so = o->getStride(COMPONENT_Y), sr = r->getStride(COMPONENT_Y), w = 64 >> depth;
      unsigned sse = 0;
      for (int y = 0; y < w; y++) for (int x = 0; x < w; x++) { const int e = po[y * so + x] - pr[y * sr + x]; sse += (unsigned)(e * e); }
      out->dist_luma = sse;
    }
    return 0;
  }
  TComDataCU *cu = position_cu(ctu, zidx, depth);
  entropy_to_goon();
  g_goOn->load(g_rdSbac[depth][CI_CURR_BEST]);                                     /* TEncSlice.cpp:1417 / TEncCu.cpp:1343-1347 */
  g_yOrg[depth]->copyFromPicYuv(g_pic->getPicYuvOrg(), ctu, zidx);                  /* TEncCu.cpp:474 */
  cu->setSkipFlagSubParts(false, 0, depth);
  cu->setPartSizeSubParts(PartSize(partSize), 0, depth);
  cu->setPredModeSubParts(MODE_INTRA, 0, depth);
  cu->setChromaQpAdjSubParts(0, 0, depth);
  static Pel resiLuma[NUMBER_OF_STORED_RESIDUAL_TYPES][MAX_CU_SIZE * MAX_CU_SIZE];
  g_search->estIntraPredLumaQT(cu, g_yOrg[depth], g_yPred[depth], g_yResi[depth], g_yReco[depth], resiLuma);
  out->dist_luma = cu->getTotalDistortion();
  if (stage >= 2) {
    g_yReco[depth]->copyToPicComponent(COMPONENT_Y, g_pic->getPicYuvRec(), ctu, zidx);
    g_search->estIntraPredChromaQT(cu, g_yOrg[depth], g_yPred[depth], g_yResi[depth], g_yReco[depth], resiLuma);
    g_ent->resetBits();
    g_ent->encodeSkipFlag(cu, 0, true);
    g_ent->encodePredMode(cu, 0, true);
    g_ent->encodePartSize(cu, 0, depth, true);
    g_ent->encodePredInfo(cu, 0);
    g_ent->encodeIPCMInfo(cu, 0, true);
    Bool dqp = false, cqa = false;
    g_ent->encodeCoeff(cu, 0, depth, dqp, cqa);
    g_goOn->store(g_rdSbac[depth][CI_TEMP_BEST]);
    cu->getTotalBits() = g_ent->getNumberOfWrittenBits();
    cu->getTotalBins() = g_goOnBin->getBinsCoded();
    cu->getTotalCost() = g_rd->calcRdCost(cu->getTotalBits(), cu->getTotalDistortion());
  }
  read_cu(cu, depth, out);
  return 0;
}


/* ==== P slices: one reference picture (list 0), TMVP off, MaxNumMergeCand 5 ================================================== */
static TComPic *g_refpic = 0;
/* slice type, reference picture planes and the slice lambda (TEncSlice::setUpLambda) */
int ref_setup_p(const unsigned char *ry, const unsigned char *ru, const unsigned char *rv, double lambda)
{
  if (!g_pic) return -1;
  if (!g_refpic) { g_refpic = new TComPic(); g_refpic->create(g_sps, g_pps, 64, 64, 4, false); }
  const unsigned char *pl[3] = { ry, ru, rv };
  for (int c = 0; c < 3; c++) {
    TComPicYuv *r = g_refpic->getPicYuvRec(); const ComponentID id = ComponentID(c);
    Pel *p = r->getAddr(id); const int s = r->getStride(id), w = r->getWidth(id), h = r->getHeight(id);
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) p[y * s + x] = pl[c][y * w + x];
  }
  g_refpic->getPicYuvRec()->setBorderExtension(false);
  g_refpic->getPicYuvRec()->extendPicBorder();                /* TEncGOP / TComPic::compressMotion path: references are padded */
  g_refpic->getSlice(0)->setPOC(0);
  g_slice->setSliceType(P_SLICE); g_slice->setPOC(1);
  g_slice->setNumRefIdx(REF_PIC_LIST_0, 1); g_slice->setNumRefIdx(REF_PIC_LIST_1, 0);
  g_slice->setRefPic(g_refpic, REF_PIC_LIST_0, 0);
  g_slice->setRefPOCList();
  g_slice->setEnableTMVPFlag(false);
  g_slice->setMaxNumMergeCand(5);
  const int qpc = (int)g_aucChromaScale[CHROMA_420][g_qp];
  const double w = pow(2.0, (g_qp - qpc) / 3.0);
  g_rd->setLambda(lambda);
  g_rd->setDistortionWeight(COMPONENT_Cb, w); g_rd->setDistortionWeight(COMPONENT_Cr, w);
  double lambdas[3] = { lambda, lambda / w, lambda / w };
  g_trq->setLambdas(lambdas);
  return 0;
}
/* Several reference pictures: RefPicList0[k] = planes[3k..3k+2] at POC pocs[k] (k < n <= 4), the current picture at curPoc.
 * The neighbours' and the slice's reference POCs come from TComSlice::setRefPOCList, as in TEncGOP.cpp:1006. */
static TComPic *g_refpics[4] = { 0, 0, 0, 0 };
} /* extern "C" */
static void release_refpics(void)
{
  for (int k = 1; k < 4; k++) if (g_refpics[k] && g_refpics[k] != g_refpic) { g_refpics[k]->destroy(); delete g_refpics[k]; }
  if (g_refpic) { g_refpic->destroy(); delete g_refpic; }
  g_refpic = 0; for (int k = 0; k < 4; k++) g_refpics[k] = 0;
}
extern "C" {
int ref_setup_p_multi(int n, const unsigned char *const *planes, const int *pocs, int curPoc, double lambda)
{
  if (!g_pic || n < 1 || n > 4) return -1;
  if (ref_setup_p(planes[0], planes[1], planes[2], lambda) != 0) return -1;
  g_refpics[0] = g_refpic;
  for (int k = 1; k < n; k++) {
    if (!g_refpics[k]) { g_refpics[k] = new TComPic(); g_refpics[k]->create(g_sps, g_pps, 64, 64, 4, false); }
    for (int c = 0; c < 3; c++) {
      TComPicYuv *r = g_refpics[k]->getPicYuvRec(); const ComponentID id = ComponentID(c);
      Pel *p = r->getAddr(id); const int s = r->getStride(id), w = r->getWidth(id), h = r->getHeight(id);
      for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) p[y * s + x] = planes[3 * k + c][y * w + x];
    }
    g_refpics[k]->getPicYuvRec()->setBorderExtension(false);
    g_refpics[k]->getPicYuvRec()->extendPicBorder();
  }
  for (int k = 0; k < n; k++) { g_refpics[k]->getSlice(0)->setPOC(pocs[k]); g_refpics[k]->setIsLongTerm(false); }
  g_slice->setPOC(curPoc);
  g_slice->setNumRefIdx(REF_PIC_LIST_0, n);
  for (int k = 0; k < n; k++) g_slice->setRefPic(g_refpics[k], REF_PIC_LIST_0, k);
  g_slice->setRefPOCList();
  return 0;
}
/* TMVP with several references: the collocated picture (RefPicList0[0]) holds its own POC and the POCs its list 0 named */
void ref_col_finish_multi(int curPoc, int colPoc, const int *colRefPocs, int n)
{
  g_refpic->compressMotion();
  TComSlice *cs = g_refpic->getSlice(0);
  cs->setPOC(colPoc);
  for (int k = 0; k < n; k++) { cs->setRefPOC(colRefPocs[k], REF_PIC_LIST_0, k); cs->setIsUsedAsLongTerm(REF_PIC_LIST_0, k, false); }
  cs->setSliceType(P_SLICE); cs->setNumRefIdx(REF_PIC_LIST_0, n);     /* (read by the adapter; xGetColMVP asks the CUs, not the slice type) */
  g_slice->setPOC(curPoc); g_slice->setRefPOCList();
  g_slice->setEnableTMVPFlag(true); g_slice->setColFromL0Flag(1); g_slice->setColRefIdx(0); g_slice->setCheckLDC(true);
}
/* TMVP: the reference picture as the collocated picture.  Its decided CTUs' prediction modes and list-0 motion are loaded,
 * TComPic::compressMotion is run on it (as TEncGOP does after a picture is coded, TEncGOP.cpp:1497) and the current slice
 * gets the collocated-picture syntax HM's lowdelay configuration produces (TMVP on, collocated_from_l0, collocated_ref_idx 0). */
void ref_set_col_ctu(int ctu, const signed char *predMode, const short *mv, const signed char *refIdx)
{
  TComDataCU *c = g_refpic->getCtu(ctu);
  if (c->getPic() == 0) c->initCtu(g_refpic, ctu);
  TComCUMvField *f0 = c->getCUMvField(REF_PIC_LIST_0), *f1 = c->getCUMvField(REF_PIC_LIST_1);
  for (int i = 0; i < 256; i++) {
    const bool decided = predMode[i] == MODE_INTER || predMode[i] == MODE_INTRA;
    c->getPredictionMode()[i] = decided ? predMode[i] : (signed char)NUMBER_OF_PREDICTION_MODES;
    c->getPartitionSize()[i] = decided ? (signed char)SIZE_2Nx2N : (signed char)NUMBER_OF_PART_SIZES;
    f0->m_pcMv[i].set(mv[2 * i], mv[2 * i + 1]); f0->m_piRefIdx[i] = predMode[i] == MODE_INTER ? refIdx[i] : -1;
    f1->m_pcMv[i].set(0, 0); f1->m_piRefIdx[i] = -1;
  }
}
void ref_col_finish(int poc)
{
  g_refpic->compressMotion();
  TComSlice *cs = g_refpic->getSlice(0);
  cs->setPOC(poc - 1); cs->setRefPOC(poc - 2, REF_PIC_LIST_0, 0); cs->setIsUsedAsLongTerm(REF_PIC_LIST_0, 0, false);
  cs->setSliceType(P_SLICE); cs->setNumRefIdx(REF_PIC_LIST_0, 1);
  g_slice->setPOC(poc); g_slice->setRefPOCList();
  g_slice->setEnableTMVPFlag(true); g_slice->setColFromL0Flag(1); g_slice->setColRefIdx(0); g_slice->setCheckLDC(true);
}
/* inter fields of a decided CTU: skip flags, inter direction, list-0 motion */
void ref_set_ctu_inter(int ctu, const unsigned char *skip, const unsigned char *interDir, const unsigned char *mergeFlag, const short *mv, const signed char *refIdx)
{
  TComDataCU *c = g_pic->getCtu(ctu);
  TComCUMvField *f = c->getCUMvField(REF_PIC_LIST_0);
  for (int i = 0; i < 256; i++) {
    c->getSkipFlag()[i] = skip[i] != 0; c->getInterDir()[i] = interDir[i]; c->getMergeFlag()[i] = mergeFlag[i] != 0;
    f->m_pcMv[i].set(mv[2 * i], mv[2 * i + 1]); f->m_piRefIdx[i] = refIdx[i];
  }
}
/* One inter candidate through the reference: TEncCu::xCheckRDCostInter itself (TEncCu.cpp:2025-2062) on a TEncCuRef object */
int ref_inter_cu(int ctu, int zidx, int depth, int partSizeArg, RefCuOut *out)
{
  const int partSize = partSizeArg & 15; const bool useMRG = ((partSizeArg >> 4) & 1) != 0;       /* bit 4: bUseMRG (AMP_MRG: merge estimation only) */
  cu_ref_setup();
  cu_ref_position(ctu, zidx, depth);
  entropy_to_goon();
  g_goOn->load(g_rdSbac[depth][CI_CURR_BEST]);
  TComDataCU *&best = g_cuRef->m_ppcBestCU[depth], *&temp = g_cuRef->m_ppcTempCU[depth];
  g_cuRef->xCheckRDCostInter(best, temp, PartSize(partSize), useMRG);      /* the reference's function; the fresh best CU loses to any candidate */
  read_cu_ref(best, depth, g_cuRef->m_ppcRecoYuvBest[depth], g_cuRef->m_ppcPredYuvBest[depth], out);
  out->dist_luma = 0;
  return 0;
}
/* One merge candidate, with or without residual: the loop body of TEncCu::xCheckRDCostMerge2Nx2N (TEncCu.cpp:1941-1975).
 * cands5x3 receives the candidate list (mv hor, mv ver, refIdx) of getInterMergeCandidates. */
int ref_merge_cu(int ctu, int zidx, int depth, int cand, int noResidual, RefCuOut *out, int *cands5x3)
{
  TComDataCU *cu = position_cu(ctu, zidx, depth);
  entropy_to_goon();
  g_goOn->load(g_rdSbac[depth][CI_CURR_BEST]);
  g_yOrg[depth]->copyFromPicYuv(g_pic->getPicYuvOrg(), ctu, zidx);
  TComMvField mvf[2 * MRG_MAX_NUM_CANDS]; UChar dirs[MRG_MAX_NUM_CANDS]; Int numValid = 0;
  for (UInt ui = 0; ui < cu->getSlice()->getMaxNumMergeCand(); ++ui) dirs[ui] = 0;
  cu->setPartSizeSubParts(SIZE_2Nx2N, 0, depth);
  cu->getInterMergeCandidates(0, 0, mvf, dirs, numValid);
  for (int i = 0; i < 5; i++) { cands5x3[3 * i] = mvf[2 * i].getHor(); cands5x3[3 * i + 1] = mvf[2 * i].getVer(); cands5x3[3 * i + 2] = mvf[2 * i].getRefIdx(); }
  if (cand >= numValid) return -1;
  cu->setPredModeSubParts(MODE_INTER, 0, depth);
  cu->setCUTransquantBypassSubParts(false, 0, depth);
  cu->setChromaQpAdjSubParts(0, 0, depth);
  cu->setPartSizeSubParts(SIZE_2Nx2N, 0, depth);
  cu->setMergeFlagSubParts(true, 0, 0, depth);
  cu->setMergeIndexSubParts(cand, 0, 0, depth);
  cu->setInterDirSubParts(dirs[cand], 0, 0, depth);
  cu->getCUMvField(REF_PIC_LIST_0)->setAllMvField(mvf[0 + 2 * cand], SIZE_2Nx2N, 0, 0);
  cu->getCUMvField(REF_PIC_LIST_1)->setAllMvField(mvf[1 + 2 * cand], SIZE_2Nx2N, 0, 0);
  g_search->motionCompensation(cu, g_yPred[depth]);
  g_search->encodeResAndCalcRdInterCU(cu, g_yOrg[depth], g_yPred[depth], g_yResi[depth], g_yResiBest[depth], g_yReco[depth], noResidual != 0);
  read_cu(cu, depth, out);
  out->dist_luma = 0;
  return numValid;
}

/* RDOQ / RDOQTS switches of the quantiser (TComTrQuant::init, TEncTop.cpp; xQuant reads them, TComTrQuant.cpp:1145-1147)
 * and the slice type, which selects the rounding offset of the plain quantiser (:1199) */
/* AMP on / off in the SPS (TEncTop::xInitSPS: setUseAMP): changes the part-size syntax and admits the asymmetric sizes */
void ref_set_amp(int on)
{
  TComSPS *sps = const_cast<TComSPS *>(g_slice->getSPS());
  sps->setUseAMP(on != 0); g_sps.setUseAMP(on != 0);
}
/* cabac_init_flag machinery: the table index the encoder chose after the previous slice (TEncSlice.cpp:1750-1753) reaches
 * TEncSbac::resetEntropy through the slice (TEncSbac.cpp:111-115); cabac_init_present_flag must be on in the PPS */
void ref_set_cabac_table(int bTable)
{
  TComPPS *pps = const_cast<TComPPS *>(g_slice->getPPS());
  pps->setCabacInitPresentFlag(true); g_pps.setCabacInitPresentFlag(true);
  g_slice->setEncCABACTableIdx(bTable ? B_SLICE : P_SLICE);
}
void ref_set_rdoq(int rdoq, int rdoqTS) { g_trq->m_useRDOQ = rdoq != 0; g_trq->m_useRDOQTS = rdoqTS != 0; }
void ref_set_slice_type(int isP) { g_slice->setSliceType(isP ? P_SLICE : I_SLICE); }

/* TZ search (FastSearch 1) starts non-2Nx2N / deeper searches from the integer vector of the last 2Nx2N search
 * (m_integerMv2Nx2N, TEncSearch.cpp:3822-3833): encoder state that the caller carries over from its own search */
void ref_set_int_mv(int refIdx, int x, int y) { g_search->m_integerMv2Nx2N[0][refIdx].set(x, y); }

/* ---- sample adaptive offset: the reference's own TEncSampleAdaptiveOffset on the picture held by the driver -------------
 * PicYuvOrg / PicYuvRec (the deblocked picture) are loaded with ref_set_org / ref_set_rec.  The call sequence is TEncGOP's
 * (TEncGOP.cpp:1427-1441): initRDOCabacCoder(go-on coder, slice) then SAOProcess(pic, sliceEnabled, slice lambdas, SaoCtuBoundary
 * = false).  slice_ctus > 0 cuts the picture into slices of that many CTUs (SliceMode 1) the way TEncGOP / TEncSlice leave
 * them in the TComPicSym: one TComSlice per slice, every CTU pointing at its own, LFCrossSliceBoundaryFlag 1.
 * layer = pic->getSlice(0)->getDepth() (temporal layer of the GOP entry); disabledRate[comp] preloads
 * m_saoDisabledRate[comp][layer - 1] (what earlier pictures would have left, :895-917).
 * out_params[ctu][comp][0..34] = modeIdc, typeIdc, typeAuxInfo, offset[32]; out_stats[ctu][comp][type][0..63] = diff[32], count[32];
 * out_misc[0..2] = sliceEnabled, out_rate[comp] = m_saoDisabledRate[comp][layer] afterwards. */
int ref_sao(int sliceType, int qp, const double *lambdas, int slice_ctus, int layer, const double *disabledRate,
            int *out_params, long long *out_stats, int *out_misc, double *out_rate)
{
  const int n = (int)g_pic->getNumberOfCtusInFrame();
  g_slice->setSliceType(sliceType == 0 ? I_SLICE : P_SLICE); g_slice->setSliceQp(qp); g_slice->setDepth(layer);
  g_slice->setLFCrossSliceBoundaryFlag(true);
  g_slice->setLambdas(lambdas);
  if (slice_ctus > 0) {
    const int n_sl = (n + slice_ctus - 1) / slice_ctus;
    for (int k = 1; k < n_sl; k++) {
      g_pic->allocateNewSlice();
      TComSlice *sl = g_pic->getSlice(k);
      sl->copySliceInfo(g_slice);
      sl->setSPS(g_slice->getSPS()); sl->setPPS(g_slice->getPPS()); sl->setPic(g_pic);
      sl->setLFCrossSliceBoundaryFlag(true);
    }
    for (int k = 0; k < n_sl; k++) {
      TComSlice *sl = g_pic->getSlice(k);
      const int a = k * slice_ctus, b = std::min(n, a + slice_ctus);
      sl->setSliceCurStartCtuTsAddr(a); sl->setSliceCurEndCtuTsAddr(b);
      sl->setSliceSegmentCurStartCtuTsAddr(a); sl->setSliceSegmentCurEndCtuTsAddr(b);
      for (int c = a; c < b; c++) g_pic->getCtu(c)->m_pcSlice = sl;
    }
  }
  TEncSampleAdaptiveOffset *sao = new TEncSampleAdaptiveOffset();
  sao->create(g_sps.getPicWidthInLumaSamples(), g_sps.getPicHeightInLumaSamples(), CHROMA_420, 64, 64, 4, 0, 0);   /* TEncTop.cpp:100-104: SaoLumaOffsetBitShift 0 */
  sao->createEncData(false);
  if (layer > 0) for (int c = 0; c < 3; c++) sao->m_saoDisabledRate[c][layer - 1] = disabledRate[c];
  TEncSbac *goOn = new TEncSbac(); TEncBinCABACCounter *bin = new TEncBinCABACCounter(); TComBitCounter *bits = new TComBitCounter();
  goOn->init(bin); bits->resetBits(); goOn->setBitstream(bits);
  sao->initRDOCabacCoder(goOn, g_slice);
  Bool enabled[MAX_NUM_COMPONENT];
  sao->SAOProcess(g_pic, enabled, g_slice->getLambdas(), false);
  SAOBlkParam *bp = g_pic->getPicSym()->getSAOBlkParam();
  for (int a = 0; a < n; a++) for (int c = 0; c < 3; c++) {
    int *o = out_params + (a * 3 + c) * 35; const SAOOffset &p = bp[a][ComponentID(c)];
    o[0] = p.modeIdc; o[1] = p.typeIdc; o[2] = p.typeAuxInfo;
    for (int k = 0; k < 32; k++) o[3 + k] = p.offset[k];
    if (out_stats) for (int t = 0; t < NUM_SAO_NEW_TYPES; t++) {
      long long *q = out_stats + ((a * 3 + c) * 5 + t) * 64; const SAOStatData &sd = sao->m_statData[a][c][t];
      for (int k = 0; k < 32; k++) { q[k] = sd.diff[k]; q[32 + k] = sd.count[k]; }
    }
  }
  for (int c = 0; c < 3; c++) { out_misc[c] = enabled[c] ? 1 : 0; out_rate[c] = sao->m_saoDisabledRate[c][layer]; }
  return n;
}

/* ---- the HM adapter of this repository (adapter/TEncCuFcu.cpp, adapter/fcu_marshal.h) against the reference's classes ----
 * ref_adapter_marshal: fcu_ctu_out bytes -> the picture's TComDataCU of that CTU through the adapter's own marshalling.
 * ref_adapter_encode_ctu: the adapter's TEncCu::encodeCtu (its xEncodeCU / finishCU walk) on the reference's TEncEntropy
 * with the bit counter attached, as TEncSlice::compressSlice runs it after compressCtu (TEncSlice.cpp:1474-1487). */
static TEncCu *g_cuEnc = 0;
void ref_adapter_marshal(int ctu, const void *ctu_out) { fcu_adapter::marshal_ctu(*(const fcu_ctu_out *)ctu_out, g_pic->getCtu(ctu)); }
void ref_adapter_encode_ctu(int ctu)
{
  if (!g_cuEnc) { g_cuEnc = new TEncCu(); g_cuEnc->create(4, 64, 64, CHROMA_420); }
  g_cuEnc->m_pcEntropyCoder = g_ent;
  g_cuEnc->encodeCtu(g_pic->getCtu(ctu));
}
/* The adapter's TEncCu::compressCtu itself -- device calls included -- on the reference's objects, as TEncSlice::compressSlice
 * calls it (TEncSlice.cpp:1468): the members TEncCu::init takes from TEncTop are bound to this driver's objects, the slice
 * carries its lambdas (TEncSlice::setUpLambda -> TComSlice::setLambdas) and the TEncCfg the switches the adapter reads.
 * Needs a GPU (tests/test_gpu_adapter.py); the decided CTU lands in the picture's TComDataCU and PicYuvRec. */
static int g_adapterSliceArg = 0;
/* SliceMode 1 for the adapter runs: SliceArgument CTUs per slice (0: one slice per picture), and the slice the next CTUs belong
 * to -- what TEncGOP's slice loop (TEncGOP.cpp:1102-1138) sets on the TComSlice before TEncSlice::compressSlice */
void ref_adapter_slices(int arg) { g_adapterSliceArg = arg; }
void ref_set_slice_range(int first, int count)
{
  if (first > 0) {                                             /* a further slice of the picture: its own TComSlice, as TEncGOP.cpp:1130-1137 makes one */
    TComSlice *prev = g_slice;
    g_pic->allocateNewSlice();
    const UInt idx = g_pic->getNumAllocatedSlice() - 1;
    g_pic->setCurrSliceIdx(idx);
    g_slice = g_pic->getSlice(idx);
    g_slice->copySliceInfo(prev);
    g_slice->setSliceIdx(idx);
  }
  g_slice->setSliceCurStartCtuTsAddr(first); g_slice->setSliceCurEndCtuTsAddr(first + count);
  g_slice->setSliceSegmentCurStartCtuTsAddr(first); g_slice->setSliceSegmentCurEndCtuTsAddr(first + count);
  for (int a = first; a < first + count; a++) g_pic->getCtu(a)->initCtu(g_pic, a);      /* binds the CTUs to the current slice (TEncSlice.cpp:1395) */
  g_ent->setEntropyCoder(g_sbac, g_slice);
}
int ref_adapter_compress_ctu(int ctu)
{
  if (!g_cfg || !g_pic) return -1;
  if (!g_cuEnc) { g_cuEnc = new TEncCu(); g_cuEnc->create(4, 64, 64, CHROMA_420); }
  g_cfg->m_sliceMode = g_adapterSliceArg > 0 ? FIXED_NUMBER_OF_CTU : NO_SLICES; g_cfg->m_sliceArgument = g_adapterSliceArg; g_cfg->m_useFastDecisionForMerge = true;
  g_cuEnc->m_pcEncCfg = g_cfg; g_cuEnc->m_pcRdCost = g_rd; g_cuEnc->m_pcTrQuant = g_trq; g_cuEnc->m_pcEntropyCoder = g_ent;
  const double l = g_rd->getLambda(), w = g_rd->getChromaWeight();
  const double ls[3] = { l, l / w, l / w };
  g_slice->setLambdas(ls);
  g_cuEnc->compressCtu(g_pic->getCtu(ctu));
  return 0;
}
void ref_adapter_release(void) { if (g_cuEnc) g_cuEnc->destroy(); }
void ref_set_poc(int poc) { g_slice->setPOC(poc); }
/* slice lambda of a picture whose lambda is not the all-intra one (TEncSlice::setUpLambda) */
void ref_set_lambda(double lambda)
{
  const int qpc = (int)g_aucChromaScale[CHROMA_420][g_qp];
  const double w = pow(2.0, (g_qp - qpc) / 3.0);
  g_rd->setLambda(lambda);
  g_rd->setDistortionWeight(COMPONENT_Cb, w); g_rd->setDistortionWeight(COMPONENT_Cr, w);
  double lambdas[3] = { lambda, lambda / w, lambda / w };
  g_trq->setLambdas(lambdas);
}
int ref_adapter_isl_cost(int ctu, int w, int h) { if (!g_cuEnc) { g_cuEnc = new TEncCu(); g_cuEnc->create(4, 64, 64, CHROMA_420); } return g_cuEnc->updateCtuDataISlice(g_pic->getCtu(ctu), w, h); }
/* planes through the adapter's converters: 8-bit plane -> PicYuvRec block by block (widen_ctu_block), and back (narrow_plane) */
void ref_adapter_planes_roundtrip(int comp, const unsigned char *in, unsigned char *out)
{
  const ComponentID c = ComponentID(comp);
  for (UInt a = 0; a < g_pic->getNumberOfCtusInFrame(); a++) fcu_adapter::widen_ctu_block(in, g_pic->getPicYuvRec(), c, a, g_pic->getFrameWidthInCtus());
  fcu_adapter::narrow_plane(g_pic->getPicYuvRec(), c, out);
}

/* flags: bit 0 P slice, bit 1 AMP enabled.  log[k] = (call id << 8) | changed flag of xCheckBestMode where the call returns it
 * (intra), call ids: 1 merge, 2 inter (part size in bits 16..19, bUseMRG bit 20), 3 intra.  Returns the number of calls. */
int ref_cu_run(int ctu, int zidx, int depth, int parentPartSize, int flags, RefCuOut *out, int *log)
{
  cu_ref_setup();
  const bool isP = (flags & 1) != 0, amp = (flags & 2) != 0;
  cu_ref_position(ctu, zidx, depth);
  entropy_to_goon();
  g_goOn->load(g_rdSbac[depth][CI_CURR_BEST]);
  TComDataCU *&best = g_cuRef->m_ppcBestCU[depth], *&temp = g_cuRef->m_ppcTempCU[depth];
  int n = 0;
  bool tryIntra = true;
  if (isP) {
    Bool esd = false;
    g_cuRef->xCheckRDCostMerge2Nx2N(best, temp, &esd); temp->initEstData(depth, g_qp, false); log[n++] = 1 << 8;      /* TEncCu.cpp:774-775 */
    g_cuRef->xCheckRDCostInter(best, temp, SIZE_2Nx2N, false); temp->initEstData(depth, g_qp, false); log[n++] = (2 << 8) | (SIZE_2Nx2N << 16);   /* :780 */
    g_cuRef->xCheckRDCostInter(best, temp, SIZE_Nx2N, false); temp->initEstData(depth, g_qp, false); log[n++] = (2 << 8) | (SIZE_Nx2N << 16);     /* :826 */
    g_cuRef->xCheckRDCostInter(best, temp, SIZE_2NxN, false); temp->initEstData(depth, g_qp, false); log[n++] = (2 << 8) | (SIZE_2NxN << 16);     /* :835 */
    if (amp && depth < 3) {                                       /* :843-943 (AMP_ENC_SPEEDUP, AMP_MRG) */
      Bool hor = false, ver = false, mhor = false, mver = false;
      g_cuRef->deriveTestModeAMP(best, PartSize(parentPartSize), hor, ver, mhor, mver);
      const PartSize hs[2] = { SIZE_2NxnU, SIZE_2NxnD }, vs[2] = { SIZE_nLx2N, SIZE_nRx2N };
      if (hor || mhor) for (int k = 0; k < 2; k++) { g_cuRef->xCheckRDCostInter(best, temp, hs[k], !hor); temp->initEstData(depth, g_qp, false); log[n++] = (2 << 8) | (hs[k] << 16) | ((!hor) << 20); }
      if (ver || mver) for (int k = 0; k < 2; k++) { g_cuRef->xCheckRDCostInter(best, temp, vs[k], !ver); temp->initEstData(depth, g_qp, false); log[n++] = (2 << 8) | (vs[k] << 16) | ((!ver) << 20); }
    }
    tryIntra = best->getCbf(0, COMPONENT_Y) != 0 || best->getCbf(0, COMPONENT_Cb) != 0 || best->getCbf(0, COMPONENT_Cr) != 0;   /* :1033-1036 */
  }
  if (tryIntra) {
    Double cost = 0;
    Bool ch = g_cuRef->xCheckRDCostIntra(best, temp, cost, SIZE_2Nx2N); temp->initEstData(depth, g_qp, false); log[n++] = (3 << 8) | (SIZE_2Nx2N << 16) | (ch ? 1 : 0);   /* :1040 */
    if (depth == 3) { ch = g_cuRef->xCheckRDCostIntra(best, temp, cost, SIZE_NxN); temp->initEstData(depth, g_qp, false); log[n++] = (3 << 8) | (SIZE_NxN << 16) | (ch ? 1 : 0); }   /* :1143 */
  }
  read_cu_ref(best, depth, g_cuRef->m_ppcRecoYuvBest[depth], g_cuRef->m_ppcPredYuvBest[depth], out);
  out->dist_luma = 0;
  return n;
}

/* ---- the fork's pre-pass: TEncSlice::getOutlierWithDCT (TEncSlice.cpp:878-1173), the reference's own member function.
 * It touches no member of TEncSlice (only its TComPic argument, globals, partialButterfly and TCMprocessOneSequence), so it
 * runs on zeroed storage of the class's size: constructing a TEncSlice would need TEncTop / TEncGOP, which are not built.
 * The source plane is g_pic's PicYuvOrg (ref_set_org); out = the OBF count map, (width / 4) x (height / 4). */
void ref_obf(short *out)
{
  TEncSlice *s = (TEncSlice *)calloc(1, sizeof(TEncSlice));
  s->getOutlierWithDCT(g_pic);
  free(s);
  TComPicYuv *o = g_pic->getOBF(); const Pel *p = o->getAddr(COMPONENT_Y); const int st = o->getStride(COMPONENT_Y);
  const int w = g_sps.getPicWidthInLumaSamples() / 4, h = g_sps.getPicHeightInLumaSamples() / 4;
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) out[y * w + x] = p[y * st + x];
}

} /* extern "C" */

/* the two output streams the reference's main program defines (App/TAppEncoder/encmain.cpp:52-53) and getOutlierWithDCT
 * writes its picture dumps to: the driver plays main here; they are never opened, so the writes go nowhere */
ofstream OutlierYuvFile;
ofstream OBFFile;
