/*
 * TEncCuFcu.cpp -- drop-in replacement of TLibEncoder/TEncCu.cpp for HM-16.3 / the Fast-CU-Decision fork: the six public
 * methods of class TEncCu (TLibEncoder/TEncCu.h:104-118) implemented over libfcu.so, plus the protected members the
 * bitstream pass needs (xEncodeCU / finishCU).  The maintainer builds HM with this file instead of TEncCu.cpp and links
 * libfcu.so + the HIP runtime; nothing else in HM changes: TEncSlice::compressSlice keeps calling compressCtu / encodeCtu per
 * CTU (TEncSlice.cpp:1468,1482) and encodeSlice keeps calling encodeCtu (TEncSlice.cpp:1707).
 *
 *   compressCtu : the CU depth / mode RDO of one CTU on the GPU.  Slice setup happens here, on the first CTU of a slice
 *                 (source planes once per picture, reference picture of a P slice, QP / lambdas from the objects
 *                 TEncSlice::initEncSlice / setUpLambda have prepared, chain range = the slice).  The result is written where
 *                 xCompressCU leaves it: the CTU's TComDataCU arrays, PicYuvRec, and the contexts of
 *                 m_pppcRDSbacCoder[0][CI_CURR_BEST] are advanced by the device as compressSlice expects
 *                 (the engine replays encodeCtu on its own coder; HM's own replay at TEncSlice.cpp:1474-1487 still runs on
 *                 the host with the marshalled data and arrives at the same state).
 *   encodeCtu   : HM's syntax walk over the decided CTU with whatever entropy coder is attached -- written here from the
 *                 syntax order of the standard (7.3.8.4 / 7.3.8.5), as TEncCu.cpp:359-373,1679-1778 has it.
 *   PicYuvPred  : not written.  The fork's xCopyYuv2Pic also stores the best prediction there (TEncCu.cpp:2298); nothing in
 *                 Lib/ or App/ reads TComPic::getPicYuvPred() (grep), so the plane is dead output.
 * Only the fork's default control is behind the ABI (Training state = HM's exhaustive RDO, or the Naive decision switches
 * through fcu_chain_set_decision); see INTEGRATION.md for the per-picture calls around this class.
 *
 * Compile check in this repository (no HM build system is run):  adapter/build_check.sh
 */
#include <vector>
#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <hip/hip_runtime_api.h>
#include "TLibEncoder/TEncTop.h"
#include <cstring>
#include "TLibEncoder/TEncCu.h"
#include "fcu_marshal.h"

namespace {

/* device side of one encoder instance (TEncCu is a singleton inside TEncTop; HM encodes one picture at a time) */
struct FcuState {
  fcu_ctx *ctx = nullptr;
  int width = 0, height = 0, n_ctu = 0;
  uint8_t *d_org[3] = { nullptr, nullptr, nullptr }, *d_rec[3] = { nullptr, nullptr, nullptr };
  uint8_t *d_refsrc[3] = { nullptr, nullptr, nullptr };
  /* padded reference pictures resident in HBM, keyed by POC: a picture is uploaded and padded once, however many later pictures
   * name it (the lowdelay cfg keeps the GOP-boundary pictures for up to three GOPs) */
  enum { N_SLOT = FCU_MAX_REF + 1 };
  uint8_t *d_ref[N_SLOT][3] = {}; Int ref_poc[N_SLOT]; Bool ref_valid[N_SLOT] = {};
  fcu_ctu_out *d_out = nullptr;                                /* decisions of the picture being coded ...                          */
  fcu_ctu_out *d_out_prev = nullptr; Int poc_prev = -1 << 30;  /* ... and of the picture coded before it: the TMVP motion field     */
  std::vector<uint8_t> h_plane[3];
  Int poc_loaded = -1 << 30;
  fcu_ctu_out h_out;
  /* Picture-ahead (SliceMode 1): the slices of a picture are independent chains -- nothing a slice's decisions read comes from
   * another slice of the same picture -- so with the first CTU of the first slice all of them are bound with that slice's
   * parameters and decided in ONE launch; compressCtu then hands out the results CTU by CTU.  A later slice whose parameters
   * turn out to differ (a rate control that moves the QP inside the picture) is decided again on its own. */
  /* TEncSearch::m_integerMv2Nx2N (the TZ search's carried start point) lives as long as the encoder: the adapter takes it out of
   * the chain before it rebinds it and puts it back afterwards, so that slices and pictures run one after the other see what HM's
   * search would.  Picture-ahead is used only where that state cannot matter: no slice may begin with a CTU too small for a
   * 64x64 CU (such a slice's first search would read what the slice before it left). */
  int32_t search[2 * FCU_MAX_REF] = {}; bool bound0 = false;
  int max_chains = 0;
  std::vector<fcu_ctu_out> h_all; std::vector<uint8_t> h_rec[3];
  const fcu_ctu_out *last = nullptr;                           /* the record the last compressCtu marshalled (tests) */
  int ahead_last = 0;
  bool ahead_valid = false, serve_ahead = false; Int ahead_poc = -1 << 30; fcu_frame_params ahead_fp; int ahead_nref = 0, ahead_pocs[FCU_MAX_REF];
};
FcuState g_fcu;

void die(const char *what, int rc) { fprintf(stderr, "TEncCuFcu: %s failed (%d): %s\n", what, rc, fcu_last_error()); exit(1); }   /* no CPU fallback */
#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "TEncCuFcu: %s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

size_t plane_bytes(int w, int h, int c) { return c ? (size_t)(w / 2) * (h / 2) : (size_t)w * h; }

void ensure_context(const TComSPS *sps, int chains)
{
  FcuState &S = g_fcu;
  const int w = (int)sps->getPicWidthInLumaSamples(), h = (int)sps->getPicHeightInLumaSamples();
  if (S.ctx && S.width == w && S.height == h && S.max_chains >= chains) return;
  if (S.ctx) {                                                 /* another picture size or more slices per picture: start over */
    for (int c = 0; c < 3; c++) { hipFree(S.d_org[c]); hipFree(S.d_rec[c]); hipFree(S.d_refsrc[c]); for (int k = 0; k < FcuState::N_SLOT; k++) { hipFree(S.d_ref[k][c]); S.ref_valid[k] = false; } }
    hipFree(S.d_out); hipFree(S.d_out_prev); fcu_destroy(S.ctx); S.ctx = nullptr; S.poc_prev = S.poc_loaded = -1 << 30; S.bound0 = false; S.ahead_valid = false;
  }
  if (sps->getChromaFormatIdc() != CHROMA_420 || sps->getBitDepth(CHANNEL_TYPE_LUMA) != 8 || sps->getMaxCUWidth() != 64) { fprintf(stderr, "TEncCuFcu: 8-bit 4:2:0 with 64x64 CTUs only\n"); exit(1); }
  fcu_seq_params sp = { w, h, /*max_chains*/ chains, /*device*/ 0 };
  S.max_chains = chains; S.ahead_valid = false;
  int rc = fcu_create(&sp, &S.ctx);
  if (rc != FCU_OK) die("fcu_create", rc);
  S.width = w; S.height = h; S.n_ctu = fcu_num_ctus(S.ctx);
  size_t pad[3]; fcu_pad_sizes(S.ctx, pad);
  for (int c = 0; c < 3; c++) {
    HIPOK(hipMalloc((void **)&S.d_org[c], plane_bytes(w, h, c))); HIPOK(hipMalloc((void **)&S.d_rec[c], plane_bytes(w, h, c)));
    HIPOK(hipMalloc((void **)&S.d_refsrc[c], plane_bytes(w, h, c)));
    for (int k = 0; k < FcuState::N_SLOT; k++) { HIPOK(hipMalloc((void **)&S.d_ref[k][c], pad[c])); S.ref_valid[k] = false; }
    S.h_plane[c].resize(plane_bytes(w, h, c));
  }
  HIPOK(hipMalloc((void **)&S.d_out, sizeof(fcu_ctu_out) * (size_t)S.n_ctu));
  HIPOK(hipMalloc((void **)&S.d_out_prev, sizeof(fcu_ctu_out) * (size_t)S.n_ctu));
}

void upload(TComPicYuv *pic, uint8_t *const d[3])
{
  FcuState &S = g_fcu;
  for (int c = 0; c < 3; c++) {
    fcu_adapter::narrow_plane(pic, ComponentID(c), S.h_plane[c].data());
    HIPOK(hipMemcpy(d[c], S.h_plane[c].data(), S.h_plane[c].size(), hipMemcpyHostToDevice));
  }
}

/* first CTU of a slice: bind the chain to the picture with the slice's parameters and restrict it to the slice */
void begin_slice(TComDataCU *pCtu, TComRdCost *rd, TComTrQuant *trq, TEncCfg *cfg)
{
  FcuState &S = g_fcu;
  TComPic *pic = pCtu->getPic(); TComSlice *slice = pCtu->getSlice();
  const int nCtuPic = (int)pic->getNumberOfCtusInFrame();
  const int sliceArg = cfg->getSliceMode() == FIXED_NUMBER_OF_CTU ? cfg->getSliceArgument() : 0;
  int nSlices = (sliceArg > 0 && sliceArg < nCtuPic) ? (nCtuPic + sliceArg - 1) / sliceArg : 1;
  {
    const int wc = (int)pic->getFrameWidthInCtus(), W = (int)slice->getSPS()->getPicWidthInLumaSamples(), H = (int)slice->getSPS()->getPicHeightInLumaSamples();
    for (int k = 0; k < nSlices && nSlices > 1; k++) {           /* picture-ahead only if every slice starts on a CTU that holds a 64x64 CU */
      const int a = k * sliceArg, x0 = (a % wc) * 64, y0 = (a / wc) * 64;
      if (x0 + 64 > W || y0 + 64 > H) nSlices = 1;
    }
    static const bool noAhead = getenv("FCU_ADAPTER_NO_AHEAD") != nullptr;     /* switch: slice by slice, as HM runs them */
    if (noAhead) nSlices = 1;
  }
  if (S.ctx && S.bound0) { int r0 = fcu_chain_get_search_state(S.ctx, S.ahead_valid ? S.ahead_last : 0, S.search); if (r0 != FCU_OK) die("fcu_chain_get_search_state", r0); }
  ensure_context(slice->getSPS(), nSlices);
  S.serve_ahead = false;
  if (pic->getPOC() != S.poc_loaded) {                          /* once per picture: the source planes; the finished picture's decisions stay resident */
    upload(pic->getPicYuvOrg(), S.d_org);
    std::swap(S.d_out, S.d_out_prev); S.poc_prev = S.poc_loaded;
    S.poc_loaded = pic->getPOC(); S.ahead_valid = false;
    for (int k = 0; k < FcuState::N_SLOT; k++) if (S.ref_valid[k] && S.ref_poc[k] == S.poc_loaded) S.ref_valid[k] = false;   /* a POC coded again (new IDR period) */
  }
  fcu_frame_params fp;
  memset(&fp, 0, sizeof(fp));                                   /* (compared bytewise with the picture-ahead parameters below) */
  fcu_default_frame_params(&fp, slice->getSliceQp());
  fp.lambda = rd->getLambda(); fp.sqrt_lambda = rd->getSqrtLambda(); fp.chroma_weight = rd->getChromaWeight();   /* TEncSlice::setUpLambda */
  for (int c = 0; c < 3; c++) fp.rdoq_lambda[c] = slice->getLambdas()[c];
  const TComPPS *pps = slice->getPPS();
  fp.transform_skip = pps->getUseTransformSkip(); fp.transform_skip_fast = cfg->getUseTransformSkipFast();
  fp.sign_hiding = pps->getSignHideFlag(); fp.strong_intra_smoothing = slice->getSPS()->getUseStrongIntraSmoothing();
  fp.rdoq = cfg->getUseRDOQ(); fp.rdoq_ts = cfg->getUseRDOQTS();
  const int first = (int)pic->getPicSym()->getCtuTsToRsAddrMap(slice->getSliceCurStartCtuTsAddr());
  const int count = (int)(slice->getSliceCurEndCtuTsAddr() - slice->getSliceCurStartCtuTsAddr());
  fp.slice_ctus = cfg->getSliceMode() == FIXED_NUMBER_OF_CTU ? cfg->getSliceArgument() : 0;
  if (!slice->isIntra()) {
    if (slice->isInterB() || slice->getNumRefIdx(REF_PIC_LIST_0) < 1 || slice->getNumRefIdx(REF_PIC_LIST_0) > FCU_MAX_REF) { fprintf(stderr, "TEncCuFcu: P slices with 1..%d reference pictures only\n", FCU_MAX_REF); exit(1); }
    fp.slice_type = FCU_SLICE_P;
    fp.search_range = cfg->getSearchRange(); fp.fast_enc = cfg->getUseFastEnc(); fp.hadamard_me = cfg->getUseHADME();
    fp.fast_merge_decision = cfg->getUseFastDecisionForMerge(); fp.max_merge_cand = slice->getMaxNumMergeCand();
    fp.fast_search = cfg->getFastSearch() ? 1 : 0;
    fp.tmvp = slice->getEnableTMVPFlag() ? 1 : 0;
    fp.amp = slice->getSPS()->getUseAMP() ? 1 : 0;              /* part sizes 4..7 come back with HM's own PartSize values */
    /* cabac_init_flag: the RD coders of this slice were reset with the table TEncSbac::determineCabacInitIdx chose after the
     * previous slice (TEncSlice.cpp:1750-1753 -> TComSlice::getEncCABACTableIdx, TEncSbac::resetEntropy TEncSbac.cpp:111-115) */
    fp.cabac_b_table = (pps->getCabacInitPresentFlag() && slice->getEncCABACTableIdx() == B_SLICE) ? 1 : 0;
  }
  int rc = FCU_OK;
  const int nRef = fp.slice_type == FCU_SLICE_P ? slice->getNumRefIdx(REF_PIC_LIST_0) : 0;
  const uint8_t *planes[3 * FCU_MAX_REF]; int pocs[FCU_MAX_REF], slot[FCU_MAX_REF];
  int colPocs[FCU_MAX_REF], nCol = 0, colPoc = 0;
  if (fp.slice_type == FCU_SLICE_P) {                           /* list 0: the filtered reconstructions HM holds */
    for (int r = 0; r < nRef; r++) {
      pocs[r] = slice->getRefPOC(REF_PIC_LIST_0, r); slot[r] = -1;
      for (int k = 0; k < FcuState::N_SLOT; k++) if (S.ref_valid[k] && S.ref_poc[k] == pocs[r]) slot[r] = k;
    }
    for (int r = 0; r < nRef; r++) {
      if (slot[r] < 0) {                                        /* not resident yet: a slot no picture of this list occupies */
        int k = 0;
        for (; k < FcuState::N_SLOT; k++) { bool used = false; for (int q = 0; q < nRef; q++) used |= slot[q] == k; if (!used) break; }
        upload(slice->getRefPic(REF_PIC_LIST_0, r)->getPicYuvRec(), S.d_refsrc);
        rc = fcu_pad_reference(S.ctx, S.d_refsrc[0], S.d_refsrc[1], S.d_refsrc[2], S.d_ref[k][0], S.d_ref[k][1], S.d_ref[k][2], nullptr);
        if (rc != FCU_OK) die("fcu_pad_reference", rc);
        S.ref_poc[k] = pocs[r]; S.ref_valid[k] = true; slot[r] = k;
      }
      for (int c = 0; c < 3; c++) planes[3 * r + c] = S.d_ref[slot[r]][c];
    }
    if (fp.tmvp) {
      /* collocated picture = list 0, collocated_ref_idx 0 (the encoder's choice for low-delay P): the picture coded before this
       * one, whose fcu_ctu_out array is still in HBM.  Its own list 0 gives the POCs its vectors point at; the engine scales by
       * the two POC distances (TComDataCU::xGetColMVP, TComDataCU.cpp:3242-3310). */
      TComPic *col = slice->getRefPic(RefPicList(slice->getColFromL0Flag() ? 0 : 1), slice->getColRefIdx());
      TComSlice *cs = col->getSlice(0);
      if (!slice->getColFromL0Flag() || slice->getColRefIdx() != 0 || col->getPOC() != S.poc_prev || (!cs->isIntra() && cs->getNumRefIdx(REF_PIC_LIST_0) > FCU_MAX_REF)) {
        fprintf(stderr, "TEncCuFcu: TMVP needs the collocated picture to be list 0 index 0 and the previously coded picture\n"); exit(1);
      }
      colPoc = col->getPOC();
      if (!cs->isIntra()) { nCol = cs->getNumRefIdx(REF_PIC_LIST_0); for (int k = 0; k < nCol; k++) colPocs[k] = cs->getRefPOC(REF_PIC_LIST_0, k); }
    }
  }
  /* bind chain k to the picture with this slice's parameters, restricted to CTUs [a, a + n) (n = 0: the whole picture) */
  auto bind_chain = [&](int k, int a, int n) {
    int r2 = fcu_chain_begin(S.ctx, k, &fp, S.d_org[0], S.d_org[1], S.d_org[2], S.d_rec[0], S.d_rec[1], S.d_rec[2], S.d_out);
    if (r2 != FCU_OK) die("fcu_chain_begin", r2);
    if (fp.slice_type == FCU_SLICE_P) {
      r2 = fcu_chain_set_references(S.ctx, k, nRef, planes, pocs, slice->getPOC());
      if (r2 != FCU_OK) die("fcu_chain_set_references", r2);
      if (fp.tmvp) {
        r2 = fcu_chain_set_collocated(S.ctx, k, S.d_out_prev);
        if (r2 != FCU_OK) die("fcu_chain_set_collocated", r2);
        if (nCol) { r2 = fcu_chain_set_collocated_pocs(S.ctx, k, colPoc, colPocs, nCol); if (r2 != FCU_OK) die("fcu_chain_set_collocated_pocs", r2); }
      }
    }
    if (n > 0) { r2 = fcu_chain_set_range(S.ctx, k, a, n); if (r2 != FCU_OK) die("fcu_chain_set_range", r2); }
    if (k == 0) { r2 = fcu_chain_set_search_state(S.ctx, 0, S.search); if (r2 != FCU_OK) die("fcu_chain_set_search_state", r2); S.bound0 = true; }
  };
  auto same_lists = [&]() { if (S.ahead_nref != nRef) return false; for (int r = 0; r < nRef; r++) if (S.ahead_pocs[r] != pocs[r]) return false; return true; };
  if (nSlices > 1) {
    if (first == 0) {                                           /* first slice of the picture: decide all of them now */
      for (int k = 0; k < nSlices; k++) bind_chain(k, k * sliceArg, std::min(sliceArg, nCtuPic - k * sliceArg));
      rc = fcu_compress_chains(S.ctx, 0, nSlices, sliceArg, nullptr);
      if (rc != FCU_OK) die("fcu_compress_chains", rc);
      rc = fcu_sync(S.ctx);
      if (rc != FCU_OK) die("fcu_sync", rc);
      S.h_all.resize((size_t)nCtuPic);
      HIPOK(hipMemcpy(S.h_all.data(), S.d_out, sizeof(fcu_ctu_out) * (size_t)nCtuPic, hipMemcpyDeviceToHost));
      for (int c = 0; c < 3; c++) { S.h_rec[c].resize(plane_bytes(S.width, S.height, c)); HIPOK(hipMemcpy(S.h_rec[c].data(), S.d_rec[c], S.h_rec[c].size(), hipMemcpyDeviceToHost)); }
      S.ahead_last = nSlices - 1;                               /* HM's search state after the picture is the last slice's */
      S.ahead_valid = true; S.ahead_poc = pic->getPOC(); S.ahead_fp = fp; S.ahead_nref = nRef; for (int r = 0; r < nRef; r++) S.ahead_pocs[r] = pocs[r];
      S.serve_ahead = true;
      return;
    }
    if (S.ahead_valid && S.ahead_poc == pic->getPOC() && memcmp(&S.ahead_fp, &fp, sizeof(fp)) == 0 && same_lists()) { S.serve_ahead = true; return; }
    S.ahead_valid = false;                                      /* this slice was given other parameters: decide it (and the rest) slice by slice */
  }
  bind_chain(0, first, fp.slice_ctus > 0 ? count : 0);
}

} /* namespace */

/* test hook: the fcu_ctu_out record behind the CTU the last compressCtu filled, and whether it came from a picture-ahead launch */
extern "C" int fcu_adapter_last_record(void *dst) { if (!g_fcu.last) return -1; memcpy(dst, g_fcu.last, sizeof(fcu_ctu_out)); return g_fcu.serve_ahead ? 1 : 0; }

/* ---- the six public methods ------------------------------------------------------------------------------------------ */

Void TEncCu::init(TEncTop *pcEncTop)
{
  m_pcEncCfg = pcEncTop;
  m_pcPredSearch = pcEncTop->getPredSearch();
  m_pcTrQuant = pcEncTop->getTrQuant();
  m_pcRdCost = pcEncTop->getRdCost();
  m_pcEntropyCoder = pcEncTop->getEntropyCoder();
  m_pcBinCABAC = pcEncTop->getBinCABAC();
  m_pppcRDSbacCoder = pcEncTop->getRDSbacCoder();
  m_pcRDGoOnSbacCoder = pcEncTop->getRDGoOnSbacCoder();
  m_pcRateCtrl = pcEncTop->getRateCtrl();
}

Void TEncCu::create(UChar uhTotalDepth, UInt, UInt, ChromaFormat)
{
  /* the per-depth best / temp CU and YUV buffers of TEncCu.cpp live in device memory now (csrc/fcu_engine.h: Scratch);
   * the context needs the picture size, which arrives with the first slice (ensure_context) */
  m_uhTotalDepth = uhTotalDepth + 1;
  m_ppcBestCU = m_ppcTempCU = NULL;
  m_ppcPredYuvBest = m_ppcResiYuvBest = m_ppcRecoYuvBest = m_ppcPredYuvTemp = m_ppcResiYuvTemp = m_ppcRecoYuvTemp = m_ppcOrigYuv = NULL;
  m_bEncodeDQP = false; m_CodeChromaQpAdjFlag = false; m_ChromaQpAdjIdc = 0;
}

Void TEncCu::destroy()
{
  FcuState &S = g_fcu;
  for (int c = 0; c < 3; c++) { hipFree(S.d_org[c]); hipFree(S.d_rec[c]); hipFree(S.d_refsrc[c]); S.d_org[c] = S.d_rec[c] = S.d_refsrc[c] = nullptr;
    for (int k = 0; k < FcuState::N_SLOT; k++) { hipFree(S.d_ref[k][c]); S.d_ref[k][c] = nullptr; S.ref_valid[k] = false; } }
  hipFree(S.d_out); hipFree(S.d_out_prev); S.d_out = S.d_out_prev = nullptr; S.poc_prev = -1 << 30;
  if (S.ctx) { fcu_destroy(S.ctx); S.ctx = nullptr; }
  S.max_chains = 0; S.ahead_valid = S.serve_ahead = false; S.bound0 = false;
  for (int k = 0; k < 2 * FCU_MAX_REF; k++) S.search[k] = 0;      /* a new encoder */
  S.width = S.height = 0; S.poc_loaded = -1 << 30;
}

Void TEncCu::compressCtu(TComDataCU *pCtu)
{
  FcuState &S = g_fcu;
  TComPic *pic = pCtu->getPic(); TComSlice *slice = pCtu->getSlice();
  const UInt rs = pCtu->getCtuRsAddr();
  if (rs == pic->getPicSym()->getCtuTsToRsAddrMap(slice->getSliceCurStartCtuTsAddr())) begin_slice(pCtu, m_pcRdCost, m_pcTrQuant, m_pcEncCfg);
  const UInt wcA = pic->getFrameWidthInCtus();
  if (S.serve_ahead) {                                          /* decided with the first slice of the picture (picture-ahead) */
    fcu_adapter::marshal_ctu(S.h_all[rs], pCtu); S.last = &S.h_all[rs];
    for (int c = 0; c < 3; c++) fcu_adapter::widen_ctu_block(S.h_rec[c].data(), pic->getPicYuvRec(), ComponentID(c), rs, wcA);
    return;
  }
  const int rc = fcu_compress_ctu(S.ctx, 0, rs, &S.h_out);      /* compressCtu + the context replay of encodeCtu, result to the host */
  if (rc != FCU_OK) die("fcu_compress_ctu", rc);
  fcu_adapter::marshal_ctu(S.h_out, pCtu); S.last = &S.h_out;
  /* PicYuvRec: the CTU's block of the three planes (neighbouring CTUs and the loop filter read it on the host) */
  const UInt wc = pic->getFrameWidthInCtus();
  const int x0 = (int)(rs % wc) * 64, y0 = (int)(rs / wc) * 64;
  for (int c = 0; c < 3; c++) {
    const int sh = c ? 1 : 0, w = S.width >> sh, bx = x0 >> sh, by = y0 >> sh, bw = std::min(64 >> sh, w - bx), bh = std::min(64 >> sh, (S.height >> sh) - by);
    HIPOK(hipMemcpy2D(S.h_plane[c].data() + (size_t)by * w + bx, (size_t)w, S.d_rec[c] + (size_t)by * w + bx, (size_t)w, (size_t)bw, (size_t)bh, hipMemcpyDeviceToHost));
    fcu_adapter::widen_ctu_block(S.h_plane[c].data(), pic->getPicYuvRec(), ComponentID(c), rs, wc);
  }
}

Void TEncCu::encodeCtu(TComDataCU *pCtu)
{
  if (pCtu->getSlice()->getPPS()->getUseDQP()) setdQPFlag(true);
  if (pCtu->getSlice()->getUseChromaQpAdj()) setCodeChromaQpAdjFlag(true);
  xEncodeCU(pCtu, 0, 0);
}

/* rate control's intra complexity of a CTU: sum over its whole 8x8 blocks of the original luma of the 8x8 Hadamard
 * amplitude without DC, (sum + 2) >> 2 per block (TEncCu.cpp:1792-1893) */
Int TEncCu::updateCtuDataISlice(TComDataCU *pCtu, Int width, Int height)
{
  const Pel *org = pCtu->getPic()->getPicYuvOrg()->getAddr(COMPONENT_Y, pCtu->getCtuRsAddr(), 0);
  const Int stride = pCtu->getPic()->getPicYuvOrg()->getStride(COMPONENT_Y);
  Int total = 0;
  for (Int by = 0; by + 8 <= height; by += 8) for (Int bx = 0; bx + 8 <= width; bx += 8) {
    Int m[64];
    for (Int y = 0; y < 8; y++) for (Int x = 0; x < 8; x++) m[y * 8 + x] = org[(by + y) * stride + bx + x];
    for (Int pass = 0; pass < 2; pass++) {                     /* rows, then columns: three butterfly stages each */
      const Int step = pass ? 8 : 1, line = pass ? 1 : 8;
      for (Int l = 0; l < 8; l++) for (Int span = 4; span >= 1; span >>= 1)
        for (Int base = 0; base < 8; base += 2 * span) for (Int k = 0; k < span; k++) {
          Int &a = m[l * line + (base + k) * step], &b = m[l * line + (base + k + span) * step];
          const Int s = a + b, d = a - b; a = s; b = d;
        }
    }
    Int sum = 0;
    for (Int i = 1; i < 64; i++) sum += abs(m[i]);
    total += (sum + 2) >> 2;
  }
  return total;
}

/* ---- bitstream pass over a decided CTU -------------------------------------------------------------------------------- */

/* end of a coding unit: every CTU but the last of its slice segment ends with a zero terminating bin */
Void TEncCu::finishCU(TComDataCU *pcCU, UInt uiAbsPartIdx, UInt)
{
  if (!pcCU->isLastSubCUOfCtu(uiAbsPartIdx)) return;
  TComPic *pic = pcCU->getPic();
  const TComSlice *slice = pic->getSlice(pic->getCurrSliceIdx());
  const Int ts = (Int)pic->getPicSym()->getCtuRsToTsAddrMap(pcCU->getCtuRsAddr());
  if ((Int)slice->getSliceSegmentCurEndCtuTsAddr() != ts + 1) m_pcEntropyCoder->encodeTerminatingBit(0);
}

/* coding_quadtree() / coding_unit() of the CTU in the order of the standard.  A node inside the picture signals its split
 * flag; a node that crosses the picture border is split without one, and children that start outside do not exist. */
Void TEncCu::xEncodeCU(TComDataCU *pcCU, UInt uiAbsPartIdx, UInt uiDepth)
{
  const TComSlice *slice = pcCU->getSlice();
  const TComSPS *sps = slice->getSPS(); const TComPPS *pps = slice->getPPS();
  const UInt size = g_uiMaxCUWidth >> uiDepth, raster = g_auiZscanToRaster[uiAbsPartIdx];
  const UInt x = pcCU->getCUPelX() + g_auiRasterToPelX[raster], y = pcCU->getCUPelY() + g_auiRasterToPelY[raster];
  const Bool inside = x + size <= sps->getPicWidthInLumaSamples() && y + size <= sps->getPicHeightInLumaSamples();
  if (inside) m_pcEntropyCoder->encodeSplitFlag(pcCU, uiAbsPartIdx, uiDepth);
  const Bool dqpLevel = pps->getUseDQP() && size == (g_uiMaxCUWidth >> pps->getMaxCuDQPDepth());
  const Bool cqaLevel = slice->getUseChromaQpAdj() && size == (g_uiMaxCUWidth >> pps->getMaxCuChromaQpAdjDepth());
  if (!inside || (uiDepth < pcCU->getDepth(uiAbsPartIdx) && uiDepth < g_uiMaxCUDepth - g_uiAddCUDepth)) {
    if (dqpLevel) setdQPFlag(true);
    if (cqaLevel) setCodeChromaQpAdjFlag(true);
    const UInt quarter = (pcCU->getPic()->getNumPartitionsInCtu() >> (2 * uiDepth)) >> 2;
    for (UInt k = 0; k < 4; k++) {
      const UInt part = uiAbsPartIdx + k * quarter, r = g_auiZscanToRaster[part];
      if (pcCU->getCUPelX() + g_auiRasterToPelX[r] < sps->getPicWidthInLumaSamples() && pcCU->getCUPelY() + g_auiRasterToPelY[r] < sps->getPicHeightInLumaSamples())
        xEncodeCU(pcCU, part, uiDepth + 1);
    }
    return;
  }
  if (pps->getUseDQP() && size >= (g_uiMaxCUWidth >> pps->getMaxCuDQPDepth())) setdQPFlag(true);
  if (slice->getUseChromaQpAdj() && size >= (g_uiMaxCUWidth >> pps->getMaxCuChromaQpAdjDepth())) setCodeChromaQpAdjFlag(true);
  if (pps->getTransquantBypassEnableFlag()) m_pcEntropyCoder->encodeCUTransquantBypassFlag(pcCU, uiAbsPartIdx);
  if (!slice->isIntra()) m_pcEntropyCoder->encodeSkipFlag(pcCU, uiAbsPartIdx);
  if (pcCU->isSkipped(uiAbsPartIdx)) {
    m_pcEntropyCoder->encodeMergeIndex(pcCU, uiAbsPartIdx);
    finishCU(pcCU, uiAbsPartIdx, uiDepth);
    return;
  }
  m_pcEntropyCoder->encodePredMode(pcCU, uiAbsPartIdx);
  m_pcEntropyCoder->encodePartSize(pcCU, uiAbsPartIdx, uiDepth);
  if (pcCU->isIntra(uiAbsPartIdx) && pcCU->getPartitionSize(uiAbsPartIdx) == SIZE_2Nx2N) {
    m_pcEntropyCoder->encodeIPCMInfo(pcCU, uiAbsPartIdx);
    if (pcCU->getIPCMFlag(uiAbsPartIdx)) { finishCU(pcCU, uiAbsPartIdx, uiDepth); return; }
  }
  m_pcEntropyCoder->encodePredInfo(pcCU, uiAbsPartIdx);
  Bool codeDQP = getdQPFlag(), codeCQA = getCodeChromaQpAdjFlag();
  m_pcEntropyCoder->encodeCoeff(pcCU, uiAbsPartIdx, uiDepth, codeDQP, codeCQA);
  setCodeChromaQpAdjFlag(codeCQA);
  setdQPFlag(codeDQP);
  finishCU(pcCU, uiAbsPartIdx, uiDepth);
}
