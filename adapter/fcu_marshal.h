/*
 * fcu_marshal.h -- the data conversions of the HM adapter (adapter/TEncCuFcu.cpp), header only, no device calls:
 *   * fcu_ctu_out  ->  the picture's TComDataCU of that CTU, i.e. what xCompressCU's copyToPic calls leave behind
 *     (TComDataCU::copyToPic, TLibCommon/TComDataCU.cpp:992-1065; arrays TComDataCU.h:72-164);
 *   * TComPicYuv planes (Pel = int16, stride = width + 2 * margin, TComPicYuv.cpp:81-97)  <->  the dense 8-bit planes
 *     of the C ABI (include/fcu.h).
 * Compiled against the reference's own headers; nothing here restates an algorithm of the reference.
 */
#pragma once
#include <string.h>
#include <stdint.h>
#include "TLibCommon/CommonDef.h"
#include "TLibCommon/TComRom.h"
#include "TLibCommon/TComPic.h"
#include "TLibCommon/TComPicYuv.h"
#include "TLibCommon/TComDataCU.h"
#include "fcu.h"

namespace fcu_adapter {

/* Every per-partition array of the CTU's TComDataCU (256 partitions of 4x4 luma samples, z-order) and its totals.  Fields
 * the configurations behind the ABI never switch on get the value TComDataCU::initCtu gives them (TComDataCU.cpp:449-520). */
inline void marshal_ctu(const fcu_ctu_out &o, TComDataCU *ctu)
{
  const UInt n = ctu->getTotalNumPart();                       /* 256 */
  memcpy(ctu->getDepth(), o.depth, n);
  memcpy(ctu->getWidth(), o.width, n);
  memcpy(ctu->getHeight(), o.height, n);
  memcpy(ctu->getPartitionSize(), o.part_size, n);
  memcpy(ctu->getPredictionMode(), o.pred_mode, n);
  memcpy(ctu->getQP(), o.qp, n);
  memcpy(ctu->getChromaQpAdj(), o.chroma_qp_adj, n);
  memcpy(ctu->getTransformIdx(), o.tr_idx, n);
  memcpy(ctu->getIntraDir(CHANNEL_TYPE_LUMA), o.intra_dir[0], n);
  memcpy(ctu->getIntraDir(CHANNEL_TYPE_CHROMA), o.intra_dir[1], n);
  memcpy(ctu->getMergeIndex(), o.merge_idx, n);
  memcpy(ctu->getInterDir(), o.inter_dir, n);
  for (UInt i = 0; i < n; i++) {                               /* Bool arrays */
    ctu->getSkipFlag()[i] = o.skip[i] != 0;
    ctu->getCUTransquantBypass()[i] = o.tq_bypass[i] != 0;
    ctu->getMergeFlag()[i] = o.merge_flag[i] != 0;
    ctu->getIPCMFlag()[i] = o.ipcm[i] != 0;
  }
  for (UInt c = 0; c < MAX_NUM_COMPONENT; c++) {
    const ComponentID comp = ComponentID(c);
    memcpy(ctu->getCbf(comp), o.cbf[c], n);
    memcpy(ctu->getTransformSkip(comp), o.tskip[c], n);
    memset(ctu->getCrossComponentPredictionAlpha(comp), 0, n);
    memset(ctu->getExplicitRdpcmMode(comp), NUMBER_OF_RDPCM_MODES, n);
  }
  /* motion of list 0; list 1 stays "no reference" (P slices) */
  TComCUMvField *l0 = ctu->getCUMvField(REF_PIC_LIST_0), *l1 = ctu->getCUMvField(REF_PIC_LIST_1);
  for (UInt i = 0; i < n; i++) {
    const bool inter = o.pred_mode[i] == MODE_INTER;
    l0->setAllMv(TComMv(inter ? o.mv[i][0] : 0, inter ? o.mv[i][1] : 0), SIZE_2Nx2N, (Int)i, 4);      /* depth 4: one partition */
    l0->setAllMvd(TComMv(inter ? o.mvd[i][0] : 0, inter ? o.mvd[i][1] : 0), SIZE_2Nx2N, (Int)i, 4);
    l0->setAllRefIdx(inter ? o.ref_idx[i] : NOT_VALID, SIZE_2Nx2N, (Int)i, 4);
    l1->setAllMv(TComMv(0, 0), SIZE_2Nx2N, (Int)i, 4); l1->setAllMvd(TComMv(0, 0), SIZE_2Nx2N, (Int)i, 4);
    l1->setAllRefIdx(NOT_VALID, SIZE_2Nx2N, (Int)i, 4);
    ctu->getMVPIdx(REF_PIC_LIST_0)[i] = inter ? o.mvp_idx[i] : -1;
    ctu->getMVPNum(REF_PIC_LIST_0)[i] = (inter && !o.merge_flag[i]) ? AMVP_MAX_NUM_CANDS : -1;
    ctu->getMVPIdx(REF_PIC_LIST_1)[i] = -1; ctu->getMVPNum(REF_PIC_LIST_1)[i] = -1;
  }
  /* quantised levels: TCoeff is int32 like the ABI's arrays, TU-contiguous at partition * 16 (luma) / * 4 (chroma) */
  memcpy(ctu->getCoeff(COMPONENT_Y), o.coeff_y, sizeof(o.coeff_y));
  memcpy(ctu->getCoeff(COMPONENT_Cb), o.coeff_cb, sizeof(o.coeff_cb));
  memcpy(ctu->getCoeff(COMPONENT_Cr), o.coeff_cr, sizeof(o.coeff_cr));
  ctu->getTotalCost() = o.total_cost;
  ctu->getTotalDistortion() = o.total_dist;
  ctu->getTotalBits() = o.total_bits;
  ctu->getTotalBins() = o.total_bins;
}

/* whole plane of a TComPicYuv -> dense 8-bit plane (stride = plane width).  The reference holds 8-bit video in Pel. */
inline void narrow_plane(TComPicYuv *pic, ComponentID c, uint8_t *dst)
{
  const Pel *s = pic->getAddr(c); const Int stride = pic->getStride(c), w = pic->getWidth(c), h = pic->getHeight(c);
  for (Int y = 0; y < h; y++, s += stride, dst += w) for (Int x = 0; x < w; x++) dst[x] = (uint8_t)s[x];
}
/* the block of CTU `ctuRsAddr` of a dense 8-bit plane -> the same block of a TComPicYuv (PicYuvRec after compressCtu) */
inline void widen_ctu_block(const uint8_t *src, TComPicYuv *pic, ComponentID c, UInt ctuRsAddr, UInt frameWidthInCtus)
{
  const Int sx = pic->getComponentScaleX(c), sy = pic->getComponentScaleY(c);
  const Int w = pic->getWidth(c), h = pic->getHeight(c), stride = pic->getStride(c);
  const Int x0 = (Int)((ctuRsAddr % frameWidthInCtus) * g_uiMaxCUWidth) >> sx, y0 = (Int)((ctuRsAddr / frameWidthInCtus) * g_uiMaxCUHeight) >> sy;
  const Int bw = std::min<Int>((Int)g_uiMaxCUWidth >> sx, w - x0), bh = std::min<Int>((Int)g_uiMaxCUHeight >> sy, h - y0);
  Pel *d = pic->getAddr(c) + (size_t)y0 * stride + x0;
  const uint8_t *s = src + (size_t)y0 * w + x0;
  for (Int y = 0; y < bh; y++, d += stride, s += w) for (Int x = 0; x < bw; x++) d[x] = (Pel)s[x];
}

} /* namespace fcu_adapter */
