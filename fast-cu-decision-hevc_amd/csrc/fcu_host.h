/*
 * fcu_host.h -- host-side derivation of the per-chain parameters that HM computes once per
 * slice with libm (lambda, sqrt(lambda), chroma distortion weight, RDOQ lambdas, RDOQ error
 * scales, sign-hiding rdFactor).  They are handed to the kernels as f64/i64 bit patterns so
 * that no transcendental is ever evaluated on the device.
 */
#pragma once
#include <math.h>
#include "fcu_engine.h"

namespace fcu {

inline void fill_params(Params &p, int width, int height, const fcu_frame_params &fp)
{
  p.width = width; p.height = height; p.qp = fp.qp; p.slice_ctus = fp.slice_ctus;
  p.slice_type = fp.slice_type; p.search_range = fp.search_range; p.fast_enc = fp.fast_enc; p.had_me = fp.hadamard_me;
  p.fast_search = fp.fast_search; p.rdoq = fp.rdoq; p.rdoq_ts = fp.rdoq_ts; p.tmvp = fp.tmvp; p.amp = fp.amp != 0;
  p.cabac_b_table = fp.slice_type == FCU_SLICE_P && fp.cabac_b_table != 0;
  p.fdm = fp.fast_merge_decision; p.max_merge_cand = fp.max_merge_cand > 0 ? (fp.max_merge_cand > 5 ? 5 : fp.max_merge_cand) : 5;
  p.transform_skip = fp.transform_skip; p.ts_fast = fp.transform_skip_fast;
  p.sign_hiding = fp.sign_hiding; p.strong_smoothing = fp.strong_intra_smoothing;
  /* chroma QP: g_aucChromaScale[CHROMA_420] (TLibCommon/TComRom.cpp:507), offsets 0 */
  static const unsigned char chroma_scale[58] = {
    0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29,
    29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37, 37, 38, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 51 };
  p.qp_c = fp.qp < 0 ? fp.qp : chroma_scale[fp.qp > 57 ? 57 : fp.qp];
  if (fp.lambda > 0.0) {
    const double w = fp.chroma_weight > 0.0 ? fp.chroma_weight : pow(2.0, (fp.qp - p.qp_c) / 3.0);     /* setUpLambda, TEncSlice.cpp:496-524 */
    p.lambda = fp.lambda; p.sqrt_lambda = fp.sqrt_lambda > 0.0 ? fp.sqrt_lambda : sqrt(fp.lambda); p.chroma_weight = w;
    p.rdoq_lambda[0] = fp.rdoq_lambda[0] > 0.0 ? fp.rdoq_lambda[0] : fp.lambda;
    p.rdoq_lambda[1] = fp.rdoq_lambda[1] > 0.0 ? fp.rdoq_lambda[1] : fp.lambda / w;
    p.rdoq_lambda[2] = fp.rdoq_lambda[2] > 0.0 ? fp.rdoq_lambda[2] : fp.lambda / w;
  } else {
    /* TEncSlice::initEncSlice, I slice: lambda = 0.57 * 2^((QP-12)/3)  (TEncSlice.cpp:686-706) */
    const double lambda = 0.57 * pow(2.0, ((double)fp.qp - 12) / 3.0);
    const double w = pow(2.0, (fp.qp - p.qp_c) / 3.0);             /* setUpLambda, TEncSlice.cpp:496-524 */
    p.lambda = lambda; p.sqrt_lambda = sqrt(lambda); p.chroma_weight = w;
    p.rdoq_lambda[0] = lambda; p.rdoq_lambda[1] = lambda / w; p.rdoq_lambda[2] = lambda / w;
  }
  p.lambda_motion_sad = (uint32_t)floor(65536.0 * sqrt(p.lambda));      /* TComRdCost::setLambda, TComRdCost.cpp:194-219 */
  for (int ch = 0; ch < 2; ch++) {
    const int qp = ch ? p.qp_c : p.qp, per = qp / 6, rem = qp % 6;
    static const int quant_scales[6] = { 26214, 23302, 20560, 18396, 16384, 14564 };
    static const int inv_quant_scales[6] = { 40, 45, 51, 57, 64, 72 };
    for (int l = 0; l < 4; l++) {
      /* setErrScaleCoeff, TComTrQuant.cpp:3018-3040: 2^15 * 2^(-2*transformShift) / q / q */
      const int tshift = 15 - 8 - (l + 2);
      double e = (double)(1 << 15);
      e = e * ldexp(1.0, -2 * tshift);
      e = e / quant_scales[rem] / quant_scales[rem] / 1;
      p.err_scale[ch][l] = e;
    }
    /* rdFactor of sign-bit hiding, TComTrQuant.cpp:2444-2447 (lambda = m_dLambda of the component) */
    const double invQ = (double)inv_quant_scales[rem];
    const double lam = p.rdoq_lambda[ch ? 1 : 0];
    p.rd_factor[ch] = (long long)(invQ * invQ * (double)(1 << (2 * per)) / lam / 16 / 1 + 0.5);
  }
}

inline void default_frame_params(fcu_frame_params &fp, int qp)
{
  memset(&fp, 0, sizeof(fp));
  fp.qp = qp; fp.slice_ctus = 0; fp.transform_skip = 1; fp.transform_skip_fast = 1; fp.sign_hiding = 1; fp.strong_intra_smoothing = 1;
  fp.slice_type = FCU_SLICE_I; fp.search_range = 64; fp.fast_enc = 1; fp.hadamard_me = 1; fp.fast_merge_decision = 1; fp.max_merge_cand = 5; fp.fast_search = 0; fp.tmvp = 0; fp.rdoq = 1; fp.rdoq_ts = 1; fp.amp = 0;
}
/* TEncSlice::initEncSlice for HM's lowdelay_P GOP table (GOPSize 4): slice type, QP = base + offset, lambda =
 * QPFactor * 2^((QP-12)/3), x Clip3(2, 4, (QP-12)/6) at temporal depth > 0 (POC % 4 != 0); the I picture's factor is
 * 0.57 * (1 - 0.05 * (GOPSize - 1)) (TEncSlice.cpp:560-740) */
inline void ldp_slice(fcu_frame_params &fp, int base_qp, int poc)
{
  static const int qp_off[4] = { 3, 2, 3, 1 };
  static const double qp_fac[4] = { 0.4624, 0.4624, 0.4624, 0.578 };
  default_frame_params(fp, base_qp);
  if (poc == 0) { fp.lambda = 0.57 * (1.0 - 0.05 * 3) * pow(2.0, ((double)base_qp - 12) / 3.0); return; }
  const int g = (poc - 1) % 4, qp = base_qp + qp_off[g];
  double lambda = qp_fac[g] * pow(2.0, ((double)qp - 12) / 3.0);
  if (poc % 4 != 0) { double f = ((double)qp - 12) / 6.0; f = f < 2.0 ? 2.0 : (f > 4.0 ? 4.0 : f); lambda *= f; }
  fp.slice_type = FCU_SLICE_P; fp.qp = qp; fp.lambda = lambda;
}

} // namespace fcu
