/*
 * sao_emu.cpp -- TEST-ONLY: the SAO kernel source (csrc/fcu_sao.h) compiled for the CPU with the HIP keywords defined
 * away and every grid run as a loop (workgroup phases in order, threads inside a phase in order), so that indexing and
 * arithmetic can be checked against the oracle / the reference's golden vectors without a GPU.  Not part of libfcu.so.
 */
#define FCU_EMU 1
#include <vector>
#include <cstring>
#include "../../fast-cu-decision-hevc_amd/csrc/fcu_host.h"
#define __device__
struct Dim3 { unsigned x, y, z; };
static thread_local Dim3 blockIdx, threadIdx;
static inline void atomicAdd(int32_t *p, int v) { *p += v; }
#include "../../fast-cu-decision-hevc_amd/csrc/fcu_sao.h"

using namespace fcu;

/* one picture; mirrors the launches of fcu_sao (fcu_kernels.hip) */
extern "C" void sao_emu(int w, int h, int slice_type, int qp, int slice_ctus, const double *lambda, const int *enabled,
                        const uint8_t *oy, const uint8_t *ou, const uint8_t *ov, uint8_t *ry, uint8_t *ru, uint8_t *rv,
                        fcu_sao_ctu *coded, int32_t *off_count, int32_t *stats_out)
{
  const int w_ctu = (w + 63) / 64, n_ctu = w_ctu * ((h + 63) / 64);
  const size_t plane[3] = { (size_t)w * h, (size_t)(w / 2) * (h / 2), (size_t)(w / 2) * (h / 2) };
  std::vector<uint8_t> src[3];
  uint8_t *rec[3] = { ry, ru, rv }; const uint8_t *org[3] = { oy, ou, ov };
  SaoPic P;
  for (int k = 0; k < 3; k++) {
    src[k].assign(rec[k], rec[k] + plane[k]);
    P.org[k] = org[k]; P.rec[k] = rec[k]; P.src[k] = src[k].data(); P.lambda[k] = lambda[k]; P.enabled[k] = enabled[k];
  }
  P.slice_type = slice_type; P.qp = qp; P.slice_ctus = slice_ctus;
  std::vector<int32_t> stats((size_t)n_ctu * 3 * SAO_STAT_INTS);
  std::vector<SaoCand> cands((size_t)n_ctu * 15);
  std::vector<fcu_sao_ctu> recon((size_t)n_ctu);
  int32_t hist[SAO_STAT_INTS];
  blockIdx.z = 0;
  for (unsigned a = 0; a < (unsigned)n_ctu; a++) for (unsigned comp = 0; comp < 3; comp++) {
    blockIdx.x = a; blockIdx.y = comp;
    for (unsigned t = 0; t < SAO_THREADS; t++) { threadIdx.x = t; sao_stats_phase<0>(hist, &P, stats.data(), w, h, w_ctu, n_ctu); }
    for (unsigned t = 0; t < SAO_THREADS; t++) { threadIdx.x = t; sao_stats_phase<1>(hist, &P, stats.data(), w, h, w_ctu, n_ctu); }
    for (unsigned t = 0; t < SAO_THREADS; t++) { threadIdx.x = t; sao_stats_phase<2>(hist, &P, stats.data(), w, h, w_ctu, n_ctu); }
  }
  blockIdx.y = 0;
  for (unsigned b = 0; b < ((unsigned)n_ctu * 15 + SAO_THREADS - 1) / SAO_THREADS; b++)
    for (unsigned t = 0; t < SAO_THREADS; t++) { blockIdx.x = b; threadIdx.x = t; sao_cands_thread(&P, stats.data(), cands.data(), n_ctu, 1); }
  static SaoDecideLds lds;
  sao_decide_picture(P, stats.data(), cands.data(), coded, recon.data(), off_count, w_ctu, n_ctu, lds);
  for (unsigned a = 0; a < (unsigned)n_ctu; a++) for (unsigned comp = 0; comp < 3; comp++)
    for (unsigned t = 0; t < SAO_THREADS; t++) { blockIdx.x = a; blockIdx.y = comp; threadIdx.x = t; sao_apply_block(&P, recon.data(), w, h, w_ctu, n_ctu); }
  if (stats_out) memcpy(stats_out, stats.data(), stats.size() * sizeof(int32_t));
}
