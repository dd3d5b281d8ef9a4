/*
 * fcu_obf.h -- the fork's pre-pass: outlier-block-flag (OBF) map of a luma plane
 * (TEncSlice::getOutlierWithDCT, Lib/TLibEncoder/TEncSlice.cpp:878-1173; called once per picture from
 * TEncGOP.cpp:1096, read per CU at TEncCu.cpp:585-603).  Included by fcu_kernels.hip only.
 *
 * Two HBM-bound kernels around a tiny host step:
 *   obf_hist   every 4x4 block: forward 4x4 DCT (partialButterfly x2, shifts 1 and 8), amplitude |coef/8| of each
 *              of the 15 AC frequencies into a per-frame histogram (LDS-privatised low bins, global atomics above)
 *   host       per frequency: the transparent-composite-model fit TCMprocessOneSequence (:343-392) on the
 *              histogram -> threshold Yc.  It is double arithmetic with exp/log; it runs on the host with the C
 *              library the reference itself would use, so the argmax over likelihoods is bit-identical
 *   obf_count  every block again (recomputing the DCT costs less traffic than storing 15 coefficients per block):
 *              OBF = number of AC coefficients with |coef| >= Yc*8 (BINARIZE_OBF 0), one int16 per block
 * Algorithmic bytes per frame: read W*H (hist) + read W*H, write W*H/8 (count).
 */
#pragma once
#include <math.h>
#include <thread>
#include <vector>

namespace fcu {

enum { OBF_HB = 4096, OBF_LB = 256, OBF_THREADS = 256, OBF_GROUPS_PER_THREAD = 2 };   /* |coef/8| <= 4080 for 8-bit sources */

/* forward 4x4 DCT of one block of source samples: xTrMxN with partialButterfly4 (TComTrQuant.cpp:388-412,860-915),
 * first stage shift 1 (log2 + bitDepth + 6 - 15), second stage shift 8 (log2 + 6); coef[k2*4 + k1].
 * r[y] = the block's row y as four packed samples. */
__device__ static inline void obf_dct4(const uint32_t r[4], int coef[16])
{
  int tmp[16];
#pragma unroll
  for (int y = 0; y < 4; y++) {
    const uint32_t w = r[y];
    const int s0 = w & 255, s1 = (w >> 8) & 255, s2 = (w >> 16) & 255, s3 = w >> 24;
    const int e0 = s0 + s3, o0 = s0 - s3, e1 = s1 + s2, o1 = s1 - s2;
    tmp[0 * 4 + y] = (64 * e0 + 64 * e1 + 1) >> 1;
    tmp[2 * 4 + y] = (64 * e0 - 64 * e1 + 1) >> 1;
    tmp[1 * 4 + y] = (83 * o0 + 36 * o1 + 1) >> 1;
    tmp[3 * 4 + y] = (36 * o0 - 83 * o1 + 1) >> 1;
  }
#pragma unroll
  for (int k1 = 0; k1 < 4; k1++) {
    const int e0 = tmp[k1 * 4 + 0] + tmp[k1 * 4 + 3], o0 = tmp[k1 * 4 + 0] - tmp[k1 * 4 + 3];
    const int e1 = tmp[k1 * 4 + 1] + tmp[k1 * 4 + 2], o1 = tmp[k1 * 4 + 1] - tmp[k1 * 4 + 2];
    coef[0 * 4 + k1] = (64 * e0 + 64 * e1 + 128) >> 8;
    coef[2 * 4 + k1] = (64 * e0 - 64 * e1 + 128) >> 8;
    coef[1 * 4 + k1] = (83 * o0 + 36 * o1 + 128) >> 8;
    coef[3 * 4 + k1] = (36 * o0 - 83 * o1 + 128) >> 8;
  }
}

/* A thread owns groups of four horizontally adjacent blocks: four 16-byte row loads (consecutive lanes read
 * consecutive 16 bytes of a picture row), four DCTs.  Width is a multiple of 8 (SPS), so a row holds an even
 * number of blocks; the last group of a row may hold fewer than four. */
struct ObfGroup { uint4 row[4]; int nb; };
__device__ static inline ObfGroup obf_load_group(const uint8_t *plane, int w, int gpr, int gidx)
{
  ObfGroup g;
  const int gy = gidx / gpr, gx = gidx - gy * gpr, bw = w >> 2;
  g.nb = bw - gx * 4 < 4 ? bw - gx * 4 : 4;
  const uint8_t *p = plane + (size_t)(gy * 4) * w + gx * 16;
#pragma unroll
  for (int y = 0; y < 4; y++) {
    if (g.nb == 4) g.row[y] = *(const uint4 *)(p + (size_t)y * w);
    else { const uint2 t = *(const uint2 *)(p + (size_t)y * w); g.row[y] = make_uint4(t.x, t.y, 0u, 0u); }
  }
  return g;
}
__device__ static inline uint32_t obf_word(const uint4 &v, int j) { return j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w)); }

/* grid.y = frame; hist[frame][15][OBF_HB].  Zero amplitudes (the bulk on smooth content) are counted in registers and
 * reduced over the wave; the rest go through LDS atomics (amplitudes < OBF_LB) or global atomics. */
__global__ void __launch_bounds__(OBF_THREADS) obf_hist(const uint8_t *y, int w, int h, size_t frame_bytes, unsigned *hist)
{
  __shared__ unsigned lh[15 * OBF_LB];
  for (int i = threadIdx.x; i < 15 * OBF_LB; i += OBF_THREADS) lh[i] = 0;
  __syncthreads();
  const int gpr = ((w >> 2) + 3) >> 2, ngrp = gpr * (h >> 2);
  const uint8_t *plane = y + (size_t)blockIdx.y * frame_bytes;
  unsigned *gh = hist + (size_t)blockIdx.y * 15 * OBF_HB;
  unsigned zeros[15];
#pragma unroll
  for (int x = 0; x < 15; x++) zeros[x] = 0;
  const int lane = threadIdx.x & 63;
  for (int k = 0; k < OBF_GROUPS_PER_THREAD; k++) {
    const int gi = (blockIdx.x * OBF_GROUPS_PER_THREAD + k) * OBF_THREADS + threadIdx.x;
    const bool live = gi < ngrp;
    ObfGroup g; g.nb = 0;
    if (live) g = obf_load_group(plane, w, gpr, gi);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const bool on = j < g.nb;
      int coef[16];
      const uint32_t r[4] = { obf_word(g.row[0], j), obf_word(g.row[1], j), obf_word(g.row[2], j), obf_word(g.row[3], j) };
      obf_dct4(r, coef);
#pragma unroll
      for (int x = 1; x < 16; x++) {
        const int c = coef[x] / 8;                              /* CoeffFrequency = coeff / DctScaling, truncated (:955) */
        const int a = on ? (c < 0 ? -c : c) : -1;
        zeros[x - 1] += (a == 0);
        if (a > 0) {
          if (a < OBF_LB) atomicAdd(&lh[(x - 1) * OBF_LB + a], 1u);
          else atomicAdd(&gh[(x - 1) * OBF_HB + (a < OBF_HB ? a : OBF_HB - 1)], 1u);
        }
      }
    }
  }
#pragma unroll
  for (int x = 0; x < 15; x++) {                                /* zero counts: reduce over the wave, one LDS atomic per wave */
    unsigned z = zeros[x];
    for (int o = 32; o > 0; o >>= 1) z += __shfl_down(z, o, 64);
    if (lane == 0 && z) atomicAdd(&lh[x * OBF_LB], z);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 15 * OBF_LB; i += OBF_THREADS) {
    const unsigned v = lh[i];
    if (v) atomicAdd(&gh[(i / OBF_LB) * OBF_HB + (i % OBF_LB)], v);
  }
}

/* thr[frame][16] = Yc * 8 as an integer (Yc is a bucket index); OBF = number of AC coefficients with |c| >= thr,
 * c != 0  <=>  not (c < Yc*8 && c > -Yc*8)  (:1009-1038) */
__global__ void __launch_bounds__(OBF_THREADS) obf_count(const uint8_t *y, int w, int h, size_t frame_bytes, const int *thr, int16_t *obf)
{
  const int bw = w >> 2, gpr = (bw + 3) >> 2, ngrp = gpr * (h >> 2);
  const uint8_t *plane = y + (size_t)blockIdx.y * frame_bytes;
  int t[16];
#pragma unroll
  for (int x = 0; x < 16; x++) t[x] = thr[blockIdx.y * 16 + x];
  int16_t *out = obf + (size_t)blockIdx.y * bw * (h >> 2);
  for (int k = 0; k < OBF_GROUPS_PER_THREAD; k++) {
    const int gi = (blockIdx.x * OBF_GROUPS_PER_THREAD + k) * OBF_THREADS + threadIdx.x;
    if (gi >= ngrp) break;
    const ObfGroup g = obf_load_group(plane, w, gpr, gi);
    const int gy = gi / gpr, gx = gi - gy * gpr;
    int16_t res[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      int coef[16];
      const uint32_t r[4] = { obf_word(g.row[0], j), obf_word(g.row[1], j), obf_word(g.row[2], j), obf_word(g.row[3], j) };
      obf_dct4(r, coef);
      int n = 0;
#pragma unroll
      for (int x = 1; x < 16; x++) { const int c = coef[x], a = c < 0 ? -c : c; n += (c != 0 && a >= t[x]); }
      res[j] = (int16_t)n;
    }
    int16_t *o = out + (size_t)gy * bw + gx * 4;
    if (g.nb == 4) *(uint2 *)o = make_uint2((uint32_t)(uint16_t)res[0] | ((uint32_t)(uint16_t)res[1] << 16), (uint32_t)(uint16_t)res[2] | ((uint32_t)(uint16_t)res[3] << 16));
    else { o[0] = res[0]; o[1] = res[1]; }
  }
}

/* ---- host: TCMprocessOneSequence and helpers, TEncSlice.cpp:194-392 ---------------------------------------- */
struct TcmBucket { int count; double acum_abs_amp, acum_samp_num, prob, lambda, likelyhood; };

static double tcm_lambda_given_yc(double yc, double sum_yi, double total)
{
  const double c = sum_yi / total;
  double lambda, lambda_old;
  if (c / yc >= 0.95) return -1.0;                              /* too far from Laplacian */
  lambda_old = c;
  lambda = c - yc * (1.0 - 1.0 / (1.0 - exp(-yc / lambda_old)));
  for (int k = 0; k < 5; k++) { lambda_old = lambda; lambda = c - yc * (1.0 - 1.0 / (1.0 - exp(-yc / lambda_old))); }
  while (fabs(lambda - lambda_old) > 0.1) { lambda_old = lambda; lambda = c - yc * (1.0 - 1.0 / (1.0 - exp(-yc / lambda_old))); }
  return lambda;
}
static void tcm_likelyhood(int point, int n, TcmBucket *b, int peak)
{
  const double n1 = b[point].acum_samp_num, n2 = n - n1, yc = point;
  const double sum_yi = b[point].acum_abs_amp, total = b[point].acum_samp_num;
  const double lambda = tcm_lambda_given_yc(yc, sum_yi, total);
  const double prob = (double)n1 / (double)n;
  if (lambda > 0) {
    b[point].likelyhood = n2 * log(1 - prob) + n1 * log(prob) - n2 * log((peak - yc) * 2.0)
                          - n1 * log(1 - exp(-yc / lambda))
                          - n1 * log(2 * lambda) - sum_yi / lambda;
    b[point].lambda = lambda; b[point].prob = prob;
  } else { b[point].likelyhood = 1.e30; b[point].lambda = lambda; b[point].prob = 1; }   /* -MinLikelyhood (:276) */
}
/* Yc of one frequency from its amplitude histogram (len samples); 0 when the data are all zero */
static double tcm_threshold(const unsigned *hist, int len)
{
  int peak = 0;
  for (int a = OBF_HB - 1; a > 0; a--) if (hist[a]) { peak = a; break; }
  if (peak == 0) return 0.0;
  std::vector<TcmBucket> b((size_t)peak + 1);
  for (int k = 0; k <= peak; k++) { b[k].count = (int)hist[k]; b[k].acum_abs_amp = 0; b[k].acum_samp_num = 0; }
  b[0].acum_samp_num = b[0].count;
  for (int k = 1; k <= peak; k++) {
    b[k].acum_abs_amp = b[k - 1].acum_abs_amp + k * b[k].count;
    b[k].acum_samp_num = b[k - 1].acum_samp_num + b[k].count;
  }
  int start;                                                    /* FindStartPoint (:224-247) */
  for (start = peak; start > 0; start--) {
    if (b[start].count == 0) continue;
    if (b[start].acum_samp_num < len * (1.0 - 0.1)) break;
  }
  const int q = len / 100;
  auto cnt = [&](int k) { return k <= peak ? b[k].count : 0; };
  if (cnt(0) > q && cnt(1) > q && cnt(2) > q && cnt(3) > q) { if (start < 3) start = 3; }
  else if (cnt(0) > q && cnt(1) > q && cnt(2) > q) { if (start < 2) start = 2; }
  else { if (start < 1) start = 1; }
  if (start > peak) start = peak;                               /* the reference would read a cleared bucket beyond the peak */
  tcm_likelyhood(start, len, b.data(), peak);
  double max_l = b[start].likelyhood; int max_pos = start;
  for (int k = start + 1; k <= peak; k++) {
    if (b[k].count == 0) continue;
    tcm_likelyhood(k, len, b.data(), peak);
    if (b[k].likelyhood > max_l) { max_pos = k; max_l = b[k].likelyhood; }
  }
  return max_l > -1.e30 ? (double)max_pos : 0.0;
}

} // namespace fcu
