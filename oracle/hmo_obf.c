/*
 * hmo_obf.c -- ORACLE (test infrastructure, never shipped / never on the product path).
 *
 * CPU restatement of the fork's "outlier block flag" pre-pass, TEncSlice::getOutlierWithDCT
 * (Lib/TLibEncoder/TEncSlice.cpp:878-1173) with the transparent-composite-model threshold fit
 * TCMprocessOneSequence (:343-392) and its helpers ComputeLambdaGivenYc (:194-221),
 * FindStartPoint (:224-247), ComputeLikelyhood (:249-281).  Build switches as in the reference:
 * GEN_OUTLIER 1, DCT_SIZE_IS_FOUR 1, BINARIZE_OBF 0 (TypeDef.h:119-128), StartFreq 1.
 *
 * Parity: UNPINNED above the transform (the pre-pass lives in TEncSlice.cpp, which cannot be built here);
 * the 4x4 forward DCT it calls (partialButterfly, TComTrQuant.cpp:388) is the leaf-pinned hmo_fwd_transform.
 * The reference's bucket array has 10000 entries (:346): amplitudes >= 10000 are undefined behaviour there and
 * are reported as an error here (they cannot occur for 8-bit sources: |coeff/8| <= 4080).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "hmo.h"

#define TCM_MAX_AMP 65536          /* MaxAmp, TEncSlice.cpp:178 */
#define TCM_LAMBDA_DELTA 0.1       /* LambdaDelta :179 */
#define TCM_MIN_LIKELYHOOD (-1.e30)/* MinLikelyhood :180 */
#define TCM_START_POINT_PROB 0.1   /* StartPointProb :182 */
#define TCM_BUCKETS 10000

typedef struct { int count; double acum_abs_amp, acum_samp_num, prob, lambda, likelyhood; } Bucket;

static double lambda_given_yc(double yc, double sum_yi, double total)          /* :194-221 */
{
  const double c = sum_yi / total;
  double lambda, lambda_old;
  if (c / yc >= 0.95) return -1.0;
  lambda_old = c;
  lambda = c - yc * (1.0 - 1.0 / (1.0 - exp(-yc / lambda_old)));
  for (int k = 0; k < 5; k++) { lambda_old = lambda; lambda = c - yc * (1.0 - 1.0 / (1.0 - exp(-yc / lambda_old))); }
  while (fabs(lambda - lambda_old) > TCM_LAMBDA_DELTA) { lambda_old = lambda; lambda = c - yc * (1.0 - 1.0 / (1.0 - exp(-yc / lambda_old))); }
  return lambda;
}

static int find_start_point(const Bucket *b, int peak, int n)                   /* :224-247 */
{
  int k;
  for (k = peak; k > 0; k--) {
    if (b[k].count == 0) continue;
    if (b[k].acum_samp_num < n * (1.0 - TCM_START_POINT_PROB)) break;
  }
  /* buckets above the peak are not initialised in the reference (:360-365); they count as empty here */
#define CNT(i) ((i) <= peak ? b[i].count : 0)
  if (CNT(0) > n / 100 && CNT(1) > n / 100 && CNT(2) > n / 100 && CNT(3) > n / 100) { if (k < 3) k = 3; }
  else if (CNT(0) > n / 100 && CNT(1) > n / 100 && CNT(2) > n / 100) { if (k < 2) k = 2; }
  else { if (k < 1) k = 1; }
#undef CNT
  if (k > peak) k = peak;
  return k;
}

static void compute_likelyhood(int point, int n, Bucket *b, int peak)           /* :249-281 */
{
  const double n1 = b[point].acum_samp_num, n2 = n - n1, yc = point;
  const double sum_yi = b[point].acum_abs_amp, total = b[point].acum_samp_num;
  const double lambda = lambda_given_yc(yc, sum_yi, total);
  const double prob = (double)n1 / (double)n;
  if (lambda > 0) {
    b[point].likelyhood = n2 * log(1 - prob) + n1 * log(prob) - n2 * log((peak - yc) * 2.0)
                          - n1 * log(1 - exp(-yc / lambda))
                          - n1 * log(2 * lambda) - sum_yi / lambda;
    b[point].lambda = lambda; b[point].prob = prob;
  } else {
    b[point].likelyhood = -TCM_MIN_LIKELYHOOD;
    b[point].lambda = lambda; b[point].prob = 1;
  }
}

/* TCMprocessOneSequence on the histogram of |C[k]| (hist[a] = number of samples with amplitude a, a <= peak);
 * returns Yc.  *err is set when the reference would run past its bucket array. */
double hmo_tcm_threshold(const int *hist, int peak, int len, int *err)
{
  static Bucket b[TCM_BUCKETS];
  if (peak == 0 || peak >= TCM_MAX_AMP) return 0.0;
  if (peak >= TCM_BUCKETS) { if (err) *err = 1; return 0.0; }
  for (int k = 0; k <= peak; k++) { b[k].count = hist[k]; b[k].acum_abs_amp = 0; b[k].acum_samp_num = 0; }
  b[0].acum_samp_num = b[0].count;
  for (int k = 1; k <= peak; k++) {
    b[k].acum_abs_amp = b[k - 1].acum_abs_amp + k * b[k].count;
    b[k].acum_samp_num = b[k - 1].acum_samp_num + b[k].count;
  }
  const int start = find_start_point(b, peak, len);
  compute_likelyhood(start, len, b, peak);
  double max_l = b[start].likelyhood; int max_pos = start;
  for (int k = start + 1; k <= peak; k++) {
    if (b[k].count == 0) continue;
    compute_likelyhood(k, len, b, peak);
    if (b[k].likelyhood > max_l) { max_pos = k; max_l = b[k].likelyhood; }
  }
  return max_l > TCM_MIN_LIKELYHOOD ? (double)max_pos : 0.0;
}

/* getOutlierWithDCT for the luma plane: obf[(h/4) x (w/4)] = number of AC frequencies of the block's 4x4 DCT that
 * survive the per-frequency threshold Yc[x] * DctScaling; yc16[x] receives the thresholds (yc16[0] = 0).
 * Returns 0, or 1 when an amplitude exceeds the reference's bucket array. */
int hmo_obf_prepass(const uint8_t *y, int w, int h, int stride, int16_t *obf, double *yc16)
{
  const int bw = w / 4, bh = h / 4, nblk = bw * bh;
  int32_t *org = (int32_t *)malloc(sizeof(int32_t) * 16 * (size_t)nblk);     /* CoeffFrequencyOrg[x][blk] */
  int *hist = (int *)calloc(TCM_MAX_AMP, sizeof(int));
  int err = 0;
  for (int by = 0; by < bh; by++)
    for (int bx = 0; bx < bw; bx++) {
      int16_t blk[16]; int32_t coef[16];
      for (int yy = 0; yy < 4; yy++) for (int xx = 0; xx < 4; xx++) blk[yy * 4 + xx] = y[(by * 4 + yy) * stride + bx * 4 + xx];
      hmo_fwd_transform(blk, 4, coef, 2, 0);                 /* partialButterfly x2, shifts 1 and 8 (:920-923,951-952) */
      for (int x = 0; x < 16; x++) org[(size_t)x * nblk + by * bw + bx] = coef[x];
    }
  yc16[0] = 0.0;
  for (int x = 1; x < 16; x++) {                             /* StartFreq = 1: DC is cleared (:985-989) */
    int peak = 0;
    memset(hist, 0, sizeof(int) * TCM_MAX_AMP);
    for (int i = 0; i < nblk; i++) {
      const int c = (int)(org[(size_t)x * nblk + i] / 8.0);  /* CoeffFrequency = coeff / DctScaling, truncated (:955) */
      const int a = abs(c);
      if (a > peak) peak = a;
      hist[a < TCM_MAX_AMP ? a : TCM_MAX_AMP - 1]++;
    }
    yc16[x] = hmo_tcm_threshold(hist, peak, nblk, &err);
  }
  for (int i = 0; i < nblk; i++) {
    int n = 0;
    for (int x = 1; x < 16; x++) {
      const int32_t c = org[(size_t)x * nblk + i];
      const int zeroed = (c < yc16[x] * 8.0 && c > -yc16[x] * 8.0);           /* :1009-1013 */
      if (!zeroed && c != 0) n++;                            /* OBF counts the surviving frequencies (:1023-1038) */
    }
    obf[i] = (int16_t)n;
  }
  free(org); free(hist);
  return err;
}
