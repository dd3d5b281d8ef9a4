/*
 * hmo_pred.c -- ORACLE (test infrastructure).  Intra reference samples, the 35 intra
 * predictors, SATD and SSE.
 *
 * Reference samples are held as ONE linear array ref[0..4N]:
 *   ref[0]        = p[-1][2N-1]  (bottom of the below-left column)
 *   ref[2N-1]     = p[-1][0]
 *   ref[2N]       = p[-1][-1]    (corner)
 *   ref[2N+1+x]   = p[x][-1]     x = 0..2N-1
 * which is the walk order of HM's substitution/filter code (TComPattern.cpp:314-521).
 */
#include "hmo_int.h"
#include <stdlib.h>
#include <string.h>

/* z-scan availability (H.265 6.4.1) == getPULeft/Above/AboveLeft/AboveRightAdi/BelowLeftAdi,
 * TComDataCU.cpp:1071-1390, with bEnforceSliceRestriction. (lx,ly) = luma position of the
 * neighbouring 4x4 unit, (cx,cy) = luma position of the current block. */
static int unit_available(const HmoEnc *e, int lx, int ly, int cx, int cy)
{
  if (lx < 0 || ly < 0 || lx >= e->p.width || ly >= e->p.height) return 0;
  int ctuN = (ly >> 6) * e->w_ctu + (lx >> 6);
  int ctuC = (cy >> 6) * e->w_ctu + (cx >> 6);
  if (ctuN < e->slice_start) return 0;
  if (ctuN < ctuC) return 1;
  if (ctuN > ctuC) return 0;
  int zn = hmo_r2z[((ly & 63) >> 2) * 16 + ((lx & 63) >> 2)];
  int zc = hmo_r2z[((cy & 63) >> 2) * 16 + ((cx & 63) >> 2)];
  return zn < zc;
}

/* initAdiPatternChType + fillReferenceSamples, TComPattern.cpp:104-521 (unfiltered part).
 * (px,py) = block position in the component plane, log2 = block size in that plane. */
void hmo_build_ref(HmoEnc *e, int comp, int px, int py, int log2, int unused, uint8_t *ref)
{
  (void)unused;
  const int N = 1 << log2, sh = comp ? 1 : 0, unit = 4 >> sh;
  const int lx0 = px << sh, ly0 = py << sh;       /* luma position of the block */
  const uint8_t *rec = e->rec[comp];
  const int stride = e->stride[comp];
  const int total = 4 * N + 1;
  uint8_t avail[4 * 64 + 1];
  int nAvail = 0;
  /* left + below-left, bottom to top */
  for (int i = 0; i < 2 * N; i++) {
    int y = 2 * N - 1 - i;                        /* row in component samples */
    int a = unit_available(e, lx0 - 4, ((py + y) / unit * unit) << sh, lx0, ly0);
    avail[i] = (uint8_t)a; nAvail += a;
    if (a) ref[i] = rec[(py + y) * stride + px - 1];
  }
  { int a = unit_available(e, lx0 - 4, ly0 - 4, lx0, ly0);
    avail[2 * N] = (uint8_t)a; nAvail += a;
    if (a) ref[2 * N] = rec[(py - 1) * stride + px - 1]; }
  for (int x = 0; x < 2 * N; x++) {
    int a = unit_available(e, ((px + x) / unit * unit) << sh, ly0 - 4, lx0, ly0);
    avail[2 * N + 1 + x] = (uint8_t)a; nAvail += a;
    if (a) ref[2 * N + 1 + x] = rec[(py - 1) * stride + px + x];
  }
  if (nAvail == 0) { memset(ref, 128, (size_t)total); return; }
  if (nAvail == total) return;
  if (!avail[0]) {
    int j = 1;
    while (j < total && !avail[j]) j++;
    for (int i = 0; i < j; i++) ref[i] = ref[j];
  }
  for (int i = 1; i < total; i++) if (!avail[i]) ref[i] = ref[i - 1];
}

/* reference smoothing incl. strong (bilinear) 32x32 filter, TComPattern.cpp:172-290 */
void hmo_filter_ref(const uint8_t *ref, uint8_t *out, int N, int strongAllowed)
{
  const int n4 = 4 * N;
  int strong = 0;
  if (strongAllowed && N >= 32) {
    int bl = ref[0], tl = ref[2 * N], tr = ref[n4];
    int bilLeft = abs(bl + tl - 2 * ref[N]) < 8;
    int bilAbove = abs(tl + tr - 2 * ref[3 * N]) < 8;
    strong = bilLeft && bilAbove;
  }
  out[0] = ref[0]; out[n4] = ref[n4];
  if (strong) {
    int bl = ref[0], tl = ref[2 * N], tr = ref[n4], shift = 0;
    while ((1 << shift) < 2 * N) shift++;
    for (int i = 1; i < 2 * N; i++) out[i] = (uint8_t)(((2 * N - i) * bl + i * tl + N) >> shift);
    out[2 * N] = (uint8_t)tl;
    for (int i = 1; i < 2 * N; i++) out[2 * N + i] = (uint8_t)(((2 * N - i) * tl + i * tr + N) >> shift);
  } else {
    for (int i = 1; i < n4; i++) out[i] = (uint8_t)((ref[i - 1] + 2 * ref[i] + ref[i + 1] + 2) >> 2);
  }
}

/* TComPrediction::filteringIntraReferenceSamples, TComPattern.cpp:523-548 */
int hmo_use_filtered_ref(int mode, int log2, int isLuma)
{
  if (!isLuma) return 0;                          /* 4:2:0 chroma never smoothed */
  if (mode == HMO_DC) return 0;
  int d1 = abs(mode - HMO_HOR), d2 = abs(mode - HMO_VER);
  int diff = d1 < d2 ? d1 : d2;
  return diff > hmo_intra_filter_thr[log2 - 2];
}

static inline int clip8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

/* predIntraAng / xPredIntraAng / xPredIntraPlanar / xDCPredFiltering,
 * TComPrediction.cpp:183-496,755-841.  `ref` must already be the array selected by
 * hmo_use_filtered_ref(); `refFilt` is unused and kept for symmetry. */
void hmo_intra_pred(const uint8_t *ref, const uint8_t *refFilt, int log2, int mode, int isLuma,
                    uint8_t *dst, int ds)
{
  (void)refFilt;
  const int N = 1 << log2;
  const uint8_t *corner = ref + 2 * N;
#define LEFT(y) (corner[-1 - (y)])               /* y = -1 .. 2N-1 */
#define TOP(x)  (corner[1 + (x)])                /* x = -1 .. 2N-1 */
  if (mode == HMO_PLANAR) {
    int bl = LEFT(N), tr = TOP(N);
    for (int y = 0; y < N; y++)
      for (int x = 0; x < N; x++)
        dst[y * ds + x] = (uint8_t)(((N - 1 - x) * LEFT(y) + (x + 1) * tr + (N - 1 - y) * TOP(x) + (y + 1) * bl + N) >> (log2 + 1));
    return;
  }
  if (mode == HMO_DC) {
    int sum = 0;
    for (int i = 0; i < N; i++) sum += TOP(i) + LEFT(i);
    int dc = (sum + N) >> (log2 + 1);
    for (int y = 0; y < N; y++) memset(dst + y * ds, dc, (size_t)N);
    if (isLuma && N <= 16) {
      dst[0] = (uint8_t)((TOP(0) + LEFT(0) + 2 * dc + 2) >> 2);
      for (int x = 1; x < N; x++) dst[x] = (uint8_t)((TOP(x) + 3 * dc + 2) >> 2);
      for (int y = 1; y < N; y++) dst[y * ds] = (uint8_t)((LEFT(y) + 3 * dc + 2) >> 2);
    }
    return;
  }
  const int isVer = mode >= 18;
  const int angMode = isVer ? mode - HMO_VER : -(mode - HMO_HOR);
  const int absAng = abs(angMode), sign = angMode < 0 ? -1 : 1;
  const int angle = sign * hmo_ang_table[absAng];
  const int invAngle = hmo_inv_ang_table[absAng];
  int16_t buf[3 * 64 + 2];
  int16_t *refMain = buf + 64;                    /* index -N .. 2N */
  /* main / side arrays: main[0] = corner */
  if (angle < 0) {
    for (int i = 0; i <= N; i++) refMain[i] = isVer ? TOP(i - 1) : LEFT(i - 1);
    int invSum = 128;
    for (int k = -1; k > ((N * angle) >> 5); k--) {
      invSum += invAngle;
      int s = invSum >> 8;                        /* index into side array, side[0] = corner */
      refMain[k] = isVer ? LEFT(s - 1) : TOP(s - 1);
    }
  } else {
    for (int i = 0; i <= 2 * N; i++) refMain[i] = isVer ? TOP(i - 1) : LEFT(i - 1);
  }
  for (int y = 0; y < N; y++) {
    int deltaPos = (y + 1) * angle, di = deltaPos >> 5, df = deltaPos & 31;
    for (int x = 0; x < N; x++) {
      int v;
      if (angle == 0) v = refMain[x + 1];
      else if (df) v = ((32 - df) * refMain[x + di + 1] + df * refMain[x + di + 2] + 16) >> 5;
      else v = refMain[x + di + 1];
      if (isVer) dst[y * ds + x] = (uint8_t)v; else dst[x * ds + y] = (uint8_t)v;
    }
  }
  if (angle == 0 && isLuma && N <= 16) {          /* edge filter of pure hor/ver */
    for (int y = 0; y < N; y++) {
      if (isVer) dst[y * ds] = (uint8_t)clip8(dst[y * ds] + ((LEFT(y) - corner[0]) >> 1));
      else       dst[y]      = (uint8_t)clip8(dst[y] + ((TOP(y) - corner[0]) >> 1));
    }
  }
#undef LEFT
#undef TOP
}

/* xCalcHADs8x8 / xCalcHADs4x4 / xGetHADs, TComRdCost.cpp:1343-1604 */
static uint32_t had(const uint8_t *o, int so, const uint8_t *p, int sp, int n)
{
  int m[64];
  for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) m[y * n + x] = o[y * so + x] - p[y * sp + x];
  for (int y = 0; y < n; y++)                     /* rows */
    for (int len = 1; len < n; len <<= 1)
      for (int i = 0; i < n; i += 2 * len)
        for (int j = i; j < i + len; j++) { int a = m[y * n + j], b = m[y * n + j + len]; m[y * n + j] = a + b; m[y * n + j + len] = a - b; }
  for (int x = 0; x < n; x++)                     /* columns */
    for (int len = 1; len < n; len <<= 1)
      for (int i = 0; i < n; i += 2 * len)
        for (int j = i; j < i + len; j++) { int a = m[j * n + x], b = m[(j + len) * n + x]; m[j * n + x] = a + b; m[(j + len) * n + x] = a - b; }
  uint32_t s = 0;
  for (int i = 0; i < n * n; i++) s += (uint32_t)abs(m[i]);
  return n == 8 ? ((s + 2) >> 2) : ((s + 1) >> 1);
}
uint32_t hmo_satd(const uint8_t *org, int so, const uint8_t *pred, int sp, int w, int h)
{
  uint32_t sum = 0;
  int n = ((w % 8) == 0 && (h % 8) == 0) ? 8 : 4;
  for (int y = 0; y < h; y += n) for (int x = 0; x < w; x += n) sum += had(org + y * so + x, so, pred + y * sp + x, sp, n);
  return sum;
}
/* xGetSSE*, TComRdCost.cpp:970-1315 */
uint32_t hmo_sse(const uint8_t *org, int so, const uint8_t *rec, int sr, int w, int h)
{
  uint32_t s = 0;
  for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) { int d = org[y * so + x] - rec[y * sr + x]; s += (uint32_t)(d * d); }
  return s;
}
