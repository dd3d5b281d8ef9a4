/*
 * fcu_sao.h -- sample adaptive offset of decided, deblocked pictures on the device
 * (TEncSampleAdaptiveOffset::SAOProcess, Lib/TLibEncoder/TEncSampleAdaptiveOffset.cpp:257-287, called per picture at
 * TEncGOP.cpp:1434; offset pass TComSampleAdaptiveOffset::offsetCTU, Lib/TLibCommon/TComSampleAdaptiveOffset.cpp:558-616).
 * Included by fcu_kernels.hip only (and by the test-only CPU build tests/emu/sao_emu.cpp).
 *
 * Four launches per batch of pictures; everything but the third is data parallel:
 *   sao_stats    one workgroup per (CTU, component, picture): getBlkStats (:922-1381) -- every sample is classified for the
 *                four edge directions and the band table, (org - src) and 1 accumulate in an LDS histogram (integer LDS
 *                atomics: order-free, so the sums are exact whatever the interleaving).  Reads org + src once (the 3x3
 *                neighbourhood comes from L2), writes 1 280 B per block.  Band statistics by LDS atomics, edge statistics in
 *                registers with one wave reduction per class.
 *   sao_cands    one thread per (CTU, component, type, picture): deriveOffsets / estIterOffset (:441-591) and the
 *                distortion of the derived offsets (:397-433) -- they depend on the statistics and lambda only.
 *   sao_decide   one wave per picture, CTUs in coding order: what is left of decideBlkParams (:790-920) is the rate of
 *                each candidate on the CABAC bit counter (two adaptive contexts carried from CTU to CTU), the new / merge
 *                choice, and the reconstruction of merged parameters (see sao_decide_picture below).
 *   sao_apply    one workgroup per (CTU, component, picture): offsetBlock (:317-556) from the untouched copy of the
 *                deblocked picture into the reconstruction.  HBM-bound: reads src, writes rec.
 * 8-bit 4:2:0, 64x64 CTUs, LFCrossSliceBoundaryFlag 1, no tiles: neighbour availability is the picture boundary
 * (TComPicSym.cpp:357-376); merge candidates stay inside the CTU's slice (TComPic.cpp:138-143).
 * Algorithmic bytes per picture: stats 2 x 1.5 W H read; apply 1.5 W H read + <= 1.5 W H written.
 */
#pragma once

namespace fcu {

enum { SAO_THREADS = 256, SAO_OFF = 0, SAO_NEW = 1, SAO_MERGE = 2, SAO_BO = 4, SAO_NTYPES = 5, SAO_MAXQ = 7,
       SAO_STAT_INTS = SAO_NTYPES * 2 * 32 };             /* per (CTU, component): [type][diff, count][class] */

struct SaoPic {                                            /* one picture of a batch (device copy) */
  const uint8_t *org[3]; uint8_t *rec[3]; uint8_t *src[3];
  double lambda[3]; int enabled[3]; int slice_type, qp, slice_ctus;
};
struct SaoCand { int16_t aux, ep; int32_t off[5]; long long dist; };   /* EO: off[class]; BO: off[i] of band aux + i; ep: bypass bins of the offsets as coded (sao_cand_ep) */

__device__ static inline int sao_sgn(int v) { return (v > 0) - (v < 0); }

/* ---- statistics: phase 0 clears the histogram, 1 accumulates, 2 writes it out -------------------------------------- */
template <int PHASE>
__device__ static inline void sao_stats_phase(int32_t *hist, const SaoPic *pics, int32_t *stats, int width, int height, int w_ctu, int n_ctu)
{
  const int t = (int)threadIdx.x, a = (int)blockIdx.x, comp = (int)blockIdx.y, pic = (int)blockIdx.z;
  if (PHASE == 0) { for (int i = t; i < SAO_STAT_INTS; i += SAO_THREADS) hist[i] = 0; return; }
  if (PHASE == 2) { int32_t *o = stats + ((size_t)(pic * n_ctu + a) * 3 + comp) * SAO_STAT_INTS; for (int i = t; i < SAO_STAT_INTS; i += SAO_THREADS) o[i] = hist[i]; return; }
  const SaoPic &P = pics[pic];
  const int sh = comp ? 1 : 0, cx = a % w_ctu, cy = a / w_ctu, x0 = cx * 64, y0 = cy * 64;
  const int bw = (x0 + 64 > width ? width - x0 : 64) >> sh, bh = (y0 + 64 > height ? height - y0 : 64) >> sh, stride = width >> sh;
  const int L = cx > 0, R = x0 + 64 < width, A = cy > 0, B = y0 + 64 < height, AL = A && L;
  const int skipR = comp ? 3 : 5, skipB = comp ? 2 : 4;
  const int sx = L ? 0 : 1, ex = R ? bw - skipR : bw - 1, exFull = R ? bw - skipR : bw;
  const int eyFull = B ? bh - skipB : bh, ey = B ? bh - skipB : bh - 1;
  const size_t o0 = (size_t)(y0 >> sh) * stride + (x0 >> sh);
  const uint8_t *src = P.src[comp] + o0, *org = P.org[comp] + o0;
  /* The five classes of the four edge directions are summed per thread in registers (compare-selects, no addressing) and reduced
   * once per wave: most samples of a picture fall into the flat class, and sixty-four lanes adding to one LDS word serialise.
   * The 32 bands spread over their addresses and keep the LDS atomic. */
  int32_t ed[4][5], ec[4][5];
#pragma unroll
  for (int ty = 0; ty < 4; ty++)
#pragma unroll
    for (int q = 0; q < 5; q++) { ed[ty][q] = 0; ec[ty][q] = 0; }
  for (int i = t; i < bw * bh; i += SAO_THREADS) {
    const int y = i / bw, x = i - y * bw;
    const uint8_t *s = src + (size_t)y * stride + x;
    const int c = s[0], d = (int)org[(size_t)y * stride + x] - c;
#define SAO_EDGE(type, k) do { const int k_ = (k); _Pragma("unroll") for (int q = 0; q < 5; q++) { const int m_ = k_ == q; ed[type][q] += m_ ? d : 0; ec[type][q] += m_; } } while (0)
    if (x < exFull && y < eyFull) { atomicAdd(&hist[(SAO_BO * 2) * 32 + (c >> 3)], d); atomicAdd(&hist[(SAO_BO * 2 + 1) * 32 + (c >> 3)], 1); }
    if (x >= sx && x < ex && y < eyFull) SAO_EDGE(0, 2 + sao_sgn(c - s[-1]) + sao_sgn(c - s[1]));
    if (x < exFull && y >= (A ? 0 : 1) && y < ey) SAO_EDGE(1, 2 + sao_sgn(c - s[-stride]) + sao_sgn(c - s[stride]));
    if (y == 0 ? (A && x >= (AL ? 0 : 1) && x < ex) : (y < ey && x >= sx && x < ex)) SAO_EDGE(2, 2 + sao_sgn(c - s[-stride - 1]) + sao_sgn(c - s[stride + 1]));
    if (y == 0 ? (A && x >= sx && x < ex) : (y < ey && x >= sx && x < ex)) SAO_EDGE(3, 2 + sao_sgn(c - s[-stride + 1]) + sao_sgn(c - s[stride - 1]));
#undef SAO_EDGE
  }
#pragma unroll
  for (int ty = 0; ty < 4; ty++)
#pragma unroll
    for (int q = 0; q < 5; q++) {
#ifdef FCU_EMU
      hist[(ty * 2) * 32 + q] += ed[ty][q]; hist[(ty * 2 + 1) * 32 + q] += ec[ty][q];
#else
      const uint32_t sd = fcu_wave_sum((uint32_t)ed[ty][q]), sc = fcu_wave_sum((uint32_t)ec[ty][q]);      /* (two's complement: the signed sum) */
      if ((t & 63) == 0) { atomicAdd(&hist[(ty * 2) * 32 + q], (int32_t)sd); atomicAdd(&hist[(ty * 2 + 1) * 32 + q], (int32_t)sc); }
#endif
    }
}

/* ---- offsets of one (CTU, component, type) ------------------------------------------------------------------------- */
__device__ static inline long long sao_est_dist(long long count, long long off, long long diff) { return count * off * off - diff * off * 2; }
/* estIterOffset, :441-472 */
__device__ static inline int sao_iter_offset(int type, double lambda, int offIn, long long count, long long diff, long long *bestDist, double *bestCost)
{
  int it = offIn, out = 0;
  double minCost = lambda;
  while (it != 0) {
    const int ab = it < 0 ? -it : it;
    long long rate = type == SAO_BO ? ab + 2 : ab + 1;
    if (ab == SAO_MAXQ) rate--;
    const long long dist = sao_est_dist(count, it, diff);
    const double cost = (double)dist + lambda * (double)rate;
    if (cost < minCost) { minCost = cost; out = it; *bestDist = dist; *bestCost = cost; }
    it = it > 0 ? it - 1 : it + 1;
  }
  return out;
}
__device__ static inline int sao_initial_offset(int diff, int count)
{
  if (count == 0) return 0;
  const double x = (double)diff / (double)count;
  const int v = x >= 0 ? (int)(x + 0.5) : (int)(x - 0.5);    /* xRoundIbdi at 8 bit, :54-57 */
  return v < -SAO_MAXQ ? -SAO_MAXQ : (v > SAO_MAXQ ? SAO_MAXQ : v);
}
/* bypass bins codeSAOOffsetParam (TEncSbac.cpp:1620-1676) spends on a new-mode candidate after sao_type_idx: four offsets
 * (codeSaoMaxUvlc :1545-1572: |v| = 0 -> 1 bin, else |v| + 1 bins, one less at the maximum), then signs of the non-zero band
 * offsets + 5 bits of band position, or 2 bits of edge class (not for Cr, which shares Cb's) */
__device__ static inline int sao_cand_ep(int type, int comp, const SaoCand &c)
{
  int n = 0, nz = 0;
  for (int i = 0; i < 4; i++) {
    const int v = type == SAO_BO ? c.off[i] : c.off[i < 2 ? i : i + 1], ab = v < 0 ? -v : v;
    n += ab == 0 ? 1 : 1 + (ab - 1) + (SAO_MAXQ > ab ? 1 : 0);
    nz += v != 0;
  }
  return n + (type == SAO_BO ? nz + 5 : (comp != 2 ? 2 : 0));
}
/* deriveOffsets (:474-591) + getDistortion (:397-433) of one type from its statistics */
__device__ static inline void sao_cand_one(int type, int comp, const int32_t *st /* [2][32] of the type */, double lambda, SaoCand *out)
{
  const int32_t *diff = st, *count = st + 32;
  long long dist = 0;
  if (type != SAO_BO) {
    for (int k = 0; k < 5; k++) {
      int q = k == 2 ? 0 : sao_initial_offset(diff[k], count[k]);
      if (k < 2 && q < 0) q = 0;
      if (k > 2 && q > 0) q = 0;
      long long dd; double cc;
      if (q != 0) q = sao_iter_offset(type, lambda, q, count[k], diff[k], &dd, &cc);
      out->off[k] = q;
      dist += sao_est_dist(count[k], q, diff[k]);
    }
    out->aux = 0;
  } else {
    /* the four-band window with the least cost: costs of the 32 bands first (a band's cost is lambda when its offset is 0) */
    double cost[32]; int8_t q[32];
    for (int k = 0; k < 32; k++) {
      cost[k] = lambda;
      int v = sao_initial_offset(diff[k], count[k]);
      long long dd = 0;
      if (v != 0) v = sao_iter_offset(type, lambda, v, count[k], diff[k], &dd, &cost[k]);
      q[k] = (int8_t)v;
    }
    double minCost = 1.7e+308; int band0 = 0;
    for (int band = 0; band < 32 - 4 + 1; band++) {
      double c = cost[band]; c += cost[band + 1]; c += cost[band + 2]; c += cost[band + 3];
      if (c < minCost) { minCost = c; band0 = band; }
    }
    out->aux = (int16_t)band0; out->off[4] = 0;
    for (int i = 0; i < 4; i++) { const int b = band0 + i; out->off[i] = q[b]; dist += sao_est_dist(count[b], q[b], diff[b]); }
  }
  out->dist = dist;
  out->ep = (int16_t)sao_cand_ep(type, comp, *out);
}
__device__ static inline void sao_cands_thread(const SaoPic *pics, const int32_t *stats, SaoCand *cands, int n_ctu, int n_pics)
{
  const long long id = (long long)blockIdx.x * SAO_THREADS + threadIdx.x;
  if (id >= (long long)n_pics * n_ctu * 15) return;
  const int type = (int)(id % 5), comp = (int)((id / 5) % 3);
  const long long blk = id / 15;                            /* pic * n_ctu + ctu */
  const int pic = (int)(blk / n_ctu);
  sao_cand_one(type, comp, stats + ((size_t)blk * 3 + comp) * SAO_STAT_INTS + type * 64, pics[pic].lambda[comp], &cands[id]);
}

/* ---- decision ------------------------------------------------------------------------------------------------------ */
/* The coder of the SAO syntax is two adaptive contexts (sao_merge_flag, sao_type_idx) and the Q15 bit counter of
 * TEncBinCABACCounter; bits written = counter >> 15, and resetBits() keeps the remainder (counter & 32767).
 * codeSAOOffsetParam (TEncSbac.cpp:1602-1677): sao_type_idx = one context-coded bin (+ one bypass bin when not off), then
 * bypass bins only -- the offsets (codeSaoMaxUvlc, :1545-1572), signs + band position or the edge class.  Those bypass bins
 * are SaoCand::ep.  codeSAOBlkParam (:1679-1714): merge-left / merge-up flags (context 0), then the three components. */
__device__ static inline uint8_t sao_ctx_init(int iv, int qp)
{
  qp = qp < 0 ? 0 : (qp > 51 ? 51 : qp);
  const int slope = (iv >> 4) * 5 - 45, offset = ((iv & 15) << 3) - 16;
  int st = ((slope * qp) >> 4) + offset; st = st < 1 ? 1 : (st > 126 ? 126 : st);
  const int mps = st >= 64;
  return (uint8_t)(((mps ? (st - 64) : (63 - st)) << 1) + mps);
}
/* ---- decideBlkParams (:790-920) of one picture on one wave ------------------------------------------------------------
 * What is serial in the reference's loop over the CTUs is small: the two adaptive contexts and the Q15 remainder of the bit
 * counter carried from CTU to CTU, and the parameters of the left / above CTU (merge candidates).  Everything else was
 * hoisted: a candidate's distortion and its bypass bins (SaoCand::ep) come from sao_cands, so that the rate of a candidate is
 * `ep` plus at most three context-coded bins.  Per CTU: all 64 lanes fetch the next CTU's statistics (3 840 B) and candidates
 * (480 B) into registers; the costs of deriveModeNewRDO's candidates and the distortions deriveModeMergeRDO needs are
 * computed one per lane on the copies in LDS (sao_ctu_costs; the neighbours' parameters come from a ring of the last RING
 * CTUs in LDS); lane 0 makes the choices in the reference's order and advances the coder (sao_ctu_choose); the fetched words
 * go to the other LDS buffer, and lanes 32..58 write the CTU's two 108-byte records (coded / reconstructed parameters) a
 * dword each.  No per-CTU round trip to HBM is on the
 * serial path.  The comparison order and every floating-point expression are the reference's. */
/* reconstructed parameters of one component of one CTU, packed: off4 = the four coded offsets as bytes (EO: classes 0, 1, 3, 4;
 * BO: bands band .. band + 3) */
struct SaoPar { int8_t on, type, band, pad; uint32_t off4; };
__device__ static inline int sao_par_off_i(const SaoPar &p, int i) { return (int)(int8_t)(p.off4 >> (8 * i)); }
enum { SAO_RING = 256, SAO_PRE = 17, SAO_CAND_WORDS = 15 * 8 };   /* the ring holds w_ctu + 1 CTUs: pictures up to 255 CTUs wide */
struct SaoDecideLds {
  int32_t stats[2][3 * SAO_STAT_INTS];
  SaoCand cand[2][15];
  SaoPar ring[SAO_RING][3];
  uint32_t bin[256];                                         /* k_bin: (bits << 8) | next state */
  int32_t merge[2];                                          /* the decision of the CTU whose records are being written: -1 new, 0 left, 1 above */
  int32_t n_off[3];
  uint32_t coder[3];                                         /* carried from CTU to CTU: sao_merge_flag context, sao_type_idx context, Q15 counter */
  double cost_l[5], cost_c[2][6];                            /* sao_ctu_costs -> sao_ctu_choose */
  double norm_l[5], norm_c[2][5], norm_m[2][3];
  double lambda[3]; int32_t en[3];                           /* of the picture: on chip, because they are indexed by component */              /* distortion / lambda of the luma types, the Cb / Cr types, the merge candidates */
};
static_assert(sizeof(SaoCand) == 32 && sizeof(SaoPar) == 8 && sizeof(fcu_sao_ctu) == 108, "layouts the word-wise copies rely on");
#ifdef FCU_EMU
#define SAO_PHASE for (int lane = 0; lane < 64; lane++)
#define SAO_SYNC() do { } while (0)
#define SAO_L lane
enum { SAO_NL = 64 };
#else
#define SAO_PHASE for (int lane = (int)threadIdx.x, once_ = 1; once_; once_ = 0)
/* one wave per workgroup: LDS accesses of a wave execute in program order, so the hand-over between its lanes needs no
 * barrier -- and must not get __syncthreads()' memory fence, which would wait for the statistics prefetch in flight */
#define SAO_SYNC() __builtin_amdgcn_wave_barrier()
#define SAO_L 0
enum { SAO_NL = 1 };
#endif
/* words of CTU a's statistics and candidates that `lane` fetches (SAO_PRE words per lane: 15 of statistics, 2 of candidates) */
__device__ static inline void sao_fetch(int lane, const int32_t *stats, const SaoCand *cands, int a, int32_t *pre)
{
  const int32_t *s = stats + (size_t)a * 3 * SAO_STAT_INTS, *c = (const int32_t *)(cands + (size_t)a * 15);
#pragma unroll
  for (int k = 0; k < 15; k++) pre[k] = s[k * 64 + lane];
#pragma unroll
  for (int k = 0; k < 2; k++) pre[15 + k] = (k * 64 + lane < SAO_CAND_WORDS) ? c[k * 64 + lane] : 0;
}
__device__ static inline void sao_stash(int lane, const int32_t *pre, int32_t *stats, SaoCand *cand)
{
#pragma unroll
  for (int k = 0; k < 15; k++) stats[k * 64 + lane] = pre[k];
#pragma unroll
  for (int k = 0; k < 2; k++) if (k * 64 + lane < SAO_CAND_WORDS) ((int32_t *)cand)[k * 64 + lane] = pre[15 + k];
}
__device__ static inline SaoPar sao_par_off() { SaoPar p; p.on = 0; p.type = 0; p.band = 0; p.pad = 0; p.off4 = 0; return p; }
__device__ static inline SaoPar sao_par_from_cand(int type, const SaoCand &cd)
{
  SaoPar p; p.on = 1; p.type = (int8_t)type; p.band = (int8_t)cd.aux; p.pad = 0; p.off4 = 0;
  for (int i = 0; i < 4; i++) p.off4 |= (uint32_t)((type == SAO_BO ? cd.off[i] : cd.off[i < 2 ? i : i + 1]) & 255) << (8 * i);
  return p;
}
/* getDistortion of already reconstructed offsets against this CTU's statistics (merge candidates, :760-766) */
/* A class of a 64x64 CTU holds <= 4096 samples, |diff| <= 255 * 4096 and |offset| <= 7: every term and the sum of five
 * fit 32 bits, so the 64-bit arithmetic of sao_est_dist is not needed here. */
__device__ static inline long long sao_merge_dist(const SaoPar &m, const int32_t *st /* [5][2][32] */)
{
  const int32_t *diff = st + m.type * 64, *count = diff + 32;
  int32_t d = 0;
  for (int i = 0; i < 4; i++) {                              /* the edge class without an offset (2) adds nothing */
    const int b = m.type == SAO_BO ? ((m.band + i) & 31) : (i < 2 ? i : i + 1), o = sao_par_off_i(m, i);
    d += count[b] * o * o - diff[b] * o * 2;
  }
  return d;
}
/* dword r (0..8) of the fcu_sao_offset of packed parameters: r = 0 the header (merge >= 0: signalled as a merge of that kind,
 * the coded record), r > 0 offset bytes 4 (r - 1) .. 4 (r - 1) + 3 -- EO: classes 0, 1, (2 = 0), 3 | 4; BO: bands from `band` */
__device__ static inline uint32_t sao_record_word(const SaoPar &p, int r, int merge)
{
  if (r == 0) {
    const int mode = merge >= 0 ? SAO_MERGE : (p.on ? SAO_NEW : SAO_OFF), type = merge >= 0 ? merge : p.type;
    return (uint32_t)(mode & 255) | ((uint32_t)(type & 255) << 8) | ((uint32_t)(p.band & 255) << 16);
  }
  if (!p.on) return 0;
  if (p.type != SAO_BO) return r == 1 ? ((p.off4 & 0xffffu) | ((p.off4 & 0xff0000u) << 8)) : (r == 2 ? (p.off4 >> 24) : 0u);
  const int sh = p.band * 8 - 32 * (r - 1);                   /* the four bytes start `sh` bits into this dword */
  return sh >= 32 || sh <= -32 ? 0u : (sh >= 0 ? p.off4 << sh : p.off4 >> -sh);
}
#define SAO_BIN(ctxv, fracv, binv) do { const uint32_t e_ = L.bin[(ctxv) * 2 + (binv)]; (fracv) += e_ >> 8; (ctxv) = e_ & 255u; } while (0)
struct SaoNb { int left, above; };                         /* merge candidates: inside the picture and the CTU's slice (TComPic.cpp:138-143) */
/* First half of a CTU, one quantity per lane, all from the carried coder L.coder = (merge context, type context, counter):
 *   lanes 0..4    cost of luma type `lane` (deriveModeNewRDO :620-652);
 *   lanes 8..13 / 16..21   cost of chroma type 0..4 and of "chroma off" (index 5) if luma ends up off / new (:654-730): the
 *                 coder chroma starts from depends on luma only through the sao_type_idx bin luma coded -- the counter is
 *                 reset to its Q15 remainder, which whole bypass bins do not change -- so there are two cases, not six;
 *   lanes 32..37  distortion of the left / above CTU's parameters on this CTU's statistics, per component (:760-766). */
__device__ static inline void sao_ctu_costs(int lane, SaoDecideLds &L, const int32_t *st, const SaoCand *cd, int a, int w_ctu, const SaoNb nb)
{
  const int en0 = L.en[0], en1 = L.en[1], en2 = L.en[2];
  if (!en0 && !en1 && !en2) return;
  if (lane < 22) {
    uint32_t m0 = L.coder[0], mf = L.coder[2];
    if (nb.left) SAO_BIN(m0, mf, 0);
    if (nb.above) SAO_BIN(m0, mf, 0);
    uint32_t x1 = L.coder[1], xf = mf & 32767u;
    if (lane < 5) {
      if (!en0) return;
      SAO_BIN(x1, xf, 1); xf += 32768u * (uint32_t)(1 + cd[lane].ep);
      L.cost_l[lane] = (double)cd[lane].dist + L.lambda[0] * (double)(int)(xf >> 15);
      L.norm_l[lane] = (double)cd[lane].dist / L.lambda[0];            /* its term of the new-mode cost in bits (:700-706), should it win */
    } else if (lane >= 8 && (lane & 7) < 6) {
      const int sc = (lane >> 3) - 1, t = lane & 7;
      if (sc == 0) { if (en0) SAO_BIN(x1, xf, 0); } else SAO_BIN(x1, xf, 1);
      const uint32_t m1 = x1, base = xf & 32767u;
      uint32_t prev = 0, now;
      double cost = 0;
      xf = base;
      if (t == 5) {
        if (en1) SAO_BIN(x1, xf, 0);
        now = xf >> 15; cost += L.lambda[1] * (double)(now - prev); prev = now;
        now = xf >> 15; cost += L.lambda[2] * (double)(now - prev); prev = now;     /* Cr shares the type with Cb: nothing is coded for "off" */
      } else {
        for (int c = 1; c < 3; c++) {
          if (!(c == 1 ? en1 : en2)) continue;
          const SaoCand &d = cd[c * 5 + t];
          if (c == 1) { x1 = m1; SAO_BIN(x1, xf, 1); xf += 32768u; }
          xf += 32768u * (uint32_t)d.ep;
          now = xf >> 15;
          cost += (double)d.dist + L.lambda[c] * (double)(now - prev); prev = now;
          if (sc == 0) L.norm_c[c - 1][t] = (double)d.dist / L.lambda[c];
        }
      }
      L.cost_c[sc][t] = cost;
    }
  } else if (lane >= 32 && lane < 38) {
    const int mt = (lane - 32) / 3, c = (lane - 32) % 3;
    long long d = 0;
    if (mt == 0 ? nb.left : nb.above) {
      const SaoPar *m = &L.ring[(mt == 0 ? a - 1 : a - w_ctu) % SAO_RING][c];
      if (m->on) d = sao_merge_dist(*m, st + c * SAO_STAT_INTS);
    }
    L.norm_m[mt][c] = (double)d / L.lambda[c];
  }
}
/* Second half, lane 0: the choices in the reference's order, the coder after the CTU, the CTU's reconstructed parameters */
__device__ static inline void sao_ctu_choose(SaoDecideLds &L, const SaoCand *cd, int a, int w_ctu, const SaoNb nb)
{
  const int en0 = L.en[0], en1 = L.en[1], en2 = L.en[2];
  SaoPar *out = L.ring[a % SAO_RING];
  if (!en0 && !en1 && !en2) { for (int c = 0; c < 3; c++) { out[c] = sao_par_off(); L.n_off[c]++; } L.merge[a & 1] = -1; return; }
  const uint32_t c0 = L.coder[0], c1 = L.coder[1], fr = L.coder[2];
  /* ---- deriveModeNewRDO, :593-734 */
  uint32_t g0 = c0, g1 = c1, gf = fr & 32767u, mf = fr;
  if (nb.left) { SAO_BIN(g0, gf, 0); }
  if (nb.above) { SAO_BIN(g0, gf, 0); }
  mf = fr + (gf - (fr & 32767u));                           /* mid: the same two flags on the counter that was not reset */
  int tL = -1, tC = -1;
  {
    uint32_t x1 = c1, xf = mf & 32767u;
    if (en0) SAO_BIN(x1, xf, 0);
    double minCost = L.lambda[0] * (double)(xf >> 15);
    if (en0) for (int type = 0; type < SAO_NTYPES; type++) { const double cost = L.cost_l[type]; if (cost < minCost) { minCost = cost; tL = type; } }
  }
  {
    const double *cc = L.cost_c[tL >= 0 ? 1 : 0];
    double minCost = cc[5];
    for (int type = 0; type < SAO_NTYPES; type++) if (cc[type] < minCost) { minCost = cc[type]; tC = type; }
  }
  double normNew = 0;                                        /* sum of modeDist[c] / lambda[c] (a component that stays off adds 0 / lambda) */
  normNew += tL >= 0 ? L.norm_l[tL] : 0.0; normNew += (tC >= 0 && en1) ? L.norm_c[0][tC] : 0.0; normNew += (tC >= 0 && en2) ? L.norm_c[1][tC] : 0.0;
  if (en0) { if (tL < 0) SAO_BIN(g1, gf, 0); else { SAO_BIN(g1, gf, 1); gf += 32768u * (uint32_t)(1 + cd[tL].ep); } }
  if (en1) { if (tC < 0) SAO_BIN(g1, gf, 0); else { SAO_BIN(g1, gf, 1); gf += 32768u * (uint32_t)(1 + cd[5 + tC].ep); } }
  if (en2 && tC >= 0) gf += 32768u * (uint32_t)cd[10 + tC].ep;
  normNew += (double)(gf >> 15);
  double minCost = 1.7e+308;
  int merge = -1; uint32_t n0 = g0, n1 = g1, nf = gf;
  if (normNew < minCost) minCost = normNew;
  /* ---- deriveModeMergeRDO, :736-788 */
  {
    double best = 1.7e+308; int bm = -1; uint32_t b0 = 0, bf = 0;
    for (int mt = 0; mt < 2; mt++) {
      if (!(mt == 0 ? nb.left : nb.above)) continue;
      const SaoPar *m = L.ring[(mt == 0 ? a - 1 : a - w_ctu) % SAO_RING];
      double normDist = 0;
      for (int c = 0; c < 3; c++) if (m[c].on) normDist += L.norm_m[mt][c];
      uint32_t x0 = c0, xf = fr & 32767u;
      int isLeft = 0;
      if (nb.left) { isLeft = mt == 0; SAO_BIN(x0, xf, isLeft); }
      if (nb.above && !isLeft) SAO_BIN(x0, xf, mt == 1);
      const double cost = normDist + (double)(int)(xf >> 15);
      if (cost < best) { best = cost; bm = mt; b0 = x0; bf = xf; }
    }
    if (best < minCost) { minCost = best; merge = bm; n0 = b0; n1 = c1; nf = bf; }
  }
  L.coder[0] = n0; L.coder[1] = n1; L.coder[2] = nf;
  /* ---- the CTU's parameters after reconstructBlkSAOParam (TComSampleAdaptiveOffset.cpp:252-288) */
  if (merge >= 0) { const SaoPar *m = L.ring[(merge == 0 ? a - 1 : a - w_ctu) % SAO_RING]; for (int c = 0; c < 3; c++) out[c] = m[c]; }
  else {
    out[0] = tL >= 0 ? sao_par_from_cand(tL, cd[tL]) : sao_par_off();
    out[1] = (en1 && tC >= 0) ? sao_par_from_cand(tC, cd[5 + tC]) : sao_par_off();
    out[2] = (en2 && tC >= 0) ? sao_par_from_cand(tC, cd[10 + tC]) : sao_par_off();
  }
  for (int c = 0; c < 3; c++) L.n_off[c] += !out[c].on;
  L.merge[a & 1] = merge;
}
/* coded[] = parameters as signalled, recon[] = after reconstructBlkSAOParam, off_count[comp] = CTUs whose reconstructed mode is
 * OFF (-> m_saoDisabledRate) */
__device__ static inline void sao_decide_picture(const SaoPic &P_, const int32_t *stats, const SaoCand *cands, fcu_sao_ctu *coded, fcu_sao_ctu *recon,
                                                 int32_t *off_count, int w_ctu, int n_ctu, SaoDecideLds &L)
{
  const int slice_ctus = P_.slice_ctus;                      /* the descriptor lives in HBM: read once, not per CTU behind the record stores */
  int32_t pre[SAO_NL][SAO_PRE];
  int cx = 0, cy = 0, sliceStart = 0;
  SAO_PHASE {
    for (int k = lane; k < 256; k += 64) L.bin[k] = k_bin[k];
    if (lane < 3) { L.n_off[lane] = 0; L.lambda[lane] = P_.lambda[lane]; L.en[lane] = P_.enabled[lane]; }
    sao_fetch(lane, stats, cands, 0, pre[SAO_L]);
    sao_stash(lane, pre[SAO_L], L.stats[0], L.cand[0]);
    if (lane == 0) {
      L.coder[0] = sao_ctx_init(153, P_.qp);                              /* INIT_SAO_MERGE_FLAG, ContextTables.h:444-450 */
      L.coder[1] = sao_ctx_init(P_.slice_type == SLICE_I ? 200 : 185, P_.qp);   /* INIT_SAO_TYPE_IDX [I] / [P], :452-458 */
      L.coder[2] = 0;
    }
  }
  SAO_SYNC();
  for (int a = 0; a < n_ctu; a++) {
    SAO_PHASE { if (a + 1 < n_ctu) sao_fetch(lane, stats, cands, a + 1, pre[SAO_L]); }
    if (slice_ctus > 0 && a == sliceStart + slice_ctus) sliceStart = a;
    SaoNb nb; nb.above = cy > 0 && a - w_ctu >= sliceStart; nb.left = cx > 0 && a - 1 >= sliceStart;
    if (++cx == w_ctu) { cx = 0; cy++; }
    SAO_PHASE { sao_ctu_costs(lane, L, L.stats[a & 1], L.cand[a & 1], a, w_ctu, nb); }
    SAO_SYNC();
    SAO_PHASE { if (lane == 0) sao_ctu_choose(L, L.cand[a & 1], a, w_ctu, nb); }
    SAO_PHASE { if (a + 1 < n_ctu) sao_stash(lane, pre[SAO_L], L.stats[(a + 1) & 1], L.cand[(a + 1) & 1]); }
    SAO_SYNC();
    SAO_PHASE {                                               /* the CTU's two records, a dword per lane */
      const int j = lane - 32;
      if (j >= 0 && j < 27) {
        const SaoPar &p = L.ring[a % SAO_RING][j / 9];
        const uint32_t w = sao_record_word(p, j % 9, -1);        /* the two records differ in the headers only */
        ((uint32_t *)(recon + a))[j] = w;
        ((uint32_t *)(coded + a))[j] = (j % 9 == 0 && L.merge[a & 1] >= 0) ? sao_record_word(p, 0, L.merge[a & 1]) : w;
      }
    }
  }
  SAO_SYNC();
  SAO_PHASE { if (lane < 3) off_count[lane] = L.n_off[lane]; }
}

/* ---- offsetBlock, TComSampleAdaptiveOffset.cpp:317-556 -------------------------------------------------------------- */
__device__ static inline void sao_apply_block(const SaoPic *pics, const fcu_sao_ctu *recon, int width, int height, int w_ctu, int n_ctu)
{
  const int t = (int)threadIdx.x, a = (int)blockIdx.x, comp = (int)blockIdx.y, pic = (int)blockIdx.z;
  const fcu_sao_offset &p = recon[(size_t)pic * n_ctu + a].c[comp];
  if (p.mode == SAO_OFF) return;
  const SaoPic &P = pics[pic];
  const int sh = comp ? 1 : 0, cx = a % w_ctu, cy = a / w_ctu, x0 = cx * 64, y0 = cy * 64;
  const int w = (x0 + 64 > width ? width - x0 : 64) >> sh, h = (y0 + 64 > height ? height - y0 : 64) >> sh, stride = width >> sh;
  const int L = cx > 0, R = x0 + 64 < width, A = cy > 0, B = y0 + 64 < height, AL = A && L, AR = A && R, BL = B && L, BR = B && R;
  const int sx = L ? 0 : 1, ex = R ? w : w - 1, type = p.type;
  const size_t o0 = (size_t)(y0 >> sh) * stride + (x0 >> sh);
  const uint8_t *src = P.src[comp] + o0; uint8_t *res = P.rec[comp] + o0;
  for (int i = t; i < w * h; i += SAO_THREADS) {
    const int y = i / w, x = i - y * w;
    const uint8_t *s = src + (size_t)y * stride + x;
    const int c = s[0];
    int k = -1;
    if (type == 0) { if (x >= sx && x < ex) k = 2 + sao_sgn(c - s[-1]) + sao_sgn(c - s[1]); }
    else if (type == 1) { if (y >= (A ? 0 : 1) && y < (B ? h : h - 1)) k = 2 + sao_sgn(c - s[-stride]) + sao_sgn(c - s[stride]); }
    else if (type == 2) {
      int ok;
      if (y == 0) ok = A && x >= (AL ? 0 : 1) && x < ex;
      else if (y == h - 1) ok = x >= (B ? sx : w - 1) && x < (BR ? w : w - 1);
      else ok = x >= sx && x < ex;
      if (ok) k = 2 + sao_sgn(c - s[-stride - 1]) + sao_sgn(c - s[stride + 1]);
    } else if (type == 3) {
      int ok;
      if (y == 0) ok = x >= (A ? sx : w - 1) && x < (AR ? w : w - 1);
      else if (y == h - 1) ok = B && x >= (BL ? 0 : 1) && x < ex;
      else ok = x >= sx && x < ex;
      if (ok) k = 2 + sao_sgn(c - s[-stride + 1]) + sao_sgn(c - s[stride - 1]);
    } else k = c >> 3;
    if (k >= 0) { const int v = c + p.offset[k]; res[(size_t)y * stride + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
  }
}

#ifndef FCU_EMU
__global__ void __launch_bounds__(SAO_THREADS) sao_stats(const SaoPic *pics, int32_t *stats, int width, int height, int w_ctu, int n_ctu)
{
  __shared__ int32_t hist[SAO_STAT_INTS];
  sao_stats_phase<0>(hist, pics, stats, width, height, w_ctu, n_ctu);
  __syncthreads();
  sao_stats_phase<1>(hist, pics, stats, width, height, w_ctu, n_ctu);
  __syncthreads();
  sao_stats_phase<2>(hist, pics, stats, width, height, w_ctu, n_ctu);
}
__global__ void __launch_bounds__(SAO_THREADS) sao_cands(const SaoPic *pics, const int32_t *stats, SaoCand *cands, int n_ctu, int n_pics)
{ sao_cands_thread(pics, stats, cands, n_ctu, n_pics); }
__global__ void __launch_bounds__(64) sao_decide(const SaoPic *pics, const int32_t *stats, const SaoCand *cands, fcu_sao_ctu *coded, fcu_sao_ctu *recon,
                                                 int32_t *off_count, int w_ctu, int n_ctu, int n_pics)
{
  __shared__ SaoDecideLds L;                                 /* one wave per picture */
  const int pic = (int)blockIdx.x;
  sao_decide_picture(pics[pic], stats + (size_t)pic * n_ctu * 3 * SAO_STAT_INTS, cands + (size_t)pic * n_ctu * 15,
                     coded + (size_t)pic * n_ctu, recon + (size_t)pic * n_ctu, off_count + pic * 3, w_ctu, n_ctu, L);
}
__global__ void __launch_bounds__(SAO_THREADS) sao_apply(const SaoPic *pics, const fcu_sao_ctu *recon, int width, int height, int w_ctu, int n_ctu)
{ sao_apply_block(pics, recon, width, height, w_ctu, n_ctu); }
#endif

} /* namespace fcu */
