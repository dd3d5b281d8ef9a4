/*
 * fcu_deblock.h -- in-loop deblocking of a decided intra or P picture on the device
 * (TComLoopFilter::loopFilterPic, Lib/TLibCommon/TComLoopFilter.cpp:130-155; called per picture at TEncGOP.cpp:1160).
 * Included by fcu_kernels.hip only.
 *
 * HBM-bound, two launches per picture because every vertical edge of the picture is filtered before the first
 * horizontal one (:133-154):
 *   dbk_pass<0>  one thread per (8-sample grid column x, 4-row segment): 4 rows x (4 + 4) luma samples as two aligned
 *                32-bit words per row; neighbouring lanes own neighbouring edges, so a wave reads 512 contiguous bytes
 *                of each row.  Every second edge column is also a chroma edge (8-sample chroma grid): 2 rows x (2 + 2)
 *                samples of Cb and Cr.
 *   dbk_pass<1>  one thread per (4-column segment, 8-sample grid row y): 8 rows x 4 luma samples, one 32-bit word per
 *                row, 256 contiguous bytes per row and wave; chroma 4 rows x 2 samples.
 * Edges 8 samples apart never touch each other's samples (3 modified, 4 read per side), so the threads of a pass are
 * independent and the filter runs in place like the reference's.  Edge flags come straight from the TComDataCU arrays
 * of fcu_ctu_out: the partition on the Q side starts a transform unit there (xSetEdgefilterTU / xSetEdgefilterPU,
 * :270-343) or a prediction unit (2NxN / Nx2N halves), is inside the picture and not on its border (xSetLoopfilterParam,
 * :346-405).  Boundary strength (xGetBoundaryStrengthSingle, :405-553): 2 next to an intra CU; else 1 across a transform
 * edge with a coded luma block on either side, or across different motion (P slices: one list; |dmv| >= 4 quarter samples
 * or different reference); else 0.  Chroma is filtered at Bs 2 only.
 * Algorithmic bytes per pass: read 1.5*W*H samples + 4 bytes of CU data per 4x4 partition, write <= 1.5*W*H.
 */
#pragma once

namespace fcu {

__device__ static const uint8_t k_dbk_tc[54] = {      /* sm_tcTable, TComLoopFilter.cpp:59-62 */
  0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,1,1,1,1,1,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,5,5,6,6,7,8,9,10,11,13,14,16,18,20,22,24 };
__device__ static const uint8_t k_dbk_beta[52] = {    /* sm_betaTable, :64-67 */
  0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,6,7,8,9,10,11,12,13,14,15,16,17,18,20,22,24,26,28,30,32,34,36,38,40,42,44,46,48,50,52,54,56,58,60,62,64 };

enum { DBK_THREADS = 256 };

__device__ static inline int dbk_clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ static inline int dbk_abs(int v) { return v < 0 ? -v : v; }

/* one line across a luma edge, xPelFilterLuma (:805-869): m[0..3] = P side (m[3] next to the edge), m[4..7] = Q side */
__device__ static inline void dbk_line_luma(int m[8], int tc, int sw, int thrCut, int filtP, int filtQ)
{
  const int m0 = m[0], m1 = m[1], m2 = m[2], m3 = m[3], m4 = m[4], m5 = m[5], m6 = m[6], m7 = m[7];
  if (sw) {
    m[3] = dbk_clip3(m3 - 2 * tc, m3 + 2 * tc, (m1 + 2 * m2 + 2 * m3 + 2 * m4 + m5 + 4) >> 3);
    m[4] = dbk_clip3(m4 - 2 * tc, m4 + 2 * tc, (m2 + 2 * m3 + 2 * m4 + 2 * m5 + m6 + 4) >> 3);
    m[2] = dbk_clip3(m2 - 2 * tc, m2 + 2 * tc, (m1 + m2 + m3 + m4 + 2) >> 2);
    m[5] = dbk_clip3(m5 - 2 * tc, m5 + 2 * tc, (m3 + m4 + m5 + m6 + 2) >> 2);
    m[1] = dbk_clip3(m1 - 2 * tc, m1 + 2 * tc, (2 * m0 + 3 * m1 + m2 + m3 + m4 + 4) >> 3);
    m[6] = dbk_clip3(m6 - 2 * tc, m6 + 2 * tc, (m3 + m4 + m5 + 3 * m6 + 2 * m7 + 4) >> 3);
  } else {
    int delta = (9 * (m4 - m3) - 3 * (m5 - m2) + 8) >> 4;
    if (dbk_abs(delta) < thrCut) {
      delta = dbk_clip3(-tc, tc, delta);
      m[3] = dbk_clip3(0, 255, m3 + delta);
      m[4] = dbk_clip3(0, 255, m4 - delta);
      const int tc2 = tc >> 1;
      if (filtP) m[2] = dbk_clip3(0, 255, m2 + dbk_clip3(-tc2, tc2, ((((m1 + m3 + 1) >> 1) - m2 + delta) >> 1)));
      if (filtQ) m[5] = dbk_clip3(0, 255, m5 + dbk_clip3(-tc2, tc2, ((((m6 + m4 + 1) >> 1) - m5 - delta) >> 1)));
    }
  }
}
/* the four lines of one luma segment: the loop body of xEdgeFilterLuma (:597-665).  Returns 0 when nothing is filtered. */
__device__ static inline int dbk_segment_luma(int m[4][8], int qp, int betaOff, int tcOff, int bs)
{
  const int tc = k_dbk_tc[dbk_clip3(0, 53, qp + 2 * (bs - 1) + tcOff * 2)];           /* + DEFAULT_INTRA_TC_OFFSET * (Bs - 1) */
  const int beta = k_dbk_beta[dbk_clip3(0, 51, qp + betaOff * 2)];
  const int side = (beta + (beta >> 1)) >> 3, thrCut = tc * 10;
  const int dp0 = dbk_abs(m[0][1] - 2 * m[0][2] + m[0][3]), dq0 = dbk_abs(m[0][4] - 2 * m[0][5] + m[0][6]);
  const int dp3 = dbk_abs(m[3][1] - 2 * m[3][2] + m[3][3]), dq3 = dbk_abs(m[3][4] - 2 * m[3][5] + m[3][6]);
  const int d0 = dp0 + dq0, d3 = dp3 + dq3, d = d0 + d3;
  if (d >= beta) return 0;
  const int filtP = (dp0 + dp3) < side, filtQ = (dq0 + dq3) < side;
  /* xUseStrongFiltering on lines 0 and 3 (:921-931) */
  const int s0 = (dbk_abs(m[0][0] - m[0][3]) + dbk_abs(m[0][7] - m[0][4]) < (beta >> 3)) && (2 * d0 < (beta >> 2)) && (dbk_abs(m[0][3] - m[0][4]) < ((tc * 5 + 1) >> 1));
  const int s3 = (dbk_abs(m[3][0] - m[3][3]) + dbk_abs(m[3][7] - m[3][4]) < (beta >> 3)) && (2 * d3 < (beta >> 2)) && (dbk_abs(m[3][3] - m[3][4]) < ((tc * 5 + 1) >> 1));
  const int sw = s0 && s3;
#pragma unroll
  for (int i = 0; i < 4; i++) dbk_line_luma(m[i], tc, sw, thrCut, filtP, filtQ);
  return 1;
}
/* xPelFilterChroma (:881-905): c = { m2, m3 | m4, m5 } */
__device__ static inline void dbk_line_chroma(int c[4], int tc)
{
  const int delta = dbk_clip3(-tc, tc, ((((c[2] - c[1]) * 4) + c[0] - c[3] + 4) >> 3));
  c[1] = dbk_clip3(0, 255, c[1] + delta);
  c[2] = dbk_clip3(0, 255, c[2] - delta);
}

/* z-order index of the 4x4 partition (x4, y4) inside its CTU: the bits of x and y interleaved, x in the even positions
 * (g_auiRasterToZscan, initZscanToRaster TComRom.cpp) -- computed instead of looked up, which saves a dependent gather */
__device__ static inline int dbk_zidx(int x4, int y4)
{
  int x = x4 & 15, y = y4 & 15;
  x = (x | (x << 2)) & 0x33; x = (x | (x << 1)) & 0x55;
  y = (y | (y << 2)) & 0x33; y = (y | (y << 1)) & 0x55;
  return x | (y << 1);
}
struct DbkPart { int flag, qp, bs; };
/* CU data of the partition (x4, y4) (4-sample units of the picture) for direction DIR: is its left / top border a
 * filtered edge, and its QP */
template <int DIR>
__device__ static inline DbkPart dbk_part(const fcu_ctu_out *out, int w_ctu, int x4, int y4)
{
  const fcu_ctu_out *c = &out[(y4 >> 4) * w_ctu + (x4 >> 4)];
  const int z = dbk_zidx(x4, y4);
  DbkPart r;
  r.qp = c->qp[z];
  const int pos = (DIR == 0 ? x4 : y4) * 4;
  const int cu = CTU >> c->depth[z], tu = cu >> c->tr_idx[z], ps = c->part_size[z];
  r.flag = 0; r.bs = 0;
  if (ps == SIZE_NONE || pos == 0) return r;
  if ((pos & (tu - 1)) == 0) r.flag = 1;                                   /* transform-unit / CU edge */
  else if ((pos & (cu - 1)) == (cu >> 1) && (ps == SIZE_NxN || (DIR == 0 ? ps == SIZE_Nx2N : ps == SIZE_2NxN))) r.flag = 2;   /* PU edge only */
  /* asymmetric partitions: the edge at a quarter of the CU (xSetEdgefilterPU, TComLoopFilter.cpp:331-350); only 32x32 and
   * 64x64 CUs put it on the 8-sample grid the filter visits */
  else if ((pos & (cu - 1)) == (cu >> 2) && (DIR == 0 ? ps == SIZE_nLx2N : ps == SIZE_2NxnU)) r.flag = 2;
  else if ((pos & (cu - 1)) == cu - (cu >> 2) && (DIR == 0 ? ps == SIZE_nRx2N : ps == SIZE_2NxnD)) r.flag = 2;
  if (!r.flag) return r;
  /* boundary strength against the partition on the P side */
  const int px4 = DIR == 0 ? x4 - 1 : x4, py4 = DIR == 0 ? y4 : y4 - 1;
  const fcu_ctu_out *p = &out[(py4 >> 4) * w_ctu + (px4 >> 4)];
  const int zp = dbk_zidx(px4, py4);
  if (p->pred_mode[zp] == MODE_INTRA || c->pred_mode[z] == MODE_INTRA) { r.bs = 2; return r; }
  if (r.flag == 1 && ((((c->cbf[0][z] >> c->tr_idx[z]) & 1) != 0) || (((p->cbf[0][zp] >> p->tr_idx[zp]) & 1) != 0))) { r.bs = 1; return r; }
  const int rp = p->ref_idx[zp], rq = c->ref_idx[z];
  const int mpx = rp < 0 ? 0 : p->mv[zp][0], mpy = rp < 0 ? 0 : p->mv[zp][1], mqx = rq < 0 ? 0 : c->mv[z][0], mqy = rq < 0 ? 0 : c->mv[z][1];
  r.bs = ((rp < 0) != (rq < 0) || (rp >= 0 && rp != rq) || dbk_abs(mqx - mpx) >= 4 || dbk_abs(mqy - mpy) >= 4) ? 1 : 0;
  return r;
}
__device__ static inline int dbk_qp_of(const fcu_ctu_out *out, int w_ctu, int x4, int y4)
{
  return out[(y4 >> 4) * w_ctu + (x4 >> 4)].qp[dbk_zidx(x4, y4)];
}

template <int DIR>
__global__ void __launch_bounds__(DBK_THREADS)
dbk_pass(const fcu_ctu_out *out, uint8_t *Y, uint8_t *U, uint8_t *V, int w, int h, int w_ctu, int betaOff, int tcOff)
{
  const int id = (int)(blockIdx.x * DBK_THREADS + threadIdx.x);
  const int cw = w >> 1;
  if (DIR == 0) {
    const int ne = w >> 3, x8 = id % ne, y4 = id / ne;
    if (y4 >= (h >> 2)) return;
    const int x4 = x8 * 2;
    const DbkPart q = dbk_part<0>(out, w_ctu, x4, y4);
    if (!q.bs) return;                                         /* no edge here (incl. the picture border x = 0, :358-365) or Bs 0 */
    const int qp = (dbk_qp_of(out, w_ctu, x4 - 1, y4) + q.qp + 1) >> 1;
    int m[4][8];
    uint8_t *p = Y + (size_t)(y4 * 4) * w + x4 * 4 - 4;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t a = *(const uint32_t *)(p + (size_t)i * w), b = *(const uint32_t *)(p + (size_t)i * w + 4);
#pragma unroll
      for (int k = 0; k < 4; k++) { m[i][k] = (a >> (8 * k)) & 255; m[i][4 + k] = (b >> (8 * k)) & 255; }
    }
    if (dbk_segment_luma(m, qp, betaOff, tcOff, q.bs)) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        *(uint32_t *)(p + (size_t)i * w) = (uint32_t)m[i][0] | ((uint32_t)m[i][1] << 8) | ((uint32_t)m[i][2] << 16) | ((uint32_t)m[i][3] << 24);
        *(uint32_t *)(p + (size_t)i * w + 4) = (uint32_t)m[i][4] | ((uint32_t)m[i][5] << 8) | ((uint32_t)m[i][6] << 16) | ((uint32_t)m[i][7] << 24);
      }
    }
    if ((x4 & 3) == 0 && q.bs == 2) {                          /* 8-sample chroma grid (:216-221,700-707), intra edges only (:723) */
      const int tc = k_dbk_tc[dbk_clip3(0, 53, (int)k_chroma_scale[qp] + 2 + tcOff * 2)];      /* cb / cr QP offsets 0 (:747-766) */
#pragma unroll
      for (int comp = 0; comp < 2; comp++) {
        uint8_t *cp = (comp ? V : U) + (size_t)(y4 * 2) * cw + x4 * 2 - 2;
#pragma unroll
        for (int i = 0; i < 2; i++) {
          const uint32_t a = *(const uint16_t *)(cp + (size_t)i * cw), b = *(const uint16_t *)(cp + (size_t)i * cw + 2);
          int c[4] = { (int)(a & 255), (int)(a >> 8), (int)(b & 255), (int)(b >> 8) };
          dbk_line_chroma(c, tc);
          cp[(size_t)i * cw + 1] = (uint8_t)c[1]; cp[(size_t)i * cw + 2] = (uint8_t)c[2];
        }
      }
    }
  } else {
    const int ns = w >> 2, x4 = id % ns, y8 = id / ns;
    if (y8 >= (h >> 3)) return;
    const int y4 = y8 * 2;
    if (y8 == 0) return;                                       /* picture border: never filtered (:383-390) */
    /* the samples are requested before the CU data is known, so that both round trips overlap */
    uint8_t *p = Y + (size_t)(y4 * 4 - 4) * w + x4 * 4;
    uint32_t ra[8];
#pragma unroll
    for (int k = 0; k < 8; k++) ra[k] = *(const uint32_t *)(p + (size_t)k * w);
    const DbkPart q = dbk_part<1>(out, w_ctu, x4, y4);
    if (!q.bs) return;
    const int qp = (dbk_qp_of(out, w_ctu, x4, y4 - 1) + q.qp + 1) >> 1;
    int m[4][8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
#pragma unroll
      for (int i = 0; i < 4; i++) m[i][k] = (ra[k] >> (8 * i)) & 255;
    }
    if (dbk_segment_luma(m, qp, betaOff, tcOff, q.bs)) {
#pragma unroll
      for (int k = 1; k < 7; k++)
        *(uint32_t *)(p + (size_t)k * w) = (uint32_t)m[0][k] | ((uint32_t)m[1][k] << 8) | ((uint32_t)m[2][k] << 16) | ((uint32_t)m[3][k] << 24);
    }
    if ((y4 & 3) == 0 && q.bs == 2) {
      const int tc = k_dbk_tc[dbk_clip3(0, 53, (int)k_chroma_scale[qp] + 2 + tcOff * 2)];
#pragma unroll
      for (int comp = 0; comp < 2; comp++) {
        uint8_t *cp = (comp ? V : U) + (size_t)(y4 * 2 - 2) * cw + x4 * 2;
        uint32_t r[4];
#pragma unroll
        for (int k = 0; k < 4; k++) r[k] = *(const uint16_t *)(cp + (size_t)k * cw);
#pragma unroll
        for (int i = 0; i < 2; i++) {
          int c[4] = { (int)((r[0] >> (8 * i)) & 255), (int)((r[1] >> (8 * i)) & 255), (int)((r[2] >> (8 * i)) & 255), (int)((r[3] >> (8 * i)) & 255) };
          dbk_line_chroma(c, tc);
          cp[(size_t)1 * cw + i] = (uint8_t)c[1]; cp[(size_t)2 * cw + i] = (uint8_t)c[2];
        }
      }
    }
  }
}

} // namespace fcu
