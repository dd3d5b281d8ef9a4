/*
 * fcu_inter.h -- the P-slice half of the CTU engine (BASELINE configs[4]); included by fcu_engine.h inside namespace fcu.
 *
 * Configuration built: one to four reference pictures in list 0 (padded planes of pictures after the loop
 * filters, with their POCs: reference-index loop and syntax, POC-scaled predictors), TMVP and AMP optional (frame parameters), TZ search (FastSearch 1) or full integer search with FEN
 * sub-sampling, half / quarter refinement on Hadamard cost, FDM, MaxNumMergeCand 5, QuadtreeTUMaxDepthInter 3.  The inter
 * candidates of a CU that predict the same block share one residual coding (Scratch::memo_*).  Replaces, per routine
 * (paths in the reference):
 *   TEncCu::xCheckRDCostMerge2Nx2N / xCheckRDCostInter                     TEncCu.cpp:1900-2062
 *   TEncSearch::predInterSearch, xEstimateMvPredAMVP, xMotionEstimation,
 *     xPatternSearch, xPatternSearchFracDIF, xCheckBestMVP, xMergeEstimation TEncSearch.cpp:2905-4375
 *   TEncSearch::encodeResAndCalcRdInterCU, xEstimateInterResidualQT,
 *     xEncodeInterResidualQT, xSetInterResidualQTData, xAddSymbolBitsInter TEncSearch.cpp:4380-5421
 *   TComDataCU::getInterMergeCandidates, fillMvpCand, clipMv               TComDataCU.cpp:2340-2942
 *   TComPrediction::motionCompensation + TComInterpolationFilter            TComPrediction.cpp:518-705
 *   TComRdCost::xGetSAD* / getCost / xGetComponentBits                     TComRdCost.cpp:278-292,465-964, TComRdCost.h:163-189
 *
 * How the wave is used: the integer search puts one CANDIDATE POSITION per lane (packed 4-sample SADs, v_sad_u8) and
 * picks the winner with a cross-lane min; interpolation, Hadamard, SSE and the transforms spread samples over lanes and
 * reduce with cross-lane sums (FCU_WAVE_ADD); the Y / Cb / Cr (x transform-skip) trials of a transform unit all restart
 * from the same CABAC snapshot, so their RDOQ and bit counts run side by side, one variant per lane.
 */

struct Pu { int addr, ox, oy, w, h; };
FCU_DEV int pu_count(int ps) { return ps == SIZE_2Nx2N ? 1 : (ps == SIZE_NxN ? 4 : 2); }
/* getPartIndexAndSize / getPartPosition (TComDataCU.cpp:2165-2240,2706-2772) as a table: per (part size, PU) the offset and size in
 * quarters of the CU size and the partition address in sixteenths of the CU's partitions (g_auiPUOffset), packed
 * ox | oy << 4 | w << 8 | h << 12 | addr << 16.  One scalar load instead of a branch ladder per shape. */
FCU_TABLE uint32_t k_pu_geom[8][4] = { { 0x04400, 0x04400, 0x04400, 0x04400 }, { 0x02400, 0x82420, 0x00000, 0x00000 }, { 0x04200, 0x44202, 0x00000, 0x00000 }, { 0x02200, 0x42202, 0x82220, 0xc2222 }, { 0x01400, 0x23410, 0x00000, 0x00000 }, { 0x03400, 0xa1430, 0x00000, 0x00000 }, { 0x04100, 0x14301, 0x00000, 0x00000 }, { 0x04300, 0x54103, 0x00000, 0x00000 } };
FCU_DEV Pu pu_geom(int depth, int ps, int pu)
{
  const int s = CTU >> depth, n = NPART >> (2 * depth);
  const uint32_t e = k_pu_geom[ps][pu];
  Pu g; g.ox = (int)((e & 7) * s) >> 2; g.oy = (int)(((e >> 4) & 7) * s) >> 2; g.w = (int)(((e >> 8) & 7) * s) >> 2; g.h = (int)(((e >> 12) & 7) * s) >> 2;
  g.addr = (int)(((e >> 16) & 15) * n) >> 4;
  return g;
}
/* does partition i (relative to the CU, n partitions) belong to prediction unit pu?  (TComCUMvField::setAll / setSubPart) */
FCU_DEV int pu_covers(int n, int ps, int pu, int i)
{
  if (ps == SIZE_2Nx2N) return 1;
  if (ps == SIZE_2NxN) return (i >= (n >> 1)) == (pu != 0);
  if (ps == SIZE_Nx2N) return ((i / (n >> 2)) & 1) == pu;
  if (ps == SIZE_NxN) return i / (n >> 2) == pu;
  /* asymmetric: position of partition i inside the CU (a CU is aligned to its size, so the z-order index is local) against
   * the quarter line; n partitions = (s / 4)^2 */
  const int lx = part_x(i), ly = part_y(i), s = n == 256 ? 64 : (n == 64 ? 32 : (n == 16 ? 16 : 8)), q = s >> 2;
  if (ps == SIZE_2NxnU) return (ly >= q) == (pu != 0);
  if (ps == SIZE_2NxnD) return (ly >= s - q) == (pu != 0);
  if (ps == SIZE_nLx2N) return (lx >= q) == (pu != 0);
  return (lx >= s - q) == (pu != 0);
}
/* motion fields of one PU: all lanes */
FCU_DEV void pu_set_motion(CuObj *cu, int ps, int pu, int lane, int mvx, int mvy, int ref)
{ const int n = cu->nparts; for (int i = lane; i < n; i += 64) if (pu_covers(n, ps, pu, i)) { cu->mv[i][0] = (int16_t)mvx; cu->mv[i][1] = (int16_t)mvy; cu->ref_idx[i] = (int8_t)ref; } }
FCU_DEV void pu_set_info(CuObj *cu, int ps, int pu, int lane, int mergeFlag, int mergeIdx, int mvdx, int mvdy, int mvpIdx)
{
  const int n = cu->nparts;
  for (int i = lane; i < n; i += 64) if (pu_covers(n, ps, pu, i)) {
    cu->merge_flag[i] = (uint8_t)mergeFlag; cu->merge_idx[i] = (uint8_t)mergeIdx; cu->inter_dir[i] = 1;
    cu->mvd[i][0] = (int16_t)mvdx; cu->mvd[i][1] = (int16_t)mvdy; cu->mvp_idx[i] = (int8_t)mvpIdx;
  }
}

/* neighbour motion (getPULeft / Above / AboveRight / BelowLeft / AboveLeft): inside the picture, in this slice, earlier in
 * z-scan than the corner partition (cx, cy) the lookup starts from; data from the working CU when inside it */
struct Nb { int avail, inter, skip, mvx, mvy, ref; };
FCU_DEV Nb nb_motion(const Env E, const CuObj *cu, int nx, int ny, int cx, int cy)
{
  Nb r; r.avail = 0; r.inter = 0; r.skip = 0; r.mvx = 0; r.mvy = 0; r.ref = -1;
  const Params &P = E.C->p;
  if (nx < 0 || ny < 0 || nx >= P.width || ny >= P.height) return r;
  const int ctuN = (ny >> 6) * E.C->w_ctu + (nx >> 6), ctuC = (cy >> 6) * E.C->w_ctu + (cx >> 6);
  if (ctuN < E.slice_start || ctuN > ctuC) return r;
  if (ctuN == ctuC && !(zidx_of(nx, ny) < zidx_of(cx, cy))) return r;
  r.avail = 1;
  if (inside_cu(cu, nx, ny)) {
    const int p = zidx_of(nx, ny) - cu->zidx;
    r.inter = cu->pred_mode[p] == MODE_INTER; r.skip = cu->skip[p]; r.mvx = cu->mv[p][0]; r.mvy = cu->mv[p][1]; r.ref = cu->ref_idx[p];
  } else {
    const fcu_ctu_out *c = &E.C->out[ctuN]; const int p = zidx_of(nx, ny);
    r.inter = c->pred_mode[p] == MODE_INTER; r.skip = c->skip[p]; r.mvx = c->mv[p][0]; r.mvy = c->mv[p][1]; r.ref = c->ref_idx[p];
  }
  return r;
}
FCU_DEV int same_motion(const Nb &a, const Nb &b) { return a.mvx == b.mvx && a.mvy == b.mvy && a.ref == b.ref; }

/* getInterMergeCandidates (P slice, no TMVP) -> g_S.mrg_mv / mrg_ref; one lane */
/* ---- TMVP: the reference picture is the collocated picture (collocated_from_l0, collocated_ref_idx 0).  xGetColMVP
 * (TComDataCU.cpp:3175-3242): the motion its fcu_ctu_out array holds at the top-left 4x4 partition of the 16x16 block that
 * contains the position (what TComPic::compressMotion keeps); unavailable where that partition is intra.  The vector is
 * scaled when the two POC distances differ (col_mvp below). */
/* xGetDistScaleFactor (TComDataCU.cpp:3312-3329) and TComMv::scaleMv (TComMv.h:145-150): vectors of neighbours / of the
 * collocated picture that point at another reference picture are scaled by the ratio of the POC distances */
FCU_DEV int dist_scale(int curPoc, int curRefPoc, int colPoc, int colRefPoc)
{
  const int dD = colPoc - colRefPoc, dB = curPoc - curRefPoc;
  if (dD == dB) return 4096;
  const int tdb = clip3i(-128, 127, dB), tdd = clip3i(-128, 127, dD);
  const int x = (0x4000 + iabs(tdd / 2)) / tdd;
  return clip3i(-4096, 4095, (tdb * x + 32) >> 6);
}
FCU_DEV void scale_mv(int sc, int &x, int &y)
{
  if (sc == 4096) return;
  x = clip3i(-32768, 32767, (sc * x + 127 + (sc * x < 0)) >> 8); y = clip3i(-32768, 32767, (sc * y + 127 + (sc * y < 0)) >> 8);
}
FCU_DEV int col_mvp(const Env E, int x, int y, int refIdx, int &mvx, int &mvy)
{
  const fcu_ctu_out *col = E.C->col;
  if (!col) return 0;
  const int xc = x & ~15, yc = y & ~15;
  const fcu_ctu_out *c = &col[(yc >> 6) * E.C->w_ctu + (xc >> 6)];
  const int z = zidx_of(xc & 63, yc & 63);
  if (c->pred_mode[z] != MODE_INTER || c->ref_idx[z] < 0) return 0;
  mvx = c->mv[z][0]; mvy = c->mv[z][1];
  const int cr = c->ref_idx[z] < FCU_MAX_REF ? c->ref_idx[z] : 0;
  scale_mv(dist_scale(E.C->poc, E.C->ref_poc[refIdx], E.C->col_poc, E.C->col_ref_poc[cr]), mvx, mvy);
  return 1;
}
/* bottom-right neighbour H when inside the picture and the CTU row (:2528-2563 / :2863-2900), else the PU centre */
FCU_DEV int temporal_candidate(const Env E, int xP, int yP, int w, int h, int refIdx, int &mvx, int &mvy)
{
  if (!E.C->p.tmvp) return 0;
  const int bx = xP + w, by = yP + h;
  if (bx < E.C->p.width && by < E.C->p.height && (by & 63) != 0 && col_mvp(E, bx, by, refIdx, mvx, mvy)) return 1;
  return col_mvp(E, xP + ((w >> 3) << 2), yP + ((h >> 3) << 2), refIdx, mvx, mvy);
}

FCU_DEV FCU_NOINLINE void merge_candidates(const CuObj *cu, int ps, int pu)
{
  const Env E = env_get(); cu = FCU_UNI(cu); ps = FCU_UNI(ps); pu = FCU_UNI(pu);
  const Pu g = pu_geom(cu->depth_cu, ps, pu);
  const int xP = cu->x + g.ox, yP = cu->y + g.oy, w = g.w, h = g.h, maxc = E.C->p.max_merge_cand;
  const int lbx = xP, lby = yP + h - 1, rtx = xP + w - 1, rty = yP;
  int n = 0;
#define FCU_ADD_MRG(nb) do { g_S.mrg_mv[n][0] = (nb).mvx; g_S.mrg_mv[n][1] = (nb).mvy; g_S.mrg_ref[n] = (nb).ref; n++; } while (0)
  const Nb a1 = nb_motion(E, cu, xP - 1, yP + h - 1, lbx, lby);
  const int okA1 = a1.avail && !(pu == 1 && (ps == SIZE_Nx2N || ps == SIZE_nLx2N || ps == SIZE_nRx2N)) && a1.inter;
  if (okA1) FCU_ADD_MRG(a1);
  const Nb b1 = nb_motion(E, cu, xP + w - 1, yP - 1, rtx, rty);
  const int okB1 = b1.avail && !(pu == 1 && (ps == SIZE_2NxN || ps == SIZE_2NxnU || ps == SIZE_2NxnD)) && b1.inter;
  if (n < maxc && okB1 && (!okA1 || !same_motion(a1, b1))) FCU_ADD_MRG(b1);
  const Nb b0 = nb_motion(E, cu, xP + w, yP - 1, rtx, rty);
  const int okB0 = b0.avail && b0.inter;
  if (n < maxc && okB0 && (!okB1 || !same_motion(b1, b0))) FCU_ADD_MRG(b0);
  const Nb a0 = nb_motion(E, cu, xP - 1, yP + h, lbx, lby);
  const int okA0 = a0.avail && a0.inter;
  if (n < maxc && okA0 && (!okA1 || !same_motion(a1, a0))) FCU_ADD_MRG(a0);
  if (n < maxc && n < 4) {
    const Nb b2 = nb_motion(E, cu, xP - 1, yP - 1, xP, yP);
    if (b2.avail && b2.inter && (!okA1 || !same_motion(a1, b2)) && (!okB1 || !same_motion(b1, b2))) FCU_ADD_MRG(b2);
  }
  if (n < maxc) { int tx, ty; if (temporal_candidate(E, xP, yP, w, h, 0, tx, ty)) { g_S.mrg_mv[n][0] = tx; g_S.mrg_mv[n][1] = ty; g_S.mrg_ref[n] = 0; n++; } }   /* references index 0 (:2566) */
  for (int r = 0, refcnt = 0; n < maxc; n++) {               /* zero candidates walk the reference indices (:2672-2686) */
    g_S.mrg_mv[n][0] = g_S.mrg_mv[n][1] = 0; g_S.mrg_ref[n] = r;
    if (refcnt == E.C->n_ref - 1) r = 0; else { ++r; ++refcnt; }
  }
#undef FCU_ADD_MRG
}
/* fillMvpCand (TComDataCU.cpp:2781-2925) for RefPicList0[refIdx] -> g_S.amvp; one lane.  The left pair (A0, A1) and the above
 * triple (B0, B1, B2), each first for a neighbour that references the same picture (xAddMVPCand), then -- left: if that found
 * nothing; above: only if no left neighbour was inter -- any inter neighbour with its vector scaled by the POC distances
 * (xAddMVPCandOrder); equal pair pruned, temporal candidate appended, cut / padded to two. */
FCU_DEV int mvp_same(const Env E, const Nb &nb, int refIdx, int (*c)[2], int &n)
{ if (nb.avail && nb.ref >= 0 && E.C->ref_poc[nb.ref] == E.C->ref_poc[refIdx]) { c[n][0] = nb.mvx; c[n][1] = nb.mvy; n++; return 1; } return 0; }
FCU_DEV int mvp_scaled(const Env E, const Nb &nb, int refIdx, int (*c)[2], int &n)
{
  if (!(nb.avail && nb.ref >= 0)) return 0;
  int x = nb.mvx, y = nb.mvy; scale_mv(dist_scale(E.C->poc, E.C->ref_poc[refIdx], E.C->poc, E.C->ref_poc[nb.ref]), x, y);
  c[n][0] = x; c[n][1] = y; n++; return 1;
}
FCU_DEV FCU_NOINLINE void amvp_candidates(const CuObj *cu, int ps, int pu, int refIdx)
{
  const Env E = env_get(); cu = FCU_UNI(cu); ps = FCU_UNI(ps); pu = FCU_UNI(pu); refIdx = FCU_UNI(refIdx);
  const Pu g = pu_geom(cu->depth_cu, ps, pu);
  const int xP = cu->x + g.ox, yP = cu->y + g.oy, w = g.w, h = g.h;
  const int lbx = xP, lby = yP + h - 1, rtx = xP + w - 1, rty = yP;
  int n = 0; int c3[4][2];
  const Nb a0 = nb_motion(E, cu, xP - 1, yP + h, lbx, lby), a1 = nb_motion(E, cu, xP - 1, yP + h - 1, lbx, lby);
  const int addedSmvp = (a0.avail && a0.inter) || (a1.avail && a1.inter);
  int added = mvp_same(E, a0, refIdx, c3, n);
  if (!added) added = mvp_same(E, a1, refIdx, c3, n);
  if (!added) { added = mvp_scaled(E, a0, refIdx, c3, n); if (!added) mvp_scaled(E, a1, refIdx, c3, n); }
  const Nb b0 = nb_motion(E, cu, xP + w, yP - 1, rtx, rty), b1 = nb_motion(E, cu, xP + w - 1, yP - 1, rtx, rty), b2 = nb_motion(E, cu, xP - 1, yP - 1, xP, yP);
  added = mvp_same(E, b0, refIdx, c3, n);
  if (!added) added = mvp_same(E, b1, refIdx, c3, n);
  if (!added) mvp_same(E, b2, refIdx, c3, n);
  if (!addedSmvp) {
    added = mvp_scaled(E, b0, refIdx, c3, n);
    if (!added) added = mvp_scaled(E, b1, refIdx, c3, n);
    if (!added) mvp_scaled(E, b2, refIdx, c3, n);
  }
  if (n == 2 && c3[0][0] == c3[1][0] && c3[0][1] == c3[1][1]) n = 1;
  if (n < 3) { int tx, ty; if (temporal_candidate(E, xP, yP, w, h, refIdx, tx, ty)) { c3[n][0] = tx; c3[n][1] = ty; n++; } }
  if (n > 2) n = 2;
  while (n < 2) { c3[n][0] = c3[n][1] = 0; n++; }
  g_S.amvp[0][0] = c3[0][0]; g_S.amvp[0][1] = c3[0][1]; g_S.amvp[1][0] = c3[1][0]; g_S.amvp[1][1] = c3[1][1];
}
FCU_DEV void clip_mv(const Params &P, const CuObj *cu, int &x, int &y)            /* clipMv, TComDataCU.cpp:2930-2942 */
{
  const int hmax = (P.width + 8 - cu->x - 1) << 2, hmin = (-64 - 8 - cu->x + 1) << 2;
  const int vmax = (P.height + 8 - cu->y - 1) << 2, vmin = (-64 - 8 - cu->y + 1) << 2;
  x = x > hmax ? hmax : (x < hmin ? hmin : x); y = y > vmax ? vmax : (y < vmin ? vmin : y);
}

/* ---- interpolation (TComInterpolationFilter as xPredInterBlk applies it): every path equals
 * clip8((sum_v sum_h c_v c_h s + 2048) >> 12) with the {64} filter for a zero fraction; the reference planes are padded */
FCU_TABLE int8_t k_luma_filter[4][8] = { { 0, 0, 0, 64, 0, 0, 0, 0 }, { -1, 4, -10, 58, 17, -5, 1, 0 }, { -1, 4, -11, 40, 40, -11, 4, -1 }, { 0, 1, -5, 17, 58, -10, 4, -1 } };
FCU_TABLE int8_t k_chroma_filter[8][4] = { { 0, 64, 0, 0 }, { -2, 58, 10, -2 }, { -4, 54, 16, -2 }, { -6, 46, 28, -4 }, { -4, 36, 36, -4 }, { -4, 28, 46, -6 }, { -2, 16, 54, -4 }, { -2, 10, 58, -2 } };
/* sample (x, y) of the component plane displaced by (mvx, mvy) (luma: quarter units, chroma: eighth units) */
FCU_DEV int interp_sample(const uint8_t *ref, int rs, int comp, int x, int y, int mvx, int mvy)
{
  const int sh = comp ? 3 : 2, msk = (1 << sh) - 1, fx = mvx & msk, fy = mvy & msk;
  const uint8_t *p = ref + (y + (mvy >> sh)) * rs + x + (mvx >> sh);
  if (!fx && !fy) return p[0];
  if (comp == 0) {
    if (!fy) { int s = 0; uint8_t b[8]; __builtin_memcpy(b, p - 3, 8);
#pragma unroll
      for (int t = 0; t < 8; t++) s += k_luma_filter[fx][t] * b[t];
      return clip8((s + 32) >> 6); }
    if (!fx) { int s = 0;
#pragma unroll
      for (int t = 0; t < 8; t++) s += k_luma_filter[fy][t] * p[(t - 3) * rs];
      return clip8((s + 32) >> 6); }
    int s = 0;
    for (int j = 0; j < 8; j++) { int h = 0; uint8_t b[8]; __builtin_memcpy(b, p + (j - 3) * rs - 3, 8);
#pragma unroll
      for (int t = 0; t < 8; t++) h += k_luma_filter[fx][t] * b[t];
      s += k_luma_filter[fy][j] * h; }
    return clip8((s + 2048) >> 12);
  }
  int s = 0;
  for (int j = 0; j < 4; j++) { int h = 0; const uint8_t *q = p + (j - 1) * rs;
#pragma unroll
    for (int t = 0; t < 4; t++) h += k_chroma_filter[fx][t] * q[t - 1];
    s += k_chroma_filter[fy][j] * h; }
  return clip8((s + 2048) >> 12);
}
/* block of the component plane at (bx, by), size w x h, displaced by mv -> dst (all lanes; caller supplies the phase) */
FCU_DEV void mc_block_lanes(const Env E, int lane, int refIdx, int comp, int bx, int by, int w, int h, int mvx, int mvy, uint8_t *dst, int ds)
{
  const uint8_t *ref = E.C->refs[refIdx][comp]; const int rs = E.C->ref_stride[comp];
  for (int i = lane; i < w * h; i += 64) { const int y = i / w, x = i - y * w; dst[y * ds + x] = (uint8_t)interp_sample(ref, rs, comp, bx + x, by + y, mvx, mvy); }
}
/* motionCompensation of one PU into a CU-sized buffer (the vector is clipped first, xPredInterUni) */
FCU_DEV FCU_NOINLINE void mc_pu(const CuObj *cu, int ps, int pu, Yuv *dst, int lumaOnly)
{
  const Env E = env_get(); cu = FCU_UNI(cu); ps = FCU_UNI(ps); pu = FCU_UNI(pu); dst = FCU_UNI(dst); lumaOnly = FCU_UNI(lumaOnly);
  const Pu g = pu_geom(cu->depth_cu, ps, pu);
  int mvx = FCU_UNI((int)cu->mv[g.addr][0]), mvy = FCU_UNI((int)cu->mv[g.addr][1]);
  int r = FCU_UNI((int)cu->ref_idx[g.addr]); if (r < 0) r = 0;
  clip_mv(E.C->p, cu, mvx, mvy);
  FCU_FOR_LANES {
    mc_block_lanes(E, lane, r, 0, cu->x + g.ox, cu->y + g.oy, g.w, g.h, mvx, mvy, dst->y + g.oy * 64 + g.ox, 64);
    if (!lumaOnly) {
      mc_block_lanes(E, lane, r, 1, (cu->x + g.ox) >> 1, (cu->y + g.oy) >> 1, g.w >> 1, g.h >> 1, mvx, mvy, dst->u + (g.oy >> 1) * 32 + (g.ox >> 1), 32);
      mc_block_lanes(E, lane, r, 2, (cu->x + g.ox) >> 1, (cu->y + g.oy) >> 1, g.w >> 1, g.h >> 1, mvx, mvy, dst->v + (g.oy >> 1) * 32 + (g.ox >> 1), 32);
    }
  }
}

/* ---- motion-vector cost: unsigned 32-bit arithmetic as in the reference */
FCU_DEV uint32_t mv_comp_bits(int v) { uint32_t len = 1, t = (v <= 0) ? (((uint32_t)(-v)) << 1) + 1 : ((uint32_t)v << 1); while (t != 1) { t >>= 1; len += 2; } return len; }
FCU_DEV uint32_t mv_bits(int x, int y, int px, int py, int scale) { return mv_comp_bits((x << scale) - px) + mv_comp_bits((y << scale) - py); }
FCU_DEV uint32_t motion_cost(const Params &P, uint32_t bits) { return (P.lambda_motion_sad * bits) >> 16; }

#ifdef FCU_EMU
static inline uint32_t fcu_sad_u8(uint32_t a, uint32_t b, uint32_t acc)
{ for (int k = 0; k < 4; k++) { const int d = (int)((a >> (8 * k)) & 255) - (int)((b >> (8 * k)) & 255); acc += (uint32_t)(d < 0 ? -d : d); } return acc; }
#else
__device__ inline uint32_t fcu_sad_u8(uint32_t a, uint32_t b, uint32_t acc) { return __builtin_amdgcn_sad_u8(a, b, acc); }
#endif
/* SAD of the w x h source block (stride 64, in the chain's scratch) against the reference at `r`, rows 0, step, 2*step.. */
FCU_DEV uint32_t sad_block(const uint8_t *org, const uint8_t *r, int rs, int w, int h, int step)
{
  uint32_t s = 0;
  for (int y = 0; y < h; y += step) {
    const uint8_t *o = org + y * 64, *q = r + y * rs;
    if (!(w & 7)) for (int x = 0; x < w; x += 8) { uint32_t a[2], b[2]; __builtin_memcpy(a, o + x, 8); __builtin_memcpy(b, q + x, 8); s = fcu_sad_u8(a[0], b[0], s); s = fcu_sad_u8(a[1], b[1], s); }
    else for (int x = 0; x < w; x += 4) { uint32_t a, b; __builtin_memcpy(&a, o + x, 4); __builtin_memcpy(&b, q + x, 4); s = fcu_sad_u8(a, b, s); }
  }
  return s;
}
/* Hadamard SATD of one USZ x USZ block from two sample buffers (xCalcHADs8x8 / 4x4), same scheme as satd_unit */
template <int USZ>
FCU_DEV uint32_t had_unit(const uint8_t *org, int so, const uint8_t *pred, int sp)
{
  int acc[USZ * USZ];
#pragma unroll
  for (int i = 0; i < USZ * USZ; i++) acc[i] = 0;
  for (int y = 0; y < USZ; y++) {
    int row[USZ]; uint8_t o[USZ], p[USZ];
    __builtin_memcpy(o, org + y * so, USZ); __builtin_memcpy(p, pred + y * sp, USZ);      /* one 8- / 4-byte load per row and operand */
#pragma unroll
    for (int x = 0; x < USZ; x++) row[x] = (int)o[x] - (int)p[x];
#pragma unroll
    for (int len = 1; len < USZ; len <<= 1)
#pragma unroll
      for (int i = 0; i < USZ; i += 2 * len)
#pragma unroll
        for (int j = i; j < i + len; j++) { const int a = row[j], b = row[j + len]; row[j] = a + b; row[j + len] = a - b; }
#pragma unroll
    for (int k = 0; k < USZ; k++) {
      const int neg = -(__builtin_popcount((unsigned)(k & y)) & 1);
#pragma unroll
      for (int j = 0; j < USZ; j++) acc[k * USZ + j] += (row[j] ^ neg) - neg;
    }
  }
  int s = 0;
#pragma unroll
  for (int i = 0; i < USZ * USZ; i++) s += iabs(acc[i]);
  return (uint32_t)(USZ == 8 ? ((s + 2) >> 2) : ((s + 1) >> 1));
}
/* xGetHADs of a w x h block: 8x8 units when both dimensions allow, else 4x4; lanes share the units, the sum lands in *accp */
FCU_DEV void had_block_lanes(int lane, const uint8_t *org, int so, const uint8_t *pred, int sp, int w, int h, uint32_t *accp)
{
  uint32_t part = 0;
  if (!(w & 7) && !(h & 7)) { const int bw = w >> 3, nb = bw * (h >> 3); for (int u = lane; u < nb; u += 64) part += had_unit<8>(org + (u / bw) * 8 * so + (u % bw) * 8, so, pred + (u / bw) * 8 * sp + (u % bw) * 8, sp); }
  else { const int bw = w >> 2, nb = bw * (h >> 2); for (int u = lane; u < nb; u += 64) part += had_unit<4>(org + (u / bw) * 4 * so + (u % bw) * 4, so, pred + (u / bw) * 4 * sp + (u % bw) * 4, sp); }
  FCU_WAVE_ADD(accp, part);
}

/* xGetInterPredictionError: MC of the PU (luma) + Hadamard against the source -> return value (uniform) */
FCU_DEV FCU_NOINLINE uint32_t inter_pred_error(const CuObj *cu, int ps, int pu)
{
  const Env E = env_get(); cu = FCU_UNI(cu); ps = FCU_UNI(ps); pu = FCU_UNI(pu);
  const Pu g = pu_geom(cu->depth_cu, ps, pu);
  mc_pu(cu, ps, pu, &E.G->tmp_pred, 1);
  FCU_SERIAL g_S.acc[0] = 0;
  const uint8_t *org = E.G->org[cu->depth_cu].y + g.oy * 64 + g.ox, *p = E.G->tmp_pred.y + g.oy * 64 + g.ox;
  FCU_FOR_LANES {
    if (E.C->p.had_me) had_block_lanes(lane, org, 64, p, 64, g.w, g.h, &g_S.acc[0]);
    else { uint32_t s = 0; for (int i = lane; i < g.w * g.h; i += 64) s += (uint32_t)iabs(org[(i / g.w) * 64 + i % g.w] - p[(i / g.w) * 64 + i % g.w]); FCU_WAVE_ADD(&g_S.acc[0], s); }
  }
  return FCU_UNI(g_S.acc[0]);
}

/* ---- xEstimateMvPredAMVP + xGetTemplateCost: best of the two candidates -> returns its index (uniform) */
FCU_DEV FCU_NOINLINE int estimate_mvp(const CuObj *cu, int ps, int pu, int refIdx)
{
  const Env E = env_get(); cu = FCU_UNI(cu); ps = FCU_UNI(ps); pu = FCU_UNI(pu); refIdx = FCU_UNI(refIdx);
  const Params &P = E.C->p;
  const Pu g = pu_geom(cu->depth_cu, ps, pu);
  FCU_SERIAL { amvp_candidates(cu, ps, pu, refIdx); g_S.acc[0] = g_S.acc[1] = 0; }
  const uint8_t *org = E.G->org[cu->depth_cu].y + g.oy * 64 + g.ox;
  for (int i = 0; i < 2; i++) {
    int cx = FCU_UNI(g_S.amvp[i][0]), cy = FCU_UNI(g_S.amvp[i][1]);
    clip_mv(P, cu, cx, cy);
    FCU_FOR_LANES {
      uint32_t s = 0;
      for (int k = lane; k < g.w * g.h; k += 64) { const int y = k / g.w, x = k - y * g.w; s += (uint32_t)iabs(org[y * 64 + x] - interp_sample(E.C->refs[refIdx][0], E.C->ref_stride[0], 0, cu->x + g.ox + x, cu->y + g.oy + y, cx, cy)); }
      FCU_WAVE_ADD(&g_S.acc[i], s);
    }
  }
  /* calcRdCost(bits = 1, SAD, false, DF_SAD): floor(SAD + floor(lambdaMotionSAD + 0.5) / 65536) */
  const double add = FCU_FLOOR((double)P.lambda_motion_sad + 0.5) / 65536.0;
  const uint32_t c0 = (uint32_t)FCU_FLOOR((double)FCU_UNI(g_S.acc[0]) + add), c1 = (uint32_t)FCU_FLOOR((double)FCU_UNI(g_S.acc[1]) + add);
  return c1 < c0 ? 1 : 0;
}

/* ---- xMotionEstimation: full search (xPatternSearch) + xPatternSearchFracDIF.  Results in g_S.me_out: [0] mvx, [1] mvy,
 * g_S.acc[12] bits, g_S.acc[13] cost. */
FCU_TABLE int8_t k_refine_h[9][2] = { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, 0 }, { 1, 0 }, { -1, -1 }, { 1, -1 }, { -1, 1 }, { 1, 1 } };
FCU_TABLE int8_t k_refine_q[9][2] = { { 0, 0 }, { 0, -1 }, { 0, 1 }, { -1, -1 }, { 1, -1 }, { -1, 0 }, { 1, 0 }, { -1, 1 }, { 1, 1 } };
/* ---- xTZSearch (FastSearch 1), TEncSearch.cpp:3981-4180 with TZ_SEARCH_CONFIGURATION (:301-317).  The search is a sequence
 * of rounds; a round tests a short list of positions in a fixed order and keeps the first strict improvement chain
 * (xTZSearchHelp).  Lane 0 writes the round's list (the reference's order and border checks), the lanes take one position
 * each (SAD over the PU + vector cost), and the wave's minimum of (cost << 32 | list index) is the position the reference's
 * sequential scan would have ended on. */
struct TzCtx { const uint8_t *org, *ref0; int rs, px, py, w, h, step, predx, predy; };
FCU_DEV void tz_list_reset() { g_S.tz_n = 0; }
FCU_DEV void tz_list_add(int x, int y, int point, int dist) { const int n = g_S.tz_n; if (n < 16) { g_S.tz_x[n] = (int16_t)x; g_S.tz_y[n] = (int16_t)y; g_S.tz_pt[n] = (uint8_t)point; g_S.tz_d[n] = (uint8_t)dist; g_S.tz_n = n + 1; } }
/* runs the listed positions; list written by lane 0 beforehand */
FCU_DEV void tz_run(const Params &P, const TzCtx &t)
{
  FCU_SERIAL g_S.me_best = ~0ull;
  FCU_FOR_LANES {
    unsigned long long key = ~0ull;
    if (lane < g_S.tz_n) {
      const int x = g_S.tz_x[lane], y = g_S.tz_y[lane];
      uint32_t s = sad_block(t.org, t.ref0 + (t.py + y) * t.rs + t.px + x, t.rs, t.w, t.h, t.step);
      if (t.step == 2) s <<= 1;
      s += motion_cost(P, mv_bits(x, y, t.predx, t.predy, 2));
      key = ((unsigned long long)s << 32) | (unsigned)lane;
    }
    FCU_WAVE_MIN64(&g_S.me_best, key);
  }
  FCU_SERIAL {
    const uint32_t s = (uint32_t)(g_S.me_best >> 32); const int i = (int)(g_S.me_best & 0xffffffffull);
    if (g_S.tz_n > 0 && s < g_S.tz_best) { g_S.tz_best = s; g_S.tz_bx = g_S.tz_x[i]; g_S.tz_by = g_S.tz_y[i]; g_S.tz_dist = g_S.tz_d[i]; g_S.tz_round = 0; g_S.tz_point = g_S.tz_pt[i]; }
  }
}
/* xTZ8PointDiamondSearch, :625-805: the list of one diamond (lane 0) */
FCU_DEV void tz_diamond_list(int ltx, int lty, int rbx, int rby, int sx, int sy, int d)
{
  const int top = sy - d, bot = sy + d, left = sx - d, right = sx + d;
  tz_list_reset();
  g_S.tz_round += 1;
  if (d == 1) {
    if (top >= lty) tz_list_add(sx, top, 2, d);
    if (left >= ltx) tz_list_add(left, sy, 4, d);
    if (right <= rbx) tz_list_add(right, sy, 5, d);
    if (bot <= rby) tz_list_add(sx, bot, 7, d);
  } else if (d <= 8) {
    const int h = d >> 1, top2 = sy - h, bot2 = sy + h, left2 = sx - h, right2 = sx + h;
    const int all = top >= lty && left >= ltx && right <= rbx && bot <= rby;
    if (all || top >= lty) tz_list_add(sx, top, 2, d);
    if (all || top2 >= lty) { if (all || left2 >= ltx) tz_list_add(left2, top2, 1, h); if (all || right2 <= rbx) tz_list_add(right2, top2, 3, h); }
    if (all || left >= ltx) tz_list_add(left, sy, 4, d);
    if (all || right <= rbx) tz_list_add(right, sy, 5, d);
    if (all || bot2 <= rby) { if (all || left2 >= ltx) tz_list_add(left2, bot2, 6, h); if (all || right2 <= rbx) tz_list_add(right2, bot2, 8, h); }
    if (all || bot <= rby) tz_list_add(sx, bot, 7, d);
  } else {
    const int q = d >> 2;
    const int all = top >= lty && left >= ltx && right <= rbx && bot <= rby;
    if (all || top >= lty) tz_list_add(sx, top, 0, d);
    if (all || left >= ltx) tz_list_add(left, sy, 0, d);
    if (all || right <= rbx) tz_list_add(right, sy, 0, d);
    if (all || bot <= rby) tz_list_add(sx, bot, 0, d);
    for (int i = 1; i < 4; i++) {
      const int yt = top + q * i, yb = bot - q * i, xl = sx - q * i, xr = sx + q * i;
      if (all || yt >= lty) { if (all || xl >= ltx) tz_list_add(xl, yt, 0, d); if (all || xr <= rbx) tz_list_add(xr, yt, 0, d); }
      if (all || yb <= rby) { if (all || xl >= ltx) tz_list_add(xl, yb, 0, d); if (all || xr <= rbx) tz_list_add(xr, yb, 0, d); }
    }
  }
}
/* xTZ2PointSearch, :442-573 */
FCU_DEV void tz_two_point_list(int ltx, int lty, int rbx, int rby)
{
  const int x = g_S.tz_bx, y = g_S.tz_by;
  const int L = x - 1 >= ltx, R = x + 1 <= rbx, T = y - 1 >= lty, B = y + 1 <= rby;
  tz_list_reset();
  switch (g_S.tz_point) {
  case 1: if (L) tz_list_add(x - 1, y, 0, 2); if (T) tz_list_add(x, y - 1, 0, 2); break;
  case 2: if (T) { if (L) tz_list_add(x - 1, y - 1, 0, 2); if (R) tz_list_add(x + 1, y - 1, 0, 2); } break;
  case 3: if (T) tz_list_add(x, y - 1, 0, 2); if (R) tz_list_add(x + 1, y, 0, 2); break;
  case 4: if (L) { if (B) tz_list_add(x - 1, y + 1, 0, 2); if (T) tz_list_add(x - 1, y - 1, 0, 2); } break;
  case 5: if (R) { if (T) tz_list_add(x + 1, y - 1, 0, 2); if (B) tz_list_add(x + 1, y + 1, 0, 2); } break;
  case 6: if (L) tz_list_add(x - 1, y, 0, 2); if (B) tz_list_add(x, y + 1, 0, 2); break;
  case 7: if (B) { if (L) tz_list_add(x - 1, y + 1, 0, 2); if (R) tz_list_add(x + 1, y + 1, 0, 2); } break;
  case 8: if (R) tz_list_add(x + 1, y, 0, 2); if (B) tz_list_add(x, y + 1, 0, 2); break;
  default: break;
  }
}
/* the whole search; result in g_S.tz_bx / tz_by.  Diamonds use the window of the predictor (lt, rb); only the raster
 * search uses the window re-centred on the best start point when a 2Nx2N integer vector was offered (:4023-4037). */
FCU_DEV void tz_search(const Params &P, const CuObj *cu, const TzCtx &t, int ltx, int lty, int rbx, int rby, int useIntMv, int imx, int imy)
{
  const int range = P.search_range, raster = 5;
  int rltx = ltx, rlty = lty, rrbx = rbx, rrby = rby;
  int stx = t.predx, sty = t.predy; clip_mv(P, cu, stx, sty); stx >>= 2; sty >>= 2;
  int mx = imx << 2, my = imy << 2; clip_mv(P, cu, mx, my); mx >>= 2; my >>= 2;
  FCU_SERIAL {
    g_S.tz_best = 0xffffffffu; g_S.tz_bx = g_S.tz_by = 0; g_S.tz_dist = 0; g_S.tz_round = 0; g_S.tz_point = 0;
    tz_list_reset(); tz_list_add(stx, sty, 0, 0); tz_list_add(0, 0, 0, 0);               /* predictor, zero vector */
    if (useIntMv) tz_list_add(mx, my, 0, 0);
  }
  tz_run(P, t);
  if (useIntMv) {
    int cx = FCU_UNI(g_S.tz_bx) << 2, cy = FCU_UNI(g_S.tz_by) << 2; clip_mv(P, cu, cx, cy);
    rltx = cx - (range << 2); rlty = cy - (range << 2); rrbx = cx + (range << 2); rrby = cy + (range << 2);
    clip_mv(P, cu, rltx, rlty); clip_mv(P, cu, rrbx, rrby);
    rltx >>= 2; rlty >>= 2; rrbx >>= 2; rrby >>= 2;
  }
  int sx = FCU_UNI(g_S.tz_bx), sy = FCU_UNI(g_S.tz_by);
  for (int d = 1; d <= range; d *= 2) {                        /* first search: stops three rounds after the last improvement */
    FCU_SERIAL tz_diamond_list(ltx, lty, rbx, rby, sx, sy, d);
    tz_run(P, t);
    if (FCU_UNI(g_S.tz_round) >= 3) break;
  }
  if (FCU_UNI(g_S.tz_dist) == 1) { FCU_SERIAL { g_S.tz_dist = 0; tz_two_point_list(ltx, lty, rbx, rby); } tz_run(P, t); }
  if (FCU_UNI(g_S.tz_dist) > raster) {                         /* raster search, step 5: one position per lane, raster order wins ties */
    const int nx = (rrbx - rltx) / raster + 1, ny = (rrby - rlty) / raster + 1;
    FCU_SERIAL { g_S.tz_dist = raster; g_S.me_best = ~0ull; }
    FCU_FOR_LANES {
      unsigned long long best = ~0ull;
      for (int p = lane; p < nx * ny; p += 64) {
        const int yy = p / nx, xx = p - yy * nx, x = rltx + xx * raster, y = rlty + yy * raster;
        uint32_t s = sad_block(t.org, t.ref0 + (t.py + y) * t.rs + t.px + x, t.rs, t.w, t.h, t.step);
        if (t.step == 2) s <<= 1;
        s += motion_cost(P, mv_bits(x, y, t.predx, t.predy, 2));
        const unsigned long long key = ((unsigned long long)s << 32) | (unsigned)p;
        if (key < best) best = key;
      }
      FCU_WAVE_MIN64(&g_S.me_best, best);
    }
    FCU_SERIAL {
      const uint32_t s = (uint32_t)(g_S.me_best >> 32); const int p = (int)(g_S.me_best & 0xffffffffull);
      if (nx > 0 && ny > 0 && s < g_S.tz_best) { g_S.tz_best = s; g_S.tz_bx = rltx + (p % nx) * raster; g_S.tz_by = rlty + (p / nx) * raster; g_S.tz_dist = raster; g_S.tz_round = 0; g_S.tz_point = 0; }
    }
  }
  while (FCU_UNI(g_S.tz_dist) > 0) {                           /* star refinement */
    sx = FCU_UNI(g_S.tz_bx); sy = FCU_UNI(g_S.tz_by);
    FCU_SERIAL { g_S.tz_dist = 0; g_S.tz_point = 0; }
    for (int d = 1; d < range + 1; d *= 2) { FCU_SERIAL tz_diamond_list(ltx, lty, rbx, rby, sx, sy, d); tz_run(P, t); }
    if (FCU_UNI(g_S.tz_dist) == 1) {
      const int two = FCU_UNI(g_S.tz_point) != 0;
      FCU_SERIAL { g_S.tz_dist = 0; if (two) tz_two_point_list(ltx, lty, rbx, rby); }
      if (two) tz_run(P, t);
    }
  }
}

FCU_DEV FCU_NOINLINE void motion_estimation(const CuObj *cu, int ps, int pu, int refIdx, int predx, int predy, uint32_t bitsIn)
{
  const Env E = env_get(); cu = FCU_UNI(cu); ps = FCU_UNI(ps); pu = FCU_UNI(pu); refIdx = FCU_UNI(refIdx); predx = FCU_UNI(predx); predy = FCU_UNI(predy); bitsIn = FCU_UNI(bitsIn);
  const Params &P = E.C->p; Scratch *G = E.G;
  const Pu g = pu_geom(cu->depth_cu, ps, pu);
  const uint8_t *org = G->org[cu->depth_cu].y + g.oy * 64 + g.ox;
  const int px = cu->x + g.ox, py = cu->y + g.oy, rng = P.search_range, rs = E.C->ref_stride[0];
  const uint8_t *ref0 = E.C->refs[refIdx][0];
  /* xSetSearchRange */
  int cx = predx, cy = predy; clip_mv(P, cu, cx, cy);
  int ltx = cx - (rng << 2), lty = cy - (rng << 2), rbx = cx + (rng << 2), rby = cy + (rng << 2);
  clip_mv(P, cu, ltx, lty); clip_mv(P, cu, rbx, rby);
  ltx >>= 2; lty >>= 2; rbx >>= 2; rby >>= 2;
  const int nx = rbx - ltx + 1, ny = rby - lty + 1, step = (P.fast_enc && g.h > 8) ? 2 : 1;
  int bx, by;
  FCU_TIC(pi_);
  if (P.fast_search) {                                       /* xPatternSearchFast -> xTZSearch; m_integerMv2Nx2N, TEncSearch.cpp:3822-3833 */
    TzCtx t; t.org = org; t.ref0 = ref0; t.rs = rs; t.px = px; t.py = py; t.w = g.w; t.h = g.h; t.step = step; t.predx = predx; t.predy = predy;
    const int usePred = ps != SIZE_2Nx2N || cu->depth_cu != 0;
    tz_search(P, cu, t, ltx, lty, rbx, rby, usePred, FCU_UNI(E.C->int_mv_r[refIdx][0]), FCU_UNI(E.C->int_mv_r[refIdx][1]));
    bx = FCU_UNI(g_S.tz_bx); by = FCU_UNI(g_S.tz_by);
    if (ps == SIZE_2Nx2N) FCU_SERIAL { E.C->int_mv_r[refIdx][0] = bx; E.C->int_mv_r[refIdx][1] = by; }
  } else {
  FCU_SERIAL g_S.me_best = ~0ull;
  FCU_FOR_LANES {                                            /* one candidate position per lane */
    unsigned long long best = ~0ull;
    for (int p = lane; p < nx * ny; p += 64) {
      const int yy = p / nx, xx = p - yy * nx, x = ltx + xx, y = lty + yy;
      uint32_t s = sad_block(org, ref0 + (py + y) * rs + px + x, rs, g.w, g.h, step);
      if (step == 2) s <<= 1;
      s += motion_cost(P, mv_bits(x, y, predx, predy, 2));
      const unsigned long long key = ((unsigned long long)s << 32) | (unsigned)p;     /* ties: the earlier raster position wins */
      if (key < best) best = key;
    }
    FCU_WAVE_MIN64(&g_S.me_best, best);
  }
  const unsigned bp = (unsigned)FCU_UNI((int)(unsigned)(g_S.me_best & 0xffffffffull));
  bx = ltx + (int)(bp % (unsigned)nx); by = lty + (int)(bp / (unsigned)nx);
  }
  FCU_ITOC(E, pi_, 13);                                        /* profile: integer search | sub-sample refinement */
  FCU_TIC(pf_);
  /* half-sample round, then quarter-sample round around the winner */
  int hx = 0, hy = 0, qx = 0, qy = 0; uint32_t bestD = 0;
  for (int round = 0; round < 2; round++) {
    const int basex = (bx << 2) + 2 * hx, basey = (by << 2) + 2 * hy, stp = round == 0 ? 2 : 1;
    /* The nine interpolated blocks, separably (as xExtDIFUpSamplingH / Q build them, TEncSearch.cpp:5431-5560): the round has
     * three horizontal positions, so three horizontally filtered planes (unshifted 8-tap sums, <= 15 bits) over the rows any
     * of its vertical positions touches, then nine vertical 8-tap passes.  Same integers as the two-dimensional sum
     * clip8((sum_v sum_h c_v c_h s + 2048) >> 12) that interp_sample evaluates, at 16 instead of 64 multiplies per sample. */
    const int iyMin = (basey - stp) >> 2, iyMax = (basey + stp) >> 2, rows = g.h + 7 + (iyMax - iyMin);
    FCU_FOR_LANES {
      if (lane < 9) g_S.acc[lane] = 0;
      for (int k = lane; k < 3 * rows * g.w; k += 64) {
        const int pl = k / (rows * g.w), r = k - pl * rows * g.w, yy = r / g.w, x = r - yy * g.w;
        const int mvx = basex + (pl - 1) * stp, fx = mvx & 3;
        const uint8_t *p = ref0 + (py + iyMin - 3 + yy) * rs + px + x + (mvx >> 2);
        int s;
        if (!fx) s = 64 * p[0];
        else { uint8_t b[8]; __builtin_memcpy(b, p - 3, 8); s = 0;                    /* the eight taps' samples in one load */
#pragma unroll
          for (int t = 0; t < 8; t++) s += k_luma_filter[fx][t] * b[t]; }
        G->me_h[pl][yy * 64 + x] = (int16_t)s;
      }
    }
    /* vertical passes: a lane owns one column of one block and slides an eight-row window of the horizontal plane down it
     * in registers -- one load per output sample instead of eight (the kernel waits on memory three quarters of the time) */
    FCU_FOR_LANES {
      for (int it = lane; it < 9 * g.w; it += 64) {
        const int c = it / g.w, x = it - c * g.w;
        const int8_t *t = round == 0 ? k_refine_h[c] : k_refine_q[c];
        const int mvy = basey + stp * t[1], fy = mvy & 3;
        const int16_t *q = G->me_h[t[0] + 1] + ((mvy >> 2) - iyMin) * 64 + x;           /* row of tap j of output row y: y + j (stored from iyMin - 3) */
        uint8_t *o = G->me_pred[c] + x;
        if (!fy) { for (int y = 0; y < g.h; y++) o[y * 64] = (uint8_t)clip8((64 * q[(y + 3) * 64] + 2048) >> 12); }
        else {
          const int f0 = k_luma_filter[fy][0], f1 = k_luma_filter[fy][1], f2 = k_luma_filter[fy][2], f3 = k_luma_filter[fy][3],
                    f4 = k_luma_filter[fy][4], f5 = k_luma_filter[fy][5], f6 = k_luma_filter[fy][6], f7 = k_luma_filter[fy][7];
          int w0 = q[0], w1 = q[64], w2 = q[128], w3 = q[192], w4 = q[256], w5 = q[320], w6 = q[384];
          for (int y = 0; y < g.h; y++) {
            const int w7 = q[(y + 7) * 64];
            const int s = f0 * w0 + f1 * w1 + f2 * w2 + f3 * w3 + f4 * w4 + f5 * w5 + f6 * w6 + f7 * w7;
            o[y * 64] = (uint8_t)clip8((s + 2048) >> 12);
            w0 = w1; w1 = w2; w2 = w3; w3 = w4; w4 = w5; w5 = w6; w6 = w7;
          }
        }
      }
    }
    if (P.had_me) {                                          /* Hadamard units of all nine blocks share the lanes: 9 x units items, one per lane */
      FCU_FOR_LANES {
        const int u8 = !(g.w & 7) && !(g.h & 7), us = u8 ? 8 : 4, bw = g.w / us, nb = bw * (g.h / us);
        for (int it = lane; it < 9 * nb; it += 64) {
          const int c = it / nb, u = it - c * nb, uy = u / bw, ux = u - uy * bw, o = uy * us * 64 + ux * us;
          const uint32_t s = u8 ? had_unit<8>(org + o, 64, G->me_pred[c] + o, 64) : had_unit<4>(org + o, 64, G->me_pred[c] + o, 64);
          FCU_ATOMIC_ADD(&g_S.acc[c], s);
        }
      }
    } else for (int c = 0; c < 9; c++) {
      FCU_FOR_LANES {
        if (P.had_me) had_block_lanes(lane, org, 64, G->me_pred[c], 64, g.w, g.h, &g_S.acc[c]);
        else { uint32_t s = 0; for (int i = lane; i < g.w * g.h; i += 64) s += (uint32_t)iabs(org[(i / g.w) * 64 + i % g.w] - G->me_pred[c][(i / g.w) * 64 + i % g.w]); FCU_WAVE_ADD(&g_S.acc[c], s); }
      }
    }
    uint32_t best = 0xffffffffu; int bi = 0;
    for (int c = 0; c < 9; c++) {
      const int8_t *t = round == 0 ? k_refine_h[c] : k_refine_q[c];
      uint32_t d = FCU_UNI(g_S.acc[c]);
      if (round == 0) d += motion_cost(P, mv_bits((bx << 1) + t[0], (by << 1) + t[1], predx, predy, 1));
      else d += motion_cost(P, mv_bits(basex + t[0], basey + t[1], predx, predy, 0));
      if (d < best) { best = d; bi = c; }
    }
    if (round == 0) { hx = k_refine_h[bi][0]; hy = k_refine_h[bi][1]; }
    else { qx = k_refine_q[bi][0]; qy = k_refine_q[bi][1]; bestD = best; }
  }
  const int mvx = (bx << 2) + 2 * hx + qx, mvy = (by << 2) + 2 * hy + qy;
  const uint32_t mvBits = mv_bits(mvx, mvy, predx, predy, 0), bits = bitsIn + mvBits;
  const uint32_t cost = (uint32_t)(FCU_FLOOR(1.0 * ((double)bestD - (double)motion_cost(P, mvBits))) + (double)motion_cost(P, bits));
  FCU_ITOC(E, pf_, 14);
  FCU_SERIAL { g_S.me_out[0] = mvx; g_S.me_out[1] = mvy; g_S.acc[12] = bits; g_S.acc[13] = cost; }
}

/* ---- predInterSearch (P slice, list 0 with 1..4 reference pictures): motion of every PU of the CU + its prediction in predt[d] */
FCU_DEV FCU_NOINLINE void pred_inter_search(CuObj *cu, int ps, int useMrg)
{
  const Env E = env_get(); cu = FCU_UNI(cu); ps = FCU_UNI(ps); useMrg = FCU_UNI(useMrg);
  const Params &P = E.C->p; Scratch *G = E.G;
  const int d = cu->depth_cu, npu = pu_count(ps);
  const int normalMC = !(useMrg && (CTU >> d) > 8 && npu == 2);   /* AMP_MRG: merge estimation only (TEncSearch.cpp:3098-3103) */
  for (int pu = 0; pu < npu; pu++) {
    const uint32_t mbBits = (ps == SIZE_2Nx2N || ps == SIZE_NxN) ? 1 : 3;      /* xGetBlkBits, P slice */
    int mvx = 0, mvy = 0, refBest = 0; uint32_t bitsT = mbBits;
    if (!normalMC) {                                         /* the cleared motion field (:3356-3363) */
      FCU_FOR_LANES { pu_set_motion(cu, ps, pu, lane, 0, 0, -1); pu_set_info(cu, ps, pu, lane, 0, 0, 0, 0, -1); }
    } else {
    /* uni-directional prediction, list 0: every reference index in turn, the first strictly cheapest wins (:3110-3190) */
    const int nRef = FCU_UNI(E.C->n_ref);
    uint32_t costBest = 0xffffffffu; int bMvx = 0, bMvy = 0, bPredx = 0, bPredy = 0, bMvp = 0; uint32_t bBits = mbBits;
    for (int refIdx = 0; refIdx < nRef; refIdx++) {
    uint32_t refBits = mbBits;
    if (nRef > 1) { refBits += (uint32_t)refIdx + 1; if (refIdx == nRef - 1) refBits--; }      /* ref_idx bins (:3114-3121) */
    FCU_TIC(p5_); int mvpIdx = estimate_mvp(cu, ps, pu, refIdx); FCU_ITOC(E, p5_, 5);
    int predx = FCU_UNI(g_S.amvp[mvpIdx][0]), predy = FCU_UNI(g_S.amvp[mvpIdx][1]);
    { FCU_TIC(p_); motion_estimation(cu, ps, pu, refIdx, predx, predy, refBits + 1); FCU_ITOC(E, p_, 6); }
    mvx = FCU_UNI(g_S.me_out[0]); mvy = FCU_UNI(g_S.me_out[1]);
    bitsT = FCU_UNI(g_S.acc[12]); uint32_t costT = FCU_UNI(g_S.acc[13]);
    {                                                        /* xCheckBestMVP */
      const int orgBits = (int)mv_bits(mvx, mvy, predx, predy, 0) + 1;
      int bestBits = orgBits, bestIdx = mvpIdx;
      for (int i = 0; i < 2; i++) {
        if (i == mvpIdx) continue;
        const int b = (int)mv_bits(mvx, mvy, FCU_UNI(g_S.amvp[i][0]), FCU_UNI(g_S.amvp[i][1]), 0) + 1;
        if (b < bestBits) { bestBits = b; bestIdx = i; }
      }
      if (bestIdx != mvpIdx) {
        predx = FCU_UNI(g_S.amvp[bestIdx][0]); predy = FCU_UNI(g_S.amvp[bestIdx][1]); mvpIdx = bestIdx;
        const uint32_t org = bitsT;
        bitsT = org - (uint32_t)orgBits + (uint32_t)bestBits;
        costT = (costT - motion_cost(P, org)) + motion_cost(P, bitsT);
      }
    }
    if (costT < costBest) { costBest = costT; bMvx = mvx; bMvy = mvy; bPredx = predx; bPredy = predy; bMvp = mvpIdx; bBits = bitsT; refBest = refIdx; }
    }
    mvx = bMvx; mvy = bMvy; bitsT = bBits;
    { const int predx = bPredx, predy = bPredy, mvpIdx = bMvp, rb = refBest;
      FCU_FOR_LANES { pu_set_motion(cu, ps, pu, lane, mvx, mvy, rb); pu_set_info(cu, ps, pu, lane, 0, 0, mvx - predx, mvy - predy, mvpIdx); } }
    }
    FCU_TIC(p7_);
    if (ps != SIZE_2Nx2N) {                                  /* merge estimation of the PU (TEncSearch.cpp:3448-3498) */
      /* the prediction error of the PU is a function of its vector: candidates that repeat the motion search's result or an
       * earlier candidate's vector (the usual case under coherent motion) reuse the error instead of predicting again */
      int seenX[6], seenY[6], seenR[6], nSeen = 0; uint32_t seenE[6];
      uint32_t meCost = 0xffffffffu;
      if (normalMC) { const uint32_t e = inter_pred_error(cu, ps, pu); meCost = e + motion_cost(P, bitsT); seenX[0] = mvx; seenY[0] = mvy; seenR[0] = refBest; seenE[0] = e; nSeen = 1; }
      FCU_SERIAL merge_candidates(cu, ps, pu);
      uint32_t mrgCost = 0xffffffffu; int mrgIdx = 0;
      const int nc = P.max_merge_cand;
      for (int c = 0; c < nc; c++) {                          /* xMergeEstimation */
        const int cx = FCU_UNI(g_S.mrg_mv[c][0]), cy = FCU_UNI(g_S.mrg_mv[c][1]), cr = FCU_UNI(g_S.mrg_ref[c]);
        int found = -1;
#pragma unroll
        for (int k = 0; k < 6; k++) if (k < nSeen && seenX[k] == cx && seenY[k] == cy && seenR[k] == cr) found = k;
        uint32_t cc;
        if (found >= 0) cc = seenE[found];
        else {
          FCU_FOR_LANES pu_set_motion(cu, ps, pu, lane, cx, cy, cr);
          cc = inter_pred_error(cu, ps, pu);
#pragma unroll
          for (int k = 0; k < 6; k++) if (k == nSeen) { seenX[k] = cx; seenY[k] = cy; seenR[k] = cr; seenE[k] = cc; }
          nSeen++;
        }
        uint32_t b = (uint32_t)c + 1; if (c == nc - 1) b--;
        cc += motion_cost(P, b);
        if (cc < mrgCost) { mrgCost = cc; mrgIdx = c; }
      }
      if (mrgCost < meCost) {
        const int cx = FCU_UNI(g_S.mrg_mv[mrgIdx][0]), cy = FCU_UNI(g_S.mrg_mv[mrgIdx][1]), cr = FCU_UNI(g_S.mrg_ref[mrgIdx]);
        FCU_FOR_LANES { pu_set_motion(cu, ps, pu, lane, cx, cy, cr); pu_set_info(cu, ps, pu, lane, 1, mrgIdx, 0, 0, -1); }
      } else {
        const int rb = refBest;
        FCU_FOR_LANES { pu_set_motion(cu, ps, pu, lane, mvx, mvy, rb); const int n = cu->nparts; for (int i = lane; i < n; i += 64) if (pu_covers(n, ps, pu, i)) { cu->merge_flag[i] = 0; cu->merge_idx[i] = 0; cu->inter_dir[i] = 1; } }
      }
    }
    FCU_ITOC(E, p7_, 7);
    { FCU_TIC(p_); mc_pu(cu, ps, pu, &G->predt[d], 0); FCU_ITOC(E, p_, 8); }
  }
}

/* ---- inter syntax (one lane, coder id c) ------------------------------------------------ */
FCU_DEV void code_skip_flag(const Env E, int c, const CuObj *cu, int part)                   /* TEncSbac.cpp:540-555 */
{
  const int z = cu->zidx + part, lx = (cu->x & ~63) + part_x(z), ly = (cu->y & ~63) + part_y(z);
  const Nb l = nb_motion(E, cu, lx - 1, ly, lx, ly), a = nb_motion(E, cu, lx, ly - 1, lx, ly);
  cab_bin(c, cu->skip[part], CTX_SKIP + (l.avail ? l.skip : 0) + (a.avail ? a.skip : 0));
}
FCU_DEV void code_pred_mode(int c, const CuObj *cu, int part) { cab_bin(c, cu->pred_mode[part] == MODE_INTRA, CTX_PRED_MODE); }
FCU_DEV void code_merge_index(const Env E, int c, const CuObj *cu, int part)
{
  const int idx = cu->merge_idx[part], n = E.C->p.max_merge_cand;
  for (int ui = 0; ui < n - 1; ui++) { const int sym = ui == idx ? 0 : 1; if (ui == 0) cab_bin(c, sym, CTX_MERGE_IDX); else cab_ep(c, 1); if (!sym) break; }
}
FCU_DEV void code_part_size_inter(const Env E, int c, const CuObj *cu, int part, int depth)              /* TEncSbac.cpp:436-520 */
{
  const int ps = cu->part_size[part], amp = E.C->p.amp && depth < MAXDEPTH;
  if (ps == SIZE_2Nx2N) { cab_bin(c, 1, CTX_PARTSIZE); return; }
  cab_bin(c, 0, CTX_PARTSIZE);
  const int hor = ps == SIZE_2NxN || ps == SIZE_2NxnU || ps == SIZE_2NxnD;
  cab_bin(c, hor, CTX_PARTSIZE1);
  if (!hor && depth == MAXDEPTH && !((CTU >> depth) == 8)) cab_bin(c, 1, CTX_PARTSIZE1 + 1);
  if (amp) {                                                   /* fourth part_mode context, then one bypass bin: 2NxnU / nLx2N 0, 2NxnD / nRx2N 1 */
    if (ps == SIZE_2NxN || ps == SIZE_Nx2N) cab_bin(c, 1, CTX_PARTSIZE1 + 2);
    else { cab_bin(c, 0, CTX_PARTSIZE1 + 2); cab_ep(c, 1); }
  }
}
FCU_DEV int ep_exgolomb_bins(uint32_t symbol, uint32_t count) { int n = 0; while (symbol >= (1u << count)) { n++; symbol -= 1u << count; count++; } return n + 1 + (int)count; }
/* codeRefFrmIdx, TEncSbac.cpp:743-775: first bin on context 0, second on context 1, the rest bypass (truncated unary) */
FCU_DEV void code_ref_idx(int c, int refIdx, int nRef)
{
  cab_bin(c, refIdx == 0 ? 0 : 1, CTX_REF);
  if (refIdx > 0) {
    const int refNum = nRef - 2; refIdx--;
    for (int ui = 0; ui < refNum; ui++) {
      const int sym = ui == refIdx ? 0 : 1;
      if (ui == 0) cab_bin(c, sym, CTX_REF + 1); else cab_ep(c, 1);
      if (!sym) break;
    }
  }
}
FCU_DEV void code_mvd(int c, int hor, int ver)                                             /* TEncSbac.cpp:780-830 */
{
  cab_bin(c, hor != 0, CTX_MVD); cab_bin(c, ver != 0, CTX_MVD);
  const uint32_t ah = (uint32_t)iabs(hor), av = (uint32_t)iabs(ver);
  if (hor) cab_bin(c, ah > 1, CTX_MVD + 1);
  if (ver) cab_bin(c, av > 1, CTX_MVD + 1);
  if (hor) { if (ah > 1) cab_ep(c, ep_exgolomb_bins(ah - 2, 1)); cab_ep(c, 1); }
  if (ver) { if (av > 1) cab_ep(c, ep_exgolomb_bins(av - 2, 1)); cab_ep(c, 1); }
}
FCU_DEV void code_pu_wise(const Env E, int c, const CuObj *cu, int part)                    /* TEncEntropy.cpp:456-507 */
{
  const int ps = cu->part_size[part], npu = pu_count(ps), n = NPART >> (2 * cu->depth[part]);
  const int off = (((0x51a24480u >> (4 * ps)) & 15) * n) >> 4;   /* g_auiPUOffset {0, 8, 4, 4, 2, 10, 1, 5} sixteenths of the CU (TEncEntropy.cpp:339) */
  for (int pu = 0, sp = part; pu < npu; pu++, sp += off) {
    cab_bin(c, cu->merge_flag[sp], CTX_MERGE_FLAG);
    if (cu->merge_flag[sp]) code_merge_index(E, c, cu, sp);
    else { if (E.C->n_ref > 1) code_ref_idx(c, cu->ref_idx[sp], E.C->n_ref); code_mvd(c, cu->mvd[sp][0], cu->mvd[sp][1]); cab_bin(c, cu->mvp_idx[sp], CTX_MVP_IDX); }
  }
}
FCU_DEV int qt_root_cbf(const CuObj *cu, int part) { return (cu->cbf[0][part] | cu->cbf[1][part] | cu->cbf[2][part]) & 1; }
FCU_DEV int min_tu_log2_inter(int depth) { int l = 6 - depth; if (l < LOG2_MINTU + 2) return LOG2_MINTU; l -= 2; return l > LOG2_MAXTU ? LOG2_MAXTU : l; }   /* QuadtreeTUMaxDepthInter 3 */
FCU_DEV int qt_cbf_ctx(const TU &tu, int comp) { return comp ? CTX_CBF_CHROMA + tu.tr_depth : CTX_CBF_LUMA + (tu.tr_depth == 0 ? 1 : 0); }

/* final-order transform tree of an inter CU (xEncodeTransform, inter branch): iterative, one lane */
FCU_DEV FCU_NOINLINE void encode_transform_inter(int c, const CuObj *cu, int cuPart, uint32_t root_k)
{
  const Env E = env_get(); c = FCU_UNI(c); cu = FCU_UNI(cu); cuPart = FCU_UNI(cuPart); const TU root = tu_of_key(FCU_UNI(root_k));
  TU *st = g_S.wk_st; int *ci = g_S.wk_ci; int sp = 0;
  st[0] = root; ci[0] = -1;
  while (sp >= 0) {
    const TU tu = st[sp];
    if (ci[sp] < 0) {
      const int part = cuPart + tu.part, trIdx = tu.tr_depth, subdiv = cu->tr_idx[part] > trIdx;
      if (tu.log2 > LOG2_MAXTU) { }
      else if (tu.log2 == LOG2_MINTU) { }
      else if (tu.log2 == min_tu_log2_inter(cu->depth[part])) { }
      else cab_bin(c, subdiv, CTX_SUBDIV + 5 - tu.log2);
      const int first = trIdx == 0;
      for (int comp = 1; comp < 3; comp++)
        if (first || tu.c_code_all)
          if (first || ((cu->cbf[comp][part] >> (trIdx - 1)) & 1)) {
            const int lowest = trIdx + ((subdiv && !(tu.cwo >= 8)) ? 1 : 0);
            cab_bin(c, (cu->cbf[comp][cuPart + tu_part_c(tu)] >> lowest) & 1, CTX_CBF_CHROMA + trIdx);
          }
      if (!subdiv) {
        if (!(trIdx == 0 && !((cu->cbf[1][part] & 1) || (cu->cbf[2][part] & 1)))) cab_bin(c, (cu->cbf[0][part] >> trIdx) & 1, CTX_CBF_LUMA + (trIdx == 0 ? 1 : 0));
        for (int comp = 0; comp < 3; comp++) {
          if (comp && tu.cw == 0) continue;
          if (!((cu->cbf[comp][part] >> trIdx) & 1)) continue;
          const int log2 = comp ? ilog2(tu.cw) : tu.log2, pc = cuPart + (comp ? tu_part_c(tu) : tu.part);
          const int16_t *coef = cu->coef[comp] + (comp ? (cuPart * 4 + tu.off_c) : (cuPart * 16 + tu.off_y));
          code_coeff_nxn<1>(c, coef, 1, -1, log2, comp, 0, cu->tskip[comp][pc], E.C->p, g_S.lane_abs[0]);
        }
        sp--; continue;
      }
      ci[sp] = 0;
    }
    if (ci[sp] >= 4) { sp--; continue; }
    { TU ch; tu_child(ch, st[sp], ci[sp], 1); ci[sp]++; sp++; st[sp] = ch; ci[sp] = -1; }
  }
}
FCU_DEV void encode_cu_syntax_inter(const Env E, int c, const CuObj *cu, int cuPart, int depth)   /* xAddSymbolBitsInter / xEncodeCU */
{
  code_skip_flag(E, c, cu, cuPart);
  if (cu->skip[cuPart]) { code_merge_index(E, c, cu, cuPart); return; }
  code_pred_mode(c, cu, cuPart);
  code_part_size_inter(E, c, cu, cuPart, depth);
  code_pu_wise(E, c, cu, cuPart);
  if (!(cu->merge_flag[cuPart] && cu->part_size[cuPart] == SIZE_2Nx2N)) cab_bin(c, qt_root_cbf(cu, cuPart), CTX_ROOT_CBF);
  if (!qt_root_cbf(cu, cuPart)) return;
  TU root; tu_root(root, depth);
  encode_transform_inter(c, cu, cuPart, tu_key(root));
}

/* xEncodeInterResidualQT on the search-time buffers: pass 3 = subdivision / cbf flags, pass 0..2 = coefficients of a
 * component; iterative, one lane */
FCU_DEV FCU_NOINLINE void encode_inter_residual_qt(int c, const CuObj *cu, uint32_t root_k, int pass)
{
  const Env E = env_get(); c = FCU_UNI(c); cu = FCU_UNI(cu); const TU root = tu_of_key(FCU_UNI(root_k)); pass = FCU_UNI(pass);
  TU *st = g_S.wk_st; int *ci = g_S.wk_ci; int sp = 0;
  st[0] = root; ci[0] = -1;
  while (sp >= 0) {
    const TU tu = st[sp];
    if (ci[sp] < 0) {
      const int cur = tu.tr_depth, trMode = cu->tr_idx[tu.part], subdiv = cur != trMode, log2 = tu.log2;
      if (pass == 3) {
        if (log2 <= LOG2_MAXTU && log2 > min_tu_log2_inter(cu->depth_cu)) cab_bin(c, subdiv, CTX_SUBDIV + 5 - log2);
        const int first = cur == 0;
        for (int comp = 1; comp < 3; comp++)
          if (first || tu.c_code_all)
            if (first || ((cu->cbf[comp][tu.part] >> (cur - 1)) & 1)) {
              const int lowest = cur + ((subdiv && !(tu.cw >= 8)) ? 1 : 0);
              cab_bin(c, (cu->cbf[comp][tu_part_c(tu)] >> lowest) & 1, CTX_CBF_CHROMA + cur);
            }
        if (!subdiv) cab_bin(c, (cu->cbf[0][tu.part] >> cur) & 1, CTX_CBF_LUMA + (cur == 0 ? 1 : 0));
      }
      if (!subdiv) {
        if (pass != 3 && !(pass && tu.cw == 0) && ((cu->cbf[pass][tu.part] >> trMode) & 1)) {
          const int layer = LOG2_MAXTU - log2, l2 = pass ? ilog2(tu.cw) : log2;
          code_coeff_nxn<1>(c, E.G->qt_coef[pass][layer] + (pass ? tu.off_c : tu.off_y), 1, -1, l2, pass, 0, cu->tskip[pass][pass ? tu_part_c(tu) : tu.part], E.C->p, g_S.lane_abs[0]);
        }
        sp--; continue;
      }
      if (!(pass == 3 || ((cu->cbf[pass][tu.part] >> cur) & 1))) { sp--; continue; }
      ci[sp] = 0;
    }
    if (ci[sp] >= 4) { sp--; continue; }
    { TU ch; tu_child(ch, st[sp], ci[sp], 0); ci[sp]++; sp++; st[sp] = ch; ci[sp] = -1; }
  }
}

/* ---- one transform unit coded whole: the bCheckFull part of xEstimateInterResidualQT.  Its Y / Cb / Cr (x transform-skip)
 * variants all restart from the snapshot in the go-on coder, so transform, RDOQ, reconstruction and bit count of all of
 * them run side by side.  Leaves the chosen levels / residual in the RQT layer buffers, the flags in the CU, the coder
 * after the TU's syntax in the go-on coder and (single bits, single distortion) in g_S.iv_bits[0] / g_S.iv_dist[0]. */
FCU_DEV FCU_NOINLINE void inter_tu_trials(CuObj *cu, uint32_t tu_k, int addZero)
{
  const Env E = env_get(); cu = FCU_UNI(cu); const TU tu = tu_of_key(FCU_UNI(tu_k)); addZero = FCU_UNI(addZero);
  Scratch *G = E.G; const Params &P = E.C->p;
  const int d = cu->depth_cu, trMode = tu.tr_depth, depth = d + trMode, log2 = tu.log2, layer = LOG2_MAXTU - log2, part = tu.part;
  const int NY = 1 << log2, NC = tu.cw, lc = NC ? ilog2(NC) : 0, ncomp = NC ? 3 : 1;
  const int tsY = P.transform_skip && NY == 4, tsC = P.transform_skip && NC == 4;
  /* variant table (uniform): v = 2 * comp + ts; block b = comp at p_resi[boff[b]] */
  const int bN[3] = { NY, NC, NC }, bl2[3] = { log2, lc, lc }, boff[3] = { 0, NY * NY, NY * NY + NC * NC };
  const int btot = NY * NY + (NC ? 2 * NC * NC : 0);
  auto voff = [&](int v) { return (v & 1) ? 1536 + boff[v >> 1] : boff[v >> 1]; };        /* transform-skip variants (4x4 only) live behind the coded ones */
  auto vok = [&](int v) { const int comp = v >> 1; return comp < ncomp && (!(v & 1) || (comp ? tsC : tsY)); };
  FCU_QTIC(q_);
  FCU_FOR_LANES {                                            /* residual blocks + first transform stage */
    for (int i = lane; i < tu.nparts; i += 64) cu->tr_idx[part + i] = (uint8_t)trMode;
    if (lane < 6) { g_S.iv_top[lane] = -1; g_S.acc[lane] = 0; g_S.acc[6 + (lane >> 1)] = 0; }
    for (int i = lane; i < btot; i += 64) {
      const int b = i < boff[1] ? 0 : (i < boff[2] ? 1 : 2), k = i - boff[b], N = bN[b], y = k / N, x = k - y * N;
      const int16_t *r = (b == 0 ? G->resi_cu.y + tu.y * 64 + tu.x : (b == 1 ? G->resi_cu.u : G->resi_cu.v) + tu.cy * 32 + tu.cx);
      G->p_resi[i] = r[y * (b ? 32 : 64) + x];
    }
  }
  FCU_FOR_LANES {
    uint32_t z[3] = { 0, 0, 0 };
    for (int i = lane; i < btot; i += 64) { const int b = i < boff[1] ? 0 : (i < boff[2] ? 1 : 2); const int r = G->p_resi[i]; z[b] += (uint32_t)(r * r); }
    for (int b = 0; b < ncomp; b++)
      by_log2(bl2[b], [&](auto L) { constexpr int LG = decltype(L)::value; RowCache<(1 << LG), int16_t> rc; for (int k = lane; k < bN[b] * bN[b]; k += 64) G->p_tmp[boff[b] + k] = fwd1<LG>(rc, G->p_resi + boff[b], 0, k); });
    for (int b = 0; b < 3; b++) FCU_WAVE_ADD(&g_S.acc[6 + b], z[b]);
  }
  FCU_FOR_LANES {                                            /* second stage -> level_double in scan order (SCAN_DIAG) */
    est_build(CAB_GOON, lane);
    for (int v = 0; v < 6; v++) {
      if (!vok(v)) continue;
      const int comp = v >> 1, N = bN[comp], l2 = bl2[comp], n2 = N * N, o = voff(v);
      const int qp = comp ? P.qp_c : P.qp, qbits = rdoq_qbits(l2, qp), qscale = k_quant_scales[qp % 6];
      const uint16_t *iscan = k_iscan + k_scan_off[0 * 4 + l2 - 2];
      auto put = [&](int k, int32_t t) {
        const int sp = iscan[k]; const int32_t ld = level_double(t, qscale, qbits);
        G->p_lscan[o + sp] = ld;
        if (level_nonzero(ld, qbits)) FCU_ATOMIC_MAX(&g_S.iv_top[v], sp);
      };
      if (v & 1) { for (int k = lane; k < n2; k += 64) put(k, (int32_t)G->p_resi[boff[comp] + k] << (15 - 8 - l2)); }
      else by_log2(l2, [&](auto L) { constexpr int LG = decltype(L)::value; RowCache<(1 << LG), int32_t> rc; for (int k = lane; k < n2; k += 64) put(k, fwd2<LG>(rc, G->p_tmp + boff[comp], 0, k)); });
    }
  }
  FCU_QTOC(E, q_, 0);
  /* RDOQ, all variants priced against the snapshot.  Blocks of 8x8 and more (they have no transform-skip variant): one after
   * the other with the whole wave (rdoq_wave); 4x4 blocks: one variant per lane -- rdoq() and code_coeff_nxn() take block size
   * and channel type as wave-uniform (scalar) arguments, so the luma variants and the chroma variants go in two rounds */
  const int waveY = FCU_UNI((int)(log2 >= 3 && P.rdoq)), waveC = FCU_UNI((int)(NC && lc >= 3 && P.rdoq));
  for (int v = 0; v < 6; v += 2) {
    const int comp = v >> 1;
    if (!(comp ? waveC : waveY) || comp >= ncomp) continue;
    const int cbfCtx = (comp == 0 && trMode == 0) ? EST_ROOT_CBF : qt_cbf_ctx(tu, comp);      /* blockRootCbpBits, TComTrQuant.cpp:2358 */
    rdoq_wave(CAB_GOON, G->p_lscan + voff(v), G->p_qscan + voff(v), FCU_UNI(g_S.iv_top[v]), bl2[comp], comp ? 1 : 0, 0, cbfCtx, P, G->r_rec + voff(v), G->r_cg + v * 64);
    FCU_SERIAL { g_S.iv_abs[v] = g_S.rw_abs; g_S.iv_lsp[v] = g_S.rw_lsp; }
  }
  for (int ch = 0; ch < (NC ? 2 : 1); ch++) {
    if (ch ? waveC : waveY) continue;
    FCU_FOR_LANES {
      if (lane < 6 && vok(lane) && ((lane >> 1) != 0) == (ch != 0)) {
        const int comp = lane >> 1, l2 = bl2[comp], o = voff(lane);
        const int cbfCtx = (comp == 0 && trMode == 0) ? EST_ROOT_CBF : qt_cbf_ctx(tu, comp);      /* blockRootCbpBits, TComTrQuant.cpp:2358 */
        const RdoqOut r = ((lane & 1) ? P.rdoq_ts : P.rdoq)
          ? rdoq<0, 1>(CAB_GOON, G->p_lscan + o, G->p_qscan + o, 1, g_S.iv_top[lane], l2, comp ? 1 : 0, 0, cbfCtx, P, G->r_rec + o, G->r_cg + lane * 64)   /* Cb and Cr share every parameter the call reads */
          : quant_plain(G->p_lscan + o, G->p_qscan + o, 1, g_S.iv_top[lane], l2, comp ? 1 : 0, P);
        g_S.iv_abs[lane] = r.abs_sum; g_S.iv_lsp[lane] = r.last;
      }
    }
  }
  FCU_SERIAL { E.C->n_tu_trials += (unsigned long long)(ncomp + (tsY ? 1 : 0) + (tsC ? 2 : 0)); }
  FCU_QTOC(E, q_, 1);
  FCU_FOR_LANES {                                            /* dequantisation (transposed for the inverse stages) */
    for (int v = 0; v < 6; v++) {
      if (!vok(v) || g_S.iv_abs[v] <= 0) continue;
      const int comp = v >> 1, N = bN[comp], l2 = bl2[comp], n2 = N * N, o = voff(v);
      const DeqParams dq = deq_params(l2, comp ? P.qp_c : P.qp);
      const uint16_t *iscan = k_iscan + k_scan_off[0 * 4 + l2 - 2];
      const int cgEnd = ((g_S.iv_top[v] >> 4) + 1) << 4;
      for (int k = lane; k < n2; k += 64) { const int sp = iscan[k]; const int q = sp < cgEnd ? G->p_qscan[o + sp] : 0; G->p_tmp[2048 + o + ((v & 1) ? k : tr_index(k, l2))] = dequant1(q, dq); }
    }
  }
  FCU_FOR_LANES {
    for (int v = 0; v < 6; v++) {
      if (!vok(v) || g_S.iv_abs[v] <= 0 || (v & 1)) continue;
      const int comp = v >> 1, l2 = bl2[comp], n2 = bN[comp] * bN[comp], o = voff(v);
      by_log2(l2, [&](auto L) { constexpr int LG = decltype(L)::value; RowCache<(1 << LG), int32_t> rc; for (int k = lane; k < n2; k += 64) G->p_tcoef[o + k] = inv1<LG>(rc, G->p_tmp + 2048 + o, 0, k); });
    }
  }
  FCU_FOR_LANES {                                            /* reconstructed residual of every variant + its SSE against the residual */
    for (int v = 0; v < 6; v++) {
      if (!vok(v)) continue;
      const int comp = v >> 1, l2 = bl2[comp], n2 = bN[comp] * bN[comp], o = voff(v);
      uint32_t sse = 0;
      if (g_S.iv_abs[v] > 0) {
        for (int k = lane; k < n2; k += 64) {
          int rr;
          if (v & 1) { const int s = 15 - 8 - l2; rr = (int16_t)((G->p_tmp[2048 + o + k] + (1 << (s - 1))) >> s); }
          else by_log2(l2, [&](auto L) { rr = (int16_t)inv2<decltype(L)::value>(G->p_tcoef + o, 0, k); });
          G->p_resi[2048 + o + k] = (int16_t)rr;
          const int e = rr - G->p_resi[boff[comp] + k]; sse += (uint32_t)(e * e);
        }
      }
      FCU_WAVE_ADD(&g_S.acc[v], sse);
    }
  }
  FCU_QTOC(E, q_, 2);
  for (int ch = 0; ch < (NC ? 2 : 1); ch++) {                  /* bits of (cbf, coefficients) per variant on lane-private coders */
    FCU_FOR_LANES {
      if (lane < 6 && vok(lane) && ((lane >> 1) != 0) == (ch != 0) && g_S.iv_abs[lane] > 0) {
        const int comp = lane >> 1, l2 = bl2[comp], o = voff(lane), c = CAB_LANE0 + lane;
        cab_copy1(&g_S.cab[c], &g_S.cab[CAB_GOON]); cab_reset_bits(c);
        cab_bin(c, 1, qt_cbf_ctx(tu, comp));
        code_coeff_nxn<0>(c, G->p_qscan + o, 1, g_S.iv_lsp[lane], l2, comp ? 1 : 0, 0, lane & 1, P, g_S.lane_abs[lane]);
        g_S.iv_bits[lane] = cab_bits(c);
      }
    }
  }
  FCU_QTOC(E, q_, 3);
  FCU_SERIAL {                                               /* per component: coded / skipped / transform-skip (TEncSearch.cpp:4640-4900) */
    const uint64_t low = g_S.cab[CAB_GOON].frac & 32767;
    for (int comp = 0; comp < ncomp; comp++) {
      const uint32_t sseZ = g_S.acc[6 + comp];
      const uint32_t nonDist = comp ? (uint32_t)(P.chroma_weight * (double)sseZ) : sseZ;
      const uint32_t nonBits = (uint32_t)((low + (uint64_t)ctx_bits(CAB_GOON, qt_cbf_ctx(tu, comp), 0)) >> 15);
      const double nonCost = rd_cost(P, nonBits, nonDist);
      if (addZero) g_S.iq_zero += nonDist;
      double minCost = FCU_MAX_DOUBLE; int bAbs = 0, bTs = 0; uint32_t bDist = 0;
      const int nModes = (comp ? tsC : tsY) ? 2 : 1;
      for (int ts = 0; ts < nModes; ts++) {
        const int v = 2 * comp + ts; int curAbs = g_S.iv_abs[v];
        uint32_t curDist; double curCost;
        if (curAbs > 0) { const uint32_t s = g_S.acc[v]; curDist = comp ? (uint32_t)(P.chroma_weight * (double)s) : s; curCost = rd_cost(P, g_S.iv_bits[v], curDist); }
        else if (ts == 1) { curDist = 0; curCost = FCU_MAX_DOUBLE; }
        else { curDist = nonDist; curCost = nonCost; }
        if (curCost < minCost || (ts == 1 && curCost == minCost)) {
          if (ts == 0 && (nonCost < curCost || curAbs == 0)) { curAbs = 0; curDist = nonDist; curCost = nonCost; }
          bAbs = curAbs; bDist = curDist; minCost = curCost; bTs = ts;
        }
      }
      g_S.it_abs[comp] = bAbs; g_S.it_ts[comp] = bTs; g_S.it_dist[comp] = bDist;
    }
  }
  FCU_FOR_LANES {                                            /* publish the winners: levels (scan order), residual, flags */
    for (int comp = 0; comp < ncomp; comp++) {
      const int N = bN[comp], n2 = N * N, v = 2 * comp + g_S.it_ts[comp], o = voff(v), coded = g_S.it_abs[comp] > 0;
      const int cgEnd = ((g_S.iv_top[v] >> 4) + 1) << 4;
      int16_t *cf = G->qt_coef[comp][layer] + (comp ? tu.off_c : tu.off_y);
      int16_t *rq = (comp == 0 ? G->qt_resi[layer].y + tu.y * 64 + tu.x : (comp == 1 ? G->qt_resi[layer].u : G->qt_resi[layer].v) + tu.cy * 32 + tu.cx);
      for (int k = lane; k < n2; k += 64) {
        cf[k] = (coded && k < cgEnd) ? G->p_qscan[o + k] : (int16_t)0;
        rq[(k / N) * (comp ? 32 : 64) + k % N] = coded ? G->p_resi[2048 + o + k] : (int16_t)0;
      }
      const int cpart = comp ? tu_part_c(tu) : part, cnp = comp ? tu_nparts_c(tu) : tu.nparts;
      for (int i = lane; i < cnp; i += 64) { cu->tskip[comp][cpart + i] = (uint8_t)g_S.it_ts[comp]; cu->cbf[comp][cpart + i] = (uint8_t)((coded ? 1 : 0) << trMode); }
    }
  }
  FCU_QTOC(E, q_, 4);
  FCU_SERIAL {                                               /* syntax of the whole TU from the snapshot: subdivision flag, cbf Cb Cr Y, coefficients Y Cb Cr */
    const int c = CAB_GOON;
    cab_reset_bits(c);
    if (log2 > min_tu_log2_inter(d)) cab_bin(c, 0, CTX_SUBDIV + 5 - log2);
    for (int k = 0; k < 3; k++) { const int comp = (k + 1) % 3; if (comp >= ncomp) continue; cab_bin(c, g_S.it_abs[comp] > 0, qt_cbf_ctx(tu, comp)); }
    uint32_t dist = 0;
    for (int comp = 0; comp < ncomp; comp++) {
      if (g_S.it_abs[comp] > 0)
        code_coeff_nxn<1>(c, G->qt_coef[comp][layer] + (comp ? tu.off_c : tu.off_y), 1, -1, bl2[comp], comp, 0, g_S.it_ts[comp], P, g_S.lane_abs[0]);
      dist += g_S.it_dist[comp];
    }
    g_S.iv_bits[0] = cab_bits(c); g_S.iv_dist[0] = dist;
  }
  FCU_QTOC(E, q_, 5);
  (void)depth;
}

/* ---- xEstimateInterResidualQT.  LEVEL = recursion level; adds (cost, bits, distortion) to g_S.iq_*[LEVEL]. */
template <int LEVEL>
FCU_DEV FCU_NOINLINE void est_inter_residual_qt(CuObj *cu, uint32_t tu_k, int addZero)
{
  const Env E = env_get(); cu = FCU_UNI(cu); const TU tu = tu_of_key(FCU_UNI(tu_k)); addZero = FCU_UNI(addZero);
  const Params &P = E.C->p;
  const int d = cu->depth_cu, part = tu.part, trMode = tu.tr_depth, depth = d + trMode, log2 = tu.log2;
  const int checkFull = log2 <= LOG2_MAXTU, checkSplit = log2 > min_tu_log2_inter(d);
  double singleCost = FCU_MAX_DOUBLE; uint32_t singleBits = 0, singleDist = 0;
  int bestTS[3] = { 0, 0, 0 };
  FCU_FOR_LANES cab_copy(slot_ptr(E, depth, CI_QT_TRAFO_ROOT), &g_S.cab[CAB_GOON], lane);
  if (checkFull) {
    inter_tu_trials(cu, tu_key(tu), addZero);
    singleBits = FCU_UNI(g_S.iv_bits[0]); singleDist = FCU_UNI(g_S.iv_dist[0]);
    singleCost = rd_cost(P, singleBits, singleDist);
    for (int c = 0; c < 3; c++) bestTS[c] = FCU_UNI(g_S.it_ts[c]);
  }
  if (checkSplit) {
    if constexpr (LEVEL < 2) {
      if (checkFull) { FCU_FOR_LANES cab_copy(slot_ptr(E, depth, CI_QT_TRAFO_TEST), &g_S.cab[CAB_GOON], lane); FCU_FOR_LANES cab_copy(&g_S.cab[CAB_GOON], slot_ptr(E, depth, CI_QT_TRAFO_ROOT), lane); }
      int bestCbf[3];
      for (int c = 0; c < 3; c++) bestCbf[c] = FCU_UNI((int)((cu->cbf[c][part] >> trMode) & 1));
      FCU_SERIAL { g_S.iq_cost[LEVEL + 1] = 0; g_S.iq_bits[LEVEL + 1] = 0; g_S.iq_dist[LEVEL + 1] = 0; }
      for (int i = 0; i < 4; i++) { TU c; tu_child(c, tu, i, 0); est_inter_residual_qt<LEVEL + 1>(cu, tu_key(c), checkFull ? 0 : addZero); }
      const int q = tu.nparts >> 2;
      FCU_SERIAL {
        int any = 0;
        for (int c = 0; c < 3; c++) { int yuv = 0; for (int i = 0; i < 4; i++) yuv |= (cu->cbf[c][part + i * q] >> (trMode + 1)) & 1; g_S.uni[c] = yuv; any |= yuv; }
        g_S.uni[3] = any;
      }
      FCU_FOR_LANES { for (int c = 0; c < 3; c++) for (int o = lane; o < 4 * q; o += 64) cu->cbf[c][part + o] |= (uint8_t)(g_S.uni[c] << trMode); cab_copy(&g_S.cab[CAB_GOON], slot_ptr(E, depth, CI_QT_TRAFO_ROOT), lane); }
      { FCU_QTIC(q6_);
      FCU_SERIAL {
        cab_reset_bits(CAB_GOON);
        encode_inter_residual_qt(CAB_GOON, cu, tu_key(tu), 3);
        for (int c = 0; c < 3; c++) encode_inter_residual_qt(CAB_GOON, cu, tu_key(tu), c);
        g_S.iv_bits[1] = cab_bits(CAB_GOON);
      }
      FCU_QTOC(E, q6_, 6); }
      const uint32_t subBits = FCU_UNI(g_S.iv_bits[1]), subDist = FCU_UNI(g_S.iq_dist[LEVEL + 1]);
      const double subCost = rd_cost(P, subBits, subDist);
      const int cbfAny = FCU_UNI(g_S.uni[3]);
      if (!checkFull || (cbfAny && subCost < singleCost)) { FCU_SERIAL { g_S.iq_cost[LEVEL] += subCost; g_S.iq_bits[LEVEL] += subBits; g_S.iq_dist[LEVEL] += subDist; } return; }
      FCU_FOR_LANES {
        for (int i = lane; i < tu.nparts; i += 64) { cu->tr_idx[part + i] = (uint8_t)trMode; for (int c = 0; c < 3; c++) { cu->cbf[c][part + i] = (uint8_t)(bestCbf[c] << trMode); cu->tskip[c][part + i] = (uint8_t)bestTS[c]; } }
        cab_copy(&g_S.cab[CAB_GOON], slot_ptr(E, depth, CI_QT_TRAFO_TEST), lane);
      }
    }
  }
  FCU_SERIAL { g_S.iq_cost[LEVEL] += singleCost; g_S.iq_bits[LEVEL] += singleBits; g_S.iq_dist[LEVEL] += singleDist; }
}

/* xSetInterResidualQTData: leaves of the chosen tree -> CU levels (spatial 0) / residual samples (spatial 1); iterative */
FCU_DEV FCU_NOINLINE void set_inter_residual_qt_data(CuObj *cu, int spatial)
{
  const Env E = env_get(); cu = FCU_UNI(cu); spatial = FCU_UNI(spatial);
  Scratch *G = E.G;
  TU *st = g_S.wk_st; int *ci = g_S.wk_ci; int sp = 0;
  tu_root(st[0], cu->depth_cu); ci[0] = -1;
  while (sp >= 0) {
    const TU tu = st[sp];
    if (ci[sp] < 0) {
      if (tu.tr_depth == cu->tr_idx[tu.part]) {
        const int layer = LOG2_MAXTU - tu.log2;
        FCU_FOR_LANES {
          for (int comp = 0; comp < 3; comp++) {
            if (comp && tu.cw == 0) continue;
            const int N = comp ? tu.cw : (1 << tu.log2), bs = comp ? 32 : 64, bx = comp ? tu.cx : tu.x, by = comp ? tu.cy : tu.y, off = comp ? tu.off_c : tu.off_y;
            if (spatial) {
              const int16_t *s = (comp == 0 ? G->qt_resi[layer].y : (comp == 1 ? G->qt_resi[layer].u : G->qt_resi[layer].v)) + by * bs + bx;
              int16_t *t = (comp == 0 ? G->resi_best.y : (comp == 1 ? G->resi_best.u : G->resi_best.v)) + by * bs + bx;
              for (int k = lane; k < N * N; k += 64) t[(k / N) * bs + k % N] = s[(k / N) * bs + k % N];
            } else for (int k = lane; k < N * N; k += 64) cu->coef[comp][off + k] = G->qt_coef[comp][layer][off + k];
          }
        }
        sp--; continue;
      }
      ci[sp] = 0;
    }
    if (ci[sp] >= 4) { sp--; continue; }
    { TU ch; tu_child(ch, st[sp], ci[sp], 0); ci[sp]++; sp++; st[sp] = ch; ci[sp] = -1; }
  }
}

/* ---- encodeResAndCalcRdInterCU: the prediction of the whole CU is in predt[d] */
FCU_DEV FCU_NOINLINE void encode_res_and_calc_rd_inter_cu(CuObj *cu, int skipResidual)
{
  const Env E = env_get(); cu = FCU_UNI(cu); skipResidual = FCU_UNI(skipResidual);
  Scratch *G = E.G; const Params &P = E.C->p;
  const int d = cu->depth_cu, s = CTU >> d, n = cu->nparts, hs = s >> 1;
  Yuv *org = &G->org[d], *pred = &G->predt[d], *rec = &G->reco[d][1 - g_S.reco_best_idx[d]];
  if (skipResidual) {
    FCU_FOR_LANES {
      if (lane < 3) g_S.acc[lane] = 0;
      for (int i = lane; i < n; i += 64) cu->skip[i] = 1;
    }
    FCU_FOR_LANES {
      uint32_t e0 = 0, e1 = 0, e2 = 0;
      for (int i = lane; i < s * s; i += 64) { const int o = (i / s) * 64 + i % s; const int v = pred->y[o]; rec->y[o] = (uint8_t)v; const int e = org->y[o] - v; e0 += (uint32_t)(e * e); }
      for (int i = lane; i < hs * hs; i += 64) { const int o = (i / hs) * 32 + i % hs; int v = pred->u[o]; rec->u[o] = (uint8_t)v; int e = org->u[o] - v; e1 += (uint32_t)(e * e); v = pred->v[o]; rec->v[o] = (uint8_t)v; e = org->v[o] - v; e2 += (uint32_t)(e * e); }
      FCU_WAVE_ADD(&g_S.acc[0], e0); FCU_WAVE_ADD(&g_S.acc[1], e1); FCU_WAVE_ADD(&g_S.acc[2], e2);
    }
    FCU_FOR_LANES cab_copy(&g_S.cab[CAB_GOON], slot_ptr(E, d, CI_CURR_BEST), lane);
    FCU_SERIAL {
      cab_reset_bits(CAB_GOON);
      code_skip_flag(E, CAB_GOON, cu, 0); code_merge_index(E, CAB_GOON, cu, 0);
      cu->bits = cab_bits(CAB_GOON);
      cu->dist = g_S.acc[0] + (uint32_t)(P.chroma_weight * (double)g_S.acc[1]) + (uint32_t)(P.chroma_weight * (double)g_S.acc[2]);
      cu->cost = rd_cost(P, cu->bits, cu->dist);
    }
    FCU_FOR_LANES cab_copy(slot_ptr(E, d, CI_TEMP_BEST), &g_S.cab[CAB_GOON], lane);
    return;
  }
  /* memo lookup (Scratch::memo_*): a slot of this depth whose motion field equals this candidate's */
  const int memoBase = d == 0 ? 0 : (d == 1 ? 6144 : (d == 2 ? 6144 + 1536 : 6144 + 1536 + 384)), ySz = s * s, cSz = hs * hs;
  int hit = -1;
  for (int k = 0; k < MEMO_K && hit < 0; k++) {
    if (!FCU_UNI(G->memo_valid[d][k])) continue;
    FCU_SERIAL g_S.uni[5] = 0;
    FCU_FOR_LANES {
      int bad = 0;
      for (int i = lane; i < n; i += 64) bad |= (G->memo_mv[d][k][i][0] != cu->mv[i][0]) | (G->memo_mv[d][k][i][1] != cu->mv[i][1]) | (G->memo_ref[d][k][i] != cu->ref_idx[i]);
      if (bad) g_S.uni[5] = 1;
    }
    if (!FCU_UNI(g_S.uni[5])) hit = k;
  }
  int zeroOut;
  if (hit >= 0) {
    zeroOut = FCU_UNI(G->memo_zero[d][hit]);
    const int16_t *mc = G->memo_coef + hit * MEMO_POOL + memoBase, *mr = G->memo_resi + hit * MEMO_POOL + memoBase;
    FCU_FOR_LANES {
      for (int i = lane; i < n; i += 64) { cu->tr_idx[i] = G->memo_tr_idx[d][hit][i]; for (int c = 0; c < 3; c++) { cu->cbf[c][i] = G->memo_cbf[d][hit][c][i]; cu->tskip[c][i] = G->memo_tskip[d][hit][c][i]; } }
      if (!zeroOut) {
        for (int i = lane; i < ySz; i += 64) { cu->coef[0][i] = mc[i]; G->resi_best.y[(i / s) * 64 + i % s] = mr[i]; }
        for (int i = lane; i < cSz; i += 64) {
          cu->coef[1][i] = mc[ySz + i]; cu->coef[2][i] = mc[ySz + cSz + i];
          const int o = (i / hs) * 32 + i % hs; G->resi_best.u[o] = mr[ySz + i]; G->resi_best.v[o] = mr[ySz + cSz + i];
        }
      }
    }
  } else {
  FCU_FOR_LANES {                                            /* residual of the CU */
    for (int i = lane; i < s * s; i += 64) { const int o = (i / s) * 64 + i % s; G->resi_cu.y[o] = (int16_t)(org->y[o] - pred->y[o]); }
    for (int i = lane; i < hs * hs; i += 64) { const int o = (i / hs) * 32 + i % hs; G->resi_cu.u[o] = (int16_t)(org->u[o] - pred->u[o]); G->resi_cu.v[o] = (int16_t)(org->v[o] - pred->v[o]); }
    cab_copy(&g_S.cab[CAB_GOON], slot_ptr(E, d, CI_CURR_BEST), lane);
    if (lane == 0) { g_S.iq_cost[0] = 0; g_S.iq_bits[0] = 0; g_S.iq_dist[0] = 0; g_S.iq_zero = 0; }
  }
  TU root; tu_root(root, d);
  { FCU_QTIC(q8_); est_inter_residual_qt<0>(cu, tu_key(root), 1); FCU_QTOC(E, q8_, 8); }
  FCU_SERIAL {
    cab_reset_bits(CAB_GOON); cab_bin(CAB_GOON, 0, CTX_ROOT_CBF);                            /* encodeQtRootCbfZero */
    const double zeroCost = rd_cost(P, cab_bits(CAB_GOON), g_S.iq_zero);
    g_S.uni[4] = (zeroCost < g_S.iq_cost[0] || !qt_root_cbf(cu, 0)) ? 1 : 0;
  }
  zeroOut = FCU_UNI(g_S.uni[4]);
  if (zeroOut) { FCU_FOR_LANES { for (int i = lane; i < n; i += 64) { cu->tr_idx[i] = 0; for (int c = 0; c < 3; c++) { cu->cbf[c][i] = 0; cu->tskip[c][i] = 0; } } } }
  else set_inter_residual_qt_data(cu, 0);
  }
  FCU_FOR_LANES {
    cab_copy(&g_S.cab[CAB_GOON], slot_ptr(E, d, CI_CURR_BEST), lane);
    if (cu->merge_flag[0] && cu->part_size[0] == SIZE_2Nx2N && zeroOut) for (int i = lane; i < n; i += 64) cu->skip[i] = 1;   /* xAddSymbolBitsInter */
  }
  { FCU_QTIC(q7_); FCU_SERIAL { cab_reset_bits(CAB_GOON); encode_cu_syntax_inter(E, CAB_GOON, cu, 0, d); cu->bits = cab_bits(CAB_GOON); } FCU_QTOC(E, q7_, 7); }
  if (!zeroOut && hit < 0) set_inter_residual_qt_data(cu, 1);
  if (hit < 0) {                                             /* memo store: next slot of this depth */
    const int k = FCU_UNI(G->memo_next[d]);
    int16_t *mc = G->memo_coef + k * MEMO_POOL + memoBase, *mr = G->memo_resi + k * MEMO_POOL + memoBase;
    FCU_FOR_LANES {
      for (int i = lane; i < n; i += 64) {
        G->memo_mv[d][k][i][0] = cu->mv[i][0]; G->memo_mv[d][k][i][1] = cu->mv[i][1]; G->memo_ref[d][k][i] = cu->ref_idx[i];
        G->memo_tr_idx[d][k][i] = cu->tr_idx[i]; for (int c = 0; c < 3; c++) { G->memo_cbf[d][k][c][i] = cu->cbf[c][i]; G->memo_tskip[d][k][c][i] = cu->tskip[c][i]; }
      }
      if (!zeroOut) {
        for (int i = lane; i < ySz; i += 64) { mc[i] = cu->coef[0][i]; mr[i] = G->resi_best.y[(i / s) * 64 + i % s]; }
        for (int i = lane; i < cSz; i += 64) {
          mc[ySz + i] = cu->coef[1][i]; mc[ySz + cSz + i] = cu->coef[2][i];
          const int o = (i / hs) * 32 + i % hs; mr[ySz + i] = G->resi_best.u[o]; mr[ySz + cSz + i] = G->resi_best.v[o];
        }
      }
      if (lane == 0) { G->memo_zero[d][k] = zeroOut; G->memo_valid[d][k] = 1; G->memo_next[d] = k + 1 == MEMO_K ? 0 : k + 1; }
    }
  }
  FCU_FOR_LANES { cab_copy(slot_ptr(E, d, CI_TEMP_BEST), &g_S.cab[CAB_GOON], lane); if (lane < 3) g_S.acc[lane] = 0; }
  FCU_FOR_LANES {                                            /* reconstruction + final distortion */
    uint32_t e0 = 0, e1 = 0, e2 = 0;
    for (int i = lane; i < s * s; i += 64) { const int o = (i / s) * 64 + i % s; const int v = clip8(pred->y[o] + (zeroOut ? 0 : G->resi_best.y[o])); rec->y[o] = (uint8_t)v; const int e = org->y[o] - v; e0 += (uint32_t)(e * e); }
    for (int i = lane; i < hs * hs; i += 64) {
      const int o = (i / hs) * 32 + i % hs;
      int v = clip8(pred->u[o] + (zeroOut ? 0 : G->resi_best.u[o])); rec->u[o] = (uint8_t)v; int e = org->u[o] - v; e1 += (uint32_t)(e * e);
      v = clip8(pred->v[o] + (zeroOut ? 0 : G->resi_best.v[o])); rec->v[o] = (uint8_t)v; e = org->v[o] - v; e2 += (uint32_t)(e * e);
    }
    FCU_WAVE_ADD(&g_S.acc[0], e0); FCU_WAVE_ADD(&g_S.acc[1], e1); FCU_WAVE_ADD(&g_S.acc[2], e2);
  }
  FCU_SERIAL {
    cu->dist = g_S.acc[0] + (uint32_t)(P.chroma_weight * (double)g_S.acc[1]) + (uint32_t)(P.chroma_weight * (double)g_S.acc[2]);
    cu->cost = rd_cost(P, cu->bits, cu->dist);
  }
}
