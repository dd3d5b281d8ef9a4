/*
 * fcu_emu.cpp -- TEST-ONLY build of the engine source as a plain C++ wave emulator
 * (-DFCU_EMU: the 64 lanes of each phase run as a loop).  It exists so that the engine's
 * decision logic can be compared with the oracle in `-m "not gpu"` tests where no GPU is
 * present.  It is NOT part of libfcu.so and nothing in the product path can reach it.
 */
#define FCU_EMU 1
#include "../../fast-cu-decision-hevc_amd/csrc/fcu_host.h"
#include <stdlib.h>
#include <new>

using namespace fcu;

struct EmuChain { Chain c; Scratch *g; };

extern "C" {
/* tools: bit 0 transform_skip, 1 transform_skip_fast, 2 sign_hiding, 3 strong_intra_smoothing; -1 = defaults */
void *fcu_emu_create(int width, int height, int qp, int slice_ctus, int tools, const uint8_t *oy, const uint8_t *ou, const uint8_t *ov,
                     uint8_t *ry, uint8_t *ru, uint8_t *rv, fcu_ctu_out *out)
{
  EmuChain *e = new EmuChain();
  memset(&e->c, 0, sizeof(e->c));
  fcu_frame_params fp; default_frame_params(fp, qp); fp.slice_ctus = slice_ctus;
  if (tools >= 0) { fp.transform_skip = tools & 1; fp.transform_skip_fast = (tools >> 1) & 1; fp.sign_hiding = (tools >> 2) & 1; fp.strong_intra_smoothing = (tools >> 3) & 1; }
  fill_params(e->c.p, width, height, fp);
  e->c.org[0] = oy; e->c.org[1] = ou; e->c.org[2] = ov; e->c.rec[0] = ry; e->c.rec[1] = ru; e->c.rec[2] = rv;
  e->c.stride[0] = width; e->c.stride[1] = e->c.stride[2] = width / 2;
  e->c.out = out;
  e->c.w_ctu = (width + 63) / 64; e->c.h_ctu = (height + 63) / 64; e->c.n_ctu = e->c.w_ctu * e->c.h_ctu;
  load_hot_tables();
  e->g = (Scratch *)calloc(1, sizeof(Scratch));
  return e;
}
/* P picture: lambda of the slice (fcu_ldp_slice on the host), search range, and the padded reference planes
 * (luma margin FCU_REF_MARGIN, pointers to the first byte of each padded plane) */
void fcu_emu_set_p(void *h, int qp, double lambda, int search_range, int fast_search, const uint8_t *py, const uint8_t *pu, const uint8_t *pv)
{
  EmuChain *e = (EmuChain *)h;
  const int width = e->c.p.width, height = e->c.p.height;
  fcu_frame_params fp; default_frame_params(fp, qp);
  fp.slice_ctus = e->c.p.slice_ctus; fp.slice_type = FCU_SLICE_P; fp.lambda = lambda; fp.search_range = search_range; fp.fast_search = fast_search;
  fill_params(e->c.p, width, height, fp);
  const int m = FCU_REF_MARGIN, sy = width + 2 * m, sc = width / 2 + m;
  e->c.ref_stride[0] = sy; e->c.ref_stride[1] = e->c.ref_stride[2] = sc;
  e->c.ref[0] = py + (size_t)m * sy + m; e->c.ref[1] = pu + (size_t)(m / 2) * sc + m / 2; e->c.ref[2] = pv + (size_t)(m / 2) * sc + m / 2;
  for (int k = 0; k < 3; k++) e->c.refs[0][k] = e->c.ref[k];
  e->c.n_ref = 1; e->c.poc = 1; e->c.ref_poc[0] = 0; e->c.col_poc = 0; e->c.col_ref_poc[0] = -1;
}
/* several reference pictures: planes[3r .. 3r+2] = padded Y, U, V of RefPicList0[r] (after fcu_emu_set_p with planes[0..2]) */
void fcu_emu_set_refs(void *h, int n, const uint8_t *const *planes, const int *pocs, int poc, const int *col_ref_pocs, int n_col)
{
  EmuChain *e = (EmuChain *)h;
  const int m = FCU_REF_MARGIN, sy = e->c.p.width + 2 * m, sc = e->c.p.width / 2 + m;
  for (int r = 0; r < n; r++) {
    e->c.refs[r][0] = planes[3 * r] + (size_t)m * sy + m; e->c.refs[r][1] = planes[3 * r + 1] + (size_t)(m / 2) * sc + m / 2; e->c.refs[r][2] = planes[3 * r + 2] + (size_t)(m / 2) * sc + m / 2;
    e->c.ref_poc[r] = pocs[r];
  }
  e->c.n_ref = n; e->c.poc = poc; e->c.col_poc = pocs[0];
  for (int k = 0; k < FCU_MAX_REF; k++) e->c.col_ref_poc[k] = k < n_col ? col_ref_pocs[k] : pocs[0] - 1;
}
/* lambda of an I picture that is not the intra_main default (lowdelay_P: 0.57 * 0.85) */
void fcu_emu_set_lambda(void *h, int qp, double lambda)
{
  EmuChain *e = (EmuChain *)h;
  fcu_frame_params fp; default_frame_params(fp, qp); fp.slice_ctus = e->c.p.slice_ctus; fp.lambda = lambda;
  fill_params(e->c.p, e->c.p.width, e->c.p.height, fp);
}
void fcu_emu_get_state_full(void *h, uint8_t *ctx, uint64_t *frac) { EmuChain *e = (EmuChain *)h; memcpy(ctx, e->c.state.ctx, NCTX); *frac = e->c.state.frac; }
void fcu_emu_destroy(void *h) { EmuChain *e = (EmuChain *)h; free(e->g); delete e; }
void fcu_emu_compress_ctu(void *h, int a) { EmuChain *e = (EmuChain *)h; compress_ctu(&e->c, e->g, a); e->c.next_ctu = a + 1; }
void fcu_emu_get_state(void *h, uint8_t *ctx, uint64_t *frac) { EmuChain *e = (EmuChain *)h; memcpy(ctx, e->c.state.ctx, NCTX_INTRA); *frac = e->c.state.frac; }
void fcu_emu_set_decision(void *h, int state, const uint8_t *sw_skip, const uint8_t *sw_term, int depth_exception, const int16_t *obf)
{
  Chain &c = ((EmuChain *)h)->c;
  c.dec_state = state; c.depth_exception = depth_exception; c.obf = obf; c.obf_stride = c.p.width / 4;
  for (int d = 0; d < 4; d++) { c.sw_skip[d] = sw_skip[d]; c.sw_term[d] = sw_term[d]; }
  memset(c.ver, 0, sizeof(c.ver));
}
void fcu_emu_set_col(void *h, const fcu_ctu_out *col) { ((EmuChain *)h)->c.col = col; ((EmuChain *)h)->c.p.tmvp = col != nullptr; }
void fcu_emu_set_amp(void *h, int amp) { ((EmuChain *)h)->c.p.amp = amp != 0; }
void fcu_emu_set_cabac_b(void *h, int on) { ((EmuChain *)h)->c.p.cabac_b_table = on != 0; }
void fcu_emu_set_rdoq(void *h, int rdoq, int rdoq_ts) { ((EmuChain *)h)->c.p.rdoq = rdoq; ((EmuChain *)h)->c.p.rdoq_ts = rdoq_ts; }
void fcu_emu_set_pu_trace(void *h, fcu_pu_trace *buf) { ((EmuChain *)h)->c.pu_trace = buf; }
void fcu_emu_get_verify(void *h, double *out24) { memcpy(out24, ((EmuChain *)h)->c.ver, sizeof(double) * 24); }
unsigned long long fcu_emu_tu_trials(void *h) { return ((EmuChain *)h)->c.n_tu_trials; }
int fcu_emu_sizes(int which) { return which == 0 ? (int)sizeof(Scratch) : which == 1 ? (int)sizeof(Shared) : (int)sizeof(Chain); }
}
