/*
 * fcu_kernels.hip -- gfx950 kernel entry + the C ABI of libfcu.so (include/fcu.h).
 *
 * Launch geometry: one 64-thread workgroup (= one wavefront) per chain, grid = number of
 * chains advanced by the call.  Chains are dealt round-robin over the 8 XCDs by the
 * dispatcher; a chain's scratch (fcu::Scratch, ~0.7 MB) is touched only by its own wave,
 * so it stays in that XCD's L2 / the Infinity Cache with no cross-XCD traffic.  The hot CABAC
 * coders, reference samples and per-PU mailboxes live in LDS (fcu::Shared + fcu::HotTables, ~10 KB per chain).
 *
 * There is no CPU fallback: every entry point returns FCU_ERR_NO_DEVICE without a GPU.
 */
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "fcu_host.h"
#include "fcu_obf.h"
#include "fcu_deblock.h"
#include "fcu_sao.h"

using namespace fcu;

/* the chain loop lives in a callable so that the kernel body itself keeps no state across the call
 * (its SGPR spills would otherwise cost two extra VGPRs and the fourth wave per SIMD) */
__device__ __noinline__ static void run_chain(Chain *C, Scratch *G, int ctus)
{
  load_hot_tables();
  for (int k = 0; k < ctus; k++) {
    const int a = C->next_ctu;
    if (a >= C->end_ctu || C->out == nullptr) break;        /* every wave reaches this exit */
    compress_ctu(C, G, a);
    FCU_SERIAL { C->next_ctu = a + 1; }
  }
}

__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(FCU_WAVES_PER_EU, FCU_WAVES_PER_EU)))
fcu_ctu_engine(Chain *chains, Scratch *scratch, int first, int ctus)
{
  run_chain(&chains[first + blockIdx.x], &scratch[first + blockIdx.x], ctus);
}

/* Reference picture padding (TComPicYuv::extendPicBorder): every thread writes 16 consecutive bytes of one row of a
 * padded plane; the source coordinate is clamped into the picture.  HBM-bound: reads W*H*1.5, writes (W+160)*(H+160)*1.5/.. */
__global__ void __launch_bounds__(256) fcu_pad_plane(const uint8_t *src, int w, int h, uint8_t *dst, int margin)
{
  const int pw = w + 2 * margin, ph = h + 2 * margin, chunks = (pw + 15) >> 4;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= chunks * ph) return;
  const int y = t / chunks, x0 = (t - y * chunks) << 4;
  int sy = y - margin; sy = sy < 0 ? 0 : (sy >= h ? h - 1 : sy);
  const uint8_t *row = src + (size_t)sy * w;
  uint8_t v[16];
  const int sx0 = x0 - margin;
  if (sx0 >= 0 && sx0 + 16 <= w) __builtin_memcpy(v, row + sx0, 16);
  else for (int k = 0; k < 16; k++) { int sx = sx0 + k; sx = sx < 0 ? 0 : (sx >= w ? w - 1 : sx); v[k] = row[sx]; }
  uint8_t *o = dst + (size_t)y * pw + x0;
  if (x0 + 16 <= pw) __builtin_memcpy(o, v, 16); else for (int k = 0; x0 + k < pw; k++) o[k] = v[k];
}

/* ---------------------------------------------------------------------------------------- */
struct fcu_ctx {
  fcu_seq_params sp;
  int n_ctu;
  Chain *d_chains; Scratch *d_scratch;
  std::vector<Chain> h_chains;
  std::vector<int> h_pos;
  std::vector<hipEvent_t> ev;      /* start/stop pairs of launches not yet harvested (bounded, see harvest_events) */
  double ms_acc; int launches;
  /* persistent scratch of fcu_obf_prepass (grown on demand, freed by fcu_destroy) */
  unsigned *d_hist; size_t hist_cap; int *d_thr; size_t thr_cap;
  /* persistent scratch of fcu_sao: picture descriptors, copy of the deblocked planes, statistics, candidates, reconstructed parameters, off counters */
  void *d_sao; size_t sao_cap;
};

/* Launch timing keeps two events per launch until they are read.  A long-running caller that never asks for
 * fcu_kernel_ms must not grow that list without bound: past FCU_MAX_PENDING_EVENTS pairs the oldest are folded
 * into the accumulator (they have long completed; hipEventSynchronize on them returns at once) and destroyed. */
enum { FCU_MAX_PENDING_EVENTS = 64 };
static void harvest_events(fcu_ctx *c, size_t keep_pairs)
{
  while (c->ev.size() > 2 * keep_pairs) {
    float ms = 0.f;
    hipEventSynchronize(c->ev[1]);
    if (hipEventElapsedTime(&ms, c->ev[0], c->ev[1]) == hipSuccess) { c->ms_acc += ms; c->launches++; }
    hipEventDestroy(c->ev[0]); hipEventDestroy(c->ev[1]);
    c->ev.erase(c->ev.begin(), c->ev.begin() + 2);
  }
}

/* one buffer per host thread: a failure text never races with another thread's call */
static thread_local char g_err[256] = "";
static int fail(int code, const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg); return code; }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { snprintf(g_err, sizeof(g_err), "%s: %s", #x, hipGetErrorString(e_)); return FCU_ERR_HIP; } } while (0)

extern "C" {

const char *fcu_last_error(void) { return g_err; }

/* compiler + flags this library was built with (recorded by the build recipe in __graft_entry__.py).  The engine's
 * correctness depends on `-mllvm -amdgpu-remove-redundant-endcf=false`; tests/test_cabi.py asserts it. */
#ifndef FCU_BUILD_FLAGS
#define FCU_BUILD_FLAGS "unrecorded"
#endif
#define FCU_STR2(x) #x
#define FCU_STR(x) FCU_STR2(x)
int fcu_abi_sizeof(int which)
{
  switch (which) {
  case FCU_ABI_CTU_OUT: return (int)sizeof(fcu_ctu_out);
  case FCU_ABI_SEQ_PARAMS: return (int)sizeof(fcu_seq_params);
  case FCU_ABI_FRAME_PARAMS: return (int)sizeof(fcu_frame_params);
  case FCU_ABI_DECISION_PARAMS: return (int)sizeof(fcu_decision_params);
  case FCU_ABI_VERIFY_COUNTS: return (int)sizeof(fcu_verify_counts);
  case FCU_ABI_SAO_CTU: return (int)sizeof(fcu_sao_ctu);
  case FCU_ABI_SAO_PARAMS: return (int)sizeof(fcu_sao_params);
  case FCU_ABI_PU_TRACE: return (int)sizeof(fcu_pu_trace);
  default: return -1;
  }
}
const char *fcu_build_info(void)
{
  return "hipcc clang " __clang_version__ " HIP " FCU_STR(HIP_VERSION_MAJOR) "." FCU_STR(HIP_VERSION_MINOR) "." FCU_STR(HIP_VERSION_PATCH)
         " waves_per_eu=" FCU_STR(FCU_WAVES_PER_EU) " flags: " FCU_BUILD_FLAGS;
}

void fcu_default_frame_params(fcu_frame_params *fp, int qp) { default_frame_params(*fp, qp); }

int fcu_create(const fcu_seq_params *sp, fcu_ctx **out)
{
  if (!sp || !out || sp->width <= 0 || sp->height <= 0 || (sp->width & 7) || (sp->height & 7) || sp->max_chains <= 0) return fail(FCU_ERR_ARG, "bad sequence parameters");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || sp->device >= ndev) return fail(FCU_ERR_NO_DEVICE, "no HIP device: libfcu has no CPU fallback");
  HIPCHK(hipSetDevice(sp->device));
  fcu_ctx *c = new fcu_ctx();
  c->sp = *sp; c->n_ctu = ((sp->width + 63) / 64) * ((sp->height + 63) / 64);
  c->ms_acc = 0; c->launches = 0;
  c->d_chains = nullptr; c->d_scratch = nullptr; c->d_hist = nullptr; c->hist_cap = 0; c->d_thr = nullptr; c->thr_cap = 0; c->d_sao = nullptr; c->sao_cap = 0;
  struct Guard { fcu_ctx *c; ~Guard() { if (c) { hipFree(c->d_chains); hipFree(c->d_scratch); delete c; } } } guard{ c };   /* frees on every early return */
  HIPCHK(hipMalloc((void **)&c->d_chains, sizeof(Chain) * (size_t)sp->max_chains));
  HIPCHK(hipMalloc((void **)&c->d_scratch, sizeof(Scratch) * (size_t)sp->max_chains));
  HIPCHK(hipMemset(c->d_chains, 0, sizeof(Chain) * (size_t)sp->max_chains));
  guard.c = nullptr;
  c->h_chains.resize((size_t)sp->max_chains);
  memset(c->h_chains.data(), 0, sizeof(Chain) * (size_t)sp->max_chains);
  c->h_pos.assign((size_t)sp->max_chains, 0);
  *out = c;
  return FCU_OK;
}

void fcu_destroy(fcu_ctx *c)
{
  if (!c) return;
  hipSetDevice(c->sp.device);
  hipDeviceSynchronize();
  for (hipEvent_t e : c->ev) hipEventDestroy(e);
  hipFree(c->d_chains); hipFree(c->d_scratch); hipFree(c->d_hist); hipFree(c->d_thr); hipFree(c->d_sao);
  delete c;
}

int fcu_num_ctus(const fcu_ctx *c) { return c ? c->n_ctu : 0; }

int fcu_chain_begin(fcu_ctx *c, int chain, const fcu_frame_params *fp,
                    const uint8_t *oy, const uint8_t *ou, const uint8_t *ov, uint8_t *ry, uint8_t *ru, uint8_t *rv, fcu_ctu_out *dev_out)
{
  if (!c || !fp || chain < 0 || chain >= c->sp.max_chains || !oy || !ou || !ov || !ry || !ru || !rv || !dev_out) return fail(FCU_ERR_ARG, "fcu_chain_begin: bad argument");
  if (fp->qp < 0 || fp->qp > 51 || fp->slice_ctus < 0) return fail(FCU_ERR_ARG, "fcu_chain_begin: QP / slice_ctus out of range");
  if (fp->slice_type != FCU_SLICE_I && fp->slice_type != FCU_SLICE_P) return fail(FCU_ERR_ARG, "fcu_chain_begin: unknown slice type");
  if (fp->slice_type == FCU_SLICE_P && (!(fp->lambda > 0.0) || fp->search_range < 1 || fp->search_range > 64)) return fail(FCU_ERR_ARG, "fcu_chain_begin: a P slice needs its lambda (fcu_ldp_slice) and 1 <= search_range <= 64");
  HIPCHK(hipSetDevice(c->sp.device));
  Chain &h = c->h_chains[(size_t)chain];
  memset(&h, 0, sizeof(h));
  fill_params(h.p, c->sp.width, c->sp.height, *fp);
  h.org[0] = oy; h.org[1] = ou; h.org[2] = ov; h.rec[0] = ry; h.rec[1] = ru; h.rec[2] = rv;
  h.stride[0] = c->sp.width; h.stride[1] = h.stride[2] = c->sp.width / 2;
  h.out = dev_out;
  h.w_ctu = (c->sp.width + 63) / 64; h.h_ctu = (c->sp.height + 63) / 64; h.n_ctu = h.w_ctu * h.h_ctu;
  h.next_ctu = 0; h.end_ctu = h.n_ctu;
  c->h_pos[(size_t)chain] = 0;
  HIPCHK(hipMemcpy(&c->d_chains[chain], &h, sizeof(Chain), hipMemcpyHostToDevice));
  return FCU_OK;
}

void fcu_ldp_slice(fcu_frame_params *fp, int base_qp, int poc) { if (fp) ldp_slice(*fp, base_qp, poc); }

void fcu_pad_sizes(const fcu_ctx *c, size_t *out3)
{
  if (!c || !out3) return;
  out3[0] = (size_t)(c->sp.width + 2 * FCU_REF_MARGIN) * (size_t)(c->sp.height + 2 * FCU_REF_MARGIN);
  out3[1] = out3[2] = (size_t)(c->sp.width / 2 + FCU_REF_MARGIN) * (size_t)(c->sp.height / 2 + FCU_REF_MARGIN);
}
int fcu_ldp_layer(int poc) { static const int layer[4] = { 0, 2, 1, 2 }; return poc < 0 ? 0 : layer[poc & 3]; }

int fcu_pad_reference(fcu_ctx *c, const uint8_t *dy, const uint8_t *du, const uint8_t *dv, uint8_t *py, uint8_t *pu, uint8_t *pv, void *hip_stream)
{
  if (!c || !dy || !du || !dv || !py || !pu || !pv) return fail(FCU_ERR_ARG, "fcu_pad_reference: bad argument");
  HIPCHK(hipSetDevice(c->sp.device));
  hipStream_t st = (hipStream_t)hip_stream;
  const uint8_t *src[3] = { dy, du, dv }; uint8_t *dst[3] = { py, pu, pv };
  for (int k = 0; k < 3; k++) {
    const int w = k ? c->sp.width / 2 : c->sp.width, h = k ? c->sp.height / 2 : c->sp.height, m = k ? FCU_REF_MARGIN / 2 : FCU_REF_MARGIN;
    const int n = ((w + 2 * m + 15) >> 4) * (h + 2 * m);
    hipLaunchKernelGGL(fcu_pad_plane, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src[k], w, h, dst[k], m);
    HIPCHK(hipGetLastError());
  }
  return FCU_OK;
}

int fcu_chain_set_reference(fcu_ctx *c, int chain, const uint8_t *py, const uint8_t *pu, const uint8_t *pv)
{
  if (!c || chain < 0 || chain >= c->sp.max_chains || !py || !pu || !pv) return fail(FCU_ERR_ARG, "fcu_chain_set_reference: bad argument");
  Chain &h = c->h_chains[(size_t)chain];
  if (h.out == nullptr) return fail(FCU_ERR_STATE, "fcu_chain_set_reference: chain not bound (fcu_chain_begin)");
  HIPCHK(hipSetDevice(c->sp.device));
  const int m = FCU_REF_MARGIN, sy = c->sp.width + 2 * m, sc = c->sp.width / 2 + m;
  h.ref_stride[0] = sy; h.ref_stride[1] = h.ref_stride[2] = sc;
  h.ref[0] = py + (size_t)m * sy + m; h.ref[1] = pu + (size_t)(m / 2) * sc + m / 2; h.ref[2] = pv + (size_t)(m / 2) * sc + m / 2;
  static_assert(offsetof(Chain, ref_stride) == offsetof(Chain, ref) + 3 * sizeof(void *), "ref / ref_stride are adjacent");
  /* this entry point: one reference picture at POC distance 1 (no vector is ever scaled), list 0 = { this picture } */
  for (int k = 0; k < 3; k++) h.refs[0][k] = h.ref[k];
  h.n_ref = 1; h.poc = 1; h.ref_poc[0] = 0; h.col_poc = 0; h.col_ref_poc[0] = -1;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy((char *)&c->d_chains[chain] + offsetof(Chain, ref), &h.ref[0], 3 * sizeof(void *) + 3 * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy((char *)&c->d_chains[chain] + offsetof(Chain, refs), (const char *)&h + offsetof(Chain, refs), offsetof(Chain, int_mv_r) - offsetof(Chain, refs), hipMemcpyHostToDevice));
  return FCU_OK;
}

int fcu_chain_set_references(fcu_ctx *c, int chain, int n_ref, const uint8_t *const *dev_pad_planes, const int *ref_pocs, int cur_poc)
{
  if (!c || chain < 0 || chain >= c->sp.max_chains || n_ref < 1 || n_ref > FCU_MAX_REF || !dev_pad_planes || !ref_pocs) return fail(FCU_ERR_ARG, "fcu_chain_set_references: bad argument");
  for (int k = 0; k < 3 * n_ref; k++) if (!dev_pad_planes[k]) return fail(FCU_ERR_ARG, "fcu_chain_set_references: null plane");
  for (int a = 0; a < n_ref; a++) { if (ref_pocs[a] == cur_poc) return fail(FCU_ERR_ARG, "fcu_chain_set_references: a reference picture cannot have the current POC");
    for (int b = 0; b < a; b++) if (ref_pocs[a] == ref_pocs[b]) return fail(FCU_ERR_ARG, "fcu_chain_set_references: the same picture twice in the list"); }
  int rc = fcu_chain_set_reference(c, chain, dev_pad_planes[0], dev_pad_planes[1], dev_pad_planes[2]);
  if (rc != FCU_OK) return rc;
  Chain &h = c->h_chains[(size_t)chain];
  const int m = FCU_REF_MARGIN, sy = c->sp.width + 2 * m, sc = c->sp.width / 2 + m;
  for (int r = 0; r < n_ref; r++) {
    h.refs[r][0] = dev_pad_planes[3 * r] + (size_t)m * sy + m;
    h.refs[r][1] = dev_pad_planes[3 * r + 1] + (size_t)(m / 2) * sc + m / 2; h.refs[r][2] = dev_pad_planes[3 * r + 2] + (size_t)(m / 2) * sc + m / 2;
    h.ref_poc[r] = ref_pocs[r];
  }
  h.n_ref = n_ref; h.poc = cur_poc; h.col_poc = ref_pocs[0]; h.col_ref_poc[0] = ref_pocs[0] - 1;
  HIPCHK(hipMemcpy((char *)&c->d_chains[chain] + offsetof(Chain, refs), (const char *)&h + offsetof(Chain, refs), offsetof(Chain, int_mv_r) - offsetof(Chain, refs), hipMemcpyHostToDevice));
  return FCU_OK;
}

int fcu_chain_get_search_state(fcu_ctx *c, int chain, int32_t *xy)
{
  if (!c || chain < 0 || chain >= c->sp.max_chains || !xy) return fail(FCU_ERR_ARG, "fcu_chain_get_search_state: bad argument");
  static_assert(sizeof(((Chain *)0)->int_mv_r) == 2 * FCU_MAX_REF * sizeof(int32_t), "search state layout");
  HIPCHK(hipSetDevice(c->sp.device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(xy, (const char *)&c->d_chains[chain] + offsetof(Chain, int_mv_r), 2 * FCU_MAX_REF * sizeof(int32_t), hipMemcpyDeviceToHost));
  return FCU_OK;
}
int fcu_chain_set_search_state(fcu_ctx *c, int chain, const int32_t *xy)
{
  if (!c || chain < 0 || chain >= c->sp.max_chains || !xy) return fail(FCU_ERR_ARG, "fcu_chain_set_search_state: bad argument");
  if (c->h_chains[(size_t)chain].out == nullptr) return fail(FCU_ERR_STATE, "fcu_chain_set_search_state: chain not bound (fcu_chain_begin)");
  HIPCHK(hipSetDevice(c->sp.device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy((char *)&c->d_chains[chain] + offsetof(Chain, int_mv_r), xy, 2 * FCU_MAX_REF * sizeof(int32_t), hipMemcpyHostToDevice));
  return FCU_OK;
}

int fcu_chain_set_collocated_pocs(fcu_ctx *c, int chain, int col_poc, const int *col_ref_pocs, int n)
{
  if (!c || chain < 0 || chain >= c->sp.max_chains || !col_ref_pocs || n < 1 || n > FCU_MAX_REF) return fail(FCU_ERR_ARG, "fcu_chain_set_collocated_pocs: bad argument");
  Chain &h = c->h_chains[(size_t)chain];
  if (h.out == nullptr) return fail(FCU_ERR_STATE, "fcu_chain_set_collocated_pocs: chain not bound (fcu_chain_begin)");
  HIPCHK(hipSetDevice(c->sp.device));
  h.col_poc = col_poc;
  for (int k = 0; k < FCU_MAX_REF; k++) { h.col_ref_poc[k] = k < n ? col_ref_pocs[k] : col_poc - 1; if (h.col_ref_poc[k] == col_poc) return fail(FCU_ERR_ARG, "fcu_chain_set_collocated_pocs: a reference of the collocated picture has its own POC"); }
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy((char *)&c->d_chains[chain] + offsetof(Chain, refs), (const char *)&h + offsetof(Chain, refs), offsetof(Chain, int_mv_r) - offsetof(Chain, refs), hipMemcpyHostToDevice));
  return FCU_OK;
}

int fcu_chain_set_range(fcu_ctx *c, int chain, int first_ctu, int n_ctus)
{
  if (!c || chain < 0 || chain >= c->sp.max_chains) return fail(FCU_ERR_ARG, "fcu_chain_set_range: bad argument");
  Chain &h = c->h_chains[(size_t)chain];
  if (h.out == nullptr) return fail(FCU_ERR_STATE, "fcu_chain_set_range: chain not bound (fcu_chain_begin)");
  const int sl = h.p.slice_ctus;
  if (first_ctu < 0 || n_ctus <= 0 || first_ctu + n_ctus > h.n_ctu) return fail(FCU_ERR_ARG, "fcu_chain_set_range: range outside the frame");
  /* a chain may only start where the reference resets its entropy coder and cuts the neighbourhood: at a slice start */
  if (first_ctu != 0 && (sl <= 0 || first_ctu % sl != 0)) return fail(FCU_ERR_ARG, "fcu_chain_set_range: a chain must start at a slice boundary");
  if (first_ctu + n_ctus != h.n_ctu && (sl <= 0 || (first_ctu + n_ctus) % sl != 0)) return fail(FCU_ERR_ARG, "fcu_chain_set_range: a chain must end at a slice boundary");
  HIPCHK(hipSetDevice(c->sp.device));
  h.next_ctu = first_ctu; h.end_ctu = first_ctu + n_ctus;
  c->h_pos[(size_t)chain] = first_ctu;
  /* only the range: the chain's coder state, verification counters and trial count on the device stay as they are */
  static_assert(offsetof(Chain, end_ctu) == offsetof(Chain, next_ctu) + sizeof(int), "next_ctu / end_ctu are adjacent");
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy((char *)&c->d_chains[chain] + offsetof(Chain, next_ctu), &h.next_ctu, 2 * sizeof(int), hipMemcpyHostToDevice));
  return FCU_OK;
}

int fcu_compress_chains(fcu_ctx *c, int first, int n, int ctus, void *hip_stream)
{
  if (!c || first < 0 || n <= 0 || first + n > c->sp.max_chains || ctus <= 0) return fail(FCU_ERR_ARG, "fcu_compress_chains: bad range");
  for (int i = first; i < first + n; i++) {
    if (c->h_chains[(size_t)i].out == nullptr) return fail(FCU_ERR_STATE, "fcu_compress_chains: chain not bound (fcu_chain_begin)");
    if (c->h_chains[(size_t)i].p.slice_type == SLICE_P && c->h_chains[(size_t)i].ref[0] == nullptr) return fail(FCU_ERR_STATE, "fcu_compress_chains: P chain without reference picture (fcu_chain_set_reference)");
  }
  HIPCHK(hipSetDevice(c->sp.device));
  hipStream_t st = (hipStream_t)hip_stream;
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
  HIPCHK(hipEventRecord(e0, st));
  hipLaunchKernelGGL(fcu_ctu_engine, dim3((unsigned)n), dim3(64), 0, st, c->d_chains, c->d_scratch, first, ctus);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(e1, st));
  c->ev.push_back(e0); c->ev.push_back(e1);
  harvest_events(c, FCU_MAX_PENDING_EVENTS);
  for (int i = first; i < first + n; i++) { int &p = c->h_pos[(size_t)i]; p += ctus; if (p > c->h_chains[(size_t)i].end_ctu) p = c->h_chains[(size_t)i].end_ctu; }
  return FCU_OK;
}

int fcu_sync(fcu_ctx *c)
{
  if (!c) return fail(FCU_ERR_ARG, "null ctx");
  HIPCHK(hipSetDevice(c->sp.device));
  HIPCHK(hipDeviceSynchronize());
  return FCU_OK;
}

double fcu_kernel_ms(fcu_ctx *c, int *launches)
{
  if (!c) return 0.0;
  hipSetDevice(c->sp.device);
  hipDeviceSynchronize();
  harvest_events(c, 0);
  const int n = c->launches; const double avg = n ? c->ms_acc / n : 0.0;
  if (launches) *launches = n;
  c->ms_acc = 0; c->launches = 0;
  return avg;
}

int fcu_chain_position(fcu_ctx *c, int chain)
{
  if (!c || chain < 0 || chain >= c->sp.max_chains) return -1;
  return c->h_pos[(size_t)chain];
}

int fcu_compress_ctu(fcu_ctx *c, int chain, uint32_t ctuRsAddr, fcu_ctu_out *host_out)
{
  if (!c || chain < 0 || chain >= c->sp.max_chains || !host_out) return fail(FCU_ERR_ARG, "fcu_compress_ctu: bad argument");
  if ((int)ctuRsAddr != c->h_pos[(size_t)chain] || (int)ctuRsAddr >= c->h_chains[(size_t)chain].end_ctu) return fail(FCU_ERR_STATE, "fcu_compress_ctu: CTUs of a chain must be decided in raster order");
  int r = fcu_compress_chains(c, chain, 1, 1, nullptr);
  if (r != FCU_OK) return r;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(host_out, c->h_chains[(size_t)chain].out + ctuRsAddr, sizeof(fcu_ctu_out), hipMemcpyDeviceToHost));
  return FCU_OK;
}

/* diagnostic: per-chain section timers (only meaningful in a -DFCU_PROFILE build) and TU-trial count */
int fcu_debug_counters(fcu_ctx *c, int chain, unsigned long long *out17)
{
  if (!c || chain < 0 || chain >= c->sp.max_chains || !out17) return fail(FCU_ERR_ARG, "fcu_debug_counters: bad argument");
  HIPCHK(hipSetDevice(c->sp.device));
  HIPCHK(hipDeviceSynchronize());
  Chain h;
  HIPCHK(hipMemcpy(&h, &c->d_chains[chain], sizeof(Chain), hipMemcpyDeviceToHost));
  for (int i = 0; i < 16; i++) out17[i] = h.prof[i];
  out17[16] = h.n_tu_trials;
  return FCU_OK;
}

int fcu_get_ctx_state(fcu_ctx *c, int chain, uint8_t *ctx160, uint64_t *frac_bits)
{
  if (!c || chain < 0 || chain >= c->sp.max_chains || !ctx160 || !frac_bits) return fail(FCU_ERR_ARG, "fcu_get_ctx_state: bad argument");
  HIPCHK(hipSetDevice(c->sp.device));
  HIPCHK(hipDeviceSynchronize());
  Chain h;
  HIPCHK(hipMemcpy(&h, &c->d_chains[chain], sizeof(Chain), hipMemcpyDeviceToHost));
  memcpy(ctx160, h.state.ctx, NCTX_INTRA);
  *frac_bits = h.state.frac;
  return FCU_OK;
}

int fcu_get_ctx_state_full(fcu_ctx *c, int chain, uint8_t *ctx176, uint64_t *frac_bits)
{
  if (!c || chain < 0 || chain >= c->sp.max_chains || !ctx176 || !frac_bits) return fail(FCU_ERR_ARG, "fcu_get_ctx_state_full: bad argument");
  HIPCHK(hipSetDevice(c->sp.device));
  HIPCHK(hipDeviceSynchronize());
  Chain h;
  HIPCHK(hipMemcpy(&h, &c->d_chains[chain], sizeof(Chain), hipMemcpyDeviceToHost));
  memcpy(ctx176, h.state.ctx, NCTX);
  *frac_bits = h.state.frac;
  return FCU_OK;
}

int fcu_chain_set_decision(fcu_ctx *c, int chain, const fcu_decision_params *dp)
{
  if (!c || !dp || chain < 0 || chain >= c->sp.max_chains) return fail(FCU_ERR_ARG, "fcu_chain_set_decision: bad argument");
  if (dp->state < FCU_TRAINING || dp->state > FCU_TESTING) return fail(FCU_ERR_ARG, "fcu_chain_set_decision: unknown state");
  if (dp->state != FCU_TRAINING && !dp->dev_obf) return fail(FCU_ERR_ARG, "fcu_chain_set_decision: Verifying / Testing need the frame's OBF map (fcu_obf_prepass)");
  Chain &h = c->h_chains[(size_t)chain];
  if (h.out == nullptr) return fail(FCU_ERR_STATE, "fcu_chain_set_decision: chain not bound (fcu_chain_begin)");
  HIPCHK(hipSetDevice(c->sp.device));
  h.dec_state = dp->state; h.depth_exception = dp->depth_exception != 0; h.obf = dp->dev_obf; h.obf_stride = c->sp.width / 4;
  for (int d = 0; d < 4; d++) { h.sw_skip[d] = dp->sw_skip2nx2n[d] != 0; h.sw_term[d] = dp->sw_terminate[d] != 0; }
  memset(h.ver, 0, sizeof(h.ver));
  /* only the decision block of the descriptor: the chain's position and context state on the device stay as they are */
  const size_t off = offsetof(Chain, dec_state);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy((char *)&c->d_chains[chain] + off, (const char *)&h + off, sizeof(Chain) - off, hipMemcpyHostToDevice));
  return FCU_OK;
}

int fcu_chain_set_collocated(fcu_ctx *c, int chain, const fcu_ctu_out *dev_col_out)
{
  if (!c || chain < 0 || chain >= c->sp.max_chains) return fail(FCU_ERR_ARG, "fcu_chain_set_collocated: bad argument");
  Chain &h = c->h_chains[(size_t)chain];
  if (h.out == nullptr) return fail(FCU_ERR_STATE, "fcu_chain_set_collocated: chain not bound (fcu_chain_begin)");
  if (dev_col_out == h.out) return fail(FCU_ERR_ARG, "fcu_chain_set_collocated: the collocated picture's array is the chain's own output array");
  HIPCHK(hipSetDevice(c->sp.device));
  h.col = dev_col_out;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy((char *)&c->d_chains[chain] + offsetof(Chain, col), &h.col, sizeof(h.col), hipMemcpyHostToDevice));
  return FCU_OK;
}
int fcu_pu_index(int depth, int nxn, int zidx) { return nxn ? 85 + zidx : (depth <= 0 ? 0 : depth == 1 ? 1 + (zidx >> 6) : depth == 2 ? 5 + (zidx >> 4) : 21 + (zidx >> 2)); }
int fcu_chain_set_pu_trace(fcu_ctx *c, int chain, fcu_pu_trace *dev_trace)
{
  if (!c || chain < 0 || chain >= c->sp.max_chains) return fail(FCU_ERR_ARG, "fcu_chain_set_pu_trace: bad argument");
  Chain &h = c->h_chains[(size_t)chain];
  if (h.out == nullptr) return fail(FCU_ERR_STATE, "fcu_chain_set_pu_trace: chain not bound (fcu_chain_begin)");
  HIPCHK(hipSetDevice(c->sp.device));
  h.pu_trace = dev_trace;
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy((char *)&c->d_chains[chain] + offsetof(Chain, pu_trace), &h.pu_trace, sizeof(h.pu_trace), hipMemcpyHostToDevice));
  return FCU_OK;
}

int fcu_get_verify_counts(fcu_ctx *c, int first, int n, fcu_verify_counts *host_sum)
{
  if (!c || !host_sum || first < 0 || n <= 0 || first + n > c->sp.max_chains) return fail(FCU_ERR_ARG, "fcu_get_verify_counts: bad argument");
  HIPCHK(hipSetDevice(c->sp.device));
  HIPCHK(hipDeviceSynchronize());
  std::vector<Chain> h((size_t)n);
  HIPCHK(hipMemcpy(h.data(), &c->d_chains[first], sizeof(Chain) * (size_t)n, hipMemcpyDeviceToHost));
  memset(host_sum, 0, sizeof(*host_sum));
  for (int i = 0; i < n; i++) for (int d = 0; d < 4; d++) for (int k = 0; k < 6; k++) host_sum->n[d][k] += h[(size_t)i].ver[d][k];
  return FCU_OK;
}

void fcu_decision_switch(const fcu_verify_counts *v, const double th_skip[4], const double th_term[4], uint8_t sw_skip[4], uint8_t sw_term[4])
{
  for (int d = 0; d < 4; d++) {
    const double tp = v->n[d][0], fp = v->n[d][1], tn = v->n[d][2], fn = v->n[d][3];
    const double ths = (th_skip && th_skip[d] != 0) ? th_skip[d] : 0.8, tht = (th_term && th_term[d] != 0) ? th_term[d] : 0.8;
    const double ps = (tp + fp == 0) ? 0.0 : tp / (tp + fp);            /* getSkipPrecision, tools_YS.cpp:1251-1258 */
    const double pt = (tn + fn == 0) ? 0.0 : tn / (tn + fn);            /* getTermPrecision, :1259-1266 */
    sw_skip[d] = ps > ths; sw_term[d] = pt > tht;
  }
}

int fcu_frame_state(int poc, int period, int n_training, int n_verifying)
{
  if (period <= 0) return FCU_TRAINING;
  const int r = poc % period;
  return r < n_training ? FCU_TRAINING : (r < n_training + n_verifying ? FCU_VERIFYING : FCU_TESTING);
}

/* diagnostic: resident workgroups (= chains) per CU the runtime grants the engine kernel */
int fcu_chains_per_cu(void)
{
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fcu_ctu_engine, 64, 0) != hipSuccess) return -1;
  return n;
}

/* the host step of the pre-pass on its own: Yc of one frequency from its amplitude histogram (pure host arithmetic, no GPU) */
double fcu_tcm_threshold(const unsigned *hist, int hist_len, int n_samples)
{
  if (!hist || hist_len <= 0 || n_samples <= 0) return 0.0;
  std::vector<unsigned> h((size_t)OBF_HB, 0u);
  for (int i = 0; i < hist_len && i < OBF_HB; i++) h[(size_t)i] = hist[i];
  return tcm_threshold(h.data(), n_samples);
}

/* ---- fork pre-pass: OBF maps of n luma planes ------------------------------------------------------------ */
int fcu_obf_prepass(fcu_ctx *c, int n_frames, const uint8_t *dev_y, int16_t *dev_obf, double *host_yc, float *kernel_ms2, void *hip_stream)
{
  if (!c || n_frames <= 0 || !dev_y || !dev_obf) return fail(FCU_ERR_ARG, "fcu_obf_prepass: bad argument");
  HIPCHK(hipSetDevice(c->sp.device));
  hipStream_t st = (hipStream_t)hip_stream;
  const int w = c->sp.width, h = c->sp.height, nblk = (w / 4) * (h / 4);
  const size_t frame_bytes = (size_t)w * h, hist_n = (size_t)n_frames * 15 * OBF_HB;
  /* histogram / threshold buffers belong to the context and only grow: no allocation in the steady state */
  if (c->hist_cap < hist_n) { hipFree(c->d_hist); c->d_hist = nullptr; c->hist_cap = 0; HIPCHK(hipMalloc((void **)&c->d_hist, hist_n * sizeof(unsigned))); c->hist_cap = hist_n; }
  if (c->thr_cap < (size_t)n_frames * 16) { hipFree(c->d_thr); c->d_thr = nullptr; c->thr_cap = 0; HIPCHK(hipMalloc((void **)&c->d_thr, (size_t)n_frames * 16 * sizeof(int))); c->thr_cap = (size_t)n_frames * 16; }
  unsigned *d_hist = c->d_hist; int *d_thr = c->d_thr;
  HIPCHK(hipMemsetAsync(d_hist, 0, hist_n * sizeof(unsigned), st));
  hipEvent_t e[4] = { nullptr, nullptr, nullptr, nullptr };
  struct EventGuard { hipEvent_t *e; ~EventGuard() { for (int i = 0; i < 4; i++) if (e[i]) hipEventDestroy(e[i]); } } guard{ e };
  for (int i = 0; i < 4; i++) HIPCHK(hipEventCreate(&e[i]));
  const int ngrp = (((w / 4) + 3) / 4) * (h / 4);               /* groups of four blocks along a row */
  const dim3 grid((unsigned)((ngrp + OBF_THREADS * OBF_GROUPS_PER_THREAD - 1) / (OBF_THREADS * OBF_GROUPS_PER_THREAD)), (unsigned)n_frames);
  HIPCHK(hipEventRecord(e[0], st));
  hipLaunchKernelGGL(obf_hist, grid, dim3(OBF_THREADS), 0, st, dev_y, w, h, frame_bytes, d_hist);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(e[1], st));
  std::vector<unsigned> hist(hist_n);
  HIPCHK(hipMemcpyAsync(hist.data(), d_hist, hist_n * sizeof(unsigned), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  /* threshold fit on the host (doubles + libm exp/log, as the reference), frames spread over a few threads */
  std::vector<double> yc((size_t)n_frames * 16, 0.0);
  std::vector<int> thr((size_t)n_frames * 16, 0);
  {
    const int nt = n_frames < 8 ? n_frames : 8;
    std::vector<std::thread> pool;
    for (int t = 0; t < nt; t++)
      pool.emplace_back([&, t]() {
        for (int f = t; f < n_frames; f += nt)
          for (int x = 1; x < 16; x++) {
            const double v = tcm_threshold(&hist[((size_t)f * 15 + (x - 1)) * OBF_HB], nblk);
            yc[(size_t)f * 16 + x] = v; thr[(size_t)f * 16 + x] = (int)(v * 8.0);
          }
      });
    for (auto &th : pool) th.join();
  }
  for (size_t k = 0; k < (size_t)n_frames * 15; k++)           /* the clamp bin is unreachable for 8-bit sources (|coef/8| <= 4080) */
    if (hist[k * OBF_HB + OBF_HB - 1]) { return fail(FCU_ERR_ARG, "fcu_obf_prepass: amplitude beyond the histogram (not an 8-bit plane?)"); }
  HIPCHK(hipMemcpyAsync(d_thr, thr.data(), thr.size() * sizeof(int), hipMemcpyHostToDevice, st));
  HIPCHK(hipEventRecord(e[2], st));
  hipLaunchKernelGGL(obf_count, grid, dim3(OBF_THREADS), 0, st, dev_y, w, h, frame_bytes, d_thr, dev_obf);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(e[3], st));
  HIPCHK(hipStreamSynchronize(st));
  if (kernel_ms2) { hipEventElapsedTime(&kernel_ms2[0], e[0], e[1]); hipEventElapsedTime(&kernel_ms2[1], e[2], e[3]); }
  if (host_yc) memcpy(host_yc, yc.data(), yc.size() * sizeof(double));
  return FCU_OK;
}

int fcu_deblock(fcu_ctx *c, const fcu_ctu_out *dev_out, uint8_t *dev_rec_y, uint8_t *dev_rec_u, uint8_t *dev_rec_v,
                int beta_offset_div2, int tc_offset_div2, float *kernel_ms2, void *hip_stream)
{
  if (!c || !dev_out || !dev_rec_y || !dev_rec_u || !dev_rec_v) return fail(FCU_ERR_ARG, "fcu_deblock: bad argument");
  if (beta_offset_div2 < -6 || beta_offset_div2 > 6 || tc_offset_div2 < -6 || tc_offset_div2 > 6) return fail(FCU_ERR_ARG, "fcu_deblock: offsets are limited to [-6, 6]");
  HIPCHK(hipSetDevice(c->sp.device));
  hipStream_t st = (hipStream_t)hip_stream;
  const int w = c->sp.width, h = c->sp.height, w_ctu = (w + 63) / 64;
  const unsigned n0 = (unsigned)((w >> 3) * (h >> 2)), n1 = (unsigned)((w >> 2) * (h >> 3));
  hipEvent_t e[3] = { nullptr, nullptr, nullptr };
  struct EventGuard { hipEvent_t *e; ~EventGuard() { for (int i = 0; i < 3; i++) if (e[i]) hipEventDestroy(e[i]); } } guard{ e };
  if (kernel_ms2) { for (int i = 0; i < 3; i++) HIPCHK(hipEventCreate(&e[i])); HIPCHK(hipEventRecord(e[0], st)); }
  /* all vertical edges of the picture before the first horizontal one (TComLoopFilter.cpp:133-154): stream order */
  hipLaunchKernelGGL(dbk_pass<0>, dim3((n0 + DBK_THREADS - 1) / DBK_THREADS), dim3(DBK_THREADS), 0, st, dev_out, dev_rec_y, dev_rec_u, dev_rec_v, w, h, w_ctu, beta_offset_div2, tc_offset_div2);
  HIPCHK(hipGetLastError());
  if (kernel_ms2) HIPCHK(hipEventRecord(e[1], st));
  hipLaunchKernelGGL(dbk_pass<1>, dim3((n1 + DBK_THREADS - 1) / DBK_THREADS), dim3(DBK_THREADS), 0, st, dev_out, dev_rec_y, dev_rec_u, dev_rec_v, w, h, w_ctu, beta_offset_div2, tc_offset_div2);
  HIPCHK(hipGetLastError());
  if (kernel_ms2) {
    HIPCHK(hipEventRecord(e[2], st));
    HIPCHK(hipStreamSynchronize(st));
    hipEventElapsedTime(&kernel_ms2[0], e[0], e[1]); hipEventElapsedTime(&kernel_ms2[1], e[1], e[2]);
  }
  return FCU_OK;
}

int fcu_sao(fcu_ctx *c, int n_pics, const fcu_sao_params *params, const uint8_t *const *dev_org, uint8_t *const *dev_rec,
            fcu_sao_ctu *dev_coded, int32_t *off_count, float *kernel_ms4, void *hip_stream)
{
  if (!c || n_pics <= 0 || !params || !dev_org || !dev_rec || !dev_coded) return fail(FCU_ERR_ARG, "fcu_sao: bad argument");
  for (int i = 0; i < 3 * n_pics; i++) if (!dev_org[i] || !dev_rec[i]) return fail(FCU_ERR_ARG, "fcu_sao: null plane");
  for (int i = 0; i < n_pics; i++) {
    if (params[i].slice_type != FCU_SLICE_I && params[i].slice_type != FCU_SLICE_P) return fail(FCU_ERR_ARG, "fcu_sao: slice type");
    if (params[i].qp < 0 || params[i].qp > 51 || params[i].slice_ctus < 0) return fail(FCU_ERR_ARG, "fcu_sao: qp / slice_ctus");
    if (!(params[i].lambda[0] > 0) || params[i].lambda[1] < 0 || params[i].lambda[2] < 0) return fail(FCU_ERR_ARG, "fcu_sao: lambda[0] must be positive");
  }
  HIPCHK(hipSetDevice(c->sp.device));
  hipStream_t st = (hipStream_t)hip_stream;
  const int w = c->sp.width, h = c->sp.height, w_ctu = (w + 63) / 64, n_ctu = c->n_ctu;
  if (w_ctu + 1 > SAO_RING) return fail(FCU_ERR_ARG, "fcu_sao: pictures wider than 255 CTUs are not supported (sao_decide's neighbour ring)");
  const size_t plane[3] = { (size_t)w * h, (size_t)(w / 2) * (h / 2), (size_t)(w / 2) * (h / 2) }, pic_bytes = plane[0] + 2 * plane[1];
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  const size_t o_pics = 0, o_src = up(o_pics + sizeof(SaoPic) * n_pics), o_stats = up(o_src + pic_bytes * n_pics),
               o_cand = up(o_stats + sizeof(int32_t) * SAO_STAT_INTS * 3 * (size_t)n_ctu * n_pics),
               o_recon = up(o_cand + sizeof(SaoCand) * 15 * (size_t)n_ctu * n_pics),
               o_off = up(o_recon + sizeof(fcu_sao_ctu) * (size_t)n_ctu * n_pics), total = up(o_off + sizeof(int32_t) * 3 * n_pics);
  if (total > c->sao_cap) {
    if (c->d_sao) { HIPCHK(hipStreamSynchronize(st)); hipFree(c->d_sao); c->d_sao = nullptr; c->sao_cap = 0; }
    HIPCHK(hipMalloc(&c->d_sao, total)); c->sao_cap = total;
  }
  uint8_t *base = (uint8_t *)c->d_sao;
  SaoPic *d_pics = (SaoPic *)(base + o_pics); int32_t *d_stats = (int32_t *)(base + o_stats); SaoCand *d_cand = (SaoCand *)(base + o_cand);
  fcu_sao_ctu *d_recon = (fcu_sao_ctu *)(base + o_recon); int32_t *d_off = (int32_t *)(base + o_off);
  std::vector<SaoPic> hp((size_t)n_pics);
  for (int i = 0; i < n_pics; i++) {
    uint8_t *s = base + o_src + pic_bytes * (size_t)i;
    for (int k = 0; k < 3; k++) {
      hp[i].org[k] = dev_org[3 * i + k]; hp[i].rec[k] = dev_rec[3 * i + k]; hp[i].src[k] = s;
      HIPCHK(hipMemcpyAsync(s, dev_rec[3 * i + k], plane[k], hipMemcpyDeviceToDevice, st));      /* resYuv->copyToPic(srcYuv), :265 */
      s += plane[k];
      hp[i].enabled[k] = params[i].enabled[k] ? 1 : 0;
    }
    fcu_frame_params fp; default_frame_params(fp, params[i].qp); fp.lambda = params[i].lambda[0];
    Params pp; fill_params(pp, w, h, fp);                      /* chroma lambdas = lambda / chroma weight (setUpLambda) unless given */
    for (int k = 0; k < 3; k++) hp[i].lambda[k] = params[i].lambda[k] > 0 ? params[i].lambda[k] : pp.rdoq_lambda[k];
    hp[i].slice_type = params[i].slice_type; hp[i].qp = params[i].qp; hp[i].slice_ctus = params[i].slice_ctus;
  }
  HIPCHK(hipMemcpyAsync(d_pics, hp.data(), sizeof(SaoPic) * n_pics, hipMemcpyHostToDevice, st));
  hipEvent_t e[5] = { nullptr, nullptr, nullptr, nullptr, nullptr };
  struct EventGuard { hipEvent_t *e; ~EventGuard() { for (int i = 0; i < 5; i++) if (e[i]) hipEventDestroy(e[i]); } } guard{ e };
  if (kernel_ms4) { for (int i = 0; i < 5; i++) HIPCHK(hipEventCreate(&e[i])); HIPCHK(hipEventRecord(e[0], st)); }
  hipLaunchKernelGGL(sao_stats, dim3(n_ctu, 3, n_pics), dim3(SAO_THREADS), 0, st, d_pics, d_stats, w, h, w_ctu, n_ctu);
  HIPCHK(hipGetLastError());
  if (kernel_ms4) HIPCHK(hipEventRecord(e[1], st));
  const long long n_cand = (long long)n_pics * n_ctu * 15;
  hipLaunchKernelGGL(sao_cands, dim3((unsigned)((n_cand + SAO_THREADS - 1) / SAO_THREADS)), dim3(SAO_THREADS), 0, st, d_pics, d_stats, d_cand, n_ctu, n_pics);
  HIPCHK(hipGetLastError());
  if (kernel_ms4) HIPCHK(hipEventRecord(e[2], st));
  hipLaunchKernelGGL(sao_decide, dim3(n_pics), dim3(64), 0, st, d_pics, d_stats, d_cand, dev_coded, d_recon, d_off, w_ctu, n_ctu, n_pics);
  HIPCHK(hipGetLastError());
  if (kernel_ms4) HIPCHK(hipEventRecord(e[3], st));
  hipLaunchKernelGGL(sao_apply, dim3(n_ctu, 3, n_pics), dim3(SAO_THREADS), 0, st, d_pics, d_recon, w, h, w_ctu, n_ctu);
  HIPCHK(hipGetLastError());
  if (kernel_ms4) HIPCHK(hipEventRecord(e[4], st));
  if (off_count) HIPCHK(hipMemcpyAsync(off_count, d_off, sizeof(int32_t) * 3 * n_pics, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));                          /* hp (the host descriptors) and off_count are done with */
  if (kernel_ms4) for (int i = 0; i < 4; i++) hipEventElapsedTime(&kernel_ms4[i], e[i], e[i + 1]);
  return FCU_OK;
}

void fcu_sao_enabled(const double rate[3][8], int layer, int32_t enabled[3])
{
  static const double thr[3] = { 0.75, 0.5, 0.5 };           /* SAO_ENCODING_RATE / SAO_ENCODING_RATE_CHROMA, TypeDef.h:201-204 */
  for (int k = 0; k < 3; k++) enabled[k] = (layer > 0 && layer <= 8 && rate[k][layer - 1] > thr[k]) ? 0 : 1;
}
void fcu_sao_update_rate(double rate[3][8], int layer, const int32_t off_count[3], int num_ctus)
{
  if (layer < 0 || layer >= 8 || num_ctus <= 0) return;
  for (int k = 0; k < 3; k++) rate[k][layer] = (double)off_count[k] / (double)num_ctus;
}

} /* extern "C" */
