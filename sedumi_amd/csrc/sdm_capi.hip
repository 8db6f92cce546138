// sdm_capi.hip -- extern "C" entry points of libsedumi_hip.so (include/sedumi_hip.h).
#include "../../include/sedumi_hip.h"
#include "sdm_plan.h"
#include <algorithm>
#include <cstring>
#include <memory>

namespace sdm {
static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
}  // namespace sdm
using namespace sdm;

// every read-back of a plan passes through here after its stream synchronise: a spin inside one of THIS plan's panel
// launches (or merged sweep launches, sdm_solve.hip: merged_wait) that gave up makes the results unusable (the plan is marked not factored by chol_wait_timeouts)
static void check_plan_health(sdm_plan *p) {
  if (chol_wait_timeouts(p)) throw std::runtime_error("blkchol: a workgroup timed out waiting for another one inside a launch (factor panel or merged sweep)");
}

#define SDM_TRY try {
#define SDM_CATCH                                              \
  }                                                            \
  catch (const std::exception &e) { set_error(e.what()); return 1; } \
  catch (...) { set_error("unknown error"); return 1; }        \
  return 0;

extern "C" {

const char *sdm_last_error(void) { return g_err.c_str(); }
#ifdef SDM_EMU
const char *sdm_backend(void) { return "emu"; }
#else
const char *sdm_backend(void) { return "hip-gfx950"; }
#endif
int sdm_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
int sdm_set_device(int dev) {
  SDM_TRY
  SDM_HIP_CHECK(hipSetDevice(dev));
  SDM_CATCH
}

// ------------------------------------------------------------------ plan
sdm_plan *sdm_plan_create(int device, void *stream) {
  try {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { set_error("no HIP device available (libsedumi_hip has no CPU fallback)"); return nullptr; }
    SDM_HIP_CHECK(hipSetDevice(device));
    sdm_plan *p = new sdm_plan();
    p->device = device;
    if (stream) { p->stream = (hipStream_t)stream; p->own_stream = false; }
    else { SDM_HIP_CHECK(hipStreamCreate(&p->stream)); p->own_stream = true; }
    return p;
  } catch (const std::exception &e) { set_error(e.what()); return nullptr; }
}
void sdm_plan_destroy(sdm_plan *p) {
  if (!p) return;
  (void)hipSetDevice(p->device);
  (void)hipStreamSynchronize(p->stream);
  chol_forget_plan(p);
  for (auto g : p->graphs) (void)hipGraphExecDestroy(g);
  for (int i = 0; i < 16; i++) { if (p->ev_begin[i]) (void)hipEventDestroy(p->ev_begin[i]); if (p->ev_end[i]) (void)hipEventDestroy(p->ev_end[i]); }
  if (p->own_stream && p->stream) (void)hipStreamDestroy(p->stream);
  delete p;
}
int sdm_plan_sync(sdm_plan *p) {
  SDM_TRY
  SDM_HIP_CHECK(hipSetDevice(p->device));
  SDM_HIP_CHECK(hipStreamSynchronize(p->stream));
  check_plan_health(p);
  SDM_CATCH
}

int sdm_plan_set_chol(sdm_plan *p, sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const sdm_int *perm,
                      sdm_int nsuper, const sdm_int *xsuper, const sdm_int *ADAjc, const sdm_int *ADAir) {
  SDM_TRY
  SDM_HIP_CHECK(hipSetDevice(p->device));
  p->dense = sdm::DensePlan();            // dense-column tables are sized for the factor they were set with: gone with it
  chol_build(p, m, Ljc, Lir, perm, nsuper, xsuper, ADAjc, ADAir);
  p->ada_jc.assign(ADAjc, ADAjc + m + 1);
  p->ada_ir.assign(ADAir, ADAir + ADAjc[m]);
  SDM_CATCH
}

int sdm_plan_set_ada(sdm_plan *p, sdm_int N, sdm_int m, const sdm_int *Ajc, const sdm_int *Air, const double *Apr,
                     const sdm_int *Ajc_psd, const sdm_cone *K, const sdm_int *qblkstart, const sdm_int *psd_blkstart,
                     const sdm_int *Qjc, const sdm_int *Qir) {
  SDM_TRY
  SDM_HIP_CHECK(hipSetDevice(p->device));
  if (!p->has_chol) throw std::runtime_error("sdm_plan_set_ada: call sdm_plan_set_chol first (it carries the ADA pattern)");
  if (m != p->chol.m) throw std::runtime_error("sdm_plan_set_ada: m mismatch");
  ada_build(p, N, m, Ajc, Air, Apr, Ajc_psd, K->lpN, K->lorN, K->lorNL, K->sdpN, K->rsdpN, K->sdpNL, qblkstart,
            psd_blkstart, Qjc, Qir, p->ada_jc.data(), p->ada_ir.data());
  // host copy of At for the operators of sdm_pcg.hip (their device tables are built on first use)
  p->ada.h_Ajc.assign(Ajc, Ajc + m + 1); p->ada.h_Air.assign(Air, Air + Ajc[m]); p->ada.h_Apr.assign(Apr, Apr + Ajc[m]);
  p->ada.pcg_ready = false; p->ada.aden_n = 0;
  SDM_CATCH
}

static DevBuf<double> *plan_buf(sdm_plan *p, const char *name) {
  std::string s(name);
  if (s == "ada") return &p->ada_val;
  if (s == "absd") return &p->absd;
  if (s == "d") return &p->chol.d;
  if (s == "dsolve") return &p->chol.dsolve;
  if (s == "y") return &p->y;
  if (s == "rhs") return &p->rhs;
  if (s == "lpr") return &p->lpr;
  if (s == "udsqr") return &p->ada.udsqr;
  if (s == "u") { if (p->ada.ufac.n < (size_t)std::max<sdm_int>(p->ada.lenud, 1)) p->ada.ufac.alloc((size_t)std::max<sdm_int>(p->ada.lenud, 1)); return &p->ada.ufac; }
  if (s == "dl") return &p->ada.dl;
  if (s == "ddet") return &p->ada.ddet;
  if (s == "qpr") return &p->ada.qpr;
  if (s == "q1") return &p->ada.q1;
  if (s == "q2") return &p->ada.q2;
  if (s == "xN") { if (p->ada.xN.n == 0) throw std::runtime_error("buffer xN exists after the first sdm_plan_amul / vecsym / psdscale / pcg_init"); return &p->ada.xN; }
  if (s == "psd") { if (p->ada.psd.n == 0) throw std::runtime_error("buffer psd exists after sdm_plan_pcg_init"); return &p->ada.psd; }
  if (s == "ad") return &p->dense.ad;       // dense columns Ad (m x nden, column major; deninfac.m:58-59)
  if (s == "lad") return &p->dense.lad;     // LAD = L \ Ad(perm,:) of the last sdm_plan_deninfac
  if (s == "dden") return &p->dense.dden;   // Ld of the last sdm_plan_deninfac
  // the arenas the multi-GPU layer exchanges slices of (sedumi_amd.dist.SeparatorShardedSolver)
  if (s == "fronts") return &p->chol.fronts;
  if (s == "wvec") return &p->chol.wvec;
  if (s == "xfin") return &p->chol.xfin;
  if (s == "ub") return &p->chol.ub;
  if (s == "panelrec") { if (p->chol.panelrec.n == 0) p->chol.panelrec.alloc((size_t)(4 * sdm::NB + 2 + sdm::NB * sdm::NB)); return &p->chol.panelrec; }
  throw std::runtime_error("unknown plan buffer: " + s);
}
void *sdm_plan_devptr(sdm_plan *p, const char *name, sdm_int *nelem) {
  try {
    DevBuf<double> *b = plan_buf(p, name);
    if (nelem) *nelem = (sdm_int)b->n;
    return b->p;
  } catch (const std::exception &e) { set_error(e.what()); return nullptr; }
}
int sdm_plan_upload(sdm_plan *p, const char *name, const double *src, sdm_int nelem) {
  SDM_TRY
  DevBuf<double> *b = plan_buf(p, name);
  if ((size_t)nelem > b->n) throw std::runtime_error(std::string("upload: too many elements for buffer ") + name);
  SDM_HIP_CHECK(hipMemcpyAsync(b->p, src, (size_t)nelem * sizeof(double), hipMemcpyHostToDevice, p->stream));
  SDM_HIP_CHECK(hipStreamSynchronize(p->stream));
  SDM_CATCH
}
int sdm_plan_download(sdm_plan *p, const char *name, double *dst, sdm_int nelem) {
  SDM_TRY
  DevBuf<double> *b = plan_buf(p, name);
  if (std::string(name) == "lpr") chol_extract(p, p->lpr.p);
  if ((size_t)nelem > b->n) throw std::runtime_error(std::string("download: too many elements for buffer ") + name);
  // small read-backs (the vectors a host-driven loop fetches a hundred times per iteration) through the plan's own pinned buffer: a copy into
  // pageable memory goes through the runtime's staging and took 37 us per call of the product driver, this one 12 (profiles/r08w_*)
  constexpr size_t DL_STAGE_BYTES = 64 * 1024;
  const size_t nbytes = (size_t)nelem * sizeof(double);
  if (nbytes > 0 && nbytes <= DL_STAGE_BYTES && !p->capturing) {
    p->dl_stage.ensure(DL_STAGE_BYTES / sizeof(int));
    SDM_HIP_CHECK(hipMemcpyAsync(p->dl_stage.p, b->p, nbytes, hipMemcpyDeviceToHost, p->stream));
    SDM_HIP_CHECK(hipStreamSynchronize(p->stream));
    memcpy(dst, p->dl_stage.p, nbytes);
  } else {
    SDM_HIP_CHECK(hipMemcpyAsync(dst, b->p, nbytes, hipMemcpyDeviceToHost, p->stream));
    SDM_HIP_CHECK(hipStreamSynchronize(p->stream));
  }
  check_plan_health(p);
  SDM_CATCH
}

int sdm_plan_getada(sdm_plan *p) {
  SDM_TRY
  if (!p->has_ada) throw std::runtime_error("sdm_plan_getada: no ADA data set");
  if (ada_lq_q(p, p->ada_val.p)) { ada_psd(p, p->ada_val.p, nullptr, false, true); return 0; }
  p->ada.zero_defer = true;                  // (a zero LP / Lorentz part may be cleared by the stage-1 launch of ada_psd: ada_zero_flush)
  try {
    ada_lq(p, p->ada_val.p, nullptr, false);
    ada_q(p, p->ada_val.p, nullptr, true);
    ada_psd(p, p->ada_val.p, nullptr, false);
  } catch (...) { p->ada.zero_defer = false; ada_zero_flush(p); throw; }
  p->ada.zero_defer = false;
  ada_zero_flush(p);
  SDM_CATCH
}
int sdm_plan_getdatq(sdm_plan *p) {
  SDM_TRY
  if (!p->has_ada) throw std::runtime_error("sdm_plan_getdatq: no ADA data set");
  ada_datq(p);
  SDM_CATCH
}
int sdm_plan_getada_cols(sdm_plan *p, sdm_int j0, sdm_int j1) {
  SDM_TRY
  if (!p->has_ada) throw std::runtime_error("sdm_plan_getada_cols: no ADA data set");
  if (j0 < 0 || j1 > p->ada.m || j0 > j1) throw std::runtime_error("sdm_plan_getada_cols: column range out of bounds");
  p->ada.col0 = j0; p->ada.col1 = j1;
  try {
    ada_lq(p, p->ada_val.p, nullptr, false);
    ada_q(p, p->ada_val.p, nullptr, true);
    ada_psd(p, p->ada_val.p, nullptr, false);
  } catch (...) { p->ada.col0 = 0; p->ada.col1 = p->ada.m; throw; }
  p->ada.col0 = 0; p->ada.col1 = p->ada.m;
  SDM_CATCH
}
int sdm_plan_copy(sdm_plan *p, const char *name, void *devptr, sdm_int offset, sdm_int nelem, int to_plan) {
  SDM_TRY
  DevBuf<double> *b = plan_buf(p, name);
  if (offset < 0 || nelem < 0 || (size_t)(offset + nelem) > b->n) throw std::runtime_error(std::string("sdm_plan_copy: range outside buffer ") + name);
  if (std::string(name) == "lpr" && !to_plan) chol_extract(p, p->lpr.p);
  if (to_plan) SDM_HIP_CHECK(hipMemcpyAsync(b->p + offset, devptr, (size_t)nelem * sizeof(double), hipMemcpyDeviceToDevice, p->stream));
  else SDM_HIP_CHECK(hipMemcpyAsync(devptr, b->p + offset, (size_t)nelem * sizeof(double), hipMemcpyDeviceToDevice, p->stream));
  SDM_HIP_CHECK(hipStreamSynchronize(p->stream));
  check_plan_health(p);
  SDM_CATCH
}
int sdm_plan_blkchol(sdm_plan *p, const sdm_cholpars *pars, int use_absd) {
  SDM_TRY
  if (!p->has_chol) throw std::runtime_error("sdm_plan_blkchol: no symbolic factor set");
  sdm_cholpars q = {1e-12, 5e2, 1e-20};           // blkchol.c:292-294 defaults
  if (pars) q = *pars;
  if (q.abstol < 0.0) q.abstol = 0.0;              // blkchol.c:303
  p->dense.factored = false;                        // the dense-column factors belong to the previous L, d
  chol_factor(p, q.canceltol, q.maxu, q.abstol, use_absd);
  SDM_CATCH
}
int sdm_plan_blkchol_wait(sdm_plan *p, const sdm_cholpars *pars, int use_absd) {
  SDM_TRY
  for (int attempt = 0; ; attempt++) {
    if (sdm_plan_blkchol(p, pars, use_absd)) throw std::runtime_error(g_err);
    SDM_HIP_CHECK(hipStreamSynchronize(p->stream));
    if (!chol_wait_timeouts(p)) break;                               // (a time-out switches the plan to the launch-per-panel path)
    if (attempt == 1) throw std::runtime_error("blkchol: a workgroup timed out waiting for another one inside a launch (factor panel or merged sweep)");
  }
  SDM_CATCH
}
// ---- the factorisation and the solve level by level (include/sedumi_hip.h: "Separator fronts across GPUs")
int sdm_plan_set_active_supernodes(sdm_plan *p, const int *active, sdm_int nsuper) {
  SDM_TRY
  p->chol.sn_active.assign(active ? active : nullptr, active ? active + nsuper : nullptr);
  for (auto &c : p->chol.sn_active) c = c ? 1 : 0;
  SDM_CATCH
}
int sdm_plan_front_layout(sdm_plan *p, sdm_int *nlevels, sdm_int *level, sdm_int *foff, sdm_int *fsize, sdm_int *woff, sdm_int *ms, sdm_int *first, sdm_int *ns) {
  SDM_TRY
  if (!p->has_chol) throw std::runtime_error("sdm_plan_front_layout: no symbolic factor set");
  const sdm::CholPlan &C = p->chol;
  if (nlevels) *nlevels = C.nlevels;
  for (sdm_int s = 0; s < C.nsuper; s++) {
    if (level) level[s] = C.sn_level[s];
    if (foff) foff[s] = C.sn_foff[s];
    if (fsize) fsize[s] = (sdm_int)C.sn_ld[s] * C.sn_ms[s];
    if (woff) woff[s] = C.sn_woff[s];
    if (ms) ms[s] = C.sn_ms[s];
    if (first) first[s] = C.sn_first[s];
    if (ns) ns[s] = C.sn_ns[s];
  }
  SDM_CATCH
}
int sdm_plan_blkchol_begin(sdm_plan *p, const sdm_cholpars *pars, int use_absd) {
  SDM_TRY
  if (!p->has_chol) throw std::runtime_error("sdm_plan_blkchol_begin: no symbolic factor set");
  sdm_cholpars q = {1e-12, 5e2, 1e-20};
  if (pars) q = *pars;
  if (q.abstol < 0.0) q.abstol = 0.0;
  p->dense.factored = false;
  chol_begin(p, q.canceltol, q.maxu, q.abstol, use_absd);
  SDM_CATCH
}
// ---- one dense front factored block-column-cyclically by several ranks (include/sedumi_hip.h: "One front across GPUs")
int sdm_plan_set_column_owner(sdm_plan *p, int world, int rank, int blk) {
  SDM_TRY
  if (world < 1 || world > 255 || rank < 0 || rank >= world || blk < 1 || blk > 255) throw std::runtime_error("sdm_plan_set_column_owner: need 1 <= world <= 255, 0 <= rank < world, 1 <= blk <= 255");
  p->chol.own = world == 1 ? 0 : (world | rank << 8 | blk << 16);
  SDM_CATCH
}
int sdm_plan_blkchol_panels(sdm_plan *p, sdm_int l0, sdm_int l1, sdm_int pan0, sdm_int pan1) {
  SDM_TRY
  if (!p->has_chol) throw std::runtime_error("sdm_plan_blkchol_panels: no symbolic factor set");
  for (int l = (int)std::max<sdm_int>(l0, 0); l < (int)std::min<sdm_int>(l1, p->chol.nlevels); l++)
    if (p->chol.lev_persist[l] && !p->chol.front_disabled)
      throw std::runtime_error("sdm_plan_blkchol_panels: the level is planned as ONE launch; call sdm_plan_set_one_launch_fronts(plan, 0) before sdm_plan_set_chol");
  chol_levels(p, (int)l0, (int)l1, false, (int)pan0, (int)std::min<sdm_int>(pan1, 1 << 30));
  SDM_CATCH
}
int sdm_plan_panel_record(sdm_plan *p, sdm_int panel, int unpack, sdm_int *front_offset, sdm_int *front_nelem) {
  SDM_TRY
  if (!p->has_chol) throw std::runtime_error("sdm_plan_panel_record: no symbolic factor set");
  if (unpack >= 0) chol_panel_record(p, (int)panel, unpack);            // (unpack < 0: only the slice of "fronts" is asked for)
  const CholPlan &C = p->chol;
  if (C.nsuper != 1 || panel < 0 || panel * NB >= C.sn_ns[0]) throw std::runtime_error("sdm_plan_panel_record: one dense front, 0 <= panel < its panels");
  const sdm_int k0 = panel * NB, kb = std::min<sdm_int>(NB, C.sn_ns[0] - k0);
  if (front_offset) *front_offset = C.sn_foff[0] + k0 * (sdm_int)C.sn_ld[0];
  if (front_nelem) *front_nelem = kb * (sdm_int)C.sn_ld[0];
  SDM_CATCH
}
int sdm_plan_blkchol_levels(sdm_plan *p, sdm_int l0, sdm_int l1, int extend_only) {
  SDM_TRY
  chol_levels(p, (int)l0, (int)l1, extend_only != 0);
  SDM_CATCH
}
int sdm_plan_blkchol_end(sdm_plan *p) {
  SDM_TRY
  chol_end(p);
  SDM_CATCH
}
int sdm_plan_solve_levels(sdm_plan *p, int what, sdm_int l0, sdm_int l1) {
  SDM_TRY
  if (!p->factored) throw std::runtime_error("sdm_plan_solve_levels: no factor resident");
  if (p->dense.factored) throw std::runtime_error("sdm_plan_solve_levels: not with a resident dense-column factor");
  solve_levels(p, what, (int)l0, (int)l1);
  SDM_CATCH
}
int sdm_plan_load_factor(sdm_plan *p, const double *Lpr, const double *d) {
  SDM_TRY
  if (!p->has_chol) throw std::runtime_error("sdm_plan_load_factor: no symbolic factor set");
  SDM_HIP_CHECK(hipSetDevice(p->device));
  p->dense.factored = false;
  chol_load_factor(p, Lpr, d);
  SDM_CATCH
}
int sdm_plan_pivots(sdm_plan *p, sdm_int *nskip, sdm_int *skip_idx, double *skip_val, sdm_int *nadd,
                    sdm_int *add_idx, double *add_val) {
  SDM_TRY
  const sdm_int m = p->chol.m;
  std::vector<int> st(m);
  std::vector<double> val(m);
  SDM_HIP_CHECK(hipSetDevice(p->device));
  SDM_HIP_CHECK(hipStreamSynchronize(p->stream));
  check_plan_health(p);
  SDM_HIP_CHECK(hipMemcpy(st.data(), p->chol.pivstat.p, m * sizeof(int), hipMemcpyDeviceToHost));
  SDM_HIP_CHECK(hipMemcpy(val.data(), p->chol.pivval.p, m * sizeof(double), hipMemcpyDeviceToHost));
  sdm_int ns = 0, na = 0;
  for (sdm_int j = 0; j < m; j++) {
    if (st[j] == 1) { if (skip_idx) skip_idx[ns] = j; if (skip_val) skip_val[ns] = val[j]; ns++; }
    else if (st[j] == 2) { if (add_idx) add_idx[na] = j; if (add_val) add_val[na] = val[j]; na++; }
  }
  if (nskip) *nskip = ns;
  if (nadd) *nadd = na;
  SDM_CATCH
}
int sdm_plan_fwsolve(sdm_plan *p) {
  SDM_TRY
  if (!p->factored) throw std::runtime_error("fwsolve: no factor resident");
  solve_run(p, p->rhs.p, p->y.p, 1);
  SDM_CATCH
}
int sdm_plan_bwsolve(sdm_plan *p) {
  SDM_TRY
  if (!p->factored) throw std::runtime_error("bwsolve: no factor resident");
  solve_run(p, p->rhs.p, p->y.p, 4);
  SDM_CATCH
}
int sdm_plan_ldlsolve(sdm_plan *p) {
  SDM_TRY
  if (!p->factored) throw std::runtime_error("ldlsolve: no factor resident");
  solve_run(p, p->rhs.p, p->y.p, 7);
  SDM_CATCH
}
int sdm_plan_set_growth_max(sdm_plan *p, double growth_max) {
  SDM_TRY
  if (!(growth_max >= 0.0)) throw std::runtime_error("growth_max must be >= 0");
  p->chol.growth_max = growth_max;
  SDM_CATCH
}
int sdm_plan_set_refinement(sdm_plan *p, int mode, double refine_max) {
  SDM_TRY
  if (mode < 0 || mode > 2) throw std::runtime_error("sdm_plan_set_refinement: mode must be 0, 1 or 2");
  if (!(refine_max >= 0.0)) throw std::runtime_error("sdm_plan_set_refinement: refine_max must be >= 0");
  p->chol.refine_mode = mode; p->chol.refine_max = refine_max;
  SDM_CATCH
}
int sdm_plan_set_one_launch_fronts(sdm_plan *p, int on) {
  SDM_TRY
  p->chol.front_off_req = !on;
  SDM_CATCH
}
int sdm_plan_set_tile_workgroups(sdm_plan *p, int n) {
  SDM_TRY
  if (n < 0) throw std::runtime_error("sdm_plan_set_tile_workgroups: n must be >= 0");
  p->chol.tile_wgs_req = n;
  SDM_CATCH
}
int sdm_plan_set_one_launch_inverse(sdm_plan *p, int on) {
  SDM_TRY
  p->chol.sprep_off = !on;
  SDM_CATCH
}
int sdm_plan_set_solve_width(sdm_plan *p, sdm_int width) {
  SDM_TRY
  if (width != 0 && (width < sdm::SBW_MIN || width > sdm::SBW_MAX || (width & (width - 1)) != 0))
    throw std::runtime_error("solve width must be 0 (automatic) or a power of two in 256 .. 2048");
  p->chol.sbw_req = (int)width;
  SDM_CATCH
}
int sdm_plan_get_solve_width(sdm_plan *p, sdm_int *width) {
  SDM_TRY
  if (!p->has_chol) throw std::runtime_error("sdm_plan_get_solve_width: no symbolic factor set");
  *width = p->chol.sbw;
  SDM_CATCH
}
int sdm_plan_solve_stats(sdm_plan *p, sdm_int *nblocks, sdm_int *nbad, double *max_growth) {
  SDM_TRY
  if (!p->has_chol) throw std::runtime_error("sdm_plan_solve_stats: no symbolic factor set");
  solve_stats(p, nblocks, nbad, max_growth);
  SDM_CATCH
}
// ---- the operators around the solves in wrapPcg.m / loopPcg.m
int sdm_plan_pcg_init(sdm_plan *p, sdm_int nden, const sdm_int *dense_cols, const double *denseA) {
  SDM_TRY
  if (!p->has_ada) throw std::runtime_error("sdm_plan_pcg_init: no problem data set (sdm_plan_set_ada)");
  SDM_HIP_CHECK(hipSetDevice(p->device));
  pcg_set_dense(p, nden, dense_cols, denseA);
  SDM_CATCH
}
int sdm_plan_amul(sdm_plan *p, int transp) {
  SDM_TRY
  if (!p->has_ada) throw std::runtime_error("sdm_plan_amul: no problem data set (sdm_plan_set_ada)");
  pcg_amul(p, transp);
  SDM_CATCH
}
int sdm_plan_vecsym(sdm_plan *p) {
  SDM_TRY
  if (!p->has_ada) throw std::runtime_error("sdm_plan_vecsym: no problem data set (sdm_plan_set_ada)");
  pcg_vecsym(p);
  SDM_CATCH
}
int sdm_plan_psdscale(sdm_plan *p, int transp, int use_perm) {
  SDM_TRY
  if (!p->has_ada) throw std::runtime_error("sdm_plan_psdscale: no problem data set (sdm_plan_set_ada)");
  pcg_psdscale(p, transp, use_perm != 0);
  SDM_CATCH
}
// ---- resident dense-column unit (deninfac.m:58-94)
int sdm_plan_set_dense(sdm_plan *p, sdm_int nden, const sdm_int *LADjc, const sdm_int *LADir, const sdm_int *dzjc, const sdm_int *dzir,
                       const sdm_int *colperm, const sdm_int *first) {
  SDM_TRY
  if (!p->has_chol) throw std::runtime_error("sdm_plan_set_dense: call sdm_plan_set_chol first");
  SDM_HIP_CHECK(hipSetDevice(p->device));
  dense_set(p, nden, LADjc, LADir, dzjc, dzir, colperm, first);
  SDM_CATCH
}
int sdm_plan_deninfac(sdm_plan *p, const double *smult, double maxuden, int *host_fallback) {
  SDM_TRY
  if (p->capturing) throw std::runtime_error("sdm_plan_deninfac synchronises (it reads the stability flag): not inside a graph capture");
  dense_factor(p, smult, maxuden, host_fallback);
  SDM_CATCH
}
int sdm_plan_lden(sdm_plan *p, sdm_int *betajc, double *beta, double *pv, sdm_int *pivperm, sdm_int *npivperm, sdm_int *dopiv, double *Ld) {
  SDM_TRY
  DensePlan &D = p->dense;
  if (!D.factored) throw std::runtime_error("sdm_plan_lden: no dense-column factor resident");
  dense_fetch_tables(p->stream, D);                                   // (lengths, row-order flags: device tables of the factorisation)
  const sdm_int nden = D.nden, nb = D.betajc[nden], np = D.permoff[nden];
  for (sdm_int k = 0; k <= nden; k++) betajc[k] = D.betajc[k];
  for (sdm_int k = 0; k < nden; k++) dopiv[k] = D.dopiv[k];
  if (nb) SDM_HIP_CHECK(hipMemcpy(beta, D.beta.p, nb * sizeof(double), hipMemcpyDeviceToHost));
  if (D.pnnz) SDM_HIP_CHECK(hipMemcpy(pv, D.p.p, D.pnnz * sizeof(double), hipMemcpyDeviceToHost));
  std::vector<int> pp((size_t)std::max<sdm_int>(np, 1));
  if (np) SDM_HIP_CHECK(hipMemcpy(pp.data(), D.d_pivperm.p, np * sizeof(int), hipMemcpyDeviceToHost));
  for (sdm_int i = 0; i < np; i++) pivperm[i] = pp[i];
  *npivperm = np;
  SDM_HIP_CHECK(hipMemcpy(Ld, D.dden.p, p->chol.m * sizeof(double), hipMemcpyDeviceToHost));
  SDM_CATCH
}
// ---- hipGraph capture of a launch-bound sequence of plan calls (e.g. one whole iteration unit): everything the plan
// enqueues on its stream between begin and end becomes one executable graph; replay costs one launch.
int sdm_plan_graph_begin(sdm_plan *p) {
  SDM_TRY
  if (p->capturing) throw std::runtime_error("graph capture already in progress");
  if (p->kprof.enabled) throw std::runtime_error("per-kernel timing must be off during graph capture");
  SDM_HIP_CHECK(hipStreamSynchronize(p->stream));
  SDM_HIP_CHECK(hipStreamBeginCapture(p->stream, hipStreamCaptureModeThreadLocal));
  p->capturing = true;
  SDM_CATCH
}
int sdm_plan_graph_end(sdm_plan *p, int *graph_id) {
  SDM_TRY
  if (!p->capturing) throw std::runtime_error("no graph capture in progress");
  p->capturing = false;
  hipGraph_t g = nullptr;
  SDM_HIP_CHECK(hipStreamEndCapture(p->stream, &g));
  hipGraphExec_t ge = nullptr;
  hipError_t e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  SDM_HIP_CHECK(e);
  p->graphs.push_back(ge);
  if (graph_id) *graph_id = (int)p->graphs.size() - 1;
  SDM_CATCH
}
int sdm_plan_graph_launch(sdm_plan *p, int graph_id) {
  SDM_TRY
  if (graph_id < 0 || graph_id >= (int)p->graphs.size()) throw std::runtime_error("unknown graph id");
  SDM_HIP_CHECK(hipGraphLaunch(p->graphs[graph_id], p->stream));
  SDM_CATCH
}

int sdm_plan_timer_begin(sdm_plan *p, int slot) {
  SDM_TRY
  if (slot < 0 || slot >= 16) throw std::runtime_error("timer slot out of range");
  if (!p->ev_begin[slot]) { SDM_HIP_CHECK(hipEventCreate(&p->ev_begin[slot])); SDM_HIP_CHECK(hipEventCreate(&p->ev_end[slot])); }
  SDM_HIP_CHECK(hipEventRecord(p->ev_begin[slot], p->stream));
  SDM_CATCH
}
int sdm_plan_timer_end(sdm_plan *p, int slot) {
  SDM_TRY
  if (slot < 0 || slot >= 16 || !p->ev_end[slot]) throw std::runtime_error("timer slot not started");
  SDM_HIP_CHECK(hipEventRecord(p->ev_end[slot], p->stream));
  SDM_CATCH
}
int sdm_plan_timer_ms(sdm_plan *p, int slot, float *ms) {
  SDM_TRY
  if (slot < 0 || slot >= 16 || !p->ev_end[slot]) throw std::runtime_error("timer slot not started");
  SDM_HIP_CHECK(hipEventSynchronize(p->ev_end[slot]));
  SDM_HIP_CHECK(hipEventElapsedTime(ms, p->ev_begin[slot], p->ev_end[slot]));
  SDM_CATCH
}

// per-kernel timing (HIP events around every launch of the plan while enabled)
int sdm_plan_kprof_enable(sdm_plan *p, int on) {
  SDM_TRY
  SDM_HIP_CHECK(hipStreamSynchronize(p->stream));
  for (auto &r : p->kprof.recs) { p->kprof.pool.push_back(r.a); p->kprof.pool.push_back(r.b); }
  p->kprof.recs.clear();
  p->kprof.enabled = on != 0;
  SDM_CATCH
}
int sdm_plan_kprof_get(sdm_plan *p, const char *kernel, sdm_int *calls, double *total_ms) {
  SDM_TRY
  SDM_HIP_CHECK(hipStreamSynchronize(p->stream));
  sdm_int n = 0; double tot = 0.0;
  for (auto &r : p->kprof.recs)
    if (std::string(r.name) == kernel) { float ms = 0; SDM_HIP_CHECK(hipEventElapsedTime(&ms, r.a, r.b)); tot += ms; n++; }
  if (calls) *calls = n;
  if (total_ms) *total_ms = tot;
  SDM_CATCH
}
// writes "name:calls:ms;" records for every kernel seen into buf (truncated to buflen)
int sdm_plan_kprof_summary(sdm_plan *p, char *buf, sdm_int buflen) {
  SDM_TRY
  SDM_HIP_CHECK(hipStreamSynchronize(p->stream));
  std::vector<std::string> names; std::vector<sdm_int> cnt; std::vector<double> tot;
  for (auto &r : p->kprof.recs) {
    float ms = 0; SDM_HIP_CHECK(hipEventElapsedTime(&ms, r.a, r.b));
    size_t i = 0; for (; i < names.size(); i++) if (names[i] == r.name) break;
    if (i == names.size()) { names.push_back(r.name); cnt.push_back(0); tot.push_back(0.0); }
    cnt[i]++; tot[i] += ms;
  }
  std::string out;
  for (size_t i = 0; i < names.size(); i++) out += names[i] + ":" + std::to_string(cnt[i]) + ":" + std::to_string(tot[i]) + ";";
  if (buflen > 0) { size_t n = std::min<size_t>(out.size(), (size_t)buflen - 1); memcpy(buf, out.data(), n); buf[n] = 0; }
  SDM_CATCH
}

// ------------------------------------------------- tier (1): MEX equivalents
struct PlanGuard {
  sdm_plan *p;
  PlanGuard() : p(sdm_plan_create(0, nullptr)) { if (!p) throw std::runtime_error(g_err); }
  ~PlanGuard() { sdm_plan_destroy(p); }
};
}  // extern "C"

namespace sdm {
void gw_upload_invperm(DevBuf<int> &buf, const sdm_int *perm, sdm_int m) {
  std::vector<int> ip(m);
  for (sdm_int k = 0; k < m; k++) {
    if (perm[k] < 0 || perm[k] >= m) throw std::runtime_error("permutation entry out of range");
    ip[perm[k]] = (int)k;
  }
  buf.upload(ip);
}
// a trivial symbolic factor (diagonal) so that an ADA-only plan can be built
static void set_trivial_chol(sdm_plan *p, sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir) {
  std::vector<sdm_int> Ljc(m + 1), Lir(m), perm(m), xs(m + 1);
  for (sdm_int j = 0; j <= m; j++) { Ljc[j] = j; xs[j] = j; }
  for (sdm_int j = 0; j < m; j++) { Lir[j] = j; perm[j] = j; }
  if (sdm_plan_set_chol(p, m, Ljc.data(), Lir.data(), perm.data(), m, xs.data(), ADAjc, ADAir)) throw std::runtime_error(g_err);
}
// ---- gateway-shaped plans: a plan that holds exactly what ONE of the getada gateways needs (its slice of At / DAt.q, the ADA
// pattern).  The stateless tier-1 entry points below build one per call; the MEX cache (sdm_mexcache.hip) keeps them.
// ADA' values live in p->ada_val, absd in p->absd.
void gw_build_getada1(sdm_plan *p, sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir, const sdm_int *Ajc, const sdm_int *Air,
                      const double *Apr, const sdm_int *Ajc2, sdm_int lpN, sdm_int lorN, const sdm_int *qblkstart) {
  set_trivial_chol(p, m, ADAjc, ADAir);
  // only the LP/Lorentz rows matter here: present At as if it had no PSD part (rows >= qblkstart[lorN] are never touched:
  // Ajc2 is the end of the LP/Lorentz nonzeros)
  std::vector<sdm_int> Qjc(m + 1, 0);
  const sdm_int nlq = lorN > 0 ? qblkstart[lorN] : lpN;
  ada_build(p, nlq, m, Ajc, Air, Apr, Ajc2, lpN, lorN, nullptr, 0, 0, nullptr, qblkstart, nullptr, Qjc.data(), nullptr, ADAjc, ADAir);
}
void gw_run_getada1(sdm_plan *p, const int *d_invperm, const double *dl, const double *ddet) {
  AdaPlan &A = p->ada;
  if (A.lpN) SDM_HIP_CHECK(hipMemcpyAsync(A.dl.p, dl, A.lpN * sizeof(double), hipMemcpyHostToDevice, p->stream));
  if (A.lorN) SDM_HIP_CHECK(hipMemcpyAsync(A.ddet.p, ddet, A.lorN * sizeof(double), hipMemcpyHostToDevice, p->stream));
  SDM_HIP_CHECK(hipMemsetAsync(p->ada_val.p, 0, p->ada_val.n * sizeof(double), p->stream));   // getada1.c:222-225
  if (A.nnz_lq == 0) return;     // no LP / Lorentz nonzeros at all (MAXCUT): ADA' stays zero -- nnz(ADA') empty sparse dots took 0.6 ms at n = 4000
  ada_lq(p, p->ada_val.p, d_invperm, false);
}
bool gw_getada1_is_zero(sdm_plan *p) { return p->ada.nnz_lq == 0; }
void gw_build_getada2(sdm_plan *p, sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir, sdm_int lorN, const sdm_int *Qjc, const sdm_int *Qir) {
  set_trivial_chol(p, m, ADAjc, ADAir);
  std::vector<sdm_int> Ajc(m + 1, 0), qb(lorN + 1, 0);
  ada_build(p, lorN, m, Ajc.data(), nullptr, nullptr, Ajc.data(), 0, lorN, nullptr, 0, 0, nullptr, qb.data(), nullptr, Qjc, Qir, ADAjc, ADAir);
}
void gw_run_getada2(sdm_plan *p, const int *d_invperm, const double *Qpr) {   // p->ada_val in/out
  if (p->ada.nnzQ) SDM_HIP_CHECK(hipMemcpyAsync(p->ada.qpr.p, Qpr, p->ada.nnzQ * sizeof(double), hipMemcpyHostToDevice, p->stream));
  ada_q(p, p->ada_val.p, d_invperm, true);
}
void gw_build_getada3(sdm_plan *p, sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir, sdm_int N, const sdm_int *Ajc, const sdm_int *Air,
                      const double *Apr, const sdm_int *Ajc1, const sdm_cone *K, const sdm_int *psd_blkstart) {
  set_trivial_chol(p, m, ADAjc, ADAir);
  std::vector<sdm_int> Qjc(m + 1, 0), qb(K->lorN + 1, 0);
  // the LP/Lorentz rows are not read by getada3: describe them as plain LP rows up to the first PSD row
  const sdm_int nlq = K->sdpN > 0 ? psd_blkstart[0] : N;
  ada_build(p, N, m, Ajc, Air, Apr, Ajc1, nlq, 0, nullptr, K->sdpN, K->rsdpN, K->sdpNL, qb.data(), psd_blkstart, Qjc.data(), nullptr, ADAjc, ADAir);
}
void gw_run_getada3(sdm_plan *p, const double *udsqr, bool input_is_zero) {   // p->ada_val in/out, p->absd out
  // input_is_zero: the ADA' handed in is known to be the zero matrix (getada1 / getada2 had nothing to add): nothing to symmetrise
  // (k_symmetrize + the copy back: 0.28 ms and 1.8 GB of traffic at n = 4000)
  if (p->ada.lenud) SDM_HIP_CHECK(hipMemcpyAsync(p->ada.udsqr.p, udsqr, p->ada.lenud * sizeof(double), hipMemcpyHostToDevice, p->stream));
  ada_psd(p, p->ada_val.p, nullptr, !input_is_zero);
}
// the whole ADA' of a problem WITHOUT PSD blocks (getada.m:13-40)
void gw_build_getada(sdm_plan *p, sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir, const sdm_int *Ajc, const sdm_int *Air, const double *Apr,
                     sdm_int lpN, sdm_int lorN, const sdm_int *qblkstart, const sdm_int *Qjc, const sdm_int *Qir) {
  set_trivial_chol(p, m, ADAjc, ADAir);
  const sdm_int nlq = lorN > 0 ? qblkstart[lorN] : lpN;             // = K.mainblks(3)-1: rows of Alq (getada.m:25)
  std::vector<sdm_int> ajc2(m), zq(m + 1, 0);
  for (sdm_int j = 0; j < m; j++) ajc2[j] = std::lower_bound(Air + Ajc[j], Air + Ajc[j + 1], nlq) - Air;
  ada_build(p, nlq, m, Ajc, Air, Apr, ajc2.data(), lpN, lorN, nullptr, 0, 0, nullptr, qblkstart, nullptr,
            lorN > 0 ? Qjc : zq.data(), lorN > 0 ? Qir : nullptr, ADAjc, ADAir);
}
void gw_run_getada(sdm_plan *p, const double *dl, const double *ddet, const double *Qpr) {
  AdaPlan &A = p->ada;
  if (A.lpN) SDM_HIP_CHECK(hipMemcpyAsync(A.dl.p, dl, A.lpN * sizeof(double), hipMemcpyHostToDevice, p->stream));
  if (A.lorN) SDM_HIP_CHECK(hipMemcpyAsync(A.ddet.p, ddet, A.lorN * sizeof(double), hipMemcpyHostToDevice, p->stream));
  if (A.lorN && A.nnzQ) SDM_HIP_CHECK(hipMemcpyAsync(A.qpr.p, Qpr, A.nnzQ * sizeof(double), hipMemcpyHostToDevice, p->stream));
  if (sdm_plan_getada(p)) throw std::runtime_error(g_err);
}
// values of ADA' (and absd) back to the host once the plan's stream has drained
void gw_download(sdm_plan *p, double *ADApr, double *absd) {
  if (ADApr) SDM_HIP_CHECK(hipMemcpyAsync(ADApr, p->ada_val.p, p->ada_val.n * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  if (absd) SDM_HIP_CHECK(hipMemcpyAsync(absd, p->absd.p, (size_t)p->chol.m * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  SDM_HIP_CHECK(hipStreamSynchronize(p->stream));
}
}  // namespace sdm

extern "C" {

int sdm_getada1(sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir, sdm_int N, const sdm_int *Ajc,
                const sdm_int *Air, const double *Apr, const sdm_int *Ajc2, const sdm_int *perm, sdm_int lpN,
                const double *dl, sdm_int lorN, const double *ddet, const sdm_int *qblkstart, double *ADApr) {
  SDM_TRY
  (void)N;
  PlanGuard G; sdm_plan *p = G.p;
  gw_build_getada1(p, m, ADAjc, ADAir, Ajc, Air, Apr, Ajc2, lpN, lorN, qblkstart);
  DevBuf<int> ip; gw_upload_invperm(ip, perm, m);
  gw_run_getada1(p, ip.p, dl, ddet);
  gw_download(p, ADApr, nullptr);
  SDM_CATCH
}

int sdm_getada2(sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir, double *ADApr, sdm_int lorN,
                const sdm_int *Qjc, const sdm_int *Qir, const double *Qpr, const sdm_int *qperm) {
  SDM_TRY
  if (lorN <= 0) return 0;                          // getada2.c:154-155: nothing to do without Lorentz cones
  PlanGuard G; sdm_plan *p = G.p;
  gw_build_getada2(p, m, ADAjc, ADAir, lorN, Qjc, Qir);
  SDM_HIP_CHECK(hipMemcpy(p->ada_val.p, ADApr, (size_t)ADAjc[m] * sizeof(double), hipMemcpyHostToDevice));
  DevBuf<int> ip; gw_upload_invperm(ip, qperm, m);
  gw_run_getada2(p, ip.p, Qpr);
  gw_download(p, ADApr, nullptr);
  SDM_CATCH
}

int sdm_getada3(sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir, double *ADApr, sdm_int N, const sdm_int *Ajc,
                const sdm_int *Air, const double *Apr, const sdm_int *Ajc1, const sdm_int *sperm, const double *udsqr,
                const sdm_cone *K, const sdm_int *psd_blkstart, double *absd) {
  SDM_TRY
  (void)sperm;   // the sum is independent of the fill order (see sdm_ada.hip header)
  PlanGuard G; sdm_plan *p = G.p;
  gw_build_getada3(p, m, ADAjc, ADAir, N, Ajc, Air, Apr, Ajc1, K, psd_blkstart);
  SDM_HIP_CHECK(hipMemcpy(p->ada_val.p, ADApr, (size_t)ADAjc[m] * sizeof(double), hipMemcpyHostToDevice));
  gw_run_getada3(p, udsqr, false);
  gw_download(p, ADApr, absd);
  SDM_CATCH
}

// absd = getada(A, K, d, DAt) with the global ADA_sedumi_ : the whole ADA' of a problem WITHOUT PSD blocks
// (sedumi.m:446-448 -> getada.m:13-40):  ADA = DAt.q' DAt.q + Alq' diag([d.l; -d.det; d.det per norm-bound row]) Alq
// on the pattern of the global (both triangles), absd = diag(ADA) (getada.m:40).
int sdm_getada(sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir, sdm_int N, const sdm_int *Ajc, const sdm_int *Air,
               const double *Apr, sdm_int lpN, const double *dl, sdm_int lorN, const double *ddet, const sdm_int *qblkstart,
               const sdm_int *Qjc, const sdm_int *Qir, const double *Qpr, double *ADApr, double *absd) {
  SDM_TRY
  (void)N;
  PlanGuard G; sdm_plan *p = G.p;
  gw_build_getada(p, m, ADAjc, ADAir, Ajc, Air, Apr, lpN, lorN, qblkstart, Qjc, Qir);
  gw_run_getada(p, dl, ddet, Qpr);
  gw_download(p, ADApr, absd);
  SDM_CATCH
}

int sdm_blkchol(sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const sdm_int *perm, sdm_int nsuper,
                const sdm_int *xsuper, const sdm_int *Xjc, const sdm_int *Xir, const double *Xpr,
                const sdm_cholpars *pars, const double *absd, double *Lpr, double *d, sdm_int *nskip,
                sdm_int *skip_idx, double *skip_val, sdm_int *nadd, sdm_int *add_idx, double *add_val) {
  SDM_TRY
  PlanGuard G; sdm_plan *p = G.p;
  if (sdm_plan_set_chol(p, m, Ljc, Lir, perm, nsuper, xsuper, Xjc, Xir)) throw std::runtime_error(g_err);
  SDM_HIP_CHECK(hipMemcpy(p->ada_val.p, Xpr, (size_t)Xjc[m] * sizeof(double), hipMemcpyHostToDevice));
  if (absd) SDM_HIP_CHECK(hipMemcpy(p->absd.p, absd, m * sizeof(double), hipMemcpyHostToDevice));
  if (sdm_plan_blkchol_wait(p, pars, absd ? 1 : 0)) throw std::runtime_error(g_err);
  if (sdm_plan_download(p, "lpr", Lpr, Ljc[m])) throw std::runtime_error(g_err);
  if (sdm_plan_download(p, "d", d, m)) throw std::runtime_error(g_err);
  if (sdm_plan_pivots(p, nskip, skip_idx, skip_val, nadd, add_idx, add_val)) throw std::runtime_error(g_err);
  SDM_CATCH
}

static void solve_common(bool fw, sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const double *Lpr,
                         const sdm_int *perm, sdm_int nsuper, const sdm_int *xsuper, sdm_int nrhs, const double *b,
                         double *y) {
  PlanGuard G; sdm_plan *p = G.p;
  // the ADA pattern is irrelevant for a solve: use an empty one
  std::vector<sdm_int> zj(m + 1, 0), idperm;
  if (!perm) { idperm.resize(m); for (sdm_int i = 0; i < m; i++) idperm[i] = i; perm = idperm.data(); }
  if (sdm_plan_set_chol(p, m, Ljc, Lir, perm, nsuper, xsuper, zj.data(), nullptr)) throw std::runtime_error(g_err);
  chol_load_factor(p, Lpr);
  for (sdm_int c = 0; c < nrhs; c++) {
    SDM_HIP_CHECK(hipMemcpyAsync(p->rhs.p, b + c * m, m * sizeof(double), hipMemcpyHostToDevice, p->stream));
    if ((fw ? sdm_plan_fwsolve(p) : sdm_plan_bwsolve(p))) throw std::runtime_error(g_err);
    SDM_HIP_CHECK(hipMemcpyAsync(y + c * m, p->y.p, m * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    SDM_HIP_CHECK(hipStreamSynchronize(p->stream));
  }
}
int sdm_fwblkslv(sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const double *Lpr, const sdm_int *perm,
                 sdm_int nsuper, const sdm_int *xsuper, sdm_int nrhs, const double *b, double *y) {
  SDM_TRY
  solve_common(true, m, Ljc, Lir, Lpr, perm, nsuper, xsuper, nrhs, b, y);
  SDM_CATCH
}
int sdm_bwblkslv(sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const double *Lpr, const sdm_int *perm,
                 sdm_int nsuper, const sdm_int *xsuper, sdm_int nrhs, const double *b, double *y) {
  SDM_TRY
  solve_common(false, m, Ljc, Lir, Lpr, perm, nsuper, xsuper, nrhs, b, y);
  SDM_CATCH
}

// sparse right-hand sides: the pattern (Yjc,Yir) from symbfwblk is closed under the solve, so the
// dense solve restricted to it gives exactly the entries selfwsolve / selbwsolve define.
static void solve_sparse(bool fw, sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const double *Lpr,
                         const sdm_int *perm, sdm_int nsuper, const sdm_int *xsuper, sdm_int n, const sdm_int *Bjc,
                         const sdm_int *Bir, const double *Bpr, const sdm_int *Yjc, const sdm_int *Yir, double *Ypr) {
  PlanGuard G; sdm_plan *p = G.p;
  std::vector<sdm_int> zj(m + 1, 0), idperm(m);
  for (sdm_int i = 0; i < m; i++) idperm[i] = i;
  // forward variant maps b through invperm == dense fwblkslv on a dense copy of b (fwblkslv.c:308-309);
  // backward variant uses no permutation at all (bwblkslv.c:279-291)
  if (sdm_plan_set_chol(p, m, Ljc, Lir, fw ? perm : idperm.data(), nsuper, xsuper, zj.data(), nullptr)) throw std::runtime_error(g_err);
  chol_load_factor(p, Lpr);
  std::vector<double> col(m), out(m);
  for (sdm_int c = 0; c < n; c++) {
    std::fill(col.begin(), col.end(), 0.0);
    for (sdm_int t = Bjc[c]; t < Bjc[c + 1]; t++) col[Bir[t]] = Bpr[t];
    SDM_HIP_CHECK(hipMemcpyAsync(p->rhs.p, col.data(), m * sizeof(double), hipMemcpyHostToDevice, p->stream));
    if ((fw ? sdm_plan_fwsolve(p) : sdm_plan_bwsolve(p))) throw std::runtime_error(g_err);
    SDM_HIP_CHECK(hipMemcpyAsync(out.data(), p->y.p, m * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    SDM_HIP_CHECK(hipStreamSynchronize(p->stream));
    for (sdm_int t = Yjc[c]; t < Yjc[c + 1]; t++) Ypr[t] = out[Yir[t]];
  }
}
int sdm_fwblkslv_sparse(sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const double *Lpr, const sdm_int *perm,
                        sdm_int nsuper, const sdm_int *xsuper, sdm_int n, const sdm_int *Bjc, const sdm_int *Bir,
                        const double *Bpr, const sdm_int *Yjc, const sdm_int *Yir, double *Ypr) {
  SDM_TRY
  solve_sparse(true, m, Ljc, Lir, Lpr, perm, nsuper, xsuper, n, Bjc, Bir, Bpr, Yjc, Yir, Ypr);
  SDM_CATCH
}
int sdm_bwblkslv_sparse(sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const double *Lpr, sdm_int nsuper,
                        const sdm_int *xsuper, sdm_int n, const sdm_int *Bjc, const sdm_int *Bir, const double *Bpr,
                        const sdm_int *Yjc, const sdm_int *Yir, double *Ypr) {
  SDM_TRY
  solve_sparse(false, m, Ljc, Lir, Lpr, nullptr, nsuper, xsuper, n, Bjc, Bir, Bpr, Yjc, Yir, Ypr);
  SDM_CATCH
}


// ---- invcholfac
static void cone_blocks(const sdm_cone *K, std::vector<int> &ns, sdm_int &lenud, sdm_int &plen) {
  ns.resize((size_t)K->sdpN); lenud = 0; plen = 0;
  for (sdm_int k = 0; k < K->sdpN; k++) {
    ns[k] = (int)K->sdpNL[k];
    lenud += (k < K->rsdpN ? 1 : 2) * K->sdpNL[k] * K->sdpNL[k];
    plen += K->sdpNL[k];
  }
}
static void perm32(const sdm_int *perm, const std::vector<int> &ns, std::vector<int> &out) {
  out.clear();
  size_t o = 0;
  for (int n : ns) {
    std::vector<char> seen((size_t)n, 0);
    for (int i = 0; i < n; i++) {
      const sdm_int v = perm[o + i];
      if (v < 0 || v >= n || seen[(size_t)v]) throw std::runtime_error("perm is not a permutation of the block");
      seen[(size_t)v] = 1; out.push_back((int)v);
    }
    o += (size_t)n;
  }
}
int sdm_invcholfac(const sdm_cone *K, const double *u, const sdm_int *perm, double *y) {
  SDM_TRY
  std::vector<int> ns, p32; sdm_int lenud, plen;
  cone_blocks(K, ns, lenud, plen);
  if (lenud == 0) return 0;
  PlanGuard G; sdm_plan *p = G.p;
  DevBuf<double> du, dy; DevBuf<int> dp, dn, dpo; DevBuf<int64_t> doff;
  du.upload(u, (size_t)lenud); dy.alloc((size_t)lenud);
  if (perm) { perm32(perm, ns, p32); dp.upload(p32); }
  psd_invcholfac(p->stream, du.p, dy.p, perm ? dp.p : nullptr, ns, (int)K->rsdpN, dn, doff, dpo, false);
  SDM_HIP_CHECK(hipStreamSynchronize(p->stream));
  SDM_HIP_CHECK(hipMemcpy(y, dy.p, (size_t)lenud * sizeof(double), hipMemcpyDeviceToHost));
  SDM_CATCH
}
int sdm_plan_invcholfac(sdm_plan *p, const sdm_int *perm) {
  SDM_TRY
  AdaPlan &A = p->ada;
  if (A.lenud == 0) return 0;
  if (A.ufac.n < (size_t)A.lenud) throw std::runtime_error("invcholfac: upload buffer \"u\" first");
  std::vector<int> ns(A.psd_n.begin(), A.psd_n.end()), p32;
  if (perm) {
    // the permutation is staged in a plan-owned pinned buffer that outlives the call (the copy is asynchronous);
    // inside a graph capture the copy node would keep reading that buffer at every replay, so a NEW permutation
    // cannot be handed over there -- upload it before the capture
    if (p->capturing) throw std::runtime_error("sdm_plan_invcholfac: a permutation cannot be uploaded inside a graph capture (the captured copy would replay from the staging buffer)");
    perm32(perm, ns, p32);
    if (A.ic_perm.n < p32.size()) A.ic_perm.alloc(p32.size());
    A.ic_perm_host.ensure(p32.size());
    SDM_HIP_CHECK(hipStreamSynchronize(p->stream));                 // an earlier copy out of the staging buffer may be pending
    memcpy(A.ic_perm_host.p, p32.data(), p32.size() * sizeof(int));
    SDM_HIP_CHECK(hipMemcpyAsync(A.ic_perm.p, A.ic_perm_host.p, p32.size() * sizeof(int), hipMemcpyHostToDevice, p->stream));
    A.ic_has_perm = true;
  }
  const bool ready = A.ic_n.n == ns.size() && !ns.empty();           // block tables: once per plan (set_ada resets them)
  psd_invcholfac(p->stream, A.ufac.p, A.udsqr.p, perm ? A.ic_perm.p : nullptr, ns, (int)A.rsdpN, A.ic_n, A.ic_off, A.ic_poff, ready);
  SDM_CATCH
}
}  // extern "C"
