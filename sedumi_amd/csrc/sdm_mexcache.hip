// sdm_mexcache.hip -- the process-wide resident state behind the mexFunction shims (INTEGRATION.md "Keeping data on the
// device across MEX calls").  Every .mex binary is its own shared object with its own statics, so a cache kept inside the
// shims would never be shared between getada3.mex, blkchol.mex and fwblkslv.mex; libsedumi_hip.so is loaded once per
// process, so the cache lives here, behind the C ABI.
//
// What a MATLAB session hands the gateways of one solve, call after call (sedumi.m:450-462, wrapPcg.m:56-59):
//   * the SAME problem data every iteration -- At, K, the ADA pattern, the symbolic factor: recognised by fingerprints
//     (below), their device-side analysis (ada_build, chol_build) is done once and kept;
//   * the value arrays one gateway returned as the next one's input -- ADA from getada1 -> getada2 -> getada3 -> blkchol,
//     L.L from blkchol -> fwblkslv / bwblkslv: the device still holds them, so they are not uploaded again.
// Residency is never assumed from a host address alone (MATLAB may free an array and hand the same address to another one
// of the same shape): every reuse is backed by a content fingerprint of what the host array holds NOW.
//
// Fingerprints.  `Finger` records length, a sampled hash (SAMPLE evenly spaced words) and, up to FULL_MAX words, the hash
// of every word, plus the addresses at which this content has been seen.  A candidate matches if its sampled hash agrees
// and (a) it sits at a known address, or (b) its full hash agrees (the address is then remembered).  Arrays beyond
// FULL_MAX words at an unknown address do not match: the analysis is redone / the values are uploaded (correct, slower).
// Limit of (a): an edit IN PLACE of an array the cache has seen, at words the sample does not touch, goes unnoticed --
// MATLAB's value semantics make that an exclusive-owner `X(i) = v` between two gateway calls, which sedumi.m never does;
// sdm_mexcache_set_strict(1) hashes every word in case (a) as well.
#include "sdm_plan.h"
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace sdm;

namespace {
typedef unsigned long long u64;
constexpr sdm_int SAMPLE = 512;            // words of the sampled hash (each a cache and TLB miss on a big array: 4096 of them cost 0.7 ms per check on a 64 MB factor)
constexpr sdm_int FULL_MAX = 1 << 22;      // full hash only up to 4M words (32 MB, ~2 ms on the host)
constexpr u64 P1 = 11400714785074694791ull, P2 = 14029467366897019727ull;
inline u64 rotl(u64 x, int r) { return (x << r) | (x >> (64 - r)); }
inline u64 lane(u64 acc, u64 v) { return rotl(acc + v * P2, 31) * P1; }
// every word, position dependent, as eight independent wrap-around sums of 32 x 32 -> 64 bit products: the compiler vectorises
// it on the host, and -- sums commute -- the same value comes out of a parallel reduction on the device (k_words_checksum:
// what blkchol leaves in HBM is fingerprinted there for free instead of by a pass over the host copy).  A change detector, not a
// cryptographic hash: any single changed word changes its sum.
#define SDM_CK_C1 {0x9E3779B1u, 0x85EBCA77u, 0xC2B2AE3Du, 0x27D4EB2Fu, 0x165667B1u, 0xD3A2646Du, 0xFD7046C5u, 0xB55A4F09u}
#define SDM_CK_C2 {0x8DA6B343u, 0xD8163841u, 0xCB1AB31Fu, 0x9F6B3F4Bu, 0xA54FF53Bu, 0x3C6EF373u, 0xBB67AE85u, 0x6A09E667u}
u64 fold_sums(const u64 *acc, sdm_int n) {
  u64 h = (u64)n;
  for (int k = 0; k < 8; k++) h = lane(h, acc[k]);
  return h;
}
u64 hash_full(const void *pv, sdm_int n) {
  const u64 *v = (const u64 *)pv;
  static const u64 C1[8] = SDM_CK_C1, C2[8] = SDM_CK_C2;
  u64 acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  sdm_int i = 0;
  for (; i + 8 <= n; i += 8)
    for (int k = 0; k < 8; k++) {
      const u64 w = v[i + k];
      acc[k] += (u64)((unsigned)w ^ (unsigned)i) * C1[k] + (u64)((unsigned)(w >> 32) ^ (unsigned)i) * C2[k];
    }
  for (int k = 0; i + k < n; k++) {
    const u64 w = v[i + k];
    acc[k] += (u64)((unsigned)w ^ (unsigned)i) * C1[k] + (u64)((unsigned)(w >> 32) ^ (unsigned)i) * C2[k];
  }
  return fold_sums(acc, n);
}
// the same eight sums of n words in device memory, added into acc8 (zeroed by the caller): work-item t owns the words
// t, t + stride, ... with stride a multiple of 8, i.e. always the same sum
__global__ void k_words_checksum(const unsigned long long *v, long long n, unsigned long long *acc8) {
  const unsigned long long C1[8] = SDM_CK_C1, C2[8] = SDM_CK_C2;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
  const int k = (int)(t & 7);
  unsigned long long s = 0;
  for (long long i = t; i < n; i += stride) {
    const unsigned long long w = v[i];
    const unsigned base = (unsigned)(i & ~7ll);
    s += (unsigned long long)((unsigned)w ^ base) * C1[k] + (unsigned long long)((unsigned)(w >> 32) ^ base) * C2[k];
  }
  if (s) atomicAdd(&acc8[k], s);
}
u64 hash_sampled(const void *pv, sdm_int n) {                         // always the first and the last word
  const u64 *v = (const u64 *)pv;
  if (n <= 0) return 7;
  const sdm_int step = n <= SAMPLE ? 1 : (n + SAMPLE - 1) / SAMPLE;
  u64 h = (u64)n;
  for (sdm_int i = 0; i < n; i += step) h = lane(h, v[i]);
  return lane(h, v[n - 1]);
}
bool strict = false;      // sdm_mexcache_set_strict: every word of an array at a known address is hashed too (in-place edits between calls)
struct Finger {
  sdm_int n = -1;
  u64 sample = 0, full = 0;
  bool have_full = false;
  const void *at[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // (the last eight addresses it was seen at)
  int nat = 0;
  void seen_at(const void *p) {
    for (int i = 0; i < 8; i++) if (at[i] == p) return;
    at[nat++ & 7] = p;
  }
  // want_full = false: the content is only ever recognised at this address (value arrays that travel from one gateway to
  // the next by reference; at another address they are simply uploaded again)
  // any_size: the full checksum whatever the length (patterns: once per solve; without it an array beyond FULL_MAX is only ever
  // recognised at an address it has been seen at)
  void take(const void *p, sdm_int len, bool want_full = true, bool any_size = false) {
    n = len; sample = hash_sampled(p, len);
    have_full = want_full && (any_size || len <= FULL_MAX);
    full = have_full ? hash_full(p, len) : 0;
    for (auto &a : at) a = nullptr;
    nat = 0; seen_at(p);
  }
  // the full hash computed elsewhere (on the device copy of the same values)
  void take_with_full(const void *p, sdm_int len, u64 fullhash) {
    take(p, len, false);
    have_full = true; full = fullhash;
  }
  bool same(const void *p, sdm_int len) {
    if (len != n || n < 0) return false;
    if (hash_sampled(p, len) != sample) return false;
    for (int i = 0; i < 8; i++) if (at[i] == p && p) return !(strict && have_full) || hash_full(p, len) == full;
    if (!have_full || hash_full(p, len) != full) return false;
    seen_at(p);
    return true;
  }
  void forget() { n = -1; }
};

// ---- index patterns (CSC jc / ir pairs): interned, so that the slots below compare small integers.  An id is never reused.
struct Pattern { sdm_int ncol = -1; Finger jc, ir; u64 id = 0, used = 0; };
Pattern g_pat[12];
u64 g_next_id = 1, g_clock = 0;
u64 intern(sdm_int ncol, const sdm_int *jc, const sdm_int *ir) {
  const sdm_int nnz = jc[ncol];
  for (auto &p : g_pat)
    if (p.id && p.ncol == ncol && p.jc.same(jc, ncol + 1) && p.ir.same(ir, nnz)) { p.used = ++g_clock; return p.id; }
  Pattern *v = &g_pat[0];
  for (auto &p : g_pat) if (p.used < v->used) v = &p;                 // least recently used (empty entries first: used = 0)
  v->ncol = ncol; v->jc.take(jc, ncol + 1, true, true); v->ir.take(ir, nnz, true, true); v->id = g_next_id++; v->used = ++g_clock;
  return v->id;
}
// `ir` is a copy of the row indices of pattern `id` that a shim made for the array it returns: the next call presents it
void alias(u64 id, const sdm_int *ir) {
  if (!ir) return;
  for (auto &p : g_pat) if (p.id == id) p.ir.seen_at(ir);
}

sdm_int g_stat[16];       // counters for tests and the bench (sdm_mexcache_stats)
enum { ST_ADA_BUILD = 0, ST_ADA_REUSE = 1, ST_ADA_UPLOAD = 2, ST_ADA_RESIDENT = 3, ST_CHOL_BUILD = 4, ST_CHOL_REUSE = 5,
       ST_X_UPLOAD = 6, ST_X_RESIDENT = 7, ST_SOLVE_RESIDENT = 8, ST_SOLVE_STATELESS = 9, ST_AT_UPLOAD = 10 };

int device() { const char *dev = getenv("SEDUMI_HIP_DEVICE"); return dev ? atoi(dev) : 0; }
sdm_plan *new_plan() {
  sdm_plan *p = sdm_plan_create(device(), nullptr);
  if (!p) throw std::runtime_error(sdm_last_error());
  return p;
}

// ---- one slot per getada gateway: the plan that holds its slice of the problem, and what it was built from
struct AdaSlot {
  sdm_plan *plan = nullptr;
  bool built = false;
  u64 pat_ada = 0, pat_a = 0, pat_q = 0;
  Finger apr;
  std::vector<sdm_int> split, ints, perm;      // Ajc1 / Ajc2; the scalar and small-vector arguments (cone, block starts); the last permutation
  DevBuf<int> invperm;
  void drop() { if (plan) sdm_plan_destroy(plan); plan = nullptr; built = false; perm.clear(); invperm.release(); }
  void set_perm(const sdm_int *pm, sdm_int m) {
    if ((sdm_int)perm.size() == m && (m == 0 || memcmp(perm.data(), pm, (size_t)m * sizeof(sdm_int)) == 0) && invperm.p) return;
    gw_upload_invperm(invperm, pm, m);
    perm.assign(pm, pm + m);
  }
};
AdaSlot g_s1, g_s2, g_s3, g_s0;

// whose ada_val holds the ADA values most recently handed back to the host, and what they were
struct { sdm_plan *plan = nullptr; Finger vals; } g_last;
void returned(sdm_plan *p, const double *pr, sdm_int nnz) { g_last.plan = p; g_last.vals.take(pr, nnz, false); }
// the input values of a gateway -> p->ada_val: from the device if they are what the previous gateway returned, else from the host
void ada_input(sdm_plan *p, const double *pr, sdm_int nnz) {
  if (g_last.plan && g_last.vals.same(pr, nnz)) {
    g_stat[ST_ADA_RESIDENT]++;
    if (g_last.plan != p)
      SDM_HIP_CHECK(hipMemcpyAsync(p->ada_val.p, g_last.plan->ada_val.p, (size_t)nnz * sizeof(double), hipMemcpyDeviceToDevice, p->stream));
    return;
  }
  g_stat[ST_ADA_UPLOAD]++;
  SDM_HIP_CHECK(hipMemcpyAsync(p->ada_val.p, pr, (size_t)nnz * sizeof(double), hipMemcpyHostToDevice, p->stream));
}
void forget_plan(sdm_plan *p) { if (g_last.plan == p) { g_last.plan = nullptr; g_last.vals.forget(); } }

// ---- the factorisation's plan
struct Chol {
  sdm_plan *plan = nullptr;
  u64 pat_l = 0, pat_x = 0;
  std::vector<sdm_int> perm, xsuper;
  bool have_factor = false;
  Finger lpr;
  DevBuf<u64> ck;                        // eight device words: checksum of the factor's values (k_words_checksum)
} g;

void drop_chol() {
  if (g.plan) { forget_plan(g.plan); sdm_plan_destroy(g.plan); }
  g.ck.release();
  g.plan = nullptr; g.pat_l = g.pat_x = 0; g.perm.clear(); g.xsuper.clear(); g.have_factor = false; g.lpr.forget();
}
void drop_all() {
  for (AdaSlot *s : {&g_s1, &g_s2, &g_s3, &g_s0}) { if (s->plan) forget_plan(s->plan); s->drop(); }
  drop_chol();
  for (auto &p : g_pat) p = Pattern();
}
void at_exit_once() {
  // the cached plans own streams, events and device memory: they go before the HIP runtime's own exit handlers run (registered
  // by the first HIP call of the process -- at the latest the plan creation that precedes this -- so this one, registered after
  // them, runs first).  A plan left alive across process exit crashed there (r03g).
  static bool done = false;
  if (!done) { done = true; std::atexit([] { drop_all(); }); }
}
bool vec_is(const std::vector<sdm_int> &v, const sdm_int *p, sdm_int n) {
  return (sdm_int)v.size() == n && (n == 0 || memcmp(v.data(), p, (size_t)n * sizeof(sdm_int)) == 0);
}

sdm_plan *chol_plan(sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const sdm_int *perm, sdm_int nsuper, const sdm_int *xsuper,
                    const sdm_int *Xjc, const sdm_int *Xir) {
  const u64 pl = intern(m, Ljc, Lir), px = intern(m, Xjc, Xir);
  if (g.plan && g.pat_l == pl && g.pat_x == px && vec_is(g.perm, perm, m) && vec_is(g.xsuper, xsuper, nsuper + 1)) { g_stat[ST_CHOL_REUSE]++; return g.plan; }
  drop_chol();
  g.plan = new_plan();
  at_exit_once();
  g_stat[ST_CHOL_BUILD]++;
  if (sdm_plan_set_chol(g.plan, m, Ljc, Lir, perm, nsuper, xsuper, Xjc, Xir)) { std::string e = sdm_last_error(); drop_chol(); throw std::runtime_error(e); }
  g.pat_l = pl; g.pat_x = px; g.perm.assign(perm, perm + m); g.xsuper.assign(xsuper, xsuper + nsuper + 1);
  return g.plan;
}

#define MC_TRY try { SDM_HIP_CHECK(hipSetDevice(device()));
#define MC_CATCH                                                    \
  }                                                                 \
  catch (const std::exception &e) { set_error(e.what()); return 1; } \
  catch (...) { set_error("unknown error"); return 1; }             \
  return 0;
}  // namespace

extern "C" {

void sdm_mexcache_clear(void) { drop_all(); }
// on != 0: an array presented at an address the cache knows is compared word for word (full hash) instead of by its sampled
// hash -- for callers that edit arrays IN PLACE between gateway calls (sedumi.m never does); costs ~0.7 ns per word and call
void sdm_mexcache_set_strict(int on) { strict = on != 0; }
void sdm_mexcache_stats(sdm_int *out, sdm_int n) { for (sdm_int i = 0; i < n && i < 16; i++) out[i] = g_stat[i]; }

// ADA = getada1(ADA, A, Ajc2, perm, d, blkstart) on the cache (same arguments as sdm_getada1; ADAir_out: the row indices of
// the array the shim returns, a copy of ADAir, or NULL)
int sdm_mexcache_getada1(sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir, sdm_int N, const sdm_int *Ajc, const sdm_int *Air,
                         const double *Apr, const sdm_int *Ajc2, const sdm_int *perm, sdm_int lpN, const double *dl, sdm_int lorN,
                         const double *ddet, const sdm_int *qblkstart, double *ADApr, const sdm_int *ADAir_out) {
  MC_TRY
  AdaSlot &S = g_s1;
  const u64 pa = intern(m, ADAjc, ADAir), pA = intern(m, Ajc, Air);
  std::vector<sdm_int> ints = {m, N, lpN, lorN};
  ints.insert(ints.end(), qblkstart, qblkstart + lorN + 1);
  if (!(S.built && S.pat_ada == pa && S.pat_a == pA && S.ints == ints && vec_is(S.split, Ajc2, m) && S.apr.same(Apr, Ajc[m]))) {
    if (!S.plan) S.plan = new_plan();
    at_exit_once();
    S.built = false; forget_plan(S.plan);
    g_stat[ST_ADA_BUILD]++; g_stat[ST_AT_UPLOAD]++;
    gw_build_getada1(S.plan, m, ADAjc, ADAir, Ajc, Air, Apr, Ajc2, lpN, lorN, qblkstart);
    S.pat_ada = pa; S.pat_a = pA; S.ints = ints; S.split.assign(Ajc2, Ajc2 + m); S.apr.take(Apr, Ajc[m]); S.perm.clear();
    S.built = true;
  } else g_stat[ST_ADA_REUSE]++;
  S.set_perm(perm, m);
  gw_run_getada1(S.plan, S.invperm.p, dl, ddet);
  gw_download(S.plan, ADApr, nullptr);
  returned(S.plan, ADApr, ADAjc[m]);
  alias(pa, ADAir_out);
  MC_CATCH
}

// ADA = getada2(ADA, DAt, Aord, K): ADApr_in the values of the input array, ADApr (out) those of the copy the shim returns
int sdm_mexcache_getada2(sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir, const double *ADApr_in, double *ADApr, sdm_int lorN,
                         const sdm_int *Qjc, const sdm_int *Qir, const double *Qpr, const sdm_int *qperm, const sdm_int *ADAir_out) {
  MC_TRY
  AdaSlot &S = g_s2;
  const u64 pa = intern(m, ADAjc, ADAir), pq = intern(m, Qjc, Qir);
  std::vector<sdm_int> ints = {m, lorN};
  if (!(S.built && S.pat_ada == pa && S.pat_q == pq && S.ints == ints)) {
    if (!S.plan) S.plan = new_plan();
    at_exit_once();
    S.built = false; forget_plan(S.plan);
    g_stat[ST_ADA_BUILD]++;
    gw_build_getada2(S.plan, m, ADAjc, ADAir, lorN, Qjc, Qir);
    S.pat_ada = pa; S.pat_q = pq; S.ints = ints; S.perm.clear();
    S.built = true;
  } else g_stat[ST_ADA_REUSE]++;
  S.set_perm(qperm, m);
  ada_input(S.plan, ADApr_in, ADAjc[m]);
  gw_run_getada2(S.plan, S.invperm.p, Qpr);
  gw_download(S.plan, ADApr, nullptr);
  returned(S.plan, ADApr, ADAjc[m]);
  alias(pa, ADAir_out);
  MC_CATCH
}
// the copy getada2 returns when there is nothing to add (getada2.c:154-155): the device copy stays the current one
void sdm_mexcache_getada2_passthrough(sdm_int nnz, const double *ADApr_in, const double *ADApr_out, const sdm_int *ADAir_in, const sdm_int *ADAir_out) {
  if (g_last.plan && g_last.vals.same(ADApr_in, nnz)) g_last.vals.seen_at(ADApr_out);
  // (the copy's row indices too: a pattern too big for the full checksum is only recognised at an address the cache knows -- with an
  // allocator that does not hand out the same address every iteration getada3 took the copy for a new pattern and rebuilt its analysis)
  if (ADAir_in && ADAir_out)
    for (auto &p : g_pat) if (p.id && p.ir.n == nnz && p.ir.same(ADAir_in, nnz)) { p.ir.seen_at(ADAir_out); break; }
}

// [ADA, absd] = getada3(ADA, A, Ajc1, Aord, udsqr, K)
int sdm_mexcache_getada3(sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir, const double *ADApr_in, double *ADApr, sdm_int N,
                         const sdm_int *Ajc, const sdm_int *Air, const double *Apr, const sdm_int *Ajc1, const double *udsqr,
                         const sdm_cone *K, const sdm_int *psd_blkstart, double *absd, const sdm_int *ADAir_out) {
  MC_TRY
  AdaSlot &S = g_s3;
  const u64 pa = intern(m, ADAjc, ADAir), pA = intern(m, Ajc, Air);
  std::vector<sdm_int> ints = {m, N, K->lorN, K->sdpN, K->rsdpN};
  ints.insert(ints.end(), K->sdpNL, K->sdpNL + K->sdpN);
  ints.insert(ints.end(), psd_blkstart, psd_blkstart + K->sdpN + (K->sdpN > 0 ? 1 : 0));
  if (!(S.built && S.pat_ada == pa && S.pat_a == pA && S.ints == ints && vec_is(S.split, Ajc1, m) && S.apr.same(Apr, Ajc[m]))) {
    if (!S.plan) S.plan = new_plan();
    at_exit_once();
    S.built = false; forget_plan(S.plan);
    g_stat[ST_ADA_BUILD]++; g_stat[ST_AT_UPLOAD]++;
    gw_build_getada3(S.plan, m, ADAjc, ADAir, N, Ajc, Air, Apr, Ajc1, K, psd_blkstart);
    S.pat_ada = pa; S.pat_a = pA; S.ints = ints; S.split.assign(Ajc1, Ajc1 + m); S.apr.take(Apr, Ajc[m]);
    S.built = true;
  } else g_stat[ST_ADA_REUSE]++;
  ada_input(S.plan, ADApr_in, ADAjc[m]);
  gw_run_getada3(S.plan, udsqr);
  gw_download(S.plan, ADApr, absd);
  returned(S.plan, ADApr, ADAjc[m]);
  alias(pa, ADAir_out);
  MC_CATCH
}

// absd = getada(A, K, d, DAt) [global ADA_sedumi_]: the whole ADA' of a problem without PSD blocks (getada.m:13-40)
int sdm_mexcache_getada(sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir, sdm_int N, const sdm_int *Ajc, const sdm_int *Air,
                        const double *Apr, sdm_int lpN, const double *dl, sdm_int lorN, const double *ddet, const sdm_int *qblkstart,
                        const sdm_int *Qjc, const sdm_int *Qir, const double *Qpr, double *ADApr, double *absd, const sdm_int *ADAir_out) {
  MC_TRY
  AdaSlot &S = g_s0;
  const u64 pa = intern(m, ADAjc, ADAir), pA = intern(m, Ajc, Air), pq = lorN > 0 ? intern(m, Qjc, Qir) : 0;
  std::vector<sdm_int> ints = {m, N, lpN, lorN};
  ints.insert(ints.end(), qblkstart, qblkstart + (lorN > 0 ? lorN + 1 : 0));
  if (!(S.built && S.pat_ada == pa && S.pat_a == pA && S.pat_q == pq && S.ints == ints && S.apr.same(Apr, Ajc[m]))) {
    if (!S.plan) S.plan = new_plan();
    at_exit_once();
    S.built = false; forget_plan(S.plan);
    g_stat[ST_ADA_BUILD]++; g_stat[ST_AT_UPLOAD]++;
    gw_build_getada(S.plan, m, ADAjc, ADAir, Ajc, Air, Apr, lpN, lorN, qblkstart, Qjc, Qir);
    S.pat_ada = pa; S.pat_a = pA; S.pat_q = pq; S.ints = ints; S.apr.take(Apr, Ajc[m]);
    S.built = true;
  } else g_stat[ST_ADA_REUSE]++;
  gw_run_getada(S.plan, dl, ddet, Qpr);
  gw_download(S.plan, ADApr, absd);
  returned(S.plan, ADApr, ADAjc[m]);
  alias(pa, ADAir_out);
  MC_CATCH
}

// [L.L, L.d, L.skip, L.add] = blkchol(L, X, pars, absd) on the cache (arguments of sdm_blkchol; Lir_out: the row indices of the
// L.L the shim returns, a copy of Lir, or NULL).  The factor stays resident for the solves.
int sdm_mexcache_blkchol(sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const sdm_int *perm, sdm_int nsuper, const sdm_int *xsuper,
                         const sdm_int *Xjc, const sdm_int *Xir, const double *Xpr, const sdm_cholpars *pars, const double *absd,
                         double *Lpr, double *d, sdm_int *nskip, sdm_int *skip_idx, double *skip_val, sdm_int *nadd, sdm_int *add_idx,
                         double *add_val, const sdm_int *Lir_out) {
  MC_TRY
  sdm_plan *p = chol_plan(m, Ljc, Lir, perm, nsuper, xsuper, Xjc, Xir);
  g.have_factor = false;             // the resident factor is about to be overwritten: whatever the solves are handed before this call has returned is not it
  const sdm_int nnzX = Xjc[m], nnzL = Ljc[m];
  if (g_last.plan && g_last.vals.same(Xpr, nnzX)) {
    g_stat[ST_X_RESIDENT]++;
    if (g_last.plan != p)
      SDM_HIP_CHECK(hipMemcpyAsync(p->ada_val.p, g_last.plan->ada_val.p, (size_t)nnzX * sizeof(double), hipMemcpyDeviceToDevice, p->stream));
  } else {
    g_stat[ST_X_UPLOAD]++;
    SDM_HIP_CHECK(hipMemcpyAsync(p->ada_val.p, Xpr, (size_t)nnzX * sizeof(double), hipMemcpyHostToDevice, p->stream));
  }
  if (absd) SDM_HIP_CHECK(hipMemcpyAsync(p->absd.p, absd, (size_t)m * sizeof(double), hipMemcpyHostToDevice, p->stream));
  if (sdm_plan_blkchol_wait(p, pars, absd ? 1 : 0)) throw std::runtime_error(sdm_last_error());   // (waited for, repeated once on the launch-per-panel path after a time-out)
  // L.L values, L.d and the fingerprint of the factor (summed on the device: no pass over the host copy) in one drain of the stream
  u64 acc[8];
  if (!g.ck.p) g.ck.alloc(8);
  chol_extract(p, p->lpr.p);
  SDM_HIP_CHECK(hipMemsetAsync(g.ck.p, 0, 8 * sizeof(u64), p->stream));
  SDM_LAUNCH(k_words_checksum, dim3(64), dim3(256), 0, p->stream, (const unsigned long long *)p->lpr.p, (long long)nnzL, g.ck.p);
  SDM_HIP_CHECK(hipMemcpyAsync(Lpr, p->lpr.p, (size_t)nnzL * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  SDM_HIP_CHECK(hipMemcpyAsync(d, p->chol.d.p, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  SDM_HIP_CHECK(hipMemcpyAsync(acc, g.ck.p, 8 * sizeof(u64), hipMemcpyDeviceToHost, p->stream));
  if (sdm_plan_pivots(p, nskip, skip_idx, skip_val, nadd, add_idx, add_val)) throw std::runtime_error(sdm_last_error());   // (drains the stream)
  g.lpr.take_with_full(Lpr, nnzL, fold_sums(acc, nnzL));
  g.have_factor = true;
  alias(g.pat_l, Lir_out);
  MC_CATCH
}

// y = fwblkslv(L, b) / bwblkslv(L, b), dense b: on the resident factor when the L.L values handed over ARE the factor the last
// blkchol left on the device (content fingerprint), else stateless (sdm_fwblkslv / sdm_bwblkslv)
int sdm_mexcache_solve(int fw, sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const double *Lpr, const sdm_int *perm, sdm_int nsuper,
                       const sdm_int *xsuper, sdm_int nrhs, const double *b, double *y) {
  MC_TRY
  bool hit = g.plan && g.have_factor && vec_is(g.xsuper, xsuper, nsuper + 1) && (!perm || vec_is(g.perm, perm, m));
  hit = hit && intern(m, Ljc, Lir) == g.pat_l && g.lpr.same(Lpr, Ljc[m]);
  if (!hit) {
    g_stat[ST_SOLVE_STATELESS]++;
    if ((fw ? sdm_fwblkslv : sdm_bwblkslv)(m, Ljc, Lir, Lpr, perm, nsuper, xsuper, nrhs, b, y)) throw std::runtime_error(sdm_last_error());
    return 0;
  }
  g_stat[ST_SOLVE_RESIDENT]++;
  sdm_plan *p = g.plan;
  for (sdm_int c = 0; c < nrhs; c++) {
    SDM_HIP_CHECK(hipMemcpyAsync(p->rhs.p, b + c * m, (size_t)m * sizeof(double), hipMemcpyHostToDevice, p->stream));
    if (fw ? sdm_plan_fwsolve(p) : sdm_plan_bwsolve(p)) throw std::runtime_error(sdm_last_error());
    SDM_HIP_CHECK(hipMemcpyAsync(y + c * m, p->y.p, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    SDM_HIP_CHECK(hipStreamSynchronize(p->stream));
  }
  if (sdm_plan_sync(p)) throw std::runtime_error(sdm_last_error());     // (time-out flags of the plan)
  MC_CATCH
}

// ---- the round-3 entry points, kept for callers that drive the cached plan themselves
sdm_plan *sdm_mexcache_plan(sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const sdm_int *perm, sdm_int nsuper,
                            const sdm_int *xsuper, const sdm_int *Xjc, const sdm_int *Xir) {
  try { SDM_HIP_CHECK(hipSetDevice(device())); return chol_plan(m, Ljc, Lir, perm, nsuper, xsuper, Xjc, Xir); }
  catch (const std::exception &e) { set_error(e.what()); return nullptr; }
}
void sdm_mexcache_remember_factor(const double *Lpr_host, sdm_int nnz) {
  if (!g.plan || !Lpr_host || nnz != g.plan->chol.nnzL) { g.have_factor = false; return; }      // (NULL: invalidate -- a refactorisation is starting)
  g.have_factor = true;
  g.lpr.take(Lpr_host, nnz);
}
sdm_plan *sdm_mexcache_factor_plan(sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const double *Lpr, const sdm_int *perm,
                                   sdm_int nsuper, const sdm_int *xsuper) {
  if (!g.plan || !g.have_factor || !vec_is(g.xsuper, xsuper, nsuper + 1)) return nullptr;
  if (perm && !vec_is(g.perm, perm, m)) return nullptr;
  if (intern(m, Ljc, Lir) != g.pat_l) return nullptr;
  if (!g.lpr.same(Lpr, Ljc[m])) return nullptr;                          // content differs: not the resident factor
  return g.plan;
}

}  // extern "C"
