// sdm_mexcache.cpp -- process-wide cache of ONE resident plan for the mexFunction shims (INTEGRATION.md "Keeping data
// on the device across MEX calls").  Every .mex binary is its own shared object with its own statics, so a cache
// kept inside the shims would never be shared between blkchol.mex and fwblkslv.mex; libsedumi_hip.so is loaded
// once per process, so the cache lives here, behind the C ABI.
//
// Residency of a factor is never assumed from the host address of L.L alone (MATLAB may free that array and hand
// the same address to another L.L of the same shape): blkchol records a content fingerprint of the values it
// returned, and the solves present the values they were given.
#include "../../include/sedumi_hip.h"
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace sdm { void set_error(const std::string &msg); }

namespace {
typedef unsigned long long u64;
inline u64 mix(u64 h, u64 v) { h ^= v; h *= 1099511628211ull; return h; }
u64 hash_idx(const sdm_int *v, sdm_int n) {
  u64 h = 1469598103934665603ull;
  for (sdm_int i = 0; i < n; i++) h = mix(h, (u64)v[i]);
  return h;
}
// bit patterns of at most `cap` evenly spaced values (always the first and the last one)
u64 hash_vals(const double *v, sdm_int n, sdm_int cap) {
  u64 h = 1469598103934665603ull;
  if (n <= 0) return h;
  const sdm_int step = n <= cap ? 1 : (n + cap - 1) / cap;
  for (sdm_int i = 0; i < n; i += step) { u64 b; memcpy(&b, v + i, 8); h = mix(h, b); }
  u64 b; memcpy(&b, v + n - 1, 8);
  return mix(h, b);
}
constexpr sdm_int SAMPLE = 1 << 14;        // sampled fingerprint: 16K values
constexpr sdm_int FULL_MAX = 1 << 22;      // full fingerprint only up to 4M values (32 MB), ~4 ms on the host

struct Cache {
  sdm_plan *plan = nullptr;
  sdm_int m = -1, nnzL = -1, nsuper = -1, nnzX = -1;
  u64 hperm = 0, hxs = 0, hLjc = 0, hLir = 0, hXjc = 0, hXir = 0;
  // the factor the plan holds
  bool have_factor = false;
  const double *Lpr = nullptr;
  u64 fp_sample = 0, fp_full = 0;
  bool have_full = false;
} g;

void drop() {
  if (g.plan) sdm_plan_destroy(g.plan);
  g = Cache();
}
}  // namespace

extern "C" {

void sdm_mexcache_clear(void) { drop(); }

sdm_plan *sdm_mexcache_plan(sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const sdm_int *perm, sdm_int nsuper,
                            const sdm_int *xsuper, const sdm_int *Xjc, const sdm_int *Xir) {
  const sdm_int nnzL = Ljc[m], nnzX = Xjc[m];
  const u64 hp = hash_idx(perm, m), hx = hash_idx(xsuper, nsuper + 1), hj = hash_idx(Ljc, m + 1), hi = hash_idx(Lir, nnzL),
            hxj = hash_idx(Xjc, m + 1), hxi = hash_idx(Xir, nnzX);
  if (g.plan && g.m == m && g.nnzL == nnzL && g.nsuper == nsuper && g.nnzX == nnzX && g.hperm == hp && g.hxs == hx &&
      g.hLjc == hj && g.hLir == hi && g.hXjc == hxj && g.hXir == hxi)
    return g.plan;
  drop();
  const char *dev = getenv("SEDUMI_HIP_DEVICE");
  g.plan = sdm_plan_create(dev ? atoi(dev) : 0, nullptr);
  if (!g.plan) return nullptr;
  // the cached plan owns streams, events and device memory: it goes before the HIP runtime's own exit handlers run (registered
  // by the first HIP call of the process -- at the latest the ones above -- so this one, registered after them, runs first).
  // A plan left alive across process exit crashed there (r03g: segmentation fault after the last test of a process that
  // had called blkchol.mex, when a late mexAtExit teardown reached sdm_plan_destroy behind the runtime's own shutdown).
  static bool at_exit = false;
  if (!at_exit) { at_exit = true; std::atexit([] { drop(); }); }
  if (sdm_plan_set_chol(g.plan, m, Ljc, Lir, perm, nsuper, xsuper, Xjc, Xir)) { drop(); return nullptr; }
  g.m = m; g.nnzL = nnzL; g.nsuper = nsuper; g.nnzX = nnzX;
  g.hperm = hp; g.hxs = hx; g.hLjc = hj; g.hLir = hi; g.hXjc = hxj; g.hXir = hxi;
  return g.plan;
}

void sdm_mexcache_remember_factor(const double *Lpr_host, sdm_int nnz) {
  if (!g.plan || !Lpr_host || nnz != g.nnzL) { g.have_factor = false; return; }      // (NULL: invalidate -- a refactorisation is starting)
  g.have_factor = true;
  g.Lpr = Lpr_host;
  g.fp_sample = hash_vals(Lpr_host, nnz, SAMPLE);
  g.have_full = nnz <= FULL_MAX;
  g.fp_full = g.have_full ? hash_vals(Lpr_host, nnz, nnz) : 0;
}

sdm_plan *sdm_mexcache_factor_plan(sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const double *Lpr, const sdm_int *perm,
                                   sdm_int nsuper, const sdm_int *xsuper) {
  if (!g.plan || !g.have_factor || g.m != m || g.nsuper != nsuper || g.nnzL != Ljc[m]) return nullptr;
  if (g.hxs != hash_idx(xsuper, nsuper + 1) || g.hLjc != hash_idx(Ljc, m + 1)) return nullptr;
  if (perm && g.hperm != hash_idx(perm, m)) return nullptr;
  if (g.hLir != hash_idx(Lir, Ljc[m])) return nullptr;
  if (g.fp_sample != hash_vals(Lpr, Ljc[m], SAMPLE)) return nullptr;      // content differs: not the resident factor
  // same sampled content: accept the very array blkchol returned; any other array must match in full
  if (Lpr == g.Lpr) return g.plan;
  if (g.have_full && g.fp_full == hash_vals(Lpr, Ljc[m], Ljc[m])) return g.plan;
  return nullptr;
}

}  // extern "C"
