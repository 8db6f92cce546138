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
// Residency is decided by CONTENT: an array is taken for the one the device holds only if a checksum over EVERY word of what the
// host array holds now equals the checksum of the resident data (for arrays a gateway produced: summed on the device over what
// it left in HBM, k_words_checksum).  The one shortcut, and its limit:
//   * arrays of up to `full_below` words (default 65 536 = 512 KB, ~15 us) are checksummed completely at EVERY presentation;
//   * a larger array is checksummed completely the first time an address presents it in an EPOCH (one epoch = the interval
//     between two blkchol calls, i.e. one IPM iteration); further presentations AT THAT ADDRESS IN THAT EPOCH are accepted on
//     length + a sampled hash (SAMPLE evenly spaced words + first and last).  An address is trusted only for the epoch in
//     which this cache itself READ every word there and found the resident content.  Not trusted: the output buffer a gateway has
//     just written (the host may copy the result and free the buffer -- Octave and this package's own MEX host do -- and hand the
//     address to another array; GPUTEST r04 / r05a failed on exactly that); nothing is remembered across epochs.
// Limit of the shortcut (stated in DESIGN.md 1a / INTEGRATION.md, asserted by tests/test_mexshims.py): a large array that is
// edited IN PLACE -- or freed and replaced at the same address by an array of the same length -- at words the sample does not
// touch, between two presentations inside one epoch, is still taken for the resident one.  sedumi.m never does that
// (sedumi.m:450-462, wrapPcg.m:56-59 hand the gateways' outputs on untouched).  sdm_mexcache_set_strict(1) removes the shortcut:
// every presentation is checksummed completely, whatever the size.
#include "sdm_plan.h"
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace sdm;

namespace {
typedef unsigned long long u64;
constexpr sdm_int SAMPLE = 512;            // words of the sampled hash
constexpr sdm_int THREADS_FROM = 1 << 20;  // arrays from 1M words (8 MB) on are checksummed by several host threads
constexpr u64 P1 = 11400714785074694791ull, P2 = 14029467366897019727ull;
inline u64 rotl(u64 x, int r) { return (x << r) | (x >> (64 - r)); }
inline u64 lane(u64 acc, u64 v) { return rotl(acc + v * P2, 31) * P1; }
// every word, position dependent, as eight independent wrap-around sums of 32 x 32 -> 64 bit products: the compiler vectorises
// it on the host, and -- sums commute -- the same value comes out of a parallel reduction on the device (k_words_checksum:
// what a gateway leaves in HBM is fingerprinted there instead of by a pass over the host copy) or of several host threads.
// A change detector, not a cryptographic hash: any single changed word changes its sum.
#define SDM_CK_C1 {0x9E3779B1u, 0x85EBCA77u, 0xC2B2AE3Du, 0x27D4EB2Fu, 0x165667B1u, 0xD3A2646Du, 0xFD7046C5u, 0xB55A4F09u}
#define SDM_CK_C2 {0x8DA6B343u, 0xD8163841u, 0xCB1AB31Fu, 0x9F6B3F4Bu, 0xA54FF53Bu, 0x3C6EF373u, 0xBB67AE85u, 0x6A09E667u}
u64 fold_sums(const u64 *acc, sdm_int n) {
  u64 h = (u64)n;
  for (int k = 0; k < 8; k++) h = lane(h, acc[k]);
  return h;
}
// the eight sums over the words [i0, i1) of v (i0 a multiple of 8), added into acc; one body, compiled for the baseline ISA and for
// AVX2 / AVX-512 (the host of a GPU box has them; picked once at run time)
#define SDM_CK_BODY                                                                                                          \
  static const u64 C1[8] = SDM_CK_C1, C2[8] = SDM_CK_C2;                                                                     \
  u64 a[8] = {0, 0, 0, 0, 0, 0, 0, 0};                                                                                       \
  sdm_int i = i0;                                                                                                            \
  for (; i + 8 <= i1; i += 8)                                                                                                \
    for (int k = 0; k < 8; k++) {                                                                                            \
      const u64 w = v[i + k];                                                                                                \
      a[k] += (u64)((unsigned)w ^ (unsigned)i) * C1[k] + (u64)((unsigned)(w >> 32) ^ (unsigned)i) * C2[k];                   \
    }                                                                                                                        \
  for (int k = 0; i + k < i1; k++) {                                                                                         \
    const u64 w = v[i + k];                                                                                                  \
    a[k] += (u64)((unsigned)w ^ (unsigned)i) * C1[k] + (u64)((unsigned)(w >> 32) ^ (unsigned)i) * C2[k];                     \
  }                                                                                                                          \
  for (int k = 0; k < 8; k++) acc[k] += a[k];
void sums_base(const u64 *v, sdm_int i0, sdm_int i1, u64 *acc) { SDM_CK_BODY }
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__) && !defined(SDM_NO_MULTIVERSION)
__attribute__((target("avx2"))) void sums_avx2(const u64 *v, sdm_int i0, sdm_int i1, u64 *acc) { SDM_CK_BODY }
__attribute__((target("avx512f,avx512dq,avx512vl"))) void sums_avx512(const u64 *v, sdm_int i0, sdm_int i1, u64 *acc) { SDM_CK_BODY }
typedef void (*sums_fn)(const u64 *, sdm_int, sdm_int, u64 *);
sums_fn pick_sums() {
  __builtin_cpu_init();
  if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl")) return sums_avx512;
  if (__builtin_cpu_supports("avx2")) return sums_avx2;
  return sums_base;
}
const sums_fn sums = pick_sums();
#else
#define sums sums_base
#endif
void pool_shutdown();
// A few helper threads for the checksums of arrays from POOL_FROM words on: what a gateway is handed has usually just been written by
// the DMA engine (or by a memcpy with streaming stores) and comes from DRAM at 8 - 10 GB/s through one core.  The helpers are created
// at the first such array, sleep on a condition variable in between (woken in ~10 us) and are joined when the cache is torn down.
constexpr sdm_int POOL_FROM = 1 << 17;     // 128K words = 1 MB
class SumPool {
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv_go, cv_done;
  const u64 *v = nullptr;
  sdm_int n = 0, chunk = 0;
  int nparts = 0, pending = 0;
  u64 gen = 0;
  bool stop = false;
  u64 part[16][8];
  void worker(int id, u64 seen) {                              // seen: the generation current when the helper was created (by the one thread that advances it)
    for (;;) {
      std::unique_lock<std::mutex> lk(mu);
      cv_go.wait(lk, [&] { return stop || gen != seen; });
      if (stop) return;
      seen = gen;
      const bool mine = id < nparts;
      const u64 *vv = v; const sdm_int i0 = std::min(n, (sdm_int)id * chunk), i1 = std::min(n, i0 + chunk);
      lk.unlock();
      if (mine) { for (int k = 0; k < 8; k++) part[id][k] = 0; sums(vv, i0, i1, part[id]); }
      lk.lock();
      if (mine && --pending == 0) cv_done.notify_one();
    }
  }
 public:
  // the eight sums of v[0, n) into acc, by `want` threads (the caller is one of them)
  void run(const u64 *vp, sdm_int len, int want, u64 *acc) {
    want = std::max(1, std::min(want, 16));
    if (th.empty()) { static bool reg = false; if (!reg) { reg = true; std::atexit([] { pool_shutdown(); }); } }   // sleeping helpers are joined before the process ends
    while ((int)th.size() < want - 1) { const int id = (int)th.size() + 1; const u64 g0 = gen; th.emplace_back([this, id, g0] { worker(id, g0); }); }
    const sdm_int ch = ((len + want - 1) / want + 7) & ~(sdm_int)7;
    {
      std::lock_guard<std::mutex> lk(mu);
      v = vp; n = len; chunk = ch; nparts = want; pending = want - 1; gen++;
    }
    cv_go.notify_all();
    for (int k = 0; k < 8; k++) part[0][k] = 0;
    sums(vp, 0, std::min(len, ch), part[0]);
    {
      std::unique_lock<std::mutex> lk(mu);
      cv_done.wait(lk, [&] { return pending == 0; });
    }
    for (int t = 0; t < want; t++) for (int k = 0; k < 8; k++) acc[k] += part[t][k];
  }
  void shutdown() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv_go.notify_all();
    for (auto &t : th) t.join();
    th.clear(); stop = false;
  }
  ~SumPool() { shutdown(); }
} g_pool;
void pool_shutdown() { g_pool.shutdown(); }
int g_threads = -1;        // sdm_mexcache_set_threads: threads of a big checksum (-1: 4 from POOL_FROM words, up to 8 from 1M; never more than the host has)
sdm_int g_hash_ns = 0, g_hash_calls = 0;      // host time spent in complete checksums since the last clear (stats[13], [14])
u64 hash_full(const void *pv, sdm_int n) {
  const u64 *v = (const u64 *)pv;
  u64 acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const auto t0 = std::chrono::steady_clock::now();
  int nt = 1;
  if (n >= POOL_FROM) {
    const int hw = (int)std::thread::hardware_concurrency();
    nt = g_threads >= 0 ? g_threads : (n >= THREADS_FROM ? 8 : 4);
    nt = std::max(1, std::min(nt, std::max(1, hw)));
  }
  if (nt <= 1) sums(v, 0, n, acc);
  else g_pool.run(v, n, nt, acc);
  g_hash_ns += (sdm_int)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); g_hash_calls++;
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
// Checksums a shim has just computed itself while it copied an array (sdm_mexcache_copy_words: the pattern of the array it returns is a copy
// of its input's -- one pass over the input for both): valid for the gateway call in progress only, dropped when it returns.
struct Note { const void *p; sdm_int n; u64 sum; };
Note g_notes[4];
int g_nnotes = 0;
void drop_notes() { g_nnotes = 0; }
bool strict = false;            // sdm_mexcache_set_strict: no shortcut, every presentation is checksummed completely
sdm_int full_below = 1 << 16;   // sdm_mexcache_set_full_below: arrays up to this many words are checksummed completely at every presentation
u64 g_epoch = 1;                // advances with every blkchol call: the lifetime of a trusted address (file header)
sdm_int g_fullsum_words = 0;    // words checksummed on the host since the last sdm_mexcache_clear (the cost of the content checks; stats[11])

// What is known about one array the device holds a copy of (or the analysis of): its length, the sampled hash, the checksum of every
// word, and the addresses at which THIS epoch every word has been read (or written) by the cache itself.
struct Finger {
  sdm_int n = -1;
  u64 sample = 0, full = 0;
  struct { const void *p; u64 epoch; } ok[8] = {};
  int nok = 0;
  bool trusted(const void *p) const {
    for (int i = 0; i < 8; i++) if (p && ok[i].p == p && ok[i].epoch == g_epoch) return true;
    return false;
  }
  // every word at p was read (or written) by the cache in this epoch and is this content
  void verified(const void *p) {
    if (!p || trusted(p)) return;
    for (int i = 0; i < 8; i++) if (ok[i].p == p) { ok[i].epoch = g_epoch; return; }
    ok[nok & 7].p = p; ok[nok++ & 7].epoch = g_epoch;
  }
  void take(const void *p, sdm_int len) {
    n = len; sample = hash_sampled(p, len); full = hash_full(p, len); g_fullsum_words += len;
    for (auto &a : ok) a.p = nullptr;
    nok = 0; verified(p);
  }
  // the checksum computed elsewhere: on the device copy of the values that have just been downloaded to p.  No address is trusted
  // yet: whoever presents this content has it read completely once (file header)
  void take_with_full(const void *p, sdm_int len, u64 fullhash) {
    n = len; sample = hash_sampled(p, len); full = fullhash;
    for (auto &a : ok) a.p = nullptr;
    nok = 0;
  }
  bool same(const void *p, sdm_int len) {
    if (len != n || n < 0) return false;
    if (!strict && len > full_below && trusted(p)) return hash_sampled(p, len) == sample;   // the shortcut of the file header
    u64 sum = 0;
    bool noted = false;
    for (int i = 0; i < g_nnotes; i++) if (g_notes[i].p == p && g_notes[i].n == len) { sum = g_notes[i].sum; noted = true; }
    if (!noted) { g_fullsum_words += len; sum = hash_full(p, len); }
    if (sum != full) return false;
    verified(p);
    return true;
  }
  void forget() { n = -1; }
};

// ---- index patterns (CSC jc / ir pairs): interned, so that the slots below compare small integers.  An id is never reused.
struct Pattern { sdm_int ncol = -1; Finger jc, ir; u64 id = 0, used = 0; std::vector<sdm_int> jcc, irc; };   // (jcc / irc: a host copy, lazy mode only)
Pattern g_pat[12];
u64 g_next_id = 1, g_clock = 0;
// ---- lazy intermediates (SEDUMI_HIP_LAZY = 0 | 1 | 2, or sdm_mexcache_set_lazy; default 2 since round 6).  sedumi.m:450-458 hands ADA' from getada1 to getada2
// to getada3 to blkchol and never looks at it.  Level 1: getada1.mex / getada2.mex return, instead of ADA' (values + a copy of the pattern,
// 2 x 128 MB for MAXCUT-4000, each read again by the next gateway's content check), a TOKEN: an m x m sparse matrix with the one nonzero
// (1,1) = LAZY_BASE + a serial number that is never reused; the values stay on the device.  A gateway handed a token takes the
// device's ADA' if the token is the current one and fails loudly otherwise (there are no values to fall back on).  getada3.mex
// materialises: its ADA' and absd are the reference's.  Level 2: getada3.mex returns a token as well (absd is real) and blkchol.mex
// factors the device's ADA' -- what crosses PCIe per iteration is then absd, L.L, L.d, the pivot lists and the solves' vectors.
constexpr double LAZY_BASE = 6755399441055744.0;             // 1.5 * 2^52: token = LAZY_BASE + serial, exact in a double
int g_lazy = -1;                                             // -1: not read from the environment yet
u64 g_serial = 0;
struct { u64 serial = 0, pat = 0; sdm_int m = 0, nnz = 0; } g_tok;      // the current token: which pattern the device's ADA' (g_last.plan) has
int lazy_level() {
  // default (round 6): level 2 -- every call site of the reference hands the array on untouched (sedumi.m:450-458, optstep.m:68-76; symbchol.m:62-73
  // and sedumi.m:401 read the global before the first getada1): tests/test_mexshims.py runs whole solves through the shims at levels 0 and 2
  // and compares the logs.  SEDUMI_HIP_LAZY=0 gives the reference's arrays back at every gateway.
  if (g_lazy < 0) { const char *e = getenv("SEDUMI_HIP_LAZY"); g_lazy = e ? std::max(0, std::min(2, atoi(e))) : 2; }
  return g_lazy;
}
// the tokens handed out last (serial -> pattern): getada1.mex needs only the PATTERN of the ADA' it is given (it starts from zero, getada1.c:222-225),
// so it also accepts a token that is no longer current -- sedumi.m's global still holds the token of ITS last getada3 when optstep.m has run the
// three gateways on its own copy in between
struct TokRec { u64 serial = 0, pat = 0; sdm_int m = 0, nnz = 0; };
TokRec g_tok_hist[16];
void remember_token(u64 serial, u64 pat, sdm_int m, sdm_int nnz) { TokRec &r = g_tok_hist[serial & 15]; r.serial = serial; r.pat = pat; r.m = m; r.nnz = nnz; }
bool is_token(double t) { return t > LAZY_BASE && t < LAZY_BASE + 4294967296.0; }
Pattern *pattern_by_id(u64 id) { for (auto &p : g_pat) if (p.id == id) return &p; return nullptr; }
u64 intern(sdm_int ncol, const sdm_int *jc, const sdm_int *ir) {
  const sdm_int nnz = jc[ncol];
  for (auto &p : g_pat)
    if (p.id && p.ncol == ncol && p.jc.same(jc, ncol + 1) && p.ir.same(ir, nnz)) {
      p.used = ++g_clock;
      if (lazy_level() > 0 && p.irc.empty() && nnz > 0) { p.jcc.assign(jc, jc + ncol + 1); p.irc.assign(ir, ir + nnz); }
      return p.id;
    }
  Pattern *v = &g_pat[0];
  for (auto &p : g_pat) if (p.used < v->used) v = &p;                 // least recently used (empty entries first: used = 0)
  v->ncol = ncol; v->jc.take(jc, ncol + 1); v->ir.take(ir, nnz); v->id = g_next_id++; v->used = ++g_clock;
  v->jcc.clear(); v->irc.clear();
  if (lazy_level() > 0) { v->jcc.assign(jc, jc + ncol + 1); v->irc.assign(ir, ir + nnz); }
  return v->id;
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
struct { sdm_plan *plan = nullptr; Finger vals; bool zero = false; } g_last;   // zero: those values are known to be all zero (getada1 with nothing to add)
DevBuf<u64> g_ck;                       // eight device words: checksum of values a gateway leaves in HBM (k_words_checksum)
double *g_pin = nullptr;                // pinned staging for the solves' right-hand side and solution (sdm_mexcache_solve)
sdm_int g_pin_n = 0;
u64 *g_ck_host = nullptr;               // ... and where they land on the host: pinned, so that the copy is queued like the rest and the one
                                        // synchronisation of the gateway (its download) covers it (to pageable memory every copy blocked the host)
// the eight sums of n device words -> g_ck_host, queued on the plan's stream: valid once the stream has been drained
void device_checksum(sdm_plan *p, const double *v, sdm_int n) {
  if (!g_ck.p) g_ck.alloc(8);
  if (!g_ck_host) SDM_HIP_CHECK(hipHostMalloc((void **)&g_ck_host, 8 * sizeof(u64), 0));
  const sdm_int wg = std::min<sdm_int>(2048, std::max<sdm_int>(64, n / 8192));
  SDM_HIP_CHECK(hipMemsetAsync(g_ck.p, 0, 8 * sizeof(u64), p->stream));
  SDM_LAUNCH(k_words_checksum, dim3((unsigned)wg), dim3(256), 0, p->stream, (const unsigned long long *)v, (long long)n, g_ck.p);
  SDM_HIP_CHECK(hipMemcpyAsync(g_ck_host, g_ck.p, 8 * sizeof(u64), hipMemcpyDeviceToHost, p->stream));
}
// ADA' values and absd of plan p -> the host arrays the shim returns, together with the checksum of the values summed on the device:
// the array at pr is, from now on, known to hold what p->ada_val holds
void download_returned(sdm_plan *p, double *pr, double *absd, sdm_int nnz) {
  device_checksum(p, p->ada_val.p, nnz);
  gw_download(p, pr, absd);                                           // (drains the stream)
  g_last.plan = p; g_last.vals.take_with_full(pr, nnz, fold_sums(g_ck_host, nnz)); g_last.zero = false;
  g_tok.serial = 0;                                                   // real arrays went out: a token issued earlier is stale from here on (refused loudly)
}
// a gateway leaves its ADA' on the device and hands out a token for it (lazy mode)
double returned_token(sdm_plan *p, u64 pat, sdm_int m, sdm_int nnz) {
  SDM_HIP_CHECK(hipStreamSynchronize(p->stream));
  g_last.plan = p; g_last.vals.forget(); g_last.zero = false;
  g_tok.serial = ++g_serial; g_tok.pat = pat; g_tok.m = m; g_tok.nnz = nnz;
  remember_token(g_tok.serial, pat, m, nnz);
  return LAZY_BASE + (double)g_tok.serial;
}
// the pattern id of the ADA' a token stands for; throws unless it is the current token
u64 token_pattern(double tok, sdm_int m) {
  if (!g_last.plan || !is_token(tok) || tok != LAZY_BASE + (double)g_tok.serial || g_tok.m != m)
    throw std::runtime_error("lazy ADA token is not the current one (SEDUMI_HIP_LAZY: the array a getada gateway returned must be handed to the next gateway untouched)");
  return g_tok.pat;
}
// the pattern id behind ANY of the last sixteen tokens (getada1.mex: pattern only); throws when it is not remembered
u64 token_pattern_any(double tok, sdm_int m) {
  if (is_token(tok)) {
    const u64 serial = (u64)(tok - LAZY_BASE);
    const TokRec &r = g_tok_hist[serial & 15];
    if (serial != 0 && r.serial == serial && r.m == m) return r.pat;
  }
  throw std::runtime_error("lazy ADA token is not one of the last sixteen handed out (SEDUMI_HIP_LAZY: getada1 needs the pattern behind the token it is given)");
}
void ada_input_token(sdm_plan *p, sdm_int nnz) {
  g_stat[ST_ADA_RESIDENT]++;
  if (g_last.plan != p)
    SDM_HIP_CHECK(hipMemcpyAsync(p->ada_val.p, g_last.plan->ada_val.p, (size_t)nnz * sizeof(double), hipMemcpyDeviceToDevice, p->stream));
}
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
void forget_plan(sdm_plan *p) { if (g_last.plan == p) { g_last.plan = nullptr; g_last.vals.forget(); g_last.zero = false; g_tok.serial = 0; } }

// ---- the factorisation's plan
struct Chol {
  sdm_plan *plan = nullptr;
  u64 pat_l = 0, pat_x = 0;
  std::vector<sdm_int> perm, xsuper;
  bool have_factor = false;
  Finger lpr;
} g;

void drop_chol() {
  if (g.plan) { forget_plan(g.plan); sdm_plan_destroy(g.plan); }
  g.plan = nullptr; g.pat_l = g.pat_x = 0; g.perm.clear(); g.xsuper.clear(); g.have_factor = false; g.lpr.forget();
}
void drop_all() {
  for (AdaSlot *s : {&g_s1, &g_s2, &g_s3, &g_s0}) { if (s->plan) forget_plan(s->plan); s->drop(); }
  drop_chol();
  g_pool.shutdown();
  g_ck.release();
  if (g_ck_host) { (void)hipHostFree(g_ck_host); g_ck_host = nullptr; }
  if (g_pin) { (void)hipHostFree(g_pin); g_pin = nullptr; g_pin_n = 0; }
  for (auto &p : g_pat) p = Pattern();
  for (auto &t : g_tok_hist) t = TokRec();
  g_fullsum_words = 0; g_hash_ns = 0; g_hash_calls = 0;
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
                    const sdm_int *Xjc, const sdm_int *Xir, u64 px = 0) {
  const u64 pl = intern(m, Ljc, Lir);
  if (!px) px = intern(m, Xjc, Xir);
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
  drop_notes();                                                     \
  }                                                                 \
  catch (const std::exception &e) { drop_notes(); set_error(e.what()); return 1; } \
  catch (...) { drop_notes(); set_error("unknown error"); return 1; }             \
  return 0;
}  // namespace

extern "C" {

void sdm_mexcache_clear(void) { drop_all(); drop_notes(); }
void sdm_mexcache_forget_notes(void) { drop_notes(); }
// dst[0 .. n) = src[0 .. n) (8-byte words) and the content checksum of src in the same pass, noted for the gateway call that follows: the
// cache then does not read src a second time to check it (a shim copies the pattern of its input into the array it returns anyway)
unsigned long long sdm_mexcache_copy_words(void *dst, const void *src, sdm_int n) {
  u64 sum;
  if (n >= POOL_FROM) { memcpy(dst, src, (size_t)n * 8); sum = hash_full(src, n); g_fullsum_words += n; }
  else {
    const u64 *v = (const u64 *)src; u64 *o = (u64 *)dst;
    static const u64 C1[8] = SDM_CK_C1, C2[8] = SDM_CK_C2;
    u64 a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    sdm_int i = 0;
    for (; i + 8 <= n; i += 8)
      for (int k = 0; k < 8; k++) {
        const u64 w = v[i + k];
        o[i + k] = w;
        a[k] += (u64)((unsigned)w ^ (unsigned)i) * C1[k] + (u64)((unsigned)(w >> 32) ^ (unsigned)i) * C2[k];
      }
    for (int k = 0; i + k < n; k++) {
      const u64 w = v[i + k];
      o[i + k] = w;
      a[k] += (u64)((unsigned)w ^ (unsigned)i) * C1[k] + (u64)((unsigned)(w >> 32) ^ (unsigned)i) * C2[k];
    }
    sum = fold_sums(a, n);
  }
  if (g_nnotes < 4) g_notes[g_nnotes++] = Note{src, n, sum};
  return sum;
}
// the content checksum of n 8-byte words on the host (what residency is decided on; tests, and the bench's statement of its cost)
unsigned long long sdm_mexcache_checksum(const void *words, sdm_int n) { return hash_full(words, n); }
// on != 0: every presentation of every array is checksummed completely (no address is ever trusted) -- for callers that edit arrays
// IN PLACE between gateway calls of one iteration (sedumi.m never does); costs a pass over the host array per call
void sdm_mexcache_set_strict(int on) { strict = on != 0; }
void sdm_mexcache_stats(sdm_int *out, sdm_int n) {
  g_stat[11] = g_fullsum_words; g_stat[12] = (sdm_int)g_epoch; g_stat[13] = g_hash_ns; g_stat[14] = g_hash_calls;
  for (sdm_int i = 0; i < n && i < 16; i++) out[i] = g_stat[i];
}
// arrays of up to `words` words are checksummed completely at every presentation (default 65 536); larger ones once per address and epoch
void sdm_mexcache_set_full_below(sdm_int words) { full_below = words < 0 ? 0 : words; }
// lazy intermediates (the block comment at LAZY_BASE): level 0 | 1 | 2 (default); -1: as the environment variable SEDUMI_HIP_LAZY says (unset: 2)
void sdm_mexcache_set_lazy(int level) { g_lazy = level < 0 ? -1 : std::min(level, 2); }
int sdm_mexcache_lazy(void) { return lazy_level(); }
// m x m of the ADA' a token stands for and its number of nonzeros; returns 1 (with sdm_last_error) unless `token` is the current token
int sdm_mexcache_token_info(double token, sdm_int m, sdm_int *nnz) {
  try { token_pattern(token, m); *nnz = g_tok.nnz; return 0; } catch (const std::exception &e) { set_error(e.what()); return 1; }
}
// the pattern behind the current token into the caller's arrays (m + 1 and nnz entries)
int sdm_mexcache_token_pattern(double token, sdm_int m, sdm_int *jc_out, sdm_int *ir_out) {
  try {
    Pattern *pp = pattern_by_id(token_pattern(token, m));
    if (!pp || pp->irc.empty()) throw std::runtime_error("lazy ADA token: its pattern is not cached any more");
    memcpy(jc_out, pp->jcc.data(), pp->jcc.size() * sizeof(sdm_int)); memcpy(ir_out, pp->irc.data(), pp->irc.size() * sizeof(sdm_int));
    return 0;
  } catch (const std::exception &e) { set_error(e.what()); return 1; }
}
double sdm_mexcache_token_base(void) { return LAZY_BASE; }
// threads of the checksum of an array from 128K words on (the caller's included): -1 = automatic (4, and 8 from 1M words), 1 = none
void sdm_mexcache_set_threads(int n) { g_threads = n; }

// ADA = getada1(ADA, A, Ajc2, perm, d, blkstart) on the cache (same arguments as sdm_getada1)
int sdm_mexcache_getada1(sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir, sdm_int N, const sdm_int *Ajc, const sdm_int *Air,
                         const double *Apr, const sdm_int *Ajc2, const sdm_int *perm, sdm_int lpN, const double *dl, sdm_int lorN,
                         const double *ddet, const sdm_int *qblkstart, double *ADApr, double token_in, double *token_out) {
  // token_in != 0: the ADA' handed in is a lazy token (ADAjc / ADAir may be NULL: only its pattern matters and the cache knows it);
  // token_out != NULL: the values stay on the device and *token_out stands for them (ADApr may be NULL)
  MC_TRY
  AdaSlot &S = g_s1;
  const u64 pa = token_in != 0.0 ? token_pattern_any(token_in, m) : intern(m, ADAjc, ADAir), pA = intern(m, Ajc, Air);
  if (token_in != 0.0) { Pattern *pp = pattern_by_id(pa); if (!pp || pp->irc.empty()) throw std::runtime_error("lazy ADA token: its pattern is not cached any more"); ADAjc = pp->jcc.data(); ADAir = pp->irc.data(); }
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
  if (token_out) *token_out = returned_token(S.plan, pa, m, ADAjc[m]);
  else download_returned(S.plan, ADApr, nullptr, ADAjc[m]);
  g_last.zero = gw_getada1_is_zero(S.plan);
  MC_CATCH
}

// ADA = getada2(ADA, DAt, Aord, K): ADApr_in the values of the input array, ADApr (out) those of the copy the shim returns
int sdm_mexcache_getada2(sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir, const double *ADApr_in, double *ADApr, sdm_int lorN,
                         const sdm_int *Qjc, const sdm_int *Qir, const double *Qpr, const sdm_int *qperm, double token_in, double *token_out) {
  MC_TRY
  AdaSlot &S = g_s2;
  const u64 pa = token_in != 0.0 ? token_pattern(token_in, m) : intern(m, ADAjc, ADAir), pq = intern(m, Qjc, Qir);
  if (token_in != 0.0) { Pattern *pp = pattern_by_id(pa); if (!pp || pp->irc.empty()) throw std::runtime_error("lazy ADA token: its pattern is not cached any more"); ADAjc = pp->jcc.data(); ADAir = pp->irc.data(); }
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
  if (token_in != 0.0) ada_input_token(S.plan, ADAjc[m]); else ada_input(S.plan, ADApr_in, ADAjc[m]);
  g_last.zero = false;                                                // (Lorentz terms are being added)
  gw_run_getada2(S.plan, S.invperm.p, Qpr);
  if (token_out) *token_out = returned_token(S.plan, pa, m, ADAjc[m]);
  else download_returned(S.plan, ADApr, nullptr, ADAjc[m]);
  MC_CATCH
}
// [ADA, absd] = getada3(ADA, A, Ajc1, Aord, udsqr, K)
int sdm_mexcache_getada3(sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir, const double *ADApr_in, double *ADApr, sdm_int N,
                         const sdm_int *Ajc, const sdm_int *Air, const double *Apr, const sdm_int *Ajc1, const double *udsqr,
                         const sdm_cone *K, const sdm_int *psd_blkstart, double *absd, double token_in, double *token_out) {
  MC_TRY
  AdaSlot &S = g_s3;
  const u64 pa = token_in != 0.0 ? token_pattern(token_in, m) : intern(m, ADAjc, ADAir), pA = intern(m, Ajc, Air);
  if (token_in != 0.0) { Pattern *pp = pattern_by_id(pa); if (!pp || pp->irc.empty()) throw std::runtime_error("lazy ADA token: its pattern is not cached any more"); ADAjc = pp->jcc.data(); ADAir = pp->irc.data(); }
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
  const sdm_int up0 = g_stat[ST_ADA_UPLOAD];
  const bool was_zero = g_last.zero;
  if (token_in != 0.0) ada_input_token(S.plan, ADAjc[m]); else ada_input(S.plan, ADApr_in, ADAjc[m]);
  const bool zero_in = was_zero && g_stat[ST_ADA_UPLOAD] == up0;      // the device's ADA' was taken, and it is the zero matrix getada1 left
  g_last.zero = false;
  gw_run_getada3(S.plan, udsqr, zero_in);
  if (token_out) { gw_download(S.plan, nullptr, absd); *token_out = returned_token(S.plan, pa, m, ADAjc[m]); }
  else download_returned(S.plan, ADApr, absd, ADAjc[m]);
  MC_CATCH
}

// absd = getada(A, K, d, DAt) [global ADA_sedumi_]: the whole ADA' of a problem without PSD blocks (getada.m:13-40)
int sdm_mexcache_getada(sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir, sdm_int N, const sdm_int *Ajc, const sdm_int *Air,
                        const double *Apr, sdm_int lpN, const double *dl, sdm_int lorN, const double *ddet, const sdm_int *qblkstart,
                        const sdm_int *Qjc, const sdm_int *Qir, const double *Qpr, double *ADApr, double *absd) {
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
  download_returned(S.plan, ADApr, absd, ADAjc[m]);
  MC_CATCH
}

// [L.L, L.d, L.skip, L.add] = blkchol(L, X, pars, absd) on the cache (arguments of sdm_blkchol).  The factor stays resident for the solves.
int sdm_mexcache_blkchol(sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const sdm_int *perm, sdm_int nsuper, const sdm_int *xsuper,
                         const sdm_int *Xjc, const sdm_int *Xir, const double *Xpr, const sdm_cholpars *pars, const double *absd,
                         double *Lpr, double *d, sdm_int *nskip, sdm_int *skip_idx, double *skip_val, sdm_int *nadd, sdm_int *add_idx,
                         double *add_val, double token_in) {
  // token_in != 0: X is a lazy token (level 2: Xjc / Xir / Xpr may be NULL)
  MC_TRY
  // X is what getada3 returned in the epoch that ends here: its values and pattern are looked at before the epoch advances
  const u64 px = token_in != 0.0 ? token_pattern(token_in, m) : intern(m, Xjc, Xir);
  if (token_in != 0.0) { Pattern *pp = pattern_by_id(px); if (!pp || pp->irc.empty()) throw std::runtime_error("lazy ADA token: its pattern is not cached any more"); Xjc = pp->jcc.data(); Xir = pp->irc.data(); }
  const sdm_int nnzX = Xjc[m], nnzL = Ljc[m];
  const bool x_resident = token_in != 0.0 || (g_last.plan && g_last.vals.same(Xpr, nnzX));
  g_epoch++;                         // a new factor: every address trusted so far has to present its complete content again (file header)
  sdm_plan *p = chol_plan(m, Ljc, Lir, perm, nsuper, xsuper, Xjc, Xir, px);
  g.have_factor = false;             // the resident factor is about to be overwritten: whatever the solves are handed before this call has returned is not it
  if (x_resident) {
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
  chol_extract(p, p->lpr.p);
  device_checksum(p, p->lpr.p, nnzL);
  SDM_HIP_CHECK(hipMemcpyAsync(Lpr, p->lpr.p, (size_t)nnzL * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  SDM_HIP_CHECK(hipMemcpyAsync(d, p->chol.d.p, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  if (sdm_plan_pivots(p, nskip, skip_idx, skip_val, nadd, add_idx, add_val)) throw std::runtime_error(sdm_last_error());   // (drains the stream)
  g.lpr.take_with_full(Lpr, nnzL, fold_sums(g_ck_host, nnzL));
  g.have_factor = true;
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
    drop_notes();
    return 0;
  }
  g_stat[ST_SOLVE_RESIDENT]++;
  sdm_plan *p = g.plan;
  // right-hand side and solution travel through a pinned buffer: copies between pageable host memory and the device block the host once
  // each, and a solve of a small factor is two such copies around 10 us of kernels
  if (g_pin_n < 2 * m) {
    if (g_pin) (void)hipHostFree(g_pin);
    g_pin = nullptr; g_pin_n = 0;
    SDM_HIP_CHECK(hipHostMalloc((void **)&g_pin, (size_t)2 * m * sizeof(double), 0));
    g_pin_n = 2 * m;
  }
  for (sdm_int c = 0; c < nrhs; c++) {
    memcpy(g_pin, b + c * m, (size_t)m * sizeof(double));
    SDM_HIP_CHECK(hipMemcpyAsync(p->rhs.p, g_pin, (size_t)m * sizeof(double), hipMemcpyHostToDevice, p->stream));
    if (fw ? sdm_plan_fwsolve(p) : sdm_plan_bwsolve(p)) throw std::runtime_error(sdm_last_error());
    SDM_HIP_CHECK(hipMemcpyAsync(g_pin + m, p->y.p, (size_t)m * sizeof(double), hipMemcpyDeviceToHost, p->stream));
    SDM_HIP_CHECK(hipStreamSynchronize(p->stream));
    memcpy(y + c * m, g_pin + m, (size_t)m * sizeof(double));
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
  g_epoch++;                         // a (re)factorisation on this API too: no address stays trusted across it (file header)
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
