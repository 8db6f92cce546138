// sdm_plan.h -- internal structures of the resident plan (not part of the ABI).
#pragma once
#include "../../include/sedumi_hip.h"
#include "sdm_rt.h"
#include <mutex>
#include <string>
#include <vector>

namespace sdm {

constexpr int NB = 64;     // factor panel width (columns)
constexpr int TILE = 64;   // trailing-update tile (MFMA 4 waves x 32x32)
constexpr int UTP = 72;    // LDS pitch (doubles) of the update's operand tiles As[k][row]: the four k rows one ds_read_b64 of an MFMA operand touches start 16 banks apart (64 would put all four on the same banks)
constexpr int S1_MAXN = 96;   // PSD blocks up to this order take the matrix-core stage 1 of ADA'
constexpr int S1_KC = 48;     // slots (nonzero columns of A_jk) per GEMM chunk
constexpr int S1_WAVES = 8;   // wavefronts per task
constexpr int S1_NZ = 1536;   // nonzeros of a chunk of slots staged in LDS (bigger chunks read At directly)
constexpr int S1_MAXT = ((S1_MAXN / 16) * (S1_MAXN / 16) + S1_WAVES - 1) / S1_WAVES;   // 16x16 tiles of Z per wavefront
constexpr int S1_GEN_LDS = 74 * 1024;   // LDS target per task of the generic stage-1 kernel (bytes): two 512-work-item tasks per compute unit
constexpr int ELL_WAVES = 8;  // wavefronts per workgroup of the ELL stage-2 kernel of ADA'
constexpr int SOLVE_LDS_MAX = 3072;   // doubles of a product-form right-hand side kept in LDS (k_pr1_solve)
constexpr int FUSE_MAX_TILES = 1 << 20; // trailing updates of at most this many tiles ride along with the next diagonal-block launch (in effect: all)
constexpr int SBW_MIN = 256;          // narrowest super-block of the solves (columns whose diagonal block is applied as ONE explicit inverse)
constexpr int SBW_MAX = 2048;         // widest one (CholPlan::sbw = the power-of-two multiple of SBW_MIN that covers the widest front, capped here)
constexpr int SINV_MAXLEV = 4;        // combine levels above the 128-column leaves: half widths 128, 256, 512, 1024
constexpr int SPREP_NCNT = 2 + 2 * SINV_MAXLEV;   // completion counters per super-block of k_sprep (leaves, then T / X per level)
constexpr int SROWS = 16;             // rows (forward) / columns (backward) of a front handled by one workgroup of the solve kernels
#ifndef SDM_SW
#define SDM_SW 8
#endif
constexpr int SW = SDM_SW;                 // columns of the diagonal block swept in registers at a time (readlane chain), rest via LDS
constexpr int LDL_THREADS = 512;      // workgroup of the diagonal-block kernel: wavefront 0 sweeps, the other 7 apply the previous sweep
constexpr int PANEL_THREADS = 256;    // workgroup of the row-solve kernel: 4 wavefronts, one per SIMD (the solve is issue bound)
constexpr int ROWS_BATCH = 16 * (PANEL_THREADS / 64);   // rows per workgroup of the row-solve kernel (16 per wavefront on the matrix cores)
constexpr int TRSM_ROWS = 128;        // up to this many rows below the diagonal block are solved by the diagonal-block kernel itself
// fronts with fewer rows below their first panel use the bit-faithful row substitution, one row per work-item; the others the blocked
// matrix-core row solve, and -- levels that fit the device -- the one-launch path (k_ldl_front).  256 until round 4; with 48 the
// reference's small examples take the one-launch path: arch0 (174 rows) factor + inverse 0.178 -> 0.085 ms, nb (123) 0.106 -> 0.068
// (profiles/r04x_small_fronts.txt; golden fixtures and 150 rank-deficient fronts of 100 .. 330 rows identical in decisions)
#ifndef SDM_MFMA_MIN_ROWS
#define SDM_MFMA_MIN_ROWS 48
#endif
constexpr int MFMA_MIN_ROWS = SDM_MFMA_MIN_ROWS;
constexpr int CHK = 16;               // column chunk of the row substitution held in registers
constexpr int64_t ASM_FULL_MAX = 4 << 20;   // arenas of up to this many entries are assembled with the zero fill folded in
constexpr int FRONT_MAXT = 64;        // fronts of up to this many 64-row tile rows (and at least MFMA_MIN_ROWS + NB rows) are factored by ONE launch (k_ldl_front)
constexpr int FRONT_XCNT_OFF = 2 * FRONT_MAXT + FRONT_MAXT * FRONT_MAXT;   // behind the counters of k_ldl_front: those of k_sinv_follow (tiles of the inverse done)
constexpr int FRONT_CNT = FRONT_XCNT_OFF + FRONT_MAXT * FRONT_MAXT;   // counters per front of k_ldl_front: rows solved, update steps, updates per tile; then k_sinv_follow's
constexpr int FRONT_POOL = 1;         // tiles per tile workgroup of k_ldl_front (levels that would need more keep the launch-per-panel path; SDM_FRONT_POOL overrides)
constexpr int PANEL_RB = (LDL_THREADS / 64) * NB * 17;     // doubles: max(Lc 64x64, Xs 48 x TRSM_ROWS, 8 wave tiles 64x17)
constexpr size_t PANEL_LDS = (size_t)(NB * (NB + 1) + PANEL_RB) * sizeof(double);
constexpr size_t PANEL_LDS_RIDE = std::max(PANEL_LDS, (size_t)4 * NB * UTP * sizeof(double));   // two update tiles side by side
constexpr int FRONT_CV_OFF = NB * (NB + 1) + (LDL_THREADS / 64) * NB * 17;          // k_ldl_front: doubles, behind S and the wave tiles: the next diagonal tile's values
constexpr size_t FRONT_LDS = (size_t)(FRONT_CV_OFF + NB * TILE) * sizeof(double);
static_assert(PANEL_RB >= NB * NB && PANEL_RB >= (NB - CHK) * TRSM_ROWS && TRSM_ROWS == 16 * (LDL_THREADS / 64) && ROWS_BATCH <= TRSM_ROWS, "panel LDS layout");

template <class T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  DevBuf() = default;
  DevBuf(const DevBuf &) = delete;
  DevBuf &operator=(const DevBuf &) = delete;
  DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  DevBuf &operator=(DevBuf &&o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr; n = 0;
  }
  void alloc(size_t count) {
    release();
    n = count;
    SDM_HIP_CHECK(hipMalloc((void **)&p, (count ? count : 1) * sizeof(T)));
  }
  void upload(const std::vector<T> &h) {
    alloc(h.size());
    if (!h.empty()) SDM_HIP_CHECK(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  }
  void upload(const T *h, size_t count) {
    alloc(count);
    if (count) SDM_HIP_CHECK(hipMemcpy(p, h, count * sizeof(T), hipMemcpyHostToDevice));
  }
};

// Arguments of k_ldl_panel that only its diagonal-block role reads (pivot thresholds and report, the never-fail rule's probe inputs):
// kept in device memory behind one pointer -- as sixteen more scalar kernel arguments they were live across the whole kernel and the
// register allocator parked them in a VGPR's lanes (sgpr spills, and a VGPR spill around the role's call)
struct PanelCtx {
  double *lb; const double *ubp; int *pivstat; double *pivval; double *colbuf; const double *ada; const int *asm_src; const int64_t *Ljc;
  int mtot, pad;
};

// one int in pinned host memory that kernels of a plan can raise (rare error reports: no copy, no symbol lookup;
// the host reads it after a stream synchronise)
struct HostFlag {
  int *host = nullptr;
  HostFlag() = default;
  HostFlag(const HostFlag &) = delete;
  HostFlag &operator=(const HostFlag &) = delete;
  HostFlag(HostFlag &&o) noexcept : host(o.host) { o.host = nullptr; }
  HostFlag &operator=(HostFlag &&o) noexcept { if (this != &o) { if (host) (void)hipHostFree(host); host = o.host; o.host = nullptr; } return *this; }
  ~HostFlag() { if (host) (void)hipHostFree(host); }
  void ensure() {
    if (host) return;
    SDM_HIP_CHECK(hipHostMalloc((void **)&host, 4 * sizeof(int), 0));      // (CholPlan::noted uses two of them)
    for (int i = 0; i < 4; i++) host[i] = 0;
  }
  int *dev() {
    ensure();
    void *d = nullptr;
    SDM_HIP_CHECK(hipHostGetDevicePointer(&d, host, 0));
    return (int *)d;
  }
};

// pinned host staging buffer of ints
struct PinnedInts {
  int *p = nullptr; size_t n = 0;
  PinnedInts() = default;
  PinnedInts(const PinnedInts &) = delete;
  PinnedInts &operator=(const PinnedInts &) = delete;
  ~PinnedInts() { if (p) (void)hipHostFree(p); }
  void ensure(size_t count) {
    if (count <= n) return;
    if (p) (void)hipHostFree(p);
    p = nullptr; n = 0;
    SDM_HIP_CHECK(hipHostMalloc((void **)&p, (count ? count : 1) * sizeof(int), 0));
    n = count;
  }
};

// ---------------------------------------------------------------- chol plan
// Supernodal elimination tree of the symbolic factor L laid
// out for a level-scheduled multifrontal LDL': every supernode s owns a dense
// m_s x m_s column-major "front" in one HBM arena; its first n_s columns end up
// holding the columns of L (unit diagonal stored as 1.0, blkchol2.c:136).
struct LevelLaunch {
  int level, panel;     // panel index p (columns p*NB .. ) of the fronts in level
  int nactive;          // fronts (prefix of the level list, sorted by n_s desc) with n_s > p*NB
  int maxrows;          // max over active fronts of rows below the panel's diagonal block
  int maxtiles;         // max number of TILE x TILE lower tiles in the trailing update
  int lasttiles;        // the same over the fronts whose LAST panel this is (stand-alone update launch)
  int q0;               // first panel of the level whose diagonal-block launch carries the previous update's tiles
  int ride_wgs;         // workgroups of k_ldl_panel beyond workgroup 0 (row solves + pairs of update tiles), max over fronts
};

// one etree level of the solves (sdm_solve.hip)
struct SolveLevel {
  int nfronts = 0, maxns = 0, maxms = 0, nsb = 0;   // nsb = super-blocks of the widest front
  bool children = false;                             // some front of the level has children (assembly launch needed)
  bool below = false;                                // some front has rows below its own columns
  std::vector<int> slabs_fw;                         // grid.x of the forward step launch behind super-block P (0: none)
};

struct CholPlan {
  sdm_int m = 0, nsuper = 0, nnzL = 0, nnzADA = 0;
  int nlevels = 0;
  int64_t fsize = 0, wsize = 0, tsize = 0;
  int maxms = 0, maxns = 0;
  std::vector<sdm_int> Ljc, perm;
  std::vector<int> sn_first, sn_ns, sn_ms, sn_ld, sn_parent, sn_level;
  std::vector<int64_t> sn_foff, sn_xl, sn_woff, sn_roff, sn_toff;
  std::vector<int> childptr, childlist, levptr, levlist, lev_T;
  std::vector<char> sn_active;           // empty: every supernode; else the supernodes THIS plan factors and solves (the level lists hold only
                                         // these: the others keep their place in the arenas and are never touched by a launch -- sedumi_amd.dist)
  double pars_canceltol = 0, pars_maxu = 0, pars_abstol = 0; int pars_use_absd = 0;   // of the factorisation in progress (chol_begin .. chol_end)
  std::vector<LevelLaunch> launches;     // factor panel launches in execution order
  std::vector<int> lev_first_launch;     // index into launches per level (+ sentinel)
  // device copies
  DevBuf<int> d_first, d_ns, d_ms, d_ld, d_parent, d_childptr, d_childlist, d_levlist, d_lindx, d_relidx, d_perm;
  DevBuf<int64_t> d_foff, d_xl, d_woff, d_roff, d_toff, d_asm_dst, d_asm_dstT, d_Ljc;
  DevBuf<int> d_asm_src;
  DevBuf<double> fronts, frontsT, wvec, colbuf, d, dsolve, lb, pivval, ub;
  DevBuf<int> pivstat;
  DevBuf<int> diag_cnt;    // per front: panels whose factored diagonal block has been published (k_ldl_panel)
  DevBuf<int> d_asm_fsrc;  // arena entry -> ADA value index (k_assemble_full), empty for arenas above ASM_FULL_MAX entries
  DevBuf<int> front_cnt;   // k_ldl_front: FRONT_CNT counters per front of a one-launch level (slot d_fslot[s]): rows solved through
                           // panel / update steps finished per tile row, updates applied per tile
  DevBuf<int> d_fslot;     // front -> its slot in front_cnt (fronts of other levels: 0, unused)
  std::vector<char> lev_persist;   // level factored by ONE k_ldl_front launch (all its fronts qualify)
  std::vector<int> lev_maxT;       // its grid: tile rows of the tallest front ...
  std::vector<int> lev_ntw;        // ... plus this many tile workgroups per front
  DevBuf<int> upd_cnt;     // per front: finished tile workgroups of the updates that rode along with k_ldl_panel
  int own = 0;             // block-cyclic ranks (sdm_plan_set_column_owner): world | rank << 8 | blk << 16, 0 = this plan owns every tile column (sdm_chol.hip: owns_col)
  DevBuf<double> panelrec; // the record of one finished panel between those ranks (chol_panel_record)
  DevBuf<PanelCtx> panel_ctx;   // what only workgroup 0 of k_ldl_panel needs, behind ONE kernel argument (uploaded when it changes)
  PanelCtx panel_ctx_host = {};
  HostFlag tmo;            // raised by a spin inside a panel launch of THIS plan that gave up (chol_wait_timeouts)
  // ---- solves (sdm_solve.hip): per front and super-block of sbw columns one nb x nb array in the arena S = the explicit
  // inverse of that diagonal block of L (block P of front s at sn_soff[s] + P * sbw * sn_sld[s], leading dimension sn_sld[s])
  int sbw = SBW_MIN;                 // super-block width of this plan (solve_build: covers the widest front, at most SBW_MAX)
  int sbw_req = 0;                   // != 0: width asked for through sdm_plan_set_solve_width (before set_chol)
  bool front_off_req = false;        // sdm_plan_set_one_launch_fronts(p, 0): the next set_chol plans no k_ldl_front level
  bool sprep_off = false;            // sdm_plan_set_one_launch_inverse(0): the inverses of small problems by a launch per stage too (k_sinv128 + k_stile, not k_sprep)
  int tile_wgs_req = 0;              // sdm_plan_set_tile_workgroups: workgroups a big front's update tiles are dealt to (0 = the device's compute units)
  int64_t ssize = 0; int nsbtot = 0;
  std::vector<int64_t> sn_soff; std::vector<int> sn_sld, sn_sboff;
  DevBuf<int64_t> d_soff; DevBuf<int> d_sld, d_sboff;
  DevBuf<double> S, xfin, zdiv;
  DevBuf<double> ST;                 // the same blocks transposed (ST[r*sld + c] = inverse(r, c)): the forward sweep reads ROWS of an inverse, the
                                     // backward sweep columns -- with both copies every product of the sweeps is a dot product along contiguous memory
  DevBuf<double> LT;                 // fronts of several super-blocks: the rows of L below super-block P against its columns, transposed likewise
                                     // (LT block (s, P) at sn_ltoff[s] + lt_boff(P), (ns - (P+1) sbw) rows of nb_P <= sbw entries, pitch sbw): forward step
  std::vector<int64_t> sn_ltoff; DevBuf<int64_t> d_ltoff;
  DevBuf<int> l_lt; int n_lt = 0;    // 64x64 tiles of that transposition (4 ints each: s, P, I, J)
  DevBuf<double> Tarena;             // scratch of the inversion, same layout as S: T = B inv(A) of every combine step
  DevBuf<int> sweep_cnt;             // counters of the merged sweep launches (two sets of MC_N, a 128-byte line each: sdm_solve.hip, merged_count)
  DevBuf<unsigned long long> sb_g;   // per super-block: bit patterns of max|inverse| and max|L block| (growth check); behind them the counters of k_sprep
  DevBuf<int> l_i128, l_items;       // work lists of the inversion: 128-column leaves (4 ints each), combine tiles (8 ints each, sorted by stage)
  int n_i128 = 0, n_items = 0;
  bool follow = false;               // every front is factored by k_ldl_front and inverted behind it by k_sinv_follow (no solve_prepare launches)
  bool front_disabled = false;       // a launch of this plan timed out: its later factorisations take the launch-per-panel path (chol_wait_timeouts)
  std::vector<int> lev_followT;      // grid.x of k_sinv_follow per level: tiles of the inverse of its widest front
  std::vector<int> stage_ptr;        // combine tiles of stage st (= 2 * level + (0: T, 1: X)) are l_items[stage_ptr[st] .. stage_ptr[st+1])
  std::vector<SolveLevel> slev;
  DevBuf<double> prep_part; DevBuf<int> prep_ticket;   // k_begin_factor: maxima per bounds workgroup, the ticket of the last one
  double growth_max = 1e4;           // a super-block whose max|inv| * max|L| exceeds this is solved by substitution
  double growth_used = 1e4;          // the bound in force at the last solve_prepare
  // blocks beyond growth_max but within refine_max: inverse + REFINE_STEPS steps of iterative refinement against the factor
  // (k_sfw_resid / k_sbw_resid) when the sweeps run their refinement launches.  refine_mode 1 (default): they do while such blocks
  // keep turning up.  Every sweep has a number (sweep_seq); a launch that meets such a block stores the sweep's number in
  // noted[0], the last diagonal-block launch of every sweep stores it in noted[1] (pinned host memory, read without
  // synchronising when the next sweep is enqueued): refinement is switched ON when the latest sweep known to have run (or the one
  // before it) met such a block, OFF when two sweeps have run since the last one that did, and stays as it is otherwise.  The
  // first sweeps that meet such a block -- those enqueued before the news arrives -- substitute.  0: never refine; 2: always.
  double refine_max = 1e10;
  int refine_mode = 1;
  bool refine_on = false;
  int sweep_seq = 0;
  HostFlag noted;
};

// device view of the front tables
struct FrontTab {
  const int *first, *ns, *ms, *ld;
  const int64_t *foff, *xl, *woff, *roff, *toff;
  const int *childptr, *childlist, *lindx, *relidx;
  const int64_t *soff; const int *sld, *sboff;
  const int64_t *ltoff = nullptr;
  const int *fslot = nullptr;   // k_ldl_front: the front's counter slot
  // levels of ONE front (solve kernels): its descriptor rides along as kernel arguments, so the first data load of a
  // launch does not wait for two dependent table loads (list[..] -> ns[s], soff[s], ...)
  int one = 0, o_s = 0, o_ns = 0, o_ms = 0, o_ld = 0, o_first = 0, o_sld = 0, o_sboff = 0;
  int64_t o_foff = 0, o_soff = 0, o_woff = 0, o_xl = 0, o_ltoff = 0;
};

// ----------------------------------------------------------------- ada plan
struct AdaPlan {
  sdm_int N = 0, m = 0, nnzA = 0, lpN = 0, lorN = 0, sdpN = 0, rsdpN = 0;
  sdm_int nlq = 0;       // number of LP + Lorentz rows (= first PSD row)
  sdm_int lenud = 0, ntask = 0, zlen = 0, nnzQ = 0, nnz_lq = 0;
  int maxn = 0;
  bool thread_per_row = false;
  sdm_int col0 = 0, col1 = 0;            // column range of ADA' formed by this plan (sdm_plan_getada_cols)
  double *zero_ptr = nullptr; long long zero_n = 0; bool zero_defer = false;   // a clearing of ADA' left to the next stage (ada_zero_flush, sdm_ada.hip)
  std::vector<int64_t> h_taskptr;       // host copy of c_taskptr
  std::vector<sdm_int> psd_n, psd_start, psd_udoff;
  DevBuf<int64_t> d_Ajc, d_Ajc_psd, d_Qjc, d_ADAjc;
  DevBuf<int> d_Air, d_Qir, d_ADAir, d_ADAT;
  DevBuf<double> d_Apr;
  DevBuf<int> d_Ablk, d_Aupos;            // per PSD nonzero of At: block id, position in U_k
  // stage-1 tasks (constraint j, PSD block k)
  DevBuf<int> t_col, t_blk, t_n, t_nslot, t_ulen, t_herm, t_order, t_order_xcd;
  DevBuf<int64_t> t_slotptr, t_udoff, t_uoff, t_zoff;
  DevBuf<int> s_col;                      // per slot: column of X_jk
  DevBuf<int64_t> s_nzptr;                // per slot (+1): nonzero range in At
  int64_t s1_maxnz = 0;                   // nonzeros of the largest stage-1 task
  int64_t lq_maxcol = 0, q_maxcol = 0;    // longest column of the LP + Lorentz part of At / of DAt.q (k_ada_spdot: lanes per pattern entry)
  bool one_task_per_col = false;          // every constraint touches at most one PSD block (its z_j is one task's output: stage 2 can ride in that task)
  int s1_maxulen = 0;                     // targets of the largest union pattern
  DevBuf<int> u_rc;                       // the same targets as (r << 16) | c (real blocks)
  DevBuf<int> u_pos;                      // concatenated target lists U_k (position r + c*n_k [+ n_k^2 for Im])
  DevBuf<int64_t> c_taskptr;              // per constraint: its tasks
  DevBuf<double> zbuf, dsqr, symtmp;
  DevBuf<int> dsqr_code;
  // dense-column form of the LP / Lorentz part (ada_build decides): At(0:nlq, :) and DAt.q as dense column-major
  // arrays, ADA' contributions as weighted Gram matrices on the FP64 matrix cores (split-K, fixed-order reduction)
  bool lq_dense = false, q_dense = false;
  int gram_split = 1;
  DevBuf<double> Alq_d, Q_d, gram_part;
  DevBuf<int64_t> q_dst;
  DevBuf<int> q_src;                      // Q_d entry -> index into qpr, -1 = zero (k_lq_q_prep; fused LP + Lorentz Gram form)
  DevBuf<int64_t> t_end, d_psd_start;
  DevBuf<double> dl, ddet, qpr, udsqr;
  DevBuf<double> q1, q2;                  // d.q1 (per Lorentz cone), d.q2 (norm-bound parts): inputs of sdm_plan_getdatq
  DevBuf<int64_t> d_qblk;                 // qblkstart (0-based rows of the norm-bound parts), lorN+1
  DevBuf<double> ufac;                    // d.u of the scaling (input of sdm_plan_invcholfac), lenud doubles
  DevBuf<int> ic_n, ic_poff, ic_perm; DevBuf<int64_t> ic_off;   // invcholfac block tables
  PinnedInts ic_perm_host; bool ic_has_perm = false;            // staging of the permutation handed to sdm_plan_invcholfac
  // ---- Amul / vecsym / psdscale (sdm_pcg.hip): host copy of At (set by sdm_plan_set_ada), its transposed
  // device copy, the cone-space work vectors and the PSD block tables -- built on first use
  std::vector<sdm_int> h_Ajc, h_Air; std::vector<double> h_Apr;
  bool pcg_ready = false;
  DevBuf<int64_t> d_Tjc, pb_off; DevBuf<int> d_Tir, pb_n, pb_herm, pb_poff, pb_items, aden_cols;
  DevBuf<double> d_Tpr, xN, psd, psdtmp, aden;
  int pcg_ntiles = 0, aden_n = 0;
  size_t stage1_lds = 0;
  // stage-2 fast path: interleaved (ELL) copy of the PSD nonzeros, rows sorted by length, groups of 64
  bool ell_ok = false;
  int ell_ng = 0;
  int64_t zmax = 0, zmaxj = 0;
  DevBuf<int64_t> c_zlen;
  DevBuf<int> g_row, g_len, g_bu;
  DevBuf<long long> cdc64, cdp64;          // k_psd_stage2_ell: one record per column, by column / by ELL position (sdm_ada.hip, set-up)
  DevBuf<int> cdc32, cdp32;
  DevBuf<int> d_Azpos, t_zdst;             // k_psd_stage2_ell: position in the full-length z vector of every PSD nonzero / of every task's block
  DevBuf<int> ell_pos;                     // constraint -> position in the ELL row order (its group = pos / 64)
  DevBuf<int> ell_order;                   // position -> constraint
  bool ell_full = false;                   // the ADA' pattern is full (symmetric half-sweep of stage 2 possible)
  DevBuf<int64_t> g_off, d_uoff;
  DevBuf<double> g_val;
};

}  // namespace sdm

namespace sdm {
// optional per-kernel HIP-event timing (bench.py roofline leg): every launch made through SDM_KLAUNCH is
// bracketed by two events on the plan's stream while enabled.
struct KProf {
  bool enabled = false;
  struct Rec { const char *name; hipEvent_t a, b; };
  std::vector<Rec> recs;
  std::vector<hipEvent_t> pool;
  hipEvent_t get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e; SDM_HIP_CHECK(hipEventCreate(&e)); return e;
  }
};
}  // namespace sdm

namespace sdm {
// resident dense-column unit: the dense columns of A re-enter the normal equations as a product of rank-1 factors
// (deninfac.m:58-94); symbolic data once per solve (dense_set), LAD = L \ Ad and the product-form factors every iteration
struct DensePlan {
  sdm_int nden = 0, dznnz = 0, pnnz = 0, mrows = 0;
  bool active = false, factored = false;
  bool tables_on_host = false;                    // betajc / permoff / dopiv below mirror the device tables of the last factorisation
  std::vector<sdm_int> LADjc, LADir, dzjc, dzir, colperm, first;
  std::vector<int64_t> poff, betajc, permoff;
  std::vector<int> dopiv;
  std::vector<int> later_ptr;                     // per factor k: the later columns its inverse is applied to (first <= k), d_later[later_ptr[k] ..)
  DevBuf<int> d_dzir, d_colperm, d_pivperm, d_dopiv, d_later;
  DevBuf<int64_t> d_dzjc, d_poff, d_betajc, d_permoff;
  DevBuf<double> ad, lad, wvb, p, beta, dgat, smult, dden;
  DevBuf<double> w_psq, w_mu, w_key;              // scratch of k_dpr1_general
  DevBuf<int> w_ord, w_ord2, w_acc, w_post, w_dep, w_st;
};
}  // namespace sdm

struct sdm_plan {
  int device = 0;
  sdm::DensePlan dense;
  sdm::KProf kprof;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  bool has_chol = false, has_ada = false, factored = false;
  sdm::CholPlan chol;
  sdm::AdaPlan ada;
  sdm::DevBuf<double> ada_val, absd, rhs, y, ywork, lpr;
  std::vector<sdm_int> ada_jc, ada_ir;   // host copy of the ADA pattern
  hipEvent_t ev_begin[16] = {}, ev_end[16] = {};
  bool capturing = false;
  sdm::PinnedInts dl_stage;          // pinned staging of small read-backs (sdm_plan_download: DL_STAGE_BYTES)
  std::vector<hipGraphExec_t> graphs;   // captured launch sequences (sdm_plan_graph_*)
};

#define SDM_KLAUNCH_ON(P, st_, kernel, grid, block, shmem, ...)                           \
  do {                                                                                     \
    if ((P)->kprof.enabled) {                                                              \
      sdm::KProf::Rec r_; r_.name = #kernel; r_.a = (P)->kprof.get(); r_.b = (P)->kprof.get(); \
      SDM_HIP_CHECK(hipEventRecord(r_.a, (st_)));                                          \
      SDM_LAUNCH(kernel, grid, block, shmem, (st_), __VA_ARGS__);                          \
      SDM_HIP_CHECK(hipEventRecord(r_.b, (st_)));                                          \
      (P)->kprof.recs.push_back(r_);                                                       \
    } else {                                                                               \
      SDM_LAUNCH(kernel, grid, block, shmem, (st_), __VA_ARGS__);                          \
    }                                                                                      \
  } while (0)
#define SDM_KLAUNCH(P, kernel, grid, block, shmem, ...) SDM_KLAUNCH_ON(P, (P)->stream, kernel, grid, block, shmem, __VA_ARGS__)
#ifdef SDM_EMU
// (tests/hipemu only) a launch whose workgroups wait for each other, as one process per workgroup: emu_launch_concurrent
#define SDM_KLAUNCH_CONCURRENT(P, kernel, grid, block, shmem, ...)                             \
  do {                                                                                         \
    sdm::KProf::Rec r_; r_.name = #kernel;                                                     \
    if ((P)->kprof.enabled) { r_.a = (P)->kprof.get(); r_.b = (P)->kprof.get(); SDM_HIP_CHECK(hipEventRecord(r_.a, (P)->stream)); } \
    SDM_LAUNCH_CONCURRENT(kernel, grid, block, shmem, __VA_ARGS__);                            \
    if ((P)->kprof.enabled) { SDM_HIP_CHECK(hipEventRecord(r_.b, (P)->stream)); (P)->kprof.recs.push_back(r_); } \
  } while (0)
#endif

namespace sdm {
// Two launches whose workgroups all have to be resident (k_ldl_front) of different plans (streams) must
// not share the device: each needs ALL its workgroups resident and waits inside, so two half-dispatched ones could hold the compute units the other is waiting for.  Launches of one
// process take turns per device: a plan that follows ANOTHER plan's launch first waits (on the device: an event recorded
// on that plan's stream, hipStreamWaitEvent on its own) -- a plan that has the device to itself pays a mutex and nothing
// else.  Kernels whose workgroups only wait for workgroups dispatched before them (k_ldl_panel, k_sprep) need no turn.
// (Header-defined: the function-local statics are one instance per process.)
// Not inside a graph capture (an event of another stream cannot be captured): replayed graphs of several plans that
// contain such launches must be ordered by the caller.
struct PersistTurn {
#ifdef SDM_EMU
  explicit PersistTurn(sdm_plan *) {}
  static void forget(sdm_plan *) {}
#else
  static constexpr int MAXDEV = 64;
  static std::mutex &mtx() { static std::mutex m; return m; }
  static hipEvent_t *events() { static hipEvent_t ev[MAXDEV] = {}; return ev; }
  static sdm_plan **owners() { static sdm_plan *pl[MAXDEV] = {}; return pl; }
  sdm_plan *P;
  bool on;
  std::unique_lock<std::mutex> lk;
  explicit PersistTurn(sdm_plan *p) : P(p), on(!p->capturing && p->device >= 0 && p->device < MAXDEV) {
    if (!on) return;
    lk = std::unique_lock<std::mutex>(mtx());
    sdm_plan *prev = owners()[P->device];
    if (prev && prev != P && !prev->capturing) {
      hipEvent_t &ev = events()[P->device];
      if (!ev) SDM_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
      SDM_HIP_CHECK(hipEventRecord(ev, prev->stream));
      SDM_HIP_CHECK(hipStreamWaitEvent(P->stream, ev, 0));
    }
  }
  ~PersistTurn() { if (on) owners()[P->device] = P; }
  static void forget(sdm_plan *p) {                                    // the plan is going away: nobody waits for its stream any more
    std::lock_guard<std::mutex> g(mtx());
    for (int dv = 0; dv < MAXDEV; dv++) if (owners()[dv] == p) owners()[dv] = nullptr;
  }
#endif
};
void set_error(const std::string &msg);
// sdm_chol.hip
void chol_build(sdm_plan *P, sdm_int m, const sdm_int *Ljc, const sdm_int *Lir, const sdm_int *perm,
                sdm_int nsuper, const sdm_int *xsuper, const sdm_int *ADAjc, const sdm_int *ADAir);
void chol_factor(sdm_plan *P, const double canceltol, const double maxu, const double abstol, int use_absd);
// the same in three steps (sdm_plan_blkchol_begin / _levels / _end: the multi-GPU layer reduces update matrices between levels)
void chol_begin(sdm_plan *P, const double canceltol, const double maxu, const double abstol, int use_absd);
void chol_levels(sdm_plan *P, int l0, int l1, bool extend_only, int pan0 = 0, int pan1 = 1 << 30);
void chol_panel_record(sdm_plan *P, int panel, int unpack);
void chol_end(sdm_plan *P);
void chol_forget_plan(sdm_plan *P);    // the plan is being destroyed (turn-taking of k_ldl_front launches)
int chol_wait_timeouts(sdm_plan *P);   // non-zero: a spin inside a panel launch of this plan gave up since the last call (call after a stream sync)
void chol_extract(sdm_plan *P, double *d_Lpr_out);           // device pointer, nnzL doubles
void chol_load_factor(sdm_plan *P, const double *h_Lpr, const double *h_d = nullptr);   // host L values (and d) -> fronts (stand-alone solves)
void vec_gather(sdm_plan *P, double *dst, const double *src, bool forward);  // dst[k]=src[perm[k]] / dst[perm[k]]=src[k]
void vec_divd(sdm_plan *P, double *v);
FrontTab front_tab(CholPlan &C);
// sdm_solve.hip: inverse-block solves
void solve_build(sdm_plan *P);                      // host tables + buffers (end of chol_build)
void solve_prepare(sdm_plan *P, bool sb_g_is_zero);  // after a factorisation: inverses of the diagonal super-blocks
void solve_follow(sdm_plan *P, int level, hipStream_t st);   // the same for the fronts of a k_ldl_front level, launched NEXT to that kernel (CholPlan::follow)
const double *solve_d(sdm_plan *P);                  // the d the solves divide by: L.d (skipped pivots act as 1), or Ld of deninfac
void solve_run(sdm_plan *P, const double *rhs, double *yout, int mode);   // mode bits 1 fw | 2 ./d | 4 bw
// the fw, ./d, bw solve of solve_run(mode 7) level by level: what = 1 assembly launches of the forward sweep of levels l0 .. l1-1 only,
// 2 their forward sweep without the assembly, 3 both, 4 the backward sweep of levels l1-1 down to l0
void solve_levels(sdm_plan *P, int what, int l0, int l1);
void solve_stats(sdm_plan *P, sdm_int *nblocks, sdm_int *nbad, double *max_growth);
void solve_fw_batch(sdm_plan *P, const double *rhs, int64_t rhs_stride, double *y, int64_t y_stride, double *wv, int nrhs,
                    double *zdiv = nullptr, const double *dscale = nullptr, int l0 = 0, int l1 = -1, int what = 3);
// sdm_pcg.hip: Amul / vecsym / psdscale on the plan
void pcg_amul(sdm_plan *P, int transp);
void pcg_set_dense(sdm_plan *P, sdm_int nden, const sdm_int *cols, const double *Aden);
void pcg_vecsym(sdm_plan *P);
void pcg_psdscale(sdm_plan *P, int transp, bool with_perm);
// sdm_dpr1.hip: resident dense-column unit (deninfac.m:58-94)
void dense_set(sdm_plan *P, sdm_int nden, const sdm_int *LADjc, const sdm_int *LADir, const sdm_int *dzjc, const sdm_int *dzir,
               const sdm_int *colperm, const sdm_int *first);
void dense_factor(sdm_plan *P, const double *smult, double maxuden, int *host_fallback);
void dense_tables(DensePlan &D, sdm_int m, sdm_int nden, const sdm_int *dzjc, const sdm_int *dzir, const sdm_int *colperm, const sdm_int *first);
void dense_prodformfact(sdm_plan *P, hipStream_t st, DensePlan &D, const double *smult, double maxu);   // the product-form factorisation on the device
void dense_fetch_tables(hipStream_t st, DensePlan &D);
void dense_prodform(sdm_plan *P, double *y, bool with_divide);       // y <- bwdpr1(Lden, fwdpr1(Lden, y) ./ Ld)   (permuted order)
// sdm_capi.hip: gateway-shaped plans -- what ONE getada gateway needs (tier 1 builds one per call, sdm_mexcache.hip keeps them)
void gw_upload_invperm(DevBuf<int> &buf, const sdm_int *perm, sdm_int m);
void gw_build_getada1(sdm_plan *p, sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir, const sdm_int *Ajc, const sdm_int *Air,
                      const double *Apr, const sdm_int *Ajc2, sdm_int lpN, sdm_int lorN, const sdm_int *qblkstart);
void gw_run_getada1(sdm_plan *p, const int *d_invperm, const double *dl, const double *ddet);
void gw_build_getada2(sdm_plan *p, sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir, sdm_int lorN, const sdm_int *Qjc, const sdm_int *Qir);
void gw_run_getada2(sdm_plan *p, const int *d_invperm, const double *Qpr);
void gw_build_getada3(sdm_plan *p, sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir, sdm_int N, const sdm_int *Ajc, const sdm_int *Air,
                      const double *Apr, const sdm_int *Ajc1, const sdm_cone *K, const sdm_int *psd_blkstart);
void gw_run_getada3(sdm_plan *p, const double *udsqr, bool input_is_zero);
bool gw_getada1_is_zero(sdm_plan *p);   // getada1 on this plan has nothing to add (no LP / Lorentz nonzeros)
void gw_build_getada(sdm_plan *p, sdm_int m, const sdm_int *ADAjc, const sdm_int *ADAir, const sdm_int *Ajc, const sdm_int *Air, const double *Apr,
                     sdm_int lpN, sdm_int lorN, const sdm_int *qblkstart, const sdm_int *Qjc, const sdm_int *Qir);
void gw_run_getada(sdm_plan *p, const double *dl, const double *ddet, const double *Qpr);
void gw_download(sdm_plan *p, double *ADApr, double *absd);
// sdm_ada.hip
void ada_build(sdm_plan *P, sdm_int N, sdm_int m, const sdm_int *Ajc, const sdm_int *Air, const double *Apr,
               const sdm_int *Ajc_psd, sdm_int lpN, sdm_int lorN, const sdm_int *lorNL, sdm_int sdpN,
               sdm_int rsdpN, const sdm_int *sdpNL, const sdm_int *qblkstart, const sdm_int *psd_blkstart,
               const sdm_int *Qjc, const sdm_int *Qir, const sdm_int *ADAjc, const sdm_int *ADAir);
// mode bits: 1 = LP/Lorentz-det part (getada1), 2 = Lorentz rank-1 part (getada2), 4 = PSD part (getada3)
// tri_perm (device, length m, inverse permutation) != nullptr -> only entries with invperm[i] <= invperm[j]
// are touched (the reference's triangular bookkeeping); symmetrize -> spmakesym afterwards.
void ada_zero_flush(sdm_plan *P);
void ada_lq(sdm_plan *P, double *ada, const int *d_invperm, bool accumulate);
void ada_q(sdm_plan *P, double *ada, const int *d_invperm, bool accumulate);
void ada_psd(sdm_plan *P, double *ada, const int *d_invperm, bool sym_input, bool absd_done = false);
bool ada_lq_q(sdm_plan *P, double *ada);   // ada_lq + ada_q of a full ADA' in three launches when both take the dense Gram form
void ada_datq(sdm_plan *P);     // qpr = values of DAt.q (getDAtm.m:39-44) from the resident d.q1, d.q2
// y(perm,perm) = u'u per PSD block (invcholfac.c); u, y device (lenud doubles), perm device int32 0-based or null
void psd_invcholfac(hipStream_t st, const double *u, double *y, const int *perm, const std::vector<int> &ns, int rsdpN,
                    DevBuf<int> &d_n, DevBuf<int64_t> &d_off, DevBuf<int> &d_poff, bool tables_ready);
}  // namespace sdm
