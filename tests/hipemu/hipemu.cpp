/*
 * hipemu.cpp -- TEST INFRASTRUCTURE ONLY (tests/hipemu).  See hipemu.h.
 * Cooperative-fiber execution of one workgroup at a time (x86-64 only); for kernels whose workgroups wait for each
 * other, optionally one forked process per workgroup over shared "device" memory (emu_launch_concurrent).
 */
#include "hipemu.h"
#include <chrono>
#include <stdexcept>
#include <string>
#include <vector>
#include <execinfo.h>
#include <sched.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

extern "C" void emu_ctx_switch(void **save_sp, void *new_sp);
asm(R"(
.text
.globl emu_ctx_switch
.type emu_ctx_switch,@function
emu_ctx_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size emu_ctx_switch, .-emu_ctx_switch
)");

namespace {
constexpr size_t STACK_BYTES = 96 * 1024;
struct Fiber {
  void *sp = nullptr;
  char *stack = nullptr;
  bool done = false;
  emu_uint3 tid{0, 0, 0};
  int lin = 0;
};
struct Barrier { int count = 0; unsigned gen = 0; int need = 0; };

std::vector<Fiber> g_fibers;
std::vector<char *> g_stack_pool;
void *g_main_sp = nullptr;
Fiber *g_cur = nullptr;
const std::function<void()> *g_body = nullptr;
Barrier g_block_bar;
std::vector<Barrier> g_wave_bar;
std::vector<double> g_wave_scr;   /* per wave: 3*64 doubles */
std::vector<char> g_dyn_smem;
unsigned long g_progress = 0;
int g_reverse = 0;
int g_concurrent = 0;        /* tests: launches of kernels with inter-workgroup waits run one process per workgroup */
bool g_in_wg_process = false;
bool g_group_open = false;   /* launches between emu_group_begin / _end run side by side: their processes are waited for at the end */
std::vector<pid_t> g_group_kids;
std::string g_group_names;
bool g_spinning = false;     /* some work-item of the round polled another workgroup (emu_spin_pause) */
constexpr size_t SHARED_HDR = 256;

void yield_to_main() { emu_ctx_switch(&g_cur->sp, g_main_sp); }

void fiber_entry() {
  (*g_body)();
  g_cur->done = true;
  g_progress++;
  yield_to_main();
  abort(); /* never resumed */
}

void barrier_wait(Barrier &b) {
  unsigned gen = b.gen;
  if (++b.count == b.need) { b.count = 0; b.gen++; g_progress++; return; }
  while (b.gen == gen) yield_to_main();
}
}  // namespace

emu_uint3 emu_blockIdx{0, 0, 0}, emu_blockDim{1, 1, 1}, emu_gridDim{1, 1, 1};
static emu_uint3 g_dummy_tid{0, 0, 0};
emu_uint3 &emu_threadIdx_ref() { return g_cur ? g_cur->tid : g_dummy_tid; }
char *emu_dyn_smem() { return g_dyn_smem.data(); }
void emu_set_reverse(int rev) { g_reverse = rev; }

void emu_syncthreads() { barrier_wait(g_block_bar); }

static inline int cur_wave() { return g_cur->lin >> 6; }
static inline int cur_lane() { return g_cur->lin & 63; }

double emu_shfl(double v, int arg, int mode) {
  int w = cur_wave(), l = cur_lane();
  double *scr = &g_wave_scr[(size_t)w * 192];
  scr[l] = v;
  barrier_wait(g_wave_bar[w]);
  int src = mode == 0 ? arg : (mode == 1 ? l + arg : (l ^ arg));
  double r = (src >= 0 && src < g_wave_bar[w].need) ? scr[src] : v;
  barrier_wait(g_wave_bar[w]);
  return r;
}

emu_double4 emu_mfma_f64_16x16x4(double a, double b, emu_double4 c) {
  int w = cur_wave(), l = cur_lane();
  double *sa = &g_wave_scr[(size_t)w * 192], *sb = sa + 64;
  if (g_wave_bar[w].need != 64) { fprintf(stderr, "hipemu: MFMA in a partial wave\n"); abort(); }
  sa[l] = a; sb[l] = b;
  barrier_wait(g_wave_bar[w]);
  emu_double4 d = c;
  int col = l & 15;
  for (int r = 0; r < 4; r++) {
    int row = (l >> 4) + 4 * r;
    double acc = d[r];
    for (int k = 0; k < 4; k++) acc = std::fma(sa[row + 16 * k], sb[col + 16 * k], acc);
    d[r] = acc;
  }
  barrier_wait(g_wave_bar[w]);
  return d;
}

void *emu_shared_alloc(size_t n) {
  const size_t total = n + SHARED_HDR;
  void *m = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
  if (m == MAP_FAILED) return nullptr;
  *(size_t *)m = total;
  return (char *)m + SHARED_HDR;
}
void emu_shared_free(void *p) {
  if (!p) return;
  char *m = (char *)p - SHARED_HDR;
  munmap(m, *(size_t *)m);
}
void emu_set_concurrent(int on) { g_concurrent = on; }
/* concurrent launches: fork the workgroups' processes from the LAST workgroup down (waiters before the workgroups they wait for) */
static int g_reverse_blocks = 0;
void emu_set_reverse_blocks(int on) { g_reverse_blocks = on; }
static int g_injected_timeouts = 0;
void emu_inject_timeouts(int n) { g_injected_timeouts = n; }
int emu_take_injected_timeout() { if (g_injected_timeouts <= 0) return 0; g_injected_timeouts--; return 1; }
int emu_concurrent() { return g_concurrent; }
void emu_spin_pause() {
  if (!g_in_wg_process) { fprintf(stderr, "hipemu: waiting on a workgroup that has not run\n"); abort(); }
  /* a work-item polling another workgroup's progress must let its own workgroup's other work-items run meanwhile (on the
     device they run beside it): e.g. work-item 0 may still owe the increment of a counter that the awaited workgroup is
     waiting for.  So: back to the scheduler; it pauses the process when a whole round made no progress (run_block). */
  g_spinning = true;
  yield_to_main();
}

namespace {
void prepare_launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body, int nthr) {
  if (nthr > 1024) { fprintf(stderr, "hipemu: block too large\n"); abort(); }
  while ((int)g_stack_pool.size() < nthr) g_stack_pool.push_back((char *)malloc(STACK_BYTES));
  if (g_dyn_smem.size() < shmem + 64) g_dyn_smem.resize(shmem + 64);
  g_wave_scr.assign((size_t)((nthr + 63) / 64) * 192, 0.0);
  emu_blockDim = {block.x, block.y, block.z};
  emu_gridDim = {grid.x, grid.y, grid.z};
  g_body = &body;
}
void run_block(unsigned bx, unsigned by, unsigned bz, dim3 block, int nthr, const char *name) {
  const int nwaves = (nthr + 63) / 64;
      {
        emu_blockIdx = {bx, by, bz};
        g_fibers.assign(nthr, Fiber());
        g_block_bar = Barrier(); g_block_bar.need = nthr;
        g_wave_bar.assign(nwaves, Barrier());
        for (int w = 0; w < nwaves; w++) g_wave_bar[w].need = std::min(64, nthr - 64 * w);
        for (int t = 0; t < nthr; t++) {
          Fiber &f = g_fibers[t];
          f.stack = g_stack_pool[t];
          f.lin = t;
          f.tid.x = t % block.x; f.tid.y = (t / block.x) % block.y; f.tid.z = t / (block.x * block.y);
          uintptr_t top = ((uintptr_t)(f.stack + STACK_BYTES)) & ~(uintptr_t)15;
          void **sp = (void **)top;
          *(--sp) = nullptr;               /* fake return address for fiber_entry */
          *(--sp) = (void *)&fiber_entry;  /* popped by 'ret' in emu_ctx_switch */
          for (int i = 0; i < 6; i++) *(--sp) = nullptr;
          f.sp = sp;
        }
        int ndone = 0;
        unsigned idle_rounds = 0;
        while (ndone < nthr) {
          unsigned long before = g_progress;
          ndone = 0;
          for (int i = 0; i < nthr; i++) {
            int t = g_reverse ? nthr - 1 - i : i;
            Fiber &f = g_fibers[t];
            if (f.done) { ndone++; continue; }
            g_cur = &f;
            emu_ctx_switch(&g_main_sp, f.sp);
            if (f.done) ndone++;
          }
          g_cur = nullptr;
          if (ndone < nthr && g_progress == before && g_spinning) {
            /* nobody moved, somebody polls another workgroup: give the processor away -- briefly at first, then for longer (the
               kernels give a poll 2^21 tries before they raise their time-out flag: this stretches that budget to minutes) */
            g_spinning = false;
            if (++idle_rounds < 2000) sched_yield(); else usleep(100);
            continue;
          }
          if (g_progress != before) idle_rounds = 0;
          g_spinning = false;
          if (ndone < nthr && g_progress == before) {
            fprintf(stderr, "hipemu: deadlock (divergent barrier?) in kernel %s block (%u,%u,%u)\n", name, bx, by, bz);
            abort();
          }
        }
      }
}
}  // namespace

void emu_launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body, const char *name) {
  const int nthr = (int)(block.x * block.y * block.z);
  if (nthr <= 0 || grid.x * grid.y * grid.z == 0) return;
  if (g_group_open) { emu_launch_concurrent(grid, block, shmem, body, name); return; }
  prepare_launch(grid, block, shmem, body, nthr);
  for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
      for (unsigned bx = 0; bx < grid.x; bx++) run_block(bx, by, bz, block, nthr, name);
  g_body = nullptr;
}

/* wait for exactly these processes; a failed one or a stall (10 minutes) ends them all */
static void wait_for(const std::vector<pid_t> &kids, const char *name) {
  const auto t0 = std::chrono::steady_clock::now();
  size_t left = kids.size();
  std::vector<char> done(kids.size(), 0);
  bool failed = false, asked = false;
  auto t_last = t0;
  while (left > 0 && !failed) {
    bool any = false;
    for (size_t i = 0; i < kids.size(); i++) {
      if (done[i]) continue;
      int st = 0;
      pid_t r = waitpid(kids[i], &st, WNOHANG);
      if (r == kids[i]) {
        done[i] = 1; left--; any = true;
        if (!(WIFEXITED(st) && WEXITSTATUS(st) == 0)) {
          failed = true;
          fprintf(stderr, "hipemu: workgroup process %zu of %s ended with %s %d\n", i, name, WIFSIGNALED(st) ? "signal" : "exit code", WIFSIGNALED(st) ? WTERMSIG(st) : WEXITSTATUS(st));
        }
      }
    }
    const auto now = std::chrono::steady_clock::now();
    if (any) t_last = now; else usleep(2000);
    if (!asked && std::chrono::duration<double>(now - t_last).count() > 20.0) {
      asked = true;                               /* no workgroup has finished for 20 s: ask the remaining ones where they are (once) */
      for (size_t i = 0; i < kids.size(); i++) if (!done[i]) kill(kids[i], SIGUSR1);
    }
    if (std::chrono::duration<double>(now - t0).count() > 600.0) {
      failed = true;
      fprintf(stderr, "hipemu: %s stalled (%zu workgroups still running after 10 minutes)\n", name, left);
    }
  }
  if (failed) {
    for (size_t i = 0; i < kids.size(); i++) if (!done[i]) { kill(kids[i], SIGKILL); waitpid(kids[i], nullptr, 0); }
    throw std::runtime_error(std::string("hipemu: a workgroup process of ") + name + " failed or the launch stalled");
  }
}

void emu_report_timeout() {                        /* a spin of a kernel gave up (sdm_raise_flag): say which workgroup and where */
  void *bt[16];
  const int n = backtrace(bt, 16);
  fprintf(stderr, "hipemu: wait timed out in block (%u,%u,%u), work-item %d\n", emu_blockIdx.x, emu_blockIdx.y, emu_blockIdx.z, g_cur ? g_cur->lin : -1);
  backtrace_symbols_fd(bt, n, 2);
}

static void wg_where(int) {                        /* SIGUSR1 from the parent of a launch that makes no progress: where is this workgroup? */
  void *bt[24];
  const int n = backtrace(bt, 24);
  fprintf(stderr, "hipemu: block (%u,%u,%u) is in work-item %d at\n", emu_blockIdx.x, emu_blockIdx.y, emu_blockIdx.z, g_cur ? g_cur->lin : -1);
  backtrace_symbols_fd(bt, n, 2);
}

static void wg_crash(int) {                        /* an out-of-bounds access of a workgroup: say where, then end the process */
  void *bt[32];
  const int n = backtrace(bt, 32);
  fprintf(stderr, "hipemu: SIGSEGV in block (%u,%u,%u), work-item %d\n", emu_blockIdx.x, emu_blockIdx.y, emu_blockIdx.z, g_cur ? g_cur->lin : -1);
  backtrace_symbols_fd(bt, n, 2);
  _exit(139);
}

void emu_launch_concurrent(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body, const char *name) {
  const int nthr = (int)(block.x * block.y * block.z);
  const unsigned nblk = grid.x * grid.y * grid.z;
  if (nthr <= 0 || nblk == 0) return;
  if (nblk > 256) throw std::runtime_error("hipemu: concurrent launch of more than 256 workgroups");
  prepare_launch(grid, block, shmem, body, nthr);
  fflush(nullptr);
  std::vector<pid_t> kids;
  for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
      for (unsigned bi = 0; bi < grid.x; bi++) {
        const unsigned bx = g_reverse_blocks ? grid.x - 1 - bi : bi;
        pid_t pid = fork();
        if (pid < 0) {
          kids.insert(kids.end(), g_group_kids.begin(), g_group_kids.end());
          g_group_kids.clear(); g_group_open = false;
          for (pid_t k : kids) { kill(k, SIGKILL); waitpid(k, nullptr, 0); }
          throw std::runtime_error("hipemu: fork failed");
        }
        if (pid == 0) {                              /* the workgroup's process: its statics (= LDS) are its own from here on */
          g_in_wg_process = true;
          signal(SIGSEGV, wg_crash);
          signal(SIGUSR1, wg_where);
          run_block(bx, by, bz, block, nthr, name);
          _exit(0);
        }
        kids.push_back(pid);
      }
  g_body = nullptr;
  if (g_group_open) { g_group_kids.insert(g_group_kids.end(), kids.begin(), kids.end()); g_group_names += std::string(g_group_names.empty() ? "" : " + ") + name; return; }
  wait_for(kids, name);
}

void emu_group_begin() { g_group_open = true; g_group_kids.clear(); g_group_names.clear(); }
void emu_group_end() {
  g_group_open = false;
  std::vector<pid_t> kids;
  kids.swap(g_group_kids);
  if (!kids.empty()) wait_for(kids, g_group_names.c_str());
}

struct emu_event { std::chrono::steady_clock::time_point t; };
hipError_t hipEventCreate(hipEvent_t *e) { *e = new emu_event(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
