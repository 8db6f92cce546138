/*
 * hipemu.cpp -- TEST INFRASTRUCTURE ONLY (tests/hipemu).  See hipemu.h.
 * Cooperative-fiber execution of one workgroup at a time (x86-64 only).
 */
#include "hipemu.h"
#include <chrono>
#include <vector>

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

void emu_launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body, const char *name) {
  const int nthr = (int)(block.x * block.y * block.z);
  if (nthr <= 0 || grid.x * grid.y * grid.z == 0) return;
  if (nthr > 1024) { fprintf(stderr, "hipemu: block too large\n"); abort(); }
  while ((int)g_stack_pool.size() < nthr) g_stack_pool.push_back((char *)malloc(STACK_BYTES));
  if (g_dyn_smem.size() < shmem + 64) g_dyn_smem.resize(shmem + 64);
  const int nwaves = (nthr + 63) / 64;
  g_wave_scr.assign((size_t)nwaves * 192, 0.0);
  emu_blockDim = {block.x, block.y, block.z};
  emu_gridDim = {grid.x, grid.y, grid.z};
  g_body = &body;
  for (unsigned bz = 0; bz < grid.z; bz++)
    for (unsigned by = 0; by < grid.y; by++)
      for (unsigned bx = 0; bx < grid.x; bx++) {
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
          if (ndone < nthr && g_progress == before) {
            fprintf(stderr, "hipemu: deadlock (divergent barrier?) in kernel %s block (%u,%u,%u)\n", name, bx, by, bz);
            abort();
          }
        }
      }
  g_body = nullptr;
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
