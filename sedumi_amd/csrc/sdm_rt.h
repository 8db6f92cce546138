// sdm_rt.h -- device runtime include for the sedumi_amd HIP sources.
//
// Product build: hipcc --offload-arch=gfx950 (plain HIP for CDNA4, no other
// back-ends).  The SDM_EMU branch exists only so tests/hipemu can run the same
// kernel sources through its CPU fiber emulator inside the GPU-less build
// container; it is never compiled into libsedumi_hip.so.
#pragma once
#ifdef SDM_EMU
#include "hipemu.h"
typedef emu_double4 sdm_double4;
struct sdm_double2 { double x, y; };
#define SDM_MFMA_F64_16x16x4(a, b, c) emu_mfma_f64_16x16x4((a), (b), (c))
#define SDM_DYN_SMEM(name) char *name = emu_dyn_smem()
// value of `v` in lane `lane` (lane uniform across the wavefront), delivered to every lane
inline double sdm_bcast_lane(double v, int lane) { return emu_shfl(v, lane, 0); }
// wave-synchronous LDS hand-off between lanes of one wavefront: the emulator runs lanes as fibers and needs a
// real rendezvous; on the GPU the lanes are in lockstep and LDS operations of a wave complete in order
#define SDM_WAVE_SYNC() ((void)emu_shfl(0.0, 0, 0))
#define SDM_SETPRIO(n) do {} while (0)
// predicate of lane `lane` (uniform), delivered to every lane
inline bool sdm_lane_pred(bool pred, int lane) { return emu_shfl(pred ? 1.0 : 0.0, lane, 0) != 0.0; }
// completion counters and published data between workgroups of one launch.  The emulator runs workgroups one after the other --
// or, for a launch whose workgroups wait for each other (emu_launch_concurrent), as processes side by side over shared memory:
// hence real atomics, volatile accesses and fences here (x86-64: aligned 8-byte accesses are single copies, stores stay in order)
inline void sdm_signal_add(int *p, int n = 1) { __atomic_fetch_add(p, n, __ATOMIC_SEQ_CST); }
inline int sdm_ticket_take(int *p) { return __atomic_fetch_add(p, 1, __ATOMIC_SEQ_CST); }
inline void sdm_signal_reset(int *p) { __atomic_store_n(p, 0, __ATOMIC_SEQ_CST); }
inline void sdm_store_wt(double *p, double v) { *(volatile double *)p = v; }
inline double sdm_load_wt(const double *p) { return *(const volatile double *)p; }
inline void sdm_store_wt2(double *p, double a, double b) { ((volatile double *)p)[0] = a; ((volatile double *)p)[1] = b; }
inline unsigned long long sdm_load_wt_u64(const unsigned long long *p) { return *(const volatile unsigned long long *)p; }
// ("this WAVEFRONT's stores have been acknowledged": the lanes of a wavefront issue their stores together, so lane 0 may signal
// for all of them afterwards -- here the lanes are fibers that run one after the other, and the rendezvous is what makes the
// other lanes' stores precede lane 0's signal.  Every call site is wave-uniform; one that is not would stall the emulator.)
#define SDM_STORES_DONE() do { (void)emu_shfl(0.0, 0, 0); __atomic_thread_fence(__ATOMIC_SEQ_CST); } while (0)
inline int sdm_signal_load(const int *p) { return __atomic_load_n(p, __ATOMIC_SEQ_CST); }
inline void sdm_raise_flag(int *p) { __atomic_store_n(p, 1, __ATOMIC_SEQ_CST); emu_report_timeout(); }
inline bool sdm_flag_raised(const int *p) { return __atomic_load_n(p, __ATOMIC_SEQ_CST) != 0; }
inline void sdm_host_note(int *p, int v) { __atomic_store_n(p, v, __ATOMIC_SEQ_CST); }
#define SDM_UNIFORM_INT(x) (x)
#define SDM_ACQUIRE_FENCE() __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define SDM_COMPILER_BARRIER() __asm__ __volatile__("" ::: "memory")
#define SDM_SPIN_PAUSE() emu_spin_pause()
#else
#include <hip/hip_runtime.h>
typedef double sdm_double4 __attribute__((ext_vector_type(4)));
typedef double2 sdm_double2;
#define SDM_MFMA_F64_16x16x4(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
#define SDM_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#define SDM_LAUNCH(kernel, grid, block, shmem, stream, ...) \
  hipLaunchKernelGGL(kernel, (grid), (block), (shmem), (stream), __VA_ARGS__)
#define SDM_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
// issue priority of the calling wavefront (s_setprio): the wave on a kernel's dependency chain ahead of its helpers
#define SDM_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
// completion counters between workgroups of one launch: release on the producer side (after __threadfence() and a
// barrier), acquire on the consumer side, device scope (the workgroups may sit on different XCDs / L2s)
// The producers' data stores are write-through to the device-wide coherence point (sdm_store_wt: agent-scope relaxed
// atomic stores -- a release fence instead would write back the whole L2 of the XCD, microseconds per workgroup when a
// big trailing matrix is dirty); SDM_STORES_DONE waits for this wavefront's stores (an explicit s_waitcnt vmcnt(0):
// a workgroup-scope release fence emits NOTHING for global stores on gfx950, and inline asm is invisible to the
// compiler pass that drops waits it believes redundant -- MI355X_MICROARCH.md "Compiler hazard"), a barrier collects
// the workgroup, then one relaxed increment publishes it.  tests/test_abi.py checks the disassembly for the wait.
__device__ __forceinline__ void sdm_signal_add(int *p, int n = 1) { __hip_atomic_fetch_add(p, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// a ticket: the value before the increment (device scope)
__device__ __forceinline__ int sdm_ticket_take(int *p) { return __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// a counter back to 0 for a LATER launch (device scope: the increments act at the coherence point, the zero must be there too)
__device__ __forceinline__ void sdm_signal_reset(int *p) { __hip_atomic_store(p, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sdm_store_wt(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 16 bytes write-through in one instruction (p 16-byte aligned): a row of published data as full-width fabric writes instead of
// one 8-byte write per lane and word (MI355X_MICROARCH.md: scalar sc1 stores cost 2.7x the dwordx4 time per byte)
__device__ __forceinline__ void sdm_store_wt2(double *p, double a, double b) {
  typedef double v2d_ __attribute__((ext_vector_type(2)));
  v2d_ v; v.x = a; v.y = b;
  // (the s_nop: a VMEM store of more than 8 bytes must not be followed at once by a write of its data registers -- a hazard
  // the compiler pads for its own stores and cannot see inside inline assembly; without it the rows came out corrupted)
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}
// the matching read: an sc1 load bypasses this CU's L1 (served by L2 / memory), so data another workgroup published with
// sdm_store_wt needs NO acquire fence (1.7 us: MI355X_MICROARCH.md, price list) before it is read this way
__device__ __forceinline__ double sdm_load_wt(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long sdm_load_wt_u64(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#define SDM_STORES_DONE() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// (relaxed: a poll must not invalidate caches -- with dozens of workgroups polling, acquire loads kept every L2 of the device
// cold; the ONE acquire fence a consumer needs comes after its wait has ended, SDM_ACQUIRE_FENCE in spin_until / prep_wait)
__device__ __forceinline__ int sdm_signal_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// error flag in pinned host memory (HostFlag): one system-scope store, read by the host after a stream synchronise
__device__ __forceinline__ void sdm_raise_flag(int *p) { __hip_atomic_store(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ bool sdm_flag_raised(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0; }
// a note for the host in pinned memory that is no error (read without synchronising: late is fine)
__device__ __forceinline__ void sdm_host_note(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
#define SDM_ACQUIRE_FENCE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
// a wave-uniform integer the compiler cannot prove uniform (e.g. threadIdx.x >> 6): moved to a scalar register, so that
// addresses built from it stay scalar and loads through them become s_load
#define SDM_UNIFORM_INT(x) __builtin_amdgcn_readfirstlane(x)
#define SDM_SPIN_PAUSE() __builtin_amdgcn_s_sleep(4)
// no memory access moves across this point at compile time
#define SDM_COMPILER_BARRIER() asm volatile("" ::: "memory")
// predicate of lane `lane` (uniform), delivered to every lane: one compare into a lane mask, one scalar bit test
__device__ __forceinline__ bool sdm_lane_pred(bool pred, int lane) { return (__ballot(pred) >> lane) & 1ull; }
// v_readlane_b32 x2: a scalar broadcast, no LDS crossbar round trip (ds_bpermute) on the dependency chain
__device__ __forceinline__ double sdm_bcast_lane(double v, int lane) {
  union { double d; int i[2]; } u;
  u.d = v;
  u.i[0] = __builtin_amdgcn_readlane(u.i[0], lane);
  u.i[1] = __builtin_amdgcn_readlane(u.i[1], lane);
  return u.d;
}
#endif

// Bounded waits between workgroups of one launch: poll number `it` of a wait gives up -- true -- at the bound (2^19 polls of
// 0.5 - 1.5 us each: a fraction of a second, against legitimate waits of at most a few milliseconds; it raises the plan's time-out
// flag in pinned host memory), or as soon as ANOTHER wait of the plan has given up: the flag is looked at (a PCIe round trip) when
// a wait has lasted 64, 128, 256, ... polls -- never in a healthy hand-over on the chain, a few times in the long idle waits --
// so a starved launch drains within milliseconds of its first time-out instead of paying the bound once per wait still ahead.
#ifdef SDM_EMU
#define SDM_SPIN_MAX (1L << 21)          // (processes sharing a few cores: polls are cheap here and legitimate waits long)
inline
#else
#ifndef SDM_SPIN_MAX
#define SDM_SPIN_MAX (1L << 19)
#endif
__device__ __forceinline__
#endif
bool sdm_spin_giveup(long it, int *tmo) {
  if (it + 1 >= SDM_SPIN_MAX) { sdm_raise_flag(tmo); return true; }
  return it >= 63 && ((it + 1) & it) == 0 && sdm_flag_raised(tmo);
}

// Statement-level switch: keep a*b-c as two roundings (gcc -O2 on x86-64 emits no FMA for the reference's
// daxpy loops); used where a pivot accept/skip decision depends on noise-level values (blkchol2.c:114-161).
#ifdef SDM_EMU
#define SDM_FP_STRICT do {} while (0)
#define SDM_PIN(x) do {} while (0)
#define SDM_ZERO_AFTER(v) 0
#else
#define SDM_FP_STRICT _Pragma("clang fp contract(off)")
// keep a value materialised in a VGPR at this point (stops the scheduler from sinking its load next to the use)
#define SDM_PIN(x) asm volatile("" : "+v"(x))
// an integer 0 the compiler must assume to depend on `v`: added to an address it keeps the load BEHIND the point where
// v is final (stops the scheduler from hoisting every batch of coefficients to the top and spilling them)
__device__ __forceinline__ int sdm_zero_after(double v) { int z = 0; asm volatile("" : "+v"(z) : "v"(v)); return z; }
#define SDM_ZERO_AFTER(v) sdm_zero_after(v)
#endif

// Optional in-kernel phase clocks (tools/ubench builds only, -DSDM_PHASES): work-item 0 accumulates wall_clock64
// ticks (100 MHz) between marks into sdm_phase_acc[].  Compiled out of the product library.
#if defined(SDM_PHASES) && !defined(SDM_EMU)
static __device__ unsigned long long sdm_phase_acc[32];
#define SDM_PHASE_BEGIN() long long ph_t_ = wall_clock64()
#define SDM_PHASE(n) do { const long long t_ = wall_clock64(); if (threadIdx.x == 0) atomicAdd(&sdm_phase_acc[n], (unsigned long long)(t_ - ph_t_)); ph_t_ = t_; } while (0)
// the same through an LDS accumulator, flushed once at the end: no global atomics inside the timed region (their
// completion would be waited for by whatever vmcnt wait the compiler placed in the loop)
#define SDM_LPHASE_BEGIN() __shared__ unsigned long long ph_l_[32]; if (threadIdx.x < 32) ph_l_[threadIdx.x] = 0; long long ph_t_ = wall_clock64()
#define SDM_LPHASE(n) do { const long long t_ = wall_clock64(); if (threadIdx.x == 0) ph_l_[n] += (unsigned long long)(t_ - ph_t_); ph_t_ = t_; } while (0)
#define SDM_LPHASE_END() do { if (threadIdx.x == 0) for (int i_ = 0; i_ < 32; i_++) if (ph_l_[i_]) atomicAdd(&sdm_phase_acc[i_], ph_l_[i_]); } while (0)
// time stamps of single events (tools/trace_front.py): sdm_trace_buf[slot] = wall_clock64() by work-item 0 of the workgroup
static __device__ long long sdm_trace_buf[2048];
#define SDM_TRACE(slot) do { if (threadIdx.x == 0 && (slot) >= 0 && (slot) < 2048) sdm_trace_buf[(slot)] = wall_clock64(); } while (0)
// the same for lane 0 of whichever wavefront executes it (the caller picks the wavefront)
#define SDM_WPHASE_BEGIN() long long wph_t_ = wall_clock64()
#define SDM_WPHASE(n) do { const long long t_ = wall_clock64(); if ((threadIdx.x & 63) == 0) atomicAdd(&sdm_phase_acc[n], (unsigned long long)(t_ - wph_t_)); wph_t_ = t_; } while (0)
#else
#define SDM_WPHASE_BEGIN() do {} while (0)
#define SDM_WPHASE(n) do {} while (0)
#define SDM_TRACE(slot) do {} while (0)
#define SDM_PHASE_BEGIN() do {} while (0)
#define SDM_PHASE(n) do {} while (0)
#define SDM_LPHASE_BEGIN() do {} while (0)
#define SDM_LPHASE(n) do {} while (0)
#define SDM_LPHASE_END() do {} while (0)

#endif

// End the wavefront from inside a device function whose caller has nothing left to do (the roles of k_ldl_panel): the function's
// epilogue -- reloading the callee-saved registers its prologue parked in scratch, then the return -- is dead code that way.
// SDM_NORETURN marks such a function for its callers: nothing of theirs has to survive the call.
#ifdef SDM_EMU
#define SDM_ENDPGM() return
#define SDM_NORETURN
#else
#define SDM_ENDPGM() __builtin_amdgcn_endpgm()
#define SDM_NORETURN __attribute__((noreturn))
#endif

// A pointer that a __noinline__ device function receives as a plain parameter is a GENERIC pointer to the compiler: every access
// through it becomes a FLAT instruction, and flat instructions count in lgkmcnt as well as vmcnt -- so the next wait for an LDS
// read also waits for every outstanding global load and for the acknowledgement of every write-through store (microseconds on
// the chain of the factor kernels).  Such functions therefore take their global-memory pointers as SDM_GP(T) -- address space 1,
// what kernel arguments are -- and convert them to plain pointers at the top: the address-space inference follows the conversion
// and the accesses are global_load / global_store / global_atomic again.  Callers cast: (SDM_GP(T))p.
#ifdef SDM_EMU
#define SDM_GP(T) T *
#else
#define SDM_GP(T) __attribute__((address_space(1))) T *
#endif

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>

#define SDM_HIP_CHECK(expr)                                                          \
  do {                                                                               \
    hipError_t e_ = (expr);                                                          \
    if (e_ != hipSuccess)                                                            \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e_) + \
                               " at " __FILE__ ":" + std::to_string(__LINE__));      \
  } while (0)

typedef int64_t sdm_int;  // host-side index type of the C ABI
