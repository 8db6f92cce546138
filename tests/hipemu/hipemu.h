/*
 * hipemu.h -- TEST INFRASTRUCTURE ONLY (tests/hipemu).
 *
 * A tiny single-threaded emulator of the subset of the HIP programming model
 * that sedumi_amd/csrc uses, so that the *same kernel sources* can be compiled
 * with g++ (-DSDM_EMU) and executed on the CPU-only build container to test
 * kernel logic and host orchestration before spending scarce MI355X time.
 * Every workgroup is run as a ring of cooperative fibers (one per work-item);
 * __syncthreads() and the wave-level operations (shuffles, MFMA) are barrier
 * points at which a fiber yields to the next one.
 *
 * This is NOT a product code path: sedumi_amd/ only ever loads
 * libsedumi_hip.so (built by hipcc for gfx950) and fails loudly when it is
 * missing.  The emulated library (tests/hipemu/libsedumi_hipemu.so) is built
 * and loaded by the tests alone.
 */
#ifndef SDM_HIPEMU_H
#define SDM_HIPEMU_H

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <algorithm>
using std::min;
using std::max;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __noinline__
#define __launch_bounds__(...)
#define __restrict__

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct emu_uint3 { unsigned x, y, z; };

extern emu_uint3 emu_blockIdx, emu_blockDim, emu_gridDim;
emu_uint3 &emu_threadIdx_ref();
#define threadIdx (emu_threadIdx_ref())
#define blockIdx emu_blockIdx
#define blockDim emu_blockDim
#define gridDim emu_gridDim

/* __shared__: one workgroup runs at a time, so a static is exactly "per block" */
#define __shared__ static
char *emu_dyn_smem();

void emu_syncthreads();
#define __syncthreads() emu_syncthreads()
inline void __threadfence() {}

typedef int hipError_t;
typedef void *hipStream_t;
typedef struct emu_event *hipEvent_t;
enum { hipSuccess = 0, hipErrorUnknown = 999 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };

inline const char *hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
/* "device" and pinned memory: anonymous SHARED mappings, so that the workgroup processes of a concurrent launch
   (emu_launch_concurrent) and the host see the same bytes */
void *emu_shared_alloc(size_t n);
void emu_shared_free(void *p);
inline hipError_t hipMalloc(void **p, size_t n) { *p = emu_shared_alloc(n ? n : 1); return *p ? hipSuccess : hipErrorUnknown; }
template <class T> inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
inline hipError_t hipFree(void *p) { emu_shared_free(p); return hipSuccess; }
inline hipError_t hipHostMalloc(void **p, size_t n, unsigned = 0) { *p = emu_shared_alloc(n ? n : 1); return *p ? hipSuccess : hipErrorUnknown; }
inline hipError_t hipHostFree(void *p) { emu_shared_free(p); return hipSuccess; }
inline hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned = 0) { *d = h; return hipSuccess; }
inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = 0) { if (n) memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void *d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = 0) { if (n) memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t *s) { *s = 0; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = 0);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);

/* graphs: the emulator executes eagerly, so a "captured" sequence has already run once; replay is not supported */
typedef struct emu_graph *hipGraph_t;
typedef struct emu_graphexec *hipGraphExec_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1 };
inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorUnknown; }
inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t *) { return hipErrorUnknown; }
inline hipError_t hipGraphInstantiate(hipGraphExec_t *, hipGraph_t, void *, void *, size_t) { return hipErrorUnknown; }
inline hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
inline hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
inline hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorUnknown; }

/* launch: runs every block sequentially, each as blockDim.x*y*z fibers */
void emu_launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body, const char *name = "?");
#define SDM_LAUNCH(kernel, grid, block, shmem, stream, ...) \
  emu_launch((grid), (block), (shmem), [&]() { kernel(__VA_ARGS__); }, #kernel)
/* launch of a kernel whose workgroups WAIT FOR EACH OTHER (k_ldl_front): one forked process per workgroup, all running
   at once over the shared "device" memory, each a ring of fibers as above; throws if one of them fails or they stall.
   Only used when the test asked for it (emu_set_concurrent): the default remains the phase-by-phase sequential form. */
void emu_launch_concurrent(dim3 grid, dim3 block, size_t shmem, const std::function<void()> &body, const char *name = "?");
#define SDM_LAUNCH_CONCURRENT(kernel, grid, block, shmem, ...) \
  emu_launch_concurrent((grid), (block), (shmem), [&]() { kernel(__VA_ARGS__); }, #kernel)
/* launches between the two run side by side (a kernel and the follower that polls its progress from a second stream): every
   launch in between forks its workgroups and returns; _end waits for all of them */
void emu_group_begin();
void emu_group_end();
void emu_set_concurrent(int on);
/* fault injection for the host-side recovery paths: the next n launches that ask (emu_take_injected_timeout) are told that
   one of their workgroups gave up waiting */
void emu_inject_timeouts(int n);
int emu_take_injected_timeout();
int emu_concurrent();
void emu_set_reverse_blocks(int on);
/* inside a spin loop on another workgroup's progress: yields the processor in a workgroup process, aborts in a sequential
   launch (there the other workgroup has either run already or never will) */
void emu_spin_pause();
void emu_report_timeout();

/* ---- wave-level operations (wave = 64 consecutive linear thread ids) ---- */
double emu_shfl(double v, int srcLane, int mode);  /* mode 0: idx, 1: down(delta), 2: xor(mask) */
inline double __shfl(double v, int src) { return emu_shfl(v, src, 0); }
inline double __shfl_down(double v, int d) { return emu_shfl(v, d, 1); }
inline double __shfl_xor(double v, int m) { return emu_shfl(v, m, 2); }
inline int __shfl(int v, int src) { return (int)emu_shfl((double)v, src, 0); }
inline int __shfl_down(int v, int d) { return (int)emu_shfl((double)v, d, 1); }
inline int __shfl_xor(int v, int m) { return (int)emu_shfl((double)v, m, 2); }

struct emu_double4 {
  double v[4];
  double &operator[](int i) { return v[i]; }
  const double &operator[](int i) const { return v[i]; }
};
/* v_mfma_f64_16x16x4_f64: lane l supplies A[l&15][l>>4], B[l>>4][l&15];
   result reg r of lane l is D[(l>>4)+4r][l&15]  (cdna_hip_programming.md section 3) */
emu_double4 emu_mfma_f64_16x16x4(double a, double b, emu_double4 c);

inline double atomicAdd(double *p, double v) { double o = *p; *p = o + v; return o; }   /* (no kernel of a concurrent launch adds doubles) */
inline int atomicAdd(int *p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned atomicAdd(unsigned *p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline int atomicMax(int *p, int v) { int o = *p; if (v > o) *p = v; return o; }
inline unsigned long long atomicMax(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; if (v > o) *p = v; return o; }
inline int atomicExch(int *p, int v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }

/* scheduling order knob: 0 ascending lanes, 1 descending (exposes missing barriers) */
void emu_set_reverse(int rev);

#endif
