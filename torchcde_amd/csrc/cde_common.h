// cde_common.h -- device helpers shared by every kernel of libcde_mi355x.so (gfx950 only).
//
// The library is compiled with -ffp-contract=off: wherever the reference's float result is
// defined by a sequence of separately rounded torch ops (knot lookup, spline Horner steps, RK
// stage combinations, output interpolation) the kernels reproduce that sequence literally;
// fused multiply-adds appear only where written explicitly (dot products, MFMA).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/cde_mi355x.h"

namespace cde {

// ---------------------------------------------------------------- interval lookup
// index = clamp(bucketize(t, knots, right=False) - 1, 0, n_intervals - 1);  frac = t - knots[index]
// (reference interpolation_cubic.py:315-322).  bucketize(right=False) is lower_bound with the
// comparison torch uses, `!(knots[mid] >= t)`, so a query exactly on knot k resolves to interval
// k-1 and NaN queries behave identically.
template <typename T>
__device__ __forceinline__ int64_t locate(const T* __restrict__ knots, int64_t n_intervals, T t, T& frac) {
  int64_t lo = 0, hi = n_intervals + 1;  // n_intervals + 1 knots
  while (lo < hi) {
    const int64_t mid = lo + ((hi - lo) >> 1);
    if (!(knots[mid] >= t)) lo = mid + 1; else hi = mid;
  }
  int64_t idx = lo - 1;
  idx = idx < 0 ? 0 : idx;
  idx = idx > n_intervals - 1 ? n_intervals - 1 : idx;
  frac = t - knots[idx];
  return idx;
}

// locate() when the answer of a nearby earlier query is known (the stages of one adaptive step almost always share
// their interval): two comparisons instead of the search; identical result in every case, NaN queries included
// (interior idx <=> !(knots[idx] >= t) && knots[idx+1] >= t; the first / last interval also take everything
// below / above them because of the clamp).
template <typename T>
__device__ __forceinline__ int64_t locate_near(const T* __restrict__ knots, int64_t n_intervals, T t, int64_t hint, T& frac) {
  if (hint >= 0) {
    const T k0 = knots[hint], k1 = knots[hint + 1];
    const bool above = hint == 0 || !(k0 >= t);
    const bool below = hint == n_intervals - 1 ? !(k0 >= t) || hint == 0 : k1 >= t;
    if (above && below) { frac = t - k0; return hint; }
  }
  return locate(knots, n_intervals, t, frac);
}

// ... and when the answer is `hint` or one of its neighbours (consecutive steps of an adaptive solve): four
// independent loads instead of the search's chain of dependent ones.
template <typename T>
__device__ __forceinline__ int64_t locate_around(const T* __restrict__ knots, int64_t n_intervals, T t, int64_t hint, T& frac) {
  if (hint >= 0 && hint < n_intervals) {
    const int64_t lo = hint > 0 ? hint - 1 : 0, hi = hint < n_intervals - 1 ? hint + 1 : n_intervals - 1;
    T k[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) k[j] = knots[lo + j <= n_intervals ? lo + j : n_intervals];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int64_t cand = lo + j;
      if (cand > hi) break;
      const bool above = cand == 0 || !(k[j] >= t);
      const bool below = cand == n_intervals - 1 ? !(k[j] >= t) || cand == 0 : k[j + 1] >= t;
      if (above && below) { frac = t - k[j]; return cand; }
    }
  }
  return locate(knots, n_intervals, t, frac);
}

// locate_around in two halves, for kernels that know the hint long before they know the query: the four knots around
// `hint` are REQUESTED early (knot_window), the comparisons happen when t is known (locate_window: same result as locate()
// in every case -- it falls back to the search when t is in none of the three intervals).
template <typename T>
struct KnotWindow { T k[4]; int64_t lo, hi; bool valid; };
template <typename T>
__device__ __forceinline__ KnotWindow<T> knot_window(const T* __restrict__ knots, int64_t n_intervals, int64_t hint) {
  KnotWindow<T> w;
  w.valid = hint >= 0 && hint < n_intervals;
  const int64_t h = w.valid ? hint : 0;
  w.lo = h > 0 ? h - 1 : 0;
  w.hi = h < n_intervals - 1 ? h + 1 : n_intervals - 1;
#pragma unroll
  for (int j = 0; j < 4; ++j) w.k[j] = knots[w.lo + j <= n_intervals ? w.lo + j : n_intervals];
  return w;
}
template <typename T>
__device__ __forceinline__ int64_t locate_window(const KnotWindow<T>& w, const T* __restrict__ knots, int64_t n_intervals, T t, T& frac) {
  if (w.valid) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int64_t cand = w.lo + j;
      if (cand > w.hi) break;
      const bool above = cand == 0 || !(w.k[j] >= t);
      const bool below = cand == n_intervals - 1 ? !(w.k[j] >= t) || cand == 0 : w.k[j + 1] >= t;
      if (above && below) { frac = t - w.k[j]; return cand; }
    }
  }
  return locate(knots, n_intervals, t, frac);
}

// ---------------------------------------------------------------- RK4 3/8-rule stage clock
// torchdiffeq rk4_alt_step_func: stage times t0, t0 + dt*(1/3), t0 + dt*(2/3), t1 formed in the
// grid's dtype (python floats 1/3, 2/3 rounded to that dtype), then cast to the state dtype.
template <typename TT>
struct StageClock {
  TT t0, t1, dt;
  __device__ __forceinline__ StageClock(TT a, TT b) : t0(a), t1(b), dt(b - a) {}
  __device__ __forceinline__ TT time(int stage) const {
    const TT third = (TT)(1.0 / 3.0), two_thirds = (TT)(2.0 / 3.0);
    switch (stage) {
      case 0: return t0;
      case 1: return t0 + dt * third;
      case 2: return t0 + dt * two_thirds;
      default: return t1;
    }
  }
};

// ---------------------------------------------------------------- control derivative of one channel
// cubic row layout (one interval of one series): [a(C) | b(C) | two_c(C) | three_d(C)]
//   derivative = b + (two_c + three_d*frac)*frac          (interpolation_cubic.py:334-335)
//   value      = a + (b + (0.5*two_c + three_d*frac/3)*frac)*frac   (:327-329)
template <typename T>
__device__ __forceinline__ T cubic_derivative(T b, T two_c, T three_d, T frac) {
  const T inner = two_c + three_d * frac;
  return b + inner * frac;
}
template <typename T>
__device__ __forceinline__ T cubic_value(T a, T b, T two_c, T three_d, T frac) {
  T inner = (T)0.5 * two_c + three_d * frac / (T)3;
  inner = b + inner * frac;
  return a + inner * frac;
}

__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float tanh_t(float x) { return tanhf(x); }
__device__ __forceinline__ double tanh_t(double x) { return tanh(x); }

// The tuning table (include/cde_mi355x.h, CDE_OPT_*): written by cde_set_option only, read by the layout functions and the
// launchers.  Defined in interp_kernels.hip.
int64_t option(int key);

inline int check_launch() { return hipGetLastError() == hipSuccess ? CDE_OK : CDE_ERR_LAUNCH; }

// Zero-fill as a KERNEL, never hipMemsetAsync: captured into a hipGraph, the memset node of this ROCm zeroes its target on
// the first replay only (found by tests/test_gpu_08_frontend.py::test_solver_calls_are_graph_capturable: the second replay of
// the wide adjoint added to the first one's sums).  `bytes` must be a multiple of 4.
static __global__ __launch_bounds__(256) void zero_words_kernel(unsigned* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = 0u;
}
static inline int zero_async(void* p, size_t bytes, hipStream_t s) {
  const size_t n = bytes / 4;
  if (n == 0) return CDE_OK;
  size_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  zero_words_kernel<<<(unsigned)blocks, 256, 0, s>>>((unsigned*)p, n);
  return CDE_OK;
}

// ------------------------------------------------------------------------------------------ phase trace (debug builds)
// A runtime stage number as a compile-time constant: `fn(std::integral_constant<int, i>{})` for i in 0..6.  Lets a ROLLED
// stage loop (small enough for the instruction cache) index register-resident arrays and Butcher rows statically.
#define CDE_STATIC_STAGE(i, fn)                                                                                       \
  switch (i) {                                                                                                        \
    case 0: fn(std::integral_constant<int, 0>{}); break;                                                              \
    case 1: fn(std::integral_constant<int, 1>{}); break;                                                              \
    case 2: fn(std::integral_constant<int, 2>{}); break;                                                              \
    case 3: fn(std::integral_constant<int, 3>{}); break;                                                              \
    case 4: fn(std::integral_constant<int, 4>{}); break;                                                              \
    case 5: fn(std::integral_constant<int, 5>{}); break;                                                              \
    default: fn(std::integral_constant<int, 6>{}); break;                                                             \
  }

// CDE_PHASE_TRACE=1 builds libcde_mi355x_trace.so (torchcde_amd/_lib.py): the attempt kernels stamp the 100 MHz
// wall clock (s_memrealtime: one time base for the whole chip) at their phase boundaries, wave 0 of every workgroup
// writes its stamps to a ring indexed by the attempt number, and scripts/phase_trace.py turns the ring into the
// per-phase split of an attempt (launch gap, prologue, stages, drain).  The product build compiles none of it.
#ifdef CDE_PHASE_TRACE
constexpr int TRACE_SLOTS = 40, TRACE_BLOCKS = 512, TRACE_RING = 32;   // slots 20..39: a second wave's stamps (CDE_STAMP_FLUSH2)
struct PhaseStamps { unsigned long long t[TRACE_SLOTS]; };
#define CDE_STAMP_DECL ::cde::PhaseStamps stamps_ = {}
#define CDE_STAMP(slot) do { __builtin_amdgcn_sched_barrier(0); stamps_.t[slot] = wall_clock64(); \
                             __builtin_amdgcn_sched_barrier(0); } while (0)
#define CDE_STAMP_FLUSH(buf, attempt) do { if (threadIdx.x == 0 && blockIdx.x < ::cde::TRACE_BLOCKS) { \
    unsigned long long* dst_ = (buf) + ((size_t)((attempt) % ::cde::TRACE_RING) * ::cde::TRACE_BLOCKS + blockIdx.x) * ::cde::TRACE_SLOTS; \
    for (int s_ = 0; s_ < 20; ++s_) dst_[s_] = stamps_.t[s_]; } } while (0)
// a second stamping wave (its lane 0 is thread `tid2`) owns slots 20..39
#define CDE_STAMP_FLUSH2(buf, attempt, tid2) do { if (threadIdx.x == (tid2) && blockIdx.x < ::cde::TRACE_BLOCKS) { \
    unsigned long long* dst_ = (buf) + ((size_t)((attempt) % ::cde::TRACE_RING) * ::cde::TRACE_BLOCKS + blockIdx.x) * ::cde::TRACE_SLOTS; \
    for (int s_ = 20; s_ < ::cde::TRACE_SLOTS; ++s_) dst_[s_] = stamps_.t[s_]; } } while (0)
#define CDE_STAMP_IF(cond, slot) do { if (cond) CDE_STAMP(slot); } while (0)
#else
#define CDE_STAMP_FLUSH2(buf, attempt, tid2) do {} while (0)
#define CDE_STAMP_IF(cond, slot) do {} while (0)
#define CDE_STAMP_DECL do {} while (0)
#define CDE_STAMP(slot) do {} while (0)
#define CDE_STAMP_FLUSH(buf, attempt) do {} while (0)
#endif

}  // namespace cde
