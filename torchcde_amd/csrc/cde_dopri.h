// cde_dopri.h -- controller state and Dormand-Prince tableau shared by the adaptive kernels (dopri5.hip,
// dopri5_adjoint.hip).
#pragma once
#include "cde_mfma.h"

namespace cde {

struct DopriCtrl {
  double t_lo, t_hi, dt;        // dense-output interval of the last accepted step, next step size
  double t1_try, dt_try;        // the attempt whose partial sums are pending
  double h0;                    // Hairer initial step, phase 1 -> 2
  int64_t i_out, i_jump;        // next output index, next jump index
  int64_t n_accept, n_reject;
  int32_t phase;                // 0 start, 1 after f0 norms, 2 after f1 norm, 3 stepping, 4 done
  int32_t on_jump;              // pending attempt was clipped to a jump time
  int32_t refresh;              // k0 must be recomputed just after t_hi (we stepped onto a jump)
  int32_t pad;
  int32_t slot;                 // MFMA attempt kernel: which of the two (y, k) state slots holds the step's start
  int32_t stored;               // ... and what the pending attempt left in the other one: bit 0 k6, bit 1 the midpoint
  int32_t hint_lo, hint_hi;     // ... and the knot intervals of the pending attempt's first and last stage times: the next
                                // launch requests the control rows of its two likely intervals BEFORE it knows the decision
};
static_assert(sizeof(DopriCtrl) == 112, "cde_dopri5_status (include/cde_mi355x.h) and _lib.DopriStatus mirror this layout");

// wave-uniform copies: the controller's outputs derive from LDS reads (the block sums), so the compiler keeps them --
// and everything computed from them -- in vector registers; read back through lane 0 they live in scalar registers
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uni(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }
__device__ __forceinline__ double uni(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
__device__ __forceinline__ int64_t uni(int64_t v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// Sums of NV doubles over the workgroup, the result in every thread.  Fixed order (xor-shuffle tree inside each wave,
// then the waves in index order): bit-identical in every workgroup that sums the same values, and run to run.
// `red`: NV * (blockDim.x / 64) doubles of LDS.
template <int NV>
__device__ __forceinline__ void block_total(double (&v)[NV], double* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v[k] += __shfl_xor(v[k], off, 64);
  __syncthreads();                                               // `red` may still be in use by an earlier call
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < NV; ++k) red[k * nw + wave] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    double s = 0.0;
    for (int w = 0; w < nw; ++w) s += red[k * nw + w];
    v[k] = s;
  }
}

__device__ __forceinline__ float next_toward(float x, float dir) { return nextafterf(x, x + dir); }
__device__ __forceinline__ double next_toward(double x, double dir) { return nextafter(x, x + dir); }

// Dormand-Prince tableau (identical numbers to oracle/odeint.py)
__device__ constexpr double DP_ALPHA[6] = {1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1.0, 1.0};
__device__ constexpr double DP_BETA[6][6] = {
    {1.0 / 5, 0, 0, 0, 0, 0},
    {3.0 / 40, 9.0 / 40, 0, 0, 0, 0},
    {44.0 / 45, -56.0 / 15, 32.0 / 9, 0, 0, 0},
    {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729, 0, 0},
    {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656, 0},
    {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84}};
__device__ constexpr double DP_CERR[7] = {35.0 / 384 - 1951.0 / 21600, 0, 500.0 / 1113 - 22642.0 / 50085,
                                          125.0 / 192 - 451.0 / 720, -2187.0 / 6784 - -12231.0 / 42400,
                                          11.0 / 84 - 649.0 / 6300, -1.0 / 60};
__device__ constexpr double DP_CMID[7] = {6025192743.0 / 30085553152.0 / 2, 0, 51252292925.0 / 65400821598.0 / 2,
                                          -2691868925.0 / 45128329728.0 / 2, 187940372067.0 / 1594534317056.0 / 2,
                                          -1776094331.0 / 19743644256.0 / 2, 11237099.0 / 235043384.0 / 2};

}  // namespace cde
