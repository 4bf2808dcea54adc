// interp_kernels.hip -- K1 (Hermite fit), K1b (interval lookup + path evaluation), contraction.
// HBM-bound streaming kernels: one lane per 16-byte output vector, fully coalesced stores.
#include "cde_common.h"

namespace cde {

template <typename T, int V>
struct Vec {
  T v[V];
};

// ------------------------------------------------------------------------------------------ K1
// One lane produces V consecutive floats of one output row [a|b|2c|3d]; V divides C, so a lane's
// elements share one coefficient kind and cover V consecutive channels.  Consecutive lanes write
// consecutive 16-B pieces: a wave stores 1 KiB contiguous.  x is re-read from L1/L2 (each x row is
// touched by the 3 neighbouring intervals x 4 kinds); HBM sees x once and coeffs once.
//
// Arithmetic follows interpolation_hermite_cubic_bdiff.py:39 and :10-18 literally:
//   secant_i = (x[i+1]-x[i]) / (t[i+1]-t[i])
//   enter_i  = secant_{i-1}   (enter_0 = secant_0)
//   two_c    = 2*(3*((x[i+1]-x[i])/h - enter) - secant + enter) / h
//   three_d  = (1/h**2)*(secant - enter) - two_c/h
template <typename T, int V>
__global__ __launch_bounds__(256) void hermite_bdiff_kernel(const T* __restrict__ x, const T* __restrict__ t,
                                                            T* __restrict__ out, int64_t B, int64_t L, int64_t C) {
  const int64_t row_vecs = 4 * C / V;
  const int64_t n_rows = B * (L - 1);
  const int64_t total = n_rows * row_vecs;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = e / row_vecs;
    const int64_t q = (e - row * row_vecs) * V;  // first element inside the 4C-wide row
    const int kind = (int)(q / C);
    const int64_t c = q - (int64_t)kind * C;
    const int64_t b = row / (L - 1);
    const int64_t i = row - b * (L - 1);
    const T* xi = x + (b * L + i) * C + c;
    const T h = t[i + 1] - t[i];
    const T h_prev = i > 0 ? t[i] - t[i - 1] : h;
    Vec<T, V> lo = *reinterpret_cast<const Vec<T, V>*>(xi);
    Vec<T, V> res;
    if (kind == 0) {
      res = lo;
    } else {
      Vec<T, V> hi = *reinterpret_cast<const Vec<T, V>*>(xi + C);
      Vec<T, V> before = lo;
      if (i > 0) before = *reinterpret_cast<const Vec<T, V>*>(xi - C);
#pragma unroll
      for (int k = 0; k < V; ++k) {
        const T rise = hi.v[k] - lo.v[k];
        const T secant = rise / h;
        const T enter = i > 0 ? (lo.v[k] - before.v[k]) / h_prev : secant;
        if (kind == 1) {
          res.v[k] = enter;
        } else {
          const T two_c = (T)2 * ((T)3 * (rise / h - enter) - secant + enter) / h;
          if (kind == 2) {
            res.v[k] = two_c;
          } else {
            res.v[k] = ((T)1 / (h * h)) * (secant - enter) - two_c / h;
          }
        }
      }
    }
    *reinterpret_cast<Vec<T, V>*>(out + row * 4 * C + q) = res;
  }
}

template <typename T>
static int launch_hermite(const void* x, const void* t, void* out, int64_t B, int64_t L, int64_t C, hipStream_t s) {
  constexpr int VMAX = 16 / sizeof(T);
  const int64_t total_scalar = B * (L - 1) * 4 * C;
  if (total_scalar == 0) return CDE_OK;
  const bool aligned = ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0);
  auto grid_for = [](int64_t n) { int64_t g = (n + 255) / 256; return (unsigned)(g > 262144 ? 262144 : g); };
  if (aligned && C % VMAX == 0) {
    hermite_bdiff_kernel<T, VMAX><<<grid_for(total_scalar / VMAX), 256, 0, s>>>((const T*)x, (const T*)t, (T*)out, B, L, C);
  } else {
    hermite_bdiff_kernel<T, 1><<<grid_for(total_scalar), 256, 0, s>>>((const T*)x, (const T*)t, (T*)out, B, L, C);
  }
  return check_launch();
}

// ------------------------------------------------------------------------------------------ K1b
template <typename T>
__global__ void interpret_t_kernel(const T* __restrict__ knots, int64_t n_intervals, const T* __restrict__ tq,
                                   int64_t nq, int64_t* __restrict__ index_out, T* __restrict__ frac_out) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= nq) return;
  T frac;
  const int64_t idx = locate(knots, n_intervals, tq[q], frac);
  index_out[q] = idx;
  frac_out[q] = frac;
}

// out[b, q, c]; one lane per output element, c fastest (coalesced stores, row-contiguous loads).
template <typename T, int DEGREE, int WHAT>
__global__ __launch_bounds__(256) void path_eval_kernel(const T* __restrict__ coeffs, const T* __restrict__ knots,
                                                        const T* __restrict__ tq, int64_t nq, T* __restrict__ out,
                                                        int64_t B, int64_t n_intervals, int64_t C) {
  const int64_t total = B * nq * C;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = e % C;
    const int64_t q = (e / C) % nq;
    const int64_t b = e / (C * nq);
    T frac;
    const int64_t idx = locate(knots, n_intervals, tq[q], frac);
    T r;
    if (DEGREE == CDE_PATH_CUBIC) {
      const T* row = coeffs + (b * n_intervals + idx) * 4 * C;
      if (WHAT == CDE_EVAL_DERIVATIVE) {
        r = cubic_derivative(row[C + c], row[2 * C + c], row[3 * C + c], frac);
      } else {
        r = cubic_value(row[c], row[C + c], row[2 * C + c], row[3 * C + c], frac);
      }
    } else {
      // linear: coeffs are the knot values (B, n_intervals+1, C)
      const T* lo = coeffs + (b * (n_intervals + 1) + idx) * C;
      const T width = knots[idx + 1] - knots[idx];
      if (WHAT == CDE_EVAL_DERIVATIVE) {
        r = (lo[C + c] - lo[c]) / width;                       // interpolation_linear.py:189
      } else {
        r = lo[c] + frac * (lo[C + c] - lo[c]) / width;        // interpolation_linear.py:220
      }
    }
    out[e] = r;
  }
}

template <typename T>
static int launch_path_eval(const void* coeffs, const void* knots, const void* tq, int64_t nq, void* out, int64_t B,
                            int64_t n_intervals, int64_t C, int degree, int what, hipStream_t s) {
  const int64_t total = B * nq * C;
  if (total == 0) return CDE_OK;
  int64_t g = (total + 255) / 256;
  const unsigned grid = (unsigned)(g > 65536 ? 65536 : g);
#define CDE_PE(D, W) \
  path_eval_kernel<T, D, W><<<grid, 256, 0, s>>>((const T*)coeffs, (const T*)knots, (const T*)tq, nq, (T*)out, B, n_intervals, C)
  if (degree == CDE_PATH_CUBIC && what == CDE_EVAL_DERIVATIVE) CDE_PE(CDE_PATH_CUBIC, CDE_EVAL_DERIVATIVE);
  else if (degree == CDE_PATH_CUBIC && what == CDE_EVAL_VALUE) CDE_PE(CDE_PATH_CUBIC, CDE_EVAL_VALUE);
  else if (degree == CDE_PATH_LINEAR && what == CDE_EVAL_DERIVATIVE) CDE_PE(CDE_PATH_LINEAR, CDE_EVAL_DERIVATIVE);
  else if (degree == CDE_PATH_LINEAR && what == CDE_EVAL_VALUE) CDE_PE(CDE_PATH_LINEAR, CDE_EVAL_VALUE);
  else return CDE_ERR_UNSUPPORTED;
#undef CDE_PE
  return check_launch();
}

// ------------------------------------------------------------------------------------------ contraction
// out[b,h] = sum_c F[b,h,c]*dX[b,c]   (solver.py:130); one lane per (b,h), F rows are C contiguous floats.
template <typename T>
__global__ __launch_bounds__(256) void contract_kernel(const T* __restrict__ F, const T* __restrict__ dX,
                                                       T* __restrict__ out, int64_t B, int64_t H, int64_t C) {
  const int64_t total = B * H;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = e / H;
    const T* f = F + e * C;
    const T* d = dX + b * C;
    T acc = (T)0;
    for (int64_t c = 0; c < C; ++c) acc = fma_t(f[c], d[c], acc);
    out[e] = acc;
  }
}

}  // namespace cde

// ================================================================================================ C ABI
extern "C" int cde_hermite_bdiff_coeffs(const void* x, const void* t, void* coeffs, int64_t B, int64_t L, int64_t C,
                                        int dtype, void* stream) {
  if (B < 0 || L < 2 || C < 1) return CDE_ERR_SHAPE;
  if (B == 0) return CDE_OK;
  if (!x || !t || !coeffs) return CDE_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CDE_F32) return cde::launch_hermite<float>(x, t, coeffs, B, L, C, s);
  if (dtype == CDE_F64) return cde::launch_hermite<double>(x, t, coeffs, B, L, C, s);
  return CDE_ERR_DTYPE;
}

extern "C" int cde_interpret_t(const void* knots, int64_t n_intervals, const void* tq, int64_t nq, int64_t* index_out,
                               void* frac_out, int dtype, void* stream) {
  if (n_intervals < 1 || nq < 0) return CDE_ERR_SHAPE;
  if (nq == 0) return CDE_OK;
  if (!knots || !tq || !index_out || !frac_out) return CDE_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)((nq + 255) / 256);
  if (dtype == CDE_F32)
    cde::interpret_t_kernel<float><<<grid, 256, 0, s>>>((const float*)knots, n_intervals, (const float*)tq, nq, index_out, (float*)frac_out);
  else if (dtype == CDE_F64)
    cde::interpret_t_kernel<double><<<grid, 256, 0, s>>>((const double*)knots, n_intervals, (const double*)tq, nq, index_out, (double*)frac_out);
  else
    return CDE_ERR_DTYPE;
  return cde::check_launch();
}

extern "C" int cde_path_eval(const void* coeffs, const void* knots, const void* tq, int64_t nq, void* out, int64_t B,
                             int64_t n_intervals, int64_t C, int degree, int what, int dtype, void* stream) {
  if (B < 0 || n_intervals < 1 || C < 1 || nq < 0) return CDE_ERR_SHAPE;
  if (B == 0 || nq == 0) return CDE_OK;
  if (!coeffs || !knots || !tq || !out) return CDE_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CDE_F32) return cde::launch_path_eval<float>(coeffs, knots, tq, nq, out, B, n_intervals, C, degree, what, s);
  if (dtype == CDE_F64) return cde::launch_path_eval<double>(coeffs, knots, tq, nq, out, B, n_intervals, C, degree, what, s);
  return CDE_ERR_DTYPE;
}

extern "C" int cde_contract(const void* F, const void* dX, void* out, int64_t B, int64_t H, int64_t C, int dtype,
                            void* stream) {
  if (B < 0 || H < 1 || C < 1) return CDE_ERR_SHAPE;
  if (B == 0) return CDE_OK;
  if (!F || !dX || !out) return CDE_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  int64_t g = (B * H + 255) / 256;
  const unsigned grid = (unsigned)(g > 65536 ? 65536 : g);
  if (dtype == CDE_F32)
    cde::contract_kernel<float><<<grid, 256, 0, s>>>((const float*)F, (const float*)dX, (float*)out, B, H, C);
  else if (dtype == CDE_F64)
    cde::contract_kernel<double><<<grid, 256, 0, s>>>((const double*)F, (const double*)dX, (double*)out, B, H, C);
  else
    return CDE_ERR_DTYPE;
  return cde::check_launch();
}

extern "C" int cde_abi_version(void) { return CDE_ABI_VERSION; }

extern "C" const char* cde_error_string(int code) {
  switch (code) {
    case CDE_OK: return "ok";
    case CDE_ERR_NULL: return "a required pointer argument is NULL";
    case CDE_ERR_DTYPE: return "unknown dtype enum (expected CDE_F32 or CDE_F64)";
    case CDE_ERR_SHAPE: return "a size argument is out of range";
    case CDE_ERR_UNSUPPORTED: return "this (dtype, shape, activation, variant) combination is not implemented";
    case CDE_ERR_WORKSPACE: return "workspace smaller than cde_rk4_adjoint_workspace_bytes()";
    case CDE_ERR_LAUNCH: return "HIP kernel launch failed";
    default: return "unknown error code";
  }
}
