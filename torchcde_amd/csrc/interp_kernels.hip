// interp_kernels.hip -- K1 (Hermite fit), K1b (interval lookup + path evaluation), contraction.
// HBM-bound streaming kernels: one lane per 16-byte output vector, fully coalesced stores.
#include "cde_common.h"

namespace cde {

template <typename T, int V>
struct Vec {
  T v[V];
};

// ------------------------------------------------------------------------------------------ K1
// One lane produces, for one interval of one series, all four coefficient kinds of V consecutive channels
// (V = 16 B worth): the IEEE divisions -- the expensive part, they must stay true divisions for bit parity --
// are shared between the kinds (4 per channel instead of 10), and the knot derivative entering the interval is
// taken from the lane that owns the previous interval of the same series when that lane is in the same wave.
//
// Arithmetic follows interpolation_hermite_cubic_bdiff.py:39 and :10-18 literally:
//   secant_i = (x[i+1]-x[i]) / (t[i+1]-t[i])
//   enter_i  = secant_{i-1}   (enter_0 = secant_0)
//   two_c    = 2*(3*((x[i+1]-x[i])/h - enter) - secant + enter) / h
//   three_d  = (1/h**2)*(secant - enter) - two_c/h
// Unit knot spacing (the reference's default t = linspace(0, L-1, L): every h is exactly 1.0): q / 1.0 == q and
// 1 / (1*1) == 1 in IEEE arithmetic, so the divisions are skipped -- same bits, and the kernel is then a pure stream.
// Stores: the 64 lanes of a wave own 64 consecutive (interval, channel group) entries = one contiguous span of
// 256*V output values, but lane by lane the four kinds of a row sit 4C values apart.  The tile goes through a
// per-wave LDS buffer and leaves as four fully coalesced stores (64 lanes x 16 B contiguous each).
// CHECK: raise *nan_flag to `mark` (atomic max) when some input value is NaN (the reference scans x for NaNs before
// fitting, interpolation_linear.py:169; here the scan rides on the loads the fit does anyway).  mark = 1 on a zeroed
// flag is the plain "OR 1"; a call counter as `mark` needs no zero-fill between calls.
// gate: when non-null the launch does nothing unless *gate == gate_value (the on-device "repair" pass of
// cde_hermite_bdiff_coeffs_nonblocking: it runs only if the checked pass of the same call raised the flag).
template <typename T, int V, bool CHECK>
__global__ __launch_bounds__(256) void hermite_bdiff_kernel(const T* __restrict__ x, const T* __restrict__ t,
                                                            T* __restrict__ out, int64_t B, int64_t L, int64_t C,
                                                            int* __restrict__ nan_flag, int mark = 1,
                                                            const int* __restrict__ gate = nullptr, int gate_value = 0) {
  __shared__ __attribute__((aligned(16))) T stage[V > 1 ? 4 * 256 * V : 1];
  if (gate && *gate != gate_value) return;
  const int64_t groups = C / V;                       // lanes per interval row
  const int64_t total = B * (L - 1) * groups;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = e < total;
  const int64_t ec = live ? e : total - 1;
  const int64_t row = ec / groups;
  const int64_t c = (ec - row * groups) * V;
  const int64_t b = row / (L - 1);
  const int64_t i = row - b * (L - 1);
  const T* xi = x + (b * L + i) * C + c;
  const T h = t[i + 1] - t[i];
  const Vec<T, V> lo = *reinterpret_cast<const Vec<T, V>*>(xi);
  const Vec<T, V> hi = *reinterpret_cast<const Vec<T, V>*>(xi + C);
  if (CHECK) {
    bool bad = false;
#pragma unroll
    for (int k = 0; k < V; ++k) bad = bad || (lo.v[k] != lo.v[k]) || (hi.v[k] != hi.v[k]);
    if (__builtin_amdgcn_ballot_w64(bad) != 0 && (threadIdx.x & 63) == 0) atomicMax(nan_flag, mark);
  }
  const bool unit = h == (T)1;
  Vec<T, V> rise, secant;
#pragma unroll
  for (int k = 0; k < V; ++k) { rise.v[k] = hi.v[k] - lo.v[k]; secant.v[k] = unit ? rise.v[k] : rise.v[k] / h; }
  // previous interval of the same series lives `groups` lanes below (same wave) unless i == 0
  const int lane = threadIdx.x & 63;
  const bool from_neighbour = groups <= 32 && lane >= (int)groups && i > 0;
  Vec<T, V> enter;
#pragma unroll
  for (int k = 0; k < V; ++k) enter.v[k] = __shfl_up(secant.v[k], (unsigned)(groups <= 32 ? groups : 1), 64);
  if (!from_neighbour) {
    if (i > 0) {
      const Vec<T, V> before = *reinterpret_cast<const Vec<T, V>*>(xi - C);
      const T h_prev = t[i] - t[i - 1];
#pragma unroll
      for (int k = 0; k < V; ++k) enter.v[k] = h_prev == (T)1 ? lo.v[k] - before.v[k] : (lo.v[k] - before.v[k]) / h_prev;
    } else {
      enter = secant;
    }
  }
  Vec<T, V> two_c, three_d;
  if (unit) {
#pragma unroll
    for (int k = 0; k < V; ++k) {
      two_c.v[k] = (T)2 * ((T)3 * (secant.v[k] - enter.v[k]) - secant.v[k] + enter.v[k]);
      three_d.v[k] = (secant.v[k] - enter.v[k]) - two_c.v[k];
    }
  } else {
    const T inv_h2 = (T)1 / (h * h);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      two_c.v[k] = (T)2 * ((T)3 * (secant.v[k] - enter.v[k]) - secant.v[k] + enter.v[k]) / h;   // rise/h == secant
      three_d.v[k] = inv_h2 * (secant.v[k] - enter.v[k]) - two_c.v[k] / h;
    }
  }
  if (V == 1) {                                       // narrow / unaligned shapes: direct stores
    if (!live) return;
    T* o = out + row * 4 * C + c;
    o[0] = lo.v[0]; o[C] = enter.v[0]; o[2 * C] = two_c.v[0]; o[3 * C] = three_d.v[0];
    return;
  }
  // stage the wave's tile: value index inside the tile = (row - first row of the wave) * 4C + kind * C + c
  const int wave = threadIdx.x >> 6;
  T* tile = stage + wave * 256 * V;
  const int64_t e0 = e - lane;                        // first entry of this wave
  const int64_t row0 = e0 / groups;                   // its first row
  const int64_t rel = (row - row0) * 4 * C + c;       // this lane's kind-0 piece inside the tile
  if ((64 % groups) == 0) {
    // whole rows per wave (C*sizeof(T) divides 1 KB): the tile is exactly 256 V contiguous values of `out`
    if (live) {
      *reinterpret_cast<Vec<T, V>*>(tile + rel) = lo;
      *reinterpret_cast<Vec<T, V>*>(tile + rel + C) = enter;
      *reinterpret_cast<Vec<T, V>*>(tile + rel + 2 * C) = two_c;
      *reinterpret_cast<Vec<T, V>*>(tile + rel + 3 * C) = three_d;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    T* o = out + row0 * 4 * C;
    const int64_t limit = (B * (L - 1) - row0) * 4 * C;                  // values of `out` from row0 to the end
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t at = (int64_t)(j * 64 + lane) * V;
      if (at < limit) {                                                  // written once, never re-read here: streaming store
        using Raw = __attribute__((ext_vector_type(4))) unsigned;
        static_assert(sizeof(Vec<T, V>) == 16 || V == 1, "16-byte tiles");
        __builtin_nontemporal_store(*reinterpret_cast<const Raw*>(tile + at), reinterpret_cast<Raw*>(o + at));
      }
    }
    return;
  }
  if (!live) return;
  T* o = out + row * 4 * C + c;
  *reinterpret_cast<Vec<T, V>*>(o) = lo;
  *reinterpret_cast<Vec<T, V>*>(o + C) = enter;
  *reinterpret_cast<Vec<T, V>*>(o + 2 * C) = two_c;
  *reinterpret_cast<Vec<T, V>*>(o + 3 * C) = three_d;
}

template <typename T>
static int launch_hermite(const void* x, const void* t, void* out, int64_t B, int64_t L, int64_t C, int* nan_flag,
                          hipStream_t s, int mark = 1, const int* gate = nullptr, int gate_value = 0) {
  constexpr int VMAX = 16 / sizeof(T);
  if (B * (L - 1) * C == 0) return CDE_OK;
  const bool aligned = ((uintptr_t)x % 16 == 0) && ((uintptr_t)out % 16 == 0);
  auto grid_for = [](int64_t n) { return (unsigned)((n + 255) / 256); };
#define CDE_K1(V, CHECK, N)                                                                                         \
  hermite_bdiff_kernel<T, V, CHECK><<<grid_for(N), 256, 0, s>>>((const T*)x, (const T*)t, (T*)out, B, L, C, nan_flag, mark, \
                                                               gate, gate_value)
  if (aligned && C % VMAX == 0) {
    if (nan_flag) CDE_K1(VMAX, true, B * (L - 1) * (C / VMAX)); else CDE_K1(VMAX, false, B * (L - 1) * (C / VMAX));
  } else {
    if (nan_flag) CDE_K1(1, true, B * (L - 1) * C); else CDE_K1(1, false, B * (L - 1) * C);
  }
#undef CDE_K1
  return check_launch();
}

// K1 backward: the fit is linear in x, so dL/dx is the transposed map applied to dL/dcoeffs.  With
//   u_j = secant_j - enter_j:   two_c_j = 4 u_j / h_j,   three_d_j = -3 u_j / h_j^2   (what :17-18 reduce to)
//   dL/du_j = 4 gc_j / h_j - 3 gd_j / h_j^2,   dL/denter_j = gb_j - dL/du_j,
//   dL/dsecant_j = dL/du_j + dL/denter_{j+1} (+ dL/denter_0 for j = 0)          [enter_j = secant_{j-1}, enter_0 = secant_0]
//   dL/dx_i = ga_i + dL/dsecant_{i-1} / h_{i-1} - dL/dsecant_i / h_i
// One lane per element of dL/dx gathers the (at most) three coefficient rows it depends on: no atomics.
template <typename T>
__global__ __launch_bounds__(256) void hermite_bdiff_backward_kernel(const T* __restrict__ g, const T* __restrict__ t,
                                                                     T* __restrict__ gx, int64_t B, int64_t L, int64_t C) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * L * C) return;
  const int64_t c = e % C, bi = e / C, i = bi % L, b = bi / L;
  const T* gs = g + b * (L - 1) * 4 * C + c;            // + j*4C + {0, C, 2C, 3C}
  auto d_enter = [&](int64_t j, T& du) {                 // returns dL/denter_j, sets dL/du_j
    const T h = t[j + 1] - t[j];
    const T* r = gs + j * 4 * C;
    du = (T)4 * r[2 * C] / h - (T)3 * r[3 * C] / (h * h);
    return r[C] - du;
  };
  auto d_secant_over_h = [&](int64_t j) {                // dL/dsecant_j / h_j
    T du, dun;
    const T de = d_enter(j, du);
    T ds = du;
    if (j + 1 <= L - 2) ds += d_enter(j + 1, dun);
    if (j == 0) ds += de;
    return ds / (t[j + 1] - t[j]);
  };
  T acc = (T)0;
  if (i <= L - 2) acc = gs[i * 4 * C] - d_secant_over_h(i);
  if (i >= 1) acc += d_secant_over_h(i - 1);
  gx[e] = acc;
}

// dL/dh_j of the same fit, h_j = t_{j+1} - t_j, one lane per (series, interval, channel); the caller sums over series
// and channels and distributes dL/dt_{j+1} += dL/dh_j, dL/dt_j -= dL/dh_j.  With s_j = rise_j / h_j, e_j = s_{j-1}
// (e_0 = s_0), u_j = s_j - e_j the fit is b_j = e_j, 2c_j = 4 u_j / h_j, 3d_j = -3 u_j / h_j^2 (the reference's
// expressions :17-18 simplify to these), so
//   dL/dh_j = dL/ds_j (-s_j / h_j) + g2c_j (-4 u_j / h_j^2) + g3d_j (6 u_j / h_j^3).
template <typename T>
__global__ __launch_bounds__(256) void hermite_bdiff_backward_dt_kernel(const T* __restrict__ g, const T* __restrict__ x,
                                                                        const T* __restrict__ t, T* __restrict__ gh,
                                                                        int64_t B, int64_t L, int64_t C) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * (L - 1) * C) return;
  const int64_t c = e % C, bj = e / C, j = bj % (L - 1), b = bj / (L - 1);
  const T* gs = g + b * (L - 1) * 4 * C + c;
  const T* xs = x + b * L * C + c;
  auto secant = [&](int64_t k) { return (xs[(k + 1) * C] - xs[k * C]) / (t[k + 1] - t[k]); };
  auto d_u = [&](int64_t k) { const T h = t[k + 1] - t[k]; return (T)4 * gs[k * 4 * C + 2 * C] / h - (T)3 * gs[k * 4 * C + 3 * C] / (h * h); };
  const T h = t[j + 1] - t[j];
  const T sj = secant(j), ej = j == 0 ? sj : secant(j - 1), uj = sj - ej;
  T gsj = d_u(j);                                            // dL/ds_j: through u_j, e_{j+1} (and e_0 for j == 0)
  if (j + 1 <= L - 2) gsj += gs[(j + 1) * 4 * C + C] - d_u(j + 1);
  if (j == 0) gsj += gs[C] - d_u(0);
  gh[e] = gsj * (-sj / h) + gs[j * 4 * C + 2 * C] * ((T)-4 * uj / (h * h)) + gs[j * 4 * C + 3 * C] * ((T)6 * uj / (h * h * h));
}

// ------------------------------------------------------------------------------------------ K0 missing values
// linear_interpolation_coeffs on data with NaNs (interpolation_linear.py:13-84): every scalar path (one series,
// one channel) is filled independently -- observed values stay, a gap is the straight line between its nearest
// OBSERVED neighbours  prev + ((t - t_prev)/(t_next - t_prev)) * (next - prev)  (same association as :68-69),
// leading / trailing gaps take the first / last observation, an all-NaN path becomes zeros.  One lane per scalar
// path walks the L samples once; lanes of a wave cover adjacent channels / series, so the strided reads of a wave
// land in the same cache lines.  (The reference does this with Python loops per scalar path: ~47 series/s.)
template <typename T>
__global__ __launch_bounds__(256) void linear_fill_kernel(const T* __restrict__ x, const T* __restrict__ t,
                                                          T* __restrict__ out, int64_t B, int64_t L, int64_t C,
                                                          const int* __restrict__ gate = nullptr, int gate_value = 0) {
  if (gate && *gate != gate_value) return;             // (see hermite_bdiff_kernel)
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * C) return;
  const int64_t b = e / C, c = e - b * C;
  const T* src = x + b * L * C + c;
  T* dst = out + b * L * C + c;
  int64_t first = -1, last = -1;
  for (int64_t i = 0; i < L; ++i) {
    const T v = src[i * C];
    if (v == v) { if (first < 0) first = i; last = i; }
  }
  if (first < 0) {
    for (int64_t i = 0; i < L; ++i) dst[i * C] = (T)0;
    return;
  }
  const T head = src[0], tail = src[(L - 1) * C];
  const T v0 = head == head ? head : src[first * C];
  const T vL = tail == tail ? tail : src[last * C];
  dst[0] = v0;
  int64_t lo = 0;
  T x_lo = v0;
  for (int64_t i = 1; i < L; ++i) {
    const T xi = i == L - 1 ? vL : src[i * C];
    if (xi == xi) {
      const T t_lo = t[lo], span = t[i] - t_lo;
      for (int64_t j = lo + 1; j < i; ++j) {
        const T ratio = (t[j] - t_lo) / span;
        dst[j * C] = x_lo + ratio * (xi - x_lo);
      }
      dst[i * C] = xi;
      lo = i;
      x_lo = xi;
    }
  }
}

// Backward of linear_fill_kernel w.r.t. the observed values: every filled entry is x_lo + r (x_hi - x_lo) of its two
// anchors (or a copy of the first / last observation), so its incoming gradient goes to them with weights (1 - r, r).
// One lane per scalar path walks the gaps exactly like the forward kernel and adds in place: fixed order, no atomics.
template <typename T>
__global__ __launch_bounds__(256) void linear_fill_backward_kernel(const T* __restrict__ grad_out, const T* __restrict__ x,
                                                                   const T* __restrict__ t, T* __restrict__ grad_x,
                                                                   int64_t B, int64_t L, int64_t C) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * C) return;
  const int64_t b = e / C, c = e - b * C;
  const T* src = x + b * L * C + c;
  const T* g = grad_out + b * L * C + c;
  T* dst = grad_x + b * L * C + c;
  int64_t first = -1, last = -1;
  for (int64_t i = 0; i < L; ++i) {
    const T v = src[i * C];
    dst[i * C] = (T)0;
    if (v == v) { if (first < 0) first = i; last = i; }
  }
  if (first < 0) return;                                    // an all-NaN path is filled with constants
  const T head = src[0], tail = src[(L - 1) * C];
  const int64_t s0 = head == head ? 0 : first;              // where the values at the two ends come from
  const int64_t sL = tail == tail ? L - 1 : last;
  dst[s0 * C] += g[0];
  int64_t lo = 0, src_lo = s0;
  for (int64_t i = 1; i < L; ++i) {
    const bool anchor = i == L - 1 || src[i * C] == src[i * C];
    if (anchor) {
      const int64_t src_i = i == L - 1 ? sL : i;
      const T t_lo = t[lo], span = t[i] - t_lo;
      T to_lo = (T)0, to_i = g[i * C];
      for (int64_t j = lo + 1; j < i; ++j) {
        const T ratio = (t[j] - t_lo) / span;
        const T gj = g[j * C];
        to_lo += gj - gj * ratio;
        to_i += gj * ratio;
      }
      dst[src_lo * C] += to_lo;
      dst[src_i * C] += to_i;
      lo = i;
      src_lo = src_i;
    }
  }
}

// ------------------------------------------------------------------------------------------ K0b / K0c
// K0b forward fill along the length axis (reference misc.py:103-126: gather at the cummax of the observed-count
// cumsum): every NaN takes the latest earlier observation of its scalar path; leading NaNs stay NaN.
// K0c rectilinear preparation (interpolation_linear.py:86-128): forward fill, every sample repeated twice with the
// time channel advanced by one row, last row dropped -> 2L-1 rows, so that LINEAR interpolation of the result is the
// rectilinear (causal) interpolation of the data:   out[j][c] = filled[j/2][c],  out[j][time] = time[(j+1)/2].
// Pure data movement (bit-exact); one lane per scalar path walks its L samples once, as in K0.
template <typename T>
__global__ __launch_bounds__(256) void forward_fill_kernel(const T* __restrict__ x, T* __restrict__ out, int64_t B,
                                                           int64_t L, int64_t C) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * C) return;
  const int64_t b = e / C, c = e - b * C;
  const T* src = x + b * L * C + c;
  T* dst = out + b * L * C + c;
  T held = src[0];
  for (int64_t i = 0; i < L; ++i) {
    const T v = src[i * C];
    if (v == v) held = v;
    dst[i * C] = held;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void rectilinear_prepare_kernel(const T* __restrict__ x, T* __restrict__ out,
                                                                  int64_t B, int64_t L, int64_t C, int64_t time_index) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * C) return;
  const int64_t b = e / C, c = e - b * C;
  const T* src = x + b * L * C + c;
  T* dst = out + b * (2 * L - 1) * C + c;
  if (c == time_index) {
    // the host has checked that the time column holds no NaN (the reference asserts it)
    for (int64_t i = 0; i < L; ++i) {
      const T v = src[i * C];
      dst[(2 * i) * C] = v;
      if (i > 0) dst[(2 * i - 1) * C] = v;
    }
    return;
  }
  T held = src[0];
  for (int64_t i = 0; i < L; ++i) {
    const T v = src[i * C];
    if (v == v) held = v;
    dst[(2 * i) * C] = held;
    if (i + 1 < L) dst[(2 * i + 1) * C] = held;
  }
}

// Backward of the two kernels above (the reference's gathers are differentiable): every output entry was copied from
// one input entry, so the gradients of all the copies of an observation flow back to it.  One lane per scalar path,
// walking backwards; RECT: grad_out has 2L-1 rows, rows 2i and 2i+1 are copies of filled row i (time channel: rows
// 2i-1 and 2i are copies of row i).  Missing entries get 0.
template <typename T, bool RECT>
__global__ __launch_bounds__(256) void forward_fill_backward_kernel(const T* __restrict__ grad_out,
                                                                    const T* __restrict__ x, T* __restrict__ grad_x,
                                                                    int64_t B, int64_t L, int64_t C, int64_t time_index) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * C) return;
  const int64_t b = e / C, c = e - b * C;
  const T* src = x + b * L * C + c;
  const T* g = grad_out + b * (RECT ? 2 * L - 1 : L) * C + c;
  T* dst = grad_x + b * L * C + c;
  if (RECT && c == time_index) {
    for (int64_t i = 0; i < L; ++i) dst[i * C] = i > 0 ? g[(2 * i - 1) * C] + g[(2 * i) * C] : g[0];
    return;
  }
  int64_t first = L;                                        // leading NaNs are copies of themselves (the reference's
  for (int64_t i = 0; i < L; ++i) { const T v = src[i * C]; if (v == v) { first = i; break; } }   // gather index)
  T run = (T)0;
  for (int64_t i = L - 1; i >= 0; --i) {
    T own;
    if (RECT) own = i + 1 < L ? g[(2 * i + 1) * C] + g[(2 * i) * C] : g[(2 * i) * C];
    else own = g[i * C];
    run = run + own;
    const T v = src[i * C];
    if (v == v) { dst[i * C] = run; run = (T)0; }
    else dst[i * C] = i < first ? own : (T)0;
  }
}

// ------------------------------------------------------------------------------------------ K1n natural cubic splines
// natural_cubic_coeffs / natural_cubic_spline_coeffs (interpolation_cubic.py:7-266 with the tridiagonal solve of
// misc.py:14-67), missing values included.  One lane per scalar path (series, channel), three sequential passes over
// its L samples, every floating-point operation in the reference's order (bit-exact):
//   1. forward sweep of the Thomas algorithm over the KEPT points (observed, or imputed at the ends: version 0 copies
//      the first/last observation to the two end points, version 1 fills everything before/after them); the system
//      rows need the next kept point, so a row is finished one kept point late.  Temporaries live in the path's own
//      output rows at the compact index k:  a-slot = value_k, b-slot = new_b[k], 2c-slot = new_diag[k], 3d-slot = time_k
//   2. backward substitution, emitting the coefficients of compact piece k in place as soon as both knot derivatives
//      are known
//   3. (paths with gaps) expansion from compact pieces to all L-1 intervals, back to front so that nothing unread is
//      overwritten: interval j re-centres the piece that contains it (:149-160).
template <typename T>
__global__ __launch_bounds__(256) void natural_cubic_kernel(const T* __restrict__ x, const T* __restrict__ t,
                                                            T* __restrict__ out, int64_t B, int64_t L, int64_t C,
                                                            int version, int expand_all) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * C) return;
  const int64_t b = e / C, c = e - b * C;
  const T* src = x + b * L * C + c;
  T* row0 = out + b * (L - 1) * 4 * C + c;
  const int64_t RS = 4 * C;                                // row stride; slots a, b, 2c, 3d at +0, +C, +2C, +3C
  int64_t first = -1, last = -1;
  for (int64_t i = 0; i < L; ++i) { const T v = src[i * C]; if (v == v) { if (first < 0) first = i; last = i; } }
  if (first < 0) {
    for (int64_t j = 0; j < L - 1; ++j) { T* r = row0 + j * RS; r[0] = (T)0; r[C] = (T)0; r[2 * C] = (T)0; r[3 * C] = (T)0; }
    return;
  }
  const T v_first = src[first * C], v_last = src[last * C];
  auto value = [&](int64_t i, bool& kept) -> T {
    const T v = src[i * C];
    kept = true;
    if (v == v) return v;
    if (version == 0) { if (i == 0) return v_first; if (i == L - 1) return v_last; }
    else { if (i < first) return v_first; if (i > last) return v_last; }
    kept = false;
    return v;
  };
  // ---- pass 1
  int64_t m = 0;
  T tau_pp = 0, v_pp = 0;          // kept point k-1 (the row being finished)
  T tau_p = 0, v_p = 0;            // kept point k   (the newest)
  T r_prev = 0, s_prev = 0;        // r_{k-2}, scaled_{k-2}: piece before the row being finished
  T nd_prev = 0, nb_prev = 0;      // new_diag / new_b of the row before the one being finished
  for (int64_t i = 0; i < L; ++i) {
    bool kept;
    const T v = value(i, kept);
    if (!kept) continue;
    const T tau = t[i];
    if (m >= 1) {
      // piece (m-1): between kept points m-1 (tau_p, v_p) and m (tau, v); finishes row m-1
      const T r = (T)1 / (tau - tau_p);
      const T scaled = ((T)3 * (v - v_p)) * (r * r);
      const int64_t k = m - 1;
      T diag, rhs;
      if (k == 0) { diag = r * (T)2; rhs = scaled; }
      else { diag = (r + r_prev) * (T)2; rhs = scaled + s_prev; }
      T nd, nb;
      if (k == 0) { nd = diag; nb = rhs; }
      else { const T w = r_prev / nd_prev; nd = diag - w * r_prev; nb = rhs - w * nb_prev; }
      T* slot = row0 + k * RS;
      slot[0] = v_p; slot[C] = nb; slot[2 * C] = nd; slot[3 * C] = tau_p;
      nd_prev = nd; nb_prev = nb; r_prev = r; s_prev = scaled;
    }
    tau_pp = tau_p; v_pp = v_p; tau_p = tau; v_p = v;
    ++m;
  }
  (void)tau_pp; (void)v_pp;
  // last row (k = m-1): diag = (0 + r_{m-2}) * 2, rhs = 0 + scaled_{m-2}
  if (m == 2) {
    T* slot = row0;
    const T a0 = slot[0], t0 = slot[3 * C];
    slot[C] = (v_p - a0) / (tau_p - t0); slot[2 * C] = (T)0; slot[3 * C] = (T)0;
  } else {
    const T w = r_prev / nd_prev;
    const T nd_last = ((T)0 + r_prev) * (T)2 - w * r_prev;
    const T nb_last = ((T)0 + s_prev) - w * nb_prev;
    // ---- pass 2
    T kd_next = nb_last / nd_last, tau_next = tau_p, v_next = v_p;
    for (int64_t k = m - 2; k >= 0; --k) {
      T* slot = row0 + k * RS;
      const T vk = slot[0], nb = slot[C], nd = slot[2 * C], tk = slot[3 * C];
      const T r = (T)1 / (tau_next - tk);
      const T kd = (nb - r * kd_next) / nd;
      const T six = (T)2 * ((T)3 * (v_next - vk));
      const T r2 = r * r;
      slot[C] = kd;
      slot[2 * C] = (six * r - (T)4 * kd - (T)2 * kd_next) * r;
      slot[3 * C] = (-six * r + (T)3 * (kd + kd_next)) * r2;
      kd_next = kd; tau_next = tk; v_next = vk;
    }
  }
  // ---- pass 3: only when the batch had gaps (the reference then re-centres EVERY interval, :149-160)
  if (!expand_all) return;
  int64_t k = m - 2, kidx = L - 1;
  { bool kept = false; kidx = L - 2; while (true) { (void)value(kidx, kept); if (kept) break; --kidx; } }   // kept point <= L-2
  T pa = row0[k * RS], pb = row0[k * RS + C], pc = row0[k * RS + 2 * C], pd = row0[k * RS + 3 * C];
  for (int64_t j = L - 2; j >= 0; --j) {
    if (kidx > j) {
      bool kept = false;
      kidx = j;
      while (true) { (void)value(kidx, kept); if (kept) break; --kidx; }
      --k;
      pa = row0[k * RS]; pb = row0[k * RS + C]; pc = row0[k * RS + 2 * C]; pd = row0[k * RS + 3 * C];
    }
    const T offset = t[kidx] - t[j];
    const T inner = ((T)0.5 * pc - pd * offset / (T)3) * offset;
    T* slot = row0 + j * RS;
    slot[0] = pa + (inner - pb) * offset;
    slot[C] = pb + (pd * offset - pc) * offset;
    slot[2 * C] = pc - (T)2 * pd * offset;
    slot[3 * C] = pd;
  }
}

// Backward of natural_cubic_kernel w.r.t. the values, paths WITHOUT missing entries (what autograd produces through
// interpolation_cubic.py:7-54 and the tridiagonal solve of misc.py:14-67; reference test/test_tricks.py:21-49
// differentiates through the coefficient construction).  The coefficients are linear in the values for fixed knots:
//   kd = T^-1 rhs(v),  b_k = kd_k,  2c_k = 6 dv_k r_k^2 - (4 kd_k + 2 kd_{k+1}) r_k,  3d_k = -6 dv_k r_k^3 + 3 (kd_k + kd_{k+1}) r_k^2
// so the gradient is the transposed map: collect dL/dkd from the three blocks, solve with the same (symmetric)
// tridiagonal matrix, push the solution through rhs_k = s_{k-1} + s_k, s_k = 3 dv_k r_k^2, dv_k = v_{k+1} - v_k.
// The elimination factors depend on the knots only: natural_cubic_aux_kernel computes them once per call
// (aux[k] = c'_k, aux[L + k] = 1 / pivot_k); one lane per scalar path then does a forward and a backward sweep, its
// own output row doubling as the temporary of the forward sweep.
template <typename T>
__global__ void natural_cubic_aux_kernel(const T* __restrict__ t, T* __restrict__ aux, int64_t L) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  T c_prev = 0;
  for (int64_t k = 0; k < L; ++k) {
    const T r_lo = k >= 1 ? (T)1 / (t[k] - t[k - 1]) : (T)0, r_hi = k <= L - 2 ? (T)1 / (t[k + 1] - t[k]) : (T)0;
    const T pivot = (r_lo + r_hi) * (T)2 - r_lo * c_prev;
    const T inv = (T)1 / pivot;
    c_prev = r_hi * inv;
    aux[k] = c_prev;
    aux[L + k] = inv;
  }
}

// WITH_T: also the gradient w.r.t. the knot times, as one partial row per scalar path (summed over the paths by the
// caller): the solve's solution kd is recomputed from the values into `kd_buf` (B, L, C), and with y = T^-1 dL/dkd
//   dL/dr_k = gc_k [12 dv_k r_k - (4 kd_k + 2 kd_{k+1})] + gd_k [-18 dv_k r_k^2 + 6 (kd_k + kd_{k+1}) r_k]
//           + 6 dv_k r_k (y_k + y_{k+1}) - [2 y_k kd_k + 2 y_{k+1} kd_{k+1} + y_k kd_{k+1} + y_{k+1} kd_k],
//   r_k = 1 / (t_{k+1} - t_k):  dL/dt_{k+1} -= r_k^2 dL/dr_k,  dL/dt_k += r_k^2 dL/dr_k.
template <typename T, bool WITH_T>
__global__ __launch_bounds__(256) void natural_cubic_backward_kernel(const T* __restrict__ grad, const T* __restrict__ t,
                                                                     const T* __restrict__ aux, T* __restrict__ grad_x,
                                                                     int64_t B, int64_t L, int64_t C,
                                                                     const T* __restrict__ x, T* __restrict__ kd_buf,
                                                                     T* __restrict__ grad_t_rows) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * C) return;
  const int64_t b = e / C, c = e - b * C;
  const T* g = grad + b * (L - 1) * 4 * C + c;              // piece k: a, b, 2c, 3d at g[k*4C + {0, C, 2C, 3C}]
  T* out = grad_x + b * L * C + c;
  const T* v = WITH_T ? x + b * L * C + c : nullptr;
  T* kd = WITH_T ? kd_buf + b * L * C + c : nullptr;
  T* gt = WITH_T ? grad_t_rows + b * L * C + c : nullptr;
  const int64_t RS = 4 * C;
  if (L == 2) {
    const T gb = g[C] / (t[1] - t[0]);
    out[0] = g[0] - gb;
    out[C] = gb;
    if (WITH_T) {                                           // b_0 = (v_1 - v_0) / (t_1 - t_0)
      const T w = g[C] * (v[C] - v[0]) / ((t[1] - t[0]) * (t[1] - t[0]));
      gt[0] = w; gt[C] = -w;
    }
    return;
  }
  if (WITH_T) {                                             // the forward solve again: kd = T^-1 rhs(v)
    T d_prev = 0, r_prev = 0, s_prev = 0;
    for (int64_t k = 0; k < L; ++k) {
      T r = 0, sc = 0;
      if (k <= L - 2) { r = (T)1 / (t[k + 1] - t[k]); sc = (T)3 * (v[(k + 1) * C] - v[k * C]) * r * r; }
      const T d = ((s_prev + sc) - r_prev * d_prev) * aux[L + k];
      kd[k * C] = d;
      d_prev = d; r_prev = r; s_prev = sc;
    }
    T next = kd[(L - 1) * C];
    gt[(L - 1) * C] = (T)0;
    for (int64_t k = L - 2; k >= 0; --k) { next = kd[k * C] - aux[k] * next; kd[k * C] = next; gt[k * C] = (T)0; }
  }
  // forward sweep of T y = dL/dkd; out[k] <- d'_k
  T d_prev = 0, r_prev = 0, gc_prev = 0, gd_prev = 0;
  for (int64_t k = 0; k < L; ++k) {
    T gkd = 0, r = 0, gc = 0, gd = 0;
    if (k <= L - 2) {
      r = (T)1 / (t[k + 1] - t[k]);
      gc = g[k * RS + 2 * C]; gd = g[k * RS + 3 * C];
      gkd = g[k * RS + C] - (T)4 * r * gc + (T)3 * r * r * gd;
    }
    if (k >= 1) gkd += (T)3 * r_prev * r_prev * gd_prev - (T)2 * r_prev * gc_prev;
    const T d = (gkd - r_prev * d_prev) * aux[L + k];
    out[k * C] = d;
    d_prev = d; r_prev = r; gc_prev = gc; gd_prev = gd;
  }
  // backward sweep: y_k, and with y_k, y_{k+1} the gradient of dv_k = v_{k+1} - v_k
  T y_next = out[(L - 1) * C];                              // y_{L-1} = d'_{L-1}
  T gv_next = 0;                                            // what v_{k+1} has collected so far (-(dL/d dv_{k+1}))
  for (int64_t k = L - 2; k >= 0; --k) {
    const T y = out[k * C] - aux[k] * y_next;
    const T r = (T)1 / (t[k + 1] - t[k]), r2 = r * r;
    const T gdv = (T)6 * r2 * g[k * RS + 2 * C] - (T)6 * r2 * r * g[k * RS + 3 * C] + (T)3 * r2 * (y + y_next);
    out[(k + 1) * C] = gv_next + gdv + (k + 1 <= L - 2 ? g[(k + 1) * RS] : (T)0);
    gv_next = -gdv;
    if (WITH_T) {
      const T dv = v[(k + 1) * C] - v[k * C], k0 = kd[k * C], k1 = kd[(k + 1) * C];
      const T gc = g[k * RS + 2 * C], gd = g[k * RS + 3 * C];
      const T gr = gc * ((T)12 * dv * r - ((T)4 * k0 + (T)2 * k1)) + gd * ((T)-18 * dv * r2 + (T)6 * (k0 + k1) * r) +
                   (T)6 * dv * r * (y + y_next) - ((T)2 * y * k0 + (T)2 * y_next * k1 + y * k1 + y_next * k0);
      gt[(k + 1) * C] -= r2 * gr;
      gt[k * C] += r2 * gr;
    }
    y_next = y;
  }
  out[0] = gv_next + g[0];
}

// Backward of natural_cubic_kernel w.r.t. the values for batches WITH missing entries (autograd through
// interpolation_cubic.py:83-166: index selection of the observed points, the compact solve, the re-centring of every
// interval).  One lane per scalar path, `work` (B, L-1, 4C) holds the path's temporaries at the COMPACT piece index:
//   1. transposed re-centring: the gradient of interval j folds into the gradient of the compact piece that contains
//      it (offset o = t_kept - t_j:  a_j = pa - pb o + pc o^2 / 2 - pd o^3 / 3,  b_j = pb - pc o + pd o^2,
//      c_j = pc - 2 pd o,  d_j = pd)
//   2. the compact solve transposed, as in natural_cubic_backward_kernel but with the path's OWN kept knots: forward
//      sweep over the kept points (a row is finished when the next kept point is known), the elimination factor c'_k
//      parked in the b slot of piece k, d'_k in grad_x at the point's original index; backward sweep from the last
//      kept point
//   3. imputed end points hand their gradient to the observation they copied (:109-131 version 0, :226-266 version 1);
//      missing entries get 0.
template <typename T>
__global__ __launch_bounds__(256) void natural_cubic_backward_missing_kernel(const T* __restrict__ grad,
                                                                             const T* __restrict__ x,
                                                                             const T* __restrict__ t,
                                                                             T* __restrict__ grad_x, T* __restrict__ work,
                                                                             int64_t B, int64_t L, int64_t C, int version) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * C) return;
  const int64_t b = e / C, c = e - b * C;
  const T* src = x + b * L * C + c;
  const T* g = grad + b * (L - 1) * 4 * C + c;
  T* w = work + b * (L - 1) * 4 * C + c;
  T* out = grad_x + b * L * C + c;
  const int64_t RS = 4 * C;
  int64_t first = -1, last = -1;
  for (int64_t i = 0; i < L; ++i) { const T v = src[i * C]; if (v == v) { if (first < 0) first = i; last = i; } out[i * C] = (T)0; }
  if (first < 0) return;
  auto kept = [&](int64_t i) -> bool {
    const T v = src[i * C];
    if (v == v) return true;
    return version == 0 ? (i == 0 || i == L - 1) : (i < first || i > last);
  };
  // ---- 1. fold the interval gradients into the compact pieces (positions 0 and L-1 are always kept)
  int64_t k = -1, kidx = 0;
  for (int64_t j = 0; j <= L - 2; ++j) {
    if (kept(j)) { ++k; kidx = j; T* slot = w + k * RS; slot[0] = (T)0; slot[C] = (T)0; slot[2 * C] = (T)0; slot[3 * C] = (T)0; }
    const T o = t[kidx] - t[j], o2 = o * o;
    const T ga = g[j * RS], gb = g[j * RS + C], gc = g[j * RS + 2 * C], gd = g[j * RS + 3 * C];
    T* slot = w + k * RS;
    slot[0] += ga;
    slot[C] += gb - o * ga;
    slot[2 * C] += gc - o * gb + (T)0.5 * o2 * ga;
    slot[3 * C] += gd - (T)2 * o * gc + o2 * gb - o2 * o * ga / (T)3;
  }
  const int64_t m = k + 2;                                    // kept points
  if (m == 2) {                                               // one straight piece: a = v_0, b = (v_1 - v_0) / (tau_1 - tau_0)
    const T gb = w[C] / (t[L - 1] - t[0]);
    out[0] = w[0] - gb;
    out[(L - 1) * C] = gb;
  } else {
    // ---- 2a. forward sweep of T y = dL/dkd over the kept points
    int64_t i_prev = -1;
    k = -1;
    T tau_p = 0, r_prev = 0, gc_prev = 0, gd_prev = 0, d_prev = 0, c_prev = 0;
    for (int64_t i = 0; i < L; ++i) {
      if (!kept(i)) continue;
      const T tau = t[i];
      if (k >= 0) {
        T* slot = w + k * RS;
        const T r = (T)1 / (tau - tau_p);
        const T gc = slot[2 * C], gd = slot[3 * C];
        T gkd = slot[C] - (T)4 * r * gc + (T)3 * r * r * gd;
        if (k >= 1) gkd += (T)3 * r_prev * r_prev * gd_prev - (T)2 * r_prev * gc_prev;
        const T inv = (T)1 / ((r_prev + r) * (T)2 - r_prev * c_prev);
        const T d = (gkd - r_prev * d_prev) * inv;
        c_prev = r * inv;
        slot[C] = c_prev;
        out[i_prev * C] = d;
        d_prev = d; r_prev = r; gc_prev = gc; gd_prev = gd;
      }
      ++k; i_prev = i; tau_p = tau;
    }
    // last row (kept point m-1, no piece to its right)
    T y_next;
    {
      const T gkd = (T)3 * r_prev * r_prev * gd_prev - (T)2 * r_prev * gc_prev;
      y_next = (gkd - r_prev * d_prev) / (r_prev * (T)2 - r_prev * c_prev);
    }
    // ---- 2b. backward sweep
    int64_t i_next = i_prev;
    T tau_next = tau_p, gv_next = 0;
    k = m - 2;
    for (int64_t i = i_prev - 1; i >= 0; --i) {
      if (!kept(i)) continue;
      T* slot = w + k * RS;
      const T y = out[i * C] - slot[C] * y_next;
      const T r = (T)1 / (tau_next - t[i]), r2 = r * r;
      const T gdv = (T)6 * r2 * slot[2 * C] - (T)6 * r2 * r * slot[3 * C] + (T)3 * r2 * (y + y_next);
      out[i_next * C] = gv_next + gdv + (k + 1 <= m - 2 ? w[(k + 1) * RS] : (T)0);
      gv_next = -gdv;
      y_next = y; i_next = i; tau_next = t[i];
      --k;
    }
    out[0] = gv_next + w[0];
  }
  // ---- 3. imputed end points
  for (int64_t i = 0; i < L; ++i) {
    const T v = src[i * C];
    if (v == v || !kept(i)) continue;
    const int64_t to = i < first ? first : last;
    out[to * C] += out[i * C];
    out[i * C] = (T)0;
  }
}

// ------------------------------------------------------------------------------------------ K5 log-ODE windows
// logsig_windows / logsignature_windows (reference log_ode.py:15-133) after the host has merged the window
// boundaries into the series and filled them linearly: for every window the logsignature (depth <= 3) of the
// piecewise-linear path between two boundary rows, optionally scaled, accumulated along the windows.
// The arithmetic the reference gets from the `signatory` package (absent here; see oracle/logsig.py): signature by
// Chen's identity  S <- S (x) exp(d)  over the increments, tensor-algebra logarithm, coefficients of the Lyndon words
// (`words`: (level, flat index) pairs in signatory's order, built by the host).
// The signature levels live in per-lane arrays, so the kernel is instantiated for the (channels, depth) envelopes that
// occur: up to 8 channels to depth 3 (config 5 and the examples), up to 5 channels to depth 4 (the reference's test
// runs depth 1-4 on 1-3 channels), up to 32 channels to depth 2.
template <int N, int P> struct IPow { static constexpr int value = N * IPow<N, P - 1>::value; };
template <int N> struct IPow<N, 0> { static constexpr int value = 1; };

// pass 1: one lane per (series, window) -- the windows of a series are independent until the running sum
template <typename T, int MAXC, int MAXD>
__global__ __launch_bounds__(64) void logsig_windows_kernel(const T* __restrict__ x, const int64_t* __restrict__ rows,
                                                            const T* __restrict__ scale, const int32_t* __restrict__ words,
                                                            T* __restrict__ out, int64_t B, int64_t L, int C, int depth,
                                                            int64_t n_windows, int n_words) {
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= B * n_windows) return;
  const int64_t b = id / n_windows, win = id - b * n_windows;
  const T* src = x + b * L * C;
  T* dst = out + (b * (n_windows + 1) + win + 1) * n_words;
  T S1[MAXC], S2[MAXD >= 2 ? IPow<MAXC, 2>::value : 1], S3[MAXD >= 3 ? IPow<MAXC, 3>::value : 1],
      S4[MAXD >= 4 ? IPow<MAXC, 4>::value : 1];
  const int C2 = C * C, C3 = C2 * C;
  for (int i = 0; i < C; ++i) S1[i] = (T)0;
  if (MAXD >= 2 && depth >= 2) for (int i = 0; i < C2; ++i) S2[i] = (T)0;
  if (MAXD >= 3 && depth >= 3) for (int i = 0; i < C3; ++i) S3[i] = (T)0;
  if (MAXD >= 4 && depth >= 4) for (int i = 0; i < C3 * C; ++i) S4[i] = (T)0;
  for (int64_t r = rows[win]; r < rows[win + 1]; ++r) {
    T d[MAXC];
    for (int i = 0; i < C; ++i) d[i] = src[(r + 1) * C + i] - src[r * C + i];
    // levels of S (x) exp(d), highest first (they read the old lower levels); exp(d): e1 = d, e2 = e1 (x) d / 2, ...
    // level k = S_k + e_k + S_1 (x) e_(k-1) + ... + S_(k-1) (x) e_1, added in that order (oracle/logsig.py)
    if (MAXD >= 4 && depth >= 4)
      for (int i = 0; i < C; ++i)
        for (int j = 0; j < C; ++j) {
          const T e2ij = d[i] * d[j] / (T)2;
          for (int k = 0; k < C; ++k) {
            const T e3ijk = e2ij * d[k] / (T)3;
            const T e2jk = d[j] * d[k] / (T)2;
            for (int l = 0; l < C; ++l) {
              const int at = ((i * C + j) * C + k) * C + l;
              T acc = S4[at] + e3ijk * d[l] / (T)4;
              acc = acc + S1[i] * (e2jk * d[l] / (T)3);
              acc = acc + S2[i * C + j] * (d[k] * d[l] / (T)2);
              acc = acc + S3[(i * C + j) * C + k] * d[l];
              S4[at] = acc;
            }
          }
        }
    if (MAXD >= 3 && depth >= 3)
      for (int i = 0; i < C; ++i)
        for (int j = 0; j < C; ++j) {
          const T e2 = d[i] * d[j] / (T)2;
          for (int k = 0; k < C; ++k) {
            T acc = S3[(i * C + j) * C + k] + e2 * d[k] / (T)3;
            acc = acc + S1[i] * (d[j] * d[k] / (T)2);
            acc = acc + S2[i * C + j] * d[k];
            S3[(i * C + j) * C + k] = acc;
          }
        }
    if (MAXD >= 2 && depth >= 2)
      for (int i = 0; i < C; ++i)
        for (int j = 0; j < C; ++j) S2[i * C + j] = (S2[i * C + j] + d[i] * d[j] / (T)2) + S1[i] * d[j];
    for (int i = 0; i < C; ++i) S1[i] = S1[i] + d[i];
  }
  // logarithm: log(1 + S) = S - S^2/2 + S^3/3 - S^4/4, level by level, then the Lyndon-word coordinates
  const T sc = scale[win];
  for (int w = 0; w < n_words; ++w) {
    const int level = words[2 * w], flat = words[2 * w + 1];
    T value;
    if (level == 1) value = S1[flat];
    else if (level == 2) {
      const int i = flat / C, j = flat - i * C;
      value = S2[flat] + (-(S1[i] * S1[j])) / (T)2;
    } else if (level == 3) {
      const int i = flat / C2, jk = flat - i * C2, j = jk / C, k = jk - j * C;
      const T p2 = (S1[i] * S2[j * C + k]) + S2[i * C + j] * S1[k];          // (S^2)_3
      const T p3 = (S1[i] * S1[j]) * S1[k];                                   // (S^3)_3
      value = (S3[flat] + (-p2) / (T)2) + p3 / (T)3;
    } else {
      const int i = flat / C3, jkl = flat - i * C3, j = jkl / C2, kl = jkl - j * C2, k = kl / C, l = kl - k * C;
      const int ij = i * C + j, ijk = ij * C + k;
      const T p2 = ((S1[i] * S3[jkl]) + S2[ij] * S2[kl]) + S3[ijk] * S1[l];                      // (S^2)_4
      const T s2_3 = (S1[i] * S2[j * C + k]) + S2[ij] * S1[k];                                   // (S^2)_3 at ijk
      const T p3 = ((S1[i] * S1[j]) * S2[kl]) + s2_3 * S1[l];                                    // (S^3)_4
      const T p4 = ((S1[i] * S1[j]) * S1[k]) * S1[l];                                            // (S^4)_4
      value = ((S4[flat] + (-p2) / (T)2) + p3 / (T)3) + (-p4) / (T)4;
    }
    dst[w] = value * sc;
  }
}

// pass 2: the running sum of log_ode.py:63 (sequential, like torch.cumsum on the CPU), one lane per (series, coordinate);
// row 0 = the first observation padded with zeros (log_ode.py:50-52)
template <typename T>
__global__ __launch_bounds__(256) void logsig_accumulate_kernel(const T* __restrict__ x, T* __restrict__ out, int64_t B,
                                                                int64_t L, int C, int64_t n_windows, int n_words) {
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= B * n_words) return;
  const int64_t b = id / n_words;
  const int w = (int)(id - b * n_words);
  T* col = out + b * (n_windows + 1) * n_words + w;
  T run = w < C ? x[b * L * C + w] : (T)0;
  col[0] = run;
  for (int64_t win = 1; win <= n_windows; ++win) { run = run + col[win * n_words]; col[win * n_words] = run; }
}

// ---- K5 backward (autograd through signatory's logsignature and the running sum of log_ode.py:53-63)
// pass 1: the running sum transposed -- suffix sums of grad_out along the windows, one lane per (series, coordinate);
// row k of `gsum` = sum of the rows >= k.  Row 0 is the gradient of the first observation (first C coordinates).
template <typename T>
__global__ __launch_bounds__(256) void logsig_suffix_kernel(const T* __restrict__ grad_out, T* __restrict__ gsum,
                                                            T* __restrict__ grad_x, int64_t B, int64_t L, int C,
                                                            int64_t n_windows, int n_words) {
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= B * n_words) return;
  const int64_t b = id / n_words;
  const int w = (int)(id - b * n_words);
  const T* src = grad_out + b * (n_windows + 1) * n_words + w;
  T* dst = gsum + b * (n_windows + 1) * n_words + w;
  T run = (T)0;
  for (int64_t win = n_windows; win >= 0; --win) { run = run + src[win * n_words]; dst[win * n_words] = run; }
  if (w < C) grad_x[b * L * C + w] = run;                  // grad_x was zeroed by the caller; pass 2 adds to it
}

// pass 2: one lane per (series, window).  The signature levels below the top one are rebuilt (the logarithm's and the
// Chen step's derivatives never read the top level), the word coordinates and the logarithm are differentiated into
// gS, and the Chen recursion is walked BACKWARDS: before increment r is differentiated the signature is stepped back
// with  S <- S (x) exp(-d_r)  (the reversibility signatory's own backward relies on), then
//   new_k = S_k + e_k(d) + sum_j S_j (x) e_(k-j)(d),  e_m(d) = d^(x m) / m!
// is transposed level by level, LOWEST level first (level k reads gS_k of the new signature, which the lower levels'
// updates have not touched, and adds to the lower gS).  No atomics, fixed summation order: rows interior to a window get
// both of their increments' contributions from this lane (one plain store); of a boundary row's two contributions the
// one from the window BELOW it (that window's last increment) is parked in the first C slots of the window's own `gsum`
// row -- dead once the lane has read it -- and the one from the window above it is added in place by that window's lane,
// the row's only writer in this pass; pass 3 then adds the parked values window by window.
template <typename T, int MAXC, int MAXD>
__global__ __launch_bounds__(64) void logsig_windows_backward_kernel(T* gsum, const T* __restrict__ x,
                                                                     const int64_t* __restrict__ rows,
                                                                     const T* __restrict__ scale,
                                                                     const int32_t* __restrict__ words,
                                                                     T* __restrict__ grad_x, int64_t B, int64_t L, int C,
                                                                     int depth, int64_t n_windows, int n_words) {
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= B * n_windows) return;
  const int64_t b = id / n_windows, win = id - b * n_windows;
  const T* src = x + b * L * C;
  T* gx = grad_x + b * L * C;
  T* gw = gsum + (b * (n_windows + 1) + win + 1) * n_words;
  T S1[MAXC], S2[MAXD >= 3 ? IPow<MAXC, 2>::value : 1], S3[MAXD >= 4 ? IPow<MAXC, 3>::value : 1];
  T g1[MAXC], g2[MAXD >= 2 ? IPow<MAXC, 2>::value : 1], g3[MAXD >= 3 ? IPow<MAXC, 3>::value : 1],
      g4[MAXD >= 4 ? IPow<MAXC, 4>::value : 1];
  const int C2 = C * C, C3 = C2 * C;
  for (int i = 0; i < C; ++i) { S1[i] = (T)0; g1[i] = (T)0; }
  if (MAXD >= 2 && depth >= 2) for (int i = 0; i < C2; ++i) g2[i] = (T)0;
  if (MAXD >= 3 && depth >= 3) for (int i = 0; i < C2; ++i) S2[i] = (T)0;
  if (MAXD >= 3 && depth >= 3) for (int i = 0; i < C3; ++i) g3[i] = (T)0;
  if (MAXD >= 4 && depth >= 4) for (int i = 0; i < C3; ++i) S3[i] = (T)0;
  if (MAXD >= 4 && depth >= 4) for (int i = 0; i < C3 * C; ++i) g4[i] = (T)0;
  const int64_t r_lo = rows[win], r_hi = rows[win + 1];
  // ---- the signature of the window, levels 1 .. depth-1 (same operations as the forward kernel)
  for (int64_t r = r_lo; r < r_hi; ++r) {
    T d[MAXC];
    for (int i = 0; i < C; ++i) d[i] = src[(r + 1) * C + i] - src[r * C + i];
    if (MAXD >= 4 && depth >= 4)
      for (int i = 0; i < C; ++i)
        for (int j = 0; j < C; ++j) {
          const T e2 = d[i] * d[j] / (T)2;
          for (int k = 0; k < C; ++k) {
            T acc = S3[(i * C + j) * C + k] + e2 * d[k] / (T)3;
            acc = acc + S1[i] * (d[j] * d[k] / (T)2);
            acc = acc + S2[i * C + j] * d[k];
            S3[(i * C + j) * C + k] = acc;
          }
        }
    if (MAXD >= 3 && depth >= 3)
      for (int i = 0; i < C; ++i)
        for (int j = 0; j < C; ++j) S2[i * C + j] = (S2[i * C + j] + d[i] * d[j] / (T)2) + S1[i] * d[j];
    for (int i = 0; i < C; ++i) S1[i] = S1[i] + d[i];
  }
  // ---- word coordinates and logarithm, transposed
  const T sc = scale[win];
  for (int w = 0; w < n_words; ++w) {
    const int level = words[2 * w], flat = words[2 * w + 1];
    const T g = gw[w] * sc;
    if (level == 1) g1[flat] += g;
    else if (level == 2) {
      if (MAXD >= 2) {
        const int i = flat / C, j = flat - i * C;
        g2[flat] += g;
        const T h = -g / (T)2;
        const T si = S1[i], sj = S1[j];
        g1[i] += h * sj; g1[j] += h * si;
      }
    } else if (level == 3) {
      if (MAXD >= 3) {
        const int i = flat / C2, jk = flat - i * C2, j = jk / C, k = jk - j * C, ij = i * C + j;
        g3[flat] += g;
        const T h2 = -g / (T)2, h3 = g / (T)3;
        const T si = S1[i], sj = S1[j], sk = S1[k], sjk = S2[jk], sij = S2[ij];
        g1[i] += h2 * sjk + h3 * sj * sk;
        g1[j] += h3 * si * sk;
        g1[k] += h2 * sij + h3 * si * sj;
        g2[jk] += h2 * si;
        g2[ij] += h2 * sk;
      }
    } else {
      if (MAXD >= 4) {
        const int i = flat / C3, jkl = flat - i * C3, j = jkl / C2, kl = jkl - j * C2, k = kl / C, l = kl - k * C;
        const int ij = i * C + j, jk = j * C + k, ijk = ij * C + k;
        g4[flat] += g;
        const T h2 = -g / (T)2, h3 = g / (T)3, h4 = -g / (T)4;
        const T si = S1[i], sj = S1[j], sk = S1[k], sl = S1[l];
        const T sij = S2[ij], sjk = S2[jk], skl = S2[kl], sijk = S3[ijk], sjkl = S3[jkl];
        // (S^2)_4 = S1_i S3_jkl + S2_ij S2_kl + S3_ijk S1_l
        g1[i] += h2 * sjkl; g3[jkl] += h2 * si;
        g2[ij] += h2 * skl; g2[kl] += h2 * sij;
        g3[ijk] += h2 * sl; g1[l] += h2 * sijk;
        // (S^3)_4 = S1_i S1_j S2_kl + (S1_i S2_jk + S2_ij S1_k) S1_l
        const T s23 = si * sjk + sij * sk, hs = h3 * sl;
        g1[i] += h3 * sj * skl + hs * sjk;
        g1[j] += h3 * si * skl;
        g2[kl] += h3 * si * sj;
        g1[l] += h3 * s23;
        g2[jk] += hs * si;
        g2[ij] += hs * sk;
        g1[k] += hs * sij;
        // (S^4)_4 = S1_i S1_j S1_k S1_l
        g1[i] += h4 * sj * sk * sl; g1[j] += h4 * si * sk * sl; g1[k] += h4 * si * sj * sl; g1[l] += h4 * si * sj * sk;
      }
    }
  }
  // ---- Chen's recursion backwards
  T carry[MAXC];                                            // -(dL/dd) of the increment above: what row r+1 still owes
  for (int i = 0; i < C; ++i) carry[i] = (T)0;
  for (int64_t r = r_hi - 1; r >= r_lo; --r) {
    T d[MAXC], gd[MAXC];
    for (int i = 0; i < C; ++i) d[i] = src[(r + 1) * C + i] - src[r * C + i];
    // step the signature back: S <- S (x) exp(-d), highest level first
    if (MAXD >= 4 && depth >= 4)
      for (int i = 0; i < C; ++i)
        for (int j = 0; j < C; ++j) {
          const T e2 = d[i] * d[j] / (T)2;
          for (int k = 0; k < C; ++k)
            S3[(i * C + j) * C + k] = ((S3[(i * C + j) * C + k] - e2 * d[k] / (T)3) + S1[i] * (d[j] * d[k] / (T)2)) - S2[i * C + j] * d[k];
        }
    if (MAXD >= 3 && depth >= 3)
      for (int i = 0; i < C; ++i)
        for (int j = 0; j < C; ++j) S2[i * C + j] = (S2[i * C + j] + d[i] * d[j] / (T)2) - S1[i] * d[j];
    if (depth >= 2) for (int i = 0; i < C; ++i) S1[i] = S1[i] - d[i];
    // transposed step, lowest level first
    for (int i = 0; i < C; ++i) gd[i] = g1[i];
    if (MAXD >= 2 && depth >= 2)
      for (int i = 0; i < C; ++i)
        for (int j = 0; j < C; ++j) {
          const T G = g2[i * C + j];
          gd[i] += G * d[j] / (T)2;
          gd[j] += G * (d[i] / (T)2 + S1[i]);
          g1[i] += G * d[j];
        }
    if (MAXD >= 3 && depth >= 3)
      for (int i = 0; i < C; ++i)
        for (int j = 0; j < C; ++j)
          for (int k = 0; k < C; ++k) {
            const T G = g3[(i * C + j) * C + k];
            const T s1 = S1[i], s2 = S2[i * C + j];
            gd[i] += G * d[j] * d[k] / (T)6;
            gd[j] += G * (d[i] * d[k] / (T)6 + s1 * d[k] / (T)2);
            gd[k] += G * (d[i] * d[j] / (T)6 + s1 * d[j] / (T)2 + s2);
            g1[i] += G * d[j] * d[k] / (T)2;
            g2[i * C + j] += G * d[k];
          }
    if (MAXD >= 4 && depth >= 4)
      for (int i = 0; i < C; ++i)
        for (int j = 0; j < C; ++j)
          for (int k = 0; k < C; ++k)
            for (int l = 0; l < C; ++l) {
              const T G = g4[((i * C + j) * C + k) * C + l];
              const T s1 = S1[i], s2 = S2[i * C + j], s3 = S3[(i * C + j) * C + k];
              gd[i] += G * d[j] * d[k] * d[l] / (T)24;
              gd[j] += G * (d[i] * d[k] * d[l] / (T)24 + s1 * d[k] * d[l] / (T)6);
              gd[k] += G * (d[i] * d[j] * d[l] / (T)24 + s1 * d[j] * d[l] / (T)6 + s2 * d[l] / (T)2);
              gd[l] += G * (d[i] * d[j] * d[k] / (T)24 + s1 * d[j] * d[k] / (T)6 + s2 * d[k] / (T)2 + s3);
              g1[i] += G * d[j] * d[k] * d[l] / (T)6;
              g2[i * C + j] += G * d[k] * d[l] / (T)2;
              g3[(i * C + j) * C + k] += G * d[l];
            }
    // d = x_{r+1} - x_r
    if (r == r_hi - 1) for (int i = 0; i < C; ++i) { gw[i] = gd[i]; carry[i] = -gd[i]; }       // boundary row above: parked
    else for (int i = 0; i < C; ++i) { gx[(r + 1) * C + i] = gd[i] + carry[i]; carry[i] = -gd[i]; }
  }
  if (r_hi > r_lo) for (int i = 0; i < C; ++i) gx[r_lo * C + i] += carry[i];
  else for (int i = 0; i < C; ++i) gw[i] = (T)0;           // empty window: nothing parked
}

// pass 3: the parked boundary contributions, one lane per (series, channel), windows in order (several empty windows may
// share a row)
template <typename T>
__global__ __launch_bounds__(256) void logsig_boundary_kernel(const T* __restrict__ gsum, const int64_t* __restrict__ rows,
                                                              T* __restrict__ grad_x, int64_t B, int64_t L, int C,
                                                              int64_t n_windows, int n_words) {
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= B * C) return;
  const int64_t b = id / C;
  const int i = (int)(id - b * C);
  T* gx = grad_x + b * L * C + i;
  const T* parked = gsum + b * (n_windows + 1) * n_words + i;
  for (int64_t win = 0; win < n_windows; ++win) gx[rows[win + 1] * C] += parked[(win + 1) * n_words];
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

// Backward of path_eval w.r.t. the coefficients: grad_coeffs (zeroed by the caller) += d out / d coeffs ^T grad_out.
// Several query times may fall into one interval, so one lane owns a (series, channel) and walks the queries in order:
// no atomics, deterministic.
template <typename T, int DEGREE, int WHAT>
__global__ __launch_bounds__(256) void path_eval_backward_kernel(const T* __restrict__ grad_out, const T* __restrict__ knots,
                                                                 const T* __restrict__ tq, int64_t nq,
                                                                 T* __restrict__ grad_coeffs, int64_t B,
                                                                 int64_t n_intervals, int64_t C) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * C) return;
  const int64_t b = e / C, c = e - b * C;
  for (int64_t q = 0; q < nq; ++q) {
    T frac;
    const int64_t idx = locate(knots, n_intervals, tq[q], frac);
    const T g = grad_out[(b * nq + q) * C + c];
    if (DEGREE == CDE_PATH_CUBIC) {
      T* row = grad_coeffs + (b * n_intervals + idx) * 4 * C + c;
      if (WHAT == CDE_EVAL_DERIVATIVE) {            // b + (2c + 3d*frac)*frac
        row[C] += g; row[2 * C] += g * frac; row[3 * C] += g * frac * frac;
      } else {                                      // a + (b + (0.5*2c + 3d*frac/3)*frac)*frac
        row[0] += g; row[C] += g * frac; row[2 * C] += g * ((T)0.5 * frac * frac); row[3 * C] += g * (frac * frac * frac / (T)3);
      }
    } else {
      T* lo = grad_coeffs + (b * (n_intervals + 1) + idx) * C + c;
      const T width = knots[idx + 1] - knots[idx];
      const T w = WHAT == CDE_EVAL_DERIVATIVE ? (T)1 / width : frac / width;
      if (WHAT == CDE_EVAL_VALUE) lo[0] += g;
      lo[0] -= g * w; lo[C] += g * w;
    }
  }
}

template <typename T>
static int launch_path_eval_backward(const void* grad_out, const void* knots, const void* tq, int64_t nq, void* grad_coeffs,
                                     int64_t B, int64_t n_intervals, int64_t C, int degree, int what, hipStream_t s) {
  if (B * C == 0 || nq == 0) return CDE_OK;
  const unsigned grid = (unsigned)((B * C + 255) / 256);
#define CDE_PB(D, W) \
  path_eval_backward_kernel<T, D, W><<<grid, 256, 0, s>>>((const T*)grad_out, (const T*)knots, (const T*)tq, nq, (T*)grad_coeffs, B, n_intervals, C)
  if (degree == CDE_PATH_CUBIC && what == CDE_EVAL_DERIVATIVE) CDE_PB(CDE_PATH_CUBIC, CDE_EVAL_DERIVATIVE);
  else if (degree == CDE_PATH_CUBIC && what == CDE_EVAL_VALUE) CDE_PB(CDE_PATH_CUBIC, CDE_EVAL_VALUE);
  else if (degree == CDE_PATH_LINEAR && what == CDE_EVAL_DERIVATIVE) CDE_PB(CDE_PATH_LINEAR, CDE_EVAL_DERIVATIVE);
  else if (degree == CDE_PATH_LINEAR && what == CDE_EVAL_VALUE) CDE_PB(CDE_PATH_LINEAR, CDE_EVAL_VALUE);
  else return CDE_ERR_UNSUPPORTED;
#undef CDE_PB
  return check_launch();
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
  if (dtype == CDE_F32) return cde::launch_hermite<float>(x, t, coeffs, B, L, C, nullptr, s);
  if (dtype == CDE_F64) return cde::launch_hermite<double>(x, t, coeffs, B, L, C, nullptr, s);
  return CDE_ERR_DTYPE;
}

extern "C" int cde_hermite_bdiff_coeffs_checked(const void* x, const void* t, void* coeffs, int64_t B, int64_t L,
                                                int64_t C, int dtype, int* nan_flag, void* stream) {
  if (B < 0 || L < 2 || C < 1) return CDE_ERR_SHAPE;
  if (B == 0) return CDE_OK;
  if (!x || !t || !coeffs || !nan_flag) return CDE_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CDE_F32) return cde::launch_hermite<float>(x, t, coeffs, B, L, C, nan_flag, s);
  if (dtype == CDE_F64) return cde::launch_hermite<double>(x, t, coeffs, B, L, C, nan_flag, s);
  return CDE_ERR_DTYPE;
}

// K1 without a host round trip: the checked fit, then -- gated ON THE DEVICE by the flag the first launch may have raised
// to `generation` -- the missing-value fill of x into `scratch` and the fit of the filled series over `coeffs`.
extern "C" int cde_hermite_bdiff_coeffs_nonblocking(const void* x, const void* t, void* coeffs, void* scratch, int64_t B,
                                                    int64_t L, int64_t C, int dtype, int* nan_flag, int generation,
                                                    void* stream) {
  if (B < 0 || L < 2 || C < 1 || generation < 1) return CDE_ERR_SHAPE;
  if (B == 0) return CDE_OK;
  if (!x || !t || !coeffs || !scratch || !nan_flag) return CDE_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)((B * C + 255) / 256);
  int rc;
  if (dtype == CDE_F32) {
    rc = cde::launch_hermite<float>(x, t, coeffs, B, L, C, nan_flag, s, generation);
    if (rc != CDE_OK) return rc;
    cde::linear_fill_kernel<float><<<grid, 256, 0, s>>>((const float*)x, (const float*)t, (float*)scratch, B, L, C, nan_flag,
                                                        generation);
    return cde::launch_hermite<float>(scratch, t, coeffs, B, L, C, nullptr, s, 1, nan_flag, generation);
  }
  if (dtype == CDE_F64) {
    rc = cde::launch_hermite<double>(x, t, coeffs, B, L, C, nan_flag, s, generation);
    if (rc != CDE_OK) return rc;
    cde::linear_fill_kernel<double><<<grid, 256, 0, s>>>((const double*)x, (const double*)t, (double*)scratch, B, L, C,
                                                         nan_flag, generation);
    return cde::launch_hermite<double>(scratch, t, coeffs, B, L, C, nullptr, s, 1, nan_flag, generation);
  }
  return CDE_ERR_DTYPE;
}

extern "C" int cde_linear_fill_missing(const void* x, const void* t, void* out, int64_t B, int64_t L, int64_t C, int dtype,
                                       void* stream) {
  if (B < 0 || L < 2 || C < 1) return CDE_ERR_SHAPE;
  if (B == 0) return CDE_OK;
  if (!x || !t || !out) return CDE_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)((B * C + 255) / 256);
  if (dtype == CDE_F32)
    cde::linear_fill_kernel<float><<<grid, 256, 0, s>>>((const float*)x, (const float*)t, (float*)out, B, L, C);
  else if (dtype == CDE_F64)
    cde::linear_fill_kernel<double><<<grid, 256, 0, s>>>((const double*)x, (const double*)t, (double*)out, B, L, C);
  else
    return CDE_ERR_DTYPE;
  return cde::check_launch();
}

extern "C" int cde_linear_fill_missing_backward(const void* grad_out, const void* x, const void* t, void* grad_x,
                                                int64_t B, int64_t L, int64_t C, int dtype, void* stream) {
  if (B < 0 || L < 2 || C < 1) return CDE_ERR_SHAPE;
  if (B == 0) return CDE_OK;
  if (!grad_out || !x || !t || !grad_x) return CDE_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)((B * C + 255) / 256);
  if (dtype == CDE_F32)
    cde::linear_fill_backward_kernel<float><<<grid, 256, 0, s>>>((const float*)grad_out, (const float*)x, (const float*)t,
                                                                 (float*)grad_x, B, L, C);
  else if (dtype == CDE_F64)
    cde::linear_fill_backward_kernel<double><<<grid, 256, 0, s>>>((const double*)grad_out, (const double*)x,
                                                                  (const double*)t, (double*)grad_x, B, L, C);
  else
    return CDE_ERR_DTYPE;
  return cde::check_launch();
}


extern "C" int cde_hermite_bdiff_coeffs_backward(const void* grad_coeffs, const void* t, void* grad_x, int64_t B,
                                                 int64_t L, int64_t C, int dtype, void* stream) {
  if (B < 0 || L < 2 || C < 1) return CDE_ERR_SHAPE;
  if (B == 0) return CDE_OK;
  if (!grad_coeffs || !t || !grad_x) return CDE_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)((B * L * C + 255) / 256);
  if (dtype == CDE_F32)
    cde::hermite_bdiff_backward_kernel<float><<<grid, 256, 0, s>>>((const float*)grad_coeffs, (const float*)t, (float*)grad_x, B, L, C);
  else if (dtype == CDE_F64)
    cde::hermite_bdiff_backward_kernel<double><<<grid, 256, 0, s>>>((const double*)grad_coeffs, (const double*)t, (double*)grad_x, B, L, C);
  else return CDE_ERR_DTYPE;
  return cde::check_launch();
}

// dL/dh (B, L-1, C) of the fit, h_j = t_{j+1} - t_j (data without missing values): sum over B and C, then
// dL/dt_{j+1} += dL/dh_j and dL/dt_j -= dL/dh_j give the gradient w.r.t. the knot times.
extern "C" int cde_hermite_bdiff_coeffs_backward_dt(const void* grad_coeffs, const void* x, const void* t, void* grad_h,
                                                    int64_t B, int64_t L, int64_t C, int dtype, void* stream) {
  if (B < 0 || L < 2 || C < 1) return CDE_ERR_SHAPE;
  if (B == 0) return CDE_OK;
  if (!grad_coeffs || !x || !t || !grad_h) return CDE_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)((B * (L - 1) * C + 255) / 256);
  if (dtype == CDE_F32)
    cde::hermite_bdiff_backward_dt_kernel<float><<<grid, 256, 0, s>>>((const float*)grad_coeffs, (const float*)x,
                                                                      (const float*)t, (float*)grad_h, B, L, C);
  else if (dtype == CDE_F64)
    cde::hermite_bdiff_backward_dt_kernel<double><<<grid, 256, 0, s>>>((const double*)grad_coeffs, (const double*)x,
                                                                       (const double*)t, (double*)grad_h, B, L, C);
  else return CDE_ERR_DTYPE;
  return cde::check_launch();
}

extern "C" int cde_natural_cubic_coeffs(const void* x, const void* t, void* coeffs, int64_t B, int64_t L, int64_t C,
                                        int version, int has_missing, int dtype, void* stream) {
  if (B < 0 || L < 2 || C < 1 || (version != 0 && version != 1)) return CDE_ERR_SHAPE;
  if (B == 0) return CDE_OK;
  if (!x || !t || !coeffs) return CDE_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)((B * C + 255) / 256);
  if (dtype == CDE_F32)
    cde::natural_cubic_kernel<float><<<grid, 256, 0, s>>>((const float*)x, (const float*)t, (float*)coeffs, B, L, C, version, has_missing);
  else if (dtype == CDE_F64)
    cde::natural_cubic_kernel<double><<<grid, 256, 0, s>>>((const double*)x, (const double*)t, (double*)coeffs, B, L, C, version, has_missing);
  else return CDE_ERR_DTYPE;
  return cde::check_launch();
}

extern "C" size_t cde_natural_cubic_coeffs_backward_workspace_bytes(int64_t L, int dtype) {
  return (size_t)2 * (size_t)L * (dtype == CDE_F64 ? 8 : 4);
}

// grad_t_rows == NULL: gradient w.r.t. the values only.  Otherwise also `x` (the forward input), `kd_scratch` and
// `grad_t_rows`, each (B, L, C): grad_t_rows receives one partial dL/dt row per scalar path (sum them over B and C).
extern "C" int cde_natural_cubic_coeffs_backward(const void* grad_coeffs, const void* t, void* grad_x, void* workspace,
                                                 size_t workspace_bytes, int64_t B, int64_t L, int64_t C, int dtype,
                                                 const void* x, void* kd_scratch, void* grad_t_rows, void* stream) {
  if (B < 0 || L < 2 || C < 1) return CDE_ERR_SHAPE;
  if (B == 0) return CDE_OK;
  if (!grad_coeffs || !t || !grad_x || !workspace) return CDE_ERR_NULL;
  if (grad_t_rows && (!x || !kd_scratch)) return CDE_ERR_NULL;
  if (dtype != CDE_F32 && dtype != CDE_F64) return CDE_ERR_DTYPE;
  if (workspace_bytes < cde_natural_cubic_coeffs_backward_workspace_bytes(L, dtype)) return CDE_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)((B * C + 255) / 256);
#define CDE_NCB(T)                                                                                                  \
  do {                                                                                                              \
    cde::natural_cubic_aux_kernel<T><<<1, 64, 0, s>>>((const T*)t, (T*)workspace, L);                               \
    if (grad_t_rows)                                                                                                \
      cde::natural_cubic_backward_kernel<T, true><<<grid, 256, 0, s>>>((const T*)grad_coeffs, (const T*)t,          \
          (const T*)workspace, (T*)grad_x, B, L, C, (const T*)x, (T*)kd_scratch, (T*)grad_t_rows);                  \
    else                                                                                                            \
      cde::natural_cubic_backward_kernel<T, false><<<grid, 256, 0, s>>>((const T*)grad_coeffs, (const T*)t,         \
          (const T*)workspace, (T*)grad_x, B, L, C, nullptr, nullptr, nullptr);                                     \
  } while (0)
  if (dtype == CDE_F32) CDE_NCB(float); else CDE_NCB(double);
#undef CDE_NCB
  return cde::check_launch();
}


// Batches with missing entries (values only): `x` is the forward input, `workspace` has the shape of the coefficients.
extern "C" int cde_natural_cubic_coeffs_backward_missing(const void* grad_coeffs, const void* x, const void* t,
                                                         void* grad_x, void* workspace, int64_t B, int64_t L, int64_t C,
                                                         int version, int dtype, void* stream) {
  if (B < 0 || L < 2 || C < 1 || (version != 0 && version != 1)) return CDE_ERR_SHAPE;
  if (B == 0) return CDE_OK;
  if (!grad_coeffs || !x || !t || !grad_x || !workspace) return CDE_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)((B * C + 255) / 256);
  if (dtype == CDE_F32)
    cde::natural_cubic_backward_missing_kernel<float><<<grid, 256, 0, s>>>((const float*)grad_coeffs, (const float*)x,
        (const float*)t, (float*)grad_x, (float*)workspace, B, L, C, version);
  else if (dtype == CDE_F64)
    cde::natural_cubic_backward_missing_kernel<double><<<grid, 256, 0, s>>>((const double*)grad_coeffs, (const double*)x,
        (const double*)t, (double*)grad_x, (double*)workspace, B, L, C, version);
  else return CDE_ERR_DTYPE;
  return cde::check_launch();
}


extern "C" int cde_logsig_windows(const void* x, const int64_t* rows, const void* scale, const int32_t* words, void* out,
                                  int64_t B, int64_t L, int64_t C, int depth, int64_t n_windows, int n_words, int dtype,
                                  void* stream) {
  if (B < 0 || L < 1 || C < 1 || n_windows < 0 || n_words < 1) return CDE_ERR_SHAPE;
  // envelopes of the per-lane signature arrays: (8 channels, depth 3), (5, 4), (32, 2)
  const int env = (depth >= 1 && depth <= 3 && C <= 8) ? 0 : (depth == 4 && C <= 5) ? 1 : (depth >= 1 && depth <= 2 && C <= 32) ? 2 : -1;
  if (env < 0) return CDE_ERR_UNSUPPORTED;
  if (B == 0) return CDE_OK;
  if (!x || !rows || !scale || !words || !out) return CDE_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)((B * n_windows + 63) / 64), grid2 = (unsigned)((B * n_words + 255) / 256);
#define CDE_LS(T, MAXC, MAXD)                                                                                          \
  cde::logsig_windows_kernel<T, MAXC, MAXD><<<grid, 64, 0, s>>>((const T*)x, rows, (const T*)scale, words, (T*)out, B, L, \
                                                                (int)C, depth, n_windows, n_words)
  if (dtype == CDE_F32) {
    if (n_windows > 0) { if (env == 0) CDE_LS(float, 8, 3); else if (env == 1) CDE_LS(float, 5, 4); else CDE_LS(float, 32, 2); }
    cde::logsig_accumulate_kernel<float><<<grid2, 256, 0, s>>>((const float*)x, (float*)out, B, L, (int)C, n_windows, n_words);
  } else if (dtype == CDE_F64) {
    if (n_windows > 0) { if (env == 0) CDE_LS(double, 8, 3); else if (env == 1) CDE_LS(double, 5, 4); else CDE_LS(double, 32, 2); }
    cde::logsig_accumulate_kernel<double><<<grid2, 256, 0, s>>>((const double*)x, (double*)out, B, L, (int)C, n_windows, n_words);
  } else return CDE_ERR_DTYPE;
#undef CDE_LS
  return cde::check_launch();
}

// grad_out (B, n_windows + 1, n_words) -> grad_x (B, L, C) w.r.t. the filled series the forward call was given;
// `workspace` has the size of grad_out.
extern "C" int cde_logsig_windows_backward(const void* grad_out, const void* x, const int64_t* rows, const void* scale,
                                           const int32_t* words, void* grad_x, void* workspace, int64_t B, int64_t L,
                                           int64_t C, int depth, int64_t n_windows, int n_words, int dtype, void* stream) {
  if (B < 0 || L < 1 || C < 1 || n_windows < 0 || n_words < 1) return CDE_ERR_SHAPE;
  const int env = (depth >= 1 && depth <= 3 && C <= 8) ? 0 : (depth == 4 && C <= 5) ? 1 : (depth >= 1 && depth <= 2 && C <= 32) ? 2 : -1;
  if (env < 0) return CDE_ERR_UNSUPPORTED;
  if (B == 0) return CDE_OK;
  if (!grad_out || !x || !rows || !scale || !words || !grad_x || !workspace) return CDE_ERR_NULL;
  if (dtype != CDE_F32 && dtype != CDE_F64) return CDE_ERR_DTYPE;
  hipStream_t s = (hipStream_t)stream;
  cde::zero_async(grad_x, (size_t)(B * L * C) * (dtype == CDE_F64 ? 8 : 4), s);
  const unsigned grid = (unsigned)((B * n_windows + 63) / 64), grid2 = (unsigned)((B * n_words + 255) / 256);
#define CDE_LSB(T, MAXC, MAXD)                                                                                         \
  do {                                                                                                                 \
    cde::logsig_windows_backward_kernel<T, MAXC, MAXD><<<grid, 64, 0, s>>>((T*)workspace, (const T*)x, rows,           \
        (const T*)scale, words, (T*)grad_x, B, L, (int)C, depth, n_windows, n_words);                                  \
    cde::logsig_boundary_kernel<T><<<(unsigned)((B * C + 255) / 256), 256, 0, s>>>((const T*)workspace, rows,          \
        (T*)grad_x, B, L, (int)C, n_windows, n_words);                                                                 \
  } while (0)
  if (dtype == CDE_F32) {
    cde::logsig_suffix_kernel<float><<<grid2, 256, 0, s>>>((const float*)grad_out, (float*)workspace, (float*)grad_x, B, L, (int)C, n_windows, n_words);
    if (n_windows > 0) { if (env == 0) CDE_LSB(float, 8, 3); else if (env == 1) CDE_LSB(float, 5, 4); else CDE_LSB(float, 32, 2); }
  } else {
    cde::logsig_suffix_kernel<double><<<grid2, 256, 0, s>>>((const double*)grad_out, (double*)workspace, (double*)grad_x, B, L, (int)C, n_windows, n_words);
    if (n_windows > 0) { if (env == 0) CDE_LSB(double, 8, 3); else if (env == 1) CDE_LSB(double, 5, 4); else CDE_LSB(double, 32, 2); }
  }
#undef CDE_LSB
  return cde::check_launch();
}

extern "C" int cde_forward_fill(const void* x, void* out, int64_t B, int64_t L, int64_t C, int dtype, void* stream) {
  if (B < 0 || L < 1 || C < 1) return CDE_ERR_SHAPE;
  if (B == 0) return CDE_OK;
  if (!x || !out) return CDE_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)((B * C + 255) / 256);
  if (dtype == CDE_F32) cde::forward_fill_kernel<float><<<grid, 256, 0, s>>>((const float*)x, (float*)out, B, L, C);
  else if (dtype == CDE_F64) cde::forward_fill_kernel<double><<<grid, 256, 0, s>>>((const double*)x, (double*)out, B, L, C);
  else return CDE_ERR_DTYPE;
  return cde::check_launch();
}

extern "C" int cde_rectilinear_prepare(const void* x, void* out, int64_t B, int64_t L, int64_t C, int64_t time_index,
                                       int dtype, void* stream) {
  if (B < 0 || L < 1 || C < 1 || time_index < 0 || time_index >= C) return CDE_ERR_SHAPE;
  if (B == 0) return CDE_OK;
  if (!x || !out) return CDE_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)((B * C + 255) / 256);
  if (dtype == CDE_F32)
    cde::rectilinear_prepare_kernel<float><<<grid, 256, 0, s>>>((const float*)x, (float*)out, B, L, C, time_index);
  else if (dtype == CDE_F64)
    cde::rectilinear_prepare_kernel<double><<<grid, 256, 0, s>>>((const double*)x, (double*)out, B, L, C, time_index);
  else return CDE_ERR_DTYPE;
  return cde::check_launch();
}

extern "C" int cde_forward_fill_backward(const void* grad_out, const void* x, void* grad_x, int64_t B, int64_t L, int64_t C,
                                         int dtype, void* stream) {
  if (B < 0 || L < 1 || C < 1) return CDE_ERR_SHAPE;
  if (B == 0) return CDE_OK;
  if (!grad_out || !x || !grad_x) return CDE_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)((B * C + 255) / 256);
  if (dtype == CDE_F32)
    cde::forward_fill_backward_kernel<float, false><<<grid, 256, 0, s>>>((const float*)grad_out, (const float*)x, (float*)grad_x, B, L, C, -1);
  else if (dtype == CDE_F64)
    cde::forward_fill_backward_kernel<double, false><<<grid, 256, 0, s>>>((const double*)grad_out, (const double*)x, (double*)grad_x, B, L, C, -1);
  else return CDE_ERR_DTYPE;
  return cde::check_launch();
}

extern "C" int cde_rectilinear_prepare_backward(const void* grad_out, const void* x, void* grad_x, int64_t B, int64_t L,
                                                int64_t C, int64_t time_index, int dtype, void* stream) {
  if (B < 0 || L < 1 || C < 1 || time_index < 0 || time_index >= C) return CDE_ERR_SHAPE;
  if (B == 0) return CDE_OK;
  if (!grad_out || !x || !grad_x) return CDE_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)((B * C + 255) / 256);
  if (dtype == CDE_F32)
    cde::forward_fill_backward_kernel<float, true><<<grid, 256, 0, s>>>((const float*)grad_out, (const float*)x, (float*)grad_x, B, L, C, time_index);
  else if (dtype == CDE_F64)
    cde::forward_fill_backward_kernel<double, true><<<grid, 256, 0, s>>>((const double*)grad_out, (const double*)x, (double*)grad_x, B, L, C, time_index);
  else return CDE_ERR_DTYPE;
  return cde::check_launch();
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

extern "C" int cde_path_eval_backward(const void* grad_out, const void* knots, const void* tq, int64_t nq,
                                      void* grad_coeffs, int64_t B, int64_t n_intervals, int64_t C, int degree, int what,
                                      int dtype, void* stream) {
  if (B < 0 || n_intervals < 1 || C < 1 || nq < 0) return CDE_ERR_SHAPE;
  if (B == 0 || nq == 0) return CDE_OK;
  if (!grad_out || !knots || !tq || !grad_coeffs) return CDE_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  if (dtype == CDE_F32) return cde::launch_path_eval_backward<float>(grad_out, knots, tq, nq, grad_coeffs, B, n_intervals, C, degree, what, s);
  if (dtype == CDE_F64) return cde::launch_path_eval_backward<double>(grad_out, knots, tq, nq, grad_coeffs, B, n_intervals, C, degree, what, s);
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

// ---------------------------------------------------------------------------------------------- tuning table
#include <atomic>
namespace cde {
static constexpr int64_t OPTION_DEFAULTS[CDE_OPT_COUNT] = {
    /* K3_FORM */ 0, /* K3_WAVES */ 0, /* K3D_WAVES */ 0, /* K2M_NO_SPLIT */ 0, /* K3M_NO_SPLIT */ 0, /* K3M_SPLIT4 */ 0,
    /* K3M_S8_TILES */ -1, /* K4_NO_SPLIT */ 0, /* K4M_NO_SPLIT */ 0, /* K4M_SPLIT_TILES */ -1, /* K4AM_WAVES */ 0,
    /* K4AM_S8_TILES */ -1, /* K4AM_SPLIT4 */ 0, /* K4AM_NO_SPLIT */ 0, /* K4AM_NO_SMALL_REDUCE */ 0, /* K4AM_SPS */ 0,
    /* K4AM_NO_FSAL */ 0, /* WIDE_SCRATCH_BYTES */ 0};
static std::atomic<int64_t> g_options[CDE_OPT_COUNT] = {
    {0}, {0}, {0}, {0}, {0}, {0}, {-1}, {0}, {0}, {-1}, {0}, {-1}, {0}, {0}, {0}, {0}, {0}, {0}};
int64_t option(int key) { return g_options[key].load(std::memory_order_relaxed); }
}  // namespace cde

extern "C" int cde_set_option(int key, int64_t value) {
  if (key < 0 || key >= CDE_OPT_COUNT) return CDE_ERR_SHAPE;
  cde::g_options[key].store(value, std::memory_order_relaxed);
  return CDE_OK;
}
extern "C" int64_t cde_get_option(int key) {
  if (key < 0 || key >= CDE_OPT_COUNT) return INT64_MIN;
  return cde::option(key);
}
extern "C" int cde_reset_options(void) {
  for (int k = 0; k < CDE_OPT_COUNT; ++k) cde::g_options[k].store(cde::OPTION_DEFAULTS[k], std::memory_order_relaxed);
  return CDE_OK;
}

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
