// rk4_generic.hip -- fused RK4 (3/8 rule) CDE solve + continuous adjoint, VALU version.
//
// Any hidden size H <= 256, any channel count C, f32 or f64 state, f32 or f64 time grid, cubic or
// linear control, ACT_NONE or ACT_TANH.  This is the correctness path (small configs, float64,
// odd shapes) and the on-device cross-check of the MFMA kernels; the headline configuration
// (f32, H=32, C=8) runs rk4_mfma.hip instead.
//
// Work decomposition: a workgroup of NT = NS*H lanes owns NS series at a time (grid-stride over
// series tiles); lane (s, h) keeps its own RK state for hidden unit h of series s in registers.
// Per stage the stage state and the control derivative are exchanged through LDS, every lane
// forms its row of f(z) dX/dt with W streamed from L1/L2 (W is at most a few tens of KB).
#include "cde_common.h"

namespace cde {

template <typename T>
struct GenericArgs {
  const T* coeffs; const T* knots; int64_t n_intervals;
  const T* W; const T* bias; int act;
  int64_t B, C, H;
  int NS;  // series per workgroup tile
};

// derivative of channel c of series `series` at (idx, frac)
template <typename T, int DEGREE>
__device__ __forceinline__ T control_derivative(const GenericArgs<T>& g, int64_t series, int64_t idx, T frac, int64_t c) {
  if (DEGREE == CDE_PATH_CUBIC) {
    const T* row = g.coeffs + (series * g.n_intervals + idx) * 4 * g.C;
    return cubic_derivative(row[g.C + c], row[2 * g.C + c], row[3 * g.C + c], frac);
  } else {
    const T* lo = g.coeffs + (series * (g.n_intervals + 1) + idx) * g.C;
    return (lo[g.C + c] - lo[c]) / (g.knots[idx + 1] - g.knots[idx]);
  }
}

// f_h = sum_c act(bias[hC+c] + sum_k W[hC+c][k] z_k) * dX_c for lane (s,h); z, dX in LDS.
template <typename T>
__device__ __forceinline__ T field_row(const GenericArgs<T>& g, const T* zs, const T* dx, int h) {
  const int H = (int)g.H, C = (int)g.C;
  T acc = (T)0;
  for (int c = 0; c < C; ++c) {
    const T* w = g.W + ((int64_t)h * C + c) * H;
    T y = g.bias[h * C + c];
    for (int k = 0; k < H; ++k) y = fma_t(w[k], zs[k], y);
    if (g.act == CDE_ACT_TANH) y = tanh_t(y);
    acc = fma_t(y, dx[c], acc);
  }
  return acc;
}

// ------------------------------------------------------------------------------------------ forward
template <typename T, typename TT, int DEGREE>
__global__ void rk4_forward_generic(GenericArgs<T> g, const T* __restrict__ z0, const TT* __restrict__ grid,
                                    int64_t n_grid, const TT* __restrict__ t_out, int64_t n_out, T* __restrict__ z_out,
                                    const int64_t* __restrict__ stage_index, const T* __restrict__ stage_frac) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T* zs = reinterpret_cast<T*>(smem_raw);          // [NS][H]
  T* dx = zs + (int64_t)g.NS * g.H;                // [NS][C]
  const int H = (int)g.H, C = (int)g.C, NS = g.NS;
  const int tid = threadIdx.x;
  const int s = tid / H, h = tid - s * H;
  const bool lane_on = s < NS;
  const int64_t n_tiles = (g.B + NS - 1) / NS;

  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t series = tile * NS + s;
    const bool valid = lane_on && series < g.B;
    T y0 = valid ? z0[series * H + h] : (T)0;
    if (valid) z_out[(series * n_out + 0) * H + h] = y0;
    int64_t j = 1;
    for (int64_t k = 0; k + 1 < n_grid; ++k) {
      const StageClock<TT> clk(grid[k], grid[k + 1]);
      const T dt = (T)clk.dt;
      T k1 = 0, k2 = 0, k3 = 0, k4 = 0, zst = y0;
#pragma unroll 1
      for (int stage = 0; stage < 4; ++stage) {
        const int64_t idx = stage_index[4 * k + stage];   // table from stage_table_kernel (api.hip)
        const T frac = stage_frac[4 * k + stage];
        // stage state (torchdiffeq rk4_alt_step_func association order)
        if (stage == 1) zst = y0 + dt * k1 * (T)(1.0 / 3.0);
        else if (stage == 2) zst = y0 + dt * (k2 - k1 * (T)(1.0 / 3.0));
        else if (stage == 3) zst = y0 + dt * (k1 - k2 + k3);
        __syncthreads();  // previous stage's LDS readers are done
        if (lane_on) zs[s * H + h] = zst;
        for (int e = tid; e < NS * C; e += blockDim.x) {
          const int s2 = e / C, c = e - s2 * C;
          int64_t ser = tile * NS + s2;
          ser = ser < g.B ? ser : g.B - 1;
          dx[e] = control_derivative<T, DEGREE>(g, ser, idx, frac, c);
        }
        __syncthreads();
        T f = (T)0;
        if (lane_on) f = field_row(g, zs + s * H, dx + s * C, h);
        if (stage == 0) k1 = f; else if (stage == 1) k2 = f; else if (stage == 2) k3 = f; else k4 = f;
      }
      const T y1 = y0 + (k1 + (T)3 * (k2 + k3) + k4) * dt * (T)0.125;
      // outputs landing in (t0, t1]: linear interpolation, end points verbatim
      while (j < n_out && clk.t1 >= t_out[j]) {
        const TT tj = t_out[j];
        T v;
        if (tj == clk.t0) v = y0;
        else if (tj == clk.t1) v = y1;
        else { const T slope = (T)((tj - clk.t0) / (clk.t1 - clk.t0)); v = y0 + slope * (y1 - y0); }
        if (valid) z_out[(series * n_out + j) * H + h] = v;
        ++j;
      }
      y0 = y1;
    }
  }
}

// ------------------------------------------------------------------------------------------ adjoint
// Reverse-time (s = -t) RK4 on (y, a, a_params):  dy/ds = -f(-s, y),  da/ds = a^T df/dy,
// d a_p/ds = a^T df/dp.  Parameter gradients do not feed back into the dynamics, so their RK
// quadrature is accumulated stage by stage with the 3/8 weights (1,3,3,1)/8 * ds.
// Per-workgroup partial sums live in `partial[blockIdx][H*C*H + H*C]`; reduce_partials() adds them
// in block order (deterministic).
template <typename T, typename TT, int DEGREE>
__global__ void rk4_adjoint_generic(GenericArgs<T> g, const T* __restrict__ z_saved, const T* __restrict__ grad_out,
                                    const TT* __restrict__ sgrid, const int64_t* __restrict__ seg_off, int64_t n_out,
                                    T* __restrict__ grad_z0, T* __restrict__ partial,
                                    const int64_t* __restrict__ stage_index, const T* __restrict__ stage_frac) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int H = (int)g.H, C = (int)g.C, NS = g.NS;
  T* ys = reinterpret_cast<T*>(smem_raw);   // [NS][H]   stage state
  T* dx = ys + NS * H;                      // [NS][C]
  T* gy = dx + NS * C;                      // [NS][H*C] a_h * dX_c * act'(Y_hc)
  const int tid = threadIdx.x;
  const int s = tid / H, h = tid - s * H;
  const bool lane_on = s < NS;
  const int64_t n_tiles = (g.B + NS - 1) / NS;
  const int64_t P = (int64_t)H * C * H + (int64_t)H * C;
  T* mine = partial + (int64_t)blockIdx.x * P;
  for (int64_t e = tid; e < P; e += blockDim.x) mine[e] = (T)0;

  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t series = tile * NS + s;
    const bool valid = lane_on && series < g.B;
    T y0 = valid ? z_saved[(series * n_out + (n_out - 1)) * H + h] : (T)0;
    T a0 = valid ? grad_out[(series * n_out + (n_out - 1)) * H + h] : (T)0;
    for (int64_t p = 0; p + 1 < n_out; ++p) {
      const int64_t i_out = n_out - 1 - p;  // integrating from t_out[i_out] down to t_out[i_out-1]
      for (int64_t k = seg_off[p]; k + 1 < seg_off[p + 1]; ++k) {
        const StageClock<TT> clk(sgrid[k], sgrid[k + 1]);
        const T ds = (T)clk.dt;
        T ky1 = 0, ky2 = 0, ky3 = 0, ky4 = 0, ka1 = 0, ka2 = 0, ka3 = 0, ka4 = 0, yst = y0, ast = a0;
#pragma unroll 1
        for (int stage = 0; stage < 4; ++stage) {
          // table entry for t = -s (torchdiffeq _ReverseFunc evaluates base_func(-t, y))
          const int64_t idx = stage_index[4 * k + stage];
          const T frac = stage_frac[4 * k + stage];
          if (stage == 1) { yst = y0 + ds * ky1 * (T)(1.0 / 3.0); ast = a0 + ds * ka1 * (T)(1.0 / 3.0); }
          else if (stage == 2) { yst = y0 + ds * (ky2 - ky1 * (T)(1.0 / 3.0)); ast = a0 + ds * (ka2 - ka1 * (T)(1.0 / 3.0)); }
          else if (stage == 3) { yst = y0 + ds * (ky1 - ky2 + ky3); ast = a0 + ds * (ka1 - ka2 + ka3); }
          __syncthreads();
          if (lane_on) ys[s * H + h] = yst;
          for (int e = tid; e < NS * C; e += blockDim.x) {
            const int s2 = e / C, c = e - s2 * C;
            int64_t ser = tile * NS + s2;
            ser = ser < g.B ? ser : g.B - 1;
            dx[e] = control_derivative<T, DEGREE>(g, ser, idx, frac, c);
          }
          __syncthreads();
          // forward field row + gY row for (s,h)
          T f = (T)0;
          if (lane_on) {
            const T* zrow = ys + s * H;
            const T* drow = dx + s * C;
            for (int c = 0; c < C; ++c) {
              const T* w = g.W + ((int64_t)h * C + c) * H;
              T y = g.bias[h * C + c];
              for (int kk = 0; kk < H; ++kk) y = fma_t(w[kk], zrow[kk], y);
              T dact = (T)1;
              if (g.act == CDE_ACT_TANH) { y = tanh_t(y); dact = (T)1 - y * y; }
              f = fma_t(y, drow[c], f);
              gy[((int64_t)s * H + h) * C + c] = valid ? ast * drow[c] * dact : (T)0;
            }
          }
          __syncthreads();
          // vjp_y for (s, k=h): sum over (h', c) of gY[s][h'][c] * W[h'C+c][k]
          T va = (T)0;
          if (lane_on) {
            const T* grow = gy + (int64_t)s * H * C;
            for (int hc = 0; hc < H * C; ++hc) va = fma_t(grow[hc], g.W[(int64_t)hc * H + h], va);
          }
          // parameter-gradient quadrature for this stage
          const T wq = ((stage == 0 || stage == 3) ? (T)0.125 : (T)0.375) * ds;
          for (int64_t e = tid; e < P; e += blockDim.x) {
            T sum = (T)0;
            if (e < (int64_t)H * C * H) {
              const int hc = (int)(e / H), kk = (int)(e - (int64_t)hc * H);
              for (int s2 = 0; s2 < NS; ++s2) sum = fma_t(gy[(int64_t)s2 * H * C + hc], ys[s2 * H + kk], sum);
            } else {
              const int hc = (int)(e - (int64_t)H * C * H);
              for (int s2 = 0; s2 < NS; ++s2) sum += gy[(int64_t)s2 * H * C + hc];
            }
            mine[e] = fma_t(wq, sum, mine[e]);
          }
          const T ky = -f, ka = va;
          if (stage == 0) { ky1 = ky; ka1 = ka; } else if (stage == 1) { ky2 = ky; ka2 = ka; }
          else if (stage == 2) { ky3 = ky; ka3 = ka; } else { ky4 = ky; ka4 = ka; }
        }
        y0 = y0 + (ky1 + (T)3 * (ky2 + ky3) + ky4) * ds * (T)0.125;
        a0 = a0 + (ka1 + (T)3 * (ka2 + ka3) + ka4) * ds * (T)0.125;
      }
      // torchdiffeq adjoint: re-seed y from the stored forward value, add the incoming gradient
      if (valid) {
        y0 = z_saved[(series * n_out + (i_out - 1)) * H + h];
        a0 += grad_out[(series * n_out + (i_out - 1)) * H + h];
      }
    }
    if (valid) grad_z0[series * H + h] = a0;
    __syncthreads();
  }
}

template <typename T>
__global__ void reduce_partials_kernel(const T* __restrict__ partial, int64_t n_blocks, int64_t P, int64_t n_w,
                                       T* __restrict__ grad_W, T* __restrict__ grad_b) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= P) return;
  T sum = (T)0;
  for (int64_t b = 0; b < n_blocks; ++b) sum += partial[b * P + e];
  if (e < n_w) grad_W[e] = sum; else grad_b[e - n_w] = sum;
}

// ------------------------------------------------------------------------------------------ host side
// series per workgroup tile: at most 256 lanes, at most 16 series, and the tile's LDS (stage state, control slope and
// -- adjoint -- the H*C dL/dY values per series) within 64 KB; 0 = the shape does not fit at all
constexpr size_t GENERIC_LDS_LIMIT = 64 * 1024;
static inline int generic_ns(int64_t H, int64_t C, size_t elem, bool adjoint) {
  if (H < 1 || H > 256 || C < 1) return 0;
  int ns = (int)(256 / H);
  ns = ns > 16 ? 16 : ns;
  const size_t per_series = (size_t)(adjoint ? H + C + H * C : H + C) * elem;
  const size_t fit = GENERIC_LDS_LIMIT / per_series;
  return fit < (size_t)ns ? (int)fit : ns;
}
static inline int64_t generic_blocks(int64_t B, int ns) {
  int64_t tiles = (B + ns - 1) / ns;
  return tiles < 1 ? 1 : (tiles > 1024 ? 1024 : tiles);
}

bool generic_applicable(int64_t C, int64_t H, size_t elem, bool adjoint) { return generic_ns(H, C, elem, adjoint) >= 1; }

size_t generic_adjoint_workspace_bytes(int64_t B, int64_t C, int64_t H, size_t elem) {
  const int ns = generic_ns(H, C, elem, true);
  if (ns < 1) return 0;
  return (size_t)generic_blocks(B, ns) * (size_t)(H * C * H + H * C) * elem;
}

template <typename T, typename TT>
int launch_forward_generic(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                           const void* bias, int act, const void* z0, const void* grid, int64_t n_grid,
                           const void* t_out, int64_t n_out, void* z_out, int64_t B, int64_t C, int64_t H,
                           const int64_t* stage_index, const void* stage_frac, hipStream_t s) {
  const int ns = generic_ns(H, C, sizeof(T), false);
  if (ns < 1) return CDE_ERR_SHAPE;
  GenericArgs<T> g{(const T*)coeffs, (const T*)knots, n_intervals, (const T*)W, (const T*)bias, act, B, C, H, ns};
  const int nt = ((g.NS * (int)H + 63) / 64) * 64;
  const size_t lds = (size_t)g.NS * (H + C) * sizeof(T);
  const unsigned blocks = (unsigned)generic_blocks(B, ns);
  if (degree == CDE_PATH_CUBIC)
    rk4_forward_generic<T, TT, CDE_PATH_CUBIC><<<blocks, nt, lds, s>>>(g, (const T*)z0, (const TT*)grid, n_grid, (const TT*)t_out, n_out, (T*)z_out, stage_index, (const T*)stage_frac);
  else if (degree == CDE_PATH_LINEAR)
    rk4_forward_generic<T, TT, CDE_PATH_LINEAR><<<blocks, nt, lds, s>>>(g, (const T*)z0, (const TT*)grid, n_grid, (const TT*)t_out, n_out, (T*)z_out, stage_index, (const T*)stage_frac);
  else
    return CDE_ERR_UNSUPPORTED;
  return check_launch();
}

template <typename T, typename TT>
int launch_adjoint_generic(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                           const void* bias, int act, const void* z_saved, const void* grad_out, const void* sgrid,
                           const int64_t* seg_off, int64_t n_out, void* grad_z0, void* grad_W, void* grad_b,
                           int64_t B, int64_t C, int64_t H, const int64_t* stage_index, const void* stage_frac,
                           void* partial_v, hipStream_t s) {
  const int ns = generic_ns(H, C, sizeof(T), true);
  if (ns < 1) return CDE_ERR_SHAPE;
  GenericArgs<T> g{(const T*)coeffs, (const T*)knots, n_intervals, (const T*)W, (const T*)bias, act, B, C, H, ns};
  const int nt = ((g.NS * (int)H + 63) / 64) * 64;
  const size_t lds = (size_t)g.NS * (H + C + H * C) * sizeof(T);
  const int64_t blocks = generic_blocks(B, ns);
  const int64_t P = H * C * H + H * C;
  T* partial = (T*)partial_v;
  if (degree == CDE_PATH_CUBIC)
    rk4_adjoint_generic<T, TT, CDE_PATH_CUBIC><<<(unsigned)blocks, nt, lds, s>>>(g, (const T*)z_saved, (const T*)grad_out, (const TT*)sgrid, seg_off, n_out, (T*)grad_z0, partial, stage_index, (const T*)stage_frac);
  else if (degree == CDE_PATH_LINEAR)
    rk4_adjoint_generic<T, TT, CDE_PATH_LINEAR><<<(unsigned)blocks, nt, lds, s>>>(g, (const T*)z_saved, (const T*)grad_out, (const TT*)sgrid, seg_off, n_out, (T*)grad_z0, partial, stage_index, (const T*)stage_frac);
  else
    return CDE_ERR_UNSUPPORTED;
  int rc = check_launch();
  if (rc != CDE_OK) return rc;
  reduce_partials_kernel<T><<<(unsigned)((P + 255) / 256), 256, 0, s>>>(partial, blocks, P, H * C * H, (T*)grad_W, (T*)grad_b);
  return check_launch();
}

// explicit instantiations used by api.hip
#define CDE_INST(T, TT)                                                                                                  \
  template int launch_forward_generic<T, TT>(const void*, const void*, int64_t, int, const void*, const void*, int,    \
                                             const void*, const void*, int64_t, const void*, int64_t, void*, int64_t,  \
                                             int64_t, int64_t, const int64_t*, const void*, hipStream_t);              \
  template int launch_adjoint_generic<T, TT>(const void*, const void*, int64_t, int, const void*, const void*, int,    \
                                             const void*, const void*, const void*, const int64_t*, int64_t, void*,    \
                                             void*, void*, int64_t, int64_t, int64_t, const int64_t*, const void*,     \
                                             void*, hipStream_t);
CDE_INST(float, float)
CDE_INST(float, double)
CDE_INST(double, double)
CDE_INST(double, float)
#undef CDE_INST

}  // namespace cde
