// dopri5.hip -- K4: adaptive Dormand-Prince 5(4) CDE solve for the affine vector-field family.
//
// Replaces torchdiffeq.odeint(method='dopri5') behind reference solver.py:226-227 (torchcde's DEFAULT method:
// cdeint passes no `method` unless the user does, README.md:174) together with _VectorField.forward
// (solver.py:117-135) and the control derivative inside it.  Semantics restated in oracle/odeint.py (_Dopri5):
//   * ONE step size for the whole batch: error ratio = RMS over all B*H elements of err/(atol+rtol*max|y0|,|y1|)
//   * time-like quantities (t, dt, tolerances, controller) in float64, state in `dtype`
//   * stage times formed in the state dtype: t0 + alpha_i*dt, the two alpha == 1 stages at nextafter(t1, -inf)
//   * jump_t: steps are clipped to land exactly on the next jump time and f is re-evaluated just after it
//   * first step by Hairer's rule (two extra evaluations), outputs by the 4th-order dense interpolant
//
// Execution model.  A batch-global decision per attempted step needs a grid-wide reduction; instead of a host
// round trip or a cooperative grid barrier, every launch of `dopri5_attempt_kernel` is "finish the previous
// attempt, start the next": in its prologue each workgroup re-derives the accept/reject decision from the
// per-workgroup error partials the previous launch left in global memory (summed in a fixed order: the decision
// is bit-identical in every workgroup and run-to-run), commits its series' state, emits any outputs the accepted
// step covered, and then runs the 6 new stages.  The controller state ping-pongs between two structs.  The host
// only queues launches and looks at a done flag every few dozen of them.
#include "cde_dopri.h"
#include "cde_split.h"

namespace cde {

template <typename T>
struct DopriArgs {
  const T* coeffs; const T* knots; int64_t n_intervals; int degree;
  const T* W; const T* bias; int act;
  const T* z0; const double* t_out; int64_t n_out; const double* jump_t; int64_t n_jump;
  double rtol, atol, safety, ifactor, dfactor;
  T* z_out; int64_t B, C, H; int NS;
  DopriCtrl* ctrl;              // [2]
  T* state;                     // [2][5][B*H]: y0, y1, k0 (f at t0), k6 (f at t1), y_mid (dense-output midpoint)
  const float* w16;             // MFMA kernels: the two 16x16x4 weight images, built once per solve
  double* partial;              // [2][n_blocks][2], accumulated in float64 whatever the state dtype
  int64_t n_blocks_alloc;
  const T* W1; const T* bias1; int width;      // two-layer fields: the hidden layer (W, bias are then the output layer)
  double* trace;                // [CDE_DOPRI5_TRACE_STEPS][3]: (t0, t1, clipped onto a jump time) of every accepted step
  const double* ext_sums;       // sharded batch: the pending sums, already added up over ALL shards (else nullptr)
  int64_t B_global;             // number of series the error norm runs over (0: this call's B)
};

// vector field row for lane (s,h): sum_c act(bias + W z) dX_c, control derivative at time ts
template <typename T>
__device__ __forceinline__ T dopri_field(const DopriArgs<T>& g, T* zs, T* dx, T zval, T ts, int64_t tile, int s, int h,
                                         bool lane_on) {
  const int H = (int)g.H, C = (int)g.C, NS = g.NS;
  T frac;
  const int64_t idx = locate(g.knots, g.n_intervals, ts, frac);
  __syncthreads();
  if (lane_on) zs[s * H + h] = zval;
  for (int e = threadIdx.x; e < NS * C; e += blockDim.x) {
    const int s2 = e / C, c = e - s2 * C;
    int64_t ser = tile * NS + s2;
    ser = ser < g.B ? ser : g.B - 1;
    T d;
    if (g.degree == CDE_PATH_CUBIC) {
      const T* row = g.coeffs + (ser * g.n_intervals + idx) * 4 * C;
      d = cubic_derivative(row[C + c], row[2 * C + c], row[3 * C + c], frac);
    } else {
      const T* lo = g.coeffs + (ser * (g.n_intervals + 1) + idx) * C;
      d = (lo[C + c] - lo[c]) / (g.knots[idx + 1] - g.knots[idx]);
    }
    dx[e] = d;
  }
  __syncthreads();
  T acc = (T)0;
  if (lane_on) {
    const T* zrow = zs + s * H;
    const T* drow = dx + s * C;
    for (int c = 0; c < C; ++c) {
      const T* w = g.W + ((int64_t)h * C + c) * H;
      T y = g.bias[h * C + c];
      for (int k = 0; k < H; ++k) y = fma_t(w[k], zrow[k], y);
      if (g.act == CDE_ACT_TANH) y = tanh_t(y);
      acc = fma_t(y, drow[c], acc);
    }
  }
  return acc;
}

// block-wide sum of two values, result valid in every thread (fixed tree order)
__device__ __forceinline__ void block_sum2(double& a, double& b, double* red) {
  const int tid = threadIdx.x, n = blockDim.x;
  __syncthreads();
  red[tid] = a; red[n + tid] = b;
  __syncthreads();
  for (int off = 1; off < n; off <<= 1) {
    double va = 0.0, vb = 0.0;
    const bool act = (tid % (2 * off)) == 0 && tid + off < n;
    if (act) { va = red[tid + off]; vb = red[n + tid + off]; }
    __syncthreads();
    if (act) { red[tid] += va; red[n + tid] += vb; }
    __syncthreads();
  }
  a = red[0]; b = red[n];
  __syncthreads();
}

// What one launch has to do, derived identically by every thread from the controller struct and the pending sums.
template <typename T>
struct DopriPlan {
  bool accept;                  // decision on the pending attempt (phase 3 only)
  int mode;                     // this launch: 0 = f0 norms, 1 = f1 norm, 2 = attempt, 3 = nothing more
  double t0, t1, dt, dt_done;   // new attempt [t0, t1]; dt_done = step size of the attempt just decided
  T h0_state;
  int64_t emit_from, emit_to;   // outputs covered by the step just accepted
};

template <typename T>
__device__ __forceinline__ DopriPlan<T> dopri_controller(const DopriArgs<T>& g, DopriCtrl& c, double sum0, double sum1) {
  const double n_elems = (double)((g.B_global > 0 ? g.B_global : g.B) * g.H);
  bool accept = false;
  int mode;                     // what this launch computes: 0 = f0 norms, 1 = f1 norm, 2 = attempt, 3 = nothing more
  double t0 = 0, t1 = 0, dt = 0;
  T h0_state = (T)0;
  int on_jump = 0;
  if (c.phase == 0) {
    mode = 0;
    c.t_lo = c.t_hi = g.t_out[0];
    c.i_out = 1; c.i_jump = 0; c.n_accept = c.n_reject = 0; c.refresh = 0;
    // first jump strictly after nothing: torchdiffeq keeps jump times >= t0 and starts at bisect(jump_t, t0)
    int64_t j = 0;
    while (j < g.n_jump && g.jump_t[j] < c.t_hi) ++j;          // drop jumps before t0
    int64_t first = j;
    while (j < g.n_jump && g.jump_t[j] <= c.t_hi) ++j;         // bisect_right
    c.i_jump = j - first;
    if (g.n_jump - first > 0 && c.i_jump > g.n_jump - first - 1) c.i_jump = g.n_jump - first - 1;
    c.pad = (int32_t)first;                                    // offset of the first kept jump time
  } else if (c.phase == 1) {
    // Hairer: d0 = ||y0/scale||, d1 = ||f0/scale||
    const T d0 = (T)sqrt(sum0 / n_elems), d1 = (T)sqrt(sum1 / n_elems);
    T h0;
    if (d0 < (T)1e-5 || d1 < (T)1e-5) h0 = (T)1e-6; else h0 = (T)0.01 * d0 / d1;
    h0 = h0 < 0 ? -h0 : h0;
    c.h0 = (double)h0;
    h0_state = h0;
    c.dt = (double)d1;                                          // park d1 for phase 2
    mode = 1;
  } else if (c.phase == 2) {
    const T h0 = (T)c.h0, d1 = (T)c.dt;
    const T d2 = (T)sqrt(sum0 / n_elems) / h0;
    T h1;
    if (d1 <= (T)1e-15 && d2 <= (T)1e-15) {
      const T a = (T)1e-6, b = h0 * (T)1e-3;
      h1 = a > b ? a : b;
    } else {
      // torch: (0.01 / max(d1, d2)) ** (1/5) on a 0-d tensor of the state dtype
      const T m = d1 > d2 ? d1 : d2;
      if (sizeof(T) == 4) h1 = (T)powf((float)((T)0.01 / m), (float)(1.0 / 5.0));
      else h1 = (T)pow((double)((T)0.01 / m), 1.0 / 5.0);
    }
    h1 = h1 < 0 ? -h1 : h1;
    const T hundred = (T)100 * h0;
    c.dt = (double)(hundred < h1 ? hundred : h1);
    mode = 2;
  } else {
    // decide the pending attempt: ratio = sqrt(mean((err/tol)^2))
    const T ratio_t = (T)sqrt(sum0 / n_elems);
    accept = ratio_t <= (T)1;
    // (min_step = 0, max_step = inf: the extra accept/reject overrides of torchdiffeq never fire)
    if (accept) {
      c.n_accept++;
      c.t_lo = c.t_hi; c.t_hi = c.t1_try;
      if (g.trace && blockIdx.x == 0 && threadIdx.x == 0 && c.n_accept <= CDE_DOPRI5_TRACE_STEPS) {
        g.trace[3 * (c.n_accept - 1)] = c.t_lo;                  // the step sequence of the solve (tests replay it
        g.trace[3 * (c.n_accept - 1) + 1] = c.t_hi;              // through the oracle; sharded runs can compare it)
        g.trace[3 * (c.n_accept - 1) + 2] = c.on_jump ? 1.0 : 0.0;
      }
      c.refresh = 0;
      if (c.on_jump) {
        const int64_t kept = g.n_jump - c.pad;
        if (c.i_jump != kept - 1) c.i_jump++;
        c.refresh = 1;
      }
    } else {
      c.n_reject++;
      c.t_lo = c.t_hi;                                          // oracle: t_lo, t_hi = t0, t0
    }
    // next step size (float64)
    const double ratio = (double)ratio_t;
    double factor;
    if (ratio == 0.0) factor = g.ifactor;
    else {
      const double dfac = ratio < 1.0 ? 1.0 : g.dfactor;
      double f = g.safety / pow(ratio, 1.0 / 5.0);
      f = f > dfac ? f : dfac;
      factor = g.ifactor < f ? g.ifactor : f;
    }
    c.dt = c.dt_try * factor;
    mode = 2;
  }

  // how many outputs does the accepted interval [t_lo, t_hi] cover?
  int64_t emit_from = c.i_out, emit_to = c.i_out;
  if (c.phase == 3 && accept) {
    while (emit_to < g.n_out && !(g.t_out[emit_to] > c.t_hi)) ++emit_to;
    c.i_out = emit_to;
  }
  const bool finished = (c.phase == 3 && c.i_out >= g.n_out);
  if (finished) mode = 3;
  const double dt_done = c.dt_try;                              // step size of the attempt just decided (dense output)

  if (mode == 2) {
    // new attempt from t_hi
    t0 = c.t_hi;
    dt = c.dt;
    if (!(dt == dt) || dt > 1e300 || dt < -1e300) dt = 0.0;     // non-finite -> min_step (0)
    t1 = t0 + dt;
    const int64_t kept = g.n_jump - c.pad;
    if (kept > 0) {
      const double nxt = g.jump_t[c.pad + c.i_jump];
      if (t0 < nxt && nxt < t0 + dt) { on_jump = 1; t1 = nxt; dt = t1 - t0; }
    }
    c.t1_try = t1; c.dt_try = dt; c.on_jump = on_jump;
  }

  DopriPlan<T> plan;
  plan.accept = accept; plan.mode = mode; plan.t0 = t0; plan.t1 = t1; plan.dt = dt; plan.dt_done = dt_done;
  plan.h0_state = h0_state; plan.emit_from = emit_from; plan.emit_to = emit_to;
  return plan;
}

template <typename T>
__global__ void dopri5_attempt_kernel(DopriArgs<T> g, int parity) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int H = (int)g.H, C = (int)g.C, NS = g.NS;
  T* zs = reinterpret_cast<T*>(smem_raw);
  T* dx = zs + NS * H;
  double* red = reinterpret_cast<double*>(smem_raw + (((size_t)(NS * (H + C)) * sizeof(T) + 15) / 16) * 16);   // 2 * blockDim doubles
  const int tid = threadIdx.x;
  const int s = tid / H, h = tid - s * H;
  const bool lane_on = s < NS;
  const int64_t n_tiles = (g.B + NS - 1) / NS;
  const int64_t BH = g.B * g.H;
  const int p = parity, q = parity ^ 1;
  DopriCtrl c = g.ctrl[p];
  if (c.phase == 4) {                                           // finished: keep the flag alive in both structs
    if (blockIdx.x == 0 && tid == 0) g.ctrl[q] = c;
    return;
  }
  T* Sp = g.state + (int64_t)p * 5 * BH;                        // what the previous launch produced
  T* Sq = g.state + (int64_t)q * 5 * BH;                        // what this launch produces
  const double* Pp = g.partial + (int64_t)p * g.n_blocks_alloc * 2;
  double* Pq = g.partial + (int64_t)q * g.n_blocks_alloc * 2;
  const T rtol = (T)g.rtol, atol = (T)g.atol;

  // ---- pending global sums of the previous launch (fixed order -> identical in every workgroup)
  double sum0 = 0.0, sum1 = 0.0;
  if (c.phase != 0 && g.ext_sums) {                               // one controller for all shards of the batch
    sum0 = g.ext_sums[0]; sum1 = g.ext_sums[1];
  } else if (c.phase != 0) {
    for (int64_t b = tid; b < (int64_t)gridDim.x; b += blockDim.x) { sum0 += Pp[2 * b]; sum1 += Pp[2 * b + 1]; }
    block_sum2(sum0, sum1, red);
  }

  DopriPlan<T> plan = dopri_controller<T>(g, c, sum0, sum1);
  const bool accept = plan.accept;
  const int mode = plan.mode;
  const double t0 = plan.t0, t1 = plan.t1, dt = plan.dt, dt_done = plan.dt_done;
  const T h0_state = plan.h0_state;
  const int64_t emit_from = plan.emit_from, emit_to = plan.emit_to;
  double acc0 = 0.0, acc1 = 0.0;
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t series = tile * NS + s;
    const bool valid = lane_on && series < g.B;
    const int64_t e = valid ? series * H + h : 0;
    T y = (T)0, k0 = (T)0;
    if (c.phase == 0) {
      y = valid ? g.z0[e] : (T)0;
      if (valid) g.z_out[(series * g.n_out) * H + h] = y;
    } else if (c.phase == 1 || c.phase == 2) {
      y = valid ? Sp[0 * BH + e] : (T)0;
      k0 = valid ? Sp[2 * BH + e] : (T)0;
    } else {
      // commit the pending attempt
      const T y0p = valid ? Sp[0 * BH + e] : (T)0, y1p = valid ? Sp[1 * BH + e] : (T)0;
      const T f0 = valid ? Sp[2 * BH + e] : (T)0, f1 = valid ? Sp[3 * BH + e] : (T)0;
      if (accept) {
        // dense output over [t_lo, t_hi] for every output time the step covered (oracle _fit_dense/_eval_dense)
        if (emit_to > emit_from) {
          const T dtf = (T)dt_done;
          const T ymid = valid ? Sp[4 * BH + e] : (T)0;
          const T ca = (T)2 * dtf * (f1 - f0) - (T)8 * (y1p + y0p) + (T)16 * ymid;
          const T cb = dtf * ((T)5 * f0 - (T)3 * f1) + (T)18 * y0p + (T)14 * y1p - (T)32 * ymid;
          const T cc = dtf * (f1 - (T)4 * f0) - (T)11 * y0p - (T)5 * y1p + (T)16 * ymid;
          const T cd = dtf * f0;
          for (int64_t io = emit_from; io < emit_to; ++io) {
            const T x = (T)((g.t_out[io] - c.t_lo) / (c.t_hi - c.t_lo));
            T total = y0p + x * cd;
            T xp = x;
            xp = xp * x; total = total + xp * cc;
            xp = xp * x; total = total + xp * cb;
            xp = xp * x; total = total + xp * ca;
            if (valid) g.z_out[(series * g.n_out + io) * H + h] = total;
          }
        }
        y = y1p; k0 = f1;
      } else {
        y = y0p; k0 = f0;
      }
    }
    if (mode == 3) continue;

    if (mode == 0) {
      const T ts = (T)c.t_hi;
      k0 = dopri_field(g, zs, dx, y, ts, tile, s, h, lane_on);
      const T scale = atol + (y < 0 ? -y : y) * rtol;
      if (valid) {
        const T a = y / scale, b = k0 / scale;
        acc0 += (double)(a * a); acc1 += (double)(b * b);
        Sq[0 * BH + e] = y; Sq[2 * BH + e] = k0;
      }
    } else if (mode == 1) {
      const T yy = y + h0_state * k0;
      const T ts = (T)(c.t_hi + (double)h0_state);              // t0 (float64) + h0, cast by the field wrapper
      const T f1 = dopri_field(g, zs, dx, yy, ts, tile, s, h, lane_on);
      const T scale = atol + (y < 0 ? -y : y) * rtol;
      if (valid) {
        const T a = (f1 - k0) / scale;
        acc0 += (double)(a * a);
        Sq[0 * BH + e] = y; Sq[2 * BH + e] = k0;
      }
    } else {
      const T t0f = (T)t0, dtf = (T)dt, t1f = (T)t1;
      if (c.refresh) k0 = dopri_field(g, zs, dx, y, next_toward(t0f, (T)1), tile, s, h, lane_on);   // just after the jump
      T kk[7];
      kk[0] = k0;
      T yi = y;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        T ti;
        if (i >= 4) ti = next_toward(t1f, (T)-1); else ti = t0f + (T)DP_ALPHA[i] * dtf;
        T inc = (T)0;
#pragma unroll
        for (int j = 0; j <= i; ++j) inc += kk[j] * ((T)DP_BETA[i][j] * dtf);
        yi = y + inc;
        kk[i + 1] = dopri_field(g, zs, dx, yi, ti, tile, s, h, lane_on);
      }
      const T y1 = yi;
      T err = (T)0, mid = (T)0;
#pragma unroll
      for (int j = 0; j < 7; ++j) { err += kk[j] * (dtf * (T)DP_CERR[j]); mid += kk[j] * (dtf * (T)DP_CMID[j]); }
      const T ay = y < 0 ? -y : y, ay1 = y1 < 0 ? -y1 : y1;
      const T tol = atol + rtol * (ay > ay1 ? ay : ay1);
      if (valid) {
        const T r = err / tol;
        acc0 += (double)(r * r);
        Sq[0 * BH + e] = y; Sq[1 * BH + e] = y1; Sq[2 * BH + e] = kk[0]; Sq[3 * BH + e] = kk[6];
        Sq[4 * BH + e] = y + mid;                                  // y_mid = y0 + k @ (dt * c_mid)
      }
    }
  }
  // ---- publish this launch's partial sums and the controller state for the next launch
  block_sum2(acc0, acc1, red);
  if (tid == 0) { Pq[2 * blockIdx.x] = acc0; Pq[2 * blockIdx.x + 1] = acc1; }
  if (blockIdx.x == 0 && tid == 0) {
    if (mode == 0) c.phase = 1;
    else if (mode == 1) c.phase = 2;
    else if (mode == 2) c.phase = 3;
    else c.phase = 4;
    g.ctrl[q] = c;
  }
}

constexpr int64_t DOPRI_MAX_LDS_KNOTS = 8192;
constexpr int64_t DOPRI_MAX_LDS_KNOTS_MLP = 1536;   // 160 KB LDS - 145.5 KB of weight images - 8 KB reduction scratch

// ------------------------------------------------------------------------------------------ MFMA attempt kernel
// f32, H = 32, C = 8, no activation: 16 series per wave on v_mfma_f32_16x16x4_f32 exactly like K2 (field16), six
// stage evaluations per launch.  Same controller, same state / partial layout as the generic kernel.
__device__ __forceinline__ f32x4 abs4(const f32x4& v) { return f32x4{fabsf(v[0]), fabsf(v[1]), fabsf(v[2]), fabsf(v[3])}; }
__device__ __forceinline__ f32x4 max4(const f32x4& a, const f32x4& b) {
  return f32x4{fmaxf(a[0], b[0]), fmaxf(a[1], b[1]), fmaxf(a[2], b[2]), fmaxf(a[3], b[3])};
}
__device__ __forceinline__ double sq4(const f32x4& v) {
  return (double)(v[0] * v[0]) + (double)(v[1] * v[1]) + (double)(v[2] * v[2]) + (double)(v[3] * v[3]);
}

#ifdef CDE_PHASE_TRACE
__device__ unsigned long long k4_phase_trace[TRACE_RING * TRACE_BLOCKS * TRACE_SLOTS];
#endif

// SPLIT (two-layer field, at most one tile per CU): the workgroup's 8 waves share ONE 16-series tile and split layer 2 of
// every evaluation by unit group (cde_mfma.h: field_mlp16<..., SPLIT>); all of them carry the state, wave 0 alone stores
// it, writes outputs and contributes to the error sums.
constexpr int DOPRI_XWIN_FLOATS = 8 * 64;
constexpr int64_t DOPRI_MLP_SPLIT_TILES = 768;   // two-layer field, 8-channel tiles: the eight waves of a workgroup share a tile (forward 8192 series: 105 -> 51 ms, 16384: tie)
// HI (round 6): the two-layer field with 17..32 hidden units on the 16-channel layout (cde_mfma.h: field_mlp16<.., HI>)
template <int DEGREE, int ACT, bool MLP = false, int CT = MC, bool SPLIT = false, bool HI = false>
__global__ __launch_bounds__(512, 2) void dopri5_attempt_mfma(DopriArgs<float> g, int parity) {
  static_assert(CT == MC || MLP, "16-channel tiles: two-layer fields only");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using T = float;
  const int tid = threadIdx.x;
  const int p = parity, q2 = parity ^ 1;
  CDE_STAMP_DECL;
  CDE_STAMP(0);
  // Everything a launch needs before it can decide the pending attempt is requested at once, ahead of the first wait:
  // the previous launch's partial sums (their address depends on the launch parity only), the controller block, the
  // weight image and the knots.  (profiles/r03_phase_k4.log: taken one after the other these dependent round trips were
  // 5.7 us of a 40 us attempt.)
  const double* Pp = g.partial + (int64_t)p * g.n_blocks_alloc * 2;
  double sums[2] = {0.0, 0.0};
  if (!g.ext_sums)
    for (int64_t b = tid; b < (int64_t)gridDim.x; b += blockDim.x) { sums[0] += Pp[2 * b]; sums[1] += Pp[2 * b + 1]; }
  DopriCtrl c = g.ctrl[p];
  if (c.phase == 4) {
    if (blockIdx.x == 0 && tid == 0) g.ctrl[q2] = c;
    return;
  }
  CDE_STAMP(1);
  // weight images: built once per solve (w16_image_kernel / wy16_image_kernel).  Product form: one coalesced
  // 16-byte load per group and lane into registers; activation form: copied to LDS (field_act16 reads it there).
  constexpr bool PRODUCT = ACT == CDE_ACT_NONE && !MLP;
  constexpr int STRIDE = PRODUCT ? 1 : 4;
  constexpr int IMG_FLOATS = MLP ? MLP16_LDS_FLOATS : ACT16_LDS_FLOATS;
  constexpr int64_t MAX_LDS_KNOTS = MLP ? DOPRI_MAX_LDS_KNOTS_MLP : DOPRI_MAX_LDS_KNOTS;
  float4 wA[PRODUCT ? W16_GROUPS : 1], wB[PRODUCT ? W16_GROUPS : 1];
  const Dims dims{(int)g.H, (int)g.C};
  const int Hr = dims.H;
  // two-layer field with more than 16 hidden units on the 16-channel layout: unit groups 4..7 from the raw output layer
  static_assert(!HI || (MLP && CT == 16), "the upper half: two-layer field, 16-channel layout");
  const MlpHi mlp_hi = HI ? MlpHi{(const float*)g.W, (const float*)g.bias, dims.H, dims.C, g.width} : MlpHi{};
  double* red = reinterpret_cast<double*>(lds);                    // 2 * 512 doubles
  // The knot search of every stage time is a chain of dependent loads: from global memory that is ~7 x 0.3 us per
  // stage (it dominated this kernel); the knots are copied to LDS once per launch instead.
  float* knots_lds = lds + 2 * 512 * 2;
  const bool knots_in_lds = g.n_intervals + 1 <= MAX_LDS_KNOTS;
  if (knots_in_lds) for (int64_t i = tid; i <= g.n_intervals; i += blockDim.x) knots_lds[i] = g.knots[i];
  const float* kn = knots_in_lds ? knots_lds : g.knots;
  float* img_lds = knots_lds + (knots_in_lds ? (g.n_intervals + 4) / 4 * 4 : 0);          // 16-byte aligned
  float* xwin = PRODUCT ? img_lds : img_lds + IMG_FLOATS;             // (SPLIT only: the exchange window behind the image)
  if constexpr (!PRODUCT) {
    const float4* src = reinterpret_cast<const float4*>(g.w16);
    float4* dst = reinterpret_cast<float4*>(img_lds);
    for (int i = tid; i < IMG_FLOATS / 4; i += blockDim.x) dst[i] = src[i];
  }
  const int64_t BH = g.B * g.H;
  // State: two (y, k) slots and one midpoint array.  Slot `c.slot` holds the start of the pending attempt (y0, k0 = f
  // at t0), the other slot what the attempt produced (y1, and k6 = f at t1 when that is ever read: a step clipped
  // onto a jump time re-evaluates f just after the jump instead).  Accepting a step swaps the roles, rejecting it
  // changes nothing, so an attempt moves 2 + 2 arrays of B*H floats through HBM (the round-1 layout rewrote all five
  // arrays every attempt: 38 MB and 8 us per attempt on the 32768-series shard, all of it after the last MFMA).
  float* const Ys[2] = {g.state, g.state + BH};
  float* const Ks[2] = {g.state + 2 * BH, g.state + 3 * BH};
  float* const Mid = g.state + 4 * BH;
  double* Pq = g.partial + (int64_t)q2 * g.n_blocks_alloc * 2;
  const T rtol = (T)g.rtol, atol = (T)g.atol;

  const int lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int64_t series = SPLIT ? (int64_t)blockIdx.x * 16 + n : ((int64_t)blockIdx.x * 8 + wave) * 16 + n;
  const bool in_range = series < g.B;
  const bool valid = in_range && (!SPLIT || wave == 0);               // who stores / counts (every wave LOADS its series)
  const int64_t sc = in_range ? series : g.B - 1;
  const int64_t e = sc * Hr;                                          // this series' row in the state arrays
  // this lane's 8 hidden units in two groups of 4: product form 8q..8q+7, activation form q, 4+q, .., 28+q
  const int u0 = PRODUCT ? 8 * q : q, u1 = PRODUCT ? 8 * q + 4 : 16 + q;

  // the state the next attempt starts from if the pending one is accepted (nearly always): requested before the
  // partial sums and the controller, whose latency then hides the loads
  const int phase_in = uni(c.phase), old_slot = uni(c.slot), cand = old_slot ^ 1, pending_stored = uni(c.stored);
  f32x4 ya, yb, k0a, k0b;
  if (phase_in == 3) {
    ya = load_units4<STRIDE>(Ys[cand] + e, u0, Hr); yb = load_units4<STRIDE>(Ys[cand] + e, u1, Hr);
    if (pending_stored & 1) { k0a = load_units4<STRIDE>(Ks[cand] + e, u0, Hr); k0b = load_units4<STRIDE>(Ks[cand] + e, u1, Hr); }
  }

  // The control rows of the two intervals the next attempt is likely to start in (the pending attempt's own start
  // interval if it is rejected or stays inside it, the one it ended in -- or the next, after a jump -- if it is
  // accepted) are TOUCHED now: one dword per lane, first and last of each row, pulls their lines from HBM into this
  // XCD's L2 while the sums are reduced and the controller runs.  The row load proper can only be issued once the
  // controller has decided; it used to be an HBM round trip in front of the first stage (profiles/r03_phase_k4.log:
  // stage 1 took 8.0 us, the others 3.5), now it hits the L2.  (Holding both candidate rows in registers instead
  // spilled: the weight image already fills the register file.)
  float touched = 0.f;
  if (phase_in == 3) {
    const int hint_a = uni(c.hint_lo), hint_b = uni(c.hint_hi == c.hint_lo ? c.hint_lo + 1 : c.hint_hi);
    const int cand = q < 2 ? hint_a : hint_b;
    if (cand >= 0 && cand < g.n_intervals) {
      constexpr int PARTS = DEGREE == CDE_PATH_CUBIC ? 3 : 2;
      const float* rowp = DEGREE == CDE_PATH_CUBIC ? g.coeffs + ((sc * g.n_intervals + cand) * 4 + 1) * dims.C
                                                   : g.coeffs + (sc * (g.n_intervals + 1) + cand) * dims.C;
      touched = *reinterpret_cast<const volatile float*>(rowp + ((q & 1) ? PARTS * dims.C - 1 : 0));
    }
  }

  // pending sums of the whole batch: fixed order (xor tree in each wave, then the waves in index order), so the decision
  // is identical in every workgroup and run to run.  One barrier serves the knots, the weight image and the sums.
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sums[k] += __shfl_xor(sums[k], off, 64);
  if (lane == 0) { red[wave] = sums[0]; red[8 + wave] = sums[1]; }
  __syncthreads();
  CDE_STAMP(2);
  // (the register image of the weights is requested here, not at entry: it fills the register file, and its L2 latency
  // hides behind the controller just as well)
  float4 sg0, sg1, sh0, sh1, sba, sbb;                               // (PRODUCT && SPLIT: this wave's groups only)
  if constexpr (PRODUCT && SPLIT) {
    const float4* img = reinterpret_cast<const float4*>(g.w16) + (tid & 63);
    const int pw_ = uni((int)(tid >> 6));
    sg0 = img[(2 * pw_) * 64]; sg1 = img[(2 * pw_ + 1) * 64];
    sh0 = img[(W16_GROUPS + 2 * pw_) * 64]; sh1 = img[(W16_GROUPS + 2 * pw_ + 1) * 64];
    sba = img[16 * 64]; sbb = img[(W16_GROUPS + 16) * 64];
  } else if constexpr (PRODUCT) {
    const float4* img = reinterpret_cast<const float4*>(g.w16) + (tid & 63);
#pragma unroll
    for (int grp = 0; grp < W16_GROUPS; ++grp) { wA[grp] = img[grp * 64]; wB[grp] = img[(W16_GROUPS + grp) * 64]; }
  }
  if (phase_in != 0 && g.ext_sums) {                              // one controller for all shards of the batch
    sums[0] = g.ext_sums[0]; sums[1] = g.ext_sums[1];
  } else {
    sums[0] = sums[1] = 0.0;
#pragma unroll
    for (int w8 = 0; w8 < 8; ++w8) { sums[0] += red[w8]; sums[1] += red[8 + w8]; }
  }
  // (two-layer field, eight waves per tile: the 8 KB of `red` are the evaluations' u window from here on -- nobody may
  //  still be reading the block sums)
  if constexpr (MLP && SPLIT && CT == MC) __syncthreads();
  CDE_STAMP(3);
  const DopriPlan<T> plan = dopri_controller<T>(g, c, sums[0], sums[1]);
  // The controller's outputs derive from LDS reads (the block sums), so the compiler would keep them -- and every
  // stage time, interval index and slot pointer computed from them -- in vector registers; read back through lane 0
  // they are scalars.
  const int mode = uni(plan.mode);
  const bool accepted = uni((int)plan.accept) != 0;
  const int slot = (phase_in == 3 && accepted) ? cand : old_slot;      // where the next attempt starts from
  const int other = slot ^ 1;
  const int64_t emit_from = uni(plan.emit_from), emit_to = uni(plan.emit_to);
  const bool refresh = uni((int)c.refresh) != 0;
  const T plan_t0 = uni((T)plan.t0), plan_dt = uni((T)plan.dt), plan_t1 = uni((T)plan.t1), h0f = uni(plan.h0_state);
  const T dt_done = uni((T)plan.dt_done);
  const double t_lo = uni(c.t_lo), t_hi = uni(c.t_hi);
  const bool will_emit = mode == 2 && c.i_out < g.n_out && !(g.t_out[c.i_out] > plan.t1);
  const int stored = uni(mode == 2 ? (((!c.on_jump || will_emit) ? 1 : 0) | (will_emit ? 2 : 0)) : 0);
  if (blockIdx.x == 0 && tid == 0) {                               // the controller block for the next launch
    c.phase = mode == 0 ? 1 : mode == 1 ? 2 : mode == 2 ? 3 : 4;
    c.slot = slot; c.stored = stored;
    g.ctrl[q2] = c;
  }
#ifdef CDE_PHASE_TRACE
  const int attempt_no = uni((int)(c.n_accept + c.n_reject));
#endif
  CDE_STAMP(4);
  const float4* wy = reinterpret_cast<const float4*>(img_lds) + lane;
  const float4* by = reinterpret_cast<const float4*>(img_lds + WY_FLOATS) + q;
  auto field = [&](const f32x4& za, const f32x4& zb, const float (&dXv)[CT], f32x4& fa, f32x4& fb) {
    if constexpr (MLP) field_mlp16<ACT, CT, SPLIT, HI>(img_lds, lane, q, za, zb, dXv, fa, fb, wave, xwin, reinterpret_cast<float*>(red), mlp_hi);
    else if constexpr (CT == MC) {
      if constexpr (PRODUCT && SPLIT) field16_split(sg0, sg1, sh0, sh1, sba, sbb, za, zb, dXv, q, fa, fb, wave, xwin, lane);
      else if constexpr (PRODUCT) field16(wA, wB, za, zb, dXv, q, fa, fb);
      else if constexpr (SPLIT) field_act16_split<ACT>(wy, by, za, zb, dXv, fa, fb, wave, xwin, lane);
      else field_act16<ACT>(wy, by, za, zb, dXv, fa, fb);
    }
  };

  // control derivative at a (wave-uniform) time; the row is re-fetched only when the interval changes
  int64_t row_idx = -1, first_idx = -1;
  Row<DEGREE, CT> row;
  float dX_lin[CT];                                   // piecewise-linear control: the slope of the interval in use
  asm volatile("" ::"v"(touched));                    // (the touch loads are waited for here at the latest)
  auto slope_at = [&](T ts, float (&dX)[CT]) {
    T frac;
    // the stages of an attempt almost always share their interval (always, with jump_t on the knots): two comparisons
    // against the interval in use instead of the 8 dependent LDS reads of the search (6.5 us per attempt)
    const int64_t idx = locate_near(kn, g.n_intervals, ts, row_idx, frac);
    if (idx != row_idx) {
      row = load_row<DEGREE, CT>(g.coeffs, sc, g.n_intervals, idx, dims.C);
      if (first_idx < 0) first_idx = idx;
      row_idx = idx;
      // the 8 IEEE divisions of a linear slope once per interval, not once per stage (with jump_t on the knots all
      // stages of an attempt share the interval: the divisions were a fifth of this kernel's VALU instructions)
      if (DEGREE == CDE_PATH_LINEAR) control_slope<DEGREE, CT>(row, frac, kn[idx + 1] - kn[idx], dX_lin);
    }
    if (DEGREE == CDE_PATH_LINEAR) {
#pragma unroll
      for (int cc = 0; cc < CT; ++cc) dX[cc] = dX_lin[cc];
    } else {
      control_slope<DEGREE, CT>(row, frac, 1.f, dX);
    }
  };

  double acc[2] = {0.0, 0.0};
  if (phase_in == 0) {
    ya = load_units4<STRIDE>(g.z0 + e, u0, Hr); yb = load_units4<STRIDE>(g.z0 + e, u1, Hr);
    if (valid) { store_units4<STRIDE>(g.z_out + (series * g.n_out) * Hr, u0, Hr, ya); store_units4<STRIDE>(g.z_out + (series * g.n_out) * Hr, u1, Hr, yb); }
    k0a = k0b = f32x4{0.f, 0.f, 0.f, 0.f};
  } else if (phase_in == 1 || phase_in == 2) {
    ya = load_units4<STRIDE>(Ys[slot] + e, u0, Hr); yb = load_units4<STRIDE>(Ys[slot] + e, u1, Hr);
    k0a = load_units4<STRIDE>(Ks[slot] + e, u0, Hr); k0b = load_units4<STRIDE>(Ks[slot] + e, u1, Hr);
  } else if (accepted) {
    // (ya, yb) = y1 and -- unless f is re-evaluated after a jump -- (k0a, k0b) = k6 of the accepted step are in flight
    if (emit_to > emit_from) {
      // outputs covered by the accepted step: 4th-order dense interpolant (oracle _fit_dense / _eval_dense); such an
      // attempt always stores k6 and the midpoint
      const T dtf = dt_done;
      const f32x4 y0a = load_units4<STRIDE>(Ys[old_slot] + e, u0, Hr), y0b = load_units4<STRIDE>(Ys[old_slot] + e, u1, Hr);
      const f32x4 f0a = load_units4<STRIDE>(Ks[old_slot] + e, u0, Hr), f0b = load_units4<STRIDE>(Ks[old_slot] + e, u1, Hr);
      const f32x4 ma = load_units4<STRIDE>(Mid + e, u0, Hr), mb = load_units4<STRIDE>(Mid + e, u1, Hr);
      const f32x4 y1a = ya, y1b = yb, f1a = k0a, f1b = k0b;
      const f32x4 caa = 2.f * dtf * (f1a - f0a) - 8.f * (y1a + y0a) + 16.f * ma;
      const f32x4 cab = 2.f * dtf * (f1b - f0b) - 8.f * (y1b + y0b) + 16.f * mb;
      const f32x4 cba = dtf * (5.f * f0a - 3.f * f1a) + 18.f * y0a + 14.f * y1a - 32.f * ma;
      const f32x4 cbb = dtf * (5.f * f0b - 3.f * f1b) + 18.f * y0b + 14.f * y1b - 32.f * mb;
      const f32x4 cca = dtf * (f1a - 4.f * f0a) - 11.f * y0a - 5.f * y1a + 16.f * ma;
      const f32x4 ccb = dtf * (f1b - 4.f * f0b) - 11.f * y0b - 5.f * y1b + 16.f * mb;
      const f32x4 cda = dtf * f0a, cdb = dtf * f0b;
      for (int64_t io = emit_from; io < emit_to; ++io) {
        const T x = (T)((g.t_out[io] - t_lo) / (t_hi - t_lo));
        f32x4 ta = y0a + x * cda, tb = y0b + x * cdb;
        T xp = x;
        xp = xp * x; ta = ta + xp * cca; tb = tb + xp * ccb;
        xp = xp * x; ta = ta + xp * cba; tb = tb + xp * cbb;
        xp = xp * x; ta = ta + xp * caa; tb = tb + xp * cab;
        if (valid) { store_units4<STRIDE>(g.z_out + (series * g.n_out + io) * Hr, u0, Hr, ta); store_units4<STRIDE>(g.z_out + (series * g.n_out + io) * Hr, u1, Hr, tb); }
      }
    }
  } else {                                                            // rejected: back to the start of that attempt
    ya = load_units4<STRIDE>(Ys[slot] + e, u0, Hr); yb = load_units4<STRIDE>(Ys[slot] + e, u1, Hr);
    k0a = load_units4<STRIDE>(Ks[slot] + e, u0, Hr); k0b = load_units4<STRIDE>(Ks[slot] + e, u1, Hr);
  }

  float dX[CT];
  CDE_STAMP(5);
  if (mode == 0) {
    slope_at((T)t_hi, dX);
    field(ya, yb, dX, k0a, k0b);
    const f32x4 sa = atol + abs4(ya) * rtol, sb = atol + abs4(yb) * rtol;
    if (valid) {
      acc[0] = sq4(ya / sa) + sq4(yb / sb);
      acc[1] = sq4(k0a / sa) + sq4(k0b / sb);
      store_units4<STRIDE>(Ys[slot] + e, u0, Hr, ya); store_units4<STRIDE>(Ys[slot] + e, u1, Hr, yb);
      store_units4<STRIDE>(Ks[slot] + e, u0, Hr, k0a); store_units4<STRIDE>(Ks[slot] + e, u1, Hr, k0b);
    }
  } else if (mode == 1) {
    const T h0 = h0f;
    const f32x4 za = ya + h0 * k0a, zb = yb + h0 * k0b;
    slope_at((T)(t_hi + (double)h0), dX);
    f32x4 f1a, f1b;
    field(za, zb, dX, f1a, f1b);
    const f32x4 sa = atol + abs4(ya) * rtol, sb = atol + abs4(yb) * rtol;
    if (valid) acc[0] = sq4((f1a - k0a) / sa) + sq4((f1b - k0b) / sb);
  } else if (mode == 2) {
    const T t0f = plan_t0, dtf = plan_dt, t1f = plan_t1;
    if (refresh) {                                                    // just after the jump we landed on
      slope_at(next_toward(t0f, 1.f), dX);
      field(ya, yb, dX, k0a, k0b);
      if (valid) {                                                    // a rejected attempt restarts from this k0
        store_units4<STRIDE>(Ks[slot] + e, u0, Hr, k0a); store_units4<STRIDE>(Ks[slot] + e, u1, Hr, k0b);
      }
    }
    // (`stored`: will the step, if accepted, cover an output time?  Only then are k6 -- when the step ends on a jump --
    // and the midpoint ever read)
    f32x4 ka[7], kb[7];
    ka[0] = k0a; kb[0] = k0b;
    f32x4 zia = ya, zib = yb;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const T ti = i >= 4 ? next_toward(t1f, -1.f) : t0f + (T)DP_ALPHA[i] * dtf;
      f32x4 ia = {0.f, 0.f, 0.f, 0.f}, ib = ia;
#pragma unroll
      for (int j = 0; j <= i; ++j) {                      // torchdiffeq forms this sum inside a matmul: fused
        const T w = (T)DP_BETA[i][j] * dtf;               // multiply-adds are as faithful as separate roundings
        if (DP_BETA[i][j] == 0.0) continue;
        const f32x4 wv = {w, w, w, w};
        ia = __builtin_elementwise_fma(ka[j], wv, ia); ib = __builtin_elementwise_fma(kb[j], wv, ib);
      }
      zia = ya + ia; zib = yb + ib;
      slope_at(ti, dX);
      field(zia, zib, dX, ka[i + 1], kb[i + 1]);
      CDE_STAMP(6 + i);
    }
    f32x4 ea = {0.f, 0.f, 0.f, 0.f}, eb = ea;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const T we = dtf * (T)DP_CERR[j];
      if (DP_CERR[j] == 0.0) continue;                    // c_err[1] == c_mid[1] == 0
      const f32x4 wev = {we, we, we, we};
      ea = __builtin_elementwise_fma(ka[j], wev, ea); eb = __builtin_elementwise_fma(kb[j], wev, eb);
    }
    const f32x4 ta = atol + rtol * max4(abs4(ya), abs4(zia)), tb = atol + rtol * max4(abs4(yb), abs4(zib));
    if (valid) {
      acc[0] = sq4(ea / ta) + sq4(eb / tb);
      store_units4<STRIDE>(Ys[other] + e, u0, Hr, zia); store_units4<STRIDE>(Ys[other] + e, u1, Hr, zib);
      if (stored & 1) { store_units4<STRIDE>(Ks[other] + e, u0, Hr, ka[6]); store_units4<STRIDE>(Ks[other] + e, u1, Hr, kb[6]); }
    }
    if (stored & 2) {
      f32x4 ma = {0.f, 0.f, 0.f, 0.f}, mb = ma;
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const T wm = dtf * (T)DP_CMID[j];
        if (DP_CMID[j] == 0.0) continue;
        const f32x4 wmv = {wm, wm, wm, wm};
        ma = __builtin_elementwise_fma(ka[j], wmv, ma); mb = __builtin_elementwise_fma(kb[j], wmv, mb);
      }
      if (valid) { store_units4<STRIDE>(Mid + e, u0, Hr, ya + ma); store_units4<STRIDE>(Mid + e, u1, Hr, yb + mb); }
    }
  }
  CDE_STAMP(12);
  block_total<2>(acc, red);
  CDE_STAMP(13);
  if (tid == 0) { Pq[2 * blockIdx.x] = acc[0]; Pq[2 * blockIdx.x + 1] = acc[1]; }
  if (blockIdx.x == 0 && tid == 0) {                               // (the rest of the block was written before the stages)
    g.ctrl[q2].hint_lo = (int32_t)first_idx; g.ctrl[q2].hint_hi = (int32_t)row_idx;
  }
  CDE_STAMP_FLUSH(k4_phase_trace, attempt_no);
}

// ------------------------------------------------------------------------------------------ wide attempt kernel
// K4 for the shapes of rk4_wide.hip (one-layer field, f32, H <= 64 and C <= 8, or H <= 32 and C <= 16): a workgroup of
// NW waves per 16-series tile, wave w owns hidden units 8w..8w+7 and lane (n = l & 15, q = l >> 4) units ua = 8w + q,
// ub = ua + 4 of series n -- their state and their seven slopes live in that lane; Y = W z + b from a register image
// (v_mfma_f32_16x16x4_f32), the stage state crosses the waves through LDS once per evaluation.  A bounded grid of
// workgroups walks the tiles; the control derivative of a tile at all stage times and the tile's state are requested
// one tile ahead.  Same controller, same two state slots, same partial sums as dopri5_attempt_mfma.
constexpr int64_t DOPRI_MAX_LDS_KNOTS_WIDE = 8192;
constexpr int DOPRI_WIDE_RED_FLOATS = 128;               // 2 * (waves) doubles of reduction scratch, padded

template <int NW, int NB>
constexpr size_t dopri_wide_lds_bytes(int64_t n_knots) {
  using G = Wide<NW, NB>;
  return (size_t)(DOPRI_WIDE_RED_FLOATS + 2 * G::ZBUF + 2 * 7 * G::DX + G::GC) * sizeof(float) +
         (n_knots <= DOPRI_MAX_LDS_KNOTS_WIDE ? (size_t)((n_knots + 3) / 4 * 4) * sizeof(float) : 0);
}

template <int DEGREE, int ACT, int NW, int NB>
__global__ __launch_bounds__(64 * NW, 2) void dopri5_attempt_wide(DopriArgs<float> g, int parity) {
  using G = Wide<NW, NB>;
  using T = float;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int p = parity, q2 = parity ^ 1;
  DopriCtrl c = g.ctrl[p];
  if (c.phase == 4) {
    if (blockIdx.x == 0 && tid == 0) g.ctrl[q2] = c;
    return;
  }
  const int Hr = (int)g.H, Cr = (int)g.C;
  const int lane = tid & 63, w = tid >> 6;
  const int n = lane & 15, q = lane >> 4;
  double* red = reinterpret_cast<double*>(lds);
  float* zbuf = lds + DOPRI_WIDE_RED_FLOATS;
  float* dxb = zbuf + 2 * G::ZBUF;                                   // [tile parity][evaluation][series][channel]
  float* bias_lds = dxb + 2 * 7 * G::DX;                             // zero-padded [unit][CT]: the C/D rows of a lane
  float* knots_lds = bias_lds + G::GC;
  for (int e = tid; e < G::GC; e += blockDim.x) {
    const int h = e / G::CT, cc = e % G::CT;
    bias_lds[e] = (h < Hr && cc < Cr) ? g.bias[h * Cr + cc] : 0.f;
  }
  const bool knots_in_lds = g.n_intervals + 1 <= DOPRI_MAX_LDS_KNOTS_WIDE;
  if (knots_in_lds) for (int64_t i = tid; i <= g.n_intervals; i += blockDim.x) knots_lds[i] = g.knots[i];
  const float* kn = knots_in_lds ? knots_lds : g.knots;

  // Y image of this wave: tile T = NB*P + tb, row i <-> (h = 8w + 4P + (i >> 2), c = 4 tb + (i & 3)); K step s: unit 4s + q
  float wy[G::NT][G::KS];
#pragma unroll
  for (int Tt = 0; Tt < G::NT; ++Tt) {
    const int P = Tt / NB, tb = Tt % NB;
    const int hA = 8 * w + 4 * P + (n >> 2), cA = 4 * tb + (n & 3);
#pragma unroll
    for (int s = 0; s < G::KS; ++s) {
      const int k = 4 * s + q;
      wy[Tt][s] = (hA < Hr && cA < Cr && k < Hr) ? g.W[(hA * Cr + cA) * Hr + k] : 0.f;
    }
  }
  __syncthreads();

  const int64_t BH = g.B * g.H;
  float* const Ys[2] = {g.state, g.state + BH};
  float* const Ks[2] = {g.state + 2 * BH, g.state + 3 * BH};
  float* const Mid = g.state + 4 * BH;
  const double* Pp = g.partial + (int64_t)p * g.n_blocks_alloc * 2;
  double* Pq = g.partial + (int64_t)q2 * g.n_blocks_alloc * 2;
  const T rtol = (T)g.rtol, atol = (T)g.atol;

  double sums[2] = {0.0, 0.0};
  if (c.phase != 0 && g.ext_sums) {
    sums[0] = g.ext_sums[0]; sums[1] = g.ext_sums[1];
  } else if (c.phase != 0) {
    for (int64_t b = tid; b < (int64_t)gridDim.x; b += blockDim.x) { sums[0] += Pp[2 * b]; sums[1] += Pp[2 * b + 1]; }
    block_total<2>(sums, red);
  }
  const int phase_in = uni(c.phase), old_slot = uni(c.slot), pending_stored = uni(c.stored);
  const DopriPlan<T> plan = dopri_controller<T>(g, c, sums[0], sums[1]);
  const int mode = uni(plan.mode);
  const bool accepted = uni((int)plan.accept) != 0;
  const bool commit = phase_in == 3 && accepted;
  const int slot = commit ? (old_slot ^ 1) : old_slot;               // where the next attempt starts from
  const int other = slot ^ 1;
  const int64_t emit_from = uni(plan.emit_from), emit_to = uni(plan.emit_to);
  const bool emits = commit && emit_to > emit_from;
  const bool refresh = uni((int)c.refresh) != 0;
  const T t0f = uni((T)plan.t0), dtf = uni((T)plan.dt), t1f = uni((T)plan.t1), h0f = uni(plan.h0_state);
  const T dt_done = uni((T)plan.dt_done);
  const double t_lo = uni(c.t_lo), t_hi = uni(c.t_hi);
  const bool will_emit = mode == 2 && c.i_out < g.n_out && !(g.t_out[c.i_out] > plan.t1);
  const int stored = uni(mode == 2 ? (((!c.on_jump || will_emit) ? 1 : 0) | (will_emit ? 2 : 0)) : 0);
  // the controller block for the next launch: everything in it is known now, and `c` need not stay live
  if (blockIdx.x == 0 && tid == 0) {
    c.phase = mode == 0 ? 1 : mode == 1 ? 2 : mode == 2 ? 3 : 4;
    c.slot = slot; c.stored = stored;
    g.ctrl[q2] = c;
  }

  // ---- evaluation times of this launch (wave-uniform): e = 0 the refresh / Hairer evaluation, e = 1..6 the stages
  const bool use0 = mode == 0 || mode == 1 || (mode == 2 && refresh);
  const int n_eval = mode == 2 ? 7 : (mode == 3 ? 0 : 1);
  int eidx[7];
  float efrac[7];
  {
    int64_t hint = -1;
#pragma unroll
    for (int e = 0; e < 7; ++e) {
      T ts;
      if (e == 0) ts = mode == 0 ? (T)t_hi : mode == 1 ? (T)(t_hi + (double)h0f) : next_toward(t0f, 1.f);
      else ts = e - 1 >= 4 ? next_toward(t1f, -1.f) : t0f + (T)DP_ALPHA[e - 1] * dtf;
      T frac = 0.f;
      int64_t idx = 0;
      if (e < n_eval && (e > 0 || use0)) { idx = locate_near(kn, g.n_intervals, ts, hint, frac); hint = idx; }
      eidx[e] = uni((int)idx); efrac[e] = uni(frac);
    }
  }

  const int ua = 8 * w + q, ub = ua + 4;
  const bool has_a = ua < Hr, has_b = ub < Hr;
  const int fc = G::CPW * w + (q % G::CPW);                          // the control channel this lane feeds
  const bool feeds = q < G::CPW;
  const int fcc = fc < Cr ? fc : Cr - 1;
  const int64_t n_tiles = (g.B + 15) / 16;
  float* zw = zbuf + n * G::ZROW + q * G::KS + 2 * w;
  const float* zr = zbuf + n * G::ZROW + q * G::KS;
  int par = 0, dbuf = 0;

  // control derivative of a tile at every evaluation time: requested one tile ahead, finished into dxb when that tile is next
  float raw[7][3];
  auto feed_request = [&](int64_t tile) {
    const int64_t series = tile * 16 + n;
    const int64_t sc = series < g.B ? series : g.B - 1;
#pragma unroll
    for (int e = 0; e < 7; ++e) {
      if (e < n_eval && (e > 0 || use0)) {
        if (DEGREE == CDE_PATH_CUBIC) {
          const float* pr = g.coeffs + ((sc * g.n_intervals + eidx[e]) * 4 + 1) * Cr + fcc;
          raw[e][0] = pr[0]; raw[e][1] = pr[Cr]; raw[e][2] = pr[2 * Cr];
        } else {
          const float* pr = g.coeffs + (sc * (g.n_intervals + 1) + eidx[e]) * Cr + fcc;
          raw[e][0] = pr[0]; raw[e][1] = pr[Cr]; raw[e][2] = kn[eidx[e] + 1] - kn[eidx[e]];
        }
      }
    }
  };
  auto feed_store = [&](int buf) {
#pragma unroll
    for (int e = 0; e < 7; ++e) {
      if (e < n_eval && (e > 0 || use0)) {
        float v = DEGREE == CDE_PATH_CUBIC ? cubic_derivative(raw[e][0], raw[e][1], raw[e][2], efrac[e])
                                           : (raw[e][1] - raw[e][0]) / raw[e][2];
        if (fc >= Cr) v = 0.f;
        if (feeds) dxb[(buf * 7 + e) * G::DX + n * G::DXROW + fc] = v;
      }
    }
  };
  // the state a tile starts from (its lane's two units), requested one tile ahead
  const bool k_known = phase_in == 1 || phase_in == 2 || (phase_in == 3 && (!accepted || (pending_stored & 1)));
  auto state_request = [&](int64_t tile, float (&st)[4]) {
    const int64_t series = tile * 16 + n;
    const int64_t sc = series < g.B ? series : g.B - 1;
    const int64_t ea = sc * Hr + (has_a ? ua : 0), eb = sc * Hr + (has_b ? ub : 0);
    if (phase_in == 0) { st[0] = g.z0[ea]; st[1] = g.z0[eb]; st[2] = 0.f; st[3] = 0.f; }
    else {
      st[0] = Ys[slot][ea]; st[1] = Ys[slot][eb];
      if (k_known) { st[2] = Ks[slot][ea]; st[3] = Ks[slot][eb]; } else { st[2] = 0.f; st[3] = 0.f; }
    }
  };

  // one evaluation of the vector field for the tile: publish the lane's two units, read everybody's, Y tiles, contraction
  auto evaluate = [&](float za, float zb, int e, float& fa, float& fb) {
    *reinterpret_cast<float2*>(zw + par * G::ZBUF) = make_float2(za, zb);
    spl_barrier();
    float4 z4[G::KS / 4], d4[NB];
#pragma unroll
    for (int i = 0; i < G::KS / 4; ++i) z4[i] = *reinterpret_cast<const float4*>(zr + par * G::ZBUF + 4 * i);
#pragma unroll
    for (int tb = 0; tb < NB; ++tb)
      d4[tb] = *reinterpret_cast<const float4*>(dxb + (dbuf * 7 + e) * G::DX + n * G::DXROW + 4 * tb);
    f32x4 y[G::NT];                                                  // bias rows of the lane's two units (LDS)
#pragma unroll
    for (int Tt = 0; Tt < G::NT; ++Tt) {
      const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + ((Tt / NB) ? ub : ua) * G::CT + 4 * (Tt % NB));
      y[Tt] = f32x4{b4.x, b4.y, b4.z, b4.w};
    }
#pragma unroll
    for (int i = 0; i < G::KS / 4; ++i) {
      const float zs[4] = {z4[i].x, z4[i].y, z4[i].z, z4[i].w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int Tt = 0; Tt < G::NT; ++Tt) y[Tt] = mfma16(wy[Tt][4 * i + j], zs[j], y[Tt]);
    }
    f32x2 fpa = {0.f, 0.f}, fpb = {0.f, 0.f};
#pragma unroll
    for (int tb = 0; tb < NB; ++tb) {
      const f32x2 d01 = {d4[tb].x, d4[tb].y}, d23 = {d4[tb].z, d4[tb].w};
      fpa = __builtin_elementwise_fma(activate2<ACT>(y[tb][0], y[tb][1]), d01, fpa);
      fpb = __builtin_elementwise_fma(activate2<ACT>(y[NB + tb][0], y[NB + tb][1]), d01, fpb);
      fpa = __builtin_elementwise_fma(activate2<ACT>(y[tb][2], y[tb][3]), d23, fpa);
      fpb = __builtin_elementwise_fma(activate2<ACT>(y[NB + tb][2], y[NB + tb][3]), d23, fpb);
    }
    fa = fpa[0] + fpa[1]; fb = fpb[0] + fpb[1];
    par ^= 1;
  };

  double acc[2] = {0.0, 0.0};
  float st_next[4] = {0.f, 0.f, 0.f, 0.f};
  if ((int64_t)blockIdx.x < n_tiles) {
    state_request(blockIdx.x, st_next);
    if (mode != 3) { feed_request(blockIdx.x); feed_store(0); }
  }
  for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t series = tile * 16 + n;
    const bool valid = series < g.B;
    const int64_t sc = valid ? series : g.B - 1;
    const bool ona = valid && has_a, onb = valid && has_b;
    const int64_t ea = sc * Hr + (has_a ? ua : 0), eb = sc * Hr + (has_b ? ub : 0);
    float ya = has_a ? st_next[0] : 0.f, yb = has_b ? st_next[1] : 0.f;
    float k0a = has_a ? st_next[2] : 0.f, k0b = has_b ? st_next[3] : 0.f;
    const bool has_next = tile + gridDim.x < n_tiles;
    if (has_next) { state_request(tile + gridDim.x, st_next); if (mode != 3) feed_request(tile + gridDim.x); }
    if (phase_in == 0) {
      if (ona) g.z_out[(series * g.n_out) * Hr + ua] = ya;
      if (onb) g.z_out[(series * g.n_out) * Hr + ub] = yb;
    }
    if (emits) {
      // outputs covered by the accepted step: 4th-order dense interpolant (oracle _fit_dense / _eval_dense); (ya, yb) is
      // y1 and (k0a, k0b) the stored k6 of that step (an attempt that covers an output always stores k6 and the midpoint)
      const T dtd = dt_done;
      const int old = slot ^ 1;
      const float y0a = has_a ? Ys[old][ea] : 0.f, y0b = has_b ? Ys[old][eb] : 0.f;
      const float f0a = has_a ? Ks[old][ea] : 0.f, f0b = has_b ? Ks[old][eb] : 0.f;
      const float ma = has_a ? Mid[ea] : 0.f, mb = has_b ? Mid[eb] : 0.f;
      const float y1a = ya, y1b = yb, f1a = k0a, f1b = k0b;
      const float caa = 2.f * dtd * (f1a - f0a) - 8.f * (y1a + y0a) + 16.f * ma;
      const float cab = 2.f * dtd * (f1b - f0b) - 8.f * (y1b + y0b) + 16.f * mb;
      const float cba = dtd * (5.f * f0a - 3.f * f1a) + 18.f * y0a + 14.f * y1a - 32.f * ma;
      const float cbb = dtd * (5.f * f0b - 3.f * f1b) + 18.f * y0b + 14.f * y1b - 32.f * mb;
      const float cca = dtd * (f1a - 4.f * f0a) - 11.f * y0a - 5.f * y1a + 16.f * ma;
      const float ccb = dtd * (f1b - 4.f * f0b) - 11.f * y0b - 5.f * y1b + 16.f * mb;
      const float cda = dtd * f0a, cdb = dtd * f0b;
      for (int64_t io = emit_from; io < emit_to; ++io) {
        const T x = (T)((g.t_out[io] - t_lo) / (t_hi - t_lo));
        float ta = y0a + x * cda, tb = y0b + x * cdb;
        T xp = x;
        xp = xp * x; ta = ta + xp * cca; tb = tb + xp * ccb;
        xp = xp * x; ta = ta + xp * cba; tb = tb + xp * cbb;
        xp = xp * x; ta = ta + xp * caa; tb = tb + xp * cab;
        if (ona) g.z_out[(series * g.n_out + io) * Hr + ua] = ta;
        if (onb) g.z_out[(series * g.n_out + io) * Hr + ub] = tb;
      }
    }
    if (mode == 3) continue;

    if (mode == 0) {
      evaluate(ya, yb, 0, k0a, k0b);
      const float sa = atol + fabsf(ya) * rtol, sb = atol + fabsf(yb) * rtol;
      if (ona) { const float u = ya / sa, v = k0a / sa; acc[0] += (double)(u * u); acc[1] += (double)(v * v); Ys[slot][ea] = ya; Ks[slot][ea] = k0a; }
      if (onb) { const float u = yb / sb, v = k0b / sb; acc[0] += (double)(u * u); acc[1] += (double)(v * v); Ys[slot][eb] = yb; Ks[slot][eb] = k0b; }
    } else if (mode == 1) {
      const T h0 = h0f;
      float f1a, f1b;
      evaluate(ya + h0 * k0a, yb + h0 * k0b, 0, f1a, f1b);
      const float sa = atol + fabsf(ya) * rtol, sb = atol + fabsf(yb) * rtol;
      if (ona) { const float u = (f1a - k0a) / sa; acc[0] += (double)(u * u); }
      if (onb) { const float u = (f1b - k0b) / sb; acc[0] += (double)(u * u); }
    } else {
      if (refresh) {                                                   // just after the jump we landed on
        evaluate(ya, yb, 0, k0a, k0b);
        if (ona) Ks[slot][ea] = k0a;                                   // a rejected attempt restarts from this k0
        if (onb) Ks[slot][eb] = k0b;
      }
      float ka[7], kb[7];
      ka[0] = k0a; kb[0] = k0b;
      float zia = ya, zib = yb;
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        float ia = 0.f, ib = 0.f;
#pragma unroll
        for (int j = 0; j <= i; ++j) {                    // torchdiffeq forms this sum inside a matmul: fused
          if (DP_BETA[i][j] == 0.0) continue;             // multiply-adds are as faithful as separate roundings
          const T wgt = (T)DP_BETA[i][j] * dtf;
          ia = __builtin_fmaf(ka[j], wgt, ia); ib = __builtin_fmaf(kb[j], wgt, ib);
        }
        zia = ya + ia; zib = yb + ib;
        evaluate(zia, zib, i + 1, ka[i + 1], kb[i + 1]);
      }
      float era = 0.f, erb = 0.f;
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        if (DP_CERR[j] == 0.0) continue;
        const T we = dtf * (T)DP_CERR[j];
        era = __builtin_fmaf(ka[j], we, era); erb = __builtin_fmaf(kb[j], we, erb);
      }
      const float ta = atol + rtol * fmaxf(fabsf(ya), fabsf(zia)), tb = atol + rtol * fmaxf(fabsf(yb), fabsf(zib));
      if (ona) { const float u = era / ta; acc[0] += (double)(u * u); Ys[other][ea] = zia; if (stored & 1) Ks[other][ea] = ka[6]; }
      if (onb) { const float u = erb / tb; acc[0] += (double)(u * u); Ys[other][eb] = zib; if (stored & 1) Ks[other][eb] = kb[6]; }
      if (stored & 2) {
        float ma = 0.f, mb = 0.f;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          if (DP_CMID[j] == 0.0) continue;
          const T wm = dtf * (T)DP_CMID[j];
          ma = __builtin_fmaf(ka[j], wm, ma); mb = __builtin_fmaf(kb[j], wm, mb);
        }
        if (ona) Mid[ea] = ya + ma;
        if (onb) Mid[eb] = yb + mb;
      }
    }
    if (has_next) feed_store(dbuf ^ 1);
    dbuf ^= 1;
  }
  block_total<2>(acc, red);
  if (tid == 0) { Pq[2 * blockIdx.x] = acc[0]; Pq[2 * blockIdx.x + 1] = acc[1]; }
}

__global__ void w16_image_kernel(const float* __restrict__ W, const float* __restrict__ bias, float* __restrict__ img, Dims d) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= W16_FLOATS) return;
  const int q4 = e & 3, l = (e >> 2) & 63, g = e >> 8;
  const int T = g / W16_GROUPS, grp = g - T * W16_GROUPS;
  img[e] = w16_image(W, bias, T, grp * 4 + q4, l, d);
}

static inline int dopri_ns(int64_t H) { int ns = (int)(256 / H); return ns < 1 ? 1 : (ns > 16 ? 16 : ns); }
static inline int64_t dopri_blocks(int64_t B, int64_t H) {
  const int ns = dopri_ns(H);
  int64_t tiles = (B + ns - 1) / ns;
  return tiles < 1 ? 1 : (tiles > 2048 ? 2048 : tiles);
}
static inline size_t al256(size_t x) { return (x + 255) / 256 * 256; }

// sharded batches: this shard's pending partial sums, added up in block order (what the next launch would do itself)
__global__ __launch_bounds__(64) void dopri_pending_sums_kernel(const double* __restrict__ partial, int64_t n_blocks,
                                                                int width, double* __restrict__ out) {
  const int k = threadIdx.x;
  if (k >= width) return;
  double s = 0.0;
  for (int64_t b = 0; b < n_blocks; ++b) s += partial[b * width + k];
  out[k] = s;
}
__global__ void wy16_image_kernel(const float* __restrict__ W, const float* __restrict__ bias, float* __restrict__ img, Dims d) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < WY_FLOATS) {
    const int j = e & 3, l = (e >> 2) & 63, g = e >> 8;
    img[e] = wy16_image(W, g >> 1, 4 * (g & 1) + j, l, d);
  } else if (e < ACT16_LDS_FLOATS) {
    const int b = e - WY_FLOATS;
    img[e] = by16_image(bias, b >> 4, (b >> 2) & 3, b & 3, d);
  }
}
__global__ void mlp16_image_kernel(const float* __restrict__ W1, const float* __restrict__ b1,
                                   const float* __restrict__ W2, const float* __restrict__ b2, float* __restrict__ img,
                                   MlpDims d, int nb) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < MLP16_LDS_FLOATS) img[e] = mlp16_image(W1, b1, W2, b2, e, d, nb);
}
static_assert(ACT16_LDS_FLOATS <= W16_FLOATS && W16_FLOATS <= MLP16_LDS_FLOATS, "all image forms share one workspace slot");
constexpr size_t DOPRI_IMAGE_BYTES = (size_t)MLP16_LDS_FLOATS * sizeof(float);

static inline bool dopri_use_mfma(int64_t C, int64_t H, int dtype, int act, int variant) {
  return variant != CDE_VARIANT_GENERIC && dtype == CDE_F32 && H <= MH && C <= MC &&
         (act == CDE_ACT_NONE || act == CDE_ACT_TANH);
}
bool wide_applicable(int64_t C, int64_t H, int dtype, int act);         // rk4_wide.hip
// one-layer fields beyond the 32 x 8 tiles (H <= 64, C <= 8 or H <= 32, C <= 16): the wide attempt kernel under AUTO
static inline bool dopri_use_wide(int64_t C, int64_t H, int dtype, int act, int variant) {
  return variant == CDE_VARIANT_AUTO && !dopri_use_mfma(C, H, dtype, act, variant) && wide_applicable(C, H, dtype, act);
}
static inline int64_t dopri_wide_grid(int64_t B, int64_t C) {           // workgroups walking the 16-series tiles
  const int64_t tiles = (B + 15) / 16, cap = C > MC ? 512 : 256;
  return tiles < cap ? tiles : cap;
}
static inline int64_t dopri_blocks_any(int64_t B, int64_t H) {          // partial buffer must fit every kernel's grid
  const int64_t a = dopri_blocks(B, H), b = (B + 127) / 128;
  const int64_t m = a > b ? a : b;
  return m > 512 ? m : 512;
}

}  // namespace cde

#ifdef CDE_PHASE_TRACE
// debug builds only (cde_common.h, "phase trace"): the stamp ring of dopri5_attempt_mfma, [ring][workgroup][slot]
extern "C" int cde_debug_k4_phase_trace(void* host_out, size_t bytes) {
  if (bytes > sizeof(unsigned long long) * cde::TRACE_RING * cde::TRACE_BLOCKS * cde::TRACE_SLOTS) return CDE_ERR_SHAPE;
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(cde::k4_phase_trace), bytes) == hipSuccess ? CDE_OK : CDE_ERR_LAUNCH;
}
#endif

// ================================================================================================ C ABI
extern "C" size_t cde_dopri5_trace_offset(int64_t B, int64_t C, int64_t H, int dtype) {
  (void)C;
  const size_t elem = dtype == CDE_F64 ? 8 : 4;
  return cde::al256(cde::al256(2 * sizeof(cde::DopriCtrl)) +
                    cde::al256((size_t)2 * cde::dopri_blocks_any(B, H) * 2 * sizeof(double)) +
                    (size_t)2 * 5 * B * H * elem + cde::al256(cde::DOPRI_IMAGE_BYTES));
}

extern "C" size_t cde_dopri5_workspace_bytes(int64_t B, int64_t C, int64_t H, int dtype) {
  return cde_dopri5_trace_offset(B, C, H, dtype) + cde::al256((size_t)CDE_DOPRI5_TRACE_STEPS * 3 * sizeof(double));
}

// W1 == nullptr: one-layer field (W, bias); otherwise W1/bias1/width is the hidden layer and W/bias the output layer
static int dopri5_advance_impl(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W1,
                               const void* bias1, int64_t width, const void* W, const void* bias, int act,
                               const void* z0, const double* t_out, int64_t n_out, const double* jump_t, int64_t n_jump,
                               double rtol, double atol, double safety, double ifactor, double dfactor, void* z_out,
                               int64_t B, int64_t C, int64_t H, int dtype, int variant, void* workspace,
                               size_t workspace_bytes, int64_t first_launch, int64_t n_launches, void* stream,
                               const double* ext_sums = nullptr, int64_t B_global = 0) {
  const bool mlp = W1 != nullptr;
  if (B < 1 || C < 1 || H < 1 || H > 256 || n_intervals < 1 || n_out < 1 || n_launches < 0 || n_jump < 0) return CDE_ERR_SHAPE;
  if (mlp && (width < 1 || !bias1)) return width < 1 ? CDE_ERR_SHAPE : CDE_ERR_NULL;
  const bool mlp_upper = mlp && cde::mlp_shape_hi(C, H, width) && ((uintptr_t)W & 15) == 0;     // 32 units x 16 channels (cde_mfma.h: MlpHi)
  if (mlp && (dtype != CDE_F32 || !(cde::mlp_shape_ok(C, H, width) || mlp_upper) || variant == CDE_VARIANT_GENERIC))
    return CDE_ERR_UNSUPPORTED;
  if (act != CDE_ACT_NONE && act != CDE_ACT_TANH) return CDE_ERR_UNSUPPORTED;
  if (degree != CDE_PATH_CUBIC && degree != CDE_PATH_LINEAR) return CDE_ERR_UNSUPPORTED;
  if (!coeffs || !knots || !W || !bias || !z0 || !t_out || !z_out || !workspace) return CDE_ERR_NULL;
  if (n_jump > 0 && !jump_t) return CDE_ERR_NULL;
  if (workspace_bytes < cde_dopri5_workspace_bytes(B, C, H, dtype)) return CDE_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const bool use_mfma = mlp || cde::dopri_use_mfma(C, H, dtype, act, variant);
  if (variant == CDE_VARIANT_MFMA && !use_mfma) return CDE_ERR_UNSUPPORTED;
  const int64_t blocks = cde::dopri_blocks_any(B, H);          // allocation stride of the partial sums
  unsigned char* base = (unsigned char*)workspace;
  cde::DopriCtrl* ctrl = (cde::DopriCtrl*)base;
  double* partial = (double*)(base + cde::al256(2 * sizeof(cde::DopriCtrl)));
  float* w16 = (float*)(base + cde::al256(2 * sizeof(cde::DopriCtrl)) + cde::al256((size_t)2 * blocks * 2 * sizeof(double)));
  void* state = (unsigned char*)w16 + cde::al256(cde::DOPRI_IMAGE_BYTES);
  double* trace = (double*)(base + cde_dopri5_trace_offset(B, C, H, dtype));
  if (first_launch == 0) {
    cde::zero_async(ctrl, 2 * sizeof(cde::DopriCtrl), s);                                                // phase 0
  }
  const int ns = cde::dopri_ns(H);
  const int nt = ((ns * (int)H + 63) / 64) * 64;
#define CDE_DOPRI(T)                                                                                              \
  do {                                                                                                            \
    cde::DopriArgs<T> g{(const T*)coeffs, (const T*)knots, n_intervals, degree, (const T*)W, (const T*)bias, act, \
                        (const T*)z0, t_out, n_out, jump_t, n_jump, rtol, atol, safety, ifactor, dfactor,         \
                        (T*)z_out, B, C, H, ns, ctrl, (T*)state, nullptr, partial, blocks};                        \
    g.trace = trace; g.ext_sums = ext_sums; g.B_global = B_global;                                                \
    const size_t lds = (((size_t)ns * (H + C) * sizeof(T) + 15) / 16) * 16 + 2 * nt * sizeof(double);                                 \
    for (int64_t i = 0; i < n_launches; ++i)                                                                      \
      cde::dopri5_attempt_kernel<T><<<(unsigned)cde::dopri_blocks(B, H), nt, lds, s>>>(g, (int)((first_launch + i) & 1)); \
  } while (0)
  if (!mlp && cde::dopri_use_wide(C, H, dtype, act, variant)) {
    cde::DopriArgs<float> g{(const float*)coeffs, (const float*)knots, n_intervals, degree, (const float*)W,
                            (const float*)bias, act, (const float*)z0, t_out, n_out, jump_t, n_jump, rtol, atol, safety,
                            ifactor, dfactor, (float*)z_out, B, C, H, 16, ctrl, (float*)state, w16, partial, blocks,
                            nullptr, nullptr, 0, trace, ext_sums, B_global};
    const unsigned grid = (unsigned)cde::dopri_wide_grid(B, C);
    const int64_t n_knots = n_intervals + 1;
#define CDE_WIDE(D, A, NWV, NBV)                                                                                   \
  do {                                                                                                             \
    const size_t lds = cde::dopri_wide_lds_bytes<NWV, NBV>(n_knots);                                               \
    (void)hipFuncSetAttribute((const void*)cde::dopri5_attempt_wide<D, A, NWV, NBV>,                               \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                               \
    for (int64_t i = 0; i < n_launches; ++i)                                                                       \
      cde::dopri5_attempt_wide<D, A, NWV, NBV><<<grid, 64 * NWV, lds, s>>>(g, (int)((first_launch + i) & 1));      \
  } while (0)
#define CDE_WIDE_SHAPE(D, A)                                                                                       \
  do {                                                                                                             \
    if (C <= cde::MC) CDE_WIDE(D, A, 8, 2); else CDE_WIDE(D, A, 4, 4);                                             \
  } while (0)
    if (act == CDE_ACT_NONE) {
      if (degree == CDE_PATH_CUBIC) CDE_WIDE_SHAPE(CDE_PATH_CUBIC, CDE_ACT_NONE); else CDE_WIDE_SHAPE(CDE_PATH_LINEAR, CDE_ACT_NONE);
    } else {
      if (degree == CDE_PATH_CUBIC) CDE_WIDE_SHAPE(CDE_PATH_CUBIC, CDE_ACT_TANH); else CDE_WIDE_SHAPE(CDE_PATH_LINEAR, CDE_ACT_TANH);
    }
#undef CDE_WIDE_SHAPE
#undef CDE_WIDE
    return cde::check_launch();
  }
  if (use_mfma) {
    cde::DopriArgs<float> g{(const float*)coeffs, (const float*)knots, n_intervals, degree, (const float*)W,
                            (const float*)bias, act, (const float*)z0, t_out, n_out, jump_t, n_jump, rtol, atol, safety,
                            ifactor, dfactor, (float*)z_out, B, C, H, 16, ctrl, (float*)state, w16, partial, blocks,
                            (const float*)W1, (const float*)bias1, (int)width, trace, ext_sums, B_global};
    const cde::Dims dims{(int)H, (int)C};
    const unsigned grid = (unsigned)((B + 127) / 128);
    const int64_t n_knots = n_intervals + 1;
    if (mlp) {
      if (first_launch == 0)
        cde::mlp16_image_kernel<<<(cde::MLP16_LDS_FLOATS + 255) / 256, 256, 0, s>>>(
            (const float*)W1, (const float*)bias1, (const float*)W, (const float*)bias, w16,
            cde::MlpDims{(int)H, (int)C, (int)width}, C > cde::MC ? 4 : 2);
      const size_t lds = 2 * 512 * sizeof(double) +
                         (n_knots <= cde::DOPRI_MAX_LDS_KNOTS_MLP ? (size_t)((n_knots + 3) / 4 * 4) * sizeof(float) : 0) +
                         cde::DOPRI_IMAGE_BYTES;
      // up to 256 tiles (one workgroup per CU): the 8 waves of a workgroup share a tile
      const int64_t tiles = (B + 15) / 16;
      const int64_t split_req = cde::option(CDE_OPT_K4M_SPLIT_TILES);           // (measurements; -1: the default)
      // (an override can only LOWER the threshold: the split form was measured and tested up to these tile counts)
      const int64_t split_max = C > cde::MC ? 256 : cde::DOPRI_MLP_SPLIT_TILES;
      const int64_t split_tiles = split_req >= 0 && split_req < split_max ? split_req : split_max;
      const bool split = tiles <= split_tiles && !ext_sums && B_global == 0 && !cde::option(CDE_OPT_K4M_NO_SPLIT);
      const size_t lds_split = lds + (size_t)cde::DOPRI_XWIN_FLOATS * sizeof(float);
#define CDE_MLP_CT(D, A, CTV, HIV)                                                                                 \
  do {                                                                                                             \
    if (split) {                                                                                                   \
      (void)hipFuncSetAttribute((const void*)cde::dopri5_attempt_mfma<D, A, true, CTV, true, HIV>,                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_split);                       \
      for (int64_t i = 0; i < n_launches; ++i)                                                                     \
        cde::dopri5_attempt_mfma<D, A, true, CTV, true, HIV><<<(unsigned)tiles, 512, lds_split, s>>>(              \
            g, (int)((first_launch + i) & 1));                                                                     \
    } else {                                                                                                       \
      (void)hipFuncSetAttribute((const void*)cde::dopri5_attempt_mfma<D, A, true, CTV, false, HIV>,                \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                             \
      for (int64_t i = 0; i < n_launches; ++i)                                                                     \
        cde::dopri5_attempt_mfma<D, A, true, CTV, false, HIV><<<grid, 512, lds, s>>>(g, (int)((first_launch + i) & 1)); \
    }                                                                                                              \
  } while (0)
#define CDE_MLP(D, A)                                                                                              \
  do {                                                                                                             \
    if (mlp_upper) CDE_MLP_CT(D, A, 16, true); else if (C > cde::MC) CDE_MLP_CT(D, A, 16, false); else CDE_MLP_CT(D, A, cde::MC, false); \
  } while (0)
      if (act == CDE_ACT_NONE) {
        if (degree == CDE_PATH_CUBIC) CDE_MLP(CDE_PATH_CUBIC, CDE_ACT_NONE); else CDE_MLP(CDE_PATH_LINEAR, CDE_ACT_NONE);
      } else {
        if (degree == CDE_PATH_CUBIC) CDE_MLP(CDE_PATH_CUBIC, CDE_ACT_TANH); else CDE_MLP(CDE_PATH_LINEAR, CDE_ACT_TANH);
      }
#undef CDE_MLP
#undef CDE_MLP_CT
      return cde::check_launch();
    }
    if (first_launch == 0) {
      if (act == CDE_ACT_NONE)
        cde::w16_image_kernel<<<(cde::W16_FLOATS + 255) / 256, 256, 0, s>>>((const float*)W, (const float*)bias, w16, dims);
      else
        cde::wy16_image_kernel<<<(cde::ACT16_LDS_FLOATS + 255) / 256, 256, 0, s>>>((const float*)W, (const float*)bias, w16, dims);
    }
    const size_t lds = 2 * 512 * sizeof(double) +
                       (n_knots <= cde::DOPRI_MAX_LDS_KNOTS ? (size_t)((n_knots + 3) / 4 * 4) * sizeof(float) : 0) +
                       (act == CDE_ACT_NONE ? 0 : cde::ACT16_LDS_FLOATS * sizeof(float));
    // small batches (at most one tile per CU): the 8 waves of a workgroup share a tile -- tanh fields one unit group each,
    // identity fields one K group each
    const int64_t tiles_act = (B + 15) / 16;
    const bool split_act = tiles_act <= 256 && !ext_sums && B_global == 0 && !cde::option(CDE_OPT_K4_NO_SPLIT);
    const size_t lds_split = lds + (size_t)8 * 64 * 9 * sizeof(float);
    // every form may ask for more than the 64 KB a kernel gets by default (knot buffer up to 32 KB + the 33.8 KB tanh
    // image + the split forms' 18 KB exchange window): the limit is raised per instantiation, as for the other families
#define CDE_K4_ONE(D, A)                                                                                           \
  do {                                                                                                             \
    if (split_act) {                                                                                               \
      (void)hipFuncSetAttribute((const void*)cde::dopri5_attempt_mfma<D, A, false, cde::MC, true>,                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_split);                       \
      for (int64_t i = 0; i < n_launches; ++i)                                                                     \
        cde::dopri5_attempt_mfma<D, A, false, cde::MC, true><<<(unsigned)tiles_act, 512, lds_split, s>>>(          \
            g, (int)((first_launch + i) & 1));                                                                     \
    } else {                                                                                                       \
      (void)hipFuncSetAttribute((const void*)cde::dopri5_attempt_mfma<D, A>,                                       \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                             \
      for (int64_t i = 0; i < n_launches; ++i)                                                                     \
        cde::dopri5_attempt_mfma<D, A><<<grid, 512, lds, s>>>(g, (int)((first_launch + i) & 1));                   \
    }                                                                                                              \
  } while (0)
    if (act == CDE_ACT_NONE) {
      if (degree == CDE_PATH_CUBIC) CDE_K4_ONE(CDE_PATH_CUBIC, CDE_ACT_NONE); else CDE_K4_ONE(CDE_PATH_LINEAR, CDE_ACT_NONE);
    } else {
      if (degree == CDE_PATH_CUBIC) CDE_K4_ONE(CDE_PATH_CUBIC, CDE_ACT_TANH); else CDE_K4_ONE(CDE_PATH_LINEAR, CDE_ACT_TANH);
    }
#undef CDE_K4_ONE
    return cde::check_launch();
  }
  if (dtype == CDE_F32) CDE_DOPRI(float);
  else if (dtype == CDE_F64) CDE_DOPRI(double);
  else return CDE_ERR_DTYPE;
#undef CDE_DOPRI
  return cde::check_launch();
}

extern "C" int cde_dopri5_advance(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                                  const void* bias, int act, const void* z0, const double* t_out, int64_t n_out,
                                  const double* jump_t, int64_t n_jump, double rtol, double atol, double safety,
                                  double ifactor, double dfactor, void* z_out, int64_t B, int64_t C, int64_t H,
                                  int dtype, int variant, void* workspace, size_t workspace_bytes,
                                  int64_t first_launch, int64_t n_launches, void* stream) {
  return dopri5_advance_impl(coeffs, knots, n_intervals, degree, nullptr, nullptr, 0, W, bias, act, z0, t_out, n_out,
                             jump_t, n_jump, rtol, atol, safety, ifactor, dfactor, z_out, B, C, H, dtype, variant,
                             workspace, workspace_bytes, first_launch, n_launches, stream);
}

// Sharded batches with ONE step controller (torchdiffeq's semantics for the whole batch): per attempted step every
// shard calls cde_dopri5_pending_sums, the 2 doubles are all-reduced (sum) across the shards, and every shard runs ONE
// launch of cde_dopri5_advance_sharded with the reduced sums and the global batch size.
extern "C" int cde_dopri5_pending_sums(const void* workspace, size_t workspace_bytes, int64_t B, int64_t C, int64_t H,
                                       int dtype, int variant, int act, int64_t total_launches, double* sums, void* stream) {
  if (B < 1 || C < 1 || H < 1) return CDE_ERR_SHAPE;
  if (!workspace || !sums) return CDE_ERR_NULL;
  if (workspace_bytes < cde_dopri5_workspace_bytes(B, C, H, dtype)) return CDE_ERR_WORKSPACE;
  const int64_t stride = cde::dopri_blocks_any(B, H);
  const bool use_mfma = cde::dopri_use_mfma(C, H, dtype, act, variant);
  const int64_t live = use_mfma ? (B + 127) / 128
                       : cde::dopri_use_wide(C, H, dtype, act, variant) ? cde::dopri_wide_grid(B, C)
                                                                        : cde::dopri_blocks(B, H);   // the grid of the attempt kernel
  const double* partial = (const double*)((const unsigned char*)workspace + cde::al256(2 * sizeof(cde::DopriCtrl))) +
                          (total_launches & 1) * stride * 2;
  cde::dopri_pending_sums_kernel<<<1, 64, 0, (hipStream_t)stream>>>(partial, live, 2, sums);
  return cde::check_launch();
}

extern "C" int cde_dopri5_advance_sharded(const void* coeffs, const void* knots, int64_t n_intervals, int degree,
                                          const void* W, const void* bias, int act, const void* z0, const double* t_out,
                                          int64_t n_out, const double* jump_t, int64_t n_jump, double rtol, double atol,
                                          double safety, double ifactor, double dfactor, void* z_out, int64_t B,
                                          int64_t C, int64_t H, int dtype, int variant, void* workspace,
                                          size_t workspace_bytes, int64_t first_launch, const double* reduced_sums,
                                          int64_t B_global, void* stream) {
  if (!reduced_sums || B_global < B) return reduced_sums ? CDE_ERR_SHAPE : CDE_ERR_NULL;
  return dopri5_advance_impl(coeffs, knots, n_intervals, degree, nullptr, nullptr, 0, W, bias, act, z0, t_out, n_out,
                             jump_t, n_jump, rtol, atol, safety, ifactor, dfactor, z_out, B, C, H, dtype, variant,
                             workspace, workspace_bytes, first_launch, 1, stream, reduced_sums, B_global);
}

extern "C" int cde_dopri5_advance_mlp(const void* coeffs, const void* knots, int64_t n_intervals, int degree,
                                      const void* W1, const void* bias1, int64_t width, const void* W2,
                                      const void* bias2, int act, const void* z0, const double* t_out, int64_t n_out,
                                      const double* jump_t, int64_t n_jump, double rtol, double atol, double safety,
                                      double ifactor, double dfactor, void* z_out, int64_t B, int64_t C, int64_t H,
                                      int dtype, void* workspace, size_t workspace_bytes, int64_t first_launch,
                                      int64_t n_launches, void* stream) {
  if (!W1) return CDE_ERR_NULL;
  return dopri5_advance_impl(coeffs, knots, n_intervals, degree, W1, bias1, width, W2, bias2, act, z0, t_out, n_out,
                             jump_t, n_jump, rtol, atol, safety, ifactor, dfactor, z_out, B, C, H, dtype,
                             CDE_VARIANT_AUTO, workspace, workspace_bytes, first_launch, n_launches, stream);
}

// The same two calls for the two-layer field (round 4): one controller across the shards of a sharded batch for the
// call every example of the reference makes (cde_dopri5_advance_mlp's arguments + the reduced sums / global batch).
extern "C" int cde_dopri5_pending_sums_mlp(const void* workspace, size_t workspace_bytes, int64_t B, int64_t C, int64_t H,
                                           int dtype, int64_t total_launches, double* sums, void* stream) {
  if (B < 1 || C < 1 || H < 1) return CDE_ERR_SHAPE;
  if (!workspace || !sums) return CDE_ERR_NULL;
  if (workspace_bytes < cde_dopri5_workspace_bytes(B, C, H, dtype)) return CDE_ERR_WORKSPACE;
  const int64_t stride = cde::dopri_blocks_any(B, H);
  const int64_t live = (B + 127) / 128;                 // the grid of the (never split, when sharded) two-layer attempt kernel
  const double* partial = (const double*)((const unsigned char*)workspace + cde::al256(2 * sizeof(cde::DopriCtrl))) +
                          (total_launches & 1) * stride * 2;
  cde::dopri_pending_sums_kernel<<<1, 64, 0, (hipStream_t)stream>>>(partial, live, 2, sums);
  return cde::check_launch();
}

extern "C" int cde_dopri5_advance_mlp_sharded(const void* coeffs, const void* knots, int64_t n_intervals, int degree,
                                              const void* W1, const void* bias1, int64_t width, const void* W2,
                                              const void* bias2, int act, const void* z0, const double* t_out,
                                              int64_t n_out, const double* jump_t, int64_t n_jump, double rtol, double atol,
                                              double safety, double ifactor, double dfactor, void* z_out, int64_t B,
                                              int64_t C, int64_t H, int dtype, void* workspace, size_t workspace_bytes,
                                              int64_t first_launch, const double* reduced_sums, int64_t B_global,
                                              void* stream) {
  if (!W1) return CDE_ERR_NULL;
  if (!reduced_sums || B_global < B) return reduced_sums ? CDE_ERR_SHAPE : CDE_ERR_NULL;
  return dopri5_advance_impl(coeffs, knots, n_intervals, degree, W1, bias1, width, W2, bias2, act, z0, t_out, n_out,
                             jump_t, n_jump, rtol, atol, safety, ifactor, dfactor, z_out, B, C, H, dtype,
                             CDE_VARIANT_AUTO, workspace, workspace_bytes, first_launch, 1, stream, reduced_sums, B_global);
}
