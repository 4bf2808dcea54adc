// cde_dopri_adj.h -- the step controller of the adaptive ADJOINT solves (K4a: dopri5_adjoint.hip, K4am:
// dopri5_mlp_adjoint.hip), shared so that both kernels take torchdiffeq's decisions with the same arithmetic.
//
// torchdiffeq.odeint_adjoint's backward (behind reference solver.py:226; tolerances solver.py:199-203) integrates the
// augmented state (vjp_t, y, a, dL/dtheta_1, ..) in reversed time, one odeint call per output interval.  Its step
// controller sees the WHOLE augmented state through the default "mixed" norm
//     max(|e_t|, rms(e_y), rms(e_a), max_p rms(e_theta_p))                (e = error / tolerance, elementwise)
// -- `adjoint_options=dict(norm="seminorm")` drops the parameter blocks -- and the solve of an interval ends by stepping
// PAST the interval end and evaluating the 4th-order dense interpolant there (restated in oracle/odeint.py: _Dopri5,
// _Adjoint, odeint_adjoint).  Round 2's K4a used the state-only norm and clipped the last step: both deviations are gone.
//
// What the parameter blocks need per attempted step is a grid-wide sum of per-workgroup gradient images, too large to
// redo in every workgroup's prologue; a small "R" kernel after every attempt kernel owns the running total G, adds the
// accepted attempt's increment to it, and leaves per-block sums of (e_theta)^2 that the next attempt kernel's prologue
// adds up together with the state sums.  vjp_t (a scalar: the field depends on t through dX/dt(t), and torchdiffeq
// always carries it -- DESIGN.md section 4) rides in the state sums.
#pragma once
#include "cde_dopri.h"

namespace cde {

constexpr int ADJ_NS = 8;                // pending state sums per launch (see adj_controller)
constexpr int ADJ_MAX_PT = 6;            // tensors in the mixed norm: W, b (one-layer) or W1, b1, W2, b2 -- and, when they are
                                         // among adjoint_params, the control's coefficient tensor and its knot times
constexpr int ADJ_CTRL_STRIDE = 256;     // bytes between the two controller blocks at the head of the workspace
constexpr int ADJ_MAX_RBLOCKS = 256;     // blocks of the R kernel (partial parameter sums per launch)
constexpr int ADJ_TRACE_ATTEMPTS = 16384;  // rows of the attempt trace: (t0, t1, clipped onto a jump, accepted, error ratio)

struct AdjCtrl {
  DopriCtrl c;                  // K4's controller block (the host reads it through cde_dopri5_status)
  double T;                     // vjp_t, committed
  double x_end;                 // pending attempt: (s1 - t0) / (t1 - t0) when it reaches the interval end
  int32_t over;                 // pending attempt reaches s1 (t1 >= s1): accepting it ends the interval
  int32_t commit;               // R kernel of THIS launch: 0 nothing, 1 add the previous attempt's A to the running
                                // totals, 2 add THIS launch's image (mode 3: the dense output at the interval end)
  int32_t mode;                 // what this launch computes: 0 f0 norms, 1 f1 norm, 2 attempt, 3 dense output at s1
  int32_t kind0;                // stage-0 time perturbation of the pending attempt (mode 3 repeats that attempt)
  // K4am, eight-wave form (first-same-as-last, as torchdiffeq keeps f0 / f1): where the pending attempt's FIRST stage lives
  // -- factor-row block and slope stash `src0` (0: evaluated by that launch; 5 / 6: the last stage of the step accepted
  // before it) -- and where its LAST stage went (`six`: 5 or 6, never `src0`).  The other forms set (0, 5): every stage
  // evaluated, the last one in block 5.
  int32_t src0, six;            // (six & 15: the block; six & ADJ_FRESH0: this launch evaluated its first stage -- its factor
                                //  rows and its stage image are new; a separate word would cost the attempt kernels a register)
};
constexpr int ADJ_FRESH0 = 16;
static_assert(sizeof(AdjCtrl) <= ADJ_CTRL_STRIDE, "controller block outgrew its slot");

struct AdjCommon {
  double s0, s1;                           // the interval in reversed time, s0 < s1
  const double* jump_s; int64_t n_jump;    // jump times in reversed time, ascending
  double rtol, atol, safety, ifactor, dfactor;
  int64_t n_state;                         // elements of y (= of a) the state norms run over: B_global * H
  int64_t n_param[ADJ_MAX_PT]; int n_pt;   // true element counts of the parameter tensors
  int norm_kind;                           // 0: torchdiffeq's default mixed norm, 1: "seminorm"
  double* trace;                           // [CDE_DOPRI5_TRACE_STEPS][3]
  double* trace_all;                       // [ADJ_TRACE_ATTEMPTS][5]: EVERY decided attempt with its error ratio (tests
                                           // replay them through the oracle: same state -> same ratio -> same decision)
  double* carry;                           // [0]: vjp_t across the output intervals of one backward pass
};

struct AdjPlan {
  bool accept;          // decision on the pending attempt (phase 3)
  int mode;             // this launch: 0 = f0 norms, 1 = f1 norm, 2 = attempt, 3 = interval finished
  double t0, t1, dt;
  float h0;
  int kind0;            // perturbation of the stage-0 time: 0 none, -1 just before, +1 just after
  float x_end;          // mode 3: where in the step the interval end lies
};

// S[0..7]: sums the previous launch left (its meaning depends on what that launch computed)
//   mode 0:  S0 = sum (y/sc)^2   S1 = sum (a/sc)^2   S2 = sum (f_y/sc)^2   S3 = sum (f_a/sc)^2   S4 = f_t
//   mode 1:  S0 = sum ((f1-f0)_y/sc)^2, S1 likewise for a, S4 = (f1 - f0)_t
//   mode 2:  S0 = sum (err_y/tol)^2, S1 for a, S4 = dt sum c_sol k_t, S5 = dt sum c_err k_t
//   mode 3:  S4 = dt sum w_D k_t (consumed by the R kernel, not here)
// Q[2p], Q[2p+1]: the same two slots for parameter tensor p (from the R kernel).
__device__ __forceinline__ AdjPlan adj_controller(const AdjCommon& g, AdjCtrl& k, const double* S, const double* Q) {
  DopriCtrl& c = k.c;
  const double n_elems = (double)g.n_state;
  auto rms = [](double s, double n) { return (float)sqrt(s / n); };
  auto maxf = [](float a, float b) { return a > b ? a : b; };
  const float rtol = (float)g.rtol, atol = (float)g.atol;
  auto params = [&](int slot) {
    float m = 0.f;
    if (g.norm_kind == 0)
      for (int p = 0; p < g.n_pt; ++p) m = maxf(m, rms(Q[2 * p + slot], (double)g.n_param[p]));
    return m;
  };
  AdjPlan plan{};
  bool accept = false;
  int mode;
  k.commit = 0;
  if (c.phase == 0) {
    mode = 0;
    c.t_lo = c.t_hi = g.s0;
    c.i_out = 1; c.n_accept = c.n_reject = 0; c.refresh = 0; c.on_jump = 0;
    int64_t j = 0;
    while (j < g.n_jump && g.jump_s[j] < c.t_hi) ++j;             // torchdiffeq keeps jump times >= t0 ...
    const int64_t first = j;
    while (j < g.n_jump && g.jump_s[j] <= c.t_hi) ++j;            // ... and starts at bisect_right(jump_t, t0)
    c.i_jump = j - first;
    if (g.n_jump - first > 0 && c.i_jump > g.n_jump - first - 1) c.i_jump = g.n_jump - first - 1;
    c.pad = (int32_t)first;
    k.T = g.carry[0];
    k.over = 0; k.x_end = 0.0;
  } else if (c.phase == 1) {
    const float T = (float)k.T, sct = atol + fabsf(T) * rtol;
    const float d0 = maxf(maxf(fabsf(T / sct), params(0)), maxf(rms(S[0], n_elems), rms(S[1], n_elems)));
    const float d1 = maxf(maxf(fabsf((float)S[4] / sct), params(1)), maxf(rms(S[2], n_elems), rms(S[3], n_elems)));
    float h0 = (d0 < 1e-5f || d1 < 1e-5f) ? 1e-6f : 0.01f * d0 / d1;
    h0 = h0 < 0 ? -h0 : h0;
    c.h0 = (double)h0;
    c.dt = (double)d1;                                            // parked for phase 2
    plan.h0 = h0;
    mode = 1;
  } else if (c.phase == 2) {
    const float T = (float)k.T, sct = atol + fabsf(T) * rtol;
    const float h0 = (float)c.h0, d1 = (float)c.dt;
    const float d2 = maxf(maxf(fabsf((float)S[4] / sct), params(0)), maxf(rms(S[0], n_elems), rms(S[1], n_elems))) / h0;
    float h1;
    if (d1 <= 1e-15f && d2 <= 1e-15f) { const float a = 1e-6f, b = h0 * 1e-3f; h1 = a > b ? a : b; }
    else h1 = powf(0.01f / maxf(d1, d2), (float)(1.0 / 5.0));
    h1 = h1 < 0 ? -h1 : h1;
    const float hundred = 100.f * h0;
    c.dt = (double)(hundred < h1 ? hundred : h1);
    mode = 2;
  } else {
    const float T = (float)k.T, T1 = T + (float)S[4];
    const float tol_t = atol + rtol * maxf(fabsf(T), fabsf(T1));
    const float ratio_t = maxf(maxf(fabsf((float)S[5] / tol_t), params(0)), maxf(rms(S[0], n_elems), rms(S[1], n_elems)));
    accept = ratio_t <= 1.f;
    if (g.trace_all && blockIdx.x == 0 && threadIdx.x == 0 && c.n_accept + c.n_reject < ADJ_TRACE_ATTEMPTS) {
      double* row = g.trace_all + 5 * (c.n_accept + c.n_reject);
      row[0] = c.t_hi; row[1] = c.t1_try; row[2] = c.on_jump ? 1.0 : 0.0; row[3] = accept ? 1.0 : 0.0; row[4] = (double)ratio_t;
    }
    if (accept) {
      c.n_accept++;
      c.t_lo = c.t_hi; c.t_hi = c.t1_try;
      if (g.trace && blockIdx.x == 0 && threadIdx.x == 0 && c.n_accept <= CDE_DOPRI5_TRACE_STEPS) {
        g.trace[3 * (c.n_accept - 1)] = c.t_lo;
        g.trace[3 * (c.n_accept - 1) + 1] = c.t_hi;
        g.trace[3 * (c.n_accept - 1) + 2] = c.on_jump ? 1.0 : 0.0;
      }
      c.refresh = 0;
      if (c.on_jump) {
        const int64_t kept = g.n_jump - c.pad;
        if (c.i_jump != kept - 1) c.i_jump++;
        c.refresh = 1;
      }
      // a step that reached s1 ends the interval: its increment is NOT committed -- this launch repeats the step and
      // accumulates the dense output at s1 instead (mode 3), which the R kernel adds to the running totals
      k.commit = k.over ? 2 : 1;
      if (!k.over) k.T = (double)T1;
    } else {
      c.n_reject++;
      c.t_lo = c.t_hi;
    }
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
  double t0 = 0, t1 = 0, dt = 0;
  if (c.phase == 3 && accept && k.over) {                         // torchdiffeq: `while next_t > t1: step` has ended
    mode = 3;
    t0 = c.t_lo; t1 = c.t_hi; dt = c.dt_try;                      // the accepted step, once more
    plan.kind0 = k.kind0; plan.x_end = (float)k.x_end;
  }
  if (mode == 2) {
    k.over = 0; k.x_end = 0.0;
    t0 = c.t_hi;
    dt = c.dt;
    if (!(dt == dt) || dt > 1e300 || dt < -1e300) dt = 0.0;
    t1 = t0 + dt;
    int on_jump = 0;
    const int64_t kept = g.n_jump - c.pad;
    if (kept > 0) {
      const double nxt = g.jump_s[c.pad + c.i_jump];
      if (t0 < nxt && nxt < t0 + dt) { on_jump = 1; t1 = nxt; dt = t1 - t0; }
    }
    c.t1_try = t1; c.dt_try = dt; c.on_jump = on_jump;
    // no clipping at the interval end: the step passes it and the dense interpolant is evaluated there
    if (!(t1 < g.s1)) { k.over = 1; k.x_end = (g.s1 - t0) / (t1 - t0); }
    k.kind0 = c.refresh ? 1 : (c.n_accept > 0 ? -1 : 0);
    plan.kind0 = k.kind0;
  }
  k.mode = mode;
  plan.accept = accept; plan.mode = mode; plan.t0 = t0; plan.t1 = t1; plan.dt = dt;
  return plan;
}

// Stage weights of the two linear functionals of the seven stage slopes a launch accumulates, for the parameter
// gradients (gradient images) and for vjp_t alike:
//   S  mode 0: k_0     mode 1: k_1 - k_0     mode 2: the step's increment dt sum c_sol k
//      mode 3: the dense output at the interval end minus the step's start value.  The quartic of torchdiffeq's
//      _interp_fit is linear in (y0, y1, f0, f1, y_mid); with y1 = y0 + dt sum c_sol k, y_mid = y0 + dt sum c_mid k,
//      f0 = k_0, f1 = k_6 the y0 terms cancel and
//        w_D[s] = dt ( x d_s0 + x^2 (d_s6 - 4 d_s0 - 5 c_sol[s] + 16 c_mid[s]) + x^3 (5 d_s0 - 3 d_s6 + 14 c_sol[s] - 32 c_mid[s])
//                      + x^4 (2 d_s6 - 2 d_s0 - 8 c_sol[s] + 16 c_mid[s]) )            (x = 1: dt c_sol;  x = 1/2: dt c_mid)
//   E  mode 2: the error estimate dt sum c_err k (only when the parameter blocks / vjp_t take part in the norm)
__device__ __forceinline__ void adj_stage_weights(int mode, float dtf, float x, float (&wS)[7], float (&wE)[7]) {
#pragma unroll
  for (int s = 0; s < 7; ++s) { wS[s] = 0.f; wE[s] = 0.f; }
  if (mode == 0) wS[0] = 1.f;
  else if (mode == 1) { wS[0] = -1.f; wS[1] = 1.f; }
  else {
    const float x2 = x * x, x3 = x2 * x, x4 = x3 * x;
#pragma unroll
    for (int s = 0; s < 7; ++s) {
      const float cs = s < 6 ? (float)DP_BETA[5][s] : 0.f, cm = (float)DP_CMID[s];
      const float d0 = s == 0 ? 1.f : 0.f, d6 = s == 6 ? 1.f : 0.f;
      if (mode == 2) { wS[s] = dtf * cs; wE[s] = dtf * (float)DP_CERR[s]; }
      else
        wS[s] = dtf * (x * d0 + x2 * (d6 - 4.f * d0 - 5.f * cs + 16.f * cm) + x3 * (5.f * d0 - 3.f * d6 + 14.f * cs - 32.f * cm) +
                       x4 * (2.f * d6 - 2.f * d0 - 8.f * cs + 16.f * cm));
    }
  }
}

// One element of a parameter tensor in the R kernel: commit what the controller decided, then this launch's contribution
// to the two norm slots.  g: running total (returned updated), s_prev: the previous attempt's S sum, (S, E): this launch's.
__device__ __forceinline__ float adj_param_element(const AdjCtrl& k, float rtol, float atol, float g, float s_prev, float S,
                                                   float E, double& q0, double& q1) {
  if (k.commit == 1) g += s_prev;
  else if (k.commit == 2) g += S;                      // mode 3: S is the dense output at the interval end minus the start
  if (k.mode == 0) {
    const float sc = atol + fabsf(g) * rtol, u = g / sc, v = S / sc;
    q0 += (double)(u * u); q1 += (double)(v * v);
  } else if (k.mode == 1) {
    const float sc = atol + fabsf(g) * rtol, v = S / sc;
    q0 += (double)(v * v);
  } else if (k.mode == 2) {
    const float tol = atol + rtol * fmaxf(fabsf(g), fabsf(g + S)), v = E / tol;
    q0 += (double)(v * v);
  }
  return g;
}

}  // namespace cde
