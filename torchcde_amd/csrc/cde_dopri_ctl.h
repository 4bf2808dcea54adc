// cde_dopri_ctl.h -- control gradients through the adaptive backward kernels (K4a: dopri5_adjoint.hip, K4am:
// dopri5_mlp_adjoint.hip): what the two attempt kernels leave per launch and the small kernel that turns it into the
// coefficient / knot-time blocks of torchdiffeq's adjoint state.
#pragma once
#include "cde_dopri_adj.h"

namespace cde {

// DCTRL (control gradients: adjoint_params holds the coefficient tensor the path was built from, reference solver.py:207-222,
// README.md:251-270).  torchdiffeq then integrates one more block of the augmented state -- dL/dcoeffs, the size of the
// coefficient tensor -- and measures it in the mixed norm like the parameter blocks.  Its integrand is local to a series:
// per stage gx_c = sum_h a_h act(Y)_hc = d(a.f)/d(dX_c), chained to the coefficient row in use (cubic: 1, frac, frac^2 on
// b, 2c, 3d; linear: -+1/width on the two knot values).  The chain waves (product form only: the cached-Jacobian chain
// never forms act(Y)) leave their two units' shares of gx in an LDS tile, helper waves 0 / 1 add the 16 shares of a
// (series, channel) one stage behind and store the seven UNWEIGHTED values of the attempt; `adjoint_control_kernel` (one
// thread per (series, channel), right after the R kernel) applies the stage weights of the launch's record, commits to the
// gradient tensor itself -- which is the running total G of this block -- and leaves the block's norm sums.
constexpr int ADJ_GX_TILE = 16 * 16 * 8;                         // [chain wave * 4 + lane quarter][series][channel]
constexpr int ADJ_GX_ROW = 64;                                   // floats per series in the pending buffer: [channel][8 stages]
                                                                 // (8-channel tiles; the 16-channel tiles of K4am: 128)
constexpr int ADJ_KT_MAX_WG = 1024;                              // workgroups whose per-stage time-term sums the knot block adds up
struct AdjStageRec {                                             // what a launch computed, for the control kernel (uniform)
  int32_t mode, ns;
  int32_t sidx[7];
  float sfrac[7], wS[7], wE[7];
};
constexpr int ADJ_REC_STRIDE = 128;
static_assert(sizeof(AdjStageRec) <= ADJ_REC_STRIDE, "stage record outgrew its slot");

// ------------------------------------------------------------------------------------------ the control kernel (DCTRL)
// One thread per (series, channel), launched after the R kernel of every attempt launch.  The gradient tensor `G` (layout of
// the coefficient tensor, zeroed by the caller) IS the running total of the block; per launch the thread
//   1. commits what the controller decided: the previous attempt's increment (commit 1: its record and pending values are
//      still in the other parity's slots) or this launch's dense-output functional (commit 2, mode 3);
//   2. adds this launch's contribution to the block's two norm slots -- over the entries the launch's stages touch (elsewhere
//      its S and E are zero), and in mode 0 over the whole tensor (Hairer's d0 = |G / scale|) -- with the float arithmetic
//      of adj_param_element.
// Entries: cubic (row, j) <- sum over the stages in that row of w gx frac^j (b, 2c, 3d: interpolation_cubic.py:334-335);
// linear knot value e <- sum of +- w gx / width over the stages whose interval ends / starts there
// (interpolation_linear.py:186-191, :222-225).
struct AdjControlArgs {
  unsigned char* ctrl; const unsigned char* rec; const float* gx; float* G; const float* knots; double* cq;
  int64_t B, n_intervals;
  int C, degree, norm_kind;
  float rtol, atol;
  // the knot-time block (nullptr: not requested): G_knots (n_intervals + 1 floats, zeroed by the caller) is its running total
  float* G_knots; const double* ktp; int n_wg, kt_stride;          // ktp: [2][kt_stride workgroups][8]
};

// The knot times as a block (adjoint_params = (.., coeffs, t), reference test/test_tricks.py:21-49).  f depends on knot j
// through frac = t - t_j (cubic: df/dt_j = -F d2X/dt2 for the stages in interval j, interpolation_cubic.py:315-336) or through
// the widths of the slopes (linear: d slope_j / d h_j = -slope_j / h_j, interpolation_linear.py:186-191), so per stage ONE
// batch sum drives it: KT(s) = sum over the series of a . F d2X/dt2 (cubic) or of a . f (linear), left per workgroup by the
// attempt kernel.  Entry e of the launch described by `rc`:
//   cubic :  - sum_{s: idx(s) == e} w(s) KT(s)
//   linear:  + sum_{s: idx(s) == e} w(s) KT(s) / h_e  -  sum_{s: idx(s) + 1 == e} w(s) KT(s) / h_{e-1}
template <int DEGREE>
__device__ __forceinline__ void knot_entry(const AdjStageRec& rc, const double (&KT)[7], const float* __restrict__ knots, int e,
                                           float& S, float& E) {
  S = 0.f; E = 0.f;
#pragma unroll
  for (int s = 0; s < 7; ++s) {
    if (s < rc.ns) {
      float chain = 0.f;
      if (DEGREE == CDE_PATH_CUBIC) { if (rc.sidx[s] == e) chain = -1.f; }
      else {
        const float width = knots[rc.sidx[s] + 1] - knots[rc.sidx[s]];
        if (rc.sidx[s] == e) chain = 1.f / width; else if (rc.sidx[s] + 1 == e) chain = -1.f / width;
      }
      const float v = (float)KT[s] * chain;
      S = __builtin_fmaf(rc.wS[s], v, S);
      E = __builtin_fmaf(rc.wE[s], v, E);
    }
  }
}

template <int DEGREE>
__device__ __forceinline__ void control_entry(const AdjStageRec& rc, const float (&gxv)[8], const float* __restrict__ knots,
                                              int e, int j, float& S, float& E) {
  S = 0.f; E = 0.f;
#pragma unroll
  for (int s = 0; s < 7; ++s) {
    if (s < rc.ns) {
      float chain = 0.f;
      if (DEGREE == CDE_PATH_CUBIC) {
        if (rc.sidx[s] == e) chain = j == 0 ? 1.f : j == 1 ? rc.sfrac[s] : rc.sfrac[s] * rc.sfrac[s];
      } else {
        const float width = knots[rc.sidx[s] + 1] - knots[rc.sidx[s]];
        if (rc.sidx[s] == e) chain = -1.f / width; else if (rc.sidx[s] + 1 == e) chain = 1.f / width;
      }
      const float v = gxv[s] * chain;
      S = __builtin_fmaf(rc.wS[s], v, S);
      E = __builtin_fmaf(rc.wE[s], v, E);
    }
  }
}

template <int DEGREE, int CT = 8>
__global__ __launch_bounds__(256) void adjoint_control_kernel(AdjControlArgs r, int parity) {
  __shared__ double red[2][4];
  const int p2 = parity ^ 1;
  const AdjCtrl k = *reinterpret_cast<const AdjCtrl*>(r.ctrl + p2 * ADJ_CTRL_STRIDE);      // written by the attempt launch just before this one
  if (k.c.phase == 4 && k.commit == 0) return;
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t series = id / CT;
  const int c = (int)(id % CT);
  const bool on = series < r.B && c < r.C;
  constexpr int NJ = DEGREE == CDE_PATH_CUBIC ? 3 : 1;
  double q0 = 0.0, q1 = 0.0;
  if (on) {
    const AdjStageRec cur = *reinterpret_cast<const AdjStageRec*>(r.rec + parity * ADJ_REC_STRIDE);
    auto pending = [&](int which, float (&v)[8]) {
      const float* src = r.gx + (((int64_t)which * r.B + series) * CT + c) * 8;
      const float4 lo = *reinterpret_cast<const float4*>(src), hi = *reinterpret_cast<const float4*>(src + 4);
      v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
    };
    auto element = [&](int e, int j) -> float* {
      return DEGREE == CDE_PATH_CUBIC ? r.G + ((series * r.n_intervals + e) * 4 + 1 + j) * r.C + c
                                      : r.G + (series * (r.n_intervals + 1) + e) * r.C + c;
    };
    auto span = [&](const AdjStageRec& rc, int& lo, int& hi) {
      lo = rc.sidx[0]; hi = rc.sidx[0];
      for (int s = 1; s < rc.ns; ++s) { lo = rc.sidx[s] < lo ? rc.sidx[s] : lo; hi = rc.sidx[s] > hi ? rc.sidx[s] : hi; }
      if (DEGREE == CDE_PATH_LINEAR) hi += 1;
    };
    float gxc[8];
    pending(parity, gxc);
    // ---- 1. commit
    if (k.commit == 1) {
      const AdjStageRec prev = *reinterpret_cast<const AdjStageRec*>(r.rec + p2 * ADJ_REC_STRIDE);
      float gxp[8];
      pending(p2, gxp);
      int lo, hi;
      span(prev, lo, hi);
      for (int e = lo; e <= hi; ++e)
        for (int j = 0; j < NJ; ++j) {
          float S, E;
          control_entry<DEGREE>(prev, gxp, r.knots, e, j, S, E);
          *element(e, j) += S;
        }
    } else if (k.commit == 2) {
      int lo, hi;
      span(cur, lo, hi);
      for (int e = lo; e <= hi; ++e)
        for (int j = 0; j < NJ; ++j) {
          float S, E;
          control_entry<DEGREE>(cur, gxc, r.knots, e, j, S, E);
          *element(e, j) += S;
        }
    }
    // ---- 2. this launch's share of the block's norm slots
    if (r.norm_kind == 0 && k.mode != 3) {
      if (k.mode == 0) {
        const int n_e = (int)(DEGREE == CDE_PATH_CUBIC ? r.n_intervals : r.n_intervals + 1);
        for (int e = 0; e < n_e; ++e)
          for (int j = 0; j < NJ; ++j) {
            const float g = *element(e, j), sc = r.atol + fabsf(g) * r.rtol, u = g / sc;
            q0 += (double)(u * u);
          }
      }
      int lo, hi;
      span(cur, lo, hi);
      for (int e = lo; e <= hi; ++e)
        for (int j = 0; j < NJ; ++j) {
          float S, E;
          control_entry<DEGREE>(cur, gxc, r.knots, e, j, S, E);
          const float g = *element(e, j);
          if (k.mode == 0) { const float sc = r.atol + fabsf(g) * r.rtol, v = S / sc; q1 += (double)(v * v); }
          else if (k.mode == 1) { const float sc = r.atol + fabsf(g) * r.rtol, v = S / sc; q0 += (double)(v * v); }
          else { const float tol = r.atol + r.rtol * fmaxf(fabsf(g), fabsf(g + S)), v = E / tol; q0 += (double)(v * v); }
        }
    }
  }
  // ---- the knot-time block: one wave of block 0 (batch sums of the per-stage time term -> entries -> commit -> norm slots)
  if (r.G_knots && blockIdx.x == 0 && threadIdx.x < 64) {
    const int lane = threadIdx.x;
    auto stage_sums = [&](int which, double (&KT)[7]) {
#pragma unroll
      for (int s = 0; s < 7; ++s) {
        double t = 0.0;
        for (int b = lane; b < r.n_wg; b += 64) t += r.ktp[((int64_t)which * r.kt_stride + b) * 8 + s];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off, 64);
        KT[s] = t;
      }
    };
    const AdjStageRec cur = *reinterpret_cast<const AdjStageRec*>(r.rec + parity * ADJ_REC_STRIDE);
    double KTc[7];
    stage_sums(parity, KTc);
    auto span = [&](const AdjStageRec& rc, int& lo, int& hi) {
      lo = rc.sidx[0]; hi = rc.sidx[0];
      for (int s = 1; s < rc.ns; ++s) { lo = rc.sidx[s] < lo ? rc.sidx[s] : lo; hi = rc.sidx[s] > hi ? rc.sidx[s] : hi; }
      if (DEGREE == CDE_PATH_LINEAR) hi += 1;
    };
    if (k.commit == 1) {
      const AdjStageRec prev = *reinterpret_cast<const AdjStageRec*>(r.rec + p2 * ADJ_REC_STRIDE);
      double KTp[7];
      stage_sums(p2, KTp);
      int lo, hi;
      span(prev, lo, hi);
      if (lane == 0)
        for (int e = lo; e <= hi; ++e) { float S, E; knot_entry<DEGREE>(prev, KTp, r.knots, e, S, E); r.G_knots[e] += S; }
    } else if (k.commit == 2) {
      int lo, hi;
      span(cur, lo, hi);
      if (lane == 0)
        for (int e = lo; e <= hi; ++e) { float S, E; knot_entry<DEGREE>(cur, KTc, r.knots, e, S, E); r.G_knots[e] += S; }
    }
    if (r.norm_kind == 0 && k.mode != 3) {
      double k0 = 0.0, k1 = 0.0;
      if (k.mode == 0) {
        const int n_e = (int)r.n_intervals + 1;
        for (int e = lane; e < n_e; e += 64) {
          const float g = r.G_knots[e], sc = r.atol + fabsf(g) * r.rtol, u = g / sc;
          k0 += (double)(u * u);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) k0 += __shfl_xor(k0, off, 64);
      }
      if (lane == 0) {
        int lo, hi;
        span(cur, lo, hi);
        for (int e = lo; e <= hi; ++e) {
          float S, E;
          knot_entry<DEGREE>(cur, KTc, r.knots, e, S, E);
          const float g = r.G_knots[e];
          if (k.mode == 0) { const float sc = r.atol + fabsf(g) * r.rtol, v = S / sc; k1 += (double)(v * v); }
          else if (k.mode == 1) { const float sc = r.atol + fabsf(g) * r.rtol, v = S / sc; k0 += (double)(v * v); }
          else { const float tol = r.atol + r.rtol * fmaxf(fabsf(g), fabsf(g + S)), v = E / tol; k0 += (double)(v * v); }
        }
        r.cq[((int64_t)p2 * (gridDim.x + 1) + gridDim.x) * 2] = k0;
        r.cq[((int64_t)p2 * (gridDim.x + 1) + gridDim.x) * 2 + 1] = k1;
      }
    }
  }
  if (r.norm_kind != 0 || k.mode == 3) return;
  // the block's sums (fixed order)
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { q0 += __shfl_xor(q0, off, 64); q1 += __shfl_xor(q1, off, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = q0; red[1][threadIdx.x >> 6] = q1; }
  __syncthreads();
  if (threadIdx.x < 2) {
    const int i = threadIdx.x;
    // slot p2: where the NEXT attempt launch (parity p2) looks for the sums pending on it, like the R kernel's
    r.cq[((int64_t)p2 * (gridDim.x + 1) + blockIdx.x) * 2 + i] = (red[i][0] + red[i][1]) + (red[i][2] + red[i][3]);
  }
}


}  // namespace cde
