// dopri5_adjoint.hip -- K4a: the continuous-adjoint backward of an adaptive (dopri5) solve, fused.
//
// Replaces the backward of torchdiffeq.odeint_adjoint(method='dopri5') behind reference solver.py:226 -- torchcde's
// DEFAULT call, cdeint(X, func, z0, t) with adjoint=True and no method (solver.py:144,199-203, README.md:174) -- for
// the affine vector-field family (identity / tanh), f32, H <= 32, C <= 8.  Per output interval [t_{i-1}, t_i]
// (processed last to first by the host) the augmented state (vjp_t, y, a, dL/dW, dL/db) is integrated in reversed time
// s = -t with Dormand-Prince 5(4) and torchdiffeq's batch-global step controller (semantics restated in
// oracle/odeint.py: _Dopri5 + _Adjoint; controller: cde_dopri_adj.h, float64 times, state-dtype norms).
//
// Execution model = K4's ("finish the previous attempt, start the next" per launch, no host round trip, no grid
// barrier) on the workgroup-per-tile decomposition of rk4_split.hip:
//   * a workgroup of 8 waves (4 chain + 4 helper, two per SIMD) owns 16 series at a time and walks its share of the
//     tiles; the 7 stage evaluations of an attempt (the FSAL stage is RE-EVALUATED: its dL/dW contribution needs this
//     attempt's dt, and a recomputation with the same inputs gives the same bits as the stored value would) run
//     exactly like the stages of rk4_adjoint_split8: Y tiles, f, g, va partials through LDS, one barrier per stage;
//     each lane keeps the RK bookkeeping of its two hidden units (7 stage slopes of y and of a: 28 registers);
//   * the helper waves accumulate, over ALL tiles of the workgroup and in registers, two LINEAR FUNCTIONALS of the seven
//     stage slopes of (dL/dW, dL/db) -- S: the step's increment (dt c_sol), E: its error estimate (dt c_err)
//     (cde_dopri_adj.h: adj_stage_weights) -- and leave them as per-workgroup images; `adjoint_reduce_kernel` (one small
//     launch after every attempt launch) adds the images up in a fixed order, owns the running total G, commits the
//     previous attempt's S to it when the controller accepted that attempt, and leaves per-block sums of (E / tol)^2
//     for the next launch's controller;
//   * error control = torchdiffeq's default MIXED norm over the whole augmented state,
//         max(|e_t|, rms(e_y), rms(e_a), rms(e_W), rms(e_b)),
//     or, with adjoint_options=dict(norm="seminorm"), the same without the parameter blocks.  vjp_t (a scalar; the field
//     depends on t through dX/dt(t): d f / dt = F(z) d2X/dt2) is accumulated by the chain waves next to the state sums;
//   * the solve of an interval ends as torchdiffeq's does: the last step passes the interval end and the 4th-order dense
//     interpolant is evaluated there.  The launch that accepts such a step REPEATS it (mode 3: same start state, same
//     stage times, hence the same slopes) with S := the dense-output functional: `a(s1)` in the chain waves (the
//     expression order of torchdiffeq's _interp_fit / _interp_evaluate), (dL/dW, dL/db) and vjp_t at s1 through the
//     image / the state sums, which the R kernel adds to the running totals.
// Round 2's two stated deviations (state-only norm, clipped last step) are gone.
#include "cde_dopri_adj.h"
#include "cde_dopri_ctl.h"
#include "cde_split.h"

namespace cde {

constexpr int ADJ_IMAGE = 36;                                    // per helper lane: 32 dW accumulators + 4 bias sums
constexpr int ADJ_IMAGE_FLOATS = 4 * 64 * ADJ_IMAGE;             // per workgroup and functional (A, E, D)
constexpr int ADJ_MAX_WG = 256;
constexpr int ADJ_LDS_FLOATS = 2 * SPL_ZBUF + 2 * SPL_ZT + 2 * SPL_VA + 2 * 2 * 7 * SPL_DX + 2 * 4 * SPL_GT;
constexpr size_t ADJ_LDS_BYTES = (size_t)ADJ_LDS_FLOATS * sizeof(float) + 4 * 512 * sizeof(double);
constexpr int ADJ_RBLOCKS = ADJ_IMAGE_FLOATS / 16;               // blocks of the R kernel: 16 image slots each
struct DopriAdjArgs {
  const float* coeffs; const float* knots; int64_t n_intervals;
  const float* W; const float* bias; Dims dims;
  int64_t B, n_tiles;
  unsigned char* ctrl;              // [2] AdjCtrl, ADJ_CTRL_STRIDE bytes apart
  float* state;                     // [2][4][B*H]: committed y, a; attempted y1, a1
  const float* y_init; const float* a_init;
  float* a_out;                     // a at the end of the interval (dense output of the attempt that reaches it)
  double* partial;                  // [2][ADJ_MAX_WG][ADJ_NS]
  double* pq;                       // [2][ADJ_RBLOCKS][4]: the R kernel's partial sums, slots (q0, q1) of W then of b
  float* att;                       // [ADJ_MAX_WG][2][ADJ_IMAGE_FLOATS]: this launch's S and E images
  AdjCommon com;
  const double* ext_sums;           // sharded batch: the ADJ_NS pending sums added up over all shards (else nullptr)
  // DCTRL
  float* gx;                        // [2][B][ADJ_GX_ROW]: the launch's per-stage d(a.f)/d(dX_c), unweighted
  unsigned char* rec;               // [2] AdjStageRec, ADJ_REC_STRIDE bytes apart
  const double* cq;                 // [2][n_cblocks + 1][2]: the control kernel's norm sums (last entry: the knot block)
  int n_cblocks;
  double* ktp;                      // [2][ADJ_MAX_WG][8]: per-workgroup sums over the series of the per-stage time term
                                    // (cubic: a . F d2X/dt2; linear: a . f) -- what the knot-time block is made of
  int with_knots;                   // the knot times are a block of the norm too (n_pt == 4)
};

__device__ __forceinline__ AdjCtrl* adj_ctrl(unsigned char* base, int which) {
  return reinterpret_cast<AdjCtrl*>(base + which * ADJ_CTRL_STRIDE);
}

#ifdef CDE_PHASE_TRACE
__device__ unsigned long long k4a_phase_trace[TRACE_RING * TRACE_BLOCKS * TRACE_SLOTS];
#endif

template <int DEGREE, int ACT, bool DCTRL = false>
__global__ __launch_bounds__(512, 1) void dopri5_adjoint_attempt(DopriAdjArgs g, int parity) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int p = parity, p2 = parity ^ 1;
  CDE_STAMP_DECL;
  CDE_STAMP(0);
  // Everything the launch needs before it can decide the pending attempt is requested at once, ahead of the first wait
  // (profiles/r03_phase_k4a.log: one after the other, these round trips were 19 us of a 131 us attempt): the pending
  // sums of the previous attempt launch and of its R kernel (addresses depend on the launch parity only), the controller
  // block, then -- once that is here -- the knots around the previous attempt's interval, the chain waves' weights and a
  // touch of the control rows the first tile is likely to need.
  const double* Pp = g.partial + (int64_t)p * ADJ_MAX_WG * ADJ_NS;
  const double* Qp = g.pq + (int64_t)p * ADJ_RBLOCKS * 4;
  constexpr int NQ = DCTRL ? 8 : 4;                                 // norm slots of the parameter blocks: W, b (, coefficients)
  double sum[ADJ_NS + NQ];
#pragma unroll
  for (int i = 0; i < ADJ_NS + NQ; ++i) sum[i] = 0.0;
  if (!g.ext_sums) {
    for (int64_t b = tid; b < (int64_t)gridDim.x; b += blockDim.x) {
#pragma unroll
      for (int i = 0; i < ADJ_NS; ++i) sum[i] += Pp[ADJ_NS * b + i];
    }
  }
  for (int b = tid; b < ADJ_RBLOCKS; b += blockDim.x) {
#pragma unroll
    for (int i = 0; i < 4; ++i) sum[ADJ_NS + i] += Qp[4 * b + i];
  }
  if constexpr (DCTRL) {
    const double* Cp = g.cq + (int64_t)p * (g.n_cblocks + 1) * 2;
    for (int b = tid; b < g.n_cblocks; b += blockDim.x) { sum[ADJ_NS + 4] += Cp[2 * b]; sum[ADJ_NS + 5] += Cp[2 * b + 1]; }
    if (tid == 0) { sum[ADJ_NS + 6] = Cp[2 * g.n_cblocks]; sum[ADJ_NS + 7] = Cp[2 * g.n_cblocks + 1]; }     // the knot block
  }
  AdjCtrl k = *adj_ctrl(g.ctrl, p);
  DopriCtrl& c = k.c;
  if (c.phase == 4) {
    if (blockIdx.x == 0 && tid == 0) { k.commit = 0; k.mode = 3; *adj_ctrl(g.ctrl, p2) = k; }
    return;
  }
  CDE_STAMP(1);
  const int Hr = g.dims.H, Cr = g.dims.C;
  const int lane = tid & 63, wave = tid >> 6;
  const int w = wave & 3;
  const bool helper = __builtin_amdgcn_readfirstlane(wave) >= 4;
  // The chain waves are the critical path of a stage (two dependent products, the activation and the LDS exchange
  // between them); a helper wave shares its SIMD -- and the matrix pipe -- with one of them and is ready the moment the
  // stage barrier opens.  With equal priorities its 32 (64 with the E image) MFMAs went first and the chain wave's
  // product queued behind them: every helper MFMA was on the critical path (the E image cost exactly its own MFMA
  // time per stage).  Priority 3 for the chain waves: helper MFMAs fill the pipe only while the chain wave is busy
  // with VALU / LDS work.
  if (!helper) __builtin_amdgcn_s_setprio(3);
  const int n = lane & 15, q = lane >> 4;
  float* zbuf = lds;
  float* ztb = lds + 2 * SPL_ZBUF;
  float* vab = ztb + 2 * SPL_ZT;
  float* dxb = vab + 2 * SPL_VA;                                   // [2 tiles in flight][7 stages][16 series][SPL_DXROW]
  float* d2b = dxb + 2 * 7 * SPL_DX;                               // the same for d2X/dt2 (cubic controls: vjp_t)
  float* gT = d2b + 2 * 7 * SPL_DX + w * SPL_GT;                   // + (parity) * 4 * SPL_GT
  double* red = reinterpret_cast<double*>(lds + ADJ_LDS_FLOATS);
  float* gxb = lds + ADJ_LDS_FLOATS + 4096;                         // DCTRL: [2][ADJ_GX_TILE], behind the 16 KB of `red`
  const int64_t BH = g.B * g.dims.H;
  const float* Sp = g.state + (int64_t)p * 4 * BH;
  float* Sq = g.state + (int64_t)p2 * 4 * BH;
  double* Pq = g.partial + (int64_t)p2 * ADJ_MAX_WG * ADJ_NS;
  const float rtol = (float)g.com.rtol, atol = (float)g.com.atol;
  const int phase_in = c.phase;

  // the four knots around the previous attempt's first stage time (reversed time: the field lives at t = -s)
  const KnotWindow<float> window = knot_window(g.knots, g.n_intervals, phase_in == 0 ? (int64_t)-1 : (int64_t)c.slot);
  // chain waves: weights into registers (L2 hits, but 80 scattered loads per lane whose latency used to sit between the
  // controller and the first stage)
  // JC (affine field, piecewise-linear control): the chain waves work from the Jacobian J = sum_c dX_c W_c of THEIR rows
  // (the shared-Jacobian form of rk4_split.hip) -- and dX is constant inside a knot interval, so J is formed once per
  // tile and interval (32 MFMAs) and every further stage in that interval costs the chain wave no MFMA at all: with
  // jump_t on the knots (steps never cross one) that is every stage but the first of a tile.
  constexpr bool JC = DEGREE == CDE_PATH_LINEAR && ACT == CDE_ACT_NONE && !DCTRL;
  float wy[4][8], wv[2][16];
  f32x4 by[4];
  float wj[JC ? 16 : 1][2];
  f32x2 bja[4], bjb[4];
  if (!helper) {
    if constexpr (JC) spl_load_wj(g.W, g.bias, w, n, q, g.dims, wj, bja, bjb);
    else {
      spl_load_wy(g.W, g.bias, w, n, q, g.dims, wy, by);
      spl_load_wv(g.W, w, n, q, g.dims, wv);
    }
  }
  // helper wave 0: one dword of the first and the last 16 bytes of the control rows the first tile will most likely read
  // (the previous attempt's interval, or the one before it in reversed time) pulls their lines from HBM into this XCD's L2
  // while the sums are reduced and the controller runs
  float touched = 0.f;
  if (helper && w == 0 && phase_in == 3 && (int64_t)blockIdx.x < g.n_tiles) {
    const int64_t series = (int64_t)blockIdx.x * 16 + n;
    const int64_t sc = series < g.B ? series : g.B - 1;
    const int cand = c.slot - (q >> 1);
    if (cand >= 0 && cand < g.n_intervals) {
      const float* rowp = DEGREE == CDE_PATH_CUBIC ? g.coeffs + ((sc * g.n_intervals + cand) * 4 + 1) * Cr
                                                   : g.coeffs + (sc * (g.n_intervals + 1) + cand) * Cr;
      touched = *reinterpret_cast<const volatile float*>(rowp + ((q & 1) ? (DEGREE == CDE_PATH_CUBIC ? 3 : 2) * Cr - 1 : 0));
    }
  }

  // ---- pending global sums (fixed order: the decision is identical in every workgroup and run to run): the state
  // sums of the previous attempt launch and the parameter sums its R kernel left
  if (phase_in != 0) {
    block_total<ADJ_NS + NQ>(sum, red);
    if (g.ext_sums) {
#pragma unroll
      for (int i = 0; i < ADJ_NS; ++i) sum[i] = g.ext_sums[i];
    }
  }
  CDE_STAMP(2);
  asm volatile("" ::"v"(touched));
  const AdjPlan plan = adj_controller(g.com, k, sum, sum + ADJ_NS);
  const int mode = plan.mode;
  // mode 3 repeats the accepted step from ITS start state (the dense output at the interval end needs its slopes)
  const bool commit = phase_in == 3 && plan.accept && mode != 3;
  const int ns = mode == 0 ? 1 : mode == 1 ? 2 : 7;

  // ---- stage times (reversed time), their knot intervals and the stage weights, all wave-uniform
  const float t0f = (float)plan.t0, dtf = (float)plan.dt, t1f = (float)plan.t1;
  int sidx[7];
  float sfrac[7];
  {
    float ts = 0.f;
    const int i = lane & 7;
    if (mode == 0) ts = (float)c.t_hi;
    else if (mode == 1) ts = i == 0 ? (float)c.t_hi : (float)(c.t_hi + (double)plan.h0);
    else if (i == 0) ts = plan.kind0 == 0 ? t0f : next_toward(t0f, plan.kind0 > 0 ? 1.f : -1.f);
    else if (i <= 4) ts = t0f + (float)DP_ALPHA[i - 1] * dtf;
    else ts = next_toward(t1f, -1.f);
    float frac;
    // the field lives at t = -s.  Consecutive steps sit in the same or in neighbouring intervals: the interval of the
    // previous launch's stage 0 (kept in the controller block) turns the search's eight dependent global loads -- paid
    // by every launch before anything else can start -- into four independent ones
    const int idx = (int)locate_window(window, g.knots, g.n_intervals, -ts, frac);
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      sidx[j] = __builtin_amdgcn_readlane(idx, j);
      sfrac[j] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(frac), j));
    }
  }
  // the controller block of the next launch (written now: 36 registers the stages do not have to carry)
  if (blockIdx.x == 0 && tid == 0) {
    c.phase = mode == 0 ? 1 : mode == 1 ? 2 : mode == 2 ? 3 : 4;
    c.slot = sidx[0];                                              // search hint for the next launch's stage times
    *adj_ctrl(g.ctrl, p2) = k;
  }
#ifdef CDE_PHASE_TRACE
  const int attempt_no = uni((int)(c.n_accept + c.n_reject));
#endif
  // bc[i][j]: weight of slope j in the state handed to stage i.  Everything here is wave-uniform but derives from
  // memory / LDS reads, so the compiler would hold it in vector registers (70 of them): read back through lane 0
  float bc[7][6];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
#pragma unroll
    for (int j = 0; j < 6; ++j) bc[i][j] = (i >= 1 && j < i) ? uni((float)DP_BETA[i - 1][j] * dtf) : 0.f;
  }
  if (mode == 1) bc[1][0] = uni(plan.h0);
  // the two functionals of the stage slopes this launch accumulates (parameter gradients in the helper waves, vjp_t
  // in the chain waves): S and E of cde_dopri_adj.h; wE doubles as the error weights of the state
  float wS[7], wE[7];
  adj_stage_weights(mode, dtf, plan.x_end, wS, wE);
#pragma unroll
  for (int j = 0; j < 7; ++j) { wS[j] = uni(wS[j]); wE[j] = uni(wE[j]); }
  const bool img_e = mode == 2 && g.com.norm_kind == 0;            // the E image: only when the parameter blocks are in the norm
  if constexpr (DCTRL) {
    if (blockIdx.x == 0 && tid == 0) {
      AdjStageRec rc;
      rc.mode = mode; rc.ns = ns;
#pragma unroll
      for (int j = 0; j < 7; ++j) { rc.sidx[j] = sidx[j]; rc.sfrac[j] = sfrac[j]; rc.wS[j] = wS[j]; rc.wE[j] = wE[j]; }
      *reinterpret_cast<AdjStageRec*>(g.rec + p * ADJ_REC_STRIDE) = rc;
    }
  }

  int par = 0, gpar = 0, dbuf = 0;
  double acc[ADJ_NS] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  double ktd[DCTRL ? 7 : 1] = {};                                  // DCTRL with knots: the per-stage time term, summed over the series
  CDE_STAMP(3);

  if (helper) {
    // ------------------------------------------------------------------------------------------ helper wave
    f32x4 accA[4][2], accE[4][2];
    float gbA[4] = {0.f, 0.f, 0.f, 0.f}, gbE[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int Tm = 0; Tm < 4; ++Tm) {
      accA[Tm][0] = f32x4{0.f, 0.f, 0.f, 0.f}; accA[Tm][1] = accA[Tm][0];
      accE[Tm][0] = accA[Tm][0]; accE[Tm][1] = accA[Tm][0];
    }
    float* my_att = g.att + (((int64_t)blockIdx.x * 2) * 256 + w * 64 + lane) * ADJ_IMAGE;      // image S; E follows
    const float* ztr = ztb + n * SPL_TROW + 4 * q;
    const float* gr = gT + n * SPL_TROW + 4 * q;
    const int fc = 2 * w + (q & 1);                                // the control channel this lane feeds
    const bool feeds = q < 2;
    const int fcc = fc < Cr ? fc : Cr - 1;
    // control derivative of a tile at every stage time: requested one tile ahead (its global-load latency would
    // otherwise be exposed once per tile and attempt), finished and stored to dxb[buffer][stage] when that tile is next
    float raw[7][3];
    auto feed_request = [&](int64_t tile) {
      const int64_t series = tile * 16 + n;
      const int64_t sc = series < g.B ? series : g.B - 1;
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        if (i < ns) {
          if (DEGREE == CDE_PATH_CUBIC) {
            const float* pr = g.coeffs + ((sc * g.n_intervals + sidx[i]) * 4 + 1) * Cr + fcc;
            raw[i][0] = pr[0]; raw[i][1] = pr[Cr]; raw[i][2] = pr[2 * Cr];
          } else {
            const float* pr = g.coeffs + (sc * (g.n_intervals + 1) + sidx[i]) * Cr + fcc;
            raw[i][0] = pr[0]; raw[i][1] = pr[Cr]; raw[i][2] = g.knots[sidx[i] + 1] - g.knots[sidx[i]];
          }
        }
      }
    };
    auto feed_store = [&](int buf) {
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        if (i < ns) {
          const float v = DEGREE == CDE_PATH_CUBIC ? cubic_derivative(raw[i][0], raw[i][1], raw[i][2], sfrac[i])
                                                   : (raw[i][1] - raw[i][0]) / raw[i][2];
          if (feeds) dxb[(buf * 7 + i) * SPL_DX + n * SPL_DXROW + fc] = fc < Cr ? v : 0.f;
          if (DEGREE == CDE_PATH_CUBIC) {
            // d2X/dt2 = 2c + 2 (3d) frac: what autograd gets from interpolation_cubic.py:334-335 through frac = t - t_i
            const float v2 = raw[i][1] + 2.f * raw[i][2] * sfrac[i];
            if (feeds) d2b[(buf * 7 + i) * SPL_DX + n * SPL_DXROW + fc] = fc < Cr ? v2 : 0.f;
          }
        }
      }
    };
    if ((int64_t)blockIdx.x < g.n_tiles) { feed_request(blockIdx.x); feed_store(0); }
    // DCTRL: helper waves 0 / 1 own (series 8w + (lane >> 3), channel lane & 7): the 16 chain-wave shares of a stage, in a
    // fixed order, one stage behind like the g tile
    const bool sums_gx = DCTRL && w < 2;
    const int gx_at = (8 * w + (lane >> 3)) * 8 + (lane & 7);
    auto gx_sum = [&](int gp) {
      float t = 0.f;
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) t += gxb[gp * ADJ_GX_TILE + kk * 128 + gx_at];
      return t;
    };
#ifdef CDE_PHASE_TRACE
    int htile = -1;
#endif
    for (int64_t tile = blockIdx.x; tile < g.n_tiles; tile += gridDim.x) {
#ifdef CDE_PHASE_TRACE
      ++htile;
#endif
      const bool has_next = tile + gridDim.x < g.n_tiles;
      if (has_next) feed_request(tile + gridDim.x);
      spl_barrier();
      f32x2 zr[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};      // z of the stage whose g tile is next
      float gxs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      // one stage's contribution to the images: g^T (w z) for each functional with a non-zero weight on that stage
#ifdef CDE_PHASE_TRACE
      int hstage = -1;
#endif
      auto dw_round = [&](int gp, float a_w, float e_w) {
        if (a_w == 0.f && e_w == 0.f) return;
        float4 ga4[4];
#pragma unroll
        for (int Tm = 0; Tm < 4; ++Tm) ga4[Tm] = *reinterpret_cast<const float4*>(gr + gp * 4 * SPL_GT + Tm * 16 * SPL_TROW);
#ifdef CDE_PHASE_TRACE
        if (htile == 1 && hstage == 3) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); CDE_STAMP(21); }
#endif
        float rs[4];
#pragma unroll
        for (int Tm = 0; Tm < 4; ++Tm) rs[Tm] = (ga4[Tm].x + ga4[Tm].y) + (ga4[Tm].z + ga4[Tm].w);
        auto add = [&](float wgt, f32x4 (&accW)[4][2], float (&gb)[4]) {
          const f32x2 z0 = zr[0] * wgt, z1 = zr[1] * wgt, z2 = zr[2] * wgt, z3 = zr[3] * wgt;
          const float b0[4] = {z0[0], z0[1], z1[0], z1[1]}, b1[4] = {z2[0], z2[1], z3[0], z3[1]};
#pragma unroll
          for (int Tm = 0; Tm < 4; ++Tm) {
            const float ga[4] = {ga4[Tm].x, ga4[Tm].y, ga4[Tm].z, ga4[Tm].w};
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              accW[Tm][0] = mfma16(ga[s], b0[s], accW[Tm][0]);
              accW[Tm][1] = mfma16(ga[s], b1[s], accW[Tm][1]);
            }
            gb[Tm] = __builtin_fmaf(rs[Tm], wgt, gb[Tm]);
          }
        };
        if (a_w != 0.f) add(a_w, accA, gbA);
        if (e_w != 0.f) add(e_w, accE, gbE);
      };
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        if (i < ns) {
#ifdef CDE_PHASE_TRACE
          hstage = i;
#endif
          CDE_STAMP_IF(htile == 1 && i == 3, 20);
          const float4 zt0 = *reinterpret_cast<const float4*>(ztr + par * SPL_ZT);
          const float4 zt1 = *reinterpret_cast<const float4*>(ztr + par * SPL_ZT + 16 * SPL_TROW);
          if (i >= 1) dw_round(gpar ^ 1, wS[i - 1], img_e ? wE[i - 1] : 0.f);      // stage i-1's tile
          if constexpr (DCTRL) { if (i >= 1 && sums_gx) gxs[i - 1] = gx_sum(gpar ^ 1); }
#ifdef CDE_PHASE_TRACE
          if (htile == 1 && i == 3) { asm volatile("s_nop 0" : "+v"(accA[3][1]), "+v"(accE[3][1])); CDE_STAMP(22); }
#endif
          zr[0] = f32x2{zt0.x, zt0.y}; zr[1] = f32x2{zt0.z, zt0.w};
          zr[2] = f32x2{zt1.x, zt1.y}; zr[3] = f32x2{zt1.z, zt1.w};
          CDE_STAMP_IF(htile == 1 && i == 3, 23);
          spl_barrier();
          CDE_STAMP_IF(htile == 1 && i == 3, 24);
          par ^= 1; gpar ^= 1;
        }
      }
      // the last stage's tile: the chain waves overwrite this g buffer two barriers from now at the earliest
      if (ns == 1) dw_round(gpar ^ 1, wS[0], 0.f);
      else if (ns == 2) dw_round(gpar ^ 1, wS[1], 0.f);
      else dw_round(gpar ^ 1, wS[6], img_e ? wE[6] : 0.f);
      if constexpr (DCTRL) {
        if (sums_gx) {
          const float last = gx_sum(gpar ^ 1);
          if (ns == 1) gxs[0] = last; else if (ns == 2) gxs[1] = last; else gxs[6] = last;
          const int64_t series = tile * 16 + 8 * w + (lane >> 3);
          if (series < g.B) {
            float* dst = g.gx + ((int64_t)p * g.B + series) * ADJ_GX_ROW + (lane & 7) * 8;
            *reinterpret_cast<float4*>(dst) = make_float4(gxs[0], gxs[1], gxs[2], gxs[3]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(gxs[4], gxs[5], gxs[6], 0.f);
          }
        }
      }
      if (has_next) feed_store(dbuf ^ 1);
      dbuf ^= 1;
    }
    {
      auto put = [&](float* dst, f32x4 (&accW)[4][2], float (&gb)[4]) {
#pragma unroll
        for (int Tm = 0; Tm < 4; ++Tm) {
#pragma unroll
          for (int Tn = 0; Tn < 2; ++Tn)
            *reinterpret_cast<float4*>(dst + (Tm * 2 + Tn) * 4) = make_float4(accW[Tm][Tn][0], accW[Tm][Tn][1], accW[Tm][Tn][2], accW[Tm][Tn][3]);
        }
        *reinterpret_cast<float4*>(dst + 32) = make_float4(gb[0], gb[1], gb[2], gb[3]);
      };
      put(my_att, accA, gbA);
      if (img_e) put(my_att + ADJ_IMAGE_FLOATS, accE, gbE);
    }
  } else {
    // ------------------------------------------------------------------------------------------ chain wave
    const int ua = 8 * w + q, ub = ua + 4;
    const int pos = spl_pos(n);
    float* zw = zbuf + n * SPL_ZROW + q * 8 + 2 * w;
    const float* zr = zbuf + n * SPL_ZROW + q * 8;
    float* ztw = ztb + ua * SPL_TROW + pos;
    float* vw = vab + (q * 16 + n) * SPL_VROW + 2 * w;
    const float* vr = vab + ((w * 4 + q) * 16 + n) * SPL_VROW;
    float* gw_ = gT + (4 * q) * SPL_TROW + pos;
    const float* dxr = dxb + n * SPL_DXROW;
    const float* d2r = d2b + n * SPL_DXROW;
    auto publish = [&](int pp, float za, float zb) {
      *reinterpret_cast<float2*>(zw + pp * SPL_ZBUF) = make_float2(za, zb);
      ztw[pp * SPL_ZT] = za;
      ztw[pp * SPL_ZT + 4 * SPL_TROW] = zb;
    };
    auto read_ka = [&](int pp, float& kaa, float& kab) {
      const float4 p03 = *reinterpret_cast<const float4*>(vr + pp * SPL_VA);
      const float4 p47 = *reinterpret_cast<const float4*>(vr + pp * SPL_VA + 4);
      kaa = (p03.x + p03.z) + (p47.x + p47.z);
      kab = (p03.y + p03.w) + (p47.y + p47.w);
    };
    // the committed state of this lane's two units, requested one tile ahead
    auto state_request = [&](int64_t tile, float (&st)[4]) {
      const int64_t series = tile * 16 + n;
      const int64_t sc = series < g.B ? series : g.B - 1;
      const int64_t ea = sc * Hr + (ua < Hr ? ua : 0), eb = sc * Hr + (ub < Hr ? ub : 0);
      if (phase_in == 0) {
        st[0] = g.y_init[ea]; st[1] = g.y_init[eb]; st[2] = g.a_init[ea]; st[3] = g.a_init[eb];
      } else {
        const int off = commit ? 2 : 0;                            // (mode 3: the accepted step's own start state)
        st[0] = Sp[(off + 0) * BH + ea]; st[1] = Sp[(off + 0) * BH + eb];
        st[2] = Sp[(off + 1) * BH + ea]; st[3] = Sp[(off + 1) * BH + eb];
      }
    };
    float st_next[4] = {0.f, 0.f, 0.f, 0.f};
    if ((int64_t)blockIdx.x < g.n_tiles) state_request(blockIdx.x, st_next);
    float vtS = 0.f, vtE = 0.f;                                   // this lane's share of the vjp_t functionals
    float ktv[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};           // DCTRL: ... and of the unweighted per-stage time term
    CDE_STAMP(4);
#ifdef CDE_PHASE_TRACE
    int tile_no = 0;
#endif
    for (int64_t tile = blockIdx.x; tile < g.n_tiles; tile += gridDim.x) {
      const int64_t series = tile * 16 + n;
      const bool valid = series < g.B;
      const int64_t sc = valid ? series : g.B - 1;
      const bool ona = valid && ua < Hr, onb = valid && ub < Hr;
      const int64_t ea = sc * Hr + (ua < Hr ? ua : 0), eb = sc * Hr + (ub < Hr ? ub : 0);
      float y0a = st_next[0], y0b = st_next[1], a0a = st_next[2], a0b = st_next[3];
      if (tile + gridDim.x < g.n_tiles) state_request(tile + gridDim.x, st_next);
      if (ua >= Hr) { y0a = 0.f; a0a = 0.f; }
      if (ub >= Hr) { y0b = 0.f; a0b = 0.f; }
      if (!valid) { a0a = 0.f; a0b = 0.f; }                      // padded lanes add nothing to dL/dW
      if (mode != 3) {
        if (ona) { Sq[0 * BH + ea] = y0a; Sq[1 * BH + ea] = a0a; }
        if (onb) { Sq[0 * BH + eb] = y0b; Sq[1 * BH + eb] = a0b; }
      }
      publish(par, y0a, y0b);
      spl_barrier();
      float kya[7], kyb[7], kaa[7], kab[7];
      float ysa = y0a, ysb = y0b, asa = a0a, asb = a0b;          // state handed to the current stage
      f32x4 Jr[JC ? 16 : 1];                                     // JC: this wave's Jacobian rows for the tile's series ...
      int jc_idx = -1;                                           // ... valid for this knot interval
      float cba = 0.f, cbb = 0.f;
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        if (i < ns) {
          CDE_STAMP_IF(tile_no == 1 && i == 3, 15);
          const float4 z03 = *reinterpret_cast<const float4*>(zr + par * SPL_ZBUF);
          const float4 z47 = *reinterpret_cast<const float4*>(zr + par * SPL_ZBUF + 4);
          const float4 d03 = *reinterpret_cast<const float4*>(dxr + (dbuf * 7 + i) * SPL_DX);
          const float4 d47 = *reinterpret_cast<const float4*>(dxr + (dbuf * 7 + i) * SPL_DX + 4);
          if (i >= 1) {
            // a path: the slope stage i-1 left open, then the state of this stage
            read_ka(par, kaa[i - 1], kab[i - 1]);
            float sa = 0.f, sb = 0.f;
#pragma unroll
            for (int j = 0; j < i; ++j) { sa = __builtin_fmaf(bc[i][j], kaa[j], sa); sb = __builtin_fmaf(bc[i][j], kab[j], sb); }
            asa = a0a + sa; asb = a0b + sb;
          }
          const float zs[8] = {z03.x, z03.y, z03.z, z03.w, z47.x, z47.y, z47.z, z47.w};
          const float dX[MC] = {d03.x, d03.y, d03.z, d03.w, d47.x, d47.y, d47.z, d47.w};
          f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
          if constexpr (JC) {
            float* gwp = gw_ + gpar * 4 * SPL_GT;
            // ---- g = a (x) dX for the helper waves' images (the lane's two units, all channels)
#pragma unroll
            for (int T = 0; T < 4; ++T) {
              const float aown = (T >> 1) ? asb : asa;
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                const f32x2 gq2 = f32x2{dX[4 * (T & 1) + 2 * j], dX[4 * (T & 1) + 2 * j + 1]} * aown;
                gwp[(T * 16 + 2 * j) * SPL_TROW] = gq2[0];
                gwp[(T * 16 + 2 * j + 1) * SPL_TROW] = gq2[1];
              }
            }
            // ---- the Jacobian rows of this wave for the tile's series: new only when the knot interval changed
            if (sidx[i] != jc_idx) {
              jc_idx = sidx[i];
              const float bq0 = q == 0 ? dX[0] : q == 1 ? dX[1] : q == 2 ? dX[2] : dX[3];
              const float bq1 = q == 0 ? dX[4] : q == 1 ? dX[5] : q == 2 ? dX[6] : dX[7];
              const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
              for (int t = 0; t < 16; ++t) {
                f32x4 jt = mfma16(wj[t][0], bq0, zero);
                Jr[t] = mfma16(wj[t][1], bq1, jt);
              }
              f32x2 fb2 = bja[0] * f32x2{dX[0], dX[1]}, fb3 = bjb[0] * f32x2{dX[0], dX[1]};
#pragma unroll
              for (int j = 1; j < 4; ++j) {
                fb2 = __builtin_elementwise_fma(bja[j], f32x2{dX[2 * j], dX[2 * j + 1]}, fb2);
                fb3 = __builtin_elementwise_fma(bjb[j], f32x2{dX[2 * j], dX[2 * j + 1]}, fb3);
              }
              cba = fb2[0] + fb2[1]; cbb = fb3[0] + fb3[1];        // (b dX) of the lane's two units
            }
            // ---- a of the wave's 8 units in every lane quarter
            float a8[8];
            {
              float ea = asa, oa = asa, eb = asb, ob = asb;
              swap16s(ea, oa);
              swap16s(eb, ob);
              float e0 = ea, e2 = ea, o1 = oa, o3 = oa, e4 = eb, e6 = eb, o5 = ob, o7 = ob;
              swap32s(e0, e2); swap32s(o1, o3); swap32s(e4, e6); swap32s(o5, o7);
              a8[0] = e0; a8[1] = o1; a8[2] = e2; a8[3] = o3; a8[4] = e4; a8[5] = o5; a8[6] = e6; a8[7] = o7;
            }
            // ---- f_h = J[h][.] . z (this quarter's share) and a's slope sum_h a_h J[h][.], from the cached rows
            const f32x2 z01 = {zs[0], zs[1]}, z23 = {zs[2], zs[3]}, z45 = {zs[4], zs[5]}, z67 = {zs[6], zs[7]};
            f32x2 va01 = {0.f, 0.f}, va23 = va01, vb01 = va01, vb23 = va01;
            float p[8];
#pragma unroll
            for (int hi = 0; hi < 8; ++hi) {
              const f32x4 j0 = Jr[2 * hi], j1 = Jr[2 * hi + 1];
              const f32x2 ah = {a8[hi], a8[hi]};
              f32x2 pp = f32x2{j0[0], j0[1]} * z01;
              pp = __builtin_elementwise_fma(f32x2{j0[2], j0[3]}, z23, pp);
              pp = __builtin_elementwise_fma(f32x2{j1[0], j1[1]}, z45, pp);
              pp = __builtin_elementwise_fma(f32x2{j1[2], j1[3]}, z67, pp);
              p[hi] = pp[0] + pp[1];
              va01 = __builtin_elementwise_fma(f32x2{j0[0], j0[1]}, ah, va01);
              va23 = __builtin_elementwise_fma(f32x2{j0[2], j0[3]}, ah, va23);
              vb01 = __builtin_elementwise_fma(f32x2{j1[0], j1[1]}, ah, vb01);
              vb23 = __builtin_elementwise_fma(f32x2{j1[2], j1[3]}, ah, vb23);
            }
            v0 = f32x4{va01[0], va01[1], va23[0], va23[1]};
            v1 = f32x4{vb01[0], vb01[1], vb23[0], vb23[1]};
            swap32s(p[0], p[2]); swap32s(p[1], p[3]); swap32s(p[4], p[6]); swap32s(p[5], p[7]);
            float s0 = p[0] + p[2], s1 = p[1] + p[3], s4 = p[4] + p[6], s5 = p[5] + p[7];
            swap16s(s0, s1); swap16s(s4, s5);
            kya[i] = -((s0 + s1) + cba); kyb[i] = -((s4 + s5) + cbb);          // reverse time: dy/ds = -f
            if (i + 1 < ns) {
              float sa = 0.f, sb = 0.f;
#pragma unroll
              for (int j = 0; j <= i; ++j) { sa = __builtin_fmaf(bc[i + 1][j], kya[j], sa); sb = __builtin_fmaf(bc[i + 1][j], kyb[j], sb); }
              ysa = y0a + sa; ysb = y0b + sb;
              publish(par ^ 1, ysa, ysb);
            }
          } else {
          f32x4 yt[4] = {by[0], by[1], by[2], by[3]};
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            yt[0] = mfma16(wy[0][s], zs[s], yt[0]);
            yt[1] = mfma16(wy[1][s], zs[s], yt[1]);
            yt[2] = mfma16(wy[2][s], zs[s], yt[2]);
            yt[3] = mfma16(wy[3][s], zs[s], yt[3]);
          }
#ifdef CDE_PHASE_TRACE
          if (tile_no == 1 && i == 3) { asm volatile("s_nop 0" : "+v"(yt[0]), "+v"(yt[1]), "+v"(yt[2]), "+v"(yt[3])); CDE_STAMP(16); }
#endif
          f32x2 gq[4][2];
          f32x2 fpa = {0.f, 0.f}, fpb = {0.f, 0.f};
          f32x2 hpa = {0.f, 0.f}, hpb = {0.f, 0.f};                // the same contraction with d2X/dt2: d f / dt
          float d2X[MC];
          if (DEGREE == CDE_PATH_CUBIC) {
            const float4 e03 = *reinterpret_cast<const float4*>(d2r + (dbuf * 7 + i) * SPL_DX);
            const float4 e47 = *reinterpret_cast<const float4*>(d2r + (dbuf * 7 + i) * SPL_DX + 4);
            d2X[0] = e03.x; d2X[1] = e03.y; d2X[2] = e03.z; d2X[3] = e03.w;
            d2X[4] = e47.x; d2X[5] = e47.y; d2X[6] = e47.z; d2X[7] = e47.w;
          }
          float* gwp = gw_ + gpar * 4 * SPL_GT;
          f32x2 gx2[4];                                            // DCTRL: a_ua act(Y)_(ua, c) + a_ub act(Y)_(ub, c), channel pairs
#pragma unroll
          for (int T = 0; T < 4; ++T) {
            const float aown = (T >> 1) ? asb : asa;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const f32x2 dx = {dX[4 * (T & 1) + 2 * j], dX[4 * (T & 1) + 2 * j + 1]};
              const f32x2 t = activate2<ACT>(yt[T][2 * j], yt[T][2 * j + 1]);
              if constexpr (DCTRL) {
                if (T >> 1) gx2[(T & 1) * 2 + j] = __builtin_elementwise_fma(t, f32x2{aown, aown}, gx2[(T & 1) * 2 + j]);
                else gx2[(T & 1) * 2 + j] = t * aown;
              }
              if (T >> 1) fpb = __builtin_elementwise_fma(t, dx, fpb); else fpa = __builtin_elementwise_fma(t, dx, fpa);
              if (DEGREE == CDE_PATH_CUBIC) {
                const f32x2 d2 = {d2X[4 * (T & 1) + 2 * j], d2X[4 * (T & 1) + 2 * j + 1]};
                if (T >> 1) hpb = __builtin_elementwise_fma(t, d2, hpb); else hpa = __builtin_elementwise_fma(t, d2, hpa);
              }
              if (ACT == CDE_ACT_NONE) gq[T][j] = dx * aown;
              else gq[T][j] = (f32x2{spl_slope<ACT>(t[0]), spl_slope<ACT>(t[1])} * dx) * aown;
              gwp[(T * 16 + 2 * j) * SPL_TROW] = gq[T][j][0];
              gwp[(T * 16 + 2 * j + 1) * SPL_TROW] = gq[T][j][1];
            }
          }
          if constexpr (DCTRL) {
            float* gxw = gxb + gpar * ADJ_GX_TILE + ((w * 4 + q) * 16 + n) * 8;
            *reinterpret_cast<float4*>(gxw) = make_float4(gx2[0][0], gx2[0][1], gx2[1][0], gx2[1][1]);
            *reinterpret_cast<float4*>(gxw + 4) = make_float4(gx2[2][0], gx2[2][1], gx2[3][0], gx2[3][1]);
          }
          kya[i] = -(fpa[0] + fpa[1]); kyb[i] = -(fpb[0] + fpb[1]);          // reverse time: dy/ds = -f
          // d vjp_t / ds = + a . (df/dt) with the stage value of a (padded lanes carry a == 0)
          if (DEGREE == CDE_PATH_CUBIC) {
            const float kt = asa * (hpa[0] + hpa[1]) + asb * (hpb[0] + hpb[1]);
            vtS = __builtin_fmaf(wS[i], kt, vtS); vtE = __builtin_fmaf(wE[i], kt, vtE);
            if constexpr (DCTRL) ktv[i] += kt;
          } else if constexpr (DCTRL) {
            // piecewise-linear control: the knot times act through the widths; per stage sum_c gx_c dX_c = a . f
            ktv[i] += asa * (fpa[0] + fpa[1]) + asb * (fpb[0] + fpb[1]);
          }
          if (i + 1 < ns) {
            // y path: the state of the next stage does not wait for anything else
            float sa = 0.f, sb = 0.f;
#pragma unroll
            for (int j = 0; j <= i; ++j) { sa = __builtin_fmaf(bc[i + 1][j], kya[j], sa); sb = __builtin_fmaf(bc[i + 1][j], kyb[j], sb); }
            ysa = y0a + sa; ysb = y0b + sb;
            publish(par ^ 1, ysa, ysb);
          }
          CDE_STAMP_IF(tile_no == 1 && i == 3, 17);
#pragma unroll
          for (int sp = 0; sp < 16; ++sp) {
            const float gv = gq[2 * (sp >> 3) + ((sp >> 2) & 1)][(sp >> 1) & 1][sp & 1];
            v0 = mfma16(wv[0][sp], gv, v0);
            v1 = mfma16(wv[1][sp], gv, v1);
          }
#ifdef CDE_PHASE_TRACE
          if (tile_no == 1 && i == 3) { asm volatile("s_nop 0" : "+v"(v0), "+v"(v1)); CDE_STAMP(18); }
#endif
          }
          float* vwp = vw + (par ^ 1) * SPL_VA;
          *reinterpret_cast<float2*>(vwp) = make_float2(v0[0], v0[1]);
          *reinterpret_cast<float2*>(vwp + 64 * SPL_VROW) = make_float2(v0[2], v0[3]);
          *reinterpret_cast<float2*>(vwp + 2 * 64 * SPL_VROW) = make_float2(v1[0], v1[1]);
          *reinterpret_cast<float2*>(vwp + 3 * 64 * SPL_VROW) = make_float2(v1[2], v1[3]);
          spl_barrier();
          CDE_STAMP_IF(tile_no == 1 && i == 3, 19);
          par ^= 1; gpar ^= 1;
        }
      }
      {                                                          // the slope the last stage left open
        float ra, rb;
        read_ka(par, ra, rb);
        if (ns == 1) { kaa[0] = ra; kab[0] = rb; } else if (ns == 2) { kaa[1] = ra; kab[1] = rb; } else { kaa[6] = ra; kab[6] = rb; }
      }
      // ---- what this launch owes the controller
      const float sca = atol + fabsf(y0a) * rtol, scb = atol + fabsf(y0b) * rtol;      // Hairer's scale
      const float saa = atol + fabsf(a0a) * rtol, sab = atol + fabsf(a0b) * rtol;
      auto sq = [](float v) { return (double)(v * v); };
      if (mode == 0) {
        if (ona) { acc[0] += sq(y0a / sca); acc[1] += sq(a0a / saa); acc[2] += sq(kya[0] / sca); acc[3] += sq(kaa[0] / saa); }
        if (onb) { acc[0] += sq(y0b / scb); acc[1] += sq(a0b / sab); acc[2] += sq(kyb[0] / scb); acc[3] += sq(kab[0] / sab); }
      } else if (mode == 1) {
        if (ona) { acc[0] += sq((kya[1] - kya[0]) / sca); acc[1] += sq((kaa[1] - kaa[0]) / saa); }
        if (onb) { acc[0] += sq((kyb[1] - kyb[0]) / scb); acc[1] += sq((kab[1] - kab[0]) / sab); }
      } else if (mode == 2) {
        // the step: y1 / a1 are the states handed to stage 6 (FSAL row == solution weights)
        float eya = 0.f, eyb = 0.f, eaa = 0.f, eab = 0.f;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          eya = __builtin_fmaf(wE[j], kya[j], eya); eyb = __builtin_fmaf(wE[j], kyb[j], eyb);
          eaa = __builtin_fmaf(wE[j], kaa[j], eaa); eab = __builtin_fmaf(wE[j], kab[j], eab);
        }
        const float tya = atol + rtol * fmaxf(fabsf(y0a), fabsf(ysa)), tyb = atol + rtol * fmaxf(fabsf(y0b), fabsf(ysb));
        const float taa = atol + rtol * fmaxf(fabsf(a0a), fabsf(asa)), tab = atol + rtol * fmaxf(fabsf(a0b), fabsf(asb));
        if (ona) { acc[0] += sq(eya / tya); acc[1] += sq(eaa / taa); Sq[2 * BH + ea] = ysa; Sq[3 * BH + ea] = asa; }
        if (onb) { acc[0] += sq(eyb / tyb); acc[1] += sq(eab / tab); Sq[2 * BH + eb] = ysb; Sq[3 * BH + eb] = asb; }
      } else {
        {
          // the accepted step that reached the interval end, once more: a(s1) by torchdiffeq's dense output
          // (_interp_fit / _interp_evaluate, oracle/odeint.py: _fit_dense / _eval_dense, same expression order)
          float ma = 0.f, mb = 0.f;
#pragma unroll
          for (int j = 0; j < 7; ++j) {
            const float wm = dtf * (float)DP_CMID[j];
            ma = __builtin_fmaf(wm, kaa[j], ma); mb = __builtin_fmaf(wm, kab[j], mb);
          }
          auto dense = [&](float y0, float y1, float f0, float f1, float mid) {
            const float ym = y0 + mid;
            const float ca = 2.f * dtf * (f1 - f0) - 8.f * (y1 + y0) + 16.f * ym;
            const float cb = dtf * (5.f * f0 - 3.f * f1) + 18.f * y0 + 14.f * y1 - 32.f * ym;
            const float cc = dtf * (f1 - 4.f * f0) - 11.f * y0 - 5.f * y1 + 16.f * ym;
            const float cd = dtf * f0;
            const float x = plan.x_end;
            float total = y0 + x * cd;
            float xp = x;
            xp = xp * x; total = total + xp * cc;
            xp = xp * x; total = total + xp * cb;
            xp = xp * x; total = total + xp * ca;
            return total;
          };
          if (ona) g.a_out[ea] = dense(a0a, asa, kaa[0], kaa[6], ma);
          if (onb) g.a_out[eb] = dense(a0b, asb, kab[0], kab[6], mb);
        }
      }
      dbuf ^= 1;
#ifdef CDE_PHASE_TRACE
      if (tile_no < 8) { __builtin_amdgcn_sched_barrier(0); stamps_.t[5 + tile_no] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); }
      ++tile_no;
#endif
    }
    acc[4] = (double)vtS; acc[5] = (double)vtE;
    if constexpr (DCTRL) {
      if (g.with_knots) {
#pragma unroll
        for (int j = 0; j < 7; ++j) ktd[j] = (double)ktv[j];
      }
    }
  }
  CDE_STAMP(13);
  // ---- publish this launch's partial sums and the controller state for the next launch
  block_total<ADJ_NS>(acc, red);
  CDE_STAMP(14);
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < ADJ_NS; ++i) Pq[ADJ_NS * blockIdx.x + i] = acc[i];
  }
  if constexpr (DCTRL) {
    if (g.with_knots) {
      block_total<7>(ktd, red);
      if (tid == 0) {
        double* dst = g.ktp + ((int64_t)p * ADJ_MAX_WG + blockIdx.x) * 8;
#pragma unroll
        for (int j = 0; j < 7; ++j) dst[j] = ktd[j];
      }
    }
  }
  CDE_STAMP_FLUSH(k4a_phase_trace, attempt_no);
  CDE_STAMP_FLUSH2(k4a_phase_trace, attempt_no, 256);
}

// ------------------------------------------------------------------------------------------ the R kernel
// One thread per slot of the helper-wave image layout (coalesced sums over the workgroups' images, fixed order).
// Slot (lane l = (w, q, n), register s): s < 32 is a dL/dW element (each exactly once); s = 32 + Tm is lane (q, n)'s
// share of the bias row sum of (h = 8w + 4(Tm>>1) + (n>>2), c = 4(Tm&1) + (n&3)): the q == 0 lane adds the four shares.
//   stage 0 (fused) : sums -> commit -> norms        (unsharded solves: one launch per attempt)
//   stage 1         : sums only, into `sums_out`     (sharded: the host all-reduces them over the ranks ...)
//   stage 2         : commit + norms from `sums_in`  (... and every rank then holds the same reduced images)
struct AdjReduceArgs {
  unsigned char* ctrl; const float* att; int n_wg;
  float* G;                 // [IMAGE]: running total (sharded: of the GLOBAL batch -- the norm needs that)
  float* G_local;           // sharded only: this shard's own running total (what the caller gets back); else nullptr
  float* prevS;             // [2][IMAGE]: the S sums of a launch, kept for the commit one launch later
  float* prevS_local;       // the same for this shard's own sums (sharded)
  double* pq;               // [2][ADJ_RBLOCKS][4]
  const double* partial;    // [2][ADJ_MAX_WG][ADJ_NS]: the attempt launches' state sums (vjp_t at the interval end)
  double* carry;
  double* sums_out;         // stage 1: [2][IMAGE] doubles (S, E)
  const double* sums_in;    // stage 2: the reduced buffer (ADJ_NS state sums, then the images)
  float rtol, atol;
};

// Launch shape: 16 image slots per block x 16 "parts"; part p adds the images of workgroups p, p + 16, .. (the loads of a
// thread are independent of each other -- a single thread walking all 256 images was a 256-deep chain of L2 latencies,
// 250 us per attempted step), the 16 partial sums of a slot meet in LDS and are added in a fixed order.
__global__ __launch_bounds__(256) void adjoint_reduce_kernel(AdjReduceArgs r, int parity, int stage) {
  __shared__ float psum[2][16][17];
  __shared__ double red[4][16];
  const int p2 = parity ^ 1;
  const AdjCtrl k = *adj_ctrl(r.ctrl, p2);                          // written by the attempt launch just before this one
  if (k.c.phase == 4 && k.commit == 0) return;                     // the interval was finished (and committed) earlier
  const int slot = threadIdx.x & 15, part = threadIdx.x >> 4;
  const int j = blockIdx.x * 16 + slot;                            // image slot
  const int s = j % ADJ_IMAGE, l = j / ADJ_IMAGE, q = (l >> 4) & 3;
  const bool bias_share = s >= 32;
  const bool owner = !bias_share || q == 0;
  const bool need_e = k.mode == 2;
  if (stage != 2) {
    float a = 0.f, e = 0.f;
    if (owner) {
      const int reps = bias_share ? 4 : 1;
      for (int rep = 0; rep < reps; ++rep) {
        const float* img = r.att + j + rep * 16 * ADJ_IMAGE;        // lanes q = 1, 2, 3 of the same (w, n)
        // (n_wg <= 256: at most 16 images per part, all loads in flight at once)
        float av[16], ev[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int b = part + 16 * u;
          const bool on = b < r.n_wg;
          av[u] = on ? img[(int64_t)b * 2 * ADJ_IMAGE_FLOATS] : 0.f;
          ev[u] = on && need_e ? img[(int64_t)b * 2 * ADJ_IMAGE_FLOATS + ADJ_IMAGE_FLOATS] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) { a += av[u]; e += ev[u]; }
      }
    }
    psum[0][slot][part] = a; psum[1][slot][part] = e;
  }
  __syncthreads();
  double qv[4] = {0.0, 0.0, 0.0, 0.0};
  if (part == 0) {
    float S = 0.f, E = 0.f;
    if (stage == 2) { S = (float)r.sums_in[ADJ_NS + j]; E = (float)r.sums_in[ADJ_NS + ADJ_IMAGE_FLOATS + j]; }
    else
      for (int pp = 0; pp < 16; ++pp) { S += psum[0][slot][pp]; E += psum[1][slot][pp]; }
    if (stage == 1) {
      r.sums_out[j] = S; r.sums_out[ADJ_IMAGE_FLOATS + j] = E;
      r.prevS_local[parity * ADJ_IMAGE_FLOATS + j] = S;             // this shard's own increment, for its own running total
    } else {
      if (k.mode == 3 && j == 0) {
        // vjp_t at the interval end: its committed value at the step's start + the dense-output functional of its slopes
        double vt = 0.0;
        if (stage == 2) vt = r.sums_in[4];
        else
          for (int b = 0; b < r.n_wg; ++b) vt += r.partial[((int64_t)p2 * ADJ_MAX_WG + b) * ADJ_NS + 4];
        r.carry[0] = (double)((float)k.T + (float)vt);
      }
      double q0 = 0.0, q1 = 0.0;
      const float gn = adj_param_element(k, r.rtol, r.atol, r.G[j], r.prevS[p2 * ADJ_IMAGE_FLOATS + j], S, E, q0, q1);
      if (k.commit) r.G[j] = gn;
      if (r.G_local && k.commit)
        r.G_local[j] += k.commit == 1 ? r.prevS_local[p2 * ADJ_IMAGE_FLOATS + j] : r.prevS_local[parity * ADJ_IMAGE_FLOATS + j];
      r.prevS[parity * ADJ_IMAGE_FLOATS + j] = S;
      if (owner) { qv[bias_share ? 2 : 0] = q0; qv[bias_share ? 3 : 1] = q1; }
    }
  }
  if (stage == 1 || k.mode == 3) return;
  // the block's sums over its 16 slots (fixed order)
  if (part == 0)
#pragma unroll
    for (int i = 0; i < 4; ++i) red[i][slot] = qv[i];
  __syncthreads();
  if (threadIdx.x < 4) {
    const int i = threadIdx.x;
    double t = 0.0;
    for (int sl = 0; sl < 16; ++sl) t += red[i][sl];
    // slot p2: where the NEXT attempt launch (parity p2) looks for the sums pending on it, like the state sums
    r.pq[((int64_t)p2 * ADJ_RBLOCKS + blockIdx.x) * 4 + i] = t;
  }
}

// dL/dW, dL/db from the running total in image layout.  Image of helper wave w, lane (n = l & 15, q = l >> 4): register
// (Tm*2 + Tn)*4 + r = dW[h = 8w + 4(Tm>>1) + q][c = 4(Tm&1) + r][k = 16 Tn + n]; register 32 + Tm of the q == 0 lane =
// the bias gradient of h = 8w + 4(Tm>>1) + (n>>2), c = 4(Tm&1) + (n&3).
__global__ __launch_bounds__(256) void dopri5_adjoint_finish_kernel(const float* __restrict__ G, float* __restrict__ grad_W,
                                                                    float* __restrict__ grad_b, Dims d) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  const int n_w = d.H * d.C * d.H, n_b = d.H * d.C;
  if (id >= n_w + n_b) return;
  if (id < n_w) {
    const int kk = id % d.H, hc = id / d.H, c = hc % d.C, h = hc / d.C;
    const int w = h >> 3, Tm = 2 * ((h >> 2) & 1) + (c >> 2), q = h & 3, r = c & 3, Tn = kk >> 4, n = kk & 15;
    grad_W[id] = G[(w * 64 + q * 16 + n) * ADJ_IMAGE + (Tm * 2 + Tn) * 4 + r];
  } else {
    const int hc = id - n_w, c = hc % d.C, h = hc / d.C;
    const int w = h >> 3, Tm = 2 * ((h >> 2) & 1) + (c >> 2), n = (h & 3) * 4 + (c & 3);
    grad_b[hc] = G[(w * 64 + n) * ADJ_IMAGE + 32 + Tm];
  }
}

static inline size_t a256(size_t x) { return (x + 255) / 256 * 256; }
static inline int adj_grid(int64_t B) { const int64_t t = (B + 15) / 16; return (int)(t < ADJ_MAX_WG ? t : ADJ_MAX_WG); }

}  // namespace cde

#ifdef CDE_PHASE_TRACE
// debug builds only (cde_common.h, "phase trace"): the stamp ring of dopri5_adjoint_attempt, [ring][workgroup][slot]
extern "C" int cde_debug_k4a_phase_trace(void* host_out, size_t bytes) {
  if (bytes > sizeof(unsigned long long) * cde::TRACE_RING * cde::TRACE_BLOCKS * cde::TRACE_SLOTS) return CDE_ERR_SHAPE;
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(cde::k4a_phase_trace), bytes) == hipSuccess ? CDE_OK : CDE_ERR_LAUNCH;
}
#endif

// ================================================================================================ C ABI
// workspace: [ctrl x2][state sums][parameter sums][carry][state 2x4xBxH][G][G_local][prev A, D (x2, global + local)]
//            [attempt images][reduced-sum scratch][trace]
namespace {
struct AdjLayout {
  size_t partial, pq, carry, state, G, G_local, prev, att, trace, trace_all, total;
  size_t rec, cq, ktp, gx, total_dcontrol;
  int n_cblocks;
};
AdjLayout adj_layout(int64_t B, int64_t H) {
  using namespace cde;
  AdjLayout L;
  L.partial = a256(2 * ADJ_CTRL_STRIDE);
  L.pq = L.partial + a256((size_t)2 * ADJ_MAX_WG * ADJ_NS * sizeof(double));
  L.carry = L.pq + a256((size_t)2 * ADJ_RBLOCKS * 4 * sizeof(double));
  L.state = L.carry + 256;
  L.G = L.state + a256((size_t)2 * 4 * B * H * sizeof(float));
  L.G_local = L.G + a256((size_t)ADJ_IMAGE_FLOATS * sizeof(float));
  L.prev = L.G_local + a256((size_t)ADJ_IMAGE_FLOATS * sizeof(float));
  L.att = L.prev + a256((size_t)4 * ADJ_IMAGE_FLOATS * sizeof(float));        // prevS, prevS_local, each [2]
  L.trace = L.att + a256((size_t)ADJ_MAX_WG * 2 * ADJ_IMAGE_FLOATS * sizeof(float));
  L.trace_all = L.trace + a256((size_t)CDE_DOPRI5_TRACE_STEPS * 3 * sizeof(double));
  L.total = L.trace_all + a256((size_t)ADJ_TRACE_ATTEMPTS * 5 * sizeof(double));
  // control gradients (cde_dopri5_adjoint_advance_dcontrol): behind everything else, so the plain layout is a prefix
  L.n_cblocks = (int)((B * 8 + 255) / 256);
  L.rec = L.total;
  L.cq = L.rec + a256(2 * ADJ_REC_STRIDE);
  L.ktp = L.cq + a256((size_t)2 * (L.n_cblocks + 1) * 2 * sizeof(double));
  L.gx = L.ktp + a256((size_t)2 * ADJ_MAX_WG * 8 * sizeof(double));
  L.total_dcontrol = L.gx + a256((size_t)2 * B * ADJ_GX_ROW * sizeof(float));
  return L;
}
}  // namespace

extern "C" size_t cde_dopri5_adjoint_trace_offset(int64_t B, int64_t C, int64_t H) {
  (void)C;
  return adj_layout(B, H).trace;
}
extern "C" size_t cde_dopri5_adjoint_workspace_bytes(int64_t B, int64_t C, int64_t H) {
  (void)C;
  return adj_layout(B, H).total;
}
extern "C" size_t cde_dopri5_adjoint_attempt_trace_offset(int64_t B, int64_t C, int64_t H) {
  (void)C;
  return adj_layout(B, H).trace_all;
}
extern "C" size_t cde_dopri5_adjoint_status_stride(void) { return cde::ADJ_CTRL_STRIDE; }
extern "C" size_t cde_dopri5_adjoint_carry_offset(int64_t B, int64_t C, int64_t H) {
  (void)C;
  return adj_layout(B, H).carry;
}
extern "C" size_t cde_dopri5_adjoint_reduced_count(void) { return (size_t)cde::ADJ_NS + 2 * (size_t)cde::ADJ_IMAGE_FLOATS; }

static cde::AdjReduceArgs adj_reduce_args(unsigned char* base, const AdjLayout& L, int64_t B, double rtol, double atol,
                                          bool sharded) {
  using namespace cde;
  AdjReduceArgs r;
  r.ctrl = base; r.att = (const float*)(base + L.att); r.n_wg = adj_grid(B);
  r.G = (float*)(base + L.G);
  r.G_local = sharded ? (float*)(base + L.G_local) : nullptr;
  float* prev = (float*)(base + L.prev);
  r.prevS = prev; r.prevS_local = prev + 2 * ADJ_IMAGE_FLOATS;
  r.pq = (double*)(base + L.pq);
  r.partial = (const double*)(base + L.partial);
  r.carry = (double*)(base + L.carry);
  r.sums_out = nullptr; r.sums_in = nullptr;
  r.rtol = (float)rtol; r.atol = (float)atol;
  return r;
}

extern "C" size_t cde_dopri5_adjoint_dcontrol_workspace_bytes(int64_t B, int64_t C, int64_t H) {
  (void)C;
  return adj_layout(B, H).total_dcontrol;
}

static int adjoint_advance(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                           const void* bias, int act, const void* y_init, const void* a_init, double s0, double s1,
                           const double* jump_s, int64_t n_jump, double rtol, double atol, double safety, double ifactor,
                           double dfactor, int norm_kind, void* a_out, int64_t B, int64_t C, int64_t H, int dtype,
                           int first_interval, void* workspace, size_t workspace_bytes, int64_t first_launch,
                           int64_t n_launches, const double* reduced_sums, int64_t B_global, void* grad_coeffs,
                           int64_t control_numel, void* grad_knots, void* stream) {
  if (B < 1 || C < 1 || H < 1 || n_intervals < 1 || n_launches < 0 || n_jump < 0 || !(s0 < s1)) return CDE_ERR_SHAPE;
  if (dtype != CDE_F32) return dtype == CDE_F64 ? CDE_ERR_UNSUPPORTED : CDE_ERR_DTYPE;
  if (H > cde::MH || C > cde::MC) return CDE_ERR_UNSUPPORTED;
  if (act != CDE_ACT_NONE && act != CDE_ACT_TANH) return CDE_ERR_UNSUPPORTED;
  if (degree != CDE_PATH_CUBIC && degree != CDE_PATH_LINEAR) return CDE_ERR_UNSUPPORTED;
  if (norm_kind != 0 && norm_kind != 1) return CDE_ERR_UNSUPPORTED;
  if (!coeffs || !knots || !W || !bias || !y_init || !a_init || !a_out || !workspace) return CDE_ERR_NULL;
  if (n_jump > 0 && !jump_s) return CDE_ERR_NULL;
  const bool dctrl = grad_coeffs != nullptr;
  if (workspace_bytes < (dctrl ? cde_dopri5_adjoint_dcontrol_workspace_bytes(B, C, H) : cde_dopri5_adjoint_workspace_bytes(B, C, H)))
    return CDE_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  unsigned char* base = (unsigned char*)workspace;
  const AdjLayout L = adj_layout(B, H);
  const bool sharded = reduced_sums != nullptr || B_global > 0;
  if (dctrl && (sharded || control_numel < 1)) return CDE_ERR_UNSUPPORTED;          // control gradients: one controller per solve
  cde::DopriAdjArgs g;
  g.coeffs = (const float*)coeffs; g.knots = (const float*)knots; g.n_intervals = n_intervals;
  g.W = (const float*)W; g.bias = (const float*)bias; g.dims = cde::Dims{(int)H, (int)C};
  g.B = B; g.n_tiles = (B + 15) / 16;
  g.ctrl = base;
  g.partial = (double*)(base + L.partial);
  g.pq = (double*)(base + L.pq);
  g.state = (float*)(base + L.state);
  g.att = (float*)(base + L.att);
  g.y_init = (const float*)y_init; g.a_init = (const float*)a_init; g.a_out = (float*)a_out;
  g.com.s0 = s0; g.com.s1 = s1; g.com.jump_s = jump_s; g.com.n_jump = n_jump;
  g.com.rtol = rtol; g.com.atol = atol; g.com.safety = safety; g.com.ifactor = ifactor; g.com.dfactor = dfactor;
  g.com.n_state = (B_global > 0 ? B_global : B) * H;
  if (grad_knots && !dctrl) return CDE_ERR_UNSUPPORTED;
  g.com.n_pt = dctrl ? (grad_knots ? 4 : 3) : 2; g.com.n_param[0] = H * C * H; g.com.n_param[1] = H * C;
  g.com.n_param[2] = dctrl ? control_numel : 1; g.com.n_param[3] = grad_knots ? n_intervals + 1 : 1;
  g.gx = (float*)(base + L.gx); g.rec = base + L.rec; g.cq = (const double*)(base + L.cq); g.n_cblocks = L.n_cblocks;
  g.ktp = (double*)(base + L.ktp); g.with_knots = grad_knots ? 1 : 0;
  g.com.norm_kind = norm_kind;
  g.com.trace = (double*)(base + L.trace);
  g.com.trace_all = (double*)(base + L.trace_all);
  g.com.carry = (double*)(base + L.carry);
  g.ext_sums = reduced_sums;
  if (sharded && (n_launches != 1 || B_global < B)) return CDE_ERR_SHAPE;      // sharded: one launch per all-reduce
  if (first_launch > 0 && sharded && !reduced_sums) return CDE_ERR_NULL;
  const int grid = cde::adj_grid(B);
  if (first_launch == 0) {
    cde::zero_async(base, 2 * cde::ADJ_CTRL_STRIDE, s);                                                // phase 0
    if (first_interval & 1) {
      // (bit 1: the caller has set vjp_t itself -- output times that require a gradient: torchdiffeq starts every interval
      //  at vjp_t - f(t_i, y_i) . dL/dy_i, cde_dopri5_adjoint_carry_offset)
      if (!(first_interval & 2)) cde::zero_async(base + L.carry, 256, s);
      cde::zero_async(base + L.G, L.att - L.G, s);                                // G, G_local, the prev buffers
    }
    if (dctrl) cde::zero_async(base + L.rec, L.gx - L.rec, s);                    // stage records, control norm sums
  }
  // sharded under "seminorm": the parameter blocks take no part in the decision, so only the 8 state sums travel between the
  // shards (cde_dopri5_adjoint_state_sums / _apply_state_sums) and the gradient images stay LOCAL -- reduced, committed and
  // returned per shard like an unsharded solve's (the caller all-reduces gradients once, as for any data-parallel step)
  const bool images_local = sharded && norm_kind == 1;
  cde::AdjReduceArgs r = adj_reduce_args(base, L, B, rtol, atol, sharded && !images_local);
  cde::AdjControlArgs cr;
  cr.ctrl = base; cr.rec = base + L.rec; cr.gx = (const float*)(base + L.gx); cr.G = (float*)grad_coeffs;
  cr.knots = (const float*)knots; cr.cq = (double*)(base + L.cq); cr.B = B; cr.n_intervals = n_intervals;
  cr.C = (int)C; cr.degree = degree; cr.norm_kind = norm_kind; cr.rtol = (float)rtol; cr.atol = (float)atol;
  cr.G_knots = (float*)grad_knots; cr.ktp = (const double*)(base + L.ktp); cr.n_wg = grid; cr.kt_stride = cde::ADJ_MAX_WG;
  const size_t lds_dc = cde::ADJ_LDS_BYTES + (size_t)2 * cde::ADJ_GX_TILE * sizeof(float);
#define CDE_ADJ(D, A)                                                                                                \
  do {                                                                                                               \
    (void)hipFuncSetAttribute((const void*)cde::dopri5_adjoint_attempt<D, A>,                                        \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)cde::ADJ_LDS_BYTES);                  \
    (void)hipFuncSetAttribute((const void*)cde::dopri5_adjoint_attempt<D, A, true>,                                  \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dc);                              \
    for (int64_t i = 0; i < n_launches; ++i) {                                                                       \
      const int parity = (int)((first_launch + i) & 1);                                                              \
      if (dctrl) cde::dopri5_adjoint_attempt<D, A, true><<<grid, 512, lds_dc, s>>>(g, parity);                       \
      else cde::dopri5_adjoint_attempt<D, A><<<grid, 512, cde::ADJ_LDS_BYTES, s>>>(g, parity);                       \
      if (!sharded || images_local) cde::adjoint_reduce_kernel<<<cde::ADJ_RBLOCKS, 256, 0, s>>>(r, parity, 0);       \
      if (dctrl) cde::adjoint_control_kernel<D, 8><<<L.n_cblocks, 256, 0, s>>>(cr, parity);                          \
    }                                                                                                                \
  } while (0)
  if (act == CDE_ACT_NONE) {
    if (degree == CDE_PATH_CUBIC) CDE_ADJ(CDE_PATH_CUBIC, CDE_ACT_NONE); else CDE_ADJ(CDE_PATH_LINEAR, CDE_ACT_NONE);
  } else {
    if (degree == CDE_PATH_CUBIC) CDE_ADJ(CDE_PATH_CUBIC, CDE_ACT_TANH); else CDE_ADJ(CDE_PATH_LINEAR, CDE_ACT_TANH);
  }
#undef CDE_ADJ
  return cde::check_launch();
}

extern "C" int cde_dopri5_adjoint_advance(const void* coeffs, const void* knots, int64_t n_intervals, int degree,
                                          const void* W, const void* bias, int act, const void* y_init,
                                          const void* a_init, double s0, double s1, const double* jump_s, int64_t n_jump,
                                          double rtol, double atol, double safety, double ifactor, double dfactor,
                                          int norm_kind, void* a_out, int64_t B, int64_t C, int64_t H, int dtype,
                                          int first_interval, void* workspace, size_t workspace_bytes,
                                          int64_t first_launch, int64_t n_launches, const double* reduced_sums,
                                          int64_t B_global, void* stream) {
  return adjoint_advance(coeffs, knots, n_intervals, degree, W, bias, act, y_init, a_init, s0, s1, jump_s, n_jump, rtol, atol,
                         safety, ifactor, dfactor, norm_kind, a_out, B, C, H, dtype, first_interval, workspace,
                         workspace_bytes, first_launch, n_launches, reduced_sums, B_global, nullptr, 0, nullptr, stream);
}

// The same with adjoint_params naming the coefficient tensor the path was built from (reference solver.py:207-222,
// README.md:251-270): `grad_coeffs` (layout of `coeffs`, zeroed by the caller before the first interval, the same tensor
// for every interval of a backward pass) receives dL/dcoeffs and is the running total of that block of torchdiffeq's
// mixed norm; `control_numel` = the element count of the tensor the caller passed in adjoint_params (the block's rms
// runs over all of them).  One controller per solve: no sharded form.
extern "C" int cde_dopri5_adjoint_advance_dcontrol(const void* coeffs, const void* knots, int64_t n_intervals, int degree,
                                                   const void* W, const void* bias, int act, const void* y_init,
                                                   const void* a_init, double s0, double s1, const double* jump_s,
                                                   int64_t n_jump, double rtol, double atol, double safety, double ifactor,
                                                   double dfactor, int norm_kind, void* a_out, int64_t B, int64_t C,
                                                   int64_t H, int dtype, int first_interval, void* workspace,
                                                   size_t workspace_bytes, int64_t first_launch, int64_t n_launches,
                                                   void* grad_coeffs, int64_t control_numel, void* grad_knots,
                                                   void* stream) {
  if (!grad_coeffs) return CDE_ERR_NULL;
  return adjoint_advance(coeffs, knots, n_intervals, degree, W, bias, act, y_init, a_init, s0, s1, jump_s, n_jump, rtol, atol,
                         safety, ifactor, dfactor, norm_kind, a_out, B, C, H, dtype, first_interval, workspace,
                         workspace_bytes, first_launch, n_launches, nullptr, 0, grad_coeffs, control_numel, grad_knots, stream);
}

// sharded batches (one controller for all shards), after the single attempt launch `total_launches - 1`: this shard's
// pending sums -- ADJ_NS state sums, then the S and E gradient images (cde_dopri5_adjoint_reduced_count() doubles) -- to
// be all-reduced (sum) over the shards and handed to cde_dopri5_adjoint_apply_reduced and to the next advance call.
__global__ __launch_bounds__(64) void dopri_adjoint_pending_sums_kernel(const double* __restrict__ partial, int n_wg,
                                                                        double* __restrict__ out) {
  const int i = threadIdx.x;
  if (i >= cde::ADJ_NS) return;
  double s = 0.0;
  for (int b = 0; b < n_wg; ++b) s += partial[cde::ADJ_NS * b + i];
  out[i] = s;
}

extern "C" int cde_dopri5_adjoint_pending_sums(void* workspace, size_t workspace_bytes, int64_t B, int64_t C,
                                               int64_t H, int64_t total_launches, double* sums, void* stream) {
  if (B < 1 || C < 1 || H < 1 || total_launches < 1) return CDE_ERR_SHAPE;
  if (!workspace || !sums) return CDE_ERR_NULL;
  if (workspace_bytes < cde_dopri5_adjoint_workspace_bytes(B, C, H)) return CDE_ERR_WORKSPACE;
  unsigned char* base = (unsigned char*)workspace;
  const AdjLayout L = adj_layout(B, H);
  const int parity = (int)((total_launches - 1) & 1);               // the launch whose sums are pending
  const double* partial = (const double*)(base + L.partial) + (int64_t)(parity ^ 1) * cde::ADJ_MAX_WG * cde::ADJ_NS;
  hipStream_t s = (hipStream_t)stream;
  dopri_adjoint_pending_sums_kernel<<<1, 64, 0, s>>>(partial, cde::adj_grid(B), sums);
  cde::AdjReduceArgs r = adj_reduce_args(base, L, B, 0.0, 0.0, true);
  r.sums_out = sums + cde::ADJ_NS;
  cde::adjoint_reduce_kernel<<<cde::ADJ_RBLOCKS, 256, 0, s>>>(r, parity, 1);
  return cde::check_launch();
}

// The "seminorm" form of the two calls: only the ADJ_NS state sums are pending on the other shards (the attempt's launch
// already ran the R kernel on this shard's own images); after the all-reduce the one number the R kernel took from LOCAL
// sums -- vjp_t at the end of an interval -- is redone from the reduced ones.
__global__ void dopri_adjoint_carry_kernel(const unsigned char* __restrict__ ctrl, int p2, const double* __restrict__ reduced,
                                           double* __restrict__ carry) {
  const cde::AdjCtrl k = *reinterpret_cast<const cde::AdjCtrl*>(ctrl + p2 * cde::ADJ_CTRL_STRIDE);
  if (k.c.phase == 4 && k.commit == 0) return;
  if (k.mode == 3) carry[0] = (double)((float)k.T + (float)reduced[4]);
}

extern "C" int cde_dopri5_adjoint_state_sums(void* workspace, size_t workspace_bytes, int64_t B, int64_t C, int64_t H,
                                             int64_t total_launches, double* sums, void* stream) {
  if (B < 1 || C < 1 || H < 1 || total_launches < 1) return CDE_ERR_SHAPE;
  if (!workspace || !sums) return CDE_ERR_NULL;
  if (workspace_bytes < cde_dopri5_adjoint_workspace_bytes(B, C, H)) return CDE_ERR_WORKSPACE;
  unsigned char* base = (unsigned char*)workspace;
  const AdjLayout L = adj_layout(B, H);
  const int parity = (int)((total_launches - 1) & 1);
  const double* partial = (const double*)(base + L.partial) + (int64_t)(parity ^ 1) * cde::ADJ_MAX_WG * cde::ADJ_NS;
  dopri_adjoint_pending_sums_kernel<<<1, 64, 0, (hipStream_t)stream>>>(partial, cde::adj_grid(B), sums);
  return cde::check_launch();
}

extern "C" int cde_dopri5_adjoint_apply_state_sums(void* workspace, size_t workspace_bytes, int64_t B, int64_t C, int64_t H,
                                                   int64_t total_launches, const double* reduced, void* stream) {
  if (B < 1 || C < 1 || H < 1 || total_launches < 1) return CDE_ERR_SHAPE;
  if (!workspace || !reduced) return CDE_ERR_NULL;
  if (workspace_bytes < cde_dopri5_adjoint_workspace_bytes(B, C, H)) return CDE_ERR_WORKSPACE;
  unsigned char* base = (unsigned char*)workspace;
  const AdjLayout L = adj_layout(B, H);
  const int parity = (int)((total_launches - 1) & 1);
  dopri_adjoint_carry_kernel<<<1, 1, 0, (hipStream_t)stream>>>(base, parity ^ 1, reduced, (double*)(base + L.carry));
  return cde::check_launch();
}

// ... and after the all-reduce: commit / norms with the reduced images (every shard holds the same ones)
extern "C" int cde_dopri5_adjoint_apply_reduced(void* workspace, size_t workspace_bytes, int64_t B, int64_t C, int64_t H,
                                                double rtol, double atol, int64_t total_launches,
                                                const double* reduced, void* stream) {
  if (B < 1 || C < 1 || H < 1 || total_launches < 1) return CDE_ERR_SHAPE;
  if (!workspace || !reduced) return CDE_ERR_NULL;
  if (workspace_bytes < cde_dopri5_adjoint_workspace_bytes(B, C, H)) return CDE_ERR_WORKSPACE;
  unsigned char* base = (unsigned char*)workspace;
  const AdjLayout L = adj_layout(B, H);
  cde::AdjReduceArgs r = adj_reduce_args(base, L, B, rtol, atol, true);
  r.sums_in = reduced;
  cde::adjoint_reduce_kernel<<<cde::ADJ_RBLOCKS, 256, 0, (hipStream_t)stream>>>(r, (int)((total_launches - 1) & 1), 2);
  return cde::check_launch();
}

extern "C" int cde_dopri5_adjoint_finish(const void* workspace, size_t workspace_bytes, void* grad_W, void* grad_b,
                                         int64_t B, int64_t C, int64_t H, int sharded, void* stream) {
  if (B < 1 || C < 1 || H < 1 || H > cde::MH || C > cde::MC) return CDE_ERR_SHAPE;
  if (!workspace || !grad_W || !grad_b) return CDE_ERR_NULL;
  if (workspace_bytes < cde_dopri5_adjoint_workspace_bytes(B, C, H)) return CDE_ERR_WORKSPACE;
  const AdjLayout L = adj_layout(B, H);
  const float* G = (const float*)((const unsigned char*)workspace + (sharded ? L.G_local : L.G));
  const int n = (int)(H * C * H + H * C);
  cde::dopri5_adjoint_finish_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(G, (float*)grad_W, (float*)grad_b,
                                                                                     cde::Dims{(int)H, (int)C});
  return cde::check_launch();
}
