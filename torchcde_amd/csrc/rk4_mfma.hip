// rk4_mfma.hip -- the headline kernels: fused RK4 (3/8) CDE solve and continuous-adjoint sweep for
//   f32 state, H = 32 hidden, C = 8 channels, affine vector field f(z) = reshape_{HxC}(W z + b)
// on the exact-f32 matrix pipe of gfx950 (v_mfma_f32_32x32x2_f32: 64 cycles issue = dependent
// latency, bitwise an fmaf chain; 157.3 TFLOP/s chip peak).
//
// Decomposition.  One wave owns 32 series for the WHOLE solve (all steps, all stages): there is no
// inter-wave communication, a workgroup (4 waves, one per SIMD) only shares the weight image in
// LDS.  Lane l = (n = l & 31, half = l >> 5) holds, for series n, the 16 hidden units
// h = 2r + half (r = 0..15) of every state vector in registers -- exactly the C/D fragment of the
// 32x32 MFMA when output rows are permuted by rho (below), and exactly what the NEXT stage's
// B operand needs, so state never moves between lanes.
//
// Per stage (all operands f32):
//   f   = W (z (x) dX) + b dX      132 MFMAs:  D[h][n] += A[h][(j,c),hk] * B[(j,c),hk][n]
//                                    A = W[(h*C+c)][2j+hk]  (LDS image, ds_read_b128 = 4 steps)
//                                    B = z_n[2j+hk] * dX_n[c] (one v_mul per MFMA, in-lane)
//   adjoint only:
//   a^T df/dz                      128 MFMAs, same shape with A = W[(2j+hk)*C+c][k]
//   dL/dW += (a (x) dX)^T z        128 MFMAs with the series index as the MFMA K dimension;
//                                    the (series -> K) transpose of z, a, dX goes through a
//                                    9.5 KB per-wave LDS scratch (stride-33 rows, conflict-free)
//   dL/db                          128 v_fma in the shadow of the MFMAs
// The interval index / fractional part of every stage time comes from the stage table written by
// stage_table_kernel (api.hip) -- the same numbers CubicSpline._interpret_t would produce.
#include "cde_mfma.h"

namespace cde {

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ void swap32(float& x, float& y) {      // x[lanes 32..63] <-> y[lanes 0..31]
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  x = __uint_as_float(r[0]);
  y = __uint_as_float(r[1]);
}

// ============================================================================================ forward
// K2.  One wave owns 16 series; lane l = (n = l & 15, q = l >> 4) keeps hidden units 8q..8q+7 of series n.
// v_mfma_f32_16x16x4_f32 (32-cycle issue, 40-cycle dependent latency): two M-tiles (hidden 32 = 2 x 16
// rows) alternate, so the pipe never waits on its own accumulator.  2 waves per SIMD (<= 256 registers):
// while one wave is in its serial RK "tail" (stage combination, next control derivative) the other wave's
// 132-MFMA chain owns the matrix pipe -- measured MFMA-busy rose from 68 % (one 32-series wave per SIMD,
// 32x32x2) to what profiles/ reports for this version.
//
//   tile T row i  <->  hidden unit 8*(i>>2) + 4*T + (i&3)   =>  acc_T[r] of lane (n,q) is unit 8q + 4T + r
//   step s = (m, c), lane quarter kq feeds K index kq = input unit 8*kq + m, channel c
//   bias steps 64, 65: lane quarter kq feeds channel 4*(s-64) + kq
//
// ACT != CDE_ACT_NONE: same skeleton on the pre-activation form (cde_mfma.h: field_act16) -- the lane then owns
// units q, 4+q, .., 28+q instead of 8q..8q+7.
// MLP: two-layer field (cde_mfma.h: field_mlp16), W1/bias1/width describe the hidden layer.
// Workgroup shape: 512 threads, 2 waves per SIMD for every form.  (Tried for the one-layer activation form, which
// is VALU-co-limited: 256-thread workgroups at 3 waves per SIMD, <= 168 registers -- 4.29 ms vs 4.0-4.2 ms, the
// spills it forces cost more than the third wave hides.)
template <int ACT, bool MLP> constexpr int fwd_block_threads() { return 512; }
template <int ACT, bool MLP> constexpr int fwd_waves_per_simd() { return 2; }

// SPLIT (two-layer field, at most one tile per CU): the workgroup's 8 waves carry ONE tile together and split layer 2 of
// every evaluation by unit group (cde_mfma.h: field_mlp16<..., SPLIT>); wave 0 alone writes the outputs.
// SAVE (affine field, the backward pass of adjoint=False: rk4_backprop.hip): the state handed to EVERY field evaluation
// is also written to `stages` (B, n_steps, 4, 32), the 32 units in K3's lane order (evens, then odds: a lane's units
// 8q .. 8q+7 are two 16-byte stores) -- what reverse-mode autograd would keep of torchdiffeq's rk4 step.
// METHOD (CDE_METHOD_*): the same skeleton for torchdiffeq's other fixed-grid methods (reference test/test_cdeint.py:49-63
// runs `midpoint`) -- two stages (midpoint) or one (euler) per step instead of the four of the 3/8 rule.
// HI (round 6): the two-layer field with 17..32 hidden units on the 16-channel layout (cde_mfma.h: field_mlp16<.., HI>)
template <typename TT, int DEGREE, int ACT, bool MLP = false, int CT = MC, bool SPLIT = false, bool SAVE = false,
          int METHOD = CDE_METHOD_RK4, bool HI = false>
__global__ __launch_bounds__((fwd_block_threads<ACT, MLP>()), (fwd_waves_per_simd<ACT, MLP>())) void rk4_forward_mfma(
    const float* __restrict__ coeffs, const float* __restrict__ knots, int64_t n_intervals,
    const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ z0,
    const TT* __restrict__ grid, int64_t n_grid, const TT* __restrict__ t_out, int64_t n_out,
    float* __restrict__ z_out, int64_t B, const int64_t* __restrict__ stage_index,
    const float* __restrict__ stage_frac, Dims dims, const float* __restrict__ W1 = nullptr,
    const float* __restrict__ bias1 = nullptr, int width = 0, float* __restrict__ stages = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // the A-operand image is loop invariant: staged once through LDS, then it lives in registers
  constexpr bool PRODUCT = ACT == CDE_ACT_NONE && !MLP;
  static_assert(!SAVE || (!MLP && CT == MC) || (MLP && !SPLIT), "stage states: the one-layer fields' kernel, the two-layer field's one-wave form");
  static_assert(METHOD == CDE_METHOD_RK4 || !SAVE, "stage states are stored for the 3/8 rule only");
  constexpr int NS = METHOD == CDE_METHOD_RK4 ? 4 : METHOD == CDE_METHOD_MIDPOINT ? 2 : 1;     // stages per step
  constexpr int STRIDE = PRODUCT ? 1 : 4;          // distance between a lane's consecutive hidden units
  float4 wA[PRODUCT ? W16_GROUPS : 1], wB[PRODUCT ? W16_GROUPS : 1];
  if constexpr (PRODUCT) load_w16(W, bias, lds, wA, wB, dims);
  else if constexpr (MLP) stage_mlp16(W1, bias1, W, bias, lds, MlpDims{dims.H, dims.C, width}, CT / 4);
  else stage_wy16(W, bias, lds, dims);
  const int Hr = dims.H, Cr = dims.C;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, q = lane >> 4;
  static_assert(!SPLIT || MLP, "the split form exists for the two-layer field");
  const int64_t tile = SPLIT ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * (fwd_block_threads<ACT, MLP>() / 64) + wave;
  if (tile * 16 >= B) return;
  // Waves w and w+4 of a 512-thread workgroup share a SIMD and do identical work: left alone they run in
  // lockstep and both sit in their MFMA-free RK tail at the same time.  Half a stage of head start for one of
  // them keeps the matrix pipe fed by the other.  (Not in the split form: its waves meet at barriers.)
  if (!SPLIT && __builtin_amdgcn_readfirstlane(wave) >= 4) __builtin_amdgcn_s_sleep(FWD_STAGGER_SLEEP);
  const int64_t series = tile * 16 + n;
  const bool in_range = series < B;
  const bool valid = in_range && (!SPLIT || wave == 0);           // who writes (every wave of a split tile LOADS its series)
  const int64_t sc = in_range ? series : B - 1;
  float* xwin = lds + MLP16_LDS_FLOATS;                           // (SPLIT only: 8 x 64 floats behind the images)
  // two-layer field with more than 16 hidden units on the 16-channel layout: unit groups 4..7 from the raw output layer
  static_assert(!HI || (MLP && CT == 16), "the upper half: two-layer field, 16-channel layout");
  const MlpHi mlp_hi = HI ? MlpHi{W, bias, dims.H, dims.C, width} : MlpHi{};

  // this lane's 8 hidden units in two groups of 4 (zero beyond the real hidden size)
  const int ua = PRODUCT ? 8 * q : q, ub = PRODUCT ? 8 * q + 4 : 16 + q;
  const float4* wy = reinterpret_cast<const float4*>(lds) + lane;              // activation form: LDS images
  const float4* by = reinterpret_cast<const float4*>(lds + WY_FLOATS) + q;
  f32x4 ya = load_units4<STRIDE>(z0 + sc * Hr, ua, Hr), yb = load_units4<STRIDE>(z0 + sc * Hr, ub, Hr);
  auto store = [&](int64_t j, const f32x4& a, const f32x4& b) {
    if (valid) {
      float* row = z_out + (series * n_out + j) * Hr;
      store_units4<STRIDE>(row, ua, Hr, a);
      store_units4<STRIDE>(row, ub, Hr, b);
    }
  };
  store(0, ya, yb);
  int64_t jout = 1;
  const int64_t n_steps = n_grid - 1;
  if (n_steps <= 0) return;

  int64_t idx = stage_index[0];
  float frac = stage_frac[0];
  Row<DEGREE, CT> row = load_row<DEGREE, CT>(coeffs, sc, n_intervals, idx, Cr);

  for (int64_t k = 0; k < n_steps; ++k) {
    TT t0 = grid[k];
    const TT t1 = grid[k + 1];
    // (euler instantiations: with both grid points in scalar registers this compiler emits v_add_f32 with two SGPR operands,
    //  "violates constant bus restriction" on gfx950 -- one of them is pinned to a vector register)
    if constexpr (METHOD == CDE_METHOD_EULER) asm volatile("" : "+v"(t0));
    const float dt = (float)(t1 - t0);
    f32x4 k1a, k1b, k2a, k2b, pqa, pqb, za = ya, zb = yb;
#pragma unroll
    for (int stage = 0; stage < NS; ++stage) {
      float dX[CT];
      const float width = DEGREE == CDE_PATH_LINEAR ? knots[idx + 1] - knots[idx] : 1.f;
      control_slope<DEGREE, CT>(row, frac, width, dX);
      // prefetch the next stage's table entry and (if the interval changes) its control row
      const int64_t e_next = stage + 1 < NS ? 4 * k + stage + 1 : 4 * (k + 1);
      const bool more = e_next < 4 * n_steps;
      const int64_t nidx = more ? stage_index[e_next] : idx;
      const float nfrac = more ? stage_frac[e_next] : frac;
      // product form: the next row is fetched before the MFMA chain (its latency hides behind it); the activation
      // forms are register-bound, so they fetch it after the field evaluation (it then lands during the RK tail of
      // this wave / the MFMA phase of the other waves on the SIMD)
      Row<DEGREE, CT> nrow = row;
      if constexpr (PRODUCT) { if (nidx != idx) nrow = load_row<DEGREE, CT>(coeffs, sc, n_intervals, nidx, Cr); }
      if constexpr (SAVE) {
        if (valid) {
          float* srow = stages + ((series * n_steps + k) * 4 + stage) * 32;
          if constexpr (PRODUCT) {           // K3's lane order: evens, then odds
            *reinterpret_cast<float4*>(srow + 4 * q) = make_float4(za[0], za[2], zb[0], zb[2]);          // units 8q, 8q+2, ..
            *reinterpret_cast<float4*>(srow + 16 + 4 * q) = make_float4(za[1], za[3], zb[1], zb[3]);     // units 8q+1, 8q+3, ..
          } else {                           // tanh / two-layer field: plain unit order (the lane owns units q, 4+q, .., 28+q)
#pragma unroll
            for (int i = 0; i < 4; ++i) { srow[q + 4 * i] = za[i]; srow[16 + q + 4 * i] = zb[i]; }
          }
        }
      }

      f32x4 fa, fb;
      if constexpr (PRODUCT) { if constexpr (CT == MC) field16(wA, wB, za, zb, dX, q, fa, fb); }
      else if constexpr (MLP) field_mlp16<ACT, CT, SPLIT, HI>(lds, lane, q, za, zb, dX, fa, fb, wave, xwin, xwin + 8 * 64, mlp_hi);
      else { if constexpr (CT == MC) field_act16<ACT>(wy, by, za, zb, dX, fa, fb); }
      if constexpr (!PRODUCT) {
        __builtin_amdgcn_sched_barrier(0);
        if (nidx != idx) nrow = load_row<DEGREE, CT>(coeffs, sc, n_intervals, nidx, Cr);
      }

      // torchdiffeq rk4_alt_step_func (3/8 rule), association order preserved
      const float third = (float)(1.0 / 3.0);
      if constexpr (METHOD == CDE_METHOD_EULER) {                       // y1 = y0 + dt * f(t0, y0)
        za = ya + dt * fa; zb = yb + dt * fb;
      } else if constexpr (METHOD == CDE_METHOD_MIDPOINT) {             // y_mid = y0 + f(t0, y0) * half_dt; y1 = y0 + dt * f(t0 + half_dt, y_mid)
        const float half_dt = 0.5f * dt;
        if (stage == 0) { za = ya + fa * half_dt; zb = yb + fb * half_dt; }
        else { za = ya + dt * fa; zb = yb + dt * fb; }
      } else if (stage == 0) {
        k1a = fa; k1b = fb;
        za = ya + dt * k1a * third; zb = yb + dt * k1b * third;
      } else if (stage == 1) {
        k2a = fa; k2b = fb;
        za = ya + dt * (k2a - k1a * third); zb = yb + dt * (k2b - k1b * third);
      } else if (stage == 2) {
        za = ya + dt * (k1a - k2a + fa); zb = yb + dt * (k1b - k2b + fb);
        pqa = k1a + 3.f * (k2a + fa); pqb = k1b + 3.f * (k2b + fb);
      } else {
        za = ya + (pqa + fa) * dt * 0.125f; zb = yb + (pqb + fb) * dt * 0.125f;
      }
      row = nrow; idx = nidx; frac = nfrac;
    }
    const f32x4 y1a = za, y1b = zb;
    while (jout < n_out && t1 >= t_out[jout]) {
      const TT tj = t_out[jout];
      if (tj == t0) store(jout, ya, yb);
      else if (tj == t1) store(jout, y1a, y1b);
      else {
        const float slope = (float)((tj - t0) / (t1 - t0));
        store(jout, ya + slope * (y1a - ya), yb + slope * (y1b - yb));
      }
      ++jout;
    }
    ya = y1a; yb = y1b;
  }
}

// ============================================================================================ adjoint
// partial layout per wave: [gW: (h*C+c)*H + k  (8192 floats)] [gb: h*C+c (256 floats)]
constexpr int64_t PARTIAL_FLOATS = MH * MC * MH + MH * MC;

template <typename TT, int DEGREE>
__global__ __launch_bounds__(256, 1) void rk4_adjoint_mfma(
    const float* __restrict__ coeffs, const float* __restrict__ knots, int64_t n_intervals,
    const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ z_saved,
    const float* __restrict__ grad_out, const TT* __restrict__ sgrid, const int64_t* __restrict__ seg_off,
    int64_t n_out, float* __restrict__ grad_z0, float* __restrict__ partial, int64_t B,
    const int64_t* __restrict__ stage_index, const float* __restrict__ stage_frac, Dims dims) {
  const int Hr = dims.H, Cr = dims.C;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* w1f = lds;
  float* w2f = lds + W1_FLOATS;
  for (int e = threadIdx.x; e < W1_FLOATS; e += 256) {
    const int s4 = e >> 8, l = (e >> 2) & 63, q = e & 3;
    w1f[e] = w1_image(W, bias, s4 * 4 + q, l, dims);
  }
  for (int e = threadIdx.x; e < W2_FLOATS; e += 256) {
    const int s4 = e >> 8, l = (e >> 2) & 63, q = e & 3;
    w2f[e] = w2_image(W, s4 * 4 + q, l, dims);
  }
  __syncthreads();
  const float4* w1 = reinterpret_cast<const float4*>(w1f);
  const float4* w2 = reinterpret_cast<const float4*>(w2f);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, half = lane >> 5;
  float* scr_y = lds + W1_FLOATS + W2_FLOATS + wave * SCR_FLOATS;

  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  float* my_partial = partial + tile * PARTIAL_FLOATS;
  if (tile * 32 >= B) return;   // host sizes `partial` by the number of live tiles only
  const int64_t series = tile * 32 + n;
  const bool valid = series < B;
  const int64_t sc = valid ? series : B - 1;

  f32x16 accW[MC];
  f32x2 gbp[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};   // dL/db partials, channel pairs
#pragma unroll
  for (int c = 0; c < MC; ++c) {
#pragma unroll
    for (int r = 0; r < 16; ++r) accW[c][r] = 0.f;
  }

  f32x16 y0, a0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int u = 2 * r + half;
    const bool on = u < Hr;
    y0[r] = on ? z_saved[(sc * n_out + (n_out - 1)) * Hr + u] : 0.f;
    a0[r] = (valid && on) ? grad_out[(sc * n_out + (n_out - 1)) * Hr + u] : 0.f;   // a == 0 stays 0: padded lanes add nothing to dL/dW
  }

  // Per-wave LDS scratch for the (series -> MFMA K index) transpose of the dL/dW product, laid out so that
  // the reader side is 16-byte vector loads:
  //   scr_zt[(par*32 + u)*20 + s] = z_u of series 2s+par      (row stride 20 floats = 80 B: b128-aligned and
  //   scr_at[(par*32 + u)*20 + s] = a_u of series 2s+par       conflict-free for the 16-lane b128 groups)
  //   scr_dw[series*8 + c]        = (quadrature weight * ds) * dX_c of that series
  // Instruction economy matters more than placement here: a single wave hides nothing behind its own f32
  // MFMAs (scripts/ubench/mfma_issue.hip), so every product is a packed v_pk_mul_f32, operands of the two
  // chains come straight from registers, and LDS is touched with b128 only.
  float* scr_zt = scr_y;                    // 64 rows x 20
  float* scr_at = scr_y + 64 * 20;          // 64 rows x 20
  float* scr_dw = scr_y + 2 * 64 * 20;      // 32 x 8

  for (int64_t p = 0; p + 1 < n_out; ++p) {
    const int64_t i_out = n_out - 1 - p;
    const int64_t k_begin = seg_off[p], k_end = seg_off[p + 1] - 1;   // steps k_begin .. k_end-1
    if (k_end > k_begin) {
      int64_t idx = stage_index[4 * k_begin];
      float frac = stage_frac[4 * k_begin];
      Row<DEGREE> row = load_row<DEGREE>(coeffs, sc, n_intervals, idx, Cr);
      for (int64_t k = k_begin; k < k_end; ++k) {
        const float ds = (float)(sgrid[k + 1] - sgrid[k]);
        f32x16 ky1, ky2, ka1, ka2, yst = y0, ast = a0;
#pragma unroll
        for (int stage = 0; stage < 4; ++stage) {
          float dX[MC];
          const float width = DEGREE == CDE_PATH_LINEAR ? knots[idx + 1] - knots[idx] : 1.f;
          control_slope<DEGREE>(row, frac, width, dX);
          // prefetch the next stage's table entry and (if the interval changes) its control row
          const int64_t e_next = 4 * k + stage + 1;
          const bool more = e_next < 4 * k_end;
          const int64_t nidx = more ? stage_index[e_next] : idx;
          const float nfrac = more ? stage_frac[e_next] : frac;
          if (nidx != idx) row = load_row<DEGREE>(coeffs, sc, n_intervals, nidx, Cr);

          const f32x2 d01 = {dX[0], dX[1]}, d23 = {dX[2], dX[3]}, d45 = {dX[4], dX[5]}, d67 = {dX[6], dX[7]};
          // ---- stage state -> scratch (transposed), weighted control derivative
          {
            const float wq = ((stage == 0 || stage == 3) ? 0.125f : 0.375f) * ds;   // 3/8-rule quadrature weight
            float* wz = scr_zt + ((n & 1) * 32 + half) * 20 + (n >> 1);              // + 2r*20
            float* wa = scr_at + ((n & 1) * 32 + half) * 20 + (n >> 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) { wz[r * 40] = yst[r]; wa[r * 40] = ast[r]; }
            const f32x2 w0 = (half ? d45 : d01) * wq, w1 = (half ? d67 : d23) * wq;
            *reinterpret_cast<float4*>(scr_dw + n * 8 + 4 * half) = make_float4(w0[0], w0[1], w1[0], w1[1]);
            wave_lds_sync();
          }

          // ---- f = W (z (x) dX) + b dX  and  va = a^T df/dz : two independent accumulator chains
          f32x16 f = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          f32x16 va = f;
          {
            const float4* p1 = w1 + lane;
            const float4* p2 = w2 + lane;
            float4 fa = p1[0], fb = p1[64], ga = p2[0], gb4 = p2[64];
            f32x2 zp = {yst[0], yst[1]}, ap = {ast[0], ast[1]};
            f32x2 z01 = pk_mul_lo(d01, zp), z23 = pk_mul_lo(d23, zp), z45 = pk_mul_lo(d45, zp), z67 = pk_mul_lo(d67, zp);
            f32x2 a01 = pk_mul_lo(d01, ap), a23 = pk_mul_lo(d23, ap), a45 = pk_mul_lo(d45, ap), a67 = pk_mul_lo(d67, ap);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
              // operands of group j+1: A images from LDS (4 x b128), products from registers (8 x pk_mul)
              float4 nfa = fa, nfb = fb, nga = ga, ngb = gb4;
              f32x2 y01 = z01, y23 = z23, y45 = z45, y67 = z67, b01 = a01, b23 = a23, b45 = a45, b67 = a67;
              if (j < 15) {
                nfa = p1[(2 * j + 2) * 64]; nfb = p1[(2 * j + 3) * 64];
                nga = p2[(2 * j + 2) * 64]; ngb = p2[(2 * j + 3) * 64];
                const int jn = j + 1;
                const f32x2 zq = {yst[jn & ~1], yst[jn | 1]}, aq = {ast[jn & ~1], ast[jn | 1]};
                if (jn & 1) {
                  y01 = pk_mul_hi(d01, zq); y23 = pk_mul_hi(d23, zq); y45 = pk_mul_hi(d45, zq); y67 = pk_mul_hi(d67, zq);
                  b01 = pk_mul_hi(d01, aq); b23 = pk_mul_hi(d23, aq); b45 = pk_mul_hi(d45, aq); b67 = pk_mul_hi(d67, aq);
                } else {
                  y01 = pk_mul_lo(d01, zq); y23 = pk_mul_lo(d23, zq); y45 = pk_mul_lo(d45, zq); y67 = pk_mul_lo(d67, zq);
                  b01 = pk_mul_lo(d01, aq); b23 = pk_mul_lo(d23, aq); b45 = pk_mul_lo(d45, aq); b67 = pk_mul_lo(d67, aq);
                }
              }
              __builtin_amdgcn_sched_barrier(0);
              f = mfma(fa.x, z01[0], f);   va = mfma(ga.x, a01[0], va);
              f = mfma(fa.y, z01[1], f);   va = mfma(ga.y, a01[1], va);
              f = mfma(fa.z, z23[0], f);   va = mfma(ga.z, a23[0], va);
              f = mfma(fa.w, z23[1], f);   va = mfma(ga.w, a23[1], va);
              f = mfma(fb.x, z45[0], f);   va = mfma(gb4.x, a45[0], va);
              f = mfma(fb.y, z45[1], f);   va = mfma(gb4.y, a45[1], va);
              f = mfma(fb.z, z67[0], f);   va = mfma(gb4.z, a67[0], va);
              f = mfma(fb.w, z67[1], f);   va = mfma(gb4.w, a67[1], va);
              __builtin_amdgcn_sched_barrier(0);
              fa = nfa; fb = nfb; ga = nga; gb4 = ngb;
              z01 = y01; z23 = y23; z45 = y45; z67 = y67; a01 = b01; a23 = b23; a45 = b45; a67 = b67;
            }
            const float4 wc = p1[32 * 64];               // bias steps: lane half hk contributes channel 2*sp + hk
            f = mfma(wc.x, half ? dX[1] : dX[0], f);
            f = mfma(wc.y, half ? dX[3] : dX[2], f);
            f = mfma(wc.z, half ? dX[5] : dX[4], f);
            f = mfma(wc.w, half ? dX[7] : dX[6], f);
          }

          // ---- dL/dW tile c: D[h][k] += sum_series (w ds a_h dX_c)[series] * z_k[series]; this lane feeds MFMA
          // K index `half` of K-step s2, i.e. series 2*s2 + half, row h = n, column k = n.
          {
            const float4* zt4 = reinterpret_cast<const float4*>(scr_zt + (half * 32 + n) * 20);
            const float4* at4 = reinterpret_cast<const float4*>(scr_at + (half * 32 + n) * 20);
            const float4* dw4 = reinterpret_cast<const float4*>(scr_dw + half * 8);      // + s2*4 (16 floats per s2)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const float4 zq = zt4[g4], aq = at4[g4];                       // K-steps 4*g4 .. 4*g4+3
              const f32x2 ap0 = {aq.x, aq.y}, ap1 = {aq.z, aq.w};
              const float zs[4] = {zq.x, zq.y, zq.z, zq.w};
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const int s2 = 4 * g4 + i;
                const float4 e0 = dw4[s2 * 4], e1 = dw4[s2 * 4 + 1];
                const f32x2 e01 = {e0.x, e0.y}, e23 = {e0.z, e0.w}, e45 = {e1.x, e1.y}, e67 = {e1.z, e1.w};
                const f32x2 asrc = i < 2 ? ap0 : ap1;
                f32x2 v01, v23, v45, v67;
                if (i & 1) {
                  v01 = pk_mul_hi(e01, asrc); v23 = pk_mul_hi(e23, asrc); v45 = pk_mul_hi(e45, asrc); v67 = pk_mul_hi(e67, asrc);
                  pk_fma_hi(gbp[0], e01, asrc); pk_fma_hi(gbp[1], e23, asrc); pk_fma_hi(gbp[2], e45, asrc); pk_fma_hi(gbp[3], e67, asrc);
                } else {
                  v01 = pk_mul_lo(e01, asrc); v23 = pk_mul_lo(e23, asrc); v45 = pk_mul_lo(e45, asrc); v67 = pk_mul_lo(e67, asrc);
                  pk_fma_lo(gbp[0], e01, asrc); pk_fma_lo(gbp[1], e23, asrc); pk_fma_lo(gbp[2], e45, asrc); pk_fma_lo(gbp[3], e67, asrc);
                }
                __builtin_amdgcn_sched_barrier(0);
                const float zb = zs[i];
                accW[0] = mfma(v01[0], zb, accW[0]); accW[1] = mfma(v01[1], zb, accW[1]);
                accW[2] = mfma(v23[0], zb, accW[2]); accW[3] = mfma(v23[1], zb, accW[3]);
                accW[4] = mfma(v45[0], zb, accW[4]); accW[5] = mfma(v45[1], zb, accW[5]);
                accW[6] = mfma(v67[0], zb, accW[6]); accW[7] = mfma(v67[1], zb, accW[7]);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }
          wave_lds_sync();   // scratch reads retired before the next stage overwrites it

          // ---- reverse-time dynamics: dy/ds = -f, da/ds = +a^T df/dz.  3/8 rule in two slots per variable:
          // after stage 2 slot 1 holds k1 + 3*(k2+k3) (same association as torchdiffeq).
          const f32x16 ky = -f, ka = va;
          const float third = (float)(1.0 / 3.0);
          if (stage == 0) {
            ky1 = ky; ka1 = ka;
            yst = y0 + ds * ky1 * third;
            ast = a0 + ds * ka1 * third;
          } else if (stage == 1) {
            ky2 = ky; ka2 = ka;
            yst = y0 + ds * (ky2 - ky1 * third);
            ast = a0 + ds * (ka2 - ka1 * third);
          } else if (stage == 2) {
            yst = y0 + ds * (ky1 - ky2 + ky);
            ast = a0 + ds * (ka1 - ka2 + ka);
            ky1 = ky1 + 3.f * (ky2 + ky);
            ka1 = ka1 + 3.f * (ka2 + ka);
          } else {
            yst = y0 + (ky1 + ky) * ds * 0.125f;
            ast = a0 + (ka1 + ka) * ds * 0.125f;
          }
          idx = nidx; frac = nfrac;
        }
        y0 = yst; a0 = ast;
      }
    }
    // torchdiffeq adjoint: re-seed y from the stored forward value, add the incoming gradient
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int u = 2 * r + half;
      if (u < Hr) {
        y0[r] = z_saved[(sc * n_out + (i_out - 1)) * Hr + u];
        if (valid) a0[r] += grad_out[(sc * n_out + (i_out - 1)) * Hr + u];
      }
    }
  }
  if (valid) {
#pragma unroll
    for (int r = 0; r < 16; ++r) if (2 * r + half < Hr) grad_z0[series * Hr + 2 * r + half] = a0[r];
  }
  // per-wave partial parameter gradients (summed in tile order by reduce_mfma_partials)
#pragma unroll
  for (int c = 0; c < MC; ++c) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int h = (r & 3) + 8 * (r >> 2) + 4 * half;
      my_partial[(h * MC + c) * MH + n] = accW[c][r];
    }
    // gb partial of lane (h = n, half): add the two halves through a lane exchange
    const float mine_gb = gbp[c >> 1][c & 1];
    const float other = __shfl_xor(mine_gb, 32, 64);
    if (half == 0) my_partial[MH * MC * MH + n * MC + c] = mine_gb + other;
  }
}

// ============================================================================================ adjoint, shared Jacobian
// K3j.  The vector field of the headline configuration is AFFINE in z: f(z) = (W z + b) dX, so its Jacobian
//     J = df/dz = sum_c dX_c W_c          (W_c[h][k] = W[(h, c), k]; H x H per series and stage time)
// serves both products the adjoint needs: f = J z + (b dX) and a^T df/dz = a^T J.  K3 evaluates them as two GEMMs
// against W ((z (x) dX) and (a (x) dX) as the B operands: 2 x 16384 flop per series and stage); here ONE GEMM of the
// same size forms J -- M = (h, k) = 1024 rows, K = the 8 channels, N = the series, i.e. 4 MFMAs per row h that leave
// J[h][k], k = the lane's own 16 units, in the lane -- and the two H x H matrix-vector products run on the vector
// pipe straight from the MFMA result (16 packed FMAs per row h, two half-lane exchanges per unit pair per stage).
// Per stage and 32 series: 132 + 128 MFMAs instead of 260 + 128.  The reassociation moves roundings, not the
// algorithm: same stages, same quadrature, same dL/dW product (tests: the same tolerances as K3 against the oracle).
constexpr int WJ_ROWS = MH + 1;                          // 32 rows of J + the bias rows
constexpr int WJ_FLOATS = WJ_ROWS * 64 * 4;
// A operand of row `h`, K step s (channels 2s, 2s + 1), lane l: MFMA row i = l & 31 is unit k = rho(i), so that register
// r of half-lane `half` receives k = 2r + half -- the layout the state lives in.  Row 32: bias, MFMA row i = unit rho(i).
__device__ __forceinline__ float wj_image(const float* __restrict__ W, const float* __restrict__ bias, int h, int s, int l,
                                          Dims d) {
  const int k = rho(l & 31), c = 2 * s + (l >> 5);
  if (c >= d.C || k >= d.H) return 0.f;
  if (h < MH) return h < d.H ? W[(h * d.C + c) * d.H + k] : 0.f;
  return bias[k * d.C + c];
}

// BX (variant "bf16x3"): J's GEMM on the bf16 matrix pipe at float32 accuracy.  Every float32 operand is split into three
// bf16 pieces (x = x1 + x2 + x3) and the product takes the six piece products with i + j <= 4 (csrc/rk4_bf16x3.hip).  J's
// GEMM has K = 8 channels, a bf16 MFMA K = 16: the second half of K carries a second PIECE of dX, so three
// v_mfma_f32_32x32x16_bf16 (8 passes each) replace the four 16-pass f32 MFMAs of a row:
//     W1 (d1 | d2)  +  W2 (d1 | d2)  +  (W1 | W3) (d3 | d1)          (A | B: lower | upper half of the K index)
// The weight pieces are split once per launch into the LDS image, dX's eight values once per stage.
using bf16x8j = __attribute__((ext_vector_type(8))) __bf16;
using u32x4j = __attribute__((ext_vector_type(4))) unsigned;
constexpr int WJB_U4 = WJ_ROWS * 3 * 64;                 // 16-byte entries: [row][MFMA][lane]
constexpr int WJB_FLOATS = WJB_U4 * 4;
__device__ __forceinline__ void split3j(float x, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)x;
  const float r1 = x - (float)a;
  b = (__bf16)r1;
  c = (__bf16)(r1 - (float)b);
}
__device__ __forceinline__ u32x4j wjb_image(const float* __restrict__ W, const float* __restrict__ bias, int h, int m, int l, Dims d) {
  const int k = rho(l & 31), upper = l >> 5;
  bf16x8j out;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    float w = 0.f;
    if (c < d.C && k < d.H) w = h < MH ? (h < d.H ? W[(h * d.C + c) * d.H + k] : 0.f) : bias[k * d.C + c];
    __bf16 p1, p2, p3;
    split3j(w, p1, p2, p3);
    out[c] = m == 0 ? p1 : m == 1 ? p2 : (upper ? p3 : p1);
  }
  return __builtin_bit_cast(u32x4j, out);
}

template <typename TT, int DEGREE, bool BX = false>
__global__ __launch_bounds__(256, 1) void rk4_adjoint_jacobian(
    const float* __restrict__ coeffs, const float* __restrict__ knots, int64_t n_intervals,
    const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ z_saved,
    const float* __restrict__ grad_out, const TT* __restrict__ sgrid, const int64_t* __restrict__ seg_off,
    int64_t n_out, float* __restrict__ grad_z0, float* __restrict__ partial, int64_t B,
    const int64_t* __restrict__ stage_index, const float* __restrict__ stage_frac, Dims dims) {
  const int Hr = dims.H, Cr = dims.C;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if constexpr (BX) {
    u32x4j* img = reinterpret_cast<u32x4j*>(lds);
    for (int e = threadIdx.x; e < WJB_U4; e += 256) img[e] = wjb_image(W, bias, e / 192, (e >> 6) % 3, e & 63, dims);
  } else {
    for (int e = threadIdx.x; e < WJ_FLOATS; e += 256) lds[e] = wj_image(W, bias, e >> 8, e & 3, (e >> 2) & 63, dims);
  }
  __syncthreads();
  const float4* wj = reinterpret_cast<const float4*>(lds);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, half = lane >> 5;
  float* scr_y = lds + (BX ? WJB_FLOATS : WJ_FLOATS) + wave * SCR_FLOATS;

  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  float* my_partial = partial + tile * PARTIAL_FLOATS;
  if (tile * 32 >= B) return;   // host sizes `partial` by the number of live tiles only
  const int64_t series = tile * 32 + n;
  const bool valid = series < B;
  const int64_t sc = valid ? series : B - 1;

  f32x16 accW[MC];
  f32x2 gbp[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};   // dL/db partials, channel pairs
#pragma unroll
  for (int c = 0; c < MC; ++c) {
#pragma unroll
    for (int r = 0; r < 16; ++r) accW[c][r] = 0.f;
  }

  f32x16 y0, a0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int u = 2 * r + half;
    const bool on = u < Hr;
    y0[r] = on ? z_saved[(sc * n_out + (n_out - 1)) * Hr + u] : 0.f;
    a0[r] = (valid && on) ? grad_out[(sc * n_out + (n_out - 1)) * Hr + u] : 0.f;   // a == 0 stays 0: padded lanes add nothing to dL/dW
  }

  // Per-wave LDS scratch for the (series -> MFMA K index) transpose of the dL/dW product, laid out so that
  // the reader side is 16-byte vector loads:
  //   scr_zt[(par*32 + u)*20 + s] = z_u of series 2s+par      (row stride 20 floats = 80 B: b128-aligned and
  //   scr_at[(par*32 + u)*20 + s] = a_u of series 2s+par       conflict-free for the 16-lane b128 groups)
  //   scr_dw[series*8 + c]        = (quadrature weight * ds) * dX_c of that series
  // Instruction economy matters more than placement here: a single wave hides nothing behind its own f32
  // MFMAs (scripts/ubench/mfma_issue.hip), so every product is a packed v_pk_mul_f32, operands of the two
  // chains come straight from registers, and LDS is touched with b128 only.
  float* scr_zt = scr_y;                    // 64 rows x 20
  float* scr_at = scr_y + 64 * 20;          // 64 rows x 20
  float* scr_dw = scr_y + 2 * 64 * 20;      // 32 x 8

  for (int64_t p = 0; p + 1 < n_out; ++p) {
    const int64_t i_out = n_out - 1 - p;
    const int64_t k_begin = seg_off[p], k_end = seg_off[p + 1] - 1;   // steps k_begin .. k_end-1
    if (k_end > k_begin) {
      int64_t idx = stage_index[4 * k_begin];
      float frac = stage_frac[4 * k_begin];
      Row<DEGREE> row = load_row<DEGREE>(coeffs, sc, n_intervals, idx, Cr);
      for (int64_t k = k_begin; k < k_end; ++k) {
        const float ds = (float)(sgrid[k + 1] - sgrid[k]);
        f32x16 ky1, ky2, ka1, ka2, yst = y0, ast = a0;
#pragma unroll
        for (int stage = 0; stage < 4; ++stage) {
          float dX[MC];
          const float width = DEGREE == CDE_PATH_LINEAR ? knots[idx + 1] - knots[idx] : 1.f;
          control_slope<DEGREE>(row, frac, width, dX);
          // prefetch the next stage's table entry and (if the interval changes) its control row
          const int64_t e_next = 4 * k + stage + 1;
          const bool more = e_next < 4 * k_end;
          const int64_t nidx = more ? stage_index[e_next] : idx;
          const float nfrac = more ? stage_frac[e_next] : frac;
          if (nidx != idx) row = load_row<DEGREE>(coeffs, sc, n_intervals, nidx, Cr);

          const f32x2 d01 = {dX[0], dX[1]}, d23 = {dX[2], dX[3]}, d45 = {dX[4], dX[5]}, d67 = {dX[6], dX[7]};
          // ---- stage state -> scratch (transposed), weighted control derivative
          {
            const float wq = ((stage == 0 || stage == 3) ? 0.125f : 0.375f) * ds;   // 3/8-rule quadrature weight
            float* wz = scr_zt + ((n & 1) * 32 + half) * 20 + (n >> 1);              // + 2r*20
            float* wa = scr_at + ((n & 1) * 32 + half) * 20 + (n >> 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) { wz[r * 40] = yst[r]; wa[r * 40] = ast[r]; }
            const f32x2 w0 = (half ? d45 : d01) * wq, w1 = (half ? d67 : d23) * wq;
            *reinterpret_cast<float4*>(scr_dw + n * 8 + 4 * half) = make_float4(w0[0], w0[1], w1[0], w1[1]);
            wave_lds_sync();
          }

          // ---- J = sum_c dX_c W_c one row (hidden unit h) at a time: 4 MFMAs leave J[h][k] of this lane's series in
          // the lane, k = the 16 units it owns; f_h += J[h][.] . z and va += a_h J[h][.] follow on the vector pipe.
          f32x16 f, va;
          {
            const float bs0 = half ? dX[1] : dX[0], bs1 = half ? dX[3] : dX[2], bs2 = half ? dX[5] : dX[4], bs3 = half ? dX[7] : dX[6];
            f32x2 va2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) va2[j] = f32x2{0.f, 0.f};
            int opaque = 0;                               // the image reads are loop invariant: without this the
            asm volatile("" : "+v"(opaque));              // compiler hoists all 33 out of the solve (132 registers)
            const float4* wp = wj + lane + opaque;
            // One row of J = 4 dependent MFMAs into VECTOR registers.  Written as asm because the compiler gives MFMA
            // results of this kernel accumulator registers (the 128 of dL/dW have to live there) and would then move
            // every J value back with a v_accvgpr_read: 16 extra instructions per row, each one costing the wave
            // matrix-pipe time.  Hazards: the rows are issued one ahead of their use -- J of row h is read only after
            // the 4 MFMAs of row h + 1 have ISSUED, and those queue behind row h in the same in-order pipe (each waits
            // for its predecessor's accumulator), so row h has long been written; the empty asm on J after the next
            // issue pins that order.  The last issue of a stage (bias rows) is followed by explicit wait states.
            // (BX) B operands: the three bf16 pieces of this lane's 8 channel values; lower half-lanes feed d1 (rows 1, 2)
            // and d3 (row 3), upper ones d2 and d1
            u32x4j b01 = {0u, 0u, 0u, 0u}, b2 = b01;
            if constexpr (BX) {
              bf16x8j q01, q2;
#pragma unroll
              for (int c = 0; c < 8; ++c) {
                __bf16 p1, p2, p3;
                split3j(dX[c], p1, p2, p3);
                q01[c] = half ? p2 : p1;
                q2[c] = half ? p1 : p3;
              }
              b01 = __builtin_bit_cast(u32x4j, q01);
              b2 = __builtin_bit_cast(u32x4j, q2);
            }
            const u32x4j* wpb = reinterpret_cast<const u32x4j*>(wj) + lane + opaque;
            struct RowImage { float4 a; u32x4j m0, m1, m2; };
            auto image = [&](int h) {
              RowImage r;
              if constexpr (BX) { r.m0 = wpb[(3 * h) * 64]; r.m1 = wpb[(3 * h + 1) * 64]; r.m2 = wpb[(3 * h + 2) * 64]; }
              else r.a = wp[h * 64];
              return r;
            };
            auto issue = [&](f32x16& J, const RowImage& a) {
              __builtin_amdgcn_sched_barrier(0);           // everything that still reads the old J stays above
              if constexpr (BX) {
                asm volatile("s_nop 1\n\t"                                     // (operands may be fresh VALU results)
                             "v_mfma_f32_32x32x16_bf16 %0, %3, %5, 0\n\t"      // smallest terms first
                             "v_mfma_f32_32x32x16_bf16 %0, %2, %4, %0\n\t"
                             "v_mfma_f32_32x32x16_bf16 %0, %1, %4, %0"
                             : "=&v"(J) : "v"(a.m0), "v"(a.m1), "v"(a.m2), "v"(b01), "v"(b2));
              } else {
                asm volatile("s_nop 1\n\t"                                     // (operands may be fresh VALU results)
                             "v_mfma_f32_32x32x2_f32 %0, %1, %5, 0\n\t"
                             "v_mfma_f32_32x32x2_f32 %0, %2, %6, %0\n\t"
                             "v_mfma_f32_32x32x2_f32 %0, %3, %7, %0\n\t"
                             "v_mfma_f32_32x32x2_f32 %0, %4, %8, %0"
                             : "=&v"(J) : "v"(a.a.x), "v"(a.a.y), "v"(a.a.z), "v"(a.a.w), "v"(bs0), "v"(bs1), "v"(bs2), "v"(bs3));
              }
              __builtin_amdgcn_sched_barrier(0);
            };
            // the two uses of a row: this half-lane's share of f_h = J[h][.] . z, and va += a_h J[h][.]
            auto consume = [&](const f32x16& J, float ah) {
              const f32x2 ah2 = {ah, ah};
              f32x2 p2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                const f32x2 m = {J[2 * j], J[2 * j + 1]};
                if (j & 1) q2 = __builtin_elementwise_fma(m, f32x2{yst[2 * j], yst[2 * j + 1]}, q2);
                else p2 = __builtin_elementwise_fma(m, f32x2{yst[2 * j], yst[2 * j + 1]}, p2);
                va2[j] = __builtin_elementwise_fma(m, ah2, va2[j]);
              }
              // (va is only read after the last row: left alone, the optimiser sinks all 256 of these FMAs below the
              // last row and keeps every J alive in scratch until then)
#pragma unroll
              for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(va2[j]));
              p2 = p2 + q2;
              return p2[0] + p2[1];
            };
            f32x16 Je, Jo;                                 // rows 2r / 2r + 1 in flight
            RowImage a_cur = image(0), a_nxt = image(1);
            issue(Je, a_cur);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              a_cur = image(2 * r + 2);                    // image of row 2r + 2 (r = 15: the bias rows)
              issue(Jo, a_nxt);
              asm volatile("" : "+v"(Je));
              // a of units 2r, 2r + 1 in both half-lanes (the lower half-lanes own unit 2r, the upper ones 2r + 1)
              float ae = ast[r], ao = ast[r];
              swap32(ae, ao);
              float X = consume(Je, ae);
              if (r < 15) a_nxt = image(2 * r + 3);
              issue(Je, a_cur);
              asm volatile("" : "+v"(Jo));
              float Y = consume(Jo, ao);
              swap32(X, Y);                                // ... and each half-lane collects the f of the unit it owns
              f[r] = X + Y;
            }
            // Je: (b dX)_h for the units this lane owns.  16-pass MFMA result -> VALU read: 20 wait states
            asm volatile("s_nop 15\n\ts_nop 7" : "+v"(Je));
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              f[r] += Je[r];
              va[r] = va2[r >> 1][r & 1];
            }
          }

          // ---- dL/dW tile c: D[h][k] += sum_series (w ds a_h dX_c)[series] * z_k[series]; this lane feeds MFMA
          // K index `half` of K-step s2, i.e. series 2*s2 + half, row h = n, column k = n.
          {
            const float4* zt4 = reinterpret_cast<const float4*>(scr_zt + (half * 32 + n) * 20);
            const float4* at4 = reinterpret_cast<const float4*>(scr_at + (half * 32 + n) * 20);
            const float4* dw4 = reinterpret_cast<const float4*>(scr_dw + half * 8);      // + s2*4 (16 floats per s2)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const float4 zq = zt4[g4], aq = at4[g4];                       // K-steps 4*g4 .. 4*g4+3
              const f32x2 ap0 = {aq.x, aq.y}, ap1 = {aq.z, aq.w};
              const float zs[4] = {zq.x, zq.y, zq.z, zq.w};
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const int s2 = 4 * g4 + i;
                const float4 e0 = dw4[s2 * 4], e1 = dw4[s2 * 4 + 1];
                const f32x2 e01 = {e0.x, e0.y}, e23 = {e0.z, e0.w}, e45 = {e1.x, e1.y}, e67 = {e1.z, e1.w};
                const f32x2 asrc = i < 2 ? ap0 : ap1;
                f32x2 v01, v23, v45, v67;
                if (i & 1) {
                  v01 = pk_mul_hi(e01, asrc); v23 = pk_mul_hi(e23, asrc); v45 = pk_mul_hi(e45, asrc); v67 = pk_mul_hi(e67, asrc);
                  pk_fma_hi(gbp[0], e01, asrc); pk_fma_hi(gbp[1], e23, asrc); pk_fma_hi(gbp[2], e45, asrc); pk_fma_hi(gbp[3], e67, asrc);
                } else {
                  v01 = pk_mul_lo(e01, asrc); v23 = pk_mul_lo(e23, asrc); v45 = pk_mul_lo(e45, asrc); v67 = pk_mul_lo(e67, asrc);
                  pk_fma_lo(gbp[0], e01, asrc); pk_fma_lo(gbp[1], e23, asrc); pk_fma_lo(gbp[2], e45, asrc); pk_fma_lo(gbp[3], e67, asrc);
                }
                __builtin_amdgcn_sched_barrier(0);
                const float zb = zs[i];
                accW[0] = mfma(v01[0], zb, accW[0]); accW[1] = mfma(v01[1], zb, accW[1]);
                accW[2] = mfma(v23[0], zb, accW[2]); accW[3] = mfma(v23[1], zb, accW[3]);
                accW[4] = mfma(v45[0], zb, accW[4]); accW[5] = mfma(v45[1], zb, accW[5]);
                accW[6] = mfma(v67[0], zb, accW[6]); accW[7] = mfma(v67[1], zb, accW[7]);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }
          wave_lds_sync();   // scratch reads retired before the next stage overwrites it

          // ---- reverse-time dynamics: dy/ds = -f, da/ds = +a^T df/dz.  3/8 rule in two slots per variable:
          // after stage 2 slot 1 holds k1 + 3*(k2+k3) (same association as torchdiffeq).
          const f32x16 ky = -f, ka = va;
          const float third = (float)(1.0 / 3.0);
          if (stage == 0) {
            ky1 = ky; ka1 = ka;
            yst = y0 + ds * ky1 * third;
            ast = a0 + ds * ka1 * third;
          } else if (stage == 1) {
            ky2 = ky; ka2 = ka;
            yst = y0 + ds * (ky2 - ky1 * third);
            ast = a0 + ds * (ka2 - ka1 * third);
          } else if (stage == 2) {
            yst = y0 + ds * (ky1 - ky2 + ky);
            ast = a0 + ds * (ka1 - ka2 + ka);
            ky1 = ky1 + 3.f * (ky2 + ky);
            ka1 = ka1 + 3.f * (ka2 + ka);
          } else {
            yst = y0 + (ky1 + ky) * ds * 0.125f;
            ast = a0 + (ka1 + ka) * ds * 0.125f;
          }
          idx = nidx; frac = nfrac;
        }
        y0 = yst; a0 = ast;
      }
    }
    // torchdiffeq adjoint: re-seed y from the stored forward value, add the incoming gradient
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int u = 2 * r + half;
      if (u < Hr) {
        y0[r] = z_saved[(sc * n_out + (i_out - 1)) * Hr + u];
        if (valid) a0[r] += grad_out[(sc * n_out + (i_out - 1)) * Hr + u];
      }
    }
  }
  if (valid) {
#pragma unroll
    for (int r = 0; r < 16; ++r) if (2 * r + half < Hr) grad_z0[series * Hr + 2 * r + half] = a0[r];
  }
  // per-wave partial parameter gradients (summed in tile order by reduce_mfma_partials)
#pragma unroll
  for (int c = 0; c < MC; ++c) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int h = (r & 3) + 8 * (r >> 2) + 4 * half;
      my_partial[(h * MC + c) * MH + n] = accW[c][r];
    }
    // gb partial of lane (h = n, half): add the two halves through a lane exchange
    const float mine_gb = gbp[c >> 1][c & 1];
    const float other = __shfl_xor(mine_gb, 32, 64);
    if (half == 0) my_partial[MH * MC * MH + n * MC + c] = mine_gb + other;
  }
}

// ============================================================================================ adjoint, fields with an activation
// K3a.  f = reshape(act(W z + b)) dX.  Same ownership as K3 (one wave = 32 series, lane (n, half) keeps hidden
// units 2r + half), but the GEMMs are the pre-activation ones, processed in 8 row tiles of 4 hidden units x 8
// channels (32x32x2 MFMAs, 48 per tile, 384 per stage):
//   Y_T   = W_T z + b_T               16 MFMAs: row i <-> (h = 4T + (i>>3), c = i&7); K step s: half hk feeds unit 2s+hk
//                                       => lane (n, half) register r = Y[h = 4T + (r>>2)][c = (r&3) + 4*half]
//   in-lane: t = act(Y), partial f_h = sum_c t dX_c over this half's 4 channels, u = act'(Y) dX_c
//            v_permlane32_swap adds the two halves' partial sums and leaves f for unit 2r+half in its owner;
//            it also broadcasts the adjoint state of the tile's 4 units to both halves: g = a_h u  (= dL/dY)
//   va   += W_T^T g                   16 MFMAs: K step r: half hk feeds (h = 4T + (r>>2), c = (r&3) + 4hk) = its OWN
//                                       register r; row i <-> output unit rho(i)
//   dW_T += (w ds g)^T z              16 MFMAs with the series as K: g goes through a per-wave LDS transpose
//                                       (rows of 16 series + 4 pad, ds_read_b128 on the reader side); dL/db is the
//                                       row sum of the same transposed tile.
template <int ACT>
__device__ __forceinline__ float activate_slope(float t) { return ACT == CDE_ACT_TANH ? __builtin_fmaf(-t, t, 1.f) : 1.f; }

constexpr int WV32_FLOATS = 8 * 16 * 64;          // one 32x32x2 image: 8 tiles x 16 K steps x 64 lanes
constexpr int BY32_FLOATS = 8 * 2 * 16;           // bias image [tile][half][register]
constexpr int SCRA_FLOATS = 2 * 64 * 20;          // per wave: z^T and one transposed g tile
constexpr int ACT_ADJ_LDS_FLOATS = 2 * WV32_FLOATS + BY32_FLOATS + 4 * SCRA_FLOATS;

__device__ __forceinline__ float wy32_image(const float* __restrict__ W, int T, int s, int l, Dims d) {
  const int i = l & 31, hk = l >> 5;
  const int h = 4 * T + (i >> 3), c = i & 7, k = 2 * s + hk;
  return (h < d.H && c < d.C && k < d.H) ? W[(h * d.C + c) * d.H + k] : 0.f;
}
__device__ __forceinline__ float wv32_image(const float* __restrict__ W, int T, int r, int l, Dims d) {
  const int k = rho(l & 31), hk = l >> 5;
  const int h = 4 * T + (r >> 2), c = (r & 3) + 4 * hk;
  return (h < d.H && c < d.C && k < d.H) ? W[(h * d.C + c) * d.H + k] : 0.f;
}


//
// DCOEFF: also produce dL/d(control coefficients) (adjoint_params containing the coefficient tensor, reference
// solver.py:207-222).  f depends on the control only through dX_c, and d(a.f)/d(dX_c) = sum_h a_h act(Y)_hc is a sum
// over rows this lane already holds (its 4 channels, all 32 hidden units after the 8 tiles): no cross-lane traffic.
// The chain to the row of the interval in use -- cubic: d/db = 1, d/d(2c) = frac, d/d(3d) = frac^2; linear:
// -+1/width on the two knots -- is accumulated in registers while the interval stays the same and added to
// `grad_coeffs` (zeroed by the caller, same layout as `coeffs`) when it changes; one lane owns a (series, channel), so
// the read-modify-write needs no atomics and the result is run-to-run deterministic.
template <typename TT, int DEGREE, int ACT, bool DCOEFF = false>
__global__ __launch_bounds__(256, 1) void rk4_adjoint_act_mfma(
    const float* __restrict__ coeffs, const float* __restrict__ knots, int64_t n_intervals,
    const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ z_saved,
    const float* __restrict__ grad_out, const TT* __restrict__ sgrid, const int64_t* __restrict__ seg_off,
    int64_t n_out, float* __restrict__ grad_z0, float* __restrict__ partial, int64_t B,
    const int64_t* __restrict__ stage_index, const float* __restrict__ stage_frac, Dims dims,
    float* __restrict__ grad_coeffs = nullptr) {
  const int Hr = dims.H, Cr = dims.C;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* wyf = lds;
  float* wvf = lds + WV32_FLOATS;
  float* byf = lds + 2 * WV32_FLOATS;
  for (int e = threadIdx.x; e < WV32_FLOATS; e += 256) {
    const int j = e & 3, l = (e >> 2) & 63, g = e >> 8;             // g = 4T + (step >> 2)
    wyf[e] = wy32_image(W, g >> 2, 4 * (g & 3) + j, l, dims);
    wvf[e] = wv32_image(W, g >> 2, 4 * (g & 3) + j, l, dims);
  }
  for (int e = threadIdx.x; e < BY32_FLOATS; e += 256) {
    const int r = e & 15, hf = (e >> 4) & 1, T = e >> 5;
    const int h = 4 * T + (r >> 2), c = (r & 3) + 4 * hf;
    byf[e] = (h < Hr && c < Cr) ? bias[h * Cr + c] : 0.f;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, half = lane >> 5;
  const float4* wy = reinterpret_cast<const float4*>(wyf) + lane;
  const float4* wv = reinterpret_cast<const float4*>(wvf) + lane;
  const float4* by = reinterpret_cast<const float4*>(byf) + 4 * half;
  float* scr_zt = lds + 2 * WV32_FLOATS + BY32_FLOATS + wave * SCRA_FLOATS;      // 64 rows x 20
  float* scr_g = scr_zt + 64 * 20;                                               // 64 rows x 20

  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  float* my_partial = partial + tile * PARTIAL_FLOATS;
  if (tile * 32 >= B) return;
  const int64_t series = tile * 32 + n;
  const bool valid = series < B;
  const int64_t sc = valid ? series : B - 1;

  f32x16 accW[8];
  float gb[8];
#pragma unroll
  for (int T = 0; T < 8; ++T) {
    gb[T] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) accW[T][r] = 0.f;
  }
  f32x16 y0, a0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int u = 2 * r + half;
    const bool on = u < Hr;
    y0[r] = on ? z_saved[(sc * n_out + (n_out - 1)) * Hr + u] : 0.f;
    a0[r] = (valid && on) ? grad_out[(sc * n_out + (n_out - 1)) * Hr + u] : 0.f;
  }

  // dL/d(coefficient row in use), this lane's 4 channels: cubic (b, 2c, 3d), linear (left knot, right knot)
  float gc0[4] = {0.f, 0.f, 0.f, 0.f}, gc1[4] = {0.f, 0.f, 0.f, 0.f}, gc2[4] = {0.f, 0.f, 0.f, 0.f};
  auto flush_control_grad = [&](int64_t at) {
    if constexpr (DCOEFF) {
#pragma unroll
      for (int cl = 0; cl < 4; ++cl) {
        const int c = cl + 4 * half;
        if (valid && c < Cr) {
          if (DEGREE == CDE_PATH_CUBIC) {
            float* g = grad_coeffs + (series * n_intervals + at) * 4 * Cr;
            g[Cr + c] += gc0[cl]; g[2 * Cr + c] += gc1[cl]; g[3 * Cr + c] += gc2[cl];
          } else {
            float* g = grad_coeffs + (series * (n_intervals + 1) + at) * Cr;
            g[c] += gc0[cl]; g[Cr + c] += gc1[cl];
          }
        }
        gc0[cl] = 0.f; gc1[cl] = 0.f; gc2[cl] = 0.f;
      }
    }
  };

  for (int64_t p = 0; p + 1 < n_out; ++p) {
    const int64_t i_out = n_out - 1 - p;
    const int64_t k_begin = seg_off[p], k_end = seg_off[p + 1] - 1;
    if (k_end > k_begin) {
      int64_t idx = stage_index[4 * k_begin];
      float frac = stage_frac[4 * k_begin];
      Row<DEGREE> row = load_row<DEGREE>(coeffs, sc, n_intervals, idx, Cr);
      for (int64_t k = k_begin; k < k_end; ++k) {
        const float ds = (float)(sgrid[k + 1] - sgrid[k]);
        f32x16 ky1, ky2, ka1, ka2, yst = y0, ast = a0;
#pragma unroll
        for (int stage = 0; stage < 4; ++stage) {
          float dX[MC];
          const float width = DEGREE == CDE_PATH_LINEAR ? knots[idx + 1] - knots[idx] : 1.f;
          control_slope<DEGREE>(row, frac, width, dX);
          const int64_t e_next = 4 * k + stage + 1;
          const bool more = e_next < 4 * k_end;
          const int64_t nidx = more ? stage_index[e_next] : idx;
          const float nfrac = more ? stage_frac[e_next] : frac;
          if (nidx != idx) row = load_row<DEGREE>(coeffs, sc, n_intervals, nidx, Cr);
          // this half's 4 channels
          const float dh[4] = {half ? dX[4] : dX[0], half ? dX[5] : dX[1], half ? dX[6] : dX[2], half ? dX[7] : dX[3]};
          const float wq = ((stage == 0 || stage == 3) ? 0.125f : 0.375f) * ds;     // 3/8-rule quadrature weight

          // z^T for the dL/dW products (same layout as K3): scr_zt[(par*32 + u)*20 + s] = z_u of series 2s+par
          {
            float* wz = scr_zt + ((n & 1) * 32 + half) * 20 + (n >> 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) wz[r * 40] = yst[r];
            wave_lds_sync();
          }
          float zB[16];
          {
            const float4* zt4 = reinterpret_cast<const float4*>(scr_zt + (half * 32 + n) * 20);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const float4 v = zt4[g4];
              zB[4 * g4] = v.x; zB[4 * g4 + 1] = v.y; zB[4 * g4 + 2] = v.z; zB[4 * g4 + 3] = v.w;
            }
          }

          f32x16 f, va = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
          float gdx[4] = {0.f, 0.f, 0.f, 0.f};      // d(a.f)/d(dX_c), this lane's 4 channels (DCOEFF)
          int opaque = 0;
          asm volatile("" : "+v"(opaque));          // keeps the image reads inside the stage (no hoisting)
          const float4* wys = wy + opaque;
          const float4* wvs = wv + opaque;
          const float4* bys = by + opaque;
#pragma unroll
          for (int T = 0; T < 8; ++T) {
            // ---- Y tile
            f32x16 y;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const float4 b4 = bys[T * 8 + g4];
              y[4 * g4] = b4.x; y[4 * g4 + 1] = b4.y; y[4 * g4 + 2] = b4.z; y[4 * g4 + 3] = b4.w;
            }
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const float4 a4 = wys[(4 * T + g4) * 64];
              y = mfma(a4.x, yst[4 * g4], y);
              y = mfma(a4.y, yst[4 * g4 + 1], y);
              y = mfma(a4.z, yst[4 * g4 + 2], y);
              y = mfma(a4.w, yst[4 * g4 + 3], y);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- activation, contraction with dX, dL/dY
            // adjoint state of units 4T..4T+3 in every lane: swap32 of two copies broadcasts both halves' values
            float ae0 = ast[2 * T], ao0 = ast[2 * T], ae1 = ast[2 * T + 1], ao1 = ast[2 * T + 1];
            swap32(ae0, ao0);
            swap32(ae1, ao1);
            const float a4u[4] = {ae0, ao0, ae1, ao1};
            float ps[4], g[16];
#pragma unroll
            for (int hl = 0; hl < 4; ++hl) {
              float acc = 0.f;
              const f32x2 tp[2] = {activate2<ACT>(y[4 * hl], y[4 * hl + 1]), activate2<ACT>(y[4 * hl + 2], y[4 * hl + 3])};
#pragma unroll
              for (int cl = 0; cl < 4; ++cl) {
                const float t = tp[cl >> 1][cl & 1];
                acc = cl == 0 ? t * dh[0] : __builtin_fmaf(t, dh[cl], acc);
                g[4 * hl + cl] = a4u[hl] * (dh[cl] * activate_slope<ACT>(t));
                if constexpr (DCOEFF) gdx[cl] = __builtin_fmaf(a4u[hl], t, gdx[cl]);
              }
              ps[hl] = acc;
            }
            swap32(ps[0], ps[1]);
            swap32(ps[2], ps[3]);
            f[2 * T] = ps[0] + ps[1];
            f[2 * T + 1] = ps[2] + ps[3];
            __builtin_amdgcn_sched_barrier(0);
            // ---- va += W_T^T g
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const float4 a4 = wvs[(4 * T + g4) * 64];
              va = mfma(a4.x, g[4 * g4], va);
              va = mfma(a4.y, g[4 * g4 + 1], va);
              va = mfma(a4.z, g[4 * g4 + 2], va);
              va = mfma(a4.w, g[4 * g4 + 3], va);
            }
            // ---- dW_T += (wq g)^T z through the transposed scratch tile
            {
              float* wg = scr_g + ((n & 1) * 32 + 4 * half) * 20 + (n >> 1);     // + row(r) * 20, row = (r&3) + 8*(r>>2)
#pragma unroll
              for (int r = 0; r < 16; ++r) wg[((r & 3) + 8 * (r >> 2)) * 20] = g[r] * wq;
              wave_lds_sync();
              const float4* g4p = reinterpret_cast<const float4*>(scr_g + (half * 32 + n) * 20);
              float gA[16];
#pragma unroll
              for (int g4 = 0; g4 < 4; ++g4) {
                const float4 v = g4p[g4];
                gA[4 * g4] = v.x; gA[4 * g4 + 1] = v.y; gA[4 * g4 + 2] = v.z; gA[4 * g4 + 3] = v.w;
              }
              wave_lds_sync();                       // reads retired before the next tile overwrites the scratch
              float rs = 0.f;
#pragma unroll
              for (int s2 = 0; s2 < 16; ++s2) {
                accW[T] = mfma(gA[s2], zB[s2], accW[T]);
                rs += gA[s2];
              }
              gb[T] += rs;
            }
            __builtin_amdgcn_sched_barrier(0);       // one tile at a time: bounds the live registers
          }

          if constexpr (DCOEFF) {
#pragma unroll
            for (int cl = 0; cl < 4; ++cl) {
              const float w = wq * gdx[cl];
              if (DEGREE == CDE_PATH_CUBIC) { gc0[cl] += w; gc1[cl] += w * frac; gc2[cl] += w * frac * frac; }
              else { gc0[cl] -= w / width; gc1[cl] += w / width; }
            }
            if (nidx != idx) flush_control_grad(idx);
          }
          const f32x16 ky = -f, ka = va;
          const float third = (float)(1.0 / 3.0);
          if (stage == 0) {
            ky1 = ky; ka1 = ka;
            yst = y0 + ds * ky1 * third;
            ast = a0 + ds * ka1 * third;
          } else if (stage == 1) {
            ky2 = ky; ka2 = ka;
            yst = y0 + ds * (ky2 - ky1 * third);
            ast = a0 + ds * (ka2 - ka1 * third);
          } else if (stage == 2) {
            yst = y0 + ds * (ky1 - ky2 + ky);
            ast = a0 + ds * (ka1 - ka2 + ka);
            ky1 = ky1 + 3.f * (ky2 + ky);
            ka1 = ka1 + 3.f * (ka2 + ka);
          } else {
            yst = y0 + (ky1 + ky) * ds * 0.125f;
            ast = a0 + (ka1 + ka) * ds * 0.125f;
          }
          idx = nidx; frac = nfrac;
        }
        y0 = yst; a0 = ast;
      }
      flush_control_grad(idx);                 // end of this output interval's sweep
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int u = 2 * r + half;
      if (u < Hr) {
        y0[r] = z_saved[(sc * n_out + (i_out - 1)) * Hr + u];
        if (valid) a0[r] += grad_out[(sc * n_out + (i_out - 1)) * Hr + u];
      }
    }
  }
  if (valid) {
#pragma unroll
    for (int r = 0; r < 16; ++r) if (2 * r + half < Hr) grad_z0[series * Hr + 2 * r + half] = a0[r];
  }
  // per-wave partials in the K3 layout: tile T register r of lane (n, half) is dW[h = 4T + (r>>2)][c = (r&3) + 4 half][k = n]
#pragma unroll
  for (int T = 0; T < 8; ++T) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int h = 4 * T + (r >> 2), c = (r & 3) + 4 * half;
      my_partial[(h * MC + c) * MH + n] = accW[T][r];
    }
    // row sums: lane (i = n, half) summed row i = (h = 4T + (n>>3), c = n&7) over the series of its parity
    const float other = __shfl_xor(gb[T], 32, 64);
    if (half == 0) my_partial[MH * MC * MH + 32 * T + n] = gb[T] + other;
  }
}

// Sum the per-wave partials in a FIXED order (run-to-run deterministic: reference test/test_tricks.py:111-131 compares
// gradients bitwise), in two passes: pass 1 adds up groups of consecutive tiles -- one lane per entry of the padded
// (32, 8, 32) + (32, 8) layout and group, the group's sum written over its first tile (a lane only ever touches its own
// entry) -- pass 2 adds the at most 16 group sums in group order, one lane per REAL gradient entry.  (Round 4: one pass
// with 33 workgroups walking all 1,024 tiles took 83 us at 32768 series -- 0.4 TB/s; pass 1 now runs 33 x 16 workgroups.)
constexpr int REDUCE_GROUPS = 16;
__global__ __launch_bounds__(256) void reduce_mfma_partials_groups(float* __restrict__ partial, int64_t n_tiles, int64_t group) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t t0 = (int64_t)blockIdx.y * group;
  if (e >= PARTIAL_FLOATS || t0 >= n_tiles) return;
  const int64_t t1 = t0 + group < n_tiles ? t0 + group : n_tiles;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int64_t t = t0;
  for (; t + 3 < t1; t += 4) {
    s0 += partial[(t + 0) * PARTIAL_FLOATS + e];
    s1 += partial[(t + 1) * PARTIAL_FLOATS + e];
    s2 += partial[(t + 2) * PARTIAL_FLOATS + e];
    s3 += partial[(t + 3) * PARTIAL_FLOATS + e];
  }
  for (; t < t1; ++t) s0 += partial[t * PARTIAL_FLOATS + e];
  partial[t0 * PARTIAL_FLOATS + e] = (s0 + s1) + (s2 + s3);
}
__global__ __launch_bounds__(256) void reduce_mfma_partials(const float* __restrict__ partial, int64_t n_tiles, int64_t group,
                                                            float* __restrict__ grad_W, float* __restrict__ grad_b,
                                                            Dims d) {
  // one lane per REAL gradient entry; `e` is its position in the padded (32, 8, 32) + (32, 8) partial layout
  const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t n_w = (int64_t)d.H * d.C * d.H, n_b = (int64_t)d.H * d.C;
  if (id >= n_w + n_b) return;
  int64_t e;
  if (id < n_w) { const int64_t k = id % d.H, hc = id / d.H, c = hc % d.C, h = hc / d.C; e = (h * MC + c) * MH + k; }
  else { const int64_t hc = id - n_w, c = hc % d.C, h = hc / d.C; e = (int64_t)MH * MC * MH + h * MC + c; }
  float sum = 0.f;
  for (int64_t t = 0; t < n_tiles; t += group) sum += partial[t * PARTIAL_FLOATS + e];
  if (id < n_w) grad_W[id] = sum; else grad_b[id - n_w] = sum;
}

// ------------------------------------------------------------------------------------------ host side
// shapes the two-layer kernels take: the 16 output tiles hold 32 units x 8 channels or 16 units x 16 channels
bool mlp_shape_ok(int64_t C, int64_t H, int64_t width) {
  return width >= 1 && width <= MW && H >= 1 && C >= 1 && ((C <= MC && H <= MH) || (C <= 16 && H <= 16));
}
// ... and 32 units x 16 channels (cde_mfma.h: MlpHi): the kernels that take the upper half from the raw tensors
bool mlp_shape_hi(int64_t C, int64_t H, int64_t width) {
  return width >= 4 && width <= MW && (width & 3) == 0 && C > MC && C <= 16 && H > 16 && H <= MH;
}
// ... and the sweeps, which read a zero-padded copy of the upper rows (cde_mlp_adj.h: mlp_adj_hi): any width
bool mlp_shape_upper(int64_t C, int64_t H, int64_t width) {
  return width >= 1 && width <= MW && C > MC && C <= 16 && H > 16 && H <= MH;
}

int launch_reduce_partials(const float* partial, int64_t n_tiles, void* grad_W, void* grad_b, int H, int C, hipStream_t s) {
  // (`partial` is the caller's scratch: pass 1 overwrites the first tile of every group with the group's sum)
  const int64_t group = (n_tiles + REDUCE_GROUPS - 1) / REDUCE_GROUPS > 0 ? (n_tiles + REDUCE_GROUPS - 1) / REDUCE_GROUPS : 1;
  const unsigned n_groups = (unsigned)((n_tiles + group - 1) / group);
  if (n_tiles > 0)
    reduce_mfma_partials_groups<<<dim3((unsigned)((PARTIAL_FLOATS + 255) / 256), n_groups), 256, 0, s>>>(
        const_cast<float*>(partial), n_tiles, group);
  reduce_mfma_partials<<<(unsigned)((H * C * H + H * C + 255) / 256), 256, 0, s>>>(partial, n_tiles, group, (float*)grad_W,
                                                                                 (float*)grad_b, Dims{H, C});
  return check_launch();
}

bool mfma_applicable(int64_t C, int64_t H, int dtype, int act, bool adjoint) {
  (void)adjoint;
  const bool act_ok = act == CDE_ACT_NONE || act == CDE_ACT_TANH;
  return dtype == CDE_F32 && H >= 1 && H <= MH && C >= 1 && C <= MC && act_ok;
}

size_t mfma_adjoint_partial_bytes(int64_t B) { return (size_t)((B + 31) / 32) * PARTIAL_FLOATS * sizeof(float); }

template <typename TT>
int launch_forward_mfma(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                        const void* bias, int act, const void* z0, const void* grid, int64_t n_grid, const void* t_out,
                        int64_t n_out, void* z_out, int64_t B, int64_t C, int64_t H, const int64_t* stage_index,
                        const void* stage_frac, hipStream_t s) {
  const Dims dims{(int)H, (int)C};
  // product form: 8 waves x 16 series per workgroup (one per CU at B = 32768); activation form: 4 waves x 16 series
#define CDE_FWD(D, A)                                                                                               \
  rk4_forward_mfma<TT, D, A><<<(unsigned)((B + fwd_block_threads<A, false>() / 4 - 1) / (fwd_block_threads<A, false>() / 4)), \
                               fwd_block_threads<A, false>(),                                                       \
                               (A == CDE_ACT_NONE ? W16_FLOATS : ACT16_LDS_FLOATS) * sizeof(float), s>>>(           \
      (const float*)coeffs, (const float*)knots, n_intervals, (const float*)W, (const float*)bias, (const float*)z0, \
      (const TT*)grid, n_grid, (const TT*)t_out, n_out, (float*)z_out, B, stage_index, (const float*)stage_frac, dims)
  if (degree != CDE_PATH_CUBIC && degree != CDE_PATH_LINEAR) return CDE_ERR_UNSUPPORTED;
  if (act == CDE_ACT_NONE) {
    if (degree == CDE_PATH_CUBIC) CDE_FWD(CDE_PATH_CUBIC, CDE_ACT_NONE); else CDE_FWD(CDE_PATH_LINEAR, CDE_ACT_NONE);
  } else if (act == CDE_ACT_TANH) {
    if (degree == CDE_PATH_CUBIC) CDE_FWD(CDE_PATH_CUBIC, CDE_ACT_TANH); else CDE_FWD(CDE_PATH_LINEAR, CDE_ACT_TANH);
  } else return CDE_ERR_UNSUPPORTED;
#undef CDE_FWD
  return check_launch();
}

template <typename TT>
int launch_forward_mlp(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W1,
                       const void* bias1, int64_t width, const void* W2, const void* bias2, int act, const void* z0,
                       const void* grid, int64_t n_grid, const void* t_out, int64_t n_out, void* z_out, int64_t B,
                       int64_t C, int64_t H, const int64_t* stage_index, const void* stage_frac, hipStream_t s) {
  // (32 units x 16 channels: the upper unit groups straight from W2 / bias2 -- 16-byte rows)
  const bool upper = mlp_shape_hi(C, H, width) && ((uintptr_t)W2 & 15) == 0;
  if (!mlp_shape_ok(C, H, width) && !upper) return CDE_ERR_UNSUPPORTED;
  if (degree != CDE_PATH_CUBIC && degree != CDE_PATH_LINEAR) return CDE_ERR_UNSUPPORTED;
  if (act != CDE_ACT_NONE && act != CDE_ACT_TANH) return CDE_ERR_UNSUPPORTED;
  const Dims dims{(int)H, (int)C};
  const unsigned blocks = (unsigned)((B + 127) / 128);
  const size_t lds = (size_t)MLP16_LDS_FLOATS * sizeof(float);
  const bool wide = C > MC;                 // 16 channels x 16 (or, `upper`, 32) hidden units on the same 16 tiles
  // up to 768 tiles (three rounds of one workgroup per CU still beat 8 tiles per workgroup on a quarter of the CUs): the 8
  // waves of a workgroup share a tile (K2m's split form)
  const int64_t tiles = (B + 15) / 16;
  const bool split = tiles <= 768 && !option(CDE_OPT_K2M_NO_SPLIT);
  const size_t lds_split = lds + (8 * 64 + 8 * 64 * 4) * sizeof(float);     // f window + (8-channel tiles) the u window
#define CDE_FWD_CT(D, A, CTV, HIV)                                                                                  \
  do {                                                                                                              \
    if (split) {                                                                                                    \
      (void)hipFuncSetAttribute((const void*)rk4_forward_mfma<TT, D, A, true, CTV, true, false, CDE_METHOD_RK4, HIV>, \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_split);                        \
      rk4_forward_mfma<TT, D, A, true, CTV, true, false, CDE_METHOD_RK4, HIV><<<(unsigned)tiles, 512, lds_split, s>>>( \
          (const float*)coeffs, (const float*)knots, n_intervals, (const float*)W2, (const float*)bias2,            \
          (const float*)z0, (const TT*)grid, n_grid, (const TT*)t_out, n_out, (float*)z_out, B, stage_index,        \
          (const float*)stage_frac, dims, (const float*)W1, (const float*)bias1, (int)width);                       \
    } else {                                                                                                        \
      (void)hipFuncSetAttribute((const void*)rk4_forward_mfma<TT, D, A, true, CTV, false, false, CDE_METHOD_RK4, HIV>, \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                              \
      rk4_forward_mfma<TT, D, A, true, CTV, false, false, CDE_METHOD_RK4, HIV><<<blocks, 512, lds, s>>>(            \
          (const float*)coeffs, (const float*)knots, n_intervals, (const float*)W2, (const float*)bias2,            \
          (const float*)z0, (const TT*)grid, n_grid, (const TT*)t_out, n_out, (float*)z_out, B, stage_index,        \
          (const float*)stage_frac, dims, (const float*)W1, (const float*)bias1, (int)width);                       \
    }                                                                                                               \
  } while (0)
#define CDE_FWD(D, A)                                                                                               \
  do {                                                                                                              \
    if (upper) CDE_FWD_CT(D, A, 16, true); else if (wide) CDE_FWD_CT(D, A, 16, false); else CDE_FWD_CT(D, A, MC, false); \
  } while (0)
  if (act == CDE_ACT_NONE) {
    if (degree == CDE_PATH_CUBIC) CDE_FWD(CDE_PATH_CUBIC, CDE_ACT_NONE); else CDE_FWD(CDE_PATH_LINEAR, CDE_ACT_NONE);
  } else {
    if (degree == CDE_PATH_CUBIC) CDE_FWD(CDE_PATH_CUBIC, CDE_ACT_TANH); else CDE_FWD(CDE_PATH_LINEAR, CDE_ACT_TANH);
  }
#undef CDE_FWD
#undef CDE_FWD_CT
  return check_launch();
}

// K2 with the stage states stored (adjoint=False backward: rk4_backprop.hip).  Affine field, f32, H <= 32, C <= 8.
template <typename TT>
int launch_forward_mfma_stages(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                               const void* bias, int act, const void* z0, const void* grid, int64_t n_grid, const void* t_out,
                               int64_t n_out, void* z_out, void* stages, int64_t B, int64_t C, int64_t H,
                               const int64_t* stage_index, const void* stage_frac, hipStream_t s) {
  const Dims dims{(int)H, (int)C};
  constexpr int threads = fwd_block_threads<CDE_ACT_NONE, false>();
#define CDE_FWD_S(D, A)                                                                                               \
  rk4_forward_mfma<TT, D, A, false, MC, false, true>                                                                  \
      <<<(unsigned)((B + threads / 4 - 1) / (threads / 4)), threads,                                                  \
         (A == CDE_ACT_NONE ? W16_FLOATS : ACT16_LDS_FLOATS) * sizeof(float), s>>>(                                   \
          (const float*)coeffs, (const float*)knots, n_intervals, (const float*)W, (const float*)bias,                \
          (const float*)z0, (const TT*)grid, n_grid, (const TT*)t_out, n_out, (float*)z_out, B, stage_index,          \
          (const float*)stage_frac, dims, nullptr, nullptr, 0, (float*)stages)
  if (degree != CDE_PATH_CUBIC && degree != CDE_PATH_LINEAR) return CDE_ERR_UNSUPPORTED;
  if (act == CDE_ACT_NONE) {
    if (degree == CDE_PATH_CUBIC) CDE_FWD_S(CDE_PATH_CUBIC, CDE_ACT_NONE); else CDE_FWD_S(CDE_PATH_LINEAR, CDE_ACT_NONE);
  } else if (act == CDE_ACT_TANH) {
    if (degree == CDE_PATH_CUBIC) CDE_FWD_S(CDE_PATH_CUBIC, CDE_ACT_TANH); else CDE_FWD_S(CDE_PATH_LINEAR, CDE_ACT_TANH);
  } else return CDE_ERR_UNSUPPORTED;
#undef CDE_FWD_S
  return check_launch();
}
template int launch_forward_mfma_stages<float>(const void*, const void*, int64_t, int, const void*, const void*, int,
                                               const void*, const void*, int64_t, const void*, int64_t, void*, void*, int64_t,
                                               int64_t, int64_t, const int64_t*, const void*, hipStream_t);
template int launch_forward_mfma_stages<double>(const void*, const void*, int64_t, int, const void*, const void*, int,
                                                const void*, const void*, int64_t, const void*, int64_t, void*, void*, int64_t,
                                                int64_t, int64_t, const int64_t*, const void*, hipStream_t);

// K2m with the stage states stored (adjoint=False backward of the two-layer field): the one-wave-per-tile form at any batch
template <typename TT>
int launch_forward_mlp_stages(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W1,
                              const void* bias1, int64_t width, const void* W2, const void* bias2, int act, const void* z0,
                              const void* grid, int64_t n_grid, const void* t_out, int64_t n_out, void* z_out, void* stages,
                              int64_t B, int64_t C, int64_t H, const int64_t* stage_index, const void* stage_frac,
                              hipStream_t s) {
  if (!mlp_shape_ok(C, H, width) && !(mlp_shape_hi(C, H, width) && ((uintptr_t)W2 & 15) == 0)) return CDE_ERR_UNSUPPORTED;
  if (degree != CDE_PATH_CUBIC && degree != CDE_PATH_LINEAR) return CDE_ERR_UNSUPPORTED;
  if (act != CDE_ACT_NONE && act != CDE_ACT_TANH) return CDE_ERR_UNSUPPORTED;
  const Dims dims{(int)H, (int)C};
  const unsigned blocks = (unsigned)((B + 127) / 128);
  const size_t lds = (size_t)MLP16_LDS_FLOATS * sizeof(float);
#define CDE_FWD_MS(D, A, CTV, HIV)                                                                                  \
  do {                                                                                                              \
    (void)hipFuncSetAttribute((const void*)rk4_forward_mfma<TT, D, A, true, CTV, false, true, CDE_METHOD_RK4, HIV>, \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                \
    rk4_forward_mfma<TT, D, A, true, CTV, false, true, CDE_METHOD_RK4, HIV><<<blocks, 512, lds, s>>>(               \
        (const float*)coeffs, (const float*)knots, n_intervals, (const float*)W2, (const float*)bias2,              \
        (const float*)z0, (const TT*)grid, n_grid, (const TT*)t_out, n_out, (float*)z_out, B, stage_index,          \
        (const float*)stage_frac, dims, (const float*)W1, (const float*)bias1, (int)width, (float*)stages);         \
  } while (0)
#define CDE_FWD_MSD(D, A)                                                                                           \
  do {                                                                                                              \
    if (H > 16 && C > MC) CDE_FWD_MS(D, A, 16, true); else if (C > MC) CDE_FWD_MS(D, A, 16, false); else CDE_FWD_MS(D, A, MC, false); \
  } while (0)
  if (act == CDE_ACT_NONE) {
    if (degree == CDE_PATH_CUBIC) CDE_FWD_MSD(CDE_PATH_CUBIC, CDE_ACT_NONE); else CDE_FWD_MSD(CDE_PATH_LINEAR, CDE_ACT_NONE);
  } else {
    if (degree == CDE_PATH_CUBIC) CDE_FWD_MSD(CDE_PATH_CUBIC, CDE_ACT_TANH); else CDE_FWD_MSD(CDE_PATH_LINEAR, CDE_ACT_TANH);
  }
#undef CDE_FWD_MSD
#undef CDE_FWD_MS
  return check_launch();
}
template int launch_forward_mlp_stages<float>(const void*, const void*, int64_t, int, const void*, const void*, int64_t,
                                              const void*, const void*, int, const void*, const void*, int64_t, const void*,
                                              int64_t, void*, void*, int64_t, int64_t, int64_t, const int64_t*, const void*,
                                              hipStream_t);
template int launch_forward_mlp_stages<double>(const void*, const void*, int64_t, int, const void*, const void*, int64_t,
                                               const void*, const void*, int, const void*, const void*, int64_t, const void*,
                                               int64_t, void*, void*, int64_t, int64_t, int64_t, const int64_t*, const void*,
                                               hipStream_t);

// midpoint / euler through the same kernel (affine field, f32, H <= 32, C <= 8)
template <typename TT>
int launch_forward_mfma_method(int method, const void* coeffs, const void* knots, int64_t n_intervals, int degree,
                               const void* W, const void* bias, const void* z0, const void* grid, int64_t n_grid,
                               const void* t_out, int64_t n_out, void* z_out, int64_t B, int64_t C, int64_t H,
                               const int64_t* stage_index, const void* stage_frac, hipStream_t s) {
  const Dims dims{(int)H, (int)C};
  constexpr int threads = fwd_block_threads<CDE_ACT_NONE, false>();
#define CDE_FWD_M(D, M)                                                                                               \
  rk4_forward_mfma<TT, D, CDE_ACT_NONE, false, MC, false, false, M>                                                   \
      <<<(unsigned)((B + threads / 4 - 1) / (threads / 4)), threads, W16_FLOATS * sizeof(float), s>>>(                \
          (const float*)coeffs, (const float*)knots, n_intervals, (const float*)W, (const float*)bias,                \
          (const float*)z0, (const TT*)grid, n_grid, (const TT*)t_out, n_out, (float*)z_out, B, stage_index,          \
          (const float*)stage_frac, dims)
  if (degree != CDE_PATH_CUBIC && degree != CDE_PATH_LINEAR) return CDE_ERR_UNSUPPORTED;
  if (method == CDE_METHOD_MIDPOINT) {
    if (degree == CDE_PATH_CUBIC) CDE_FWD_M(CDE_PATH_CUBIC, CDE_METHOD_MIDPOINT); else CDE_FWD_M(CDE_PATH_LINEAR, CDE_METHOD_MIDPOINT);
  } else if (method == CDE_METHOD_EULER) {
    if (degree == CDE_PATH_CUBIC) CDE_FWD_M(CDE_PATH_CUBIC, CDE_METHOD_EULER); else CDE_FWD_M(CDE_PATH_LINEAR, CDE_METHOD_EULER);
  } else return CDE_ERR_UNSUPPORTED;
#undef CDE_FWD_M
  return check_launch();
}
template int launch_forward_mfma_method<float>(int, const void*, const void*, int64_t, int, const void*, const void*, const void*,
                                               const void*, int64_t, const void*, int64_t, void*, int64_t, int64_t, int64_t,
                                               const int64_t*, const void*, hipStream_t);
template int launch_forward_mfma_method<double>(int, const void*, const void*, int64_t, int, const void*, const void*, const void*,
                                                const void*, int64_t, const void*, int64_t, void*, int64_t, int64_t, int64_t,
                                                const int64_t*, const void*, hipStream_t);

// which of the two adjoint kernels of the affine field runs: K3j (shared Jacobian) unless CDE_OPT_K3_FORM = 1 asks for K3
// (the tests run both against the oracle; scripts compare their timings)
static bool k3_form_jacobian() {
  return option(CDE_OPT_K3_FORM) != 1;
}

// K3p (rk4_adjoint_pair.hip): K3j as a chain wave + a helper wave per tile, two waves per SIMD.  CDE_OPT_K3_WAVES = 1 / 2 picks the form (tests run both; bitwise the same results)
template <typename TT>
int launch_adjoint_jacobian_pair(const void*, const void*, int64_t, int, const void*, const void*, const void*, const void*,
                                 const void*, const int64_t*, int64_t, void*, void*, void*, int64_t, int64_t, int64_t,
                                 const int64_t*, const void*, float*, hipStream_t, int method);
constexpr bool K3_PAIR_DEFAULT = true;       // 5.26 -> 5.01 ms on the headline workload (profiles/r05_k3_pair_b.log)
static bool k3_form_pair() {
  const int64_t e = option(CDE_OPT_K3_WAVES);
  return e ? e == 2 : K3_PAIR_DEFAULT;
}

template <typename TT>
int launch_adjoint_mfma(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                        const void* bias, int act, const void* z_saved, const void* grad_out, const void* sgrid,
                        const int64_t* seg_off, int64_t n_out, void* grad_z0, void* grad_W, void* grad_b, int64_t B,
                        int64_t C, int64_t H, const int64_t* stage_index, const void* stage_frac, float* partial,
                        void* grad_coeffs, hipStream_t s) {
  const Dims dims{(int)H, (int)C};
  const unsigned blocks = (unsigned)((B + 127) / 128);
  if (degree != CDE_PATH_CUBIC && degree != CDE_PATH_LINEAR) return CDE_ERR_UNSUPPORTED;
#define CDE_ADJ(KERNEL, LDS_FLOATS)                                                                                  \
  do {                                                                                                               \
    const size_t lds = (size_t)(LDS_FLOATS) * sizeof(float);                                                         \
    (void)hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
    KERNEL<<<blocks, 256, lds, s>>>((const float*)coeffs, (const float*)knots, n_intervals, (const float*)W,         \
                                    (const float*)bias, (const float*)z_saved, (const float*)grad_out,               \
                                    (const TT*)sgrid, seg_off, n_out, (float*)grad_z0, partial, B, stage_index,      \
                                    (const float*)stage_frac, dims);                                                 \
  } while (0)
#define CDE_ADJ_DX(KERNEL)                                                                                           \
  do {                                                                                                               \
    const size_t lds = (size_t)(ACT_ADJ_LDS_FLOATS) * sizeof(float);                                                 \
    (void)hipFuncSetAttribute((const void*)KERNEL, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);            \
    KERNEL<<<blocks, 256, lds, s>>>((const float*)coeffs, (const float*)knots, n_intervals, (const float*)W,         \
                                    (const float*)bias, (const float*)z_saved, (const float*)grad_out,               \
                                    (const TT*)sgrid, seg_off, n_out, (float*)grad_z0, partial, B, stage_index,      \
                                    (const float*)stage_frac, dims, (float*)grad_coeffs);                            \
  } while (0)
  if (grad_coeffs) {                       // control gradients: the pre-activation kernel (it holds act(Y) in-lane)
    if (act == CDE_ACT_NONE) {
      if (degree == CDE_PATH_CUBIC) CDE_ADJ_DX((rk4_adjoint_act_mfma<TT, CDE_PATH_CUBIC, CDE_ACT_NONE, true>));
      else CDE_ADJ_DX((rk4_adjoint_act_mfma<TT, CDE_PATH_LINEAR, CDE_ACT_NONE, true>));
    } else if (act == CDE_ACT_TANH) {
      if (degree == CDE_PATH_CUBIC) CDE_ADJ_DX((rk4_adjoint_act_mfma<TT, CDE_PATH_CUBIC, CDE_ACT_TANH, true>));
      else CDE_ADJ_DX((rk4_adjoint_act_mfma<TT, CDE_PATH_LINEAR, CDE_ACT_TANH, true>));
    } else return CDE_ERR_UNSUPPORTED;
  } else if (act == CDE_ACT_NONE && k3_form_jacobian() && k3_form_pair()) {
    return launch_adjoint_jacobian_pair<TT>(coeffs, knots, n_intervals, degree, W, bias, z_saved, grad_out, sgrid, seg_off,
                                            n_out, grad_z0, grad_W, grad_b, B, C, H, stage_index, stage_frac, partial, s,
                                            CDE_METHOD_RK4);
  } else if (act == CDE_ACT_NONE && k3_form_jacobian()) {
    if (degree == CDE_PATH_CUBIC) CDE_ADJ((rk4_adjoint_jacobian<TT, CDE_PATH_CUBIC>), WJ_FLOATS + 4 * SCR_FLOATS);
    else CDE_ADJ((rk4_adjoint_jacobian<TT, CDE_PATH_LINEAR>), WJ_FLOATS + 4 * SCR_FLOATS);
  } else if (act == CDE_ACT_NONE) {
    if (degree == CDE_PATH_CUBIC) CDE_ADJ((rk4_adjoint_mfma<TT, CDE_PATH_CUBIC>), W1_FLOATS + W2_FLOATS + 4 * SCR_FLOATS);
    else CDE_ADJ((rk4_adjoint_mfma<TT, CDE_PATH_LINEAR>), W1_FLOATS + W2_FLOATS + 4 * SCR_FLOATS);
  } else if (act == CDE_ACT_TANH) {
    if (degree == CDE_PATH_CUBIC) CDE_ADJ((rk4_adjoint_act_mfma<TT, CDE_PATH_CUBIC, CDE_ACT_TANH>), ACT_ADJ_LDS_FLOATS);
    else CDE_ADJ((rk4_adjoint_act_mfma<TT, CDE_PATH_LINEAR, CDE_ACT_TANH>), ACT_ADJ_LDS_FLOATS);
  } else return CDE_ERR_UNSUPPORTED;
#undef CDE_ADJ
#undef CDE_ADJ_DX
  int rc = check_launch();
  if (rc != CDE_OK) return rc;
  return launch_reduce_partials(partial, (B + 31) / 32, grad_W, grad_b, (int)H, (int)C, s);
}

// variant "bf16x3": K3j with its J rows on the bf16 pipe (three-piece operands)
template <typename TT>
int launch_adjoint_jacobian_bx(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                               const void* bias, const void* z_saved, const void* grad_out, const void* sgrid,
                               const int64_t* seg_off, int64_t n_out, void* grad_z0, void* grad_W, void* grad_b, int64_t B,
                               int64_t C, int64_t H, const int64_t* stage_index, const void* stage_frac, float* partial,
                               hipStream_t s) {
  const Dims dims{(int)H, (int)C};
  const unsigned blocks = (unsigned)((B + 127) / 128);
  const size_t lds = (size_t)(WJB_FLOATS + 4 * SCR_FLOATS) * sizeof(float);
#define CDE_ADJ_BX(D)                                                                                                \
  do {                                                                                                               \
    (void)hipFuncSetAttribute((const void*)rk4_adjoint_jacobian<TT, D, true>,                                        \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                 \
    rk4_adjoint_jacobian<TT, D, true><<<blocks, 256, lds, s>>>(                                                      \
        (const float*)coeffs, (const float*)knots, n_intervals, (const float*)W, (const float*)bias,                 \
        (const float*)z_saved, (const float*)grad_out, (const TT*)sgrid, seg_off, n_out, (float*)grad_z0, partial,   \
        B, stage_index, (const float*)stage_frac, dims);                                                             \
  } while (0)
  if (degree == CDE_PATH_CUBIC) CDE_ADJ_BX(CDE_PATH_CUBIC);
  else if (degree == CDE_PATH_LINEAR) CDE_ADJ_BX(CDE_PATH_LINEAR);
  else return CDE_ERR_UNSUPPORTED;
#undef CDE_ADJ_BX
  const int rc = check_launch();
  if (rc != CDE_OK) return rc;
  return launch_reduce_partials(partial, (B + 31) / 32, grad_W, grad_b, (int)H, (int)C, s);
}
template int launch_adjoint_jacobian_bx<float>(const void*, const void*, int64_t, int, const void*, const void*, const void*,
                                               const void*, const void*, const int64_t*, int64_t, void*, void*, void*,
                                               int64_t, int64_t, int64_t, const int64_t*, const void*, float*, hipStream_t);
template int launch_adjoint_jacobian_bx<double>(const void*, const void*, int64_t, int, const void*, const void*, const void*,
                                                const void*, const void*, const int64_t*, int64_t, void*, void*, void*,
                                                int64_t, int64_t, int64_t, const int64_t*, const void*, float*, hipStream_t);

template int launch_forward_mfma<float>(const void*, const void*, int64_t, int, const void*, const void*, int,
                                        const void*, const void*, int64_t, const void*, int64_t, void*, int64_t, int64_t,
                                        int64_t, const int64_t*, const void*, hipStream_t);
template int launch_forward_mfma<double>(const void*, const void*, int64_t, int, const void*, const void*, int,
                                         const void*, const void*, int64_t, const void*, int64_t, void*, int64_t,
                                         int64_t, int64_t, const int64_t*, const void*, hipStream_t);
template int launch_forward_mlp<float>(const void*, const void*, int64_t, int, const void*, const void*, int64_t,
                                       const void*, const void*, int, const void*, const void*, int64_t, const void*,
                                       int64_t, void*, int64_t, int64_t, int64_t, const int64_t*, const void*,
                                       hipStream_t);
template int launch_forward_mlp<double>(const void*, const void*, int64_t, int, const void*, const void*, int64_t,
                                        const void*, const void*, int, const void*, const void*, int64_t, const void*,
                                        int64_t, void*, int64_t, int64_t, int64_t, const int64_t*, const void*,
                                        hipStream_t);
template int launch_adjoint_mfma<float>(const void*, const void*, int64_t, int, const void*, const void*, int,
                                        const void*, const void*, const void*, const int64_t*, int64_t, void*, void*,
                                        void*, int64_t, int64_t, int64_t, const int64_t*, const void*, float*, void*,
                                        hipStream_t);
template int launch_adjoint_mfma<double>(const void*, const void*, int64_t, int, const void*, const void*, int,
                                         const void*, const void*, const void*, const int64_t*, int64_t, void*, void*,
                                         void*, int64_t, int64_t, int64_t, const int64_t*, const void*, float*, void*,
                                         hipStream_t);

}  // namespace cde
