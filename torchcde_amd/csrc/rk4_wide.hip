// rk4_wide.hip -- Kw: RK4 (3/8) solve and continuous-adjoint sweep for affine vector fields BEYOND the 32 x 8 tiles of
// K2/K3 (rk4_mfma.hip) and K2s/K3s (rk4_split.hip):
//     H <= 64, C <= 8      NW = 8 waves per 16-series tile, NB = 2 channel blocks of 4
//     H <= 32, C <= 16     NW = 4,                          NB = 4
// f(t, z) = reshape_{HxC}(act(W z + b)) dX/dt, act = identity or tanh (reference solver.py:117-135 under
// torchdiffeq's rk4 / odeint_adjoint, as K2/K3).  Until round 2 these shapes fell to the one-lane-per-element VALU
// kernels of rk4_generic.hip: 0.9 s per forward solve and 2.3 s per forward + backward at 32768 x 128 x 8, H = 64.
//
// Same decomposition as rk4_split.hip -- ONE WORKGROUP owns 16 series, wave w owns hidden units 8w..8w+7, lane
// (n = l & 15, q = l >> 4) owns units ua = 8w + q and ub = ua + 4 of series n; v_mfma_f32_16x16x4_f32, pre-activation
// form Y = W z + b; the stage state crosses the waves through LDS once per stage -- with the tile counts as template
// parameters:
//   Y tile T = NB*P + tb (P = unit half, tb = channel block): row i <-> (h = 8w + 4P + (i >> 2), c = 4 tb + (i & 3)),
//     so lane (n, q) ends up with all CT = 4 NB channels of its two units after NT = 2 NB tiles x KS = H/4 K steps.
//   adjoint: va partial = W_w^T g over the wave's own 8 CT rows (K step sp = CT*P + c <-> the lane's OWN g register),
//     all HP = 8 NW output units in MV = NW/2 tiles whose rows are permuted so that lane (n, q) holds, for each
//     destination wave, the partial sums of that wave's units; the NW partials meet in LDS.
// What does NOT carry over is K3s' register budget: the two A images (Y and va) are 64 + 64 registers per lane at
// either shape, and dL/dW (H C x H: 128 KB at H = 64) on top of them is three quarters of a CU's register file.  So
// the sweep keeps no parameter gradients: like K3m (rk4_mlp_adjoint.hip) it only integrates (z, a) backwards and
// STREAMS the per-stage factors to HBM,
//     G [row][HP*CT]   w ds dL/dY   (column h*CT + c)          row = (stage, series)
//     Z [row][HP]      the stage value of z
// and the split-K MFMA reduction of mlp_grad_reduce.hip turns them into  dW | db = G^T [Z | 1]  chunk by chunk
// (2.3 KB per series and stage; the host code below sweeps in chunks of steps that fit the workspace).
#include <stdlib.h>

#include <vector>

#include "cde_split.h"

namespace cde {

// ============================================================================================ forward
// rk4_forward_split with the tile counts as parameters (see there for the wave-local K order: local step j <-> global
// step (2w + j) mod KS, so that steps 0 and 1 take the wave's own units and can be issued around the barrier).
template <typename TT, int DEGREE, int ACT, int NW, int NB>
__global__ __launch_bounds__(64 * NW, 2) void rk4_forward_wide(
    const float* __restrict__ coeffs, const float* __restrict__ knots, int64_t n_intervals,
    const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ z0,
    const TT* __restrict__ grid, int64_t n_grid, const TT* __restrict__ t_out, int64_t n_out,
    float* __restrict__ z_out, int64_t B, const int64_t* __restrict__ stage_index,
    const float* __restrict__ stage_frac, Dims dims) {
  using G = Wide<NW, NB>;
  __shared__ __attribute__((aligned(16))) float zbuf[2 * G::ZBUF];
  __shared__ __attribute__((aligned(16))) float dxb[2 * G::DX];
  const int Hr = dims.H, Cr = dims.C;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int64_t series = (int64_t)blockIdx.x * 16 + n;
  const bool valid = series < B;
  const int64_t sc = valid ? series : B - 1;

  float wy[G::NT][G::KS];               // [tile][LOCAL K step]
  f32x4 by[G::NT];
#pragma unroll
  for (int T = 0; T < G::NT; ++T) {
    const int P = T / NB, tb = T % NB;
    const int hA = 8 * w + 4 * P + (n >> 2), cA = 4 * tb + (n & 3);
#pragma unroll
    for (int j = 0; j < G::KS; ++j) {
      const int k = 4 * ((2 * w + j) & (G::KS - 1)) + q;
      wy[T][j] = (hA < Hr && cA < Cr && k < Hr) ? W[(hA * Cr + cA) * Hr + k] : 0.f;
    }
    const int hD = 8 * w + 4 * P + q;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cD = 4 * tb + r;
      by[T][r] = (hD < Hr && cD < Cr) ? bias[hD * Cr + cD] : 0.f;
    }
  }

  const int ua = 8 * w + q, ub = ua + 4;
  float ya = ua < Hr ? z0[sc * Hr + ua] : 0.f, yb = ub < Hr ? z0[sc * Hr + ub] : 0.f;
  auto store = [&](int64_t j, float a, float b) {
    if (valid) {
      float* row = z_out + (series * n_out + j) * Hr;
      if (ua < Hr) row[ua] = a;
      if (ub < Hr) row[ub] = b;
    }
  };
  store(0, ya, yb);
  const int64_t n_steps = n_grid - 1;
  if (n_steps <= 0) return;

  float* zw = zbuf + n * G::ZROW + q * G::KS + 2 * w;       // writer: global K steps 2w, 2w + 1 of quarter q
  const float* zr = zbuf + n * G::ZROW + q * G::KS;         // reader: quarter q
  Feed<DEGREE> feed;
  feed.init(coeffs, knots, stage_index, stage_frac, n_intervals, sc, Cr, G::CPW * w + (q % G::CPW));
  float* dxw = dxb + n * G::DXROW + G::CPW * w + (q % G::CPW);
  const float* dxr = dxb + n * G::DXROW;
  const bool feeds = q < G::CPW;
  int par = 0;
  f32x4 y[G::NT];
  auto reset = [&]() {
#pragma unroll
    for (int T = 0; T < G::NT; ++T) y[T] = by[T];
  };
  auto own_step = [&](int j, float v) {
#pragma unroll
    for (int T = 0; T < G::NT; ++T) y[T] = mfma16(wy[T][j], v, y[T]);
  };
  reset();
  {
    const float d0 = feed.begin(0, (int)(4 * n_steps - 1));
    *reinterpret_cast<float2*>(zw) = make_float2(ya, yb);
    if (feeds) dxw[0] = d0;
    own_step(0, ya);
  }
  __syncthreads();

  int64_t jout = 1;
  for (int64_t k = 0; k < n_steps; ++k) {
    const TT t0 = grid[k], t1 = grid[k + 1];
    const float dt = (float)(t1 - t0);
    float k1a = 0.f, k1b = 0.f, k2a = 0.f, k2b = 0.f, pqa = 0.f, pqb = 0.f, za = ya, zb = yb;
#pragma unroll
    for (int stage = 0; stage < 4; ++stage) {
      float2 zo[NW - 1];                                      // the other waves' units: local steps (2jj, 2jj + 1)
#pragma unroll
      for (int jj = 1; jj < NW; ++jj)
        zo[jj - 1] = *reinterpret_cast<const float2*>(zr + par * G::ZBUF + ((2 * w + 2 * jj) & (G::KS - 1)));
      float4 d4[NB];
#pragma unroll
      for (int tb = 0; tb < NB; ++tb) d4[tb] = *reinterpret_cast<const float4*>(dxr + par * G::DX + 4 * tb);
      __builtin_amdgcn_sched_barrier(0);
      own_step(1, zb);
      __builtin_amdgcn_sched_barrier(0);
      const float dnext = feed.advance((int)(4 * k) + stage);      // next stage's control derivative
      if (feeds) dxw[(par ^ 1) * G::DX] = dnext;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int jj = 1; jj < NW; ++jj) { own_step(2 * jj, zo[jj - 1].x); own_step(2 * jj + 1, zo[jj - 1].y); }
      // contraction with dX: all CT channels of the lane's two units are in its own registers
      f32x2 fpa = {0.f, 0.f}, fpb = {0.f, 0.f};
#pragma unroll
      for (int tb = 0; tb < NB; ++tb) {
        const f32x2 d01 = {d4[tb].x, d4[tb].y}, d23 = {d4[tb].z, d4[tb].w};
        fpa = __builtin_elementwise_fma(activate2<ACT>(y[tb][0], y[tb][1]), d01, fpa);
        fpb = __builtin_elementwise_fma(activate2<ACT>(y[NB + tb][0], y[NB + tb][1]), d01, fpb);
        fpa = __builtin_elementwise_fma(activate2<ACT>(y[tb][2], y[tb][3]), d23, fpa);
        fpb = __builtin_elementwise_fma(activate2<ACT>(y[NB + tb][2], y[NB + tb][3]), d23, fpb);
      }
      const float fa = fpa[0] + fpa[1], fb = fpb[0] + fpb[1];
      // torchdiffeq rk4_alt_step_func (3/8 rule), association order as in K2
      const float third = (float)(1.0 / 3.0);
      if (stage == 0) {
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
      *reinterpret_cast<float2*>(zw + (par ^ 1) * G::ZBUF) = make_float2(za, zb);
      __builtin_amdgcn_sched_barrier(0);
      reset();
      own_step(0, za);                                        // first K step of the next stage
      __syncthreads();
      par ^= 1;
    }
    const float y1a = za, y1b = zb;
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

// ============================================================================================ adjoint sweep
// One chunk of RK steps [k_begin, k_end) of one output interval, reversed time: (z, a) of every series come from and
// go back to y_state / a_state (B x H); the per-stage factors go to Gout / Zout (row = (step - k_begin, stage, series),
// series padded to a multiple of 16).
// All NW waves are alike (rk4_split.hip's chain waves without their helpers: there is no dW product to hand over).
// (4 waves x 4 channel blocks: one wave per SIMD -- the register allocator then has the accumulation registers as spill
// space.  With two workgroups per CU the same kernel spilled ~2 registers per stage to scratch, and every scratch
// reload waits on vmcnt, i.e. for ALL the factor stores in flight.)
template <typename TT, int DEGREE, int ACT, int NW, int NB>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 1 : 2) void rk4_adjoint_wide_sweep(
    const float* __restrict__ coeffs, const float* __restrict__ knots, int64_t n_intervals,
    const float* __restrict__ W, const float* __restrict__ bias, float* __restrict__ y_state,
    float* __restrict__ a_state, const TT* __restrict__ sgrid, int64_t k_begin, int64_t k_end,
    const int64_t* __restrict__ stage_index, const float* __restrict__ stage_frac, float* __restrict__ Gout,
    float* __restrict__ Zout, int64_t B, Dims dims) {
  using G = Wide<NW, NB>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* zbuf = lds;
  float* vab = zbuf + 2 * G::ZBUF;
  float* dxb = vab + 2 * G::VA;
  float* bias_lds = dxb + 2 * G::DX;                                // zero-padded [unit][CT]: the C/D rows of a lane
  const int Hr = dims.H, Cr = dims.C;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int64_t series = (int64_t)blockIdx.x * 16 + n;
  const bool valid = series < B;
  const int64_t sc = valid ? series : B - 1;
  const int64_t B16 = (int64_t)gridDim.x * 16;                      // rows per stage in Gout / Zout
  const float third = (float)(1.0 / 3.0);
  for (int e = threadIdx.x; e < G::GC; e += blockDim.x) {
    const int h = e / G::CT, c = e % G::CT;
    bias_lds[e] = (h < Hr && c < Cr) ? bias[h * Cr + c] : 0.f;
  }

  // A images: Y tiles [tile][K step s: input units 4s + q], va tiles [tile][K step sp = CT*P + c]
  float wy[G::NT][G::KS], wv[G::MV][G::KV];
#pragma unroll
  for (int T = 0; T < G::NT; ++T) {
    const int P = T / NB, tb = T % NB;
    const int hA = 8 * w + 4 * P + (n >> 2), cA = 4 * tb + (n & 3);
#pragma unroll
    for (int s = 0; s < G::KS; ++s) {
      const int k = 4 * s + q;
      wy[T][s] = (hA < Hr && cA < Cr && k < Hr) ? W[(hA * Cr + cA) * Hr + k] : 0.f;
    }
  }
#pragma unroll
  for (int T = 0; T < G::MV; ++T) {
    const int qi = n >> 2, r = n & 3;
    const int k_out = 8 * (2 * T + (r >> 1)) + 4 * (r & 1) + qi;    // row i of tile T -> wave 2T + (r >> 1), unit r & 1, quarter qi
#pragma unroll
    for (int sp = 0; sp < G::KV; ++sp) {
      const int h = 8 * w + 4 * (sp / G::CT) + q, c = sp % G::CT;
      wv[T][sp] = (h < Hr && c < Cr && k_out < Hr) ? W[(h * Cr + c) * Hr + k_out] : 0.f;
    }
  }
  // bias fragment of tile T, register r: bias[h = 8w + 4P + q][c = 4 tb + r]: re-read from LDS every stage (4 NT
  // registers less to hold across the whole sweep)
  const int ua = 8 * w + q, ub = ua + 4;

  float y0a = ua < Hr ? y_state[sc * Hr + ua] : 0.f, y0b = ub < Hr ? y_state[sc * Hr + ub] : 0.f;
  float a0a = (valid && ua < Hr) ? a_state[sc * Hr + ua] : 0.f;     // a == 0 stays 0: padded lanes add nothing to G
  float a0b = (valid && ub < Hr) ? a_state[sc * Hr + ub] : 0.f;
  float* zw = zbuf + n * G::ZROW + q * G::KS + 2 * w;
  const float* zr = zbuf + n * G::ZROW + q * G::KS;
  float* vw = vab + (q * 16 + n) * G::VROW + 2 * w;                 // + w_dst * 64 * VROW
  const float* vr = vab + ((w * 4 + q) * 16 + n) * G::VROW;
  Feed<DEGREE> feed;
  feed.init(coeffs, knots, stage_index, stage_frac, n_intervals, sc, Cr, G::CPW * w + (q % G::CPW));
  float* dxw = dxb + n * G::DXROW + G::CPW * w + (q % G::CPW);
  const float* dxr = dxb + n * G::DXROW;
  const bool feeds = q < G::CPW;
  int par = 0;
  auto read_ka = [&](int pp, float& kaa, float& kab) {
    float ea = 0.f, eb = 0.f, oa = 0.f, ob = 0.f;                   // two chains each, partials of source waves 2i, 2i + 1
#pragma unroll
    for (int i = 0; i < NW / 2; ++i) {
      const float4 p4 = *reinterpret_cast<const float4*>(vr + pp * G::VA + 4 * i);
      ea += p4.x; eb += p4.y; oa += p4.z; ob += p4.w;
    }
    kaa = ea + oa; kab = eb + ob;
  };

  if (k_end <= k_begin) return;
  {
    const float d0 = feed.begin((int)(4 * k_begin), (int)(4 * k_end - 1));
    if (feeds) dxw[par * G::DX] = d0;
    *reinterpret_cast<float2*>(zw + par * G::ZBUF) = make_float2(y0a, y0b);
  }
  spl_barrier();
  float ds_prev = 0.f;
  float ka1a = 0.f, ka1b = 0.f, ka2a = 0.f, ka2b = 0.f, asa = a0a, asb = a0b;
  float pza = y0a, pzb = y0b;                                       // stage value of the lane's own units
  for (int64_t k = k_begin; k < k_end; ++k) {
    const float ds = (float)(sgrid[k + 1] - sgrid[k]);
    float ky1a = 0.f, ky1b = 0.f, ky2a = 0.f, ky2b = 0.f;
#pragma unroll
    for (int stage = 0; stage < 4; ++stage) {
      float4 z4[G::KS / 4];
#pragma unroll
      for (int i = 0; i < G::KS / 4; ++i) z4[i] = *reinterpret_cast<const float4*>(zr + par * G::ZBUF + 4 * i);
      int opaque = 0;
      asm volatile("" : "+v"(opaque));                               // keeps the bias reads below inside the stage
      float kaa = 0.f, kab = 0.f;
      const bool first = stage == 0 && k == k_begin;                 // no previous stage in this chunk
      if (!first) read_ka(par, kaa, kab);
      // ---- a path: RK update that stage e-1 left open
      if (stage == 1) {
        ka1a = kaa; ka1b = kab;
        asa = a0a + ds * ka1a * third; asb = a0b + ds * ka1b * third;
      } else if (stage == 2) {
        ka2a = kaa; ka2b = kab;
        asa = a0a + ds * (ka2a - ka1a * third); asb = a0b + ds * (ka2b - ka1b * third);
      } else if (stage == 3) {
        asa = a0a + ds * (ka1a - ka2a + kaa); asb = a0b + ds * (ka1b - ka2b + kab);
        ka1a = ka1a + 3.f * (ka2a + kaa); ka1b = ka1b + 3.f * (ka2b + kab);
      } else if (!first) {
        asa = a0a + (ka1a + kaa) * ds_prev * 0.125f; asb = a0b + (ka1b + kab) * ds_prev * 0.125f;
        a0a = asa; a0b = asb;
      }
      const float wq = ((stage == 0 || stage == 3) ? 0.125f : 0.375f) * ds;     // 3/8-rule quadrature weight
      // (stage, series) -- rows exist for the padding lanes of the last tile too (their a is 0, so g is 0 and they
      // add nothing): the factor stores are then branch-free and the compiler's s_waitcnt bookkeeping stays exact
      // (under `if (valid)` it could not count them and put an s_waitcnt vmcnt(2) in front of every store).
      const int64_t out_row = ((k - k_begin) * 4 + stage) * B16 + (int64_t)blockIdx.x * 16 + n;
      {
        float* zrow = Zout + out_row * G::HP;
        zrow[ua] = pza; zrow[ub] = pzb;
      }
      // ---- one unit half (NB tiles) at a time: Y tiles, activation, f, g = dL/dY, and that half's K steps of the
      //      va partial (this wave's 8 CT (h, c) rows -> all HP output units; K step sp = CT*P + c takes the lane's
      //      OWN g register) -- g of a half is dead before the next one starts
      f32x2 fp[2] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
      f32x4 v[G::MV];
#pragma unroll
      for (int T = 0; T < G::MV; ++T) v[T] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int P = 0; P < 2; ++P) {
        f32x4 yt[NB];
#pragma unroll
        for (int tb = 0; tb < NB; ++tb) {                            // bias row of the lane's unit (LDS)
          const float4 b4 = *reinterpret_cast<const float4*>(bias_lds + opaque + (P ? ub : ua) * G::CT + 4 * tb);
          yt[tb] = f32x4{b4.x, b4.y, b4.z, b4.w};
        }
#pragma unroll
        for (int i = 0; i < G::KS / 4; ++i) {
          const float zs[4] = {z4[i].x, z4[i].y, z4[i].z, z4[i].w};
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tb = 0; tb < NB; ++tb) yt[tb] = mfma16(wy[NB * P + tb][4 * i + j], zs[j], yt[tb]);
        }
        const float aown = P ? asb : asa;
#pragma unroll
        for (int tb = 0; tb < NB; ++tb) {                            // tile by tile: g of a tile lives in 4 registers
          const float4 d4 = *reinterpret_cast<const float4*>(dxr + opaque + par * G::DX + 4 * tb);
          const float dv[4] = {d4.x, d4.y, d4.z, d4.w};
          f32x2 gq[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const f32x2 dx = {dv[2 * j], dv[2 * j + 1]};
            const f32x2 t = activate2<ACT>(yt[tb][2 * j], yt[tb][2 * j + 1]);
            fp[P] = __builtin_elementwise_fma(t, dx, fp[P]);
            if (ACT == CDE_ACT_NONE) gq[j] = dx * aown;
            else gq[j] = (f32x2{spl_slope<ACT>(t[0]), spl_slope<ACT>(t[1])} * dx) * aown;
          }
          {   // plain stores: as non-temporal ones the four 16-byte pieces of a unit's 64-byte row (16 channels) reach
              // memory as four partial writes -- measured 67 ms instead of 25 ms per forward + backward at H = 32, C = 16
            const f32x2 g0 = gq[0] * wq, g1 = gq[1] * wq;
            *reinterpret_cast<f32x4*>(Gout + out_row * G::GC + (P ? ub : ua) * G::CT + 4 * tb) = f32x4{g0[0], g0[1], g1[0], g1[1]};
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {                              // K steps c = 4 tb + r of this half
            const float gv = gq[r >> 1][r & 1];
#pragma unroll
            for (int T = 0; T < G::MV; ++T) v[T] = mfma16(wv[T][G::CT * P + 4 * tb + r], gv, v[T]);
          }
        }
      }
      const float kya = -(fp[0][0] + fp[0][1]), kyb = -(fp[1][0] + fp[1][1]);      // reverse time: dy/ds = -f
      float nya, nyb;
      if (stage == 0) {
        ky1a = kya; ky1b = kyb;
        nya = y0a + ds * ky1a * third; nyb = y0b + ds * ky1b * third;
      } else if (stage == 1) {
        ky2a = kya; ky2b = kyb;
        nya = y0a + ds * (ky2a - ky1a * third); nyb = y0b + ds * (ky2b - ky1b * third);
      } else if (stage == 2) {
        nya = y0a + ds * (ky1a - ky2a + kya); nyb = y0b + ds * (ky1b - ky2b + kyb);
        ky1a = ky1a + 3.f * (ky2a + kya); ky1b = ky1b + 3.f * (ky2b + kyb);
      } else {
        nya = y0a + (ky1a + kya) * ds * 0.125f; nyb = y0b + (ky1b + kyb) * ds * 0.125f;
      }
      *reinterpret_cast<float2*>(zw + (par ^ 1) * G::ZBUF) = make_float2(nya, nyb);
      pza = nya; pzb = nyb;
      const float dnext = feed.advance((int)(4 * k) + stage);
      if (feeds) dxw[(par ^ 1) * G::DX] = dnext;
      // register r of tile T -> destination wave 2T + (r >> 1), its unit j = r & 1
      float* vwp = vw + (par ^ 1) * G::VA;
#pragma unroll
      for (int T = 0; T < G::MV; ++T) {
        *reinterpret_cast<float2*>(vwp + (2 * T) * 64 * G::VROW) = make_float2(v[T][0], v[T][1]);
        *reinterpret_cast<float2*>(vwp + (2 * T + 1) * 64 * G::VROW) = make_float2(v[T][2], v[T][3]);
      }
      spl_barrier();
      par ^= 1;
    }
    y0a = pza; y0b = pzb;
    ds_prev = ds;
  }
  {                                                                   // the last stage's a-path update
    float kaa, kab;
    read_ka(par, kaa, kab);
    a0a = a0a + (ka1a + kaa) * ds_prev * 0.125f; a0b = a0b + (ka1b + kab) * ds_prev * 0.125f;
  }
  if (valid) {
    if (ua < Hr) { y_state[series * Hr + ua] = y0a; a_state[series * Hr + ua] = a0a; }
    if (ub < Hr) { y_state[series * Hr + ub] = y0b; a_state[series * Hr + ub] = a0b; }
  }
}

// (z, a) at an output time, torchdiffeq's adjoint: z is re-seeded from the stored forward solution, the incoming
// gradient of that output is added to a (first: a starts from it)
__global__ void wide_seed_kernel(float* __restrict__ y_state, float* __restrict__ a_state,
                                 const float* __restrict__ z_saved, const float* __restrict__ grad_out, int64_t j,
                                 int64_t n_out, int64_t B, int H, int first) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * H) return;
  const int64_t b = e / H;
  const int h = (int)(e - b * H);
  const int64_t src = (b * n_out + j) * H + h;
  y_state[e] = z_saved[src];
  a_state[e] = first ? grad_out[src] : a_state[e] + grad_out[src];
}

// acc [m = h*CT + c][HP + 1] = [dW | db]  ->  grad_W (H*C, H), grad_b (H*C)
__global__ void wide_unpack_kernel(const float* __restrict__ acc, float* __restrict__ grad_W, float* __restrict__ grad_b,
                                   int H, int C, int HP, int CT) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= H * C * (H + 1)) return;
  const int m = e / (H + 1), k = e - m * (H + 1);
  const int h = m / C, c = m - h * C;
  const float v = acc[(h * CT + c) * (HP + 1) + (k < H ? k : HP)];
  if (k < H) grad_W[m * H + k] = v; else grad_b[m] = v;
}

// ------------------------------------------------------------------------------------------ host side
// defined in mlp_grad_reduce.hip: acc (M, N + 1) += G^T [Z | 1] over `rows` rows, G (rows, M), Z (rows, N)
int launch_wide_grad_reduce(const float* G, const float* Z, int64_t rows, int M, int N, float* acc, float* partial,
                            hipStream_t s);
size_t wide_grad_reduce_partial_bytes(int M, int N);

bool wide_applicable(int64_t C, int64_t H, int dtype, int act) {
  const bool act_ok = act == CDE_ACT_NONE || act == CDE_ACT_TANH;
  return dtype == CDE_F32 && act_ok && H >= 1 && C >= 1 && ((H <= 64 && C <= 8) || (H <= 32 && C <= 16));
}
static inline bool wide_tall(int64_t C) { return C <= 8; }           // 8 waves x 2 channel blocks, else 4 x 4

template <int NW, int NB>
static size_t wide_sweep_lds() {
  using G = Wide<NW, NB>;
  return (size_t)(2 * G::ZBUF + 2 * G::VA + 2 * G::DX + G::GC) * sizeof(float);
}

template <typename TT>
int launch_forward_wide(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                        const void* bias, int act, const void* z0, const void* grid, int64_t n_grid, const void* t_out,
                        int64_t n_out, void* z_out, int64_t B, int64_t C, int64_t H, const int64_t* stage_index,
                        const void* stage_frac, hipStream_t s) {
  const Dims dims{(int)H, (int)C};
  const unsigned blocks = (unsigned)((B + 15) / 16);
  if (degree != CDE_PATH_CUBIC && degree != CDE_PATH_LINEAR) return CDE_ERR_UNSUPPORTED;
  if (act != CDE_ACT_NONE && act != CDE_ACT_TANH) return CDE_ERR_UNSUPPORTED;
#define CDE_FWD(D, A, NWV, NBV)                                                                                      \
  rk4_forward_wide<TT, D, A, NWV, NBV><<<blocks, 64 * NWV, 0, s>>>(                                                  \
      (const float*)coeffs, (const float*)knots, n_intervals, (const float*)W, (const float*)bias, (const float*)z0, \
      (const TT*)grid, n_grid, (const TT*)t_out, n_out, (float*)z_out, B, stage_index, (const float*)stage_frac, dims)
#define CDE_FWD_SHAPE(D, A)                                                                                          \
  do {                                                                                                               \
    if (wide_tall(C)) CDE_FWD(D, A, 8, 2); else CDE_FWD(D, A, 4, 4);                                                 \
  } while (0)
  if (act == CDE_ACT_NONE) {
    if (degree == CDE_PATH_CUBIC) CDE_FWD_SHAPE(CDE_PATH_CUBIC, CDE_ACT_NONE); else CDE_FWD_SHAPE(CDE_PATH_LINEAR, CDE_ACT_NONE);
  } else {
    if (degree == CDE_PATH_CUBIC) CDE_FWD_SHAPE(CDE_PATH_CUBIC, CDE_ACT_TANH); else CDE_FWD_SHAPE(CDE_PATH_LINEAR, CDE_ACT_TANH);
  }
#undef CDE_FWD_SHAPE
#undef CDE_FWD
  return check_launch();
}

// workspace of the backward pass behind the stage tables: [y | a | acc | reduction partials | G chunk | Z chunk]
struct WideLayout {
  size_t off_y, off_a, off_acc, off_partial, off_G, off_Z, total;
  int64_t chunk_steps;
  int HP, CT;
};
static inline size_t w256(size_t x) { return (x + 255) / 256 * 256; }
constexpr size_t WIDE_SCRATCH_BYTES = (size_t)4 << 30;                // factor chunk: at most 4 GB (288 GB of HBM per GPU)
static size_t wide_scratch_bytes() {                                  // CDE_OPT_WIDE_SCRATCH_BYTES overrides (tests: tiny chunks)
  const int64_t v = option(CDE_OPT_WIDE_SCRATCH_BYTES);
  return v > 0 ? (size_t)v : WIDE_SCRATCH_BYTES;
}

WideLayout wide_layout(int64_t B, int64_t C, int64_t H, int64_t n_steps) {
  WideLayout L;
  L.HP = wide_tall(C) ? 64 : 32;
  L.CT = wide_tall(C) ? 8 : 16;
  const size_t row_bytes = (size_t)(L.HP * L.CT + L.HP) * sizeof(float);
  const int64_t B16 = (B + 15) / 16 * 16;                             // the sweep writes rows for the padding lanes too
  int64_t steps = (int64_t)(wide_scratch_bytes() / (row_bytes * 4 * (size_t)(B16 > 0 ? B16 : 1)));
  steps = steps < 1 ? 1 : steps;
  steps = steps > n_steps ? (n_steps > 0 ? n_steps : 1) : steps;
  L.chunk_steps = steps;
  const size_t rows = (size_t)steps * 4 * (size_t)B16;
  L.off_y = 0;
  L.off_a = L.off_y + w256((size_t)B * H * sizeof(float));
  L.off_acc = L.off_a + w256((size_t)B * H * sizeof(float));
  L.off_partial = L.off_acc + w256((size_t)L.HP * L.CT * (L.HP + 1) * sizeof(float));
  L.off_G = L.off_partial + w256(wide_grad_reduce_partial_bytes(L.HP * L.CT, L.HP));
  L.off_Z = L.off_G + w256(rows * L.HP * L.CT * sizeof(float));
  L.total = L.off_Z + w256(rows * L.HP * sizeof(float));
  return L;
}
size_t wide_adjoint_workspace_bytes(int64_t B, int64_t C, int64_t H, int64_t n_steps) { return wide_layout(B, C, H, n_steps).total; }

template <typename TT>
int launch_adjoint_wide(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                        const void* bias, int act, const void* z_saved, const void* grad_out, const void* sgrid,
                        int64_t n_sgrid, const int64_t* seg_off_host, int64_t n_out, void* grad_z0, void* grad_W,
                        void* grad_b, int64_t B, int64_t C, int64_t H, const int64_t* stage_index,
                        const void* stage_frac, void* scratch, hipStream_t s) {
  const Dims dims{(int)H, (int)C};
  if (degree != CDE_PATH_CUBIC && degree != CDE_PATH_LINEAR) return CDE_ERR_UNSUPPORTED;
  if (act != CDE_ACT_NONE && act != CDE_ACT_TANH) return CDE_ERR_UNSUPPORTED;
  // the chunk loop runs on the host and reads the segment offsets from the caller's HOST copy (`seg_off_host` of the C
  // ABI): no device-to-host copy, no stream synchronisation -- the call only queues work and is graph-capturable
  if (n_out > 1 && !seg_off_host) return CDE_ERR_NULL;
  const WideLayout L = wide_layout(B, C, H, n_sgrid - 1);
  unsigned char* base = (unsigned char*)scratch;
  float* y_state = (float*)(base + L.off_y);
  float* a_state = (float*)(base + L.off_a);
  float* acc = (float*)(base + L.off_acc);
  float* partial = (float*)(base + L.off_partial);
  float* Gbuf = (float*)(base + L.off_G);
  float* Zbuf = (float*)(base + L.off_Z);
  const unsigned blocks = (unsigned)((B + 15) / 16);
  const unsigned seed_blocks = (unsigned)((B * H + 255) / 256);
  zero_async(acc, (size_t)L.HP * L.CT * (L.HP + 1) * sizeof(float), s);
  wide_seed_kernel<<<seed_blocks, 256, 0, s>>>(y_state, a_state, (const float*)z_saved, (const float*)grad_out,
                                               n_out - 1, n_out, B, (int)H, 1);
#define CDE_SWEEP(D, A, NWV, NBV)                                                                                    \
  do {                                                                                                               \
    const size_t lds = wide_sweep_lds<NWV, NBV>();                                                                   \
    (void)hipFuncSetAttribute((const void*)rk4_adjoint_wide_sweep<TT, D, A, NWV, NBV>,                               \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                 \
    rk4_adjoint_wide_sweep<TT, D, A, NWV, NBV><<<blocks, 64 * NWV, lds, s>>>(                                        \
        (const float*)coeffs, (const float*)knots, n_intervals, (const float*)W, (const float*)bias, y_state,        \
        a_state, (const TT*)sgrid, k, ke, stage_index, (const float*)stage_frac, Gbuf, Zbuf, B, dims);               \
  } while (0)
#define CDE_SWEEP_SHAPE(D, A)                                                                                        \
  do {                                                                                                               \
    if (wide_tall(C)) CDE_SWEEP(D, A, 8, 2); else CDE_SWEEP(D, A, 4, 4);                                             \
  } while (0)
  for (int64_t p = 0; p + 1 < n_out; ++p) {
    int64_t k = seg_off_host[p];
    const int64_t k_end = seg_off_host[p + 1] - 1;
    while (k < k_end) {
      const int64_t ke = k + L.chunk_steps < k_end ? k + L.chunk_steps : k_end;
      if (act == CDE_ACT_NONE) {
        if (degree == CDE_PATH_CUBIC) CDE_SWEEP_SHAPE(CDE_PATH_CUBIC, CDE_ACT_NONE); else CDE_SWEEP_SHAPE(CDE_PATH_LINEAR, CDE_ACT_NONE);
      } else {
        if (degree == CDE_PATH_CUBIC) CDE_SWEEP_SHAPE(CDE_PATH_CUBIC, CDE_ACT_TANH); else CDE_SWEEP_SHAPE(CDE_PATH_LINEAR, CDE_ACT_TANH);
      }
      int rc = check_launch();
      if (rc != CDE_OK) return rc;
      rc = launch_wide_grad_reduce(Gbuf, Zbuf, 4 * (ke - k) * ((B + 15) / 16 * 16), L.HP * L.CT, L.HP, acc, partial, s);
      if (rc != CDE_OK) return rc;
      k = ke;
    }
    wide_seed_kernel<<<seed_blocks, 256, 0, s>>>(y_state, a_state, (const float*)z_saved, (const float*)grad_out,
                                                 n_out - 2 - p, n_out, B, (int)H, 0);
  }
#undef CDE_SWEEP_SHAPE
#undef CDE_SWEEP
  if (hipMemcpyAsync(grad_z0, a_state, (size_t)B * H * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess)
    return CDE_ERR_LAUNCH;
  wide_unpack_kernel<<<(unsigned)((H * C * (H + 1) + 255) / 256), 256, 0, s>>>(acc, (float*)grad_W, (float*)grad_b, (int)H,
                                                                               (int)C, L.HP, L.CT);
  return check_launch();
}

template int launch_forward_wide<float>(const void*, const void*, int64_t, int, const void*, const void*, int,
                                        const void*, const void*, int64_t, const void*, int64_t, void*, int64_t, int64_t,
                                        int64_t, const int64_t*, const void*, hipStream_t);
template int launch_forward_wide<double>(const void*, const void*, int64_t, int, const void*, const void*, int,
                                         const void*, const void*, int64_t, const void*, int64_t, void*, int64_t, int64_t,
                                         int64_t, const int64_t*, const void*, hipStream_t);
template int launch_adjoint_wide<float>(const void*, const void*, int64_t, int, const void*, const void*, int,
                                        const void*, const void*, const void*, int64_t, const int64_t*, int64_t, void*,
                                        void*, void*, int64_t, int64_t, int64_t, const int64_t*, const void*, void*,
                                        hipStream_t);
template int launch_adjoint_wide<double>(const void*, const void*, int64_t, int, const void*, const void*, int,
                                         const void*, const void*, const void*, int64_t, const int64_t*, int64_t, void*,
                                         void*, void*, int64_t, int64_t, int64_t, const int64_t*, const void*, void*,
                                         hipStream_t);

}  // namespace cde
