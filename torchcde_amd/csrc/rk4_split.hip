// rk4_split.hip -- latency-oriented RK4 (3/8) solve and continuous-adjoint sweep for SMALL per-GPU batches
// (strong scaling: 32768 series over 8 GPUs leave 4096 per GPU = 16 series per CU).
//
// K2/K3 (rk4_mfma.hip) give one wave 16/32 series for the whole 508-stage chain, so a batch below 32768 leaves SIMDs
// idle and takes as long as a full one.  Here ONE WORKGROUP owns a tile of 16 series and its waves split every
// stage's GEMMs between them; the stage state crosses the waves through LDS once per stage (one s_barrier per stage).
// 4096 series = 256 workgroups = every SIMD of the chip busy.
//
// f32, H <= 32, C <= 8 (zero padded), v_mfma_f32_16x16x4_f32, pre-activation form Y = W z + b (identity or tanh):
//   wave w (0..3, one per SIMD) owns hidden units 8w..8w+7; lane (n = l & 15, q = l >> 4) owns units ua = 8w + q,
//   ub = 8w + 4 + q of series n: their RK bookkeeping (y, k1, k2, ...) lives in that lane only.
//   Y tile T = 2P + tb (P, tb in {0,1}): row i <-> (h = 8w + 4P + (i >> 2), c = 4 tb + (i & 3))
//     => C/D fragment of lane (n, q), register r: Y[h = 8w + 4P + q][c = 4 tb + r]: all 8 channels of the lane's
//     two units after 4 tiles x 8 K steps = 32 MFMAs (a quarter of the 128 of the whole evaluation);
//     K step s: lane quarter kq feeds input unit 4s + kq, read from the stage-state buffer
//     zbuf[series][kq*8 + s]; each lane publishes its two units with one ds_write_b64.
//   adjoint only (96 MFMAs per SIMD and stage, 8 waves: see rk4_adjoint_split8):
//   va partial = W_w^T g            32 MFMAs, K = this wave's 64 (h, c) rows: K step s' = 8P + c, quarter kq <-> the
//                                     lane's OWN register g[h = 8w + 4P + kq][c] (no data movement); output rows are
//                                     permuted so that lane (n, q) holds, for every destination wave w', the partial
//                                     sums of w's units 8w' + q and 8w' + 4 + q; the four partials meet in LDS.
//   dW_w += g^T (wq z)              32 MFMAs, series = MFMA K: g goes through an LDS transpose, z through a second
//                                     (transposed) copy of the stage state; dL/db = row sums of the same tile.
// The A images (Y: 32, va: 32 values per lane) and the dW accumulators (32) live in registers for the whole solve.
// Arithmetic per series is the same as in K2a/K3a up to summation order; per-workgroup partial parameter gradients go
// through the same fixed-order reduction (reduce_mfma_partials), so results are run-to-run deterministic.
// On gfx950 f32 MFMA and VALU instructions of a wave do not overlap (scripts/ubench/mfma_issue.hip): instruction count
// is time, hence packed VALU, VGPR-form MFMA (no v_accvgpr moves; -mllvm -amdgpu-mfma-vgpr-form for this file) and
// the control derivative computed once per tile and stage.
#include "cde_split.h"

namespace cde {

// ============================================================================================ forward
// K steps in a wave-local order: local step j <-> global step (2w + j) & 7, so that local steps 0 and 1 take the wave's
// OWN units (8w + kq, 8w + 4 + kq: exactly what lane (n, kq) computes) as B operands.  Those 8 MFMAs do not depend on
// the other waves: step 0 of the NEXT stage is issued before the barrier (covers the LDS write latency and the
// barrier skew), step 1 right after it (covers the latency of reading the other waves' units back).
template <typename TT, int DEGREE, int ACT>
__global__ __launch_bounds__(256, 2) void rk4_forward_split(
    const float* __restrict__ coeffs, const float* __restrict__ knots, int64_t n_intervals,
    const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ z0,
    const TT* __restrict__ grid, int64_t n_grid, const TT* __restrict__ t_out, int64_t n_out,
    float* __restrict__ z_out, int64_t B, const int64_t* __restrict__ stage_index,
    const float* __restrict__ stage_frac, Dims dims) {
  __shared__ __attribute__((aligned(16))) float zbuf[2 * SPL_ZBUF];
  __shared__ __attribute__((aligned(16))) float dxb[2 * SPL_DX];
  const int Hr = dims.H, Cr = dims.C;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int64_t series = (int64_t)blockIdx.x * 16 + n;
  const bool valid = series < B;
  const int64_t sc = valid ? series : B - 1;

  float wy[4][8];                       // [tile][LOCAL K step]
  f32x4 by[4];
#pragma unroll
  for (int T = 0; T < 4; ++T) {
    const int hA = 8 * w + 4 * (T >> 1) + (n >> 2), cA = 4 * (T & 1) + (n & 3);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 4 * ((2 * w + j) & 7) + q;
      wy[T][j] = (hA < Hr && cA < Cr && k < Hr) ? W[(hA * Cr + cA) * Hr + k] : 0.f;
    }
    const int hD = 8 * w + 4 * (T >> 1) + q;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cD = 4 * (T & 1) + r;
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
  touch_stage_table(stage_index, stage_frac, 0, 4 * n_steps);

  float* zw = zbuf + n * SPL_ZROW + q * 8 + 2 * w;          // writer: global K steps 2w, 2w + 1 of quarter q
  const float* zr = zbuf + n * SPL_ZROW + q * 8;            // reader: quarter q
  const int o1 = (2 * w + 2) & 7, o2 = (2 * w + 4) & 7, o3 = (2 * w + 6) & 7;   // local steps (2,3), (4,5), (6,7)
  Feed<DEGREE> feed;
  feed.init(coeffs, knots, stage_index, stage_frac, n_intervals, sc, Cr, 2 * w + (q & 1));
  float* dxw = dxb + n * SPL_DXROW + 2 * w + (q & 1);
  const float* dxr = dxb + n * SPL_DXROW;
  const bool feeds = q < 2;
  int par = 0;
  f32x4 y00 = by[0], y01 = by[1], y10 = by[2], y11 = by[3];
  auto own_step = [&](int j, float v) {
    y00 = mfma16(wy[0][j], v, y00);
    y01 = mfma16(wy[1][j], v, y01);
    y10 = mfma16(wy[2][j], v, y10);
    y11 = mfma16(wy[3][j], v, y11);
  };
  {
    const float d0 = feed.begin(0, (int)(4 * n_steps - 1));
    *reinterpret_cast<float2*>(zw) = make_float2(ya, yb);
    if (feeds) dxw[0] = d0;
    own_step(0, ya);
  }
  __syncthreads();

  int64_t jout = 1;
  TT t_due = n_out > 1 ? t_out[1] : (TT)0;                    // the next output time: reloaded when an output is written only
  for (int64_t k = 0; k < n_steps; ++k) {
    const TT t0 = grid[k], t1 = grid[k + 1];
    const float dt = (float)(t1 - t0);
    float k1a = 0.f, k1b = 0.f, k2a = 0.f, k2b = 0.f, pqa = 0.f, pqb = 0.f, za = ya, zb = yb;
#pragma unroll
    for (int stage = 0; stage < 4; ++stage) {
      const float2 z23 = *reinterpret_cast<const float2*>(zr + par * SPL_ZBUF + o1);
      const float2 z45 = *reinterpret_cast<const float2*>(zr + par * SPL_ZBUF + o2);
      const float2 z67 = *reinterpret_cast<const float2*>(zr + par * SPL_ZBUF + o3);
      const float4 d03 = *reinterpret_cast<const float4*>(dxr + par * SPL_DX);
      const float4 d47 = *reinterpret_cast<const float4*>(dxr + par * SPL_DX + 4);
      __builtin_amdgcn_sched_barrier(0);
      own_step(1, zb);
      __builtin_amdgcn_sched_barrier(0);
      // next stage's control derivative
      const float dnext = feed.advance((int)(4 * k) + stage);
      if (feeds) dxw[(par ^ 1) * SPL_DX] = dnext;
      __builtin_amdgcn_sched_barrier(0);
      const float zs[6] = {z23.x, z23.y, z45.x, z45.y, z67.x, z67.y};
#pragma unroll
      for (int j = 2; j < 8; ++j) own_step(j, zs[j - 2]);
      // contraction with dX: pairs = two neighbouring channels (consecutive registers), then the two halves
      f32x2 fpa = {0.f, 0.f}, fpb = {0.f, 0.f};
      fpa = __builtin_elementwise_fma(activate2<ACT>(y00[0], y00[1]), f32x2{d03.x, d03.y}, fpa);
      fpb = __builtin_elementwise_fma(activate2<ACT>(y10[0], y10[1]), f32x2{d03.x, d03.y}, fpb);
      fpa = __builtin_elementwise_fma(activate2<ACT>(y00[2], y00[3]), f32x2{d03.z, d03.w}, fpa);
      fpb = __builtin_elementwise_fma(activate2<ACT>(y10[2], y10[3]), f32x2{d03.z, d03.w}, fpb);
      fpa = __builtin_elementwise_fma(activate2<ACT>(y01[0], y01[1]), f32x2{d47.x, d47.y}, fpa);
      fpb = __builtin_elementwise_fma(activate2<ACT>(y11[0], y11[1]), f32x2{d47.x, d47.y}, fpb);
      fpa = __builtin_elementwise_fma(activate2<ACT>(y01[2], y01[3]), f32x2{d47.z, d47.w}, fpa);
      fpb = __builtin_elementwise_fma(activate2<ACT>(y11[2], y11[3]), f32x2{d47.z, d47.w}, fpb);
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
      *reinterpret_cast<float2*>(zw + (par ^ 1) * SPL_ZBUF) = make_float2(za, zb);
      __builtin_amdgcn_sched_barrier(0);
      y00 = by[0]; y01 = by[1]; y10 = by[2]; y11 = by[3];
      own_step(0, za);                                        // first K step of the next stage
      __syncthreads();
      par ^= 1;
    }
    const float y1a = za, y1b = zb;
    while (jout < n_out && t1 >= t_due) {
      const TT tj = t_due;
      if (tj == t0) store(jout, ya, yb);
      else if (tj == t1) store(jout, y1a, y1b);
      else {
        const float slope = (float)((tj - t0) / (t1 - t0));
        store(jout, ya + slope * (y1a - ya), yb + slope * (y1b - yb));
      }
      ++jout;
      if (jout < n_out) t_due = t_out[jout];
    }
    ya = y1a; yb = y1b;
  }
}

// ============================================================================================ adjoint
// Eight waves per tile, two per SIMD, with different ROLES (waves i and i + 4 of a workgroup share a SIMD):
//   chain wave w  (0..3): the sequential part -- Y tiles, f, g, y-/a-path RK updates, va partials: 64 MFMAs per stage
//   helper wave w (4..7): everything that is off the critical path -- dW_w += g(e-1)^T (wq z(e-1)) (32 MFMAs, one stage
//                         behind, g^T handed over through a double-buffered LDS tile), dL/db row sums, the control
//                         feed for stage e+1.
// A single wave cannot overlap its own VALU / LDS / barrier time with its own MFMAs (measured with a 4-wave version of
// this kernel that did all three products itself: the matrix pipe idled 35 % of the time at one tile per CU, 1.04 ms
// at B = 4096 against 0.87 ms here); the helper's MFMA chain runs in exactly those gaps.
constexpr int SPL8_LDS_FLOATS = 2 * SPL_ZBUF + 2 * SPL_ZT + 2 * SPL_VA + 2 * SPL_DX + 2 * 4 * SPL_GT;

// JF (affine field only): the chain waves evaluate f and a^T df/dz through the shared Jacobian J = sum_c dX_c W_c (see K3j,
// rk4_mfma.hip) instead of the two products Y = W z and va = W_w^T g.  Wave w forms the 8 rows of J it owns,
//     tile (hi, kh), MFMA row i = 4 q' + r  <->  J[h = 8w + hi][k = 4 (4 kh + r) + q'],      K = the 8 channels (2 steps)
// = 32 MFMAs (as many as Y alone), and lane (n, q) then holds J[h][k] for the very units k = 4 s + q whose z it already has
// in registers as the Y product's B operand (s = 4 kh + r).  f_h = J[h][.] . z is completed over the four lane quarters by
// two half-/row-swap rounds; a's slope sum_h a_h J[h][k] lands in the (tile kh, register r) pattern of the old va tiles, so
// the cross-wave exchange through vab is unchanged.  64 MFMAs per SIMD and stage instead of 96.

template <typename TT, int DEGREE, int ACT, bool JF = false>
__global__ __launch_bounds__(512, 1) void rk4_adjoint_split8(
    const float* __restrict__ coeffs, const float* __restrict__ knots, int64_t n_intervals,
    const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ z_saved,
    const float* __restrict__ grad_out, const TT* __restrict__ sgrid, const int64_t* __restrict__ seg_off,
    int64_t n_out, float* __restrict__ grad_z0, float* __restrict__ partial, int64_t B,
    const int64_t* __restrict__ stage_index, const float* __restrict__ stage_frac, Dims dims) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int Hr = dims.H, Cr = dims.C;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int w = wave & 3;
  const bool helper = __builtin_amdgcn_readfirstlane(wave) >= 4;
  const int n = lane & 15, q = lane >> 4;
  float* zbuf = lds;
  float* ztb = lds + 2 * SPL_ZBUF;
  float* vab = ztb + 2 * SPL_ZT;
  float* dxb = vab + 2 * SPL_VA;
  float* gT = dxb + 2 * SPL_DX + w * SPL_GT;                       // + (stage parity) * 4 * SPL_GT
  const int64_t series = (int64_t)blockIdx.x * 16 + n;
  const bool valid = series < B;
  const int64_t sc = valid ? series : B - 1;
  const float third = (float)(1.0 / 3.0);
  int par = 0, gpar = 0;                                           // parity of the state buffers / of the g^T tile
  if (n_out >= 2) touch_stage_table(stage_index, stage_frac, 0, 4 * (seg_off[n_out - 1] - 1));

  if (helper) {
    // ------------------------------------------------------------------------------------------ helper wave
    f32x4 accW[4][2];
    float gb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int Tm = 0; Tm < 4; ++Tm) {
      accW[Tm][0] = f32x4{0.f, 0.f, 0.f, 0.f};
      accW[Tm][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float* ztr = ztb + n * SPL_TROW + 4 * q;                 // N tile 1: + 16 rows
    const float* gr = gT + n * SPL_TROW + 4 * q;                   // + Tm*16 rows
    Feed<DEGREE> feed;
    feed.init(coeffs, knots, stage_index, stage_frac, n_intervals, sc, Cr, 2 * w + (q & 1));
    float* dxw = dxb + n * SPL_DXROW + 2 * w + (q & 1);
    const bool feeds = q < 2;
    f32x2 zq[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};   // wq z^T of the previous stage
    float wqh = 0.f;
    bool have = false;                                             // a g^T tile of a previous stage is waiting
    auto dw_round = [&](int gp) {
      float4 ga4[4];
#pragma unroll
      for (int Tm = 0; Tm < 4; ++Tm) ga4[Tm] = *reinterpret_cast<const float4*>(gr + gp * 4 * SPL_GT + Tm * 16 * SPL_TROW);
      const float b0[4] = {zq[0][0], zq[0][1], zq[1][0], zq[1][1]}, b1[4] = {zq[2][0], zq[2][1], zq[3][0], zq[3][1]};
#pragma unroll
      for (int Tm = 0; Tm < 4; ++Tm) {
        const float ga[4] = {ga4[Tm].x, ga4[Tm].y, ga4[Tm].z, ga4[Tm].w};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          accW[Tm][0] = mfma16(ga[s], b0[s], accW[Tm][0]);
          accW[Tm][1] = mfma16(ga[s], b1[s], accW[Tm][1]);
        }
        gb[Tm] = __builtin_fmaf((ga[0] + ga[1]) + (ga[2] + ga[3]), wqh, gb[Tm]);     // dL/db: row sums
      }
    };
    for (int64_t p = 0; p + 1 < n_out; ++p) {
      const int64_t k_begin = seg_off[p], k_end = seg_off[p + 1] - 1;
      if (k_end > k_begin) {
        {
          const float d0 = feed.begin((int)(4 * k_begin), (int)(4 * k_end - 1));
          if (feeds) dxw[par * SPL_DX] = d0;
        }
        spl_barrier();
        for (int64_t k = k_begin; k < k_end; ++k) {
          const float ds = (float)(sgrid[k + 1] - sgrid[k]);
#pragma unroll
          for (int stage = 0; stage < 4; ++stage) {
            const float4 zt0 = *reinterpret_cast<const float4*>(ztr + par * SPL_ZT);
            const float4 zt1 = *reinterpret_cast<const float4*>(ztr + par * SPL_ZT + 16 * SPL_TROW);
            if (have) dw_round(gpar ^ 1);                           // the tile the chain wave wrote one stage ago
            const float dnext = feed.advance((int)(4 * k) + stage);
            if (feeds) dxw[(par ^ 1) * SPL_DX] = dnext;
            const float wq = ((stage == 0 || stage == 3) ? 0.125f : 0.375f) * ds;   // 3/8-rule quadrature weight
            zq[0] = f32x2{zt0.x, zt0.y} * wq; zq[1] = f32x2{zt0.z, zt0.w} * wq;
            zq[2] = f32x2{zt1.x, zt1.y} * wq; zq[3] = f32x2{zt1.z, zt1.w} * wq;
            wqh = wq;
            have = true;
            spl_barrier();
            par ^= 1; gpar ^= 1;
          }
        }
      }
    }
    if (have) dw_round(gpar ^ 1);                                   // the last stage's tile
    // per-workgroup partial parameter gradients in the K3 layout (summed in tile order by reduce_mfma_partials):
    // accW[Tm][Tn] register r of lane (j = n, q) = dW[h = 8w + 4(Tm>>1) + q][c = 4(Tm&1) + r][k = 16 Tn + j]
    float* my_partial = partial + (int64_t)blockIdx.x * SPL_PARTIAL_FLOATS;
#pragma unroll
    for (int Tm = 0; Tm < 4; ++Tm) {
      const int h = 8 * w + 4 * (Tm >> 1) + q;
#pragma unroll
      for (int Tn = 0; Tn < 2; ++Tn) {
#pragma unroll
        for (int r = 0; r < 4; ++r) my_partial[(h * MC + 4 * (Tm & 1) + r) * MH + 16 * Tn + n] = accW[Tm][Tn][r];
      }
      float sgb = gb[Tm];
      sgb += __shfl_xor(sgb, 16, 64);
      sgb += __shfl_xor(sgb, 32, 64);
      if (q == 0) my_partial[MH * MC * MH + (8 * w + 4 * (Tm >> 1) + (n >> 2)) * MC + 4 * (Tm & 1) + (n & 3)] = sgb;
    }
    return;
  }

  // -------------------------------------------------------------------------------------------- chain wave
  static_assert(!JF || ACT == CDE_ACT_NONE, "the Jacobian form needs a field that is affine in z");
  float wy[4][8], wv[2][16];
  f32x4 by[4];
  float wj[JF ? 16 : 1][2];                                        // JF: A image of J, [tile = 2 hi + kh][K step]
  f32x2 bja[4], bjb[4];                                            //     bias rows of the lane's two units, channel pairs
  const int ua = 8 * w + q, ub = ua + 4;
  if constexpr (JF) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int h = 8 * w + (t >> 1), k = 4 * (4 * (t & 1) + (n & 3)) + (n >> 2);
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const int c = 4 * st + q;
        wj[t][st] = (h < Hr && c < Cr && k < Hr) ? W[(h * Cr + c) * Hr + k] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      auto bv = [&](int u, int c) { return (u < Hr && c < Cr) ? bias[u * Cr + c] : 0.f; };
      bja[j] = f32x2{bv(ua, 2 * j), bv(ua, 2 * j + 1)};
      bjb[j] = f32x2{bv(ub, 2 * j), bv(ub, 2 * j + 1)};
    }
  } else {
    spl_load_wy(W, bias, w, n, q, dims, wy, by);
    spl_load_wv(W, w, n, q, dims, wv);
  }
  const int pos = spl_pos(n);
  auto saved = [&](int64_t j, int u) { return u < Hr ? z_saved[(sc * n_out + j) * Hr + u] : 0.f; };
  auto gout = [&](int64_t j, int u) { return (valid && u < Hr) ? grad_out[(sc * n_out + j) * Hr + u] : 0.f; };
  float y0a = saved(n_out - 1, ua), y0b = saved(n_out - 1, ub);
  float a0a = gout(n_out - 1, ua), a0b = gout(n_out - 1, ub);      // a == 0 stays 0: padded lanes add nothing to dL/dW
  float* zw = zbuf + n * SPL_ZROW + q * 8 + 2 * w;
  const float* zr = zbuf + n * SPL_ZROW + q * 8;
  float* ztw = ztb + ua * SPL_TROW + pos;                          // second unit: + 4 rows
  float* vw = vab + (q * 16 + n) * SPL_VROW + 2 * w;               // + w_dst * 64 * SPL_VROW
  const float* vr = vab + ((w * 4 + q) * 16 + n) * SPL_VROW;
  float* gw_ = gT + (4 * q) * SPL_TROW + pos;                      // + (T*16 + r) rows
  const float* dxr = dxb + n * SPL_DXROW;
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

  for (int64_t p = 0; p + 1 < n_out; ++p) {
    const int64_t i_out = n_out - 1 - p;
    const int64_t k_begin = seg_off[p], k_end = seg_off[p + 1] - 1;   // steps k_begin .. k_end-1
    if (k_end > k_begin) {
      publish(par, y0a, y0b);
      spl_barrier();
      float ds_prev = 0.f;
      float ka1a = 0.f, ka1b = 0.f, ka2a = 0.f, ka2b = 0.f, asa = a0a, asb = a0b;
      for (int64_t k = k_begin; k < k_end; ++k) {
        const float ds = (float)(sgrid[k + 1] - sgrid[k]);
        float ky1a = 0.f, ky1b = 0.f, ky2a = 0.f, ky2b = 0.f;
        float ysa = y0a, ysb = y0b;
#pragma unroll
        for (int stage = 0; stage < 4; ++stage) {
          const float4 z03 = *reinterpret_cast<const float4*>(zr + par * SPL_ZBUF);
          const float4 z47 = *reinterpret_cast<const float4*>(zr + par * SPL_ZBUF + 4);
          const float4 d03 = *reinterpret_cast<const float4*>(dxr + par * SPL_DX);
          const float4 d47 = *reinterpret_cast<const float4*>(dxr + par * SPL_DX + 4);
          float kaa = 0.f, kab = 0.f;
          const bool first = stage == 0 && k == k_begin;               // no previous stage in this sweep
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
          const float zs[8] = {z03.x, z03.y, z03.z, z03.w, z47.x, z47.y, z47.z, z47.w};
          const float dX[MC] = {d03.x, d03.y, d03.z, d03.w, d47.x, d47.y, d47.z, d47.w};
          f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;                // a's slope, partial over this wave's 8 units
          float kya, kyb;
          float* gwp = gw_ + gpar * 4 * SPL_GT;
          if constexpr (JF) {
            // ---- g = a (x) dX for the helper waves' dL/dW product (the lane's two units, all channels)
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
            // ---- a of the wave's 8 units in every lane quarter: a8[hi] = a_(8w + hi) of this lane's series
            float a8[8];
            {
              float ea = asa, oa = asa, eb = asb, ob = asb;
              swap16s(ea, oa);                                     // ea: unit q & ~1, oa: unit q | 1 (of this half)
              swap16s(eb, ob);
              float e0 = ea, e2 = ea, o1 = oa, o3 = oa, e4 = eb, e6 = eb, o5 = ob, o7 = ob;
              swap32s(e0, e2); swap32s(o1, o3); swap32s(e4, e6); swap32s(o5, o7);
              a8[0] = e0; a8[1] = o1; a8[2] = e2; a8[3] = o3; a8[4] = e4; a8[5] = o5; a8[6] = e6; a8[7] = o7;
            }
            // ---- J one unit at a time: two tiles (k halves) of 2 K steps each; f_h partial and a's slope from the result
            const float bq0 = q == 0 ? dX[0] : q == 1 ? dX[1] : q == 2 ? dX[2] : dX[3];
            const float bq1 = q == 0 ? dX[4] : q == 1 ? dX[5] : q == 2 ? dX[6] : dX[7];
            const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
            const f32x2 z01 = {zs[0], zs[1]}, z23 = {zs[2], zs[3]}, z45 = {zs[4], zs[5]}, z67 = {zs[6], zs[7]};
            f32x2 va01 = {0.f, 0.f}, va23 = va01, vb01 = va01, vb23 = va01;
            float p[8];
#pragma unroll
            for (int hi = 0; hi < 8; ++hi) {
              f32x4 j0 = mfma16(wj[2 * hi][0], bq0, zero);
              f32x4 j1 = mfma16(wj[2 * hi + 1][0], bq0, zero);
              j0 = mfma16(wj[2 * hi][1], bq1, j0);
              j1 = mfma16(wj[2 * hi + 1][1], bq1, j1);
              const f32x2 ah = {a8[hi], a8[hi]};
              f32x2 pp = f32x2{j0[0], j0[1]} * z01;
              pp = __builtin_elementwise_fma(f32x2{j0[2], j0[3]}, z23, pp);
              pp = __builtin_elementwise_fma(f32x2{j1[0], j1[1]}, z45, pp);
              pp = __builtin_elementwise_fma(f32x2{j1[2], j1[3]}, z67, pp);
              asm("" : "+v"(pp));                                    // keeps the chain packed (LLVM scalarises a packed chain
              p[hi] = pp[0] + pp[1];                                 //  whose lanes are consumed separately: 8 instead of 4 per row)
              va01 = __builtin_elementwise_fma(f32x2{j0[0], j0[1]}, ah, va01);
              va23 = __builtin_elementwise_fma(f32x2{j0[2], j0[3]}, ah, va23);
              vb01 = __builtin_elementwise_fma(f32x2{j1[0], j1[1]}, ah, vb01);
              vb23 = __builtin_elementwise_fma(f32x2{j1[2], j1[3]}, ah, vb23);
            }
            v0 = f32x4{va01[0], va01[1], va23[0], va23[1]};
            v1 = f32x4{vb01[0], vb01[1], vb23[0], vb23[1]};
            // ---- f of the lane's two units: sum the four quarters' shares (units 0,1,4,5 meet in the lower half-wave,
            // 2,3,6,7 in the upper one; then even / odd units in the even / odd 16-lane rows)
            swap32s(p[0], p[2]); swap32s(p[1], p[3]); swap32s(p[4], p[6]); swap32s(p[5], p[7]);
            float s0 = p[0] + p[2], s1 = p[1] + p[3], s4 = p[4] + p[6], s5 = p[5] + p[7];
            swap16s(s0, s1); swap16s(s4, s5);
            f32x2 fb2 = bja[0] * f32x2{dX[0], dX[1]}, fb3 = bjb[0] * f32x2{dX[0], dX[1]};
#pragma unroll
            for (int j = 1; j < 4; ++j) {
              fb2 = __builtin_elementwise_fma(bja[j], f32x2{dX[2 * j], dX[2 * j + 1]}, fb2);
              fb3 = __builtin_elementwise_fma(bjb[j], f32x2{dX[2 * j], dX[2 * j + 1]}, fb3);
            }
            kya = -((s0 + s1) + (fb2[0] + fb2[1])); kyb = -((s4 + s5) + (fb3[0] + fb3[1]));      // reverse time: dy/ds = -f
          } else {
          // ---- Y tiles of this wave's 8 hidden units
          f32x4 yt[4] = {by[0], by[1], by[2], by[3]};
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            yt[0] = mfma16(wy[0][s], zs[s], yt[0]);
            yt[1] = mfma16(wy[1][s], zs[s], yt[1]);
            yt[2] = mfma16(wy[2][s], zs[s], yt[2]);
            yt[3] = mfma16(wy[3][s], zs[s], yt[3]);
          }
          // ---- activation, f, g = dL/dY (pairs = neighbouring channels of one unit: packed, no shuffles)
          f32x2 gq[4][2];
          f32x2 fpa = {0.f, 0.f}, fpb = {0.f, 0.f};
#pragma unroll
          for (int T = 0; T < 4; ++T) {
            const float aown = (T >> 1) ? asb : asa;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const f32x2 dx = {dX[4 * (T & 1) + 2 * j], dX[4 * (T & 1) + 2 * j + 1]};
              const f32x2 t = activate2<ACT>(yt[T][2 * j], yt[T][2 * j + 1]);
              if (T >> 1) fpb = __builtin_elementwise_fma(t, dx, fpb); else fpa = __builtin_elementwise_fma(t, dx, fpa);
              if (ACT == CDE_ACT_NONE) gq[T][j] = dx * aown;
              else gq[T][j] = (f32x2{spl_slope<ACT>(t[0]), spl_slope<ACT>(t[1])} * dx) * aown;
              gwp[(T * 16 + 2 * j) * SPL_TROW] = gq[T][j][0];
              gwp[(T * 16 + 2 * j + 1) * SPL_TROW] = gq[T][j][1];
            }
          }
            kya = -(fpa[0] + fpa[1]); kyb = -(fpb[0] + fpb[1]);          // reverse time: dy/ds = -f
          // ---- va partial over this wave's 64 (h, c) rows, all 32 output units (K step 8P + c)
#pragma unroll
          for (int sp = 0; sp < 16; ++sp) {
            const float gv = gq[2 * (sp >> 3) + ((sp >> 2) & 1)][(sp >> 1) & 1][sp & 1];     // g[P = sp >> 3][c = sp & 7]
            v0 = mfma16(wv[0][sp], gv, v0);
            v1 = mfma16(wv[1][sp], gv, v1);
          }
          }
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
          publish(par ^ 1, nya, nyb);
          ysa = nya; ysb = nyb;
          // register r of tile T -> destination wave 2T + (r >> 1), its unit j = r & 1
          float* vwp = vw + (par ^ 1) * SPL_VA;
          *reinterpret_cast<float2*>(vwp) = make_float2(v0[0], v0[1]);
          *reinterpret_cast<float2*>(vwp + 64 * SPL_VROW) = make_float2(v0[2], v0[3]);
          *reinterpret_cast<float2*>(vwp + 2 * 64 * SPL_VROW) = make_float2(v1[0], v1[1]);
          *reinterpret_cast<float2*>(vwp + 3 * 64 * SPL_VROW) = make_float2(v1[2], v1[3]);
          spl_barrier();
          par ^= 1; gpar ^= 1;
        }
        y0a = ysa; y0b = ysb;
        ds_prev = ds;
      }
      {                                                                 // the last stage's a-path update
        float kaa, kab;
        read_ka(par, kaa, kab);
        a0a = a0a + (ka1a + kaa) * ds_prev * 0.125f; a0b = a0b + (ka1b + kab) * ds_prev * 0.125f;
      }
    }
    // torchdiffeq adjoint: re-seed y from the stored forward value, add the incoming gradient
    y0a = saved(i_out - 1, ua); y0b = saved(i_out - 1, ub);
    a0a += gout(i_out - 1, ua); a0b += gout(i_out - 1, ub);
  }
  if (valid) {
    if (ua < Hr) grad_z0[series * Hr + ua] = a0a;
    if (ub < Hr) grad_z0[series * Hr + ub] = a0b;
  }
}

// defined in rk4_mfma.hip: fixed-order sum of the per-tile partials
int launch_reduce_partials(const float* partial, int64_t n_tiles, void* grad_W, void* grad_b, int H, int C, hipStream_t s);

// ------------------------------------------------------------------------------------------ host side
size_t split_adjoint_partial_bytes(int64_t B) { return (size_t)((B + 15) / 16) * SPL_PARTIAL_FLOATS * sizeof(float); }

template <typename TT>
int launch_forward_split(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                         const void* bias, int act, const void* z0, const void* grid, int64_t n_grid, const void* t_out,
                         int64_t n_out, void* z_out, int64_t B, int64_t C, int64_t H, const int64_t* stage_index,
                         const void* stage_frac, hipStream_t s) {
  const Dims dims{(int)H, (int)C};
  const unsigned blocks = (unsigned)((B + 15) / 16);
#define CDE_FWD(D, A)                                                                                                \
  rk4_forward_split<TT, D, A><<<blocks, 256, 0, s>>>(                                                                \
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
int launch_adjoint_split(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                         const void* bias, int act, const void* z_saved, const void* grad_out, const void* sgrid,
                         const int64_t* seg_off, int64_t n_out, void* grad_z0, void* grad_W, void* grad_b, int64_t B,
                         int64_t C, int64_t H, const int64_t* stage_index, const void* stage_frac, float* partial,
                         hipStream_t s) {
  const Dims dims{(int)H, (int)C};
  const unsigned blocks = (unsigned)((B + 15) / 16);
  const size_t lds = (size_t)SPL8_LDS_FLOATS * sizeof(float);
  if (degree != CDE_PATH_CUBIC && degree != CDE_PATH_LINEAR) return CDE_ERR_UNSUPPORTED;
#define CDE_ADJ(D, A, J)                                                                                             \
  do {                                                                                                               \
    (void)hipFuncSetAttribute((const void*)rk4_adjoint_split8<TT, D, A, J>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                              (int)lds);                                                                             \
    rk4_adjoint_split8<TT, D, A, J><<<blocks, 512, lds, s>>>(                                                        \
        (const float*)coeffs, (const float*)knots, n_intervals, (const float*)W, (const float*)bias,                 \
        (const float*)z_saved, (const float*)grad_out, (const TT*)sgrid, seg_off, n_out, (float*)grad_z0, partial,   \
        B, stage_index, (const float*)stage_frac, dims);                                                             \
  } while (0)
  const bool jacobian = option(CDE_OPT_K3_FORM) != 1;                // 1: the two-GEMM chain waves (tests, comparisons)
  if (act == CDE_ACT_NONE && jacobian) {
    if (degree == CDE_PATH_CUBIC) CDE_ADJ(CDE_PATH_CUBIC, CDE_ACT_NONE, true); else CDE_ADJ(CDE_PATH_LINEAR, CDE_ACT_NONE, true);
  } else if (act == CDE_ACT_NONE) {
    if (degree == CDE_PATH_CUBIC) CDE_ADJ(CDE_PATH_CUBIC, CDE_ACT_NONE, false); else CDE_ADJ(CDE_PATH_LINEAR, CDE_ACT_NONE, false);
  } else if (act == CDE_ACT_TANH) {
    if (degree == CDE_PATH_CUBIC) CDE_ADJ(CDE_PATH_CUBIC, CDE_ACT_TANH, false); else CDE_ADJ(CDE_PATH_LINEAR, CDE_ACT_TANH, false);
  } else return CDE_ERR_UNSUPPORTED;
#undef CDE_ADJ
  int rc = check_launch();
  if (rc != CDE_OK) return rc;
  return launch_reduce_partials(partial, (B + 15) / 16, grad_W, grad_b, (int)H, (int)C, s);
}

template int launch_forward_split<float>(const void*, const void*, int64_t, int, const void*, const void*, int,
                                         const void*, const void*, int64_t, const void*, int64_t, void*, int64_t, int64_t,
                                         int64_t, const int64_t*, const void*, hipStream_t);
template int launch_forward_split<double>(const void*, const void*, int64_t, int, const void*, const void*, int,
                                          const void*, const void*, int64_t, const void*, int64_t, void*, int64_t,
                                          int64_t, int64_t, const int64_t*, const void*, hipStream_t);
template int launch_adjoint_split<float>(const void*, const void*, int64_t, int, const void*, const void*, int,
                                         const void*, const void*, const void*, const int64_t*, int64_t, void*, void*,
                                         void*, int64_t, int64_t, int64_t, const int64_t*, const void*, float*,
                                         hipStream_t);
template int launch_adjoint_split<double>(const void*, const void*, int64_t, int, const void*, const void*, int,
                                          const void*, const void*, const void*, const int64_t*, int64_t, void*, void*,
                                          void*, int64_t, int64_t, int64_t, const int64_t*, const void*, float*,
                                          hipStream_t);

}  // namespace cde
