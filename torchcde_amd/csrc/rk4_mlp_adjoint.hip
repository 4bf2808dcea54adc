// rk4_mlp_adjoint.hip -- K3m: continuous-adjoint reverse sweep for the two-layer vector field
//   f(z) = reshape_{HxC}( act( W2 relu(W1 z + b1) + b2 ) ) dX          (example/time_series_classification.py:20-51)
//
// K3 keeps dL/dW (256 x 32 per wave) in accumulator registers for the whole sweep.  Here dL/dW2 is 256 x 128 per
// wave -- four register files' worth -- so the sweep kernel only integrates the augmented state (z, a) backwards
// and streams the per-stage factors of the parameter gradients to HBM,
//     U  [row][132]  relu(W1 z + b1) (128 padded) | 1 | 0 0 0          row = (stage, series)
//     G2 [row][256]  w ds dL/dY2      (h*8 + c, padded layout)
//     G1 [row][128]  w ds dL/dY1
//     Z  [row][36]   z (32 padded) | 1 | 0 0 0
// and the host turns them into   dW2 | db2 = G2^T U,   dW1 | db1 = G1^T Z   with library GEMMs (the "1" columns
// yield the bias gradients).  288 GB of HBM make this cheap: 2.2 KB per series and stage, swept in chunks of steps.
//
// Per stage and wave (16 series, lane (n, q) owns hidden units q, 4+q, .., 28+q of z and a; 16x16x4 MFMAs):
//   layer 1       64 MFMAs   pre1 = W1 z + b1  -> u = relu(pre1), mask            (A from LDS)
//   layer 2      512 MFMAs   Y2 = W2 u + b2 per tile pair P (4 hidden units x 8 channels) (A from LDS)
//                in-lane     t = act(Y2), f_h = sum_c t dX_c, g2 = a_h dX_c act'(Y2)
//   gu          512 MFMAs   gu += W2[(h,c), :]^T g2   K step (P, c): the lane feeds its OWN g2 register
//   va           64 MFMAs   va = W1^T (gu * mask)                                  (A from L2, 16 loads)
// W2 is needed as the A operand of two products (rows = (h,c) for Y2, rows = hidden-layer units for gu); two MFMA
// images (2 x 128 KB) do not fit the LDS and streaming one from L2 left the matrix pipe waiting on load latency
// (measured: 34 ms per sweep vs 17 ms of MFMA time).  Instead ONE plain copy of W2 lives in LDS, row stride 132
// floats, rows permuted so that both access patterns are bank-conflict free:
//     physical row pr(h, c) = ((h>>2)*4 + (c>>2)*2 + ((h>>1)&1))*8 + (2*bitrev2(h&3) + (c&3)) % 8
//   Y2:  lane (i, kq) reads 4 consecutive floats (4 K steps) of row (h = 4P + (i>>2), c = 4tb + (i&3)) at column
//        16*T1 + 4*kq  -> ds_read_b128, every 8-lane group covers all 8 bank slots
//   gu:  lane (i, kq) reads row (h = 4P + kq, c) at column 16*T1 + i -> ds_read_b32, each half-wave on 32 distinct banks
// with every (P, tb, T1) offset an instruction immediate.
// 1152 MFMAs per stage = 76.3 MFLOP per series per solve for the sweep itself.
#include "cde_mlp_adj.h"

namespace cde {

__device__ __forceinline__ float mlp_adj_image(const float* __restrict__ W1, const float* __restrict__ b1,
                                               const float* __restrict__ W2, const float* __restrict__ b2, int e,
                                               MlpDims d, int nb) {
  if (e < W1M_FLOATS + B1M_FLOATS) return mlp16_image(W1, b1, W2, b2, e, d, nb);    // same layer-1 image as K2m
  e -= W1M_FLOATS + B1M_FLOATS;
  if (e < W2P_FLOATS) {
    const int pr = e / W2P_STRIDE, col = e - pr * W2P_STRIDE;
    const int r8 = pr & 7, hi = pr >> 3, hb = hi & 1, T = hi >> 1, tb = T % nb, P = T / nb;   // tile T = nb*P + tb
    // invert w2p_residue: r8 = (2*(h&3) + g(c&3)) % 8 with (h&3)>>1 == hb
    int h3 = 0, c3 = 0;
    for (int hh = 2 * hb; hh < 2 * hb + 2; ++hh)
      for (int cc = 0; cc < 4; ++cc)
        if (w2p_residue(hh, cc) == r8) { h3 = hh; c3 = cc; }
    const int h = 4 * P + h3, c = 4 * tb + c3;
    return (col < d.width && h < d.H && c < d.C) ? W2[(h * d.C + c) * d.width + col] : 0.f;
  }
  e -= W2P_FLOATS;
  if (e < BY_FLOATS) return by16_image(b2, e >> 4, (e >> 2) & 3, e & 3, Dims{d.H, d.C}, nb);
  e -= BY_FLOATS;
  // va tile T (row i <-> z unit 4*(4T + (i&3)) + (i>>2), so register r of lane (n, q) is unit 4*(4T+r) + q),
  // K step (T1, r = j): lane quarter kq feeds hidden-layer unit 16*T1 + 4*kq + j
  const int j = e & 3, l = (e >> 2) & 63, i = l & 15, kq = l >> 4;
  const int g = (e >> 8), T = g >> 3, T1 = g & 7;
  const int unit = 16 * T1 + 4 * kq + j, k = 4 * (4 * T + (i & 3)) + (i >> 2);
  return (unit < d.width && k < d.H) ? W1[unit * d.H + k] : 0.f;
}

__global__ void mlp_adj_image_kernel(const float* __restrict__ W1, const float* __restrict__ b1,
                                     const float* __restrict__ W2, const float* __restrict__ b2,
                                     float* __restrict__ img, MlpDims d, int nb) {
  int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < MLP_ADJ_IMAGE_FLOATS) { img[e] = mlp_adj_image(W1, b1, W2, b2, e, d, nb); return; }
  if (e >= MLP_ADJ_IMAGE_HI_FLOATS) return;
  // the zero-padded copy of the upper rows (hidden units 16..31 of the 16-channel layout; cde_mlp_adj.h: mlp_adj_hi)
  const int at = e;
  e -= MLP_ADJ_IMAGE_FLOATS;
  if (e < MLP_ADJ_HI_BIAS_FLOATS) {
    const int h = 16 + (e >> 4), c = e & 15;
    img[at] = (h < d.H && c < d.C) ? b2[h * d.C + c] : 0.f;
    return;
  }
  e -= MLP_ADJ_HI_BIAS_FLOATS;
  const bool by_lane = e >= MLP_ADJ_HI_W2_FLOATS;                  // the second copy: [row][8 n + T1] = W2[row][16 T1 + n]
  if (by_lane) e -= MLP_ADJ_HI_W2_FLOATS;
  const int row = e >> 7, at_col = e & 127, h = 16 + (row >> 4), c = row & 15;
  const int col = by_lane ? 16 * (at_col & 7) + (at_col >> 3) : at_col;
  img[at] = (h < d.H && c < d.C && col < d.width) ? W2[(h * d.C + c) * d.width + col] : 0.f;
}

// DCOEFF: also accumulate dL/d(control coefficients) into `grad_coeffs` (zeroed by the caller, layout of `coeffs`),
// as K3a does: d(a.f)/d(dX_c) = sum_h a_h act(Y2)_hc, here summed in-lane over the lane's hidden units and then
// over the four lane quarters with two shuffles; quarter q carries channels (CT/4) q .. (CT/4) q + CT/4 - 1 to the
// coefficient row (the 16-channel layout, round 5: the one-wave-per-tile form only).
// SPLIT (at most one tile per CU): a 256-thread workgroup whose four waves carry ONE tile through the sweep -- layer 1,
// dL/dY1, va and the RK bookkeeping redundantly and bit-identically, the unit groups of layer 2 / dL/dY2 / gu split four
// ways, partial sums added in a fixed wave order through a 9 KB LDS window (cde_mlp_adj.h: mlp_split_allreduce, as in
// K4am's split form); wave 0 stores state, the shared factor rows and the control gradients.
// BACKPROP (round 5; adjoint=False under rk4, reference solver.py:144 + README.md:103): the same evaluation, but as
// reverse-mode differentiation of the 3/8-rule steps themselves (see rk4_backprop.hip for the recurrences): the steps
// k_end-1 .. k_begin of the FORWARD grid are walked backwards, stage 4 first; the stage states come from `stages`
// (B, n_steps_total, 4, 32: what rk4_forward_mfma<.., MLP, .., SAVE> stored), the vector the Jacobian-transpose is
// applied to is kb_i (in place of a), factor rows are streamed unweighted (kb_i carries dt/8, 3dt/8), nothing is
// re-integrated: `a_state` holds dL/dy of the grid node above the chunk on entry and of the node below it on return.
template <typename TT, int DEGREE, int ACT, bool DCOEFF = false, int CT = MC, bool SPLIT = false, bool BACKPROP = false>
__global__ __launch_bounds__(SPLIT ? 256 : 512, SPLIT ? 1 : 2) void rk4_adjoint_mlp_sweep(
    const float* __restrict__ coeffs, const float* __restrict__ knots, int64_t n_intervals,
    const float* __restrict__ img, float* __restrict__ y_state, float* __restrict__ a_state,
    const TT* __restrict__ sgrid, int64_t k_begin, int64_t k_end, const int64_t* __restrict__ stage_index,
    const float* __restrict__ stage_frac, float* __restrict__ U, float* __restrict__ G2, float* __restrict__ G1,
    float* __restrict__ Z, int64_t B, Dims dims, float* __restrict__ grad_coeffs = nullptr,
    const float* __restrict__ stages = nullptr, int64_t n_steps_total = 0) {
  static_assert(!BACKPROP || !SPLIT, "the reverse-mode form: one wave per tile");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  {
    const float4* src = reinterpret_cast<const float4*>(img);
    float4* dst = reinterpret_cast<float4*>(lds);
    for (int i = threadIdx.x; i < ADJ_LDS_FLOATS / 4; i += blockDim.x) dst[i] = src[i];
  }
  __syncthreads();
  constexpr int NB = CT / 4, NP = 16 / NB;      // channel blocks per unit group, unit groups (of 4 hidden units)
  constexpr int NJ = CT / 4;                    // control gradients: channels NJ q .. NJ q + NJ - 1 go to the coefficient row with quarter q
  // 32 units x 16 channels (round 6): unit groups 4..7 from the zero-padded copy of the upper rows behind the images
  constexpr int NPX = CT == 16 ? 8 : NP;
  const MlpHi hi = mlp_adj_hi(img, dims.H, CT);
  const bool has_hi = CT == 16 && hi.W2 != nullptr;
  const int per_wave = has_hi ? 2 : NP / 4;     // SPLIT: unit groups per wave
  // the upper groups' dL/dY2 rows go behind the lower groups' (the caller reduces the two halves separately)
  float* G2hi = G2 + (int64_t)4 * (k_end - k_begin) * B * G2_COLS;
  static_assert(!(DCOEFF && CT != MC && SPLIT), "control gradients of the 16-channel layout: one wave per tile");
  const int Hr = dims.H, Cr = dims.C;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int64_t tile = SPLIT ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * 8 + wave;
  if (tile * 16 >= B) return;
  const int pw = SPLIT ? wave : 0;
  float* xbuf = lds + ADJ_LDS_FLOATS;                                // (SPLIT only: 4 x 64 x 9 floats behind the images)
  const int64_t series = tile * 16 + n;
  const bool in_range = series < B;
  const bool valid = in_range && (!SPLIT || wave == 0);              // who stores what every wave of a split tile holds
  const int64_t sc = in_range ? series : B - 1;
  const float4* w1t_base = reinterpret_cast<const float4*>(img + ADJ_LDS_FLOATS) + lane;
  // lane offsets into the plain W2 copy (see the header): Y2 rows (h&3 = n>>2, c&3 = n&3), gu rows (h&3 = q, c&3 = cl)
  const int w2y_off = ((n >> 3) * 8 + w2p_residue(n >> 2, n & 3)) * W2P_STRIDE + 4 * q;
  int w2g_off[4];
#pragma unroll
  for (int cl = 0; cl < 4; ++cl) w2g_off[cl] = ((q >> 1) * 8 + w2p_residue(q, cl)) * W2P_STRIDE + n;

  const int ua = q, ub = 16 + q;                                     // this lane's units: q, 4+q, .., 28+q
  f32x4 ya = {0.f, 0.f, 0.f, 0.f}, yb = ya;
  if constexpr (!BACKPROP) { ya = load_units4<4>(y_state + sc * Hr, ua, Hr); yb = load_units4<4>(y_state + sc * Hr, ub, Hr); }
  f32x4 aa = load_units4<4>(a_state + sc * Hr, ua, Hr), ab = load_units4<4>(a_state + sc * Hr, ub, Hr);
  if (!in_range) { aa = f32x4{0.f, 0.f, 0.f, 0.f}; ab = aa; }       // a == 0 stays 0: padded lanes contribute nothing

  const int64_t e_first = BACKPROP ? 4 * k_end - 1 : 4 * k_begin;
  int64_t idx = stage_index[e_first];
  float frac = stage_frac[e_first];
  Row<DEGREE, CT> row = load_row<DEGREE, CT>(coeffs, sc, n_intervals, idx, Cr);

  // dL/d(coefficient row in use) for channels NJ q .. NJ q + NJ - 1: cubic (b, 2c, 3d), linear (left knot, right knot)
  float gc0[NJ], gc1[NJ], gc2[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) { gc0[j] = 0.f; gc1[j] = 0.f; gc2[j] = 0.f; }
  auto flush_control_grad = [&](int64_t at) {
    if constexpr (DCOEFF) {
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const int c = NJ * q + j;
        if (valid && c < Cr) {
          if (DEGREE == CDE_PATH_CUBIC) {
            float* g = grad_coeffs + (series * n_intervals + at) * 4 * Cr;
            g[Cr + c] += gc0[j]; g[2 * Cr + c] += gc1[j]; g[3 * Cr + c] += gc2[j];
          } else {
            float* g = grad_coeffs + (series * (n_intervals + 1) + at) * Cr;
            g[c] += gc0[j]; g[Cr + c] += gc1[j];
          }
        }
        gc0[j] = 0.f; gc1[j] = 0.f; gc2[j] = 0.f;
      }
    }
  };

  for (int64_t kk = 0; kk < k_end - k_begin; ++kk) {
    const int64_t k = BACKPROP ? k_end - 1 - kk : k_begin + kk;
    const float ds = (float)(sgrid[k + 1] - sgrid[k]);
    f32x4 ky1a, ky1b, ky2a, ky2b, ka1a, ka1b, ka2a, ka2b;
    f32x4 za = ya, zb = yb, sa = aa, sb = ab;                        // stage values of z and a
    // BACKPROP: aa / ab = dL/dy1 of this step; kb1 .. kb3 as in rk4_backprop.hip, sa / sb = the kb of the stage at hand
    f32x4 kb1a, kb1b, kb2a, kb2b, kb3a, kb3b, yba, ybb;
    if constexpr (BACKPROP) {
      const float c8 = ds * 0.125f;
      kb1a = aa * c8; kb1b = ab * c8; kb2a = aa * (3.f * c8); kb2b = ab * (3.f * c8); kb3a = kb2a; kb3b = kb2b;
      sa = kb1a; sb = kb1b;                                          // kb4 == kb1's initial value
      yba = aa; ybb = ab;
    }
#pragma unroll
    for (int si = 0; si < 4; ++si) {
      const int stage = BACKPROP ? 3 - si : si;
      float dX[CT];
      const float width = DEGREE == CDE_PATH_LINEAR ? knots[idx + 1] - knots[idx] : 1.f;
      control_slope<DEGREE, CT>(row, frac, width, dX);
      const int64_t e_next = BACKPROP ? 4 * k + stage - 1 : 4 * k + stage + 1;
      const bool more = BACKPROP ? e_next >= 4 * k_begin : e_next < 4 * k_end;
      const int64_t nidx = more ? stage_index[e_next] : idx;
      const float nfrac = more ? stage_frac[e_next] : frac;
      // `row` is dead from here to the end of the tile loop: the next stage's row is (re)loaded only then, so its 24
      // registers are free while the register pressure peaks (reloading costs 6 KB of L2 traffic per wave and stage)
      const float wq = BACKPROP ? 1.f : ((stage == 0 || stage == 3) ? 0.125f : 0.375f) * ds;     // 3/8-rule quadrature weight
      const int64_t out_row = (kk * 4 + si) * B + series;                        // (stage, series)
      if constexpr (BACKPROP) {                                      // the state the forward pass handed to this evaluation
        const float* srow = stages + ((sc * n_steps_total + k) * 4 + stage) * 32;
#pragma unroll
        for (int i = 0; i < 4; ++i) { za[i] = srow[q + 4 * i]; zb[i] = srow[16 + q + 4 * i]; }
      }

      int opaque = 0;                                                // keeps the LDS reads inside the stage
      asm volatile("" : "+v"(opaque));
      const float4* w1 = reinterpret_cast<const float4*>(lds) + lane + opaque;
      const float4* bb1 = reinterpret_cast<const float4*>(lds + W1M_FLOATS) + q + opaque;
      const float* w2p = lds + W1M_FLOATS + B1M_FLOATS + opaque;
      const float4* bb2 = reinterpret_cast<const float4*>(lds + W1M_FLOATS + B1M_FLOATS + W2P_FLOATS) + q + opaque;
      const float* w2y = w2p + w2y_off;
      const float* w2g[4] = {w2p + w2g_off[0], w2p + w2g_off[1], w2p + w2g_off[2], w2p + w2g_off[3]};
      const float4* w1t = w1t_base + opaque;                         // L2-resident image: same trick against LICM
      const float zs[8] = {za[0], za[1], za[2], za[3], zb[0], zb[1], zb[2], zb[3]};
      const float as[8] = {sa[0], sa[1], sa[2], sa[3], sb[0], sb[1], sb[2], sb[3]};

      // ---- layer 1: u = relu(W1 z + b1); `mask` bit s2 = (pre-activation of the lane's s2-th hidden unit > 0)
      float u[32];
      unsigned mask = 0;
#pragma unroll
      for (int TP = 0; TP < 4; ++TP) {
        const float4 c0 = bb1[8 * TP], c1 = bb1[8 * TP + 4];
        f32x4 y0 = {c0.x, c0.y, c0.z, c0.w}, y1 = {c1.x, c1.y, c1.z, c1.w};
        const float4 g00 = w1[(4 * TP) * 64], g01 = w1[(4 * TP + 1) * 64], g10 = w1[(4 * TP + 2) * 64], g11 = w1[(4 * TP + 3) * 64];
        const float a0[8] = {g00.x, g00.y, g00.z, g00.w, g01.x, g01.y, g01.z, g01.w};
        const float a1[8] = {g10.x, g10.y, g10.z, g10.w, g11.x, g11.y, g11.z, g11.w};
#pragma unroll
        for (int s = 0; s < 8; ++s) { y0 = mfma16(a0[s], zs[s], y0); y1 = mfma16(a1[s], zs[s], y1); }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          u[8 * TP + r] = fmaxf(y0[r], 0.f);
          u[8 * TP + 4 + r] = fmaxf(y1[r], 0.f);
          mask |= (y0[r] > 0.f ? 1u : 0u) << (8 * TP + r);
          mask |= (y1[r] > 0.f ? 1u : 0u) << (8 * TP + 4 + r);
        }
      }
      if (valid) {
        float* urow = U + out_row * U_COLS + 4 * q;                  // hidden-layer units 16*T1 + 4q + r
#pragma unroll
        for (int T1 = 0; T1 < 8; ++T1) stream_store4(urow + 16 * T1, u[4 * T1], u[4 * T1 + 1], u[4 * T1 + 2], u[4 * T1 + 3]);
        float* zrow = Z + out_row * Z_COLS;
#pragma unroll
        for (int m = 0; m < 8; ++m) if (4 * m + q < Hr) zrow[4 * m + q] = zs[m];
      }

      __builtin_amdgcn_sched_barrier(0);
      // ---- layer 2, activation, contraction, dL/dY2, and gu += W2^T dL/dY2, one tile pair at a time
      f32x4 gu[8];
#pragma unroll
      for (int T1 = 0; T1 < 8; ++T1) gu[T1] = f32x4{0.f, 0.f, 0.f, 0.f};
      f32x4 fa = {0.f, 0.f, 0.f, 0.f}, fb = fa;
      float gdx[CT];                                                 // d(a.f)/d(dX_c), this lane's hidden units (DCOEFF)
#pragma unroll
      for (int c = 0; c < CT; ++c) gdx[c] = 0.f;
      // (the unit-group body with the upper half as a compile-time flag; the four lower groups unrolled as ever, the upper ones --
      //  32 hidden units x 16 channels only -- a ROLLED loop behind one uniform branch: unrolled they doubled the stage body of
      //  every 16-channel sweep and cost the plain shape 1.9 %)
      auto group = [&](int P, auto upper_c) {                        // unit group P: 4 hidden units x CT channels = NB tiles
        constexpr bool upper = decltype(upper_c)::value;
        float as_P = as[0];
#pragma unroll
        for (int kk = 1; kk < 8; ++kk) as_P = P == kk ? as[kk] : as_P;
        f32x4 y[NB];
        const float* tp_[NB];
#pragma unroll
        for (int tb = 0; tb < NB; ++tb) {
          if constexpr (!upper) {
            const float4 c0 = bb2[4 * (NB * P + tb)];
            y[tb] = f32x4{c0.x, c0.y, c0.z, c0.w};
            tp_[tb] = w2y + 2 * (NB * P + tb) * 8 * W2P_STRIDE;      // tile T = NB*P + tb: physical rows (2T + hb)*8 + r8
          } else {
            // upper half: the lane's D rows are (h = 4P + q, c = 4 tb + r); its A rows (h = 4P + (n >> 2), c = 4 tb + (n & 3))
            const float4 c0 = *reinterpret_cast<const float4*>(hi.b2 + (4 * P + q - 16) * 16 + 4 * tb);
            y[tb] = f32x4{c0.x, c0.y, c0.z, c0.w};
            tp_[tb] = hi.W2 + ((4 * P + (n >> 2) - 16) * 16 + 4 * tb + (n & 3)) * 128 + 4 * q;
          }
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          float4 a[NB];
#pragma unroll
          for (int tb = 0; tb < NB; ++tb) a[tb] = *reinterpret_cast<const float4*>(tp_[tb] + 16 * g);
#pragma unroll
          for (int tb = 0; tb < NB; ++tb) y[tb] = mfma16(a[tb].x, u[4 * g], y[tb]);
#pragma unroll
          for (int tb = 0; tb < NB; ++tb) y[tb] = mfma16(a[tb].y, u[4 * g + 1], y[tb]);
#pragma unroll
          for (int tb = 0; tb < NB; ++tb) y[tb] = mfma16(a[tb].z, u[4 * g + 2], y[tb]);
#pragma unroll
          for (int tb = 0; tb < NB; ++tb) y[tb] = mfma16(a[tb].w, u[4 * g + 3], y[tb]);
        }
        float g2[CT];
        float f = 0.f;
#pragma unroll
        for (int tb = 0; tb < NB; ++tb) {
          const f32x2 t01 = activate2<ACT>(y[tb][0], y[tb][1]), t23 = activate2<ACT>(y[tb][2], y[tb][3]);
          const float tv[4] = {t01[0], t01[1], t23[0], t23[1]};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = 4 * tb + r;
            const float t = tv[r];
            f = c == 0 ? t * dX[0] : __builtin_fmaf(t, dX[c], f);
            const float slope = ACT == CDE_ACT_TANH ? __builtin_fmaf(-t, t, 1.f) : 1.f;
            g2[c] = as_P * (dX[c] * slope);
            if constexpr (DCOEFF) gdx[c] = __builtin_fmaf(as_P, t, gdx[c]);
          }
        }
        #pragma unroll
        for (int kk = 0; kk < 4; ++kk) { fa[kk] = P == kk ? f : fa[kk]; fb[kk] = P == 4 + kk ? f : fb[kk]; }
        if (SPLIT ? in_range : valid) {                              // (split: the rows of a unit group by the wave that owns it)
          float* grow = (!upper ? G2 + out_row * G2_COLS + 4 * CT * P : G2hi + out_row * G2_COLS + 4 * CT * (P - NP)) + CT * q;     // rows (h = 4P+q, c = 0..CT-1) of the padded layout
#pragma unroll
          for (int c4 = 0; c4 < CT; c4 += 4)
            stream_store4(grow + c4, g2[c4] * wq, g2[c4 + 1] * wq, g2[c4 + 2] * wq, g2[c4 + 3] * wq);
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) {                               // K step (P, c); 8 independent accumulator chains
          if constexpr (!upper) {
            const float* rowp = w2g[c & 3] + 2 * (NB * P + (c >> 2)) * 8 * W2P_STRIDE;
#pragma unroll
            for (int T1 = 0; T1 < 8; ++T1) gu[T1] = mfma16(rowp[16 * T1], g2[c], gu[T1]);
          } else {
            // upper half: row (h = 4P + q, c), columns 16 T1 + n -- two float4 of the copy laid out for this read
            const float4* rowq = reinterpret_cast<const float4*>(hi.W2t + ((4 * P + q - 16) * 16 + c) * 128 + 8 * n);
            const float4 r0 = rowq[0], r1 = rowq[1];
            const float rv[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
            for (int T1 = 0; T1 < 8; ++T1) gu[T1] = mfma16(rv[T1], g2[c], gu[T1]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      };
#pragma unroll
      for (int P = 0; P < NP; ++P) {
        if (SPLIT && P / per_wave != pw) continue;                   // (wave-uniform: another wave's group)
        group(P, std::false_type{});
      }
      if constexpr (NPX > NP) {
        if (has_hi) {
#pragma clang loop unroll(disable)
          for (int P = NP; P < NPX; ++P) {
            if (SPLIT && P / per_wave != pw) continue;
            group(P, std::true_type{});
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (SPLIT) {
        // the four waves' shares of gu, f (and of d(a.f)/d(dX)) meet: afterwards every wave holds the complete values
        mlp_split_allreduce(xbuf, pw, lane, gu[0], gu[1], nullptr);
        mlp_split_allreduce(xbuf, pw, lane, gu[2], gu[3], nullptr);
        mlp_split_allreduce(xbuf, pw, lane, gu[4], gu[5], nullptr);
        mlp_split_allreduce(xbuf, pw, lane, gu[6], gu[7], nullptr);
        mlp_split_allreduce(xbuf, pw, lane, fa, fb, nullptr);
        if constexpr (DCOEFF && CT == MC) {
          f32x4 g03 = {gdx[0], gdx[1], gdx[2], gdx[3]}, g47 = {gdx[4], gdx[5], gdx[6], gdx[7]};
          mlp_split_allreduce(xbuf, pw, lane, g03, g47, nullptr);
          gdx[0] = g03[0]; gdx[1] = g03[1]; gdx[2] = g03[2]; gdx[3] = g03[3];
          gdx[4] = g47[0]; gdx[5] = g47[1]; gdx[6] = g47[2]; gdx[7] = g47[3];
        }
      }
      if constexpr (DCOEFF) {
        float mine[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          // sum over the four lane quarters (lanes n, n+16, n+32, n+48), then quarter q keeps channels NJ q .. NJ q + NJ - 1
          float v[4];
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            float x = gdx[NJ * qq + j];
            x += __shfl_xor(x, 16, 64);
            x += __shfl_xor(x, 32, 64);
            v[qq] = x;
          }
          mine[j] = q == 0 ? v[0] : q == 1 ? v[1] : q == 2 ? v[2] : v[3];
          const float w = wq * mine[j];
          if (DEGREE == CDE_PATH_CUBIC) { gc0[j] += w; gc1[j] += w * frac; gc2[j] += w * frac * frac; }
          else { gc0[j] -= w / width; gc1[j] += w / width; }
        }
        if (nidx != idx) flush_control_grad(idx);
      }
      row = load_row<DEGREE, CT>(coeffs, sc, n_intervals, nidx, Cr);     // for the next stage; lands during the va phase
      // ---- dL/dY1 = gu * relu'(pre1);  va = W1^T dL/dY1
      float g1[32];
#pragma unroll
      for (int s2 = 0; s2 < 32; ++s2) g1[s2] = (mask >> s2) & 1u ? gu[s2 >> 2][s2 & 3] : 0.f;
      if (valid) {
        float* grow = G1 + out_row * G1_COLS + 4 * q;
#pragma unroll
        for (int T1 = 0; T1 < 8; ++T1)
          stream_store4(grow + 16 * T1, g1[4 * T1] * wq, g1[4 * T1 + 1] * wq, g1[4 * T1 + 2] * wq, g1[4 * T1 + 3] * wq);
      }
      f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = va;
#pragma unroll
      for (int T1 = 0; T1 < 8; ++T1) {
        const float4 a0 = w1t[T1 * 64], a1 = w1t[(8 + T1) * 64];
        va = mfma16(a0.x, g1[4 * T1], va);     vb = mfma16(a1.x, g1[4 * T1], vb);
        va = mfma16(a0.y, g1[4 * T1 + 1], va); vb = mfma16(a1.y, g1[4 * T1 + 1], vb);
        va = mfma16(a0.z, g1[4 * T1 + 2], va); vb = mfma16(a1.z, g1[4 * T1 + 2], vb);
        va = mfma16(a0.w, g1[4 * T1 + 3], va); vb = mfma16(a1.w, g1[4 * T1 + 3], vb);
      }

      __builtin_amdgcn_sched_barrier(0);
      // ---- reverse-time dynamics: dz/ds = -f, da/ds = +a^T df/dz; torchdiffeq 3/8 rule, association preserved
      const f32x4 kya = -fa, kyb = -fb, kaa = va, kab = vb;
      const float third = (float)(1.0 / 3.0);
      if constexpr (BACKPROP) {
        // reverse mode through the step (rk4_backprop.hip): va / vb = J_i^T kb_i
        const float dt3 = ds * third;
        yba = yba + va; ybb = ybb + vb;
        if (stage == 3) {
          kb1a = kb1a + ds * va; kb1b = kb1b + ds * vb;
          kb2a = kb2a - ds * va; kb2b = kb2b - ds * vb;
          kb3a = kb3a + ds * va; kb3b = kb3b + ds * vb;
          sa = kb3a; sb = kb3b;
        } else if (stage == 2) {
          kb2a = kb2a + ds * va; kb2b = kb2b + ds * vb;
          kb1a = kb1a - dt3 * va; kb1b = kb1b - dt3 * vb;
          sa = kb2a; sb = kb2b;
        } else if (stage == 1) {
          kb1a = kb1a + dt3 * va; kb1b = kb1b + dt3 * vb;
          sa = kb1a; sb = kb1b;
        } else {
          sa = yba; sb = ybb;                                        // dL/dy0: what the step hands down
        }
      } else if (stage == 0) {
        ky1a = kya; ky1b = kyb; ka1a = kaa; ka1b = kab;
        za = ya + ds * ky1a * third; zb = yb + ds * ky1b * third;
        sa = aa + ds * ka1a * third; sb = ab + ds * ka1b * third;
      } else if (stage == 1) {
        ky2a = kya; ky2b = kyb; ka2a = kaa; ka2b = kab;
        za = ya + ds * (ky2a - ky1a * third); zb = yb + ds * (ky2b - ky1b * third);
        sa = aa + ds * (ka2a - ka1a * third); sb = ab + ds * (ka2b - ka1b * third);
      } else if (stage == 2) {
        za = ya + ds * (ky1a - ky2a + kya); zb = yb + ds * (ky1b - ky2b + kyb);
        sa = aa + ds * (ka1a - ka2a + kaa); sb = ab + ds * (ka1b - ka2b + kab);
        ky1a = ky1a + 3.f * (ky2a + kya); ky1b = ky1b + 3.f * (ky2b + kyb);
        ka1a = ka1a + 3.f * (ka2a + kaa); ka1b = ka1b + 3.f * (ka2b + kab);
      } else {
        za = ya + (ky1a + kya) * ds * 0.125f; zb = yb + (ky1b + kyb) * ds * 0.125f;
        sa = aa + (ka1a + kaa) * ds * 0.125f; sb = ab + (ka1b + kab) * ds * 0.125f;
      }
      idx = nidx; frac = nfrac;
      __builtin_amdgcn_sched_barrier(0);
    }
    ya = za; yb = zb; aa = sa; ab = sb;
  }
  flush_control_grad(idx);                     // end of this chunk of steps
  if (valid) {
    if constexpr (!BACKPROP) {
      store_units4<4>(y_state + series * Hr, ua, Hr, ya);
      store_units4<4>(y_state + series * Hr, ub, Hr, yb);
    }
    store_units4<4>(a_state + series * Hr, ua, Hr, aa);
    store_units4<4>(a_state + series * Hr, ub, Hr, ab);
  }
}

// ------------------------------------------------------------------------------------------ eight waves per tile (round 4)
// The sweep for SMALL batches (at most two 16-series tiles per CU; 8-channel tiles, no control gradients) on the
// decomposition of K4am's small-batch kernel (dopri5_mlp_adjoint.hip: dopri5_mlp_adjoint_attempt_s8, cde_mlp_adj.h:
// mlp_adjoint_eval_split8): the eight waves of a workgroup share one tile and split every evaluation eight ways -- 144 MFMAs
// per wave instead of 384 in the four-wave form, two waves per SIMD -- every wave carries all of z but only ITS component of
// a (hidden unit 4w + q), wave 0 streams z / stores y, every wave its part of the factor rows and its component of a.
template <typename TT, int DEGREE, int ACT>
__global__ __launch_bounds__(512, 2) void rk4_adjoint_mlp_sweep_s8(
    const float* __restrict__ coeffs, const float* __restrict__ knots, int64_t n_intervals,
    const float* __restrict__ img, float* __restrict__ y_state, float* __restrict__ a_state,
    const TT* __restrict__ sgrid, int64_t k_begin, int64_t k_end, const int64_t* __restrict__ stage_index,
    const float* __restrict__ stage_frac, float* __restrict__ U, float* __restrict__ G2, float* __restrict__ G1,
    float* __restrict__ Z, int64_t B, Dims dims) {
  constexpr int CT = MC;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // the LDS image without its W1 part (b1 | W2 | b2), straight into LDS; this wave's W1 / W1^T tiles into registers
  for (int chunk = W1M_FLOATS / 256 + w; chunk * 256 < ADJ_LDS_FLOATS; chunk += 8) {
    if (chunk * 256 + lane * 4 < ADJ_LDS_FLOATS)
      __builtin_amdgcn_global_load_lds(img + chunk * 256 + lane * 4, lds + chunk * 256, 16, 0, 0);
  }
  float4 w1r[2], w1tr[2];
  {
    const float4* w1img = reinterpret_cast<const float4*>(img) + lane;
    w1r[0] = w1img[(2 * w) * 64]; w1r[1] = w1img[(2 * w + 1) * 64];
    const float4* w1t_base = reinterpret_cast<const float4*>(img + ADJ_LDS_FLOATS) + lane;
    w1tr[0] = w1t_base[w * 64]; w1tr[1] = w1t_base[(8 + w) * 64];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const float4 b1r = (reinterpret_cast<const float4*>(lds + W1M_FLOATS) + (lane >> 4))[4 * w];
  float* xa = lds;                                                 // exchange windows: the W1 image's 16 KB ..
  float* xb = lds + ADJ_LDS_FLOATS;                                // .. and 9 KB behind the image
  const int Hr = dims.H, Cr = dims.C;
  const int n = lane & 15, q = lane >> 4;
  const int64_t tile = blockIdx.x;
  const int64_t series = tile * 16 + n;
  const bool in_range = series < B;
  const int64_t sc = in_range ? series : B - 1;
  const int hw = 4 * w + q;
  const bool own = in_range && hw < Hr;
  const int w2y_off = ((n >> 3) * 8 + w2p_residue(n >> 2, n & 3)) * W2P_STRIDE + 4 * q;
  int w2g_off[4];
#pragma unroll
  for (int cl = 0; cl < 4; ++cl) w2g_off[cl] = ((q >> 1) * 8 + w2p_residue(q, cl)) * W2P_STRIDE + n;
  const int ua = q, ub = 16 + q;
  f32x4 ya = load_units4<4>(y_state + sc * Hr, ua, Hr), yb = load_units4<4>(y_state + sc * Hr, ub, Hr);
  float aw = own ? a_state[sc * Hr + hw] : 0.f;                    // a == 0 stays 0: padded lanes / units contribute nothing
  int64_t idx = stage_index[4 * k_begin];
  float frac = stage_frac[4 * k_begin];
  const float zero8[CT] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  for (int64_t k = k_begin; k < k_end; ++k) {
    const float ds = (float)(sgrid[k + 1] - sgrid[k]);
    f32x4 ky1a = {0.f, 0.f, 0.f, 0.f}, ky1b = ky1a, ky2a = ky1a, ky2b = ky1a;
    float ka1 = 0.f, ka2 = 0.f;
    f32x4 za = ya, zb = yb;
    float sw = aw;
#pragma clang loop unroll(disable)
    for (int stage = 0; stage < 4; ++stage) {
      const Row<DEGREE, CT> row = load_row<DEGREE, CT>(coeffs, sc, n_intervals, idx, Cr);
      float dX[CT];
      const float width = DEGREE == CDE_PATH_LINEAR ? knots[idx + 1] - knots[idx] : 1.f;
      control_slope<DEGREE, CT>(row, frac, width, dX);
      const int64_t e_next = 4 * k + stage + 1;
      const bool more = e_next < 4 * k_end;
      const int64_t nidx = more ? stage_index[e_next] : idx;
      const float nfrac = more ? stage_frac[e_next] : frac;
      const float wq = ((stage == 0 || stage == 3) ? 0.125f : 0.375f) * ds;     // 3/8-rule quadrature weight
      const int64_t out_row = ((k - k_begin) * 4 + stage) * B + series;          // (stage, series)
      const float zs[8] = {za[0], za[1], za[2], za[3], zb[0], zb[1], zb[2], zb[3]};
      f32x4 fa, fb;
      float va_w, kt;
      mlp_adjoint_eval_split8<ACT, false, true>(lds, xa, xb, lane, q, w, w1r, b1r, w1tr, w2y_off, w2g_off, zs, sw, dX, zero8,
                                                in_range, U + out_row * U_COLS + 4 * q, Z + out_row * Z_COLS,
                                                G2 + out_row * G2_COLS + CT * q, G1 + out_row * G1_COLS + 4 * q, Hr, fa, fb,
                                                va_w, kt, false, nullptr, wq);
      // ---- reverse-time dynamics: dz/ds = -f, da/ds = +a^T df/dz; torchdiffeq 3/8 rule, association preserved
      const f32x4 kya = -fa, kyb = -fb;
      const float ka = va_w;
      const float third = (float)(1.0 / 3.0);
      if (stage == 0) {
        ky1a = kya; ky1b = kyb; ka1 = ka;
        za = ya + ds * ky1a * third; zb = yb + ds * ky1b * third;
        sw = aw + ds * ka1 * third;
      } else if (stage == 1) {
        ky2a = kya; ky2b = kyb; ka2 = ka;
        za = ya + ds * (ky2a - ky1a * third); zb = yb + ds * (ky2b - ky1b * third);
        sw = aw + ds * (ka2 - ka1 * third);
      } else if (stage == 2) {
        za = ya + ds * (ky1a - ky2a + kya); zb = yb + ds * (ky1b - ky2b + kyb);
        sw = aw + ds * (ka1 - ka2 + ka);
        ky1a = ky1a + 3.f * (ky2a + kya); ky1b = ky1b + 3.f * (ky2b + kyb);
        ka1 = ka1 + 3.f * (ka2 + ka);
      } else {
        za = ya + (ky1a + kya) * ds * 0.125f; zb = yb + (ky1b + kyb) * ds * 0.125f;
        sw = aw + (ka1 + ka) * ds * 0.125f;
      }
      idx = nidx; frac = nfrac;
    }
    ya = za; yb = zb; aw = sw;
  }
  if (in_range && w == 0) {
    store_units4<4>(y_state + series * Hr, ua, Hr, ya);
    store_units4<4>(y_state + series * Hr, ub, Hr, yb);
  }
  if (own) a_state[series * Hr + hw] = aw;
}

// ------------------------------------------------------------------------------------------ host side
size_t mlp_adjoint_image_bytes() { return (size_t)MLP_ADJ_IMAGE_HI_FLOATS * sizeof(float); }

int launch_mlp_adjoint_images(const void* W1, const void* b1, int64_t width, const void* W2, const void* b2, int64_t C,
                              int64_t H, float* img, hipStream_t s) {
  mlp_adj_image_kernel<<<(MLP_ADJ_IMAGE_HI_FLOATS + 255) / 256, 256, 0, s>>>(
      (const float*)W1, (const float*)b1, (const float*)W2, (const float*)b2, img, MlpDims{(int)H, (int)C, (int)width},
      C > MC ? 4 : 2);                          // channel blocks per unit group: 32 units x 8 channels or 16 x 16
  return check_launch();
}

constexpr int64_t K3M_S8_MAX_TILES = 1536;    // 24576 series (measured: 12288: 34.4 -> 20.9 ms, 24576: 44.1 -> 41.9, 32768: 47.6 vs 52.2)

template <typename TT>
int launch_mlp_adjoint_sweep(const void* coeffs, const void* knots, int64_t n_intervals, int degree, int act,
                             const float* img, void* y_state, void* a_state, const void* sgrid, int64_t k_begin,
                             int64_t k_end, const int64_t* stage_index, const void* stage_frac, void* U, void* G2,
                             void* G1, void* Z, int64_t B, int64_t C, int64_t H, void* grad_coeffs, hipStream_t s) {
  if (k_end <= k_begin) return CDE_OK;
  const Dims dims{(int)H, (int)C};
  // up to 512 tiles (two rounds of one workgroup per CU; the 8-wave form would use a quarter of the CUs there): four
  // waves per tile (the split form)
  const int64_t tiles = (B + 15) / 16;
  // (the eight-wave form of 8-channel tiles runs ~7 ms per round of 256 tiles: it beats the one-wave-per-tile form up to
  //  CDE_OPT_K3M_S8_TILES tiles; measured crossover in profiles/NOTES.md)
  const int64_t s8_req = option(CDE_OPT_K3M_S8_TILES);                    // (-1: the default; an override can only LOWER the measured limit)
  const int64_t s8_tiles = s8_req < 0 || s8_req > K3M_S8_MAX_TILES ? K3M_S8_MAX_TILES : s8_req;
  const bool s8_shape = C <= MC && !grad_coeffs && !option(CDE_OPT_K3M_SPLIT4);
  // (control gradients of the 16-channel layout: the one-wave-per-tile form at every batch size)
  const bool split = tiles <= (s8_shape ? (s8_tiles > 512 ? s8_tiles : 512) : 512) && !option(CDE_OPT_K3M_NO_SPLIT) &&
                     !(grad_coeffs && C > MC);
  const unsigned blocks = split ? (unsigned)tiles : (unsigned)((B + 127) / 128);
  const unsigned threads = split ? 256 : 512;
  const size_t lds = (size_t)ADJ_LDS_FLOATS * sizeof(float) + (split ? (size_t)4 * 64 * 9 * sizeof(float) : 0);
  // ... eight waves per tile (everything split eight ways) for 8-channel tiles without control gradients
  if (split && s8_shape && tiles <= (s8_tiles > 512 ? s8_tiles : 512)) {
#define CDE_SWEEP8(D, A)                                                                                           \
  do {                                                                                                             \
    (void)hipFuncSetAttribute((const void*)rk4_adjoint_mlp_sweep_s8<TT, D, A>,                                     \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                               \
    rk4_adjoint_mlp_sweep_s8<TT, D, A><<<blocks, 512, lds, s>>>(                                                   \
        (const float*)coeffs, (const float*)knots, n_intervals, img, (float*)y_state, (float*)a_state,             \
        (const TT*)sgrid, k_begin, k_end, stage_index, (const float*)stage_frac, (float*)U, (float*)G2, (float*)G1, \
        (float*)Z, B, dims);                                                                                       \
  } while (0)
    if (degree != CDE_PATH_CUBIC && degree != CDE_PATH_LINEAR) return CDE_ERR_UNSUPPORTED;
    if (act == CDE_ACT_NONE) {
      if (degree == CDE_PATH_CUBIC) CDE_SWEEP8(CDE_PATH_CUBIC, CDE_ACT_NONE); else CDE_SWEEP8(CDE_PATH_LINEAR, CDE_ACT_NONE);
    } else if (act == CDE_ACT_TANH) {
      if (degree == CDE_PATH_CUBIC) CDE_SWEEP8(CDE_PATH_CUBIC, CDE_ACT_TANH); else CDE_SWEEP8(CDE_PATH_LINEAR, CDE_ACT_TANH);
    } else return CDE_ERR_UNSUPPORTED;
#undef CDE_SWEEP8
    return check_launch();
  }
#define CDE_SWEEP_L(D, A, X, CTV, SPL, GC)                                                                         \
  do {                                                                                                             \
    (void)hipFuncSetAttribute((const void*)rk4_adjoint_mlp_sweep<TT, D, A, X, CTV, SPL>,                           \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                               \
    rk4_adjoint_mlp_sweep<TT, D, A, X, CTV, SPL><<<blocks, threads, lds, s>>>(                                     \
        (const float*)coeffs, (const float*)knots, n_intervals, img, (float*)y_state, (float*)a_state,             \
        (const TT*)sgrid, k_begin, k_end, stage_index, (const float*)stage_frac, (float*)U, (float*)G2, (float*)G1, \
        (float*)Z, B, dims, GC);                                                                                   \
  } while (0)
#define CDE_SWEEP_X(D, A, X)                                                                                       \
  do {                                                                                                             \
    if (C > MC) {                              /* 16 channels x 16 units */                                        \
      if (X) CDE_SWEEP_L(D, A, X, 16, false, (float*)grad_coeffs);                                                 \
      else if (split) CDE_SWEEP_L(D, A, false, 16, true, nullptr);                                                 \
      else CDE_SWEEP_L(D, A, false, 16, false, nullptr);                                                           \
      break;                                                                                                       \
    }                                                                                                              \
    if (split) CDE_SWEEP_L(D, A, X, MC, true, (float*)grad_coeffs);                                                \
    else CDE_SWEEP_L(D, A, X, MC, false, (float*)grad_coeffs);                                                     \
  } while (0)
#define CDE_SWEEP(D, A)                                                                                            \
  do {                                                                                                             \
    if (grad_coeffs) CDE_SWEEP_X(D, A, true); else CDE_SWEEP_X(D, A, false);                                       \
  } while (0)
  if (degree != CDE_PATH_CUBIC && degree != CDE_PATH_LINEAR) return CDE_ERR_UNSUPPORTED;
  if (act == CDE_ACT_NONE) {
    if (degree == CDE_PATH_CUBIC) CDE_SWEEP(CDE_PATH_CUBIC, CDE_ACT_NONE); else CDE_SWEEP(CDE_PATH_LINEAR, CDE_ACT_NONE);
  } else if (act == CDE_ACT_TANH) {
    if (degree == CDE_PATH_CUBIC) CDE_SWEEP(CDE_PATH_CUBIC, CDE_ACT_TANH); else CDE_SWEEP(CDE_PATH_LINEAR, CDE_ACT_TANH);
  } else return CDE_ERR_UNSUPPORTED;
#undef CDE_SWEEP
#undef CDE_SWEEP_X
#undef CDE_SWEEP_L
  return check_launch();
}

// the reverse-mode form (adjoint=False): one wave per tile, steps k_end-1 .. k_begin of the forward grid
template <typename TT>
int launch_mlp_backprop_sweep(const void* coeffs, const void* knots, int64_t n_intervals, int degree, int act,
                              const float* img, const void* stages, int64_t n_steps_total, void* g_state, const void* grid,
                              int64_t k_begin, int64_t k_end, const int64_t* stage_index, const void* stage_frac, void* U,
                              void* G2, void* G1, void* Z, int64_t B, int64_t C, int64_t H, void* grad_coeffs,
                              hipStream_t s) {
  if (k_end <= k_begin) return CDE_OK;
  const Dims dims{(int)H, (int)C};
  const unsigned blocks = (unsigned)((B + 127) / 128);
  const size_t lds = (size_t)ADJ_LDS_FLOATS * sizeof(float);
#define CDE_BP_L(D, A, X, CTV)                                                                                     \
  do {                                                                                                             \
    (void)hipFuncSetAttribute((const void*)rk4_adjoint_mlp_sweep<TT, D, A, X, CTV, false, true>,                   \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                               \
    rk4_adjoint_mlp_sweep<TT, D, A, X, CTV, false, true><<<blocks, 512, lds, s>>>(                                 \
        (const float*)coeffs, (const float*)knots, n_intervals, img, nullptr, (float*)g_state, (const TT*)grid,    \
        k_begin, k_end, stage_index, (const float*)stage_frac, (float*)U, (float*)G2, (float*)G1, (float*)Z, B,    \
        dims, (float*)grad_coeffs, (const float*)stages, n_steps_total);                                           \
  } while (0)
#define CDE_BP(D, A)                                                                                               \
  do {                                                                                                             \
    if (C > MC) { if (grad_coeffs) CDE_BP_L(D, A, true, 16); else CDE_BP_L(D, A, false, 16); }                     \
    else if (grad_coeffs) CDE_BP_L(D, A, true, MC);                                                                \
    else CDE_BP_L(D, A, false, MC);                                                                                \
  } while (0)
  if (degree != CDE_PATH_CUBIC && degree != CDE_PATH_LINEAR) return CDE_ERR_UNSUPPORTED;
  if (act == CDE_ACT_NONE) {
    if (degree == CDE_PATH_CUBIC) CDE_BP(CDE_PATH_CUBIC, CDE_ACT_NONE); else CDE_BP(CDE_PATH_LINEAR, CDE_ACT_NONE);
  } else if (act == CDE_ACT_TANH) {
    if (degree == CDE_PATH_CUBIC) CDE_BP(CDE_PATH_CUBIC, CDE_ACT_TANH); else CDE_BP(CDE_PATH_LINEAR, CDE_ACT_TANH);
  } else return CDE_ERR_UNSUPPORTED;
#undef CDE_BP
#undef CDE_BP_L
  return check_launch();
}
template int launch_mlp_backprop_sweep<float>(const void*, const void*, int64_t, int, int, const float*, const void*, int64_t,
                                              void*, const void*, int64_t, int64_t, const int64_t*, const void*, void*, void*,
                                              void*, void*, int64_t, int64_t, int64_t, void*, hipStream_t);
template int launch_mlp_backprop_sweep<double>(const void*, const void*, int64_t, int, int, const float*, const void*, int64_t,
                                               void*, const void*, int64_t, int64_t, const int64_t*, const void*, void*, void*,
                                               void*, void*, int64_t, int64_t, int64_t, void*, hipStream_t);

template int launch_mlp_adjoint_sweep<float>(const void*, const void*, int64_t, int, int, const float*, void*, void*,
                                             const void*, int64_t, int64_t, const int64_t*, const void*, void*, void*,
                                             void*, void*, int64_t, int64_t, int64_t, void*, hipStream_t);
template int launch_mlp_adjoint_sweep<double>(const void*, const void*, int64_t, int, int, const float*, void*, void*,
                                              const void*, int64_t, int64_t, const int64_t*, const void*, void*, void*,
                                              void*, void*, int64_t, int64_t, int64_t, void*, hipStream_t);

}  // namespace cde
