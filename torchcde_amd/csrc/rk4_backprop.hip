// rk4_backprop.hip -- K3d: the backward pass of cdeint(..., method='rk4', adjoint=False), fused.
//
// Reference: solver.py:144,226-227 with adjoint=False calls torchdiffeq.odeint and lets autograd differentiate the
// solver's own operations ("discretise-then-optimise"; README.md:103 calls it the faster mode).  That gradient is the
// exact derivative of the discrete 3/8-rule map, NOT the continuous adjoint K3j integrates: here it is computed the way
// reverse-mode autograd would, from the stage states the forward kernel stored (rk4_forward_mfma<..., SAVE>: the state
// handed to every one of the 4 x n_steps field evaluations, 128 B per series and evaluation), for the affine field
// f(t, z) = J(t) z + beta(t),  J = sum_c dX_c(t) W_c  (the README's Linear(H, H*C); f32, H <= 32, C <= 8) -- and, on K3a's
// product-form stage instead of the shared Jacobian, for its tanh variant (rk4_backprop_act below; the two-layer field's
// reverse-mode sweep is K3m's own kernel with a BACKPROP flag, rk4_mlp_adjoint.hip).
//
// One RK step y1 = y0 + (k1 + 3 (k2 + k3) + k4) dt / 8 with
//     k1 = f(t0, s1 = y0)              k2 = f(t0 + dt/3, s2 = y0 + dt k1 / 3)
//     k3 = f(t0 + 2dt/3, s3 = y0 + dt (k2 - k1/3))     k4 = f(t1, s4 = y0 + dt (k1 - k2 + k3))
// backpropagates g = dL/dy1 as
//     kb1 = kb4 = (dt/8) g,  kb2 = kb3 = (3dt/8) g,  yb = g
//     v4 = J4^T kb4:  yb += v4,  kb1 += dt v4,  kb2 -= dt v4,  kb3 += dt v4
//     v3 = J3^T kb3:  yb += v3,  kb2 += dt v3,  kb1 -= (dt/3) v3
//     v2 = J2^T kb2:  yb += v2,  kb1 += (dt/3) v2
//     v1 = J1^T kb1:  yb += v1                         => dL/dy0 = yb
//     dL/dW[(h,c),k] += kb_i[h] dX_i[c] s_i[k],   dL/db[(h,c)] += kb_i[h] dX_i[c]        (i = 4, 3, 2, 1)
// i.e. per stage ONE Jacobian GEMM (the 32 rows of J on the matrix pipe, K3j's tiling: M = the (h, k) pairs, K = the 8
// channels, N = the series), one matrix-vector product on the vector pipe straight from the MFMA result, and the
// dL/dW product with the batch as MFMA K -- K3j's stage without its `f = J z` half and without the bias rows (the stage
// states are read, not re-integrated): 128 + 128 MFMAs and about 60 % of K3j's vector instructions per stage.
// Output interpolation (torchdiffeq's fixed-grid solvers interpolate linearly between grid points) is linear in the
// grid states: the host hands over, per grid node, the list of (output index, weight) pairs whose gradient lands there.
//
// Layout = K3j's: one wave owns 32 series for the whole sweep, lane (n = l & 31, half = l >> 5) keeps the hidden units
// 2r + half (r = 0..15) of series n; the stored stage rows hold the 32 units in that order (evens, then odds), so a
// lane's 16 values are four 16-byte loads.  Per-wave partial parameter gradients in K3j's layout, summed in tile order
// by reduce_mfma_partials (run-to-run deterministic).
#include "cde_mfma.h"

namespace cde {

int launch_reduce_partials(const float* partial, int64_t n_tiles, void* grad_W, void* grad_b, int H, int C, hipStream_t s);
size_t mfma_adjoint_partial_bytes(int64_t B);
// rk4_adjoint_pair.hip: the same sweep as a chain wave + a helper wave per tile
int launch_backprop_jacobian_pair(const void*, const void*, int64_t, int, const void*, const void*, const void*, int64_t,
                                  const float*, int64_t, const int64_t*, const int64_t*, const float*, void*, void*, void*, int64_t,
                                  int64_t, int64_t, const int64_t*, const float*, float*, hipStream_t);

namespace {

constexpr int64_t BP_PARTIAL_FLOATS = MH * MC * MH + MH * MC;     // == PARTIAL_FLOATS of rk4_mfma.hip
constexpr bool K3D_PAIR_DEFAULT = true;        // 4.68 -> 4.37 ms at the benchmark size, bitwise the same gradients
constexpr int BP_WJ_FLOATS = MH * 64 * 4;                          // the 32 rows of J's A image (no bias rows)

__device__ __forceinline__ void bp_wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void bp_swap32(float& x, float& y) {    // x[lanes 32..63] <-> y[lanes 0..31]
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  x = __uint_as_float(r[0]);
  y = __uint_as_float(r[1]);
}
// A operand of row h, K step s (channels 2s, 2s + 1), lane l: MFMA row i = l & 31 is unit k = rho(i) (rk4_mfma.hip: wj_image)
__device__ __forceinline__ float bp_wj_image(const float* __restrict__ W, int h, int s, int l, Dims d) {
  const int k = rho(l & 31), c = 2 * s + (l >> 5);
  return (c < d.C && k < d.H && h < d.H) ? W[(h * d.C + c) * d.H + k] : 0.f;
}

// ---- the tanh field, f = reshape(tanh(W z + b)) dX (example/irregular_data.py:36-46): no shared Jacobian -- the activation
// sits between the GEMM and the contraction -- so a stage is K3a's (rk4_mfma.hip: rk4_adjoint_act_mfma): per tile T of four
// hidden units the pre-activation Y_T = W_T s_i + b_T (16 MFMAs), g = kb_h dX_c (1 - tanh^2) in-lane, v += W_T^T g
// (16 MFMAs), dL/dW_T += g^T s_i with the batch as MFMA K through a transposed scratch tile (16 MFMAs).
constexpr int BP_WV32_FLOATS = 8 * 16 * 64;          // one 32x32x2 image: 8 tiles x 16 K steps x 64 lanes
constexpr int BP_BY32_FLOATS = 8 * 2 * 16;           // bias image [tile][half][register]
constexpr int BP_SCRA_FLOATS = 2 * 64 * 20;          // per wave: z^T and one transposed g tile
constexpr int BP_ACT_LDS_FLOATS = 2 * BP_WV32_FLOATS + BP_BY32_FLOATS + 4 * BP_SCRA_FLOATS;
__device__ __forceinline__ float bp_wy32_image(const float* __restrict__ W, int T, int s, int l, Dims d) {
  const int i = l & 31, hk = l >> 5;
  const int h = 4 * T + (i >> 3), c = i & 7, k = 2 * s + hk;
  return (h < d.H && c < d.C && k < d.H) ? W[(h * d.C + c) * d.H + k] : 0.f;
}
__device__ __forceinline__ float bp_wv32_image(const float* __restrict__ W, int T, int r, int l, Dims d) {
  const int k = rho(l & 31), hk = l >> 5;
  const int h = 4 * T + (r >> 2), c = (r & 3) + 4 * hk;
  return (h < d.H && c < d.C && k < d.H) ? W[(h * d.C + c) * d.H + k] : 0.f;
}

// DCOEFF: the control tensors require a gradient as well (under adjoint=False autograd reaches them through X.derivative at
// every stage, reference solver.py:117-135): the cotangent of dX_c at a stage is sum_h kb_h act(Y)_hc -- a sum over rows the
// lane holds, as in K3a -- chained to the coefficient row in use (cubic: 1, frac, frac^2 for b, 2c, 3d; linear: -+1/width on
// the two knot values) and added to `grad_coeffs` (zeroed by the caller, layout of `coeffs`) whenever the row changes; one
// lane owns a (series, channel): no atomics, run-to-run deterministic.  Both activations (the identity field's control
// gradients take this product-form stage too: the shared-Jacobian stage never forms act(Y)).
template <int DEGREE, int ACT, bool DCOEFF>
__global__ __launch_bounds__(256, 1) void rk4_backprop_act(
    const float* __restrict__ coeffs, const float* __restrict__ knots, int64_t n_intervals,
    const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ stages,
    const float* __restrict__ grad_out, int64_t n_out, const float* __restrict__ step_dt, int64_t n_steps,
    const int64_t* __restrict__ node_ptr, const int64_t* __restrict__ node_out, const float* __restrict__ node_weight,
    float* __restrict__ grad_z0, float* __restrict__ partial, int64_t B, const int64_t* __restrict__ stage_index,
    const float* __restrict__ stage_frac, Dims dims, float* __restrict__ grad_coeffs) {
  const int Hr = dims.H, Cr = dims.C;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* wyf = lds;
  float* wvf = lds + BP_WV32_FLOATS;
  float* byf = lds + 2 * BP_WV32_FLOATS;
  for (int e = threadIdx.x; e < BP_WV32_FLOATS; e += 256) {
    const int j = e & 3, l = (e >> 2) & 63, g = e >> 8;             // g = 4T + (step >> 2)
    wyf[e] = bp_wy32_image(W, g >> 2, 4 * (g & 3) + j, l, dims);
    wvf[e] = bp_wv32_image(W, g >> 2, 4 * (g & 3) + j, l, dims);
  }
  for (int e = threadIdx.x; e < BP_BY32_FLOATS; e += 256) {
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
  float* scr_zt = lds + 2 * BP_WV32_FLOATS + BP_BY32_FLOATS + wave * BP_SCRA_FLOATS;   // 64 rows x 20
  float* scr_g = scr_zt + 64 * 20;                                                     // 64 rows x 20

  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  float* my_partial = partial + tile * BP_PARTIAL_FLOATS;
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
  auto add_outputs = [&](int64_t m, f32x16& g) {
    for (int64_t e = node_ptr[m]; e < node_ptr[m + 1]; ++e) {
      const int64_t j = node_out[e];
      const float wgt = node_weight[e];
      if (valid) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int u = 2 * r + half;
          if (u < Hr) g[r] = __builtin_fmaf(wgt, grad_out[(sc * n_out + j) * Hr + u], g[r]);
        }
      }
    }
  };
  f32x16 gy;
#pragma unroll
  for (int r = 0; r < 16; ++r) gy[r] = 0.f;
  add_outputs(n_steps, gy);

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

  if (n_steps > 0) {
    int64_t idx = stage_index[4 * n_steps - 1];
    float frac = stage_frac[4 * n_steps - 1];
    Row<DEGREE> row = load_row<DEGREE>(coeffs, sc, n_intervals, idx, Cr);
    const float* srow = stages + (sc * n_steps * 4) * 32;                   // + (4 k + stage) * 32: plain unit order
    for (int64_t k = n_steps - 1; k >= 0; --k) {
      const float dt = step_dt[k];
      const float third = (float)(1.0 / 3.0);
      const float c8 = dt * 0.125f, dt3 = dt * third;
      f32x16 kb1 = gy * c8, kb2 = gy * (3.f * c8), kb3 = kb2, kbc = kb1;
      f32x16 yb = gy;
#pragma unroll
      for (int stage = 3; stage >= 0; --stage) {
        float dX[MC];
        const float width = DEGREE == CDE_PATH_LINEAR ? knots[idx + 1] - knots[idx] : 1.f;
        control_slope<DEGREE>(row, frac, width, dX);
        f32x16 sst;                                   // the stored stage state: units 2r + half
        if constexpr (ACT == CDE_ACT_TANH) {          // (the tanh forward stores its rows in plain unit order ...
          const float* sp = srow + (4 * k + stage) * 32 + half;
#pragma unroll
          for (int r = 0; r < 16; ++r) sst[r] = sp[2 * r];
        } else {                                      //  ... the product-form forward of the identity field evens, then odds)
          const float4* sp = reinterpret_cast<const float4*>(srow + (4 * k + stage) * 32 + 16 * half);
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const float4 v4 = sp[g4];
            sst[4 * g4] = v4.x; sst[4 * g4 + 1] = v4.y; sst[4 * g4 + 2] = v4.z; sst[4 * g4 + 3] = v4.w;
          }
        }
        const int64_t e_next = 4 * k + stage - 1;
        const bool more = e_next >= 0;
        const int64_t nidx = more ? stage_index[e_next] : idx;
        const float nfrac = more ? stage_frac[e_next] : frac;
        if (nidx != idx) row = load_row<DEGREE>(coeffs, sc, n_intervals, nidx, Cr);
        const float dh[4] = {half ? dX[4] : dX[0], half ? dX[5] : dX[1], half ? dX[6] : dX[2], half ? dX[7] : dX[3]};

        // z^T for the dL/dW products: scr_zt[(par*32 + u)*20 + s] = s_i's unit u of series 2s+par
        {
          float* wz = scr_zt + ((n & 1) * 32 + half) * 20 + (n >> 1);
#pragma unroll
          for (int r = 0; r < 16; ++r) wz[r * 40] = sst[r];
          bp_wave_lds_sync();
        }
        float zB[16];
        {
          const float4* zt4 = reinterpret_cast<const float4*>(scr_zt + (half * 32 + n) * 20);
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const float4 v4 = zt4[g4];
            zB[4 * g4] = v4.x; zB[4 * g4 + 1] = v4.y; zB[4 * g4 + 2] = v4.z; zB[4 * g4 + 3] = v4.w;
          }
        }
        f32x16 v = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float gdx[4] = {0.f, 0.f, 0.f, 0.f};        // cotangent of dX_c, this lane's 4 channels (DCOEFF)
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
            y = mfma(a4.x, sst[4 * g4], y);
            y = mfma(a4.y, sst[4 * g4 + 1], y);
            y = mfma(a4.z, sst[4 * g4 + 2], y);
            y = mfma(a4.w, sst[4 * g4 + 3], y);
          }
          __builtin_amdgcn_sched_barrier(0);
          // ---- activation, dL/dY: kb of units 4T..4T+3 in every lane (swap32 of two copies broadcasts both halves' values)
          float ae0 = kbc[2 * T], ao0 = kbc[2 * T], ae1 = kbc[2 * T + 1], ao1 = kbc[2 * T + 1];
          bp_swap32(ae0, ao0);
          bp_swap32(ae1, ao1);
          const float a4u[4] = {ae0, ao0, ae1, ao1};
          float g[16];
#pragma unroll
          for (int hl = 0; hl < 4; ++hl) {
            const f32x2 tp[2] = {activate2<ACT>(y[4 * hl], y[4 * hl + 1]), activate2<ACT>(y[4 * hl + 2], y[4 * hl + 3])};
#pragma unroll
            for (int cl = 0; cl < 4; ++cl) {
              const float t = tp[cl >> 1][cl & 1];
              g[4 * hl + cl] = a4u[hl] * (dh[cl] * (ACT == CDE_ACT_TANH ? __builtin_fmaf(-t, t, 1.f) : 1.f));
              if constexpr (DCOEFF) gdx[cl] = __builtin_fmaf(a4u[hl], t, gdx[cl]);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          // ---- v += W_T^T g
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const float4 a4 = wvs[(4 * T + g4) * 64];
            v = mfma(a4.x, g[4 * g4], v);
            v = mfma(a4.y, g[4 * g4 + 1], v);
            v = mfma(a4.z, g[4 * g4 + 2], v);
            v = mfma(a4.w, g[4 * g4 + 3], v);
          }
          // ---- dW_T += g^T s_i through the transposed scratch tile
          {
            float* wg = scr_g + ((n & 1) * 32 + 4 * half) * 20 + (n >> 1);     // + row(r) * 20, row = (r&3) + 8*(r>>2)
#pragma unroll
            for (int r = 0; r < 16; ++r) wg[((r & 3) + 8 * (r >> 2)) * 20] = g[r];
            bp_wave_lds_sync();
            const float4* g4p = reinterpret_cast<const float4*>(scr_g + (half * 32 + n) * 20);
            float gA[16];
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const float4 v4 = g4p[g4];
              gA[4 * g4] = v4.x; gA[4 * g4 + 1] = v4.y; gA[4 * g4 + 2] = v4.z; gA[4 * g4 + 3] = v4.w;
            }
            bp_wave_lds_sync();                    // reads retired before the next tile overwrites the scratch
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
            const float w = gdx[cl];                // (kb_i carries the step's dt / 8, 3 dt / 8: no quadrature weight here)
            if (DEGREE == CDE_PATH_CUBIC) { gc0[cl] += w; gc1[cl] += w * frac; gc2[cl] += w * frac * frac; }
            else { gc0[cl] -= w / width; gc1[cl] += w / width; }
          }
          if (nidx != idx) flush_control_grad(idx);
        }
        // ---- reverse-mode bookkeeping of the 3/8 rule (see the file header)
        yb = yb + v;
        if (stage == 3) {
          kb1 = kb1 + dt * v;
          kb2 = kb2 - dt * v;
          kb3 = kb3 + dt * v;
          kbc = kb3;
        } else if (stage == 2) {
          kb2 = kb2 + dt * v;
          kb1 = kb1 - dt3 * v;
          kbc = kb2;
        } else if (stage == 1) {
          kb1 = kb1 + dt3 * v;
          kbc = kb1;
        }
        idx = nidx; frac = nfrac;
      }
      gy = yb;
      add_outputs(k, gy);
    }
    flush_control_grad(idx);
  }
  if (valid) {
#pragma unroll
    for (int r = 0; r < 16; ++r) if (2 * r + half < Hr) grad_z0[series * Hr + 2 * r + half] = gy[r];
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

template <int DEGREE>
__global__ __launch_bounds__(256, 1) void rk4_backprop_jacobian(
    const float* __restrict__ coeffs, const float* __restrict__ knots, int64_t n_intervals,
    const float* __restrict__ W, const float* __restrict__ stages, const float* __restrict__ grad_out, int64_t n_out,
    const float* __restrict__ step_dt, int64_t n_steps, const int64_t* __restrict__ node_ptr,
    const int64_t* __restrict__ node_out, const float* __restrict__ node_weight, float* __restrict__ grad_z0,
    float* __restrict__ partial, int64_t B, const int64_t* __restrict__ stage_index,
    const float* __restrict__ stage_frac, Dims dims) {
  const int Hr = dims.H, Cr = dims.C;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int e = threadIdx.x; e < BP_WJ_FLOATS; e += 256) lds[e] = bp_wj_image(W, e >> 8, e & 3, (e >> 2) & 63, dims);
  __syncthreads();
  const float4* wj = reinterpret_cast<const float4*>(lds);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, half = lane >> 5;
  float* scr = lds + BP_WJ_FLOATS + wave * SCR_FLOATS;

  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  float* my_partial = partial + tile * BP_PARTIAL_FLOATS;
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
  // scratch of the (series -> MFMA K index) transposes, as in K3j:
  //   scr_zt[(par*32 + u)*20 + s] = s_i's unit u of series 2s+par;  scr_at likewise for kb_i;  scr_dw[series*8 + c] = dX_c
  float* scr_zt = scr;
  float* scr_at = scr + 64 * 20;
  float* scr_dw = scr + 2 * 64 * 20;

  // dL/d(grid state m) that comes straight from the outputs: sum over the node's (output, weight) pairs
  auto add_outputs = [&](int64_t m, f32x16& g) {
    for (int64_t e = node_ptr[m]; e < node_ptr[m + 1]; ++e) {
      const int64_t j = node_out[e];
      const float wgt = node_weight[e];
      if (valid) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int u = 2 * r + half;
          if (u < Hr) g[r] = __builtin_fmaf(wgt, grad_out[(sc * n_out + j) * Hr + u], g[r]);
        }
      }
    }
  };
  f32x16 gy;                                    // dL/dy at the current grid node (padded lanes / units: 0)
#pragma unroll
  for (int r = 0; r < 16; ++r) gy[r] = 0.f;
  add_outputs(n_steps, gy);

  if (n_steps > 0) {
    int64_t idx = stage_index[4 * n_steps - 1];
    float frac = stage_frac[4 * n_steps - 1];
    Row<DEGREE> row = load_row<DEGREE>(coeffs, sc, n_intervals, idx, Cr);
    const float* srow = stages + (sc * n_steps * 4) * 32 + half * 16;          // + (4 k + stage) * 32
    // the stage state the forward pass stored (this lane's 16 units: four 16-byte loads), requested ONE STAGE AHEAD: the
    // rows stream from HBM (2.1 GB per sweep at the benchmark size) and nothing else of a stage can start before them
    auto load_state = [&](int64_t e) {
      f32x16 v16;
      const float4* sp = reinterpret_cast<const float4*>(srow + e * 32);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 v = sp[i];
        v16[4 * i] = v.x; v16[4 * i + 1] = v.y; v16[4 * i + 2] = v.z; v16[4 * i + 3] = v.w;
      }
      return v16;
    };
    f32x16 snext = load_state(4 * n_steps - 1);
    for (int64_t k = n_steps - 1; k >= 0; --k) {
      const float dt = step_dt[k];
      const float third = (float)(1.0 / 3.0);
      const float c8 = dt * 0.125f, dt3 = dt * third;
      f32x16 kb1 = gy * c8, kb2 = gy * (3.f * c8), kb3 = kb2, kbc = kb1;          // kbc: the stage being processed (4 first)
      f32x16 yb = gy;
#pragma unroll
      for (int stage = 3; stage >= 0; --stage) {
        float dX[MC];
        const float width = DEGREE == CDE_PATH_LINEAR ? knots[idx + 1] - knots[idx] : 1.f;
        control_slope<DEGREE>(row, frac, width, dX);
        const f32x16 sst = snext;
        // prefetch the table entry of the stage processed next (one entry down), its stored state and, if the interval
        // changes, its row
        const int64_t e_next = 4 * k + stage - 1;
        const bool more = e_next >= 0;
        if (more) snext = load_state(e_next);
        const int64_t nidx = more ? stage_index[e_next] : idx;
        const float nfrac = more ? stage_frac[e_next] : frac;
        if (nidx != idx) row = load_row<DEGREE>(coeffs, sc, n_intervals, nidx, Cr);

        const f32x2 d01 = {dX[0], dX[1]}, d23 = {dX[2], dX[3]}, d45 = {dX[4], dX[5]}, d67 = {dX[6], dX[7]};
        // ---- stage state and kb_i -> scratch (transposed), control derivative
        {
          float* wz = scr_zt + ((n & 1) * 32 + half) * 20 + (n >> 1);              // + 2r*20
          float* wa = scr_at + ((n & 1) * 32 + half) * 20 + (n >> 1);
#pragma unroll
          for (int r = 0; r < 16; ++r) { wz[r * 40] = sst[r]; wa[r * 40] = kbc[r]; }
          const f32x2 w0 = half ? d45 : d01, w1 = half ? d67 : d23;
          *reinterpret_cast<float4*>(scr_dw + n * 8 + 4 * half) = make_float4(w0[0], w0[1], w1[0], w1[1]);
          bp_wave_lds_sync();
        }

        // ---- J = sum_c dX_c W_c one row h at a time (4 MFMAs leave J[h][k], k = the lane's 16 units, in the lane);
        // v += kb_h J[h][.] follows on the vector pipe.  Rows are issued one ahead of their use (K3j's hazard argument:
        // row h is read only after the 4 MFMAs of row h + 1 have issued behind it in the in-order pipe); the last row
        // is followed by explicit wait states.
        f32x16 v;
        {
          const float bs0 = half ? dX[1] : dX[0], bs1 = half ? dX[3] : dX[2], bs2 = half ? dX[5] : dX[4], bs3 = half ? dX[7] : dX[6];
          f32x2 v2[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v2[j] = f32x2{0.f, 0.f};
          int opaque = 0;                               // the image reads are loop invariant: keep them inside the stage
          asm volatile("" : "+v"(opaque));
          const float4* wp = wj + lane + opaque;
          auto issue = [&](f32x16& J, const float4& a) {
            __builtin_amdgcn_sched_barrier(0);           // everything that still reads the old J stays above
            asm volatile("s_nop 1\n\t"                                         // (operands may be fresh VALU results)
                         "v_mfma_f32_32x32x2_f32 %0, %1, %5, 0\n\t"
                         "v_mfma_f32_32x32x2_f32 %0, %2, %6, %0\n\t"
                         "v_mfma_f32_32x32x2_f32 %0, %3, %7, %0\n\t"
                         "v_mfma_f32_32x32x2_f32 %0, %4, %8, %0"
                         : "=&v"(J) : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(bs0), "v"(bs1), "v"(bs2), "v"(bs3));
            __builtin_amdgcn_sched_barrier(0);
          };
          auto consume = [&](const f32x16& J, float ah) {
            const f32x2 ah2 = {ah, ah};
#pragma unroll
            for (int j = 0; j < 8; ++j) v2[j] = __builtin_elementwise_fma(f32x2{J[2 * j], J[2 * j + 1]}, ah2, v2[j]);
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(v2[j]));       // (keeps the FMAs with their row)
          };
          f32x16 Je, Jo;                                 // rows 2r / 2r + 1 in flight
          float4 a_cur = wp[0], a_nxt = wp[64];
          issue(Je, a_cur);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (r < 15) a_cur = wp[(2 * r + 2) * 64];
            issue(Jo, a_nxt);
            asm volatile("" : "+v"(Je));
            // kb of units 2r, 2r + 1 in both half-lanes (the lower half-lanes own unit 2r, the upper ones 2r + 1)
            float ae = kbc[r], ao = kbc[r];
            bp_swap32(ae, ao);
            consume(Je, ae);
            if (r < 15) {
              a_nxt = wp[(2 * r + 3) * 64];
              issue(Je, a_cur);
              asm volatile("" : "+v"(Jo));
            } else {
              asm volatile("s_nop 15\n\ts_nop 7" : "+v"(Jo));              // 16-pass MFMA result -> VALU read
            }
            consume(Jo, ao);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = v2[r >> 1][r & 1];
        }

        // ---- dL/dW tile c: D[h][k] += sum_series (kb_h dX_c)[series] * s_k[series]; this lane feeds MFMA K index `half`
        // of K-step s2, i.e. series 2*s2 + half, row h = n, column k = n.
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
        bp_wave_lds_sync();   // scratch reads retired before the next stage overwrites it

        // ---- reverse-mode bookkeeping of the 3/8 rule (see the file header)
        yb = yb + v;
        if (stage == 3) {
          kb1 = kb1 + dt * v;
          kb2 = kb2 - dt * v;
          kb3 = kb3 + dt * v;
          kbc = kb3;
        } else if (stage == 2) {
          kb2 = kb2 + dt * v;
          kb1 = kb1 - dt3 * v;
          kbc = kb2;
        } else if (stage == 1) {
          kb1 = kb1 + dt3 * v;
          kbc = kb1;
        }
        idx = nidx; frac = nfrac;
      }
      gy = yb;
      add_outputs(k, gy);
    }
  }
  if (valid) {
#pragma unroll
    for (int r = 0; r < 16; ++r) if (2 * r + half < Hr) grad_z0[series * Hr + 2 * r + half] = gy[r];
  }
  // per-wave partial parameter gradients, K3j's layout
#pragma unroll
  for (int c = 0; c < MC; ++c) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int h = (r & 3) + 8 * (r >> 2) + 4 * half;
      my_partial[(h * MC + c) * MH + n] = accW[c][r];
    }
    const float mine_gb = gbp[c >> 1][c & 1];
    const float other = __shfl_xor(mine_gb, 32, 64);
    if (half == 0) my_partial[MH * MC * MH + n * MC + c] = mine_gb + other;
  }
}

}  // namespace

size_t backprop_workspace_bytes(int64_t B) { return mfma_adjoint_partial_bytes(B); }

int launch_backprop_jacobian(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                             const void* bias, int act, const void* stages, const void* grad_out, int64_t n_out,
                             const float* step_dt, int64_t n_steps,
                             const int64_t* node_ptr, const int64_t* node_out, const float* node_weight, void* grad_z0,
                             void* grad_W, void* grad_b, int64_t B, int64_t C, int64_t H, const int64_t* stage_index,
                             const float* stage_frac, float* partial, void* grad_coeffs, hipStream_t s) {
  const Dims dims{(int)H, (int)C};
  const unsigned blocks = (unsigned)((B + 127) / 128);
  if (act != CDE_ACT_NONE && act != CDE_ACT_TANH) return CDE_ERR_UNSUPPORTED;
  if (act == CDE_ACT_TANH || grad_coeffs) {
    const size_t lds_act = (size_t)BP_ACT_LDS_FLOATS * sizeof(float);
#define CDE_BPA(D, A, X)                                                                                             \
  do {                                                                                                               \
    (void)hipFuncSetAttribute((const void*)rk4_backprop_act<D, A, X>, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                              (int)lds_act);                                                                         \
    rk4_backprop_act<D, A, X><<<blocks, 256, lds_act, s>>>(                                                          \
        (const float*)coeffs, (const float*)knots, n_intervals, (const float*)W, (const float*)bias,                 \
        (const float*)stages, (const float*)grad_out, n_out, step_dt, n_steps, node_ptr, node_out, node_weight,      \
        (float*)grad_z0, partial, B, stage_index, stage_frac, dims, (float*)grad_coeffs);                            \
  } while (0)
#define CDE_BPA_D(D)                                                                                                 \
  do {                                                                                                               \
    if (grad_coeffs) { if (act == CDE_ACT_TANH) CDE_BPA(D, CDE_ACT_TANH, true); else CDE_BPA(D, CDE_ACT_NONE, true); } \
    else CDE_BPA(D, CDE_ACT_TANH, false);                                                                            \
  } while (0)
    if (degree == CDE_PATH_CUBIC) CDE_BPA_D(CDE_PATH_CUBIC);
    else if (degree == CDE_PATH_LINEAR) CDE_BPA_D(CDE_PATH_LINEAR);
    else return CDE_ERR_UNSUPPORTED;
#undef CDE_BPA_D
#undef CDE_BPA
    const int rca = check_launch();
    if (rca != CDE_OK) return rca;
    return launch_reduce_partials(partial, (B + 31) / 32, grad_W, grad_b, (int)H, (int)C, s);
  }
  {
    const int64_t e = option(CDE_OPT_K3D_WAVES);      // 1: this file's one-wave kernel, 2: the pair form (tests compare the two)
    if (e ? e == 2 : K3D_PAIR_DEFAULT)
      return launch_backprop_jacobian_pair(coeffs, knots, n_intervals, degree, W, stages, grad_out, n_out, step_dt, n_steps,
                                           node_ptr, node_out, node_weight, grad_z0, grad_W, grad_b, B, C, H, stage_index,
                                           stage_frac, partial, s);
  }
  const size_t lds = (size_t)(BP_WJ_FLOATS + 4 * SCR_FLOATS) * sizeof(float);
#define CDE_BP(D)                                                                                                    \
  do {                                                                                                               \
    (void)hipFuncSetAttribute((const void*)rk4_backprop_jacobian<D>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                              (int)lds);                                                                             \
    rk4_backprop_jacobian<D><<<blocks, 256, lds, s>>>(                                                               \
        (const float*)coeffs, (const float*)knots, n_intervals, (const float*)W, (const float*)stages,               \
        (const float*)grad_out, n_out, step_dt, n_steps, node_ptr, node_out, node_weight, (float*)grad_z0, partial,  \
        B, stage_index, stage_frac, dims);                                                                           \
  } while (0)
  if (degree == CDE_PATH_CUBIC) CDE_BP(CDE_PATH_CUBIC);
  else if (degree == CDE_PATH_LINEAR) CDE_BP(CDE_PATH_LINEAR);
  else return CDE_ERR_UNSUPPORTED;
#undef CDE_BP
  const int rc = check_launch();
  if (rc != CDE_OK) return rc;
  return launch_reduce_partials(partial, (B + 31) / 32, grad_W, grad_b, (int)H, (int)C, s);
}

}  // namespace cde
