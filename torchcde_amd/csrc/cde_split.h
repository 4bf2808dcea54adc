// cde_split.h -- pieces shared by the workgroup-per-tile kernels (rk4_split.hip, dopri5_adjoint.hip): LDS layouts,
// the per-wave MFMA A images and the small helpers.  See the header of rk4_split.hip for the tiling itself.
#pragma once
#include "cde_mfma.h"

namespace cde {

constexpr int SPL_ZROW = 36;                  // stage-state buffer: 32 units + 4 pad floats per series
constexpr int SPL_ZBUF = 16 * SPL_ZROW;
constexpr int SPL_TROW = 20;                  // transposed tiles: 16 series + 4 pad floats per row
constexpr int SPL_ZT = 32 * SPL_TROW;
constexpr int SPL_VROW = 12;                  // va partials: [w_dst][q][n][w_src*2 + j], 8 + 4 pad floats per lane
constexpr int SPL_VA = 4 * 64 * SPL_VROW;     //   (n fastest: conflict-free b64 writes per 16-lane group and b128 reads)
constexpr int SPL_DXROW = 12;                 // shared control derivative: [series][8 channels + 4 pad]
constexpr int SPL_DX = 16 * SPL_DXROW;
constexpr int SPL_GT = 64 * SPL_TROW;         // per wave: transposed weighted dL/dY tile (64 rows)
constexpr int64_t SPL_PARTIAL_FLOATS = MH * MC * MH + MH * MC;     // == PARTIAL_FLOATS of rk4_mfma.hip

// position of series n inside a transposed row: MFMA K step s, quarter kq <-> series 4s + kq is read as float4[kq][s]
__device__ __forceinline__ int spl_pos(int n) { return (n & 3) * 4 + (n >> 2); }

// Y-tile A image and bias of wave w (registers)
__device__ __forceinline__ void spl_load_wy(const float* __restrict__ W, const float* __restrict__ bias, int w, int n,
                                            int q, Dims d, float (&wy)[4][8], f32x4 (&by)[4]) {
#pragma unroll
  for (int T = 0; T < 4; ++T) {
    const int hA = 8 * w + 4 * (T >> 1) + (n >> 2), cA = 4 * (T & 1) + (n & 3);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int k = 4 * s + q;
      wy[T][s] = (hA < d.H && cA < d.C && k < d.H) ? W[(hA * d.C + cA) * d.H + k] : 0.f;
    }
    const int hD = 8 * w + 4 * (T >> 1) + q;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int cD = 4 * (T & 1) + r;
      by[T][r] = (hD < d.H && cD < d.C) ? bias[hD * d.C + cD] : 0.f;
    }
  }
}

template <int ACT>
__device__ __forceinline__ float spl_slope(float t) { return ACT == CDE_ACT_TANH ? __builtin_fmaf(-t, t, 1.f) : 1.f; }

// va image of wave w: tile T, K step s' = 8P + c: A[i = n][kq = q] = W[(h = 8w + 4P + q, c)][k_out(T, i)],
// k_out = 8 (2T + (r >> 1)) + 4 (r & 1) + qi  with (qi, r) = (i >> 2, i & 3)
__device__ __forceinline__ void spl_load_wv(const float* __restrict__ W, int w, int n, int q, Dims d, float (&wv)[2][16]) {
#pragma unroll
  for (int T = 0; T < 2; ++T) {
    const int qi = n >> 2, r = n & 3;
    const int k_out = 8 * (2 * T + (r >> 1)) + 4 * (r & 1) + qi;
#pragma unroll
    for (int sp = 0; sp < 16; ++sp) {
      const int h = 8 * w + 4 * (sp >> 3) + q, c = sp & 7;
      wv[T][sp] = (h < d.H && c < d.C && k_out < d.H) ? W[(h * d.C + c) * d.H + k_out] : 0.f;
    }
  }
}

// ---------------------------------------------------------------------------------------------- control feed
// dX/dt of the tile's 16 series is produced ONCE per stage and shared through LDS (dxb[series][channel]) instead of
// 16 times (4 waves x 4 lane quarters): lane (n, q) of wave w produces channel c = 2w + (q & 1) of series n (the
// q >= 2 copies compute the same value and do not store).  The feed runs one stage ahead of its consumers: during
// stage e the producers publish dX for stage e+1, from coefficients that were requested during stage e-1, at table
// entries that were requested during stage e-2 -- no load is waited for in the stage that issued it.
//   cubic : raw = (b, 2c, 3d) of the interval,   dX = b + (2c + 3d frac) frac      (interpolation_cubic.py:334-335)
//   linear: raw = (x_i, x_{i+1}, t_{i+1} - t_i), dX = (x_{i+1} - x_i) / width      (interpolation_linear.py:222-225),
//           the division happens once per interval (when the coefficients are installed), not per stage.
template <int DEGREE>
struct Feed {
  const float* __restrict__ base;      // this lane's (series, channel) entry of interval 0
  const float* __restrict__ knots;
  const int64_t* __restrict__ sidx;    // stage table (interval index: int64 in memory, < 2^31)
  const float* __restrict__ sfrac;
  int stride, part;                    // floats between two intervals / between the parts of one interval
  int e_last;
  bool live;                           // channel < real channel count
  int idx1, idx2;                      // table entries e+1, e+2 (uniform)
  float frac1, frac2;
  float cur[3], raw[3];
  bool pending;                        // raw holds the coefficients of idx1, not yet installed in cur (uniform)
  int ahead, idx_max;                  // rows between an installed interval and the one whose cache line is touched (signed)
  float touched;

  __device__ __forceinline__ void init(const float* coeffs, const float* knots_, const int64_t* si, const float* sf,
                                       int64_t n_intervals, int64_t sc, int Cr, int c) {
    const int cc = c < Cr ? c : Cr - 1;
    live = c < Cr;
    knots = knots_; sidx = si; sfrac = sf;
    part = Cr;
    if (DEGREE == CDE_PATH_CUBIC) { stride = 4 * Cr; base = coeffs + sc * n_intervals * 4 * Cr + Cr + cc; }
    else { stride = Cr; base = coeffs + sc * (n_intervals + 1) * Cr + cc; }
    idx_max = (int)(DEGREE == CDE_PATH_CUBIC ? n_intervals - 1 : n_intervals);
    ahead = 0;
  }
  // Every coefficient row is read exactly once, so the row of a NEW interval comes from HBM (~1 us) although it is needed
  // one stage (~0.5 us) after the stage table names it: a third of that wait was exposed every time the interval changed.
  // When a row is installed, one dword of the row the sweep will most likely need after it (the next 128-byte line in the
  // sweep's direction) is requested and never looked at: by the time request() names it, it sits in this XCD's L2.
  // (a plain load whose value is "used" by an empty asm at the NEXT install: the compiler waits for it there and nowhere
  //  else; a volatile access would also turn the stage table's scalar loads into vector loads)
  __device__ __forceinline__ void touch(int idx) {
    int at = idx + ahead;
    at = at < 0 ? 0 : (at > idx_max ? idx_max : at);
    touched = base[(int64_t)at * stride];
  }
  __device__ __forceinline__ int index_at(int e) const { return (int)sidx[e < e_last ? e : e_last]; }
  __device__ __forceinline__ float frac_at(int e) const { return sfrac[e < e_last ? e : e_last]; }
  __device__ __forceinline__ void request(int idx) {
    const float* p = base + (int64_t)idx * stride;
    if (DEGREE == CDE_PATH_CUBIC) { raw[0] = p[0]; raw[1] = p[part]; raw[2] = p[2 * part]; }
    else { raw[0] = p[0]; raw[1] = p[part]; raw[2] = knots[idx + 1] - knots[idx]; }
  }
  __device__ __forceinline__ void install() {
    if (DEGREE == CDE_PATH_CUBIC) { cur[0] = raw[0]; cur[1] = raw[1]; cur[2] = raw[2]; }
    else cur[0] = (raw[1] - raw[0]) / raw[2];
  }
  __device__ __forceinline__ float value(float frac) const {
    const float v = DEGREE == CDE_PATH_CUBIC ? cubic_derivative(cur[0], cur[1], cur[2], frac) : cur[0];
    return live ? v : 0.f;
  }

  // start at table entry e0 (entries e0 .. last belong to this sweep): returns dX of entry e0
  __device__ __forceinline__ float begin(int e0, int last) {
    e_last = last;
    const int idx0 = index_at(e0);
    const float frac0 = frac_at(e0);
    request(idx0);
    install();
    {
      const int rows_per_line = DEGREE == CDE_PATH_CUBIC ? 1 : (32 / part > 1 ? 32 / part : 1);
      ahead = index_at(last) >= idx0 ? rows_per_line : -rows_per_line;
      touch(idx0);
    }
    const float v0 = value(frac0);
    idx1 = index_at(e0 + 1); frac1 = frac_at(e0 + 1);
    idx2 = index_at(e0 + 2); frac2 = frac_at(e0 + 2);
    pending = idx1 != idx0;
    if (pending) request(idx1);
    return v0;
  }
  // during stage e: returns dX of entry e+1 and moves the pipeline on
  __device__ __forceinline__ float advance(int e) {
    const int idx3 = index_at(e + 3);
    const float frac3 = frac_at(e + 3);
    // `pending` opaque to the optimiser: otherwise it merges this install into the request of the previous stage (same
    // condition) and the load is waited for right where it was issued.  The wait (s_waitcnt vmcnt) sits INSIDE the branch:
    // stages that install nothing wait for nothing, so the line touched below has until the next install to arrive.
    int inst = __builtin_amdgcn_readfirstlane(pending ? 1 : 0);
    asm volatile("" : "+s"(inst));
    if (inst) {
      asm volatile("" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]) : "v"(touched));
      install();
      touch(idx1);
    }
    const float v = value(frac1);
    pending = idx2 != idx1;
    if (pending) request(idx2);
    idx1 = idx2; frac1 = frac2; idx2 = idx3; frac2 = frac3;
    return v;
  }
};

// tile counts of the wide kernels (rk4_wide.hip, the wide dopri5 attempt kernel): NW waves per 16-series tile (wave w
// owns hidden units 8w..8w+7), NB channel blocks of 4
template <int NW, int NB>
struct Wide {
  static constexpr int HP = 8 * NW, CT = 4 * NB, KS = 2 * NW, NT = 2 * NB, MV = NW / 2, KV = 2 * CT;
  static constexpr int ZROW = HP + 4, ZBUF = 16 * ZROW;          // stage state [series][kq * KS + s], unit k = 4 s + kq
  static constexpr int VROW = 2 * NW + 4, VA = NW * 64 * VROW;   // va partials [w_dst][q][n][2 w_src + j]
  static constexpr int DXROW = CT + 4, DX = 16 * DXROW;          // shared control derivative [series][channel]
  static constexpr int CPW = CT / NW;                            // control channels produced per wave (4 or 1)
  static constexpr int GC = HP * CT;                             // columns of a G row
};

// half-wave / 16-lane-row exchanges (shared-Jacobian chain waves: rk4_split.hip, dopri5_adjoint.hip)
__device__ __forceinline__ void swap32s(float& x, float& y) {      // x[lanes 32..63] <-> y[lanes 0..31]
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  x = __uint_as_float(r[0]);
  y = __uint_as_float(r[1]);
}
__device__ __forceinline__ void swap16s(float& x, float& y) {      // x[odd 16-lane rows] <-> y[even 16-lane rows]
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  x = __uint_as_float(r[0]);
  y = __uint_as_float(r[1]);
}

// A image of the Jacobian rows wave w owns (shared-Jacobian form, affine fields): tile t = 2 hi + kh, MFMA row i = 4 q' + r
// <-> J[h = 8w + hi][k = 4 (4 kh + r) + q'], K step st <-> channel 4 st + q; and the bias rows of the lane's two units
__device__ __forceinline__ void spl_load_wj(const float* __restrict__ W, const float* __restrict__ bias, int w, int n, int q,
                                            Dims d, float (&wj)[16][2], f32x2 (&bja)[4], f32x2 (&bjb)[4]) {
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int h = 8 * w + (t >> 1), k = 4 * (4 * (t & 1) + (n & 3)) + (n >> 2);
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      const int c = 4 * st + q;
      wj[t][st] = (h < d.H && c < d.C && k < d.H) ? W[(h * d.C + c) * d.H + k] : 0.f;
    }
  }
  const int ua = 8 * w + q, ub = ua + 4;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    auto bv = [&](int u, int c) { return (u < d.H && c < d.C) ? bias[u * d.C + c] : 0.f; };
    bja[j] = f32x2{bv(ua, 2 * j), bv(ua, 2 * j + 1)};
    bjb[j] = f32x2{bv(ub, 2 * j), bv(ub, 2 * j + 1)};
  }
}

// The stage table (interval index + fraction per stage, a few KB) is read through the scalar cache, one entry per stage, by
// every workgroup in the same stage at the same time: the first reader of a 64-byte line on an XCD waits for HBM (~1 us,
// every 8th stage) and the others with it.  Pull the sweep's part of the table into L2 before the first stage.
__device__ __forceinline__ void touch_stage_table(const int64_t* __restrict__ sidx, const float* __restrict__ sfrac,
                                                  int64_t first, int64_t count) {
  for (int64_t i = (int64_t)threadIdx.x * 8; i < count; i += (int64_t)blockDim.x * 8) {
    const int lo = (int)sidx[first + i];
    asm volatile("" ::"v"(lo));
  }
  for (int64_t i = (int64_t)threadIdx.x * 16; i < count; i += (int64_t)blockDim.x * 16) {
    const float v = sfrac[first + i];
    asm volatile("" ::"v"(v));
  }
}

// workgroup barrier that waits for this wave's LDS traffic only (no vmcnt: global loads stay in flight)
__device__ __forceinline__ void spl_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

}  // namespace cde
