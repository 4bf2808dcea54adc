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

// workgroup barrier that waits for this wave's LDS traffic only (no vmcnt: global loads stay in flight)
__device__ __forceinline__ void spl_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

}  // namespace cde
