// Bounded experiment: could the solver's GEMMs run on the bf16 matrix pipe at float32 accuracy?
//
// "bf16x3": every float32 operand is split into three bf16 pieces (x = x1 + x2 + x3, 8 + 8 + 8 mantissa bits) and the
// product a*b is taken as the six piece products a_i b_j with i + j <= 4, each a v_mfma_f32_32x32x16_bf16 accumulating in
// float32.  The bf16 pipe is 16x the f32 pipe per flop (MI355X_MICROARCH.md), so six products could still be ~2.7x faster
// than one f32 MFMA -- IF the operand splitting (VALU work a wave cannot hide behind its own MFMAs)
// does not eat the gain and IF the accuracy holds.
//
// The test GEMM is one vector-field evaluation in the pre-activation form: Y (256 x 32 series) = W (256 x 32) z (32 x 32),
// W pre-split once, z split anew in every evaluation (the solver state changes every stage):
//     f32      : 8 M-tiles x 16 K-steps of v_mfma_f32_32x32x2_f32        = 128 MFMAs
//     bf16x3   : 8 M-tiles x  2 K-steps x 6 products of 32x32x16_bf16    =  96 MFMAs + the split of 16 z values per lane
// Reported: shader cycles per evaluation (s_memtime, one wave per SIMD and two), and the error of both against a float64
// host computation, relative to max |Y|.
//     hipcc --offload-arch=gfx950 -O3 scripts/ubench/bf16x3_gemm.hip -o scripts/ubench/bf16x3_gemm && scripts/ubench/bf16x3_gemm
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__device__ __forceinline__ void split3(float x, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)x;
  const float r1 = x - (float)a;
  b = (__bf16)r1;
  const float r2 = r1 - (float)b;
  c = (__bf16)r2;
}

// W image for the bf16 path: [tile 8][kstep 2][piece 3][lane 64] x 8 bf16; lane l holds row i = l & 31, k = 16 ks + 8 (l >> 5) + 0..7
// z: lane l holds series j = l & 31, units k = 8 (l >> 5) + 0..7 (+16 for the second K step)
template <int MODE>
__global__ __launch_bounds__(256) void gemm_kernel(const float* __restrict__ W, const float* __restrict__ Z, float* __restrict__ Y,
                                                   long long* cycles, int iters) {
  const int lane = threadIdx.x & 63;
  const int i = lane & 31, half = lane >> 5;
  // ---- operands of this wave (every wave computes the same tile set on its own 32 series)
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const float* z = Z + (size_t)wave * 32 * 32;                  // [series][unit]
  float zr[2][8];
  for (int ks = 0; ks < 2; ++ks)
    for (int e = 0; e < 8; ++e) zr[ks][e] = z[i * 32 + 16 * ks + 8 * half + e];
  f32x16 acc[8];
  long long t0 = 0, t1 = 0;
  if (MODE == 0) {
    // f32 MFMA 32x32x2: A lane (row i, k = half), B lane (col i, k = half); K step s covers units 2s, 2s + 1
    float wa[8][16];
    for (int t = 0; t < 8; ++t)
      for (int s = 0; s < 16; ++s) wa[t][s] = W[(32 * t + i) * 32 + 2 * s + half];
    float zb[16];
    for (int s = 0; s < 16; ++s) zb[s] = z[i * 32 + 2 * s + half];
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        acc[t] = f32x16{0};
#pragma unroll
        for (int s = 0; s < 16; ++s) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[t][s], zb[s], acc[t], 0, 0, 0);
      }
      // the state moves with the result (keeps the loop honest): z_k += 1e-3 * Y[k-th row of tile 0]
#pragma unroll
      for (int s = 0; s < 16; ++s) zb[s] += 1e-3f * acc[0][s];
    }
    t1 = __builtin_amdgcn_s_memtime();
  } else {
    bf16x8 wp[8][2][3];
    for (int t = 0; t < 8; ++t)
      for (int ks = 0; ks < 2; ++ks) {
        for (int e = 0; e < 8; ++e) {
          __bf16 a, b, c;
          split3(W[(32 * t + i) * 32 + 16 * ks + 8 * half + e], a, b, c);
          wp[t][ks][0][e] = a; wp[t][ks][1][e] = b; wp[t][ks][2][e] = c;
        }
      }
    t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
      bf16x8 zp[2][3];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          __bf16 a, b, c;
          split3(zr[ks][e], a, b, c);
          zp[ks][0][e] = a; zp[ks][1][e] = b; zp[ks][2][e] = c;
        }
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        acc[t] = f32x16{0};
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          // smallest terms first: a3 b1, a2 b2, a1 b3, then a2 b1, a1 b2, then a1 b1
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[t][ks][2], zp[ks][0], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[t][ks][1], zp[ks][1], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[t][ks][0], zp[ks][2], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[t][ks][1], zp[ks][0], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[t][ks][0], zp[ks][1], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wp[t][ks][0], zp[ks][0], acc[t], 0, 0, 0);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) zr[ks][e] += 1e-3f * acc[0][8 * ks + e];
    }
    t1 = __builtin_amdgcn_s_memtime();
  }
  // D layout of 32x32: register r, lane l -> row 8 (r >> 2) + 4 (l >> 5) + (r & 3), column l & 31
  for (int t = 0; t < 8; ++t)
    for (int r = 0; r < 16; ++r) {
      const int row = 32 * t + 8 * (r >> 2) + 4 * half + (r & 3);
      Y[((size_t)wave * 256 + row) * 32 + i] = acc[t][r];
    }
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

int main() {
  const int n_wg = 256;
  std::vector<float> W(256 * 32), Z;
  srand(7);
  for (auto& w : W) w = (rand() / (float)RAND_MAX * 2.f - 1.f) * 0.177f;     // ~ Linear(32, 256) default init
  float *dW, *dZ, *dY;
  long long* dC;
  hipMalloc(&dW, W.size() * 4);
  hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
  hipMalloc(&dC, 8);
  for (int waves_per_wg : {4, 8}) {
    const int n_waves = n_wg * waves_per_wg;
    Z.resize((size_t)n_waves * 32 * 32);
    for (auto& z : Z) z = (rand() / (float)RAND_MAX * 2.f - 1.f) * 2.f;
    hipMalloc(&dZ, Z.size() * 4);
    hipMalloc(&dY, (size_t)n_waves * 256 * 32 * 4);
    hipMemcpy(dZ, Z.data(), Z.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> Y((size_t)n_waves * 256 * 32);
    for (int mode = 0; mode < 2; ++mode) {
      // accuracy: one evaluation
      if (mode == 0) gemm_kernel<0><<<n_wg, 64 * waves_per_wg>>>(dW, dZ, dY, dC, 1);
      else gemm_kernel<1><<<n_wg, 64 * waves_per_wg>>>(dW, dZ, dY, dC, 1);
      hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost);
      double worst = 0, scale = 0, sum2 = 0;
      size_t cnt = 0;
      for (int wv = 0; wv < 8; ++wv)                               // a sample of the waves against float64
        for (int row = 0; row < 256; ++row)
          for (int j = 0; j < 32; ++j) {
            double ref = 0;
            for (int k = 0; k < 32; ++k) ref += (double)W[row * 32 + k] * (double)Z[((size_t)wv * 32 + j) * 32 + k];
            const double err = fabs((double)Y[((size_t)wv * 256 + row) * 32 + j] - ref);
            worst = fmax(worst, err); scale = fmax(scale, fabs(ref)); sum2 += err * err; ++cnt;
          }
      // speed
      const int iters = 2000;
      long long c = 0;
      for (int rep = 0; rep < 3; ++rep) {
        if (mode == 0) gemm_kernel<0><<<n_wg, 64 * waves_per_wg>>>(dW, dZ, dY, dC, iters);
        else gemm_kernel<1><<<n_wg, 64 * waves_per_wg>>>(dW, dZ, dY, dC, iters);
        hipDeviceSynchronize();
        hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost);
      }
      printf("%d wave(s)/SIMD  %-7s  %8.1f cycles per evaluation (256 x 32 x 32)   max err / max|Y| = %.3g   rms err / max|Y| = %.3g\n",
             waves_per_wg / 4, mode == 0 ? "f32" : "bf16x3", (double)c / iters, worst / scale, sqrt(sum2 / cnt) / scale);
    }
    hipFree(dZ); hipFree(dY);
  }
  return 0;
}
