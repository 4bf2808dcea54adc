// Micro-benchmark (round 6): do bf16 MFMAs of one wave overlap with the f32 MFMAs / vector instructions of ANOTHER wave on the
// same SIMD of gfx950?  (f32 MFMA vs VALU: no -- profiles/NOTES.md, round 5.)  512-thread workgroups, waves 0..3 = role A,
// waves 4..7 = role B, one of each per SIMD; per configuration the span (first start .. last end, s_memtime ticks of 100 MHz
// converted with the measured ratio is unnecessary: only ratios matter) of role A alone, role B alone, both.
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

// what: bit 0 = role A active, bit 1 = role B active;  kindB: 0 = f32 32x32x2 MFMAs, 1 = packed FMAs (VALU)
template <int KINDB>
__global__ __launch_bounds__(512) void k(float* out, long long* span, int iters, float seed, int what) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool roleA = wave < 4;
  bf16x8 pa, pb;
  for (int e = 0; e < 8; ++e) { pa[e] = (__bf16)(seed * (lane + e)); pb[e] = (__bf16)(seed + e); }
  f32x16 c0, c1, c2, c3;
  for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; c2[r] = 0.f; c3[r] = 0.f; }
  f32x2 v[8];
  for (int i = 0; i < 8; ++i) v[i] = f32x2{seed + i, seed - i};
  const float fa = seed * lane, fb = seed + lane;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  if (roleA && (what & 1)) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 12; ++i) {                         // 48 bf16 MFMAs, four independent chains
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pb, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pb, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pb, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa, pb, c3, 0, 0, 0);
      }
    }
  }
  if (!roleA && (what & 2)) {
    for (int it = 0; it < iters; ++it) {
      if (KINDB == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {                        // 24 f32 MFMAs (64 cycles each = the 48 bf16 ones' 32)
          c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c2, 0, 0, 0);
          c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c3, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 48; ++i) {                       // 384 packed FMAs
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = __builtin_elementwise_fma(v[j], f32x2{1.0001f, 0.9999f}, f32x2{fa, fb});
        }
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = c0[0] + c1[1] + c2[2] + c3[3];
  for (int i = 0; i < 8; ++i) s += v[i][0] + v[i][1];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (lane == 0) { span[(blockIdx.x * 8 + wave) * 2] = t0; span[(blockIdx.x * 8 + wave) * 2 + 1] = t1; }
}

template <int KINDB>
static double run(int what, int iters) {
  const int blocks = 256;
  float* out; long long* span;
  hipMalloc(&out, blocks * 512 * sizeof(float));
  hipMalloc(&span, blocks * 16 * sizeof(long long));
  k<KINDB><<<blocks, 512>>>(out, span, 10, 1.0f, what);
  hipDeviceSynchronize();
  k<KINDB><<<blocks, 512>>>(out, span, iters, 1.0f, what);
  hipDeviceSynchronize();
  long long h[16];
  hipMemcpy(h, span, sizeof(h), hipMemcpyDeviceToHost);
  long long lo = h[0], hi = h[1];
  for (int w = 0; w < 8; ++w) { lo = h[2 * w] < lo ? h[2 * w] : lo; hi = h[2 * w + 1] > hi ? h[2 * w + 1] : hi; }
  hipFree(out); hipFree(span);
  return (double)(hi - lo) / iters;
}

int main() {
  const int iters = 2000;
  printf("ticks per iteration (workgroup 0: first start .. last end); A = 48 x v_mfma_f32_32x32x16_bf16 in waves 0-3\n");
  double a = run<0>(1, iters), b = run<0>(2, iters), ab = run<0>(3, iters);
  printf("B = 24 x v_mfma_f32_32x32x2_f32 in waves 4-7 :  A alone %.2f   B alone %.2f   both %.2f   (sum %.2f, max %.2f)\n", a, b, ab, a + b, a > b ? a : b);
  a = run<1>(1, iters); b = run<1>(2, iters); ab = run<1>(3, iters);
  printf("B = 384 x v_pk_fma_f32 in waves 4-7          :  A alone %.2f   B alone %.2f   both %.2f   (sum %.2f, max %.2f)\n", a, b, ab, a + b, a > b ? a : b);
  return 0;
}
