// Micro-benchmark: cycles per v_mfma_f32_32x32x2_f32 / 16x16x4_f32 under different operand-preparation patterns.
// One wave per SIMD (grid 256 x 256 threads) unless noted.  Prints shader cycles per MFMA from s_memtime.
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
#define MF32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)
#define MF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters, float seed) {
  const int lane = threadIdx.x;
  float w[8], d[8];
  for (int i = 0; i < 8; ++i) { w[i] = seed * (lane + i); d[i] = seed + i * 0.25f; }
  float z = seed * 3.f + lane;
  float pp[8];
  for (int i = 0; i < 8; ++i) pp[i] = z * d[i];
  f32x16 a0 = {0}, a1 = {0};
  f32x4 b0 = {0}, b1 = {0};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {            // dependent chain, operands ready (no VALU between)
#pragma unroll
      for (int i = 0; i < 8; ++i) a0 = MF32(w[i], d[i], a0);
    } else if (MODE == 1) {     // dependent chain, one v_mul before each MFMA
#pragma unroll
      for (int i = 0; i < 8; ++i) a0 = MF32(w[i], z * d[i], a0);
    } else if (MODE == 2) {     // two chains alternating, one v_mul before each MFMA
#pragma unroll
      for (int i = 0; i < 4; ++i) { a0 = MF32(w[i], z * d[i], a0); a1 = MF32(w[i + 4], z * d[i + 4], a1); }
    } else if (MODE == 3) {     // dependent chain, 8 products first, then 8 MFMAs back to back
      float p[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) p[i] = z * d[i];
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(p[i]));
#pragma unroll
      for (int i = 0; i < 8; ++i) a0 = MF32(w[i], p[i], a0);
    } else if (MODE == 4) {     // 16x16x4: two accumulators sharing the B operand, v_mul per pair
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float b = z * d[i]; b0 = MF16(w[i], b, b0); b1 = MF16(w[(i + 3) & 7], b, b1); }
    } else if (MODE == 5) {     // 16x16x4: products first then 16 MFMAs
      float p[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) p[i] = z * d[i];
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(p[i]));
#pragma unroll
      for (int i = 0; i < 8; ++i) { b0 = MF16(w[i], p[i], b0); b1 = MF16(w[(i + 3) & 7], p[i], b1); }
    } else if (MODE == 6) {     // two chains alternating, products first
      float p[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) p[i] = z * d[i];
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(p[i]));
#pragma unroll
      for (int i = 0; i < 4; ++i) { a0 = MF32(w[i], p[i], a0); a1 = MF32(w[i + 4], p[i + 4], a1); }
    } else if (MODE == 8) {     // two chains alternating; PACKED products for the NEXT group computed ahead
      using f2 = __attribute__((ext_vector_type(2))) float;
      f2 q0 = f2{d[0], d[1]} * z, q1 = f2{d[2], d[3]} * z, q2 = f2{d[4], d[5]} * z, q3 = f2{d[6], d[7]} * z;
      asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
      a0 = MF32(w[0], pp[0], a0); a1 = MF32(w[4], pp[4], a1);
      a0 = MF32(w[1], pp[1], a0); a1 = MF32(w[5], pp[5], a1);
      a0 = MF32(w[2], pp[2], a0); a1 = MF32(w[6], pp[6], a1);
      a0 = MF32(w[3], pp[3], a0); a1 = MF32(w[7], pp[7], a1);
      pp[0] = q0[0]; pp[1] = q0[1]; pp[2] = q1[0]; pp[3] = q1[1]; pp[4] = q2[0]; pp[5] = q2[1]; pp[6] = q3[0]; pp[7] = q3[1];
    } else if (MODE == 9) {     // 16x16x4 pairs; PACKED products for the NEXT group computed ahead
      using f2 = __attribute__((ext_vector_type(2))) float;
      f2 q0 = f2{d[0], d[1]} * z, q1 = f2{d[2], d[3]} * z, q2 = f2{d[4], d[5]} * z, q3 = f2{d[6], d[7]} * z;
      asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
#pragma unroll
      for (int i = 0; i < 8; ++i) { b0 = MF16(w[i], pp[i], b0); b1 = MF16(w[(i + 3) & 7], pp[i], b1); }
      pp[0] = q0[0]; pp[1] = q0[1]; pp[2] = q1[0]; pp[3] = q1[1]; pp[4] = q2[0]; pp[5] = q2[1]; pp[6] = q3[0]; pp[7] = q3[1];
    } else if (MODE == 7) {     // dependent chain with 3 extra independent VALU per MFMA (tail-like filler)
#pragma unroll
      for (int i = 0; i < 8; ++i) { a0 = MF32(w[i], d[i], a0); z = z * 1.0001f + 0.5f; w[(i + 1) & 7] += 1e-9f; }
    }
    z += 1e-3f;
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += a0[i] + a1[i];
  for (int i = 0; i < 4; ++i) s += b0[i] + b1[i];
  out[blockIdx.x * 256 + threadIdx.x] = s + z + w[3] + pp[5];
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE>
void run(const char* name, int per_iter, int blocks, int threads) {
  float* out; long long* cyc; hipMalloc(&out, 4 << 20); hipMalloc(&cyc, 8);
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep) { k<MODE><<<blocks, threads>>>(out, cyc, iters, 1e-6f); hipDeviceSynchronize(); }
  long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  // s_memtime ticks at a fixed 100 MHz on this part?  report raw ticks/MFMA as well as the ratio to MODE 0
  printf("%-62s blocks=%4d thr=%3d  ticks/MFMA = %8.3f\n", name, blocks, threads, (double)h / iters / per_iter);
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0>("32x32x2 dependent chain, operands ready", 8, 256, 256);
  run<1>("32x32x2 dependent chain, v_mul before each", 8, 256, 256);
  run<2>("32x32x2 two chains alternating, v_mul before each", 8, 256, 256);
  run<3>("32x32x2 dependent chain, 8 products then 8 MFMAs", 8, 256, 256);
  run<6>("32x32x2 two chains alternating, products first", 8, 256, 256);
  run<7>("32x32x2 dependent chain + 3 filler VALU per MFMA", 8, 256, 256);
  run<4>("16x16x4 two accs share B, v_mul per pair", 16, 256, 256);
  run<5>("16x16x4 two accs share B, products first", 16, 256, 256);
  run<4>("16x16x4 two accs share B, v_mul per pair, 2 waves/SIMD", 16, 512, 256);
  run<5>("16x16x4 two accs share B, products first, 2 waves/SIMD", 16, 512, 256);
  run<1>("32x32x2 dependent chain, v_mul before each, 2 waves/SIMD", 8, 512, 256);
  run<8>("32x32x2 two chains, packed products one group ahead", 8, 256, 256);
  run<9>("16x16x4 pairs, packed products one group ahead", 16, 256, 256);
  run<9>("16x16x4 pairs, packed products one group ahead, 2 waves/SIMD", 16, 512, 256);
  run<0>("32x32x2 dependent chain, operands ready, 2 waves/SIMD", 8, 512, 256);
  return 0;
}
