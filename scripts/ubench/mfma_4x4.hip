// Micro-benchmark + layout probe for v_mfma_f32_4x4x1_16b_f32 on gfx950 (round 6: is a 4-series-per-wave solve viable?).
//   layout: 16 independent 4x4 blocks, block = lane >> 2;  A[i] from lane (block, i), B[j] from lane (block, j),
//           D register r of lane (block, j) = D[row r][col j]   -- checked numerically below.
//   rate  : shader cycles per instruction for several issue patterns (one wave per SIMD unless noted).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
#define MF4(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0)
#define MF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)

__global__ void layout_probe(float* out) {
  const int lane = threadIdx.x;
  const float a = 1.f + lane, b = 100.f + 3.f * lane;
  f32x4 d = {0.f, 0.f, 0.f, 0.f};
  d = MF4(a, b, d);
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = d[r];
}

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, int iters, float seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float w[32], z[8];
  for (int i = 0; i < 32; ++i) w[i] = seed * (lane + i);
  for (int i = 0; i < 8; ++i) z[i] = seed + i * 0.25f + lane;
  f32x4 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  f32x4 h0 = {0}, h1 = {0};
  f32x2 v = {seed, seed};
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {             // 4 independent chains, 32 MFMAs, operands ready
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        a0 = MF4(w[i], z[i], a0); a1 = MF4(w[8 + i], z[i], a1); a2 = MF4(w[16 + i], z[i], a2); a3 = MF4(w[24 + i], z[i], a3);
      }
    } else if (MODE == 1) {      // one dependent chain
#pragma unroll
      for (int i = 0; i < 32; ++i) a0 = MF4(w[i], z[i & 7], a0);
    } else if (MODE == 2) {      // two chains
#pragma unroll
      for (int i = 0; i < 16; ++i) { a0 = MF4(w[i], z[i & 7], a0); a1 = MF4(w[16 + i], z[i & 7], a1); }
    } else if (MODE == 3) {      // 4 chains + one packed FMA per 4 MFMAs
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        a0 = MF4(w[i], z[i], a0); a1 = MF4(w[8 + i], z[i], a1); a2 = MF4(w[16 + i], z[i], a2); a3 = MF4(w[24 + i], z[i], a3);
        v = __builtin_elementwise_fma(v, f32x2{1.0001f, 0.9999f}, f32x2{z[i], z[7 - i]});
      }
    } else if (MODE == 4) {      // 4 chains, 32 MFMAs, then a block of 8 packed FMAs (the stage tail)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        a0 = MF4(w[i], z[i], a0); a1 = MF4(w[8 + i], z[i], a1); a2 = MF4(w[16 + i], z[i], a2); a3 = MF4(w[24 + i], z[i], a3);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) v = __builtin_elementwise_fma(v, f32x2{1.0001f, 0.9999f}, f32x2{z[i], z[7 - i]});
    } else if (MODE == 7) {      // 32 MFMAs, then ONE block of 24 independent-ish VALU (grouped by sched_barrier)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        a0 = MF4(w[i], z[i], a0); a1 = MF4(w[8 + i], z[i], a1); a2 = MF4(w[16 + i], z[i], a2); a3 = MF4(w[24 + i], z[i], a3);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 24; ++i) w[i] = __builtin_fmaf(w[i], 1.0001f, z[i & 7]);
      __builtin_amdgcn_sched_barrier(0);
    } else if (MODE == 8) {      // 8 x 16x16x4, then the same block of 24 VALU
#pragma unroll
      for (int i = 0; i < 4; ++i) { h0 = MF16(w[i], z[i], h0); h1 = MF16(w[8 + i], z[i], h1); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 24; ++i) w[i] = __builtin_fmaf(w[i], 1.0001f, z[i & 7]);
      __builtin_amdgcn_sched_barrier(0);
    } else if (MODE == 9) {      // only the 24 VALU
#pragma unroll
      for (int i = 0; i < 24; ++i) w[i] = __builtin_fmaf(w[i], 1.0001f, z[i & 7]);
      __builtin_amdgcn_sched_barrier(0);
    } else if (MODE == 10) {     // 4 MFMAs, 3 VALU, repeated 8 times (fine interleave, grouped)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        a0 = MF4(w[i], z[i], a0); a1 = MF4(w[8 + i], z[i], a1); a2 = MF4(w[16 + i], z[i], a2); a3 = MF4(w[24 + i], z[i], a3);
        __builtin_amdgcn_sched_barrier(0);
        w[i] = __builtin_fmaf(w[i], 1.0001f, z[i]); w[8 + i] = __builtin_fmaf(w[8 + i], 1.0001f, z[i]); w[16 + i] = __builtin_fmaf(w[16 + i], 1.0001f, z[i]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (MODE == 11) {     // roles: waves 0..3 = 32 x 4x4x1 + 24 VALU block; waves 4..7 = 8 x 16x16x4 + 6 VALU
      if (wave < 4) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          a0 = MF4(w[i], z[i], a0); a1 = MF4(w[8 + i], z[i], a1); a2 = MF4(w[16 + i], z[i], a2); a3 = MF4(w[24 + i], z[i], a3);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 24; ++i) w[i] = __builtin_fmaf(w[i], 1.0001f, z[i & 7]);
        __builtin_amdgcn_sched_barrier(0);
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) { h0 = MF16(w[i], z[i], h0); h1 = MF16(w[8 + i], z[i], h1); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 6; ++i) w[i] = __builtin_fmaf(w[i], 1.0001f, z[i & 7]);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (MODE == 5) {      // two roles: waves 0..3 4x4x1 (32 per iteration), waves 4..7 16x16x4 (8 per iteration = same cycles)
      if (wave < 4) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          a0 = MF4(w[i], z[i], a0); a1 = MF4(w[8 + i], z[i], a1); a2 = MF4(w[16 + i], z[i], a2); a3 = MF4(w[24 + i], z[i], a3);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) { h0 = MF16(w[i], z[i], h0); h1 = MF16(w[8 + i], z[i], h1); }
      }
    } else if (MODE == 6) {      // 16x16x4 reference: 8 per iteration, two chains
#pragma unroll
      for (int i = 0; i < 4; ++i) { h0 = MF16(w[i], z[i], h0); h1 = MF16(w[8 + i], z[i], h1); }
    }
    z[0] += 1e-3f;
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  float s = v[0] + v[1];
  for (int i = 0; i < 4; ++i) s += a0[i] + a1[i] + a2[i] + a3[i] + h0[i] + h1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0 && blockIdx.x == 0) { cyc[2 * wave] = t0; cyc[2 * wave + 1] = t1; }
}

template <int MODE>
void run(const char* name, int per_iter, int blocks, int threads) {
  float* out; long long* cyc; hipMalloc(&out, 4 << 20); hipMalloc(&cyc, 8 * 32);
  const int iters = 20000;
  for (int rep = 0; rep < 2; ++rep) { k<MODE><<<blocks, threads>>>(out, cyc, iters, 1e-6f); hipDeviceSynchronize(); }
  long long hh[32]; hipMemcpy(hh, cyc, 8 * 32, hipMemcpyDeviceToHost);
  // block 0: wave 0 alone (the oldest wave wins the issue arbitration), and first start .. last end over all its waves
  const int nw = threads / 64;
  long long lo = hh[0], hi = hh[1];
  for (int i = 0; i < nw; ++i) { lo = hh[2 * i] < lo ? hh[2 * i] : lo; hi = hh[2 * i + 1] > hi ? hh[2 * i + 1] : hi; }
  printf("%-70s blocks=%4d thr=%4d  wave0 ticks/iter = %8.2f  all waves = %8.2f  (/%d = %7.3f)\n", name, blocks, threads,
         (double)(hh[1] - hh[0]) / iters, (double)(hi - lo) / iters, per_iter, (double)(hi - lo) / iters / per_iter);
  hipFree(out); hipFree(cyc);
}

int main() {
  float* out; hipMalloc(&out, 64 * 4 * 4);
  layout_probe<<<1, 64>>>(out);
  float h[256]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int r = 0; r < 4; ++r) {
      const int blk = lane >> 2, j = lane & 3;
      const float want = (1.f + (blk * 4 + r)) * (100.f + 3.f * (blk * 4 + j));   // A from lane (blk, r), B from lane (blk, j)
      if (fabsf(h[lane * 4 + r] - want) > 1e-3f * fabsf(want)) ++bad;
    }
  printf("layout probe (D[r] of lane (blk, j) = A[lane (blk, r)] * B[lane (blk, j)]): %s (%d mismatches)\n", bad ? "DIFFERENT" : "confirmed", bad);
  if (bad) for (int lane = 0; lane < 8; ++lane) printf("  lane %d: %g %g %g %g\n", lane, h[lane * 4], h[lane * 4 + 1], h[lane * 4 + 2], h[lane * 4 + 3]);
  run<0>("4x4x1_16b four chains, operands ready (32 per iter)", 32, 256, 256);
  run<1>("4x4x1_16b one dependent chain (32 per iter)", 32, 256, 256);
  run<2>("4x4x1_16b two chains (32 per iter)", 32, 256, 256);
  run<3>("4x4x1_16b four chains + 1 pk_fma per 4 MFMAs (32 per iter)", 32, 256, 256);
  run<4>("4x4x1_16b four chains, then 8 pk_fma (32 per iter)", 32, 256, 256);
  run<6>("16x16x4 two chains (8 per iter)", 8, 256, 256);
  run<5>("roles: waves 0-3 32 x 4x4x1, waves 4-7 8 x 16x16x4 (512 threads; per SIMD)", 1, 256, 512);
  run<0>("4x4x1_16b four chains, 2 waves per SIMD (32 per iter per wave)", 32, 256, 512);
  run<0>("4x4x1_16b four chains, 4 waves per SIMD (32 per iter per wave)", 32, 256, 1024);
  run<6>("16x16x4 two chains, 2 waves per SIMD (8 per iter per wave)", 8, 256, 512);
  run<7>("32 x 4x4x1, then a block of 24 v_fma", 1, 256, 256);
  run<8>("8 x 16x16x4, then a block of 24 v_fma", 1, 256, 256);
  run<9>("only the block of 24 v_fma", 1, 256, 256);
  run<10>("8 x (4 x 4x4x1, 3 v_fma)", 1, 256, 256);
  run<7>("32 x 4x4x1, then 24 v_fma; 2 waves per SIMD", 1, 256, 512);
  run<8>("8 x 16x16x4, then 24 v_fma; 2 waves per SIMD", 1, 256, 512);
  run<11>("roles: waves 0-3 32 x 4x4x1 + 24 v_fma, waves 4-7 8 x 16x16x4 + 6 v_fma", 1, 256, 512);
  return 0;
}
