// cde_mfma.h -- device pieces shared by the MFMA kernels (rk4_mfma.hip, dopri5.hip): fragment layouts,
// weight images, packed-multiply helpers, control-row handling and the 16-series vector-field evaluation.
#pragma once
#include <type_traits>
#include "cde_common.h"

namespace cde {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;

constexpr int MH = 32;                 // hidden units the MFMA tiles are built for
constexpr int MC = 8;                  // channels the MFMA tiles are built for
// Smaller problems (H <= 32, C <= 8) run on the same tiles zero-padded: weight images return 0 outside the real
// (H, C), missing hidden units / channels are carried as zeros and never stored.  `Dims` is the real shape.
struct Dims { int H, C; };
constexpr int W1_STEPS = 132;          // 16*8 product steps + 4 bias steps
constexpr int W2_STEPS = 128;
constexpr int W1_FLOATS = W1_STEPS * 64;
constexpr int W2_FLOATS = W2_STEPS * 64;
constexpr int SCR_FLOATS = 2 * 64 * 20 + 32 * 8;     // per wave: z^T, a^T (64 rows x 20) and weighted dX (32 x 8)

// MFMA 32x32 C/D fragment: lane (col = l&31, half = l>>5) register r holds row
// i = (r&3) + 8*(r>>2) + 4*half.  rho maps an output ROW i to the hidden unit stored there so
// that register r of half `half` is hidden unit 2r + half.
__host__ __device__ __forceinline__ int rho(int i) { return 2 * ((i & 3) + 4 * (i >> 3)) + ((i >> 2) & 1); }

// A-operand images (value for MFMA step s, lane l)
__device__ __forceinline__ float w1_image(const float* __restrict__ W, const float* __restrict__ bias, int s, int l,
                                          Dims d) {
  const int h_out = rho(l & 31), hk = l >> 5;
  if (h_out >= d.H) return 0.f;
  if (s < 128) {
    const int j = s >> 3, c = s & 7, k_in = 2 * j + hk;
    return (c < d.C && k_in < d.H) ? W[(h_out * d.C + c) * d.H + k_in] : 0.f;
  }
  const int c = 2 * (s - 128) + hk;
  return c < d.C ? bias[h_out * d.C + c] : 0.f;
}
__device__ __forceinline__ float w2_image(const float* __restrict__ W, int s, int l, Dims d) {
  const int k_out = rho(l & 31), hk = l >> 5;
  const int j = s >> 3, c = s & 7, h_in = 2 * j + hk;
  return (k_out < d.H && c < d.C && h_in < d.H) ? W[(h_in * d.C + c) * d.H + k_out] : 0.f;
}

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// one control row held in registers, always in the 8-channel layout (missing channels are zero):
// cubic -> b, 2c, 3d (24 floats); linear -> x[idx], x[idx+1] (16 floats)
// CT: channels of the tile layout (8 everywhere; 16 in the two-layer kernels' wide variant: 16 channels x 16 hidden units)
template <int DEGREE, int CT = MC>
struct Row {
  float4 v[(DEGREE == CDE_PATH_CUBIC ? 3 : 2) * CT / 4];
};

template <int DEGREE, int CT = MC>
__device__ __forceinline__ Row<DEGREE, CT> load_row(const float* __restrict__ coeffs, int64_t series,
                                                     int64_t n_intervals, int64_t idx, int C = CT) {
  Row<DEGREE, CT> r;
  constexpr int NV = (DEGREE == CDE_PATH_CUBIC ? 3 : 2) * CT / 4;
  if (C == CT) {                                   // 16-byte vector loads (rows are 16-byte aligned when C == CT)
    if (DEGREE == CDE_PATH_CUBIC) {
      const float4* p = reinterpret_cast<const float4*>(coeffs + (series * n_intervals + idx) * 4 * CT + CT);
#pragma unroll
      for (int i = 0; i < NV; ++i) r.v[i] = p[i];
    } else {
      const float4* p = reinterpret_cast<const float4*>(coeffs + (series * (n_intervals + 1) + idx) * CT);
#pragma unroll
      for (int i = 0; i < NV; ++i) r.v[i] = p[i];
    }
  } else {                                         // narrower control: scalar loads into the padded layout
    float* f = reinterpret_cast<float*>(r.v);
    if (DEGREE == CDE_PATH_CUBIC) {
      const float* p = coeffs + (series * n_intervals + idx) * 4 * C;
#pragma unroll
      for (int part = 0; part < 3; ++part)
#pragma unroll
        for (int c = 0; c < CT; ++c) f[part * CT + c] = c < C ? p[(part + 1) * C + c] : 0.f;
    } else {
      const float* p = coeffs + (series * (n_intervals + 1) + idx) * C;
#pragma unroll
      for (int part = 0; part < 2; ++part)
#pragma unroll
        for (int c = 0; c < CT; ++c) f[part * CT + c] = c < C ? p[part * C + c] : 0.f;
    }
  }
  return r;
}

// 4 hidden units of one series, `unit`, `unit + STRIDE`, ... (zeros beyond the real H; vector access when they are
// consecutive and the real H is the padded one)
template <int STRIDE = 1>
__device__ __forceinline__ f32x4 load_units4(const float* __restrict__ row, int unit, int H) {
  if (STRIDE == 1 && H == MH) {
    const float4 v = *reinterpret_cast<const float4*>(row + unit);
    return f32x4{v.x, v.y, v.z, v.w};
  }
  f32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = unit + i * STRIDE < H ? row[unit + i * STRIDE] : 0.f;
  return r;
}
template <int STRIDE = 1>
__device__ __forceinline__ void store_units4(float* __restrict__ row, int unit, int H, const f32x4& v) {
  if (STRIDE == 1 && H == MH) { *reinterpret_cast<float4*>(row + unit) = make_float4(v[0], v[1], v[2], v[3]); return; }
#pragma unroll
  for (int i = 0; i < 4; ++i) if (unit + i * STRIDE < H) row[unit + i * STRIDE] = v[i];
}

template <int DEGREE, int CT = MC>
__device__ __forceinline__ void control_slope(const Row<DEGREE, CT>& r, float frac, float width, float (&dX)[CT]) {
  const float* f = reinterpret_cast<const float*>(r.v);
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    if (DEGREE == CDE_PATH_CUBIC) dX[c] = cubic_derivative(f[c], f[CT + c], f[2 * CT + c], frac);
    else dX[c] = (f[CT + c] - f[c]) / width;
  }
}

constexpr int W16_GROUPS = 17;                      // 66 steps per tile, padded to 68 = 17 float4 groups
constexpr int W16_FLOATS = 2 * W16_GROUPS * 64 * 4;
constexpr int FWD_STAGGER_SLEEP = 44;              // x64 cycles ~ half of one RK stage; measured effect: within noise (+0.4 %)

__device__ __forceinline__ float w16_image(const float* __restrict__ W, const float* __restrict__ bias, int T, int s,
                                           int l, Dims d) {
  const int i = l & 15, kq = l >> 4;
  const int unit_out = 8 * (i >> 2) + 4 * T + (i & 3);
  if (unit_out >= d.H) return 0.f;
  if (s < 64) {
    const int m = s >> 3, c = s & 7, unit_in = 8 * kq + m;
    return (c < d.C && unit_in < d.H) ? W[(unit_out * d.C + c) * d.H + unit_in] : 0.f;
  }
  if (s < 66) { const int c = 4 * (s - 64) + kq; return c < d.C ? bias[unit_out * d.C + c] : 0.f; }
  return 0.f;
}

// d * broadcast(z.lo) / d * broadcast(z.hi) in ONE VALU instruction (VOP3P op_sel selects the source half per
// result lane).  asm volatile + the callers' sched_barriers keep them a whole MFMA group ahead of their first
// consumer, so the VALU-write -> MFMA-read wait states are satisfied by construction.
__device__ __forceinline__ f32x2 pk_mul_lo(f32x2 d, f32x2 z) {
  f32x2 r;
  asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(d), "v"(z));
  return r;
}
__device__ __forceinline__ f32x2 pk_mul_hi(f32x2 d, f32x2 z) {
  f32x2 r;
  asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(d), "v"(z));
  return r;
}

// acc += d * broadcast(z.lo | z.hi), one instruction
__device__ __forceinline__ void pk_fma_lo(f32x2& acc, f32x2 d, f32x2 z) {
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(d), "v"(z));
}
__device__ __forceinline__ void pk_fma_hi(f32x2& acc, f32x2 d, f32x2 z) {
  asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(d), "v"(z));
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}


// f = W (z (x) dX) + b dX for the 16 series of a wave on v_mfma_f32_16x16x4_f32 (132 MFMAs): lane (n, q) passes
// its 8 hidden units (za: 8q..8q+3, zb: 8q+4..8q+7) and receives f for the same units.  wA / wB are the two
// M-tile weight images (17 float4 groups each, see w16_image).
//
// Measured on gfx950 (scripts/ubench/mfma_issue.hip): a wave hides NOTHING behind its own f32 MFMA -- every
// other instruction costs ~6 cycles of matrix-pipe time.  So: products are formed two at a time (v_pk_mul_f32
// with the hidden unit broadcast by op_sel, written as asm because LLVM scalarises the vector multiply when its
// lanes are consumed one by one) and one group AHEAD of the MFMAs that consume them (no VALU->MFMA hazard nops);
// each product feeds both M-tiles.  4 pk_mul per 16 MFMAs.
__device__ __forceinline__ void field16(const float4 (&wA)[W16_GROUPS], const float4 (&wB)[W16_GROUPS], const f32x4& za,
                                        const f32x4& zb, const float (&dX)[MC], int q, f32x4& fa, f32x4& fb) {
  fa = f32x4{0.f, 0.f, 0.f, 0.f};
  fb = f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x2 d01 = {dX[0], dX[1]}, d23 = {dX[2], dX[3]}, d45 = {dX[4], dX[5]}, d67 = {dX[6], dX[7]};
  const f32x2 zp[4] = {f32x2{za[0], za[1]}, f32x2{za[2], za[3]}, f32x2{zb[0], zb[1]}, f32x2{zb[2], zb[3]}};
  f32x2 p01 = pk_mul_lo(d01, zp[0]), p23 = pk_mul_lo(d23, zp[0]), p45 = pk_mul_lo(d45, zp[0]),
        p67 = pk_mul_lo(d67, zp[0]);
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    f32x2 n01 = p01, n23 = p23, n45 = p45, n67 = p67;
    if (m < 7) {
      const f32x2 zn = zp[(m + 1) >> 1];
      if ((m + 1) & 1) { n01 = pk_mul_hi(d01, zn); n23 = pk_mul_hi(d23, zn); n45 = pk_mul_hi(d45, zn); n67 = pk_mul_hi(d67, zn); }
      else { n01 = pk_mul_lo(d01, zn); n23 = pk_mul_lo(d23, zn); n45 = pk_mul_lo(d45, zn); n67 = pk_mul_lo(d67, zn); }
    }
    __builtin_amdgcn_sched_barrier(0);
    const float* ga = reinterpret_cast<const float*>(&wA[2 * m]);       // steps 8m .. 8m+7 = groups 2m, 2m+1
    const float* gb = reinterpret_cast<const float*>(&wB[2 * m]);
    fa = mfma16(ga[0], p01[0], fa); fb = mfma16(gb[0], p01[0], fb);
    fa = mfma16(ga[1], p01[1], fa); fb = mfma16(gb[1], p01[1], fb);
    fa = mfma16(ga[2], p23[0], fa); fb = mfma16(gb[2], p23[0], fb);
    fa = mfma16(ga[3], p23[1], fa); fb = mfma16(gb[3], p23[1], fb);
    fa = mfma16(ga[4], p45[0], fa); fb = mfma16(gb[4], p45[0], fb);
    fa = mfma16(ga[5], p45[1], fa); fb = mfma16(gb[5], p45[1], fb);
    fa = mfma16(ga[6], p67[0], fa); fb = mfma16(gb[6], p67[0], fb);
    fa = mfma16(ga[7], p67[1], fa); fb = mfma16(gb[7], p67[1], fb);
    __builtin_amdgcn_sched_barrier(0);
    p01 = n01; p23 = n23; p45 = n45; p67 = n67;
  }
  // bias: step 64 feeds channel kq, step 65 channel 4 + kq
  const float b0 = q == 0 ? dX[0] : q == 1 ? dX[1] : q == 2 ? dX[2] : dX[3];
  const float b1 = q == 0 ? dX[4] : q == 1 ? dX[5] : q == 2 ? dX[6] : dX[7];
  fa = mfma16(wA[16].x, b0, fa);
  fb = mfma16(wB[16].x, b0, fb);
  fa = mfma16(wA[16].y, b1, fa);
  fb = mfma16(wB[16].y, b1, fb);
}

// The split form of field16 (dopri5 forward on small batches): the 8 waves of a workgroup evaluate the SAME 16 series; wave
// `pw` takes the K steps of the lane's pw-th hidden unit (groups 2 pw, 2 pw + 1 of both tile images: 16 of the 132
// MFMAs; wave 0 also the two bias steps) and the 8 partial sums meet in `xwin` (LDS, 8 x 64 x 9 floats), added in wave
// order so that every wave continues with the same numbers.  g0 / g1: the wave's two groups of tile A, h0 / h1 of tile
// B, ba / bb: the bias groups.
__device__ __forceinline__ void field16_split(const float4& g0, const float4& g1, const float4& h0, const float4& h1,
                                              const float4& ba, const float4& bb, const f32x4& za, const f32x4& zb,
                                              const float (&dX)[MC], int q, f32x4& fa, f32x4& fb, int pw, float* xwin,
                                              int lane) {
  fa = f32x4{0.f, 0.f, 0.f, 0.f};
  fb = f32x4{0.f, 0.f, 0.f, 0.f};
  const float zall[8] = {za[0], za[1], za[2], za[3], zb[0], zb[1], zb[2], zb[3]};
  float zm = zall[0];
#pragma unroll
  for (int m = 1; m < 8; ++m) zm = pw == m ? zall[m] : zm;         // (wave-uniform select: the lane's pw-th unit)
  const float ga[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
  const float gb[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float p = dX[c] * zm;
    fa = mfma16(ga[c], p, fa);
    fb = mfma16(gb[c], p, fb);
  }
  if (pw == 0) {                                                   // bias: step 64 feeds channel kq, step 65 channel 4 + kq
    const float b0 = q == 0 ? dX[0] : q == 1 ? dX[1] : q == 2 ? dX[2] : dX[3];
    const float b1 = q == 0 ? dX[4] : q == 1 ? dX[5] : q == 2 ? dX[6] : dX[7];
    fa = mfma16(ba.x, b0, fa);
    fb = mfma16(bb.x, b0, fb);
    fa = mfma16(ba.y, b1, fa);
    fb = mfma16(bb.y, b1, fb);
  }
  float* slot = xwin + (pw * 64 + lane) * 9;
  slot[0] = fa[0]; slot[1] = fa[1]; slot[2] = fa[2]; slot[3] = fa[3];
  slot[4] = fb[0]; slot[5] = fb[1]; slot[6] = fb[2]; slot[7] = fb[3];
  __syncthreads();
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    const float* src = xwin + (w * 64 + lane) * 9;
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] += src[i];
  }
  fa = f32x4{s[0], s[1], s[2], s[3]};
  fb = f32x4{s[4], s[5], s[6], s[7]};
  __syncthreads();
}

// stage the two 16x16x4 weight images in LDS (blockDim threads), then pull this lane's copy into registers
__device__ __forceinline__ void load_w16(const float* __restrict__ W, const float* __restrict__ bias, float* lds,
                                         float4 (&wA)[W16_GROUPS], float4 (&wB)[W16_GROUPS], Dims d) {
  for (int e = threadIdx.x; e < W16_FLOATS; e += blockDim.x) {
    const int q4 = e & 3, l = (e >> 2) & 63, g = e >> 8;            // g = T*17 + group
    const int T = g / W16_GROUPS, grp = g - T * W16_GROUPS;
    lds[e] = w16_image(W, bias, T, grp * 4 + q4, l, d);
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const float4* w4 = reinterpret_cast<const float4*>(lds);
#pragma unroll
  for (int g = 0; g < W16_GROUPS; ++g) {
    wA[g] = w4[g * 64 + lane];
    wB[g] = w4[(W16_GROUPS + g) * 64 + lane];
  }
}

// ============================================================================ fields with an activation
// f(z) = reshape_{HxC}(act(W z + b)) dX cannot use the product form above (the activation sits between the GEMM
// and the contraction with dX), so the GEMM produces the pre-activation Y = W z + b itself and each lane contracts
// its own rows with dX:
//   16x16x4 tiles, tile T = 2P + tb (16 of them), row i  <->  hidden unit h = 4P + (i>>2), channel c = 4tb + (i&3)
//   => the C/D fragment of lane (n, q): tile 2P holds Y[h = 4P+q][c = 0..3], tile 2P+1 holds c = 4..7 of series n
//   K step s (8 of them), lane quarter kq feeds input unit 4s + kq  => lane (n, q) OWNS units 4m + q (m = 0..7):
//   its state registers are the B operands as they are (no products at all), and f comes back for the same units.
// 128 MFMAs per evaluation; the bias is the accumulator's initial value (LDS image, one b128 per tile).
constexpr int WY_GROUPS = 32;                       // 16 tiles x 8 K steps = 32 float4 groups per lane
constexpr int WY_FLOATS = WY_GROUPS * 64 * 4;
constexpr int BY_FLOATS = 16 * 4 * 4;               // bias image [tile][q][r]
constexpr int ACT16_LDS_FLOATS = WY_FLOATS + BY_FLOATS;

__device__ __forceinline__ float wy16_image(const float* __restrict__ W, int T, int s, int l, Dims d) {
  const int i = l & 15, kq = l >> 4;
  const int h = 4 * (T >> 1) + (i >> 2), c = 4 * (T & 1) + (i & 3), k = 4 * s + kq;
  return (h < d.H && c < d.C && k < d.H) ? W[(h * d.C + c) * d.H + k] : 0.f;
}
// nb = channel blocks of 4 per unit group: 2 (8 channels x 32 units) or 4 (16 channels x 16 units); 16 tiles either way
__device__ __forceinline__ float by16_image(const float* __restrict__ bias, int T, int q, int r, Dims d, int nb = 2) {
  const int h = 4 * (T / nb) + q, c = 4 * (T % nb) + r;
  return (h < d.H && c < d.C) ? bias[h * d.C + c] : 0.f;
}

// tanh to ~1e-7 absolute AND relative error in a dozen VALU instructions (v_exp_f32 / v_rcp_f32 are 1 ulp):
// 1 - 2/(exp(2|x|) + 1) away from zero, the odd Taylor polynomial through x^9 below 1/4 (where the first form
// cancels).  libm's tanhf costs ~3x as much and the vector field evaluates 256 of these per series per stage.
__device__ __forceinline__ float tanh_fast(float x) {
  const float ax = __builtin_fabsf(x);
  const float e = __builtin_amdgcn_exp2f(ax * 2.885390081777927f);            // exp(2|x|)
  const float big = __builtin_fmaf(-2.f, __builtin_amdgcn_rcpf(e + 1.f), 1.f);
  const float x2 = ax * ax;
  float p = __builtin_fmaf(x2, 62.f / 2835.f, -17.f / 315.f);
  p = __builtin_fmaf(x2, p, 2.f / 15.f);
  p = __builtin_fmaf(x2, p, -1.f / 3.f);
  const float small = __builtin_fmaf(ax * x2, p, ax);
  return __builtin_copysignf(ax < 0.25f ? small : big, x);
}
template <int ACT>
__device__ __forceinline__ float activate(float y) { return ACT == CDE_ACT_TANH ? tanh_fast(y) : y; }

// two at a time: the polynomial / affine parts become v_pk_mul / v_pk_add / v_pk_fma (9 packed + 12 scalar
// instructions per pair instead of 2 x 14); same operations, same results as tanh_fast
__device__ __forceinline__ f32x2 tanh_fast2(f32x2 x) {
  const f32x2 ax = {__builtin_fabsf(x[0]), __builtin_fabsf(x[1])};
  const f32x2 arg = ax * 2.885390081777927f;
  const f32x2 e = {__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])};
  const f32x2 d = e + 1.f;
  const f32x2 r = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  const f32x2 big = __builtin_elementwise_fma(f32x2{-2.f, -2.f}, r, f32x2{1.f, 1.f});
  const f32x2 x2 = ax * ax;
  f32x2 p = __builtin_elementwise_fma(x2, f32x2{62.f / 2835.f, 62.f / 2835.f}, f32x2{-17.f / 315.f, -17.f / 315.f});
  p = __builtin_elementwise_fma(x2, p, f32x2{2.f / 15.f, 2.f / 15.f});
  p = __builtin_elementwise_fma(x2, p, f32x2{-1.f / 3.f, -1.f / 3.f});
  const f32x2 small = __builtin_elementwise_fma(ax * x2, p, ax);
  return f32x2{__builtin_copysignf(ax[0] < 0.25f ? small[0] : big[0], x[0]),
               __builtin_copysignf(ax[1] < 0.25f ? small[1] : big[1], x[1])};
}
template <int ACT>
__device__ __forceinline__ f32x2 activate2(float y0, float y1) {
  return ACT == CDE_ACT_TANH ? tanh_fast2(f32x2{y0, y1}) : f32x2{y0, y1};
}

// stage the weight and bias images in LDS (they stay there: [WY_FLOATS weight][BY_FLOATS bias])
__device__ __forceinline__ void stage_wy16(const float* __restrict__ W, const float* __restrict__ bias, float* lds,
                                           Dims d) {
  for (int e = threadIdx.x; e < WY_FLOATS; e += blockDim.x) {
    const int j = e & 3, l = (e >> 2) & 63, g = e >> 8;              // g = 2T + (s>>2)
    lds[e] = wy16_image(W, g >> 1, 4 * (g & 1) + j, l, d);
  }
  for (int e = threadIdx.x; e < BY_FLOATS; e += blockDim.x) lds[WY_FLOATS + e] = by16_image(bias, e >> 4, (e >> 2) & 3, e & 3, d);
  __syncthreads();
}

// lane (n, q): za/zb = units q, 4+q, .., 28+q of series n; returns f for the same units.
// wy = weight image + lane, by = bias image + q (both LDS).  The A operands are re-read from LDS every call
// (32 x ds_read_b128): holding them in registers (128) leaves too little for the activation's temporaries, and
// with two waves per SIMD the reads sit in the shadow of the other wave's MFMAs.
template <int ACT>
__device__ __forceinline__ void field_act16(const float4* wy, const float4* by, const f32x4& za, const f32x4& zb,
                                            const float (&dX)[MC], f32x4& fa, f32x4& fb) {
  const float zs[8] = {za[0], za[1], za[2], za[3], zb[0], zb[1], zb[2], zb[3]};
  int opaque = 0;                             // unknown to the compiler: otherwise LICM hoists the reads out of the solve
  asm volatile("" : "+v"(opaque));
  wy += opaque;
  by += opaque;
  float4 g00 = wy[0], g01 = wy[64], g10 = wy[128], g11 = wy[192];
  float4 b0 = by[0], b1 = by[4];
#pragma unroll
  for (int P = 0; P < 8; ++P) {
    f32x4 y0 = {b0.x, b0.y, b0.z, b0.w}, y1 = {b1.x, b1.y, b1.z, b1.w};
    const float a0[8] = {g00.x, g00.y, g00.z, g00.w, g01.x, g01.y, g01.z, g01.w};     // tile 2P,   K steps 0..7
    const float a1[8] = {g10.x, g10.y, g10.z, g10.w, g11.x, g11.y, g11.z, g11.w};     // tile 2P+1
    if (P < 7) {                                                                     // next pair's operands
      g00 = wy[(4 * P + 4) * 64]; g01 = wy[(4 * P + 5) * 64]; g10 = wy[(4 * P + 6) * 64]; g11 = wy[(4 * P + 7) * 64];
      b0 = by[8 * P + 8]; b1 = by[8 * P + 12];
    }
    __builtin_amdgcn_sched_barrier(0);        // one region per tile pair, the two accumulator chains alternating
#pragma unroll
    for (int s = 0; s < 8; ++s) { y0 = mfma16(a0[s], zs[s], y0); y1 = mfma16(a1[s], zs[s], y1); }
    __builtin_amdgcn_sched_barrier(0);
    const f32x2 t01 = activate2<ACT>(y0[0], y0[1]), t23 = activate2<ACT>(y0[2], y0[3]);
    const f32x2 t45 = activate2<ACT>(y1[0], y1[1]), t67 = activate2<ACT>(y1[2], y1[3]);
    float f = t01[0] * dX[0];
    f = __builtin_fmaf(t01[1], dX[1], f); f = __builtin_fmaf(t23[0], dX[2], f); f = __builtin_fmaf(t23[1], dX[3], f);
    f = __builtin_fmaf(t45[0], dX[4], f); f = __builtin_fmaf(t45[1], dX[5], f);
    f = __builtin_fmaf(t67[0], dX[6], f); f = __builtin_fmaf(t67[1], dX[7], f);
    if (P < 4) fa[P] = f; else fb[P - 4] = f;
  }
}

// The split form (dopri5 forward on small batches): the 8 waves of a workgroup evaluate the SAME 16 series, wave `pw` the
// unit group P = pw alone (16 MFMAs instead of 128), and gather the 8 values of f through `xwin` (LDS, 8 x 64 floats).
template <int ACT>
__device__ __forceinline__ void field_act16_split(const float4* wy, const float4* by, const f32x4& za, const f32x4& zb,
                                                  const float (&dX)[MC], f32x4& fa, f32x4& fb, int pw, float* xwin,
                                                  int lane) {
  const float zs[8] = {za[0], za[1], za[2], za[3], zb[0], zb[1], zb[2], zb[3]};
  int opaque = 0;
  asm volatile("" : "+v"(opaque));
  wy += opaque + 4 * pw * 64;
  by += opaque + 8 * pw;
  const float4 g00 = wy[0], g01 = wy[64], g10 = wy[128], g11 = wy[192];
  const float4 b0 = by[0], b1 = by[4];
  f32x4 y0 = {b0.x, b0.y, b0.z, b0.w}, y1 = {b1.x, b1.y, b1.z, b1.w};
  const float a0[8] = {g00.x, g00.y, g00.z, g00.w, g01.x, g01.y, g01.z, g01.w};
  const float a1[8] = {g10.x, g10.y, g10.z, g10.w, g11.x, g11.y, g11.z, g11.w};
#pragma unroll
  for (int s = 0; s < 8; ++s) { y0 = mfma16(a0[s], zs[s], y0); y1 = mfma16(a1[s], zs[s], y1); }
  const f32x2 t01 = activate2<ACT>(y0[0], y0[1]), t23 = activate2<ACT>(y0[2], y0[3]);
  const f32x2 t45 = activate2<ACT>(y1[0], y1[1]), t67 = activate2<ACT>(y1[2], y1[3]);
  float f = t01[0] * dX[0];
  f = __builtin_fmaf(t01[1], dX[1], f); f = __builtin_fmaf(t23[0], dX[2], f); f = __builtin_fmaf(t23[1], dX[3], f);
  f = __builtin_fmaf(t45[0], dX[4], f); f = __builtin_fmaf(t45[1], dX[5], f);
  f = __builtin_fmaf(t67[0], dX[6], f); f = __builtin_fmaf(t67[1], dX[7], f);
  xwin[pw * 64 + lane] = f;
  __syncthreads();
#pragma unroll
  for (int P = 0; P < 8; ++P) { const float v = xwin[P * 64 + lane]; if (P < 4) fa[P] = v; else fb[P - 4] = v; }
  __syncthreads();
}

// ============================================================================ two-layer fields
// f(z) = reshape_{HxC}(act(W2 relu(W1 z + b1) + b2)) dX   (reference example/time_series_classification.py:20-51:
// Linear(H, width) -> relu -> Linear(width, H*C) -> tanh, width = 128).  Same tiling as field_act16 with a hidden
// layer in front: layer 1 is 8 tiles (width padded to 128) x 8 K steps whose C/D fragment -- lane (n, q) holds
// hidden-layer units 16*T1 + 4q + r -- is exactly the B operand of layer 2's K step (T1, r), so the hidden layer
// never leaves the registers.  64 + 512 MFMAs per evaluation; both weight images live in LDS (16 + 128 KB).
constexpr int MW = 128;                              // hidden-layer width the tiles are built for
constexpr int W1M_FLOATS = 8 * 2 * 64 * 4;           // layer 1: 8 tiles x 2 groups of 4 K steps
constexpr int B1M_FLOATS = 8 * 4 * 4;                // [tile][q][r]
constexpr int W2M_FLOATS = 16 * 8 * 64 * 4;          // layer 2: 16 tiles x 8 groups of 4 K steps
constexpr int MLP16_LDS_FLOATS = W1M_FLOATS + B1M_FLOATS + W2M_FLOATS + BY_FLOATS;
struct MlpDims { int H, C, width; };
bool mlp_shape_ok(int64_t C, int64_t H, int64_t width);     // rk4_mfma.hip
// 32 hidden units x 16 channels (round 6; config 5 at hidden size 32: 14 logsignature channels): twice the 16 tiles.  The LDS
// images hold unit groups 0..3 as before; groups 4..7 (hidden units 16..31) are read straight from the caller's output-layer
// tensors in global memory -- a lane's A operand of (row (h, c), hidden-layer columns 16 T1 + 4 kq .. + 3) is four consecutive
// floats of W2's row (h, c), so no second image is needed (width a multiple of 4, the tensor 16-byte aligned).
bool mlp_shape_hi(int64_t C, int64_t H, int64_t width);     // rk4_mfma.hip
bool mlp_shape_upper(int64_t C, int64_t H, int64_t width);  // rk4_mfma.hip: the same shape for the sweeps (padded copy: any width)
struct MlpHi { const float* W2; const float* b2; int H, C, width; int h0 = 0; const float* W2t = nullptr; };   // W2 == nullptr: no upper half; the rows of
                                                                                   // hidden unit h start at W2 + (h - h0) C width
                                                                                   // (h0 = 16: a copy of the upper rows only)

// value of the combined image [layer-1 weights | layer-1 bias | layer-2 weights | layer-2 bias] at flat index e
__device__ __forceinline__ float mlp16_image(const float* __restrict__ W1, const float* __restrict__ b1,
                                             const float* __restrict__ W2, const float* __restrict__ b2, int e,
                                             MlpDims d, int nb = 2) {
  if (e < W1M_FLOATS) {
    const int j = e & 3, l = (e >> 2) & 63, g = e >> 8;              // g = 2*T1 + (s>>2)
    const int row = 16 * (g >> 1) + (l & 15), k = 4 * (4 * (g & 1) + j) + (l >> 4);
    return (row < d.width && k < d.H) ? W1[row * d.H + k] : 0.f;
  }
  e -= W1M_FLOATS;
  if (e < B1M_FLOATS) {
    const int unit = 16 * (e >> 4) + 4 * ((e >> 2) & 3) + (e & 3);
    return unit < d.width ? b1[unit] : 0.f;
  }
  e -= B1M_FLOATS;
  if (e < W2M_FLOATS) {
    const int j = e & 3, l = (e >> 2) & 63, g = e >> 8;              // g = 8*T2 + T1, K step (T1, r = j)
    const int T2 = g >> 3, T1 = g & 7, i = l & 15, kq = l >> 4;
    const int h = 4 * (T2 / nb) + (i >> 2), c = 4 * (T2 % nb) + (i & 3), col = 16 * T1 + 4 * kq + j;
    return (h < d.H && c < d.C && col < d.width) ? W2[(h * d.C + c) * d.width + col] : 0.f;
  }
  e -= W2M_FLOATS;
  return by16_image(b2, e >> 4, (e >> 2) & 3, e & 3, Dims{d.H, d.C}, nb);
}

__device__ __forceinline__ void stage_mlp16(const float* __restrict__ W1, const float* __restrict__ b1,
                                            const float* __restrict__ W2, const float* __restrict__ b2, float* lds,
                                            MlpDims d, int nb = 2) {
  for (int e = threadIdx.x; e < MLP16_LDS_FLOATS; e += blockDim.x) lds[e] = mlp16_image(W1, b1, W2, b2, e, d, nb);
  __syncthreads();
}

// img = staged images + nothing else; lane (n, q): za/zb = units q, 4+q, .., 28+q of series n.
// CT = 8: 8 unit groups P of 2 tiles (4 units x 8 channels each); CT = 16: 4 unit groups of 4 tiles (4 units x 16
// channels; units 16.. do not exist: zb is zero in, fb zero out).  Tile index T = (CT/4) P + tb in both cases.
// SPLIT (small batches, dopri5 forward): the 8 waves of a workgroup evaluate the SAME 16 series -- layer 1 redundantly and
// bit-identically, unit group P of layer 2 by wave P alone -- and gather the NP values of f through `xwin` (LDS, 8 x 64
// floats): 128 instead of 576 MFMAs per wave and evaluation.
// HI (round 6): 16-channel layout with hidden units 16..31 as unit groups 4..7, read from the raw output layer (`hi`).  A
// template flag: as a run-time one it cost the kernels of the plain 16-channel shape 40 % (K4, config 5 at hidden size 8:
// 144 -> 201 us per attempted step -- twice the unrolled unit groups, their global loads hoisted across the stages).
template <int ACT, int CT = MC, bool SPLIT = false, bool HI = false>
__device__ __forceinline__ void field_mlp16(const float* img, int lane, int q, const f32x4& za, const f32x4& zb,
                                            const float (&dX)[CT], f32x4& fa, f32x4& fb, int pw = 0, float* xwin = nullptr,
                                            float* xu = nullptr, MlpHi hi = MlpHi{}) {
  constexpr int NB = CT / 4, NP = 16 / NB;
  static_assert(!HI || CT == 16, "the upper half exists on the 16-channel layout");
  constexpr int NPX = HI ? 8 : NP;                // unit groups 4..7: from the raw tensors `hi` names
  constexpr bool has_hi = HI;
  if constexpr (SPLIT && CT == MC) {
    // Eight waves, 8-channel tiles (round 4): layer 1 is split as well -- wave pw computes hidden-layer tile pw (8 MFMAs
    // instead of 64 redundant ones), the 8 x 16 units meet in `xu` (8 KB of LDS, one barrier) and are read from there as
    // layer 2's B operands; layer 2 of unit group pw as before.  72 instead of 128 MFMAs per wave and evaluation.
    const float4* w1 = reinterpret_cast<const float4*>(img) + lane;
    const float4* bb1 = reinterpret_cast<const float4*>(img + W1M_FLOATS) + q;
    const float4* w2 = reinterpret_cast<const float4*>(img + W1M_FLOATS + B1M_FLOATS) + lane;
    const float4* bb2 = reinterpret_cast<const float4*>(img + W1M_FLOATS + B1M_FLOATS + W2M_FLOATS) + q;
    const float zs[8] = {za[0], za[1], za[2], za[3], zb[0], zb[1], zb[2], zb[3]};
    {
      const float4 c0 = bb1[4 * pw];
      f32x4 y0 = {c0.x, c0.y, c0.z, c0.w};
      const float4 g0 = w1[(2 * pw) * 64], g1 = w1[(2 * pw + 1) * 64];
      const float a0[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
      for (int s = 0; s < 8; ++s) y0 = mfma16(a0[s], zs[s], y0);      // one chain, bias first: the bits of the unsplit form
      *reinterpret_cast<float4*>(xu + (pw * 64 + lane) * 4) =
          make_float4(fmaxf(y0[0], 0.f), fmaxf(y0[1], 0.f), fmaxf(y0[2], 0.f), fmaxf(y0[3], 0.f));
    }
    __syncthreads();
    f32x4 y[2];
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
      const float4 c0 = bb2[4 * (2 * pw + tb)];
      y[tb] = f32x4{c0.x, c0.y, c0.z, c0.w};
    }
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const float4 u4 = *reinterpret_cast<const float4*>(xu + (g * 64 + lane) * 4);
      const float4 a0 = w2[(8 * (2 * pw) + g) * 64], a1 = w2[(8 * (2 * pw + 1) + g) * 64];
      y[0] = mfma16(a0.x, u4.x, y[0]); y[1] = mfma16(a1.x, u4.x, y[1]);
      y[0] = mfma16(a0.y, u4.y, y[0]); y[1] = mfma16(a1.y, u4.y, y[1]);
      y[0] = mfma16(a0.z, u4.z, y[0]); y[1] = mfma16(a1.z, u4.z, y[1]);
      y[0] = mfma16(a0.w, u4.w, y[0]); y[1] = mfma16(a1.w, u4.w, y[1]);
    }
    float f = 0.f;
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) {
      const f32x2 t01 = activate2<ACT>(y[tb][0], y[tb][1]), t23 = activate2<ACT>(y[tb][2], y[tb][3]);
      f = (tb == 0) ? t01[0] * dX[0] : __builtin_fmaf(t01[0], dX[4 * tb], f);
      f = __builtin_fmaf(t01[1], dX[4 * tb + 1], f);
      f = __builtin_fmaf(t23[0], dX[4 * tb + 2], f);
      f = __builtin_fmaf(t23[1], dX[4 * tb + 3], f);
    }
    xwin[pw * 64 + lane] = f;
    __syncthreads();                                           // (also: every wave is done reading `xu`)
    float fs[8];
#pragma unroll
    for (int P = 0; P < 8; ++P) fs[P] = xwin[P * 64 + lane];
    fa = f32x4{fs[0], fs[1], fs[2], fs[3]};
    fb = f32x4{fs[4], fs[5], fs[6], fs[7]};
    __syncthreads();
    return;
  }
  int opaque = 0;                             // as in field_act16: keeps the LDS reads inside the call
  asm volatile("" : "+v"(opaque));
  const float4* w1 = reinterpret_cast<const float4*>(img) + lane + opaque;
  const float4* bb1 = reinterpret_cast<const float4*>(img + W1M_FLOATS) + q + opaque;
  const float4* w2 = reinterpret_cast<const float4*>(img + W1M_FLOATS + B1M_FLOATS) + lane + opaque;
  const float4* bb2 = reinterpret_cast<const float4*>(img + W1M_FLOATS + B1M_FLOATS + W2M_FLOATS) + q + opaque;
  const float zs[8] = {za[0], za[1], za[2], za[3], zb[0], zb[1], zb[2], zb[3]};
  float u[32];
  // ---- layer 1, two tiles at a time (two independent accumulator chains)
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
    for (int r = 0; r < 4; ++r) { u[8 * TP + r] = fmaxf(y0[r], 0.f); u[8 * TP + 4 + r] = fmaxf(y1[r], 0.f); }
  }
  // ---- layer 2 + activation + contraction, one unit group (4 hidden units x CT channels = NB tiles) at a time
  fa = f32x4{0.f, 0.f, 0.f, 0.f};
  fb = fa;
  // (the upper groups as a compile-time flag of the group body: LDS reads and global reads stay apart; outside the SPLIT form
  //  they run as a ROLLED loop -- four more unrolled groups per evaluation in a kernel that unrolls six evaluations spilled 500
  //  registers)
  auto group = [&](int P, auto upper_c) {
    constexpr bool upper = decltype(upper_c)::value;
    f32x4 y[NB];
    const float* hrow[NB];                                   // upper half: this lane's row (h, c) of W2, at its column 4 kq
    bool hrow_ok[NB];
#pragma unroll
    for (int tb = 0; tb < NB; ++tb) {
      if constexpr (!upper) {
        const float4 c0 = bb2[4 * (NB * P + tb)];
        y[tb] = f32x4{c0.x, c0.y, c0.z, c0.w};
        hrow[tb] = nullptr; hrow_ok[tb] = false;
      } else {
        const int hb = 4 * P + q;                            // bias of the lane's own D rows: (h = 4P + q, c = 4 tb + r)
#pragma unroll
        for (int r = 0; r < 4; ++r) y[tb][r] = (hb < hi.H && 4 * tb + r < hi.C) ? hi.b2[(hb - hi.h0) * hi.C + 4 * tb + r] : 0.f;
        const int i = lane & 15, h = 4 * P + (i >> 2), c = 4 * tb + (i & 3);
        hrow_ok[tb] = h < hi.H && c < hi.C;
        hrow[tb] = hi.W2 + (int64_t)(hrow_ok[tb] ? (h - hi.h0) * hi.C + c : 0) * hi.width + 4 * (lane >> 4);
      }
    }
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      float4 a[NB];
#pragma unroll
      for (int tb = 0; tb < NB; ++tb) {
        if constexpr (!upper) a[tb] = w2[(8 * (NB * P + tb) + g) * 64];            // tile NB*P + tb: groups 8T .. 8T+7
        else a[tb] = (hrow_ok[tb] && 16 * g + 4 * (lane >> 4) < hi.width) ? *reinterpret_cast<const float4*>(hrow[tb] + 16 * g)
                                                                          : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      // the NB accumulator chains alternate, so no MFMA waits for its own predecessor
#pragma unroll
      for (int tb = 0; tb < NB; ++tb) y[tb] = mfma16(a[tb].x, u[4 * g], y[tb]);
#pragma unroll
      for (int tb = 0; tb < NB; ++tb) y[tb] = mfma16(a[tb].y, u[4 * g + 1], y[tb]);
#pragma unroll
      for (int tb = 0; tb < NB; ++tb) y[tb] = mfma16(a[tb].z, u[4 * g + 2], y[tb]);
#pragma unroll
      for (int tb = 0; tb < NB; ++tb) y[tb] = mfma16(a[tb].w, u[4 * g + 3], y[tb]);
    }
    float f = 0.f;
#pragma unroll
    for (int tb = 0; tb < NB; ++tb) {
      const f32x2 t01 = activate2<ACT>(y[tb][0], y[tb][1]), t23 = activate2<ACT>(y[tb][2], y[tb][3]);
      f = (tb == 0) ? t01[0] * dX[0] : __builtin_fmaf(t01[0], dX[4 * tb], f);
      f = __builtin_fmaf(t01[1], dX[4 * tb + 1], f);
      f = __builtin_fmaf(t23[0], dX[4 * tb + 2], f);
      f = __builtin_fmaf(t23[1], dX[4 * tb + 3], f);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { fa[kk] = P == kk ? f : fa[kk]; fb[kk] = P == 4 + kk ? f : fb[kk]; }
    __builtin_amdgcn_sched_barrier(0);
  };
  if constexpr (SPLIT) {
#pragma unroll
    for (int P = 0; P < NP; ++P) {
      if (P != pw) continue;                                 // (wave-uniform: another wave's unit group)
      group(P, std::false_type{});
    }
    if constexpr (HI) {
#pragma unroll
      for (int P = NP; P < NPX; ++P) {
        if (P != pw) continue;
        group(P, std::true_type{});
      }
    }
  } else {
#pragma unroll
    for (int P = 0; P < NP; ++P) group(P, std::false_type{});
    if constexpr (HI) {
#pragma clang loop unroll(disable)
      for (int P = NP; P < NPX; ++P) group(P, std::true_type{});
    }
  }
  if constexpr (SPLIT) {
    // wave P holds f of unit group P (waves beyond the last group hold nothing): everybody collects all of them
    float mine = 0.f;
#pragma unroll
    for (int P = 0; P < NPX; ++P) if (P == pw) mine = P < 4 ? fa[P] : fb[P - 4];
    xwin[pw * 64 + lane] = mine;
    __syncthreads();
#pragma unroll
    for (int P = 0; P < NPX; ++P) {
      if (P >= NP && !has_hi) continue;
      const float v = xwin[P * 64 + lane];
      if (P < 4) fa[P] = v; else fb[P - 4] = v;
    }
    __syncthreads();
  }
}

}  // namespace cde
