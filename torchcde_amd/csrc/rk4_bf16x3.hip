// rk4_bf16x3.hip -- K2b / K3b: the headline solve (f32 state, H <= 32, C <= 8, affine field) with its weight GEMMs on the
// BF16 matrix pipe at float32 accuracy.  Opt-in: `variant = CDE_VARIANT_BF16X3` (cdeint(..., variant="bf16x3")).
//
// Idea ("bf16x3").  gfx950 has no xf32 / TF32 mode; its exact-f32 MFMA runs at the vector rate, 1/16 of the bf16 rate
// (MI355X_MICROARCH.md).  A float32 value splits exactly into three bf16 pieces x = x1 + x2 + x3 (8 + 8 + 8 mantissa
// bits), and a product a b is recovered to ~2^-24 relative from the six piece products a_i b_j with i + j <= 4, each a
// v_mfma_f32_32x32x16_bf16 accumulating in float32 (smallest terms first).  Six bf16 MFMAs of 32 cycles replace eight
// f32 MFMAs of 64-66 cycles per 32 x 32 x 16 block: 2.7x less matrix-pipe time.  scripts/ubench/bf16x3_gemm.hip measured
// it in isolation (profiles/r03_bf16x3_ubench.txt): 3381 against 8393 cycles per evaluation INCLUDING the operand split,
// error 1.45e-7 of max|Y| against the f32 MFMA's 1.94e-7.
//
// What makes the split cheap here is the PRE-ACTIVATION form of the field (the one the tanh kernels use):
//     f_h = sum_c (W z + b)_(h,c) dX_c          the GEMM's B operand is the state z itself: 16 values per lane and stage,
// not the 264 products z_m dX_c of K2 / K3's product form.  Likewise for the adjoint
//     (a^T df/dz)_k = sum_c dX_c (W_c^T a)_k    eight 32 x 32 blocks W_c, B operand = a: 16 values per lane and stage.
// The weight pieces are split ONCE per launch into LDS images.  The third GEMM of the adjoint, dL/dW += (a (x) dX)^T z,
// stays on the exact-f32 pipe as in K3: its 256-row operand changes every stage and splitting it (128 values per lane)
// costs what the bf16 MFMAs save.
//
// Ownership = K3's: one wave owns 32 series for the whole solve, lane (n = l & 31, half = l >> 5) keeps hidden units
// 2 r + half (r = 0..15) in registers.  Tilings are chosen so that nothing ever moves between lanes:
//   MFMA K index kappa = 16 ks + 8 half + e  <->  unit 2 (8 ks + e) + half : the lane's own register 8 ks + e
//   Y tile t (4 units x 8 channels), row rho = 8 g + 4 hf + e  <->  unit 4 t + 2 (g >> 1) + hf, channel 4 (g & 1) + e :
//       D register r of lane (n, half) = Y[unit 4 t + 2 (r >> 3) + half][channel r & 7] -- both of the lane's units of the
//       tile with all their channels: f_(2(2t)+half), f_(2(2t+1)+half) are two in-lane dot products with dX
//   W_c^T tile (channel c), row rho = 8 g + 4 hf + e  <->  output unit 2 (4 g + e) + hf : D register r = output unit 2 r + half
#include "cde_mfma.h"

namespace cde {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

constexpr int BX_IMG_U4 = 8 * 2 * 3 * 64;                 // uint4 entries of one piece image: [tile 8][ks 2][piece 3][lane 64]
constexpr int BX_BIAS_FLOATS = 8 * 2 * 16;                // [tile][half][register]
constexpr int BX_FWD_LDS_BYTES = BX_IMG_U4 * 16 + BX_BIAS_FLOATS * 4;
constexpr int BX_ADJ_LDS_BYTES = 2 * BX_IMG_U4 * 16 + BX_BIAS_FLOATS * 4 + 4 * SCR_FLOATS * 4;

__device__ __forceinline__ void bx_wave_lds_sync() {           // rk4_mfma.hip: wave_lds_sync
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ void bx_split3(float x, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)x;
  const float r1 = x - (float)a;
  b = (__bf16)r1;
  const float r2 = r1 - (float)b;
  c = (__bf16)r2;
}

// the three pieces of 8 consecutive registers as MFMA B operands
__device__ __forceinline__ void bx_split8(const f32x16& v, int base, bf16x8 (&p)[3]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    __bf16 a, b, c;
    bx_split3(v[base + e], a, b, c);
    p[0][e] = a; p[1][e] = b; p[2][e] = c;
  }
}

// images into LDS: `which` 0 = Y tiles (rows (unit, channel), K = input unit), 1 = W_c^T tiles (rows = output unit, K = unit)
__device__ __forceinline__ void bx_stage_image(const float* __restrict__ W, u32x4* img, int which, Dims d, int tid, int nthreads) {
  for (int e4 = tid; e4 < BX_IMG_U4; e4 += nthreads) {
    const int l = e4 & 63, piece = (e4 >> 6) % 3, tk = (e4 >> 6) / 3, ks = tk & 1, t = tk >> 1;
    const int rho = l & 31, hfA = l >> 5;
    const int g = rho >> 3, hf = (rho >> 2) & 1, ee = rho & 3;
    bf16x8 out;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int uin = 2 * (8 * ks + e) + hfA;                     // the unit K index kappa = 16 ks + 8 hfA + e stands for
      float w;
      if (which == 0) {
        const int uo = 4 * t + 2 * (g >> 1) + hf, c = 4 * (g & 1) + ee;
        w = (uo < d.H && c < d.C && uin < d.H) ? W[(uo * d.C + c) * d.H + uin] : 0.f;
      } else {
        const int ko = 2 * (4 * g + ee) + hf, c = t;              // tile index = channel
        w = (uin < d.H && c < d.C && ko < d.H) ? W[(uin * d.C + c) * d.H + ko] : 0.f;
      }
      __bf16 a, b, c3;
      bx_split3(w, a, b, c3);
      out[e] = piece == 0 ? a : piece == 1 ? b : c3;
    }
    img[e4] = __builtin_bit_cast(u32x4, out);
  }
}

__device__ __forceinline__ void bx_stage_bias(const float* __restrict__ bias, float* tab, Dims d, int tid, int nthreads) {
  for (int e = tid; e < BX_BIAS_FLOATS; e += nthreads) {
    const int r = e & 15, half = (e >> 4) & 1, t = e >> 5;
    const int u = 4 * t + 2 * (r >> 3) + half, c = r & 7;
    tab[e] = (u < d.H && c < d.C) ? bias[u * d.C + c] : 0.f;
  }
}

// six piece products of one 32 x 32 x 16 block for TWO independent accumulators (two tiles), interleaved so that no MFMA
// waits on its own accumulator; smallest terms first: a3 b1, a2 b2, a1 b3, a2 b1, a1 b2, a1 b1
__device__ __forceinline__ void bx_block2(const u32x4* a, const u32x4* a_other, const bf16x8 (&b)[3], f32x16& acc,
                                          f32x16& acc_other) {
  const bf16x8 a1 = __builtin_bit_cast(bf16x8, a[0]), a2 = __builtin_bit_cast(bf16x8, a[64]), a3 = __builtin_bit_cast(bf16x8, a[128]);
  const bf16x8 o1 = __builtin_bit_cast(bf16x8, a_other[0]), o2 = __builtin_bit_cast(bf16x8, a_other[64]),
               o3 = __builtin_bit_cast(bf16x8, a_other[128]);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b[0], acc, 0, 0, 0);
  acc_other = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o3, b[0], acc_other, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b[1], acc, 0, 0, 0);
  acc_other = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o2, b[1], acc_other, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b[2], acc, 0, 0, 0);
  acc_other = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o1, b[2], acc_other, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b[0], acc, 0, 0, 0);
  acc_other = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o2, b[0], acc_other, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b[1], acc, 0, 0, 0);
  acc_other = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o1, b[1], acc_other, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b[0], acc, 0, 0, 0);
  acc_other = __builtin_amdgcn_mfma_f32_32x32x16_bf16(o1, b[0], acc_other, 0, 0, 0);
}

// f (register r <-> unit 2 r + half) of the affine field at state z
__device__ __forceinline__ f32x16 bx_field(const u32x4* imgY, const float* btab, int lane, int half, const f32x16& z,
                                           const float (&dX)[MC]) {
  // the images are loop invariant: without this the compiler hoists all 48 LDS reads (192 registers) out of the time loop
  int opaque = 0;
  asm volatile("" : "+v"(opaque));
  imgY += opaque;
  bf16x8 zp0[3], zp1[3];
  bx_split8(z, 0, zp0);
  bx_split8(z, 8, zp1);
  f32x16 f;
  const float4* b4 = reinterpret_cast<const float4*>(btab) + half * 4 + opaque;
#pragma unroll
  for (int tp = 0; tp < 4; ++tp) {                                // two tiles at a time: 32 accumulator registers live
    f32x16 acc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const float4 bb = b4[(2 * tp + u) * 8 + q4];
        acc[u][4 * q4] = bb.x; acc[u][4 * q4 + 1] = bb.y; acc[u][4 * q4 + 2] = bb.z; acc[u][4 * q4 + 3] = bb.w;
      }
    const u32x4* a0 = imgY + (((2 * tp) * 2) * 3) * 64 + lane;    // tile 2 tp, K step 0; K step 1 is 3 * 64 further on
    const u32x4* a1 = imgY + (((2 * tp + 1) * 2) * 3) * 64 + lane;
    bx_block2(a0, a1, zp0, acc[0], acc[1]);
    bx_block2(a0 + 3 * 64, a1 + 3 * 64, zp1, acc[0], acc[1]);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      float s0 = acc[u][0] * dX[0], s1 = acc[u][8] * dX[0];
#pragma unroll
      for (int c = 1; c < MC; ++c) { s0 = __builtin_fmaf(acc[u][c], dX[c], s0); s1 = __builtin_fmaf(acc[u][8 + c], dX[c], s1); }
      f[2 * (2 * tp + u)] = s0; f[2 * (2 * tp + u) + 1] = s1;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  return f;
}

// a^T df/dz (register r <-> unit 2 r + half)
__device__ __forceinline__ f32x16 bx_vjp(const u32x4* imgV, int lane, const f32x16& a, const float (&dX)[MC]) {
  int opaque = 0;
  asm volatile("" : "+v"(opaque));
  imgV += opaque;
  bf16x8 ap0[3], ap1[3];
  bx_split8(a, 0, ap0);
  bx_split8(a, 8, ap1);
  f32x16 va;
#pragma unroll
  for (int r = 0; r < 16; ++r) va[r] = 0.f;
#pragma unroll
  for (int cp = 0; cp < MC / 2; ++cp) {                           // two channels at a time
    f32x16 acc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    const u32x4* a0 = imgV + (((2 * cp) * 2) * 3) * 64 + lane;
    const u32x4* a1 = imgV + (((2 * cp + 1) * 2) * 3) * 64 + lane;
    bx_block2(a0, a1, ap0, acc[0], acc[1]);
    bx_block2(a0 + 3 * 64, a1 + 3 * 64, ap1, acc[0], acc[1]);
#pragma unroll
    for (int r = 0; r < 16; ++r) va[r] = __builtin_fmaf(acc[1][r], dX[2 * cp + 1], __builtin_fmaf(acc[0][r], dX[2 * cp], va[r]));
    __builtin_amdgcn_sched_barrier(0);
  }
  return va;
}

// ============================================================================================ forward (K2b)
template <typename TT, int DEGREE>
__global__ __launch_bounds__(256, 1) void rk4_forward_bf16x3(
    const float* __restrict__ coeffs, const float* __restrict__ knots, int64_t n_intervals,
    const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ z0,
    const TT* __restrict__ grid, int64_t n_grid, const TT* __restrict__ t_out, int64_t n_out,
    float* __restrict__ z_out, int64_t B, const int64_t* __restrict__ stage_index,
    const float* __restrict__ stage_frac, Dims dims) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  u32x4* imgY = reinterpret_cast<u32x4*>(lds_raw);
  float* btab = reinterpret_cast<float*>(lds_raw + BX_IMG_U4 * 16);
  bx_stage_image(W, imgY, 0, dims, threadIdx.x, 256);
  bx_stage_bias(bias, btab, dims, threadIdx.x, 256);
  __syncthreads();
  const int Hr = dims.H, Cr = dims.C;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, half = lane >> 5;
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  if (tile * 32 >= B) return;
  const int64_t series = tile * 32 + n;
  const bool valid = series < B;
  const int64_t sc = valid ? series : B - 1;
  f32x16 y;
#pragma unroll
  for (int r = 0; r < 16; ++r) y[r] = 2 * r + half < Hr ? z0[sc * Hr + 2 * r + half] : 0.f;
  auto store = [&](int64_t j, const f32x16& v) {
    if (valid) {
      float* row = z_out + (series * n_out + j) * Hr;
#pragma unroll
      for (int r = 0; r < 16; ++r) if (2 * r + half < Hr) row[2 * r + half] = v[r];
    }
  };
  store(0, y);
  int64_t jout = 1;
  const int64_t n_steps = n_grid - 1;
  if (n_steps <= 0) return;
  int64_t idx = stage_index[0];
  float frac = stage_frac[0];
  Row<DEGREE> row = load_row<DEGREE>(coeffs, sc, n_intervals, idx, Cr);
  for (int64_t k = 0; k < n_steps; ++k) {
    const TT t0 = grid[k], t1 = grid[k + 1];
    const float dt = (float)(t1 - t0);
    f32x16 k1, k2, pq, z = y;
#pragma unroll
    for (int stage = 0; stage < 4; ++stage) {
      float dX[MC];
      const float width = DEGREE == CDE_PATH_LINEAR ? knots[idx + 1] - knots[idx] : 1.f;
      control_slope<DEGREE>(row, frac, width, dX);
      const int64_t e_next = 4 * k + stage + 1;
      const bool more = e_next < 4 * n_steps;
      const int64_t nidx = more ? stage_index[e_next] : idx;
      const float nfrac = more ? stage_frac[e_next] : frac;
      if (nidx != idx) row = load_row<DEGREE>(coeffs, sc, n_intervals, nidx, Cr);
      const f32x16 f = bx_field(imgY, btab, lane, half, z, dX);
      // torchdiffeq rk4_alt_step_func (3/8 rule), association order preserved
      const float third = (float)(1.0 / 3.0);
      if (stage == 0) { k1 = f; z = y + dt * k1 * third; }
      else if (stage == 1) { k2 = f; z = y + dt * (k2 - k1 * third); }
      else if (stage == 2) { z = y + dt * (k1 - k2 + f); pq = k1 + 3.f * (k2 + f); }
      else z = y + (pq + f) * dt * 0.125f;
      idx = nidx; frac = nfrac;
    }
    const f32x16 y1 = z;
    while (jout < n_out && t1 >= t_out[jout]) {
      const TT tj = t_out[jout];
      if (tj == t0) store(jout, y);
      else if (tj == t1) store(jout, y1);
      else {
        const float slope = (float)((tj - t0) / (t1 - t0));
        store(jout, y + slope * (y1 - y));
      }
      ++jout;
    }
    y = y1;
  }
}

// ============================================================================================ adjoint (K3b)
// K3 (rk4_mfma.hip: rk4_adjoint_mfma) with its two weight GEMMs on the bf16 pipe; the dL/dW product, the scratch
// transposes, the RK bookkeeping and the per-wave partial layout are K3's.
constexpr int64_t BX_PARTIAL_FLOATS = MH * MC * MH + MH * MC;

template <typename TT, int DEGREE>
__global__ __launch_bounds__(256, 1) void rk4_adjoint_bf16x3(
    const float* __restrict__ coeffs, const float* __restrict__ knots, int64_t n_intervals,
    const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ z_saved,
    const float* __restrict__ grad_out, const TT* __restrict__ sgrid, const int64_t* __restrict__ seg_off,
    int64_t n_out, float* __restrict__ grad_z0, float* __restrict__ partial, int64_t B,
    const int64_t* __restrict__ stage_index, const float* __restrict__ stage_frac, Dims dims) {
  const int Hr = dims.H, Cr = dims.C;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  u32x4* imgY = reinterpret_cast<u32x4*>(lds_raw);
  u32x4* imgV = imgY + BX_IMG_U4;
  float* btab = reinterpret_cast<float*>(lds_raw + 2 * BX_IMG_U4 * 16);
  float* scr_base = btab + BX_BIAS_FLOATS;
  bx_stage_image(W, imgY, 0, dims, threadIdx.x, 256);
  bx_stage_image(W, imgV, 1, dims, threadIdx.x, 256);
  bx_stage_bias(bias, btab, dims, threadIdx.x, 256);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 31, half = lane >> 5;
  float* scr_y = scr_base + wave * SCR_FLOATS;
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  float* my_partial = partial + tile * BX_PARTIAL_FLOATS;
  if (tile * 32 >= B) return;
  const int64_t series = tile * 32 + n;
  const bool valid = series < B;
  const int64_t sc = valid ? series : B - 1;

  f32x16 accW[MC];
  f32x2 gbp[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
#pragma unroll
  for (int c = 0; c < MC; ++c) {
#pragma unroll
    for (int r = 0; r < 16; ++r) accW[c][r] = 0.f;
  }
  f32x16 y0, a0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int u = 2 * r + half;
    const bool on = u < Hr;
    y0[r] = on ? z_saved[(sc * n_out + (n_out - 1)) * Hr + u] : 0.f;
    a0[r] = (valid && on) ? grad_out[(sc * n_out + (n_out - 1)) * Hr + u] : 0.f;
  }
  float* scr_zt = scr_y;                    // 64 rows x 20   (see rk4_adjoint_mfma)
  float* scr_at = scr_y + 64 * 20;
  float* scr_dw = scr_y + 2 * 64 * 20;      // 32 x 8

  for (int64_t p = 0; p + 1 < n_out; ++p) {
    const int64_t i_out = n_out - 1 - p;
    const int64_t k_begin = seg_off[p], k_end = seg_off[p + 1] - 1;
    if (k_end > k_begin) {
      int64_t idx = stage_index[4 * k_begin];
      float frac = stage_frac[4 * k_begin];
      Row<DEGREE> row = load_row<DEGREE>(coeffs, sc, n_intervals, idx, Cr);
      for (int64_t k = k_begin; k < k_end; ++k) {
        const float ds = (float)(sgrid[k + 1] - sgrid[k]);
        f32x16 ky1, ky2, ka1, ka2, yst = y0, ast = a0;
        // (NOT unrolled: with four stage bodies in one block the register allocator spilled 700 dwords; one body: 74)
#pragma unroll 1
        for (int stage = 0; stage < 4; ++stage) {
          float dX[MC];
          const float width = DEGREE == CDE_PATH_LINEAR ? knots[idx + 1] - knots[idx] : 1.f;
          control_slope<DEGREE>(row, frac, width, dX);
          const int64_t e_next = 4 * k + stage + 1;
          const bool more = e_next < 4 * k_end;
          const int64_t nidx = more ? stage_index[e_next] : idx;
          const float nfrac = more ? stage_frac[e_next] : frac;
          if (nidx != idx) row = load_row<DEGREE>(coeffs, sc, n_intervals, nidx, Cr);

          const f32x2 d01 = {dX[0], dX[1]}, d23 = {dX[2], dX[3]}, d45 = {dX[4], dX[5]}, d67 = {dX[6], dX[7]};
          // ---- stage state -> scratch (transposed), weighted control derivative (for the dL/dW product)
          {
            const float wq = ((stage == 0 || stage == 3) ? 0.125f : 0.375f) * ds;
            float* wz = scr_zt + ((n & 1) * 32 + half) * 20 + (n >> 1);
            float* wa = scr_at + ((n & 1) * 32 + half) * 20 + (n >> 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) { wz[r * 40] = yst[r]; wa[r * 40] = ast[r]; }
            const f32x2 w0 = (half ? d45 : d01) * wq, w1 = (half ? d67 : d23) * wq;
            *reinterpret_cast<float4*>(scr_dw + n * 8 + 4 * half) = make_float4(w0[0], w0[1], w1[0], w1[1]);
            bx_wave_lds_sync();
          }
          // ---- f and a^T df/dz on the bf16 pipe
          const f32x16 f = bx_field(imgY, btab, lane, half, yst, dX);
          const f32x16 va = bx_vjp(imgV, lane, ast, dX);
          // ---- dL/dW tile c: D[h][k] += sum_series (w ds a_h dX_c)[series] * z_k[series] on the exact-f32 pipe (K3's block)
          {
            const float4* zt4 = reinterpret_cast<const float4*>(scr_zt + (half * 32 + n) * 20);
            const float4* at4 = reinterpret_cast<const float4*>(scr_at + (half * 32 + n) * 20);
            const float4* dw4 = reinterpret_cast<const float4*>(scr_dw + half * 8);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const float4 zq = zt4[g4], aq = at4[g4];
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
          bx_wave_lds_sync();
          // ---- reverse-time dynamics: dy/ds = -f, da/ds = +a^T df/dz (3/8 rule, torchdiffeq's association)
          const f32x16 ky = -f, ka = va;
          const float third = (float)(1.0 / 3.0);
          if (stage == 0) {
            ky1 = ky; ka1 = ka;
            yst = y0 + ds * ky1 * third;
            ast = a0 + ds * ka1 * third;
          } else if (stage == 1) {
            ky2 = ky; ka2 = ka;
            yst = y0 + ds * (ky2 - ky1 * third);
            ast = a0 + ds * (ka2 - ka1 * third);
          } else if (stage == 2) {
            yst = y0 + ds * (ky1 - ky2 + ky);
            ast = a0 + ds * (ka1 - ka2 + ka);
            ky1 = ky1 + 3.f * (ky2 + ky);
            ka1 = ka1 + 3.f * (ka2 + ka);
          } else {
            yst = y0 + (ky1 + ky) * ds * 0.125f;
            ast = a0 + (ka1 + ka) * ds * 0.125f;
          }
          idx = nidx; frac = nfrac;
        }
        y0 = yst; a0 = ast;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int u = 2 * r + half;
      if (u < Hr) {
        y0[r] = z_saved[(sc * n_out + (i_out - 1)) * Hr + u];
        if (valid) a0[r] += grad_out[(sc * n_out + (i_out - 1)) * Hr + u];
      }
    }
  }
  if (valid) {
#pragma unroll
    for (int r = 0; r < 16; ++r) if (2 * r + half < Hr) grad_z0[series * Hr + 2 * r + half] = a0[r];
  }
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

// ------------------------------------------------------------------------------------------ host side
int launch_reduce_partials(const float* partial, int64_t n_tiles, void* grad_W, void* grad_b, int H, int C, hipStream_t s);   // rk4_mfma.hip

template <typename TT>
int launch_forward_bf16x3(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                          const void* bias, const void* z0, const void* grid, int64_t n_grid, const void* t_out,
                          int64_t n_out, void* z_out, int64_t B, int64_t C, int64_t H, const int64_t* stage_index,
                          const void* stage_frac, hipStream_t s) {
  const Dims dims{(int)H, (int)C};
  const unsigned blocks = (unsigned)((B + 127) / 128);
#define CDE_BX_FWD(D)                                                                                                \
  do {                                                                                                               \
    (void)hipFuncSetAttribute((const void*)rk4_forward_bf16x3<TT, D>, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                              BX_FWD_LDS_BYTES);                                                                     \
    rk4_forward_bf16x3<TT, D><<<blocks, 256, BX_FWD_LDS_BYTES, s>>>(                                                 \
        (const float*)coeffs, (const float*)knots, n_intervals, (const float*)W, (const float*)bias, (const float*)z0, \
        (const TT*)grid, n_grid, (const TT*)t_out, n_out, (float*)z_out, B, stage_index, (const float*)stage_frac, dims); \
  } while (0)
  if (degree == CDE_PATH_CUBIC) CDE_BX_FWD(CDE_PATH_CUBIC);
  else if (degree == CDE_PATH_LINEAR) CDE_BX_FWD(CDE_PATH_LINEAR);
  else return CDE_ERR_UNSUPPORTED;
#undef CDE_BX_FWD
  return check_launch();
}

template <typename TT>
int launch_adjoint_bf16x3(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                          const void* bias, const void* z_saved, const void* grad_out, const void* sgrid,
                          const int64_t* seg_off, int64_t n_out, void* grad_z0, void* grad_W, void* grad_b, int64_t B,
                          int64_t C, int64_t H, const int64_t* stage_index, const void* stage_frac, float* partial,
                          hipStream_t s) {
  const Dims dims{(int)H, (int)C};
  const unsigned blocks = (unsigned)((B + 127) / 128);
#define CDE_BX_ADJ(D)                                                                                                \
  do {                                                                                                               \
    (void)hipFuncSetAttribute((const void*)rk4_adjoint_bf16x3<TT, D>, hipFuncAttributeMaxDynamicSharedMemorySize,    \
                              BX_ADJ_LDS_BYTES);                                                                     \
    rk4_adjoint_bf16x3<TT, D><<<blocks, 256, BX_ADJ_LDS_BYTES, s>>>(                                                 \
        (const float*)coeffs, (const float*)knots, n_intervals, (const float*)W, (const float*)bias,                 \
        (const float*)z_saved, (const float*)grad_out, (const TT*)sgrid, seg_off, n_out, (float*)grad_z0, partial, B, \
        stage_index, (const float*)stage_frac, dims);                                                                \
  } while (0)
  if (degree == CDE_PATH_CUBIC) CDE_BX_ADJ(CDE_PATH_CUBIC);
  else if (degree == CDE_PATH_LINEAR) CDE_BX_ADJ(CDE_PATH_LINEAR);
  else return CDE_ERR_UNSUPPORTED;
#undef CDE_BX_ADJ
  const int rc = check_launch();
  if (rc != CDE_OK) return rc;
  return launch_reduce_partials(partial, (B + 31) / 32, grad_W, grad_b, (int)H, (int)C, s);
}

template int launch_forward_bf16x3<float>(const void*, const void*, int64_t, int, const void*, const void*, const void*,
                                          const void*, int64_t, const void*, int64_t, void*, int64_t, int64_t, int64_t,
                                          const int64_t*, const void*, hipStream_t);
template int launch_forward_bf16x3<double>(const void*, const void*, int64_t, int, const void*, const void*, const void*,
                                           const void*, int64_t, const void*, int64_t, void*, int64_t, int64_t, int64_t,
                                           const int64_t*, const void*, hipStream_t);
template int launch_adjoint_bf16x3<float>(const void*, const void*, int64_t, int, const void*, const void*, const void*,
                                          const void*, const void*, const int64_t*, int64_t, void*, void*, void*, int64_t,
                                          int64_t, int64_t, const int64_t*, const void*, float*, hipStream_t);
template int launch_adjoint_bf16x3<double>(const void*, const void*, int64_t, int, const void*, const void*, const void*,
                                           const void*, const void*, const int64_t*, int64_t, void*, void*, void*, int64_t,
                                           int64_t, int64_t, const int64_t*, const void*, float*, hipStream_t);

}  // namespace cde
