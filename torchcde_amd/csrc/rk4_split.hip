// rk4_split.hip -- latency-oriented RK4 (3/8) solve and continuous-adjoint sweep for SMALL per-GPU batches
// (strong scaling: 32768 series over 8 GPUs leave 4096 per GPU = 16 series per CU).
//
// K2/K3 (rk4_mfma.hip) give one wave 16/32 series for the whole 508-stage chain, so a batch below 32768 leaves SIMDs
// idle and takes as long as a full one.  Here ONE WORKGROUP of 4 waves (one per SIMD of a CU) owns a tile of 16
// series and the four waves split every stage's GEMMs between them; the stage state crosses the waves through LDS
// once per stage (one s_barrier per stage).  4096 series = 256 workgroups = every SIMD of the chip busy.
//
// f32, H <= 32, C <= 8 (zero padded), v_mfma_f32_16x16x4_f32, pre-activation form Y = W z + b (identity or tanh):
//   wave w owns hidden units 8w..8w+7; lane (n = l & 15, q = l >> 4) owns units ua = 8w + q, ub = 8w + 4 + q of
//   series n: their RK bookkeeping (y, k1, k2, ...) lives in that lane only.
//   Y tile T = 2P + tb (P, tb in {0,1}): row i <-> (h = 8w + 4P + (i >> 2), c = 4 tb + (i & 3))
//     => C/D fragment of lane (n, q), register r: Y[h = 8w + 4P + q][c = 4 tb + r]: all 8 channels of the lane's
//     two units after 4 tiles x 8 K steps = 32 MFMAs (a quarter of the 128 of the whole evaluation);
//     K step s: lane quarter kq feeds input unit 4s + kq, read from the stage-state buffer
//     zbuf[series][kq*8 + s] (2 x ds_read_b128); each lane publishes its two units with one ds_write_b64.
//   adjoint only (96 MFMAs per wave and stage):
//   va partial = W_w^T g            32 MFMAs, K = this wave's 64 (h, c) rows: K step s' = 8P + c, quarter kq <-> the
//                                     lane's OWN register g[h = 8w + 4P + kq][c] (no data movement); output rows are
//                                     permuted so that lane (n, q) holds, for every destination wave w', the partial
//                                     sums of w's units 8w' + q and 8w' + 4 + q; the four partials meet in LDS.
//   dW_w += (wq g)^T z              32 MFMAs, series = MFMA K: g goes through a wave-private LDS transpose, z through
//                                     a second (transposed) copy of the stage state; dL/db = row sums of the same tile.
// The A images (Y: 32, va: 32 values per lane) and the dW accumulators (32) live in registers for the whole solve.
// Arithmetic per series is the same as in K2a/K3a up to summation order; per-workgroup partial parameter gradients go
// through the same fixed-order reduction (reduce_mfma_partials), so results are run-to-run deterministic.
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
constexpr int SPL_ADJ_LDS_FLOATS = 2 * SPL_ZBUF + 2 * SPL_ZT + 2 * SPL_VA + 2 * SPL_DX + 4 * SPL_GT;
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
  const float* __restrict__ coeffs;
  const float* __restrict__ knots;
  const int64_t* __restrict__ sidx;
  const float* __restrict__ sfrac;
  int64_t n_intervals, sc, e_last;
  int Cr, c;
  int64_t idx1, idx2;          // table entries e+1, e+2 (uniform)
  float frac1, frac2;
  float cur[3], raw[3];
  bool pending;                // raw holds the coefficients of idx1, not yet installed in cur (uniform)

  __device__ __forceinline__ void request(int64_t idx) {
    const int cc = c < Cr ? c : Cr - 1;
    if (DEGREE == CDE_PATH_CUBIC) {
      const float* p = coeffs + (sc * n_intervals + idx) * 4 * Cr + cc;
      raw[0] = p[Cr]; raw[1] = p[2 * Cr]; raw[2] = p[3 * Cr];
    } else {
      const float* p = coeffs + (sc * (n_intervals + 1) + idx) * Cr + cc;
      raw[0] = p[0]; raw[1] = p[Cr]; raw[2] = knots[idx + 1] - knots[idx];
    }
  }
  __device__ __forceinline__ void install() {
    if (DEGREE == CDE_PATH_CUBIC) { cur[0] = raw[0]; cur[1] = raw[1]; cur[2] = raw[2]; }
    else cur[0] = (raw[1] - raw[0]) / raw[2];
  }
  __device__ __forceinline__ float value(float frac) const {
    const float v = DEGREE == CDE_PATH_CUBIC ? cubic_derivative(cur[0], cur[1], cur[2], frac) : cur[0];
    return c < Cr ? v : 0.f;
  }
  __device__ __forceinline__ int64_t clamp(int64_t e) const { return e < e_last ? e : e_last; }

  // start at table entry e0 (entries e0 .. e_last belong to this sweep): returns dX of entry e0
  __device__ __forceinline__ float begin(int64_t e0, int64_t last) {
    e_last = last;
    const int64_t idx0 = sidx[e0];
    const float frac0 = sfrac[e0];
    request(idx0);
    install();
    const float v0 = value(frac0);
    idx1 = sidx[clamp(e0 + 1)]; frac1 = sfrac[clamp(e0 + 1)];
    idx2 = sidx[clamp(e0 + 2)]; frac2 = sfrac[clamp(e0 + 2)];
    pending = idx1 != idx0;
    if (pending) request(idx1);
    return v0;
  }
  // during stage e: returns dX of entry e+1 and moves the pipeline on
  __device__ __forceinline__ float advance(int64_t e) {
    const int64_t idx3 = sidx[clamp(e + 3)];
    const float frac3 = sfrac[clamp(e + 3)];
    // opaque to the optimiser: otherwise it merges this install into the request of the previous stage (same
    // condition) and the load is waited for right where it was issued
    asm volatile("" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]));
    if (pending) install();
    const float v = value(frac1);
    pending = idx2 != idx1;
    if (pending) request(idx2);
    idx1 = idx2; frac1 = frac2; idx2 = idx3; frac2 = frac3;
    return v;
  }
};


// ============================================================================================ forward
template <typename TT, int DEGREE, int ACT>
__global__ __launch_bounds__(256, 2) void rk4_forward_split(
    const float* __restrict__ coeffs, const float* __restrict__ knots, int64_t n_intervals,
    const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ z0,
    const TT* __restrict__ grid, int64_t n_grid, const TT* __restrict__ t_out, int64_t n_out,
    float* __restrict__ z_out, int64_t B, const int64_t* __restrict__ stage_index,
    const float* __restrict__ stage_frac, Dims dims) {
  __shared__ __attribute__((aligned(16))) float zbuf[2 * SPL_ZBUF];
  __shared__ __attribute__((aligned(16))) float dxb[2 * SPL_DX];
  const int Hr = dims.H, Cr = dims.C;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int64_t series = (int64_t)blockIdx.x * 16 + n;
  const bool valid = series < B;
  const int64_t sc = valid ? series : B - 1;

  float wy[4][8];
  f32x4 by[4];
  spl_load_wy(W, bias, w, n, q, dims, wy, by);

  const int ua = 8 * w + q, ub = ua + 4;
  float ya = ua < Hr ? z0[sc * Hr + ua] : 0.f, yb = ub < Hr ? z0[sc * Hr + ub] : 0.f;
  auto store = [&](int64_t j, float a, float b) {
    if (valid) {
      float* row = z_out + (series * n_out + j) * Hr;
      if (ua < Hr) row[ua] = a;
      if (ub < Hr) row[ub] = b;
    }
  };
  store(0, ya, yb);
  const int64_t n_steps = n_grid - 1;
  if (n_steps <= 0) return;

  float* zw = zbuf + n * SPL_ZROW + q * 8 + 2 * w;          // writer: units (kq = q, s = 2w), (kq = q, s = 2w + 1)
  const float* zr = zbuf + n * SPL_ZROW + q * 8;            // reader: kq = q, s = 0..7
  Feed<DEGREE> feed;
  feed.coeffs = coeffs; feed.knots = knots; feed.sidx = stage_index; feed.sfrac = stage_frac;
  feed.n_intervals = n_intervals; feed.sc = sc; feed.Cr = Cr; feed.c = 2 * w + (q & 1);
  float* dxw = dxb + n * SPL_DXROW + feed.c;
  const float* dxr = dxb + n * SPL_DXROW;
  const bool feeds = q < 2;
  int par = 0;
  {
    const float d0 = feed.begin(0, 4 * n_steps - 1);
    *reinterpret_cast<float2*>(zw) = make_float2(ya, yb);
    if (feeds) dxw[0] = d0;
  }
  __syncthreads();

  int64_t jout = 1;
  for (int64_t k = 0; k < n_steps; ++k) {
    const TT t0 = grid[k], t1 = grid[k + 1];
    const float dt = (float)(t1 - t0);
    float k1a = 0.f, k1b = 0.f, k2a = 0.f, k2b = 0.f, pqa = 0.f, pqb = 0.f, za = ya, zb = yb;
#pragma unroll
    for (int stage = 0; stage < 4; ++stage) {
      const float4 z03 = *reinterpret_cast<const float4*>(zr + par * SPL_ZBUF);
      const float4 z47 = *reinterpret_cast<const float4*>(zr + par * SPL_ZBUF + 4);
      const float4 d03 = *reinterpret_cast<const float4*>(dxr + par * SPL_DX);
      const float4 d47 = *reinterpret_cast<const float4*>(dxr + par * SPL_DX + 4);
      const float zs[8] = {z03.x, z03.y, z03.z, z03.w, z47.x, z47.y, z47.z, z47.w};
      const float dX[MC] = {d03.x, d03.y, d03.z, d03.w, d47.x, d47.y, d47.z, d47.w};
      // next stage's control derivative (fills the LDS latency of the reads above)
      const float dnext = feed.advance(4 * k + stage);
      if (feeds) dxw[(par ^ 1) * SPL_DX] = dnext;
      __builtin_amdgcn_sched_barrier(0);

      f32x4 y00 = by[0], y01 = by[1], y10 = by[2], y11 = by[3];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        y00 = mfma16(wy[0][s], zs[s], y00);
        y01 = mfma16(wy[1][s], zs[s], y01);
        y10 = mfma16(wy[2][s], zs[s], y10);
        y11 = mfma16(wy[3][s], zs[s], y11);
      }
      float fa = 0.f, fb = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        fa = __builtin_fmaf(activate<ACT>(y00[r]), dX[r], fa);
        fb = __builtin_fmaf(activate<ACT>(y10[r]), dX[r], fb);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        fa = __builtin_fmaf(activate<ACT>(y01[r]), dX[4 + r], fa);
        fb = __builtin_fmaf(activate<ACT>(y11[r]), dX[4 + r], fb);
      }
      // torchdiffeq rk4_alt_step_func (3/8 rule), association order as in K2
      const float third = (float)(1.0 / 3.0);
      if (stage == 0) {
        k1a = fa; k1b = fb;
        za = ya + dt * k1a * third; zb = yb + dt * k1b * third;
      } else if (stage == 1) {
        k2a = fa; k2b = fb;
        za = ya + dt * (k2a - k1a * third); zb = yb + dt * (k2b - k1b * third);
      } else if (stage == 2) {
        za = ya + dt * (k1a - k2a + fa); zb = yb + dt * (k1b - k2b + fb);
        pqa = k1a + 3.f * (k2a + fa); pqb = k1b + 3.f * (k2b + fb);
      } else {
        za = ya + (pqa + fa) * dt * 0.125f; zb = yb + (pqb + fb) * dt * 0.125f;
      }
      *reinterpret_cast<float2*>(zw + (par ^ 1) * SPL_ZBUF) = make_float2(za, zb);
      __syncthreads();
      par ^= 1;
    }
    const float y1a = za, y1b = zb;
    while (jout < n_out && t1 >= t_out[jout]) {
      const TT tj = t_out[jout];
      if (tj == t0) store(jout, ya, yb);
      else if (tj == t1) store(jout, y1a, y1b);
      else {
        const float slope = (float)((tj - t0) / (t1 - t0));
        store(jout, ya + slope * (y1a - ya), yb + slope * (y1b - yb));
      }
      ++jout;
    }
    ya = y1a; yb = y1b;
  }
}

// ============================================================================================ adjoint
// Stage body, between two workgroup barriers (e = this stage, e-1 = the previous one):
//   reads of what the other waves published (z, z^T, dX, va partials of e-1)      <- LDS latency ...
//   dW += g(e-1)^T (wq z(e-1)), first half: operands were parked in registers      <- ... hidden behind 16 MFMAs
//   a-path RK update from the partials; control feed for e+1
//   Y tiles (32 MFMAs) -> f, g -> y-path RK update -> publish z(e+1), write g^T (wave-private)
//   va partials (32 MFMAs, cover the g^T write latency) -> read g^T back as the A operand of dW, publish partials
//   dW, second half of e-1 (16 MFMAs: cover the partial-write and g^T-read latency)
//   park g^T(e), wq z^T(e) for the next body; barrier
// so every LDS round trip has matrix work in front of it, and the only VALU left is what the stage really needs
// (on gfx950 f32 MFMA and VALU instructions of a wave do not overlap: instruction count is time).
template <int ACT>
__device__ __forceinline__ float spl_slope(float t) { return ACT == CDE_ACT_TANH ? __builtin_fmaf(-t, t, 1.f) : 1.f; }

template <typename TT, int DEGREE, int ACT>
__global__ __launch_bounds__(256, 1) void rk4_adjoint_split(
    const float* __restrict__ coeffs, const float* __restrict__ knots, int64_t n_intervals,
    const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ z_saved,
    const float* __restrict__ grad_out, const TT* __restrict__ sgrid, const int64_t* __restrict__ seg_off,
    int64_t n_out, float* __restrict__ grad_z0, float* __restrict__ partial, int64_t B,
    const int64_t* __restrict__ stage_index, const float* __restrict__ stage_frac, Dims dims) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int Hr = dims.H, Cr = dims.C;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int n = lane & 15, q = lane >> 4;
  float* zbuf = lds;
  float* ztb = lds + 2 * SPL_ZBUF;
  float* vab = ztb + 2 * SPL_ZT;
  float* dxb = vab + 2 * SPL_VA;
  float* gT = dxb + 2 * SPL_DX + w * SPL_GT;
  const int64_t series = (int64_t)blockIdx.x * 16 + n;
  const bool valid = series < B;
  const int64_t sc = valid ? series : B - 1;

  float wy[4][8], wv[2][16];
  f32x4 by[4];
  spl_load_wy(W, bias, w, n, q, dims, wy, by);
  // va image: tile T, K step s' = 8P + c: A[i = n][kq = q] = W[(h = 8w + 4P + q, c)][k_out(T, i)],
  // k_out = 8 (2T + (r >> 1)) + 4 (r & 1) + qi  with (qi, r) = (i >> 2, i & 3)
#pragma unroll
  for (int T = 0; T < 2; ++T) {
    const int qi = n >> 2, r = n & 3;
    const int k_out = 8 * (2 * T + (r >> 1)) + 4 * (r & 1) + qi;
#pragma unroll
    for (int sp = 0; sp < 16; ++sp) {
      const int h = 8 * w + 4 * (sp >> 3) + q, c = sp & 7;
      wv[T][sp] = (h < Hr && c < Cr && k_out < Hr) ? W[(h * Cr + c) * Hr + k_out] : 0.f;
    }
  }

  f32x4 accW[4][2];
  float gb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int Tm = 0; Tm < 4; ++Tm) {
    accW[Tm][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    accW[Tm][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  // parked operands of the deferred dW product (zeros: the first, empty product adds nothing)
  float4 gAh[4], zbh0 = make_float4(0.f, 0.f, 0.f, 0.f), zbh1 = zbh0;
  float wqh = 0.f;
#pragma unroll
  for (int Tm = 0; Tm < 4; ++Tm) gAh[Tm] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto dw_half = [&](int half) {
    const float b0[4] = {zbh0.x, zbh0.y, zbh0.z, zbh0.w}, b1[4] = {zbh1.x, zbh1.y, zbh1.z, zbh1.w};
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      const int Tm = 2 * half + t2;
      const float ga[4] = {gAh[Tm].x, gAh[Tm].y, gAh[Tm].z, gAh[Tm].w};
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        accW[Tm][0] = mfma16(ga[s], b0[s], accW[Tm][0]);
        accW[Tm][1] = mfma16(ga[s], b1[s], accW[Tm][1]);
      }
      gb[Tm] = __builtin_fmaf((ga[0] + ga[1]) + (ga[2] + ga[3]), wqh, gb[Tm]);     // dL/db: row sums
    }
  };

  const int ua = 8 * w + q, ub = ua + 4;
  const int pos = spl_pos(n);
  auto saved = [&](int64_t j, int u) { return u < Hr ? z_saved[(sc * n_out + j) * Hr + u] : 0.f; };
  auto gout = [&](int64_t j, int u) { return (valid && u < Hr) ? grad_out[(sc * n_out + j) * Hr + u] : 0.f; };
  float y0a = saved(n_out - 1, ua), y0b = saved(n_out - 1, ub);
  float a0a = gout(n_out - 1, ua), a0b = gout(n_out - 1, ub);      // a == 0 stays 0: padded lanes add nothing to dL/dW

  float* zw = zbuf + n * SPL_ZROW + q * 8 + 2 * w;
  const float* zr = zbuf + n * SPL_ZROW + q * 8;
  float* ztw = ztb + ua * SPL_TROW + pos;                          // second unit: + 4 rows
  const float* ztr = ztb + n * SPL_TROW + 4 * q;                   // N tile 1: + 16 rows
  float* vw = vab + (q * 16 + n) * SPL_VROW + 2 * w;               // + w_dst * 64 * SPL_VROW
  const float* vr = vab + ((w * 4 + q) * 16 + n) * SPL_VROW;
  float* gw_ = gT + (4 * q) * SPL_TROW + pos;                      // + (T*16 + r) rows
  const float* gr = gT + n * SPL_TROW + 4 * q;                     // + Tm*16 rows
  Feed<DEGREE> feed;
  feed.coeffs = coeffs; feed.knots = knots; feed.sidx = stage_index; feed.sfrac = stage_frac;
  feed.n_intervals = n_intervals; feed.sc = sc; feed.Cr = Cr; feed.c = 2 * w + (q & 1);
  float* dxw = dxb + n * SPL_DXROW + feed.c;
  const float* dxr = dxb + n * SPL_DXROW;
  const bool feeds = q < 2;
  int par = 0;
  auto publish = [&](int p, float za, float zb) {
    *reinterpret_cast<float2*>(zw + p * SPL_ZBUF) = make_float2(za, zb);
    ztw[p * SPL_ZT] = za;
    ztw[p * SPL_ZT + 4 * SPL_TROW] = zb;
  };
  // the four waves' partial sums of this wave's two units (fixed order): da/ds = +a^T df/dz
  auto read_ka = [&](int p, float& kaa, float& kab) {
    const float4 p03 = *reinterpret_cast<const float4*>(vr + p * SPL_VA);
    const float4 p47 = *reinterpret_cast<const float4*>(vr + p * SPL_VA + 4);
    kaa = (p03.x + p03.z) + (p47.x + p47.z);
    kab = (p03.y + p03.w) + (p47.y + p47.w);
  };
  const float third = (float)(1.0 / 3.0);

  for (int64_t p = 0; p + 1 < n_out; ++p) {
    const int64_t i_out = n_out - 1 - p;
    const int64_t k_begin = seg_off[p], k_end = seg_off[p + 1] - 1;   // steps k_begin .. k_end-1
    if (k_end > k_begin) {
      {
        const float d0 = feed.begin(4 * k_begin, 4 * k_end - 1);
        publish(par, y0a, y0b);
        if (feeds) dxw[par * SPL_DX] = d0;
      }
      __syncthreads();
      float ds_prev = 0.f;
      float ka1a = 0.f, ka1b = 0.f, ka2a = 0.f, ka2b = 0.f, asa = a0a, asb = a0b;
      for (int64_t k = k_begin; k < k_end; ++k) {
        const float ds = (float)(sgrid[k + 1] - sgrid[k]);
        float ky1a = 0.f, ky1b = 0.f, ky2a = 0.f, ky2b = 0.f;
        float ysa = y0a, ysb = y0b;
#pragma unroll
        for (int stage = 0; stage < 4; ++stage) {
          // ---- everything the other waves published before the barrier
          const float4 z03 = *reinterpret_cast<const float4*>(zr + par * SPL_ZBUF);
          const float4 z47 = *reinterpret_cast<const float4*>(zr + par * SPL_ZBUF + 4);
          const float4 d03 = *reinterpret_cast<const float4*>(dxr + par * SPL_DX);
          const float4 d47 = *reinterpret_cast<const float4*>(dxr + par * SPL_DX + 4);
          const float4 zt0 = *reinterpret_cast<const float4*>(ztr + par * SPL_ZT);
          const float4 zt1 = *reinterpret_cast<const float4*>(ztr + par * SPL_ZT + 16 * SPL_TROW);
          float kaa = 0.f, kab = 0.f;
          const bool first = stage == 0 && k == k_begin;               // no previous stage in this sweep
          if (!first) read_ka(par, kaa, kab);
          __builtin_amdgcn_sched_barrier(0);
          dw_half(0);                                                   // deferred dW of stage e-1, first half
          __builtin_amdgcn_sched_barrier(0);
          // ---- a path: RK update that stage e-1 left open
          if (stage == 1) {
            ka1a = kaa; ka1b = kab;
            asa = a0a + ds * ka1a * third; asb = a0b + ds * ka1b * third;
          } else if (stage == 2) {
            ka2a = kaa; ka2b = kab;
            asa = a0a + ds * (ka2a - ka1a * third); asb = a0b + ds * (ka2b - ka1b * third);
          } else if (stage == 3) {
            asa = a0a + ds * (ka1a - ka2a + kaa); asb = a0b + ds * (ka1b - ka2b + kab);
            ka1a = ka1a + 3.f * (ka2a + kaa); ka1b = ka1b + 3.f * (ka2b + kab);
          } else if (!first) {
            asa = a0a + (ka1a + kaa) * ds_prev * 0.125f; asb = a0b + (ka1b + kab) * ds_prev * 0.125f;
            a0a = asa; a0b = asb;
          }
          const float zs[8] = {z03.x, z03.y, z03.z, z03.w, z47.x, z47.y, z47.z, z47.w};
          const float dX[MC] = {d03.x, d03.y, d03.z, d03.w, d47.x, d47.y, d47.z, d47.w};
          const float dnext = feed.advance(4 * k + stage);
          if (feeds) dxw[(par ^ 1) * SPL_DX] = dnext;
          const float wq = ((stage == 0 || stage == 3) ? 0.125f : 0.375f) * ds;     // 3/8-rule quadrature weight
          // B operand of this stage's dW (executed during the next body): the quadrature weight rides on z
          const f32x2 zq[4] = {f32x2{zt0.x, zt0.y} * wq, f32x2{zt0.z, zt0.w} * wq, f32x2{zt1.x, zt1.y} * wq,
                               f32x2{zt1.z, zt1.w} * wq};
          __builtin_amdgcn_sched_barrier(0);

          // ---- Y tiles of this wave's 8 hidden units
          f32x4 yt[4] = {by[0], by[1], by[2], by[3]};
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            yt[0] = mfma16(wy[0][s], zs[s], yt[0]);
            yt[1] = mfma16(wy[1][s], zs[s], yt[1]);
            yt[2] = mfma16(wy[2][s], zs[s], yt[2]);
            yt[3] = mfma16(wy[3][s], zs[s], yt[3]);
          }
          // ---- activation, f for the two own units, g = dL/dY: B operand of va as it is, A operand of dW after the
          // wave-private transpose.  Pairs = two neighbouring channels of one unit (consecutive registers of a tile
          // and of the dX row): packed instructions without register shuffles.
          f32x2 gq[4][2];                    // gq[T][j] = g[unit P = T >> 1][channels 4 (T & 1) + 2j, + 1]
          f32x2 fpa = {0.f, 0.f}, fpb = {0.f, 0.f};
#pragma unroll
          for (int T = 0; T < 4; ++T) {
            const float aown = (T >> 1) ? asb : asa;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const f32x2 dx = {dX[4 * (T & 1) + 2 * j], dX[4 * (T & 1) + 2 * j + 1]};
              const f32x2 t = activate2<ACT>(yt[T][2 * j], yt[T][2 * j + 1]);
              if (T >> 1) fpb = __builtin_elementwise_fma(t, dx, fpb); else fpa = __builtin_elementwise_fma(t, dx, fpa);
              if (ACT == CDE_ACT_NONE) gq[T][j] = dx * aown;
              else gq[T][j] = (f32x2{spl_slope<ACT>(t[0]), spl_slope<ACT>(t[1])} * dx) * aown;
              gw_[(T * 16 + 2 * j) * SPL_TROW] = gq[T][j][0];
              gw_[(T * 16 + 2 * j + 1) * SPL_TROW] = gq[T][j][1];
            }
          }
          const f32x2 fp = {fpa[0] + fpa[1], fpb[0] + fpb[1]};
          // ---- the y path does not wait for anything else: next stage state of the own units -> LDS
          const float kya = -fp[0], kyb = -fp[1];          // reverse time: dy/ds = -f
          float nya, nyb;
          if (stage == 0) {
            ky1a = kya; ky1b = kyb;
            nya = y0a + ds * ky1a * third; nyb = y0b + ds * ky1b * third;
          } else if (stage == 1) {
            ky2a = kya; ky2b = kyb;
            nya = y0a + ds * (ky2a - ky1a * third); nyb = y0b + ds * (ky2b - ky1b * third);
          } else if (stage == 2) {
            nya = y0a + ds * (ky1a - ky2a + kya); nyb = y0b + ds * (ky1b - ky2b + kyb);
            ky1a = ky1a + 3.f * (ky2a + kya); ky1b = ky1b + 3.f * (ky2b + kyb);
          } else {
            nya = y0a + (ky1a + kya) * ds * 0.125f; nyb = y0b + (ky1b + kyb) * ds * 0.125f;
          }
          publish(par ^ 1, nya, nyb);
          ysa = nya; ysb = nyb;
          __builtin_amdgcn_sched_barrier(0);

          // ---- va partial over this wave's 64 (h, c) rows, all 32 output units (K step 8P + c)
          f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
#pragma unroll
          for (int sp = 0; sp < 16; ++sp) {
            const float gv = gq[2 * (sp >> 3) + ((sp >> 2) & 1)][(sp >> 1) & 1][sp & 1];     // g[P = sp >> 3][c = sp & 7]
            v0 = mfma16(wv[0][sp], gv, v0);
            v1 = mfma16(wv[1][sp], gv, v1);
          }
          // ---- g^T back as the A operand of this stage's dW (wave-private: only this wave's LDS writes are awaited,
          // and only AFTER the va chain -- the empty asm ties the wait to the chain's results, volatile asms keep
          // their order -- so the write latency is behind 32 MFMAs instead of in front of them)
          asm volatile("" : "+v"(v0), "+v"(v1));
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_wave_barrier();
          float4 gAn[4];
#pragma unroll
          for (int Tm = 0; Tm < 4; ++Tm) gAn[Tm] = *reinterpret_cast<const float4*>(gr + Tm * 16 * SPL_TROW);
          // register r of tile T -> destination wave 2T + (r >> 1), its unit j = r & 1
          float* vwp = vw + (par ^ 1) * SPL_VA;
          *reinterpret_cast<float2*>(vwp) = make_float2(v0[0], v0[1]);
          *reinterpret_cast<float2*>(vwp + 64 * SPL_VROW) = make_float2(v0[2], v0[3]);
          *reinterpret_cast<float2*>(vwp + 2 * 64 * SPL_VROW) = make_float2(v1[0], v1[1]);
          *reinterpret_cast<float2*>(vwp + 3 * 64 * SPL_VROW) = make_float2(v1[2], v1[3]);
          __builtin_amdgcn_sched_barrier(0);
          dw_half(1);                                                   // deferred dW of stage e-1, second half
          __builtin_amdgcn_sched_barrier(0);
          // ---- park this stage's operands (the quadrature weight rides on z)
#pragma unroll
          for (int Tm = 0; Tm < 4; ++Tm) gAh[Tm] = gAn[Tm];
          zbh0 = make_float4(zq[0][0], zq[0][1], zq[1][0], zq[1][1]);
          zbh1 = make_float4(zq[2][0], zq[2][1], zq[3][0], zq[3][1]);
          wqh = wq;
          __syncthreads();
          par ^= 1;
        }
        y0a = ysa; y0b = ysb;
        ds_prev = ds;
      }
      // the last stage's a-path update
      {
        float kaa, kab;
        read_ka(par, kaa, kab);
        a0a = a0a + (ka1a + kaa) * ds_prev * 0.125f; a0b = a0b + (ka1b + kab) * ds_prev * 0.125f;
      }
    }
    // torchdiffeq adjoint: re-seed y from the stored forward value, add the incoming gradient
    y0a = saved(i_out - 1, ua); y0b = saved(i_out - 1, ub);
    a0a += gout(i_out - 1, ua); a0b += gout(i_out - 1, ub);
  }
  dw_half(0);                                                           // the last stage's dW
  dw_half(1);
  if (valid) {
    if (ua < Hr) grad_z0[series * Hr + ua] = a0a;
    if (ub < Hr) grad_z0[series * Hr + ub] = a0b;
  }
  // per-workgroup partial parameter gradients in the K3 layout (summed in tile order by reduce_mfma_partials):
  // accW[Tm][Tn] register r of lane (j = n, q) = dW[h = 8w + 4(Tm>>1) + q][c = 4(Tm&1) + r][k = 16 Tn + j]
  float* my_partial = partial + (int64_t)blockIdx.x * SPL_PARTIAL_FLOATS;
#pragma unroll
  for (int Tm = 0; Tm < 4; ++Tm) {
    const int h = 8 * w + 4 * (Tm >> 1) + q;
#pragma unroll
    for (int Tn = 0; Tn < 2; ++Tn) {
#pragma unroll
      for (int r = 0; r < 4; ++r) my_partial[(h * MC + 4 * (Tm & 1) + r) * MH + 16 * Tn + n] = accW[Tm][Tn][r];
    }
    // row sums: lane (i = n, kq = q) summed row i over the series of its quarter
    float s = gb[Tm];
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    if (q == 0) my_partial[MH * MC * MH + (8 * w + 4 * (Tm >> 1) + (n >> 2)) * MC + 4 * (Tm & 1) + (n & 3)] = s;
  }
}

// defined in rk4_mfma.hip: fixed-order sum of the per-tile partials
int launch_reduce_partials(const float* partial, int64_t n_tiles, void* grad_W, void* grad_b, int H, int C, hipStream_t s);

// ------------------------------------------------------------------------------------------ host side
size_t split_adjoint_partial_bytes(int64_t B) { return (size_t)((B + 15) / 16) * SPL_PARTIAL_FLOATS * sizeof(float); }

template <typename TT>
int launch_forward_split(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                         const void* bias, int act, const void* z0, const void* grid, int64_t n_grid, const void* t_out,
                         int64_t n_out, void* z_out, int64_t B, int64_t C, int64_t H, const int64_t* stage_index,
                         const void* stage_frac, hipStream_t s) {
  const Dims dims{(int)H, (int)C};
  const unsigned blocks = (unsigned)((B + 15) / 16);
#define CDE_FWD(D, A)                                                                                                \
  rk4_forward_split<TT, D, A><<<blocks, 256, 0, s>>>(                                                                \
      (const float*)coeffs, (const float*)knots, n_intervals, (const float*)W, (const float*)bias, (const float*)z0, \
      (const TT*)grid, n_grid, (const TT*)t_out, n_out, (float*)z_out, B, stage_index, (const float*)stage_frac, dims)
  if (degree != CDE_PATH_CUBIC && degree != CDE_PATH_LINEAR) return CDE_ERR_UNSUPPORTED;
  if (act == CDE_ACT_NONE) {
    if (degree == CDE_PATH_CUBIC) CDE_FWD(CDE_PATH_CUBIC, CDE_ACT_NONE); else CDE_FWD(CDE_PATH_LINEAR, CDE_ACT_NONE);
  } else if (act == CDE_ACT_TANH) {
    if (degree == CDE_PATH_CUBIC) CDE_FWD(CDE_PATH_CUBIC, CDE_ACT_TANH); else CDE_FWD(CDE_PATH_LINEAR, CDE_ACT_TANH);
  } else return CDE_ERR_UNSUPPORTED;
#undef CDE_FWD
  return check_launch();
}

template <typename TT>
int launch_adjoint_split(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                         const void* bias, int act, const void* z_saved, const void* grad_out, const void* sgrid,
                         const int64_t* seg_off, int64_t n_out, void* grad_z0, void* grad_W, void* grad_b, int64_t B,
                         int64_t C, int64_t H, const int64_t* stage_index, const void* stage_frac, float* partial,
                         hipStream_t s) {
  const Dims dims{(int)H, (int)C};
  const unsigned blocks = (unsigned)((B + 15) / 16);
  const size_t lds = (size_t)SPL_ADJ_LDS_FLOATS * sizeof(float);
  if (degree != CDE_PATH_CUBIC && degree != CDE_PATH_LINEAR) return CDE_ERR_UNSUPPORTED;
#define CDE_ADJ(D, A)                                                                                                \
  do {                                                                                                               \
    (void)hipFuncSetAttribute((const void*)rk4_adjoint_split<TT, D, A>, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                              (int)lds);                                                                             \
    rk4_adjoint_split<TT, D, A><<<blocks, 256, lds, s>>>(                                                            \
        (const float*)coeffs, (const float*)knots, n_intervals, (const float*)W, (const float*)bias,                 \
        (const float*)z_saved, (const float*)grad_out, (const TT*)sgrid, seg_off, n_out, (float*)grad_z0, partial,   \
        B, stage_index, (const float*)stage_frac, dims);                                                             \
  } while (0)
  if (act == CDE_ACT_NONE) {
    if (degree == CDE_PATH_CUBIC) CDE_ADJ(CDE_PATH_CUBIC, CDE_ACT_NONE); else CDE_ADJ(CDE_PATH_LINEAR, CDE_ACT_NONE);
  } else if (act == CDE_ACT_TANH) {
    if (degree == CDE_PATH_CUBIC) CDE_ADJ(CDE_PATH_CUBIC, CDE_ACT_TANH); else CDE_ADJ(CDE_PATH_LINEAR, CDE_ACT_TANH);
  } else return CDE_ERR_UNSUPPORTED;
#undef CDE_ADJ
  int rc = check_launch();
  if (rc != CDE_OK) return rc;
  return launch_reduce_partials(partial, (B + 15) / 16, grad_W, grad_b, (int)H, (int)C, s);
}

template int launch_forward_split<float>(const void*, const void*, int64_t, int, const void*, const void*, int,
                                         const void*, const void*, int64_t, const void*, int64_t, void*, int64_t, int64_t,
                                         int64_t, const int64_t*, const void*, hipStream_t);
template int launch_forward_split<double>(const void*, const void*, int64_t, int, const void*, const void*, int,
                                          const void*, const void*, int64_t, const void*, int64_t, void*, int64_t,
                                          int64_t, int64_t, const int64_t*, const void*, hipStream_t);
template int launch_adjoint_split<float>(const void*, const void*, int64_t, int, const void*, const void*, int,
                                         const void*, const void*, const void*, const int64_t*, int64_t, void*, void*,
                                         void*, int64_t, int64_t, int64_t, const int64_t*, const void*, float*,
                                         hipStream_t);
template int launch_adjoint_split<double>(const void*, const void*, int64_t, int, const void*, const void*, int,
                                          const void*, const void*, const void*, const int64_t*, int64_t, void*, void*,
                                          void*, int64_t, int64_t, int64_t, const int64_t*, const void*, float*,
                                          hipStream_t);

}  // namespace cde
