// rk4_adjoint_pair.hip -- K3p: K3j (rk4_mfma.hip: the continuous-adjoint sweep of the affine field through the shared
// Jacobian) as TWO waves per 32-series tile that share a SIMD.
//
// Why.  K3j runs one wave per SIMD (its 128 dL/dW accumulators and the eight state vectors of the 3/8 rule do not fit
// 256 registers) and a wave overlaps nothing with its own f32 MFMAs (scripts/ubench/mfma_issue.hip): per stage
// 260 MFMAs x 64 cycles + ~1000 vector / LDS instructions x ~4.5 cycles = 21.3 k cycles, of which the matrix pipe is
// busy 16.6 k (0.76 of the f32 peak, profiles/r04_bench_kernel_stats.csv).  The same micro-benchmark shows a second wave
// on the SIMD taking over more than half of the non-MFMA slots (84 -> 74 cycles per [v_mul, MFMA] pair).  So the stage
// is cut along its one loose dependency -- dL/dW, dL/db only CONSUME (z, a, dX) of a stage, nothing reads them back:
//   chain wave  (waves 0..3 of the workgroup, priority 3): the 33 Jacobian rows (132 MFMAs, results in vector registers),
//               f = J z + b dX and a^T J on the vector pipe, the 3/8 rule; at the START of a stage it leaves (z, a)
//               transposed in one of two LDS buffers of its tile and picks up the stage's control derivative;
//   helper wave (waves 4..7; wave w + 4 shares a SIMD with wave w): one stage behind with dL/dW += (w ds a (x) dX)^T z
//               (128 MFMAs, batch = MFMA K) and dL/db -- 136 accumulators, kept in VECTOR registers
//               (-mllvm -amdgpu-mfma-vgpr-form for this file: the allocation of a kernel is max VGPR + max AGPR over both
//               roles, and only <= 256 in total leaves room for two waves per SIMD) -- and two stages AHEAD with the
//               control: it owns the stage table, the coefficient rows and the knot widths, forms dX/dt of every stage
//               and leaves it (plain for the chain wave, weighted by the quadrature weight for itself) in a three-slot LDS
//               ring.  That takes 30 registers and ~50 vector instructions per stage off the chain wave, which then fits
//               256 registers (12 bytes of scratch, touched once per RK step).
// One s_barrier per stage orders everything: the chain wave writes (z, a) buffer s & 1 and reads ring slot s % 3 before
// barrier s; the helper reads both between barriers s and s + 1 and writes slot (s + 2) % 3; the chain wave's next write to
// buffer s & 1 comes after barrier s + 1.  The matrix pipe sees the chain wave's dependent row chains and the helper's
// eight independent accumulator chains interleaved.
// Measured (profiles/r05_k3_pair_b.log, 32768 series): 5.26 -> 5.01 ms.  Chain / helper priorities make no difference
// (CDE_K3P_FLAGS), nor does requesting the helper's LDS operands a K step ahead: what is left is arithmetic -- on gfx950
// the f32 MFMA runs on the vector ALUs (157.3 TFLOP/s is the VECTOR peak), so the other wave's packed FMAs take matrix-pipe
// time too; the second wave only removes the ISSUE gaps a single wave leaves (~5 cycles per instruction down to its ~4 of
// ALU time): 260 MFMAs x 64 + ~950 vector instructions x ~4 = 20.4 k cycles per stage, measured ~21 k.
// Arithmetic: K3j's, operation for operation (same rows, same consume order, same transposes, same dL/dW K order), so
// results are bitwise K3j's; the per-wave partials keep K3j's layout and go through reduce_mfma_partials.
#include "cde_mfma.h"

namespace cde {

int launch_reduce_partials(const float* partial, int64_t n_tiles, void* grad_W, void* grad_b, int H, int C, hipStream_t s);

namespace {

constexpr int64_t KP_PARTIAL_FLOATS = MH * MC * MH + MH * MC;      // == PARTIAL_FLOATS of rk4_mfma.hip
constexpr int KP_WJ_ROWS = MH + 1;                                 // 32 rows of J + the bias rows
constexpr int KP_WJ_FLOATS = KP_WJ_ROWS * 64 * 4;
constexpr int KP_ZA_FLOATS = 2 * 64 * 20;                          // one stage's transposed (z, a): 64 rows x 20 each
constexpr int KP_TILE_FLOATS = 2 * KP_ZA_FLOATS + 3 * 256 + 3 * 256;   // per tile: 2 x (z^T | a^T), 3 x dX, 3 x weighted dX
constexpr int KP_LDS_FLOATS = KP_WJ_FLOATS + 4 * KP_TILE_FLOATS;

__device__ __forceinline__ void kp_swap32(float& x, float& y) {    // x[lanes 32..63] <-> y[lanes 0..31]
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
  x = __uint_as_float(r[0]);
  y = __uint_as_float(r[1]);
}
__device__ __forceinline__ float kp_wj_image(const float* __restrict__ W, const float* __restrict__ bias, int h, int s, int l,
                                             Dims d) {
  const int k = rho(l & 31), c = 2 * s + (l >> 5);
  if (c >= d.C || k >= d.H) return 0.f;
  if (h < MH) return h < d.H ? W[(h * d.C + c) * d.H + k] : 0.f;
  return bias[k * d.C + c];
}
// the stage barrier: this wave's LDS traffic done, then all eight waves meet
__device__ __forceinline__ void kp_stage_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// METHOD (CDE_METHOD_*): the 3/8 rule, or torchdiffeq's midpoint / euler on the same augmented system -- two stages / one
// stage per step, quadrature weights (0, ds) / (ds) for the parameter gradients (a zero-weight stage costs the helper nothing).
template <typename TT, int DEGREE, int METHOD = CDE_METHOD_RK4>
__global__ __launch_bounds__(512, 2) void rk4_adjoint_jacobian_pair(
    const float* __restrict__ coeffs, const float* __restrict__ knots, int64_t n_intervals,
    const float* __restrict__ W, const float* __restrict__ bias, const float* __restrict__ z_saved,
    const float* __restrict__ grad_out, const TT* __restrict__ sgrid, const int64_t* __restrict__ seg_off,
    int64_t n_out, float* __restrict__ grad_z0, float* __restrict__ partial, int64_t B,
    const int64_t* __restrict__ stage_index, const float* __restrict__ stage_frac, Dims dims, int flags) {
  const int Hr = dims.H, Cr = dims.C;
  constexpr int NS = METHOD == CDE_METHOD_RK4 ? 4 : METHOD == CDE_METHOD_MIDPOINT ? 2 : 1;     // stages per step
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int e = threadIdx.x; e < KP_WJ_FLOATS; e += 512) lds[e] = kp_wj_image(W, bias, e >> 8, e & 3, (e >> 2) & 63, dims);
  const float4* wj = reinterpret_cast<const float4*>(lds);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int slot = wave & 3;                                       // the tile of this wave within the workgroup
  const bool helper = wave >= 4;
  const int n = lane & 31, half = lane >> 5;
  // per tile: [2 x (z^T | a^T)] [3 x dX] [3 x weighted dX]          (KP_TILE_FLOATS)
  float* tile_lds = lds + KP_WJ_FLOATS + slot * KP_TILE_FLOATS;
  float* ring_dx = tile_lds + 2 * KP_ZA_FLOATS;                      // + (stage % 3) * 256: dX_c of series n at [n*8 + c]
  float* ring_dw = ring_dx + 3 * 256;                                // + (stage % 3) * 256: (quadrature weight * ds) * dX_c

  const int64_t tile = (int64_t)blockIdx.x * 4 + slot;
  const bool live = tile * 32 < B;                                  // (dead tiles keep pace with the barriers, store nothing)
  const int64_t series = tile * 32 + n;
  const bool valid = series < B;
  const int64_t sc = valid ? series : B - 1;

  if (helper) {
    // ------------------------------------------------------------------------------------------ helper wave
    f32x16 accW[MC];
    f32x2 gbp[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};   // dL/db partials, channel pairs
#pragma unroll
    for (int c = 0; c < MC; ++c) {
#pragma unroll
      for (int r = 0; r < 16; ++r) accW[c][r] = 0.f;
    }
    // the stages in sweep order: entry e = 4 k + stage of the stage table, k over the steps of segment p = 0, 1, ..
    int64_t n_stages = 0;
    for (int64_t p = 0; p + 1 < n_out; ++p) {
      const int64_t k_begin = seg_off[p], k_end = seg_off[p + 1] - 1;
      if (k_end > k_begin) n_stages += NS * (k_end - k_begin);
    }
    if (flags & 2) __builtin_amdgcn_s_setprio(3);
    // A cursor over those entries runs THREE stages ahead of the dL/dW work: the control derivative of stage st + 2 is
    // published (LDS ring of three slots; the chain wave reads slot st % 3 at the start of stage st, this wave's dL/dW of
    // stage st reads the weighted copy) from a row / fraction / step size requested one iteration earlier, and the
    // interval index of the stage after that is requested one iteration before its row -- no dependent global load is
    // waited for where it was issued.
    int64_t cur_p = -1, cur_e = 0, cur_end = 0;
    bool cur_ok = false;
    auto advance = [&]() {                     // (the table keeps four slots per step; a method uses the first NS of them)
      if (cur_ok && (cur_e & 3) + 1 < NS) { ++cur_e; return; }
      if (cur_ok && (cur_e | 3) + 1 < cur_end) { cur_e = (cur_e | 3) + 1; return; }
      cur_ok = false;
      for (int64_t p = cur_p + 1; p + 1 < n_out; ++p) {
        const int64_t k_begin = seg_off[p], k_end = seg_off[p + 1] - 1;
        if (k_end > k_begin) { cur_p = p; cur_e = 4 * k_begin; cur_end = 4 * k_end; cur_ok = true; return; }
      }
    };
    Row<DEGREE> row;
    int64_t row_idx = -1, idx_ahead = -1;
    float p_frac = 0.f, p_wq = 0.f, p_width = 1.f;
    bool p_ok = false;
    auto fetch = [&]() {                       // cursor's stage -> (row, p_*); then the cursor moves on and its index is requested
      p_ok = cur_ok;
      if (cur_ok) {
        if (idx_ahead != row_idx) { row = load_row<DEGREE>(coeffs, sc, n_intervals, idx_ahead, Cr); row_idx = idx_ahead; }
        p_frac = stage_frac[cur_e];
        if (DEGREE == CDE_PATH_LINEAR) p_width = knots[idx_ahead + 1] - knots[idx_ahead];
        const int64_t k = cur_e >> 2;
        const int stage = (int)(cur_e & 3);
        const float ds = (float)(sgrid[k + 1] - sgrid[k]);
        if constexpr (METHOD == CDE_METHOD_RK4) p_wq = ((stage == 0 || stage == 3) ? 0.125f : 0.375f) * ds;   // 3/8-rule quadrature weight
        else if constexpr (METHOD == CDE_METHOD_MIDPOINT) p_wq = stage == 1 ? ds : 0.f;                        // the midpoint evaluation alone
        else p_wq = ds;
      }
      advance();
      if (cur_ok) idx_ahead = stage_index[cur_e];
    };
    auto publish = [&](int which) {             // dX of the fetched stage -> ring slot `which` (plain and weighted)
      if (!p_ok) return;
      float dX[MC];
      control_slope<DEGREE>(row, p_frac, p_width, dX);
      const f32x2 d0 = half ? f32x2{dX[4], dX[5]} : f32x2{dX[0], dX[1]}, d1 = half ? f32x2{dX[6], dX[7]} : f32x2{dX[2], dX[3]};
      *reinterpret_cast<float4*>(ring_dx + which * 256 + n * 8 + 4 * half) = make_float4(d0[0], d0[1], d1[0], d1[1]);
      const f32x2 w0 = d0 * p_wq, w1 = d1 * p_wq;
      *reinterpret_cast<float4*>(ring_dw + which * 256 + n * 8 + 4 * half) = make_float4(w0[0], w0[1], w1[0], w1[1]);
    };
    advance();
    if (cur_ok) idx_ahead = stage_index[cur_e];
    fetch(); publish(0);                                            // stages 0 and 1 before the first barrier
    fetch(); publish(1);
    fetch();                                                        // stage 2: in flight
    __syncthreads();                                                // (also: the weight image is staged)
    int par = 0, rs = 0, cs = 0;                                    // (z, a) buffer, ring slot and stage-in-step of stage st
    for (int64_t st = 0; st < n_stages; ++st) {
      kp_stage_barrier();                                           // barrier `st`: buffer `par` holds (z, a) of stage `st`
      publish(rs >= 1 ? rs - 1 : 2);                                // stage st + 2 -> slot (st + 2) % 3
      fetch();                                                      // stage st + 3: requested, used next iteration
      const bool weighted = METHOD != CDE_METHOD_MIDPOINT || cs == 1;      // midpoint: the first evaluation carries no weight
      cs = cs + 1 == NS ? 0 : cs + 1;
      if (weighted) {
      // dL/dW tile c: D[h][k] += sum_series (w ds a_h dX_c)[series] * z_k[series]; this lane feeds MFMA K index `half`
      // of K-step s2, i.e. series 2*s2 + half, row h = n, column k = n.  Operands of K-step s2 + 1 are requested from LDS
      // before the MFMAs of K-step s2 are issued (their issue blocks this wave for as long as they take).
      const float* base = tile_lds + par * KP_ZA_FLOATS;
      const float4* zt4 = reinterpret_cast<const float4*>(base + (half * 32 + n) * 20);
      const float4* at4 = reinterpret_cast<const float4*>(base + 64 * 20 + (half * 32 + n) * 20);
      const float4* dw4 = reinterpret_cast<const float4*>(ring_dw + rs * 256 + half * 8);      // + s2*4 (16 floats per s2)
      float4 zq = zt4[0], aq = at4[0], e0 = dw4[0], e1 = dw4[1];
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) {
        const int i = s2 & 3;
        const f32x2 e01 = {e0.x, e0.y}, e23 = {e0.z, e0.w}, e45 = {e1.x, e1.y}, e67 = {e1.z, e1.w};
        const f32x2 asrc = i < 2 ? f32x2{aq.x, aq.y} : f32x2{aq.z, aq.w};
        const float zb = i == 0 ? zq.x : i == 1 ? zq.y : i == 2 ? zq.z : zq.w;
        f32x2 v01, v23, v45, v67;
        if (i & 1) {
          v01 = pk_mul_hi(e01, asrc); v23 = pk_mul_hi(e23, asrc); v45 = pk_mul_hi(e45, asrc); v67 = pk_mul_hi(e67, asrc);
          pk_fma_hi(gbp[0], e01, asrc); pk_fma_hi(gbp[1], e23, asrc); pk_fma_hi(gbp[2], e45, asrc); pk_fma_hi(gbp[3], e67, asrc);
        } else {
          v01 = pk_mul_lo(e01, asrc); v23 = pk_mul_lo(e23, asrc); v45 = pk_mul_lo(e45, asrc); v67 = pk_mul_lo(e67, asrc);
          pk_fma_lo(gbp[0], e01, asrc); pk_fma_lo(gbp[1], e23, asrc); pk_fma_lo(gbp[2], e45, asrc); pk_fma_lo(gbp[3], e67, asrc);
        }
        if (s2 < 15) {
          e0 = dw4[(s2 + 1) * 4]; e1 = dw4[(s2 + 1) * 4 + 1];
          if (i == 3) { zq = zt4[(s2 + 1) >> 2]; aq = at4[(s2 + 1) >> 2]; }
        }
        __builtin_amdgcn_sched_barrier(0);
        accW[0] = mfma(v01[0], zb, accW[0]); accW[1] = mfma(v01[1], zb, accW[1]);
        accW[2] = mfma(v23[0], zb, accW[2]); accW[3] = mfma(v23[1], zb, accW[3]);
        accW[4] = mfma(v45[0], zb, accW[4]); accW[5] = mfma(v45[1], zb, accW[5]);
        accW[6] = mfma(v67[0], zb, accW[6]); accW[7] = mfma(v67[1], zb, accW[7]);
        __builtin_amdgcn_sched_barrier(0);
      }
      }
      par ^= 1;
      rs = rs == 2 ? 0 : rs + 1;
    }
    if (live) {
      // per-wave partial parameter gradients (summed in tile order by reduce_mfma_partials), K3j's layout
      float* my_partial = partial + tile * KP_PARTIAL_FLOATS;
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
    return;
  }

  // -------------------------------------------------------------------------------------------- chain wave
  if (flags & 1) __builtin_amdgcn_s_setprio(3);  // the critical path: the helper's MFMAs fill the pipe while this wave is on the vector pipe
  f32x16 y0, a0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int u = 2 * r + half;
    const bool on = u < Hr;
    y0[r] = on ? z_saved[(sc * n_out + (n_out - 1)) * Hr + u] : 0.f;
    a0[r] = (valid && on) ? grad_out[(sc * n_out + (n_out - 1)) * Hr + u] : 0.f;   // a == 0 stays 0: padded lanes add nothing to dL/dW
  }
  __syncthreads();                               // the weight image and the helper's first two control derivatives are in LDS
  int par = 0, rs = 0;                           // (z, a) buffer and control-derivative slot of the current stage
  for (int64_t p = 0; p + 1 < n_out; ++p) {
    const int64_t i_out = n_out - 1 - p;
    const int64_t k_begin = seg_off[p], k_end = seg_off[p + 1] - 1;   // steps k_begin .. k_end-1
    for (int64_t k = k_begin; k < k_end; ++k) {
      const float ds = (float)(sgrid[k + 1] - sgrid[k]);
      f32x16 ky1, ky2, ka1, ka2, yst = y0, ast = a0;
#pragma unroll
      for (int stage = 0; stage < NS; ++stage) {
        // ---- this stage's control derivative (the helper wave left it two stages ago), the stage state -> this stage's
        // buffer (transposed); then the stage barrier
        float bs0, bs1, bs2, bs3;
        {
          const float4 dA = *reinterpret_cast<const float4*>(ring_dx + rs * 256 + n * 8);
          const float4 dB = *reinterpret_cast<const float4*>(ring_dx + rs * 256 + n * 8 + 4);
          bs0 = half ? dA.y : dA.x; bs1 = half ? dA.w : dA.z; bs2 = half ? dB.y : dB.x; bs3 = half ? dB.w : dB.z;
          float* base = tile_lds + par * KP_ZA_FLOATS;
          float* wz = base + ((n & 1) * 32 + half) * 20 + (n >> 1);                // + 2r*20
          float* wa = base + 64 * 20 + ((n & 1) * 32 + half) * 20 + (n >> 1);
#pragma unroll
          for (int r = 0; r < 16; ++r) { wz[r * 40] = yst[r]; wa[r * 40] = ast[r]; }
          kp_stage_barrier();
          par ^= 1;
          rs = rs == 2 ? 0 : rs + 1;
        }

        // ---- J = sum_c dX_c W_c one row (hidden unit h) at a time: 4 MFMAs leave J[h][k] of this lane's series in
        // the lane, k = the 16 units it owns; f_h += J[h][.] . z and va += a_h J[h][.] follow on the vector pipe
        // (K3j's code: rows issued one ahead of their use, explicit wait states after the last issue).
        f32x16 f, va;
        {
          f32x2 va2[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) va2[j] = f32x2{0.f, 0.f};
          int opaque = 0;                               // the image reads are loop invariant: keep them inside the stage
          asm volatile("" : "+v"(opaque));
          const float4* wp = wj + lane + opaque;
          auto issue = [&](f32x16& J, const float4& a) {
            __builtin_amdgcn_sched_barrier(0);           // everything that still reads the old J stays above
            asm volatile("s_nop 1\n\t"                                       // (operands may be fresh VALU results)
                         "v_mfma_f32_32x32x2_f32 %0, %1, %5, 0\n\t"
                         "v_mfma_f32_32x32x2_f32 %0, %2, %6, %0\n\t"
                         "v_mfma_f32_32x32x2_f32 %0, %3, %7, %0\n\t"
                         "v_mfma_f32_32x32x2_f32 %0, %4, %8, %0"
                         : "=&v"(J) : "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(bs0), "v"(bs1), "v"(bs2), "v"(bs3));
            __builtin_amdgcn_sched_barrier(0);
          };
          auto consume = [&](const f32x16& J, float ah) {
            const f32x2 ah2 = {ah, ah};
            f32x2 p2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const f32x2 m = {J[2 * j], J[2 * j + 1]};
              if (j & 1) q2 = __builtin_elementwise_fma(m, f32x2{yst[2 * j], yst[2 * j + 1]}, q2);
              else p2 = __builtin_elementwise_fma(m, f32x2{yst[2 * j], yst[2 * j + 1]}, p2);
              va2[j] = __builtin_elementwise_fma(m, ah2, va2[j]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(va2[j]));
            p2 = p2 + q2;
            return p2[0] + p2[1];
          };
          f32x16 Je, Jo;                                 // rows 2r / 2r + 1 in flight
          float4 a_cur = wp[0], a_nxt = wp[64];
          issue(Je, a_cur);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            a_cur = wp[(2 * r + 2) * 64];                // image of row 2r + 2 (r = 15: the bias rows)
            issue(Jo, a_nxt);
            asm volatile("" : "+v"(Je));
            float ae = ast[r], ao = ast[r];
            kp_swap32(ae, ao);
            float X = consume(Je, ae);
            if (r < 15) a_nxt = wp[(2 * r + 3) * 64];
            issue(Je, a_cur);
            asm volatile("" : "+v"(Jo));
            float Y = consume(Jo, ao);
            kp_swap32(X, Y);                             // ... and each half-lane collects the f of the unit it owns
            f[r] = X + Y;
          }
          // Je: (b dX)_h for the units this lane owns.  16-pass MFMA result -> VALU read: wait states
          asm volatile("s_nop 15\n\ts_nop 7" : "+v"(Je));
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            f[r] += Je[r];
            va[r] = va2[r >> 1][r & 1];
          }
        }

        // ---- reverse-time dynamics: dy/ds = -f, da/ds = +a^T df/dz.  3/8 rule in two slots per variable
        const f32x16 ky = -f, ka = va;
        const float third = (float)(1.0 / 3.0);
        if constexpr (METHOD == CDE_METHOD_EULER) {                    // y1 = y0 + ds * k
          yst = y0 + ds * ky;
          ast = a0 + ds * ka;
        } else if constexpr (METHOD == CDE_METHOD_MIDPOINT) {          // mid = y0 + k1 * half_ds; y1 = y0 + ds * k(mid)
          const float half_ds = 0.5f * ds;
          if (stage == 0) { yst = y0 + ky * half_ds; ast = a0 + ka * half_ds; }
          else { yst = y0 + ds * ky; ast = a0 + ds * ka; }
        } else if (stage == 0) {
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
      }
      y0 = yst; a0 = ast;
    }
    // torchdiffeq adjoint: re-seed y from the stored forward value, add the incoming gradient
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
}


// ------------------------------------------------------------------------------------------ K3d as a pair (adjoint=False)
// rk4_backprop.hip's reverse-mode sweep (identity-activation affine field) with the same cut: the chain wave reads the stored
// stage state s_i, forms the 32 Jacobian rows, v_i = J_i^T kb_i on the vector pipe and the kb bookkeeping; the helper wave
// runs dL/dW += (kb_i (x) dX_i)^T s_i one stage behind and the control derivative two stages ahead (the FORWARD stage table,
// walked backwards; one ring: the products carry no quadrature weight).  Same arithmetic as rk4_backprop_jacobian.
template <int DEGREE>
__global__ __launch_bounds__(512, 2) void rk4_backprop_jacobian_pair(
    const float* __restrict__ coeffs, const float* __restrict__ knots, int64_t n_intervals,
    const float* __restrict__ W, const float* __restrict__ stages, const float* __restrict__ grad_out, int64_t n_out,
    const float* __restrict__ step_dt, int64_t n_steps, const int64_t* __restrict__ node_ptr,
    const int64_t* __restrict__ node_out, const float* __restrict__ node_weight, float* __restrict__ grad_z0,
    float* __restrict__ partial, int64_t B, const int64_t* __restrict__ stage_index,
    const float* __restrict__ stage_frac, Dims dims) {
  const int Hr = dims.H, Cr = dims.C;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  for (int e = threadIdx.x; e < KP_WJ_FLOATS; e += 512) lds[e] = kp_wj_image(W, W, e >> 8, e & 3, (e >> 2) & 63, dims);   // (row 32 unused)
  const float4* wj = reinterpret_cast<const float4*>(lds);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int slot = wave & 3;
  const bool helper = wave >= 4;
  const int n = lane & 31, half = lane >> 5;
  float* tile_lds = lds + KP_WJ_FLOATS + slot * KP_TILE_FLOATS;
  float* ring_dx = tile_lds + 2 * KP_ZA_FLOATS;                      // + (stage counter % 3) * 256: dX_c of series n at [n*8 + c]
  const int64_t tile = (int64_t)blockIdx.x * 4 + slot;
  const bool live = tile * 32 < B;
  const int64_t series = tile * 32 + n;
  const bool valid = series < B;
  const int64_t sc = valid ? series : B - 1;
  const int64_t n_stages = 4 * n_steps;

  if (helper) {
    f32x16 accW[MC];
    f32x2 gbp[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
#pragma unroll
    for (int c = 0; c < MC; ++c) {
#pragma unroll
      for (int r = 0; r < 16; ++r) accW[c][r] = 0.f;
    }
    // cursor over the forward table's entries, last first; the interval index of the entry after the fetched one is requested
    // one iteration before its row (as in the adjoint's helper)
    int64_t cur_e = n_stages;                                       // entry whose data `fetch` takes next is cur_e - 1
    Row<DEGREE> row;
    int64_t row_idx = -1, idx_ahead = -1;
    float p_frac = 0.f, p_width = 1.f;
    bool p_ok = false;
    auto fetch = [&]() {
      --cur_e;
      p_ok = cur_e >= 0;
      if (p_ok) {
        if (idx_ahead != row_idx) { row = load_row<DEGREE>(coeffs, sc, n_intervals, idx_ahead, Cr); row_idx = idx_ahead; }
        p_frac = stage_frac[cur_e];
        if (DEGREE == CDE_PATH_LINEAR) p_width = knots[idx_ahead + 1] - knots[idx_ahead];
      }
      if (cur_e >= 1) idx_ahead = stage_index[cur_e - 1];
    };
    auto publish = [&](int which) {
      if (!p_ok) return;
      float dX[MC];
      control_slope<DEGREE>(row, p_frac, p_width, dX);
      const f32x2 d0 = half ? f32x2{dX[4], dX[5]} : f32x2{dX[0], dX[1]}, d1 = half ? f32x2{dX[6], dX[7]} : f32x2{dX[2], dX[3]};
      *reinterpret_cast<float4*>(ring_dx + which * 256 + n * 8 + 4 * half) = make_float4(d0[0], d0[1], d1[0], d1[1]);
    };
    if (n_stages > 0) idx_ahead = stage_index[n_stages - 1];
    fetch(); publish(0);
    fetch(); publish(1);
    fetch();
    __syncthreads();
    int par = 0, rs = 0;
    for (int64_t st = 0; st < n_stages; ++st) {
      kp_stage_barrier();
      // dL/dW of stage st first (its ring slot is rs), THEN stage st + 2's derivative into slot (st + 2) % 3 -- one ring
      // serves the chain wave and this wave, so the slot this stage reads must not be the one written: (st + 2) % 3 != st % 3
      publish(rs >= 1 ? rs - 1 : 2);
      fetch();
      const float* base = tile_lds + par * KP_ZA_FLOATS;
      const float4* zt4 = reinterpret_cast<const float4*>(base + (half * 32 + n) * 20);
      const float4* at4 = reinterpret_cast<const float4*>(base + 64 * 20 + (half * 32 + n) * 20);
      const float4* dw4 = reinterpret_cast<const float4*>(ring_dx + rs * 256 + half * 8);
      float4 zq = zt4[0], aq = at4[0], e0 = dw4[0], e1 = dw4[1];
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) {
        const int i = s2 & 3;
        const f32x2 e01 = {e0.x, e0.y}, e23 = {e0.z, e0.w}, e45 = {e1.x, e1.y}, e67 = {e1.z, e1.w};
        const f32x2 asrc = i < 2 ? f32x2{aq.x, aq.y} : f32x2{aq.z, aq.w};
        const float zb = i == 0 ? zq.x : i == 1 ? zq.y : i == 2 ? zq.z : zq.w;
        f32x2 v01, v23, v45, v67;
        if (i & 1) {
          v01 = pk_mul_hi(e01, asrc); v23 = pk_mul_hi(e23, asrc); v45 = pk_mul_hi(e45, asrc); v67 = pk_mul_hi(e67, asrc);
          pk_fma_hi(gbp[0], e01, asrc); pk_fma_hi(gbp[1], e23, asrc); pk_fma_hi(gbp[2], e45, asrc); pk_fma_hi(gbp[3], e67, asrc);
        } else {
          v01 = pk_mul_lo(e01, asrc); v23 = pk_mul_lo(e23, asrc); v45 = pk_mul_lo(e45, asrc); v67 = pk_mul_lo(e67, asrc);
          pk_fma_lo(gbp[0], e01, asrc); pk_fma_lo(gbp[1], e23, asrc); pk_fma_lo(gbp[2], e45, asrc); pk_fma_lo(gbp[3], e67, asrc);
        }
        if (s2 < 15) {
          e0 = dw4[(s2 + 1) * 4]; e1 = dw4[(s2 + 1) * 4 + 1];
          if (i == 3) { zq = zt4[(s2 + 1) >> 2]; aq = at4[(s2 + 1) >> 2]; }
        }
        __builtin_amdgcn_sched_barrier(0);
        accW[0] = mfma(v01[0], zb, accW[0]); accW[1] = mfma(v01[1], zb, accW[1]);
        accW[2] = mfma(v23[0], zb, accW[2]); accW[3] = mfma(v23[1], zb, accW[3]);
        accW[4] = mfma(v45[0], zb, accW[4]); accW[5] = mfma(v45[1], zb, accW[5]);
        accW[6] = mfma(v67[0], zb, accW[6]); accW[7] = mfma(v67[1], zb, accW[7]);
        __builtin_amdgcn_sched_barrier(0);
      }
      par ^= 1;
      rs = rs == 2 ? 0 : rs + 1;
    }
    if (live) {
      float* my_partial = partial + tile * KP_PARTIAL_FLOATS;
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
    return;
  }

  // -------------------------------------------------------------------------------------------- chain wave
  __builtin_amdgcn_s_setprio(3);
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
  const float* srow = stages + (sc * n_steps * 4) * 32 + half * 16;
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
  f32x16 snext = gy;
  if (n_steps > 0) snext = load_state(4 * n_steps - 1);
  __syncthreads();
  int par = 0, rs = 0;
  for (int64_t k = n_steps - 1; k >= 0; --k) {
    const float dt = step_dt[k];
    const float third = (float)(1.0 / 3.0);
    const float c8 = dt * 0.125f, dt3 = dt * third;
    f32x16 kb1 = gy * c8, kb2 = gy * (3.f * c8), kb3 = kb2, kbc = kb1;
    f32x16 yb = gy;
#pragma unroll
    for (int stage = 3; stage >= 0; --stage) {
      const f32x16 sst = snext;
      const int64_t e_next = 4 * k + stage - 1;
      if (e_next >= 0) snext = load_state(e_next);
      float bs0, bs1, bs2, bs3;
      {
        const float4 dA = *reinterpret_cast<const float4*>(ring_dx + rs * 256 + n * 8);
        const float4 dB = *reinterpret_cast<const float4*>(ring_dx + rs * 256 + n * 8 + 4);
        bs0 = half ? dA.y : dA.x; bs1 = half ? dA.w : dA.z; bs2 = half ? dB.y : dB.x; bs3 = half ? dB.w : dB.z;
        float* base = tile_lds + par * KP_ZA_FLOATS;
        float* wz = base + ((n & 1) * 32 + half) * 20 + (n >> 1);
        float* wa = base + 64 * 20 + ((n & 1) * 32 + half) * 20 + (n >> 1);
#pragma unroll
        for (int r = 0; r < 16; ++r) { wz[r * 40] = sst[r]; wa[r * 40] = kbc[r]; }
        kp_stage_barrier();
        par ^= 1;
        rs = rs == 2 ? 0 : rs + 1;
      }
      f32x16 v;
      {
        f32x2 v2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v2[j] = f32x2{0.f, 0.f};
        int opaque = 0;
        asm volatile("" : "+v"(opaque));
        const float4* wp = wj + lane + opaque;
        auto issue = [&](f32x16& J, const float4& a) {
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("s_nop 1\n\t"
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
          for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(v2[j]));
        };
        f32x16 Je, Jo;
        float4 a_cur = wp[0], a_nxt = wp[64];
        issue(Je, a_cur);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (r < 15) a_cur = wp[(2 * r + 2) * 64];
          issue(Jo, a_nxt);
          asm volatile("" : "+v"(Je));
          float ae = kbc[r], ao = kbc[r];
          kp_swap32(ae, ao);
          consume(Je, ae);
          if (r < 15) {
            a_nxt = wp[(2 * r + 3) * 64];
            issue(Je, a_cur);
            asm volatile("" : "+v"(Jo));
          } else {
            asm volatile("s_nop 15\n\ts_nop 7" : "+v"(Jo));
          }
          consume(Jo, ao);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = v2[r >> 1][r & 1];
      }
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
    }
    gy = yb;
    add_outputs(k, gy);
  }
  if (valid) {
#pragma unroll
    for (int r = 0; r < 16; ++r) if (2 * r + half < Hr) grad_z0[series * Hr + 2 * r + half] = gy[r];
  }
}

}  // namespace

// K3d's pair form (identity activation); CDE_K3D_WAVES=1 keeps the one-wave kernel of rk4_backprop.hip
int launch_backprop_jacobian_pair(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                                  const void* stages, const void* grad_out, int64_t n_out, const float* step_dt,
                                  int64_t n_steps, const int64_t* node_ptr, const int64_t* node_out, const float* node_weight,
                                  void* grad_z0, void* grad_W, void* grad_b, int64_t B, int64_t C, int64_t H,
                                  const int64_t* stage_index, const float* stage_frac, float* partial, hipStream_t s) {
  const Dims dims{(int)H, (int)C};
  const unsigned blocks = (unsigned)((B + 127) / 128);
  const size_t lds = (size_t)KP_LDS_FLOATS * sizeof(float);
#define CDE_BPP(D)                                                                                                   \
  do {                                                                                                               \
    (void)hipFuncSetAttribute((const void*)rk4_backprop_jacobian_pair<D>, hipFuncAttributeMaxDynamicSharedMemorySize,\
                              (int)lds);                                                                             \
    rk4_backprop_jacobian_pair<D><<<blocks, 512, lds, s>>>(                                                          \
        (const float*)coeffs, (const float*)knots, n_intervals, (const float*)W, (const float*)stages,               \
        (const float*)grad_out, n_out, step_dt, n_steps, node_ptr, node_out, node_weight, (float*)grad_z0, partial,  \
        B, stage_index, stage_frac, dims);                                                                           \
  } while (0)
  if (degree == CDE_PATH_CUBIC) CDE_BPP(CDE_PATH_CUBIC);
  else if (degree == CDE_PATH_LINEAR) CDE_BPP(CDE_PATH_LINEAR);
  else return CDE_ERR_UNSUPPORTED;
#undef CDE_BPP
  const int rc = check_launch();
  if (rc != CDE_OK) return rc;
  return launch_reduce_partials(partial, (B + 31) / 32, grad_W, grad_b, (int)H, (int)C, s);
}


template <typename TT>
int launch_adjoint_jacobian_pair(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                                 const void* bias, const void* z_saved, const void* grad_out, const void* sgrid,
                                 const int64_t* seg_off, int64_t n_out, void* grad_z0, void* grad_W, void* grad_b, int64_t B,
                                 int64_t C, int64_t H, const int64_t* stage_index, const void* stage_frac, float* partial,
                                 hipStream_t s, int method) {
  const Dims dims{(int)H, (int)C};
  const unsigned blocks = (unsigned)((B + 127) / 128);
  const size_t lds = (size_t)KP_LDS_FLOATS * sizeof(float);
  // bit 0: priority 3 for the chain waves, bit 1: for the helper waves (the trace build reads CDE_K3P_FLAGS once: experiments)
#ifdef CDE_PHASE_TRACE
  static const int flags = [] { const char* e = getenv("CDE_K3P_FLAGS"); return e ? atoi(e) : 1; }();
#else
  constexpr int flags = 1;
#endif
#define CDE_ADJ_P(D, M)                                                                                              \
  do {                                                                                                               \
    (void)hipFuncSetAttribute((const void*)rk4_adjoint_jacobian_pair<TT, D, M>,                                      \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                 \
    rk4_adjoint_jacobian_pair<TT, D, M><<<blocks, 512, lds, s>>>(                                                    \
        (const float*)coeffs, (const float*)knots, n_intervals, (const float*)W, (const float*)bias,                 \
        (const float*)z_saved, (const float*)grad_out, (const TT*)sgrid, seg_off, n_out, (float*)grad_z0, partial,   \
        B, stage_index, (const float*)stage_frac, dims, flags);                                                      \
  } while (0)
#define CDE_ADJ_PD(M)                                                                                                \
  do {                                                                                                               \
    if (degree == CDE_PATH_CUBIC) CDE_ADJ_P(CDE_PATH_CUBIC, M); else CDE_ADJ_P(CDE_PATH_LINEAR, M);                   \
  } while (0)
  if (degree != CDE_PATH_CUBIC && degree != CDE_PATH_LINEAR) return CDE_ERR_UNSUPPORTED;
  if (method == CDE_METHOD_RK4) CDE_ADJ_PD(CDE_METHOD_RK4);
  else if (method == CDE_METHOD_MIDPOINT) CDE_ADJ_PD(CDE_METHOD_MIDPOINT);
  else if (method == CDE_METHOD_EULER) CDE_ADJ_PD(CDE_METHOD_EULER);
  else return CDE_ERR_UNSUPPORTED;
#undef CDE_ADJ_PD
#undef CDE_ADJ_P
  const int rc = check_launch();
  if (rc != CDE_OK) return rc;
  return launch_reduce_partials(partial, (B + 31) / 32, grad_W, grad_b, (int)H, (int)C, s);
}
template int launch_adjoint_jacobian_pair<float>(const void*, const void*, int64_t, int, const void*, const void*, const void*,
                                                 const void*, const void*, const int64_t*, int64_t, void*, void*, void*,
                                                 int64_t, int64_t, int64_t, const int64_t*, const void*, float*, hipStream_t, int);
template int launch_adjoint_jacobian_pair<double>(const void*, const void*, int64_t, int, const void*, const void*, const void*,
                                                  const void*, const void*, const int64_t*, int64_t, void*, void*, void*,
                                                  int64_t, int64_t, int64_t, const int64_t*, const void*, float*, hipStream_t, int);

}  // namespace cde
