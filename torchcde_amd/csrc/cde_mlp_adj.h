// cde_mlp_adj.h -- what the two adjoint kernels of the two-layer field share (K3m: rk4_mlp_adjoint.hip, fixed grid;
// K4am: dopri5_mlp_adjoint.hip, adaptive): the LDS / image layout of rk4_mlp_adjoint.hip (see that file's header for the
// bank-conflict analysis of the plain W2 copy), the factor-row layout, and the stage evaluation as a function.
#pragma once
#include "cde_mfma.h"
#include <type_traits>

namespace cde {

constexpr int W2P_STRIDE = 132;                           // floats per row of the plain W2 copy
constexpr int W2P_FLOATS = 256 * W2P_STRIDE;
constexpr int W1T_FLOATS = 2 * 8 * 64 * 4;                // [tile][T1][lane][4]
constexpr int ADJ_LDS_FLOATS = W1M_FLOATS + B1M_FLOATS + W2P_FLOATS + BY_FLOATS;   // [W1 image | b1 | W2 plain | b2]
constexpr int MLP_ADJ_IMAGE_FLOATS = ADJ_LDS_FLOATS + W1T_FLOATS;
// 32 hidden units x 16 channels (round 6): behind the images, a zero-padded plain copy of the output layer's UPPER rows --
// hidden units 16..31 as [(h - 16) * 16 + c][128] (and their biases, [(h - 16) * 16 + c]) -- which the sweeps read from
// global memory / L2 for unit groups 4..7 (cde_mfma.h: MlpHi with h0 = 16): both access patterns of the plain LDS copy
// (four consecutive columns of a row for Y2, one column of a row for gu) are plain loads from it.
// A second copy serves the gu += W2^T dL/dY2 products: lane (n, q) needs columns n, 16 + n, .., 112 + n of row (4P + q, c) --
// eight scalar loads per row from the plain copy, two float4 from this one, [row][8 n + T1] = W2[row][16 T1 + n]
// (config 5 at hidden size 32, rk4 forward + adjoint on 32768 series: 128 -> 66 ms per step with it).
constexpr int MLP_ADJ_HI_BIAS_FLOATS = 256, MLP_ADJ_HI_W2_FLOATS = 256 * 128;
constexpr int MLP_ADJ_IMAGE_HI_FLOATS = MLP_ADJ_IMAGE_FLOATS + MLP_ADJ_HI_BIAS_FLOATS + 2 * MLP_ADJ_HI_W2_FLOATS;
__device__ __forceinline__ MlpHi mlp_adj_hi(const float* img, int H, int CT) {
  if (CT != 16 || H <= 16) return MlpHi{};
  const float* w2 = img + MLP_ADJ_IMAGE_FLOATS + MLP_ADJ_HI_BIAS_FLOATS;
  return MlpHi{w2, img + MLP_ADJ_IMAGE_FLOATS, 32, 16, 128, 16, w2 + MLP_ADJ_HI_W2_FLOATS};
}
// h3 = h&3 enters bit-reversed so that the four lane quarters of a gu read are shifted by 0, 16, 8, 24 banks: the LDS
// serves a b32 read in two half-waves (lanes 0-31 = quarters 0,1; lanes 32-63 = quarters 2,3) and each half must
// cover 32 distinct banks.  (With shifts 0, 8, 16, 24 rocprofv3 counted 1.3e8 SQ_LDS_BANK_CONFLICT cycles per launch.)
__host__ __device__ constexpr int w2p_residue(int h3, int c3) { return (2 * (((h3 & 1) << 1) | (h3 >> 1)) + c3) & 7; }
constexpr int U_COLS = 132, G2_COLS = 256, G1_COLS = 128, Z_COLS = 36;

__device__ __forceinline__ void stream_store4(float* p, float a, float b, float c, float d) {
  // written once, read once by the GEMM much later: keep it out of the way of the L2-resident weight images
  __builtin_nontemporal_store(f32x4{a, b, c, d}, reinterpret_cast<f32x4*>(p));
}
__device__ __forceinline__ void plain_store4(float* p, float a, float b, float c, float d) {
  // small batches (mlp_adjoint_eval_split8): the rows are read back a few microseconds later by the fused reduction -- from L2
  *reinterpret_cast<f32x4*>(p) = f32x4{a, b, c, d};
}

// One evaluation of the augmented dynamics of the two-layer field for the 16 series of a wave (lane (n, q) owns hidden
// units q, 4+q, .., 28+q of z and a): f = F(z) dX, va = a^T dF/dz dX, and -- when `stream` -- the UNWEIGHTED factors of
// the parameter gradients of this evaluation, one row per series:
//     U  [132]  relu(W1 z + b1) | 1 | 0 0 0        G2 [256]  dL/dY2 = a_h dX_c act'(Y2)   (padded (h, c) layout)
//     Z  [36]   z | 1 | 0 0 0                      G1 [128]  dL/dY1
// (the "1" columns are written once by the host; dW2 | db2 = G2^T U, dW1 | db1 = G1^T Z).  The body is K3m's (same
// MFMA order, same LDS reads); TGRAD adds kt = a . (F(z) d2X/dt2), the slope of vjp_t (cde_dopri_adj.h).
// SPLIT (K4am on small batches): the four waves of a workgroup evaluate the SAME 16 series.  Everything cheap is done by
// all of them redundantly and bit-identically (layer 1, dL/dY1, va, the RK bookkeeping around this call); the expensive
// middle -- layer 2, its activation, dL/dY2 and gu += W2^T dL/dY2 -- is split by unit group: wave `pw` takes the groups
// P with P / (NP / 4) == pw, streams their G2 rows, and the four partial gu (and the f / kt entries each wave produced)
// are added up in a fixed wave order through a small LDS window `xbuf` (4 waves x 64 lanes x 9 floats, five rounds), so
// every wave continues with the same numbers.  384 instead of 1152 MFMAs per wave and evaluation.
//     xchg: all-reduce 8 (+1) floats per lane over the workgroup's four waves
__device__ __forceinline__ void mlp_split_allreduce(float* xbuf, int pw, int lane, f32x4& a, f32x4& b, float* extra) {
  float* slot = xbuf + (pw * 64 + lane) * 9;
  slot[0] = a[0]; slot[1] = a[1]; slot[2] = a[2]; slot[3] = a[3];
  slot[4] = b[0]; slot[5] = b[1]; slot[6] = b[2]; slot[7] = b[3];
  if (extra) slot[8] = *extra;
  __syncthreads();
  float s[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) s[i] = 0.f;
#pragma unroll
  for (int w = 0; w < 4; ++w) {                                     // the same order in every wave
    const float* src = xbuf + (w * 64 + lane) * 9;
#pragma unroll
    for (int i = 0; i < 8; ++i) s[i] += src[i];
    if (extra) s[8] += src[8];
  }
  a = f32x4{s[0], s[1], s[2], s[3]};
  b = f32x4{s[4], s[5], s[6], s[7]};
  if (extra) *extra = s[8];
  __syncthreads();                                                  // the window is free again
}

// DCTRL (control gradients through K4am, round 6): also gxo[c] = sum_h a_h act(Y2)_hc = d(a.f)/d(dX_c) of the lane's SERIES,
// complete in every lane (the lane's own hidden units, the other waves' unit groups in the SPLIT form, the four lane quarters).
template <int ACT, int CT, bool TGRAD, bool SPLIT = false, bool DCTRL = false, bool HI = false>
__device__ __forceinline__ void mlp_adjoint_eval(const float* lds_base, const float4* w1t_base, int lane, int n, int q,
                                                 int w2y_off, const int (&w2g_off)[4], const float (&zs)[8],
                                                 const float (&as)[8], const float (&dX)[CT], const float (&d2X)[CT],
                                                 bool stream, float* urow, float* zrow, float* g2row, float* g1row, int Hr,
                                                 f32x4& fa, f32x4& fb, f32x4& va, f32x4& vb, float& kt, int pw = 0,
                                                 float* xbuf = nullptr, const float4* w1t_regs = nullptr,
                                                 bool stamp_on = false, unsigned long long* stamp = nullptr,
                                                 float* gxo = nullptr, MlpHi hi = MlpHi{}, float* g2row_hi = nullptr) {
#ifdef CDE_PHASE_TRACE
#define CDE_EVAL_STAMP(slot, ...) do { if (stamp_on) { asm volatile("s_nop 0" : __VA_ARGS__); __builtin_amdgcn_sched_barrier(0); \
                                       stamp[slot] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define CDE_EVAL_STAMP(slot, ...) do { (void)stamp; (void)stamp_on; } while (0)
#endif
  constexpr int NB = CT / 4, NP = 16 / NB;      // channel blocks per unit group, unit groups (of 4 hidden units)
  // 32 units x 16 channels (`hi`, round 6): unit groups NP .. 7 as well, their rows of W2 / b2 from the zero-padded copy behind
  // the images (mlp_adj_hi), their dL/dY2 rows into a second row block (`g2row_hi`: the same padded layout, units 16..31)
  // (HI is a template flag: the kernels of the other shapes compile exactly as before)
  static_assert(!HI || CT == 16, "the upper half exists for 16-channel tiles");
  constexpr int NPX = HI ? 8 : NP;
  constexpr bool has_hi = HI;
  const bool writer = !SPLIT || pw == 0;        // the wave that streams the factor rows every wave holds (U, Z, G1)
  int opaque = 0;                                                // keeps the LDS reads inside the evaluation
  asm volatile("" : "+v"(opaque));
  const float4* w1 = reinterpret_cast<const float4*>(lds_base) + lane + opaque;
  const float4* bb1 = reinterpret_cast<const float4*>(lds_base + W1M_FLOATS) + q + opaque;
  const float* w2p = lds_base + W1M_FLOATS + B1M_FLOATS + opaque;
  const float4* bb2 = reinterpret_cast<const float4*>(lds_base + W1M_FLOATS + B1M_FLOATS + W2P_FLOATS) + q + opaque;
  const float* w2y = w2p + w2y_off;
  const float* w2g[4] = {w2p + w2g_off[0], w2p + w2g_off[1], w2p + w2g_off[2], w2p + w2g_off[3]};
  const float4* w1t = w1t_base + opaque;                         // L2-resident image: same trick against LICM
  // (SPLIT: one wave per SIMD owns the whole register file, so the caller keeps the 16 float4 of the W1^T image in
  //  registers for the launch -- `w1t_regs`, indexed by compile-time constants only -- instead of 16 L2 loads per call)

  // ---- layer 1: u = relu(W1 z + b1); `mask` bit s2 = (pre-activation of the lane's s2-th hidden unit > 0)
  float u[32];
  unsigned mask = 0;
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
    for (int r = 0; r < 4; ++r) {
      u[8 * TP + r] = fmaxf(y0[r], 0.f);
      u[8 * TP + 4 + r] = fmaxf(y1[r], 0.f);
      mask |= (y0[r] > 0.f ? 1u : 0u) << (8 * TP + r);
      mask |= (y1[r] > 0.f ? 1u : 0u) << (8 * TP + 4 + r);
    }
  }
  if (stream && writer) {
#pragma unroll
    for (int T1 = 0; T1 < 8; ++T1) stream_store4(urow + 16 * T1, u[4 * T1], u[4 * T1 + 1], u[4 * T1 + 2], u[4 * T1 + 3]);
#pragma unroll
    for (int m = 0; m < 8; ++m) if (4 * m + q < Hr) zrow[4 * m + q] = zs[m];
  }

  CDE_EVAL_STAMP(0, "+v"(u[0]), "+v"(u[31]));
  __builtin_amdgcn_sched_barrier(0);
  // ---- layer 2, activation, contraction, dL/dY2, and gu += W2^T dL/dY2, one tile pair at a time
  f32x4 gu[8];
#pragma unroll
  for (int T1 = 0; T1 < 8; ++T1) gu[T1] = f32x4{0.f, 0.f, 0.f, 0.f};
  fa = f32x4{0.f, 0.f, 0.f, 0.f}; fb = fa;
  kt = 0.f;
  float gxl[DCTRL ? CT : 1];
#pragma unroll
  for (int c = 0; c < (DCTRL ? CT : 1); ++c) gxl[c] = 0.f;
  // SPLIT: a REAL loop over the wave's unit groups (round 4; the rolled stage loop of the caller then is ~10 KB of code).
  // Every address below is affine in P; only as[P] and the slot of f need a select chain on the (wave-uniform) P.
  auto group = [&](int P, auto upper_c) {                        // unit group P: 4 hidden units x CT channels = NB tiles
    float as_P = as[0];
#pragma unroll
    for (int kk = 1; kk < NPX; ++kk) as_P = P == kk ? as[kk] : as_P;
    constexpr bool upper = decltype(upper_c)::value;             // (a compile-time flag: LDS and global reads stay apart)
    f32x4 y[NB];
    const float* tp_[NB];
#pragma unroll
    for (int tb = 0; tb < NB; ++tb) {
      if (!upper) {
        const float4 c0 = bb2[4 * (NB * P + tb)];
        y[tb] = f32x4{c0.x, c0.y, c0.z, c0.w};
        tp_[tb] = w2y + 2 * (NB * P + tb) * 8 * W2P_STRIDE;      // tile T = NB*P + tb: physical rows (2T + hb)*8 + r8
      } else {
        // the lane's D rows are (h = 4P + q, c = 4 tb + r); its A rows (h = 4P + (n >> 2), c = 4 tb + (n & 3))
        const float4 c0 = *reinterpret_cast<const float4*>(hi.b2 + (4 * P + q - 16) * 16 + 4 * tb);
        y[tb] = f32x4{c0.x, c0.y, c0.z, c0.w};
        tp_[tb] = hi.W2 + ((4 * P + (n >> 2) - 16) * 16 + 4 * tb + (n & 3)) * 128 + 4 * q;
      }
    }
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      float4 a[NB];
#pragma unroll
      for (int tb = 0; tb < NB; ++tb) a[tb] = *reinterpret_cast<const float4*>(tp_[tb] + 16 * g);
#pragma unroll
      for (int tb = 0; tb < NB; ++tb) y[tb] = mfma16(a[tb].x, u[4 * g], y[tb]);
#pragma unroll
      for (int tb = 0; tb < NB; ++tb) y[tb] = mfma16(a[tb].y, u[4 * g + 1], y[tb]);
#pragma unroll
      for (int tb = 0; tb < NB; ++tb) y[tb] = mfma16(a[tb].z, u[4 * g + 2], y[tb]);
#pragma unroll
      for (int tb = 0; tb < NB; ++tb) y[tb] = mfma16(a[tb].w, u[4 * g + 3], y[tb]);
    }
    float g2[CT];
    float f = 0.f, h2 = 0.f;
#pragma unroll
    for (int tb = 0; tb < NB; ++tb) {
      const f32x2 t01 = activate2<ACT>(y[tb][0], y[tb][1]), t23 = activate2<ACT>(y[tb][2], y[tb][3]);
      const float tv[4] = {t01[0], t01[1], t23[0], t23[1]};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 4 * tb + r;
        const float t = tv[r];
        f = c == 0 ? t * dX[0] : __builtin_fmaf(t, dX[c], f);
        if (TGRAD) h2 = __builtin_fmaf(t, d2X[c], h2);
        if constexpr (DCTRL) gxl[c] = __builtin_fmaf(as_P, t, gxl[c]);
        const float slope = ACT == CDE_ACT_TANH ? __builtin_fmaf(-t, t, 1.f) : 1.f;
        g2[c] = as_P * (dX[c] * slope);
      }
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { fa[kk] = P == kk ? f : fa[kk]; fb[kk] = P == 4 + kk ? f : fb[kk]; }
    if (TGRAD) kt = __builtin_fmaf(as_P, h2, kt);
    if (stream) {
      float* grow = upper ? g2row_hi + 4 * CT * (P - NP) : g2row + 4 * CT * P;      // rows (h = 4P+q, c = 0..CT-1) of the padded layout
#pragma unroll
      for (int c4 = 0; c4 < CT; c4 += 4) stream_store4(grow + c4, g2[c4], g2[c4 + 1], g2[c4 + 2], g2[c4 + 3]);
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) {                               // K step (P, c); 8 independent accumulator chains
      if constexpr (upper) {
        // row (h = 4P + q, c) of the padded copy, columns 16 T1 + n: two float4 of the copy laid out for this read
        const float4* rowq = reinterpret_cast<const float4*>(hi.W2t + ((4 * P + q - 16) * 16 + c) * 128 + 8 * n);
        const float4 r0 = rowq[0], r1 = rowq[1];
        const float rv[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
        for (int T1 = 0; T1 < 8; ++T1) gu[T1] = mfma16(rv[T1], g2[c], gu[T1]);
      } else {
        const float* rowp = w2g[c & 3] + 2 * (NB * P + (c >> 2)) * 8 * W2P_STRIDE;
#pragma unroll
        for (int T1 = 0; T1 < 8; ++T1) gu[T1] = mfma16(rowp[16 * T1], g2[c], gu[T1]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  if constexpr (SPLIT) {
    constexpr int per_wave = NPX / 4;
    const int P_first = pw * per_wave;
    if (NPX > NP && P_first >= NP) {                             // (wave-uniform: waves 2, 3 take the upper groups)
      if constexpr (NPX > NP) {
#pragma clang loop unroll(disable)
        for (int P = P_first; P < P_first + per_wave; ++P) group(P, std::true_type{});
      }
    } else {
#pragma clang loop unroll(disable)
      for (int P = P_first; P < P_first + per_wave; ++P) group(P, std::false_type{});
    }
  } else {
#pragma unroll
    for (int P = 0; P < NP; ++P) group(P, std::false_type{});
    if constexpr (NPX > NP) {
      // (a ROLLED loop: the one-wave-per-tile kernels unroll their seven stages -- four more unrolled unit groups per stage,
      //  with their global loads hoisted across the stages, spilled over a thousand registers)
#pragma clang loop unroll(disable)
      for (int P = NP; P < NPX; ++P) group(P, std::true_type{});
    }
  }

  CDE_EVAL_STAMP(1, "+v"(gu[0]), "+v"(gu[7]));
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (SPLIT) {
    // ---- the four waves' shares of gu, f and kt meet: afterwards every wave holds the complete values
    mlp_split_allreduce(xbuf, pw, lane, gu[0], gu[1], nullptr);
    mlp_split_allreduce(xbuf, pw, lane, gu[2], gu[3], nullptr);
    mlp_split_allreduce(xbuf, pw, lane, gu[4], gu[5], nullptr);
    mlp_split_allreduce(xbuf, pw, lane, gu[6], gu[7], nullptr);
    mlp_split_allreduce(xbuf, pw, lane, fa, fb, &kt);
    if constexpr (DCTRL) {
#pragma unroll
      for (int c8 = 0; c8 < CT; c8 += 8) {
        f32x4 lo = {gxl[c8], gxl[c8 + 1], gxl[c8 + 2], gxl[c8 + 3]}, hi = {gxl[c8 + 4], gxl[c8 + 5], gxl[c8 + 6], gxl[c8 + 7]};
        mlp_split_allreduce(xbuf, pw, lane, lo, hi, nullptr);
#pragma unroll
        for (int r = 0; r < 4; ++r) { gxl[c8 + r] = lo[r]; gxl[c8 + 4 + r] = hi[r]; }
      }
    }
  }
  if constexpr (DCTRL) {
    // the four lane quarters hold different hidden units of the same series: fixed-order sum, the result in every lane
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      float v = gxl[c];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      gxo[c] = v;
    }
  }
  CDE_EVAL_STAMP(2, "+v"(gu[0]), "+v"(gu[7]));
  // ---- dL/dY1 = gu * relu'(pre1);  va = W1^T dL/dY1
  float g1[32];
#pragma unroll
  for (int s2 = 0; s2 < 32; ++s2) g1[s2] = (mask >> s2) & 1u ? gu[s2 >> 2][s2 & 3] : 0.f;
  if (stream && writer) {
#pragma unroll
    for (int T1 = 0; T1 < 8; ++T1)
      stream_store4(g1row + 16 * T1, g1[4 * T1], g1[4 * T1 + 1], g1[4 * T1 + 2], g1[4 * T1 + 3]);
  }
  va = f32x4{0.f, 0.f, 0.f, 0.f}; vb = va;
#pragma unroll
  for (int T1 = 0; T1 < 8; ++T1) {
    const float4 a0 = SPLIT ? w1t_regs[T1] : w1t[T1 * 64], a1 = SPLIT ? w1t_regs[8 + T1] : w1t[(8 + T1) * 64];
    va = mfma16(a0.x, g1[4 * T1], va);     vb = mfma16(a1.x, g1[4 * T1], vb);
    va = mfma16(a0.y, g1[4 * T1 + 1], va); vb = mfma16(a1.y, g1[4 * T1 + 1], vb);
    va = mfma16(a0.z, g1[4 * T1 + 2], va); vb = mfma16(a1.z, g1[4 * T1 + 2], vb);
    va = mfma16(a0.w, g1[4 * T1 + 3], va); vb = mfma16(a1.w, g1[4 * T1 + 3], vb);
  }
  CDE_EVAL_STAMP(3, "+v"(va), "+v"(vb));
  __builtin_amdgcn_sched_barrier(0);
#undef CDE_EVAL_STAMP
}

// SPLIT8 (round 4; CT = 8): the EIGHT waves of a workgroup (two per SIMD) evaluate the same 16 series and split EVERYTHING:
// wave w owns hidden-layer tile T1 = w (units 16w .. 16w+15) and unit group P = w (hidden units 4w .. 4w+3 of z).
//   layer 1   8 MFMAs  -> u for its 16 units                  -> all-gather of u            (xb, 8 KB, one barrier)
//   layer 2  64 MFMAs  -> Y2 of its group, activation, f_h, dL/dY2 (g2)  -> all-gather of g2   (xa, 16 KB, one barrier)
//   gu       64 MFMAs  -> (W2^T dL/dY2) for ITS 16 units over all 256 rows: complete, nothing to reduce
//   va        8 MFMAs  -> W1^T dL/dY1 over its 16 units: a partial of all 32 outputs -> all-reduce (xa / xb, two barriers)
// and of the adjoint state a wave carries only ITS unit (a_{4w+q}): the a half of the slope ring is 7 registers, not 56.
// 144 MFMAs per wave and evaluation instead of 384 (four waves) / 1152 (one), a quarter of the vector work per wave, and the
// second wave of each SIMD fills the issue slots the first one leaves (a single wave overlaps nothing with its own MFMAs:
// profiles/r01_mfma_issue_ubench.txt; profiles/r04_phase_k4am_*.log: the four-wave form spent 14 us per evaluation on
// 5.1 us of MFMA work).  Every wave ends with the same f, va, kt (fixed summation order), so the RK bookkeeping around
// the call stays redundant and bit-identical.  The W1 / W1^T tiles a wave needs are 4 float4 in registers; the 16 KB the
// W1 image occupied in LDS is the exchange window `xa`.
// NTSTORE / wq: the fixed-grid sweep (rk4_mlp_adjoint.hip) streams quadrature-WEIGHTED dL/dY rows that are read back only
// after a whole chunk of steps (non-temporal stores); the adaptive kernel streams unweighted rows read microseconds later.
template <int ACT, bool TGRAD, bool NTSTORE = false>
__device__ __forceinline__ void mlp_adjoint_eval_split8(const float* lds_base, float* xa, float* xb, int lane, int q, int w,
                                                        const float4 (&w1r)[2], float4 b1r, const float4 (&w1tr)[2],
                                                        int w2y_off, const int (&w2g_off)[4], const float (&zs)[8],
                                                        float as_w, const float (&dX)[8], const float (&d2X)[8],
                                                        bool stream, float* urow, float* zrow, float* g2row, float* g1row,
                                                        int Hr, f32x4& fa, f32x4& fb, float& va_w, float& kt,
                                                        bool stamp_on = false, unsigned long long* stamp = nullptr,
                                                        float wq = 1.f) {
#ifdef CDE_PHASE_TRACE
#define CDE_EVAL_STAMP(slot, ...) do { if (stamp_on) { asm volatile("s_nop 0" : __VA_ARGS__); __builtin_amdgcn_sched_barrier(0); \
                                       stamp[slot] = wall_clock64(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define CDE_EVAL_STAMP(slot, ...) do { (void)stamp; (void)stamp_on; } while (0)
#endif
  constexpr int CT = 8;
  auto store4 = [](float* p, float a, float b, float c, float d) {
    if constexpr (NTSTORE) stream_store4(p, a, b, c, d); else plain_store4(p, a, b, c, d);
  };
  const float* w2p = lds_base + W1M_FLOATS + B1M_FLOATS;
  const float4* bb2 = reinterpret_cast<const float4*>(lds_base + W1M_FLOATS + B1M_FLOATS + W2P_FLOATS) + q;
  // ---- layer 1, tile T1 = w: u = relu(W1 z + b1) for hidden-layer units 16w + 4q + r
  // (ONE accumulator chain, bias first, K steps in order: the bits of every other form of this layer -- forward kernels
  //  included -- so that the relu mask of the backward pass is the forward pass's: with a different summation order a
  //  pre-activation within an ulp of zero flips its mask and that series' gradient jumps)
  f32x4 y1 = {b1r.x, b1r.y, b1r.z, b1r.w};
  {
    const float a0[8] = {w1r[0].x, w1r[0].y, w1r[0].z, w1r[0].w, w1r[1].x, w1r[1].y, w1r[1].z, w1r[1].w};
#pragma unroll
    for (int s = 0; s < 8; ++s) y1 = mfma16(a0[s], zs[s], y1);
  }
  float uo[4];
  unsigned mask = 0;
#pragma unroll
  for (int r = 0; r < 4; ++r) { uo[r] = fmaxf(y1[r], 0.f); mask |= (y1[r] > 0.f ? 1u : 0u) << r; }
  *reinterpret_cast<float4*>(xb + (w * 64 + lane) * 4) = make_float4(uo[0], uo[1], uo[2], uo[3]);
  if (stream) {
    store4(urow + 16 * w, uo[0], uo[1], uo[2], uo[3]);
    if (w == 0) {
#pragma unroll
      for (int m = 0; m < 8; ++m) if (4 * m + q < Hr) zrow[4 * m + q] = zs[m];
    }
  }
  __syncthreads();                                               // u of all 128 units is in xb
  CDE_EVAL_STAMP(0, "+v"(uo[0]));

  // ---- layer 2 for unit group P = w (tiles 2w, 2w+1), activation, contraction with dX, dL/dY2
  f32x4 y[2];
  const float* tp_[2];
#pragma unroll
  for (int tb = 0; tb < 2; ++tb) {
    const float4 c0 = bb2[4 * (2 * w + tb)];
    y[tb] = f32x4{c0.x, c0.y, c0.z, c0.w};
    tp_[tb] = w2p + w2y_off + 2 * (2 * w + tb) * 8 * W2P_STRIDE;
  }
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const float4 u4 = *reinterpret_cast<const float4*>(xb + (g * 64 + lane) * 4);
    const float4 a0 = *reinterpret_cast<const float4*>(tp_[0] + 16 * g), a1 = *reinterpret_cast<const float4*>(tp_[1] + 16 * g);
    y[0] = mfma16(a0.x, u4.x, y[0]); y[1] = mfma16(a1.x, u4.x, y[1]);
    y[0] = mfma16(a0.y, u4.y, y[0]); y[1] = mfma16(a1.y, u4.y, y[1]);
    y[0] = mfma16(a0.z, u4.z, y[0]); y[1] = mfma16(a1.z, u4.z, y[1]);
    y[0] = mfma16(a0.w, u4.w, y[0]); y[1] = mfma16(a1.w, u4.w, y[1]);
  }
  float g2[CT];
  float f = 0.f, h2 = 0.f;
#pragma unroll
  for (int tb = 0; tb < 2; ++tb) {
    const f32x2 t01 = activate2<ACT>(y[tb][0], y[tb][1]), t23 = activate2<ACT>(y[tb][2], y[tb][3]);
    const float tv[4] = {t01[0], t01[1], t23[0], t23[1]};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 4 * tb + r;
      const float t = tv[r];
      f = c == 0 ? t * dX[0] : __builtin_fmaf(t, dX[c], f);
      if (TGRAD) h2 = __builtin_fmaf(t, d2X[c], h2);
      const float slope = ACT == CDE_ACT_TANH ? __builtin_fmaf(-t, t, 1.f) : 1.f;
      g2[c] = as_w * (dX[c] * slope);
    }
  }
  const float kt_part = TGRAD ? as_w * h2 : 0.f;
  *reinterpret_cast<float4*>(xa + ((2 * w + 0) * 64 + lane) * 4) = make_float4(g2[0], g2[1], g2[2], g2[3]);
  *reinterpret_cast<float4*>(xa + ((2 * w + 1) * 64 + lane) * 4) = make_float4(g2[4], g2[5], g2[6], g2[7]);
  if (stream) {
    float* grow = g2row + 4 * CT * w;                            // rows (h = 4w+q, c = 0..7) of the padded layout
    store4(grow, g2[0] * wq, g2[1] * wq, g2[2] * wq, g2[3] * wq);
    store4(grow + 4, g2[4] * wq, g2[5] * wq, g2[6] * wq, g2[7] * wq);
  }
  __syncthreads();                                               // dL/dY2 of all 256 rows is in xa
  CDE_EVAL_STAMP(1, "+v"(g2[0]));

  // ---- gu = W2^T dL/dY2 for this wave's 16 hidden-layer units, over all 256 rows: K step (P', c)
  f32x4 gua = {0.f, 0.f, 0.f, 0.f}, gub = gua;
#pragma unroll
  for (int Pp = 0; Pp < 8; ++Pp) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const float4 g4 = *reinterpret_cast<const float4*>(xa + ((2 * Pp + half) * 64 + lane) * 4);
      const float* rowbase = w2p + 2 * (2 * Pp + half) * 8 * W2P_STRIDE + 16 * w;
      gua = mfma16(rowbase[w2g_off[0]], g4.x, gua);
      gub = mfma16(rowbase[w2g_off[1]], g4.y, gub);
      gua = mfma16(rowbase[w2g_off[2]], g4.z, gua);
      gub = mfma16(rowbase[w2g_off[3]], g4.w, gub);
    }
  }
  const f32x4 gu = gua + gub;
  // ---- dL/dY1 = gu * relu'(pre1) for its units; its share of va = W1^T dL/dY1
  float g1[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) g1[r] = (mask >> r) & 1u ? gu[r] : 0.f;
  if (stream) store4(g1row + 16 * w, g1[0] * wq, g1[1] * wq, g1[2] * wq, g1[3] * wq);
  f32x4 pa = {0.f, 0.f, 0.f, 0.f}, pb = pa;
  pa = mfma16(w1tr[0].x, g1[0], pa); pb = mfma16(w1tr[1].x, g1[0], pb);
  pa = mfma16(w1tr[0].y, g1[1], pa); pb = mfma16(w1tr[1].y, g1[1], pb);
  pa = mfma16(w1tr[0].z, g1[2], pa); pb = mfma16(w1tr[1].z, g1[2], pb);
  pa = mfma16(w1tr[0].w, g1[3], pa); pb = mfma16(w1tr[1].w, g1[3], pb);
  __syncthreads();                                               // every wave is done reading u (xb) and dL/dY2 (xa)
  CDE_EVAL_STAMP(2, "+v"(pa), "+v"(pb));
  *reinterpret_cast<float4*>(xa + ((2 * w + 0) * 64 + lane) * 4) = make_float4(pa[0], pa[1], pa[2], pa[3]);
  *reinterpret_cast<float4*>(xa + ((2 * w + 1) * 64 + lane) * 4) = make_float4(pb[0], pb[1], pb[2], pb[3]);
  *reinterpret_cast<float2*>(xb + (w * 64 + lane) * 2) = make_float2(f, kt_part);
  __syncthreads();
  // this wave's component of va (hidden unit 4w + q: entry w & 3 of half w >> 2), all of f, kt: fixed order over the waves
  va_w = 0.f;
  kt = 0.f;
  float fs[8];
  const float* mine = xa + ((w >> 2) * 64 + lane) * 4 + (w & 3);
#pragma unroll
  for (int ww = 0; ww < 8; ++ww) {
    const float2 fk = *reinterpret_cast<const float2*>(xb + (ww * 64 + lane) * 2);
    va_w += mine[2 * ww * 64 * 4];
    fs[ww] = fk.x;
    kt += fk.y;
  }
  fa = f32x4{fs[0], fs[1], fs[2], fs[3]};
  fb = f32x4{fs[4], fs[5], fs[6], fs[7]};
  __syncthreads();                                               // the windows are free for the next evaluation
  CDE_EVAL_STAMP(3, "+v"(va_w));
#undef CDE_EVAL_STAMP
}

// host side of the images (rk4_mlp_adjoint.hip)
size_t mlp_adjoint_image_bytes();
int launch_mlp_adjoint_images(const void* W1, const void* b1, int64_t width, const void* W2, const void* b2, int64_t C,
                              int64_t H, float* img, hipStream_t s);

}  // namespace cde
