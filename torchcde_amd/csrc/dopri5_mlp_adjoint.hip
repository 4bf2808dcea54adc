// dopri5_mlp_adjoint.hip -- K4am: the continuous-adjoint backward of an ADAPTIVE (dopri5) solve for the two-layer field
//   f(z) = reshape_{HxC}( act( W2 relu(W1 z + b1) + b2 ) ) dX          (example/time_series_classification.py:20-51)
//
// This is the call every example of the reference makes to train its model: cdeint(X, func, z0, X.interval) with no
// method -- dopri5, adjoint=True (example/time_series_classification.py:83-86, example/irregular_data.py:59,
// example/logsignature_example.py via the same CDEFunc; semantics solver.py:144,199-203,226).  Until round 3 it ran
// step-wise (host controller, library GEMMs per evaluation: 34 s per 4096-series backward).
//
// Structure = K4a's (dopri5_adjoint.hip) around K3m's stage evaluation (cde_mlp_adj.h: mlp_adjoint_eval):
//   * one launch of `dopri5_mlp_adjoint_attempt` = "decide the previous attempt, make the next one": the controller of
//     cde_dopri_adj.h (torchdiffeq's default MIXED adjoint norm over (vjp_t, y, a, dW1, db1, dW2, db2), or "seminorm";
//     interval ends passed and interpolated: mode 3 repeats the last step with the dense-output functional);
//   * a wave owns 16 series for the 7 stage evaluations of an attempt (lane (n, q): hidden units q, 4+q, .., 28+q of z and
//     of a).  The stage body already fills the register file (u[32], gu[32], g1[32] on top of the MFMA operands), so the
//     7 x 16 stage slopes of a lane live in a per-wave scratch ring in global memory (written once per stage as four
//     coalesced 16-byte stores, read back by the stage combinations and the error / dense-output sums; 28 KB per wave,
//     L2-resident);
//   * dL/dW2 is 256 x 128 per wave -- no register file holds it -- so, as in K3m, every evaluation streams the
//     UNWEIGHTED factors of the parameter gradients to HBM (2.2 KB per series and stage; stage 1 has weight 0 in every
//     functional and is skipped), one block of rows per stage; the split-K MFMA reduction of mlp_grad_reduce.hip turns each
//     stage's rows into that stage's gradient image K_s (per-slab partials, slabs never straddle stages), and
//     `mlp_adjoint_reduce_kernel` -- the R kernel of this family -- forms S = sum_s wS[s] K_s and E = sum_s wE[s] K_s per
//     element (adj_stage_weights), owns the running totals, commits the accepted attempt's increment and leaves the
//     per-block sums of (E / tol)^2 of the four parameter tensors for the next launch's controller.
//   * FIRST SAME AS LAST (round 4), as torchdiffeq keeps f0 after a rejection and passes f1 on after an accepted step: an
//     attempt that follows an attempt does not evaluate its first stage -- its slopes are kept (ring slot / stash plane), its
//     factor rows stay where they were written (seven blocks of rows: the last stage alternates between blocks 5 and 6,
//     AdjCtrl::src0 / six say which is which) and its image is kept by the R kernel.  Same inputs bit for bit, so traces and
//     gradients are identical with CDE_K4AM_NO_FSAL=1 (tests).
// Three launches per attempted step (attempt, factor reduction of both layers, R; up to 128 series: two), no host round trip.
#include "cde_dopri_adj.h"
#include "cde_dopri_ctl.h"
#include "cde_mlp_adj.h"

namespace cde {

constexpr int MADJ_SLOTS = 6;                    // stages whose factors are kept: 0, 2, 3, 4, 5, 6
constexpr int MADJ_FSLOTS = 7;                   // factor-row blocks: 0 first stage, 1..4 stages 2..5, 5 and 6 the last stage
                                                 // (alternating: the accepted step's last stage IS the next step's first)
constexpr int MADJ_P2 = 256 * 129, MADJ_P1 = 128 * 33;          // elements of a layer-2 / layer-1 slab partial (bias column last)
constexpr int MADJ_ELEMS = MADJ_P2 + MADJ_P1;
constexpr int MADJ_RBLOCKS = (MADJ_ELEMS + 31) / 32;
constexpr int MADJ_MAX_SPS = 40;                 // slabs per stage of the factor reduction (80: 121 -> 133 us per attempt at 4096 series)
constexpr int64_t MADJ_S8_MAX_TILES = 768;       // eight waves per tile (8-channel tiles), several rounds: see madj_layout
constexpr int64_t MADJ_SPLIT_MAX_TILES = 256;    // batches up to 4096 series (one tile per CU): four waves per tile (K4am's split form)
constexpr int MADJ_NSUM = ADJ_NS + 2 * ADJ_MAX_PT;
// without control gradients the mixed norm has the four parameter blocks only: the prologue then carries (and reduces) 16 sums,
// not 20 -- with 20 the eight-wave kernel, which sits at the register limit with its prefetches in flight, ran 5 % slower
// (81.8 -> 86.1 us per attempted step, profiles/r06_bench_kernel_stats.csv against r05's)
constexpr int MADJ_NSUM_PLAIN = ADJ_NS + 8;
__host__ __device__ constexpr int madj_slot(int stage) { return stage == 0 ? 0 : stage - 1; }      // stage 1 is never stored

struct MlpAdjArgs {
  const float* coeffs; const float* knots; int64_t n_intervals;
  const float* img;                 // MLP_ADJ_IMAGE_FLOATS: [LDS image | W1^T image]
  Dims dims;
  int64_t B, n_tiles, rows_per_stage;
  unsigned char* ctrl;              // [2] AdjCtrl, ADJ_CTRL_STRIDE bytes apart
  float* state;                     // [2][4][B*H]: committed y, a; attempted y1, a1
  const float* y_init; const float* a_init;
  float* a_out;
  double* partial;                  // [2][n_wg_max][ADJ_NS]
  double* pq;                       // [2][MADJ_RBLOCKS][8]
  float* slopes;                    // [n_tiles][7][4][64] float4: the stage slopes of every lane
  float* U; float* G2; float* G1; float* Z;      // factor rows [block][rows_per_stage], MADJ_FSLOTS blocks
  float* stash_y; float* stash_a; float* stash_t;  // [3][B*H], [3][B*H], [3][B*4]: slopes of a first / last stage (blocks 0, 5, 6)
  AdjCommon com;
  int n_wg_max;
  const double* ext_sums;           // sharded batch: the ADJ_NS state sums of the pending attempt, added up over ALL shards
  int n_pq;                         // blocks of parameter sums the R kernel of this batch size leaves in `pq`
  int dbg;                          // instrumented (CDE_PHASE_TRACE) builds only: CDE_K4AM_DBG bit 0 = no factor stores
  // DCTRL (control gradients: cde_dopri_ctl.h) -- dopri5_mlp_adjoint_attempt<.., DCTRL = true> only
  float* gx;                        // [2][B][CT][8]: the launch's per-stage d(a.f)/d(dX_c), unweighted
  unsigned char* rec;               // [2] AdjStageRec
  const double* cq;                 // [2][n_cblocks + 1][2]: the control kernel's norm sums (last entry: the knot block)
  int n_cblocks;
  double* ktp;                      // [2][n_wg_max][8]: per-workgroup sums of the per-stage time term
  int with_knots;
  // 32 hidden units x 16 channels (round 6): the dL/dY2 rows of hidden units 16..31, a second block laid out like G2; the
  // parameter sums then come in two runs of blocks (one per R-kernel instance), `pq_blocks` per parity
  float* G2hi;
  int pq_blocks;
};

// SPLIT (small batches: fewer tiles than SIMDs): the workgroup's four waves share ONE tile and split the middle of every
// evaluation between them (cde_mlp_adj.h: mlp_adjoint_eval<..., SPLIT>); each wave keeps its own copy of the slope ring,
// wave 0 alone stores state, streams the shared factor rows and contributes to the error sums.
constexpr int MADJ_XBUF_FLOATS = 4 * 64 * 9;
#ifdef CDE_PHASE_TRACE
__device__ unsigned long long k4am_phase_trace[TRACE_RING * TRACE_BLOCKS * TRACE_SLOTS];
#endif
// HI (round 6): 32 hidden units x 16 channels -- the four-wave form with twice the unit groups (cde_mlp_adj.h)
template <int DEGREE, int ACT, int CT, int NWAVE, bool SPLIT = false, bool DCTRL = false, bool HI = false>
__global__ __launch_bounds__(64 * NWAVE, NWAVE == 8 ? 2 : 1) void dopri5_mlp_adjoint_attempt(MlpAdjArgs g, int parity) {
  static_assert(!SPLIT || NWAVE == 4, "this kernel's split form is four waves per tile (eight: dopri5_mlp_adjoint_attempt_s8)");
  static_assert(!HI || CT == 16, "the upper half: 16-channel tiles");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int p = parity, p2 = parity ^ 1;
  CDE_STAMP_DECL;
  CDE_STAMP(0);
  AdjCtrl k = *reinterpret_cast<const AdjCtrl*>(g.ctrl + p * ADJ_CTRL_STRIDE);
  DopriCtrl& c = k.c;
  if (c.phase == 4) {
    if (blockIdx.x == 0 && tid == 0) { k.commit = 0; k.mode = 3; *reinterpret_cast<AdjCtrl*>(g.ctrl + p2 * ADJ_CTRL_STRIDE) = k; }
    return;
  }
#ifdef CDE_PHASE_TRACE
  const int attempt_no = uni((int)(c.n_accept + c.n_reject));
#endif
  {
    const float4* src = reinterpret_cast<const float4*>(g.img);
    float4* dst = reinterpret_cast<float4*>(lds);
    for (int i = tid; i < ADJ_LDS_FLOATS / 4; i += blockDim.x) dst[i] = src[i];
  }
  double* red = reinterpret_cast<double*>(lds + ADJ_LDS_FLOATS);
  float* xbuf = lds + ADJ_LDS_FLOATS + 2 * MADJ_NSUM * 8;          // (behind the 16 x 8 doubles of `red`)
  const int Hr = g.dims.H, Cr = g.dims.C;
  const int lane = tid & 63, wave = tid >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int64_t BH = g.B * Hr;
  const float* Sp = g.state + (int64_t)p * 4 * BH;
  float* Sq = g.state + (int64_t)p2 * 4 * BH;
  const double* Pp = g.partial + (int64_t)p * g.n_wg_max * ADJ_NS;
  double* Pq = g.partial + (int64_t)p2 * g.n_wg_max * ADJ_NS;
  const float rtol = (float)g.com.rtol, atol = (float)g.com.atol;

  // ---- pending sums: the state sums of the previous attempt launch, the parameter sums its R kernel left
  constexpr int NSUM = DCTRL ? MADJ_NSUM : MADJ_NSUM_PLAIN;
  double sum[NSUM];
#pragma unroll
  for (int i = 0; i < NSUM; ++i) sum[i] = 0.0;
  if (c.phase != 0) {
    for (int64_t b = tid; b < (int64_t)gridDim.x; b += blockDim.x) {
#pragma unroll
      for (int i = 0; i < ADJ_NS; ++i) sum[i] += Pp[ADJ_NS * b + i];
    }
    const double* Qp = g.pq + (int64_t)p * g.pq_blocks * 8;
    for (int b = tid; b < g.n_pq; b += blockDim.x) {
#pragma unroll
      for (int i = 0; i < 8; ++i) sum[ADJ_NS + i] += Qp[8 * b + i];
    }
    if constexpr (DCTRL) {                                         // the control kernel's sums: coefficient block, knot block
      const double* Cp = g.cq + (int64_t)p * (g.n_cblocks + 1) * 2;
      for (int b = tid; b < g.n_cblocks; b += blockDim.x) { sum[ADJ_NS + 8] += Cp[2 * b]; sum[ADJ_NS + 9] += Cp[2 * b + 1]; }
      if (tid == 0) { sum[ADJ_NS + 10] = Cp[2 * g.n_cblocks]; sum[ADJ_NS + 11] = Cp[2 * g.n_cblocks + 1]; }
    }
  }
  if (g.ext_sums && c.phase != 0) {                                // one controller for all shards: the reduced state sums
#pragma unroll
    for (int i = 0; i < ADJ_NS; ++i) sum[i] = tid == 0 ? g.ext_sums[i] : 0.0;
  }
  block_total<NSUM>(sum, red);                                     // (also the barrier after the LDS image copy)
  CDE_STAMP(1);
  const int phase_in = c.phase;
  const AdjPlan plan = adj_controller(g.com, k, sum, sum + ADJ_NS);
  const int mode = uni(plan.mode);
  const bool commit = phase_in == 3 && plan.accept && mode != 3;   // mode 3 repeats the accepted step from ITS start state
  const int ns = mode == 0 ? 1 : mode == 1 ? 2 : 7;

  // ---- stage times (reversed time), their knot intervals, the stage weights: all wave-uniform, in scalar registers
  const float t0f = uni((float)plan.t0), dtf = uni((float)plan.dt), t1f = uni((float)plan.t1);
  int sidx[7];
  float sfrac[7];
  {
    float ts = 0.f;
    const int i = lane & 7;
    if (mode == 0) ts = (float)c.t_hi;
    else if (mode == 1) ts = i == 0 ? (float)c.t_hi : (float)(c.t_hi + (double)plan.h0);
    else if (i == 0) ts = plan.kind0 == 0 ? t0f : next_toward(t0f, plan.kind0 > 0 ? 1.f : -1.f);
    else if (i <= 4) ts = t0f + (float)DP_ALPHA[i - 1] * dtf;
    else ts = next_toward(t1f, -1.f);
    float frac;
    const int idx = (int)locate_around(g.knots, g.n_intervals, -ts, phase_in == 0 ? (int64_t)-1 : (int64_t)c.slot, frac);
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      sidx[j] = __builtin_amdgcn_readlane(idx, j);
      sfrac[j] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(frac), j));
    }
  }
  float bc[7][6];                                                  // bc[i][j]: weight of slope j in the state handed to stage i
#pragma unroll
  for (int i = 0; i < 7; ++i) {
#pragma unroll
    for (int j = 0; j < 6; ++j) bc[i][j] = (i >= 1 && j < i) ? uni((float)DP_BETA[i - 1][j] * dtf) : 0.f;
  }
  if (mode == 1) bc[1][0] = uni(plan.h0);
  float wS[7], wE[7];
  adj_stage_weights(mode, dtf, plan.x_end, wS, wE);
#pragma unroll
  for (int j = 0; j < 7; ++j) { wS[j] = uni(wS[j]); wE[j] = uni(wE[j]); }
  const float x_end = uni(plan.x_end);
  CDE_STAMP(2);
  if constexpr (DCTRL) {
    if (blockIdx.x == 0 && tid == 0) {
      AdjStageRec rc;
      rc.mode = mode; rc.ns = ns;
#pragma unroll
      for (int j = 0; j < 7; ++j) { rc.sidx[j] = sidx[j]; rc.sfrac[j] = sfrac[j]; rc.wS[j] = wS[j]; rc.wE[j] = wE[j]; }
      *reinterpret_cast<AdjStageRec*>(g.rec + p * ADJ_REC_STRIDE) = rc;
    }
  }

  // First same as last: an attempt that follows an attempt starts where that one started (rejected) or ended (accepted, not
  // on a jump).  One wave per tile: the slopes of that stage are in the lane's ring in memory already (slot 0, or slot 6:
  // copied to slot 0); waves sharing a tile (ring in registers): from the three-plane stash of the eight-wave kernel.  The
  // stage's factor rows are in the block the controller names.
  const int in_src0 = uni((int)k.src0), in_six = uni((int)k.six) & 15;
  const bool reuse = phase_in == 3 && mode == 2 && !(plan.accept && c.refresh) && !(g.dbg & 2);
  const int src0 = reuse ? (plan.accept ? in_six : in_src0) : 0;
  const int six = reuse ? (plan.accept ? 11 - in_six : in_six) : 5;
  if (blockIdx.x == 0 && tid == 0) {                               // the controller block for the next launch / the R kernel
    c.phase = mode == 0 ? 1 : mode == 1 ? 2 : mode == 2 ? 3 : 4;
    c.slot = sidx[0];                                              // search hint for the next launch's stage times
    k.src0 = src0; k.six = six | (reuse ? 0 : ADJ_FRESH0);
    *reinterpret_cast<AdjCtrl*>(g.ctrl + p2 * ADJ_CTRL_STRIDE) = k;
  }

  double acc[ADJ_NS] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  double ktd[DCTRL ? 7 : 1] = {};                                  // DCTRL: the per-stage time term of this wave's series
  const int64_t tile = SPLIT ? (int64_t)blockIdx.x : (int64_t)blockIdx.x * NWAVE + wave;
  const int pw = SPLIT ? wave : 0;
  const bool writer = !SPLIT || wave == 0;
  if (tile < g.n_tiles) {
    const int64_t series = tile * 16 + n;
    const bool valid = series < g.B;
    const int64_t sc = valid ? series : g.B - 1;
    const float4* w1t_base = reinterpret_cast<const float4*>(g.img + ADJ_LDS_FLOATS) + lane;
    const int w2y_off = ((n >> 3) * 8 + w2p_residue(n >> 2, n & 3)) * W2P_STRIDE + 4 * q;
    int w2g_off[4];
#pragma unroll
    for (int cl = 0; cl < 4; ++cl) w2g_off[cl] = ((q >> 1) * 8 + w2p_residue(q, cl)) * W2P_STRIDE + n;
    const int ua = q, ub = 16 + q;                                 // this lane's units: q, 4+q, .., 28+q
    const MlpHi hi = HI ? mlp_adj_hi(g.img, Hr, CT) : MlpHi{};
    // the state this launch starts from
    const float* ysrc = phase_in == 0 ? g.y_init : Sp + (commit ? 2 : 0) * BH;
    const float* asrc = phase_in == 0 ? g.a_init : Sp + (commit ? 3 : 1) * BH;
    f32x4 y0a = load_units4<4>(ysrc + sc * Hr, ua, Hr), y0b = load_units4<4>(ysrc + sc * Hr, ub, Hr);
    f32x4 a0a = load_units4<4>(asrc + sc * Hr, ua, Hr), a0b = load_units4<4>(asrc + sc * Hr, ub, Hr);
    if (!valid) { a0a = f32x4{0.f, 0.f, 0.f, 0.f}; a0b = a0a; }   // a == 0 stays 0: padded lanes contribute nothing
    if (valid && mode != 3 && writer) {
      store_units4<4>(Sq + 0 * BH + series * Hr, ua, Hr, y0a); store_units4<4>(Sq + 0 * BH + series * Hr, ub, Hr, y0b);
      store_units4<4>(Sq + 1 * BH + series * Hr, ua, Hr, a0a); store_units4<4>(Sq + 1 * BH + series * Hr, ub, Hr, a0b);
    }
    // the 7 x 16 stage slopes of a lane: a per-wave ring in global memory (L2-resident) when two waves share a SIMD; in the
    // SPLIT form a wave has the SIMD's whole register file and keeps them in registers (every index below is a compile-time
    // constant after unrolling) -- no L2 round trip per stage, and no global load left in the stage loop that would have to
    // wait, through the in-order vmcnt, for the factor stores of the previous stage
    float4* ring = reinterpret_cast<float4*>(g.slopes) + (tile * 7 * 4) * 64 + lane;      // [stage][4][64 lanes]
    float4 rreg[SPLIT ? 28 : 1];
    auto ring_get = [&](int j, int v) -> float4 { if constexpr (SPLIT) return rreg[j * 4 + v]; else return ring[(j * 4 + v) * 64]; };
    auto ring_put = [&](int j, int v, float4 x) { if constexpr (SPLIT) rreg[j * 4 + v] = x; else ring[(j * 4 + v) * 64] = x; };
    float4 w1tr[SPLIT ? 16 : 1];
    if constexpr (SPLIT) {
#pragma unroll
      for (int T1 = 0; T1 < 8; ++T1) { w1tr[T1] = w1t_base[T1 * 64]; w1tr[8 + T1] = w1t_base[(8 + T1) * 64]; }
    }
    float vtS = 0.f, vtE = 0.f;
    float ktv[DCTRL ? 7 : 1] = {};
    // DCTRL: this series' row of the pending buffer -- lane (n, q) stores channels q, 4 + q, ..
    float* gx_mine = DCTRL ? g.gx + (((int64_t)p * g.B + sc) * CT) * 8 : nullptr;
    const float* gx_prev = DCTRL ? g.gx + (((int64_t)p2 * g.B + sc) * CT) * 8 : nullptr;
    // the control row of a stage is requested one stage ahead (SPLIT): its latency hides behind the previous evaluation
    Row<DEGREE, CT> row_next;
    if constexpr (SPLIT) row_next = load_row<DEGREE, CT>(g.coeffs, sc, g.n_intervals, sidx[0], Cr);
#ifdef CDE_PHASE_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    CDE_STAMP(3);

    // ---- one stage, given the state handed to it and its scalars: control, evaluation, slopes (returned in fa .. vb)
#ifdef CDE_PHASE_TRACE
    unsigned long long est[4] = {0, 0, 0, 0};
#endif
    auto stage = [&](int i, const f32x4& za, const f32x4& zb, const f32x4& sa, const f32x4& sb, const Row<DEGREE, CT>& row,
                     int idx_i, float frac_i, float wS_i, float wE_i, f32x4& fa, f32x4& fb, f32x4& va, f32x4& vb) {
      float dX[CT], d2X[CT];
      {
        const float width = DEGREE == CDE_PATH_LINEAR ? g.knots[idx_i + 1] - g.knots[idx_i] : 1.f;
        control_slope<DEGREE, CT>(row, frac_i, width, dX);
        const float* f = reinterpret_cast<const float*>(row.v);
#pragma unroll
        for (int cc = 0; cc < CT; ++cc)
          d2X[cc] = DEGREE == CDE_PATH_CUBIC ? f[CT + cc] + 2.f * f[2 * CT + cc] * frac_i : 0.f;
      }
      const float zs[8] = {za[0], za[1], za[2], za[3], zb[0], zb[1], zb[2], zb[3]};
      const float as[8] = {sa[0], sa[1], sa[2], sa[3], sb[0], sb[1], sb[2], sb[3]};
      const bool keeps = mode == 2 && (i == 0 || i == 6);            // a first / last stage: the next attempt may start from it
      const bool stream = valid && (wS_i != 0.f || wE_i != 0.f || keeps) && !(g.dbg & 1);
      const int64_t out_row = (int64_t)(mode <= 1 ? i : i == 6 ? six : madj_slot(i)) * g.rows_per_stage + series;
      float kt;
      float gxo[DCTRL ? CT : 1];
      {
        mlp_adjoint_eval<ACT, CT, DEGREE == CDE_PATH_CUBIC, SPLIT, DCTRL, HI>(
            lds, w1t_base, lane, n, q, w2y_off, w2g_off, zs, as, dX, d2X, stream, g.U + out_row * U_COLS + 4 * q,
            g.Z + out_row * Z_COLS, g.G2 + out_row * G2_COLS + CT * q, g.G1 + out_row * G1_COLS + 4 * q, Hr, fa, fb, va, vb, kt,
            pw, xbuf, w1tr,
#ifdef CDE_PHASE_TRACE
            i == 3, est,
#else
            false, nullptr,
#endif
            gxo, hi, HI ? g.G2hi + out_row * G2_COLS + CT * q : nullptr);
      }
      if (DEGREE == CDE_PATH_CUBIC) { vtS = __builtin_fmaf(wS_i, kt, vtS); vtE = __builtin_fmaf(wE_i, kt, vtE); }
      if constexpr (DCTRL) {
        // the pending values of this stage: slot i of the series' row (channel c by lane quarter c & 3), and the time term
        if (valid && writer) {
#pragma unroll
          for (int c = 0; c < CT; ++c) if ((c & 3) == q) gx_mine[c * 8 + i] = gxo[c];
        }
        float term = kt;                                             // cubic: a . F d2X/dt2 (this lane's share)
        if (DEGREE != CDE_PATH_CUBIC) {                              // linear: a . f (the knot times act through the widths)
          term = 0.f;
#pragma unroll
          for (int m = 0; m < 4; ++m) term = __builtin_fmaf(as[m], fa[m], __builtin_fmaf(as[4 + m], fb[m], term));
        }
#pragma unroll
        for (int kk = 0; kk < 7; ++kk) ktv[kk] += i == kk ? term : 0.f;
      }
      if constexpr (!SPLIT) {
        if (keeps && valid && DEGREE == CDE_PATH_CUBIC) g.stash_t[((int64_t)(i == 0 ? 0 : 1) * g.B + series) * 4 + q] = kt;
      } else if (keeps && valid && writer) {                       // planes 0 / 1 / 2 <-> blocks 0 / 5 / 6
        const int64_t at = (int64_t)(i == 0 ? 0 : six - 4) * g.B + series;
        store_units4<4>(g.stash_y + at * Hr, ua, Hr, -fa); store_units4<4>(g.stash_y + at * Hr, ub, Hr, -fb);
        store_units4<4>(g.stash_a + at * Hr, ua, Hr, va); store_units4<4>(g.stash_a + at * Hr, ub, Hr, vb);
        g.stash_t[at * 4 + q] = kt;
      }
    };

    if constexpr (!SPLIT) {
      if (reuse) {
        // the first stage's slopes: slot 0 of the ring (after an accepted step: that step's last stage, slot 6)
        float kt0 = valid ? g.stash_t[((int64_t)(plan.accept ? 1 : 0) * g.B + sc) * 4 + q] : 0.f;
        if (plan.accept) {
#pragma unroll
          for (int v = 0; v < 4; ++v) ring_put(0, v, ring_get(6, v));
          if (valid && DEGREE == CDE_PATH_CUBIC) g.stash_t[((int64_t)0 * g.B + series) * 4 + q] = kt0;
        }
        if (DEGREE == CDE_PATH_CUBIC) { vtS = __builtin_fmaf(wS[0], kt0, vtS); vtE = __builtin_fmaf(wE[0], kt0, vtE); }
        if constexpr (DCTRL) {
          // the reused stage's pending values: the previous launch's slot 0 (rejected) or 6 (accepted); its time term from the
          // stash (cubic) or from the kept slopes (linear: a . f, with a = the step's start value)
          if (valid) {
#pragma unroll
            for (int c = 0; c < CT; ++c) if ((c & 3) == q) gx_mine[c * 8] = gx_prev[c * 8 + (plan.accept ? 6 : 0)];
          }
          float term = kt0;
          if (DEGREE != CDE_PATH_CUBIC) {
            const float4 k0 = ring_get(0, 0), k1 = ring_get(0, 1);
            term = -((a0a[0] * k0.x + a0a[1] * k0.y + a0a[2] * k0.z + a0a[3] * k0.w) +
                     (a0b[0] * k1.x + a0b[1] * k1.y + a0b[2] * k1.z + a0b[3] * k1.w));
          }
          ktv[0] += term;
        }
      }
      // one wave per tile: fully unrolled (the slope ring lives in global memory, every index is static)
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        if (i >= ns || (i == 0 && reuse)) continue;
        f32x4 za = y0a, zb = y0b, sa = a0a, sb = a0b;
        if (i > 0) {
          f32x4 ia = {0.f, 0.f, 0.f, 0.f}, ib = ia, ja = ia, jb = ia;
#pragma unroll
          for (int j = 0; j < i; ++j) {
            const float wgt = bc[i][j];
            const float4 k0 = ring_get(j, 0), k1 = ring_get(j, 1), k2 = ring_get(j, 2), k3 = ring_get(j, 3);
            const f32x4 wv4 = {wgt, wgt, wgt, wgt};
            ia = __builtin_elementwise_fma(f32x4{k0.x, k0.y, k0.z, k0.w}, wv4, ia);
            ib = __builtin_elementwise_fma(f32x4{k1.x, k1.y, k1.z, k1.w}, wv4, ib);
            ja = __builtin_elementwise_fma(f32x4{k2.x, k2.y, k2.z, k2.w}, wv4, ja);
            jb = __builtin_elementwise_fma(f32x4{k3.x, k3.y, k3.z, k3.w}, wv4, jb);
          }
          za = y0a + ia; zb = y0b + ib; sa = a0a + ja; sb = a0b + jb;
        }
        const Row<DEGREE, CT> row = load_row<DEGREE, CT>(g.coeffs, sc, g.n_intervals, sidx[i], Cr);
        f32x4 fa, fb, va, vb;
        stage(i, za, zb, sa, sb, row, sidx[i], sfrac[i], wS[i], wE[i], fa, fb, va, vb);
        ring_put(i, 0, make_float4(-fa[0], -fa[1], -fa[2], -fa[3]));
        ring_put(i, 1, make_float4(-fb[0], -fb[1], -fb[2], -fb[3]));
        ring_put(i, 2, make_float4(va[0], va[1], va[2], va[3]));
        ring_put(i, 3, make_float4(vb[0], vb[1], vb[2], vb[3]));
        CDE_STAMP(4 + i);
      }
    } else {
      // Waves sharing a tile (round 4): a REAL loop.  The slope ring lives in registers, which only static indices can
      // address, so the stage number enters through (wave-uniform) select chains instead: the Butcher row of stage i as six
      // weights (zero for the slopes stage i does not use -- the ring starts out as zeros), the new slopes into slot i by
      // 16 x 7 conditional moves.  The loop body is one evaluation: ~10 KB of code instead of the 200 KB of seven inlined
      // evaluations, and the unit-group loop inside the four-wave evaluation is a real loop as well.
#pragma unroll
      for (int e = 0; e < 28; ++e) rreg[e] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (reuse) {
        const int64_t at = (int64_t)(src0 == 0 ? 0 : src0 - 4) * g.B + sc;
        const f32x4 k0 = load_units4<4>(g.stash_y + at * Hr, ua, Hr), k1 = load_units4<4>(g.stash_y + at * Hr, ub, Hr);
        f32x4 k2 = load_units4<4>(g.stash_a + at * Hr, ua, Hr), k3 = load_units4<4>(g.stash_a + at * Hr, ub, Hr);
        if (!valid) { k2 = f32x4{0.f, 0.f, 0.f, 0.f}; k3 = k2; }  // (padded lanes: a == 0, so are its slope and vjp_t's)
        rreg[0] = make_float4(k0[0], k0[1], k0[2], k0[3]); rreg[1] = make_float4(k1[0], k1[1], k1[2], k1[3]);
        rreg[2] = make_float4(k2[0], k2[1], k2[2], k2[3]); rreg[3] = make_float4(k3[0], k3[1], k3[2], k3[3]);
        if (DEGREE == CDE_PATH_CUBIC) {
          const float kt0 = valid ? g.stash_t[at * 4 + q] : 0.f;
          vtS = __builtin_fmaf(wS[0], kt0, vtS); vtE = __builtin_fmaf(wE[0], kt0, vtE);
          if constexpr (DCTRL) ktv[0] += kt0;
        } else if constexpr (DCTRL) {
          ktv[0] -= (a0a[0] * k0[0] + a0a[1] * k0[1] + a0a[2] * k0[2] + a0a[3] * k0[3]) +
                    (a0b[0] * k1[0] + a0b[1] * k1[1] + a0b[2] * k1[2] + a0b[3] * k1[3]);
        }
        if constexpr (DCTRL) {
          if (valid && writer) {
#pragma unroll
            for (int c = 0; c < CT; ++c) if ((c & 3) == q) gx_mine[c * 8] = gx_prev[c * 8 + (plan.accept ? 6 : 0)];
          }
        }
        row_next = load_row<DEGREE, CT>(g.coeffs, sc, g.n_intervals, sidx[1], Cr);      // the first evaluated stage is stage 1
      }
#pragma clang loop unroll(disable)
      for (int i = reuse ? 1 : 0; i < ns; ++i) {
        int idx_i = sidx[0], idx_n = sidx[1];
        float frac_i = sfrac[0], wS_i = wS[0], wE_i = wE[0];
        float wj[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 1; kk < 7; ++kk) {
          const bool at = i == kk;
          idx_i = at ? sidx[kk] : idx_i; frac_i = at ? sfrac[kk] : frac_i; wS_i = at ? wS[kk] : wS_i; wE_i = at ? wE[kk] : wE_i;
          if (kk < 6) idx_n = at ? sidx[kk + 1] : idx_n;
#pragma unroll
          for (int j = 0; j < 6; ++j) wj[j] = at ? bc[kk][j] : wj[j];
        }
        f32x4 ia = {0.f, 0.f, 0.f, 0.f}, ib = ia, ja = ia, jb = ia;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const float4 k0 = rreg[4 * j], k1 = rreg[4 * j + 1], k2 = rreg[4 * j + 2], k3 = rreg[4 * j + 3];
          const f32x4 wv4 = {wj[j], wj[j], wj[j], wj[j]};
          ia = __builtin_elementwise_fma(f32x4{k0.x, k0.y, k0.z, k0.w}, wv4, ia);
          ib = __builtin_elementwise_fma(f32x4{k1.x, k1.y, k1.z, k1.w}, wv4, ib);
          ja = __builtin_elementwise_fma(f32x4{k2.x, k2.y, k2.z, k2.w}, wv4, ja);
          jb = __builtin_elementwise_fma(f32x4{k3.x, k3.y, k3.z, k3.w}, wv4, jb);
        }
        const f32x4 za = y0a + ia, zb = y0b + ib, sa = a0a + ja, sb = a0b + jb;
        const Row<DEGREE, CT> row = row_next;
        if (i + 1 < ns) row_next = load_row<DEGREE, CT>(g.coeffs, sc, g.n_intervals, idx_n, Cr);
        f32x4 fa, fb, va, vb;
        stage(i, za, zb, sa, sb, row, idx_i, frac_i, wS_i, wE_i, fa, fb, va, vb);
        const float4 n0 = make_float4(-fa[0], -fa[1], -fa[2], -fa[3]), n1 = make_float4(-fb[0], -fb[1], -fb[2], -fb[3]);
        const float4 n2 = make_float4(va[0], va[1], va[2], va[3]), n3 = make_float4(vb[0], vb[1], vb[2], vb[3]);
        auto pick = [](bool c, const float4& x, const float4& y) { return make_float4(c ? x.x : y.x, c ? x.y : y.y, c ? x.z : y.z, c ? x.w : y.w); };
#pragma unroll
        for (int kk = 0; kk < 7; ++kk) {
          const bool at = i == kk;
          rreg[4 * kk] = pick(at, n0, rreg[4 * kk]); rreg[4 * kk + 1] = pick(at, n1, rreg[4 * kk + 1]);
          rreg[4 * kk + 2] = pick(at, n2, rreg[4 * kk + 2]); rreg[4 * kk + 3] = pick(at, n3, rreg[4 * kk + 3]);
        }
#ifdef CDE_PHASE_TRACE
#pragma unroll
        for (int kk = 0; kk < 7; ++kk) if (i == kk) CDE_STAMP(4 + kk);
#endif
      }
    }
#ifdef CDE_PHASE_TRACE
    stamps_.t[13] = est[0]; stamps_.t[14] = est[1]; stamps_.t[15] = est[2]; stamps_.t[16] = est[3];
#endif

    // ---- what this launch owes the controller (one more pass over the slopes)
    auto sq4f = [](const f32x4& v) { return (double)(v[0] * v[0]) + (double)(v[1] * v[1]) + (double)(v[2] * v[2]) + (double)(v[3] * v[3]); };
    auto abs4f = [](const f32x4& v) { return f32x4{fabsf(v[0]), fabsf(v[1]), fabsf(v[2]), fabsf(v[3])}; };
    auto max4f = [](const f32x4& a, const f32x4& b) { return f32x4{fmaxf(a[0], b[0]), fmaxf(a[1], b[1]), fmaxf(a[2], b[2]), fmaxf(a[3], b[3])}; };
    auto slope = [&](int j, int v) { const float4 t = ring_get(j, v); return f32x4{t.x, t.y, t.z, t.w}; };
    if (mode == 0) {
      const f32x4 sya = atol + abs4f(y0a) * rtol, syb = atol + abs4f(y0b) * rtol;       // Hairer's scale
      const f32x4 saa = atol + abs4f(a0a) * rtol, sab = atol + abs4f(a0b) * rtol;
      if (valid && writer) {
        acc[0] = sq4f(y0a / sya) + sq4f(y0b / syb); acc[1] = sq4f(a0a / saa) + sq4f(a0b / sab);
        acc[2] = sq4f(slope(0, 0) / sya) + sq4f(slope(0, 1) / syb); acc[3] = sq4f(slope(0, 2) / saa) + sq4f(slope(0, 3) / sab);
      }
    } else if (mode == 1) {
      const f32x4 sya = atol + abs4f(y0a) * rtol, syb = atol + abs4f(y0b) * rtol;
      const f32x4 saa = atol + abs4f(a0a) * rtol, sab = atol + abs4f(a0b) * rtol;
      if (valid && writer) {
        acc[0] = sq4f((slope(1, 0) - slope(0, 0)) / sya) + sq4f((slope(1, 1) - slope(0, 1)) / syb);
        acc[1] = sq4f((slope(1, 2) - slope(0, 2)) / saa) + sq4f((slope(1, 3) - slope(0, 3)) / sab);
      }
    } else {
      // y1 / a1 = the state handed to stage 6 (FSAL row == solution weights); error estimate with c_err (mode 2)
      f32x4 iy[2], ia2[2], ey[2], ea[2], ma[2];
#pragma unroll
      for (int v = 0; v < 2; ++v) { iy[v] = f32x4{0.f, 0.f, 0.f, 0.f}; ia2[v] = iy[v]; ey[v] = iy[v]; ea[v] = iy[v]; ma[v] = iy[v]; }
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const float ws = j < 6 ? bc[6][j] : 0.f, we = wE[j], wm = dtf * (float)DP_CMID[j];
        const f32x4 ws4 = {ws, ws, ws, ws}, we4 = {we, we, we, we}, wm4 = {wm, wm, wm, wm};
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          const f32x4 ky = slope(j, v), ka = slope(j, 2 + v);
          iy[v] = __builtin_elementwise_fma(ky, ws4, iy[v]); ia2[v] = __builtin_elementwise_fma(ka, ws4, ia2[v]);
          ey[v] = __builtin_elementwise_fma(ky, we4, ey[v]); ea[v] = __builtin_elementwise_fma(ka, we4, ea[v]);
          ma[v] = __builtin_elementwise_fma(ka, wm4, ma[v]);
        }
      }
      const f32x4 y1a = y0a + iy[0], y1b = y0b + iy[1], a1a = a0a + ia2[0], a1b = a0b + ia2[1];
      if (mode == 2) {
        const f32x4 tya = atol + rtol * max4f(abs4f(y0a), abs4f(y1a)), tyb = atol + rtol * max4f(abs4f(y0b), abs4f(y1b));
        const f32x4 taa = atol + rtol * max4f(abs4f(a0a), abs4f(a1a)), tab = atol + rtol * max4f(abs4f(a0b), abs4f(a1b));
        if (valid && writer) {
          acc[0] = sq4f(ey[0] / tya) + sq4f(ey[1] / tyb); acc[1] = sq4f(ea[0] / taa) + sq4f(ea[1] / tab);
          store_units4<4>(Sq + 2 * BH + series * Hr, ua, Hr, y1a); store_units4<4>(Sq + 2 * BH + series * Hr, ub, Hr, y1b);
          store_units4<4>(Sq + 3 * BH + series * Hr, ua, Hr, a1a); store_units4<4>(Sq + 3 * BH + series * Hr, ub, Hr, a1b);
        }
      } else {
        // mode 3: a(s1) by torchdiffeq's dense output (_interp_fit / _interp_evaluate, the oracle's expression order)
        auto dense = [&](const f32x4& y0, const f32x4& y1, const f32x4& f0, const f32x4& f1, const f32x4& mid) {
          const f32x4 ym = y0 + mid;
          const f32x4 ca = 2.f * dtf * (f1 - f0) - 8.f * (y1 + y0) + 16.f * ym;
          const f32x4 cb = dtf * (5.f * f0 - 3.f * f1) + 18.f * y0 + 14.f * y1 - 32.f * ym;
          const f32x4 cc = dtf * (f1 - 4.f * f0) - 11.f * y0 - 5.f * y1 + 16.f * ym;
          const f32x4 cd = dtf * f0;
          f32x4 total = y0 + x_end * cd;
          float xp = x_end;
          xp = xp * x_end; total = total + xp * cc;
          xp = xp * x_end; total = total + xp * cb;
          xp = xp * x_end; total = total + xp * ca;
          return total;
        };
        if (valid && writer) {
          store_units4<4>(g.a_out + series * Hr, ua, Hr, dense(a0a, a1a, slope(0, 2), slope(6, 2), ma[0]));
          store_units4<4>(g.a_out + series * Hr, ub, Hr, dense(a0b, a1b, slope(0, 3), slope(6, 3), ma[1]));
        }
      }
    }
    if (writer) { acc[4] = (double)vtS; acc[5] = (double)vtE; }
    if constexpr (DCTRL) {
      if (writer) {
#pragma unroll
        for (int j = 0; j < 7; ++j) ktd[j] = (double)ktv[j];
      }
    }
  }
  CDE_STAMP(11);
  // ---- publish this launch's partial sums
  block_total<ADJ_NS>(acc, red);
  CDE_STAMP(12);
  CDE_STAMP_FLUSH(k4am_phase_trace, attempt_no);
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < ADJ_NS; ++i) Pq[ADJ_NS * blockIdx.x + i] = acc[i];
  }
  if constexpr (DCTRL) {
    if (g.with_knots) {
      block_total<7>(ktd, red);
      if (tid == 0) {
        double* dst = g.ktp + ((int64_t)p * g.n_wg_max + blockIdx.x) * 8;
#pragma unroll
        for (int j = 0; j < 7; ++j) dst[j] = ktd[j];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ eight waves per tile (round 4)
// The attempt kernel for SMALL batches (at most one 16-series tile per CU, i.e. up to 4096 series; CT = 8) -- the range the
// reference's examples train in (batch_size = 32: example/time_series_classification.py:149).  Same controller, same
// launch protocol and workspace as dopri5_mlp_adjoint_attempt; what differs is who does what inside a workgroup:
//   * the EIGHT waves (two per SIMD) share one tile and split every evaluation eight ways (cde_mlp_adj.h:
//     mlp_adjoint_eval_split8: 144 MFMAs per wave instead of 384 with four waves, a quarter of the vector work);
//   * the stage loop is a real loop around ONE evaluation body (~10 KB of code; the four-wave form had seven inlined
//     evaluations, 200 KB), the slope ring lives in registers, and of the adjoint state a wave carries only its own
//     component a_{4w+q} -- 56 + 7 ring registers instead of 112, which is what lets two waves share a SIMD without spills;
//   * wave 0 owns y (state stores, error sums), every wave its component of a.
// profiles/r04_phase_k4am_*.log: the four-wave form spent 14 us per evaluation on 5.1 us of MFMA work -- one wave per SIMD
// issues one instruction every ~5 cycles and overlaps nothing with its own MFMAs.
template <int DEGREE, int ACT>
__global__ __launch_bounds__(512, 2) void dopri5_mlp_adjoint_attempt_s8(MlpAdjArgs g, int parity) {
  constexpr int CT = 8;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  const int p = parity, p2 = parity ^ 1;
  CDE_STAMP_DECL;
  CDE_STAMP(0);
  // the pending sums are requested together with the controller block, not after it (both were written by the previous
  // launches, on other XCDs: each dependent round trip through the memory side costs ~2 us); before the first attempt
  // of an interval they are ignored
  constexpr int NSUM = MADJ_NSUM_PLAIN;                            // (this kernel has no control-gradient form)
  double sum[NSUM];
#pragma unroll
  for (int i = 0; i < NSUM; ++i) sum[i] = 0.0;
  {
    const double* Pp0 = g.partial + (int64_t)p * g.n_wg_max * ADJ_NS;
    for (int64_t b = tid; b < (int64_t)gridDim.x; b += blockDim.x) {
#pragma unroll
      for (int i = 0; i < ADJ_NS; ++i) sum[i] += Pp0[ADJ_NS * b + i];
    }
    const double* Qp = g.pq + (int64_t)p * g.pq_blocks * 8;
    for (int b = tid; b < g.n_pq; b += blockDim.x) {
#pragma unroll
      for (int i = 0; i < 8; ++i) sum[ADJ_NS + i] += Qp[8 * b + i];
    }
  }
  AdjCtrl k = *reinterpret_cast<const AdjCtrl*>(g.ctrl + p * ADJ_CTRL_STRIDE);
  DopriCtrl& c = k.c;
  if (c.phase == 4) {
    if (blockIdx.x == 0 && tid == 0) { k.commit = 0; k.mode = 3; *reinterpret_cast<AdjCtrl*>(g.ctrl + p2 * ADJ_CTRL_STRIDE) = k; }
    return;
  }
#ifdef CDE_PHASE_TRACE
  const int attempt_no = uni((int)(c.n_accept + c.n_reject));
#endif
  const int lane = tid & 63, w = tid >> 6;
  // Everything that does not depend on the controller's decision is requested before the first wait: the LDS image
  // (b1 | W2 | b2; straight into LDS, 1 KB per wave-wide load -- the W1 part is not needed there, its tile goes to
  // registers), this wave's W1 / W1^T tiles, and BOTH candidates of the start state (committed or attempted).
  for (int chunk = W1M_FLOATS / 256 + w; chunk * 256 < ADJ_LDS_FLOATS; chunk += 8) {
    if (chunk * 256 + lane * 4 < ADJ_LDS_FLOATS)
      __builtin_amdgcn_global_load_lds(g.img + chunk * 256 + lane * 4, lds + chunk * 256, 16, 0, 0);
  }
  float4 w1r[2], w1tr[2];
  {
    const float4* w1img = reinterpret_cast<const float4*>(g.img) + lane;
    w1r[0] = w1img[(2 * w) * 64]; w1r[1] = w1img[(2 * w + 1) * 64];
    const float4* w1t_base = reinterpret_cast<const float4*>(g.img + ADJ_LDS_FLOATS) + lane;
    w1tr[0] = w1t_base[w * 64]; w1tr[1] = w1t_base[(8 + w) * 64];
  }
  double* red = reinterpret_cast<double*>(lds + ADJ_LDS_FLOATS);
  float* xb = lds + ADJ_LDS_FLOATS + 2 * MADJ_NSUM * 8;            // exchange window B (9 KB), behind `red`
  float* xa = lds;                                                 // exchange window A: the W1 image's 16 KB, once it is in registers
  const int Hr = g.dims.H, Cr = g.dims.C;
  const int n = lane & 15, q = lane >> 4;
  const int64_t BH = g.B * Hr;
  const float* Sp = g.state + (int64_t)p * 4 * BH;
  float* Sq = g.state + (int64_t)p2 * 4 * BH;
  double* Pq = g.partial + (int64_t)p2 * g.n_wg_max * ADJ_NS;
  const float rtol = (float)g.com.rtol, atol = (float)g.com.atol;
  const int64_t tile = blockIdx.x;                                 // (the grid is exactly the tiles)
  const int64_t series = tile * 16 + n;
  const bool valid = series < g.B;
  const int64_t sc = valid ? series : g.B - 1;
  const int hw = 4 * w + q;                                        // the hidden unit whose adjoint component this lane carries
  const bool own = valid && hw < Hr;
  const int ua = q, ub = 16 + q;
  const bool fresh = c.phase == 0;
  const float* yc0 = fresh ? g.y_init : Sp + 0 * BH;               // start state if the pending attempt is not committed ..
  const float* ac0 = fresh ? g.a_init : Sp + 1 * BH;
  const f32x4 yk0a = load_units4<4>(yc0 + sc * Hr, ua, Hr), yk0b = load_units4<4>(yc0 + sc * Hr, ub, Hr);
  const f32x4 yk1a = load_units4<4>((fresh ? yc0 : Sp + 2 * BH) + sc * Hr, ua, Hr);           // .. and if it is
  const f32x4 yk1b = load_units4<4>((fresh ? yc0 : Sp + 2 * BH) + sc * Hr, ub, Hr);
  const float ak0 = own ? ac0[sc * Hr + hw] : 0.f, ak1 = own ? (fresh ? ac0 : Sp + 3 * BH)[sc * Hr + hw] : 0.f;
  // ... and both candidates of the FIRST STAGE's slopes (first-same-as-last): the pending attempt's own first stage if it is
  // rejected (same start state, same stage time: torchdiffeq keeps f0), its last stage if it is accepted (f1 becomes f0)
  const int in_src0 = uni((int)k.src0), in_six = uni((int)k.six) & 15;
  auto stash_of = [](int blk) { return blk == 0 ? 0 : blk - 4; };  // blocks 0, 5, 6 -> the stash's three planes
  const int64_t at_r = (int64_t)stash_of(in_src0) * g.B + sc, at_a = (int64_t)stash_of(fresh ? 0 : in_six) * g.B + sc;
  const f32x4 kr_a = load_units4<4>(g.stash_y + at_r * Hr, ua, Hr), kr_b = load_units4<4>(g.stash_y + at_r * Hr, ub, Hr);
  const f32x4 ka_a = load_units4<4>(g.stash_y + at_a * Hr, ua, Hr), ka_b = load_units4<4>(g.stash_y + at_a * Hr, ub, Hr);
  const float sr_a = own ? g.stash_a[at_r * Hr + hw] : 0.f, sa_a = own ? g.stash_a[at_a * Hr + hw] : 0.f;
  // (padded lanes repeat the last series' y, with a == 0: their share of vjp_t's slope is zero, not that series')
  const float kt_r = valid ? g.stash_t[at_r * 4 + q] : 0.f, kt_a = valid ? g.stash_t[at_a * 4 + q] : 0.f;

  // ---- pending sums, controller, stage scalars: as in dopri5_mlp_adjoint_attempt
  if (c.phase == 0) {
#pragma unroll
    for (int i = 0; i < NSUM; ++i) sum[i] = 0.0;
  } else if (g.ext_sums) {                                         // one controller for all shards: the reduced state sums
#pragma unroll
    for (int i = 0; i < ADJ_NS; ++i) sum[i] = tid == 0 ? g.ext_sums[i] : 0.0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's part of the LDS image has landed ..
  block_total<NSUM>(sum, red);                                     // .. and, past its barriers, everybody's
  CDE_STAMP(1);
  const int phase_in = c.phase;
  const AdjPlan plan = adj_controller(g.com, k, sum, sum + ADJ_NS);
  const int mode = uni(plan.mode);
  const bool commit = phase_in == 3 && plan.accept && mode != 3;
  const int ns = mode == 0 ? 1 : mode == 1 ? 2 : 7;
  const float t0f = uni((float)plan.t0), dtf = uni((float)plan.dt), t1f = uni((float)plan.t1);
  int sidx[7];
  float sfrac[7];
  {
    float ts = 0.f;
    const int i = lane & 7;
    if (mode == 0) ts = (float)c.t_hi;
    else if (mode == 1) ts = i == 0 ? (float)c.t_hi : (float)(c.t_hi + (double)plan.h0);
    else if (i == 0) ts = plan.kind0 == 0 ? t0f : next_toward(t0f, plan.kind0 > 0 ? 1.f : -1.f);
    else if (i <= 4) ts = t0f + (float)DP_ALPHA[i - 1] * dtf;
    else ts = next_toward(t1f, -1.f);
    float frac;
    const int idx = (int)locate_around(g.knots, g.n_intervals, -ts, phase_in == 0 ? (int64_t)-1 : (int64_t)c.slot, frac);
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      sidx[j] = __builtin_amdgcn_readlane(idx, j);
      sfrac[j] = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(frac), j));
    }
  }
  float bc[7][6];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
#pragma unroll
    for (int j = 0; j < 6; ++j) bc[i][j] = (i >= 1 && j < i) ? uni((float)DP_BETA[i - 1][j] * dtf) : 0.f;
  }
  if (mode == 1) bc[1][0] = uni(plan.h0);
  float wS[7], wE[7];
  adj_stage_weights(mode, dtf, plan.x_end, wS, wE);
#pragma unroll
  for (int j = 0; j < 7; ++j) { wS[j] = uni(wS[j]); wE[j] = uni(wE[j]); }
  const float x_end = uni(plan.x_end);
  CDE_STAMP(2);
  // first-same-as-last: an attempt that follows an attempt (phase 3) starts where that one started (rejected) or ended
  // (accepted) -- unless the accepted step ended on a jump: f is then evaluated just AFTER the jump, as torchdiffeq does
  const bool reuse = phase_in == 3 && mode == 2 && !(plan.accept && c.refresh) && !(g.dbg & 2) &&
                     !((g.dbg & 4) && plan.accept) && !((g.dbg & 8) && !plan.accept);
  const int src0 = reuse ? (plan.accept ? in_six : in_src0) : 0;
  const int six = reuse ? (plan.accept ? 11 - in_six : in_six) : 5;                  // 11 - 5 = 6, 11 - 6 = 5
  k.src0 = src0; k.six = six | (reuse ? 0 : ADJ_FRESH0);
  if (blockIdx.x == 0 && tid == 0) {
    c.phase = mode == 0 ? 1 : mode == 1 ? 2 : mode == 2 ? 3 : 4;
    c.slot = sidx[0];
    *reinterpret_cast<AdjCtrl*>(g.ctrl + p2 * ADJ_CTRL_STRIDE) = k;
  }

  double acc[ADJ_NS] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  const bool writer = w == 0;
  const float4 b1r = (reinterpret_cast<const float4*>(lds + W1M_FLOATS) + q)[4 * w];
  const int w2y_off = ((n >> 3) * 8 + w2p_residue(n >> 2, n & 3)) * W2P_STRIDE + 4 * q;
  int w2g_off[4];
#pragma unroll
  for (int cl = 0; cl < 4; ++cl) w2g_off[cl] = ((q >> 1) * 8 + w2p_residue(q, cl)) * W2P_STRIDE + n;
  const f32x4 y0a = commit ? yk1a : yk0a, y0b = commit ? yk1b : yk0b;
  const float a0 = commit ? ak1 : ak0;                             // a == 0 stays 0: padded lanes / units contribute nothing
  if (mode != 3) {
    if (valid && writer) {
      store_units4<4>(Sq + 0 * BH + series * Hr, ua, Hr, y0a); store_units4<4>(Sq + 0 * BH + series * Hr, ub, Hr, y0b);
    }
    if (own) Sq[1 * BH + series * Hr + hw] = a0;
  }
  // the slope ring: dy/ds of all 8 units of this lane (2 float4 per stage), da/ds of the wave's own unit
  float4 ry[14];
  float ra[7];
#pragma unroll
  for (int e = 0; e < 14; ++e) ry[e] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int e = 0; e < 7; ++e) ra[e] = 0.f;
  float vtS = 0.f, vtE = 0.f;
  if (reuse) {
    const f32x4 k0a = plan.accept ? ka_a : kr_a, k0b = plan.accept ? ka_b : kr_b;
    ry[0] = make_float4(k0a[0], k0a[1], k0a[2], k0a[3]);
    ry[1] = make_float4(k0b[0], k0b[1], k0b[2], k0b[3]);
    ra[0] = plan.accept ? sa_a : sr_a;
    if (DEGREE == CDE_PATH_CUBIC) {
      const float kt0 = plan.accept ? kt_a : kt_r;
      vtS = __builtin_fmaf(wS[0], kt0, vtS); vtE = __builtin_fmaf(wE[0], kt0, vtE);
    }
  }
#ifdef CDE_PHASE_TRACE
  unsigned long long est[4] = {0, 0, 0, 0};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  CDE_STAMP(3);

#pragma clang loop unroll(disable)
  for (int i = reuse ? 1 : 0; i < ns; ++i) {
    // the control row of stage i first: its (L2) latency runs under the select chains and the stage combination below
    int idx_i = sidx[0];
    float frac_i = sfrac[0];
#pragma unroll
    for (int kk = 1; kk < 7; ++kk) { idx_i = i == kk ? sidx[kk] : idx_i; frac_i = i == kk ? sfrac[kk] : frac_i; }
    const Row<DEGREE, CT> row = load_row<DEGREE, CT>(g.coeffs, sc, g.n_intervals, idx_i, Cr);
    const float width = DEGREE == CDE_PATH_LINEAR ? g.knots[idx_i + 1] - g.knots[idx_i] : 1.f;
    // the other scalars of stage i and its Butcher row (zero for the slopes it does not use), by wave-uniform select chains
    float wS_i = wS[0], wE_i = wE[0];
    float wj[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 1; kk < 7; ++kk) {
      const bool at = i == kk;
      wS_i = at ? wS[kk] : wS_i; wE_i = at ? wE[kk] : wE_i;
#pragma unroll
      for (int j = 0; j < 6; ++j) wj[j] = at ? bc[kk][j] : wj[j];
    }
    f32x4 ia = {0.f, 0.f, 0.f, 0.f}, ib = ia;
    float ja = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const float4 k0 = ry[2 * j], k1 = ry[2 * j + 1];
      const f32x4 wv4 = {wj[j], wj[j], wj[j], wj[j]};
      ia = __builtin_elementwise_fma(f32x4{k0.x, k0.y, k0.z, k0.w}, wv4, ia);
      ib = __builtin_elementwise_fma(f32x4{k1.x, k1.y, k1.z, k1.w}, wv4, ib);
      ja = __builtin_fmaf(ra[j], wj[j], ja);
    }
    const f32x4 za = y0a + ia, zb = y0b + ib;
    const float as_w = a0 + ja;
    float dX[CT], d2X[CT];
    {
      control_slope<DEGREE, CT>(row, frac_i, width, dX);
      const float* f = reinterpret_cast<const float*>(row.v);
#pragma unroll
      for (int cc = 0; cc < CT; ++cc)
        d2X[cc] = DEGREE == CDE_PATH_CUBIC ? f[CT + cc] + 2.f * f[2 * CT + cc] * frac_i : 0.f;
    }
    const float zs[8] = {za[0], za[1], za[2], za[3], zb[0], zb[1], zb[2], zb[3]};
    const bool keeps = mode == 2 && (i == 0 || i == 6);            // a first / last stage: the next attempt may start from it
    const bool stream = valid && (wS_i != 0.f || wE_i != 0.f || keeps) && !(g.dbg & 1);
    const int64_t out_row = (int64_t)(mode <= 1 ? i : i == 6 ? six : madj_slot(i)) * g.rows_per_stage + series;
    f32x4 fa, fb;
    float va_w, kt;
    mlp_adjoint_eval_split8<ACT, DEGREE == CDE_PATH_CUBIC>(
        lds, xa, xb, lane, q, w, w1r, b1r, w1tr, w2y_off, w2g_off, zs, as_w, dX, d2X, stream, g.U + out_row * U_COLS + 4 * q,
        g.Z + out_row * Z_COLS, g.G2 + out_row * G2_COLS + CT * q, g.G1 + out_row * G1_COLS + 4 * q, Hr, fa, fb, va_w, kt
#ifdef CDE_PHASE_TRACE
        , i == 3, est
#endif
        );
    if (DEGREE == CDE_PATH_CUBIC) { vtS = __builtin_fmaf(wS_i, kt, vtS); vtE = __builtin_fmaf(wE_i, kt, vtE); }
    if (keeps) {                                                   // (the values the ring gets below)
      const int64_t at = (int64_t)(i == 0 ? 0 : stash_of(six)) * g.B + series;
      if (valid && writer) {
        store_units4<4>(g.stash_y + at * Hr, ua, Hr, -fa); store_units4<4>(g.stash_y + at * Hr, ub, Hr, -fb);
        g.stash_t[at * 4 + q] = kt;
      }
      if (own) g.stash_a[at * Hr + hw] = va_w;
    }
    // reverse-time slopes dy/ds = -f, da/ds = +a^T df/dz into slot i of the ring
#pragma unroll
    for (int kk = 0; kk < 7; ++kk) {
      const bool at = i == kk;
      ry[2 * kk] = make_float4(at ? -fa[0] : ry[2 * kk].x, at ? -fa[1] : ry[2 * kk].y, at ? -fa[2] : ry[2 * kk].z, at ? -fa[3] : ry[2 * kk].w);
      ry[2 * kk + 1] = make_float4(at ? -fb[0] : ry[2 * kk + 1].x, at ? -fb[1] : ry[2 * kk + 1].y, at ? -fb[2] : ry[2 * kk + 1].z,
                                   at ? -fb[3] : ry[2 * kk + 1].w);
      ra[kk] = at ? va_w : ra[kk];
    }
#ifdef CDE_PHASE_TRACE
#pragma unroll
    for (int kk = 0; kk < 7; ++kk) if (i == kk) CDE_STAMP(4 + kk);
#endif
  }
#ifdef CDE_PHASE_TRACE
  stamps_.t[13] = est[0]; stamps_.t[14] = est[1]; stamps_.t[15] = est[2]; stamps_.t[16] = est[3];
#endif

  // ---- what this launch owes the controller: wave 0 the y block, every wave its component of a
  auto sq4f = [](const f32x4& v) { return (double)(v[0] * v[0]) + (double)(v[1] * v[1]) + (double)(v[2] * v[2]) + (double)(v[3] * v[3]); };
  auto abs4f = [](const f32x4& v) { return f32x4{fabsf(v[0]), fabsf(v[1]), fabsf(v[2]), fabsf(v[3])}; };
  auto max4f = [](const f32x4& a, const f32x4& b) { return f32x4{fmaxf(a[0], b[0]), fmaxf(a[1], b[1]), fmaxf(a[2], b[2]), fmaxf(a[3], b[3])}; };
  auto ys = [&](int j, int v) { const float4 t = ry[2 * j + v]; return f32x4{t.x, t.y, t.z, t.w}; };
  auto sq1 = [](float v) { return (double)(v * v); };
  const bool yw = valid && writer;
  if (mode == 0) {
    const f32x4 sya = atol + abs4f(y0a) * rtol, syb = atol + abs4f(y0b) * rtol;       // Hairer's scale
    const float sa = atol + fabsf(a0) * rtol;
    if (yw) { acc[0] = sq4f(y0a / sya) + sq4f(y0b / syb); acc[2] = sq4f(ys(0, 0) / sya) + sq4f(ys(0, 1) / syb); }
    if (valid) { acc[1] = sq1(a0 / sa); acc[3] = sq1(ra[0] / sa); }
  } else if (mode == 1) {
    const f32x4 sya = atol + abs4f(y0a) * rtol, syb = atol + abs4f(y0b) * rtol;
    const float sa = atol + fabsf(a0) * rtol;
    if (yw) acc[0] = sq4f((ys(1, 0) - ys(0, 0)) / sya) + sq4f((ys(1, 1) - ys(0, 1)) / syb);
    if (valid) acc[1] = sq1((ra[1] - ra[0]) / sa);
  } else {
    f32x4 iy[2], ey[2];
    float ia2 = 0.f, ea = 0.f, ma = 0.f;
#pragma unroll
    for (int v = 0; v < 2; ++v) { iy[v] = f32x4{0.f, 0.f, 0.f, 0.f}; ey[v] = iy[v]; }
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const float ws = j < 6 ? bc[6][j] : 0.f, we = wE[j], wm = dtf * (float)DP_CMID[j];
      const f32x4 ws4 = {ws, ws, ws, ws}, we4 = {we, we, we, we};
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const f32x4 ky = ys(j, v);
        iy[v] = __builtin_elementwise_fma(ky, ws4, iy[v]);
        ey[v] = __builtin_elementwise_fma(ky, we4, ey[v]);
      }
      ia2 = __builtin_fmaf(ra[j], ws, ia2); ea = __builtin_fmaf(ra[j], we, ea); ma = __builtin_fmaf(ra[j], wm, ma);
    }
    const f32x4 y1a = y0a + iy[0], y1b = y0b + iy[1];
    const float a1 = a0 + ia2;
    if (mode == 2) {
      const f32x4 tya = atol + rtol * max4f(abs4f(y0a), abs4f(y1a)), tyb = atol + rtol * max4f(abs4f(y0b), abs4f(y1b));
      const float ta = atol + rtol * fmaxf(fabsf(a0), fabsf(a1));
      if (yw) {
        acc[0] = sq4f(ey[0] / tya) + sq4f(ey[1] / tyb);
        store_units4<4>(Sq + 2 * BH + series * Hr, ua, Hr, y1a); store_units4<4>(Sq + 2 * BH + series * Hr, ub, Hr, y1b);
      }
      if (valid) acc[1] = sq1(ea / ta);
      if (own) Sq[3 * BH + series * Hr + hw] = a1;
    } else if (own) {
      // mode 3: a(s1) by torchdiffeq's dense output (_interp_fit / _interp_evaluate, the oracle's expression order)
      const float f0 = ra[0], f1 = ra[6];
      const float ym = a0 + ma;
      const float ca = 2.f * dtf * (f1 - f0) - 8.f * (a1 + a0) + 16.f * ym;
      const float cb = dtf * (5.f * f0 - 3.f * f1) + 18.f * a0 + 14.f * a1 - 32.f * ym;
      const float cc = dtf * (f1 - 4.f * f0) - 11.f * a0 - 5.f * a1 + 16.f * ym;
      const float cd = dtf * f0;
      float total = a0 + x_end * cd;
      float xp = x_end;
      xp = xp * x_end; total = total + xp * cc;
      xp = xp * x_end; total = total + xp * cb;
      xp = xp * x_end; total = total + xp * ca;
      g.a_out[series * Hr + hw] = total;
    }
  }
  if (writer) { acc[4] = (double)vtS; acc[5] = (double)vtE; }
  CDE_STAMP(11);
  block_total<ADJ_NS>(acc, red);
  if (tid == 0) {
#pragma unroll
    for (int i = 0; i < ADJ_NS; ++i) Pq[ADJ_NS * blockIdx.x + i] = acc[i];
  }
  CDE_STAMP(12);
  CDE_STAMP_FLUSH(k4am_phase_trace, attempt_no);
}

// ------------------------------------------------------------------------------------------ the R kernel of this family
// Works on the elements of the slab-partial layout of the factor reduction: layer 2 [256][129] (row = padded (h, c),
// columns = hidden-layer unit, last column = the bias), then layer 1 [128][33].  K_s[e] = sum over the slabs of stage s
// (fixed order); S = sum_s wS[s] K_s, E = sum_s wE[s] K_s with the weights of cde_dopri_adj.h rebuilt from the controller
// block; then the commit / norm logic of adj_param_element, per-block sums per parameter tensor (W1, b1, W2, b2).
//   stage 0 (fused) : S, E from the slab partials -> commit -> norms          (unsharded: one launch per attempt)
//   stage 1         : S, E only, into `sums_out` (doubles)                    (sharded: the host all-reduces them ...)
//   stage 2         : commit + norms from `sums_in`                           (... every rank then holds the same images)
struct MlpReduceArgs {
  unsigned char* ctrl;
  const float* part2; const float* part1;   // slab partials [slot][sps][M][N + 1]
  int sps;
  float* kst;               // [3][MADJ_ELEMS]: the stage image K_s of a first / last stage (blocks 0, 5, 6), kept because the
                            // next attempt's first stage is one of them (AdjCtrl::src0) and is not reduced again
  float* G;                 // [MADJ_ELEMS] running totals (layer 2 block, then layer 1 block): what the caller gets back
  float* prevS;             // [2][MADJ_ELEMS]: the S sums of a launch, kept for the commit one launch later
  float* Gn;                // sharded: the running totals of the GLOBAL batch (the norm needs those); else nullptr
  float* prevSn;            // sharded: the global S sums
  double* pq;               // [2][MADJ_RBLOCKS][8]
  const double* partial; int n_wg; int n_wg_max;
  double* carry;
  double* sums_out;         // stage 1: [2][MADJ_ELEMS] doubles (S, E)
  const double* sums_in;    // stage 2: the reduced buffer (ADJ_NS state sums, then the S and E images)
  float rtol, atol;
  // 32 hidden units x 16 channels: the kernel runs a second time on the images of hidden units 16..31 -- a layer-2 block only
  // (`n_elems` = MADJ_P2, its own kst / G / prevS), its parameter sums in the blocks from `pq_block0` on
  int n_elems, pq_blocks, pq_block0;
};

// Launch shape: 32 elements per block x 8 lanes; lane s < 6 adds the slabs of stored stage s (independent loads, eight
// in flight: one thread walking all 6 x 40 slabs was a 240-deep chain of L2 latencies), the stage images of an element
// meet in LDS and are weighted in stage order.
__global__ __launch_bounds__(256) void mlp_adjoint_reduce_kernel(MlpReduceArgs r, int parity, int stage = 0) {
  __shared__ float ks[8][33];
  __shared__ double red[8][32];
  const int p2 = parity ^ 1;
  const AdjCtrl k = *reinterpret_cast<const AdjCtrl*>(r.ctrl + p2 * ADJ_CTRL_STRIDE);
  if (k.c.phase == 4 && k.commit == 0) return;
  const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int e = blockIdx.x * 32 + el;
  const int n_slots = k.mode == 0 ? 1 : k.mode == 1 ? 2 : MADJ_SLOTS;
  const bool layer2 = e < MADJ_P2;
  if (stage != 2) {
    float sum = 0.f;
    if (e < r.n_elems && sl < n_slots) {
      auto plane = [](int blk) { return blk == 0 ? 0 : blk - 4; };
      if (k.mode == 2 && sl == 0 && !(k.six & ADJ_FRESH0)) sum = r.kst[(int64_t)plane(k.src0) * MADJ_ELEMS + e];
      else {
        const float* base = (layer2 ? r.part2 + e : r.part1 + (e - MADJ_P2)) + (int64_t)sl * r.sps * (layer2 ? MADJ_P2 : MADJ_P1);
        const int64_t stride = layer2 ? MADJ_P2 : MADJ_P1;
        for (int b0 = 0; b0 < r.sps; b0 += 8) {
          float v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = b0 + u < r.sps ? base[(int64_t)(b0 + u) * stride] : 0.f;
#pragma unroll
          for (int u = 0; u < 8; ++u) sum += v[u];
        }
        if (k.mode == 2 && sl == 0) r.kst[e] = sum;
        if (k.mode == 2 && sl == 5) r.kst[(int64_t)plane(k.six & 15) * MADJ_ELEMS + e] = sum;
      }
    }
    ks[sl][el] = sum;
  }
  __syncthreads();
  double qv[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  if (sl == 0) {
    if (stage != 1 && k.mode == 3 && e == 0 && r.pq_block0 == 0) {
      double vt = 0.0;
      if (stage == 2) vt = r.sums_in[4];
      else
        for (int b = 0; b < r.n_wg; ++b) vt += r.partial[((int64_t)p2 * r.n_wg_max + b) * ADJ_NS + 4];
      r.carry[0] = (double)((float)k.T + (float)vt);
    }
    if (e < r.n_elems) {
      float S = 0.f, E = 0.f;
      if (stage == 2) {
        S = (float)r.sums_in[ADJ_NS + e]; E = (float)r.sums_in[ADJ_NS + MADJ_ELEMS + e];
      } else {
        float wS[7], wE[7];
        adj_stage_weights(k.mode, (float)k.c.dt_try, (float)k.x_end, wS, wE);
        for (int slot = 0; slot < n_slots; ++slot) {
          const int stg = k.mode <= 1 ? slot : (slot == 0 ? 0 : slot + 1);
          S = __builtin_fmaf(wS[stg], ks[slot][el], S);
          E = __builtin_fmaf(wE[stg], ks[slot][el], E);
        }
      }
      if (stage == 1) {
        r.sums_out[e] = (double)S; r.sums_out[MADJ_ELEMS + e] = (double)E;
        r.prevS[parity * MADJ_ELEMS + e] = S;                     // this shard's own increment, for its own running total
      } else {
        const int col = layer2 ? e % 129 : (e - MADJ_P2) % 33;
        const int tensor = layer2 ? (col == 128 ? 3 : 2) : (col == 32 ? 1 : 0);      // 0 W1, 1 b1, 2 W2, 3 b2 (torch's order)
        double q0 = 0.0, q1 = 0.0;
        if (stage == 2) {
          // norms and commit on the GLOBAL images; the shard's own running total takes its own (local) increments
          const float gn = adj_param_element(k, r.rtol, r.atol, r.Gn[e], r.prevSn[p2 * MADJ_ELEMS + e], S, E, q0, q1);
          if (k.commit) {
            r.Gn[e] = gn;
            r.G[e] += k.commit == 1 ? r.prevS[p2 * MADJ_ELEMS + e] : r.prevS[parity * MADJ_ELEMS + e];
          }
          r.prevSn[parity * MADJ_ELEMS + e] = S;
        } else {
          const float gn = adj_param_element(k, r.rtol, r.atol, r.G[e], r.prevS[p2 * MADJ_ELEMS + e], S, E, q0, q1);
          if (k.commit) r.G[e] = gn;
          r.prevS[parity * MADJ_ELEMS + e] = S;
        }
        qv[2 * tensor] = q0; qv[2 * tensor + 1] = q1;
      }
    }
  }
  if (stage == 1) return;
  if (k.mode == 3) return;
  if (sl == 0)
#pragma unroll
    for (int i = 0; i < 8; ++i) red[i][el] = qv[i];
  __syncthreads();
  if (threadIdx.x < 8) {
    const int i = threadIdx.x;
    double t = 0.0;
    for (int x = 0; x < 32; ++x) t += red[i][x];
    r.pq[((int64_t)p2 * r.pq_blocks + r.pq_block0 + blockIdx.x) * 8 + i] = t;
  }
}

// ------------------------------------------------------------------------------------------ small batches: reduction + R in one
// Up to MADJ_SMALL_MAX_ROWS series (at 256 the two forms take the same time) the factor rows of an attempt are few: the split-K reduction (6 x sps
// slab partials through memory) and the R kernel (1165 blocks over those partials) are two launches of fixed latency --
// 12 + 6 us plus two launch gaps per attempted step at 32 series (profiles/r04_k4am_32_seminorm_kernel_stats_before.csv), a
// fifth of the step.  Here ONE launch does both: a wave owns a 16 x 16 tile of [dW2 | db2] (144 tiles) or [dW1 | db1] (24),
// forms the stage images K_s = G_s^T X_s of its tile for the stored stages straight from the factor rows (K = the series,
// v_mfma_f32_16x16x4_f32, operands by plain loads: the rows are L2-resident), and goes on with the R kernel's work on the
// four elements each lane then holds: S = sum wS[s] K_s, E = sum wE[s] K_s, commit, norm sums.  42 blocks of parameter
// sums instead of 1165 for the next launch's prologue.
constexpr int64_t MADJ_SMALL_MAX_ROWS = 128;
constexpr int MADJ_SMALL_TILES = 16 * 9 + 8 * 3, MADJ_SMALL_BLOCKS = MADJ_SMALL_TILES / 4;

struct MlpSmallArgs {
  MlpReduceArgs r;
  const float* G2; const float* U; const float* G1; const float* Z;
  int64_t rows_per_stage, B;
};

__global__ __launch_bounds__(256) void mlp_adjoint_small_reduce_kernel(MlpSmallArgs a, int parity) {
  __shared__ double red[4][8];
  const MlpReduceArgs& r = a.r;
  const int p2 = parity ^ 1;
  const AdjCtrl k = *reinterpret_cast<const AdjCtrl*>(r.ctrl + p2 * ADJ_CTRL_STRIDE);
  if (k.c.phase == 4 && k.commit == 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = lane & 15, kq = lane >> 4;
  const int t = blockIdx.x * 4 + wave;
  const bool layer2 = t < 16 * 9;
  const int tt = layer2 ? t : t - 16 * 9;
  const int mt = layer2 ? tt / 9 : tt / 3, nt = layer2 ? tt % 9 : tt % 3;
  const int gc = layer2 ? G2_COLS : G1_COLS, xc = layer2 ? U_COLS : Z_COLS, ncols = layer2 ? 129 : 33;
  const float* G = layer2 ? a.G2 : a.G1;
  const float* X = layer2 ? a.U : a.Z;
  const int n_slots = k.mode == 0 ? 1 : k.mode == 1 ? 2 : MADJ_SLOTS;
  const int col_load = min(16 * nt + i, xc - 1);                  // (columns past the row: any valid address, the result is dropped)
  const int ksteps = (int)((a.B + 31) / 32) * 8;                   // rows past B hold zeros (rows_per_stage >= 4 * ksteps)
  f32x4 acc[MADJ_SLOTS];
#pragma unroll
  for (int s = 0; s < MADJ_SLOTS; ++s) acc[s] = f32x4{0.f, 0.f, 0.f, 0.f};
  // the operands of four K steps (16 series) of every stored stage at a time, the next group requested before this one is
  // used: the kernel is a chain of memory round trips (the rows were streamed out by the attempt kernel), so as many loads
  // as the registers hold are kept in flight -- at 32 series all of them at once
  const float* gp = G + (int64_t)kq * gc + 16 * mt + i;
  const float* xp = X + (int64_t)kq * xc + col_load;
  const int64_t gs = a.rows_per_stage * gc, xs = a.rows_per_stage * xc;
  // (the first and the last stage of an attempt live where the controller block says: AdjCtrl::src0 / six)
  const int blk0 = k.mode <= 1 ? 0 : k.src0, blk5 = k.mode <= 1 ? 5 : (k.six & 15);
  float av[2][MADJ_SLOTS][4], bv[2][MADJ_SLOTS][4];
  auto request = [&](int buf, int k0) {
#pragma unroll
    for (int s = 0; s < MADJ_SLOTS; ++s) {
      if (s >= n_slots) continue;
      const int blk = s == 0 ? blk0 : s == 5 ? blk5 : s;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        av[buf][s][u] = gp[blk * gs + (int64_t)(k0 + u) * 4 * gc];
        bv[buf][s][u] = xp[blk * xs + (int64_t)(k0 + u) * 4 * xc];
      }
    }
  };
  auto consume = [&](int buf) {
#pragma unroll
    for (int s = 0; s < MADJ_SLOTS; ++s) {
      if (s >= n_slots) continue;
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[s] = mfma16(av[buf][s][u], bv[buf][s][u], acc[s]);
    }
  };
  request(0, 0);
  for (int k0 = 0; k0 < ksteps; k0 += 8) {                          // (ksteps is a multiple of 8)
    request(1, k0 + 4);
    consume(0);
    if (k0 + 8 < ksteps) request(0, k0 + 8);
    consume(1);
  }
  if (k.mode == 3 && t == 0 && lane == 0) {
    double vt = 0.0;
    for (int b = 0; b < r.n_wg; ++b) vt += r.partial[((int64_t)p2 * r.n_wg_max + b) * ADJ_NS + 4];
    r.carry[0] = (double)((float)k.T + (float)vt);
  }
  float wS[7], wE[7];
  adj_stage_weights(k.mode, (float)k.c.dt_try, (float)k.x_end, wS, wE);
  double qv[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  const int col = 16 * nt + i;                                     // D fragment: lane (n = i, q = kq), register rr <-> row 4 kq + rr
  if (col < ncols) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      float S = 0.f, E = 0.f;
#pragma unroll
      for (int slot = 0; slot < MADJ_SLOTS; ++slot) {
        if (slot >= n_slots) continue;
        const int stage = k.mode <= 1 ? slot : (slot == 0 ? 0 : slot + 1);
        float ws = wS[0], we = wE[0];
#pragma unroll
        for (int j = 1; j < 7; ++j) { ws = stage == j ? wS[j] : ws; we = stage == j ? wE[j] : we; }
        S = __builtin_fmaf(ws, acc[slot][rr], S);
        E = __builtin_fmaf(we, acc[slot][rr], E);
      }
      const int m = 16 * mt + 4 * kq + rr;
      const int e = layer2 ? m * 129 + col : MADJ_P2 + m * 33 + col;
      const int tensor = layer2 ? (col == 128 ? 3 : 2) : (col == 32 ? 1 : 0);      // 0 W1, 1 b1, 2 W2, 3 b2 (torch's order)
      double q0 = 0.0, q1 = 0.0;
      const float gn = adj_param_element(k, r.rtol, r.atol, r.G[e], r.prevS[p2 * MADJ_ELEMS + e], S, E, q0, q1);
      if (k.commit) r.G[e] = gn;
      r.prevS[parity * MADJ_ELEMS + e] = S;
#pragma unroll
      for (int tz = 0; tz < 4; ++tz) if (tensor == tz) { qv[2 * tz] += q0; qv[2 * tz + 1] += q1; }
    }
  }
  if (k.mode == 3) return;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) qv[j] += __shfl_xor(qv[j], off, 64);
  }
  if (lane == 0)
#pragma unroll
    for (int j = 0; j < 8; ++j) red[wave][j] = qv[j];
  __syncthreads();
  if (threadIdx.x < 8)
    r.pq[((int64_t)p2 * r.pq_blocks + blockIdx.x) * 8 + threadIdx.x] =
        (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// the "1" columns of the U and Z rows (bias gradients = column sums of G): written once per backward pass
__global__ __launch_bounds__(256) void madj_ones_kernel(float* __restrict__ U, float* __restrict__ Z, int64_t rows) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r < rows) { U[r * U_COLS + 128] = 1.f; Z[r * Z_COLS + 32] = 1.f; }
}

// from mlp_grad_reduce.hip: the split-K reduction of one attempt's factor rows, both layers, gated by the controller block
int launch_mlp_adjoint_factor_reduce(const float* G2, const float* U, const float* G1, const float* Z, int64_t rows_per_stage,
                                     int sps, int64_t rows_per_slab, float* part2, float* part1, const unsigned char* ctrl,
                                     int parity, hipStream_t s, const float* G2hi = nullptr, float* part2hi = nullptr);

static inline size_t m256(size_t x) { return (x + 255) / 256 * 256; }

}  // namespace cde

#ifdef CDE_PHASE_TRACE
// debug builds only (cde_common.h, "phase trace"): the stamp ring of dopri5_mlp_adjoint_attempt, [ring][workgroup][slot]
extern "C" int cde_debug_k4am_phase_trace(void* host_out, size_t bytes) {
  if (bytes > sizeof(unsigned long long) * cde::TRACE_RING * cde::TRACE_BLOCKS * cde::TRACE_SLOTS) return CDE_ERR_SHAPE;
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(cde::k4am_phase_trace), bytes) == hipSuccess ? CDE_OK : CDE_ERR_LAUNCH;
}
#endif

// ================================================================================================ C ABI
namespace {
struct MadjLayout {
  int64_t n_tiles, rows_per_stage, rows_per_slab;
  int sps, nwave, n_wg;
  bool split, split8, small;
  size_t partial, pq, carry, image, state, G, prev, Gn, prevn, slopes, stash, kst, part2, part1, U, G2, G1, Z, trace, trace_all, total;
  size_t rec, cq, ktp, gx, total_dcontrol;                         // control gradients: behind everything else
  int n_cblocks, ct;
  bool upper;                                                      // 32 hidden units x 16 channels: a second layer-2 instance
  size_t G_hi, prev_hi, kst_hi, part2_hi, G2_hi;
  int pq_blocks;
};
MadjLayout madj_layout(int64_t B, int64_t H, int64_t C) {
  using namespace cde;
  MadjLayout L;
  L.n_tiles = (B + 15) / 16;
  // 16384 series fill the GPU's 1024 SIMDs with one wave each; CDE_OPT_K4AM_WAVES = 8 runs the large-batch form on any batch
  // (tests: the 8-wave kernel at a size the CPU oracle can follow)
  L.upper = mlp_shape_upper(C, H, 4);
  L.pq_blocks = L.upper ? 2 * MADJ_RBLOCKS : MADJ_RBLOCKS;
  L.nwave = (L.n_tiles > 1024 || option(CDE_OPT_K4AM_WAVES) == 8) ? 8 : 4;
  // up to MADJ_SPLIT_MAX_TILES tiles (one workgroup per CU in a single round): four waves per tile, the evaluation's middle
  // split four ways
  // (8-channel tiles: the eight-wave form takes ~75 us per round of 256 tiles, the one-wave-per-tile forms 260-420 us
  //  whatever the batch -- measured per attempted step: 8192 series 320 -> 210 us, 12288: 341 -> 303, 16384: 360 vs 394;
  //  CDE_OPT_K4AM_S8_TILES overrides the threshold, for measurements)
  const int64_t s8_req = option(CDE_OPT_K4AM_S8_TILES);                   // (-1: the default; an override can only LOWER the measured limit)
  const int64_t s8_tiles = s8_req < 0 || s8_req > MADJ_S8_MAX_TILES ? MADJ_S8_MAX_TILES : s8_req;
  const bool s8_shape = C <= MC && !option(CDE_OPT_K4AM_SPLIT4);
  L.split = L.n_tiles <= (s8_shape && s8_tiles > MADJ_SPLIT_MAX_TILES ? s8_tiles : MADJ_SPLIT_MAX_TILES) &&
            !option(CDE_OPT_K4AM_NO_SPLIT);
  // ... eight (two per SIMD, everything split eight ways: mlp_adjoint_eval_split8) when the control fits the 32 x 8 tiling
  L.split8 = L.split && C <= MC && !option(CDE_OPT_K4AM_SPLIT4);
  // a few hundred rows per attempt: factor reduction + R in one launch (mlp_adjoint_small_reduce_kernel)
  L.small = B <= MADJ_SMALL_MAX_ROWS && !option(CDE_OPT_K4AM_NO_SMALL_REDUCE) && !L.upper;
  L.n_wg = L.split ? (int)L.n_tiles : (int)((L.n_tiles + L.nwave - 1) / L.nwave);
  int64_t sps = (B + 63) / 64;                     // (measured at 4096 series: 40 slabs 193 us per attempt, 16: 209, 6: 284;
  const int64_t sps_req = option(CDE_OPT_K4AM_SPS);                              // (measurements; 0: the default)
  // (20 slabs up to 4096 series are 4 % faster -- 112.4 -> 108.0 us per attempt at 4096 series, half the partials for the R
  //  kernel -- and NOT used: longer float32 chains per slab move the parameter blocks' error estimate, and on the 2048-series
  //  config-5 replay 95.5 % instead of >= 97 % of the mixed-norm error ratios stayed within 2 % of the float64 oracle's)
  const int64_t sps_default = MADJ_MAX_SPS;
  const int64_t sps_max = sps_req >= 4 && sps_req <= MADJ_MAX_SPS ? sps_req : sps_default;
  L.sps = (int)(sps < 4 ? 4 : sps > sps_max ? sps_max : sps);        //  at 64 series: 4 slabs 138, 1: 146)
  L.rows_per_slab = ((B + L.sps - 1) / L.sps + 15) / 16 * 16;
  L.rows_per_stage = L.rows_per_slab * L.sps;
  const size_t rows = (size_t)MADJ_FSLOTS * L.rows_per_stage;
  L.partial = m256(2 * ADJ_CTRL_STRIDE);
  L.pq = L.partial + m256((size_t)2 * L.n_wg * ADJ_NS * sizeof(double));
  L.carry = L.pq + m256((size_t)2 * L.pq_blocks * 8 * sizeof(double));
  L.image = L.carry + 256;
  L.state = L.image + m256(mlp_adjoint_image_bytes());
  L.G = L.state + m256((size_t)2 * 4 * B * H * sizeof(float));
  L.prev = L.G + m256((size_t)MADJ_ELEMS * sizeof(float));
  L.Gn = L.prev + m256((size_t)2 * MADJ_ELEMS * sizeof(float));        // sharded: the GLOBAL running totals and S sums
  L.prevn = L.Gn + m256((size_t)MADJ_ELEMS * sizeof(float));
  // (the upper instance's blocks sit inside the ranges the first launch zeroes: [G, slopes) and [U, trace))
  const size_t up = L.upper ? 1 : 0;
  L.G_hi = L.prevn + m256((size_t)2 * MADJ_ELEMS * sizeof(float));
  L.prev_hi = L.G_hi + up * m256((size_t)MADJ_ELEMS * sizeof(float));
  L.slopes = L.prev_hi + up * m256((size_t)2 * MADJ_ELEMS * sizeof(float));
  L.stash = L.slopes + m256((size_t)L.n_tiles * 7 * 4 * 64 * 16);
  L.kst = L.stash + m256((size_t)3 * B * (2 * H + 4) * sizeof(float));
  L.kst_hi = L.kst + m256((size_t)3 * MADJ_ELEMS * sizeof(float));
  L.part2 = L.kst_hi + up * m256((size_t)3 * MADJ_ELEMS * sizeof(float));
  L.part2_hi = L.part2 + m256((size_t)MADJ_SLOTS * L.sps * MADJ_P2 * sizeof(float));
  L.part1 = L.part2_hi + up * m256((size_t)MADJ_SLOTS * L.sps * MADJ_P2 * sizeof(float));
  L.U = L.part1 + m256((size_t)MADJ_SLOTS * L.sps * MADJ_P1 * sizeof(float));
  L.G2 = L.U + m256(rows * U_COLS * sizeof(float));
  L.G2_hi = L.G2 + m256(rows * G2_COLS * sizeof(float));
  L.G1 = L.G2_hi + up * m256(rows * G2_COLS * sizeof(float));
  L.Z = L.G1 + m256(rows * G1_COLS * sizeof(float));
  L.trace = L.Z + m256(rows * Z_COLS * sizeof(float));
  L.trace_all = L.trace + m256((size_t)CDE_DOPRI5_TRACE_STEPS * 3 * sizeof(double));
  L.total = L.trace_all + m256((size_t)ADJ_TRACE_ATTEMPTS * 5 * sizeof(double));
  // control gradients (cde_dopri5_adjoint_mlp_advance_dcontrol): the plain layout is a prefix
  L.ct = C > MC ? 16 : 8;
  L.n_cblocks = (int)((B * L.ct + 255) / 256);
  L.rec = L.total;
  L.cq = L.rec + m256(2 * ADJ_REC_STRIDE);
  L.ktp = L.cq + m256((size_t)2 * (L.n_cblocks + 1) * 2 * sizeof(double));
  L.gx = L.ktp + m256((size_t)2 * L.n_wg * 8 * sizeof(double));
  L.total_dcontrol = L.gx + m256((size_t)2 * B * L.ct * 8 * sizeof(float));
  return L;
}
}  // namespace

extern "C" size_t cde_dopri5_adjoint_mlp_workspace_bytes(int64_t B, int64_t C, int64_t H) {
  return B < 1 || H < 1 ? 0 : madj_layout(B, H, C).total;
}
extern "C" size_t cde_dopri5_adjoint_mlp_dcontrol_workspace_bytes(int64_t B, int64_t C, int64_t H) {
  return B < 1 || H < 1 ? 0 : madj_layout(B, H, C).total_dcontrol;
}
extern "C" size_t cde_dopri5_adjoint_mlp_trace_offset(int64_t B, int64_t C, int64_t H, int which) {
  const MadjLayout L = madj_layout(B, H, C);
  return which == 0 ? L.trace : L.trace_all;
}
extern "C" size_t cde_dopri5_adjoint_mlp_carry_offset(int64_t B, int64_t C, int64_t H) { return madj_layout(B, H, C).carry; }
// where the running totals live: layer 2 as [256][129] (row = padded (h, c), bias in column 128), then layer 1 as [128][33]
extern "C" size_t cde_dopri5_adjoint_mlp_gradient_offset(int64_t B, int64_t C, int64_t H) {
  return madj_layout(B, H, C).G;
}
// 32 hidden units x 16 channels: the layer-2 totals of hidden units 16..31, again [256][129]; 0 for every other shape
extern "C" size_t cde_dopri5_adjoint_mlp_gradient_upper_offset(int64_t B, int64_t C, int64_t H) {
  const MadjLayout L = madj_layout(B, H, C);
  return L.upper ? L.G_hi : 0;
}

namespace {
cde::MlpReduceArgs madj_reduce_args(unsigned char* base, const MadjLayout& L, double rtol, double atol, bool sharded) {
  cde::MlpReduceArgs q;
  q.ctrl = base; q.part2 = (const float*)(base + L.part2); q.part1 = (const float*)(base + L.part1); q.sps = L.sps;
  q.kst = (float*)(base + L.kst);
  q.G = (float*)(base + L.G); q.prevS = (float*)(base + L.prev);
  q.Gn = sharded ? (float*)(base + L.Gn) : nullptr; q.prevSn = sharded ? (float*)(base + L.prevn) : nullptr;
  q.pq = (double*)(base + L.pq);
  q.partial = (double*)(base + L.partial); q.n_wg = L.n_wg; q.n_wg_max = L.n_wg; q.carry = (double*)(base + L.carry);
  q.sums_out = nullptr; q.sums_in = nullptr;
  q.rtol = (float)rtol; q.atol = (float)atol;
  q.n_elems = cde::MADJ_ELEMS; q.pq_blocks = L.pq_blocks; q.pq_block0 = 0;
  return q;
}
}  // namespace

static int dopri5_adjoint_mlp_advance_impl(const void* coeffs, const void* knots, int64_t n_intervals, int degree,
                                           const void* W1, const void* bias1, int64_t width, const void* W2,
                                           const void* bias2, int act, const void* y_init, const void* a_init,
                                           double s0, double s1, const double* jump_s, int64_t n_jump, double rtol,
                                           double atol, double safety, double ifactor, double dfactor, int norm_kind,
                                           void* a_out, int64_t B, int64_t C, int64_t H, int dtype, int first_interval,
                                           void* workspace, size_t workspace_bytes, int64_t first_launch,
                                           int64_t n_launches, void* stream, const double* reduced_sums, int64_t B_global,
                                           void* grad_coeffs = nullptr, int64_t control_numel = 0,
                                           void* grad_knots = nullptr) {
  using namespace cde;
  const bool sharded = reduced_sums != nullptr || B_global > 0;
  const bool dctrl = grad_coeffs != nullptr;
  if (dctrl && (sharded || control_numel < 1)) return CDE_ERR_UNSUPPORTED;     // control gradients: one controller per solve
  if (grad_knots && !dctrl) return CDE_ERR_UNSUPPORTED;
  if (sharded && (n_launches != 1 || B_global < B)) return CDE_ERR_SHAPE;        // sharded: one launch per all-reduce
  if (first_launch > 0 && sharded && !reduced_sums) return CDE_ERR_NULL;
  if (B < 1 || C < 1 || H < 1 || width < 1 || n_intervals < 1 || n_launches < 0 || n_jump < 0 || !(s0 < s1)) return CDE_ERR_SHAPE;
  if (dtype != CDE_F32) return dtype == CDE_F64 ? CDE_ERR_UNSUPPORTED : CDE_ERR_DTYPE;
  if (!mlp_shape_ok(C, H, width) && !mlp_shape_upper(C, H, width)) return CDE_ERR_UNSUPPORTED;
  if (mlp_shape_upper(C, H, width) && sharded) return CDE_ERR_UNSUPPORTED;       // (one GPU's batch: no image exchange for the upper half)
  if (act != CDE_ACT_NONE && act != CDE_ACT_TANH) return CDE_ERR_UNSUPPORTED;
  if (degree != CDE_PATH_CUBIC && degree != CDE_PATH_LINEAR) return CDE_ERR_UNSUPPORTED;
  if (norm_kind != 0 && norm_kind != 1) return CDE_ERR_UNSUPPORTED;
  if (!coeffs || !knots || !W1 || !bias1 || !W2 || !bias2 || !y_init || !a_init || !a_out || !workspace) return CDE_ERR_NULL;
  if (n_jump > 0 && !jump_s) return CDE_ERR_NULL;
  if (workspace_bytes < (dctrl ? cde_dopri5_adjoint_mlp_dcontrol_workspace_bytes(B, C, H) : cde_dopri5_adjoint_mlp_workspace_bytes(B, C, H)))
    return CDE_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  unsigned char* base = (unsigned char*)workspace;
  const MadjLayout L = madj_layout(B, H, C);
  if (dctrl && grad_knots && L.n_wg > ADJ_KT_MAX_WG * 64) return CDE_ERR_UNSUPPORTED;
  MlpAdjArgs g;
  g.coeffs = (const float*)coeffs; g.knots = (const float*)knots; g.n_intervals = n_intervals;
  g.img = (const float*)(base + L.image);
  g.dims = Dims{(int)H, (int)C};
  g.B = B; g.n_tiles = L.n_tiles; g.rows_per_stage = L.rows_per_stage;
  g.ctrl = base;
  g.partial = (double*)(base + L.partial); g.pq = (double*)(base + L.pq);
  g.state = (float*)(base + L.state);
  g.y_init = (const float*)y_init; g.a_init = (const float*)a_init; g.a_out = (float*)a_out;
  g.slopes = (float*)(base + L.slopes);
  g.U = (float*)(base + L.U); g.G2 = (float*)(base + L.G2); g.G1 = (float*)(base + L.G1); g.Z = (float*)(base + L.Z);
  g.stash_y = (float*)(base + L.stash); g.stash_a = g.stash_y + 3 * B * H; g.stash_t = g.stash_a + 3 * B * H;
  g.n_wg_max = L.n_wg;
  { const int64_t e = option(CDE_OPT_K4AM_NO_FSAL); g.dbg = e == 1 ? 2 : e == 2 ? 4 : e == 3 ? 8 : 0; }   // bit 1: evaluate every first stage (tests compare the two); 4 / 8: after accepted / rejected steps only
#ifdef CDE_PHASE_TRACE
  { const char* d = getenv("CDE_K4AM_DBG"); g.dbg |= d ? atoi(d) & 1 : 0; }      // timing experiments (wrong gradients!)
#endif
  g.n_pq = L.small && (!sharded || norm_kind == 1) ? MADJ_SMALL_BLOCKS : L.pq_blocks;
  g.pq_blocks = L.pq_blocks;
  g.G2hi = L.upper ? (float*)(base + L.G2_hi) : nullptr;
  g.com.s0 = s0; g.com.s1 = s1; g.com.jump_s = jump_s; g.com.n_jump = n_jump;
  g.com.rtol = rtol; g.com.atol = atol; g.com.safety = safety; g.com.ifactor = ifactor; g.com.dfactor = dfactor;
  g.com.n_state = (B_global > 0 ? B_global : B) * H;
  g.ext_sums = reduced_sums;
  g.com.n_pt = dctrl ? (grad_knots ? 6 : 5) : 4;
  g.com.n_param[0] = width * H; g.com.n_param[1] = width; g.com.n_param[2] = H * C * width; g.com.n_param[3] = H * C;
  g.com.n_param[4] = dctrl ? control_numel : 1; g.com.n_param[5] = grad_knots ? n_intervals + 1 : 1;
  g.gx = (float*)(base + L.gx); g.rec = base + L.rec; g.cq = (const double*)(base + L.cq); g.n_cblocks = L.n_cblocks;
  g.ktp = (double*)(base + L.ktp); g.with_knots = grad_knots ? 1 : 0;
  g.com.norm_kind = norm_kind;
  g.com.trace = (double*)(base + L.trace);
  g.com.trace_all = (double*)(base + L.trace_all);
  g.com.carry = (double*)(base + L.carry);
  if (first_launch == 0) {
    zero_async(base, 2 * ADJ_CTRL_STRIDE, s);                                                     // phase 0
    if (first_interval & 1) {
      // vjp_t, the running totals and the factor rows (padding rows must hold zeros; the "1" columns are set below)
      // (bit 1: the caller has set vjp_t itself -- output-time gradients, as for K4a: cde_dopri5_adjoint_mlp_carry_offset)
      if (!(first_interval & 2)) zero_async(base + L.carry, 256, s);
      zero_async(base + L.G, L.slopes - L.G, s);
      zero_async(base + L.U, L.trace - L.U, s);
      const int64_t rows = (int64_t)MADJ_FSLOTS * L.rows_per_stage;
      madj_ones_kernel<<<(unsigned)((rows + 255) / 256), 256, 0, s>>>(g.U, g.Z, rows);
      const int rc = launch_mlp_adjoint_images(W1, bias1, width, W2, bias2, C, H, (float*)(base + L.image), s);
      if (rc != CDE_OK) return rc;
    }
    if (dctrl) zero_async(base + L.rec, L.gx - L.rec, s);                         // stage records, control norm sums, time terms
  }
  AdjControlArgs cr;
  cr.ctrl = base; cr.rec = base + L.rec; cr.gx = (const float*)(base + L.gx); cr.G = (float*)grad_coeffs;
  cr.knots = (const float*)knots; cr.cq = (double*)(base + L.cq); cr.B = B; cr.n_intervals = n_intervals;
  cr.C = (int)C; cr.degree = degree; cr.norm_kind = norm_kind; cr.rtol = (float)rtol; cr.atol = (float)atol;
  cr.G_knots = (float*)grad_knots; cr.ktp = (const double*)(base + L.ktp); cr.n_wg = L.n_wg; cr.kt_stride = L.n_wg;
  if (dctrl && L.split8 && L.n_tiles > MADJ_SPLIT_MAX_TILES) cr.n_wg = (int)((L.n_tiles + 3) / 4);
  // sharded under "seminorm": only the 8 state sums travel between the shards (cde_dopri5_adjoint_mlp_state_sums /
  // _apply_state_sums); the gradient images are reduced, committed and returned per shard like an unsharded solve's
  const bool images_local = sharded && norm_kind == 1;
  MlpReduceArgs r = madj_reduce_args(base, L, rtol, atol, sharded && !images_local);
  // control gradients on eight-channel tiles take the four-wave form where the eight-wave form would run -- up to the 256 tiles
  // that form is used (and tested) on; beyond that one wave per tile, on fewer workgroups than the layout provides for
  const bool dctrl_one_wave = dctrl && L.split8 && L.n_tiles > MADJ_SPLIT_MAX_TILES;
  const int grid = dctrl_one_wave ? (int)((L.n_tiles + 3) / 4) : L.n_wg;
  r.n_wg = grid;
  MlpReduceArgs r_hi = r;
  if (L.upper) {
    r_hi.part2 = (const float*)(base + L.part2_hi); r_hi.part1 = nullptr;
    r_hi.kst = (float*)(base + L.kst_hi); r_hi.G = (float*)(base + L.G_hi); r_hi.prevS = (float*)(base + L.prev_hi);
    r_hi.n_elems = MADJ_P2; r_hi.pq_block0 = MADJ_RBLOCKS;
  }
  MlpSmallArgs sm;
  sm.r = r; sm.G2 = g.G2; sm.U = g.U; sm.G1 = g.G1; sm.Z = g.Z; sm.rows_per_stage = L.rows_per_stage; sm.B = B;
  // after an attempt launch: the split-K reduction of its factor rows + the R kernel, or (small batches) both in one launch
  auto after_attempt = [&](int parity) -> int {
    if (sharded && !images_local) return CDE_OK;   // the caller goes on with cde_dopri5_adjoint_mlp_pending_sums / _apply_reduced
    if (L.small && !(sharded && !images_local)) {
      mlp_adjoint_small_reduce_kernel<<<MADJ_SMALL_BLOCKS, 256, 0, s>>>(sm, parity);
      return CDE_OK;
    }
    const int rc = launch_mlp_adjoint_factor_reduce(g.G2, g.U, g.G1, g.Z, L.rows_per_stage, L.sps, L.rows_per_slab,
                                                    (float*)(base + L.part2), (float*)(base + L.part1), base, parity, s,
                                                    g.G2hi, L.upper ? (float*)(base + L.part2_hi) : nullptr);
    if (rc != CDE_OK) return rc;
    mlp_adjoint_reduce_kernel<<<MADJ_RBLOCKS, 256, 0, s>>>(r, parity);
    if (L.upper) mlp_adjoint_reduce_kernel<<<MADJ_RBLOCKS, 256, 0, s>>>(r_hi, parity);
    return CDE_OK;
  };
  // ... and (control gradients) the kernel that owns the coefficient / knot-time blocks (cde_dopri_ctl.h)
  auto control_after = [&](int parity) {
    if (!dctrl) return;
    if (degree == CDE_PATH_CUBIC) {
      if (C > MC) adjoint_control_kernel<CDE_PATH_CUBIC, 16><<<L.n_cblocks, 256, 0, s>>>(cr, parity);
      else adjoint_control_kernel<CDE_PATH_CUBIC, 8><<<L.n_cblocks, 256, 0, s>>>(cr, parity);
    } else {
      if (C > MC) adjoint_control_kernel<CDE_PATH_LINEAR, 16><<<L.n_cblocks, 256, 0, s>>>(cr, parity);
      else adjoint_control_kernel<CDE_PATH_LINEAR, 8><<<L.n_cblocks, 256, 0, s>>>(cr, parity);
    }
  };
  const size_t lds_bytes = (size_t)ADJ_LDS_FLOATS * sizeof(float) + (size_t)MADJ_NSUM * 8 * sizeof(double) +
                           (L.split ? (size_t)MADJ_XBUF_FLOATS * sizeof(float) : 0);
#define CDE_MADJ_LAUNCH(D, A, CTV, NWV, SPL, HIV)                                                                    \
  do {                                                                                                               \
    (void)hipFuncSetAttribute((const void*)dopri5_mlp_adjoint_attempt<D, A, CTV, NWV, SPL, false, HIV>,              \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);                           \
    (void)hipFuncSetAttribute((const void*)dopri5_mlp_adjoint_attempt<D, A, CTV, NWV, SPL, true, HIV>,               \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);                           \
    for (int64_t i = 0; i < n_launches; ++i) {                                                                       \
      const int parity = (int)((first_launch + i) & 1);                                                              \
      if (dctrl) dopri5_mlp_adjoint_attempt<D, A, CTV, NWV, SPL, true, HIV><<<grid, 64 * NWV, lds_bytes, s>>>(g, parity); \
      else dopri5_mlp_adjoint_attempt<D, A, CTV, NWV, SPL, false, HIV><<<L.n_wg, 64 * NWV, lds_bytes, s>>>(g, parity); \
      const int rc = after_attempt(parity);                                                                          \
      if (rc != CDE_OK) return rc;                                                                                   \
      control_after(parity);                                                                                         \
    }                                                                                                                \
  } while (0)
#define CDE_MADJ_LAUNCH_S8(D, A)                                                                                     \
  do {                                                                                                               \
    (void)hipFuncSetAttribute((const void*)dopri5_mlp_adjoint_attempt_s8<D, A>,                                      \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);                           \
    for (int64_t i = 0; i < n_launches; ++i) {                                                                       \
      const int parity = (int)((first_launch + i) & 1);                                                              \
      dopri5_mlp_adjoint_attempt_s8<D, A><<<L.n_wg, 512, lds_bytes, s>>>(g, parity);                                 \
      const int rc = after_attempt(parity);                                                                          \
      if (rc != CDE_OK) return rc;                                                                                   \
    }                                                                                                                \
  } while (0)
#define CDE_MADJ_W(D, A, CTV, HIV)                                                                                   \
  do {                                                                                                               \
    if (dctrl_one_wave) CDE_MADJ_LAUNCH(D, A, CTV, 4, false, HIV);                                                   \
    else if (L.split) CDE_MADJ_LAUNCH(D, A, CTV, 4, true, HIV);                                                      \
    else if (L.nwave == 8) CDE_MADJ_LAUNCH(D, A, CTV, 8, false, HIV);                                                \
    else CDE_MADJ_LAUNCH(D, A, CTV, 4, false, HIV);                                                                  \
  } while (0)
#define CDE_MADJ(D, A)                                                                                               \
  do {                                                                                                               \
    if (L.upper) CDE_MADJ_W(D, A, 16, true);                                                                         \
    else if (C > MC) CDE_MADJ_W(D, A, 16, false);                                                                    \
    else if (L.split8 && !dctrl) CDE_MADJ_LAUNCH_S8(D, A);       /* (control gradients: the four-wave form) */          \
    else CDE_MADJ_W(D, A, 8, false);                                                                                 \
  } while (0)
  if (act == CDE_ACT_NONE) {
    if (degree == CDE_PATH_CUBIC) CDE_MADJ(CDE_PATH_CUBIC, CDE_ACT_NONE); else CDE_MADJ(CDE_PATH_LINEAR, CDE_ACT_NONE);
  } else {
    if (degree == CDE_PATH_CUBIC) CDE_MADJ(CDE_PATH_CUBIC, CDE_ACT_TANH); else CDE_MADJ(CDE_PATH_LINEAR, CDE_ACT_TANH);
  }
#undef CDE_MADJ
#undef CDE_MADJ_W
#undef CDE_MADJ_LAUNCH
#undef CDE_MADJ_LAUNCH_S8
  return check_launch();
}

extern "C" int cde_dopri5_adjoint_mlp_advance(const void* coeffs, const void* knots, int64_t n_intervals, int degree,
                                              const void* W1, const void* bias1, int64_t width, const void* W2,
                                              const void* bias2, int act, const void* y_init, const void* a_init,
                                              double s0, double s1, const double* jump_s, int64_t n_jump, double rtol,
                                              double atol, double safety, double ifactor, double dfactor, int norm_kind,
                                              void* a_out, int64_t B, int64_t C, int64_t H, int dtype, int first_interval,
                                              void* workspace, size_t workspace_bytes, int64_t first_launch,
                                              int64_t n_launches, void* stream) {
  return dopri5_adjoint_mlp_advance_impl(coeffs, knots, n_intervals, degree, W1, bias1, width, W2, bias2, act, y_init, a_init,
                                         s0, s1, jump_s, n_jump, rtol, atol, safety, ifactor, dfactor, norm_kind, a_out, B, C,
                                         H, dtype, first_interval, workspace, workspace_bytes, first_launch, n_launches,
                                         stream, nullptr, 0);
}

// K4am with control gradients (round 6; cde_mi355x.h): the coefficient tensor -- and optionally the knot times -- as further
// blocks of the adjoint state.  Eight-channel tiles then run the four-wave / one-wave forms (not the eight-wave one).
extern "C" int cde_dopri5_adjoint_mlp_advance_dcontrol(const void* coeffs, const void* knots, int64_t n_intervals, int degree,
                                                       const void* W1, const void* bias1, int64_t width, const void* W2,
                                                       const void* bias2, int act, const void* y_init, const void* a_init,
                                                       double s0, double s1, const double* jump_s, int64_t n_jump,
                                                       double rtol, double atol, double safety, double ifactor,
                                                       double dfactor, int norm_kind, void* a_out, int64_t B, int64_t C,
                                                       int64_t H, int dtype, int first_interval, void* workspace,
                                                       size_t workspace_bytes, int64_t first_launch, int64_t n_launches,
                                                       void* grad_coeffs, int64_t control_numel, void* grad_knots,
                                                       void* stream) {
  if (!grad_coeffs) return CDE_ERR_NULL;
  return dopri5_adjoint_mlp_advance_impl(coeffs, knots, n_intervals, degree, W1, bias1, width, W2, bias2, act, y_init, a_init,
                                         s0, s1, jump_s, n_jump, rtol, atol, safety, ifactor, dfactor, norm_kind, a_out, B, C,
                                         H, dtype, first_interval, workspace, workspace_bytes, first_launch, n_launches,
                                         stream, nullptr, 0, grad_coeffs, control_numel, grad_knots);
}

// ---- one step controller for a batch sharded over GPUs, two-layer field (round 4; the one-layer protocol of
// dopri5_adjoint.hip, see cde_mi355x.h).  Per attempted step n every shard runs
//   cde_dopri5_adjoint_mlp_advance_sharded   ONE attempt launch (n > 0: with the reduced buffer of step n - 1)
//   cde_dopri5_adjoint_mlp_pending_sums      its 8 state sums + its S and E gradient images -> `sums` (doubles)
//   [all-reduce `sums` over the shards]
//   cde_dopri5_adjoint_mlp_apply_reduced     commit + parameter norms on the reduced images
extern "C" size_t cde_dopri5_adjoint_mlp_reduced_count(void) { return (size_t)cde::ADJ_NS + 2 * (size_t)cde::MADJ_ELEMS; }

extern "C" int cde_dopri5_adjoint_mlp_advance_sharded(const void* coeffs, const void* knots, int64_t n_intervals, int degree,
                                                      const void* W1, const void* bias1, int64_t width, const void* W2,
                                                      const void* bias2, int act, const void* y_init, const void* a_init,
                                                      double s0, double s1, const double* jump_s, int64_t n_jump,
                                                      double rtol, double atol, double safety, double ifactor,
                                                      double dfactor, int norm_kind, void* a_out, int64_t B, int64_t C,
                                                      int64_t H, int dtype, int first_interval, void* workspace,
                                                      size_t workspace_bytes, int64_t first_launch,
                                                      const double* reduced_sums, int64_t B_global, void* stream) {
  if (B_global < B) return CDE_ERR_SHAPE;
  return dopri5_adjoint_mlp_advance_impl(coeffs, knots, n_intervals, degree, W1, bias1, width, W2, bias2, act, y_init, a_init,
                                         s0, s1, jump_s, n_jump, rtol, atol, safety, ifactor, dfactor, norm_kind, a_out, B, C,
                                         H, dtype, first_interval, workspace, workspace_bytes, first_launch, 1, stream,
                                         reduced_sums, B_global);
}

namespace cde {
__global__ __launch_bounds__(64) void madj_state_sums_kernel(const double* __restrict__ partial, int n_wg,
                                                             double* __restrict__ out) {
  const int i = threadIdx.x;
  if (i >= ADJ_NS) return;
  double s = 0.0;
  for (int b = 0; b < n_wg; ++b) s += partial[ADJ_NS * b + i];
  out[i] = s;
}
}  // namespace cde

extern "C" int cde_dopri5_adjoint_mlp_pending_sums(void* workspace, size_t workspace_bytes, int64_t B, int64_t C, int64_t H,
                                                   int64_t total_launches, double* sums, void* stream) {
  using namespace cde;
  if (B < 1 || C < 1 || H < 1 || total_launches < 1) return CDE_ERR_SHAPE;
  if (!workspace || !sums) return CDE_ERR_NULL;
  if (workspace_bytes < cde_dopri5_adjoint_mlp_workspace_bytes(B, C, H)) return CDE_ERR_WORKSPACE;
  unsigned char* base = (unsigned char*)workspace;
  const MadjLayout L = madj_layout(B, H, C);
  hipStream_t s = (hipStream_t)stream;
  const int parity = (int)((total_launches - 1) & 1);               // the launch whose sums are pending
  const double* partial = (const double*)(base + L.partial) + (int64_t)(parity ^ 1) * L.n_wg * ADJ_NS;
  madj_state_sums_kernel<<<1, 64, 0, s>>>(partial, L.n_wg, sums);
  const int rc = launch_mlp_adjoint_factor_reduce((const float*)(base + L.G2), (const float*)(base + L.U),
                                                  (const float*)(base + L.G1), (const float*)(base + L.Z), L.rows_per_stage,
                                                  L.sps, L.rows_per_slab, (float*)(base + L.part2), (float*)(base + L.part1),
                                                  base, parity, s);
  if (rc != CDE_OK) return rc;
  MlpReduceArgs r = madj_reduce_args(base, L, 0.0, 0.0, true);
  r.sums_out = sums + ADJ_NS;
  mlp_adjoint_reduce_kernel<<<MADJ_RBLOCKS, 256, 0, s>>>(r, parity, 1);
  return check_launch();
}

extern "C" int cde_dopri5_adjoint_mlp_apply_reduced(void* workspace, size_t workspace_bytes, int64_t B, int64_t C, int64_t H,
                                                    double rtol, double atol, int64_t total_launches, const double* reduced,
                                                    void* stream) {
  using namespace cde;
  if (B < 1 || C < 1 || H < 1 || total_launches < 1) return CDE_ERR_SHAPE;
  if (!workspace || !reduced) return CDE_ERR_NULL;
  if (workspace_bytes < cde_dopri5_adjoint_mlp_workspace_bytes(B, C, H)) return CDE_ERR_WORKSPACE;
  unsigned char* base = (unsigned char*)workspace;
  const MadjLayout L = madj_layout(B, H, C);
  MlpReduceArgs r = madj_reduce_args(base, L, rtol, atol, true);
  r.sums_in = reduced;
  mlp_adjoint_reduce_kernel<<<MADJ_RBLOCKS, 256, 0, (hipStream_t)stream>>>(r, (int)((total_launches - 1) & 1), 2);
  return check_launch();
}

// The "seminorm" form: only the ADJ_NS state sums are pending on the other shards; after the all-reduce vjp_t at the end of
// an interval is redone from the reduced sums (see dopri5_adjoint.hip: cde_dopri5_adjoint_state_sums).
namespace cde {
__global__ void madj_carry_kernel(const unsigned char* __restrict__ ctrl, int p2, const double* __restrict__ reduced,
                                  double* __restrict__ carry) {
  const AdjCtrl k = *reinterpret_cast<const AdjCtrl*>(ctrl + p2 * ADJ_CTRL_STRIDE);
  if (k.c.phase == 4 && k.commit == 0) return;
  if (k.mode == 3) carry[0] = (double)((float)k.T + (float)reduced[4]);
}
}  // namespace cde

extern "C" int cde_dopri5_adjoint_mlp_state_sums(void* workspace, size_t workspace_bytes, int64_t B, int64_t C, int64_t H,
                                                 int64_t total_launches, double* sums, void* stream) {
  using namespace cde;
  if (B < 1 || C < 1 || H < 1 || total_launches < 1) return CDE_ERR_SHAPE;
  if (!workspace || !sums) return CDE_ERR_NULL;
  if (workspace_bytes < cde_dopri5_adjoint_mlp_workspace_bytes(B, C, H)) return CDE_ERR_WORKSPACE;
  unsigned char* base = (unsigned char*)workspace;
  const MadjLayout L = madj_layout(B, H, C);
  const int parity = (int)((total_launches - 1) & 1);
  const double* partial = (const double*)(base + L.partial) + (int64_t)(parity ^ 1) * L.n_wg * ADJ_NS;
  madj_state_sums_kernel<<<1, 64, 0, (hipStream_t)stream>>>(partial, L.n_wg, sums);
  return check_launch();
}

extern "C" int cde_dopri5_adjoint_mlp_apply_state_sums(void* workspace, size_t workspace_bytes, int64_t B, int64_t C,
                                                       int64_t H, int64_t total_launches, const double* reduced,
                                                       void* stream) {
  using namespace cde;
  if (B < 1 || C < 1 || H < 1 || total_launches < 1) return CDE_ERR_SHAPE;
  if (!workspace || !reduced) return CDE_ERR_NULL;
  if (workspace_bytes < cde_dopri5_adjoint_mlp_workspace_bytes(B, C, H)) return CDE_ERR_WORKSPACE;
  unsigned char* base = (unsigned char*)workspace;
  const MadjLayout L = madj_layout(B, H, C);
  const int parity = (int)((total_launches - 1) & 1);
  madj_carry_kernel<<<1, 1, 0, (hipStream_t)stream>>>(base, parity ^ 1, reduced, (double*)(base + L.carry));
  return check_launch();
}

