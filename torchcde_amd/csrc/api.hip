// api.hip -- C-ABI entry points of the fused solvers: argument checks, stage table, kernel dispatch.
#include "cde_common.h"

namespace cde {

// from rk4_generic.hip
size_t generic_adjoint_workspace_bytes(int64_t B, int64_t C, int64_t H, size_t elem);
bool generic_applicable(int64_t C, int64_t H, size_t elem, bool adjoint);
template <typename T, typename TT>
int launch_forward_generic(const void*, const void*, int64_t, int, const void*, const void*, int, const void*,
                           const void*, int64_t, const void*, int64_t, void*, int64_t, int64_t, int64_t,
                           const int64_t*, const void*, hipStream_t);
template <typename T, typename TT>
int launch_adjoint_generic(const void*, const void*, int64_t, int, const void*, const void*, int, const void*,
                           const void*, const void*, const int64_t*, int64_t, void*, void*, void*, int64_t, int64_t,
                           int64_t, const int64_t*, const void*, void*, hipStream_t);
// from rk4_mfma.hip
bool mfma_applicable(int64_t C, int64_t H, int dtype, int act, bool adjoint);
size_t mfma_adjoint_partial_bytes(int64_t B);
template <typename TT>
int launch_forward_mfma(const void*, const void*, int64_t, int, const void*, const void*, int, const void*,
                        const void*, int64_t, const void*, int64_t, void*, int64_t, int64_t, int64_t, const int64_t*,
                        const void*, hipStream_t);
template <typename TT>
int launch_forward_mlp(const void*, const void*, int64_t, int, const void*, const void*, int64_t, const void*,
                       const void*, int, const void*, const void*, int64_t, const void*, int64_t, void*, int64_t,
                       int64_t, int64_t, const int64_t*, const void*, hipStream_t);
template <typename TT>
int launch_adjoint_mfma(const void*, const void*, int64_t, int, const void*, const void*, int, const void*,
                        const void*, const void*, const int64_t*, int64_t, void*, void*, void*, int64_t, int64_t,
                        int64_t, const int64_t*, const void*, float*, void*, hipStream_t);

// from rk4_split.hip: one workgroup (4 waves) per 16 series -- the latency-oriented variant for small batches
size_t split_adjoint_partial_bytes(int64_t B);
template <typename TT>
int launch_forward_split(const void*, const void*, int64_t, int, const void*, const void*, int, const void*,
                         const void*, int64_t, const void*, int64_t, void*, int64_t, int64_t, int64_t, const int64_t*,
                         const void*, hipStream_t);
template <typename TT>
int launch_adjoint_split(const void*, const void*, int64_t, int, const void*, const void*, int, const void*,
                         const void*, const void*, const int64_t*, int64_t, void*, void*, void*, int64_t, int64_t,
                         int64_t, const int64_t*, const void*, float*, hipStream_t);

bool mlp_shape_ok(int64_t C, int64_t H, int64_t width);        // rk4_mfma.hip
bool mlp_shape_upper(int64_t C, int64_t H, int64_t width);     // rk4_mfma.hip

// from rk4_wide.hip: affine fields with H <= 64, C <= 8 or H <= 32, C <= 16
bool wide_applicable(int64_t C, int64_t H, int dtype, int act);
size_t wide_adjoint_workspace_bytes(int64_t B, int64_t C, int64_t H, int64_t n_steps);
template <typename TT>
int launch_forward_wide(const void*, const void*, int64_t, int, const void*, const void*, int, const void*, const void*,
                        int64_t, const void*, int64_t, void*, int64_t, int64_t, int64_t, const int64_t*, const void*,
                        hipStream_t);
template <typename TT>
int launch_adjoint_wide(const void*, const void*, int64_t, int, const void*, const void*, int, const void*, const void*,
                        const void*, int64_t, const int64_t*, int64_t, void*, void*, void*, int64_t, int64_t, int64_t,
                        const int64_t*, const void*, void*, hipStream_t);

// from rk4_mfma.hip
template <typename TT>
int launch_adjoint_jacobian_bx(const void*, const void*, int64_t, int, const void*, const void*, const void*, const void*,
                               const void*, const int64_t*, int64_t, void*, void*, void*, int64_t, int64_t, int64_t,
                               const int64_t*, const void*, float*, hipStream_t);
// midpoint / euler: K2 and K3p with two stages / one stage per step (rk4_mfma.hip, rk4_adjoint_pair.hip)
template <typename TT>
int launch_forward_mfma_method(int, const void*, const void*, int64_t, int, const void*, const void*, const void*, const void*,
                               int64_t, const void*, int64_t, void*, int64_t, int64_t, int64_t, const int64_t*, const void*,
                               hipStream_t);
template <typename TT>
int launch_adjoint_jacobian_pair(const void*, const void*, int64_t, int, const void*, const void*, const void*, const void*,
                                 const void*, const int64_t*, int64_t, void*, void*, void*, int64_t, int64_t, int64_t,
                                 const int64_t*, const void*, float*, hipStream_t, int method);
// K2 with the stage states stored (rk4_mfma.hip) and the reverse-mode sweep over them (rk4_backprop.hip): adjoint=False
template <typename TT>
int launch_forward_mfma_stages(const void*, const void*, int64_t, int, const void*, const void*, int, const void*, const void*,
                               int64_t, const void*, int64_t, void*, void*, int64_t, int64_t, int64_t, const int64_t*,
                               const void*, hipStream_t);
size_t backprop_workspace_bytes(int64_t B);
int launch_backprop_jacobian(const void*, const void*, int64_t, int, const void*, const void*, int, const void*, const void*,
                             int64_t, const float*, int64_t, const int64_t*, const int64_t*, const float*, void*, void*, void*,
                             int64_t, int64_t, int64_t, const int64_t*, const float*, float*, void* grad_coeffs, hipStream_t);
// from rk4_bf16x3.hip
template <typename TT>
int launch_forward_bf16x3(const void*, const void*, int64_t, int, const void*, const void*, const void*, const void*, int64_t,
                          const void*, int64_t, void*, int64_t, int64_t, int64_t, const int64_t*, const void*, hipStream_t);
template <typename TT>
int launch_adjoint_bf16x3(const void*, const void*, int64_t, int, const void*, const void*, const void*, const void*,
                          const void*, const int64_t*, int64_t, void*, void*, void*, int64_t, int64_t, int64_t,
                          const int64_t*, const void*, float*, hipStream_t);
static bool bf16x3_applicable(int64_t C, int64_t H, int dtype, int act) {
  return dtype == CDE_F32 && act == CDE_ACT_NONE && H >= 1 && H <= 32 && C >= 1 && C <= 8;
}

// from rk4_mlp_adjoint.hip
size_t mlp_adjoint_image_bytes();
int launch_mlp_adjoint_images(const void*, const void*, int64_t, const void*, const void*, int64_t, int64_t, float*,
                              hipStream_t);
template <typename TT>
int launch_mlp_adjoint_sweep(const void*, const void*, int64_t, int, int, const float*, void*, void*, const void*,
                             int64_t, int64_t, const int64_t*, const void*, void*, void*, void*, void*, int64_t,
                             int64_t, int64_t, void*, hipStream_t);

// adjoint=False for the two-layer field: K2m storing its stage states, K3m's sweep as reverse mode through the steps
template <typename TT>
int launch_forward_mlp_stages(const void*, const void*, int64_t, int, const void*, const void*, int64_t, const void*,
                              const void*, int, const void*, const void*, int64_t, const void*, int64_t, void*, void*, int64_t,
                              int64_t, int64_t, const int64_t*, const void*, hipStream_t);
template <typename TT>
int launch_mlp_backprop_sweep(const void*, const void*, int64_t, int, int, const float*, const void*, int64_t, void*,
                              const void*, int64_t, int64_t, const int64_t*, const void*, void*, void*, void*, void*, int64_t,
                              int64_t, int64_t, void* grad_coeffs, hipStream_t);

// Stage table: for solver step k over [grid[k], grid[k+1]] and RK stage j, the control interval
// and fractional part at the stage time -- what CubicSpline._interpret_t (interpolation_cubic.py:
// 315-322) returns when torchdiffeq's rk4 evaluates the vector field there.  One lane per entry.
// `negate`: the reverse sweep integrates in s = -t and evaluates the field at t = -s.
// `method` (CDE_METHOD_*): rk4's 3/8-rule times; midpoint: t0, t0 + 0.5 dt (torchdiffeq's `half_dt = 0.5 * dt`); euler: t0.
// The table keeps four slots per step whatever the method (unused slots repeat t0).
template <typename T, typename TT>
__global__ void stage_table_kernel(const T* __restrict__ knots, int64_t n_intervals, const TT* __restrict__ grid,
                                   int64_t n_steps, int negate, int64_t* __restrict__ index_out,
                                   T* __restrict__ frac_out, int method) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= 4 * n_steps) return;
  const int64_t k = e >> 2;
  const StageClock<TT> clk(grid[k], grid[k + 1]);
  const int stage = (int)(e & 3);
  TT tt = clk.time(stage);
  if (method == CDE_METHOD_MIDPOINT) tt = stage == 1 ? clk.t0 + (TT)0.5 * clk.dt : clk.t0;
  else if (method == CDE_METHOD_EULER) tt = clk.t0;
  T ts = (T)tt;
  if (negate) ts = -ts;
  T frac;
  index_out[e] = locate(knots, n_intervals, ts, frac);
  frac_out[e] = frac;
}

template <typename T, typename TT>
static int fill_stage_table(const void* knots, int64_t n_intervals, const void* grid, int64_t n_steps, int negate,
                            int64_t* index_out, void* frac_out, hipStream_t s, int method = CDE_METHOD_RK4) {
  if (n_steps <= 0) return CDE_OK;
  stage_table_kernel<T, TT><<<(unsigned)((4 * n_steps + 255) / 256), 256, 0, s>>>(
      (const T*)knots, n_intervals, (const TT*)grid, n_steps, negate, index_out, (T*)frac_out, method);
  return check_launch();
}

static inline size_t align256(size_t x) { return (x + 255) / 256 * 256; }

static bool pick_mfma(int variant, int64_t C, int64_t H, int dtype, int act, bool adjoint, int* rc) {
  const bool ok = mfma_applicable(C, H, dtype, act, adjoint);
  *rc = CDE_OK;
  if (variant == CDE_VARIANT_MFMA || variant == CDE_VARIANT_SPLIT) { if (!ok) *rc = CDE_ERR_UNSUPPORTED; return ok; }
  if (variant == CDE_VARIANT_GENERIC) return false;
  if (variant != CDE_VARIANT_AUTO) { *rc = CDE_ERR_UNSUPPORTED; return false; }
  return ok;
}

// Among the MFMA kernels: the one-wave-per-series-tile kernels (K2/K3) need B/16 (B/32) waves to fill 1024 SIMDs twice
// (once); below CDE_SPLIT_MAX_BATCH series the workgroup-per-tile kernels of rk4_split.hip finish sooner.
// The tanh field is VALU-co-limited in the one-wave-per-tile kernels (K3a: 46 % MFMA-busy); the 8-wave tile kernel
// overlaps the activation work of its chain waves with the helper waves' dW products and wins at EVERY batch size
// (32768 series, forward + adjoint: 12.9 ms against 15.0 ms).
static bool pick_split(int variant, int64_t B, bool control_grad, int act = CDE_ACT_NONE) {
  if (control_grad) return false;                  // dL/dcoeffs lives in the pre-activation K3 kernel only
  if (variant == CDE_VARIANT_SPLIT) return true;
  return variant == CDE_VARIANT_AUTO && (B <= CDE_SPLIT_MAX_BATCH || act == CDE_ACT_TANH);
}

// Shapes beyond the 32 x 8 tiles (H <= 64, C <= 8 or H <= 32, C <= 16): the wide tile kernels under AUTO
static bool pick_wide(int variant, int64_t C, int64_t H, int dtype, int act) {
  return variant == CDE_VARIANT_AUTO && !mfma_applicable(C, H, dtype, act, false) && wide_applicable(C, H, dtype, act);
}

template <typename T, typename TT>
static int forward_typed(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                         const void* bias, int act, const void* z0, const void* grid, int64_t n_grid,
                         const void* t_out, int64_t n_out, void* z_out, int64_t B, int64_t C, int64_t H, int dtype,
                         int variant, int64_t* stage_index, void* stage_frac, hipStream_t s) {
  int rc = fill_stage_table<T, TT>(knots, n_intervals, grid, n_grid - 1, 0, stage_index, stage_frac, s);
  if (rc != CDE_OK) return rc;
  if (variant == CDE_VARIANT_BF16X3) {
    if (!bf16x3_applicable(C, H, dtype, act)) return CDE_ERR_UNSUPPORTED;
    return launch_forward_bf16x3<TT>(coeffs, knots, n_intervals, degree, W, bias, z0, grid, n_grid, t_out, n_out, z_out, B, C,
                                     H, stage_index, stage_frac, s);
  }
  const bool use_mfma = pick_mfma(variant, C, H, dtype, act, false, &rc);
  if (rc != CDE_OK) return rc;
  if (use_mfma && pick_split(variant, B, false, act))
    return launch_forward_split<TT>(coeffs, knots, n_intervals, degree, W, bias, act, z0, grid, n_grid, t_out, n_out,
                                    z_out, B, C, H, stage_index, stage_frac, s);
  if (use_mfma)
    return launch_forward_mfma<TT>(coeffs, knots, n_intervals, degree, W, bias, act, z0, grid, n_grid, t_out, n_out, z_out,
                                   B, C, H, stage_index, stage_frac, s);
  if (pick_wide(variant, C, H, dtype, act))
    return launch_forward_wide<TT>(coeffs, knots, n_intervals, degree, W, bias, act, z0, grid, n_grid, t_out, n_out,
                                   z_out, B, C, H, stage_index, stage_frac, s);
  return launch_forward_generic<T, TT>(coeffs, knots, n_intervals, degree, W, bias, act, z0, grid, n_grid, t_out, n_out,
                                       z_out, B, C, H, stage_index, stage_frac, s);
}

template <typename T, typename TT>
static int adjoint_typed(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                         const void* bias, int act, const void* z_saved, const void* grad_out, const void* sgrid,
                         int64_t n_sgrid, const int64_t* seg_off, const int64_t* seg_off_host, int64_t n_out,
                         void* grad_z0, void* grad_W, void* grad_b, int64_t B, int64_t C, int64_t H, int dtype,
                         int variant, void* workspace, size_t workspace_bytes, void* grad_coeffs, hipStream_t s) {
  int rc;
  if (variant == CDE_VARIANT_BF16X3) {
    if (!bf16x3_applicable(C, H, dtype, act) || grad_coeffs) return CDE_ERR_UNSUPPORTED;
    const int64_t n_steps_b = n_sgrid - 1;
    const size_t off_frac_b = align256((size_t)(4 * n_steps_b) * sizeof(int64_t));
    const size_t off_part_b = off_frac_b + align256((size_t)(4 * n_steps_b) * sizeof(T));
    if (workspace_bytes < off_part_b + mfma_adjoint_partial_bytes(B)) return CDE_ERR_WORKSPACE;
    int64_t* sidx = (int64_t*)workspace;
    void* sfrac = (unsigned char*)workspace + off_frac_b;
    rc = fill_stage_table<T, TT>(knots, n_intervals, sgrid, n_steps_b, 1, sidx, sfrac, s);
    if (rc != CDE_OK) return rc;
    // the reverse sweep: K3j with its J rows on the bf16 pipe (rk4_mfma.hip); CDE_OPT_K3_FORM = 1 keeps K3b (three GEMMs,
    // two of them on the bf16 pipe)
    if (cde::option(CDE_OPT_K3_FORM) != 1)
      return launch_adjoint_jacobian_bx<TT>(coeffs, knots, n_intervals, degree, W, bias, z_saved, grad_out, sgrid, seg_off,
                                            n_out, grad_z0, grad_W, grad_b, B, C, H, sidx, sfrac,
                                            (float*)((unsigned char*)workspace + off_part_b), s);
    return launch_adjoint_bf16x3<TT>(coeffs, knots, n_intervals, degree, W, bias, z_saved, grad_out, sgrid, seg_off, n_out,
                                     grad_z0, grad_W, grad_b, B, C, H, sidx, sfrac,
                                     (float*)((unsigned char*)workspace + off_part_b), s);
  }
  const bool use_mfma = pick_mfma(variant, C, H, dtype, act, true, &rc);
  if (rc != CDE_OK) return rc;
  if (grad_coeffs && !use_mfma) return CDE_ERR_UNSUPPORTED;      // control gradients: MFMA kernels only
  // workspace: [stage_index: 4*(n_sgrid-1) int64][stage_frac: 4*(n_sgrid-1) T][partials]
  const int64_t n_steps = n_sgrid - 1;
  const size_t off_frac = align256((size_t)(4 * n_steps) * sizeof(int64_t));
  const size_t off_part = off_frac + align256((size_t)(4 * n_steps) * sizeof(T));
  const bool use_split = use_mfma && pick_split(variant, B, grad_coeffs != nullptr, act);
  if (variant == CDE_VARIANT_SPLIT && !use_split) return CDE_ERR_UNSUPPORTED;
  const bool use_wide = !use_mfma && pick_wide(variant, C, H, dtype, act);
  const size_t part_bytes = use_split ? split_adjoint_partial_bytes(B)
                            : use_mfma ? mfma_adjoint_partial_bytes(B)
                            : use_wide ? wide_adjoint_workspace_bytes(B, C, H, n_steps)
                                       : generic_adjoint_workspace_bytes(B, C, H, sizeof(T));
  if (workspace_bytes < off_part + part_bytes) return CDE_ERR_WORKSPACE;
  int64_t* stage_index = (int64_t*)workspace;
  void* stage_frac = (unsigned char*)workspace + off_frac;
  void* partial = (unsigned char*)workspace + off_part;
  rc = fill_stage_table<T, TT>(knots, n_intervals, sgrid, n_steps, 1, stage_index, stage_frac, s);
  if (rc != CDE_OK) return rc;
  if (use_split)
    return launch_adjoint_split<TT>(coeffs, knots, n_intervals, degree, W, bias, act, z_saved, grad_out, sgrid, seg_off,
                                    n_out, grad_z0, grad_W, grad_b, B, C, H, stage_index, stage_frac, (float*)partial, s);
  if (use_mfma)
    return launch_adjoint_mfma<TT>(coeffs, knots, n_intervals, degree, W, bias, act, z_saved, grad_out, sgrid, seg_off, n_out,
                                   grad_z0, grad_W, grad_b, B, C, H, stage_index, stage_frac, (float*)partial,
                                   grad_coeffs, s);
  if (use_wide)
    return launch_adjoint_wide<TT>(coeffs, knots, n_intervals, degree, W, bias, act, z_saved, grad_out, sgrid, n_sgrid,
                                   seg_off_host, n_out, grad_z0, grad_W, grad_b, B, C, H, stage_index, stage_frac, partial, s);
  return launch_adjoint_generic<T, TT>(coeffs, knots, n_intervals, degree, W, bias, act, z_saved, grad_out, sgrid,
                                       seg_off, n_out, grad_z0, grad_W, grad_b, B, C, H, stage_index, stage_frac,
                                       partial, s);
}

}  // namespace cde

extern "C" int cde_rk4_supported(int64_t C, int64_t H, int dtype, int act, int adjoint, int variant) {
  if (C < 1 || H < 1 || (dtype != CDE_F32 && dtype != CDE_F64)) return 0;
  if (act != CDE_ACT_NONE && act != CDE_ACT_TANH) return 0;
  if (variant == CDE_VARIANT_BF16X3) return cde::bf16x3_applicable(C, H, dtype, act) ? 1 : 0;
  int rc;
  const bool mfma = cde::pick_mfma(variant, C, H, dtype, act, adjoint != 0, &rc);
  if (rc != CDE_OK) return 0;
  if (mfma) return 1;
  if (cde::pick_wide(variant, C, H, dtype, act)) return 1;
  return cde::generic_applicable(C, H, dtype == CDE_F64 ? 8 : 4, adjoint != 0) ? 1 : 0;
}

extern "C" int cde_rk4_forward_linear(const void* coeffs, const void* knots, int64_t n_intervals, int degree,
                                      const void* W, const void* bias, int act, const void* z0, const void* grid,
                                      int64_t n_grid, const void* t_out, int64_t n_out, void* z_out, int64_t B,
                                      int64_t C, int64_t H, int dtype, int time_dtype, int variant,
                                      int64_t* stage_index, void* stage_frac, void* stream) {
  if (B < 0 || C < 1 || H < 1 || n_intervals < 1 || n_grid < 1 || n_out < 1) return CDE_ERR_SHAPE;
  if (act != CDE_ACT_NONE && act != CDE_ACT_TANH) return CDE_ERR_UNSUPPORTED;
  if (B == 0) return CDE_OK;
  if (!coeffs || !knots || !W || !bias || !z0 || !grid || !t_out || !z_out) return CDE_ERR_NULL;
  if (n_grid > 1 && (!stage_index || !stage_frac)) return CDE_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
#define CDE_CALL(T, TT)                                                                                              \
  return cde::forward_typed<T, TT>(coeffs, knots, n_intervals, degree, W, bias, act, z0, grid, n_grid, t_out, n_out, \
                                   z_out, B, C, H, dtype, variant, stage_index, stage_frac, s)
  if (dtype == CDE_F32 && time_dtype == CDE_F32) CDE_CALL(float, float);
  if (dtype == CDE_F32 && time_dtype == CDE_F64) CDE_CALL(float, double);
  if (dtype == CDE_F64 && time_dtype == CDE_F64) CDE_CALL(double, double);
  if (dtype == CDE_F64 && time_dtype == CDE_F32) CDE_CALL(double, float);
#undef CDE_CALL
  return CDE_ERR_DTYPE;
}

extern "C" int cde_rk4_forward_mlp(const void* coeffs, const void* knots, int64_t n_intervals, int degree,
                                   const void* W1, const void* bias1, int64_t width, const void* W2, const void* bias2,
                                   int act, const void* z0, const void* grid, int64_t n_grid, const void* t_out,
                                   int64_t n_out, void* z_out, int64_t B, int64_t C, int64_t H, int dtype,
                                   int time_dtype, int64_t* stage_index, void* stage_frac, void* stream) {
  if (B < 0 || C < 1 || H < 1 || width < 1 || n_intervals < 1 || n_grid < 1 || n_out < 1) return CDE_ERR_SHAPE;
  if (dtype != CDE_F32) return dtype == CDE_F64 ? CDE_ERR_UNSUPPORTED : CDE_ERR_DTYPE;
  if (B == 0) return CDE_OK;
  if (!coeffs || !knots || !W1 || !bias1 || !W2 || !bias2 || !z0 || !grid || !t_out || !z_out) return CDE_ERR_NULL;
  if (n_grid > 1 && (!stage_index || !stage_frac)) return CDE_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (time_dtype == CDE_F32) {
    rc = cde::fill_stage_table<float, float>(knots, n_intervals, grid, n_grid - 1, 0, stage_index, stage_frac, s);
    if (rc != CDE_OK) return rc;
    return cde::launch_forward_mlp<float>(coeffs, knots, n_intervals, degree, W1, bias1, width, W2, bias2, act, z0, grid,
                                          n_grid, t_out, n_out, z_out, B, C, H, stage_index, stage_frac, s);
  }
  if (time_dtype == CDE_F64) {
    rc = cde::fill_stage_table<float, double>(knots, n_intervals, grid, n_grid - 1, 0, stage_index, stage_frac, s);
    if (rc != CDE_OK) return rc;
    return cde::launch_forward_mlp<double>(coeffs, knots, n_intervals, degree, W1, bias1, width, W2, bias2, act, z0, grid,
                                           n_grid, t_out, n_out, z_out, B, C, H, stage_index, stage_frac, s);
  }
  return CDE_ERR_DTYPE;
}

extern "C" size_t cde_rk4_adjoint_workspace_bytes(int64_t B, int64_t C, int64_t H, int64_t n_sgrid, int dtype,
                                                  int variant) {
  const size_t elem = dtype == CDE_F64 ? 8 : 4;
  const int64_t n_steps = n_sgrid > 1 ? n_sgrid - 1 : 0;
  size_t bytes = cde::align256((size_t)(4 * n_steps) * sizeof(int64_t)) + cde::align256((size_t)(4 * n_steps) * elem);
  if (variant == CDE_VARIANT_BF16X3) return bytes + cde::mfma_adjoint_partial_bytes(B);
  int rc;
  const bool use_mfma = cde::pick_mfma(variant, C, H, dtype, CDE_ACT_NONE, true, &rc);
  // AUTO may resolve to either kernel depending on the activation: reserve the larger need
  size_t a = cde::mfma_adjoint_partial_bytes(B);
  if (variant != CDE_VARIANT_MFMA && variant != CDE_VARIANT_GENERIC) {    // AUTO may resolve to the tile kernels (tanh: at any B)
    const size_t sp = cde::split_adjoint_partial_bytes(B);
    a = sp > a ? sp : a;
  }
  size_t b = cde::generic_adjoint_workspace_bytes(B, C, H, elem);
  if (!use_mfma && (cde::pick_wide(variant, C, H, dtype, CDE_ACT_NONE))) {
    const size_t wb = cde::wide_adjoint_workspace_bytes(B, C, H, n_steps);
    b = wb > b ? wb : b;
  }
  if (variant == CDE_VARIANT_MFMA || variant == CDE_VARIANT_SPLIT) bytes += a;
  else if (variant == CDE_VARIANT_GENERIC || !use_mfma) bytes += b;
  else bytes += (a > b ? a : b);
  return bytes;
}

static int adjoint_linear_impl(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                               const void* bias, int act, const void* z_saved, const void* grad_out, const void* sgrid,
                               int64_t n_sgrid, const int64_t* seg_off, const int64_t* seg_off_host, int64_t n_out,
                               void* grad_z0, void* grad_W, void* grad_b, void* grad_coeffs, int64_t B, int64_t C,
                               int64_t H, int dtype, int time_dtype, int variant, void* workspace,
                               size_t workspace_bytes, void* stream) {
  if (B < 1 || C < 1 || H < 1 || n_intervals < 1 || n_out < 1 || n_sgrid < 0) return CDE_ERR_SHAPE;
  if (act != CDE_ACT_NONE && act != CDE_ACT_TANH) return CDE_ERR_UNSUPPORTED;
  if (!coeffs || !knots || !W || !bias || !z_saved || !grad_out || !grad_z0 || !grad_W || !grad_b || !workspace)
    return CDE_ERR_NULL;
  if (n_out > 1 && (!sgrid || !seg_off)) return CDE_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
#define CDE_CALL(T, TT)                                                                                               \
  return cde::adjoint_typed<T, TT>(coeffs, knots, n_intervals, degree, W, bias, act, z_saved, grad_out, sgrid,        \
                                   n_sgrid, seg_off, seg_off_host, n_out, grad_z0, grad_W, grad_b, B, C, H, dtype,    \
                                   variant, workspace, workspace_bytes, grad_coeffs, s)
  if (dtype == CDE_F32 && time_dtype == CDE_F32) CDE_CALL(float, float);
  if (dtype == CDE_F32 && time_dtype == CDE_F64) CDE_CALL(float, double);
  if (dtype == CDE_F64 && time_dtype == CDE_F64) CDE_CALL(double, double);
  if (dtype == CDE_F64 && time_dtype == CDE_F32) CDE_CALL(double, float);
#undef CDE_CALL
  return CDE_ERR_DTYPE;
}

extern "C" int cde_rk4_adjoint_linear(const void* coeffs, const void* knots, int64_t n_intervals, int degree,
                                      const void* W, const void* bias, int act, const void* z_saved,
                                      const void* grad_out, const void* sgrid, int64_t n_sgrid, const int64_t* seg_off,
                                      const int64_t* seg_off_host, int64_t n_out, void* grad_z0, void* grad_W,
                                      void* grad_b, int64_t B, int64_t C, int64_t H, int dtype, int time_dtype,
                                      int variant, void* workspace, size_t workspace_bytes, void* stream) {
  return adjoint_linear_impl(coeffs, knots, n_intervals, degree, W, bias, act, z_saved, grad_out, sgrid, n_sgrid, seg_off,
                             seg_off_host, n_out, grad_z0, grad_W, grad_b, nullptr, B, C, H, dtype, time_dtype, variant, workspace,
                             workspace_bytes, stream);
}

extern "C" int cde_rk4_adjoint_linear_dcontrol(const void* coeffs, const void* knots, int64_t n_intervals, int degree,
                                               const void* W, const void* bias, int act, const void* z_saved,
                                               const void* grad_out, const void* sgrid, int64_t n_sgrid,
                                               const int64_t* seg_off, int64_t n_out, void* grad_z0, void* grad_W,
                                               void* grad_b, void* grad_coeffs, int64_t B, int64_t C, int64_t H,
                                               int dtype, int time_dtype, void* workspace, size_t workspace_bytes,
                                               void* stream) {
  if (!grad_coeffs) return CDE_ERR_NULL;
  return adjoint_linear_impl(coeffs, knots, n_intervals, degree, W, bias, act, z_saved, grad_out, sgrid, n_sgrid, seg_off,
                             nullptr, n_out, grad_z0, grad_W, grad_b, grad_coeffs, B, C, H, dtype, time_dtype, CDE_VARIANT_MFMA,
                             workspace, workspace_bytes, stream);
}

// ---------------------------------------------------------------------------------------------- midpoint / euler
extern "C" int cde_fixed_supported(int method, int64_t C, int64_t H, int dtype, int act) {
  if (method != CDE_METHOD_MIDPOINT && method != CDE_METHOD_EULER) return 0;
  return (dtype == CDE_F32 && act == CDE_ACT_NONE && cde::mfma_applicable(C, H, dtype, act, true)) ? 1 : 0;
}

extern "C" int cde_fixed_forward_linear(int method, const void* coeffs, const void* knots, int64_t n_intervals, int degree,
                                        const void* W, const void* bias, const void* z0, const void* grid, int64_t n_grid,
                                        const void* t_out, int64_t n_out, void* z_out, int64_t B, int64_t C, int64_t H,
                                        int dtype, int time_dtype, int64_t* stage_index, void* stage_frac, void* stream) {
  if (B < 0 || C < 1 || H < 1 || n_intervals < 1 || n_grid < 1 || n_out < 1) return CDE_ERR_SHAPE;
  if (dtype != CDE_F32) return dtype == CDE_F64 ? CDE_ERR_UNSUPPORTED : CDE_ERR_DTYPE;
  if (!cde_fixed_supported(method, C, H, dtype, CDE_ACT_NONE)) return CDE_ERR_UNSUPPORTED;
  if (B == 0) return CDE_OK;
  if (!coeffs || !knots || !W || !bias || !z0 || !grid || !t_out || !z_out) return CDE_ERR_NULL;
  if (n_grid > 1 && (!stage_index || !stage_frac)) return CDE_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (time_dtype == CDE_F32) {
    rc = cde::fill_stage_table<float, float>(knots, n_intervals, grid, n_grid - 1, 0, stage_index, stage_frac, s, method);
    if (rc != CDE_OK) return rc;
    return cde::launch_forward_mfma_method<float>(method, coeffs, knots, n_intervals, degree, W, bias, z0, grid, n_grid, t_out,
                                                  n_out, z_out, B, C, H, stage_index, stage_frac, s);
  }
  if (time_dtype == CDE_F64) {
    rc = cde::fill_stage_table<float, double>(knots, n_intervals, grid, n_grid - 1, 0, stage_index, stage_frac, s, method);
    if (rc != CDE_OK) return rc;
    return cde::launch_forward_mfma_method<double>(method, coeffs, knots, n_intervals, degree, W, bias, z0, grid, n_grid,
                                                   t_out, n_out, z_out, B, C, H, stage_index, stage_frac, s);
  }
  return CDE_ERR_DTYPE;
}

// workspace: [stage_index: 4*(n_sgrid-1) int64][stage_frac: 4*(n_sgrid-1) f32][per-wave partial parameter gradients]
extern "C" size_t cde_fixed_adjoint_workspace_bytes(int64_t B, int64_t n_sgrid) {
  const int64_t n_steps = n_sgrid > 1 ? n_sgrid - 1 : 0;
  return cde::align256((size_t)(4 * n_steps) * sizeof(int64_t)) + cde::align256((size_t)(4 * n_steps) * sizeof(float)) +
         (B > 0 ? cde::mfma_adjoint_partial_bytes(B) : 0);
}

extern "C" int cde_fixed_adjoint_linear(int method, const void* coeffs, const void* knots, int64_t n_intervals, int degree,
                                        const void* W, const void* bias, const void* z_saved, const void* grad_out,
                                        const void* sgrid, int64_t n_sgrid, const int64_t* seg_off, int64_t n_out,
                                        void* grad_z0, void* grad_W, void* grad_b, int64_t B, int64_t C, int64_t H, int dtype,
                                        int time_dtype, void* workspace, size_t workspace_bytes, void* stream) {
  if (B < 1 || C < 1 || H < 1 || n_intervals < 1 || n_out < 1 || n_sgrid < 0) return CDE_ERR_SHAPE;
  if (dtype != CDE_F32) return dtype == CDE_F64 ? CDE_ERR_UNSUPPORTED : CDE_ERR_DTYPE;
  if (!cde_fixed_supported(method, C, H, dtype, CDE_ACT_NONE)) return CDE_ERR_UNSUPPORTED;
  if (!coeffs || !knots || !W || !bias || !z_saved || !grad_out || !grad_z0 || !grad_W || !grad_b || !workspace)
    return CDE_ERR_NULL;
  if (n_out > 1 && (!sgrid || !seg_off)) return CDE_ERR_NULL;
  if (workspace_bytes < cde_fixed_adjoint_workspace_bytes(B, n_sgrid)) return CDE_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n_steps = n_sgrid > 1 ? n_sgrid - 1 : 0;
  unsigned char* base = (unsigned char*)workspace;
  int64_t* stage_index = (int64_t*)base;
  void* stage_frac = base + cde::align256((size_t)(4 * n_steps) * sizeof(int64_t));
  float* partial = (float*)((unsigned char*)stage_frac + cde::align256((size_t)(4 * n_steps) * sizeof(float)));
  int rc;
  if (time_dtype == CDE_F32) {
    rc = cde::fill_stage_table<float, float>(knots, n_intervals, sgrid, n_steps, 1, stage_index, stage_frac, s, method);
    if (rc != CDE_OK) return rc;
    return cde::launch_adjoint_jacobian_pair<float>(coeffs, knots, n_intervals, degree, W, bias, z_saved, grad_out, sgrid,
                                                    seg_off, n_out, grad_z0, grad_W, grad_b, B, C, H, stage_index, stage_frac,
                                                    partial, s, method);
  }
  if (time_dtype == CDE_F64) {
    rc = cde::fill_stage_table<float, double>(knots, n_intervals, sgrid, n_steps, 1, stage_index, stage_frac, s, method);
    if (rc != CDE_OK) return rc;
    return cde::launch_adjoint_jacobian_pair<double>(coeffs, knots, n_intervals, degree, W, bias, z_saved, grad_out, sgrid,
                                                     seg_off, n_out, grad_z0, grad_W, grad_b, B, C, H, stage_index, stage_frac,
                                                     partial, s, method);
  }
  return CDE_ERR_DTYPE;
}

// ---------------------------------------------------------------------------------------------- K3d (adjoint=False)
extern "C" int cde_rk4_backprop_supported(int64_t C, int64_t H, int dtype, int act) {
  return (dtype == CDE_F32 && (act == CDE_ACT_NONE || act == CDE_ACT_TANH) && cde::mfma_applicable(C, H, dtype, act, true)) ? 1 : 0;
}

extern "C" int cde_rk4_forward_linear_stages(const void* coeffs, const void* knots, int64_t n_intervals, int degree,
                                             const void* W, const void* bias, int act, const void* z0, const void* grid,
                                             int64_t n_grid, const void* t_out, int64_t n_out, void* z_out, void* stages,
                                             int64_t B, int64_t C, int64_t H, int dtype, int time_dtype,
                                             int64_t* stage_index, void* stage_frac, void* stream) {
  if (B < 0 || C < 1 || H < 1 || n_intervals < 1 || n_grid < 1 || n_out < 1) return CDE_ERR_SHAPE;
  if (dtype != CDE_F32) return dtype == CDE_F64 ? CDE_ERR_UNSUPPORTED : CDE_ERR_DTYPE;
  if (!cde_rk4_backprop_supported(C, H, dtype, act)) return CDE_ERR_UNSUPPORTED;
  if (B == 0) return CDE_OK;
  if (!coeffs || !knots || !W || !bias || !z0 || !grid || !t_out || !z_out) return CDE_ERR_NULL;
  if (n_grid > 1 && (!stage_index || !stage_frac || !stages)) return CDE_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (time_dtype == CDE_F32) {
    rc = cde::fill_stage_table<float, float>(knots, n_intervals, grid, n_grid - 1, 0, stage_index, stage_frac, s);
    if (rc != CDE_OK) return rc;
    return cde::launch_forward_mfma_stages<float>(coeffs, knots, n_intervals, degree, W, bias, act, z0, grid, n_grid, t_out,
                                                  n_out, z_out, stages, B, C, H, stage_index, stage_frac, s);
  }
  if (time_dtype == CDE_F64) {
    rc = cde::fill_stage_table<float, double>(knots, n_intervals, grid, n_grid - 1, 0, stage_index, stage_frac, s);
    if (rc != CDE_OK) return rc;
    return cde::launch_forward_mfma_stages<double>(coeffs, knots, n_intervals, degree, W, bias, act, z0, grid, n_grid, t_out,
                                                   n_out, z_out, stages, B, C, H, stage_index, stage_frac, s);
  }
  return CDE_ERR_DTYPE;
}

extern "C" size_t cde_rk4_backprop_workspace_bytes(int64_t B) { return B > 0 ? cde::backprop_workspace_bytes(B) : 0; }

static int rk4_backprop_linear_impl(const void* coeffs, const void* knots, int64_t n_intervals, int degree,
                                       const void* W, const void* bias, int act, const void* stages, const void* grad_out,
                                       int64_t n_out,
                                       const float* step_dt, int64_t n_steps, const int64_t* node_ptr,
                                       const int64_t* node_out, const float* node_weight, void* grad_z0, void* grad_W,
                                       void* grad_b, void* grad_coeffs, int64_t B, int64_t C, int64_t H, int dtype,
                                       const int64_t* stage_index, const void* stage_frac, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  if (B < 1 || C < 1 || H < 1 || n_intervals < 1 || n_out < 1 || n_steps < 0) return CDE_ERR_SHAPE;
  if (dtype != CDE_F32) return dtype == CDE_F64 ? CDE_ERR_UNSUPPORTED : CDE_ERR_DTYPE;
  if (!cde_rk4_backprop_supported(C, H, dtype, act)) return CDE_ERR_UNSUPPORTED;
  if (!coeffs || !knots || !W || !bias || !grad_out || !node_ptr || !node_out || !node_weight || !grad_z0 || !grad_W || !grad_b ||
      !workspace)
    return CDE_ERR_NULL;
  if (n_steps > 0 && (!stages || !step_dt || !stage_index || !stage_frac)) return CDE_ERR_NULL;
  if (workspace_bytes < cde_rk4_backprop_workspace_bytes(B)) return CDE_ERR_WORKSPACE;
  return cde::launch_backprop_jacobian(coeffs, knots, n_intervals, degree, W, bias, act, stages, grad_out, n_out, step_dt, n_steps,
                                       node_ptr, node_out, node_weight, grad_z0, grad_W, grad_b, B, C, H, stage_index,
                                       (const float*)stage_frac, (float*)workspace, grad_coeffs, (hipStream_t)stream);
}

extern "C" int cde_rk4_backprop_linear(const void* coeffs, const void* knots, int64_t n_intervals, int degree,
                                       const void* W, const void* bias, int act, const void* stages, const void* grad_out,
                                       int64_t n_out,
                                       const float* step_dt, int64_t n_steps, const int64_t* node_ptr,
                                       const int64_t* node_out, const float* node_weight, void* grad_z0, void* grad_W,
                                       void* grad_b, int64_t B, int64_t C, int64_t H, int dtype,
                                       const int64_t* stage_index, const void* stage_frac, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  return rk4_backprop_linear_impl(coeffs, knots, n_intervals, degree, W, bias, act, stages, grad_out, n_out, step_dt, n_steps,
                                  node_ptr, node_out, node_weight, grad_z0, grad_W, grad_b, nullptr, B, C, H, dtype,
                                  stage_index, stage_frac, workspace, workspace_bytes, stream);
}

// ... and with the gradient w.r.t. the control's coefficient tensor (`grad_coeffs`: layout of `coeffs`, ZEROED by the caller,
// accumulated): under adjoint=False autograd reaches the control through X.derivative at every stage.
extern "C" int cde_rk4_backprop_linear_dcontrol(const void* coeffs, const void* knots, int64_t n_intervals, int degree,
                                                const void* W, const void* bias, int act, const void* stages,
                                                const void* grad_out, int64_t n_out, const float* step_dt, int64_t n_steps,
                                                const int64_t* node_ptr, const int64_t* node_out, const float* node_weight,
                                                void* grad_z0, void* grad_W, void* grad_b, void* grad_coeffs, int64_t B,
                                                int64_t C, int64_t H, int dtype, const int64_t* stage_index,
                                                const void* stage_frac, void* workspace, size_t workspace_bytes,
                                                void* stream) {
  if (!grad_coeffs) return CDE_ERR_NULL;
  return rk4_backprop_linear_impl(coeffs, knots, n_intervals, degree, W, bias, act, stages, grad_out, n_out, step_dt, n_steps,
                                  node_ptr, node_out, node_weight, grad_z0, grad_W, grad_b, grad_coeffs, B, C, H, dtype,
                                  stage_index, stage_frac, workspace, workspace_bytes, stream);
}

// ---------------------------------------------------------------------------------------------- K3m
// workspace: [stage_index: 4*(n_sgrid-1) int64][stage_frac: 4*(n_sgrid-1) f32][weight images]
static inline size_t mlp_ws_frac_offset(int64_t n_steps) { return cde::align256((size_t)(4 * n_steps) * sizeof(int64_t)); }
static inline size_t mlp_ws_image_offset(int64_t n_steps) {
  return mlp_ws_frac_offset(n_steps) + cde::align256((size_t)(4 * n_steps) * sizeof(float));
}

extern "C" size_t cde_rk4_adjoint_mlp_workspace_bytes(int64_t n_sgrid) {
  const int64_t n_steps = n_sgrid > 1 ? n_sgrid - 1 : 0;
  return mlp_ws_image_offset(n_steps) + cde::align256(cde::mlp_adjoint_image_bytes());
}

extern "C" int cde_rk4_adjoint_mlp_prepare(const void* knots, int64_t n_intervals, const void* sgrid, int64_t n_sgrid,
                                           const void* W1, const void* bias1, int64_t width, const void* W2,
                                           const void* bias2, int64_t C, int64_t H, int dtype, int time_dtype,
                                           void* workspace, size_t workspace_bytes, void* stream) {
  if (C < 1 || H < 1 || width < 1 || n_intervals < 1 || n_sgrid < 0) return CDE_ERR_SHAPE;
  if (dtype != CDE_F32) return dtype == CDE_F64 ? CDE_ERR_UNSUPPORTED : CDE_ERR_DTYPE;
  if (!cde::mlp_shape_ok(C, H, width) && !cde::mlp_shape_upper(C, H, width)) return CDE_ERR_UNSUPPORTED;
  if (!knots || !W1 || !bias1 || !W2 || !bias2 || !workspace || (n_sgrid > 1 && !sgrid)) return CDE_ERR_NULL;
  if (workspace_bytes < cde_rk4_adjoint_mlp_workspace_bytes(n_sgrid)) return CDE_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n_steps = n_sgrid > 1 ? n_sgrid - 1 : 0;
  unsigned char* base = (unsigned char*)workspace;
  int rc;
  if (time_dtype == CDE_F32)
    rc = cde::fill_stage_table<float, float>(knots, n_intervals, sgrid, n_steps, 1, (int64_t*)base,
                                             base + mlp_ws_frac_offset(n_steps), s);
  else if (time_dtype == CDE_F64)
    rc = cde::fill_stage_table<float, double>(knots, n_intervals, sgrid, n_steps, 1, (int64_t*)base,
                                              base + mlp_ws_frac_offset(n_steps), s);
  else return CDE_ERR_DTYPE;
  if (rc != CDE_OK) return rc;
  return cde::launch_mlp_adjoint_images(W1, bias1, width, W2, bias2, C, H, (float*)(base + mlp_ws_image_offset(n_steps)), s);
}

// ---------------------------------------------------------------------------------------------- K3m, reverse mode (adjoint=False)
extern "C" int cde_rk4_forward_mlp_stages(const void* coeffs, const void* knots, int64_t n_intervals, int degree,
                                          const void* W1, const void* bias1, int64_t width, const void* W2,
                                          const void* bias2, int act, const void* z0, const void* grid, int64_t n_grid,
                                          const void* t_out, int64_t n_out, void* z_out, void* stages, int64_t B, int64_t C,
                                          int64_t H, int dtype, int time_dtype, int64_t* stage_index, void* stage_frac,
                                          void* stream) {
  if (B < 0 || C < 1 || H < 1 || width < 1 || n_intervals < 1 || n_grid < 1 || n_out < 1) return CDE_ERR_SHAPE;
  if (dtype != CDE_F32) return dtype == CDE_F64 ? CDE_ERR_UNSUPPORTED : CDE_ERR_DTYPE;
  if (B == 0) return CDE_OK;
  if (!coeffs || !knots || !W1 || !bias1 || !W2 || !bias2 || !z0 || !grid || !t_out || !z_out) return CDE_ERR_NULL;
  if (n_grid > 1 && (!stage_index || !stage_frac || !stages)) return CDE_ERR_NULL;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (time_dtype == CDE_F32) {
    rc = cde::fill_stage_table<float, float>(knots, n_intervals, grid, n_grid - 1, 0, stage_index, stage_frac, s);
    if (rc != CDE_OK) return rc;
    return cde::launch_forward_mlp_stages<float>(coeffs, knots, n_intervals, degree, W1, bias1, width, W2, bias2, act, z0,
                                                 grid, n_grid, t_out, n_out, z_out, stages, B, C, H, stage_index, stage_frac, s);
  }
  if (time_dtype == CDE_F64) {
    rc = cde::fill_stage_table<float, double>(knots, n_intervals, grid, n_grid - 1, 0, stage_index, stage_frac, s);
    if (rc != CDE_OK) return rc;
    return cde::launch_forward_mlp_stages<double>(coeffs, knots, n_intervals, degree, W1, bias1, width, W2, bias2, act, z0,
                                                  grid, n_grid, t_out, n_out, z_out, stages, B, C, H, stage_index, stage_frac, s);
  }
  return CDE_ERR_DTYPE;
}

extern "C" int cde_rk4_backprop_mlp_prepare(const void* knots, int64_t n_intervals, const void* grid, int64_t n_grid,
                                            const void* W1, const void* bias1, int64_t width, const void* W2,
                                            const void* bias2, int64_t C, int64_t H, int dtype, int time_dtype,
                                            void* workspace, size_t workspace_bytes, void* stream) {
  if (C < 1 || H < 1 || width < 1 || n_intervals < 1 || n_grid < 0) return CDE_ERR_SHAPE;
  if (dtype != CDE_F32) return dtype == CDE_F64 ? CDE_ERR_UNSUPPORTED : CDE_ERR_DTYPE;
  if (!cde::mlp_shape_ok(C, H, width) && !cde::mlp_shape_upper(C, H, width)) return CDE_ERR_UNSUPPORTED;
  if (!knots || !W1 || !bias1 || !W2 || !bias2 || !workspace || (n_grid > 1 && !grid)) return CDE_ERR_NULL;
  if (workspace_bytes < cde_rk4_adjoint_mlp_workspace_bytes(n_grid)) return CDE_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n_steps = n_grid > 1 ? n_grid - 1 : 0;
  unsigned char* base = (unsigned char*)workspace;
  int rc;
  if (time_dtype == CDE_F32)
    rc = cde::fill_stage_table<float, float>(knots, n_intervals, grid, n_steps, 0, (int64_t*)base,
                                             base + mlp_ws_frac_offset(n_steps), s);
  else if (time_dtype == CDE_F64)
    rc = cde::fill_stage_table<float, double>(knots, n_intervals, grid, n_steps, 0, (int64_t*)base,
                                              base + mlp_ws_frac_offset(n_steps), s);
  else return CDE_ERR_DTYPE;
  if (rc != CDE_OK) return rc;
  return cde::launch_mlp_adjoint_images(W1, bias1, width, W2, bias2, C, H, (float*)(base + mlp_ws_image_offset(n_steps)), s);
}

static int rk4_backprop_mlp_sweep_impl(const void* coeffs, const void* knots, int64_t n_intervals, int degree, int act,
                                          const void* stages, void* g_state, const void* grid, int64_t n_grid,
                                          int64_t k_begin, int64_t k_end, void* U, void* G2, void* G1, void* Z,
                                          void* grad_coeffs, int64_t B,
                                          int64_t C, int64_t H, int dtype, int time_dtype, const void* workspace,
                                          size_t workspace_bytes, void* stream) {
  if (B < 1 || C < 1 || H < 1 || n_intervals < 1 || k_begin < 0 || k_end < k_begin || k_end > n_grid - 1) return CDE_ERR_SHAPE;
  if (dtype != CDE_F32) return dtype == CDE_F64 ? CDE_ERR_UNSUPPORTED : CDE_ERR_DTYPE;
  if (!cde::mlp_shape_ok(C, H, 1) && !cde::mlp_shape_upper(C, H, 1)) return CDE_ERR_UNSUPPORTED;
  if (!coeffs || !knots || !stages || !g_state || !grid || !U || !G2 || !G1 || !Z || !workspace) return CDE_ERR_NULL;
  if (workspace_bytes < cde_rk4_adjoint_mlp_workspace_bytes(n_grid)) return CDE_ERR_WORKSPACE;
  const int64_t n_steps = n_grid - 1;
  const unsigned char* base = (const unsigned char*)workspace;
  const int64_t* stage_index = (const int64_t*)base;
  const void* stage_frac = base + mlp_ws_frac_offset(n_steps);
  const float* img = (const float*)(base + mlp_ws_image_offset(n_steps));
  hipStream_t s = (hipStream_t)stream;
  if (time_dtype == CDE_F32)
    return cde::launch_mlp_backprop_sweep<float>(coeffs, knots, n_intervals, degree, act, img, stages, n_steps, g_state, grid,
                                                 k_begin, k_end, stage_index, stage_frac, U, G2, G1, Z, B, C, H, grad_coeffs, s);
  if (time_dtype == CDE_F64)
    return cde::launch_mlp_backprop_sweep<double>(coeffs, knots, n_intervals, degree, act, img, stages, n_steps, g_state, grid,
                                                  k_begin, k_end, stage_index, stage_frac, U, G2, G1, Z, B, C, H, grad_coeffs, s);
  return CDE_ERR_DTYPE;
}

extern "C" int cde_rk4_backprop_mlp_sweep(const void* coeffs, const void* knots, int64_t n_intervals, int degree, int act,
                                          const void* stages, void* g_state, const void* grid, int64_t n_grid,
                                          int64_t k_begin, int64_t k_end, void* U, void* G2, void* G1, void* Z, int64_t B,
                                          int64_t C, int64_t H, int dtype, int time_dtype, const void* workspace,
                                          size_t workspace_bytes, void* stream) {
  return rk4_backprop_mlp_sweep_impl(coeffs, knots, n_intervals, degree, act, stages, g_state, grid, n_grid, k_begin, k_end, U,
                                     G2, G1, Z, nullptr, B, C, H, dtype, time_dtype, workspace, workspace_bytes, stream);
}

// ... and with the gradient w.r.t. the control's coefficient tensor (`grad_coeffs` zeroed by the caller before the first chunk,
// accumulated by every chunk's launch)
extern "C" int cde_rk4_backprop_mlp_sweep_dcontrol(const void* coeffs, const void* knots, int64_t n_intervals, int degree,
                                                   int act, const void* stages, void* g_state, const void* grid,
                                                   int64_t n_grid, int64_t k_begin, int64_t k_end, void* U, void* G2, void* G1,
                                                   void* Z, void* grad_coeffs, int64_t B, int64_t C, int64_t H, int dtype,
                                                   int time_dtype, const void* workspace, size_t workspace_bytes,
                                                   void* stream) {
  if (!grad_coeffs) return CDE_ERR_NULL;
  return rk4_backprop_mlp_sweep_impl(coeffs, knots, n_intervals, degree, act, stages, g_state, grid, n_grid, k_begin, k_end, U,
                                     G2, G1, Z, grad_coeffs, B, C, H, dtype, time_dtype, workspace, workspace_bytes, stream);
}

extern "C" int cde_rk4_adjoint_mlp_sweep(const void* coeffs, const void* knots, int64_t n_intervals, int degree, int act,
                                         void* y_state, void* a_state, const void* sgrid, int64_t n_sgrid,
                                         int64_t k_begin, int64_t k_end, void* U, void* G2, void* G1, void* Z,
                                         void* grad_coeffs, int64_t B, int64_t C, int64_t H, int dtype, int time_dtype,
                                         const void* workspace, size_t workspace_bytes, void* stream) {
  if (B < 1 || C < 1 || H < 1 || n_intervals < 1 || k_begin < 0 || k_end < k_begin || k_end > n_sgrid - 1) return CDE_ERR_SHAPE;
  if (dtype != CDE_F32) return dtype == CDE_F64 ? CDE_ERR_UNSUPPORTED : CDE_ERR_DTYPE;
  if (!cde::mlp_shape_ok(C, H, 1) && !cde::mlp_shape_upper(C, H, 1)) return CDE_ERR_UNSUPPORTED;
  if (!coeffs || !knots || !y_state || !a_state || !sgrid || !U || !G2 || !G1 || !Z || !workspace) return CDE_ERR_NULL;
  if (workspace_bytes < cde_rk4_adjoint_mlp_workspace_bytes(n_sgrid)) return CDE_ERR_WORKSPACE;
  const int64_t n_steps = n_sgrid - 1;
  const unsigned char* base = (const unsigned char*)workspace;
  const int64_t* stage_index = (const int64_t*)base;
  const void* stage_frac = base + mlp_ws_frac_offset(n_steps);
  const float* img = (const float*)(base + mlp_ws_image_offset(n_steps));
  hipStream_t s = (hipStream_t)stream;
  if (time_dtype == CDE_F32)
    return cde::launch_mlp_adjoint_sweep<float>(coeffs, knots, n_intervals, degree, act, img, y_state, a_state, sgrid,
                                                k_begin, k_end, stage_index, stage_frac, U, G2, G1, Z, B, C, H,
                                                grad_coeffs, s);
  if (time_dtype == CDE_F64)
    return cde::launch_mlp_adjoint_sweep<double>(coeffs, knots, n_intervals, degree, act, img, y_state, a_state, sgrid,
                                                 k_begin, k_end, stage_index, stage_frac, U, G2, G1, Z, B, C, H,
                                                grad_coeffs, s);
  return CDE_ERR_DTYPE;
}
