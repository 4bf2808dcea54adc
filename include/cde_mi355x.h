/*
 * cde_mi355x.h -- C ABI of libcde_mi355x.so: the MI355X (gfx950) Neural-CDE hot path.
 *
 * The reference (patrick-kidger/torchcde 0.2.5) is pure Python and has no FFI of its own; this
 * header is the boundary a maintainer would bind with ctypes.  Every entry
 * point names the reference code it replaces.  Conventions:
 *
 *   - all pointers are DEVICE pointers into caller-owned, contiguous, row-major buffers
 *     (the Python host passes torch.Tensor.data_ptr()); nothing is allocated or freed here;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every call is
 *     asynchronous on that stream, keeps no state between calls (the one exception is the explicit tuning table
 *     below, which only cde_set_option() writes) and is safe to capture in a hipGraph;
 *   - `dtype` / `time_dtype` are CDE_F32 or CDE_F64.  `dtype` is the arithmetic type of the
 *     state z, the control coefficients and the knots; `time_dtype` is the type the solver's
 *     time grid is stepped in (torchdiffeq keeps the grid in `t.dtype` and casts each stage
 *     time to the state dtype before evaluating the vector field);
 *   - return value: CDE_OK (0) or a negative CDE_ERR_* code; cde_error_string() explains it.
 *     The Python host raises on any non-zero code -- there is no CPU fallback.
 */
#ifndef CDE_MI355X_H
#define CDE_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CDE_ABI_VERSION 3

enum { CDE_F32 = 0, CDE_F64 = 1 };

enum {
  CDE_OK = 0,
  CDE_ERR_NULL = -1,        /* a required pointer is NULL                            */
  CDE_ERR_DTYPE = -2,       /* unknown dtype enum                                    */
  CDE_ERR_SHAPE = -3,       /* a size is out of range (e.g. L < 2, H*C*H too large)  */
  CDE_ERR_UNSUPPORTED = -4, /* combination not implemented by the requested variant  */
  CDE_ERR_WORKSPACE = -5,   /* workspace_bytes smaller than *_workspace_bytes() asks */
  CDE_ERR_LAUNCH = -6       /* hipLaunchKernel reported an error                     */
};

/* control path kinds (piecewise polynomial degree) */
enum { CDE_PATH_LINEAR = 1, CDE_PATH_CUBIC = 3 };

/* what to evaluate on a path */
enum { CDE_EVAL_VALUE = 0, CDE_EVAL_DERIVATIVE = 1 };

/* vector-field families the fused solvers understand:
 *   f(t, z) = act( reshape_{H x C}( W z + bias ) ),  W is (H*C, H) row-major, bias is (H*C)
 * CDE_ACT_NONE is the README field (reference README.md:42-49), CDE_ACT_TANH the one of
 * example/irregular_data.py:36-46.                                                       */
enum { CDE_ACT_NONE = 0, CDE_ACT_TANH = 1 };

/* kernel selection for the fused solvers */
enum {
  CDE_VARIANT_AUTO = 0,    /* f32: the MFMA kernels for H <= 32, C <= 8 (identity or tanh), the wide tile kernels for
                              H <= 64, C <= 8 or H <= 32, C <= 16 (rk4 forward / adjoint, dopri5 forward); anything
                              else: the generic kernels */
  CDE_VARIANT_GENERIC = 1, /* VALU kernel: any H, C, f32 or f64                             */
  CDE_VARIANT_MFMA = 2,    /* fail with CDE_ERR_UNSUPPORTED unless the MFMA kernel applies   */
  CDE_VARIANT_SPLIT = 3,   /* MFMA, one workgroup (4 waves) per 16 series: the latency-oriented kernels for small
                              per-GPU batches (strong scaling); AUTO picks them when B <= CDE_SPLIT_MAX_BATCH */
  CDE_VARIANT_BF16X3 = 4   /* opt-in: the weight GEMMs on the bf16 matrix pipe at float32 accuracy (every operand split
                              into three bf16 pieces, six piece products per block, f32 accumulate; csrc/rk4_bf16x3.hip).
                              f32, H <= 32, C <= 8, identity activation, rk4 forward and adjoint (no control gradients);
                              CDE_ERR_UNSUPPORTED otherwise.  Never chosen by AUTO. */
};
#define CDE_SPLIT_MAX_BATCH 16384

/* torchdiffeq's fixed-grid methods: rk4 is the 3/8 rule (rk4_alt_step_func), midpoint
 * y1 = y0 + dt f(t0 + dt/2, y0 + f(t0, y0) dt/2), euler y1 = y0 + dt f(t0, y0) */
enum { CDE_METHOD_RK4 = 0, CDE_METHOD_MIDPOINT = 1, CDE_METHOD_EULER = 2 };

/* Tuning table (tests and measurements only).  Kernel-form selectors and thresholds are NOT read from the environment:
 * they live in one process-wide table of atomics that only cde_set_option() writes (defaults = the production choice;
 * 0 everywhere unless stated).  This table is the library's only state of its own; set an option BEFORE the workspace
 * query of the solve it should apply to and hold it fixed until that solve's last launch is queued (the layout
 * functions and the launchers read it).  None of the options changes what is computed beyond summation order; every
 * form is tested against the oracle.  The host mirror is `torchcde_amd.tuning(...)` (a context manager). */
enum {
  CDE_OPT_K3_FORM = 0,          /* reverse sweep of the affine field: 0 shared Jacobian (default), 1 two GEMMs against W */
  CDE_OPT_K3_WAVES = 1,         /* its Jacobian form: 0 default (= 2), 1 one wave per tile (K3j), 2 chain + helper wave (K3p) */
  CDE_OPT_K3D_WAVES = 2,        /* the same two forms of the adjoint=False sweep K3d (identity activation) */
  CDE_OPT_K2M_NO_SPLIT = 3,     /* 1: K2m one wave per tile at every batch size */
  CDE_OPT_K3M_NO_SPLIT = 4,     /* 1: K3m one wave per tile at every batch size */
  CDE_OPT_K3M_SPLIT4 = 5,       /* 1: K3m's shared-tile form with four instead of eight waves */
  CDE_OPT_K3M_S8_TILES = 6,     /* -1 default; otherwise the tile count up to which K3m's eight-wave form runs (can only LOWER it) */
  CDE_OPT_K4_NO_SPLIT = 7,      /* 1: K4 (one-layer fields) one wave per tile at every batch size */
  CDE_OPT_K4M_NO_SPLIT = 8,     /* 1: K4 (two-layer field) one wave per tile at every batch size */
  CDE_OPT_K4M_SPLIT_TILES = 9,  /* -1 default; otherwise its shared-tile threshold (can only LOWER it) */
  CDE_OPT_K4AM_WAVES = 10,      /* 0 default; 8: the large-batch workgroup shape of K4am on any batch */
  CDE_OPT_K4AM_S8_TILES = 11,   /* -1 default; otherwise the tile count up to which K4am's eight-wave form runs (can only LOWER it) */
  CDE_OPT_K4AM_SPLIT4 = 12,     /* 1: K4am's shared-tile form with four instead of eight waves */
  CDE_OPT_K4AM_NO_SPLIT = 13,   /* 1: K4am one wave per tile at every batch size */
  CDE_OPT_K4AM_NO_SMALL_REDUCE = 14, /* 1: the split-K factor reduction + R kernel also for <= 128 series */
  CDE_OPT_K4AM_SPS = 15,        /* 0 default; 4 .. 40: slabs per stage of the factor reduction */
  CDE_OPT_K4AM_NO_FSAL = 16,    /* 0 default; 1 evaluate every first stage (bit-identical results), 2 after accepted
                                   steps only, 3 after rejected steps only */
  CDE_OPT_WIDE_SCRATCH_BYTES = 17, /* 0 default (4 GB); otherwise the chunk budget of the wide-shape sweep in bytes */
  CDE_OPT_COUNT = 18
};
/* Set / read one entry of the tuning table.  cde_set_option returns CDE_ERR_SHAPE for an unknown key; cde_get_option
 * returns INT64_MIN for one.  cde_reset_options() restores every default. */
int cde_set_option(int key, int64_t value);
int64_t cde_get_option(int key);
int cde_reset_options(void);

int cde_abi_version(void);
const char* cde_error_string(int code);

/* ---------------------------------------------------------------------------------------------
 * K1  Hermite cubic coefficients with backward differences.
 * Replaces torchcde/interpolation_hermite_cubic_bdiff.py:23-44 (and the helper at :5-20) on the
 * no-missing-value path of linear_interpolation_coeffs (interpolation_linear.py:169-171).
 *   x      (B, L, C)        observations
 *   t      (L)              strictly increasing knot times (the host materialises the default
 *                           linspace(0, L-1, L) exactly as the reference does at :35-36)
 *   coeffs (B, L-1, 4C)     out: [a | b | 2c | 3d] per interval, same floats as the reference
 * ------------------------------------------------------------------------------------------- */
int cde_hermite_bdiff_coeffs(const void* x, const void* t, void* coeffs, int64_t B, int64_t L, int64_t C, int dtype,
                             void* stream);

/* K1 with the reference's NaN scan (interpolation_linear.py:169) riding on the fit's own loads: *nan_flag (a zeroed
 * DEVICE int) is OR-ed with 1 when any value of x is NaN.  The caller then knows -- from 4 bytes instead of a second
 * pass over x -- whether `coeffs` stands or the missing values must be filled first (cde_linear_fill_missing) and the
 * fit repeated. */
int cde_hermite_bdiff_coeffs_checked(const void* x, const void* t, void* coeffs, int64_t B, int64_t L, int64_t C,
                                     int dtype, int* nan_flag, void* stream);

/* The same without any host round trip (round 4): three launches, no read-back.  (1) the checked fit of x, which raises
 * *nan_flag to `generation` (atomic max) if x holds a NaN; (2) the missing-value fill of x into `scratch` (B, L, C) and
 * (3) the fit of `scratch` over `coeffs` -- (2) and (3) do nothing unless *nan_flag == generation.  `nan_flag` is a DEVICE
 * int the caller keeps per stream and never zeroes again after creation; `generation` >= 1 must increase from call to
 * call on that flag.  Replaces the same reference lines as cde_hermite_bdiff_coeffs_checked
 * (interpolation_hermite_cubic_bdiff.py:23-44 over interpolation_linear.py:131-171) for data with or without gaps. */
int cde_hermite_bdiff_coeffs_nonblocking(const void* x, const void* t, void* coeffs, void* scratch, int64_t B, int64_t L,
                                         int64_t C, int dtype, int* nan_flag, int generation, void* stream);

/* K1 backward: dL/dx (B, L, C) from dL/dcoeffs (B, L-1, 4C) -- what autograd produces through the reference's eager
 * ops at interpolation_hermite_cubic_bdiff.py:5-44 (the fit is linear in x; this is its transpose).  Gradients
 * w.r.t. `t` are not produced. */
int cde_hermite_bdiff_coeffs_backward(const void* grad_coeffs, const void* t, void* grad_x, int64_t B, int64_t L,
                                      int64_t C, int dtype, void* stream);
/* The same fit w.r.t. the knot times (data without missing values): grad_h (B, L-1, C) = dL/d(t_{j+1} - t_j) per
 * series and channel; summed over B and C it gives dL/dt_{j+1} += . and dL/dt_j -= . (the reference's eager ops are
 * differentiable in `t` as well). */
int cde_hermite_bdiff_coeffs_backward_dt(const void* grad_coeffs, const void* x, const void* t, void* grad_h, int64_t B,
                                         int64_t L, int64_t C, int dtype, void* stream);

/* K1n  Natural cubic spline coefficients: natural_cubic_coeffs (version 1) / natural_cubic_spline_coeffs (version 0),
 * torchcde/interpolation_cubic.py:7-266 with the tridiagonal solve of misc.py:14-67; NaN = missing value.
 *   x (B, L, C), t (L) -> coeffs (B, L-1, 4C) = [a | b | 2c | 3d], same floats as the reference.
 *   has_missing: whether ANY entry of x is NaN (the reference then re-centres every interval of every path,
 *   :149-160; passing 0 for NaN-free data skips that pass -- the floats are the same either way). */
int cde_natural_cubic_coeffs(const void* x, const void* t, void* coeffs, int64_t B, int64_t L, int64_t C, int version,
                             int has_missing, int dtype, void* stream);
/* Its backward for paths WITHOUT missing entries: grad_coeffs (B, L-1, 4C) -> grad_x (B, L, C) (autograd through
 * interpolation_cubic.py:7-54 and the tridiagonal solve misc.py:14-67).  `workspace`: at least
 * cde_natural_cubic_coeffs_backward_workspace_bytes(L, dtype) bytes (the elimination factors of the knots).
 * grad_t_rows != NULL: also the gradient w.r.t. the knot times (reference test/test_tricks.py:21-49 passes the same
 * `t` to the fit and to the spline): pass `x` (the forward input) and two (B, L, C) buffers; grad_t_rows receives one
 * partial dL/dt row per scalar path, to be summed over B and C. */
size_t cde_natural_cubic_coeffs_backward_workspace_bytes(int64_t L, int dtype);
int cde_natural_cubic_coeffs_backward(const void* grad_coeffs, const void* t, void* grad_x, void* workspace,
                                      size_t workspace_bytes, int64_t B, int64_t L, int64_t C, int dtype, const void* x,
                                      void* kd_scratch, void* grad_t_rows, void* stream);
/* The same for batches WITH missing entries (gradient w.r.t. the values only; autograd through
 * interpolation_cubic.py:83-166: the compact solve over each path's own observed knots and the re-centring of every
 * interval).  `x` (B, L, C) is the forward input (NaN = missing), `version` as in the forward; `workspace` has the
 * size of the coefficients (B, L-1, 4C).  Missing entries receive 0, imputed end points pass their gradient to the
 * observation they copied. */
int cde_natural_cubic_coeffs_backward_missing(const void* grad_coeffs, const void* x, const void* t, void* grad_x,
                                              void* workspace, int64_t B, int64_t L, int64_t C, int version, int dtype,
                                              void* stream);

/* ---------------------------------------------------------------------------------------------
 * K0  Missing-value construction: the NaN path of linear_interpolation_coeffs
 * (torchcde/interpolation_linear.py:13-84, reached from :169-170 and therefore also the first step
 * of hermite_cubic_coefficients_with_backward_differences on irregular data).
 *   x (B, L, C) with NaN = missing; t (L); out (B, L, C) without NaNs, same floats as the reference.
 * ------------------------------------------------------------------------------------------- */
int cde_linear_fill_missing(const void* x, const void* t, void* out, int64_t B, int64_t L, int64_t C, int dtype,
                            void* stream);
/* Its backward w.r.t. the observed values (what autograd produces through interpolation_linear.py:58-69): grad_out
 * (B, L, C) -> grad_x (B, L, C), zero at the missing entries; `x` is the forward call's input (its NaN pattern). */
int cde_linear_fill_missing_backward(const void* grad_out, const void* x, const void* t, void* grad_x, int64_t B,
                                     int64_t L, int64_t C, int dtype, void* stream);

/* K0b  Forward fill along the length axis (torchcde/misc.py:103-126, the helper rectilinear preparation and the
 * reference's data pipelines use): x (B, L, C) -> out (B, L, C); NaNs take the latest earlier observation of their
 * scalar path, leading NaNs stay NaN.  Bit-exact (data movement only). */
int cde_forward_fill(const void* x, void* out, int64_t B, int64_t L, int64_t C, int dtype, void* stream);

/* K0c  Rectilinear preparation (torchcde/interpolation_linear.py:86-128, reached from
 * linear_interpolation_coeffs(x, rectilinear=time_index) at :152-162): x (B, L, C) -> out (B, 2L-1, C) with
 *   out[j][c] = forward_fill(x)[j/2][c]  (c != time_index),   out[j][time_index] = x[(j+1)/2][time_index].
 * The caller checks (as the reference asserts) that the time column holds no NaN.  Bit-exact. */
int cde_rectilinear_prepare(const void* x, void* out, int64_t B, int64_t L, int64_t C, int64_t time_index, int dtype,
                            void* stream);
/* Backward of K0b / K0c (the reference's index gathers are differentiable): every output entry is a copy of one input
 * entry, the gradients of all copies flow back to it; missing entries get 0.  `x` is the forward call's input.
 *   cde_forward_fill_backward:        grad_out (B, L, C)     -> grad_x (B, L, C)
 *   cde_rectilinear_prepare_backward: grad_out (B, 2L-1, C)  -> grad_x (B, L, C) */
int cde_forward_fill_backward(const void* grad_out, const void* x, void* grad_x, int64_t B, int64_t L, int64_t C, int dtype,
                              void* stream);
int cde_rectilinear_prepare_backward(const void* grad_out, const void* x, void* grad_x, int64_t B, int64_t L, int64_t C,
                                     int64_t time_index, int dtype, void* stream);

/* K5  The log-ODE transform: logsig_windows / logsignature_windows (torchcde/log_ode.py:15-133).  The caller has
 * merged the window boundaries into the series and filled them (log_ode.py:18-49; cde_linear_fill_missing).
 *   x (B, L, C) filled series;  rows (n_windows + 1) int64 boundary row of every window end (device);
 *   scale (n_windows) factor per window (1, or the window length for the deprecated variant);
 *   words (n_words, 2) int32 (level, flat index) of every Lyndon word in signatory's order;
 *   out (B, n_windows + 1, n_words): first row = first observation (padded with zeros), then the running sum of the
 *   windows' logsignatures.
 * (C <= 8, depth <= 3), (C <= 5, depth 4) or (C <= 32, depth <= 2); CDE_ERR_UNSUPPORTED otherwise.  The logsignature arithmetic replaces the
 * third-party `signatory` calls at log_ode.py:53,57,59 ("words" mode): parity with that package is unpinned. */
int cde_logsig_windows(const void* x, const int64_t* rows, const void* scale, const int32_t* words, void* out, int64_t B,
                       int64_t L, int64_t C, int depth, int64_t n_windows, int n_words, int dtype, void* stream);
/* K5 backward: grad_out (B, n_windows + 1, n_words) -> grad_x (B, L, C), the gradient w.r.t. the FILLED series the
 * forward call was given (what autograd produces through signatory's logsignature and the running sum of
 * log_ode.py:53-63; the merge and the fill in front of it have their own backward, cde_linear_fill_missing_backward).
 * Same tables and limits as the forward call; `workspace` has the size of grad_out.  The signature is stepped back
 * with S (x) exp(-d) instead of being stored (the reversibility signatory's backward uses). */
int cde_logsig_windows_backward(const void* grad_out, const void* x, const int64_t* rows, const void* scale,
                                const int32_t* words, void* grad_x, void* workspace, int64_t B, int64_t L, int64_t C,
                                int depth, int64_t n_windows, int n_words, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K1b  Interval lookup and path evaluation for a vector of query times.
 * cde_interpret_t replaces CubicSpline._interpret_t (interpolation_cubic.py:315-322) and
 * LinearInterpolation._interpret_t (interpolation_linear.py:203-210):
 *   index = clamp(bucketize(t, knots) - 1, 0, n_intervals - 1)   (int64, bit-exact)
 *   frac  = t - knots[index]
 * cde_path_eval replaces CubicSpline.evaluate/.derivative (interpolation_cubic.py:324-336) and
 * LinearInterpolation.evaluate/.derivative (interpolation_linear.py:212-225):
 *   coeffs  cubic: (B, n_intervals, 4C); linear: (B, n_intervals + 1, C) (the raw knots' values)
 *   knots   (n_intervals + 1)
 *   tq      (nq) query times, already in `dtype`
 *   out     (B, nq, C)
 * ------------------------------------------------------------------------------------------- */
int cde_interpret_t(const void* knots, int64_t n_intervals, const void* tq, int64_t nq, int64_t* index_out,
                    void* frac_out, int dtype, void* stream);
int cde_path_eval(const void* coeffs, const void* knots, const void* tq, int64_t nq, void* out, int64_t B,
                  int64_t n_intervals, int64_t C, int degree, int what, int dtype, void* stream);

/* Backward of cde_path_eval w.r.t. the coefficients (what autograd produces through the gathers of
 * interpolation_cubic.py:324-336 / interpolation_linear.py:212-225): grad_out (B, nq, C) -> grad_coeffs, shaped like
 * `coeffs` and ZEROED by the caller.  Gradients w.r.t. the query times are not produced. */
int cde_path_eval_backward(const void* grad_out, const void* knots, const void* tq, int64_t nq, void* grad_coeffs,
                           int64_t B, int64_t n_intervals, int64_t C, int degree, int what, int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Contraction of a materialised vector field with the control derivative,
 *   out[b, h] = sum_c F[b, h, c] * dX[b, c]
 * Replaces the batched mat-vec of _VectorField.forward (solver.py:130) on the step-wise path
 * used for vector fields the fused solvers do not recognise.
 * ------------------------------------------------------------------------------------------- */
int cde_contract(const void* F, const void* dX, void* out, int64_t B, int64_t H, int64_t C, int dtype, void* stream);

/* Whether the fused RK4 kernels take a vector field of this shape (1) or not (0): the MFMA kernels need f32,
 * H <= 32, C <= 8, the wide tile kernels f32 and H <= 64, C <= 8 or H <= 32, C <= 16 (variant AUTO only); the
 * generic kernels H <= 256 and one series' stage data (adjoint: H + C + H*C values) within 64 KB of LDS.  The Python host solves anything else step by step (torchcde_amd/stepwise.py) instead of failing in
 * the backward pass. */
int cde_rk4_supported(int64_t C, int64_t H, int dtype, int act, int adjoint, int variant);

/* ---------------------------------------------------------------------------------------------
 * K2  Fused fixed-grid RK4 (3/8 rule) solve of  dz/dt = f(t, z) dX/dt  for the affine family.
 * Replaces, for one cdeint call: _VectorField.forward (solver.py:117-135) x 4 per step,
 * CubicSpline.derivative / LinearInterpolation.derivative inside it, and the whole of
 * torchdiffeq.odeint(method='rk4') behind solver.py:226-227 (grid stepping, 3/8-rule stages,
 * linear interpolation onto the output times).
 *   coeffs, knots, n_intervals, degree   the control, as for cde_path_eval
 *   W (H*C, H), bias (H*C), act           the vector field
 *   z0     (B, H)
 *   grid   (n_grid)  solver grid in `time_dtype`, built by the host exactly as torchdiffeq does
 *                    (arange(n)*step + t[0], last point = t[-1]); grid[0] == t_out[0]
 *   t_out  (n_out)   output times in `time_dtype`
 *   z_out  (B, n_out, H)  out: z at the output times, already in cdeint's (..., T, H) layout
 *   stage_index (4*(n_grid-1)) int64, stage_frac (4*(n_grid-1)) `dtype`: REQUIRED device scratch.
 *                    The call first fills them with the interval index / fractional part of
 *                    every stage time (what CubicSpline._interpret_t returns for it), then the
 *                    solve kernel reads them; callers may inspect them afterwards as the trace
 *                    for bit-exact index parity checks.
 * ------------------------------------------------------------------------------------------- */
int cde_rk4_forward_linear(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                           const void* bias, int act, const void* z0, const void* grid, int64_t n_grid,
                           const void* t_out, int64_t n_out, void* z_out, int64_t B, int64_t C, int64_t H, int dtype,
                           int time_dtype, int variant, int64_t* stage_index, void* stage_frac, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K2m  K2 for the two-layer vector field of the reference's examples
 *        f(z) = reshape_{HxC}( act( W2 relu(W1 z + b1) + b2 ) )
 * (example/time_series_classification.py:20-51: Linear(H,128) -> relu -> Linear(128,H*C) -> tanh);
 * replaces the same reference code as K2 plus the user module's two nn.Linear calls per stage.
 *   W1 (width, H), bias1 (width), W2 (H*C, width), bias2 (H*C);  act applies after the second layer.
 * f32 only, width <= 128, and (H <= 32, C <= 8) or (H <= 16, C <= 16) -- the sixteen 16-row MFMA tiles of the
 * second layer hold 32 hidden units x 8 channels or 16 x 16, zero padded; otherwise CDE_ERR_UNSUPPORTED.
 * (The 16 x 16 tiling takes the 14-channel depth-3 logsignature control of example/logsignature_example.py:22.)
 * Round 6: also 16 < H <= 32 with 8 < C <= 16 (that example at hidden_channels = 32) when width is a multiple of 4 and W2 is
 * 16-byte aligned: hidden units 16..31 run as four more unit groups whose rows of W2 / bias2 are read straight from the
 * caller's tensors (no second image).
 * All other arguments as cde_rk4_forward_linear.
 * ------------------------------------------------------------------------------------------- */
int cde_rk4_forward_mlp(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W1,
                        const void* bias1, int64_t width, const void* W2, const void* bias2, int act, const void* z0,
                        const void* grid, int64_t n_grid, const void* t_out, int64_t n_out, void* z_out, int64_t B,
                        int64_t C, int64_t H, int dtype, int time_dtype, int64_t* stage_index, void* stage_frac,
                        void* stream);

/* K3m's parameter-gradient reduction (replaces the two library GEMMs of round 1): acc += G^T [X | 1] over `rows`
 * (stage, series) rows that cde_rk4_adjoint_mlp_sweep streamed to HBM.  layer = 2: G (rows, 256), X (rows, 132),
 * acc (256, 132); layer = 1: G (rows, 128), X (rows, 36), acc (128, 36); the bias gradient is column 128 / 32.
 * Split-K on the matrix pipe with a fixed-order second pass: deterministic.  `workspace`: scratch of
 * cde_mlp_grad_reduce_workspace_bytes(). */
size_t cde_mlp_grad_reduce_workspace_bytes(void);
int cde_mlp_grad_reduce(const void* G, const void* X, int64_t rows, int layer, void* acc, void* workspace,
                        size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K3m  Continuous-adjoint reverse sweep for K2m (the two-layer field), replacing torchdiffeq.odeint_adjoint's
 * backward behind solver.py:226 for that vector field.  The parameter gradient of the output layer is a
 * (H*C) x width matrix per wave -- too large for on-chip accumulation -- so the sweep integrates (z, a) backwards
 * and STREAMS the per-stage factors to caller-provided HBM scratch; the caller reduces them with two GEMMs:
 *     [dL/dW2 | dL/db2] = G2^T U        [dL/dW1 | dL/db1] = G1^T Z
 *   scratch rows: row = (local_stage * B + series), local_stage = 4*(k - k_begin) + rk_stage
 *     U  (rows, 132) f32: relu(W1 z + b1) in columns 0..127 (zero beyond `width`); the CALLER sets column 128 to 1
 *                         and 129..131 to 0 once (the kernel never writes them)
 *     G2 (rows, 256) f32: quadrature-weighted dL/dY2, column h*8 + c (zero beyond the real H, C)
 *     G1 (rows, 128) f32: quadrature-weighted dL/dY1
 *     Z  (rows, 36)  f32: z in columns 0..H-1; the CALLER zeroes the buffer and sets column 32 to 1 once
 *   cde_rk4_adjoint_mlp_prepare   once per backward: stage table of the reversed-time grid `sgrid` (as for K3) and
 *                                 the MFMA weight images, into `workspace`
 *   cde_rk4_adjoint_mlp_sweep     integrates steps k_begin .. k_end-1 of `sgrid`; y_state / a_state (B, H) hold
 *                                 (z, a) on entry and on return (the caller re-seeds z and adds the incoming
 *                                 gradient between output intervals, as torchdiffeq does); `grad_coeffs` is NULL or a
 *                                 caller-zeroed buffer shaped like `coeffs` that accumulates dL/dcoeffs over the calls
 *                                 (as cde_rk4_adjoint_linear_dcontrol)
 * f32, width <= 128, (H <= 32, C <= 8) or (H <= 16, C <= 16) as K2m; `grad_coeffs` on both tile layouts (C > 8: the one-wave-per-tile form at every batch size).  G2's
 * columns are (hidden unit)*8 + channel for C <= 8 and (hidden unit)*16 + channel for 8 < C <= 16.
 * Round 6: 16 < H <= 32 with 8 < C <= 16 (any width <= 128: the sweep reads the upper rows of W2 from zero-padded copies that
 * _prepare leaves behind the weight images).  The dL/dY2 rows of hidden units 16..31 then form a SECOND block of G2, laid out
 * like the first and 4 * (k_end - k_begin) * B rows behind its start: the caller sizes G2 for both and reduces each against U
 * (cde_mlp_grad_reduce, layer 2) into its own [256][132] image.
 * ------------------------------------------------------------------------------------------- */
size_t cde_rk4_adjoint_mlp_workspace_bytes(int64_t n_sgrid);
int cde_rk4_adjoint_mlp_prepare(const void* knots, int64_t n_intervals, const void* sgrid, int64_t n_sgrid,
                                const void* W1, const void* bias1, int64_t width, const void* W2, const void* bias2,
                                int64_t C, int64_t H, int dtype, int time_dtype, void* workspace,
                                size_t workspace_bytes, void* stream);
int cde_rk4_adjoint_mlp_sweep(const void* coeffs, const void* knots, int64_t n_intervals, int degree, int act,
                              void* y_state, void* a_state, const void* sgrid, int64_t n_sgrid, int64_t k_begin,
                              int64_t k_end, void* U, void* G2, void* G1, void* Z, void* grad_coeffs, int64_t B,
                              int64_t C, int64_t H, int dtype, int time_dtype, const void* workspace,
                              size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K3  Fused continuous-adjoint reverse sweep for K2.
 * Replaces torchdiffeq.odeint_adjoint's backward (behind solver.py:226): for every output
 * interval, RK4 (3/8) integration in reversed time of the augmented state (z, a_z, a_W, a_b),
 * re-seeding z from the stored forward solution and adding the incoming gradient at every
 * output time.
 *   z_saved  (B, n_out, H)   forward solution at the output times (K2's z_out)
 *   grad_out (B, n_out, H)   dL/dz_out
 *   sgrid    (n_sgrid) concatenated reversed-time grids in `time_dtype`, one segment per output
 *            interval, processed in the order i = n_out-1 .. 1.  Segment p = n_out-1-i covers
 *            s in [-t_out[i], -t_out[i-1]], built exactly as torchdiffeq builds the grid of
 *            odeint(..., t[i-1:i+1].flip(0)).
 *   seg_off  (n_out) DEVICE array of int64: segment p is sgrid[seg_off[p] .. seg_off[p+1])
 *            (so seg_off[0] == 0 and seg_off[n_out-1] == n_sgrid)
 *   seg_off_host  the same n_out offsets in HOST memory (the caller built them on the host anyway).  Read only by the
 *            wide-shape path below, whose chunk loop runs on the host; may be NULL for every other shape
 *            (CDE_ERR_NULL if the wide path needs it).  ABI version 2: replaces the device-to-host copy + stream
 *            synchronisation version 1 performed inside this call.
 *   grad_z0 (B, H), grad_W (H*C, H), grad_b (H*C)   out
 *   workspace / workspace_bytes : device scratch of at least cde_rk4_adjoint_workspace_bytes()
 *            bytes: the reverse-sweep stage table plus per-workgroup partial parameter
 *            gradients, which are reduced in a fixed order (run-to-run deterministic, no atomics).
 * Wide shapes (H <= 64, C <= 8 or H <= 32, C <= 16 beyond the 32 x 8 tiles; csrc/rk4_wide.hip): the sweep keeps no
 * parameter gradients in registers; it streams 2.3 KB of per-stage factors per series to the workspace (chunks of RK
 * steps, at most 4 GB at a time; CDE_WIDE_SCRATCH_BYTES in the environment overrides the bound) and a split-K MFMA
 * reduction adds each chunk up.  That chunk loop runs on the host and takes its segment bounds from `seg_off_host`.
 * Affine fields (act == CDE_ACT_NONE) on the 32 x 8 tiles: f and a^T df/dz are evaluated from the shared Jacobian
 * J = sum_c dX_c W_c (one GEMM instead of two, K3j / the chain waves of the SPLIT variant) -- a reassociation of the same
 * sums, same tolerances against the reference.  CDE_K3_FORM=product in the environment selects the two-GEMM kernels
 * (read at every call; for tests and comparisons).
 * No entry point of this library synchronises a stream or copies device memory to the host: every call only queues
 * work on `stream` and can be captured into a hipGraph (tests: test_solver_calls_are_graph_capturable).
 * ------------------------------------------------------------------------------------------- */
size_t cde_rk4_adjoint_workspace_bytes(int64_t B, int64_t C, int64_t H, int64_t n_sgrid, int dtype, int variant);
int cde_rk4_adjoint_linear(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                           const void* bias, int act, const void* z_saved, const void* grad_out, const void* sgrid,
                           int64_t n_sgrid, const int64_t* seg_off, const int64_t* seg_off_host, int64_t n_out,
                           void* grad_z0, void* grad_W, void* grad_b, int64_t B, int64_t C, int64_t H, int dtype,
                           int time_dtype, int variant, void* workspace, size_t workspace_bytes, void* stream);

/* K3 with the gradient w.r.t. the control as well (adjoint_params containing the coefficient tensor, reference
 * solver.py:207-222 / README.md:251-270):
 *   grad_coeffs   same shape and layout as `coeffs`, ZEROED by the caller; on return dL/dcoeffs (cubic: the `a`
 *                 block stays zero -- the derivative of the spline does not read it; linear: dL/d(knot values)).
 * f32, H <= 32, C <= 8 (MFMA kernels; CDE_ERR_UNSUPPORTED otherwise).  Workspace as for cde_rk4_adjoint_linear with
 * variant = CDE_VARIANT_MFMA. */
int cde_rk4_adjoint_linear_dcontrol(const void* coeffs, const void* knots, int64_t n_intervals, int degree,
                                    const void* W, const void* bias, int act, const void* z_saved,
                                    const void* grad_out, const void* sgrid, int64_t n_sgrid, const int64_t* seg_off,
                                    int64_t n_out, void* grad_z0, void* grad_W, void* grad_b, void* grad_coeffs,
                                    int64_t B, int64_t C, int64_t H, int dtype, int time_dtype, void* workspace,
                                    size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * torchdiffeq's other fixed-grid methods, `midpoint` and `euler` (reference test/test_cdeint.py:49-63 solves with
 * method='midpoint'; solver.py:226-227 forwards `method` verbatim), for the affine field with act == CDE_ACT_NONE on the
 * 32 x 8 tiles (f32, H <= 32, C <= 8: cde_fixed_supported; otherwise CDE_ERR_UNSUPPORTED and the Python host steps the
 * solve itself).  K2 / K3p with two stages / one stage per step; the continuous adjoint integrates the augmented state
 * with the same method in reversed time (parameter gradients: quadrature weights (0, ds) / (ds)).  Arguments as
 * cde_rk4_forward_linear / cde_rk4_adjoint_linear without `act`, `variant`, `seg_off_host`; the stage table keeps four
 * slots per step (a method uses the first two / one); workspace of cde_fixed_adjoint_workspace_bytes(B, n_sgrid).
 * ------------------------------------------------------------------------------------------- */
int cde_fixed_supported(int method, int64_t C, int64_t H, int dtype, int act);
int cde_fixed_forward_linear(int method, const void* coeffs, const void* knots, int64_t n_intervals, int degree,
                             const void* W, const void* bias, const void* z0, const void* grid, int64_t n_grid,
                             const void* t_out, int64_t n_out, void* z_out, int64_t B, int64_t C, int64_t H, int dtype,
                             int time_dtype, int64_t* stage_index, void* stage_frac, void* stream);
size_t cde_fixed_adjoint_workspace_bytes(int64_t B, int64_t n_sgrid);
int cde_fixed_adjoint_linear(int method, const void* coeffs, const void* knots, int64_t n_intervals, int degree,
                             const void* W, const void* bias, const void* z_saved, const void* grad_out, const void* sgrid,
                             int64_t n_sgrid, const int64_t* seg_off, int64_t n_out, void* grad_z0, void* grad_W,
                             void* grad_b, int64_t B, int64_t C, int64_t H, int dtype, int time_dtype, void* workspace,
                             size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K3d  The backward pass of cdeint(..., method='rk4', adjoint=False) for the affine field (reference solver.py:144,
 * 226-227: torchdiffeq.odeint differentiated by autograd through the solver's own operations -- the EXACT gradient of
 * the discrete 3/8-rule map, README.md:103; not the continuous adjoint of K3).  Two calls:
 *   cde_rk4_forward_linear_stages   K2, which also stores the state handed to each of the 4 * (n_grid - 1) field
 *       evaluations:  stages (B, n_grid - 1, 4, 32) f32 -- row (series, step, stage) holds the 32 (zero padded) hidden
 *       units; act == CDE_ACT_NONE: in the order u -> (u & 1) * 16 + (u >> 1) (evens, then odds: the lane order of the
 *       32 x 32 MFMA tiles), act == CDE_ACT_TANH: in plain order.  All other arguments as cde_rk4_forward_linear.
 *   cde_rk4_backprop_linear   reverse-mode sweep over the stored stages (csrc/rk4_backprop.hip): identity activation -- one
 *       Jacobian GEMM + one matrix-vector product + the dL/dW product per stage; tanh -- the pre-activation GEMM, its
 *       transpose and the dL/dW product (K3a's stage), same `act` as the forward call:
 *         grad_out (B, n_out, H)      dL/dz_out
 *         step_dt  (n_steps) f32      float32(grid[k+1] - grid[k]), the step sizes the forward kernel used
 *         node_ptr (n_steps + 2) int64, node_out / node_weight (node_ptr[n_steps + 1]) int64 / f32:  CSR lists -- grid node
 *                  m receives  sum_e node_weight[e] * grad_out[:, node_out[e]]  for e in [node_ptr[m], node_ptr[m+1]):
 *                  the transpose of torchdiffeq's linear output interpolation (an output at a grid point is one entry
 *                  of weight 1, an output inside a step two entries on the step's end nodes)
 *         stage_index / stage_frac    the table cde_rk4_forward_linear_stages filled
 *         grad_z0 (B, H), grad_W (H*C, H), grad_b (H*C)   out
 *         workspace  cde_rk4_backprop_workspace_bytes(B) bytes: per-wave partial parameter gradients, reduced in a
 *                    fixed order (deterministic)
 * f32, H <= 32, C <= 8 (cde_rk4_backprop_supported); anything else CDE_ERR_UNSUPPORTED (the Python host then
 * differentiates the step-wise solve).
 * ------------------------------------------------------------------------------------------- */
int cde_rk4_backprop_supported(int64_t C, int64_t H, int dtype, int act);
int cde_rk4_forward_linear_stages(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                                  const void* bias, int act, const void* z0, const void* grid, int64_t n_grid,
                                  const void* t_out, int64_t n_out, void* z_out, void* stages, int64_t B, int64_t C, int64_t H,
                                  int dtype, int time_dtype, int64_t* stage_index, void* stage_frac, void* stream);
size_t cde_rk4_backprop_workspace_bytes(int64_t B);
int cde_rk4_backprop_linear(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                            const void* bias, int act, const void* stages, const void* grad_out, int64_t n_out,
                            const float* step_dt,
                            int64_t n_steps, const int64_t* node_ptr, const int64_t* node_out, const float* node_weight,
                            void* grad_z0, void* grad_W, void* grad_b, int64_t B, int64_t C, int64_t H, int dtype,
                            const int64_t* stage_index, const void* stage_frac, void* workspace, size_t workspace_bytes,
                            void* stream);
/* ... with the gradient w.r.t. the control's coefficient tensor as well: under adjoint=False autograd reaches the control
 * through X.derivative at every stage (reference solver.py:117-135; test/test_tricks.py:21-49 parametrises adjoint=False).
 * `grad_coeffs`: layout of `coeffs` (cubic (B, n_intervals, 4C): the b, 2c, 3d columns receive gradients; linear
 * (B, n_intervals + 1, C): the knot values), ZEROED by the caller, accumulated.  Both activations run the product-form stage
 * (the cotangent of dX_c is sum_h kb_h act(Y)_hc, a sum over rows a lane of that stage holds). */
int cde_rk4_backprop_linear_dcontrol(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                                     const void* bias, int act, const void* stages, const void* grad_out, int64_t n_out,
                                     const float* step_dt, int64_t n_steps, const int64_t* node_ptr, const int64_t* node_out,
                                     const float* node_weight, void* grad_z0, void* grad_W, void* grad_b, void* grad_coeffs,
                                     int64_t B, int64_t C, int64_t H, int dtype, const int64_t* stage_index,
                                     const void* stage_frac, void* workspace, size_t workspace_bytes, void* stream);

/* K3d for the two-layer field of the reference's examples (adjoint=False, method='rk4'): cde_rk4_forward_mlp_stages is
 * cde_rk4_forward_mlp (one wave per tile at any batch) that also stores `stages` (B, n_grid - 1, 4, 32), the 32 zero-padded
 * hidden units in plain order; cde_rk4_backprop_mlp_prepare fills the workspace of cde_rk4_adjoint_mlp_workspace_bytes(n_grid)
 * with the stage table of the FORWARD grid and the weight images; cde_rk4_backprop_mlp_sweep walks the steps
 * k_end-1 .. k_begin backwards (stage 4 first) in reverse mode: `g_state` (B, H) holds dL/dy of grid node k_end on entry and
 * of node k_begin on return (the caller adds the output gradients that land on a node between two calls), and the
 * unweighted gradient factors of the 4 (k_end - k_begin) evaluations are streamed to U / G2 / G1 / Z exactly as
 * cde_rk4_adjoint_mlp_sweep streams them (rows ((k_end-1-k) * 4 + (3-stage)) * B + series), for cde_mlp_grad_reduce. */
int cde_rk4_forward_mlp_stages(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W1,
                               const void* bias1, int64_t width, const void* W2, const void* bias2, int act,
                               const void* z0, const void* grid, int64_t n_grid, const void* t_out, int64_t n_out,
                               void* z_out, void* stages, int64_t B, int64_t C, int64_t H, int dtype, int time_dtype,
                               int64_t* stage_index, void* stage_frac, void* stream);
int cde_rk4_backprop_mlp_prepare(const void* knots, int64_t n_intervals, const void* grid, int64_t n_grid, const void* W1,
                                 const void* bias1, int64_t width, const void* W2, const void* bias2, int64_t C, int64_t H,
                                 int dtype, int time_dtype, void* workspace, size_t workspace_bytes, void* stream);
int cde_rk4_backprop_mlp_sweep(const void* coeffs, const void* knots, int64_t n_intervals, int degree, int act,
                               const void* stages, void* g_state, const void* grid, int64_t n_grid, int64_t k_begin,
                               int64_t k_end, void* U, void* G2, void* G1, void* Z, int64_t B, int64_t C, int64_t H,
                               int dtype, int time_dtype, const void* workspace, size_t workspace_bytes, void* stream);
/* ... with the gradient w.r.t. the control's coefficient tensor (`grad_coeffs` as for cde_rk4_backprop_linear_dcontrol: zeroed by
 * the caller before the first chunk, accumulated by every chunk's call) */
int cde_rk4_backprop_mlp_sweep_dcontrol(const void* coeffs, const void* knots, int64_t n_intervals, int degree, int act,
                                        const void* stages, void* g_state, const void* grid, int64_t n_grid, int64_t k_begin,
                                        int64_t k_end, void* U, void* G2, void* G1, void* Z, void* grad_coeffs, int64_t B,
                                        int64_t C, int64_t H, int dtype, int time_dtype, const void* workspace,
                                        size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K4  Adaptive Dormand-Prince 5(4) solve (torchdiffeq's default method, what cdeint runs when the
 * caller passes no `method`: reference solver.py:226-227, README.md:174) for the affine family.
 * Replaces torchdiffeq.odeint(method='dopri5', rtol, atol, options={'jump_t': ...}) including the
 * batch-global error norm, step-size controller, jump handling and dense output.
 *   t_out  (n_out) float64 DEVICE array, strictly increasing; jump_t (n_jump) float64 DEVICE array,
 *          sorted ascending (may be NULL when n_jump == 0)
 *   z_out  (B, n_out, H)
 *   One call queues `n_launches` attempt kernels (one attempted step each; the first three launches
 *   of a solve establish the initial step).  Call with first_launch = 0 to start, then continue with
 *   first_launch += n_launches until the controller block at the head of `workspace` reports done:
 *   workspace begins with two `cde_dopri5_status`-compatible structs; the one at index
 *   (total_launches & 1) is current.
 * Attempt kernels (variant AUTO): MFMA tiles for f32, H <= 32, C <= 8; the wide tile kernel for f32 and H <= 64,
 * C <= 8 or H <= 32, C <= 16; the generic kernel otherwise (any H <= 256, f32 / f64).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  double t_lo, t_hi, dt, t1_try, dt_try, h0;
  int64_t i_out, i_jump, n_accept, n_reject;
  int32_t phase; /* 4 == done */
  int32_t on_jump, refresh, pad;
  int32_t slot, stored; /* kernel-internal: which state slot holds the start of the pending step, what the attempt stored */
  int32_t hint_lo, hint_hi; /* kernel-internal: knot intervals of the pending attempt's first / last stage time */
} cde_dopri5_status;
size_t cde_dopri5_workspace_bytes(int64_t B, int64_t C, int64_t H, int dtype);
/* The solve's step sequence: the workspace holds, at byte offset cde_dopri5_trace_offset(...), up to
 * CDE_DOPRI5_TRACE_STEPS triples of float64 (t0, t1, j), one per ACCEPTED step in order (status.n_accept of them,
 * later steps are not recorded); j = 1.0 when the step was clipped onto a jump time (f is then re-evaluated just
 * after it), else 0.0.  torchdiffeq has one controller for the whole batch; shards of a batch solved on
 * several GPUs take their own sequences, and this trace is how a caller (or a test replaying the steps through the
 * oracle) sees which. */
#define CDE_DOPRI5_TRACE_STEPS 4096
size_t cde_dopri5_trace_offset(int64_t B, int64_t C, int64_t H, int dtype);
int cde_dopri5_advance(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                       const void* bias, int act, const void* z0, const double* t_out, int64_t n_out,
                       const double* jump_t, int64_t n_jump, double rtol, double atol, double safety,
                       double ifactor, double dfactor, void* z_out, int64_t B, int64_t C, int64_t H, int dtype,
                       int variant, void* workspace, size_t workspace_bytes, int64_t first_launch,
                       int64_t n_launches, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K4a  Backward of the adaptive solve: torchdiffeq.odeint_adjoint(method='dopri5') behind reference solver.py:226,
 * i.e. loss.backward() through torchcde's DEFAULT call cdeint(X, func, z0, t) (adjoint=True, no method).
 * f32, H <= 32, C <= 8, identity or tanh.  The host walks the output intervals from the last to the first; for
 * interval [t_{i-1}, t_i] it passes the reversed-time bounds s0 = -t_i < s1 = -t_{i-1}, the jump times negated and
 * ascending, y_init = z(t_i) (B, H) as stored by the forward solve and a_init = dL/dz(t_i) accumulated so far, and
 * calls cde_dopri5_adjoint_advance with first_launch = 0, n, 2n, ... until the cde_dopri5_status at the head of the
 * workspace (block index total_launches & 1, the two blocks cde_dopri5_adjoint_status_stride() bytes apart) reports
 * phase == 4; a_out (B, H) then holds dL/dz(t_{i-1}) before the incoming gradient of that output time is added.
 * first_interval bit 0 zeroes the running parameter gradients and vjp_t; after the last interval
 * cde_dopri5_adjoint_finish writes grad_W (H*C, H) and grad_b (H*C).
 * Output times that require a gradient (torchdiffeq's time_vjps, reference test/test_tricks.py:21-49): vjp_t is one double
 * at byte cde_dopri5_adjoint_carry_offset() of the workspace, carried from interval to interval.  torchdiffeq starts interval
 * i at vjp_t - f(t_i, y_i) . dL/dy_i: the caller subtracts that (float32 arithmetic) before the first call of every
 * interval -- with first_interval bit 1 set on the first one, which then keeps the caller's value instead of zeroing it --
 * and reads dL/dt_0 there after the last interval (dL/dt_i = f(t_i, y_i) . dL/dy_i for i >= 1).
 * The accepted steps of the CURRENT interval are traced like K4's, at cde_dopri5_adjoint_trace_offset(...); EVERY decided
 * attempt (at most 16384), rejected ones included, at cde_dopri5_adjoint_attempt_trace_offset(...) as 5 doubles
 * (t0, t1, clipped onto a jump, accepted, error ratio) -- the tests replay them through the oracle.
 * This IS torchdiffeq's algorithm (ABI version 2; version 1 used the state-only norm and clipped the last step):
 *   norm_kind 0  the default MIXED adjoint norm max(|e_t|, rms e_y, rms e_a, rms e_W, rms e_b) over the augmented state
 *                (vjp_t, y, a, dL/dW, dL/db) -- every launch of the attempt kernel is followed by a small reduction
 *                kernel over the workgroups' gradient images, inside this call;
 *   norm_kind 1  adjoint_options=dict(norm="seminorm"): the same without the parameter blocks;
 *   the last step of an interval passes the interval end and the dense interpolant is evaluated there.
 * ------------------------------------------------------------------------------------------- */
size_t cde_dopri5_adjoint_workspace_bytes(int64_t B, int64_t C, int64_t H);
size_t cde_dopri5_adjoint_trace_offset(int64_t B, int64_t C, int64_t H);
size_t cde_dopri5_adjoint_attempt_trace_offset(int64_t B, int64_t C, int64_t H);
size_t cde_dopri5_adjoint_status_stride(void);
size_t cde_dopri5_adjoint_carry_offset(int64_t B, int64_t C, int64_t H);
int cde_dopri5_adjoint_advance(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                               const void* bias, int act, const void* y_init, const void* a_init, double s0, double s1,
                               const double* jump_s, int64_t n_jump, double rtol, double atol, double safety,
                               double ifactor, double dfactor, int norm_kind, void* a_out, int64_t B, int64_t C,
                               int64_t H, int dtype, int first_interval, void* workspace, size_t workspace_bytes,
                               int64_t first_launch, int64_t n_launches, const double* reduced_sums, int64_t B_global,
                               void* stream);
/* Sharded batches, ONE controller (B_global > 0, n_launches == 1).  Per attempted step every shard runs
 *   cde_dopri5_adjoint_advance(first_launch = n, 1 launch, reduced_sums = the buffer below (NULL for n == 0), B_global)
 *   cde_dopri5_adjoint_pending_sums(total_launches = n + 1) -> cde_dopri5_adjoint_reduced_count() doubles on the device:
 *       8 state sums, then the A / E / D gradient images of the attempt
 *   an all-reduce (sum) of that buffer over the shards
 *   cde_dopri5_adjoint_apply_reduced(total_launches = n + 1, the reduced buffer): commit + parameter norms
 * and all shards take THE SAME decisions (those of the unsharded batch up to the summation order of the float images:
 * shards are added as doubles, an unsharded solve adds its workgroups' float images).  cde_dopri5_adjoint_finish(sharded = 1) returns THIS shard's
 * share of dL/dW, dL/db (the caller all-reduces gradients as for any data-parallel step). */
size_t cde_dopri5_adjoint_reduced_count(void);
int cde_dopri5_adjoint_pending_sums(void* workspace, size_t workspace_bytes, int64_t B, int64_t C, int64_t H,
                                    int64_t total_launches, double* sums, void* stream);
int cde_dopri5_adjoint_apply_reduced(void* workspace, size_t workspace_bytes, int64_t B, int64_t C, int64_t H,
                                     double rtol, double atol, int64_t total_launches, const double* reduced,
                                     void* stream);
int cde_dopri5_adjoint_finish(const void* workspace, size_t workspace_bytes, void* grad_W, void* grad_b, int64_t B,
                              int64_t C, int64_t H, int sharded, void* stream);
/* The same protocol under norm_kind = 1 ("seminorm"; round 4): the parameter blocks take no part in the decision, so only
 * the 8 state sums travel.  Per attempted step: cde_dopri5_adjoint_advance as above (it then also runs the R kernel on
 * THIS shard's images), cde_dopri5_adjoint_state_sums -> 8 doubles, all-reduce, cde_dopri5_adjoint_apply_state_sums (vjp_t
 * at an interval end from the reduced sums).  64 bytes per attempted step instead of 147 KB; the gradients stay per shard:
 * cde_dopri5_adjoint_finish(sharded = 0). */
int cde_dopri5_adjoint_state_sums(void* workspace, size_t workspace_bytes, int64_t B, int64_t C, int64_t H,
                                  int64_t total_launches, double* sums, void* stream);
int cde_dopri5_adjoint_apply_state_sums(void* workspace, size_t workspace_bytes, int64_t B, int64_t C, int64_t H,
                                        int64_t total_launches, const double* reduced, void* stream);
/* K4a with CONTROL gradients (round 6): adjoint_params = the field's parameters + the coefficient tensor the path was built
 * from (reference solver.py:207-222, README.md:251-270; test/test_tricks.py:21-49 with method='dopri5').  torchdiffeq then
 * integrates dL/dcoeffs as one more block of the augmented state and measures it in the mixed norm like dL/dW, dL/db: the
 * attempt kernel leaves, per series and stage, d(a.f)/d(dX_c) = sum_h a_h act(Y)_hc; a third small launch per attempted
 * step (one thread per (series, channel)) chains it to the coefficient rows the stages touch (cubic: 1, frac, frac^2 on
 * b, 2c, 3d; linear: -+1/width on the two knot values), commits accepted steps into `grad_coeffs` -- the block's running
 * total -- and leaves the block's norm sums for the next launch's controller.
 *   grad_coeffs    layout of `coeffs`, zeroed by the caller before the first interval; the same tensor for every interval
 *   control_numel  element count of the tensor named in adjoint_params (the block's rms runs over all of them)
 *   grad_knots     NULL, or (n_intervals + 1) floats zeroed by the caller: the knot times are in adjoint_params as well
 *                  (test/test_tricks.py:21-49 passes (coeffs, t)) -- a fourth block of the norm.  f depends on knot j through
 *                  frac = t - t_j (cubic) or through the widths of the slopes (linear); per stage ONE batch sum drives the
 *                  block (sum over the series of a . F d2X/dt2, or of a . f), which the attempt kernel leaves per workgroup
 * Workspace: cde_dopri5_adjoint_dcontrol_workspace_bytes (the plain layout is a prefix: the trace / carry / status offsets
 * above hold).  One controller per solve (no sharded form); everything else as cde_dopri5_adjoint_advance. */
size_t cde_dopri5_adjoint_dcontrol_workspace_bytes(int64_t B, int64_t C, int64_t H);
int cde_dopri5_adjoint_advance_dcontrol(const void* coeffs, const void* knots, int64_t n_intervals, int degree,
                                        const void* W, const void* bias, int act, const void* y_init, const void* a_init,
                                        double s0, double s1, const double* jump_s, int64_t n_jump, double rtol,
                                        double atol, double safety, double ifactor, double dfactor, int norm_kind,
                                        void* a_out, int64_t B, int64_t C, int64_t H, int dtype, int first_interval,
                                        void* workspace, size_t workspace_bytes, int64_t first_launch, int64_t n_launches,
                                        void* grad_coeffs, int64_t control_numel, void* grad_knots, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K4am  K4a for the two-layer field Linear(H, width) -> relu -> Linear(width, H*C) -> tanh | identity of the reference's
 * examples (example/time_series_classification.py:20-51; their training call is cdeint(X, func, z0, X.interval): dopri5
 * with adjoint=True, :83-86).  f32, width <= 128, (H <= 32, C <= 8) or (H <= 16, C <= 16); round 6: also 16 < H <= 32 with
 * 8 < C <= 16 (cde_dopri5_adjoint_mlp_gradient_upper_offset below; not in the sharded protocol).  Same protocol, controller,
 * norms (norm_kind) and status / trace blocks as cde_dopri5_adjoint_advance; per attempted step it queues the attempt
 * kernel, the split-K reduction of the attempt's gradient factors (both layers) and the commit / norm kernel.
 * The running parameter gradients live in the workspace at cde_dopri5_adjoint_mlp_gradient_offset():
 *   layer 2 as float [256][129]: row = the padded (hidden unit, channel) index (32 x 8 or 16 x 16 units x channels),
 *           columns 0..width-1 = dL/dW2, column 128 = dL/db2;   then layer 1 as float [128][33]: row = hidden-layer
 *           unit, columns 0..H-1 = dL/dW1, column 32 = dL/db1.
 * cde_dopri5_adjoint_mlp_trace_offset(which = 0 accepted steps | 1 every attempt) as for K4a.
 * Workspace footprint: the attempt streams the gradient factors of its six weighted stages, (132 + 256 + 128 + 36) floats
 * per series and stage, into SEVEN blocks of rows (the last stage alternates between two: an accepted step's last stage is
 * the next step's first -- torchdiffeq's first-same-as-last -- and is neither evaluated nor reduced again) = 15.5 KB per
 * series (507 MB at 32768 series), plus 6 x 40 slab partials of 37,248 floats (36 MB), three kept stage images and
 * 3 x (2 H + 4) floats per series of kept slopes; the factor rows are zeroed once per backward pass (first_interval), not
 * per attempt.  CDE_K4AM_NO_FSAL=1 in the environment evaluates every first stage (tests compare the two: identical bits).
 * ------------------------------------------------------------------------------------------- */
size_t cde_dopri5_adjoint_mlp_workspace_bytes(int64_t B, int64_t C, int64_t H);
size_t cde_dopri5_adjoint_mlp_trace_offset(int64_t B, int64_t C, int64_t H, int which);
size_t cde_dopri5_adjoint_mlp_gradient_offset(int64_t B, int64_t C, int64_t H);
/* 16 < H <= 32 with 8 < C <= 16 (round 6; one GPU's batch, not the sharded protocol): the kernels run the hidden units 16..31 as
 * a second 16 x 16 half -- their rows of the output layer from a zero-padded copy behind the weight images, their dL/dY2
 * rows, slab partials, kept stage images and running totals as a second layer-2 instance.  The totals of that half, again
 * float [256][129], live at this offset (0 for every other shape). */
size_t cde_dopri5_adjoint_mlp_gradient_upper_offset(int64_t B, int64_t C, int64_t H);
/* vjp_t of the two-layer solve (output-time gradients; first_interval bit 1): as cde_dopri5_adjoint_carry_offset */
size_t cde_dopri5_adjoint_mlp_carry_offset(int64_t B, int64_t C, int64_t H);
int cde_dopri5_adjoint_mlp_advance(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W1,
                                   const void* bias1, int64_t width, const void* W2, const void* bias2, int act,
                                   const void* y_init, const void* a_init, double s0, double s1, const double* jump_s,
                                   int64_t n_jump, double rtol, double atol, double safety, double ifactor,
                                   double dfactor, int norm_kind, void* a_out, int64_t B, int64_t C, int64_t H, int dtype,
                                   int first_interval, void* workspace, size_t workspace_bytes, int64_t first_launch,
                                   int64_t n_launches, void* stream);

/* K4am with control gradients (round 6): as cde_dopri5_adjoint_advance_dcontrol, for the two-layer field -- the
 * coefficient tensor (and optionally the knot times) as a fifth (and sixth) block of the adjoint state.  The evaluation
 * (cde_mlp_adj.h) also leaves d(a.f)/d(dX_c) = sum_h a_h act(Y2)_hc per stage; a first stage that is reused (first same as
 * last) takes its pending values from the previous launch's slot.  Eight-channel tiles run the four-wave / one-wave forms
 * of the attempt kernel (not the eight-wave one); the layout of cde_dopri5_adjoint_mlp_workspace_bytes is a prefix of
 * cde_dopri5_adjoint_mlp_dcontrol_workspace_bytes (the trace / carry / gradient offsets hold). */
size_t cde_dopri5_adjoint_mlp_dcontrol_workspace_bytes(int64_t B, int64_t C, int64_t H);
int cde_dopri5_adjoint_mlp_advance_dcontrol(const void* coeffs, const void* knots, int64_t n_intervals, int degree,
                                            const void* W1, const void* bias1, int64_t width, const void* W2,
                                            const void* bias2, int act, const void* y_init, const void* a_init, double s0,
                                            double s1, const double* jump_s, int64_t n_jump, double rtol, double atol,
                                            double safety, double ifactor, double dfactor, int norm_kind, void* a_out,
                                            int64_t B, int64_t C, int64_t H, int dtype, int first_interval, void* workspace,
                                            size_t workspace_bytes, int64_t first_launch, int64_t n_launches,
                                            void* grad_coeffs, int64_t control_numel, void* grad_knots, void* stream);

/* K4am under ONE step controller for a batch sharded over GPUs (round 4): the protocol of the one-layer kernels above.
 * Per attempted step n every shard runs
 *   cde_dopri5_adjoint_mlp_advance_sharded(first_launch = n, reduced_sums = the buffer below (NULL for n == 0), B_global)
 *       -- ONE attempt launch, no reduction
 *   cde_dopri5_adjoint_mlp_pending_sums(total_launches = n + 1) -> cde_dopri5_adjoint_mlp_reduced_count() doubles on the
 *       device: 8 state sums, then this shard's S (increment) and E (error) images of the four parameter tensors in the
 *       gradient layout (2 x 37,248 values)
 *   an all-reduce (sum) of that buffer over the shards
 *   cde_dopri5_adjoint_mlp_apply_reduced(total_launches = n + 1, the reduced buffer): commit + parameter norms on the
 *       GLOBAL images (every shard keeps the global running total for the norm and its OWN total, which is what
 *       cde_dopri5_adjoint_mlp_gradient_offset() points at: the caller all-reduces gradients as for any data-parallel step)
 * and all shards take the same decisions.  Replaces, for a sharded batch, the same reference call as K4am
 * (example/time_series_classification.py:83-86 over solver.py:199-203,226). */
size_t cde_dopri5_adjoint_mlp_reduced_count(void);
int cde_dopri5_adjoint_mlp_advance_sharded(const void* coeffs, const void* knots, int64_t n_intervals, int degree,
                                           const void* W1, const void* bias1, int64_t width, const void* W2,
                                           const void* bias2, int act, const void* y_init, const void* a_init, double s0,
                                           double s1, const double* jump_s, int64_t n_jump, double rtol, double atol,
                                           double safety, double ifactor, double dfactor, int norm_kind, void* a_out,
                                           int64_t B, int64_t C, int64_t H, int dtype, int first_interval, void* workspace,
                                           size_t workspace_bytes, int64_t first_launch, const double* reduced_sums,
                                           int64_t B_global, void* stream);
int cde_dopri5_adjoint_mlp_pending_sums(void* workspace, size_t workspace_bytes, int64_t B, int64_t C, int64_t H,
                                        int64_t total_launches, double* sums, void* stream);
int cde_dopri5_adjoint_mlp_apply_reduced(void* workspace, size_t workspace_bytes, int64_t B, int64_t C, int64_t H,
                                         double rtol, double atol, int64_t total_launches, const double* reduced,
                                         void* stream);
/* ... and under norm_kind = 1 ("seminorm"), as for the one-layer kernels: cde_dopri5_adjoint_mlp_advance_sharded (which then
 * also runs this shard's own factor reduction and R step), cde_dopri5_adjoint_mlp_state_sums -> 8 doubles, all-reduce,
 * cde_dopri5_adjoint_mlp_apply_state_sums.  64 bytes per attempted step instead of 596 KB. */
int cde_dopri5_adjoint_mlp_state_sums(void* workspace, size_t workspace_bytes, int64_t B, int64_t C, int64_t H,
                                      int64_t total_launches, double* sums, void* stream);
int cde_dopri5_adjoint_mlp_apply_state_sums(void* workspace, size_t workspace_bytes, int64_t B, int64_t C, int64_t H,
                                            int64_t total_launches, const double* reduced, void* stream);

/* Sharded batches under ONE step controller -- torchdiffeq's semantics for the whole batch when the batch lives on
 * several GPUs.  Per attempted step every shard (1) calls cde_dopri5_pending_sums(total_launches so far) -> 2 doubles on
 * the device, (2) all-reduces them (sum) with the other shards, (3) runs ONE launch through cde_dopri5_advance_sharded
 * with the reduced sums and the global number of series.  All shards then take bit-identical decisions and the same
 * (t0, t1) sequence; without this every shard runs its own controller (a valid solve, different steps). */
int cde_dopri5_pending_sums(const void* workspace, size_t workspace_bytes, int64_t B, int64_t C, int64_t H, int dtype,
                            int variant, int act, int64_t total_launches, double* sums, void* stream);
int cde_dopri5_advance_sharded(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W,
                               const void* bias, int act, const void* z0, const double* t_out, int64_t n_out,
                               const double* jump_t, int64_t n_jump, double rtol, double atol, double safety,
                               double ifactor, double dfactor, void* z_out, int64_t B, int64_t C, int64_t H, int dtype,
                               int variant, void* workspace, size_t workspace_bytes, int64_t first_launch,
                               const double* reduced_sums, int64_t B_global, void* stream);

/* K4 for the two-layer field of K2m (W1/bias1/width = hidden layer, W2/bias2 = output layer); f32, width <= 128,
 * (H <= 32, C <= 8) or (H <= 16, C <= 16), or (round 6, as cde_rk4_forward_mlp) 16 < H <= 32 with 8 < C <= 16.  Same protocol, workspace (cde_dopri5_workspace_bytes) and status block as
 * cde_dopri5_advance. */
int cde_dopri5_advance_mlp(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W1,
                           const void* bias1, int64_t width, const void* W2, const void* bias2, int act,
                           const void* z0, const double* t_out, int64_t n_out, const double* jump_t, int64_t n_jump,
                           double rtol, double atol, double safety, double ifactor, double dfactor, void* z_out,
                           int64_t B, int64_t C, int64_t H, int dtype, void* workspace, size_t workspace_bytes,
                           int64_t first_launch, int64_t n_launches, void* stream);
/* ... and under ONE controller for a sharded batch (round 4): cde_dopri5_pending_sums / cde_dopri5_advance_sharded for the
 * two-layer field (one launch per all-reduce of the 2 pending sums). */
int cde_dopri5_pending_sums_mlp(const void* workspace, size_t workspace_bytes, int64_t B, int64_t C, int64_t H, int dtype,
                                int64_t total_launches, double* sums, void* stream);
int cde_dopri5_advance_mlp_sharded(const void* coeffs, const void* knots, int64_t n_intervals, int degree, const void* W1,
                                   const void* bias1, int64_t width, const void* W2, const void* bias2, int act,
                                   const void* z0, const double* t_out, int64_t n_out, const double* jump_t, int64_t n_jump,
                                   double rtol, double atol, double safety, double ifactor, double dfactor, void* z_out,
                                   int64_t B, int64_t C, int64_t H, int dtype, void* workspace, size_t workspace_bytes,
                                   int64_t first_launch, const double* reduced_sums, int64_t B_global, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CDE_MI355X_H */
