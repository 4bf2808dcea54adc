"""Build + ctypes binding of libcde_mi355x.so (the C ABI declared in include/cde_mi355x.h).

The shared object is built in-tree by ``build()`` (hipcc, gfx950 only) and loaded lazily.
There is NO fallback: if the library is missing or a call returns a non-zero code, a
RuntimeError is raised.
"""
import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
# CDE_PHASE_TRACE=1: the debug build whose attempt kernels stamp their phase boundaries (csrc/cde_common.h "phase trace",
# scripts/phase_trace.py).  A different file, so the product library is never the instrumented one.
PHASE_TRACE = os.environ.get("CDE_PHASE_TRACE", "") == "1"
SO_PATH = os.path.join(_HERE, "libcde_mi355x_trace.so" if PHASE_TRACE else "libcde_mi355x.so")
SOURCES = ["interp_kernels.hip", "rk4_generic.hip", "rk4_mfma.hip", "rk4_split.hip", "rk4_wide.hip", "rk4_mlp_adjoint.hip",
           "rk4_bf16x3.hip", "rk4_backprop.hip", "rk4_adjoint_pair.hip", "dopri5.hip", "dopri5_adjoint.hip", "dopri5_mlp_adjoint.hip", "mlp_grad_reduce.hip", "api.hip"]
HEADERS = [os.path.join(_CSRC, "cde_common.h"), os.path.join(_CSRC, "cde_mfma.h"), os.path.join(_CSRC, "cde_split.h"),
           os.path.join(_CSRC, "cde_dopri.h"), os.path.join(_CSRC, "cde_dopri_adj.h"), os.path.join(_CSRC, "cde_dopri_ctl.h"),
           os.path.join(_CSRC, "cde_mlp_adj.h"),
           os.path.join(_HERE, "..", "include", "cde_mi355x.h")]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC"]
if PHASE_TRACE:
    HIPCC_FLAGS.append("-DCDE_PHASE_TRACE")
# per-file additions.  rk4_split.hip: keep MFMA accumulators in VGPRs -- its tiles are consumed by VALU code right
# away, and on gfx950 every v_accvgpr_read costs matrix-pipe time (f32 MFMA and VALU do not overlap within a wave).
EXTRA_FLAGS = {"rk4_split.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               "rk4_wide.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               "rk4_adjoint_pair.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"],
               "dopri5_adjoint.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}

F32, F64 = 0, 1
PATH_LINEAR, PATH_CUBIC = 1, 3
EVAL_VALUE, EVAL_DERIVATIVE = 0, 1
ACT_NONE, ACT_TANH = 0, 1
VARIANT_AUTO, VARIANT_GENERIC, VARIANT_MFMA, VARIANT_SPLIT, VARIANT_BF16X3 = 0, 1, 2, 3, 4
METHOD_RK4, METHOD_MIDPOINT, METHOD_EULER = 0, 1, 2
FIXED_METHODS = {"rk4": METHOD_RK4, "midpoint": METHOD_MIDPOINT, "euler": METHOD_EULER}

ABI_VERSION = 3                     # == CDE_ABI_VERSION of include/cde_mi355x.h
_lib = None

# The library's tuning table (include/cde_mi355x.h, CDE_OPT_*): name -> (key, accepted spellings).  Tests and measurement
# scripts switch kernel forms with `with torchcde_amd.tuning(k3_form="product"): ...`; nothing is read from the environment.
OPTIONS = {
    "k3_form": (0, {"jacobian": 0, "product": 1}),
    "k3_waves": (1, {"default": 0, 1: 1, 2: 2}),
    "k3d_waves": (2, {"default": 0, 1: 1, 2: 2}),
    "k2m_no_split": (3, None), "k3m_no_split": (4, None), "k3m_split4": (5, None), "k3m_s8_tiles": (6, None),
    "k4_no_split": (7, None), "k4m_no_split": (8, None), "k4m_split_tiles": (9, None), "k4am_waves": (10, None),
    "k4am_s8_tiles": (11, None), "k4am_split4": (12, None), "k4am_no_split": (13, None),
    "k4am_no_small_reduce": (14, None), "k4am_sps": (15, None),
    "k4am_no_fsal": (16, {False: 0, True: 1, "accepted": 2, "rejected": 3}),
    "wide_scratch_bytes": (17, None),
}


def _option_value(name, value):
    key, spellings = OPTIONS[name]
    if spellings is not None and not isinstance(value, bool) and value in spellings:
        return key, spellings[value]
    if spellings is not None and isinstance(value, bool) and value in spellings:
        return key, spellings[value]
    return key, int(value)


def set_option(name, value):
    key, v = _option_value(name, value)
    check(load().cde_set_option(key, v), "cde_set_option(%s)" % name)


def get_option(name):
    return int(load().cde_get_option(OPTIONS[name][0]))


class tuning:
    """Context manager over the library's tuning table: `with tuning(k3_form="product", k3_waves=1): ...` sets the
    named options and restores the previous values on exit.  Process-wide (autograd runs backward on its own thread),
    so hold it around the whole solve, backward pass included."""

    def __init__(self, **options):
        for name in options:
            if name not in OPTIONS:
                raise KeyError("torchcde_amd.tuning: unknown option %r (known: %s)" % (name, ", ".join(sorted(OPTIONS))))
        self._new = options
        self._old = {}

    def __enter__(self):
        for name, value in self._new.items():
            self._old[name] = get_option(name)
            set_option(name, value)
        return self

    def __exit__(self, *exc):
        lib = load()
        for name, value in self._old.items():
            check(lib.cde_set_option(OPTIONS[name][0], value), "cde_set_option(%s)" % name)
        return False


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def _stale():
    if not os.path.exists(SO_PATH):
        return True
    built = os.path.getmtime(SO_PATH)
    deps = [os.path.join(_CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.exists(d) and os.path.getmtime(d) > built for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 into torchcde_amd/libcde_mi355x.so (cross-compiles without a GPU)."""
    if not force and not _stale():
        return SO_PATH
    # one hipcc process per translation unit, in parallel (the MFMA kernels dominate), then one link step
    compile_flags = [f for f in HIPCC_FLAGS if f != "-shared"]

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if proc.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + proc.stdout)

    # objects are cached under torchcde_amd/.build/ keyed by (source, headers, flags): editing one kernel file recompiles
    # that file only (the cache is neither tracked nor shipped to the GPU box; the linked .so is what travels).  Several
    # processes may get here at once (the ranks of a launcher over a stale library): one file lock around the whole build,
    # per-process temporary names, and the library itself is linked beside its final name and renamed into place.
    import fcntl
    import hashlib
    cache = os.path.join(_HERE, ".build")
    os.makedirs(cache, exist_ok=True)
    lock = open(os.path.join(cache, ".lock"), "w")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if not force and not _stale():            # another process built it while this one waited for the lock
            return SO_PATH
        return _build_locked(force, verbose, run, compile_flags, cache, hashlib)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _build_locked(force, verbose, run, compile_flags, cache, hashlib):
    tag = ".tmp%d" % os.getpid()
    import re

    def closure(path, seen):                      # the file and every "quoted" header it includes, transitively
        path = os.path.normpath(path)
        if path in seen or not os.path.exists(path):
            return
        seen[path] = open(path, "rb").read()
        for inc in re.findall(rb'^\s*#\s*include\s+"([^"]+)"', seen[path], flags=re.M):
            closure(os.path.join(os.path.dirname(path), inc.decode()), seen)

    objects, jobs = [], []
    for src in SOURCES:
        flags = compile_flags + EXTRA_FLAGS.get(src, [])
        seen = {}
        closure(os.path.join(_CSRC, src), seen)
        blob = b"".join(seen[k] for k in sorted(seen))
        key = hashlib.sha256(blob + " ".join(flags).encode()).hexdigest()[:20]
        obj = os.path.join(cache, "%s.%s.o" % (os.path.splitext(src)[0], key))
        objects.append(obj)
        if force or not os.path.exists(obj):
            jobs.append((src, obj, [_hipcc()] + flags + ["-c", os.path.join(_CSRC, src), "-o", obj + tag]))

    def compile_one(job):
        src, obj, cmd = job
        run(cmd)
        os.replace(obj + tag, obj)

    if jobs:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as pool:
            list(pool.map(compile_one, jobs))
    run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objects + ["-o", SO_PATH + tag])
    os.replace(SO_PATH + tag, SO_PATH)
    keep = set(objects)
    for name in os.listdir(cache):                  # drop objects of older source versions (never a lock or a temporary)
        if name.endswith(".o") and os.path.join(cache, name) not in keep:
            os.remove(os.path.join(cache, name))
    return SO_PATH


_p, _i, _i64, _sz, _d = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t, ctypes.c_double
_SIGNATURES = {
    "cde_abi_version": (_i, []),
    "cde_error_string": (ctypes.c_char_p, [_i]),
    "cde_set_option": (_i, [_i, _i64]),
    "cde_get_option": (_i64, [_i]),
    "cde_reset_options": (_i, []),
    "cde_hermite_bdiff_coeffs": (_i, [_p, _p, _p, _i64, _i64, _i64, _i, _p]),
    "cde_hermite_bdiff_coeffs_checked": (_i, [_p, _p, _p, _i64, _i64, _i64, _i, _p, _p]),
    "cde_hermite_bdiff_coeffs_nonblocking": (_i, [_p, _p, _p, _p, _i64, _i64, _i64, _i, _p, _i, _p]),
    "cde_linear_fill_missing": (_i, [_p, _p, _p, _i64, _i64, _i64, _i, _p]),
    "cde_linear_fill_missing_backward": (_i, [_p, _p, _p, _p, _i64, _i64, _i64, _i, _p]),
    "cde_interpret_t": (_i, [_p, _i64, _p, _i64, _p, _p, _i, _p]),
    "cde_hermite_bdiff_coeffs_backward": (_i, [_p, _p, _p, _i64, _i64, _i64, _i, _p]),
    "cde_hermite_bdiff_coeffs_backward_dt": (_i, [_p, _p, _p, _p, _i64, _i64, _i64, _i, _p]),
    "cde_natural_cubic_coeffs": (_i, [_p, _p, _p, _i64, _i64, _i64, _i, _i, _i, _p]),
    "cde_natural_cubic_coeffs_backward_workspace_bytes": (_sz, [_i64, _i]),
    "cde_natural_cubic_coeffs_backward": (_i, [_p, _p, _p, _p, _sz, _i64, _i64, _i64, _i, _p, _p, _p, _p]),
    "cde_natural_cubic_coeffs_backward_missing": (_i, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i, _i, _p]),
    "cde_logsig_windows": (_i, [_p, _p, _p, _p, _p, _i64, _i64, _i64, _i, _i64, _i, _i, _p]),
    "cde_logsig_windows_backward": (_i, [_p, _p, _p, _p, _p, _p, _p, _i64, _i64, _i64, _i, _i64, _i, _i, _p]),
    "cde_forward_fill": (_i, [_p, _p, _i64, _i64, _i64, _i, _p]),
    "cde_rectilinear_prepare": (_i, [_p, _p, _i64, _i64, _i64, _i64, _i, _p]),
    "cde_forward_fill_backward": (_i, [_p, _p, _p, _i64, _i64, _i64, _i, _p]),
    "cde_rectilinear_prepare_backward": (_i, [_p, _p, _p, _i64, _i64, _i64, _i64, _i, _p]),
    "cde_path_eval": (_i, [_p, _p, _p, _i64, _p, _i64, _i64, _i64, _i, _i, _i, _p]),
    "cde_path_eval_backward": (_i, [_p, _p, _p, _i64, _p, _i64, _i64, _i64, _i, _i, _i, _p]),
    "cde_contract": (_i, [_p, _p, _p, _i64, _i64, _i64, _i, _p]),
    "cde_rk4_supported": (_i, [_i64, _i64, _i, _i, _i, _i]),
    "cde_rk4_forward_linear": (_i, [_p, _p, _i64, _i, _p, _p, _i, _p, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i, _i,
                                    _i, _p, _p, _p]),
    "cde_rk4_forward_mlp": (_i, [_p, _p, _i64, _i, _p, _p, _i64, _p, _p, _i, _p, _p, _i64, _p, _i64, _p, _i64, _i64, _i64,
                                 _i, _i, _p, _p, _p]),
    "cde_rk4_adjoint_workspace_bytes": (_sz, [_i64, _i64, _i64, _i64, _i, _i]),
    "cde_fixed_supported": (_i, [_i, _i64, _i64, _i, _i]),
    "cde_fixed_forward_linear": (_i, [_i, _p, _p, _i64, _i, _p, _p, _p, _p, _i64, _p, _i64, _p, _i64, _i64, _i64, _i, _i, _p, _p,
                                      _p]),
    "cde_fixed_adjoint_workspace_bytes": (_sz, [_i64, _i64]),
    "cde_fixed_adjoint_linear": (_i, [_i, _p, _p, _i64, _i, _p, _p, _p, _p, _p, _i64, _p, _i64, _p, _p, _p, _i64, _i64, _i64,
                                      _i, _i, _p, _sz, _p]),
    "cde_rk4_forward_mlp_stages": (_i, [_p, _p, _i64, _i, _p, _p, _i64, _p, _p, _i, _p, _p, _i64, _p, _i64, _p, _p, _i64, _i64,
                                        _i64, _i, _i, _p, _p, _p]),
    "cde_rk4_backprop_mlp_prepare": (_i, [_p, _i64, _p, _i64, _p, _p, _i64, _p, _p, _i64, _i64, _i, _i, _p, _sz, _p]),
    "cde_rk4_backprop_mlp_sweep": (_i, [_p, _p, _i64, _i, _i, _p, _p, _p, _i64, _i64, _i64, _p, _p, _p, _p, _i64, _i64, _i64,
                                        _i, _i, _p, _sz, _p]),
    "cde_rk4_backprop_mlp_sweep_dcontrol": (_i, [_p, _p, _i64, _i, _i, _p, _p, _p, _i64, _i64, _i64, _p, _p, _p, _p, _p, _i64,
                                                 _i64, _i64, _i, _i, _p, _sz, _p]),
    "cde_rk4_backprop_supported": (_i, [_i64, _i64, _i, _i]),
    "cde_rk4_forward_linear_stages": (_i, [_p, _p, _i64, _i, _p, _p, _i, _p, _p, _i64, _p, _i64, _p, _p, _i64, _i64, _i64, _i,
                                           _i, _p, _p, _p]),
    "cde_rk4_backprop_workspace_bytes": (_sz, [_i64]),
    "cde_rk4_backprop_linear": (_i, [_p, _p, _i64, _i, _p, _p, _i, _p, _p, _i64, _p, _i64, _p, _p, _p, _p, _p, _p, _i64, _i64,
                                     _i64, _i, _p, _p, _p, _sz, _p]),
    "cde_rk4_backprop_linear_dcontrol": (_i, [_p, _p, _i64, _i, _p, _p, _i, _p, _p, _i64, _p, _i64, _p, _p, _p, _p, _p, _p, _p,
                                              _i64, _i64, _i64, _i, _p, _p, _p, _sz, _p]),
    "cde_dopri5_workspace_bytes": (_sz, [_i64, _i64, _i64, _i]),
    "cde_dopri5_trace_offset": (_sz, [_i64, _i64, _i64, _i]),
    "cde_dopri5_advance": (_i, [_p, _p, _i64, _i, _p, _p, _i, _p, _p, _i64, _p, _i64, _d, _d, _d, _d, _d, _p, _i64, _i64,
                                _i64, _i, _i, _p, _sz, _i64, _i64, _p]),
    "cde_dopri5_advance_mlp": (_i, [_p, _p, _i64, _i, _p, _p, _i64, _p, _p, _i, _p, _p, _i64, _p, _i64, _d, _d, _d, _d, _d,
                                    _p, _i64, _i64, _i64, _i, _p, _sz, _i64, _i64, _p]),
    "cde_dopri5_adjoint_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "cde_dopri5_adjoint_trace_offset": (_sz, [_i64, _i64, _i64]),
    "cde_dopri5_adjoint_mlp_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "cde_dopri5_adjoint_mlp_trace_offset": (_sz, [_i64, _i64, _i64, _i]),
    "cde_dopri5_adjoint_mlp_gradient_offset": (_sz, [_i64, _i64, _i64]),
    "cde_dopri5_adjoint_mlp_gradient_upper_offset": (_sz, [_i64, _i64, _i64]),
    "cde_dopri5_adjoint_mlp_advance": (_i, [_p, _p, _i64, _i, _p, _p, _i64, _p, _p, _i, _p, _p, _d, _d, _p, _i64, _d, _d, _d,
                                            _d, _d, _i, _p, _i64, _i64, _i64, _i, _i, _p, _sz, _i64, _i64, _p]),
    "cde_dopri5_adjoint_state_sums": (_i, [_p, _sz, _i64, _i64, _i64, _i64, _p, _p]),
    "cde_dopri5_adjoint_apply_state_sums": (_i, [_p, _sz, _i64, _i64, _i64, _i64, _p, _p]),
    "cde_dopri5_adjoint_mlp_dcontrol_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "cde_dopri5_adjoint_mlp_advance_dcontrol": (_i, [_p, _p, _i64, _i, _p, _p, _i64, _p, _p, _i, _p, _p, _d, _d, _p, _i64, _d, _d,
                                                     _d, _d, _d, _i, _p, _i64, _i64, _i64, _i, _i, _p, _sz, _i64, _i64, _p, _i64,
                                                     _p, _p]),
    "cde_dopri5_adjoint_mlp_state_sums": (_i, [_p, _sz, _i64, _i64, _i64, _i64, _p, _p]),
    "cde_dopri5_adjoint_mlp_apply_state_sums": (_i, [_p, _sz, _i64, _i64, _i64, _i64, _p, _p]),
    "cde_dopri5_adjoint_mlp_reduced_count": (_sz, []),
    "cde_dopri5_adjoint_mlp_advance_sharded": (_i, [_p, _p, _i64, _i, _p, _p, _i64, _p, _p, _i, _p, _p, _d, _d, _p, _i64, _d,
                                                    _d, _d, _d, _d, _i, _p, _i64, _i64, _i64, _i, _i, _p, _sz, _i64, _p,
                                                    _i64, _p]),
    "cde_dopri5_adjoint_mlp_pending_sums": (_i, [_p, _sz, _i64, _i64, _i64, _i64, _p, _p]),
    "cde_dopri5_adjoint_mlp_apply_reduced": (_i, [_p, _sz, _i64, _i64, _i64, _d, _d, _i64, _p, _p]),
    "cde_dopri5_pending_sums_mlp": (_i, [_p, _sz, _i64, _i64, _i64, _i, _i64, _p, _p]),
    "cde_dopri5_advance_mlp_sharded": (_i, [_p, _p, _i64, _i, _p, _p, _i64, _p, _p, _i, _p, _p, _i64, _p, _i64, _d, _d, _d, _d,
                                            _d, _p, _i64, _i64, _i64, _i, _p, _sz, _i64, _p, _i64, _p]),
    "cde_dopri5_adjoint_status_stride": (_sz, []),
    "cde_dopri5_adjoint_carry_offset": (_sz, [_i64, _i64, _i64]),
    "cde_dopri5_adjoint_mlp_carry_offset": (_sz, [_i64, _i64, _i64]),
    "cde_dopri5_adjoint_attempt_trace_offset": (_sz, [_i64, _i64, _i64]),
    "cde_dopri5_adjoint_reduced_count": (_sz, []),
    "cde_dopri5_adjoint_advance": (_i, [_p, _p, _i64, _i, _p, _p, _i, _p, _p, _d, _d, _p, _i64, _d, _d, _d, _d, _d, _i, _p,
                                        _i64, _i64, _i64, _i, _i, _p, _sz, _i64, _i64, _p, _i64, _p]),
    "cde_dopri5_adjoint_dcontrol_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "cde_dopri5_adjoint_advance_dcontrol": (_i, [_p, _p, _i64, _i, _p, _p, _i, _p, _p, _d, _d, _p, _i64, _d, _d, _d, _d, _d, _i,
                                                 _p, _i64, _i64, _i64, _i, _i, _p, _sz, _i64, _i64, _p, _i64, _p, _p]),
    "cde_dopri5_adjoint_pending_sums": (_i, [_p, _sz, _i64, _i64, _i64, _i64, _p, _p]),
    "cde_dopri5_adjoint_apply_reduced": (_i, [_p, _sz, _i64, _i64, _i64, _d, _d, _i64, _p, _p]),
    "cde_dopri5_pending_sums": (_i, [_p, _sz, _i64, _i64, _i64, _i, _i, _i, _i64, _p, _p]),
    "cde_dopri5_advance_sharded": (_i, [_p, _p, _i64, _i, _p, _p, _i, _p, _p, _i64, _p, _i64, _d, _d, _d, _d, _d, _p, _i64,
                                        _i64, _i64, _i, _i, _p, _sz, _i64, _p, _i64, _p]),
    "cde_dopri5_adjoint_finish": (_i, [_p, _sz, _p, _p, _i64, _i64, _i64, _i, _p]),
    "cde_mlp_grad_reduce_workspace_bytes": (_sz, []),
    "cde_mlp_grad_reduce": (_i, [_p, _p, _i64, _i, _p, _p, _sz, _p]),
    "cde_rk4_adjoint_mlp_workspace_bytes": (_sz, [_i64]),
    "cde_rk4_adjoint_mlp_prepare": (_i, [_p, _i64, _p, _i64, _p, _p, _i64, _p, _p, _i64, _i64, _i, _i, _p, _sz, _p]),
    "cde_rk4_adjoint_mlp_sweep": (_i, [_p, _p, _i64, _i, _i, _p, _p, _p, _i64, _i64, _i64, _p, _p, _p, _p, _p, _i64,
                                       _i64, _i64, _i, _i, _p, _sz, _p]),
    "cde_rk4_adjoint_linear_dcontrol": (_i, [_p, _p, _i64, _i, _p, _p, _i, _p, _p, _p, _i64, _p, _i64, _p, _p, _p, _p,
                                             _i64, _i64, _i64, _i, _i, _p, _sz, _p]),
    "cde_rk4_adjoint_linear": (_i, [_p, _p, _i64, _i, _p, _p, _i, _p, _p, _p, _i64, _p, _p, _i64, _p, _p, _p, _i64, _i64,
                                    _i64, _i, _i, _i, _p, _sz, _p]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class DopriStatus(ctypes.Structure):
    """Mirror of cde_dopri5_status (include/cde_mi355x.h)."""
    _fields_ = [("t_lo", ctypes.c_double), ("t_hi", ctypes.c_double), ("dt", ctypes.c_double),
                ("t1_try", ctypes.c_double), ("dt_try", ctypes.c_double), ("h0", ctypes.c_double),
                ("i_out", ctypes.c_int64), ("i_jump", ctypes.c_int64), ("n_accept", ctypes.c_int64),
                ("n_reject", ctypes.c_int64), ("phase", ctypes.c_int32), ("on_jump", ctypes.c_int32),
                ("refresh", ctypes.c_int32), ("pad", ctypes.c_int32), ("slot", ctypes.c_int32),
                ("stored", ctypes.c_int32), ("hint_lo", ctypes.c_int32), ("hint_hi", ctypes.c_int32)]


def load():
    """Return the ctypes handle; raises (never falls back) if the extension is unavailable."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                "torchcde_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU/eager fallback." % SO_PATH)
        lib = ctypes.CDLL(SO_PATH)
        # the version first: an older library would otherwise fail with a bare AttributeError on a symbol it lacks
        try:
            lib.cde_abi_version.restype, lib.cde_abi_version.argtypes = _i, []
            version = lib.cde_abi_version()
        except AttributeError:
            version = None
        if version != ABI_VERSION:
            raise RuntimeError("torchcde_amd: %s has ABI version %s, this package needs %d -- rebuild it "
                               "(python -c 'import __graft_entry__ as g; g.build()')" % (SO_PATH, version, ABI_VERSION))
        for name, (res, args) in _SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError:
                raise RuntimeError("torchcde_amd: %s does not export %s (ABI mismatch) -- rebuild it" % (SO_PATH, name))
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(code, what):
    if code != 0:
        msg = load().cde_error_string(code).decode()
        raise RuntimeError("torchcde_amd: %s failed: %s (code %d)" % (what, msg, code))


def dtype_enum(dtype):
    if dtype == torch.float32:
        return F32
    if dtype == torch.float64:
        return F64
    raise NotImplementedError("torchcde_amd: only float32 and float64 are supported on the native path, got %s" % dtype)


def require_gpu(tensor, what):
    if not tensor.is_cuda:
        raise RuntimeError("torchcde_amd: %s must live on a ROCm device (got device %s). This package is the MI355X "
                           "native path and has no CPU fallback." % (what, tensor.device))


def stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(tensor):
    return ctypes.c_void_p(tensor.data_ptr()) if tensor is not None else ctypes.c_void_p(0)
