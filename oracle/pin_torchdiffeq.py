"""Close the "parity unpinned" gap of oracle/odeint.py in ONE command, wherever the real torchdiffeq is importable.

    python oracle/pin_torchdiffeq.py                 # diff oracle.odeint against the real package, exit code 0 / 1
    python oracle/pin_torchdiffeq.py --write         # additionally rewrite tests/golden/cdeint.pt from the real package

TEST INFRASTRUCTURE.  The reference delegates all time stepping to the third-party ``torchdiffeq`` (reference
``setup.py:51``: ``torchdiffeq>=0.2.0``, unpinned; call sites ``torchcde/solver.py:226-227``).  The package is neither
vendored under /root/reference nor installed in the build container (no network), so ``oracle/odeint.py`` restates it
from its published algorithm and is anchored by mathematics + scipy's independent Dormand-Prince implementation
(tests/test_oracle.py).  This script is the missing pin: it needs ``import torchdiffeq`` to succeed (``pip install
torchdiffeq`` on any machine with network), and torchcde itself either from /root/reference or pip.

What it compares, problem by problem (the seeded problems of tests/golden/cdeint.pt -- README toy, mid-size rk4 in
float32/float64, half steps with outputs off the grid, irregular knots, dopri5 -- plus adaptive problems with
``jump_t`` and the adjoint's default mixed norm / ``norm="seminorm"``):

  * fixed-grid ``rk4`` (torchdiffeq's 3/8 rule), forward and ``odeint_adjoint`` gradients:  BITWISE (torch.equal)
  * ``dopri5``: the accepted-step SEQUENCE (counts of accepted / rejected steps through a counting wrapper around
    ``func``: 6 evaluations per attempt + initial-step evaluations) and the outputs / gradients to 1e-6 relative
    (bitwise is reported too, but torchdiffeq versions differ in in-place vs. out-of-place stage sums)

  * every adjoint problem once more with output times that require a gradient: ``time_vjps`` (what the fused output-time
    gradients are held to) next to the other gradients

and it prints the torchdiffeq version it pinned against.  Until it has been run somewhere, DESIGN.md section 2 keeps
the words "parity unpinned" for the solver half.
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden", "cdeint.pt")


class _Counting(torch.nn.Module):
    """Counts vector-field evaluations: the step sequence of an adaptive solve, observable from outside."""

    def __init__(self, inner):
        super().__init__()
        self.inner, self.calls, self.times = inner, 0, []

    def forward(self, t, z):
        self.calls += 1
        self.times.append(float(t.detach()))
        return self.inner(t, z)


def _field(case):
    from tests.helpers import golden_field
    return golden_field(case)


def _solve(odeint_mod, cde_cdeint, X, func, case, adjoint, extra=None, probe_calls=0, time_grad=False):
    z0 = case["z0"].clone().requires_grad_(True)
    func.zero_grad()
    kwargs = dict(extra or {})
    if case["method"] is not None:
        kwargs["method"] = case["method"]
    if case["options"] is not None:
        kwargs["options"] = dict(case["options"])
    counted = _Counting(func)
    t_out = case["t_out"].clone().requires_grad_(True) if time_grad else case["t_out"]
    out = cde_cdeint(X=X, func=counted, z0=z0, t=t_out, adjoint=adjoint, **kwargs)
    weight = torch.linspace(0.5, 1.5, out.numel(), dtype=out.dtype).view_as(out)
    # `probe_calls`: evaluations the front end itself makes before the integrator runs (the reference's compatibility
    # probe func(t[0], z0), solver.py:47-53) -- not part of the step sequence
    forward_calls = counted.calls - probe_calls
    (out * weight).sum().backward()
    return dict(out=out.detach(), gz0=z0.grad.clone(), gW=func.linear.weight.grad.clone(),
                gb=func.linear.bias.grad.clone(), gt=t_out.grad.clone() if time_grad else torch.zeros(()),
                forward_calls=forward_calls, calls=counted.calls - probe_calls, times=list(counted.times[probe_calls:]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true", help="rewrite tests/golden/cdeint.pt with the real package's outputs")
    ap.add_argument("--self-test", action="store_true",
                    help="plumbing check: register oracle.odeint itself as `torchdiffeq` (every comparison must then pass)")
    opts = ap.parse_args()
    if opts.self_test:
        import types
        from oracle import odeint as stand_in
        shim = types.ModuleType("torchdiffeq")
        shim.odeint, shim.odeint_adjoint, shim.__version__ = stand_in.odeint, stand_in.odeint_adjoint, "SELF-TEST (oracle)"
        sys.modules["torchdiffeq"] = shim
    try:
        import torchdiffeq
    except ImportError:
        print("torchdiffeq is not importable here: nothing pinned (oracle/odeint.py stays PARITY UNPINNED).\n"
              "Run this script where `pip install torchdiffeq` is possible.")
        return 2
    version = getattr(torchdiffeq, "__version__", "unknown")
    print("pinning oracle.odeint against torchdiffeq", version)

    from oracle import cde as oracle_cde, interp as oracle_interp, odeint as oracle_ode
    real = {"odeint": torchdiffeq.odeint, "odeint_adjoint": torchdiffeq.odeint_adjoint}

    # the reference's own front end (solver.py:144-245) when torchcde is importable (from /root/reference or pip; its
    # other dependency torchsde is stubbed if absent -- the torchdiffeq backend never touches it), else the oracle's
    # restatement of that front end (bit-pinned to the reference by oracle/make_golden.py); the REAL integrators behind
    # it either way
    reference_front_end = None
    try:
        import types
        if "torchsde" not in sys.modules:
            try:
                import torchsde  # noqa: F401
            except ImportError:
                sys.modules["torchsde"] = types.ModuleType("torchsde")
        if os.path.isdir("/root/reference"):
            sys.path.insert(0, "/root/reference")
        import torchcde
        reference_front_end = torchcde.cdeint
        print("front end: torchcde.cdeint from", os.path.dirname(torchcde.__file__))
    except Exception as exc:                                            # noqa: BLE001
        print("front end: oracle.cde.cdeint (torchcde not importable: %s)" % exc)

    def reference_cdeint(**kw):
        if reference_front_end is not None:
            X = kw.pop("X")
            return reference_front_end(X=torchcde.CubicSpline(X._coeffs, X._t), **kw)
        return oracle_cde.cdeint(integrators=(real["odeint"], real["odeint_adjoint"]), **kw)

    def oracle_cdeint(**kw):
        return oracle_cde.cdeint(**kw)

    probe = 1 if reference_front_end is not None else 0

    cases = torch.load(GOLDEN)
    failures = 0
    rewritten = []
    for case in cases:
        X = oracle_interp.CubicPath(case["coeffs"], case["knots"])
        record = dict(case)
        for adjoint in (False, True):
            tag = "adjoint" if adjoint else "direct"
            want = _solve(torchdiffeq, reference_cdeint, X, _field(case), case, adjoint, probe_calls=probe)
            got = _solve(oracle_ode, oracle_cdeint, X, _field(case), case, adjoint)
            fixed = case["method"] in ("rk4", "midpoint", "euler")
            bitwise = all(torch.equal(got[k], want[k]) for k in ("out", "gz0", "gW", "gb"))
            close = all(torch.allclose(got[k], want[k], rtol=1e-6, atol=1e-9) for k in ("out", "gz0", "gW", "gb"))
            same_steps = got["calls"] == want["calls"] and got["forward_calls"] == want["forward_calls"]
            same_times = got["times"] == want["times"]
            ok = bitwise if fixed else (close and same_steps and same_times)
            failures += not ok
            print("%-32s %-8s %s  bitwise=%s close=%s evaluations oracle/real = %d/%d (forward %d/%d) stage times equal=%s"
                  % (case["name"], tag, "ok  " if ok else "FAIL", bitwise, close, got["calls"], want["calls"],
                     got["forward_calls"], want["forward_calls"], same_times))
            for k in ("out", "gz0", "gW", "gb"):
                record[k + "_" + tag] = want[k]
        rewritten.append(record)

    # adaptive extras: jump_t on the knots (README.md:194-200) and the adjoint norms
    extras = [dict(options=dict(jump_t=None)), dict(adjoint_options=dict(norm="seminorm"))]
    base = [c for c in cases if c["method"] in (None, "dopri5")]
    for case in base:
        X = oracle_interp.CubicPath(case["coeffs"], case["knots"])
        for extra in extras:
            extra = dict(extra)
            if "options" in extra:
                extra["options"] = dict(jump_t=X.grid_points)
            trial = dict(case)
            if "options" in extra:
                trial["options"] = extra.pop("options")
            want = _solve(torchdiffeq, reference_cdeint, X, _field(case), trial, True, extra, probe_calls=probe)
            got = _solve(oracle_ode, oracle_cdeint, X, _field(case), trial, True, extra)
            close = all(torch.allclose(got[k], want[k], rtol=1e-6, atol=1e-9) for k in ("out", "gz0", "gW", "gb"))
            ok = close and got["times"] == want["times"]
            failures += not ok
            print("%-32s %-8s %s  close=%s evaluations %d/%d stage times equal=%s  (%s)"
                  % (case["name"], "adjoint", "ok  " if ok else "FAIL", close, got["calls"], want["calls"],
                     got["times"] == want["times"], sorted(list(extra) + (["jump_t"] if trial["options"] else []))))

    # output times that require a gradient (torchdiffeq's time_vjps; reference test/test_tricks.py:21-49): what the fused
    # output-time gradients of K3 / K4a / K4am are held to
    for case in cases:
        X = oracle_interp.CubicPath(case["coeffs"], case["knots"])
        want = _solve(torchdiffeq, reference_cdeint, X, _field(case), case, True, probe_calls=probe, time_grad=True)
        got = _solve(oracle_ode, oracle_cdeint, X, _field(case), case, True, time_grad=True)
        fixed = case["method"] in ("rk4", "midpoint", "euler")
        keys = ("out", "gz0", "gW", "gb", "gt")
        bitwise = all(torch.equal(got[k], want[k]) for k in keys)
        close = all(torch.allclose(got[k], want[k], rtol=1e-6, atol=1e-9) for k in keys)
        ok = bitwise if fixed else (close and got["times"] == want["times"])
        failures += not ok
        print("%-32s %-8s %s  bitwise=%s close=%s stage times equal=%s  (t requires grad)"
              % (case["name"], "adjoint", "ok  " if ok else "FAIL", bitwise, close, got["times"] == want["times"]))

    if opts.write and not failures and not opts.self_test:
        for record in rewritten:
            record["pinned_against"] = "torchdiffeq " + version
        torch.save(rewritten, GOLDEN)
        print("tests/golden/cdeint.pt rewritten from torchdiffeq", version)
    print("PINNED: oracle.odeint == torchdiffeq %s on every case" % version if not failures
          else "%d case(s) differ: oracle/odeint.py does NOT restate this torchdiffeq version" % failures)
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
