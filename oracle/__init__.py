"""CPU oracle for the Neural-CDE hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``torchcde_amd/`` may import this package.  The only permitted
consumers are ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` -- and there only as the *checker*, never as the thing that
is measured or shipped.

What is restated, and what pins it
----------------------------------
* ``oracle.interp``  -- reference ``torchcde/interpolation_hermite_cubic_bdiff.py:5-44``,
  ``interpolation_cubic.py:282-336``, ``interpolation_linear.py:131-225``,
  ``misc.py:70-100``.  PINNED: bit-compared in this container against the
  imported reference (``oracle/make_golden.py``), fixtures under
  ``tests/golden/``, and against the reference's own closed-form known-answer
  test (``test/test_hermite_cubic.py:6-38``).
* ``oracle.cde``     -- reference ``torchcde/solver.py:103-245`` (vector-field
  wrapper, tolerance defaults, output permute).  PINNED the same way: the real
  ``torchcde.cdeint`` is executed by ``make_golden.py`` with ``oracle.odeint``
  standing in for the missing ``torchdiffeq`` module.
* ``oracle.odeint``  -- the third-party solver ``torchdiffeq`` (requirement
  ``torchdiffeq>=0.2.0``, unpinned, reference ``setup.py:51``; call sites
  ``torchcde/solver.py:226-227``).  That package is neither vendored under
  ``/root/reference`` nor installed here, and the reference's tests hold no
  value-level vector for its output.  **PARITY UNPINNED** for this half: the
  file restates the published algorithm (fixed-grid RK4 3/8-rule, dopri5,
  continuous adjoint) and is validated by mathematics instead -- order-4
  convergence to closed-form linear-CDE solutions, adjoint-vs-autograd
  agreement and float64 gradcheck (``tests/test_oracle.py``).  What CAN be pinned without the package is pinned
  against scipy's independent Dormand-Prince implementation: the tableau rational for rational, and -- value for
  value, 1e-13 -- the step, the derivative at its end, the embedded error estimate and the dense output over the
  oracle's own accepted-step sequence (``test_dopri5_steps_and_dense_output_match_scipy_value_for_value``).
  ``python oracle/pin_torchdiffeq.py`` closes the rest in one command wherever ``import torchdiffeq`` works: it runs
  the reference's ``solver.py`` over the REAL package on the seeded problems of ``tests/golden/cdeint.pt`` and diffs
  ``oracle.odeint`` against it (rk4: bitwise; dopri5: every stage time of every attempt, forward and adjoint, mixed
  norm and seminorm, ``jump_t``), ``--write`` regenerates the fixture from the real package.  (Its plumbing is
  exercised here by ``--self-test``, tests/test_oracle.py.)
* ``oracle.logsig``  -- the third-party ``signatory`` (call sites ``torchcde/log_ode.py:53,57,59``), absent as well.
  The windowing is pinned to the reference (``make_golden.py``); the logsignature arithmetic is pinned to an EXACT
  ``fractions.Fraction`` computation in the free tensor algebra on integer paths, depth 1-4, whose signature is itself
  checked by the shuffle identity (``test_logsignature_equals_an_exact_rational_tensor_algebra_reference``).
"""
