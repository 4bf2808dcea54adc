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
  agreement and float64 gradcheck (``tests/test_oracle.py``).
"""
