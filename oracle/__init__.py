"""CPU oracle for the GaPro pseudo-label hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it,
and only as the checker; ``gapro_amd`` never imports it and has no CPU fallback.

Parity status (see DESIGN.md "Oracle"):

* ``gen_ps_oracle``  -- restates reference gapro/gen_ps_utils.py:293-482 (partition,
  pair schedule, merge, fallback, labels).  PINNED: checked against the imported
  reference itself (tests/golden/make_golden.py, fixtures in tests/golden/*.npz).
* ``svgp_oracle``    -- restates the variational GP classifier that reference
  gapro/gaussian_process_utils.py:11-25,382-445 builds out of gpytorch.  gpytorch is
  an un-vendored, un-pinned third-party dependency absent from /root/reference and
  from this image, and the reference holds no test or golden vector for it:
  **GP numerics are "parity unpinned"** -- the restatement follows gpytorch 1.x's
  published algorithm (whitened VariationalStrategy, CholeskyVariationalDistribution,
  BernoulliLikelihood with 20-point Gauss-Hermite quadrature, VariationalELBO, Adam).
"""
