"""CPU oracle for the stochopy population hot path -- TEST INFRASTRUCTURE ONLY.

This package is a numpy restatement of the per-generation path of
keurfonluu/stochopy v2.3.0 (batched objective evaluation, DE mutation /
crossover / selection, PSO / CPSO velocity-position update with competitive
restart, CMA-ES sampling and covariance update; plus, from the "next" rows,
the CMA-ES "Penalize" boundary handling and VD-CMA).  Every function cites the
reference file:line it follows (paths relative to the reference checkout).

Parity status: PINNED.  tests/test_oracle_golden.py checks this oracle against
golden vectors captured by running the reference itself
(tests/golden/make_golden.py, numpy 2.2.6): the reference's own test-suite
xrefs (tests/test_optimize.py), the objective known answers
(tests/test_factory.py), the README example, the BASELINE.json configs and
mid-size coverage cases, CMA-ES with Penalize and VD-CMA -- bit-for-bit for
DE/PSO/CPSO/CMA-ES/VD-CMA state.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
import this package, and only as the checker / the timed CPU baseline.  The
product (stochopy_amd/) never imports it; the product path fails loudly when
the HIP library is missing.

Third-party arithmetic on the path (absent from the reference tree, named
here): numpy's legacy `RandomState` MT19937 stream and pairwise `add.reduce`
(numpy, unpinned by the reference's setup.cfg:27-30; 2.2.6 here) and LAPACK
`syevd` through `numpy.linalg.eigh` (cmaes/_cmaes.py:304).  The oracle calls
numpy for those, exactly as the reference does.
"""
from .objectives import OBJECTIVES, evaluate  # noqa: F401
from .streams import LegacyStream, PhiloxStream, philox4x32_10  # noqa: F401
from .engine import minimize  # noqa: F401
