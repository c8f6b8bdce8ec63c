"""CPU oracle for the EfficientAT hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this package, and only as the checker / the timed CPU baseline -- never
as the thing shipped.  The product path (``efficientat_amd``) must not import it.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the unmodified reference
from ``/root/reference`` (with the stand-ins in ``oracle/ref_shims``) in the build
container and stores its outputs for seeded inputs/weights under ``tests/golden``;
``tests/test_oracle_golden.py`` checks this restatement against those vectors.
"""
