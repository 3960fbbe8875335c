"""CPU oracle for the PIC hot path. TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  The product path
(``optimal-control-1d-electrostatic-plasma_amd``) never does: it fails loudly when the HIP
library is missing instead of falling back to anything in here.
"""
