"""CPU ORACLE — test infrastructure, NOT product code.

Restates the reference's algorithm for the PPO hot path on the CPU so that the
HIP path can be checked against it.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this package; ``pufferlib_amd`` never does.

Parity status: PINNED against outputs of the unmodified reference run in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``; checked by
``tests/test_oracle_golden.py``).
"""
