"""CPU oracle package -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  ``polars_ols_amd`` (the product) never does; see
``oracle/pols_oracle.h`` for the scope statement and the pinning status.
"""
