"""CPU oracle for the Gaussian rasterizer hot path — TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  The product (``pf3plat_amd``) never does.  PARITY UNPINNED: see
``oracle/gsr_oracle.hpp`` for why (un-vendored CUDA dependency, no reference tests).
"""
from . import adapter, cameras, losses  # noqa: F401
from .gsr_oracle import OracleRasterizer, OracleResult, build_oracle, load_oracle  # noqa: F401
