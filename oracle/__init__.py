"""CPU oracle of the Game.step() hot path -- TEST INFRASTRUCTURE, not the product.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package (see oracle/ftl_oracle.c for the pinning statement)."""
from .oracle import OracleEnv, OracleGazebo, build_oracle, load_oracle  # noqa: F401
