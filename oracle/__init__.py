"""CPU oracle for the Harmony hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  The product (``harmonypy_amd``) never
does; it fails loudly when the HIP extension is missing.
"""
from .harmony_oracle import OracleHarmony, oracle_run_harmony, prepare_inputs  # noqa: F401
# also here: sharded_oracle.py (cells sharded over ranks, CPU/gloo), device_order.py (integer restatement of the
# engine's device-side update order, compared bit for bit with the lists the GPU builds)
