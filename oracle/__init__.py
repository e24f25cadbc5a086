"""CPU oracle for the octic hot path — TEST INFRASTRUCTURE, not product code.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
See oracle/octic_ref.py for the parity statement (pinned against tests/golden/*.npz).
"""
from . import octic_ref  # noqa: F401
