"""ORACLE -- test infrastructure, NOT product code.

CPU restatement of the reference's STARK prover path (GuildOfWeavers/distaff v0.5.1), used as the checker for the
HIP path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package;
nothing under ``distaff_amd/`` does.  The C++ sources cite the reference file:line each function follows.
"""
from .pyoracle import *  # noqa: F401,F403
