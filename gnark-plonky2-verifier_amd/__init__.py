"""MI355X-native batch Plonky2-verification engine -- host-side mirror of the reference's Go package surface.

    goldilocks.Chip, poseidon.GoldilocksChip / BN254Chip, challenger.Chip, fri.Chip, plonk.PlonkChip,
    verifier.VerifierChip (succinctlabs/gnark-plonky2-verifier)

Every operator is batch-first and runs in hand-written HIP kernels behind the C ABI of include/gpv.h (libgpv.so).
Import with importlib.import_module("gnark-plonky2-verifier_amd") (the directory name is not an identifier).
"""
from . import _lib  # noqa: F401
from . import types, variables, goldilocks, poseidon, challenger, fri, plonk, verifier  # noqa: F401
# `distributed` imports torch; load it on demand: importlib.import_module("gnark-plonky2-verifier_amd.distributed")
from ._lib import (ConfigError, Context, DeviceError, GpvError, Group, ShapeError, default_context, shard_bounds)  # noqa: F401

__all__ = ["types", "variables", "goldilocks", "poseidon", "challenger", "fri", "plonk", "verifier", "Context",
           "default_context", "Group", "shard_bounds", "GpvError", "ShapeError", "ConfigError", "DeviceError"]
