"""MI355X-native batched UAV-cellular RL environment (drop-in for the env hot path of
SamKnightGit/DRL_UAV_CellularNet: MobiEnvironment.step/reset + LTEChannel DL SINR + RPGM mobility).

Importing the package does not need a GPU; constructing an env does (no CPU fallback)."""
from ._capi import UavEnvError  # noqa: F401

__all__ = ["BatchedMobiEnv", "MobiEnvironment", "UavEnvError"]


def __getattr__(name):  # lazy: torch is only imported when an env class is requested
    if name == "BatchedMobiEnv":
        from .batched_env import BatchedMobiEnv
        return BatchedMobiEnv
    if name == "MobiEnvironment":
        from .mobile_env import MobiEnvironment
        return MobiEnvironment
    raise AttributeError(name)
