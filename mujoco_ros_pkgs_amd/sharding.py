"""Env sharding over the GPUs of one node + the only collective of the path.

Envs never interact (the reference runs exactly one, mujoco_env.h:241), so rank r simply owns the contiguous
block ``[r*E, (r+1)*E)`` (weak scaling) and no data-path collective exists.  The aggregate observation is one
all-gather of the contiguous ``sensordata[E][S]`` fp64 array (RCCL on GPUs: backend "nccl"; gloo in CPU tests)."""
from __future__ import annotations


def shard_range(rank: int, world: int, envs_per_rank: int):
    """Global env ids [lo, hi) owned by ``rank``; ``lo`` is also the Philox ``env_offset`` of the shard."""
    if not (0 <= rank < world) or envs_per_rank <= 0:
        raise ValueError("bad shard arguments")
    return rank * envs_per_rank, (rank + 1) * envs_per_rank


def gather_sensordata(local, out=None, group=None):
    """All-gather ``local`` ([E, S] fp64 tensor, contiguous) into ``out`` ([world*E, S]); returns ``out``.
    With world_size 1 (or no process group) it returns ``local`` untouched -- no collective is issued."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out
