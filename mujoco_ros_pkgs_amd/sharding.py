"""Env sharding over the GPUs of one node + the only collectives of the path.

Envs never interact (the reference runs exactly one, mujoco_env.h:241), so rank r simply owns the contiguous
block ``[r*E, (r+1)*E)`` (weak scaling) and no data-path collective exists.  What crosses xGMI is the aggregate
observation -- one all-gather of the contiguous ``sensordata[E][S]`` fp64 array -- and a 16-double metrics vector
(``mjb_metrics``: 8 additive entries reduced with SUM, 8 maxima reduced with MAX).  RCCL on GPUs (backend "nccl"),
gloo in the CPU tests.

``OverlappedExchange`` is how bench.py issues both every launch: the engine's stream copies the two send buffers
into staging tensors (a device-to-device copy of < 1 MB) and records an event; a side stream waits for it and
runs the collectives while the engine's stream is already running the next K-step launch."""
from __future__ import annotations


def shard_range(rank: int, world: int, envs_per_rank: int):
    """Global env ids [lo, hi) owned by ``rank``; ``lo`` is also the Philox ``env_offset`` of the shard."""
    if not (0 <= rank < world) or envs_per_rank <= 0:
        raise ValueError("bad shard arguments")
    return rank * envs_per_rank, (rank + 1) * envs_per_rank


def _world(group=None):
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return 1
    return dist.get_world_size(group)


def gather_sensordata(local, out=None, group=None):
    """All-gather ``local`` ([E, S] fp64 tensor, contiguous) into ``out`` ([world*E, S]); returns ``out``.
    With world_size 1 (or no process group) it returns ``local`` untouched -- no collective is issued."""
    import torch
    import torch.distributed as dist

    world = _world(group)
    if world == 1:
        return local
    if out is None:
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out


def reduce_metrics(vec16, group=None):
    """Job-wide metrics from each rank's ``mjb_metrics`` vector (16 fp64, in place): entries [0:8] are summed over the
    ranks, entries [8:16] are maximised.  World size 1: untouched."""
    import torch.distributed as dist

    if vec16.numel() != 16:
        raise ValueError("metrics vector must have 16 entries")
    if _world(group) == 1:
        return vec16
    dist.all_reduce(vec16[:8], op=dist.ReduceOp.SUM, group=group)
    dist.all_reduce(vec16[8:], op=dist.ReduceOp.MAX, group=group)
    return vec16


class OverlappedExchange:
    """Per-launch sensordata all-gather + metrics all-reduce on a side stream, overlapped with the next launch.

    ``sens_local`` / ``metrics_local`` are zero-copy torch views of the engine's HBM buffers; ``engine_stream`` is the
    raw hipStream_t the engine launches on.  ``issue()`` is called right after a launch has been enqueued; it never
    blocks the host.  ``finish()`` makes the current results visible to the host (bench fence / readers)."""

    def __init__(self, sens_local, metrics_local, engine_stream, device, force=False):
        import torch
        self.torch = torch
        self.world = _world()
        self.active = self.world > 1 or force
        self.sens_local, self.metrics_local = sens_local, metrics_local
        self.eng = torch.cuda.ExternalStream(int(engine_stream), device=device)
        self.side = torch.cuda.Stream(device=device)
        self.sens_stage = torch.empty_like(sens_local)
        self.metrics = torch.zeros(16, dtype=torch.float64, device=device)
        self.sens_all = torch.empty((self.world * sens_local.shape[0], sens_local.shape[1]), dtype=torch.float64,
                                    device=device) if self.active else sens_local
        self.staged = torch.cuda.Event()
        self.done = torch.cuda.Event()
        self.done.record(self.side)

    def issue(self):
        torch = self.torch
        import torch.distributed as dist
        with torch.cuda.stream(self.eng):
            self.eng.wait_event(self.done)  # the previous exchange has finished reading the staging buffers
            self.metrics.copy_(self.metrics_local, non_blocking=True)
            if self.active:
                self.sens_stage.copy_(self.sens_local, non_blocking=True)
            self.staged.record(self.eng)
        with torch.cuda.stream(self.side):
            self.side.wait_event(self.staged)
            if self.active:
                dist.all_gather_into_tensor(self.sens_all, self.sens_stage)
                dist.all_reduce(self.metrics[:8], op=dist.ReduceOp.SUM)
                dist.all_reduce(self.metrics[8:], op=dist.ReduceOp.MAX)
            self.done.record(self.side)

    def finish(self):
        self.side.synchronize()
        return self.sens_all, self.metrics
