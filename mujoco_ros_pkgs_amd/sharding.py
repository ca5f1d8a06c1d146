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


class DeviceStreams:
    """The stream / event layer ``OverlappedExchange`` runs on, as HIP streams through torch: the engine's own stream (raw
    hipStream_t wrapped as an ExternalStream) and one side stream for the collectives.  ``enqueue`` runs ``fn`` NOW under the
    stream's context -- everything ``fn`` issues is asynchronous device work on that stream."""

    def __init__(self, engine_stream, device):
        import torch
        self.torch = torch
        self.eng = torch.cuda.ExternalStream(int(engine_stream), device=device)
        self.side = torch.cuda.Stream(device=device)

    def new_event(self):
        return self.torch.cuda.Event()

    def stamp(self, stream):
        """A timing mark on `stream` (inside enqueue: the current stream)."""
        ev = self.torch.cuda.Event(enable_timing=True)
        ev.record(stream)
        return ev

    def between(self, a, b):
        return float(a.elapsed_time(b))  # ms; both completed (finish() synchronised their stream)

    def enqueue(self, stream, fn):
        with self.torch.cuda.stream(stream):
            fn()

    def record(self, event, stream):
        event.record(stream)

    def wait(self, stream, event):
        stream.wait_event(event)

    def synchronize(self, stream):
        stream.synchronize()


class ThreadStreams:
    """The same layer for hosts without a GPU (the world-size-2 gloo test): a stream is a FIFO worker thread, an event a
    marker that flows through it -- so work enqueued on two streams really does run concurrently and a missing ``wait`` is an
    observable race, exactly as on the device.  ``delay`` (seconds) is slept before every item of the side stream, widening
    the window in which a producer could overwrite a buffer the exchange has not read yet."""

    class _Stream:
        def __init__(self, delay=0.0):
            import queue
            import threading
            self.q = queue.Queue()
            self.delay = delay
            self.error = None
            self.t = threading.Thread(target=self._run, daemon=True)
            self.t.start()

        def _run(self):
            import time
            while True:
                fn = self.q.get()
                if fn is None:
                    return
                try:
                    if self.delay:
                        time.sleep(self.delay)
                    fn()
                except BaseException as exc:  # surfaced by synchronize()
                    self.error = exc
                finally:
                    self.q.task_done()

    class _Event:
        """Generation-counted like a device event: ``wait`` captures the most recent ``record`` at the time it is CALLED."""

        def __init__(self):
            import threading
            self.cv = threading.Condition()
            self.recorded = 0   # generations handed out by record()
            self.completed = 0  # generations their stream has reached

        def _complete(self, gen):
            with self.cv:
                if gen > self.completed:
                    self.completed = gen
                self.cv.notify_all()

        def _wait(self, gen):
            with self.cv:
                self.cv.wait_for(lambda: self.completed >= gen)

    def __init__(self, side_delay=0.0):
        self.eng = self._Stream()
        self.side = self._Stream(side_delay)

    def new_event(self):
        return self._Event()

    def stamp(self, stream):
        import time
        return time.perf_counter()  # (called from inside the stream's worker)

    def between(self, a, b):
        return 1e3 * (b - a)

    def enqueue(self, stream, fn):
        stream.q.put(fn)

    def record(self, event, stream):
        with event.cv:
            event.recorded += 1
            gen = event.recorded
        stream.q.put(lambda: event._complete(gen))

    def wait(self, stream, event):
        with event.cv:
            gen = event.recorded  # (a never-recorded event is complete, as on the device)
        stream.q.put(lambda: event._wait(gen))

    def synchronize(self, stream):
        stream.q.join()
        if stream.error is not None:
            raise stream.error

    def close(self):
        for st in (self.eng, self.side):
            st.q.put(None)


class OverlappedExchange:
    """Per-launch sensordata all-gather + metrics all-reduce on a side stream, overlapped with the next launch.

    ``sens_local`` / ``metrics_local`` are zero-copy torch views of the engine's HBM buffers; ``engine_stream`` is the
    raw hipStream_t the engine launches on.  ``issue()`` is called right after a launch has been enqueued; it never
    blocks the host.  ``finish()`` makes the current results visible to the host (bench fence / readers).

    Ordering contract (what tests/test_exchange_overlap.py drives with ``ThreadStreams``):
      engine stream:  launch k | wait(done k-1) | stage <- send buffers | record(staged k) | launch k+1 ...
      side stream:                                 wait(staged k) | collectives on the staging copies | record(done k)
    so launch k+1 may overwrite the send buffers while exchange k is in flight, and staging is not rewritten before exchange
    k-1 has read it."""

    def __init__(self, sens_local, metrics_local, engine_stream, device, force=False, streams=None):
        import torch
        self.torch = torch
        self.world = _world()
        self.active = self.world > 1 or force
        self.sens_local, self.metrics_local = sens_local, metrics_local
        self.rt = streams if streams is not None else DeviceStreams(engine_stream, device)
        self.eng, self.side = self.rt.eng, self.rt.side
        self.sens_stage = torch.empty_like(sens_local)
        self.metrics = torch.zeros(16, dtype=torch.float64, device=device)
        self.sens_all = torch.empty((self.world * sens_local.shape[0], sens_local.shape[1]), dtype=torch.float64,
                                    device=device) if self.active else sens_local
        self.staged = self.rt.new_event()
        self.done = self.rt.new_event()
        self.rt.record(self.done, self.side)
        self.issued = 0

    def _stage(self):
        self.metrics.copy_(self.metrics_local, non_blocking=True)
        if self.active:
            self.sens_stage.copy_(self.sens_local, non_blocking=True)

    def _collect(self):
        import torch.distributed as dist
        if self.active:
            dist.all_gather_into_tensor(self.sens_all, self.sens_stage)
            dist.all_reduce(self.metrics[:8], op=dist.ReduceOp.SUM)
            dist.all_reduce(self.metrics[8:], op=dist.ReduceOp.MAX)

    def _timed_collect(self):
        a = self.rt.stamp(self.side)
        self._collect()
        self._marks = (a, self.rt.stamp(self.side))

    def last_ms(self):
        """Duration of the most recent exchange on the side stream (after finish()); None before the first one."""
        marks = getattr(self, "_marks", None)
        if marks is None:
            return None
        self.rt.synchronize(self.side)
        return self.rt.between(*marks)

    def issue(self):
        rt = self.rt
        rt.wait(self.eng, self.done)  # the previous exchange has finished reading the staging buffers
        rt.enqueue(self.eng, self._stage)
        rt.record(self.staged, self.eng)
        rt.wait(self.side, self.staged)
        rt.enqueue(self.side, self._timed_collect)
        rt.record(self.done, self.side)
        self.issued += 1

    def finish(self):
        self.rt.synchronize(self.side)
        return self.sens_all, self.metrics
