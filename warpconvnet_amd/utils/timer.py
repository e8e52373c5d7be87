"""Event timers on the current HIP stream (role of reference `warpconvnet/utils/timer.py:41-77`)."""
import time

import torch


class CUDATimer:
    """`with CUDATimer() as t: ...; t.elapsed_time` (milliseconds), HIP events on the current stream."""

    def __init__(self):
        self.start_event = torch.cuda.Event(enable_timing=True)
        self.end_event = torch.cuda.Event(enable_timing=True)
        self.elapsed_time = None

    def __enter__(self):
        torch.cuda.current_stream().synchronize()
        self.start_event.record()
        return self

    def __exit__(self, *exc):
        self.end_event.record()
        self.end_event.synchronize()
        self.elapsed_time = self.start_event.elapsed_time(self.end_event)
        return False


class Timer:
    """Wall-clock timer (seconds) for CPU paths."""

    def __enter__(self):
        self._t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        self.elapsed_time = time.perf_counter() - self._t0
        return False
