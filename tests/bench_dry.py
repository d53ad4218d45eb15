"""TEST-ONLY: bench.py's control flow on a machine without a GPU.

    python tests/bench_dry.py --gpus N --steps K --warmup W [...]      (same flags as bench.py)

Runs bench.main() with a CPU stand-in for the device: the emulated kernel (tests/wave_emu), gloo instead of RCCL,
wall-clock "events".  The JSON line it prints says so in `data`; its numbers mean nothing.  bench.py itself contains
none of this."""
import contextlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402


class _WallClockEvent:
    def __init__(self):
        self.t = 0.0

    def record(self, stream=None):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class EmuBackend:
    dry = True
    dist_backend = "gloo"

    def setup(self, distributed, local_rank):
        import torch
        import torch.distributed as dist
        if distributed:
            dist.init_process_group("gloo")
        return torch.device("cpu")

    def env_kwargs(self):
        from tests.emu_backend import library
        return {"_library": library()}

    def event(self):
        return _WallClockEvent()

    def synchronize(self, device):
        pass   # nothing asynchronous on the CPU path

    def current_stream(self, device):
        return None

    def new_stream(self, device):
        return None

    def stream_ctx(self, stream):
        return contextlib.nullcontext()


if __name__ == "__main__":
    bench.main(backend=EmuBackend())
