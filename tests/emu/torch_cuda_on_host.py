"""torch.cuda on the host, for DRY RUNS of bench.py over the emulated device (tests/emu/README.md).  apply() patches torch from outside: streams, events,
synchronize and mem_get_info become host stand-ins, torch.device("cuda", i) maps to the host (the emulated device's memory IS host memory: a host tensor's
data_ptr() is a valid "device" pointer for the emulated library).  Test infrastructure; the product and bench.py know nothing of it."""
import contextlib
import os
import time


def apply():
    import torch

    _cpu = torch.device("cpu")
    _real_device = torch.device

    class _Stream:
        cuda_stream = 0

        def __init__(self, *a, **k):
            pass

        def synchronize(self):
            pass

        def wait_stream(self, other):
            pass

    class _Event:
        def __init__(self, *a, **k):
            self.t = time.perf_counter()

        def record(self, stream=None):
            self.t = time.perf_counter()

        def synchronize(self):
            pass

        def elapsed_time(self, other):
            return 1e3 * (other.t - self.t)

    def _device(*a, **k):
        if a and isinstance(a[0], str) and a[0].startswith("cuda"):
            return _cpu
        return _real_device(*a, **k)

    stream = _Stream()
    torch.device = _device
    torch.cuda.is_available = lambda: True
    n_dev = int(os.environ.get("EMU_DEVICES", "1"))          # (tests/emu/dev/hip/hip_runtime.h reads the same variable)
    torch.cuda.device_count = lambda: n_dev
    torch.cuda.set_device = lambda d: None
    torch.cuda.current_stream = lambda *a, **k: stream
    torch.cuda.Stream = _Stream
    torch.cuda.Event = _Event
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    torch.cuda.device = lambda d: contextlib.nullcontext()
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.empty_cache = lambda: None
    torch.cuda.mem_get_info = lambda *a, **k: (48 << 30, 64 << 30)
    torch.Tensor.pin_memory = lambda self, *a, **k: self
    torch.Tensor.cuda = lambda self, *a, **k: self          # tensor.cuda(): the tensor itself — host memory is the emulated device's memory
