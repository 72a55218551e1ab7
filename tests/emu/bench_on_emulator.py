"""tests/emu/bench_on_emulator.py [bench.py arguments] — bench.py end to end WITHOUT a GPU: the emulated device (tests/emu/dev, ZKGL_LIB) behind the C ABI and
host tensors standing in for device tensors (the emulated device's memory IS host memory, a tensor's data_ptr() is a valid "device" pointer).

Why: bench.py is the driver's measuring instrument, and several of its legs (realistic fixture, host-fed windows, narrow store) were written in rounds when no
GPU call was accepted — nothing had ever executed them.  A Python error in one of them would cost the driver its JSON line.  This runs every line of bench.py's
control flow and JSON assembly on a tiny configuration; the TIMES it prints are host times of the emulator and mean nothing.  Test infrastructure only: bench.py
does not know about it (torch.cuda is patched from outside), the product has no path to it.

    python tests/emu/bench_on_emulator.py --batch 2 --log2-rows 15 --steps 2 --warmup 1 --no-cpu-baseline --with-narrow-store-mode
"""
import contextlib
import os
import runpy
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.environ.get("ZKGL_LIB") or os.path.join(ROOT, "tests", "emu", "_gen", "dev_O2", "libzkgl.so")
if not os.path.exists(LIB):
    LIB = os.path.join(ROOT, "tests", "emu", "_gen", "dev", "libzkgl.so")
if not os.path.exists(LIB):
    sys.exit("build the emulated device first: bash tests/emu/dev/build.sh")
os.environ["ZKGL_LIB"] = LIB

import torch  # noqa: E402

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
    """torch.device("cuda", i) -> the host: what the emulated device's memory is"""
    if a and (a[0] == "cuda" or (isinstance(a[0], str) and a[0].startswith("cuda"))):
        return _cpu
    return _real_device(*a, **k)


_STREAM = _Stream()
torch.device = _device
torch.cuda.is_available = lambda: True
torch.cuda.device_count = lambda: 1
torch.cuda.set_device = lambda d: None
torch.cuda.current_stream = lambda *a, **k: _STREAM
torch.cuda.Stream = _Stream
torch.cuda.Event = _Event
torch.cuda.stream = lambda s: contextlib.nullcontext()
torch.cuda.device = lambda d: contextlib.nullcontext()
torch.cuda.synchronize = lambda *a, **k: None
torch.cuda.empty_cache = lambda: None
torch.cuda.mem_get_info = lambda *a, **k: (48 << 30, 64 << 30)
torch.Tensor.pin_memory = lambda self, *a, **k: self

if __name__ == "__main__":
    sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
    sys.path.insert(0, ROOT)
    runpy.run_path(sys.argv[0], run_name="__main__")
