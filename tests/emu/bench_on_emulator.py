"""tests/emu/bench_on_emulator.py [bench.py arguments] — bench.py end to end WITHOUT a GPU: the emulated device (tests/emu/dev, ZKGL_LIB) behind the C ABI and
host tensors standing in for device tensors (the emulated device's memory IS host memory, a tensor's data_ptr() is a valid "device" pointer).

Why: bench.py is the driver's measuring instrument, and several of its legs (realistic fixture, host-fed windows, narrow store) were written in rounds when no
GPU call was accepted — nothing had ever executed them.  A Python error in one of them would cost the driver its JSON line.  This runs every line of bench.py's
control flow and JSON assembly on a tiny configuration; the TIMES it prints are host times of the emulator and mean nothing.  Test infrastructure only: bench.py
does not know about it (torch.cuda is patched from outside), the product has no path to it.

    python tests/emu/bench_on_emulator.py --batch 2 --log2-rows 15 --steps 2 --warmup 1 --no-cpu-baseline --with-narrow-store-mode
    EMU_DEVICES=2 python tests/emu/bench_on_emulator.py --gpus 2 --batch 2 --log2-rows 15 --steps 2 --warmup 1 --no-cpu-baseline --headline-only
        (bench.py launches its ranks itself; they get the same stand-ins through tests/emu/site/sitecustomize.py; EMU_DEVICES=N gives every rank its own
         "device", so the step's gather is the product's zk_cs_gather_commitments over the N-process stand-in of tests/emu/dev/rccl/rccl.h)
"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.environ.get("ZKGL_LIB") or os.path.join(ROOT, "tests", "emu", "_gen", "dev_O2", "libzkgl.so")
if not os.path.exists(LIB):
    LIB = os.path.join(ROOT, "tests", "emu", "_gen", "dev", "libzkgl.so")
if not os.path.exists(LIB):
    sys.exit("build the emulated device first: bash tests/emu/dev/build.sh")
os.environ["ZKGL_LIB"] = LIB

os.environ["ZKGL_EMU_TORCH"] = "1"            # ... and for the ranks bench.py starts itself (--gpus N): tests/emu/site/sitecustomize.py
os.environ["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "tests", "emu", "site")] + ([os.environ["PYTHONPATH"]] if os.environ.get("PYTHONPATH") else []))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
import torch_cuda_on_host  # noqa: E402

torch_cuda_on_host.apply()

if __name__ == "__main__":
    sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
    sys.path.insert(0, ROOT)
    runpy.run_path(sys.argv[0], run_name="__main__")
