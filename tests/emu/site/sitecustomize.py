"""Picked up by every Python process whose PYTHONPATH holds this directory (tests/emu/bench_on_emulator.py puts it there): the ranks bench.py starts through
torch.distributed.run get the same host stand-ins for torch.cuda as the launcher.  Does nothing unless ZKGL_EMU_TORCH=1."""
import os
import sys

if os.environ.get("ZKGL_EMU_TORCH") == "1":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch_cuda_on_host

    torch_cuda_on_host.apply()
