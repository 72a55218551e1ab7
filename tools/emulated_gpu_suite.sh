#!/bin/bash
# tools/emulated_gpu_suite.sh [pytest args] — the -m gpu suite on the EMULATED DEVICE (tests/emu/README.md): the product's device source compiled for
# the host, work-items as fibers, wavefront operations as rendezvous.  No GPU involved, no performance meaning; ~30 min on 8 cores.
# Left out: test_linear_hasher_gpu (10.7 M cells per lane).  (The full-size tests — 2^16 .. 2^22 rows — were left out until round 6; they take six minutes.)  Statistics of the run: $OUT/stats.*
set -uo pipefail
cd "$(dirname "$0")/.."
EMU_OPT=-O2 bash tests/emu/dev/build.sh || exit 1
OUT=${OUT:-/tmp/emulated_gpu_suite}; mkdir -p $OUT; rm -f $OUT/stats.*
# (ZKGL_EMU_TORCH + tests/emu/site on PYTHONPATH: the tests whose device memory is a torch tensor get host tensors through the torch.cuda stand-ins, tests/emu/torch_cuda_on_host.py)
ZKGL_EMU_TORCH=1 PYTHONPATH=$PWD/tests/emu/site${PYTHONPATH:+:$PYTHONPATH} EMU_STATS=$OUT/stats ZKGL_LIB=$PWD/tests/emu/_gen/dev_O2/libzkgl.so python -m pytest tests -m gpu -q -n ${JOBS:-6} --durations=30 -p no:cacheprovider \
  --deselect tests/test_gpu_cs.py::test_linear_hasher_gpu --timeout 3000 "$@" 2>&1 | tee $OUT/pytest.log
echo "divergent wavefront-operation sites over the run (0 expected):"; cat $OUT/stats.* 2>/dev/null | grep -c "divergent x" || true
