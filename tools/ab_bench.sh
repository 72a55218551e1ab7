python -m pytest tests/test_gpu_main_vm.py tests/test_gpu_cs.py -x -q -m gpu 2>&1 | tail -4
for v in "default" "nogroups" "nt"; do
  case $v in
    default) envs="";;
    nogroups) envs="ZKGL_OP_GROUPS=0";;
    nt) envs="ZKGL_LIB=$PWD/era-zkevm_circuits_amd/libzkgl_nt.so";;
  esac
  env $envs python bench.py --steps 3 --batch 64 --no-cpu-baseline > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$v.json"))
print("$v", "step", round(d["ms_per_step"],2), "loop", round(d["roofline"]["avg_launch_ms"],2), "gates", round(d["roofline"]["other_kernels_ms"]["k_check_gates_loop"],2), "outer", round(d["roofline"]["other_kernels_ms"]["outer_post_and_checks_overlapped"],2), "seed", d["config"]["input_seeding_s"], d["config"]["commitments_equal_native_restatement"])
PY
done
