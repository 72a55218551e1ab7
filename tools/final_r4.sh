# GPU box, repo root, one gpurun call when little time is left: (1) the evidence set of the product as it stands (bench, kernel trace, PMC),
# (2) the GPU suite without the slow differential fuzz, (3) the new no-seeding tests, (4) the opt-in K8 / planes A/B.  Everything bounded.
mkdir -p gpurun_out
timeout 1500 bash tools/profile_r4.sh > gpurun_out/final_profile.log 2>&1; tail -5 gpurun_out/final_profile.log
python tools/pmc_json.py r4_final > gpurun_out/pmc_r4_final.json 2> gpurun_out/pmc_json.err || tail -2 gpurun_out/pmc_json.err
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_fused_differential.py > gpurun_out/final_suite.log 2>&1; grep -n "passed\|failed\|rror" gpurun_out/final_suite.log | tail -5
F_CIRCUITS=demux,decommit,unpacker timeout 300 python tools/f_timings.py 18 32 2>&1 | grep circuit > gpurun_out/f_timings_final.jsonl
CONFIGS=C3k,C3s,C5 timeout 600 python tests/config_timings.py 2>/dev/null | grep "^{" > gpurun_out/config_timings_final.jsonl; cut -c1-220 gpurun_out/config_timings_final.jsonl
timeout 1500 bash tools/k8_ab_r4.sh > gpurun_out/k8_ab.txt 2>&1; tail -12 gpurun_out/k8_ab.txt
