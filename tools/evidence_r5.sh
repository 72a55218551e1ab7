# GPU box, repo root, ONE gpurun call: the round-5 evidence set of the tree as it stands.
#  (a) the FULL -m gpu suite of the default library, no -x so every failure is listed
#  (b) the driver's bench command line -> r5_bench.json     (c) kernel trace + PMC of the current k_witness_loop -> r5_kernel_trace.md, pmc_r5_*.txt
#  (d) config timings C1/C3k/C3s/C4/C5 -> r5_config_timings.jsonl     (e) opt-in A/Bs from their own libraries (tools/variants_r5.sh first, in the container)
mkdir -p gpurun_out
export TAG=${TAG:-r5}
timeout 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/${TAG}_gputest.log 2>&1; grep -n "passed\|failed\|rror" gpurun_out/${TAG}_gputest.log | tail -15
BENCH_ARGS="--gpus 1 --steps 20 --warmup 5" KT_STEPS=5 timeout 1800 bash tools/profile_tag.sh > gpurun_out/${TAG}_profile.log 2>&1; tail -8 gpurun_out/${TAG}_profile.log
python tools/pmc_json.py ${TAG} > gpurun_out/pmc_${TAG}.json 2> gpurun_out/pmc_json.err || tail -2 gpurun_out/pmc_json.err
timeout 900 python tests/config_timings.py 2>gpurun_out/${TAG}_config_timings.err | grep "^{" > gpurun_out/${TAG}_config_timings.jsonl; cut -c1-260 gpurun_out/${TAG}_config_timings.jsonl
timeout 3000 bash tools/ab_r5.sh > gpurun_out/${TAG}_ab.log 2>&1; tail -40 gpurun_out/r5_ab.txt
