# GPU box, repo root, call A of round 6 (nothing experimental): the evidence set of the DEFAULT library at HEAD.
#  (a) full -m gpu suite, no -x  (b) the driver's bench command line + rocprof kernel trace + PMC passes (tools/profile_tag.sh)
#  (c) config timings C1/C3k/C3s/C4/C5
mkdir -p gpurun_out
export TAG=${TAG:-r6}
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > gpurun_out/${TAG}_gputest.log 2>&1; grep -n "passed\|failed\|rror" gpurun_out/${TAG}_gputest.log | tail -15
BENCH_ARGS="--gpus 1 --steps 20 --warmup 5" KT_STEPS=5 timeout 1500 bash tools/profile_tag.sh > gpurun_out/${TAG}_profile.log 2>&1; tail -8 gpurun_out/${TAG}_profile.log
python tools/pmc_json.py ${TAG} > gpurun_out/pmc_${TAG}.json 2> gpurun_out/pmc_json.err || tail -2 gpurun_out/pmc_json.err
timeout 900 python tests/config_timings.py 2>gpurun_out/${TAG}_config_timings.err | grep "^{" > gpurun_out/${TAG}_config_timings.jsonl; cut -c1-260 gpurun_out/${TAG}_config_timings.jsonl
