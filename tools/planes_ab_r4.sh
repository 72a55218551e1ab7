# A/B of the SELECT flag planes on ONE box (fresh processes, alternating): k_witness_loop avg ms, shader clock, value, commitment checksum
run() { env "$@" timeout 250 python bench.py --headline-only --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$*', round(r['avg_launch_ms'],2), round(r['shader_clock_mhz']), d['value'], d['commitment_checksum'])"; }
run ZKGL_FLAG_PLANES=0
run ZKGL_FLAG_PLANES=1
run ZKGL_FLAG_PLANES=0
run ZKGL_FLAG_PLANES=1
