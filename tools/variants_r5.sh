#!/bin/bash
# Container (no GPU), repo root: the A/B libraries of round 5 — every opt-in device path lives in its OWN library, never in libzkgl.so
# (the built .so files travel to the GPU box with the snapshot).  Then on the box: bash tools/ab_r5.sh
#   libzkgl_k8.so      -DZKGL_BYTEBUF_KERNEL -DZKGL_STRAND_PLANES_KERNEL   (ZK_OP_BYTEBUF_FILL on the device; flag planes in the strand form)
#   libzkgl_chains.so  -DZKGL_SELECT_CHAINS_KERNEL                        (mux-chain ops)
#   libzkgl_binv.so    -DZKGL_BATCH_INV                                   (Montgomery-batched zero-check inversions)
#   libzkgl_p2m.so     -DZKGL_P2_MERGE                                    (the gated witness-only permutations of a dependency level under one header)
#   libzkgl_p2m_binv.so  both VALU levers of the loop kernel together
#   libzkgl_sha4.so    -DZKGL_SHA4_KERNEL                                 (the reference's 4-bit-chunk SHA-256 compression as a macro-op)
set -e
cd "$(dirname "$0")/../era-zkevm_circuits_amd"
build() { ZKGL_DEFS="$2" ZKGL_OUT=../libzkgl_$1.so ZKGL_BUILD_DIR=../build/var/$1 ./build.sh 2>&1 | grep -E "error|built" || true; }
build k8 "-DZKGL_BYTEBUF_KERNEL -DZKGL_STRAND_PLANES_KERNEL" &
build chains "-DZKGL_SELECT_CHAINS_KERNEL" &
build binv "-DZKGL_BATCH_INV" &
build sha4 "-DZKGL_SHA4_KERNEL" &
build p2m "-DZKGL_P2_MERGE" &
build p2m_binv "-DZKGL_P2_MERGE -DZKGL_BATCH_INV" &
wait
ls -la libzkgl*.so
