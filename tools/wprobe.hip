// tools/wprobe.hip — write-pattern probe: the store pattern of the witness interpreter without any arithmetic.
// Every wave owns one tile [n_cells][64 lanes] of u64 and writes `n_stores` 512-byte rows of it in a given cell order:
//   mode 0: ascending cells          mode 1: pseudo-random cells         mode 2: `streams` interleaved ascending streams
// Build: hipcc --offload-arch=gfx950 -O3 tools/wprobe.hip -o gpurun_out/wprobe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
__global__ __launch_bounds__(256) void k(uint64_t* cells, uint64_t n_cells, const uint32_t* order, uint32_t n_stores, uint32_t n_lanes) {
    uint32_t lane = blockIdx.x * 256 + threadIdx.x;
    if (lane >= n_lanes) return;
    uint64_t* t = cells + (size_t)(lane >> 6) * n_cells * 64 + (lane & 63);
    uint64_t v = lane;
    for (uint32_t i = 0; i < n_stores; ++i) {
        uint32_t c = __builtin_amdgcn_readfirstlane((int)order[i]);
        t[(size_t)c << 6] = v + i;
    }
}
int main(int argc, char** argv) {
    const uint32_t n_lanes = argc > 1 ? atoi(argv[1]) : 521710, n_cells = 47724, n_stores = 34324;
    uint64_t* cells; uint32_t* d_order;
    size_t bytes = (size_t)((n_lanes + 63) / 64) * n_cells * 64 * 8;
    if (hipMalloc(&cells, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&d_order, n_stores * 4);
    for (int mode = 0; mode < 4; ++mode) {
        std::vector<uint32_t> order(n_stores);
        uint32_t streams = mode == 2 ? 8 : 64;
        for (uint32_t i = 0; i < n_stores; ++i) {
            if (mode == 0) order[i] = i;
            else if (mode == 1) order[i] = (uint32_t)(((uint64_t)i * 2654435761u) % n_cells);
            else { uint32_t s = i % streams, j = i / streams; order[i] = s * (n_cells / streams) + j; }
        }
        hipMemcpy(d_order, order.data(), n_stores * 4, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        k<<<(n_lanes + 255) / 256, 256>>>(cells, n_cells, d_order, n_stores, n_lanes);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<<<(n_lanes + 255) / 256, 256>>>(cells, n_cells, d_order, n_stores, n_lanes);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double gb = (double)n_lanes * n_stores * 8 / 1e9;
        printf("{\"mode\": \"%s\", \"lanes\": %u, \"stores_per_lane\": %u, \"ms\": %.2f, \"GBps\": %.0f}\n",
               mode == 0 ? "ascending" : mode == 1 ? "pseudo-random" : mode == 2 ? "8 streams" : "64 streams", n_lanes, n_stores, ms, gb / (ms * 1e-3));
    }
    return 0;
}
