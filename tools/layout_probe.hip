// tools/layout_probe.hip — does the PHYSICAL layout of the variable store explain the 7 % run-to-run modes of k_witness_loop?
// The store pattern of the witness interpreter without arithmetic: every wavefront writes its values in ascending slot order, 512 B per
// value, all wavefronts at about the same slot at the same time.  Layouts:
//   G = 1   (today)  wave-tiled: store[((tile * n_slots + slot) * 64 + lane)] — a wavefront owns one contiguous 9 MB tile
//   G > 1            G tiles interleaved per slot: store[(((tile / G) * n_slots + slot) * G + tile % G) * 64 + lane] — the G wavefronts of a
//                    group write G * 512 contiguous bytes per slot
// and optionally a read of the value written `back` slots earlier (operand re-read).  Prints ms / GB/s per layout, several repetitions,
// for the default and for a physically contiguous allocation.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/layout_probe.hip -o /tmp/layout_probe && /tmp/layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
template <bool READ>
__global__ __launch_bounds__(256) void k(uint64_t* cells, uint32_t n_slots, uint32_t n_lanes, uint32_t G, uint32_t back) {
    const uint32_t lane = blockIdx.x * 256 + threadIdx.x;
    if (lane >= n_lanes) return;
    const uint32_t tile = lane >> 6;
    uint64_t* base = cells + ((size_t)(tile / G) * n_slots * G + tile % G) * 64 + (lane & 63);
    const size_t stride = (size_t)G * 64;
    uint64_t v = lane;
    for (uint32_t s = 0; s < n_slots; ++s) {
        if (READ && s >= back) v += __builtin_nontemporal_load(base + (size_t)(s - back) * stride);
        base[(size_t)s * stride] = v + s;
    }
}
int main(int argc, char** argv) {
    const uint32_t n_lanes = argc > 1 ? atoi(argv[1]) : 384 * 2384, n_slots = 17548;
    const size_t bytes = (size_t)((n_lanes + 63) / 64 + 64) * n_slots * 64 * 8;
    for (int contiguous = 0; contiguous < 2; ++contiguous) {
        uint64_t* cells = nullptr;
        hipError_t e = contiguous ? hipExtMallocWithFlags((void**)&cells, bytes, hipDeviceMallocContiguous) : hipMalloc((void**)&cells, bytes);
        if (e != hipSuccess) { printf("{\"alloc\": \"%s\", \"error\": \"%s\"}\n", contiguous ? "contiguous" : "default", hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        for (int read = 0; read < 2; ++read)
            for (uint32_t G : {1u, 4u, 16u, 64u}) {
                float best = 1e9f, worst = 0;
                for (int rep = 0; rep < 4; ++rep) {
                    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                    hipEventRecord(e0);
                    if (read) k<true><<<(n_lanes + 255) / 256, 256>>>(cells, n_slots, n_lanes, G, 40);
                    else k<false><<<(n_lanes + 255) / 256, 256>>>(cells, n_slots, n_lanes, G, 0);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    if (rep) { best = ms < best ? ms : best; worst = ms > worst ? ms : worst; }
                    hipEventDestroy(e0); hipEventDestroy(e1);
                }
                const double gb = (double)n_lanes * n_slots * 8 * (read ? 2 : 1) / 1e9;
                printf("{\"alloc\": \"%s\", \"pattern\": \"%s\", \"tiles_interleaved\": %u, \"ms_best\": %.2f, \"ms_worst\": %.2f, \"GBps_best\": %.0f}\n",
                       contiguous ? "contiguous" : "default", read ? "write + read 40 slots back" : "write only", G, best, worst, gb / (best * 1e-3));
                fflush(stdout);
            }
        hipFree(cells);
    }
    return 0;
}
