// tools/rprobe.hip — READ-counter calibration probe.  FETCH_SIZE of rocprofv3 is calibrated in MI355X_MICROARCH.md for 16 B/lane
// loads (gfx950 counts half: x2); k_witness_loop issues buffer_load_dwordx2 (8 B/lane, 512 B per wavefront and value) and, over the narrow store,
// buffer_load_ubyte (64 B per wavefront and value).  This probe reads a KNOWN number of bytes in exactly those patterns from a buffer far larger than
// the caches (every byte once, wave-tiled like the store), so that  factor = known_bytes / (FETCH_SIZE x 1024)  can be put into tools/pmc_json.py.
//   mode 0: buffer_load_dwordx2, one 512 B row per wavefront per step      mode 1: buffer_load_dwordx4 (the guide's calibration pattern)
//   mode 2: buffer_load_ubyte, one 64 B row per wavefront per step        mode 3: buffer_load_dwordx2 with every row read TWICE in a row (L2 hits: not HBM)
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/rprobe.hip -o gpurun_out/rprobe && gpurun_out/rprobe
//   counters:  tools/pmc_pass.sh rprobe FETCH_SIZE   with PMC_CMD=gpurun_out/rprobe   (kernels k_read<0..3> appear by name)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
template <int MODE>
__global__ __launch_bounds__(256) void k_read(const uint8_t* base, uint32_t rows_per_wave, uint64_t* sink) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6, l = threadIdx.x & 63;
    constexpr uint32_t ROW = MODE == 1 ? 1024 : MODE == 2 ? 64 : 512;   // bytes of one row of a wavefront
    const uint8_t* tile = base + (size_t)wave * rows_per_wave * ROW;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(tile), 0, -1, 0x00020000);
    uint64_t acc = 0;
    for (uint32_t r = 0; r < rows_per_wave; ++r) {
        const uint32_t so = __builtin_amdgcn_readfirstlane(r * ROW);
        if constexpr (MODE == 0 || MODE == 3) {
            u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, l * 8, so, 0);
            acc += v.x ^ v.y;
            if constexpr (MODE == 3) { u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, l * 8, so, 0); acc += w.x; }
        } else if constexpr (MODE == 1) {
            u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, l * 16, so, 0);
            acc += v.x ^ v.y ^ v.z ^ v.w;
        } else {
            acc += __builtin_amdgcn_raw_buffer_load_b8(rsrc, l, so, 0);
        }
    }
    if (acc == 0x1234567812345678ull) sink[0] = acc;   // keep the loads
}
template <int MODE>
static void run(const uint8_t* buf, size_t bytes, uint64_t* sink, const char* name) {
    const uint32_t row = MODE == 1 ? 1024 : MODE == 2 ? 64 : 512;
    const uint32_t waves = 256 * 8 * 4;                       // 8 192 wavefronts: every SIMD busy
    const uint32_t rows = (uint32_t)(bytes / ((size_t)waves * row));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_read<MODE><<<waves / 4, 256>>>(buf, rows, sink); hipDeviceSynchronize();
    hipEventRecord(e0);
    k_read<MODE><<<waves / 4, 256>>>(buf, rows, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double known = (double)waves * rows * row;
    printf("{\"kernel\": \"k_read<%d>\", \"pattern\": \"%s\", \"known_bytes_per_launch\": %.0f, \"ms\": %.3f, \"GBps\": %.0f}\n", MODE, name, known, ms, known / (ms * 1e-3) / 1e9);
}
int main(int argc, char** argv) {
    const size_t bytes = (size_t)(argc > 1 ? atoll(argv[1]) : 8) << 30;    // 8 GiB by default: >> 256 MiB of MALL + L2
    uint8_t* buf; uint64_t* sink;
    if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&sink, 8);
    hipMemset(buf, 1, bytes);
    run<0>(buf, bytes, sink, "buffer_load_dwordx2, 512 B per wavefront and row");
    run<1>(buf, bytes, sink, "buffer_load_dwordx4, 1024 B per wavefront and row");
    run<2>(buf, bytes / 8, sink, "buffer_load_ubyte, 64 B per wavefront and row");
    run<3>(buf, bytes, sink, "buffer_load_dwordx2 twice per row (second read hits)");
    return 0;
}
