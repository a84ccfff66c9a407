// Write-only bandwidth of K2's store pattern (standalone, GPU box):  hipcc --offload-arch=gfx950 -O3 -o /tmp/store_probe tools/store_probe.hip && /tmp/store_probe
// 100 rows ("slots") of 512 KiB per frame, 40 frames; every workgroup advances `seg` bytes per row and step, one 16-byte store per lane.
// Measured: 128-byte segments (K2's) 5.4 TB/s, 512 B 5.7, 8 KiB 6.0: the scattered 128-byte row segments are not what holds K2 at 4.0.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void store_rows(char* dst, int streams, long long row_bytes, int seg, int steps) {
    const int t = blockIdx.y, c = blockIdx.x;
    char* base = dst + (long long)t * streams * row_bytes + (long long)c * steps * seg;
    const int lanes_per_seg = seg / 16, segs_per_instr = 512 / lanes_per_seg;
    const int tid = threadIdx.x, sub = tid / lanes_per_seg, off = (tid % lanes_per_seg) * 16;
    const u32x4 v = {1u, 2u, 3u, (unsigned)tid};
    for (int it = 0; it < steps; ++it)
        for (int s0 = 0; s0 < streams; s0 += segs_per_instr) {
            const int s = s0 + sub;
            if (s < streams) *reinterpret_cast<u32x4*>(base + (long long)s * row_bytes + (long long)it * seg + off) = v;
        }
}
int main() {
    const int T = 40, streams = 100, HW = 131072, chunks = 64;
    const long long row_bytes = (long long)HW * 4;
    const size_t total = (size_t)T * streams * row_bytes;
    char* d;
    if (hipMalloc(&d, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return 1;
    const int segs[] = {128, 256, 512, 1024, 2048, 8192};
    for (int seg : segs) {
        const int steps = (int)(row_bytes / chunks / seg);
        float ms = 0.f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(a);
            hipLaunchKernelGGL(store_rows, dim3(chunks, T), dim3(512), 0, 0, d, streams, row_bytes, seg, steps);
            (void)hipEventRecord(b);
            (void)hipEventSynchronize(b);
            (void)hipEventElapsedTime(&ms, a, b);
        }
        printf("segment %5d B per row and step: %.2f ms  %.0f GB/s written\n", seg, ms, total / ms / 1e6);
    }
    return hipFree(d) == hipSuccess ? 0 : 1;
}
