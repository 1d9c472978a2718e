// Does the memory-side cache (MALL / Infinity Cache, 256 MB) hold a weight set between two passes, and how much faster is a
// pass that hits it?  (round 4: a 1B draft step runs at a third of its HBM floor -- HBM idles under every kernel's ramp and
// tail; if a warm pass is much faster, a side-stream prefetch of the NEXT linear's weights would use the idle time.)
// Each of 256 workgroups (16 waves) streams its contiguous share of an S-byte buffer, 8 loads in flight per wave.
//   cold: after 2 GiB of other data went through      warm: immediately after a pass over the same S bytes
//   pre:  cold, but a 32-workgroup "toucher" kernel (one 64-B line per 128 B? no: every line) ran first
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/mall_probe.hip -o /tmp/mall_probe ; run: /tmp/mall_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(1024) void stream(const u32x4* __restrict__ buf, size_t vec_per_wg, unsigned int* sink) {
    const u32x4* src = buf + (size_t)blockIdx.x * vec_per_wg;
    const int tid = threadIdx.x, nthr = blockDim.x;
    unsigned int acc = 0;
    u32x4 r[DEPTH];
    size_t i = tid;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) r[d] = src[i + (size_t)d * nthr];
    for (i += (size_t)DEPTH * nthr; i + (size_t)DEPTH * nthr <= vec_per_wg; i += (size_t)DEPTH * nthr) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            acc += r[d][0] ^ r[d][3];
            r[d] = src[i + (size_t)d * nthr];
        }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc += r[d][1];
    if (acc == 0x12345678u) sink[0] = acc;
}

// one dword per 128-byte line: pulls the lines through the memory-side cache with little CU traffic
__global__ void touch(const unsigned int* __restrict__ buf, size_t lines, unsigned int* sink) {
    unsigned int acc = 0;
    for (size_t l = (size_t)blockIdx.x * blockDim.x + threadIdx.x; l < lines; l += (size_t)gridDim.x * blockDim.x)
        acc += __builtin_nontemporal_load(buf + l * 32);
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const size_t big_bytes = (size_t)2 << 30, max_s = (size_t)1 << 30;
    void *big, *buf;
    unsigned int* sink;
    (void)hipMalloc(&big, big_bytes);
    (void)hipMalloc(&buf, max_s);
    (void)hipMalloc(&sink, 4);
    (void)hipMemset(big, 2, big_bytes);
    (void)hipMemset(buf, 1, max_s);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    auto timed = [&](auto&& f) {
        (void)hipEventRecord(e0);
        f();
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        return ms;
    };
    auto run = [&](const void* p, size_t bytes, int nw) {
        hipLaunchKernelGGL(stream<8>, dim3(256), dim3(64 * nw), 0, 0, (const u32x4*)p, bytes / 16 / 256, sink);
    };
    auto flush = [&]() { run(big, big_bytes, 16); };
    for (int nw : {8, 16})
        for (size_t mb : {16, 32, 64, 96, 128, 192, 256, 384, 512, 1024}) {
            const size_t s = mb << 20;
            float cold = 1e9, warm = 1e9, pre = 1e9, tch = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                flush();
                float t = timed([&] { run(buf, s, nw); });
                cold = t < cold ? t : cold;
                t = timed([&] { run(buf, s, nw); });
                warm = t < warm ? t : warm;
                flush();
                t = timed([&] { hipLaunchKernelGGL(touch, dim3(64), dim3(256), 0, 0, (const unsigned int*)buf, s / 128, sink); });
                tch = t < tch ? t : tch;
                t = timed([&] { run(buf, s, nw); });
                pre = t < pre ? t : pre;
            }
            printf("waves/CU %2d  S = %4zu MB: cold %7.1f us %5.0f GB/s | warm %7.1f us %5.0f GB/s | touch(64 wgs) %7.1f us, then %7.1f us %5.0f GB/s\n",
                   nw, mb, cold * 1e3, s / cold / 1e6, warm * 1e3, s / warm / 1e6, tch * 1e3, pre * 1e3, s / pre / 1e6);
        }
    return 0;
}
