// What can one CU ingest?  (round 4: every M >= 32 GEMM of this repo sits at ~50 GB/s per CU once an activation slab is
// re-read through L2 beside a weight stream -- DESIGN.md 3.3.)  Each workgroup (NW waves) streams `bytes_per_wg` from a
// buffer with 16-byte loads, DEPTH loads in flight per wave, and adds the dwords up (so nothing is optimised away).
//   mode 0: every workgroup reads the SAME 2 MiB (L2-resident, like x)        mode 1: disjoint ranges (HBM, like W)
//   mode 2: half of the waves read the shared 2 MiB, the other half disjoint ranges (the GEMM's mix)
//   mode 3 / 4: three quarters / seven eighths of the waves read the shared buffer (x : W = 3 : 1 / 7 : 1)
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/ingest_probe.hip -o /tmp/ingest_probe ; run: /tmp/ingest_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ void probe(const u32x4* __restrict__ shared_buf, const u32x4* __restrict__ big, size_t vec_per_wg, size_t shared_vecs,
                      int mode, unsigned int* sink) {
    const int tid = threadIdx.x, nthr = blockDim.x, wave = tid >> 6;
    const bool from_shared = mode == 0 || (mode == 2 && (wave & 1) == 0) || (mode == 3 && (wave & 3) != 0) ||
                             (mode == 4 && (wave & 7) != 0);
    const u32x4* src = from_shared ? shared_buf : big + (size_t)blockIdx.x * vec_per_wg;
    const size_t n = from_shared ? shared_vecs : vec_per_wg;
    unsigned int acc = 0;
    u32x4 r[DEPTH];
    size_t i = tid;
    const size_t total = vec_per_wg;              // every thread group moves vec_per_wg vectors in total
    size_t done = 0;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) r[d] = src[(i + (size_t)d * nthr) % n];
    for (; done + (size_t)DEPTH * nthr <= total; done += (size_t)DEPTH * nthr) {
        i += (size_t)DEPTH * nthr;
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            acc += r[d][0] ^ r[d][3];
            r[d] = src[(i + (size_t)d * nthr) % n];
        }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) acc += r[d][1];
    if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
    const size_t shared_bytes = 2u << 20, big_bytes = (size_t)3 << 30;
    void *sh, *big;
    unsigned int* sink;
    (void)hipMalloc(&sh, shared_bytes);
    (void)hipMalloc(&big, big_bytes);
    (void)hipMalloc(&sink, 4);
    (void)hipMemset(sh, 1, shared_bytes);
    (void)hipMemset(big, 2, big_bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int grid = 256;
    for (int mode = 0; mode < 5; ++mode)
        for (int nw : {4, 8, 16})
            for (int depth : {4, 8}) {
                const size_t per_wg = (size_t)8 << 20;          // 8 MiB per workgroup
                const size_t vec_per_wg = per_wg / 16;
                float best = 1e9;
                for (int rep = 0; rep < 3; ++rep) {
                    (void)hipEventRecord(e0);
                    if (depth == 4)
                        hipLaunchKernelGGL(probe<4>, dim3(grid), dim3(64 * nw), 0, 0, (const u32x4*)sh, (const u32x4*)big, vec_per_wg,
                                           shared_bytes / 16, mode, sink);
                    else
                        hipLaunchKernelGGL(probe<8>, dim3(grid), dim3(64 * nw), 0, 0, (const u32x4*)sh, (const u32x4*)big, vec_per_wg,
                                           shared_bytes / 16, mode, sink);
                    (void)hipEventRecord(e1);
                    (void)hipEventSynchronize(e1);
                    float ms;
                    (void)hipEventElapsedTime(&ms, e0, e1);
                    if (ms < best) best = ms;
                }
                const double gbs = (double)per_wg * grid / best / 1e6;
                printf("mode %d (%s) waves/CU %2d depth %d: %.3f ms  %.0f GB/s chip = %.1f GB/s per CU\n", mode,
                       mode == 0 ? "shared 2 MiB, L2" : (mode == 1 ? "disjoint, HBM" : (mode == 2 ? "1 : 1 shared : HBM" : (mode == 3 ? "3 : 1 shared : HBM" : "7 : 1 shared : HBM"))), nw, depth, best, gbs,
                       gbs / grid);
            }
    return 0;
}
