// C1: one-shot sum-all-reduce of the per-layer bf16 partials over peer-mapped buffers (xGMI), for the
// tensor-parallel decode path.
//
// Replaces the two `dist.all_reduce` per layer of the reference (Engine/SnapKV/model.py:336,455 and twins; NCCL
// there).  The messages are tiny and latency-bound -- [B*(gamma+1), dim] bf16 = 2 MiB for the 8B verify step,
// 256 KiB for a 1B draft step -- so a ring (2(N-1) hops over point-to-point xGMI links) pays ~N link latencies.
// One-shot: every rank publishes its partial in an IPC-mapped buffer, raises one flag per peer, and then reads all
// N partials directly and sums them itself: one hop.
//
// Determinism: every rank adds the N partials in rank order 0..N-1 with fp32 accumulation
// and one final rounding, so all ranks obtain bit-identical results (the replicated argmax / page tables need that).
//
// Protocol per call k (flag = k, kept in device memory so that a captured hipGraph replays correctly):
//   block b copies slice b of `in` into my data buffer [k & 1]; release-stores flag k into start[b][my_rank] of
//   every peer's signal area (system scope); acquire-spins until start[b][r] >= k for every r in my own signal
//   area; then sums slice b over the N data buffers [k & 1] and writes `out`.
// Two data buffers make a closing barrier unnecessary: a rank can only reach call k+2 (which overwrites buffer
// [k & 1]) after every peer entered call k+1, i.e. after every peer finished reading in call k (stream order).
// Every call launches the same kMaxBlocks blocks so that all per-block counters stay equal (see the launcher).
// Signals live in uncached (fine-grained) memory; data buffers are ordinary device memory -- the system-scope
// release / acquire pair performs the L2 write-back / invalidate the AMDGPU memory model prescribes.
// Spins are bounded (kSpinTimeoutTicks): on timeout the kernel records an error and returns instead of hanging the GPU.
#include "md_common.h"

namespace {

constexpr int kMaxRanks = 8;
constexpr int kMaxBlocks = 64;
constexpr int kThreads = 512;
constexpr unsigned long long kSpinTimeoutTicks = 200ull * 1000 * 1000;   // wall_clock64() ticks at 100 MHz: 2 s

struct Signal {
    uint32_t start[kMaxBlocks][kMaxRanks];   // written by peers (system-scope release), read by the owner
    uint32_t flag[kMaxBlocks];               // owner only: call counter per block
    uint32_t status;                         // owner only: 0 ok, 1 = a spin timed out
};

struct ArDev {
    bf16_t* data[kMaxRanks];   // peer data buffers (2 * max_bytes each), index = rank
    Signal* sig[kMaxRanks];
    int rank, world;
    size_t buf_elems;          // elements per half buffer
};

template <int NR>
__global__ __launch_bounds__(kThreads) void oneshot_ar_kernel(const ArDev c, const bf16_t* __restrict__ in,
                                                              bf16_t* __restrict__ out, size_t n_vec) {
    const int b = blockIdx.x, tid = threadIdx.x;
    __shared__ uint32_t s_flag;
    Signal* self = c.sig[c.rank];
    if (tid == 0) {
        const uint32_t f = self->flag[b] + 1;
        self->flag[b] = f;
        s_flag = f;
    }
    __syncthreads();
    const uint32_t flag = s_flag;
    const size_t half = (flag & 1u) ? c.buf_elems : 0;
    const size_t per = (n_vec + gridDim.x - 1) / gridDim.x;
    const size_t v0 = (size_t)b * per, v1 = v0 + per < n_vec ? v0 + per : n_vec;

    // phase 0: publish my slice
    u32x4* mine = reinterpret_cast<u32x4*>(c.data[c.rank] + half);
    const u32x4* src = reinterpret_cast<const u32x4*>(in);
    for (size_t i = v0 + tid; i < v1; i += kThreads) mine[i] = src[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");   // every wave: its copies are written back before the flag is raised
    __syncthreads();
    if (tid < NR) {
        __hip_atomic_store(&c.sig[tid]->start[b][c.rank], flag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned long long t0 = wall_clock64();
        while ((int32_t)(__hip_atomic_load(&self->start[b][tid], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - flag) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > kSpinTimeoutTicks) {
                self->status = 1;
                break;
            }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");   // every wave: no stale peer data from two calls ago

    // phase 1: sum slice b over the ranks, in rank order, fp32 accumulate, one rounding
    const u32x4* peer[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) peer[r] = reinterpret_cast<const u32x4*>(c.data[r] + half);
    u32x4* dst = reinterpret_cast<u32x4*>(out);
    for (size_t i = v0 + tid; i < v1; i += kThreads) {
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const u32x4 v = __builtin_nontemporal_load(peer[r] + i);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                acc[2 * w] += __uint_as_float(v[w] << 16);
                acc[2 * w + 1] += __uint_as_float(v[w] & 0xffff0000u);
            }
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f32_to_bf16(acc[e]);
        dst[i] = *reinterpret_cast<u32x4*>(&o);
    }
}

}  // namespace

struct md_ar_comm {
    ArDev dev;
    void* my_data;
    void* my_sig;
    void* opened[2 * kMaxRanks];
    int n_opened;
    size_t max_bytes;
};

extern "C" int md_ar_create(int rank, int world, size_t max_bytes, md_ar_comm** comm_out) {
    MD_CHECK_ARG(comm_out && world >= 1 && world <= kMaxRanks && rank >= 0 && rank < world,
                 "md_ar_create: need 1 <= world <= %d and 0 <= rank < world", kMaxRanks);
    MD_CHECK_ARG(max_bytes >= 16 && max_bytes % 16 == 0, "md_ar_create: max_bytes must be a positive multiple of 16");
    md_ar_comm* c = new md_ar_comm();
    c->n_opened = 0;
    c->max_bytes = max_bytes;
    if (hipMalloc(&c->my_data, 2 * max_bytes) != hipSuccess ||
        hipExtMallocWithFlags(&c->my_sig, sizeof(Signal), hipDeviceMallocUncached) != hipSuccess) {
        md_set_error("md_ar_create: device allocation failed: %s", hipGetErrorString(hipGetLastError()));
        delete c;
        return MD_ERR_WORKSPACE;
    }
    if (hipMemset(c->my_sig, 0, sizeof(Signal)) != hipSuccess || hipMemset(c->my_data, 0, 2 * max_bytes) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess) {
        md_set_error("md_ar_create: clearing the buffers failed: %s", hipGetErrorString(hipGetLastError()));
        (void)hipFree(c->my_data);
        (void)hipFree(c->my_sig);
        delete c;
        return MD_ERR_WORKSPACE;
    }
    for (int r = 0; r < kMaxRanks; ++r) {
        c->dev.data[r] = nullptr;
        c->dev.sig[r] = nullptr;
    }
    c->dev.rank = rank;
    c->dev.world = world;
    c->dev.buf_elems = max_bytes / 2;
    c->dev.data[rank] = (bf16_t*)c->my_data;
    c->dev.sig[rank] = (Signal*)c->my_sig;
    *comm_out = c;
    return MD_OK;
}

extern "C" int md_ar_get_handles(md_ar_comm* c, void* handles_host) {
    MD_CHECK_ARG(c && handles_host, "md_ar_get_handles: null argument");
    static_assert(sizeof(hipIpcMemHandle_t) == MD_AR_HANDLE_BYTES, "IPC handle size");
    hipIpcMemHandle_t* h = (hipIpcMemHandle_t*)handles_host;
    if (hipIpcGetMemHandle(&h[0], c->my_data) != hipSuccess || hipIpcGetMemHandle(&h[1], c->my_sig) != hipSuccess) {
        md_set_error("md_ar_get_handles: hipIpcGetMemHandle failed: %s (is HSA_ENABLE_IPC_MODE_LEGACY=0 set?)",
                     hipGetErrorString(hipGetLastError()));
        return MD_ERR_LAUNCH;
    }
    return MD_OK;
}

extern "C" int md_ar_open_peers(md_ar_comm* c, const void* all_handles_host) {
    MD_CHECK_ARG(c && all_handles_host, "md_ar_open_peers: null argument");
    const hipIpcMemHandle_t* h = (const hipIpcMemHandle_t*)all_handles_host;
    for (int r = 0; r < c->dev.world; ++r) {
        if (r == c->dev.rank) continue;
        void *d = nullptr, *s = nullptr;
        if (hipIpcOpenMemHandle(&d, h[2 * r], hipIpcMemLazyEnablePeerAccess) != hipSuccess ||
            hipIpcOpenMemHandle(&s, h[2 * r + 1], hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
            md_set_error("md_ar_open_peers: hipIpcOpenMemHandle(rank %d) failed: %s", r,
                         hipGetErrorString(hipGetLastError()));
            return MD_ERR_LAUNCH;
        }
        c->opened[c->n_opened++] = d;
        c->opened[c->n_opened++] = s;
        c->dev.data[r] = (bf16_t*)d;
        c->dev.sig[r] = (Signal*)s;
    }
    return MD_OK;
}

extern "C" int md_allreduce_oneshot(md_ar_comm* c, const void* in, void* out, size_t count, md_stream_t stream) {
    MD_CHECK_ARG(c && in && out, "md_allreduce_oneshot: null argument");
    MD_CHECK_ARG(count % 8 == 0 && count * 2 <= c->max_bytes,
                 "md_allreduce_oneshot: count must be a multiple of 8 bf16 and fit the registered buffer (%zu bytes)",
                 c->max_bytes);
    MD_CHECK_ARG(((uintptr_t)in | (uintptr_t)out) % 16 == 0, "md_allreduce_oneshot: in/out must be 16-byte aligned");
    for (int r = 0; r < c->dev.world; ++r)
        MD_CHECK_ARG(c->dev.data[r] && c->dev.sig[r], "md_allreduce_oneshot: peer %d not opened (md_ar_open_peers)", r);
    if (count == 0) return MD_OK;
    const size_t n_vec = count / 8;
    // Always the full grid, whatever the message size: every block then advances its call counter on every call,
    // so all blocks of call k agree on the data-buffer half (k & 1).  (With a size-dependent grid the per-block
    // counters drift apart when message sizes alternate, and a small call could overwrite a region of the half a
    // slower peer is still reading for the previous, larger call.)  Idle blocks only run the flag handshake.
    const int blocks = kMaxBlocks;
    hipStream_t st = (hipStream_t)stream;
#define MD_AR_LAUNCH(N)                                                                                       \
    case N:                                                                                                   \
        hipLaunchKernelGGL((oneshot_ar_kernel<N>), dim3(blocks), dim3(kThreads), 0, st, c->dev, (const bf16_t*)in, \
                           (bf16_t*)out, n_vec);                                                              \
        break;
    switch (c->dev.world) {
        MD_AR_LAUNCH(1)
        MD_AR_LAUNCH(2)
        MD_AR_LAUNCH(3)
        MD_AR_LAUNCH(4)
        MD_AR_LAUNCH(5)
        MD_AR_LAUNCH(6)
        MD_AR_LAUNCH(7)
        MD_AR_LAUNCH(8)
    }
#undef MD_AR_LAUNCH
    MD_CHECK_LAUNCH("md_allreduce_oneshot");
    return MD_OK;
}

extern "C" int md_ar_status(md_ar_comm* c, int* status_host) {
    MD_CHECK_ARG(c && status_host, "md_ar_status: null argument");
    uint32_t s = 0;
    if (hipMemcpy(&s, &((Signal*)c->my_sig)->status, sizeof(s), hipMemcpyDeviceToHost) != hipSuccess) {
        md_set_error("md_ar_status: copy failed: %s", hipGetErrorString(hipGetLastError()));
        return MD_ERR_LAUNCH;
    }
    *status_host = (int)s;
    return MD_OK;
}

extern "C" int md_ar_destroy(md_ar_comm* c) {
    if (!c) return MD_OK;
    (void)hipDeviceSynchronize();
    for (int i = 0; i < c->n_opened; ++i) (void)hipIpcCloseMemHandle(c->opened[i]);
    (void)hipFree(c->my_data);
    (void)hipFree(c->my_sig);
    delete c;
    return MD_OK;
}
