// Direct all-reduce over xGMI peer mappings: one-shot for small buffers, two-shot for the 1-4 MB flat gradient
// buffers, optionally with the optimizer update fused into the epilogue.
//
// Why not a ring: an MI355X node is fully connected (7 xGMI links per GPU, ~153 GB/s each); a ring drives one
// outgoing link per step and takes 2(W-1) hops, and these messages are latency-bound anyway.  Here
// every rank owns 1/W of the buffer:
//     phase 1  (reduce-scatter by PULL): rank r reads its slice from all W staging buffers -- 7 links in parallel --
//              and sums them in rank order (fixed order: every rank later holds the same bits);
//     phase 2  (all-gather by PUSH): rank r writes the reduced slice into all W result buffers -- 7 links again;
// two hops in total.  Buffers <= `oneshot_bytes` skip the second hop: every rank pulls and reduces everything.
// The epilogue either writes sum * scale to the caller's tensor or applies the fused multi-tensor optimizer update
// (same device function as optim_kernel, csrc/hip/optim_update.h) straight from the result buffer.
//
// Memory: each rank owns ONE allocation { flags | stage[2] | result[2] } made with hipExtMallocWithFlags
// (uncached: remote and local accesses go to memory, so no cross-die cache can hold a stale line), exported with
// hipIpcGetMemHandle and mapped by every peer with hipIpcOpenMemHandle.  Stage and result are double-buffered by the
// parity of the call number, which removes any trailing barrier: when a rank enters call k+2 and overwrites the
// buffers of call k, every peer has already signalled in call k+1, i.e. finished call k.
//
// Synchronisation is per BLOCK, not per grid: block b of every rank works on the same share of every slice, so it
// only ever waits for block b of its peers (flags ready[src][b] and done[src][b], written remotely with
// system-scope release stores, polled locally).  No grid barrier, no co-residency requirement inside a rank, and
// the call number lives in device memory (advanced by the last block of a call) so a captured hipGraph replays correctly.
// Every spin is bounded by a wall-clock timeout that raises a sticky status word instead of hanging the device.
//
// This replaces the exchange sites /root/reference/DDL_Ray3.py:198,218 (pickled models and prediction matrices
// through the Ray object store) and the per-step gradient exchange the north star adds; RCCL stays the fallback,
// selected by a start-up comparison against it (parallel/p2p.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <string>

#include "../hip/optim_update.h"

#define P2P_API extern "C" __attribute__((visibility("default")))
#define P2P_MAX_WORLD 16
#define P2P_MAX_BLOCKS 512
#define P2P_THREADS 256

namespace {
thread_local std::string g_perr;

struct Flags {
    int ready[P2P_MAX_WORLD][P2P_MAX_BLOCKS];      // ready[src][b] = call number whose staging share is visible
    int done[P2P_MAX_WORLD][P2P_MAX_BLOCKS];       // done[src][b]  = call number whose result share has been pushed
};

struct P2PComm {
    int world = 1, rank = 0, device = 0;
    size_t cap4 = 0;                   // capacity of one stage / result buffer in float4
    size_t bytes = 0;
    char* base = nullptr;              // this rank's allocation
    char* peer[P2P_MAX_WORLD] = {};    // mapped allocations (peer[rank] == base)
    hipIpcMemHandle_t handle;
    int* counters = nullptr;           // [0] call number, [1] finished-block ticket (device, local)
    int* status = nullptr;             // sticky error word (device, local): 1 = a wait timed out
    int max_blocks = 128;             // measured on one MI355X (W processes sharing it): 256 blocks are slower than 128
    size_t oneshot_bytes = 256 << 10;
    long long timeout_ticks = 400000000LL;      // wall_clock64 runs at 100 MHz: 4 s
};

struct KArgs {
    Flags* flags[P2P_MAX_WORLD];
    float4* stage[P2P_MAX_WORLD];      // base of stage[0]; stage[1] = + cap4
    float4* result[P2P_MAX_WORLD];
    size_t cap4;
    int world, rank;
    int* counters;
    int* status;
    long long timeout_ticks;
    const float4* in;
    float4* out;
    long n4;
    float scale;
    int oneshot;
    // fused optimizer epilogue (FUSED kernels only)
    float4 *P, *Mo, *Vo;
    uint2* Pc;
    const int* step;
    int kind;
    float lr, b1, b2, eps, rho;
};

__device__ __forceinline__ void st_sys(int* p, int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ int ld_sys(const int* p) {
    return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Wait until flag[src] >= epoch for every src < world (lanes 0..world-1 of wave 0 poll one source each).
// Returns false on timeout (and raises *status) or when the communicator is already marked broken.
__device__ __forceinline__ bool wait_all(const int* flag_row0, int stride, int world, int epoch, int* status,
                                         long long timeout, bool dead) {
    __shared__ int s_ok;
    if (threadIdx.x == 0) s_ok = 1;
    __syncthreads();
    if (!dead && threadIdx.x < world) {
        const int* f = flag_row0 + (size_t)threadIdx.x * stride;
        const long long t0 = wall_clock64();
        while (ld_sys(f) < epoch) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > timeout) {
                atomicExch(status, 1);
                s_ok = 0;
                break;
            }
        }
    }
    __syncthreads();
    __atomic_thread_fence(__ATOMIC_ACQUIRE);           // system scope: peers' data written before their flags
    return s_ok != 0 && !dead;
}

// Share b of slice s: the float4 indices  s*q + b*T + t + k*NB*T  below the slice end.  Staging, reduction and
// epilogue all walk shares with this one pattern, so block b only ever touches bytes that block b of a peer wrote
// (and flagged).  One-shot mode is the same walk with a single slice.
// A walk hands its body U indices at a time (clamped to the last valid one, with a validity mask): the body issues
// all its loads before it consumes any, so U (x W for the reduction) 16-byte loads are in flight per lane -- these
// buffers are uncached, every load is a round trip to memory or across a link.
#define P2P_U 4
template <typename Body>
__device__ __forceinline__ void walk_share(long lo, long hi, int b, int t, long stride, Body body) {
    for (long i = lo + (long)b * P2P_THREADS + t; i < hi; i += P2P_U * stride) {
        long idx[P2P_U];
        bool ok[P2P_U];
#pragma unroll
        for (int u = 0; u < P2P_U; ++u) {
            const long j = i + u * stride;
            ok[u] = j < hi;
            idx[u] = ok[u] ? j : hi - 1;
        }
        body(idx, ok);
    }
}

__device__ __forceinline__ float4 add4(float4 a, const float4& v) {
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    return a;
}

// sum over ranks, in rank order, of U staged float4; WT = compile-time world (0: run-time loop)
template <int WT>
__device__ __forceinline__ void pull_sum(const KArgs& a, size_t par, const long (&idx)[P2P_U], float scale,
                                         float4 (&acc)[P2P_U]) {
    if constexpr (WT > 0) {
        float4 v[WT > 0 ? WT : 1][P2P_U];
#pragma unroll
        for (int p = 0; p < WT; ++p)
#pragma unroll
            for (int u = 0; u < P2P_U; ++u) v[p][u] = a.stage[p][par + idx[u]];
#pragma unroll
        for (int u = 0; u < P2P_U; ++u) {
            acc[u] = v[0][u];
#pragma unroll
            for (int p = 1; p < WT; ++p) acc[u] = add4(acc[u], v[p][u]);
        }
    } else {
#pragma unroll
        for (int u = 0; u < P2P_U; ++u) acc[u] = a.stage[0][par + idx[u]];
        for (int p = 1; p < a.world; ++p) {
            float4 v[P2P_U];
#pragma unroll
            for (int u = 0; u < P2P_U; ++u) v[u] = a.stage[p][par + idx[u]];
#pragma unroll
            for (int u = 0; u < P2P_U; ++u) acc[u] = add4(acc[u], v[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < P2P_U; ++u) { acc[u].x *= scale; acc[u].y *= scale; acc[u].z *= scale; acc[u].w *= scale; }
}

template <bool FUSED, int WT>
__global__ __launch_bounds__(P2P_THREADS) void p2p_allreduce_kernel(KArgs a) {
    const int b = blockIdx.x, t = threadIdx.x, W = a.world, r = a.rank, NB = gridDim.x;
    __shared__ int s_epoch, s_dead;
    if (t == 0) {
        // call number: one word per communicator, advanced by the last block to finish (every block has read it by
        // then), so a captured graph replays with fresh numbers and all blocks of a call agree on the parity
        s_epoch = a.counters[0] + 1;
        s_dead = *(volatile int*)a.status;
    }
    __syncthreads();
    const int epoch = s_epoch;
    const bool dead = s_dead != 0;
    const size_t par = (size_t)(epoch & 1) * a.cap4;
    const long n4 = a.n4;
    const long stride = (long)NB * P2P_THREADS;
    const int nsl = a.oneshot ? 1 : W;
    const long q = (n4 + nsl - 1) / nsl;
    auto slice_lo = [&](int s) { return (long)s * q < n4 ? (long)s * q : n4; };
    auto slice_hi = [&](int s) { return (long)s * q + q < n4 ? (long)s * q + q : n4; };

    // 1. my shares of every slice -> my staging buffer, then tell every peer (and myself)
    {
        float4* st = a.stage[r] + par;
        for (int s = 0; s < nsl; ++s)
            walk_share(slice_lo(s), slice_hi(s), b, t, stride, [&](const long (&idx)[P2P_U], const bool (&ok)[P2P_U]) {
                float4 v[P2P_U];
#pragma unroll
                for (int u = 0; u < P2P_U; ++u) v[u] = a.in[idx[u]];
#pragma unroll
                for (int u = 0; u < P2P_U; ++u) if (ok[u]) st[idx[u]] = v[u];
            });
    }
    __atomic_thread_fence(__ATOMIC_RELEASE);            // system scope
    __syncthreads();
    if (t < W) st_sys(&a.flags[t]->ready[r][b], epoch);
    // 2. every peer's shares b are staged
    bool ok_all = wait_all(&a.flags[r]->ready[0][b], P2P_MAX_BLOCKS, W, epoch, a.status, a.timeout_ticks, dead);

    OptimHyper h;
    if (FUSED) h = optim_prepare(a.kind, a.lr, a.b1, a.b2, a.eps, a.rho, a.step);

    if (a.oneshot) {
        // 3'. pull and reduce everything, in rank order
        if (ok_all) {
            walk_share(0, n4, b, t, stride, [&](const long (&idx)[P2P_U], const bool (&ok)[P2P_U]) {
                float4 acc[P2P_U];
                pull_sum<WT>(a, par, idx, a.scale, acc);
#pragma unroll
                for (int u = 0; u < P2P_U; ++u) {
                    if (!ok[u]) continue;
                    if (FUSED) optim_update4(h, a.P, a.Mo, a.Vo, a.Pc, idx[u], acc[u], nullptr, 0.f);
                    else a.out[idx[u]] = acc[u];
                }
            });
        }
    } else {
        // 3. reduce share b of MY slice (pull from all stages, rank order), 4. push it into every result buffer
        if (ok_all) {
            walk_share(slice_lo(r), slice_hi(r), b, t, stride, [&](const long (&idx)[P2P_U], const bool (&ok)[P2P_U]) {
                float4 acc[P2P_U];
                pull_sum<WT>(a, par, idx, a.scale, acc);
                for (int p = 0; p < W; ++p) {
#pragma unroll
                    for (int u = 0; u < P2P_U; ++u) if (ok[u]) a.result[p][par + idx[u]] = acc[u];
                }
            });
        }
        __atomic_thread_fence(__ATOMIC_RELEASE);
        __syncthreads();
        if (t < W) st_sys(&a.flags[t]->done[r][b], epoch);
        // 5. every owner's share b has landed in my result buffer; 6. epilogue over all slices' shares b
        ok_all = wait_all(&a.flags[r]->done[0][b], P2P_MAX_BLOCKS, W, epoch, a.status, a.timeout_ticks, dead) && ok_all;
        if (ok_all) {
            const float4* res = a.result[r] + par;
            for (int s = 0; s < W; ++s)
                walk_share(slice_lo(s), slice_hi(s), b, t, stride, [&](const long (&idx)[P2P_U], const bool (&ok)[P2P_U]) {
                    float4 v[P2P_U];
#pragma unroll
                    for (int u = 0; u < P2P_U; ++u) v[u] = res[idx[u]];
#pragma unroll
                    for (int u = 0; u < P2P_U; ++u) {
                        if (!ok[u]) continue;
                        if (FUSED) optim_update4(h, a.P, a.Mo, a.Vo, a.Pc, idx[u], v[u], nullptr, 0.f);
                        else a.out[idx[u]] = v[u];
                    }
                });
        }
    }
    __syncthreads();
    if (t == 0) {
        if (atomicAdd(&a.counters[1], 1) == NB - 1) {       // last block of this call
            a.counters[1] = 0;
            __threadfence();
            a.counters[0] = epoch;
        }
    }
}

bool hip_ok(hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    g_perr = std::string(what) + ": " + hipGetErrorString(e);
    return false;
}

size_t flags_bytes() { return (sizeof(Flags) + 255) / 256 * 256; }
}  // namespace

P2P_API const char* ddl_p2p_last_error() { return g_perr.c_str(); }
P2P_API int ddl_p2p_handle_bytes() { return (int)sizeof(hipIpcMemHandle_t); }
P2P_API int ddl_p2p_max_world() { return P2P_MAX_WORLD; }

// Allocate this rank's symmetric buffer (capacity `cap_elems` fp32 per all-reduce) and export its IPC handle.
P2P_API void* ddl_p2p_create(int world, int rank, int device, long cap_elems, void* handle_out) {
    if (world < 1 || world > P2P_MAX_WORLD || rank < 0 || rank >= world) { g_perr = "bad world / rank"; return nullptr; }
    if (!hip_ok(hipSetDevice(device), "hipSetDevice")) return nullptr;
    auto* c = new P2PComm();
    c->world = world; c->rank = rank; c->device = device;
    c->cap4 = ((size_t)cap_elems + 3) / 4;
    c->cap4 = (c->cap4 + 63) / 64 * 64;
    c->bytes = flags_bytes() + 4 * c->cap4 * sizeof(float4);
    // uncached device memory: every access, local or through a peer mapping, is served by memory -- what makes
    // in-kernel flag signalling between GPUs (and between dies of one GPU) sound
    if (!hip_ok(hipExtMallocWithFlags((void**)&c->base, c->bytes, hipDeviceMallocUncached), "hipExtMallocWithFlags")) {
        delete c;
        return nullptr;
    }
    if (!hip_ok(hipMemset(c->base, 0, c->bytes), "hipMemset") ||
        !hip_ok(hipMalloc((void**)&c->counters, (P2P_MAX_BLOCKS + 1) * sizeof(int)), "hipMalloc") ||
        !hip_ok(hipMemset(c->counters, 0, (P2P_MAX_BLOCKS + 1) * sizeof(int)), "hipMemset") ||
        !hip_ok(hipIpcGetMemHandle(&c->handle, c->base), "hipIpcGetMemHandle") ||
        !hip_ok(hipDeviceSynchronize(), "hipDeviceSynchronize")) {
        hipFree(c->base);
        if (c->counters) hipFree(c->counters);
        delete c;
        return nullptr;
    }
    c->status = c->counters + P2P_MAX_BLOCKS;
    c->peer[rank] = c->base;
    memcpy(handle_out, &c->handle, sizeof(c->handle));
    return c;
}

// `handles`: world x handle_bytes, rank-major (all-gathered by the caller).
P2P_API int ddl_p2p_connect(void* comm, const void* handles) {
    auto* c = (P2PComm*)comm;
    if (!hip_ok(hipSetDevice(c->device), "hipSetDevice")) return -1;
    for (int p = 0; p < c->world; ++p) {
        if (p == c->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, (const char*)handles + (size_t)p * sizeof(h), sizeof(h));
        void* ptr = nullptr;
        if (!hip_ok(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle")) return -1;
        c->peer[p] = (char*)ptr;
    }
    return 0;
}

P2P_API int ddl_p2p_configure(void* comm, int max_blocks, long oneshot_bytes, double timeout_s) {
    auto* c = (P2PComm*)comm;
    if (max_blocks > 0) c->max_blocks = max_blocks > P2P_MAX_BLOCKS ? P2P_MAX_BLOCKS : max_blocks;
    if (oneshot_bytes >= 0) c->oneshot_bytes = (size_t)oneshot_bytes;
    if (timeout_s > 0) c->timeout_ticks = (long long)(timeout_s * 1e8);
    return 0;
}

static int launch(P2PComm* c, KArgs& a, bool fused, hipStream_t s) {
    for (int p = 0; p < c->world; ++p) {
        if (!c->peer[p]) { g_perr = "communicator not connected"; return -1; }
        a.flags[p] = (Flags*)c->peer[p];
        a.stage[p] = (float4*)(c->peer[p] + flags_bytes());
        a.result[p] = a.stage[p] + 2 * c->cap4;
    }
    a.cap4 = c->cap4; a.world = c->world; a.rank = c->rank;
    a.counters = c->counters; a.status = c->status; a.timeout_ticks = c->timeout_ticks;
    if (a.n4 <= 0 || (size_t)a.n4 > c->cap4) { g_perr = "all-reduce larger than the communicator's capacity"; return -2; }
    a.oneshot = (size_t)a.n4 * sizeof(float4) <= c->oneshot_bytes || c->world == 1;
    // every rank derives the same grid from n (the per-block flag protocol needs identical decompositions)
    long per = (long)P2P_THREADS * P2P_U;                   // one full walk step per thread at least
    long nb = (a.n4 + per - 1) / per;                       // sized for the epilogue, which covers the whole vector
    if (nb < 1) nb = 1;
    if (nb > c->max_blocks) nb = c->max_blocks;
#define P2P_LAUNCH(WT_)                                                                        \
    do {                                                                                       \
        if (fused) p2p_allreduce_kernel<true, WT_><<<(int)nb, P2P_THREADS, 0, s>>>(a);          \
        else p2p_allreduce_kernel<false, WT_><<<(int)nb, P2P_THREADS, 0, s>>>(a);               \
    } while (0)
    switch (c->world) {
        case 1: P2P_LAUNCH(1); break;
        case 2: P2P_LAUNCH(2); break;
        case 4: P2P_LAUNCH(4); break;
        case 8: P2P_LAUNCH(8); break;
        default: P2P_LAUNCH(0); break;
    }
    return hip_ok(hipGetLastError(), "p2p_allreduce launch") ? 0 : -3;
}

// out = scale * sum over ranks of in (fp32, n a multiple of 4, 16-byte aligned; in == out allowed).
P2P_API int ddl_p2p_all_reduce(void* comm, const float* in, float* out, long n, float scale, hipStream_t s) {
    auto* c = (P2PComm*)comm;
    if (n % 4 || ((uintptr_t)in & 15) || ((uintptr_t)out & 15)) { g_perr = "need n % 4 == 0 and 16-byte aligned buffers"; return -2; }
    KArgs a{};
    a.in = (const float4*)in; a.out = (float4*)out; a.n4 = n / 4; a.scale = scale;
    return launch(c, a, false, s);
}

// All-reduce of the flat gradient buffer with the optimizer update fused into the epilogue:
// P / M / V / Pc are updated from scale * sum(G) exactly as ddl_optim_step would (the caller has advanced `step`).
P2P_API int ddl_p2p_all_reduce_update(void* comm, const float* G, long n, float scale, float* P, float* M, float* V,
                                      void* Pc, const int* step, int kind, float lr, float b1, float b2, float eps,
                                      float rho, hipStream_t s) {
    auto* c = (P2PComm*)comm;
    if (n % 4 || ((uintptr_t)G & 15) || ((uintptr_t)P & 15)) { g_perr = "need n % 4 == 0 and 16-byte aligned buffers"; return -2; }
    KArgs a{};
    a.in = (const float4*)G; a.out = nullptr; a.n4 = n / 4; a.scale = scale;
    a.P = (float4*)P; a.Mo = (float4*)M; a.Vo = (float4*)V; a.Pc = (uint2*)Pc; a.step = step;
    a.kind = kind; a.lr = lr; a.b1 = b1; a.b2 = b2; a.eps = eps; a.rho = rho;
    return launch(c, a, true, s);
}

// 0 = healthy, 1 = a wait timed out at some point (results since then are undefined). Synchronises the device.
P2P_API int ddl_p2p_status(void* comm) {
    auto* c = (P2PComm*)comm;
    int v = -1;
    if (!hip_ok(hipMemcpy(&v, c->status, sizeof(int), hipMemcpyDeviceToHost), "hipMemcpy")) return -1;
    return v;
}

P2P_API void ddl_p2p_destroy(void* comm) {
    auto* c = (P2PComm*)comm;
    if (!c) return;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    for (int p = 0; p < c->world; ++p)
        if (p != c->rank && c->peer[p]) hipIpcCloseMemHandle(c->peer[p]);
    if (c->base) hipFree(c->base);
    if (c->counters) hipFree(c->counters);
    delete c;
}
