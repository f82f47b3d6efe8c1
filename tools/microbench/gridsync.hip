// What does a device-wide rendezvous cost on MI355X (8 XCDs, one L2 each), and can the data two phases exchange skip
// the L2 write-back / invalidate that a kernel boundary or cooperative_groups::grid().sync() pays?
//
//   mode 0  two kernels per round (write phase / read phase): the boundary is the rendezvous
//   mode 1  one cooperative launch, cooperative_groups grid sync (agent-scope fences: buffer_wbl2 / buffer_inv)
//   mode 2  one cooperative launch, hand-written barrier: the exchanged array is written and read with sc1
//           (agent-coherent) accesses, nothing is flushed; a monotonic counter + agent-scope atomics
//
// Each round every workgroup writes its chunk (value derived from the round) and, after the rendezvous, reads the
// chunk of the workgroup "across" (different XCD by construction) and checks it: a stale line shows up as a miscount.
//
//   hipcc --offload-arch=gfx950 -O3 -o gridsync gridsync.hip && ./gridsync
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <vector>
namespace cg = cooperative_groups;

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr int SC1 = 16;     // cache-policy bit 4 of the raw buffer intrinsics = sc1 on gfx94x / gfx950

struct Args {
    unsigned* data;        // [nwg][chunk] uint4-aligned
    unsigned chunk;        // uints per workgroup (multiple of 4 * 256)
    unsigned* counter;     // barrier counter (mode 2)
    unsigned* errors;      // mismatches seen
    unsigned* timeout;     // set when a barrier gave up
    int rounds;
    int coherent;          // exchanged array through sc1 accesses
};

__device__ __forceinline__ unsigned value(unsigned wg, unsigned i, unsigned round) { return wg * 2654435761u + i * 40503u + round * 97u; }

template <int CP> __device__ __forceinline__ void write_chunk(const Args& a, unsigned wg, unsigned round) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(a.data + (size_t)wg * a.chunk, 0, (int)(a.chunk * 4), 0x00020000);
    for (unsigned i = threadIdx.x * 4; i < a.chunk; i += blockDim.x * 4) {
        v4u v = {value(wg, i, round), value(wg, i + 1, round), value(wg, i + 2, round), value(wg, i + 3, round)};
        __builtin_amdgcn_raw_buffer_store_b128(v, r, i * 4, 0, CP);
    }
}
template <int CP> __device__ __forceinline__ unsigned check_chunk(const Args& a, unsigned wg, unsigned round) {
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(a.data + (size_t)wg * a.chunk, 0, (int)(a.chunk * 4), 0x00020000);
    unsigned bad = 0;
    for (unsigned i = threadIdx.x * 4; i < a.chunk; i += blockDim.x * 4) {
        v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, i * 4, 0, CP);
        bad += (v.x != value(wg, i, round)) + (v.y != value(wg, i + 1, round)) + (v.z != value(wg, i + 2, round)) +
               (v.w != value(wg, i + 3, round));
    }
    return bad;
}

// partner on another XCD (workgroups go round-robin over the 8 XCDs): shift by 3 + 8 k
__device__ __forceinline__ unsigned partner(unsigned wg, unsigned nwg) { return (wg + 11u) % nwg; }

__global__ void k_write(Args a, unsigned round) { write_chunk<0>(a, blockIdx.x, round); }
__global__ void k_read(Args a, unsigned round) {
    const unsigned bad = check_chunk<0>(a, partner(blockIdx.x, gridDim.x), round);
    if (bad) atomicAdd(a.errors, bad);
}

__global__ void k_coop_cg(Args a) {
    cg::grid_group grid = cg::this_grid();
    unsigned bad = 0;
    for (int r = 0; r < a.rounds; ++r) {
        write_chunk<0>(a, blockIdx.x, r);
        grid.sync();
        bad += check_chunk<0>(a, partner(blockIdx.x, gridDim.x), r);
        grid.sync();
    }
    if (bad) atomicAdd(a.errors, bad);
}

// All stores of the workgroup done (vmcnt(0) per wave, then the workgroup barrier), one lane announces the arrival
// and polls; no cache maintenance.  `target` counts arrivals since the launch (monotonic counter, zeroed by the host).
__device__ __forceinline__ bool grid_barrier(const Args& a, unsigned& target) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    target += gridDim.x;
    __shared__ int ok;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        int good = 1;
        while ((int)(__hip_atomic_load(a.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 20000000ull ||
                __hip_atomic_load(a.timeout, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {     // ~0.2 s: give up, tell everybody
                __hip_atomic_store(a.timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                good = 0;
                break;
            }
        }
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

template <int CP> __global__ void k_coop_own(Args a) {
    unsigned bad = 0, target = 0;
    for (int r = 0; r < a.rounds; ++r) {
        write_chunk<CP>(a, blockIdx.x, r);
        if (!grid_barrier(a, target)) return;
        bad += check_chunk<CP>(a, partner(blockIdx.x, gridDim.x), r);
        if (!grid_barrier(a, target)) return;
    }
    if (bad) atomicAdd(a.errors, bad);
}

int main(int argc, char** argv) {
    const int rounds = 200;
    hipStream_t s;
    CHK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    printf("%-6s %-8s %-10s %-12s %-10s %s\n", "nwg", "KB/wg", "mode", "us/round", "errors", "timeout");
    for (int nwg : {64, 128, 256, 512}) {
        for (unsigned chunk : {1024u, 16384u}) {      // 4 KB / 64 KB per workgroup
            Args a{};
            a.chunk = chunk;
            a.rounds = rounds;
            CHK(hipMalloc(&a.data, (size_t)nwg * chunk * 4));
            CHK(hipMalloc(&a.counter, 256));
            a.errors = a.counter + 16;
            a.timeout = a.counter + 32;
            for (int mode = 0; mode < 4; ++mode) {
                CHK(hipMemsetAsync(a.counter, 0, 256, s));
                CHK(hipMemsetAsync(a.data, 0, (size_t)nwg * chunk * 4, s));
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    CHK(hipMemsetAsync(a.counter, 0, 4, s));
                    CHK(hipEventRecord(e0, s));
                    if (mode == 0) {
                        for (int r = 0; r < rounds; ++r) {
                            hipLaunchKernelGGL(k_write, dim3(nwg), dim3(256), 0, s, a, (unsigned)r);
                            hipLaunchKernelGGL(k_read, dim3(nwg), dim3(256), 0, s, a, (unsigned)r);
                        }
                    } else {
                        void* params[] = {&a};
                        const void* fn = mode == 1 ? (const void*)k_coop_cg : mode == 2 ? (const void*)k_coop_own<SC1> : (const void*)k_coop_own<0>;
                        CHK(hipLaunchCooperativeKernel(fn, dim3(nwg), dim3(256), params, 0, s));
                    }
                    CHK(hipEventRecord(e1, s));
                    CHK(hipEventSynchronize(e1));
                    float ms = 0;
                    CHK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best) best = ms;
                }
                unsigned h[48];
                CHK(hipMemcpy(h, a.counter, sizeof(h), hipMemcpyDeviceToHost));
                const char* names[] = {"2-kernel", "cg-sync", "own+sc1", "own-plain"};
                // a round = two rendezvous (write | read | next write)
                printf("%-6d %-8u %-10s %-12.2f %-10u %u\n", nwg, chunk * 4 / 1024, names[mode], best * 1e3 / rounds, h[16], h[32]);
            }
            CHK(hipFree(a.data));
            CHK(hipFree(a.counter));
        }
    }
    return 0;
}
