// ring_handoff_probe - does "FREE published after the fragment reads are ISSUED" order a later LDS-DMA write behind those reads?
//
// VERDICT r4 item 2.  conv_ring.hip's consumers publish FREE[j] with a ds_write_b32 that follows, in program order, the last
// ds_read_b128 of ring stage j % 3; a loader that has seen the counter issues `buffer_load ... lds` (LDS-DMA) into that stage.
// The claim the kernel rests on: the LDS executes a wave's instructions in issue order, so when the counter's new value is
// VISIBLE to another wave every earlier ds_read of the publishing wave has already read its bytes - a DMA issued after the look
// cannot overtake them, however tight the loader is.  The two-steps-ahead loader variant of round 4 failed with exactly this guard
// where the shipped (serial) loader is "never tight"; this program makes the hand-off as tight as the hardware allows and counts.
//
// One workgroup per CU: wave 0 consumes, wave 1 loads, waves 2..7 (optional) hammer the LDS with ds_read_b128 so that the LDS
// queue is as deep as it gets.  ONE stage (16 KB, the size of a ring stage), refilled in place: pattern i = every dword equal to i.
//   consumer, iteration i:  wait FULL >= i + 1;  issue 16 x ds_read_b128 (the whole stage);  [mode]  publish FREE = i + 1;
//                           wait for the data;  every dword must equal i (else: a DMA write overtook a read that was issued
//                           before the publication) - mismatches counted.
//   loader, iteration i:    wait FREE >= i (no sleep in the poll);  16 KB of LDS-DMA from a hot 16 KB global line set holding
//                           pattern i;  s_waitcnt vmcnt(0);  publish FULL = i + 1.
// modes:  0  publish directly behind the ISSUE of the reads (what conv_ring.hip does)
//         1  publish behind s_waitcnt lgkmcnt(0) (the data have arrived: trivially safe, the reference point)
//         2  publish BEFORE the reads are issued, then sleep ~2 us, then read (positive control: must mismatch)
// Every wait is bounded; a give-up is counted and ends the workgroup's loop.
//
//   hipcc -O3 --offload-arch=gfx950 tools/ring_handoff_probe.hip -o /tmp/ring_handoff_probe && /tmp/ring_handoff_probe [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned lds_u32;
typedef __attribute__((address_space(3))) v4u lds_v4u;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int STAGE = 16384;               // bytes
constexpr int NPAT = 8;                    // distinct source patterns (i % NPAT; stage dwords hold the pattern's index i itself)
constexpr int SPIN_LIMIT = 1 << 16;

__device__ unsigned long long g_bad[3], g_reads[3], g_giveup[3];

__device__ __forceinline__ unsigned ld_flag(const lds_u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void st_flag(lds_u32* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void cbar() { asm volatile("" ::: "memory"); }

// src: [iters][STAGE / 4] dwords would be huge; instead the loader reads pattern i from src + (i % NPAT) * STAGE, and the host
// refills nothing: dword value = pattern index (i % NPAT).  The consumer therefore expects i % NPAT.
template <int MODE>
__global__ __launch_bounds__(512) void probe(const unsigned* src, int iters, int noise) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [stage 16 KB][flags]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    lds_u32* const f_full = (lds_u32*)(lptr_t)(smem + STAGE);
    lds_u32* const f_free = f_full + 4;
    lds_u32* const f_stop = f_full + 8;
    if (tid < 16) reinterpret_cast<unsigned*>(smem + STAGE)[tid] = 0u;
    __syncthreads();
    if (wave == 0) {                                                 // ---------------- consumer
        unsigned long long bad = 0, reads = 0;
        for (int i = 0; i < iters; ++i) {
            int tries = 0;
            while (ld_flag(f_full) < (unsigned)(i + 1)) {
                if (++tries > SPIN_LIMIT) { if (lane == 0) atomicAdd(&g_giveup[MODE], 1ull); goto done; }
            }
            cbar();
            if constexpr (MODE == 2) {
                if (lane == 0) st_flag(f_free, (unsigned)(i + 1));
                cbar();
                for (int k = 0; k < 2; ++k) __builtin_amdgcn_s_sleep(127);       // 2 x 8 k cycles: far beyond a DMA round trip
            }
            v4u v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = *reinterpret_cast<const volatile lds_v4u*>((lptr_t)(smem + k * 1024 + lane * 16));
            if constexpr (MODE == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            cbar();
            if constexpr (MODE != 2) {
                if (lane == 0) st_flag(f_free, (unsigned)(i + 1));
                cbar();
            }
            const unsigned want = (unsigned)(i % NPAT);
            int nb = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) nb += (v[k].x != want) + (v[k].y != want) + (v[k].z != want) + (v[k].w != want);
            bad += (unsigned)nb;
            reads += 64;
        }
    done:
        for (int o = 32; o; o >>= 1) { bad += __shfl_xor(bad, o, 64); }
        if (lane == 0) { atomicAdd(&g_bad[MODE], bad); atomicAdd(&g_reads[MODE], reads * 64ull); st_flag(f_stop, 1u); }
    } else if (wave == 1) {                                          // ---------------- loader
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, NPAT * STAGE, 0x00020000);
        for (int i = 0; i < iters; ++i) {
            int tries = 0;
            while (ld_flag(f_free) < (unsigned)i) {
                if (++tries > SPIN_LIMIT) { if (lane == 0) atomicAdd(&g_giveup[MODE], 1ull); return; }
            }
            cbar();
            const int base = (i % NPAT) * STAGE + lane * 16;
#pragma unroll
            for (int k = 0; k < 16; ++k)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)(smem + k * 1024), 16, base + k * 1024, 0, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) st_flag(f_full, (unsigned)(i + 1));
            cbar();
        }
    } else if (wave - 2 < noise) {                                   // ---------------- LDS queue filler
        v4u acc = {0u, 0u, 0u, 0u};
        int guard = 0;
        while (ld_flag(f_stop) == 0u && ++guard < (1 << 22)) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const v4u t = *reinterpret_cast<const volatile lds_v4u*>((lptr_t)(smem + ((k * 2048 + lane * 16) & (STAGE - 1))));
                acc.x ^= t.x; acc.y ^= t.y; acc.z ^= t.z; acc.w ^= t.w;
            }
        }
        if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0xdeadbeefu) atomicAdd(&g_giveup[MODE], 1ull << 40);
    }
}

template <int MODE>
static void run(const unsigned* src, int iters, int noise, int nwg, const char* what) {
    unsigned long long z[3] = {0, 0, 0};
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_bad), z, sizeof(z)));
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_reads), z, sizeof(z)));
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_giveup), z, sizeof(z)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe<MODE>, dim3(nwg), dim3(512), STAGE + 64, 0, src, iters, noise);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long bad[3], reads[3], give[3];
    CHECK(hipMemcpyFromSymbol(bad, HIP_SYMBOL(g_bad), sizeof(bad)));
    CHECK(hipMemcpyFromSymbol(reads, HIP_SYMBOL(g_reads), sizeof(reads)));
    CHECK(hipMemcpyFromSymbol(give, HIP_SYMBOL(g_giveup), sizeof(give)));
    printf("mode %d (%s), %d LDS-noise waves, %d workgroups x %d hand-offs: %llu dwords read, %llu stale/overtaken, give-ups %llu, %.1f ms (%.0f cycles per hand-off at 2.1 GHz)\n",
           MODE, what, noise, nwg, iters, reads[MODE], bad[MODE], give[MODE], ms, ms * 1e-3 * 2.1e9 / iters);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200000;
    unsigned* h = (unsigned*)malloc((size_t)NPAT * STAGE);
    for (int p = 0; p < NPAT; ++p)
        for (int d = 0; d < STAGE / 4; ++d) h[p * (STAGE / 4) + d] = (unsigned)p;
    unsigned* src;
    CHECK(hipMalloc(&src, (size_t)NPAT * STAGE));
    CHECK(hipMemcpy(src, h, (size_t)NPAT * STAGE, hipMemcpyHostToDevice));
    CHECK(hipFuncSetAttribute((const void*)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, STAGE + 64));
    CHECK(hipFuncSetAttribute((const void*)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, STAGE + 64));
    CHECK(hipFuncSetAttribute((const void*)probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, STAGE + 64));
    const int nwg = 256;
    for (int noise = 0; noise <= 6; noise += 3) {
        run<0>(src, iters, noise, nwg, "FREE behind the ISSUE of the reads: conv_ring.hip");
        run<1>(src, iters, noise, nwg, "FREE behind lgkmcnt(0)");
    }
    run<2>(src, iters > 2000 ? 2000 : iters, 0, nwg, "positive control: FREE before the reads");
    CHECK(hipFree(src));
    free(h);
    return 0;
}
