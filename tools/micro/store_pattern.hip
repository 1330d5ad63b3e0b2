// Microbenchmark (development aid): how fast can the chip absorb the match kernel's store pattern, with no compute?
//   pattern 0: v2/v3 pattern  -- WG = (panel of 256 cols, row group); wave = 32 rows; per 64-col step 32 dword stores per lane-set
//   pattern 1: row-sweep      -- WG owns 256 rows, sweeps all M columns 64 at a time (wave = 32 rows), dword stores
//   pattern 2: linear dwordx4 fill
//   pattern 3: v3 pattern but 128 cols per step via dwordx2? (lane -> 2 consecutive cols)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) float v4f;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <bool NT>
__device__ __forceinline__ void st(float* p, float v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

template <bool NT>
__global__ __launch_bounds__(512) void pat0(float* sim, int R, int M, int G)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 31, h = lane >> 5;
    const int g = blockIdx.x % G, panel = blockIdx.x / G, col0 = panel * 256;
    const int nrb = R / 256;
    for (int rb = g; rb < nrb; rb += G) {
        const int row0 = rb * 256 + wave * 32;
        for (int cp = 0; cp < 4; ++cp) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int grow = row0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                float* o = sim + (long)grow * M + col0 + cp * 64 + lr;
                st<NT>(o, (float)reg);
                st<NT>(o + 32, (float)reg);
            }
        }
    }
}

template <bool NT>
__global__ __launch_bounds__(512) void pat1(float* sim, int R, int M)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 31, h = lane >> 5;
    const int row0 = blockIdx.x * 256 + wave * 32;
    for (int c = 0; c < M; c += 64) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int grow = row0 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            float* o = sim + (long)grow * M + c + lr;
            st<NT>(o, (float)reg);
            st<NT>(o + 32, (float)reg);
        }
    }
}

template <bool NT>
__global__ __launch_bounds__(256) void pat2(float4* sim, long n4)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        v4f v = {1.f, 2.f, 3.f, 4.f};
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(sim) + i); else reinterpret_cast<v4f*>(sim)[i] = v;
    }
}

// pattern 3: as pattern 0 but each store instruction covers 1 row x 256 contiguous bytes... (lane -> col lane, rows 2 per reg pair)
template <bool NT>
__global__ __launch_bounds__(512) void pat3(float* sim, int R, int M, int G)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = blockIdx.x % G, panel = blockIdx.x / G, col0 = panel * 256;
    const int nrb = R / 256;
    for (int rb = g; rb < nrb; rb += G) {
        const int row0 = rb * 256 + wave * 32;
        for (int cp = 0; cp < 4; ++cp) {
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                float* o = sim + (long)(row0 + r) * M + col0 + cp * 64 + lane;
                st<NT>(o, (float)r);
            }
        }
    }
}

// pattern 4: as pattern 0 but the wave writes its 32 x 256 tile row by row: 4 x dwordx4 per row (1 KB contiguous per row)
template <bool NT>
__global__ __launch_bounds__(512) void pat4(float* sim, int R, int M, int G)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = blockIdx.x % G, panel = blockIdx.x / G, col0 = panel * 256;
    const int nrb = R / 256;
    for (int rb = g; rb < nrb; rb += G) {
        const int row0 = rb * 256 + wave * 32;
#pragma unroll 8
        for (int r = 0; r < 32; ++r) {
            v4f* o = reinterpret_cast<v4f*>(sim + (long)(row0 + r) * M + col0) + lane;
            v4f v = {1.f, 2.f, 3.f, (float)r};
            if (NT) __builtin_nontemporal_store(v, o); else *o = v;
        }
    }
}

// pattern 5: swapped MFMA roles: lane (row = lane&31, h) holds 4 consecutive columns per register quad: dwordx4 stores, 32 B per row per instruction
template <bool NT>
__global__ __launch_bounds__(512) void pat5(float* sim, int R, int M, int G)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & 31, h = lane >> 5;
    const int g = blockIdx.x % G, panel = blockIdx.x / G, col0 = panel * 256;
    const int nrb = R / 256;
    for (int rb = g; rb < nrb; rb += G) {
        const int row0 = rb * 256 + wave * 32;
        for (int cp = 0; cp < 4; ++cp) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v4f* o = reinterpret_cast<v4f*>(sim + (long)(row0 + lr) * M + col0 + cp * 64 + blk * 32 + 8 * q + 4 * h);
                    v4f v = {1.f, 2.f, 3.f, (float)q};
                    if (NT) __builtin_nontemporal_store(v, o); else *o = v;
                }
        }
    }
}

int main()
{
    const int R = 32768, M = 8192;
    float* sim;
    CK(hipMalloc(&sim, (size_t)R * M * 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const double bytes = (double)R * M * 4;
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a));
        for (int i = 0; i < 10; ++i) launch();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        printf("%-44s %8.1f us  %6.0f GB/s\n", name, ms * 100, bytes / (ms / 10 * 1e-3) / 1e9);
    };
    for (int G : {8, 16}) {
        char nm[96];
        snprintf(nm, 96, "pat0 panel/rowblock dword NT G=%d", G);
        run(nm, [&] { hipLaunchKernelGGL(pat0<true>, dim3(32 * G), dim3(512), 0, 0, sim, R, M, G); });
        snprintf(nm, 96, "pat0 panel/rowblock dword plain G=%d", G);
        run(nm, [&] { hipLaunchKernelGGL(pat0<false>, dim3(32 * G), dim3(512), 0, 0, sim, R, M, G); });
        snprintf(nm, 96, "pat3 row-per-instr 256B dword NT G=%d", G);
        run(nm, [&] { hipLaunchKernelGGL(pat3<true>, dim3(32 * G), dim3(512), 0, 0, sim, R, M, G); });
        snprintf(nm, 96, "pat4 row-per-instr 1KB dwordx4 NT G=%d", G);
        run(nm, [&] { hipLaunchKernelGGL(pat4<true>, dim3(32 * G), dim3(512), 0, 0, sim, R, M, G); });
        snprintf(nm, 96, "pat5 swapped-role dwordx4 32B/row NT G=%d", G);
        run(nm, [&] { hipLaunchKernelGGL(pat5<true>, dim3(32 * G), dim3(512), 0, 0, sim, R, M, G); });
        snprintf(nm, 96, "pat5 swapped-role dwordx4 32B/row plain G=%d", G);
        run(nm, [&] { hipLaunchKernelGGL(pat5<false>, dim3(32 * G), dim3(512), 0, 0, sim, R, M, G); });
        snprintf(nm, 96, "pat4 row-per-instr 1KB dwordx4 plain G=%d", G);
        run(nm, [&] { hipLaunchKernelGGL(pat4<false>, dim3(32 * G), dim3(512), 0, 0, sim, R, M, G); });
    }
    run("pat1 row sweep dword NT", [&] { hipLaunchKernelGGL(pat1<true>, dim3(R / 256), dim3(512), 0, 0, sim, R, M); });
    run("pat1 row sweep dword plain", [&] { hipLaunchKernelGGL(pat1<false>, dim3(R / 256), dim3(512), 0, 0, sim, R, M); });
    run("pat2 linear dwordx4 NT", [&] { hipLaunchKernelGGL(pat2<true>, dim3(2048), dim3(256), 0, 0, (float4*)sim, (long)R * M / 4); });
    run("pat2 linear dwordx4 plain", [&] { hipLaunchKernelGGL(pat2<false>, dim3(2048), dim3(256), 0, 0, (float4*)sim, (long)R * M / 4); });
    CK(hipMemsetAsync(sim, 0, (size_t)R * M * 4, 0));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 10; ++i) CK(hipMemsetAsync(sim, 0, (size_t)R * M * 4, 0));
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-44s %8.1f us  %6.0f GB/s\n", "hipMemsetAsync", ms * 100, bytes / (ms / 10 * 1e-3) / 1e9);
    return 0;
}
