// store_rate.hip -- what a CU can store: the write half of a 256 x 256 bf16 GEMM tile epilogue without the GEMM.
// One workgroup of 8 waves per CU walks output tiles; a wave owns 128 rows x 64 columns of the tile and writes it with 16
// global_store_dwordx4 (8 rows x 128 bytes per instruction, as gemm_epilogue's wide path does).  Reported: microseconds per tile
// and bytes per second per CU for 16 / 64 / 256 workgroups, with and without a wait for the acknowledgements per tile, for row
// strides of N = 1024 and 4096 columns, a read-modify-write form, and a 1 KiB-contiguous form.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o store_rate store_rate.hip && ./store_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

// MODE 0: stores, s_waitcnt vmcnt(0) per tile | 1: stores, never a wait | 2: read-modify-write in two passes of 8, wait per tile
// 3: as 0 but every instruction writes 1 KiB contiguous | 4: as 0 with ~4 us of dependent VALU work per tile behind the wait
// (does a k-loop's worth of time between the bursts change the cost of a burst?) | 5: that VALU work alone
template <int MODE>
__global__ __launch_bounds__(512) void k(uint16_t* C, int N, int tiles_n, int tiles, int per_wg, float seed, float* sink) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, wr = wid >> 2, wc = wid & 3;
    uint4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = make_uint4(__float_as_uint(seed + i), lane, wid, i);
    float spin = seed;
    for (int it = 0; it < per_wg; ++it) {
        const int t = blockIdx.x + it * gridDim.x;
        if (t >= tiles) break;
        const int tm = t / tiles_n, tn = t - tm * tiles_n;
        uint16_t* base = C + (size_t)(tm * 256 + wr * 128) * N + tn * 256 + wc * 64;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint4* dst[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = h * 8 + i;
                if (MODE == 3) dst[i] = reinterpret_cast<uint4*>(C + (size_t)t * 65536 + wid * 8192 + r * 512) + lane;   // the tile stored contiguously
                else dst[i] = reinterpret_cast<uint4*>(base + (size_t)(r * 8 + (lane >> 3)) * N) + (lane & 7);
            }
            if (MODE == 2) {
                uint4 o[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = *dst[i];
#pragma unroll
                for (int i = 0; i < 8; ++i) { v[h * 8 + i].x += o[i].x; v[h * 8 + i].y ^= o[i].y; v[h * 8 + i].z += o[i].z; v[h * 8 + i].w ^= o[i].w; }
            }
            if (MODE != 5)
#pragma unroll
                for (int i = 0; i < 8; ++i) *dst[i] = v[h * 8 + i];
        }
        if (MODE != 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE >= 4) {
            for (int j = 0; j < 1500; ++j) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(spin));
        }
    }
    if (spin == 12345.f) sink[0] = spin;
}

template <int MODE>
void run(const char* name, uint16_t* C, float* sink, int M, int N) {
    const int tiles_n = N / 256, tiles = tiles_n * (M / 256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    printf("%-46s N=%4d:", name, N);
    for (int nwg : {16, 64, 256}) {
        const int per_wg = 32;                       // 32 tiles per workgroup whatever the grid
        const int used = std::min(tiles, nwg * per_wg);
        std::vector<float> ms;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k<MODE>, dim3(nwg), dim3(512), 0, 0, C, N, tiles_n, used, per_wg, 0.5f, sink);
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float t; (void)hipEventElapsedTime(&t, e0, e1);
            ms.push_back(t);
        }
        std::sort(ms.begin(), ms.end());
        const double us_tile = ms[2] * 1e3 / ((used + nwg - 1) / nwg);
        const double bytes = (MODE == 2 ? 2.0 : 1.0) * 131072.0;
        printf("  %3d WGs: %6.2f us/tile %6.1f GB/s/CU %5.2f TB/s", nwg, us_tile, bytes / us_tile * 1e-3, bytes * nwg / us_tile * 1e-6);
    }
    printf("\n");
}

int main() {
    const int M = 131072;
    uint16_t* C; float* sink;
    if (hipMalloc(&C, (size_t)M * 4096 * 2 + 4096) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    (void)hipMemset(C, 0, (size_t)M * 4096 * 2);
    for (int N : {1024, 4096}) {
        run<0>("stores, wait for the acks per tile", C, sink, M, N);
        run<1>("stores, no wait", C, sink, M, N);
        run<3>("stores of 1 KiB contiguous, wait per tile", C, sink, M, N);
        run<2>("read-modify-write (2 x 128 KiB), wait per tile", C, sink, M, N);
        run<4>("stores, wait, then ~4 us of VALU per tile", C, sink, M, N);
        run<5>("the ~4 us of VALU per tile alone", C, sink, M, N);
    }
    return 0;
}
