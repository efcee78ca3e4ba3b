// Timing harness for k_ntt_pass8 build variants (development aid; values are NOT checked here -- parity is the library's tests).
//   hipcc --offload-arch=gfx950 -O3 -DV_PB=5 -DV_MINW=3 -I../panda_amd/csrc ntt8_variants.hip -o bin/ntt8_pb5_w3
// Runs the three passes of a 2^24 BN254-Fr transform (first / middle / last pass geometry) on random data with random tables.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "ntt_radix8.h"

#ifndef V_PB
#define V_PB 5
#endif
#ifndef V_MINW
#define V_MINW 3
#endif
using namespace panda_ntt8;
typedef Bn254Fr F;

int main(int argc, char **argv)
{
    const unsigned log_n = argc > 1 ? atoi(argv[1]) : 24;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const size_t n = (size_t)1 << log_n;
    u32 *a, *b, *pq, *ta, *tb;
    hipMalloc(&a, n * 32);
    hipMalloc(&b, n * 32);
    hipMalloc(&pq, 128 * TW2_STRIDE * 4);
    hipMalloc(&ta, (size_t)65536 * TW2_STRIDE * 4);
    hipMalloc(&tb, (size_t)65536 * TW2_STRIDE * 4);
    {
        std::vector<u32> h(n * 8);
        u32 x = 12345;
        for (size_t i = 0; i < h.size(); i++) {
            x = x * 1664525u + 1013904223u;
            h[i] = (i & 7) == 7 ? (x >> 4) : x; // below 2^252 < p
        }
        hipMemcpy(a, h.data(), n * 32, hipMemcpyHostToDevice);
        std::vector<u32> t((size_t)65536 * TW2_STRIDE);
        for (size_t i = 0; i < t.size(); i++) {
            x = x * 1664525u + 1013904223u;
            t[i] = x & LIMB_MASK;
        }
        hipMemcpy(ta, t.data(), t.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(tb, t.data(), t.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(pq, t.data(), 128 * TW2_STRIDE * 4, hipMemcpyHostToDevice);
    }
    hipEvent_t e[4];
    for (auto &ev : e) hipEventCreate(&ev);
    const unsigned tiles = (unsigned)(n / ELEMS);
#ifdef P8_PERSIST
    const unsigned grid = V_PERSIST < tiles ? V_PERSIST : tiles; // workgroups that walk the tiles
#else
    const unsigned grid = tiles;
#endif
    Pass8Args p1{}, p2{}, p3{};
    p1.x = a; p1.y = b; p1.pq = pq; p1.ta = ta; p1.tb = tb; p1.log_n = log_n; p1.lgp = 0; p1.ca = 8; p1.cb = 0; p1.i2_shift = log_n - 16; p1.tiles = tiles;
    p2 = p1; p2.x = b; p2.y = a; p2.lgp = 8; p2.ca = 8; p2.cb = 8; p2.i2_shift = log_n - 24;
    p3 = p1; p3.lgp = 16; p3.ca = p3.cb = 0;
    float best[4] = {1e9f, 1e9f, 1e9f, 1e9f}, sum[4] = {0, 0, 0, 0};
    for (int r = 0; r < reps + 2; r++) {
        hipEventRecord(e[0]);
        hipLaunchKernelGGL((k_ntt_pass8<F, true, false, V_PB, V_MINW>), dim3(grid), dim3(THREADS), 0, 0, p1);
        hipEventRecord(e[1]);
        hipLaunchKernelGGL((k_ntt_pass8<F, false, false, V_PB, V_MINW>), dim3(grid), dim3(THREADS), 0, 0, p2);
        hipEventRecord(e[2]);
        hipLaunchKernelGGL((k_ntt_pass8<F, false, true, V_PB, V_MINW>), dim3(grid), dim3(THREADS), 0, 0, p3);
        hipEventRecord(e[3]);
        hipEventSynchronize(e[3]);
        if (r < 2) continue;
        float t[4];
        for (int i = 0; i < 3; i++) hipEventElapsedTime(&t[i], e[i], e[i + 1]);
        hipEventElapsedTime(&t[3], e[0], e[3]);
        for (int i = 0; i < 4; i++) {
            best[i] = t[i] < best[i] ? t[i] : best[i];
            sum[i] += t[i];
        }
    }
    printf("PB=%d MINW=%d 2^%u: first %.3f  middle %.3f  last %.3f  total %.3f ms (best); mean total %.3f ms; %s\n", V_PB, V_MINW, log_n, best[0], best[1], best[2],
           best[3], sum[3] / reps, hipGetErrorString(hipGetLastError()));
    return 0;
}
