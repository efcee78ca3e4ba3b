// Micro-benchmark: products by constants -- Montgomery (fe_mul) against the precomputed-quotient product (fe_mul_shoup) with the
// constant in vector registers (one per lane) or scalar registers (one per wave), bare and inside a radix-2 butterfly.
//   hipcc --offload-arch=gfx950 -O3 -I../panda_amd/csrc ubench_shoup.hip -o ubench_shoup
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "fe29.h"
using namespace panda29;
typedef Bn254Fr F;

#define ITERS 256
// VARIANT 0: x = x * w (Montgomery)      1: shoup, w per lane      2: shoup, w per wave
//         3: butterfly, Montgomery       4: butterfly, shoup per lane   5: butterfly, shoup per wave
//         6: x2-interleaved products, w per lane     7: two butterflies per iteration with x2-interleaved products, w per lane     8: the same, w per wave
template <int VARIANT>
__global__ void __launch_bounds__(256) k_mul(u32 *out, const u32 *in, const u32 *tw)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    Fe<F> x, z, w, x2, z2;
    FeTw<F> t;
    for (int j = 0; j < 9; j++) {
        x.l[j] = in[(i & 1023) * 18 + j] & LIMB_MASK;
        z.l[j] = in[(i & 1023) * 18 + 9 + j] & LIMB_MASK;
    }
    x.l[8] &= 0x1fffff;
    z.l[8] &= 0x1fffff;
    x2 = z;
    z2 = x;
    x2.l[0] ^= 1;
    const int ti = (VARIANT == 2 || VARIANT == 5 || VARIANT == 8) ? (blockIdx.x & 63) : (i & 63);
    for (int j = 0; j < 9; j++) {
        t.w[j] = tw[ti * 20 + j];
        t.q[j] = tw[ti * 20 + 9 + j];
        w.l[j] = t.w[j];
    }
    for (int it = 0; it < ITERS; it++) {
        if (VARIANT == 0) {
            fe_mul(x, x, w);
            fe_mul(z, z, w);
        } else if (VARIANT == 1) {
            fe_mul_shoup<F, false>(x, x, t);
            fe_mul_shoup<F, false>(z, z, t);
        } else if (VARIANT == 2) {
            fe_mul_shoup<F, true>(x, x, t);
            fe_mul_shoup<F, true>(z, z, t);
        } else if (VARIANT == 6) {
            fe_mul_shoup_x2<F, false>(x, z, x, z, t.w, t.q, t.w, t.q);
        } else if (VARIANT == 7 || VARIANT == 8) {
            // (x, z) and (z2, x2) are two independent butterflies
            Fe<F> s0, s1, r0, r1;
            fe_add(s0, x, z);
            fe_sub_raw<F, 4>(r0, x, z);
            fe_add(s1, x2, z2);
            fe_sub_raw<F, 4>(r1, x2, z2);
            if (VARIANT == 7)
                fe_mul_shoup_x2<F, false>(z, z2, r0, r1, t.w, t.q, t.w, t.q);
            else
                fe_mul_shoup_x2<F, true>(z, z2, r0, r1, t.w, t.q, t.w, t.q);
            for (int j = 0; j < 9; j++) {
                s0.l[j] &= LIMB_MASK;
                s1.l[j] &= LIMB_MASK;
            }
            s0.l[8] &= 0x1fffff;
            s1.l[8] &= 0x1fffff;
            x = s0;
            x2 = s1;
        } else {
            Fe<F> s, d, raw;
            fe_add(s, x, z);
            fe_sub_raw<F, 4>(raw, x, z);
            if (VARIANT == 3)
                fe_mul(d, raw, w);
            else if (VARIANT == 4)
                fe_mul_shoup<F, false>(d, raw, t);
            else
                fe_mul_shoup<F, true>(d, raw, t);
            // keep the sum small without changing the instruction mix of the next round: mask instead of reducing
            for (int j = 0; j < 9; j++) s.l[j] &= LIMB_MASK;
            s.l[8] &= 0x1fffff;
            x = s;
            z = d;
        }
    }
    u32 sum = 0;
    for (int j = 0; j < 9; j++) sum += x.l[j] * 3 + z.l[j] * 7 + x2.l[j] * 11 + z2.l[j] * 13;
    out[i] = sum;
}

template <int V>
static void run(const char *name, int blocks, u32 *out, u32 *in, u32 *tw)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k_mul<V><<<blocks, 256>>>(out, in, tw);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 3; r++) k_mul<V><<<blocks, 256>>>(out, in, tw);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    const double muls = (double)blocks * 256 * ITERS * ((V < 3 || V >= 6) ? 2 : 1);
    u32 chk;
    hipMemcpy(&chk, out, 4, hipMemcpyDeviceToHost);
    printf("%-28s blocks=%5d  %8.3f ms  %8.2f G %s/s  (check %08x)\n", name, blocks, ms, muls / (ms * 1e-3) * 1e-9, (V < 3 || V == 6) ? "mulmod" : "butterflies", chk);
}

int main()
{
    u32 *out, *in, *tw;
    hipMalloc(&out, 4 * 256 * 8192);
    hipMalloc(&in, 1024 * 18 * 4);
    hipMalloc(&tw, 64 * 20 * 4);
    std::vector<u32> h(1024 * 18);
    for (size_t i = 0; i < h.size(); i++) h[i] = (u32)(i * 2654435761u + 12345u);
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<u32> ht(64 * 20, 0);
    for (int e = 0; e < 64; e++) {
        Fe<F> wi;
        for (int j = 0; j < 9; j++) wi.l[j] = (u32)((e * 9 + j) * 2246822519u + 7u) & LIMB_MASK;
        wi.l[8] &= 0x1fffff;
        FeTw<F> t;
        fe_shoup_prepare(t, wi);
        for (int j = 0; j < 9; j++) {
            ht[e * 20 + j] = t.w[j];
            ht[e * 20 + 9 + j] = t.q[j];
        }
    }
    hipMemcpy(tw, ht.data(), ht.size() * 4, hipMemcpyHostToDevice);
    for (int wps : {1, 2, 3, 4}) {
        int blocks = 256 * wps;
        printf("--- %d waves/SIMD ---\n", wps);
        run<0>("montgomery", blocks, out, in, tw);
        run<1>("shoup (w per lane)", blocks, out, in, tw);
        run<2>("shoup (w per wave)", blocks, out, in, tw);
        run<3>("butterfly montgomery", blocks, out, in, tw);
        run<4>("butterfly shoup per lane", blocks, out, in, tw);
        run<5>("butterfly shoup per wave", blocks, out, in, tw);
        run<6>("shoup x2 (w per lane)", blocks, out, in, tw);
        run<7>("2 butterflies, x2 per lane", blocks, out, in, tw);
        run<8>("2 butterflies, x2 per wave", blocks, out, in, tw);
    }
    return 0;
}
