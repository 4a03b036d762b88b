// valu_probe.hip — issue cost of the candidate forms of "mantissa x multiplicity -> 64-bit histogram increment"
// (the inner loop of the table walk, vet_spatial_lut.hpp).  One wave per SIMD, dependent-free chains, cycles per
// wave-instruction from s_memtime.  Build: hipcc -O3 --offload-arch=gfx950 -o valu_probe valu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define N_ITER 4096

template <int MODE>
__global__ void probe(uint32_t* out, unsigned long long* cycles, uint32_t seed) {
    uint32_t w[8];
    for (int i = 0; i < 8; ++i) w[i] = seed * (threadIdx.x + 1 + i) | 1u;
    uint32_t m = (seed >> 3) | 1u;
    unsigned long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double dacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) {            // v_mad_u64_u32
                acc[i] += (unsigned long long)w[i] * m;
            } else if (MODE == 1) {     // 24-bit multiplies: lo 16 bits and hi 16 bits of w times a 12-bit count, recombined
                const uint32_t c = m & 0xFFFu;
                const uint32_t a = __umul24(w[i] & 0xFFFFu, c), b = __umul24(w[i] >> 16, c);
                acc[i] += ((unsigned long long)b << 16) + a;
            } else if (MODE == 2) {     // no multiply: zero-extended add (count 1, shift 0)
                acc[i] += w[i];
            } else if (MODE == 3) {     // 64-bit shift by a variable amount
                acc[i] += (unsigned long long)w[i] << (m & 31);
            } else if (MODE == 4) {     // FP64: cvt + fma
                dacc[i] = fma((double)w[i], (double)m, dacc[i]);
            } else if (MODE == 5) {     // v_mul_lo_u32 + v_mul_hi_u32
                acc[i] += ((unsigned long long)__umulhi(w[i], m) << 32) | (w[i] * m);
            } else if (MODE == 6) {     // 32-bit shifts (count 1, shift 16): {w << 16, w >> 16}
                acc[i] += ((unsigned long long)(w[i] >> 16) << 32) | (w[i] << 16);
            }
            w[i] += 0x9E3779B9u;
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    uint32_t r = 0;
    for (int i = 0; i < 8; ++i) r ^= (uint32_t)acc[i] ^ (uint32_t)(acc[i] >> 32) ^ (uint32_t)dacc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name) {
    uint32_t* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    for (int waves_per_simd : {1, 2}) {
        const int threads = 256 * waves_per_simd;
        hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, 12345u);
        hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, 12345u);
        hipDeviceSynchronize();
        unsigned long long h[256];
        hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += (double)v;
        // s_memtime counts at a fixed 100 MHz-derived rate on some parts; report raw counter units per element-op per wave
        printf("%-44s %d wave(s)/SIMD: %8.2f counter units per 8-op iteration\n", name, waves_per_simd, s / 256 / N_ITER);
    }
    hipFree(out); hipFree(cyc);
}

int main() {
    run<2>("baseline: 64-bit add of zero-extended w");
    run<0>("v_mad_u64_u32 (w * mult)");
    run<5>("v_mul_lo_u32 + v_mul_hi_u32");
    run<1>("two v_mul_u32_u24 + recombine");
    run<3>("v_lshlrev_b64 (w << s)");
    run<6>("two 32-bit shifts (w << 16 as 64 bit)");
    run<4>("v_cvt_f64_u32 + v_fma_f64");
    return 0;
}
