// vet_weights_pass.hpp — k_weights_gather: tile_weights VALUES of lattice 0 at the reference's precision
// Part of the gfx950 device code of the viewport -> tile -> entropy path (see vet_kernels.hpp for the map).
// Reference citations are relative to /root/reference/src/viewport_entropy_toolkit/.
#pragma once
#include "vet_spatial_sweep.hpp"

namespace vet {

// ------------------------------------------------------------------------------------------
// k_weights_gather — the second return value of compute_spatial_entropy (entropy_utils.py:179-192): per frame the dict
// tile -> sum over the users of calculate_tile_weights' exact FP64 weights (:124-137), whatever formulation produced
// the frame's ENTROPY (the table formulations hold block-floating-point / FP32 weights, good for the entropy contract
// only).  Off the hot path: only calls that ask for the weights output, and the fetch of a block of weight rows of a
// device-resident result, run it.
// One workgroup per frame.  Wave w takes the w-th contiguous share of the frame's users IN COLUMN ORDER and adds each
// user's exact weight row (k_wexact: ELL, FP64) into its own LDS histogram with ds_add_f64 — a wave's LDS instructions
// execute in program order and the tiles of one row are distinct, so every per-tile sum runs over the wave's users in
// column order; the waves' histograms are then added in wave order: a fixed summation order, the same for the eager
// output and for a fetched block.  G users' rows are requested before the first add (the loads of a row are
// independent of everything but the row id).
// Histograms start at -0.0 = "no key" (vet_spatial_sweep.hpp: NO_KEY_BITS): a tile whose only weights are exactly 0.0
// reads +0.0 = key with the value 0.0 -> -0.0 in the dense output, as the precise sweep writes it (include/vet.h).
// S: 64-entry chunks of a row (1, 2, 4; 0 = any number, one user at a time).
// ------------------------------------------------------------------------------------------
struct WeightsGatherParams {
    SampleSrc src;
    int U, T;
    const uint32_t* alias;      // [n_dirs] direction -> row | mirrored << 31
    const uint16_t* idx;        // [R][stride]
    const double* w;            // [R][stride]
    const uint32_t* len;        // [R]
    int stride, n;
    double* out;                // [T][n]
};

template <bool FROM_IDS, int S>
__global__ __launch_bounds__(256) void k_weights_gather(const WeightsGatherParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* hist = (double*)smem;                                  // [NW][n]
    const int NW = blockDim.x >> 6, tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const long f = blockIdx.x;
    double* h = hist + (size_t)wv * p.n;
    for (int t = lane; t < p.n; t += WAVE) ((unsigned long long*)h)[t] = NO_KEY_BITS;
    const int per = (p.U + NW - 1) / NW;
    const int u_begin = wv * per, u_end = min(p.U, u_begin + per);
    bool bad = false;
    constexpr int G = S ? 8 / S : 1, SS = S ? S : 1;
    for (int u0 = u_begin; u0 < u_end; u0 += WAVE) {
        const int u = u0 + lane;
        const int id = u < u_end ? sample_dir<FROM_IDS>(p.src, f * (long)p.U + u, bad) : -1;
        const uint32_t a = id >= 0 ? p.alias[id] : 0u;
        const int row = (int)(a & 0x7FFFFFFFu), mir = (int)(a >> 31);
        const int ln = id >= 0 ? (int)p.len[row] : 0;
        const int cnt = min(WAVE, u_end - u0);
        for (int j0 = 0; j0 < cnt; j0 += G) {
            uint16_t ti[G][SS];
            double wt[G][SS];
            int lj[G], mj[G];
            size_t base[G];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int j = min(j0 + g, cnt - 1);
                lj[g] = j0 + g < cnt ? __builtin_amdgcn_readlane(ln, j) : 0;
                mj[g] = __builtin_amdgcn_readlane(mir, j);
                base[g] = (size_t)__builtin_amdgcn_readlane(row, j) * (size_t)p.stride;
            }
            if (S) {
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int s = 0; s < SS; ++s) {
                        const int e = s * WAVE + lane;
                        const bool ok = e < lj[g];
                        ti[g][s] = ok ? p.idx[base[g] + e] : (uint16_t)0;
                        wt[g][s] = ok ? p.w[base[g] + e] : 0.0;
                    }
#pragma unroll
                for (int g = 0; g < G; ++g)
#pragma unroll
                    for (int s = 0; s < SS; ++s)
                        if (s * WAVE + lane < lj[g]) atomicAdd(&h[mj[g] ? p.n - 1 - (int)ti[g][s] : (int)ti[g][s]], wt[g][s]);
            } else {
                for (int e = lane; e < lj[0]; e += WAVE) {
                    const int t = (int)p.idx[base[0] + e];
                    atomicAdd(&h[mj[0] ? p.n - 1 - t : t], p.w[base[0] + e]);
                }
            }
        }
    }
    __syncthreads();
    for (int t = tid; t < p.n; t += blockDim.x) {
        double v = hist[t];
        for (int w2 = 1; w2 < NW; ++w2) v += hist[(size_t)w2 * p.n + t];       // -0.0 + -0.0 = -0.0: still "no key"
        const bool key = (unsigned long long)__double_as_longlong(v) != NO_KEY_BITS;
        __builtin_nontemporal_store(key ? (v == 0.0 ? -0.0 : v) : 0.0, p.out + f * (long)p.n + t);
    }
}

}  // namespace vet
