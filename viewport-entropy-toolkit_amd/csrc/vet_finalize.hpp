// vet_finalize.hpp — k_log2_table and the k_finalize kernels (mean over a plan's lattices where they ran as separate launches)
// Part of the gfx950 device code of the viewport -> tile -> entropy path (see vet_kernels.hpp for the map).
// Reference citations are relative to /root/reference/src/viewport_entropy_toolkit/.
// These small kernels are launched from more than one translation unit: internal linkage, one copy each.
#pragma once
#include "vet_common.hpp"

namespace vet {

// log2(k) for k = 1..n-1 (entry 0 = 0): integer-count entropies look their logarithms up
static __global__ void k_log2_table(double* __restrict__ tab, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) tab[i] = i ? log2((double)i) : 0.0;
}

// ------------------------------------------------------------------------------------------
// k_finalize: avg_entropy = (sum over lattices, in order) / K   (spatial_entropy.py:142-156)
// ------------------------------------------------------------------------------------------
static __global__ void k_finalize(const double* __restrict__ ent_k, int K, long rows, double* __restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < rows; i += (long)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += ent_k[(long)k * rows + i];
        out[i] = s / (double)K;
    }
}

// the same for a batch of videos: per-lattice values in [K][rows] (the videos' frames back to back, video v's from
// frame0[v]), the mean goes to every video's own output
static __global__ void k_finalize_batch(const double* __restrict__ ent_k, int K, long rows, const long* __restrict__ frame0,
                                 double* const* __restrict__ outs, int n_videos) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < rows; i += (long)gridDim.x * blockDim.x) {
        int lo = 0, hi = n_videos - 1;                         // last video with frame0 <= i
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (frame0[mid] <= i) lo = mid; else hi = mid - 1;
        }
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += ent_k[(long)k * rows + i];
        outs[lo][i - frame0[lo]] = s / (double)K;
    }
}

// the same for the frames of a resolve list only ([0] = count, then the frames)
static __global__ void k_finalize_list(const double* __restrict__ ent_k, int K, long rows, const uint32_t* __restrict__ list,
                                double* __restrict__ out) {
    const long n = (long)list[0];
    for (long j = blockIdx.x * (long)blockDim.x + threadIdx.x; j < n; j += (long)gridDim.x * blockDim.x) {
        const long i = (long)list[1 + j];
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += ent_k[(long)k * rows + i];
        out[i] = s / (double)K;
    }
}

}  // namespace vet
