// vet_plan_kernels.hpp — per-plan tables (k_grid_dirs, k_unit_dirs, k_nearest_lut), k_log2_table, k_finalize
// Part of the gfx950 device code of the viewport -> tile -> entropy path (see vet_kernels.hpp for the map).
// Reference citations are relative to /root/reference/src/viewport_entropy_toolkit/.
#pragma once
#include "vet_common.hpp"

namespace vet {

// ------------------------------------------------------------------------------------------
// k_grid_dirs: Vector.from_spherical over the pixel grid (data_types.py:204-216) from the
// host's axis tables, then the unit vector vector_angle_distance works with
// (entropy_utils.py:55-58).  raw = rounded Vector xyz (parity hook), unit = raw / |raw|.
// ------------------------------------------------------------------------------------------
__global__ void k_grid_dirs(const double* __restrict__ lon_cos, const double* __restrict__ lon_sin,
                            const double* __restrict__ lat_sin, const double* __restrict__ lat_cos,
                            int W, int H, double* __restrict__ raw, double* __restrict__ unit) {
    const long D = (long)(W + 1) * (H + 1);
    for (long d = blockIdx.x * (long)blockDim.x + threadIdx.x; d < D; d += (long)gridDim.x * blockDim.x) {
        const int py = (int)(d / (W + 1)), px = (int)(d % (W + 1));
        const double sp = lat_sin[py];
        const double x = round6(sp * lon_cos[px]);
        const double y = round6(sp * lon_sin[px]);
        const double z = round6(lat_cos[py]);
        raw[3 * d + 0] = x; raw[3 * d + 1] = y; raw[3 * d + 2] = z;
        const double len = sqrt(x * x + y * y + z * z);
        unit[3 * d + 0] = x / len; unit[3 * d + 1] = y / len; unit[3 * d + 2] = z / len;
    }
}

// explicit direction table: just the normalisation
__global__ void k_unit_dirs(const double* __restrict__ raw, long D, double* __restrict__ unit) {
    for (long d = blockIdx.x * (long)blockDim.x + threadIdx.x; d < D; d += (long)gridDim.x * blockDim.x) {
        const double x = raw[3 * d], y = raw[3 * d + 1], z = raw[3 * d + 2];
        const double len = sqrt(x * x + y * y + z * z);
        unit[3 * d + 0] = x / len; unit[3 * d + 1] = y / len; unit[3 * d + 2] = z / len;
    }
}

// ------------------------------------------------------------------------------------------
// k_nearest_lut: find_nearest_tile (entropy_utils.py:89-106) for every direction of the table.
// arccos is monotone, so arg-min distance == arg-max cosine; '>' keeps the lowest index on
// exact ties, as np.argmin does.  lane = direction, the tile walks through LDS (broadcast).
// ------------------------------------------------------------------------------------------
__global__ void k_nearest_lut(const double* __restrict__ unit, long D, const double* __restrict__ tiles,
                              int n, uint16_t* __restrict__ nearest) {
    extern __shared__ double s_tiles[];
    for (int i = threadIdx.x; i < 3 * n; i += blockDim.x) s_tiles[i] = tiles[i];
    __syncthreads();
    for (long d = blockIdx.x * (long)blockDim.x + threadIdx.x; d < D; d += (long)gridDim.x * blockDim.x) {
        const double x = unit[3 * d], y = unit[3 * d + 1], z = unit[3 * d + 2];
        double best = -2.0;
        int bi = 0;
        for (int t = 0; t < n; ++t) {
            const double c = fma(z, s_tiles[3 * t + 2], fma(y, s_tiles[3 * t + 1], x * s_tiles[3 * t]));
            if (c > best) { best = c; bi = t; }
        }
        nearest[d] = (uint16_t)bi;
    }
}


// log2(k) for k = 1..n-1 (entry 0 = 0): integer-count entropies look their logarithms up
__global__ void k_log2_table(double* __restrict__ tab, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) tab[i] = i ? log2((double)i) : 0.0;
}

// ------------------------------------------------------------------------------------------
// k_finalize: avg_entropy = (sum over lattices, in order) / K   (spatial_entropy.py:142-156)
// ------------------------------------------------------------------------------------------
__global__ void k_finalize(const double* __restrict__ ent_k, int K, long rows, double* __restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < rows; i += (long)gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += ent_k[(long)k * rows + i];
        out[i] = s / (double)K;
    }
}

// the same for a batch of videos: per-lattice values in [K][rows] (the videos' frames back to back, video v's from
// frame0[v]), the mean goes to every video's own output
__global__ void k_finalize_batch(const double* __restrict__ ent_k, int K, long rows, const long* __restrict__ frame0,
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
__global__ void k_finalize_list(const double* __restrict__ ent_k, int K, long rows, const uint32_t* __restrict__ list,
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
