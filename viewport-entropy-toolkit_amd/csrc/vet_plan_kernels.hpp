// vet_plan_kernels.hpp — per-plan tables: k_grid_dirs, k_unit_dirs, k_nearest_lut, k_angular_distances
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
// k_nearest_lut: find_nearest_tile (entropy_utils.py:89-106) for every direction of the table:
// np.argmin over arccos(clip(dot)) — the FIRST minimum, i.e. the lowest index among the tiles whose
// distance VALUE is the smallest.  arccos is monotone, so the arg-max of the cosine ('>' keeps the
// lowest index on exact ties) finds a tile of minimal distance; where the cosines of two tiles differ by
// a few ulp, arccos may map them to one double, and the reference then keeps the lower index: the second
// pass applies that rule literally to the (rare) tiles of lower index within 8 ulp of the best cosine.
// lane = direction, the tiles walk through LDS (broadcast).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double clip_unit(double c) {        // np.clip(c, -1, 1): NaN stays NaN
    return c != c ? c : fmin(fmax(c, -1.0), 1.0);
}

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
        const double near = best - 8.0 * 2.220446049250313e-16, dbest = acos(clip_unit(best));
        for (int t = 0; t < bi; ++t) {
            const double c = fma(z, s_tiles[3 * t + 2], fma(y, s_tiles[3 * t + 1], x * s_tiles[3 * t]));
            if (c >= near && acos(clip_unit(c)) <= dbest) { bi = t; break; }
        }
        nearest[d] = (uint16_t)bi;
    }
}

// ------------------------------------------------------------------------------------------
// k_angular_distances: vector_angle_distance / find_angular_distances (entropy_utils.py:41-87) for m
// vectors x n tile centres: both vectors re-normalised (v / ||v||), dot, clip to [-1, 1], arccos.  The
// arithmetic is the one k_nearest_lut and the weight kernels use (sqrt of the sum of squares, divisions,
// a fused dot, ocml acos); a zero-length vector gives NaN as numpy's 0/0 does.  raw xyz in, radians out.
// ------------------------------------------------------------------------------------------
__global__ void k_angular_distances(const double* __restrict__ vecs, long m, const double* __restrict__ tiles, int n,
                                    double* __restrict__ out) {
    const long total = m * (long)n;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long v = i / n;
        const int t = (int)(i - v * n);
        const double vx = vecs[3 * v], vy = vecs[3 * v + 1], vz = vecs[3 * v + 2];
        const double tx = tiles[3 * t], ty = tiles[3 * t + 1], tz = tiles[3 * t + 2];
        const double lv = sqrt(vx * vx + vy * vy + vz * vz), lt = sqrt(tx * tx + ty * ty + tz * tz);
        const double c = fma(vz / lv, tz / lt, fma(vy / lv, ty / lt, (vx / lv) * (tx / lt)));
        out[i] = acos(clip_unit(c));
    }
}

}  // namespace vet
