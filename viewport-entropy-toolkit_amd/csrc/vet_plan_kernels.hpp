// vet_plan_kernels.hpp — per-plan tables: k_grid_dirs, k_unit_dirs, the alias table (k_alias_*, k_canon_*), k_nearest_lut, k_angular_distances
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
// Alias table on the device (ensure_alias, vet_plan.hip): directions with the same Vector share a table row, and so does
// a direction with its mirror image (x,-y,-z).  The host form of rounds 2-5 (an unordered_map over all directions) took
// 6 ms of a first call on a 200 x 400 grid and 1.5 s on 3840 x 1920; this one is a few launches.
//   k_alias_insert   open-addressing set of direction INDICES keyed by the Vector's value (-0.0 == 0.0); a slot keeps
//                    the smallest index of its class (= the first appearance, as the host map did)
//   k_alias_lookup   alias[d] = the class's smallest index
//   k_alias_mirror   a canonical direction whose mirror image's class has a smaller index takes that row, mirrored
//   k_alias_resolve  members of a class follow their canonical direction's (possibly mirrored) row
//   k_canon_count / k_canon_scan / k_canon_fill: the canonical directions, densely numbered in ascending order (row ids)
// ------------------------------------------------------------------------------------------
constexpr uint32_t ALIAS_MIRROR = 0x80000000u;
constexpr int CANON_TILE = 2048;                               // directions per workgroup of the compaction (256 threads x 8)

__device__ __forceinline__ void vec_key(const double* raw, long d, bool mirrored, unsigned long long k[3]) {
    const double x = raw[3 * d] + 0.0, y = (mirrored ? -raw[3 * d + 1] : raw[3 * d + 1]) + 0.0,
                 z = (mirrored ? -raw[3 * d + 2] : raw[3 * d + 2]) + 0.0;          // + 0.0: -0.0 -> +0.0
    k[0] = (unsigned long long)__double_as_longlong(x);
    k[1] = (unsigned long long)__double_as_longlong(y);
    k[2] = (unsigned long long)__double_as_longlong(z);
}
__device__ __forceinline__ unsigned long long vec_hash(const unsigned long long k[3]) {
    unsigned long long h = k[0] * 0x9E3779B97F4A7C15ull;
    h ^= (k[1] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2));
    h ^= (k[2] * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2));
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    return h;
}
// slot of the class of key k in the set, or -1 (only after every insert has completed)
__device__ __forceinline__ long alias_find(const uint32_t* set, unsigned long long mask, const double* raw,
                                           const unsigned long long k[3]) {
    for (unsigned long long sl = vec_hash(k) & mask;; sl = (sl + 1) & mask) {
        const uint32_t cur = set[sl];
        if (cur == EMPTY_KEY) return -1;
        unsigned long long c[3];
        vec_key(raw, (long)cur, false, c);
        if (c[0] == k[0] && c[1] == k[1] && c[2] == k[2]) return (long)sl;
    }
}

__global__ void k_alias_insert(const double* __restrict__ raw, long D, uint32_t* __restrict__ set, unsigned long long mask) {
    for (long d = blockIdx.x * (long)blockDim.x + threadIdx.x; d < D; d += (long)gridDim.x * blockDim.x) {
        unsigned long long k[3];
        vec_key(raw, d, false, k);
        for (unsigned long long sl = vec_hash(k) & mask;; sl = (sl + 1) & mask) {
            const uint32_t cur = atomicCAS(&set[sl], EMPTY_KEY, (uint32_t)d);
            if (cur == EMPTY_KEY) break;                       // new class
            unsigned long long c[3];
            vec_key(raw, (long)cur, false, c);                 // any member of the slot's class has the class's key
            if (c[0] == k[0] && c[1] == k[1] && c[2] == k[2]) { atomicMin(&set[sl], (uint32_t)d); break; }
        }
    }
}

__global__ void k_alias_lookup(const double* __restrict__ raw, long D, const uint32_t* __restrict__ set, unsigned long long mask,
                               uint32_t* __restrict__ alias) {
    for (long d = blockIdx.x * (long)blockDim.x + threadIdx.x; d < D; d += (long)gridDim.x * blockDim.x) {
        unsigned long long k[3];
        vec_key(raw, d, false, k);
        alias[d] = set[alias_find(set, mask, raw, k)];
    }
}

__global__ void k_alias_mirror(const double* __restrict__ raw, long D, const uint32_t* __restrict__ set, unsigned long long mask,
                               uint32_t* __restrict__ alias) {
    for (long d = blockIdx.x * (long)blockDim.x + threadIdx.x; d < D; d += (long)gridDim.x * blockDim.x) {
        if (alias[d] != (uint32_t)d) continue;                 // canonical directions only (they are not written by others)
        unsigned long long k[3];
        vec_key(raw, d, true, k);
        const long sl = alias_find(set, mask, raw, k);
        if (sl >= 0 && set[sl] < (uint32_t)d) alias[d] = set[sl] | ALIAS_MIRROR;
    }
}

// alias_in: after k_alias_mirror; alias_out[d] = canonical direction of d's row | mirrored
__global__ void k_alias_resolve(const uint32_t* __restrict__ alias_in, long D, uint32_t* __restrict__ alias_out) {
    for (long d = blockIdx.x * (long)blockDim.x + threadIdx.x; d < D; d += (long)gridDim.x * blockDim.x) {
        const uint32_t a = alias_in[d];
        alias_out[d] = (!(a & ALIAS_MIRROR) && a != (uint32_t)d) ? alias_in[a] : a;
    }
}

// canonical directions (alias[d] == d) per tile of CANON_TILE directions
__global__ __launch_bounds__(256) void k_canon_count(const uint32_t* __restrict__ alias, long D, uint32_t* __restrict__ tile_count) {
    __shared__ int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    const long base = (long)blockIdx.x * CANON_TILE;
    int mine = 0;
    for (int i = threadIdx.x; i < CANON_TILE; i += 256) {
        const long d = base + i;
        mine += (d < D && alias[d] == (uint32_t)d) ? 1 : 0;
    }
    mine = wave_sum(mine);
    if (lane_id() == 0) atomicAdd(&s_cnt, mine);
    __syncthreads();
    if (threadIdx.x == 0) tile_count[blockIdx.x] = (uint32_t)s_cnt;
}
// exclusive scan of the tile counts by ONE workgroup (a 3840 x 1920 grid has 3 603 tiles); total -> tile_off[n_tiles]
__global__ __launch_bounds__(1024) void k_canon_scan(const uint32_t* __restrict__ tile_count, int n_tiles, uint32_t* __restrict__ tile_off) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_carry;
    if (threadIdx.x == 0) s_carry = 0u;
    __syncthreads();
    for (int base = 0; base < n_tiles; base += 1024) {
        const int i = base + (int)threadIdx.x;
        const uint32_t v = i < n_tiles ? tile_count[i] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o, WAVE);
            if (lane_id() >= o) incl += up;
        }
        if (lane_id() == WAVE - 1) s_wave[wave_id()] = incl;
        __syncthreads();
        uint32_t before = s_carry;
        for (int w = 0; w < wave_id(); ++w) before += s_wave[w];
        if (i < n_tiles) tile_off[i] = before + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) s_carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) tile_off[n_tiles] = s_carry;
}
// canon[row] = direction, rowid[direction] = row for the canonical directions, in ascending order of the direction
__global__ __launch_bounds__(256) void k_canon_fill(const uint32_t* __restrict__ alias, long D, const uint32_t* __restrict__ tile_off,
                                                    int* __restrict__ canon, uint32_t* __restrict__ rowid) {
    __shared__ uint32_t s_wave[4];
    const long base = (long)blockIdx.x * CANON_TILE;
    uint32_t run = tile_off[blockIdx.x];
    for (int i0 = 0; i0 < CANON_TILE; i0 += 256) {             // 256 consecutive directions per step: order preserved
        const long d = base + i0 + threadIdx.x;
        const bool is = d < D && alias[d] == (uint32_t)d;
        const unsigned long long m = __ballot(is);
        if (lane_id() == 0) s_wave[wave_id()] = (uint32_t)__popcll(m);
        __syncthreads();
        uint32_t before = run;
        for (int w = 0; w < wave_id(); ++w) before += s_wave[w];
        if (is) {
            const uint32_t r = before + (uint32_t)below(m);
            canon[r] = (int)d;
            rowid[d] = r;
        }
        run += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
}
// rowsel[d] = row of d's canonical direction | mirrored
__global__ void k_alias_rows(const uint32_t* __restrict__ alias, long D, const uint32_t* __restrict__ rowid, uint32_t* __restrict__ rowsel) {
    for (long d = blockIdx.x * (long)blockDim.x + threadIdx.x; d < D; d += (long)gridDim.x * blockDim.x) {
        const uint32_t a = alias[d];
        rowsel[d] = rowid[a & ~ALIAS_MIRROR] | (a & ALIAS_MIRROR);
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
