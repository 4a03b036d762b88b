// vet_geometry.hpp — k_fb_boundaries: tile boundary edges of a Fibonacci tiling
// Part of the gfx950 device code of the viewport -> tile -> entropy path (see vet_kernels.hpp for the map).
// Reference citations are relative to /root/reference/src/viewport_entropy_toolkit/.
#pragma once
#include "vet_common.hpp"

namespace vet {

// ------------------------------------------------------------------------------------------
// k_fb_boundaries — get_fb_tile_boundaries (utilities/data_utils.py:58-189): the boundary edges of every
// tile of a Fibonacci tiling.  For tile i: chords c_i - c_j to all other centres; the neighbours are the
// centres closer than 1.7 x the nearest one, in (chord, index) order; for each neighbour j the bisecting
// great circles of the other neighbours cut j's bisector in points of which the two nearest to c_i (chords
// rounded to 4 decimals, stable order) are an edge candidate, kept if the corner where THOSE two bisectors
// meet lies farther from c_i than the midpoint of (c_i, c_j).  One thread per tile (the reference's loop is
// O(n^2) Python per tiling, 20 s at 1001 tiles); FP64 with the reference's operation order (no contraction).
// ------------------------------------------------------------------------------------------
constexpr int FB_MAX_NEIGHBOURS = 32;

struct V3 { double x, y, z; };
__device__ __forceinline__ V3 v3_sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 v3_add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ double v3_norm(V3 a) { return sqrt((a.x * a.x + a.y * a.y) + a.z * a.z); }
__device__ __forceinline__ V3 v3_unit(V3 a) { const double l = v3_norm(a); return {a.x / l, a.y / l, a.z / l}; }
__device__ __forceinline__ V3 v3_cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ double round4(double v) { return rint(v * 1e4) / 1e4; }

// the two great circles with normals na, nb meet in +-p; the one nearer to c (chords rounded to 4 decimals,
// ties to -p: find_nearest_point, data_utils.py:483-503) and its rounded chord
__device__ __forceinline__ V3 gc_point_near(V3 na, V3 nb, V3 c, double* chord) {
    const V3 p = v3_unit(v3_cross(v3_unit(na), v3_unit(nb)));
    const V3 q = {-p.x, -p.y, -p.z};
    const double l1 = round4(v3_norm(v3_sub(c, p))), l2 = round4(v3_norm(v3_sub(c, q)));
    if (l1 < l2) { *chord = l1; return p; }
    *chord = l2;
    return q;
}

__global__ void k_fb_boundaries(const double* __restrict__ tiles, int n, int max_edges, double* __restrict__ edges,
                                int32_t* __restrict__ count, int32_t* __restrict__ err) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3 ci = {tiles[3 * i], tiles[3 * i + 1], tiles[3 * i + 2]};
    auto chord_to = [&](int j) { return v3_sub(ci, V3{tiles[3 * j], tiles[3 * j + 1], tiles[3 * j + 2]}); };
    double smallest = 1e300;
    for (int j = 0; j < n; ++j)
        if (j != i) smallest = fmin(smallest, v3_norm(chord_to(j)));
    int nb[FB_MAX_NEIGHBOURS];
    double nl[FB_MAX_NEIGHBOURS];
    int m = 0;
    bool overflow = false;
    const double limit = smallest * 1.7;
    for (int j = 0; j < n; ++j) {
        if (j == i) continue;
        const double l = v3_norm(chord_to(j));
        if (!(l >= limit)) {                       // the reference stops at the first neighbour with length >= limit
            if (m == FB_MAX_NEIGHBOURS) { overflow = true; break; }
            int k = m++;                           // stable insertion by chord length
            while (k > 0 && nl[k - 1] > l) { nl[k] = nl[k - 1]; nb[k] = nb[k - 1]; --k; }
            nl[k] = l; nb[k] = j;
        }
    }
    int ne = 0;
    for (int a = 0; a < m && !overflow; ++a) {
        const int j = nb[a];
        const V3 gj = chord_to(j);
        // the two intersections nearest to c_i, in (rounded chord, neighbour order) order
        double c1 = 1e300, c2 = 1e300;
        V3 p1 = {0, 0, 0}, p2 = {0, 0, 0};
        int k1 = -1, k2 = -1, hits = 0;
        for (int b = 0; b < m; ++b) {
            if (b == a) continue;
            double ch;
            const V3 pt = gc_point_near(gj, chord_to(nb[b]), ci, &ch);
            ++hits;
            if (ch < c1) { c2 = c1; p2 = p1; k2 = k1; c1 = ch; p1 = pt; k1 = nb[b]; }
            else if (ch < c2) { c2 = ch; p2 = pt; k2 = nb[b]; }
        }
        if (hits < 2 || k1 < 0 || k2 < 0) continue;
        double corner;
        (void)gc_point_near(chord_to(k1), chord_to(k2), ci, &corner);
        const V3 cj = {tiles[3 * j], tiles[3 * j + 1], tiles[3 * j + 2]};
        V3 mid = {(ci.x + cj.x) / 2, (ci.y + cj.y) / 2, (ci.z + cj.z) / 2};
        mid = v3_unit(mid);
        if (corner > round4(v3_norm(v3_sub(ci, mid)))) {
            if (ne == max_edges) { overflow = true; break; }
            double* e = edges + ((size_t)i * max_edges + ne) * 6;
            e[0] = p1.x; e[1] = p1.y; e[2] = p1.z; e[3] = p2.x; e[4] = p2.y; e[5] = p2.z;
            ++ne;
        }
    }
    count[i] = ne;
    if (overflow) atomicAdd(err, 1);
}

}  // namespace vet
