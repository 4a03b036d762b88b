// vet_layout.hpp — table and histogram layout constants shared by the host code and the kernels (no kernels here)
// Part of the gfx950 device code of the viewport -> tile -> entropy path (see vet_kernels.hpp for the map).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vet {

constexpr int WAVE = 64;
constexpr int MAX_LATTICES = 8;

// direction weight table (vet_weight_table.hpp)
constexpr int TAB_X = 16;                                // histogram unit 2^-(32+TAB_X): sums of < 2^16 weights <= 1 fit 64 bits
constexpr uint32_t MARKER_BITS = 0x80000000u;            // -0.0f: FP-table entry of an in-FoV tile without an FP32 value
constexpr int ROW_BITS = 19;                             // set key = row (19 bits) | mirror flag; slot = key << 12 | count
constexpr uint32_t ROW_MASK = (1u << ROW_BITS) - 1;
constexpr unsigned DEDUP_MAX_DIRS = (1u << ROW_BITS) - 1;

// ------------------------------------------------------------------------------------------
// Fused histogram layout.  The plan's K lattices (analyzers/spatial_entropy.py:142-156 loops over
// them per frame) share ONE histogram of N = Nr + 4K slots, Nr = 2 * (Hs + K), Hs = sum_k floor(n_k / 2):
//     [ 2K spare slots | first halves of lattices 0..K-1 | K centre slots | K mirrored centre slots |
//       second halves, reversed | 2K spare slots ]
// laid out so that the ONE reflection pos -> N-1-pos maps every lattice onto itself the way the Fibonacci
// lattice's mirror symmetry (x,y,z) -> (x,-y,-z) does (tile i <-> tile n_k-1-i, see ensure_alias): a direction
// and its mirror image then share one fused table row, the mirrored one adding into N-1-pos.  The centre tile
// of an odd lattice is its own mirror image; it owns two slots and the epilogue adds them.
// A distinct direction of a frame costs ONE row walk (one length word, one set-up) whatever K is, and the
// short rows of small lattices share cache lines with the others (config 4, 51+101+201 tiles: 88 entries =
// 5 lines instead of 3 rows x 3 lines + 2 meta words).
// ------------------------------------------------------------------------------------------
struct FusedLayout {
    int K;
    int n[MAX_LATTICES];        // tiles per lattice
    int off[MAX_LATTICES];      // slot of tile 0 of lattice k = 2K + sum_{j<k} floor(n_j / 2)
    int Hs, N;                  // N = 2 * (Hs + K) + 4K
    int CF;                     // 64-tile chunks per frame = sum_k ceil(n_k / 64)
    double hmax[MAX_LATTICES];
};
__host__ __device__ __forceinline__ int fused_pos(const FusedLayout& L, int k, int i) {
    const int h = L.n[k] >> 1;
    if (i < h) return L.off[k] + i;
    if (i >= L.n[k] - h) return L.N - 1 - (L.off[k] + (L.n[k] - 1 - i));
    return 2 * L.K + L.Hs + k;
}

}  // namespace vet
