// vet_weight_table.hpp — the direction weight table: k_row_stats (error bounds), k_wtab (ELL rows), k_dirrec (per-direction records)
// Part of the gfx950 device code of the viewport -> tile -> entropy path (see vet_kernels.hpp for the map).
// Reference citations are relative to /root/reference/src/viewport_entropy_toolkit/.
#pragma once
#include "vet_weights.hpp"

namespace vet {

// ------------------------------------------------------------------------------------------
// Direction weight table (ELL).  The sample domain is discrete — (W+1)(H+1) pixel directions —
// and the weight of a (direction, tile) pair depends on nothing else, so for videos with more
// samples than directions the rows  {(tile, w)} : w > 0  are evaluated once per plan (exact
// ocml acos / pow, any fov and power) and the per-frame histogram becomes a gather of rows:
//   hist[t] += count(d) * w(d, t)   for the distinct directions d of the frame's users and the ~n/4
//   tiles in d's FoV.
// Row d lives at w[d*stride .. ] (u32 mantissas) and idx[d*stride ..] (u16 tile), sorted by tile,
// zero padded.  Block floating point per ROW: with e = ceil(log2(largest weight of the row)) clamped
// to [-TAB_X, 0], entry = rint(w * 2^(32 - e)) (saturating), and the gather adds
// entry * (count << (TAB_X + e)) to a 64-bit histogram in units of 2^-(32 + TAB_X): a row whose
// weights are all small (narrow FoV, large power) keeps 32 significant bits below its own maximum
// instead of below 1.0.  meta[d] = entries in use | (TAB_X + e) << 16.
//
// k_row_stats (once per lattice, before the first weighted run) evaluates every row exactly and
// decides whether integer histograms are inside the 1e-6 relative contract for EVERY possible frame:
// with absolute step q_d on the entries of row d, k_d entries, exact row sum S_d and row entropy H_d,
//   |dH| <= 36.5 * sum_i c_i k_i q_i / S   and   H >= sum_i c_i S_i H_i / S   (entropy is concave)
// for a frame made of rows i with multiplicities c_i, hence  |dH| / H <= max_d 36.5 q_d k_d / (S_d H_d).
// Plans where that bound exceeds 1e-7 (rows with a single tile in the FoV, weights spanning many
// orders of magnitude) take the FP64 formulation (k_spatial_w<PRECISE>) instead.
// k_wtab<false> finds the longest row (conservative cone test), k_wtab<true> fills the rows.
// One wave per direction; lane = tile.
// ------------------------------------------------------------------------------------------
// (TAB_X, the histogram unit 2^-(32+TAB_X), and MARKER_BITS live in vet_layout.hpp)
//
// Key sets.  Every tile with distance < max is a key of the reference's per-frame dict (entropy_utils.py:131-136), however
// small its weight.  Integer tables therefore never store 0 for an in-FoV tile: a mantissa that rounds to 0 is stored as 1
// (the entry is then off by less than q instead of q/2; k_row_stats counts such entries twice), so "histogram slot != 0"
// is exactly "key of the reference's dict" and the epilogues need no key bitmap.  FP tables do the same with the smallest
// FP32 subnormal where that cannot move any frame's entropy by 1e-7 relative (rows whose second largest weight is within
// 2^-60 of the largest: FORCE_OK below) and keep a marker entry otherwise.

// The reference's NaN frames (entropy_utils.py:131-135, 195-198).  Every tile with distance < max is a key of the
// reference's per-frame dict, also when ((max - d) / max) ** power underflows to exactly 0.0; a key whose summed
// weight w gives fl(w / total) == 0 makes the frame's entropy NaN (0 * log2 0).  With total <= users * tiles < 2^26
// that needs w < 2^-1048: weights below ULTRA_TINY ("ultra-tiny", exact zeros included) are the only ones that can
// do it.  Plans that have such (direction, tile) pairs (k_row_stats counts them) never use an integer formulation;
// their FP table keeps every in-FoV tile whose FP32 entry would be zero — ultra-tiny or merely underflowing below
// the row's scale — as a MARKER entry (-0.0f: adds nothing), the walk records marker hits in a per-frame bitmap,
// and a frame with a marked tile whose histogram stayed 0.0 is handed to the precise sweep (exact FP64 weights and
// the exact key set), which decides NaN / not NaN as the reference does.
#define VET_ULTRA_TINY 0x1p-1048

struct StatsParams {
    const double* dir_unit;
    long D;
    const double* tiles;
    int n;
    double cos_cull;
    WeightCfg wc;
    uint8_t* row_s;             // [D+1] TAB_X + e per row (row D = the all-zero row)
    uint16_t* row_e;            // [D+1] E = -(binary exponent of the row's largest weight), unclamped (FP table), | ROW_E_FORCE_OK
    unsigned long long* crit;   // [3] bit patterns of non-negative doubles (atomicMax):
                                //   [0] max_d 36.5 q_d k_d / (S_d H_d)   with the table's q_d = 2^(e_d - 33)
                                //   [1] max_d 36.5 k_d / (S_d H_d)       (times the sweep's step, 2^(shift-53))
                                //   [2] number of in-FoV (direction, tile) pairs with an ultra-tiny weight (atomicAdd)
    // fused-table pass: rows r -> direction canon[r], the block-floating-point shift is the ROW's shared one
    // (shift_in[r], k_fuse_shifts) instead of the lattice's own; only crit[0] is produced
    const int* canon;
    const uint8_t* shift_in;
};

// row_e bit 15: the row's second largest weight is within 2^-60 of its largest.  Such a row has an entropy of at least
// ~2^-70, against which an in-FoV weight below FP32 range, stored as the smallest subnormal (2^-149 of the row's scale),
// moves no frame's entropy by more than 1e-30 relative (entropy is concave: a frame's entropy is at least the weighted
// mean of its rows').  Rows without the flag keep such tiles as marker entries (resolved by the precise sweep).
constexpr uint16_t ROW_E_FORCE_OK = 0x8000u;

__global__ void k_row_stats(const StatsParams p) {
    const int lane = lane_id();
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    double worst_tab = 0.0, worst_sweep = 0.0;
    int ultra = 0;
    for (long r = wave; r < p.D; r += nwaves) {
        const long d = p.canon ? (long)p.canon[r] : r;
        const double dx = p.dir_unit[3 * d], dy = p.dir_unit[3 * d + 1], dz = p.dir_unit[3 * d + 2];
        double S = 0.0, L = 0.0, mx = 0.0, mx2 = 0.0;      // mx2: second largest weight of the row
        int k = 0, k33 = 0;
        for (int t0 = 0; t0 < p.n; t0 += WAVE) {
            const int t = t0 + lane;
            const bool valid = t < p.n;
            const int ts = valid ? t : 0;
            const double c = fma(dz, p.tiles[3 * ts + 2], fma(dy, p.tiles[3 * ts + 1], dx * p.tiles[3 * ts]));
            if (valid && c > p.cos_cull) {
                double wt;
                if (fov_weight_cone(c, p.wc, wt) && wt < VET_ULTRA_TINY) ++ultra;
                if (wt > 0.0) {
                    ++k; S += wt; L += wt * log2(wt);
                    mx2 = fmax(mx2, fmin(mx, wt)); mx = fmax(mx, wt);
                    // a mantissa that would round to 0 is stored as 1 (k_wtab): whatever the row's shift, that needs
                    // wt < 2^-33; counted conservatively as one more entry each (off by < q instead of <= q/2)
                    if (wt < 0x1p-33) ++k33;
                }
            }
        }
        k = wave_sum(k); k33 = wave_sum(k33); S = wave_sum(S); L = wave_sum(L);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {                  // top two of the wave
            const double o1 = __shfl_xor(mx, o, WAVE), o2 = __shfl_xor(mx2, o, WAVE);
            mx2 = fmax(fmin(mx, o1), fmax(mx2, o2));
            mx = fmax(mx, o1);
        }
        int e = 0;
        if (mx > 0.0) (void)frexp(mx, &e);                  // mx <= 2^e
        if (lane == 0 && !p.canon)                          // 2^E stays finite; weights below 2^-1048 are markers
            p.row_e[d] = (uint16_t)(min(1000, max(0, -e)) | (mx2 >= mx * 0x1p-60 && mx2 > 0.0 ? ROW_E_FORCE_OK : 0));
        e = min(0, max(-TAB_X, e));
        if (lane == 0 && !p.canon) p.row_s[d] = (uint8_t)(TAB_X + e);
        if (p.canon) e = (int)p.shift_in[r] - TAB_X;
        if (k >= 1) {
            // row entropy -sum (w/S) log2(w/S) = log2 S - (sum w log2 w) / S; its own rounding error (~1e-15)
            // only matters where the bound is hopeless anyway
            const double H = k >= 2 ? fmax(log2(S) - L / S, 0.0) : 0.0;
            // entries are off by at most q/2 each — except a weight of exactly 1.0 at shift 0, whose mantissa 2^32
            // saturates to 2^32 - 1 (k_wtab): that one entry is off by q, counted here as one more entry
            const double base = H > 0.0 ? 36.5 * (double)(k + k33 + (mx >= 1.0 ? 1 : 0)) / (S * H) : __builtin_inf();
            worst_tab = fmax(worst_tab, base * ldexp(1.0, e - 33));
            worst_sweep = fmax(worst_sweep, base);
        }
    }
    ultra = wave_sum(ultra);
    if (lane == 0) {
        if (ultra) atomicAdd(&p.crit[2], (unsigned long long)ultra);
        if (p.row_s && !p.canon && wave == 0) { p.row_s[p.D] = (uint8_t)TAB_X; p.row_e[p.D] = 0; }
        if (worst_tab > 0.0) atomicMax(&p.crit[0], (unsigned long long)__double_as_longlong(worst_tab));
        if (worst_sweep > 0.0) atomicMax(&p.crit[1], (unsigned long long)__double_as_longlong(worst_sweep));
    }
}

struct WtabParams {
    const double* dir_unit;
    long D;
    const double* tiles;
    int n;
    double cos_cull;
    WeightCfg wc;
    int stride;
    uint32_t* w;
    uint16_t* idx;
    uint32_t* meta;     // [D+1] entries in use per row | row shift << 16
    const uint8_t* row_s;
    const uint16_t* row_e;
    int fp;             // FP table: entries are FP32 weights scaled by 2^E of their row, meta field = E
    int* markers;       // FP table fill: number of marker entries written (in-FoV tiles kept without a value)
    // Rows are densely numbered: row r belongs to the canonical direction canon[r] (null: row = direction); row_s / row_e
    // are read per DIRECTION when shift_by_dir is set (k_row_stats of a lattice), per row otherwise (fused shifts).
    // fused table (vet_layout.hpp, nl > 0): the row runs over the nl lattices of the plan and an entry's tile index is
    // its slot in the fused histogram.  nl == 0: one lattice (tiles, n), slot = tile.
    const int* canon;
    int shift_by_dir;
    int nl;
    const double* tiles_v[8];
    int n_v[8], off_v[8];
    int Hs, N;
    int* maxcount;
    int gs_log2;        // >= 0: well-filled blocks dealt over the 2^gs_log2 lanes of a gather group
};

// Row layout.  The gather gives every lane of a group of GS = 2^gs_log2 lanes one 16-byte chunk
// (4 slots = 4 components) per block of B = 4*GS entries, and component k of all lanes is added by
// ONE ds_add_u64 instruction.  Measured on MI355X (tools/lds_atomic_probe.hip): the LDS services
// that instruction in four groups of 16 contiguous lanes, one cycle per group when the 16 slots
// differ mod 16 (8-byte slots: bank pair = slot mod 16), one more cycle per extra slot of a class,
// two per extra lane on the same address; lanes of different groups never conflict.  With 16-lane
// gather groups a hardware group is exactly one row, so the cost is decided by the row layout:
// tile-sorted entries dealt 4 per lane put tiles ~14 apart into one instruction (mostly one or two
// classes: ~4 cycles per group).  So in a block that is at least 3/4 full (B = 64, GS = 16) the
// entries are DEALT BY CLASS: the r-th entry of a class (tile mod 16) goes to component r mod 4,
// inside a component to the next free lane; what does not fit (a component's 17th entry) and the
// block's padding fill the remaining (lane, component) places, padding on tiles of classes the
// component does not use.  Typical result: one entry per class and component = conflict-free.
// Emptier blocks (short rows of small lattices, row tails) keep the plain order, where only the
// first lanes of the group have work.
__device__ __forceinline__ bool block_interleaved(int len, int eb, int gs_log2) {
    const int B = 4 << gs_log2;
    return gs_log2 >= 0 && 4 * min(B, len - eb) >= 3 * B;
}
template <bool FILL>
__global__ void k_wtab(const WtabParams p) {
    const int lane = lane_id();
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    int longest = 0, nmark = 0;
    const bool fused = p.nl > 0;
    const int nl = fused ? p.nl : 1;
    const int slots = fused ? p.N : p.n;                    // histogram slots the padding cycles through
    for (long d = wave; d < p.D; d += nwaves) {
        const long dd = p.canon ? (long)p.canon[d] : d;     // the row's direction
        const double dx = p.dir_unit[3 * dd], dy = p.dir_unit[3 * dd + 1], dz = p.dir_unit[3 * dd + 2];
        int count = 0;
        const long si = p.shift_by_dir ? dd : d;
        const int row_shift = FILL ? (p.fp ? (int)(p.row_e[si] & 0x7FFFu) : (int)p.row_s[si]) : 0;
        const bool force_ok = FILL && p.fp && (p.row_e[si] & ROW_E_FORCE_OK) != 0;
        const double scale = p.fp ? ldexp(1.0, row_shift) : ldexp(1.0, 32 + TAB_X - row_shift);        // 2^E / 2^(32 - e)
        for (int l = 0; l < nl; ++l) {
        const double* tiles = fused ? p.tiles_v[l] : p.tiles;
        const int n = fused ? p.n_v[l] : p.n;
        for (int t0 = 0; t0 < n; t0 += WAVE) {
            const int t = t0 + lane;
            const bool valid = t < n;
            const int ts = valid ? t : 0;
            const double c = fma(dz, tiles[3 * ts + 2], fma(dy, tiles[3 * ts + 1], dx * tiles[3 * ts]));
            bool hit = valid && (c > p.cos_cull);
            unsigned w32 = 0u;
            if (FILL) {
                if (hit) {
                    double wt;
                    const bool in_fov = fov_weight_cone(c, p.wc, wt);
                    if (p.fp) {
                        // every in-FoV tile stays a key.  No FP32 value (ultra-tiny weight, or underflow below the row's
                        // scale): the smallest subnormal where that is harmless (FORCE_OK rows, not ultra-tiny), else a
                        // marker — the key survives, the value adds nothing, the precise sweep decides the frame
                        w32 = wt < VET_ULTRA_TINY ? 0u : __float_as_uint((float)(wt * scale));
                        if (in_fov && w32 == 0u) {
                            if (force_ok && wt >= VET_ULTRA_TINY) w32 = 1u;
                            else { w32 = MARKER_BITS; ++nmark; }
                        }
                    } else {
                        // integer mantissa, at least 1 for a tile in the FoV (the key set is the reference's)
                        w32 = (unsigned)fmin(rint(wt * scale), 4294967295.0);
                        if (in_fov && w32 == 0u) w32 = 1u;
                    }
                }
                hit = w32 != 0u;
            }
            const unsigned long long mask = __ballot(hit);
            if (FILL && hit) {
                const int pos = count + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                        __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                int slot = t;
                if (fused) {                                    // fused_pos
                    const int h = n >> 1;
                    slot = t < h ? p.off_v[l] + t : (t >= n - h ? p.N - 1 - (p.off_v[l] + (n - 1 - t)) : 2 * p.nl + p.Hs + l);
                }
                p.w[d * p.stride + pos] = w32;
                p.idx[d * p.stride + pos] = (uint16_t)slot;
            }
            count += __popcll(mask);
        }
        }
        if (FILL) {
            // padding: weight 0 on distinct tiles, so the gather can add every slot unconditionally
            // without piling zero adds onto one LDS address
            for (int pos = count + lane; pos < p.stride; pos += WAVE) {
                p.w[d * p.stride + pos] = 0u;
                p.idx[d * p.stride + pos] = (uint16_t)(pos % slots);
            }
            if (lane == 0) {
                if (p.meta) p.meta[d] = (uint32_t)count | ((uint32_t)row_shift << 16);
            }
            // well-filled blocks of 16- and 8-lane rows: deal the entries by class (one wave pass per block, lane =
            // sorted entry; the loads of all lanes have returned before the first store issues).
            // 16-lane rows (blocks of 64): the r-th entry of a class goes to component r mod 4 — one entry per class and
            // component: a hardware group of 16 lanes is one row and conflict-free.
            // 8-lane rows (blocks of 32; fused rows of 65..96 entries): a component has 8 places, so the classes are
            // split into two halves: the r-th entry of a class of half h goes to component (2 r + h) mod 4 — conflict-
            // free inside the row; the two rows that share a hardware group overlap by chance (the plain order put
            // one class into all eight lanes of a component: entries ~4 slots apart, four per lane).
            if (p.gs_log2 == 4 || p.gs_log2 == 3) {
                const int GSL = p.gs_log2, GS = 1 << GSL, B = 4 * GS;
                const bool active = lane < B;
                for (int eb = 0; eb < count; eb += B) {
                    if (!block_interleaved(count, eb, GSL)) continue;
                    __threadfence_block();
                    const bool real = active && eb + lane < count;
                    uint32_t wv = 0; uint16_t iv = 0;
                    if (real) { wv = p.w[d * p.stride + eb + lane]; iv = p.idx[d * p.stride + eb + lane]; }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const int cls = iv & 15;
                    int r = 0;                                   // rank among the entries of the same class
                    for (int c = 0; c < 16; ++c) {
                        const unsigned long long m = __ballot(real && cls == c);
                        if (real && cls == c) r = below(m);
                    }
                    int comp = (r * (16 >> GSL) + (cls >> GSL)) & 3, q = 0, used[4], freeb[5];
                    for (int k = 0; k < 4; ++k) {                // lane inside the component, GS places each
                        const unsigned long long m = __ballot(real && comp == k);
                        if (real && comp == k) q = below(m);
                        used[k] = min(GS, (int)__popcll(m));
                    }
                    const bool placed = real && q < GS;
                    unsigned usedmask[4];                        // classes present in each component
                    for (int k = 0; k < 4; ++k) {
                        usedmask[k] = 0;
                        for (int c = 0; c < 16; ++c)
                            if (__ballot(placed && comp == k && cls == c)) usedmask[k] |= 1u << c;
                    }
                    freeb[0] = 0;
                    for (int k = 0; k < 4; ++k) freeb[k + 1] = freeb[k] + GS - used[k];
                    const bool leftover = active && !placed;     // entries beyond a component's places, and the padding
                    const unsigned long long lm = __ballot(leftover);
                    if (leftover) {                              // they take the free places in order
                        const int j = below(lm);
                        int k = 0;
                        while (k < 3 && j >= freeb[k + 1]) ++k;
                        const int jj = j - freeb[k];
                        comp = k; q = used[k] + jj;
                        if (!real) {                             // padding: jj-th class the component lacks
                            int seen = 0, c = 0;
                            for (; c < 15; ++c) {
                                if (!((usedmask[k] >> c) & 1u)) { if (seen == jj) break; ++seen; }
                            }
                            iv = (uint16_t)c;                    // slot c has class c (more than 16 slots in such rows)
                        }
                    }
                    if (active) {
                        p.w[d * p.stride + eb + q * 4 + comp] = wv;
                        p.idx[d * p.stride + eb + q * 4 + comp] = iv;
                    }
                }
            }
        }
        longest = max(longest, count);
    }
    // a plain read first: the maximum only grows, so most waves find theirs already covered and skip
    // the same-address atomic (2048 of them cost ~100 us)
    if (!FILL && lane == 0 && longest > *(volatile int*)p.maxcount) atomicMax(p.maxcount, longest);
    if (FILL && p.markers) {
        nmark = wave_sum(nmark);
        if (lane == 0 && nmark) atomicAdd(p.markers, nmark);
    }
    if (FILL && wave == 0) {            // row D: the all-zero row idle lanes of the gather point at
        for (int pos = lane; pos < p.stride; pos += WAVE) {
            p.w[p.D * p.stride + pos] = 0u;
            // lane l of a 16-lane group adds its zeros to tile l: 16 classes, no conflict
            p.idx[p.D * p.stride + pos] = (uint16_t)(p.gs_log2 >= 3 ? (pos >> 2) & 15 : pos % slots);
        }
        if (lane == 0) {
            if (p.meta) p.meta[p.D] = p.fp ? 0u : (uint32_t)TAB_X << 16;
        }
    }
}


// Exact weight rows for the weights pass (tile_weights VALUES at the reference's precision, entropy_utils.py:124-137):
// row r = canonical direction canon[r]: every tile with distance < max, in tile order, with its exact FP64 weight
// ((max - d) / max) ** power — ocml acos / pow on the same fused dot product as everywhere else; weights that underflow
// to 0.0 stay entries (keys of the reference's dict).  One wave per row, lane = tile.
struct WexactParams {
    const double* dir_unit;
    const int* canon;
    long R;
    const double* tiles;
    int n;
    double cos_cull;
    WeightCfg wc;
    int stride;
    uint16_t* idx;      // [R][stride]
    double* w;          // [R][stride]
    uint32_t* len;      // [R]
};
__global__ void k_wexact(const WexactParams p) {
    const int lane = lane_id();
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    for (long r = wave; r < p.R; r += nwaves) {
        const long dd = (long)p.canon[r];
        const double dx = p.dir_unit[3 * dd], dy = p.dir_unit[3 * dd + 1], dz = p.dir_unit[3 * dd + 2];
        int count = 0;
        for (int t0 = 0; t0 < p.n; t0 += WAVE) {
            const int t = t0 + lane;
            const bool valid = t < p.n;
            const int ts = valid ? t : 0;
            const double c = fma(dz, p.tiles[3 * ts + 2], fma(dy, p.tiles[3 * ts + 1], dx * p.tiles[3 * ts]));
            double wt = 0.0;
            const bool in = valid && (c > p.cos_cull) && fov_weight_cone(c, p.wc, wt);
            const unsigned long long mask = __ballot(in);
            if (in) {
                const size_t pos = (size_t)r * p.stride + count + below(mask);
                p.idx[pos] = (uint16_t)t;
                p.w[pos] = wt;
            }
            count += __popcll(mask);
        }
        if (lane == 0) p.len[r] = (uint32_t)count;
    }
}

// Fused rows: one block-floating-point shift per row = the coarsest of the lattices' own shifts (k_row_stats);
// delta[k] = the most any row of lattice k loses against its own shift (the lattice's error bound grows by 2^delta)
__global__ void k_fuse_shifts(const int* __restrict__ canon, int R, int nl, const uint8_t* const* __restrict__ row_s,
                              uint8_t* __restrict__ out, int* __restrict__ delta) {
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < R; r += gridDim.x * blockDim.x) {
        const int d = canon[r];
        int s = 0;
        for (int k = 0; k < nl; ++k) s = max(s, (int)row_s[k][d]);
        out[r] = (uint8_t)s;
        for (int k = 0; k < nl; ++k)
            if (s > (int)row_s[k][d]) atomicMax(&delta[k], s - (int)row_s[k][d]);
    }
}

// Per-direction record of the table kernel's prologue: one 8-byte gather per sample instead of three
// (alias, nearest tile, row meta):  x = row (19 bits) | nearest tile bits 0..11 << 19 | mirrored << 31
//                                     y = meta of the row in lattice 0 (28 bits) | nearest tile bits 12..15 << 28
__global__ void k_dirrec(const uint32_t* __restrict__ alias, const uint16_t* __restrict__ nearest,
                         const uint32_t* __restrict__ meta0, long D, uint2* __restrict__ rec) {
    for (long d = blockIdx.x * (long)blockDim.x + threadIdx.x; d < D; d += (long)gridDim.x * blockDim.x) {
        const uint32_t a = alias[d], near = nearest[d], row = a & 0x7FFFFu;
        rec[d] = make_uint2(row | ((near & 0xFFFu) << 19) | (a & 0x80000000u), (meta0[row] & 0xFFFFFFFu) | ((near >> 12) << 28));
    }
}

}  // namespace vet
