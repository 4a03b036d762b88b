// vet_spatial_lut.hpp — k_spatial_lut: FoV-weighted spatial entropy through the direction weight table (the dominant kernel)
// Part of the gfx950 device code of the viewport -> tile -> entropy path (see vet_kernels.hpp for the map).
// Reference citations are relative to /root/reference/src/viewport_entropy_toolkit/.
#pragma once
#include "vet_common.hpp"

#ifndef VET_STAGE_CYCLES
#define VET_STAGE_CYCLES 0
#endif

namespace vet {

// ------------------------------------------------------------------------------------------
// Row walk shared by the table kernels: the workgroup adds the ELL rows of the frame's distinct
// directions into the LDS histogram hrow.  frows[j] = row << 12 | multiplicity (DEDUP) or the row
// (multiplicity 1); fmeta[j] = the row's meta word.  A group of GS = 2^gs_log2 lanes walks one
// row; UN rows per group are in flight; rows are zero padded, so a group walks to the longest of
// its UN rows only.
// ------------------------------------------------------------------------------------------

// wave-wide sums through DPP (row reductions + row broadcasts): the total in every lane, a fixed order
#define VET_DPP(v, ctrl, rmask) __builtin_amdgcn_update_dpp(0, (v), (ctrl), (rmask), 0xF, true)
__device__ __forceinline__ double wave_total(double v) {
    auto step = [&](auto mov) {
        const long long b = __double_as_longlong(v);
        const int lo = mov((int)(b & 0xFFFFFFFFll)), hi = mov((int)(b >> 32));
        return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
    };
    v += step([](int x) { return VET_DPP(x, 0xB1, 0xF); });      // quad_perm [1,0,3,2]
    v += step([](int x) { return VET_DPP(x, 0x4E, 0xF); });      // quad_perm [2,3,0,1]
    v += step([](int x) { return VET_DPP(x, 0x141, 0xF); });     // row_half_mirror
    v += step([](int x) { return VET_DPP(x, 0x140, 0xF); });     // row_mirror
    v += step([](int x) { return VET_DPP(x, 0x142, 0xA); });     // row_bcast:15 into rows 1, 3
    v += step([](int x) { return VET_DPP(x, 0x143, 0xC); });     // row_bcast:31 into rows 2, 3
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int)(b & 0xFFFFFFFFll), 63), hi = __builtin_amdgcn_readlane((int)(b >> 32), 63);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ unsigned long long wave_total(unsigned long long v) {
    auto step = [&](auto mov) {
        const int lo = mov((int)(v & 0xFFFFFFFFull)), hi = mov((int)(v >> 32));
        return ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
    };
    v += step([](int x) { return VET_DPP(x, 0xB1, 0xF); });
    v += step([](int x) { return VET_DPP(x, 0x4E, 0xF); });
    v += step([](int x) { return VET_DPP(x, 0x141, 0xF); });
    v += step([](int x) { return VET_DPP(x, 0x140, 0xF); });
    v += step([](int x) { return VET_DPP(x, 0x142, 0xA); });
    v += step([](int x) { return VET_DPP(x, 0x143, 0xC); });
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(v & 0xFFFFFFFFull), 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}

// base + (16-bit half of `pair`) * step as an LDS byte address (one VALU instruction); HIGH selects the upper half
template <bool HIGH>
__device__ __forceinline__ uint32_t lds_slot(uint32_t pair, int step, uint32_t base) {
    uint32_t a;
    if (HIGH) asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(a) : "v"(pair), "v"(step), "v"(base));
    else asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(a) : "v"(pair), "v"(step), "v"(base));
    return a;
}
__device__ __forceinline__ void lds_add_u64(uint32_t lds_byte_address, unsigned long long v) {
    __hip_atomic_fetch_add((__attribute__((address_space(3))) unsigned long long*)(size_t)lds_byte_address, v, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_WORKGROUP);
}

// FPT: FP table — entries are FP32 weights (relative precision 2^-24 each: |dH|/H <= 1.2e-7 for every frame, whatever the
// weights' dynamic range), scaled by 2^E of their row; the histogram is FP64 (ds_add_f64) in true units.
// marked / bit0: FP tables with marker entries (vet_weight_table.hpp) — bitmap of the frame's tiles (bit0 = the lattice's
// first bit) that were hit by a marker; null when the launch's tables hold none.
// GSL_IL: lanes per row (log2) of the class-dealt layout this walk is compiled for (16-lane rows; 8-lane fused rows)
template <int UN, bool INTERLEAVED, bool DEDUP, bool FPT, int GSL_IL = 4>
__device__ __forceinline__ void walk_rows(const uint32_t* frows, const uint32_t* fmeta, int nu,
                                          unsigned long long* hrow, int n,
                                          const uint32_t* __restrict__ tab_w, const uint16_t* __restrict__ tab_i,
                                          int stride, int gs_log2_rt, uint32_t zero_row,
                                          uint32_t* marked = nullptr, int bit0 = 0) {
    // Every lane takes one 4-slot chunk per block: one 16-byte load of weights, one 8-byte load of
    // tiles (stride is a multiple of the block, so chunks are 16 / 8 byte aligned), then four
    // unconditional ds_add_u64 of entry * (multiplicity << row shift): padding slots and idle lanes
    // (which walk the all-zero row) add 0 to distinct tiles — no predicates around the adds.
    const int NW = blockDim.x >> 6, lane = lane_id(), wv = wave_id();
    const int gs_log2 = INTERLEAVED ? GSL_IL : gs_log2_rt;  // class-dealt rows: the layout's own group size (ensure_wtab / ensure_fused)
    const int GS = 1 << gs_log2, UPW = WAVE >> gs_log2;
    const int sub = lane >> gs_log2, sl = lane & (GS - 1);
    const int step = NW * UPW;
    // the lane's chunk of block eb holds entries eb + 4*sl + {0..3} (plain block) or eb + sl + GS*{0..3} (class-dealt
    // block: every block with at least 3/4 of its slots in use): it has work while eb < len - cut
    const int cut = INTERLEAVED ? min(4 * sl, 3 * GS - 1) : 4 * sl;
    for (int j0 = wv * UPW; j0 < nu; j0 += UN * step) {
        uint32_t row[UN];                 // entry offsets: a table holds fewer than 2^32 entries (ensure_wtab)
        int lim[UN], sgn[UN];
        char* hb[UN];
        uint32_t mult[UN];
        double scale[UN];
        int longest = 0;
#pragma unroll
        for (int k = 0; k < UN; ++k) {
            const int j = j0 + k * step + sub;
            const bool on = j < nu;
            const uint32_t pk = on ? frows[j] : 0u, m = on ? fmeta[j] : 0u;
            const uint32_t key = DEDUP ? pk >> 12 : pk;
            // a mirrored direction (x,-y,-z) walks its partner's row into the mirrored tiles n-1-t
            const bool flip = ((DEDUP ? key >> ROW_BITS : key >> 31) & 1u) != 0u;
            const uint32_t rid = DEDUP ? key & ROW_MASK : key & 0x7FFFFFFFu;
            row[k] = on ? rid * (uint32_t)stride : zero_row;
            const int len = (int)(m & 0xFFFFu);
            lim[k] = len - cut;
            const uint32_t cnt = DEDUP ? pk & 0xFFFu : (on ? 1u : 0u);
            mult[k] = FPT ? cnt : cnt << ((m >> 16) & 0xFFFu);
            if (FPT) scale[k] = ldexp((double)cnt, -(int)((m >> 16) & 0xFFFu));
            sgn[k] = flip ? -8 : 8;
            hb[k] = (char*)hrow + (flip ? (n - 1) * 8 : 0);
            longest = max(longest, len);
        }
        for (int eb = 0; eb < longest; eb += 4 * GS) {
            // a row that has ended reads the all-zero row (same slots, one hot line) instead of its own padding
            // lines: the gather is bound by cache lines touched (TA/TD busy)
            uint32_t r[UN];
            bool any = false;
#pragma unroll
            for (int k = 0; k < UN; ++k) {
                const bool on = eb < lim[k];
                r[k] = (on ? row[k] : zero_row) + (uint32_t)(eb + 4 * sl);
                any = any || on;
            }
            if (!any) continue;
            uint4 w[UN];
            ushort4 t[UN];
#pragma unroll
            for (int k = 0; k < UN; ++k) {
                w[k] = *(const uint4*)(tab_w + r[k]);
                t[k] = *(const ushort4*)(tab_i + r[k]);
            }
#pragma unroll
            for (int k = 0; k < UN; ++k) {
                if (FPT) {
                    if (marked) {
                        const uint32_t wk[4] = {w[k].x, w[k].y, w[k].z, w[k].w};
                        const int tk[4] = {(int)t[k].x, (int)t[k].y, (int)t[k].z, (int)t[k].w};
#pragma unroll
                        for (int c = 0; c < 4; ++c)
                            if (wk[c] == MARKER_BITS) {
                                const int bit = bit0 + (sgn[k] < 0 ? n - 1 - tk[c] : tk[c]);
                                atomicOr(&marked[bit >> 5], 1u << (bit & 31));
                            }
                    }
                    atomicAdd((double*)(hb[k] + (int)t[k].x * sgn[k]), (double)__uint_as_float(w[k].x) * scale[k]);
                    atomicAdd((double*)(hb[k] + (int)t[k].y * sgn[k]), (double)__uint_as_float(w[k].y) * scale[k]);
                    atomicAdd((double*)(hb[k] + (int)t[k].z * sgn[k]), (double)__uint_as_float(w[k].z) * scale[k]);
                    atomicAdd((double*)(hb[k] + (int)t[k].w * sgn[k]), (double)__uint_as_float(w[k].w) * scale[k]);
                } else {
                    // LDS byte address = base +- 8 * tile in ONE instruction: v_mad_i32_i16 reads the 16-bit tile straight out
                    // of its half of the packed pair (op_sel), no extraction (histograms fit the LDS: fewer than 2^15 slots)
                    const uint32_t hb32 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)hb[k];
                    const uint2 tp = *(const uint2*)&t[k];
                    lds_add_u64(lds_slot<false>(tp.x, sgn[k], hb32), (unsigned long long)w[k].x * mult[k]);
                    lds_add_u64(lds_slot<true>(tp.x, sgn[k], hb32), (unsigned long long)w[k].y * mult[k]);
                    lds_add_u64(lds_slot<false>(tp.y, sgn[k], hb32), (unsigned long long)w[k].z * mult[k]);
                    lds_add_u64(lds_slot<true>(tp.y, sgn[k], hb32), (unsigned long long)w[k].w * mult[k]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_spatial_lut — compute_spatial_entropy (entropy_utils.py:147-211), FoV-weighted mode, through
// the direction weight table.  FPW frames per workgroup.
// LDS:  hist u64 [FPW][n_sum]  per-frame tile weight sums, units of 2^-(32+TAB_X)
//       hash u32 [FPW][HS]     DEDUP: open-addressing set of the frame's rows, slot = row << 12 | count;
//                              shares the histogram's space (the set is compacted before the first add)
//                              unless the users arrive in several chunks
//       rows u32 [FPW][UC]     the frame's distinct rows (slot words), or one row per present user
//       meta u32 [FPW][UC]     their meta words in the lattice being gathered
//       cnt  i32 [FPW] rows in the chunk, [FPW] users present in the frame
// Prologue: sample -> direction id -> canonical row (alias: directions with the same Vector — the pole
// row, the -180 / -90 remaps — share one row) -> set insert.  Users looking in exactly the same
// direction cost one row walk with a multiplicity instead of one each: 1024 users are ~710 distinct
// rows on the random-walk workload, ~180 on a clustered audience.
// A group of GS = 2^gs_log2 lanes walks one row (16-byte weight + 8-byte tile loads) and adds
// entry * multiplicity into the frame histogram with ds_add_u64; a wave serves 64/GS rows at once and
// two such steps are issued back to back to keep more loads in flight.
// ------------------------------------------------------------------------------------------
struct LutLattice {
    const uint32_t* tab_w;
    const uint16_t* tab_i;
    const uint32_t* tab_meta;
    int stride, gs_log2, n, interleaved;
    double hmax;
    uint32_t zrow;                // index of the table's all-zero row (the number of its rows in use)
};

// One video of a batched launch (vet_spatial_entropy_batch): many short videos share one grid,
// workgroups [block0, block0 + ceil(T / FPW)) belong to the video.
struct VideoDesc {
    const double* mu;
    const double* mv;
    int U, T;
    double* entropy;
    int32_t* assign;
    int32_t* present;
    int FPW, UC, block0, pad_;
};

struct LutParams {
    const VideoDesc* videos;      // null: single video described by the fields below
    int n_videos;
    SampleSrc src;
    int U, T;
    const uint16_t* nearest;      // lattice 0 (assign)
    const uint32_t* alias;        // [n_dirs] direction id -> canonical row | mirrored << 31
    const uint2* dirrec;          // [n_dirs] DEDUP: alias, nearest tile and lattice-0 meta in one 8-byte record (k_dirrec)
    int rec_meta;                 // the record's meta word is that of this launch's first lattice
    int K;                        // lattices handled by this launch (<= MAX_LATTICES)
    int n_sum;                    // sum of n over the K lattices
    LutLattice lat[MAX_LATTICES];
    double* entropy;              // [T] mean over the K lattices, summed in order
    int32_t* assign;
    double* weights;              // lattice 0
    int32_t* present;
    int32_t* status;
    int FPW, UC;
    int sort_words;               // FPT: words of the (row, mirrored) bitmap that orders a frame's distinct rows (0: rank sort)
    int dedup_min_users;          // DEDUP: videos of fewer users keep one row entry per user (as the launch without the set would:
                                  // an FP table's sums then do not depend on which videos share a batch)
    FusedLayout lay;              // FUSED: the launch's one "lattice" is the plan's fused table (n = lay.N slots)
    unsigned long long* timeline; // development builds: [workgroup][6] wall clock (100 MHz) at entry / after the set / lists / walk / entropy, HW id
    unsigned long long* dbg;      // development builds (-DVET_STAGE_CYCLES=1): [4] cycles of thread 0 per stage, summed over the workgroups
    uint32_t* resolve;            // FP tables with marker entries: [0] = number of frames handed to the precise sweep
                                  // (a marked tile whose histogram stayed 0.0), then the frames; null otherwise
};

// ------------------------------------------------------------------------------------------
// Entropy of integer (fixed-point) histograms — the epilogue of every integer table kernel (entropy_utils.py:194-211,
// weighted mode: normaliser log2 n).  One pass per lattice: the exact total S (sums of the 32-bit halves) and
// L = sum v log2 v together, then H = log2 S - L / S: one log2 per tile, one divide per lattice.  Against the
// term-by-term form -sum (v/S) log2 (v/S) the cancellation costs ~60 ulp ABSOLUTE (2e-14); integer formulations only
// run on plans whose every frame has an entropy above ~1e-3 (k_row_stats: the bound 36.5 q k / (2 S H) <= 1e-7 needs
// it: a frame's entropy is at least the smallest of its rows', entropy being concave), i.e. <= 2e-11 relative; a frame
// below 1e-6 is evaluated term by term all the same.  Canonical order: every lane sums its tiles lane, lane + 64, ... in order, then one fixed
// DPP tree -> the same bits whatever the kernel, the frames per workgroup or the launch geometry.
//   val(t): the histogram value of tile t of this lattice;  wrow: this frame's row of the weights output or null.
// ------------------------------------------------------------------------------------------
template <class V>
__device__ __forceinline__ double lattice_entropy_int(int n, V val, double hmax, double* wrow, double inv_unit) {
    const int lane = lane_id();
    unsigned long long hi = 0ull, lo = 0ull;
    double L = 0.0;                             // per lane over its tiles lane, lane + 64, ...; one tree at the end
    for (int t = lane; t < n; t += WAVE) {
        const unsigned long long v = val(t);
        hi += v >> 32; lo += v & 0xFFFFFFFFull;
        const double vd = (double)v;
        if (v != 0ull) L += vd * log2(vd);
        if (wrow) __builtin_nontemporal_store(vd * inv_unit, wrow + t);
    }
    L = wave_total(L);
    hi = wave_total(hi); lo = wave_total(lo);
    const double S = (double)(hi + (lo >> 32)) * 4294967296.0 + (double)(lo & 0xFFFFFFFFull);
    if (!(S > 0.0)) return 0.0;                 // no tile in any user's FoV: the reference sums over an empty dict
    double H = log2(S) - L / S;
    // The one-pass form loses ~2e-14 ABSOLUTE to cancellation.  The plans that run integer formulations keep every frame
    // above ~1e-3 (header); should a frame come out lower all the same (one key: exactly 0 in the reference), it is
    // evaluated again term by term, -sum (v/S) log2 (v/S), which has no cancellation (wave-uniform branch, same order)
    if (H < 1e-6) {
        double h = 0.0;
        for (int t = lane; t < n; t += WAVE) {
            const unsigned long long v = val(t);
            if (v != 0ull) { const double q = (double)v / S; h -= q * log2(q); }
        }
        H = wave_total(h);
    }
    return fmax(H, 0.0) / hmax;
}

// frames fl = wave, wave + NW, ... of the workgroup: K lattices laid end to end in a frame's histogram row
template <class NOf, class HmaxOf>
__device__ __forceinline__ void lut_epilogue_int(const unsigned long long* hist, size_t frame_stride, int nf, long f0,
                                                 int K, NOf n_of, HmaxOf hmax_of, const int* np_of,
                                                 double* entropy, int32_t* present, double* weights, int32_t* status) {
    const int NW = blockDim.x >> 6, lane = lane_id(), wv = wave_id();
    const double inv_unit = 1.0 / (4294967296.0 * (double)(1u << TAB_X));
    for (int fl = wv; fl < nf; fl += NW) {
        const unsigned long long* hrow = hist + (size_t)fl * frame_stride;
        double total_entropy = 0.0;
        for (int k = 0; k < K; ++k) {
            const int n = n_of(k);
            total_entropy += lattice_entropy_int(n, [&](int t) { return hrow[t]; }, hmax_of(k),
                                                 (k == 0 && weights) ? weights + (f0 + fl) * (long)n : nullptr, inv_unit);
            hrow += n;
        }
        if (lane == 0) {
            const int np = np_of[fl];
            double e = total_entropy / (double)K;
            if (np == 0) {
                e = __builtin_nan("");
                if (status) atomicAdd(&status[1], 1);
            }
            entropy[f0 + fl] = e;
            if (present) present[f0 + fl] = np;
        }
    }
}

// the same over a fused histogram (vet_layout.hpp): lattice k's tile t sits in slot fused_pos(k, t); the centre tile of
// an odd lattice owns two slots
__device__ __forceinline__ void lut_epilogue_fused(const unsigned long long* hist, int nf, long f0, const FusedLayout& lay,
                                                   const int* np_of, double* entropy, int32_t* present, double* weights,
                                                   int32_t* status) {
    const int NW = blockDim.x >> 6, lane = lane_id(), wv = wave_id();
    const double inv_unit = 1.0 / (4294967296.0 * (double)(1u << TAB_X));
    const int N = lay.N, K = lay.K;
    for (int fl = wv; fl < nf; fl += NW) {
        const unsigned long long* hrow = hist + (size_t)fl * N;
        double total_entropy = 0.0;
        for (int k = 0; k < K; ++k) {
            const int n = lay.n[k], hh = n >> 1;
            auto val = [&](int t) {
                const int pos = fused_pos(lay, k, t);
                unsigned long long v = hrow[pos];
                if (t >= hh && t < n - hh) v += hrow[N - 1 - pos];          // centre tile: both of its slots
                return v;
            };
            total_entropy += lattice_entropy_int(n, val, lay.hmax[k], (k == 0 && weights) ? weights + (f0 + fl) * (long)n : nullptr,
                                                 inv_unit);
        }
        if (lane == 0) {
            const int np = np_of[fl];
            double e = total_entropy / (double)K;
            if (np == 0) {
                e = __builtin_nan("");
                if (status) atomicAdd(&status[1], 1);
            }
            entropy[f0 + fl] = e;
            if (present) present[f0 + fl] = np;
        }
    }
}

// hash slots per frame: power of two >= 2 * UC, at least one wave's worth
__host__ __device__ __forceinline__ int lut_hash_slots(int UC) {
    int hs = 64;
    while (hs < 2 * UC) hs <<= 1;
    return hs;
}
// LDS bytes of a workgroup; the kernel and the host must agree
__host__ __device__ __forceinline__ int lut_marked_words(int n_sum) { return (n_sum + 31) >> 5; }
// priv: histograms per frame — 1, or one per wave for FP tables (each wave adds its rows of the SORTED row list into its
// own histogram in program order, the histograms are added in wave order: FP64 sums that do not depend on scheduling)
// sort_words: FP tables put every frame's row list into a canonical order through a bitmap over (row, mirrored); the bitmap
// (+ 256 scan words) lives in the histogram / set region too (the histograms are cleared after the sort)
__host__ __device__ __forceinline__ size_t lut_lds_bytes(int U, int UC, int FPW, int n_sum, bool dedup, bool marked = false,
                                                         int priv = 1, int sort_words = 0) {
    const size_t hist = (size_t)FPW * n_sum * 8 * priv, hash = dedup ? (size_t)FPW * lut_hash_slots(UC) * 4 : 0;
    const size_t srt = sort_words ? ((size_t)sort_words + 256) * 4 : 0;
    size_t a = (dedup && U <= UC) ? (hist > hash ? hist : hash) : hist + hash;
    if (dedup && U <= UC && srt > a) a = srt;
    return ((a + 15) & ~(size_t)15) + (size_t)FPW * UC * 8 + (size_t)2 * FPW * 4 + 64 +
           (marked ? (size_t)FPW * lut_marked_words(n_sum) * 4 : 0);
}

// All K lattices of the plan in one launch: the samples are read once, every row is gathered into K
// histograms, and avg_entropy = (e_0 + ... + e_{K-1}) / K is formed in lattice order as the
// reference does (spatial_entropy.py:142-156) — no per-lattice pass, no finalize.
// IL: some lattice of the plan has interleaved rows (otherwise only the plain walk is compiled in).
// OCC8: compiled for 8 workgroups of 256 threads per CU (64 VGPRs) instead of 7 (68-70 VGPRs): measured
// 2 % (random walk) to 6.5 % (clustered) faster on single-lattice plans and 7 % on batches of short
// videos, but 2-4 % slower on one multi-lattice video (profiles/r01/v6_table_occupancy.log).
// DEDUP: per-frame set of distinct rows with multiplicities (direction tables of < 2^20 rows).
// FUSED: the table is the plan's fused one (FusedLayout above): one row per distinct direction over all lattices,
// histogram slots instead of tiles, the exact total of every lattice in its total slots; the epilogue reads the
// lattices back out of the fused histogram.
template <bool FROM_IDS, int UN, bool IL, bool OCC8, bool DEDUP, bool FPT, bool FUSED = false>
__global__ __launch_bounds__(256, OCC8 ? 8 : (FPT ? 6 : 7)) void k_spatial_lut(const LutParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the video this workgroup works on: the launch's only one, or one of a batch
    SampleSrc src = p.src;
    int U = p.U, T = p.T, FPW = p.FPW, UC = p.UC;
    double* entropy = p.entropy;
    int32_t* assign = p.assign;
    int32_t* present = p.present;
    double* weights = p.weights;
    long blk = blockIdx.x;
    if (p.videos) {
        int lo = 0, hi = p.n_videos - 1;                   // last video with block0 <= blockIdx.x
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (p.videos[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        const VideoDesc& d = p.videos[lo];
        src.mu = d.mu; src.mv = d.mv;
        U = d.U; T = d.T; FPW = d.FPW; UC = d.UC;
        entropy = d.entropy; assign = d.assign; present = d.present; weights = nullptr;
        blk -= d.block0;
    }
    const int HS = DEDUP ? lut_hash_slots(UC) : 0;
    const bool merge = DEDUP && U >= p.dedup_min_users;                          // equal rows of a frame share one entry
    const bool overlay = DEDUP && U <= UC;                                       // one chunk: set and histogram share space
    const int PRIV = FPT ? (int)(blockDim.x >> 6) : 1;                            // FP table: one histogram per wave
    const size_t hist_bytes = (size_t)FPW * p.n_sum * 8 * PRIV, hash_bytes = (size_t)FPW * HS * 4;
    const size_t sort_bytes = (FPT && overlay && p.sort_words) ? ((size_t)p.sort_words + 256) * 4 : 0;
    size_t a_bytes = overlay ? (hist_bytes > hash_bytes ? hist_bytes : hash_bytes) : hist_bytes + hash_bytes;
    if (sort_bytes > a_bytes) a_bytes = sort_bytes;
    unsigned long long* hist = (unsigned long long*)smem;                        // [FPW][n_sum]
    uint32_t* hash = (uint32_t*)(smem + (overlay ? 0 : hist_bytes));             // [FPW][HS]
    uint32_t* rows = (uint32_t*)(smem + ((a_bytes + 15) & ~(size_t)15));         // [FPW][UC]
    uint32_t* meta = rows + (size_t)FPW * UC;                                    // [FPW][UC]
    int* cnt_chunk = (int*)(meta + (size_t)FPW * UC);                            // [FPW]
    int* cnt_frame = cnt_chunk + FPW;                                            // [FPW]
    const int MW = lut_marked_words(p.n_sum);
    uint32_t* marked = (FPT && p.resolve) ? (uint32_t*)(cnt_frame + FPW) : nullptr;   // [FPW][MW]
    const int NW = blockDim.x >> 6;
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const long f0 = blk * FPW;
    const int nf = (int)min((long)FPW, (long)T - f0);
#if VET_STAGE_CYCLES
    unsigned long long tdbg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = p.dbg ? __builtin_readcyclecounter() : 0ull, tsub = tlast;
    if (p.timeline && tid == 0) {
        p.timeline[(long)blockIdx.x * 6 + 0] = wall_clock64();
        p.timeline[(long)blockIdx.x * 6 + 5] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) /* HW_ID */ |
                                  ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) /* XCC_ID */ << 32);
    }
    auto stage = [&](int i) {
        if (p.timeline && tid == 0) p.timeline[(long)blockIdx.x * 6 + 1 + i] = wall_clock64();
        if (p.dbg) { const unsigned long long now = __builtin_readcyclecounter(); tdbg[i] += now - tlast; tlast = now; tsub = now; }
    };
    // parts of stage 0 (samples -> set): 4 LDS init + barriers, 5 sample loads (waited for), 6 record gathers (waited for),
    // 7 set inserts + list appends
    auto sub = [&](int i, bool wait_vm) {
        if (p.dbg) {
            if (wait_vm) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long now = __builtin_readcyclecounter(); tdbg[4 + i] += now - tsub; tsub = now;
        }
    };
#else
    auto stage = [](int) {};
    auto sub = [](int, bool) {};
#endif
    if (!overlay)
        for (int i = tid; i < FPW * p.n_sum * PRIV; i += blockDim.x) hist[i] = 0ull;
    for (int i = tid; i < 2 * FPW; i += blockDim.x) cnt_chunk[i] = 0;
    if (FPT && marked)
        for (int i = tid; i < FPW * MW; i += blockDim.x) marked[i] = 0u;
    bool bad = false;
    const int hs_shift = 32 - (31 - __clz(HS | 1));
    for (int u0 = 0; u0 < U; u0 += UC) {
        const int uc = min(UC, U - u0);
        __syncthreads();
        for (int i = tid; i < FPW; i += blockDim.x) cnt_chunk[i] = 0;
        if (merge)
            for (int i = tid; i < FPW * HS; i += blockDim.x) hash[i] = EMPTY_KEY;
        __syncthreads();
        constexpr int SPT = 4;
        const int total = nf * uc;
        // SPT samples per thread and round: all sample loads first, then the table gathers, then the LDS set
        // inserts — three waves of independent requests instead of SPT dependent chains
        sub(0, false);
        for (int i0 = tid; i0 < total; i0 += SPT * (int)blockDim.x) {
            int id[SPT], fls[SPT];
            long idxs[SPT];
            if (FROM_IDS) {
#pragma unroll
                for (int k = 0; k < SPT; ++k) {
                    const int i = i0 + k * (int)blockDim.x;
                    fls[k] = i / uc;
                    idxs[k] = (f0 + fls[k]) * (long)U + u0 + (i - fls[k] * uc);
                    id[k] = -1;
                    if (i < total) {
                        const int v = src.ids[idxs[k]];
                        if (v >= src.n_dirs) bad = true; else if (v >= 0) id[k] = v;
                    }
                }
            } else {
                double a[SPT], b[SPT];
#pragma unroll
                for (int k = 0; k < SPT; ++k) {
                    const int i = i0 + k * (int)blockDim.x;
                    fls[k] = i / uc;
                    idxs[k] = (f0 + fls[k]) * (long)U + u0 + (i - fls[k] * uc);
                    a[k] = b[k] = __builtin_nan("");
                    if (i < total) {
                        a[k] = __builtin_nontemporal_load(src.mu + idxs[k]);
                        b[k] = __builtin_nontemporal_load(src.mv + idxs[k]);
                    }
                }
                sub(1, true);
#pragma unroll
                for (int k = 0; k < SPT; ++k) id[k] = grid_dir(a[k], b[k], src.W, src.H, bad);
            }
            uint32_t row[SPT], m0[SPT];
            int near[SPT];
#pragma unroll
            for (int k = 0; k < SPT; ++k) {
                row[k] = 0u; near[k] = -1; m0[k] = 0u;
                if (id[k] >= 0) {
                    if (DEDUP) {
                        const uint2 rec = p.dirrec[id[k]];
                        row[k] = (rec.x & ROW_MASK) | ((rec.x >> 31) << ROW_BITS);
                        near[k] = (int)(((rec.x >> ROW_BITS) & 0xFFFu) | ((rec.y >> 28) << 12));
                        m0[k] = rec.y & 0xFFFFFFFu;
                    } else if (p.dirrec) {                      // small frames: no set, but the fused record
                        const uint2 rec = p.dirrec[id[k]];
                        row[k] = (rec.x & ROW_MASK) | (rec.x & 0x80000000u);
                        near[k] = (int)(((rec.x >> ROW_BITS) & 0xFFFu) | ((rec.y >> 28) << 12));
                        m0[k] = rec.y & 0xFFFFFFFu;
                    } else {
                        row[k] = p.alias[id[k]];                // canonical row | mirrored << 31
                        if (assign) near[k] = (int)p.nearest[id[k]];
                    }
                }
            }
            sub(2, true);
            if (assign) {
#pragma unroll
                for (int k = 0; k < SPT; ++k)
                    if (i0 + k * (int)blockDim.x < total) __builtin_nontemporal_store(near[k], assign + idxs[k]);
            }
#pragma unroll
            for (int k = 0; k < SPT; ++k) {
                const bool valid = id[k] >= 0;
                const int fl = fls[k];
                // users present per frame / new rows per frame: one LDS atomic per wave where the wave's
                // samples belong to one frame (always when the user count is a multiple of 64)
                const int fl0 = __builtin_amdgcn_readfirstlane(fl);
                const bool uniform = __ballot(fl != fl0) == 0ull;
                bool won = false;
                unsigned h = 0;
                if (merge) {
                    if (valid) {
                        uint32_t* tab = hash + (size_t)fl * HS;
                        h = (row[k] * 2654435761u) >> hs_shift;
                        for (;;) {
                            unsigned cur = tab[h];
                            if (cur == EMPTY_KEY) {
                                cur = atomicCAS(&tab[h], EMPTY_KEY, (row[k] << 12) | 1u);
                                if (cur == EMPTY_KEY) { won = true; break; }
                            }
                            if ((cur >> 12) == row[k]) { atomicAdd(&tab[h], 1u); break; }
                            h = (h + 1) & (unsigned)(HS - 1);
                        }
                    }
                } else {
                    won = valid;
                }
                const unsigned long long mv_ = __ballot(valid), mw = __ballot(won);
                if (uniform) {
                    int base = 0;
                    if (lane == 0) {
                        if (mv_) atomicAdd(&cnt_frame[fl0], (int)__popcll(mv_));
                        if (mw) base = atomicAdd(&cnt_chunk[fl0], (int)__popcll(mw));
                    }
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (won) {
                        const size_t pos = (size_t)fl0 * UC + base + below(mw);
                        rows[pos] = merge ? h : (DEDUP ? (row[k] << 12) | 1u : row[k]);
                        meta[pos] = m0[k];
                    }
                } else {
                    if (valid) atomicAdd(&cnt_frame[fl], 1);
                    if (won) {
                        const size_t pos = (size_t)fl * UC + atomicAdd(&cnt_chunk[fl], 1);
                        rows[pos] = merge ? h : (DEDUP ? (row[k] << 12) | 1u : row[k]);
                        meta[pos] = m0[k];
                    }
                }
            }
        }
        sub(3, false);
        __syncthreads();
        stage(0);
        if (DEDUP) {
            // slot numbers -> slot words (row << 12 | multiplicity)
            if (merge)
                for (int i = tid; i < nf * UC; i += blockDim.x) {
                    const int fl = i / UC, j = i - fl * UC;
                    if (j < cnt_chunk[fl]) rows[i] = hash[(size_t)fl * HS + rows[i]];
                }
            if (overlay && !FPT) {
                __syncthreads();
                for (int i = tid; i < FPW * p.n_sum; i += blockDim.x) hist[i] = 0ull;
            }
        }
        if (FPT) {
            // FP64 sums must not depend on the order in which the users arrived: every frame's row list is put into
            // ascending order, so that row j always goes to the same wave, lane group and turn of the walk below.
            __syncthreads();
            constexpr int EPT = 8;                               // UC <= 2048 rows over 256 threads
            const bool by_bitmap = merge && overlay && p.sort_words > 0;
            for (int fl = 0; fl < nf; ++fl) {
                const int cnt = cnt_chunk[fl];
                uint32_t* fr = rows + (size_t)fl * UC;
                uint32_t* fm = meta + (size_t)fl * UC;
                uint32_t key[EPT];
                int rank[EPT];
#pragma unroll
                for (int q = 0; q < EPT; ++q) {
                    const int e = tid + q * (int)blockDim.x;
                    key[q] = e < cnt ? fr[e] : 0xFFFFFFFFu;
                    rank[q] = 0;
                }
                if (by_bitmap) {
                    // distinct rows: rank = number of set bits below the row's own bit in a bitmap over (row, mirrored)
                    // — O(rows) instead of the O(rows^2) of a rank sort.  The bitmap sits in the histogram / set region
                    // (the set has been read out above; the histograms are cleared after the sort).
                    uint32_t* bm = (uint32_t*)smem;              // [SW]
                    uint32_t* toff = bm + p.sort_words;          // [256] bits set in the words before a thread's span
                    const int SW = p.sort_words, WPT = (SW + (int)blockDim.x - 1) / (int)blockDim.x;
                    for (int i = tid; i < SW; i += blockDim.x) bm[i] = 0u;
                    __syncthreads();
#pragma unroll
                    for (int q = 0; q < EPT; ++q)
                        if (tid + q * (int)blockDim.x < cnt) {
                            const uint32_t k20 = key[q] >> 12, bit = (k20 & ROW_MASK) * 2u + ((k20 >> ROW_BITS) & 1u);
                            atomicOr(&bm[bit >> 5], 1u << (bit & 31));
                        }
                    __syncthreads();
                    int mine = 0;
                    for (int w = tid * WPT; w < min(SW, (tid + 1) * WPT); ++w) mine += __popc(bm[w]);
                    int incl = mine;                             // inclusive scan over the workgroup's threads
#pragma unroll
                    for (int o = 1; o < WAVE; o <<= 1) {
                        const int up = __shfl_up(incl, o, WAVE);
                        if (lane >= o) incl += up;
                    }
                    int* wtot = (int*)(toff + 252);              // the last four scan words double as the waves' totals
                    __syncthreads();
                    if (lane == WAVE - 1) wtot[wv] = incl;
                    __syncthreads();
                    int before = 0;
                    for (int w2 = 0; w2 < wv; ++w2) before += wtot[w2];
                    __syncthreads();
                    toff[tid] = (uint32_t)(before + incl - mine);
                    __syncthreads();
#pragma unroll
                    for (int q = 0; q < EPT; ++q)
                        if (tid + q * (int)blockDim.x < cnt) {
                            const uint32_t k20 = key[q] >> 12, bit = (k20 & ROW_MASK) * 2u + ((k20 >> ROW_BITS) & 1u);
                            const int w = (int)(bit >> 5), t0 = w / WPT;
                            int r = (int)toff[t0];
                            for (int w2 = t0 * WPT; w2 < w; ++w2) r += __popc(bm[w2]);
                            rank[q] = r + __popc(bm[w] & ((1u << (bit & 31)) - 1u));
                        }
                } else {
                    // rank sort: few rows (no set below 128 users: equal rows may repeat, their order cannot matter),
                    // or users in several chunks
                    for (int j = 0; j < cnt; ++j) {
                        const uint32_t other = fr[j];            // broadcast read
#pragma unroll
                        for (int q = 0; q < EPT; ++q) {
                            const int e = tid + q * (int)blockDim.x;
                            rank[q] += (other < key[q] || (other == key[q] && j < e)) ? 1 : 0;
                        }
                    }
                }
                uint32_t mk[EPT];                                // (read only now: eight registers less across the ranking)
#pragma unroll
                for (int q = 0; q < EPT; ++q) mk[q] = tid + q * (int)blockDim.x < cnt ? fm[tid + q * (int)blockDim.x] : 0u;
                __syncthreads();
#pragma unroll
                for (int q = 0; q < EPT; ++q)
                    if (tid + q * (int)blockDim.x < cnt) { fr[rank[q]] = key[q]; fm[rank[q]] = mk[q]; }
                __syncthreads();
            }
            if (overlay) {
                for (int i = tid; i < FPW * p.n_sum * PRIV; i += blockDim.x) hist[i] = 0ull;
                __syncthreads();
            }
        }
        int hoff = 0;
        stage(1);
        for (int k = 0; k < p.K; ++k) {
            const LutLattice& L = p.lat[k];
            // meta words (length, shift) of this lattice for every staged row: one parallel gather, so the
            // walk below has no dependent global load in front of its row loads
            if (k) __syncthreads();
            if (!(k == 0 && p.rec_meta && (DEDUP || p.dirrec)))
                for (int i = tid; i < nf * UC; i += blockDim.x) {
                    const int fl = i / UC, j = i - fl * UC;
                    if (j < cnt_chunk[fl]) meta[i] = L.tab_meta[DEDUP ? (rows[i] >> 12) & ROW_MASK : rows[i] & 0x7FFFFFFFu];
                }
            __syncthreads();
            constexpr int GSL_IL = (FUSED && UN == 2) ? 3 : 4;      // the narrow fused kernel walks 8-lane rows
            for (int fl = 0; fl < nf; ++fl)
                if (IL && L.interleaved)
                    walk_rows<UN, true, DEDUP, FPT, GSL_IL>(rows + (size_t)fl * UC, meta + (size_t)fl * UC, cnt_chunk[fl],
                                               hist + ((size_t)fl * PRIV + (FPT ? wv : 0)) * p.n_sum + hoff, L.n, L.tab_w, L.tab_i, L.stride, L.gs_log2,
                                               L.zrow * (uint32_t)L.stride, marked ? marked + (size_t)fl * MW : nullptr, hoff);
                else
                    walk_rows<UN, false, DEDUP, FPT>(rows + (size_t)fl * UC, meta + (size_t)fl * UC, cnt_chunk[fl],
                                                hist + ((size_t)fl * PRIV + (FPT ? wv : 0)) * p.n_sum + hoff, L.n, L.tab_w, L.tab_i, L.stride, L.gs_log2,
                                                L.zrow * (uint32_t)L.stride, marked ? marked + (size_t)fl * MW : nullptr, hoff);
            hoff += L.n;
        }
    }
    __syncthreads();
    stage(2);
    // entropy (entropy_utils.py:194-211, weighted: normaliser log2 n); wave w takes frames w, w+NW, ...
    if (FUSED) {
        lut_epilogue_fused(hist, nf, f0, p.lay, cnt_frame, entropy, present, weights, p.status);
    } else if (!FPT) {
        // (the lattices' sizes and normalisers straight from the kernel arguments: a uniform index, scalar loads — a
        // private copy indexed by k would live in scratch memory)
        lut_epilogue_int(hist, (size_t)p.n_sum, nf, f0, p.K, [&](int k) { return p.lat[k].n; }, [&](int k) { return p.lat[k].hmax; },
                         cnt_frame, entropy, present, weights, p.status);
    } else
    for (int fl = wv; fl < nf; fl += NW) {
        const unsigned long long* hrow = hist + (size_t)fl * PRIV * p.n_sum;
        // FP table: the tile's weight = the waves' histograms added in wave order
        auto fp_value = [&](const unsigned long long* h, int t) {
            double v = 0.0;
            for (int w = 0; w < PRIV; ++w) v += __longlong_as_double((long long)h[(size_t)w * p.n_sum + t]);
            return v;
        };
        double total_entropy = 0.0;
        bool unresolved = false;           // a key of the reference's dict (marker hit) whose table weight sum is 0.0
        int bit0 = 0;
        for (int k = 0; k < p.K; ++k) {
            const int n = p.lat[k].n;
            // total weight can exceed 64 bits of fixed point: summed in FP64, fixed lane order + butterfly
            double totd = 0.0;
            for (int t = lane; t < n; t += WAVE) totd += fp_value(hrow, t);
            totd = wave_sum(totd);
            double h = 0.0;
            for (int t = lane; t < n; t += WAVE) {
                const double v = fp_value(hrow, t);
                if (v != 0.0) {
                    const double q = v / totd;
                    h -= q * log2(q);
                }
                if (k == 0 && weights) __builtin_nontemporal_store(v, weights + (f0 + fl) * (long)n + t);
                if (marked && v == 0.0) {
                    const int bit = bit0 + t;
                    unresolved = unresolved || ((marked[(size_t)fl * MW + (bit >> 5)] >> (bit & 31)) & 1u) != 0u;
                }
            }
            h = wave_sum(h);
            total_entropy += h / p.lat[k].hmax;
            hrow += n;
            bit0 += n;
        }
        if (marked && __ballot(unresolved) != 0ull && lane == 0)
            p.resolve[1 + atomicAdd(&p.resolve[0], 1u)] = (uint32_t)(f0 + fl);
        if (lane == 0) {
            const int np = cnt_frame[fl];
            double e = total_entropy / (double)p.K;
            if (np == 0) {
                e = __builtin_nan("");
                if (p.status) atomicAdd(&p.status[1], 1);
            }
            entropy[f0 + fl] = e;
            if (present) present[f0 + fl] = np;
        }
    }
#if VET_STAGE_CYCLES
    if (p.timeline && !p.dbg) stage(3);
    if (p.dbg) {
        stage(3);
        if (tid == 0)
            for (int i = 0; i < 8; ++i) atomicAdd(&p.dbg[i], tdbg[i]);
    }
#endif
    if (p.status) {
        const unsigned long long anybad = __ballot(bad);
        if (anybad && lane == 0) atomicAdd(&p.status[0], (int)__popcll(anybad));
    }
}

}  // namespace vet
