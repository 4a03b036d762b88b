// vet_spatial_sweep.hpp — k_spatial_w: FoV-weighted spatial entropy, sweep formulations (integer 2^-52 and FP64 precise)
// Part of the gfx950 device code of the viewport -> tile -> entropy path (see vet_kernels.hpp for the map).
// Reference citations are relative to /root/reference/src/viewport_entropy_toolkit/.
#pragma once
#include "vet_weights.hpp"

namespace vet {

// ------------------------------------------------------------------------------------------
// k_spatial_w — compute_spatial_entropy (entropy_utils.py:147-211), FoV-weighted mode, for FPW
// frames per workgroup.
//
// LDS (dynamic):   hist  u64 [FPW][n]        per-frame tile weight sums, fixed point
//                  dirs  f64 [FPW][UC][3]    unit directions of the present users (compacted)
//                  qc    f64 [NW][64(R+1)]   per-wave compaction queue: cosine
//                  qt    u16 [NW][64(R+1)]                              tile
//                  cnt   i32 [FPW] chunk-present, [FPW] frame-present
// Work item = (frame-local fl, tile group g of 64*R tiles); wave w takes items w, w+NW, ...
// In the sweep every lane owns R tiles (coordinates in registers); for each present user (LDS
// broadcast read) the wave tests the FoV cone with an FP64 dot product, appends the hits to its
// queue (ballot + mbcnt, so the acos/pow part runs on full waves only) and drains 64 entries at a
// time into the LDS histogram with ds_add_u64.  Integer adds commute, so the histogram — and with
// it the entropy — does not depend on scheduling or on the order of users.
// ------------------------------------------------------------------------------------------
struct SpatialParams {
    SampleSrc src;
    int U, T;
    const double* dir_unit;       // [n_dirs][3]
    const uint16_t* nearest;      // [n_dirs] for this lattice
    const double* tiles;          // [n][3] unit
    int n;
    double cos_cull;              // conservative: cos(max_ang) - eps (or < -1 when fov covers all)
    WeightCfg wc;
    double hmax;                  // -n*(1/n)*log2(1/n) (host, reference formula)
    double* ent_k;                // [T]
    int32_t* assign;              // [T*U] or null
    double* weights;              // [T*n] or null
    int32_t* present;             // [T] or null
    int32_t* status;              // [2] or null
    int FPW;                      // frames per workgroup
    int G;                        // tile groups per frame = ceil(n / (64*R))
    int UC;                       // users per LDS chunk
    const double* log2_tab;       // [4097] log2(k), k = 0..4096 (entry 0 is 0); k_spatial_u_lds only
    int norm_n;                   // tile count the user count is compared with (= n except binned lattices)
    int full_norm;                // unweighted kernels: always normalise by log2(n) (binned lattices
                                  // with use_weight_distribution, entropy_utils.py:442-447)
    const struct VideoDesc* videos;   // k_spatial_u_lds: a batch of videos in one launch (null: the single video above)
    int n_videos, n_blocks;
    const uint32_t* frame_list;   // precise sweep as the table kernel's resolver: [0] = number of frames, then the
                                  // frames (any order); null = all T frames, blockIdx.x * FPW onwards
};

// "no key" marker of the FP64 histograms that keep the reference's key set: -0.0.  x + (+0.0) turns it into +0.0
// (IEEE round to nearest), so a tile that only ever received weights of exactly 0.0 reads +0.0 = "key with the
// value 0.0", and every positive weight adds as if the slot had held 0.
constexpr unsigned long long NO_KEY_BITS = 0x8000000000000000ull;

// Entropy of the workgroup's frames from FP64 histograms that keep the key set (precise sweep; entropy_utils.py:
// 194-211, weighted mode).  Every key contributes -p log2 p, also p == 0 (-> NaN, as numpy's 0 * -inf).
// Dense weights output: key with the value 0.0 -> -0.0, no key -> +0.0 (include/vet.h).
__device__ __forceinline__ void keyed_frame_entropy(const double* hist, const int* cnt_frame, int nf, long f0, int n,
                                                    double hmax, double* ent_k, double* weights, int32_t* present,
                                                    int32_t* status, const uint32_t* frames) {
    const int NW = blockDim.x >> 6, lane = lane_id(), wv = wave_id();
    for (int fl = wv; fl < nf; fl += NW) {
        const double* hrow = hist + (size_t)fl * n;
        const long f = frames ? (long)frames[fl] : f0 + fl;
        double totd = 0.0;
        for (int t = lane; t < n; t += WAVE) totd += hrow[t];          // -0.0 adds nothing
        totd = wave_sum(totd);
        double h = 0.0;
        for (int t = lane; t < n; t += WAVE) {
            const double v = hrow[t];
            const bool key = (unsigned long long)__double_as_longlong(v) != NO_KEY_BITS;
            if (key) {
                const double q = v / totd;
                h -= q * log2(q);
            }
            if (weights) __builtin_nontemporal_store(key ? (v == 0.0 ? -0.0 : v) : 0.0, weights + f * (long)n + t);
        }
        h = wave_sum(h);
        if (lane == 0) {
            const int np = cnt_frame[fl];
            double e = h / hmax;
            if (np == 0) {
                e = __builtin_nan("");
                if (status) atomicAdd(&status[1], 1);
            }
            if (ent_k) ent_k[f] = e;                      // null: weights-only pass
            if (present) present[f] = np;
        }
    }
}

// PRECISE: the histogram is FP64 (ds_add_f64) and the weights are the exact ocml values, for plans whose
// entropies can be so small that no fixed-point resolution keeps them within 1e-6 relative (k_row_stats).
// Every tile is owned by one wave, users are staged in column order (absent users as NaN directions, no
// compaction) and a wave's LDS atomics execute in program order, so the per-tile sums run in the
// reference's own order (users in column order) and the result is reproducible run to run.
template <bool FROM_IDS, int WMODE, int R, bool PRECISE>
__global__ void k_spatial_w(const SpatialParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    using HT = typename std::conditional<PRECISE, double, unsigned long long>::type;
    constexpr int QC = WAVE * (R + 1);                                          // queue capacity
    const int NW = blockDim.x >> 6;
    HT* hist = (HT*)smem;                                                        // [FPW][n]
    double* dirs = (double*)(hist + (size_t)p.FPW * p.n);                       // [FPW][UC][3]
    double* qc = dirs + (size_t)p.FPW * p.UC * 3;                                // [NW][QC]
    uint16_t* qt = (uint16_t*)(qc + (size_t)NW * QC);                            // [NW][QC]
    int* cnt_chunk = (int*)(qt + (size_t)NW * QC);                               // [FPW]
    int* cnt_frame = cnt_chunk + p.FPW;                                          // [FPW]

    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    bool bad = false;
    double* my_qc = qc + wv * QC;
    uint16_t* my_qt = qt + wv * QC;
    // blocks of FPW frames: this workgroup's own (blockIdx.x), or — resolver mode — the listed frames one at a
    // time (FPW == 1), dealt round robin to the workgroups of the launch
    const long nblocks = (PRECISE && p.frame_list) ? (long)p.frame_list[0] : (long)gridDim.x;
    for (long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    const uint32_t* frames = (PRECISE && p.frame_list) ? p.frame_list + 1 + blk : nullptr;
    const long f0 = frames ? (long)frames[0] : blk * p.FPW;
    const int nf = (int)min((long)p.FPW, (long)p.T - f0);
    __syncthreads();                          // resolver mode: the previous frame's epilogue has read hist / cnt
    for (int i = tid; i < p.FPW * p.n; i += blockDim.x) {
        if (PRECISE) ((unsigned long long*)hist)[i] = NO_KEY_BITS; else hist[i] = (HT)0;
    }
    for (int i = tid; i < 2 * p.FPW; i += blockDim.x) cnt_chunk[i] = 0;

    for (int u0 = 0; u0 < p.U; u0 += p.UC) {
        const int uc = min(p.UC, p.U - u0);
        __syncthreads();                      // hist/cnt init, or previous chunk fully consumed
        for (int i = tid; i < p.FPW; i += blockDim.x) cnt_chunk[i] = 0;
        __syncthreads();
        // ---- prologue: samples -> direction ids -> unit directions in LDS, nearest tile out
        if (PRECISE && p.frame_list) {
            // resolver of the FP table: that formulation does not depend on the order of the users, so the frames it hands
            // over are summed in a canonical order too — users by ascending direction id (equal ids: equal weights)
            int* ids = cnt_frame + p.FPW;                              // [UC] (the host adds UC * 4 bytes in list mode)
            for (int i = tid; i < uc; i += blockDim.x) ids[i] = sample_dir<FROM_IDS>(p.src, f0 * (long)p.U + u0 + i, bad);
            __syncthreads();
            for (int i = tid; i < uc; i += blockDim.x) {
                const int id = ids[i];
                const unsigned key = (unsigned)id;                     // absent (-1) sorts last
                int rank = 0;
                for (int j = 0; j < uc; ++j) {
                    const unsigned other = (unsigned)ids[j];
                    rank += (other < key || (other == key && j < i)) ? 1 : 0;
                }
                double* dst = dirs + (size_t)rank * 3;
                const double nan = __builtin_nan("");
                dst[0] = id >= 0 ? p.dir_unit[3 * (long)id] : nan;
                dst[1] = id >= 0 ? p.dir_unit[3 * (long)id + 1] : nan;
                dst[2] = id >= 0 ? p.dir_unit[3 * (long)id + 2] : nan;
                if (id >= 0) atomicAdd(&cnt_chunk[0], 1);
            }
        } else
        for (int i = tid; i < nf * uc; i += blockDim.x) {
            const int fl = i / uc, uu = i - fl * uc;
            const long idx = (f0 + fl) * (long)p.U + u0 + uu;
            const int id = sample_dir<FROM_IDS>(p.src, idx, bad);
            if (PRECISE) {
                double* dst = dirs + ((size_t)fl * p.UC + uu) * 3;
                const double nan = __builtin_nan("");
                dst[0] = id >= 0 ? p.dir_unit[3 * (long)id] : nan;
                dst[1] = id >= 0 ? p.dir_unit[3 * (long)id + 1] : nan;
                dst[2] = id >= 0 ? p.dir_unit[3 * (long)id + 2] : nan;
                if (id >= 0) atomicAdd(&cnt_chunk[fl], 1);
            } else if (id >= 0) {
                const int slot = atomicAdd(&cnt_chunk[fl], 1);
                double* dst = dirs + ((size_t)fl * p.UC + slot) * 3;
                dst[0] = p.dir_unit[3 * (long)id];
                dst[1] = p.dir_unit[3 * (long)id + 1];
                dst[2] = p.dir_unit[3 * (long)id + 2];
            }
            if (p.assign) __builtin_nontemporal_store(id >= 0 ? (int)p.nearest[id] : -1, p.assign + idx);
        }
        __syncthreads();
        for (int i = tid; i < p.FPW; i += blockDim.x) cnt_frame[i] += cnt_chunk[i];
        // ---- sweep: lane = R tiles, walk the chunk's present users
        for (int item = wv; item < nf * p.G; item += NW) {
            const int fl = item / p.G, g = item - fl * p.G;
            double tx[R], ty[R], tz[R];
            int tt[R];
            bool valid[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                tt[r] = (g * R + r) * WAVE + lane;
                valid[r] = tt[r] < p.n;
                const int ts = valid[r] ? tt[r] : 0;
                tx[r] = p.tiles[3 * ts]; ty[r] = p.tiles[3 * ts + 1]; tz[r] = p.tiles[3 * ts + 2];
            }
            const int nu = PRECISE ? uc : __builtin_amdgcn_readfirstlane(cnt_chunk[fl]);
            const double* dl = dirs + (size_t)fl * p.UC * 3;
            HT* hrow = hist + (size_t)fl * p.n;
            int qn = 0;
            for (int j = 0; j < nu; ++j) {
                const double dx = dl[3 * j], dy = dl[3 * j + 1], dz = dl[3 * j + 2];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const double c = fma(dz, tz[r], fma(dy, ty[r], dx * tx[r]));
                    const bool hit = valid[r] && (c > p.cos_cull);        // NaN direction (absent): never
                    const unsigned long long mask = __ballot(hit);
                    if (hit) {
                        const int pos = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                             __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                        my_qc[pos] = c;
                        my_qt[pos] = (uint16_t)tt[r];
                    }
                    qn += __popcll(mask);
                }
                while (qn >= WAVE) {
                    qn -= WAVE;
                    __builtin_amdgcn_wave_barrier();
                    const int t = my_qt[qn + lane];
                    if (PRECISE) {
                        double w;
                        if (fov_weight_cone(my_qc[qn + lane], p.wc, w)) atomicAdd((double*)&hrow[t], w);
                    } else {
                        const unsigned long long fx = fov_weight_fx<WMODE>(my_qc[qn + lane], p.wc);
                        if (fx) atomicAdd((unsigned long long*)&hrow[t], fx);
                    }
                    __builtin_amdgcn_wave_barrier();
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (lane < qn) {
                const int t = my_qt[lane];
                if (PRECISE) {
                    double w;
                    if (fov_weight_cone(my_qc[lane], p.wc, w)) atomicAdd((double*)&hrow[t], w);
                } else {
                    const unsigned long long fx = fov_weight_fx<WMODE>(my_qc[lane], p.wc);
                    if (fx) atomicAdd((unsigned long long*)&hrow[t], fx);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();

    if (PRECISE)
        keyed_frame_entropy((const double*)hist, cnt_frame, nf, f0, p.n, p.hmax, p.ent_k, p.weights, p.present, p.status, frames);
    else
        weighted_frame_entropy<HT>(hist, cnt_frame, nf, f0, p.n, 1.0 / (double)(1ull << (52 - p.wc.shift)),
                                   p.hmax, p.ent_k, p.weights, p.present, p.status);
    }
    if (p.status) {
        const unsigned long long anybad = __ballot(bad);
        if (anybad && lane == 0) atomicAdd(&p.status[0], (int)__popcll(anybad));
    }
}

}  // namespace vet
