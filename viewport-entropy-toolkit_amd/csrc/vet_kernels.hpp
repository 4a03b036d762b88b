// vet_kernels.hpp — gfx950 (MI355X, CDNA4) device code of the viewport -> tile -> entropy path.
//
// Kernels (wave = 64 lanes everywhere):
//   plan tables (once per plan)
//     k_grid_dirs      axis tables -> rounded + normalised direction per pixel (py,px)
//     k_nearest_lut    direction -> nearest lattice tile (FP64 arg-max of the normalised dot,
//                      lowest index on ties), one LUT per lattice
//     k_wtab           direction -> ELL row of (tile, FoV weight) pairs, exact ocml acos / pow
//     k_log2_table     log2(k), k <= 4096, for the integer-count entropies
//   spatial entropy, FoV-weighted
//     k_spatial_lut    table formulation: per frame, samples -> direction ids -> gather of the
//                      users' rows into 64-bit integer LDS histograms (all lattices in one launch)
//                      -> Shannon entropy; results are order independent, hence bit-reproducible
//     k_spatial_w      sweep formulation (few samples per plan): lane = tile, FP64 cone test per
//                      (user, tile), ballot-compacted full-wave weight evaluation
//   spatial entropy, nearest-tile (unweighted) and naive lat/lon-grid mode
//     k_spatial_u_lds  persistent stream with the nearest LUT in LDS (HBM-bound)
//     k_spatial_u      generic fallback (LUT gathered from global memory)
//   transition entropy
//     k_transition_run per frame pair: (prior tile, current tile) pairs -> bucket statistics in
//                      LDS (integer atomics + one small hash table) -> transition entropy; persistent
//                      workgroups over runs of rows, every frame quantised once
//     k_transition_any the same for any number of users (bucket hash in global scratch)
//   k_finalize         mean over the plan's lattices where they ran as separate launches
//
// No MFMA: there is no dense contraction on this path.  Reference citations are relative to
// /root/reference/src/viewport_entropy_toolkit/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace vet {

constexpr int WAVE = 64;
constexpr unsigned EMPTY_KEY = 0xFFFFFFFFu;

// ------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;   // butterfly: same value, same order, in every lane
}
__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

// Workgroup barrier for LDS-only hand-offs: waits for this wave's LDS operations (lgkmcnt), not for
// its global loads/stores, so requests to HBM stay in flight across it (__syncthreads() also
// drains vmcnt).  Only for phases that exchange data through LDS.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Streamed once: non-temporal loads / stores keep the sample stream from displacing the tables in L2
__device__ __forceinline__ double2 nt_load(const double2* p) {
    double2 v;
    v.x = __builtin_nontemporal_load(&p->x);
    v.y = __builtin_nontemporal_load(&p->y);
    return v;
}
__device__ __forceinline__ void nt_store(int2* p, int2 v) {
    __builtin_nontemporal_store(v.x, &p->x);
    __builtin_nontemporal_store(v.y, &p->y);
}

// numpy-scalar round(v, 6) == rint(v * 1e6) / 1e6   (data_types.py:213-215)
__device__ __forceinline__ double round6(double v) { return rint(v * 1e6) / 1e6; }

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

// ------------------------------------------------------------------------------------------
// sample -> direction id
// ------------------------------------------------------------------------------------------
struct SampleSrc {
    const double* mu;      // [T*U] or null
    const double* mv;
    const int32_t* ids;    // [T*U] or null
    int W, H;
    long n_dirs;
};

// (mu, mv) -> direction id on the pixel grid, -1 when absent; sets bad when outside [0,1]
// (normalize_to_pixel, data_utils.py:243-261: (v * dim).astype(int) truncates toward zero)
__device__ __forceinline__ int grid_dir(double m, double v, int W, int H, bool& bad) {
    if (m != m || v != v) return -1;                           // dropna()
    if (!(m >= 0.0 && m <= 1.0 && v >= 0.0 && v <= 1.0)) { bad = true; return -1; }
    return (int)(v * (double)H) * (W + 1) + (int)(m * (double)W);
}

// returns direction id, -1 when absent; sets bad when a value is outside [0,1]
template <bool FROM_IDS, bool NT = true>
__device__ __forceinline__ int sample_dir(const SampleSrc& s, long idx, bool& bad) {
    if (FROM_IDS) {
        const int id = s.ids[idx];
        if (id >= s.n_dirs) { bad = true; return -1; }
        return id < 0 ? -1 : id;
    } else {
        if (NT) return grid_dir(__builtin_nontemporal_load(s.mu + idx), __builtin_nontemporal_load(s.mv + idx), s.W, s.H, bad);
        return grid_dir(s.mu[idx], s.mv[idx], s.W, s.H, bad);
    }
}

// ------------------------------------------------------------------------------------------
// FoV weight of one (direction, tile) pair from their cosine
// calculate_tile_weights, entropy_utils.py:124-137:  d = arccos(clip(c)); if d < max:
//   w = ((max - d) / max) ** power.  Returned in 64-bit fixed point: w * 2^(52 - shift).
// ------------------------------------------------------------------------------------------
struct WeightCfg {
    double max_ang;     // np.radians(fov/2)
    double inv_max;     // 1 / max_ang
    double power;
    int shift;          // fixed point = 2^(52-shift); shift = max(0, ceil(log2 U) - 10)
};

// WMODE: 0 generic (ocml acos, pow)   1 fast acos, power == 2   2 fast acos, power == 1
// The fast acos needs max_ang <= 60 deg (fov <= 120): then c >= 0.5 - 1e-9 and
//   theta = 2 asin(s), s = sqrt(z), z = (1 - c)/2 <= 0.2502,
//   asin(s) = s + s z P(z), P of degree 9 fitted on [0, 0.2502]: |d theta| / theta < 2e-14.
__device__ __forceinline__ double fast_theta(double c) {
    const double z = fmax((1.0 - c) * 0.5, 1e-300);
    // sqrt(z): hardware rsq seed, one Goldschmidt step and one residual correction
    const double y = __builtin_amdgcn_rsq(z);
    double s = z * y, h = 0.5 * y;
    const double e = fma(-h, s, 0.5);
    s = fma(s, e, s);
    h = fma(h, e, h);
    s = fma(fma(-s, s, z), h, s);
    double P = 2.80476016723745745e-02;
    P = fma(P, z, -3.09562448984870928e-03);
    P = fma(P, z, 1.57475990547630423e-02);
    P = fma(P, z, 1.31700206864407612e-02);
    P = fma(P, z, 1.74440881411108591e-02);
    P = fma(P, z, 2.23658455433679397e-02);
    P = fma(P, z, 3.03821932887589595e-02);
    P = fma(P, z, 4.46428521871264916e-02);
    P = fma(P, z, 7.50000000381451232e-02);
    P = fma(P, z, 1.66666666666618335e-01);
    const double a = fma(s * z, P, s);
    return a + a;
}

// w in [0, 1] -> round(w * 2^52) through the mantissa of 1 + w, then >> shift
__device__ __forceinline__ unsigned long long unit_to_fx(double w, int shift) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(1.0 + w);
    return (bits - 0x3FF0000000000000ull) >> shift;
}

template <int WMODE>
__device__ __forceinline__ unsigned long long fov_weight_fx(double c, const WeightCfg& w) {
    if (WMODE == 0) {
        c = fmin(fmax(c, -1.0), 1.0);
        const double d = acos(c);
        if (!(d < w.max_ang)) return 0ull;
        const double r = (w.max_ang - d) / w.max_ang;
        return unit_to_fx(pow(r, w.power), w.shift);
    } else {
        const double r = fmax((w.max_ang - fast_theta(c)) * w.inv_max, 0.0);   // 0 <=> not d < max
        return unit_to_fx(WMODE == 1 ? r * r : r, w.shift);
    }
}

// ------------------------------------------------------------------------------------------
// Entropy of the workgroup's frames from their fixed-point tile histograms
// (entropy_utils.py:194-211, weighted mode: normaliser log2(n)).  Wave w takes frames w, w+NW, ...
// ------------------------------------------------------------------------------------------
template <typename HT>
__device__ __forceinline__ void weighted_frame_entropy(const HT* hist, const int* cnt_frame, int nf,
                                                       long f0, int n, double inv_unit, double hmax,
                                                       double* ent_k, double* weights, int32_t* present,
                                                       int32_t* status) {
    const int NW = blockDim.x >> 6, lane = lane_id(), wv = wave_id();
    for (int fl = wv; fl < nf; fl += NW) {
        const HT* hrow = hist + (size_t)fl * n;
        // total weight: up to U*n/4 fixed-point units, which can exceed 64 bits, so it is summed
        // in FP64 (fixed lane order + butterfly => still a pure function of the histogram)
        double totd = 0.0;
        for (int t = lane; t < n; t += WAVE) totd += (double)hrow[t];
        totd = wave_sum(totd);
        double h = 0.0;
        for (int t = lane; t < n; t += WAVE) {
            const HT v = hrow[t];
            if (v != (HT)0) {
                const double q = (double)v / totd;
                h -= q * log2(q);
            }
            if (weights) __builtin_nontemporal_store((double)v * inv_unit, weights + (f0 + fl) * (long)n + t);
        }
        h = wave_sum(h);
        if (lane == 0) {
            const int np = cnt_frame[fl];
            double e = h / hmax;
            if (np == 0) {
                e = __builtin_nan("");
                if (status) atomicAdd(&status[1], 1);
            }
            ent_k[f0 + fl] = e;
            if (present) present[f0 + fl] = np;
        }
    }
}

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
};

// the reference's weight, evaluated as the reference does (entropy_utils.py:124-137): 0 when not d < max
__device__ __forceinline__ double fov_weight_exact(double c, const WeightCfg& w) {
    c = fmin(fmax(c, -1.0), 1.0);
    const double d = acos(c);
    if (!(d < w.max_ang)) return 0.0;
    return pow((w.max_ang - d) / w.max_ang, w.power);
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
    const long f0 = (long)blockIdx.x * p.FPW;
    const int nf = (int)min((long)p.FPW, (long)p.T - f0);

    for (int i = tid; i < p.FPW * p.n; i += blockDim.x) hist[i] = (HT)0;
    for (int i = tid; i < 2 * p.FPW; i += blockDim.x) cnt_chunk[i] = 0;
    bool bad = false;
    double* my_qc = qc + wv * QC;
    uint16_t* my_qt = qt + wv * QC;

    for (int u0 = 0; u0 < p.U; u0 += p.UC) {
        const int uc = min(p.UC, p.U - u0);
        __syncthreads();                      // hist/cnt init, or previous chunk fully consumed
        for (int i = tid; i < p.FPW; i += blockDim.x) cnt_chunk[i] = 0;
        __syncthreads();
        // ---- prologue: samples -> direction ids -> unit directions in LDS, nearest tile out
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
                        const double w = fov_weight_exact(my_qc[qn + lane], p.wc);
                        if (w > 0.0) atomicAdd((double*)&hrow[t], w);
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
                    const double w = fov_weight_exact(my_qc[lane], p.wc);
                    if (w > 0.0) atomicAdd((double*)&hrow[t], w);
                } else {
                    const unsigned long long fx = fov_weight_fx<WMODE>(my_qc[lane], p.wc);
                    if (fx) atomicAdd((unsigned long long*)&hrow[t], fx);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();

    weighted_frame_entropy<HT>(hist, cnt_frame, nf, f0, p.n, PRECISE ? 1.0 : 1.0 / (double)(1ull << (52 - p.wc.shift)),
                               p.hmax, p.ent_k, p.weights, p.present, p.status);
    if (p.status) {
        const unsigned long long anybad = __ballot(bad);
        if (anybad && lane == 0) atomicAdd(&p.status[0], (int)__popcll(anybad));
    }
}

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
constexpr int TAB_X = 16;       // histogram unit 2^-(32+TAB_X): sums of < 2^16 weights <= 1 fit 64 bits

struct StatsParams {
    const double* dir_unit;
    long D;
    const double* tiles;
    int n;
    double cos_cull;
    WeightCfg wc;
    uint8_t* row_s;             // [D+1] TAB_X + e per row (row D = the all-zero row)
    uint16_t* row_e;            // [D+1] E = -(binary exponent of the row's largest weight), unclamped (FP table)
    unsigned long long* crit;   // [2] bit patterns of non-negative doubles (atomicMax):
                                //   [0] max_d 36.5 q_d k_d / (S_d H_d)   with the table's q_d = 2^(e_d - 33)
                                //   [1] max_d 36.5 k_d / (S_d H_d)       (times the sweep's step, 2^(shift-53))
};

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, WAVE));
    return v;
}

__global__ void k_row_stats(const StatsParams p) {
    const int lane = lane_id();
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    double worst_tab = 0.0, worst_sweep = 0.0;
    for (long d = wave; d < p.D; d += nwaves) {
        const double dx = p.dir_unit[3 * d], dy = p.dir_unit[3 * d + 1], dz = p.dir_unit[3 * d + 2];
        double S = 0.0, L = 0.0, mx = 0.0;
        int k = 0;
        for (int t0 = 0; t0 < p.n; t0 += WAVE) {
            const int t = t0 + lane;
            const bool valid = t < p.n;
            const int ts = valid ? t : 0;
            const double c = fma(dz, p.tiles[3 * ts + 2], fma(dy, p.tiles[3 * ts + 1], dx * p.tiles[3 * ts]));
            if (valid && c > p.cos_cull) {
                const double wt = fov_weight_exact(c, p.wc);
                if (wt > 0.0) { ++k; S += wt; L += wt * log2(wt); mx = fmax(mx, wt); }
            }
        }
        k = wave_sum(k); S = wave_sum(S); L = wave_sum(L); mx = wave_max(mx);
        int e = 0;
        if (mx > 0.0) (void)frexp(mx, &e);                  // mx <= 2^e
        if (lane == 0) p.row_e[d] = (uint16_t)min(2047, max(0, -e));
        e = min(0, max(-TAB_X, e));
        if (lane == 0) p.row_s[d] = (uint8_t)(TAB_X + e);
        if (k >= 1) {
            // row entropy -sum (w/S) log2(w/S) = log2 S - (sum w log2 w) / S; its own rounding error (~1e-15)
            // only matters where the bound is hopeless anyway
            const double H = k >= 2 ? fmax(log2(S) - L / S, 0.0) : 0.0;
            const double base = H > 0.0 ? 36.5 * (double)k / (S * H) : __builtin_inf();
            worst_tab = fmax(worst_tab, base * ldexp(1.0, e - 33));
            worst_sweep = fmax(worst_sweep, base);
        }
    }
    if (lane == 0) {
        if (p.row_s && wave == 0) { p.row_s[p.D] = (uint8_t)TAB_X; p.row_e[p.D] = 0; }
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
__device__ __forceinline__ int below(unsigned long long m) {      // set bits of m below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

template <bool FILL>
__global__ void k_wtab(const WtabParams p) {
    const int lane = lane_id();
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    int longest = 0;
    for (long d = wave; d < p.D; d += nwaves) {
        const double dx = p.dir_unit[3 * d], dy = p.dir_unit[3 * d + 1], dz = p.dir_unit[3 * d + 2];
        int count = 0;
        const int row_shift = FILL ? (p.fp ? (int)p.row_e[d] : (int)p.row_s[d]) : 0;
        const double scale = p.fp ? ldexp(1.0, row_shift) : ldexp(1.0, 32 + TAB_X - row_shift);        // 2^E / 2^(32 - e)
        for (int t0 = 0; t0 < p.n; t0 += WAVE) {
            const int t = t0 + lane;
            const bool valid = t < p.n;
            const int ts = valid ? t : 0;
            const double c = fma(dz, p.tiles[3 * ts + 2], fma(dy, p.tiles[3 * ts + 1], dx * p.tiles[3 * ts]));
            bool hit = valid && (c > p.cos_cull);
            unsigned w32 = 0u;
            if (FILL) {
                if (hit) {
                    const double wt = fov_weight_exact(c, p.wc);
                    w32 = p.fp ? __float_as_uint((float)(wt * scale)) : (unsigned)fmin(rint(wt * scale), 4294967295.0);
                }
                hit = w32 != 0u;
            }
            const unsigned long long mask = __ballot(hit);
            if (FILL && hit) {
                const int pos = count + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                        __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
                p.w[d * p.stride + pos] = w32;
                p.idx[d * p.stride + pos] = (uint16_t)t;
            }
            count += __popcll(mask);
        }
        if (FILL) {
            // padding: weight 0 on distinct tiles, so the gather can add every slot unconditionally
            // without piling zero adds onto one LDS address
            for (int pos = count + lane; pos < p.stride; pos += WAVE) {
                p.w[d * p.stride + pos] = 0u;
                p.idx[d * p.stride + pos] = (uint16_t)(pos % p.n);
            }
            if (lane == 0) p.meta[d] = (uint32_t)count | ((uint32_t)row_shift << 16);
            // well-filled blocks of 16-lane rows: deal the entries by class (one wave pass per block, lane =
            // sorted entry; the loads of all lanes have returned before the first store issues)
            if (p.gs_log2 == 4) {
                for (int eb = 0; eb < count; eb += WAVE) {
                    if (!block_interleaved(count, eb, p.gs_log2)) continue;
                    __threadfence_block();
                    const bool real = eb + lane < count;
                    uint32_t wv = 0; uint16_t iv = 0;
                    if (real) { wv = p.w[d * p.stride + eb + lane]; iv = p.idx[d * p.stride + eb + lane]; }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    const int cls = iv & 15;
                    int r = 0;                                   // rank among the entries of the same class
                    for (int c = 0; c < 16; ++c) {
                        const unsigned long long m = __ballot(real && cls == c);
                        if (real && cls == c) r = below(m);
                    }
                    int comp = r & 3, q = 0, used[4], freeb[5];
                    for (int k = 0; k < 4; ++k) {                // lane inside the component, 16 places each
                        const unsigned long long m = __ballot(real && comp == k);
                        if (real && comp == k) q = below(m);
                        used[k] = min(16, (int)__popcll(m));
                    }
                    const bool placed = real && q < 16;
                    unsigned usedmask[4];                        // classes present in each component
                    for (int k = 0; k < 4; ++k) {
                        usedmask[k] = 0;
                        for (int c = 0; c < 16; ++c)
                            if (__ballot(placed && comp == k && cls == c)) usedmask[k] |= 1u << c;
                    }
                    freeb[0] = 0;
                    for (int k = 0; k < 4; ++k) freeb[k + 1] = freeb[k] + 16 - used[k];
                    if (!placed) {                               // leftovers take the free places in order
                        const int j = below(__ballot(!placed));
                        int k = 0;
                        while (k < 3 && j >= freeb[k + 1]) ++k;
                        const int jj = j - freeb[k];
                        comp = k; q = used[k] + jj;
                        if (!real) {                             // padding: jj-th class the component lacks
                            int seen = 0, c = 0;
                            for (; c < 15; ++c) {
                                if (!((usedmask[k] >> c) & 1u)) { if (seen == jj) break; ++seen; }
                            }
                            iv = (uint16_t)c;                    // tile c has class c (n > 16 for 16-lane rows)
                        }
                    }
                    p.w[d * p.stride + eb + q * 4 + comp] = wv;
                    p.idx[d * p.stride + eb + q * 4 + comp] = iv;
                }
            }
        }
        longest = max(longest, count);
    }
    // a plain read first: the maximum only grows, so most waves find theirs already covered and skip
    // the same-address atomic (2048 of them cost ~100 us)
    if (!FILL && lane == 0 && longest > *(volatile int*)p.maxcount) atomicMax(p.maxcount, longest);
    if (FILL && wave == 0) {            // row D: the all-zero row idle lanes of the gather point at
        for (int pos = lane; pos < p.stride; pos += WAVE) {
            p.w[p.D * p.stride + pos] = 0u;
            // lane l of a 16-lane group adds its zeros to tile l: 16 classes, no conflict
            p.idx[p.D * p.stride + pos] = (uint16_t)(p.gs_log2 == 4 ? (pos >> 2) & 15 : pos % p.n);
        }
        if (lane == 0) p.meta[p.D] = p.fp ? 0u : (uint32_t)TAB_X << 16;
    }
}

// ------------------------------------------------------------------------------------------
// Row walk shared by the table kernels: the workgroup adds the ELL rows of the frame's distinct
// directions into the LDS histogram hrow.  frows[j] = row << 12 | multiplicity (DEDUP) or the row
// (multiplicity 1); fmeta[j] = the row's meta word.  A group of GS = 2^gs_log2 lanes walks one
// row; UN rows per group are in flight; rows are zero padded, so a group walks to the longest of
// its UN rows only.
// ------------------------------------------------------------------------------------------
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

constexpr int ROW_BITS = 19;
constexpr uint32_t ROW_MASK = (1u << ROW_BITS) - 1;

// FPT: FP table — entries are FP32 weights (relative precision 2^-24 each: |dH|/H <= 1.2e-7 for every frame, whatever the
// weights' dynamic range), scaled by 2^E of their row; the histogram is FP64 (ds_add_f64) in true units.
template <int UN, bool INTERLEAVED, bool DEDUP, bool FPT>
__device__ __forceinline__ void walk_rows(const uint32_t* frows, const uint32_t* fmeta, int nu,
                                          unsigned long long* hrow, int n,
                                          const uint32_t* __restrict__ tab_w, const uint16_t* __restrict__ tab_i,
                                          int stride, int gs_log2_rt, uint32_t zero_row) {
    // Every lane takes one 4-slot chunk per block: one 16-byte load of weights, one 8-byte load of
    // tiles (stride is a multiple of the block, so chunks are 16 / 8 byte aligned), then four
    // unconditional ds_add_u64 of entry * (multiplicity << row shift): padding slots and idle lanes
    // (which walk the all-zero row) add 0 to distinct tiles — no predicates around the adds.
    const int NW = blockDim.x >> 6, lane = lane_id(), wv = wave_id();
    const int gs_log2 = INTERLEAVED ? 4 : gs_log2_rt;       // class-dealt rows are 16-lane rows (ensure_wtab)
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
                    atomicAdd((double*)(hb[k] + (int)t[k].x * sgn[k]), (double)__uint_as_float(w[k].x) * scale[k]);
                    atomicAdd((double*)(hb[k] + (int)t[k].y * sgn[k]), (double)__uint_as_float(w[k].y) * scale[k]);
                    atomicAdd((double*)(hb[k] + (int)t[k].z * sgn[k]), (double)__uint_as_float(w[k].z) * scale[k]);
                    atomicAdd((double*)(hb[k] + (int)t[k].w * sgn[k]), (double)__uint_as_float(w[k].w) * scale[k]);
                } else {
                    atomicAdd((unsigned long long*)(hb[k] + (int)t[k].x * sgn[k]), (unsigned long long)w[k].x * mult[k]);
                    atomicAdd((unsigned long long*)(hb[k] + (int)t[k].y * sgn[k]), (unsigned long long)w[k].y * mult[k]);
                    atomicAdd((unsigned long long*)(hb[k] + (int)t[k].z * sgn[k]), (unsigned long long)w[k].z * mult[k]);
                    atomicAdd((unsigned long long*)(hb[k] + (int)t[k].w * sgn[k]), (unsigned long long)w[k].w * mult[k]);
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
constexpr int MAX_LATTICES = 8;
constexpr unsigned DEDUP_MAX_DIRS = (1u << 19) - 1;      // set key = row (19 bits) | mirror flag; slot = key << 12 | count

struct LutLattice {
    const uint32_t* tab_w;
    const uint16_t* tab_i;
    const uint32_t* tab_meta;
    int stride, gs_log2, n, interleaved;
    double hmax;
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
};

// hash slots per frame: power of two >= 2 * UC, at least one wave's worth
__host__ __device__ __forceinline__ int lut_hash_slots(int UC) {
    int hs = 64;
    while (hs < 2 * UC) hs <<= 1;
    return hs;
}
// LDS bytes of a workgroup; the kernel and the host must agree
__host__ __device__ __forceinline__ size_t lut_lds_bytes(int U, int UC, int FPW, int n_sum, bool dedup) {
    const size_t hist = (size_t)FPW * n_sum * 8, hash = dedup ? (size_t)FPW * lut_hash_slots(UC) * 4 : 0;
    const size_t a = (dedup && U <= UC) ? (hist > hash ? hist : hash) : hist + hash;
    return ((a + 15) & ~(size_t)15) + (size_t)FPW * UC * 8 + (size_t)2 * FPW * 4 + 64;
}

// All K lattices of the plan in one launch: the samples are read once, every row is gathered into K
// histograms, and avg_entropy = (e_0 + ... + e_{K-1}) / K is formed in lattice order as the
// reference does (spatial_entropy.py:142-156) — no per-lattice pass, no finalize.
// IL: some lattice of the plan has interleaved rows (otherwise only the plain walk is compiled in).
// OCC8: compiled for 8 workgroups of 256 threads per CU (64 VGPRs) instead of 7 (68-70 VGPRs): measured
// 2 % (random walk) to 6.5 % (clustered) faster on single-lattice plans and 7 % on batches of short
// videos, but 2-4 % slower on one multi-lattice video (profiles/r01/v6_table_occupancy.log).
// DEDUP: per-frame set of distinct rows with multiplicities (direction tables of < 2^20 rows).
template <bool FROM_IDS, int UN, bool IL, bool OCC8, bool DEDUP, bool FPT>
__global__ __launch_bounds__(256, OCC8 ? 8 : 7) void k_spatial_lut(const LutParams p) {
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
    const bool overlay = DEDUP && U <= UC;                                       // one chunk: set and histogram share space
    const size_t hist_bytes = (size_t)FPW * p.n_sum * 8, hash_bytes = (size_t)FPW * HS * 4;
    const size_t a_bytes = overlay ? (hist_bytes > hash_bytes ? hist_bytes : hash_bytes) : hist_bytes + hash_bytes;
    unsigned long long* hist = (unsigned long long*)smem;                        // [FPW][n_sum]
    uint32_t* hash = (uint32_t*)(smem + (overlay ? 0 : hist_bytes));             // [FPW][HS]
    uint32_t* rows = (uint32_t*)(smem + ((a_bytes + 15) & ~(size_t)15));         // [FPW][UC]
    uint32_t* meta = rows + (size_t)FPW * UC;                                    // [FPW][UC]
    int* cnt_chunk = (int*)(meta + (size_t)FPW * UC);                            // [FPW]
    int* cnt_frame = cnt_chunk + FPW;                                            // [FPW]
    const int NW = blockDim.x >> 6;
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const long f0 = blk * FPW;
    const int nf = (int)min((long)FPW, (long)T - f0);
    if (!overlay)
        for (int i = tid; i < FPW * p.n_sum; i += blockDim.x) hist[i] = 0ull;
    for (int i = tid; i < 2 * FPW; i += blockDim.x) cnt_chunk[i] = 0;
    bool bad = false;
    const int hs_shift = 32 - (31 - __clz(HS | 1));
    for (int u0 = 0; u0 < U; u0 += UC) {
        const int uc = min(UC, U - u0);
        __syncthreads();
        for (int i = tid; i < FPW; i += blockDim.x) cnt_chunk[i] = 0;
        if (DEDUP)
            for (int i = tid; i < FPW * HS; i += blockDim.x) hash[i] = EMPTY_KEY;
        __syncthreads();
        // SPT samples per thread and round: all sample loads first, then the table gathers, then the LDS set
        // inserts — three waves of independent requests instead of SPT dependent chains
        constexpr int SPT = 4;
        const int total = nf * uc;
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
                if (DEDUP) {
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
                        rows[pos] = DEDUP ? h : row[k];
                        meta[pos] = m0[k];
                    }
                } else {
                    if (valid) atomicAdd(&cnt_frame[fl], 1);
                    if (won) {
                        const size_t pos = (size_t)fl * UC + atomicAdd(&cnt_chunk[fl], 1);
                        rows[pos] = DEDUP ? h : row[k];
                        meta[pos] = m0[k];
                    }
                }
            }
        }
        __syncthreads();
        if (DEDUP) {
            // slot numbers -> slot words (row << 12 | multiplicity)
            for (int i = tid; i < nf * UC; i += blockDim.x) {
                const int fl = i / UC, j = i - fl * UC;
                if (j < cnt_chunk[fl]) rows[i] = hash[(size_t)fl * HS + rows[i]];
            }
            if (overlay) {
                __syncthreads();
                for (int i = tid; i < FPW * p.n_sum; i += blockDim.x) hist[i] = 0ull;
            }
        }
        int hoff = 0;
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
            for (int fl = 0; fl < nf; ++fl)
                if (IL && L.interleaved)
                    walk_rows<UN, true, DEDUP, FPT>(rows + (size_t)fl * UC, meta + (size_t)fl * UC, cnt_chunk[fl],
                                               hist + (size_t)fl * p.n_sum + hoff, L.n, L.tab_w, L.tab_i, L.stride, L.gs_log2,
                                               (uint32_t)src.n_dirs * (uint32_t)L.stride);
                else
                    walk_rows<UN, false, DEDUP, FPT>(rows + (size_t)fl * UC, meta + (size_t)fl * UC, cnt_chunk[fl],
                                                hist + (size_t)fl * p.n_sum + hoff, L.n, L.tab_w, L.tab_i, L.stride, L.gs_log2,
                                                (uint32_t)src.n_dirs * (uint32_t)L.stride);
            hoff += L.n;
        }
    }
    __syncthreads();
    // entropy (entropy_utils.py:194-211, weighted: normaliser log2 n); wave w takes frames w, w+NW, ...
    const double inv_unit = 1.0 / (4294967296.0 * (double)(1u << TAB_X));
    for (int fl = wv; fl < nf; fl += NW) {
        const unsigned long long* hrow = hist + (size_t)fl * p.n_sum;
        double total_entropy = 0.0;
        for (int k = 0; k < p.K; ++k) {
            const int n = p.lat[k].n;
            // total weight can exceed 64 bits of fixed point: summed in FP64, fixed lane order + butterfly
            double totd = 0.0;
            for (int t = lane; t < n; t += WAVE) totd += FPT ? __longlong_as_double((long long)hrow[t]) : (double)hrow[t];
            totd = wave_sum(totd);
            double h = 0.0;
            for (int t = lane; t < n; t += WAVE) {
                const double v = FPT ? __longlong_as_double((long long)hrow[t]) : (double)hrow[t];
                if (v != 0.0) {
                    const double q = v / totd;
                    h -= q * log2(q);
                }
                if (k == 0 && weights) __builtin_nontemporal_store(FPT ? v : v * inv_unit, weights + (f0 + fl) * (long)n + t);
            }
            h = wave_sum(h);
            total_entropy += h / p.lat[k].hmax;
            hrow += n;
        }
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
    if (p.status) {
        const unsigned long long anybad = __ballot(bad);
        if (anybad && lane == 0) atomicAdd(&p.status[0], (int)__popcll(anybad));
    }
}

// ------------------------------------------------------------------------------------------
// k_spatial_u — the same entropy with use_weight_distribution = False: every present user adds
// weight 1.0 to its nearest tile (entropy_utils.py:139-142), so the frame histogram is an integer
// count per tile and the path is a pure stream: 16 B in, LUT gather, 4 B out per sample.
// LDS: cnt u32 [FPW][n].  Wave w owns frames w, w+NW, ... of the workgroup's FPW frames.
// ------------------------------------------------------------------------------------------
template <bool FROM_IDS>
__global__ void k_spatial_u(const SpatialParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* cnt = (unsigned*)smem;                                             // [FPW][n]
    const int NW = blockDim.x >> 6;
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const long f0 = (long)blockIdx.x * p.FPW;
    const int nf = (int)min((long)p.FPW, (long)p.T - f0);
    for (int i = tid; i < p.FPW * p.n; i += blockDim.x) cnt[i] = 0u;
    __syncthreads();
    bool bad = false;
    const long base = f0 * (long)p.U, total = (long)nf * p.U;
    if (!FROM_IDS && (p.U & 1) == 0) {
        // 16-byte loads: a lane takes two neighbouring users; four such pairs are in flight.
        // (frame, pair-in-frame) of the flat pair index is tracked incrementally: no division
        // in the loop.
        constexpr int UN = 4;
        const double2* mu2 = (const double2*)(p.src.mu + base);
        const double2* mv2 = (const double2*)(p.src.mv + base);
        int2* out2 = (int2*)(p.assign ? p.assign + base : nullptr);
        const int ppf = p.U >> 1;                                   // pairs per frame
        const int pairs = nf * ppf;
        const int dq = (int)blockDim.x / ppf, dr = (int)blockDim.x % ppf;
        int fl0 = tid / ppf, j0 = tid % ppf;
        for (int i0 = tid; i0 < pairs; i0 += UN * (int)blockDim.x) {
            double2 a[UN], b[UN];
            int fl[UN];
            int fk = fl0, jk = j0;
#pragma unroll
            for (int k = 0; k < UN; ++k) {
                const int i = i0 + k * (int)blockDim.x;
                fl[k] = fk;
                if (i < pairs) { a[k] = mu2[i]; b[k] = mv2[i]; }
                fk += dq; jk += dr;
                if (jk >= ppf) { jk -= ppf; ++fk; }
            }
            fl0 = fk; j0 = jk;
            int near[UN][2];
#pragma unroll
            for (int k = 0; k < UN; ++k) {
                near[k][0] = near[k][1] = -1;
                if (i0 + k * (int)blockDim.x < pairs) {
                    const int id0 = grid_dir(a[k].x, b[k].x, p.src.W, p.src.H, bad);
                    const int id1 = grid_dir(a[k].y, b[k].y, p.src.W, p.src.H, bad);
                    if (id0 >= 0) near[k][0] = p.nearest[id0];
                    if (id1 >= 0) near[k][1] = p.nearest[id1];
                }
            }
#pragma unroll
            for (int k = 0; k < UN; ++k) {
                const int i = i0 + k * (int)blockDim.x;
                if (i < pairs) {
                    unsigned* row = cnt + (size_t)fl[k] * p.n;
                    if (near[k][0] >= 0) atomicAdd(&row[near[k][0]], 1u);
                    if (near[k][1] >= 0) atomicAdd(&row[near[k][1]], 1u);
                    if (out2) out2[i] = make_int2(near[k][0], near[k][1]);
                }
            }
        }
    } else {
        for (long i = tid; i < total; i += blockDim.x) {
            const int fl = (int)(i / p.U);
            const long idx = base + i;
            const int id = sample_dir<FROM_IDS>(p.src, idx, bad);
            int near = -1;
            if (id >= 0) {
                near = p.nearest[id];
                atomicAdd(&cnt[(size_t)fl * p.n + near], 1u);
            }
            if (p.assign) __builtin_nontemporal_store(near, p.assign + idx);
        }
    }
    __syncthreads();
    for (int fl = wv; fl < nf; fl += NW) {
        const unsigned* row = cnt + (size_t)fl * p.n;
        int np = 0;
        for (int t = lane; t < p.n; t += WAVE) np += (int)row[t];
        np = wave_sum(np);
        const double tw = (double)np;             // total_weight == number of present users
        double h = 0.0;
        for (int t = lane; t < p.n; t += WAVE) {
            const unsigned v = row[t];
            if (v) {
                const double q = (double)v / tw;
                h -= q * log2(q);
            }
            if (p.weights) __builtin_nontemporal_store((double)v, p.weights + (f0 + fl) * (long)p.n + t);
        }
        h = wave_sum(h);
        if (lane == 0) {
            double hmax = p.hmax;                  // entropy_utils.py:201-206
            if (!(tw > (double)p.norm_n) && !p.full_norm) {
                const double mp = 1.0 / tw;
                hmax = -tw * mp * log2(mp);
            }
            double e = h / hmax;
            if (np == 0) {
                e = __builtin_nan("");
                if (p.status) atomicAdd(&p.status[1], 1);
            }
            p.ent_k[f0 + fl] = e;
            if (p.present) p.present[f0 + fl] = np;
        }
    }
    if (p.status) {
        const unsigned long long anybad = __ballot(bad);
        if (anybad && lane == 0) atomicAdd(&p.status[0], (int)__popcll(anybad));
    }
}

// ------------------------------------------------------------------------------------------
// k_spatial_u_lds — k_spatial_u for plans whose nearest-tile LUT fits the LDS (40 KB at the
// default 100x200 grid).  Measured on MI355X (tools/stream_probe.hip): the 2-byte LUT gather
// from global memory runs at about one lane per cycle per CU and costs 70 us of a 177 us
// kernel, while the same stream with the LUT in LDS reaches 5.1 TB/s.  So: persistent
// workgroups (1024 threads, 2 per CU) load the LUT into LDS once and walk the frame axis in
// blocks of FB frames (FB * U/2 <= 2048 sample pairs, two pairs per thread, 16-byte loads).
// Per sample: 16 B in, one ds_read_u16, one ds_add_u32, 4 B out.  Frame f of a round is reduced
// to its entropy by wave f.  Requires an even U <= 4096 and grid samples.
// Counts are integers <= U, so log2(v/N) is taken as lg[v] - lg[N] from an LDS table of log2(k),
// k = 1..U, copied from a per-context table (keeps ocml's log2 out of this kernel: 64 VGPRs, no spills).
// LDS: lut u16 [n_dirs] | lg f64 [U+1] | cnt u32 [FB][n]
// ------------------------------------------------------------------------------------------
// WEIGHTS: also write the per-frame tile counts (the analyzers' tile_weights).  PAIRS: 16-byte loads,
// two users per lane (even U); otherwise one user per lane with 8-byte loads, any U.
template <bool WEIGHTS, bool PAIRS>
__global__ __launch_bounds__(1024, 8) void k_spatial_u_lds(const SpatialParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int PPT = 2;
    const int NW = blockDim.x >> 6;
    const int FB = p.FPW;                                                        // frames per round
    const long D = p.src.n_dirs;
    uint16_t* lut = (uint16_t*)smem;
    double* lg = (double*)(smem + ((D * 2 + 15) & ~15L));                        // [U+1]
    unsigned* cnt = (unsigned*)(lg + p.U + 1);                                   // [FB][n]
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    for (long i = tid; i < (D + 1) / 2; i += blockDim.x) ((unsigned*)lut)[i] = ((const unsigned*)p.nearest)[i];
    for (int i = tid; i <= p.U; i += blockDim.x) lg[i] = p.log2_tab[i];
    for (int i = tid; i < FB * p.n; i += blockDim.x) cnt[i] = 0u;
    __syncthreads();
    bool bad = false;
    const int ipf = PAIRS ? p.U >> 1 : p.U;                                      // items (pairs or users) per frame
    const float inv_ipf = 1.0f / (float)ipf;
    const long nblocks = ((long)p.T + FB - 1) / FB;
    // PAIRS: the next round's samples are requested before the barriers of this round (the barriers
    // wait for LDS traffic only, see lds_barrier), so HBM loads stay in flight while the waves
    // reduce the round's histograms.
    double2 a[PPT], b[PPT];
    if (PAIRS && (long)blockIdx.x < nblocks) {
        const long f0 = (long)blockIdx.x * FB;
        const int nitems = (int)min((long)FB, (long)p.T - f0) * ipf;
        const double2* mu2 = (const double2*)(p.src.mu + f0 * (long)p.U);
        const double2* mv2 = (const double2*)(p.src.mv + f0 * (long)p.U);
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = tid + k * (int)blockDim.x;
            if (i < nitems) { a[k] = nt_load(mu2 + i); b[k] = nt_load(mv2 + i); }
        }
    }
    for (long blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
        const long f0 = blk * FB;
        const int nf = (int)min((long)FB, (long)p.T - f0);
        const int nitems = nf * ipf;
        if (PAIRS) {
            int2* out2 = (int2*)(p.assign ? p.assign + f0 * (long)p.U : nullptr);
            int near[PPT][2];
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int i = tid + k * (int)blockDim.x;
                near[k][0] = near[k][1] = -1;
                if (i < nitems) {
                    const int id0 = grid_dir(a[k].x, b[k].x, p.src.W, p.src.H, bad);
                    const int id1 = grid_dir(a[k].y, b[k].y, p.src.W, p.src.H, bad);
                    if (id0 >= 0) near[k][0] = (int)lut[id0];
                    if (id1 >= 0) near[k][1] = (int)lut[id1];
                }
            }
            // next round's loads go out ahead of this round's stores
            const long nb = blk + gridDim.x;
            if (nb < nblocks) {
                const long g0 = nb * FB;
                const int nnext = (int)min((long)FB, (long)p.T - g0) * ipf;
                const double2* mu2 = (const double2*)(p.src.mu + g0 * (long)p.U);
                const double2* mv2 = (const double2*)(p.src.mv + g0 * (long)p.U);
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    const int i = tid + k * (int)blockDim.x;
                    if (i < nnext) { a[k] = nt_load(mu2 + i); b[k] = nt_load(mv2 + i); }
                }
            }
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int i = tid + k * (int)blockDim.x;
                if (i < nitems) {
                    const int fl = (int)(((float)i + 0.5f) * inv_ipf);           // exact: i < 2^12
                    unsigned* row = cnt + (size_t)fl * p.n;
                    if (near[k][0] >= 0) atomicAdd(&row[near[k][0]], 1u);
                    if (near[k][1] >= 0) atomicAdd(&row[near[k][1]], 1u);
                    if (out2) nt_store(out2 + i, make_int2(near[k][0], near[k][1]));
                }
            }
        } else {
            const double* mu1 = p.src.mu + f0 * (long)p.U;
            const double* mv1 = p.src.mv + f0 * (long)p.U;
            int* out1 = p.assign ? p.assign + f0 * (long)p.U : nullptr;
            double a[2 * PPT], b[2 * PPT];
#pragma unroll
            for (int k = 0; k < 2 * PPT; ++k) {
                const int i = tid + k * (int)blockDim.x;
                if (i < nitems) { a[k] = __builtin_nontemporal_load(mu1 + i); b[k] = __builtin_nontemporal_load(mv1 + i); }
            }
#pragma unroll
            for (int k = 0; k < 2 * PPT; ++k) {
                const int i = tid + k * (int)blockDim.x;
                if (i < nitems) {
                    const int fl = (int)(((float)i + 0.5f) * inv_ipf);           // exact: i < 2^13
                    const int id0 = grid_dir(a[k], b[k], p.src.W, p.src.H, bad);
                    const int n0 = id0 >= 0 ? (int)lut[id0] : -1;
                    if (n0 >= 0) atomicAdd(&cnt[(size_t)fl * p.n + n0], 1u);
                    if (out1) __builtin_nontemporal_store(n0, out1 + i);
                }
            }
        }
        lds_barrier();
        // entropy (entropy_utils.py:194-211): wave f reduces frame f and clears its histogram
        for (int f = wv; f < nf; f += NW) {
            unsigned* row = cnt + (size_t)f * p.n;
            int np = 0;                            // users present = histogram total (exact)
            for (int t = lane; t < p.n; t += WAVE) np += (int)row[t];
            np = wave_sum(np);
            const double tw = (double)np, lgn = lg[np], inv_tw = 1.0 / tw;
            double h = 0.0;
            double* wout = WEIGHTS ? p.weights + (f0 + f) * (long)p.n : nullptr;
            for (int t = lane; t < p.n; t += WAVE) {
                const unsigned v = row[t];
                if (v) h -= ((double)v * inv_tw) * (lg[v] - lgn);
                if (WEIGHTS) __builtin_nontemporal_store((double)v, wout + t);
                row[t] = 0u;
            }
            h = wave_sum(h);
            if (lane == 0) {
                double hmax = p.hmax;              // entropy_utils.py:201-206
                if (!(tw > (double)p.norm_n) && !p.full_norm) hmax = -tw * (1.0 / tw) * -lgn;   // log2(1/N) = -log2 N
                double e = h / hmax;
                if (np == 0) {
                    e = __builtin_nan("");
                    if (p.status) atomicAdd(&p.status[1], 1);
                }
                p.ent_k[f0 + f] = e;
                if (p.present) p.present[f0 + f] = np;
            }
        }
        lds_barrier();
    }
    if (p.status) {
        const unsigned long long anybad = __ballot(bad);
        if (anybad && lane == 0) atomicAdd(&p.status[0], (int)__popcll(anybad));
    }
}

// ------------------------------------------------------------------------------------------
// k_transition — compute_transition_entropy (entropy_utils.py:213-332) for one frame pair per
// workgroup.  For source tile p with m users in column order a_1 < ... < a_m the reference's
// dict walk reduces to (pinned by oracle/vet_oracle.py):
//   K = 1 + #distinct destinations among a_2..a_m        (a_1 sits alone in an int-keyed bucket)
//   w = 1 if m == 1 else count among a_2..a_m of the destination whose first appearance is
//       latest                                           (stale loop variable, :307-315)
//   cell = -(m/N) * K * (w/m) * log2(w/m),  H = sum cell,  normalised by log2(n) if N > n else
//   log2(N).
// LDS: per tile  first_u, m, K-1, last_fu, w_last (u32 [n] each); hash of (p,c) buckets
//      key/fu/cnt u32 [HS]; pc u32 [U] the packed pairs.  Integer atomics only, so the result
//      does not depend on scheduling.
// ------------------------------------------------------------------------------------------
struct TransParams {
    SampleSrc src;
    int U, T;
    const uint16_t* nearest;
    int n;
    double hmax;                  // n * -(1/n) * log2(1/n)
    double* ent_k;                // [T-1]
    int32_t* pairs;               // [(T-1)*U*2] or null
    int32_t* srccount;            // [(T-1)*n] or null
    int32_t* common;              // [T-1] or null
    int32_t* status;
    const double* log2_tab;       // [4097] log2(k)
    int HS;                       // hash slots (power of two >= 2*U)
    int hs_shift;                 // 32 - log2(HS)
    uint32_t* scratch;            // k_transition_any: per-workgroup slices of 3*HS + 2*U words
    int run_q, run_r;             // k_transition_run: rows per workgroup (quotient, remainder)
};

// Per-tile words of one row in LDS (both transition kernels):
//   acc f64 [2][20]: per row parity (thread 0 finishes row r while the others initialise row r+1):
//                    [0..15] per-wave partial entropy sums, [16] (as u64) users present in both frames
//   first_u, m_cnt, k_cnt, last_fu u32 [n4]
// Row algorithm:
//   (1) every user: tiles of both frames, key = p << 16 | c, first_u[p] = min u, m[p] += 1
//   (2) non-first users: bucket insert (CAS); the creator of a bucket counts it into K[p];
//       bucket first-user = min u, bucket count += 1; the user remembers its slot
//   (3) non-first users that are the first of their bucket: last_fu[p] = max u
//   (4) the user last_fu[p] publishes w[p] = its bucket's count (into first_u[p], free by then)
//   (5) per tile: cell = -(m/N) K (w/m) log2(w/m) = -(K w / N)(log2 w - log2 m), summed per wave (xor
//       butterfly) and over the waves in order: a pure function of the row for a given workgroup size
constexpr int TRANS_ACC = 20;

__device__ __forceinline__ void trans_init(unsigned* tile_words, int n4, unsigned* hkey, unsigned* hfu, unsigned* hcnt, int HS,
                                           double* acc, int bd) {
    const uint4 ones = make_uint4(~0u, ~0u, ~0u, ~0u), zeros = make_uint4(0u, 0u, 0u, 0u);
    const int tid = threadIdx.x;
    for (int i = tid; i < n4 / 4; i += bd) {
        ((uint4*)tile_words)[i] = ones;                                   // first_u
        ((uint4*)tile_words)[i + n4 / 4] = zeros;                         // m_cnt
        ((uint4*)tile_words)[i + 2 * (n4 / 4)] = zeros;                   // k_cnt
        ((uint4*)tile_words)[i + 3 * (n4 / 4)] = zeros;                   // last_fu
    }
    for (int i = tid; i < HS / 4; i += bd) {
        ((uint4*)hkey)[i] = ones; ((uint4*)hfu)[i] = ones; ((uint4*)hcnt)[i] = zeros;
    }
    if (tid == 0) ((unsigned long long*)acc)[16] = 0ull;
}

// step (5) and the row's outputs; all threads call it after step (4) is visible
// LDS_ONLY: the row's shared words are all in LDS, the barrier need not drain global loads.
// VIA_SLOT: first_u[t] holds (user << 13 | bucket slot) of the bucket that gives w, and w = hcnt[slot]
template <bool LDS_ONLY, bool VIA_SLOT = false>
__device__ __forceinline__ void trans_cells(const TransParams& p, long r, const unsigned* first_u, const unsigned* m_cnt,
                                            const unsigned* k_cnt, double* acc, const double* log2_tab, int bd,
                                            const unsigned* hcnt = nullptr) {
    const bool tab = log2_tab != nullptr;
    const int tid = threadIdx.x, NW = bd >> 6;
    const int N = (int)((const unsigned long long*)acc)[16];
    const double inv_n = 1.0 / (double)N;
    double h = 0.0;
    for (int t0 = tid; t0 < p.n; t0 += 2 * bd) {          // two tiles per thread and iteration, their loads issued together
        const int t1 = t0 + bd;
        const bool has1 = t1 < p.n;
        const int ts1 = has1 ? t1 : t0;
        const unsigned m0 = m_cnt[t0], m1 = has1 ? m_cnt[ts1] : 0u;
        const unsigned K0 = 1u + k_cnt[t0], K1 = 1u + k_cnt[ts1];
        const unsigned f0 = first_u[t0], f1 = first_u[ts1];
        unsigned w0, w1;
        if (VIA_SLOT) { w0 = hcnt[f0 & 0x1FFFu]; w1 = hcnt[f1 & 0x1FFFu]; } else { w0 = f0; w1 = f1; }
        w0 = m0 <= 1u ? 1u : w0;
        w1 = m1 <= 1u ? 1u : w1;
        const unsigned d0 = m0 ? m0 : 1u, d1 = m1 ? m1 : 1u;
        double lq0, lq1;
        if (tab) { lq0 = log2_tab[w0] - log2_tab[d0]; lq1 = log2_tab[w1] - log2_tab[d1]; }
        else { lq0 = log2((double)w0 / (double)d0); lq1 = log2((double)w1 / (double)d1); }
        if (m0) h -= ((double)((unsigned long long)K0 * w0) * inv_n) * lq0;
        if (m1) h -= ((double)((unsigned long long)K1 * w1) * inv_n) * lq1;
        if (p.srccount) {
            p.srccount[r * (long)p.n + t0] = (int)m0;
            if (has1) p.srccount[r * (long)p.n + t1] = (int)m1;
        }
    }
    h = wave_sum(h);
    if (lane_id() == 0) acc[wave_id()] = h;
    if (LDS_ONLY) lds_barrier(); else __syncthreads();
    if (tid == 0) {
        double tot = 0.0;
        for (int i = 0; i < NW; ++i) tot += acc[i];
        double hmax = p.hmax;
        if (!(N > p.n)) {
            const double tp = 1.0 / (double)N;          // entropy_utils.py:322-327
            hmax = (double)N * -tp * (tab ? -log2_tab[N] : log2(tp));
        }
        double e = tot / hmax;
        if (N == 0) {
            e = __builtin_nan("");
            if (p.status) atomicAdd(&p.status[1], 1);
        }
        p.ent_k[r] = e;
        if (p.common) p.common[r] = N;
    }
}

// ------------------------------------------------------------------------------------------
// k_transition_any — compute_transition_entropy (entropy_utils.py:213-332) for ANY number of users:
// the bucket hash and the per-user words live in a per-workgroup slice of global scratch (L2
// resident), persistent workgroups loop over the rows; only the per-tile words stay in LDS.
// For source tile p with m users in column order a_1 < ... < a_m the reference's dict walk reduces to
// (pinned by oracle/vet_oracle.py):
//   K = 1 + #distinct destinations among a_2..a_m        (a_1 sits alone in an int-keyed bucket)
//   w = 1 if m == 1 else count among a_2..a_m of the destination whose first appearance is
//       latest                                           (stale loop variable, :307-315)
//   cell = -(m/N) * K * (w/m) * log2(w/m),  H = sum cell,  normalised by log2(n) if N > n else log2(N).
// ------------------------------------------------------------------------------------------
template <bool FROM_IDS>
__global__ void k_transition_any(const TransParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* acc2 = (double*)smem;                              // [2][TRANS_ACC]
    unsigned* first_u = (unsigned*)(acc2 + 2 * TRANS_ACC);     // [n4]
    const int n4 = (p.n + 3) & ~3;
    unsigned* m_cnt = first_u + n4;
    unsigned* k_cnt = m_cnt + n4;
    unsigned* last_fu = k_cnt + n4;
    const int tid = threadIdx.x, lane = lane_id();
    const size_t U4 = ((size_t)p.U + 3) & ~(size_t)3;
    unsigned* hkey = p.scratch + (size_t)blockIdx.x * (3 * (size_t)p.HS + 2 * U4);   // [HS]
    unsigned* hfu = hkey + p.HS;                   // [HS]
    unsigned* hcnt = hfu + p.HS;                   // [HS]
    unsigned* pc = hcnt + p.HS;                    // [U4] the packed pairs
    unsigned* uslot = pc + U4;                     // [U4] bucket slot of every non-first user
    const long R = (long)p.T - 1;
    const bool tab = p.U <= 4096;
    bool bad = false;
    int parity = 0;
    for (long r = blockIdx.x; r < R; r += gridDim.x, parity ^= 1) {
        double* acc = acc2 + TRANS_ACC * parity;
        trans_init(first_u, n4, hkey, hfu, hcnt, p.HS, acc, (int)blockDim.x);       // the barrier inside trans_cells of the previous row precedes
        __syncthreads();
        for (int u = tid; u < p.U; u += blockDim.x) {
            const int ia = sample_dir<FROM_IDS, false>(p.src, r * (long)p.U + u, bad);
            const int ib = sample_dir<FROM_IDS, false>(p.src, (r + 1) * (long)p.U + u, bad);
            unsigned packed = EMPTY_KEY;
            int pa = -1, cb = -1;
            if (ia >= 0 && ib >= 0) {           // user present in both frames (entropy_utils.py:259-261)
                pa = p.nearest[ia]; cb = p.nearest[ib];
                packed = ((unsigned)pa << 16) | (unsigned)cb;
                atomicMin(&first_u[pa], (unsigned)u);
                atomicAdd(&m_cnt[pa], 1u);
            }
            const unsigned long long both = __ballot(packed != EMPTY_KEY);
            if (lane == 0 && both) atomicAdd((unsigned long long*)acc + 16, (unsigned long long)__popcll(both));
            pc[u] = packed;
            if (p.pairs) {      // written once: non-temporal
                __builtin_nontemporal_store(pa, p.pairs + (r * (long)p.U + u) * 2);
                __builtin_nontemporal_store(cb, p.pairs + (r * (long)p.U + u) * 2 + 1);
            }
        }
        __syncthreads();
        for (int u = tid; u < p.U; u += blockDim.x) {
            const unsigned key = pc[u];
            unsigned mark = 0x40000000u;              // absent, or the first user of its source tile
            if (key != EMPTY_KEY && first_u[key >> 16] != (unsigned)u) {
                unsigned h = (key * 2654435761u) >> p.hs_shift;
                for (;;) {
                    const unsigned prev = atomicCAS(&hkey[h], EMPTY_KEY, key);
                    if (prev == EMPTY_KEY) { atomicAdd(&k_cnt[key >> 16], 1u); break; }      // a new destination of this source tile
                    if (prev == key) break;
                    h = (h + 1) & (unsigned)(p.HS - 1);
                }
                atomicMin(&hfu[h], (unsigned)u);
                atomicAdd(&hcnt[h], 1u);
                mark = h;
            }
            uslot[u] = mark;
        }
        __syncthreads();
        for (int u = tid; u < p.U; u += blockDim.x) {
            const unsigned sl = uslot[u];
            if (sl < 0x40000000u && hfu[sl] == (unsigned)u) atomicMax(&last_fu[pc[u] >> 16], (unsigned)u);
        }
        __syncthreads();
        // the user last_fu[p] publishes w[p] = its bucket's count into first_u[p] (nobody reads first_u any more)
        for (int u = tid; u < p.U; u += blockDim.x) {
            const unsigned sl = uslot[u];
            if (sl < 0x40000000u && last_fu[pc[u] >> 16] == (unsigned)u) first_u[pc[u] >> 16] = hcnt[sl];
        }
        __syncthreads();
        trans_cells<false>(p, r, first_u, m_cnt, k_cnt, acc, tab ? p.log2_tab : nullptr, (int)blockDim.x);
    }
    if (p.status) {
        const unsigned long long anybad = __ballot(bad);
        if (anybad && lane == 0) atomicAdd(&p.status[0], (int)__popcll(anybad));
    }
}

// ------------------------------------------------------------------------------------------
// k_transition_run — the same rows for U <= UPT * blockDim users (everything in LDS), software
// pipelined: a persistent workgroup takes a contiguous RUN of rows.  The current frame's tiles of row r
// stay in registers as the prior frame's tiles of row r+1 (every frame is read and quantised once
// instead of twice), and the samples of frame r+2 are requested before the bucket phases of row r, so
// the HBM latency hides behind LDS work.  Thread t owns users t, t + blockDim, ...  The kernel is
// bound by instruction issue, and the per-row fixed work (initialisation, barriers, the tile phase) is
// paid per wave: two waves with four users per lane measured best at 512 users.
// ------------------------------------------------------------------------------------------
// branch-free grid_dir for the straight-line row loop below (same results)
__device__ __forceinline__ int grid_dir_sel(double m, double v, int W, int H, bool& bad) {
    const bool ordered = (m == m) & (v == v);
    const bool in = (m >= 0.0) & (m <= 1.0) & (v >= 0.0) & (v <= 1.0);       // false for NaN
    bad |= ordered & !in;
    const double ms = in ? m : 0.0, vs = in ? v : 0.0;
    const int id = (int)(vs * (double)H) * (W + 1) + (int)(ms * (double)W);
    return in ? id : -1;
}

// EXACT: U == UPT * blockDim, no bounds checks on the user index; THREADS: the workgroup size when it is a
// compile-time constant (0: read blockDim)
template <bool FROM_IDS, int UPT, bool EXACT, int THREADS>
__global__ void k_transition_run(const TransParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* acc2 = (double*)smem;                              // [2][TRANS_ACC]
    unsigned* first_u = (unsigned*)(acc2 + 2 * TRANS_ACC);     // [n4]
    const int n4 = (p.n + 3) & ~3;
    unsigned* m_cnt = first_u + n4;
    unsigned* k_cnt = m_cnt + n4;
    unsigned* last_fu = k_cnt + n4;
    unsigned* hkey = last_fu + n4;                 // [HS]
    unsigned* hfu = hkey + p.HS;
    unsigned* hcnt = hfu + p.HS;
    const int tid = threadIdx.x, lane = lane_id();
    const int BD = THREADS ? THREADS : (int)blockDim.x;
    const long R = (long)p.T - 1;
    // runs of run_q or run_q + 1 rows (the first run_r workgroups take the longer ones): R = run_q * gridDim + run_r
    const long b = blockIdx.x;
    const long r_begin = b * p.run_q + (b < p.run_r ? b : (long)p.run_r);
    const long r_end = r_begin + p.run_q + (b < p.run_r ? 1 : 0);
    if (r_begin >= r_end || r_end > R) return;
    bool bad = false;
    int prev[UPT], cur[UPT];
    double sa[UPT], sb[UPT];                       // samples of the frame after the current one, in flight
    int si[UPT];
    bool mine[UPT];
#pragma unroll
    for (int k = 0; k < UPT; ++k) mine[k] = EXACT || tid + k * BD < p.U;
    // the next frame to request: running pointers (one 64-bit add per row instead of a 64-bit multiply per load)
    const double* next_mu = FROM_IDS ? nullptr : p.src.mu + r_begin * (long)p.U + tid;
    const double* next_mv = FROM_IDS ? nullptr : p.src.mv + r_begin * (long)p.U + tid;
    const int32_t* next_id = FROM_IDS ? p.src.ids + r_begin * (long)p.U + tid : nullptr;
    // Loads are issued unconditionally (a lane without a user reads user U-1, the row after the run's last one is
    // replaced by the last one): the compiler can then count the loads in flight and the waits for the nearest-tile
    // gathers of this row do not drain the prefetch of the next one (a conditional load forces s_waitcnt vmcnt(0)).
    int off[UPT];
#pragma unroll
    for (int k = 0; k < UPT; ++k) off[k] = mine[k] ? k * BD : p.U - 1 - tid;
    auto request = [&](bool valid) {               // issue the loads of the next frame (of the one before it if !valid)
        const long back = valid ? 0 : -(long)p.U;
#pragma unroll
        for (int k = 0; k < UPT; ++k) {
            if (FROM_IDS) si[k] = next_id[back + off[k]];
            else { sa[k] = next_mu[back + off[k]]; sb[k] = next_mv[back + off[k]]; }
        }
        if (FROM_IDS) next_id += p.U; else { next_mu += p.U; next_mv += p.U; }
    };
    auto tiles_of = [&](int* out) {                // requested samples -> direction ids -> nearest tiles (-1 absent)
        int id[UPT];
#pragma unroll
        for (int k = 0; k < UPT; ++k) {
            bool b = false;
            if (FROM_IDS) {
                b = si[k] >= p.src.n_dirs;
                id[k] = b ? -1 : si[k];
            } else {
                id[k] = grid_dir_sel(sa[k], sb[k], p.src.W, p.src.H, b);
            }
            if (!mine[k]) id[k] = -1;
            bad |= b & mine[k];
        }
        unsigned short t[UPT];
#pragma unroll
        for (int k = 0; k < UPT; ++k) t[k] = p.nearest[id[k] < 0 ? 0 : id[k]];       // unconditional loads, then selects
#pragma unroll
        for (int k = 0; k < UPT; ++k) out[k] = id[k] < 0 ? -1 : (int)t[k];
    };
    // log2(k), k <= U, in LDS: the cell phase then has no global loads, whose wait would drain the prefetch too
    double* l2 = (double*)(hcnt + p.HS);
    for (int i = tid; i <= p.U; i += BD) l2[i] = p.log2_tab[i];
    request(true);
    tiles_of(prev);
    request(true);
    int32_t* pairs_row = p.pairs ? p.pairs + (r_begin * (long)p.U + tid) * 2 : nullptr;
    int parity = 0;
    for (long r = r_begin; r < r_end; ++r, parity ^= 1) {
        double* acc = acc2 + TRANS_ACC * parity;
        trans_init(first_u, n4, hkey, hfu, hcnt, p.HS, acc, BD);      // the barrier inside trans_cells of the previous row precedes
        tiles_of(cur);
        request(r + 1 < r_end);                    // in flight during this row's LDS phases: the barriers below wait for
                                                   // LDS traffic only (lds_barrier), not for these loads
        lds_barrier();
        unsigned key[UPT];
        int present = 0;
#pragma unroll
        for (int k = 0; k < UPT; ++k) {
            const unsigned u = (unsigned)(tid + k * BD);
            key[k] = EMPTY_KEY;
            const bool both = prev[k] >= 0 && cur[k] >= 0;      // present in both frames (entropy_utils.py:259-261)
            if (both) {
                key[k] = ((unsigned)prev[k] << 16) | (unsigned)cur[k];
                if (first_u[prev[k]] > u) atomicMin(&first_u[prev[k]], u);      // later users of a crowded tile skip the atomic
                atomicAdd(&m_cnt[prev[k]], 1u);
            }
            present += (int)__popcll(__ballot(both));
            if (pairs_row && mine[k]) {       // written once: non-temporal (the compiler merges the two into one 8-byte store)
                __builtin_nontemporal_store(both ? prev[k] : -1, pairs_row + 2 * k * BD);
                __builtin_nontemporal_store(both ? cur[k] : -1, pairs_row + 2 * k * BD + 1);
            }
        }
        if (pairs_row) pairs_row += 2 * (long)p.U;
        if (lane == 0 && present) atomicAdd((unsigned long long*)acc + 16, (unsigned long long)present);
        lds_barrier();
        unsigned slot[UPT];
        bool nonfirst[UPT];
#pragma unroll
        for (int k = 0; k < UPT; ++k) {
            const unsigned u = (unsigned)(tid + k * BD);
            nonfirst[k] = key[k] != EMPTY_KEY && first_u[key[k] >> 16] != u;
            slot[k] = 0;
            if (nonfirst[k]) {
                unsigned h = (key[k] * 2654435761u) >> p.hs_shift;
                for (;;) {
                    const unsigned was = atomicCAS(&hkey[h], EMPTY_KEY, key[k]);
                    if (was == EMPTY_KEY) { atomicAdd(&k_cnt[key[k] >> 16], 1u); break; }      // a new destination of this source tile
                    if (was == key[k]) break;
                    h = (h + 1) & (unsigned)(p.HS - 1);
                }
                if (hfu[h] > u) atomicMin(&hfu[h], u);
                atomicAdd(&hcnt[h], 1u);
                slot[k] = h;
            }
        }
        lds_barrier();
#pragma unroll
        for (int k = 0; k < UPT; ++k) {
            const unsigned u = (unsigned)(tid + k * BD);
            // w of a source tile = count of the bucket whose first user is the latest: user << 13 | slot (U <= 4096, HS <= 8192)
            if (nonfirst[k] && hfu[slot[k]] == u) atomicMax(&last_fu[key[k] >> 16], (u << 13) | slot[k]);
        }
        lds_barrier();
        trans_cells<true, true>(p, r, last_fu, m_cnt, k_cnt, acc, l2, BD, hcnt);
#pragma unroll
        for (int k = 0; k < UPT; ++k) prev[k] = cur[k];
    }
    if (p.status) {
        const unsigned long long anybad = __ballot(bad);
        if (anybad && lane == 0) atomicAdd(&p.status[0], (int)__popcll(anybad));
    }
}

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

}  // namespace vet
