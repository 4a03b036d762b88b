// vet_spatial_rows.hpp — k_spatial_rows: FoV-weighted spatial entropy through the FUSED direction weight table
// (one row per distinct direction over ALL lattices of the plan), persistent workgroups, per-direction records in LDS.
// Part of the gfx950 device code of the viewport -> tile -> entropy path (see vet_kernels.hpp for the map).
// Reference citations are relative to /root/reference/src/viewport_entropy_toolkit/.
#pragma once
#include "vet_spatial_lut.hpp"

namespace vet {

constexpr int ROWS_THREADS = 1024;      // one workgroup per CU: the record table takes most of the LDS
constexpr int ROWS_SPT = 2;             // samples per thread and round at most
constexpr int ROWS_LEN_BITS = 11;       // lens word = entries in use | row shift << 11
constexpr int ROWS_RING = 3;            // rounds in flight: samples -> set (i), walk (i-1), entropy (i-2)

struct RowsParams {
    const VideoDesc* videos;      // null: the single video described below; else block0 = first round of the video
    int n_videos;
    const double* mu;
    const double* mv;
    int U, T, W, H;
    const uint32_t* rec;          // [D] direction -> fused row (15 bits) | mirrored << 15 | nearest tile of lattice 0 << 16
    const uint16_t* lens;         // [R+2] entries in use | row shift << 11   (row R = the all-zero row)
    int D, R;
    const uint32_t* tab_w;        // [R+1][stride] u32 mantissas (block floating point per row, vet_weight_table.hpp)
    const uint16_t* tab_i;        // [R+1][stride] fused histogram slots
    int stride, gs_log2, interleaved;
    FusedLayout lay;
    double* entropy;              // [T] mean over the K lattices, summed in lattice order
    int32_t* assign;
    double* weights;              // lattice 0, [T][n_0]
    int32_t* present;
    int32_t* status;
    int FB;                       // frames per round (FB * U <= ROWS_SPT * ROWS_THREADS)
    int HS;                       // hash slots per frame: power of two >= 2 * max U
    int UCAP;                     // row-list capacity per frame (max U)
    long n_rounds;
    unsigned long long* dbg;      // VET_ROWS_DEBUG: [8] cycles per phase of the workgroups' first waves (null: off)
};

__host__ __device__ __forceinline__ size_t rows_lds_static(int D, int R) {
    return (((size_t)D * 4 + 15) & ~(size_t)15) + (((size_t)(R + 2) * 2 + 15) & ~(size_t)15);
}
// two set / list buffers and two histogram buffers (the rounds of a workgroup are pipelined), small rings
__host__ __device__ __forceinline__ size_t rows_lds_round(int FB, int HS, int UCAP, int N, int CF) {
    return 2 * ((size_t)FB * N * 8 + (size_t)FB * HS * 4 + (size_t)FB * UCAP * 4) + (size_t)ROWS_RING * (2 * FB + 2) * 4 +
           (size_t)FB * CF * 8 + (size_t)FB * 4 + 64;
}

// ------------------------------------------------------------------------------------------
// k_spatial_rows — compute_spatial_entropy (entropy_utils.py:147-211), FoV-weighted, integer table formulation.
// One persistent workgroup of 1024 threads per CU; rounds of FB frames (FB * U <= 2048 samples) dealt round
// robin; the rounds of a workgroup are PIPELINED, one workgroup barrier per round: in phase i the waves work on
//   P(i)    the samples of round i (requested from HBM during phase i-1, held in registers) -> direction id ->
//           LDS record -> nearest tile out, insert into the frame's set of distinct rows; then request round i+1
//   W(i-1)  walk of the distinct rows of round i-1: row tasks of 64/GS * UN rows from an LDS queue; a group of GS
//           lanes walks one row, ds_add_u64 of entry * (multiplicity << shift) into the frame's fused histogram
//   E(i-2)  entropy of the frames of round i-2 (one wave per frame), which also clears the histogram
// in any order — the three touch different buffers — so a wave that waits on LDS round trips in P or E leaves the
// L2 -> L1 path to the waves that walk.  Every wave does its P part at its own point of the row queue.
// LDS:  rec u32 [D] | lens u16 [R+2] | hist u64 [2][FB][N] | set u32 [2][FB][HS] | list u32 [2][FB][UCAP]
//       | rings of 3: rows per frame, users present per frame, row-queue head
//   rec: direction -> row | mirrored | nearest tile (81 KB on the 100 x 200 grid: the per-sample record gather of
//        k_spatial_lut — a 128-byte line from L2 for 8 bytes — is a ds_read here)
//   set: slot = (row | mirrored << 19) << 12 | multiplicity; the walk resets a slot to EMPTY as it consumes it
// Integer adds commute, the frame total is an exact integer and -sum p log2 p runs in a canonical order
// (64-tile chunks): the result does not depend on scheduling, on the order of users or on how the frames are
// split over rounds, videos or GPUs.
// ------------------------------------------------------------------------------------------
template <bool IL, int UN>
__global__ __launch_bounds__(ROWS_THREADS) void k_spatial_rows(const RowsParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NW = ROWS_THREADS / 64;
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int FB = p.FB, HS = p.HS, UCAP = p.UCAP, N = p.lay.N, K = p.lay.K;
    uint32_t* rec = (uint32_t*)smem;
    uint16_t* lens = (uint16_t*)(smem + (((size_t)p.D * 4 + 15) & ~(size_t)15));
    unsigned long long* hist_b = (unsigned long long*)(smem + rows_lds_static(p.D, p.R));    // [2][FB][N]
    uint32_t* set_b = (uint32_t*)(hist_b + (size_t)2 * FB * N);                              // [2][FB][HS]
    uint32_t* list_b = set_b + (size_t)2 * FB * HS;                                          // [2][FB][UCAP]
    int* ring = (int*)(list_b + (size_t)2 * FB * UCAP);                                      // [3][2 FB + 2]
    const int RS = 2 * FB + 2;              // ring slot: rows [FB], present [FB], queue head, spare
    const int CF = p.lay.CF;
    int* e_done = ring + ROWS_RING * RS;                                                     // [FB] chunk tasks finished
    double* chunk_h = (double*)(((uintptr_t)(e_done + FB) + 7) & ~(uintptr_t)7);             // [FB][CF]

    for (int i = tid; i < p.D; i += ROWS_THREADS) rec[i] = p.rec[i];
    for (int i = tid; i < (p.R + 2) / 2; i += ROWS_THREADS) ((uint32_t*)lens)[i] = ((const uint32_t*)p.lens)[i];
    for (int i = tid; i < 2 * FB * N; i += ROWS_THREADS) hist_b[i] = 0ull;
    for (int i = tid; i < 2 * FB * HS; i += ROWS_THREADS) set_b[i] = EMPTY_KEY;
    for (int i = tid; i < ROWS_RING * RS + FB; i += ROWS_THREADS) ring[i] = 0;
    const int hs_shift = 32 - (31 - __clz(HS | 1));
    const uint32_t zero_row = (uint32_t)p.R * (uint32_t)p.stride;
    bool bad = false;

    // the video and frames of a round
    struct Round { const double* mu; const double* mv; int U, T; long f0; int nf; double* entropy; int32_t* assign; int32_t* present; double* weights; };
    auto locate = [&](long round) {
        Round r;
        r.mu = p.mu; r.mv = p.mv; r.U = p.U; r.T = p.T;
        r.entropy = p.entropy; r.assign = p.assign; r.present = p.present; r.weights = p.weights;
        long blk = round;
        if (p.videos) {
            int lo = 0, hi = p.n_videos - 1;                   // last video with block0 <= round
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if ((long)p.videos[mid].block0 <= round) lo = mid; else hi = mid - 1;
            }
            const VideoDesc& d = p.videos[lo];
            r.mu = d.mu; r.mv = d.mv; r.U = d.U; r.T = d.T;
            r.entropy = d.entropy; r.assign = d.assign; r.present = d.present; r.weights = nullptr;
            blk -= d.block0;
        }
        r.f0 = blk * FB;
        r.nf = (int)min((long)FB, (long)r.T - r.f0);
        return r;
    };

    double a[ROWS_SPT], b[ROWS_SPT];
    auto request = [&](const Round& r) {                       // the round's samples -> registers (no wait)
        const int total = r.nf * r.U;
        const double* mu = r.mu + r.f0 * (long)r.U;
        const double* mv = r.mv + r.f0 * (long)r.U;
#pragma unroll
        for (int k = 0; k < ROWS_SPT; ++k) {
            const int i = tid + k * ROWS_THREADS;
            a[k] = b[k] = __builtin_nan("");
            if (i < total) {
                a[k] = __builtin_nontemporal_load(mu + i);
                b[k] = __builtin_nontemporal_load(mv + i);
            }
        }
    };
    const long first = blockIdx.x, stride_r = gridDim.x;
    const long n_local = first < p.n_rounds ? (p.n_rounds - first + stride_r - 1) / stride_r : 0;
    if (n_local > 0) request(locate(first));
    unsigned long long tph[6] = {0, 0, 0, 0, 0, 0}, tprev = p.dbg ? __builtin_readcyclecounter() : 0ull;
    auto mark = [&](int ph) {
        if (p.dbg) { const unsigned long long now = __builtin_readcyclecounter(); tph[ph] += now - tprev; tprev = now; }
    };
    const int gs_log2 = IL ? 4 : p.gs_log2;
    const int GS = 1 << gs_log2, UPW = WAVE >> gs_log2, RPT = UPW * UN;       // rows per walk task
    const int sub = lane >> gs_log2, sl = lane & (GS - 1);
    const int cut = IL ? min(4 * sl, 3 * GS - 1) : 4 * sl;
    lds_barrier();

    for (long i = 0; i < n_local + 2; ++i) {
        const int rp = (int)(i % ROWS_RING), rw = (int)((i + ROWS_RING - 1) % ROWS_RING), re = (int)((i + ROWS_RING - 2) % ROWS_RING);
        int* ring_p = ring + rp * RS;                          // written by P(i)
        int* ring_w = ring + rw * RS;                          // rows of round i-1, queue head of W(i-1)
        int* ring_e = ring + re * RS;                          // present of round i-2; reset by E(i-2)
        const bool has_p = i < n_local, has_w = i >= 1 && i - 1 < n_local, has_e = i >= 2;
        // ---- P(i): this thread's samples of round i
        auto do_p = [&]() {
            const Round cur = locate(first + i * stride_r);
            const int U = cur.U, nf = cur.nf, total = nf * U;
            const float inv_u = 1.0f / (float)U;
            uint32_t* set = set_b + (size_t)(i & 1) * FB * HS;
            uint32_t* list = list_b + (size_t)(i & 1) * FB * UCAP;
#pragma unroll
            for (int k = 0; k < ROWS_SPT; ++k) {
                const int s = tid + k * ROWS_THREADS;
                if (k * ROWS_THREADS >= total) break;          // uniform
                const int fl = (int)(((float)s + 0.5f) * inv_u);   // exact: s < 2^12
                const int id = grid_dir(a[k], b[k], p.W, p.H, bad);     // NaN registers beyond the round: -1
                const bool valid = id >= 0;
                uint32_t key = 0u;
                int near = -1;
                if (valid) {
                    const uint32_t r = rec[id];
                    key = (r & 0x7FFFu) | (((r >> 15) & 1u) << ROW_BITS);
                    near = (int)(r >> 16);
                }
                if (cur.assign && s < total) __builtin_nontemporal_store(near, cur.assign + cur.f0 * (long)U + s);
                const int fl0 = __builtin_amdgcn_readfirstlane(fl);
                const bool uniform = __ballot(fl != fl0) == 0ull;
                bool won = false;
                unsigned h = 0;
                if (valid) {
                    uint32_t* tab = set + (size_t)fl * HS;
                    h = (key * 2654435761u) >> hs_shift;
                    for (;;) {
                        unsigned c = tab[h];
                        if (c == EMPTY_KEY) {
                            c = atomicCAS(&tab[h], EMPTY_KEY, (key << 12) | 1u);
                            if (c == EMPTY_KEY) { won = true; break; }
                        }
                        if ((c >> 12) == key) { atomicAdd(&tab[h], 1u); break; }
                        h = (h + 1) & (unsigned)(HS - 1);
                    }
                }
                const unsigned long long mv_ = __ballot(valid), mw = __ballot(won);
                if (uniform) {
                    int base = 0;
                    if (lane == 0) {
                        if (mv_) atomicAdd(&ring_p[FB + fl0], (int)__popcll(mv_));
                        if (mw) base = atomicAdd(&ring_p[fl0], (int)__popcll(mw));
                    }
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (won) list[(size_t)fl0 * UCAP + base + below(mw)] = h;
                } else {
                    if (valid) atomicAdd(&ring_p[FB + fl], 1);
                    if (won) list[(size_t)fl * UCAP + atomicAdd(&ring_p[fl], 1)] = h;
                }
            }
            // the next round's samples: requested now, consumed in the next phase
            if (i + 1 < n_local) request(locate(first + (i + 1) * stride_r));
        };
        // ---- E(i-2), one task per (frame, lattice, 64-tile chunk): -sum p log2 p of the chunk against the frame's
        // exact total (entropy_utils.py:194-211, weighted: normaliser log2 n); the task that finishes a frame last adds
        // the chunk sums in a canonical order, forms the mean in lattice order (spatial_entropy.py:142-156) and
        // writes the frame out.  Every task clears the histogram slots it has read.
        const Round old = has_e ? locate(first + (i - 2) * stride_r) : Round();
        const int n_e = has_e ? old.nf * CF : 0;
        auto do_e = [&](int e) {
            unsigned long long* hist = hist_b + (size_t)(i & 1) * FB * N;
            const double inv_unit = 1.0 / (4294967296.0 * (double)(1u << TAB_X));
            const int fl = e / CF;
            int k = 0, c = e - fl * CF;
            for (; k < K - 1; ++k) {
                const int ck = (p.lay.n[k] + WAVE - 1) / WAVE;
                if (c < ck) break;
                c -= ck;
            }
            unsigned long long* hrow = hist + (size_t)fl * N;
            const int n = p.lay.n[k], hh = n >> 1;
            // exact total of lattice k: hi * 2^32 + lo from the rows' pseudo entries (mirrored rows: N-1-slot)
            const unsigned long long hi = hrow[2 * k] + hrow[N - 1 - 2 * k], lo = hrow[2 * k + 1] + hrow[N - 2 - 2 * k];
            const double totd = (double)(hi + (lo >> 32)) * 4294967296.0 + (double)(lo & 0xFFFFFFFFull);
            const int t = c * WAVE + lane;
            double term = 0.0;
            if (t < n) {
                const int pos = fused_pos(p.lay, k, t);
                unsigned long long v = hrow[pos];
                hrow[pos] = 0ull;
                if (t >= hh && t < n - hh) { v += hrow[N - 1 - pos]; hrow[N - 1 - pos] = 0ull; }     // centre tile: both slots
                if (v != 0ull) {
                    const double qv = (double)v / totd;
                    term = -(qv * log2(qv));
                }
                if (k == 0 && old.weights) __builtin_nontemporal_store((double)v * inv_unit, old.weights + (old.f0 + fl) * (long)n + t);
            }
            term = wave_total(term);
            int last = 0;
            if (lane == 0) {
                chunk_h[fl * CF + (e - fl * CF)] = term;
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                last = atomicAdd(&e_done[fl], 1) == CF - 1;
            }
            if (__builtin_amdgcn_readfirstlane(last)) {
                if (lane == 0) {
                    double total_entropy = 0.0;
                    int j = 0;
                    for (int kk = 0; kk < K; ++kk) {
                        double hk = 0.0;
                        for (int cc = 0; cc * WAVE < p.lay.n[kk]; ++cc) hk += chunk_h[fl * CF + j++];
                        total_entropy += hk / p.lay.hmax[kk];
                    }
                    const int np = ring_e[FB + fl];
                    double en = total_entropy / (double)K;
                    if (np == 0) {
                        en = __builtin_nan("");
                        if (p.status) atomicAdd(&p.status[1], 1);
                    }
                    old.entropy[old.f0 + fl] = en;
                    if (old.present) old.present[old.f0 + fl] = np;
                    ring_e[fl] = 0; ring_e[FB + fl] = 0; e_done[fl] = 0;
                }
                if (lane < 4 * K) hrow[lane < 2 * K ? lane : N - 4 * K + lane] = 0ull;               // the total slots
            }
        };
        mark(0);
        if (wv == NW - 1 && lane == 0 && has_e) ring_e[2 * FB] = 0;        // queue head of the ring slot P(i+1) reuses
        mark(1);
        // ---- W(i-1): row tasks from the queue; this wave's P at its own point of the queue
        bool p_done = !has_p;
        if (has_w || has_e) {
            uint32_t* set_wr = set_b + (size_t)((i - 1) & 1) * FB * HS;
            const uint32_t* list = list_b + (size_t)((i - 1) & 1) * FB * UCAP;
            unsigned long long* hist = hist_b + (size_t)((i - 1) & 1) * FB * N;
            const int nfw = FB;                                  // frames without rows have no tasks
            int nw_tasks = 0;
            if (has_w)
                for (int f = 0; f < nfw; ++f) nw_tasks += (ring_w[f] + RPT - 1) / RPT;
            const int ntasks = n_e + nw_tasks;                   // the queue: entropy chunks first, then row tasks
            const int my_point = n_e + (int)(((long)wv * nw_tasks * 3) / (4 * NW));
            for (;;) {
                int c = 0;
                if (lane == 0) c = atomicAdd(&ring_w[2 * FB], 1);
                c = __builtin_amdgcn_readfirstlane(c);
                if (c < n_e) { do_e(c); continue; }
                if (!p_done && c >= my_point) { do_p(); p_done = true; }
                if (c >= ntasks) break;
                int fl = 0, base = c - n_e;                      // task -> (frame, first row)
                for (; fl < nfw; ++fl) {
                    const int tf = (ring_w[fl] + RPT - 1) / RPT;
                    if (base < tf) break;
                    base -= tf;
                }
                const int nu = ring_w[fl];
                uint32_t* setf = set_wr + (size_t)fl * HS;
                const uint32_t* listf = list + (size_t)fl * UCAP;
                unsigned long long* hrow = hist + (size_t)fl * N;
                uint32_t row[UN], mult[UN];
                int lim[UN], sgn[UN];
                char* hb[UN];
                int longest = 0;
#pragma unroll
                for (int k = 0; k < UN; ++k) {
                    const int j = base * RPT + k * UPW + sub;
                    const bool on = j < nu;
                    uint32_t pk = 0u;
                    if (on) {
                        const uint32_t slot = listf[j];
                        pk = setf[slot];
                        if (sl == 0) setf[slot] = EMPTY_KEY;             // consumed: the set is empty again for round i+1
                    }
                    const uint32_t key = pk >> 12;
                    const bool flip = ((key >> ROW_BITS) & 1u) != 0u;
                    const uint32_t rid = key & ROW_MASK;
                    const uint32_t m = on ? (uint32_t)lens[rid] : 0u;
                    row[k] = on ? rid * (uint32_t)p.stride : zero_row;
                    const int len = (int)(m & ((1u << ROWS_LEN_BITS) - 1u));
                    lim[k] = len - cut;
                    mult[k] = (pk & 0xFFFu) << (m >> ROWS_LEN_BITS);
                    sgn[k] = flip ? -8 : 8;
                    hb[k] = (char*)hrow + (flip ? (N - 1) * 8 : 0);
                    longest = max(longest, len);
                }
                for (int eb = 0; eb < longest; eb += 4 * GS) {
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
                        w[k] = *(const uint4*)(p.tab_w + r[k]);
                        t[k] = *(const ushort4*)(p.tab_i + r[k]);
                    }
#pragma unroll
                    for (int k = 0; k < UN; ++k) {
                        atomicAdd((unsigned long long*)(hb[k] + (int)t[k].x * sgn[k]), (unsigned long long)w[k].x * mult[k]);
                        atomicAdd((unsigned long long*)(hb[k] + (int)t[k].y * sgn[k]), (unsigned long long)w[k].y * mult[k]);
                        atomicAdd((unsigned long long*)(hb[k] + (int)t[k].z * sgn[k]), (unsigned long long)w[k].z * mult[k]);
                        atomicAdd((unsigned long long*)(hb[k] + (int)t[k].w * sgn[k]), (unsigned long long)w[k].w * mult[k]);
                    }
                }
            }
        }
        mark(2);
        if (!p_done) do_p();
        mark(3);
        lds_barrier();
        mark(4);
    }
    if (p.dbg && tid == 0)
        for (int i = 0; i < 6; ++i) atomicAdd(&p.dbg[i], tph[i]);
    if (p.status) {
        const unsigned long long anybad = __ballot(bad);
        if (anybad && lane == 0) atomicAdd(&p.status[0], (int)__popcll(anybad));
    }
}

}  // namespace vet
