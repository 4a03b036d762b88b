// vet_spatial_runs.hpp — the two-kernel form of the integer table formulation for long videos of many users:
//   k_runs          stream: samples -> direction ids -> table rows (LDS tables), per GROUP of F consecutive frames the
//                   distinct (row, mirrored, frame) triples with their user counts, sorted by row, as 8-byte units
//   k_spatial_walk  walk: a workgroup per frame group adds every distinct row ONCE per group into the histograms of
//                   the frames that look in that direction, then the entropies
// Part of the gfx950 device code of the viewport -> tile -> entropy path (see vet_kernels.hpp for the map).
// Reference citations are relative to /root/reference/src/viewport_entropy_toolkit/.
//
// Why.  k_spatial_lut (one kernel, a frame per workgroup) runs at the L2 -> L1 line rate of a CU: per frame of 1024
// users it moves 4 266 lines of table rows, 1 024 lines of per-direction records (8 bytes used of 128) and the sample
// stream, all through the same 64 outstanding L1 misses.  Here the records become two LDS tables of a persistent
// streaming kernel (no gather at all: 20 B per sample at HBM speed), and consecutive frames share row loads: the users
// of a frame group that look in the same direction in DIFFERENT frames (a viewer who does not move; viewers crossing
// each other's path) cost one row load.  Measured on the bench's random walk: distinct rows of 8 frames / sum of the
// frames' distinct rows = 0.60 (0.74 at 4 frames, 0.45 at 16; a clustered audience: 0.20 at 8).
// The histograms are integer sums — order independent — so the result is bit-identical to k_spatial_lut's.
#pragma once
#include "vet_spatial_lut.hpp"

namespace vet {

// One unit (16 bytes) = one table row with up to six (frame, mirrored, users) items of a frame group; a row with more
// items continues in the next unit; rows ascending:
//   x = row (19 bits)
//   y, z, w = items 0..5, 16 bits each;  item = users (11 bits, >= 1) | frame in group (4 bits) << 11 | mirrored << 15;  0 = none
constexpr int RUN_CNT_BITS = 11, RUN_MAX_FRAMES = 16, RUN_UNIT_ITEMS = 6;
constexpr int RUNS_THREADS = 1024, RUNS_SPT = 8;            // k_runs: samples per thread and round (F * U <= 8192)
constexpr int RUNS_MAX_CHUNKS = 8;                          // bitmap words <= 8 * 1024

struct RunsParams {
    const double* mu;
    const double* mv;
    int U, T, W, H;
    const uint2* dirrec;          // [D] per-direction record (k_dirrec): row | nearest tile | mirrored
    int D, R;
    int F, lgF;                   // frames per group (power of two <= 16)
    int BW;                       // bitmap words: ceil(R * 2F / 32)
    int G;                        // frame groups = ceil(T / F)
    long cap;                     // units reserved per group (F * U)
    uint4* units;                 // [G][cap]
    int32_t* n_units;             // [G]
    int32_t* assign;              // [T*U] or null
    int32_t* present;             // [T] users present per frame (always written: the walk's epilogue reads it)
    int32_t* status;
    unsigned long long* dbg;      // development builds (-DVET_STAGE_CYCLES=1): [8] cycles of thread 0 per stage
};

// LDS bytes of k_runs; the kernel and the host must agree
__host__ __device__ __forceinline__ size_t runs_lds_bytes(int BW, long cap) {
    return (size_t)BW * 4 /* bitmap */ + (size_t)BW * 4 /* item | unit prefix */ + (((size_t)cap + 1) / 2) * 4 /* counts */ +
           (RUN_MAX_FRAMES + RUNS_MAX_CHUNKS * (RUNS_THREADS / WAVE) + 8) * 4;
}

// k_runs: two persistent workgroups per CU (56 KB of LDS each at F = 8, 1024 users), a round = one frame group:
//   keys    samples (prefetched a round ahead) -> direction id -> per-direction record (one 8-byte gather from a 160 KB,
//           L2-resident table) -> bit (row, mirrored, frame) of the group's bitmap; nearest tile out
//   scan    items (set bits) and units before every bitmap word, in word order; word w belongs to thread w mod 1024, so
//           the crowded part of the sphere is spread over all threads
//   counts  users per item (rank = set bits before the item's own)
//   emit    one unit per row and six items
// PAIRS: U even, 16-byte loads of two users.
template <bool PAIRS>
__global__ __launch_bounds__(RUNS_THREADS) void k_runs(const RunsParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint32_t* bm = (uint32_t*)smem;                                   // [BW] bit (row * 2F + mirrored * F + frame)
    uint32_t* pre = bm + p.BW;                                        // [BW] items before the word | units before it << 16
    uint32_t* cnt = pre + p.BW;                                       // [(cap+1)/2] users per item, two 16-bit halves per word
    uint32_t* pres = cnt + (p.cap + 1) / 2;                           // [16] users present per frame of the group
    uint32_t* wtot = pres + RUN_MAX_FRAMES;                           // [chunks][16] per-wave totals of the scan | [1] total
    constexpr int NWV = RUNS_THREADS / WAVE;
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const int F = p.F, F2 = 2 * F;
    for (int i = tid; i < p.BW; i += RUNS_THREADS) bm[i] = 0u;
    for (int i = tid; i < (int)((p.cap + 1) / 2); i += RUNS_THREADS) cnt[i] = 0u;
    if (tid < RUN_MAX_FRAMES) pres[tid] = 0u;
    __syncthreads();
    bool bad = false;
    const int ipf = PAIRS ? p.U >> 1 : p.U;                            // load items (pairs or users) per frame
    constexpr int LPT = PAIRS ? RUNS_SPT / 2 : RUNS_SPT;              // load items per thread and round
    double2 a2[PAIRS ? LPT : 1], b2[PAIRS ? LPT : 1];
    double a1[PAIRS ? 1 : LPT], b1[PAIRS ? 1 : LPT];
    auto request = [&](long g) {                                       // the samples of frame group g -> registers
        const long f0 = g * F;
        const int nf = (int)min((long)F, (long)p.T - f0), nitems = nf * ipf;
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            const int i = tid + k * RUNS_THREADS;
            if (i < nitems) {
                if (PAIRS) { a2[k] = nt_load((const double2*)(p.mu + f0 * (long)p.U) + i); b2[k] = nt_load((const double2*)(p.mv + f0 * (long)p.U) + i); }
                else { a1[k] = __builtin_nontemporal_load(p.mu + f0 * (long)p.U + i); b1[k] = __builtin_nontemporal_load(p.mv + f0 * (long)p.U + i); }
            }
        }
    };
    if ((long)blockIdx.x < p.G) request(blockIdx.x);
#if VET_STAGE_CYCLES
    unsigned long long tdbg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
    auto stage = [&](int i) { const unsigned long long now = __builtin_readcyclecounter(); tdbg[i] += now - tlast; tlast = now; };
#else
    auto stage = [](int) {};
#endif
    const float inv_ipf = 1.0f / (float)ipf;
    const int CH = (p.BW + RUNS_THREADS - 1) / RUNS_THREADS;           // chunks of 1024 bitmap words
    const int fields = 32 >> (p.lgF + 1);                               // rows per bitmap word
    const uint32_t fmask = F2 == 32 ? 0xFFFFFFFFu : (1u << F2) - 1u;
    for (long g = blockIdx.x; g < p.G; g += gridDim.x) {
        const long f0 = g * F;
        const int nf = (int)min((long)F, (long)p.T - f0), nitems = nf * ipf;
        // ---- samples -> bit indices (row, mirrored, frame); nearest tile out
        int key[RUNS_SPT], ids[RUNS_SPT];
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            const int i = tid + k * RUNS_THREADS;
            const bool on = i < nitems;
#pragma unroll
            for (int h = 0; h < (PAIRS ? 2 : 1); ++h) {
                int id = -1;
                if (on) id = PAIRS ? grid_dir(h ? a2[k].y : a2[k].x, h ? b2[k].y : b2[k].x, p.W, p.H, bad)
                                   : grid_dir(a1[k], b1[k], p.W, p.H, bad);
                ids[(PAIRS ? 2 : 1) * k + h] = id;
            }
        }
        // the next group's samples travel while this group is sorted (the barriers below wait for LDS traffic only)
        if (g + gridDim.x < p.G) request(g + gridDim.x);
        uint2 rec[RUNS_SPT];
#pragma unroll
        for (int k = 0; k < RUNS_SPT; ++k) rec[k] = p.dirrec[ids[k] < 0 ? 0 : ids[k]];     // unconditional gathers, then selects
#pragma unroll
        for (int k = 0; k < LPT; ++k) {
            const int i = tid + k * RUNS_THREADS;
            const bool on = i < nitems;
            const int fl = (int)(((float)i + 0.5f) * inv_ipf);          // exact: i < 2^13
            int near[2] = {-1, -1};
#pragma unroll
            for (int h = 0; h < (PAIRS ? 2 : 1); ++h) {
                const int q = (PAIRS ? 2 : 1) * k + h;
                int kk = -1;
                if (ids[q] >= 0) {
                    const uint2 r = rec[q];
                    near[h] = (int)(((r.x >> ROW_BITS) & 0xFFFu) | ((r.y >> 28) << 12));
                    kk = (int)(r.x & ROW_MASK) * F2 + (int)(r.x >> 31) * F + fl;
                }
                key[q] = kk;
            }
            if (p.assign && on) {
                if (PAIRS) nt_store((int2*)(p.assign + f0 * (long)p.U) + i, make_int2(near[0], near[1]));
                else __builtin_nontemporal_store(near[0], p.assign + f0 * (long)p.U + i);
            }
            // users present per frame: one LDS atomic per wave where the wave's samples belong to one frame
            const int nv = (near[0] >= 0 ? 1 : 0) + (PAIRS && near[1] >= 0 ? 1 : 0);
            const int fl0 = __builtin_amdgcn_readfirstlane(fl);
            if (__ballot(on && fl != fl0) == 0ull) {
                const int tot = wave_sum(nv);
                if (lane == 0 && tot) atomicAdd(&pres[fl0], (uint32_t)tot);
            } else if (nv) atomicAdd(&pres[fl], (uint32_t)nv);
        }
        stage(0);
#pragma unroll
        for (int k = 0; k < RUNS_SPT; ++k)
            if (key[k] >= 0) atomicOr(&bm[key[k] >> 5], 1u << (key[k] & 31));
        lds_barrier();
        stage(1);
        // ---- scan in word order: word c * 1024 + tid; items | units << 16
        uint32_t mine[RUNS_MAX_CHUNKS], incl[RUNS_MAX_CHUNKS];
#pragma unroll
        for (int c = 0; c < RUNS_MAX_CHUNKS; ++c) {
            mine[c] = incl[c] = 0u;
            if (c < CH) {
                const int w = c * RUNS_THREADS + tid;
                if (w < p.BW) {
                    const uint32_t bits = bm[w];
                    uint32_t units = 0u;
                    for (int f = 0; f < fields; ++f) units += ((uint32_t)__popc((bits >> (f * F2)) & fmask) + (RUN_UNIT_ITEMS - 1)) / RUN_UNIT_ITEMS;
                    mine[c] = (uint32_t)__popc(bits) | (units << 16);
                }
                uint32_t v = mine[c];
#pragma unroll
                for (int o = 1; o < WAVE; o <<= 1) {
                    const uint32_t up = __shfl_up(v, o, WAVE);
                    if (lane >= o) v += up;
                }
                incl[c] = v;
                if (lane == WAVE - 1) wtot[c * NWV + wv] = v;
            }
        }
        lds_barrier();
        if (wv == 0) {                                                  // exclusive scan of the CH * 16 wave totals by one wave
            uint32_t carry = 0u;
            for (int i0 = 0; i0 < CH * NWV; i0 += WAVE) {
                const int i = i0 + lane;
                const uint32_t own = i < CH * NWV ? wtot[i] : 0u;
                uint32_t v = own;
#pragma unroll
                for (int o = 1; o < WAVE; o <<= 1) {
                    const uint32_t up = __shfl_up(v, o, WAVE);
                    if (lane >= o) v += up;
                }
                if (i < CH * NWV) wtot[i] = carry + v - own;
                carry += __shfl(v, WAVE - 1, WAVE);
            }
            if (lane == 0) wtot[RUNS_MAX_CHUNKS * NWV] = carry;         // the group's totals
        }
        lds_barrier();
#pragma unroll
        for (int c = 0; c < RUNS_MAX_CHUNKS; ++c)
            if (c < CH) {
                const int w = c * RUNS_THREADS + tid;
                if (w < p.BW) pre[w] = wtot[c * NWV + wv] + incl[c] - mine[c];
            }
        lds_barrier();
        stage(2);
        // ---- users per item
#pragma unroll
        for (int k = 0; k < RUNS_SPT; ++k)
            if (key[k] >= 0) {
                const int w = key[k] >> 5;
                const uint32_t r = (pre[w] & 0xFFFFu) + (uint32_t)__popc(bm[w] & ((1u << (key[k] & 31)) - 1u));
                atomicAdd(&cnt[r >> 1], 1u << ((r & 1u) << 4));
            }
        lds_barrier();
        stage(3);
        // ---- emit the units of this thread's rows
        uint4* out = p.units + g * p.cap;
#pragma unroll
        for (int c = 0; c < RUNS_MAX_CHUNKS; ++c) {
            if (c >= CH) break;
            const int w = c * RUNS_THREADS + tid;
            if (w >= p.BW || mine[c] == 0u) continue;
            const uint32_t bits = bm[w];
            uint32_t item = pre[w] & 0xFFFFu, unit = pre[w] >> 16;
            bm[w] = 0u;
            for (int f = 0; f < fields; ++f) {
                uint32_t fb = (bits >> (f * F2)) & fmask;
                if (!fb) continue;
                const uint32_t row = (uint32_t)(w * fields + f);
                uint32_t y[3] = {0u, 0u, 0u};
                int slot = 0;
                while (fb) {
                    const int b = __ffs((int)fb) - 1;
                    fb &= fb - 1u;
                    const uint32_t cc = (cnt[item >> 1] >> ((item & 1u) << 4)) & 0xFFFFu;
                    const uint32_t it = cc | ((uint32_t)(b & (F - 1)) << RUN_CNT_BITS) | ((uint32_t)(b >> p.lgF) << 15);
                    y[slot >> 1] |= it << (16 * (slot & 1));
                    ++item;
                    if (++slot == RUN_UNIT_ITEMS) { out[unit++] = make_uint4(row, y[0], y[1], y[2]); y[0] = y[1] = y[2] = 0u; slot = 0; }
                }
                if (slot) out[unit++] = make_uint4(row, y[0], y[1], y[2]);
            }
        }
        const uint32_t totals = wtot[RUNS_MAX_CHUNKS * NWV];
        if (tid == 0) p.n_units[g] = (int32_t)(totals >> 16);
        if (tid < nf) p.present[f0 + tid] = (int32_t)pres[tid];
        stage(4);
        lds_barrier();
        stage(5);
        for (int i = tid; i < (int)(((totals & 0xFFFFu) + 1u) >> 1); i += RUNS_THREADS) cnt[i] = 0u;
        if (tid < RUN_MAX_FRAMES) pres[tid] = 0u;
        lds_barrier();
        stage(6);
    }
#if VET_STAGE_CYCLES
    if (p.dbg && tid == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&p.dbg[i], tdbg[i]);
#endif
    if (p.status) {
        const unsigned long long anybad = __ballot(bad);
        if (anybad && lane == 0) atomicAdd(&p.status[0], (int)__popcll(anybad));
    }
}

// ------------------------------------------------------------------------------------------
// k_spatial_walk — compute_spatial_entropy (entropy_utils.py:147-211), FoV-weighted mode, for one frame group per
// workgroup from k_runs' units.  LDS: hist u64 [F][n] only.
// A group of GS = 2^GSL lanes takes a contiguous share of the group's units, GS units per batch (lane l holds unit l and
// its row's meta word; the walk reads them with ds_bpermute).  Per unit: the row's entries go into one of NB register
// buffers (16-byte weight + 8-byte tile loads, MAXB blocks of 4 * GS entries, unconditionally: a block past the row's
// end, or a unit past the share's end, reads the table's all-zero row), then every item adds entry * (users << row
// shift) into its frame's histogram with ds_add_u64 (mirrored items into tile n-1-t).  The loads of unit j + NB - 1 are
// issued before the adds of unit j: the code is straight-line, so the compiler counts the loads in flight (vmcnt(N))
// and a lane group keeps NB - 1 rows travelling while it adds — the L1's 64 outstanding misses stay occupied with 4-5
// workgroups per CU.  Epilogue: lut_epilogue_int / _fused.
// ------------------------------------------------------------------------------------------
struct WalkParams {
    const uint4* units;
    const int32_t* n_units;
    long cap;
    int F, U, T;
    const uint32_t* tab_w;
    const uint16_t* tab_i;
    const uint32_t* tab_meta;     // [R+1] entries | row shift << 16
    int stride, n;                // n: slots of a frame's histogram (tiles of the lattice, or the fused layout's N)
    uint32_t zero_row;            // index of the table's all-zero row (= rows in use)
    double hmax;
    FusedLayout lay;              // FUSED
    const int32_t* users;         // [T] users present per frame (k_runs)
    double* entropy;
    double* weights;
    int32_t* present;             // caller's output or null
    int32_t* status;
    unsigned long long* dbg;      // development builds: [4] cycles of thread 0 per stage (clear, walk, wait, entropy)
};

template <bool INTERLEAVED, bool FUSED, int MAXB, int GSL>
__global__ __launch_bounds__(1024) void k_spatial_walk(const WalkParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int GS = 1 << GSL, B = 4 * GS;
    static_assert(!INTERLEAVED || GSL == 4, "class-dealt rows are 16-lane rows");
    unsigned long long* hist = (unsigned long long*)smem;                        // [F][n]
    const int tid = threadIdx.x;
    const long g = blockIdx.x, f0 = g * p.F;
    const int nf = (int)min((long)p.F, (long)p.T - f0);
#if VET_STAGE_CYCLES
    unsigned long long tdbg[4] = {0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
    auto stage = [&](int i) { const unsigned long long now = __builtin_readcyclecounter(); tdbg[i] += now - tlast; tlast = now; };
#else
    auto stage = [](int) {};
#endif
    for (int i = tid; i < p.F * p.n; i += blockDim.x) hist[i] = 0ull;
    __syncthreads();
    stage(0);
    const int NG = (int)blockDim.x >> GSL;
    const int grp = tid >> GSL, sl = tid & (GS - 1);
    const long total = p.n_units[g];
    const uint4* units = p.units + g * p.cap;
    // the lane's chunk of block eb holds entries eb + 4*sl + {0..3} (plain block) or eb + sl + GS*{0..3} (class-dealt
    // block): it has work while eb < len - cut (walk_rows)
    const int cut = INTERLEAVED ? min(4 * sl, 3 * GS - 1) : 4 * sl;
    const uint32_t z0 = p.zero_row * (uint32_t)p.stride + (uint32_t)(4 * sl);
    // batches of GS units are dealt round robin to the lane groups (a row's cost depends on how many frames share it)
#ifdef VET_WALK_CONTIG
    const long kbeg = total * grp / NG, kend = total * (grp + 1) / NG, kstep = GS;
#else
    const long kbeg = (long)grp * GS, kend = total, kstep = (long)NG * GS;
#endif
    for (long k0 = kbeg; __any(k0 < kend); k0 += kstep) {
        uint4 u = make_uint4(p.zero_row, 0u, 0u, 0u);                            // past the end: the all-zero row, no item
        if (k0 + sl < kend) u = units[k0 + sl];
        const uint32_t m = p.tab_meta[u.x & ROW_MASK];
        const int nu = (int)min((long)GS, kend - k0);                             // <= 0 once this lane group is done
        for (int j = 0; j < GS; ++j) {
            if (__all(j >= nu)) break;
            const uint32_t row = (uint32_t)__shfl((int)u.x, j, GS) & ROW_MASK, mm = (uint32_t)__shfl((int)m, j, GS);
            uint32_t lo = (uint32_t)__shfl((int)u.y, j, GS), mid = (uint32_t)__shfl((int)u.z, j, GS), hi = (uint32_t)__shfl((int)u.w, j, GS);
            const int len = (int)(mm & 0xFFFFu), lim = len - cut;
            const uint32_t shift = (mm >> 16) & 31u;
            const uint32_t r0 = row * (uint32_t)p.stride + (uint32_t)(4 * sl);
            uint4 w[MAXB];
            ushort4 t[MAXB];
#pragma unroll
            for (int b = 0; b < MAXB; ++b) {
                // a lane without entries in a block reads the table's all-zero row (distinct tiles, weight 0): the adds
                // need no predicate and zero adds do not pile onto one LDS address (walk_rows)
                const uint32_t r = (b * B < lim ? r0 : z0) + (uint32_t)(b * B);
                w[b] = *(const uint4*)(p.tab_w + r);
                t[b] = *(const ushort4*)(p.tab_i + r);
            }
            // the six items as a 96-bit shift register, consumed from the low end (items are packed from slot 0)
#pragma unroll 1
            while (lo & 0xFFFFu) {
                const uint32_t it = lo & 0xFFFFu;
                lo = (lo >> 16) | (mid << 16); mid = (mid >> 16) | (hi << 16); hi >>= 16;
                const uint32_t mult = (it & ((1u << RUN_CNT_BITS) - 1)) << shift;
                const int fr = (int)((it >> RUN_CNT_BITS) & (RUN_MAX_FRAMES - 1));
                const bool flip = (it >> 15) != 0u;
                const int sgn = flip ? -8 : 8;
                char* hb = (char*)(hist + (size_t)fr * p.n) + (flip ? (p.n - 1) * 8 : 0);
#pragma unroll
                for (int b = 0; b < MAXB; ++b) {
                    if (b * B >= len) break;
                    atomicAdd((unsigned long long*)(hb + (int)t[b].x * sgn), (unsigned long long)w[b].x * mult);
                    atomicAdd((unsigned long long*)(hb + (int)t[b].y * sgn), (unsigned long long)w[b].y * mult);
                    atomicAdd((unsigned long long*)(hb + (int)t[b].z * sgn), (unsigned long long)w[b].z * mult);
                    atomicAdd((unsigned long long*)(hb + (int)t[b].w * sgn), (unsigned long long)w[b].w * mult);
                }
            }
        }
    }
    stage(1);
    __syncthreads();
    stage(2);
    if (FUSED) {
        lut_epilogue_fused(hist, nf, f0, p.lay, p.users + f0, p.entropy, p.present, p.weights, p.status);
    } else {
        const int n_of[1] = {p.n};
        const double hmax_of[1] = {p.hmax};
        lut_epilogue_int(hist, (size_t)p.n, nf, f0, 1, n_of, hmax_of, p.users + f0, p.entropy, p.present, p.weights, p.status);
    }
#if VET_STAGE_CYCLES
    stage(3);
    if (p.dbg && tid == 0)
        for (int i = 0; i < 4; ++i) atomicAdd(&p.dbg[i], tdbg[i]);
#endif
}

}  // namespace vet
