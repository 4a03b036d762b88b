// vet_transition.hpp — k_transition_run / k_transition_any: transition entropy of consecutive frame pairs
// Part of the gfx950 device code of the viewport -> tile -> entropy path (see vet_kernels.hpp for the map).
// Reference citations are relative to /root/reference/src/viewport_entropy_toolkit/.
#pragma once
#include "vet_common.hpp"

namespace vet {

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
    const struct TransVideo* videos;   // k_transition_run: a batch of videos in one launch (null: the single video above)
    int n_videos;
};

// One video of a batched transition launch (vet_transition_entropy_batch): workgroups [wg0, wg0 + n_wgs) walk its
// rows in contiguous runs of run_q (+1 for the first run_r of them) rows; a run never crosses into another video.
struct TransVideo {
    const double* mu;
    const double* mv;
    int U, T;
    double* ent;                  // [T-1]
    int32_t* pairs;               // [(T-1)*U*2] or null
    int32_t* common;              // [T-1] or null
    int wg0, n_wgs, run_q, run_r;
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
// k_transition_big — compute_transition_entropy (entropy_utils.py:213-332) for more users than k_transition_run holds
// in registers (U > 4096), with the bucket hash still in LDS.  What has to fit is not the users but the BUCKETS —
// the distinct (source tile, destination tile) pairs of a row — and a source tile of m users has at most min(m, n) of
// them.  After step (1) (first user and user count per source tile, all users) the source tiles are cut into ranges
// whose bucket bound sum min(m, n) stays below the hash's capacity, and steps (2)-(5) run once per range over an
// 8192-slot LDS hash: one pass for a real audience (viewers move little between frames: a few destinations per source
// tile), two at 9 000 uniformly scattered users, ceil(sum min(m, n) / 4 900) in general.  Per user and pass: one 4-byte
// read of its packed pair from a per-workgroup global array (written in step (1), L2 resident) and one LDS read of its
// source tile's pass.  One persistent 1024-thread workgroup per CU.  The global-scratch hash of k_transition_any
// (dependent global atomics: 0.7 ms for 512 rows of 9 000 users) stays as the fallback for lattices of more than
// TRANS_BIG_MAX_TILES tiles.
// LDS: acc f64 [2][20] | first_u, m_cnt, k_cnt, last_fu u32 [n4] | pass u16 [n4] | hkey, hfu, hcnt u32 [8192]
// ------------------------------------------------------------------------------------------
constexpr int TRANS_BIG_HS = 8192, TRANS_BIG_MAX_TILES = 2800, TRANS_BIG_THREADS = 1024;
__host__ __device__ __forceinline__ size_t trans_big_lds_bytes(int n) {
    const size_t n4 = ((size_t)n + 3) & ~(size_t)3;
    return 2 * TRANS_ACC * 8 + 4 * n4 * 4 + ((n4 * 2 + 15) & ~(size_t)15) + (size_t)3 * TRANS_BIG_HS * 4 + 64;
}

template <bool FROM_IDS>
__global__ __launch_bounds__(TRANS_BIG_THREADS) void k_transition_big(const TransParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int HS = TRANS_BIG_HS, BD = TRANS_BIG_THREADS, NW = BD / WAVE;
    double* acc2 = (double*)smem;                              // [2][TRANS_ACC]
    unsigned* first_u = (unsigned*)(acc2 + 2 * TRANS_ACC);     // [n4]
    const int n4 = (p.n + 3) & ~3;
    unsigned* m_cnt = first_u + n4;
    unsigned* k_cnt = m_cnt + n4;
    unsigned* last_fu = k_cnt + n4;
    unsigned short* pass_of = (unsigned short*)(last_fu + n4); // [n4] range (pass) of every source tile
    unsigned* hkey = (unsigned*)((unsigned char*)pass_of + ((n4 * 2 + 15) & ~15));
    unsigned* hfu = hkey + HS;
    unsigned* hcnt = hfu + HS;
    int* n_pass = (int*)(hcnt + HS);
    const int tid = threadIdx.x, lane = lane_id(), wv = wave_id();
    const size_t U4 = ((size_t)p.U + 3) & ~(size_t)3;
    unsigned* pc = p.scratch + (size_t)blockIdx.x * U4;       // [U4] the row's packed pairs
    const long R = (long)p.T - 1;
    const int cap = HS * 6 / 10 - p.n;                         // bucket bound of a range (one more tile may join: + n at most)
    const unsigned hs_shift = 32 - 13;
    bool bad = false;
    int parity = 0;
    for (long r = blockIdx.x; r < R; r += gridDim.x, parity ^= 1) {
        double* acc = acc2 + TRANS_ACC * parity;
        {   // per-tile words (the barrier at the end of the previous row precedes)
            const uint4 ones = make_uint4(~0u, ~0u, ~0u, ~0u), zeros = make_uint4(0u, 0u, 0u, 0u);
            for (int i = tid; i < n4 / 4; i += BD) {
                ((uint4*)first_u)[i] = ones; ((uint4*)m_cnt)[i] = zeros; ((uint4*)k_cnt)[i] = zeros; ((uint4*)last_fu)[i] = zeros;
            }
            if (tid == 0) ((unsigned long long*)acc)[16] = 0ull;
        }
        __syncthreads();
        // ---- (1) every user: tiles of both frames, first user and user count per source tile
        for (int u = tid; u < p.U; u += BD) {
            const int ia = sample_dir<FROM_IDS, false>(p.src, r * (long)p.U + u, bad);
            const int ib = sample_dir<FROM_IDS, false>(p.src, (r + 1) * (long)p.U + u, bad);
            unsigned packed = EMPTY_KEY;
            int pa = -1, cb = -1;
            if (ia >= 0 && ib >= 0) {           // user present in both frames (entropy_utils.py:259-261)
                pa = p.nearest[ia]; cb = p.nearest[ib];
                packed = ((unsigned)pa << 16) | (unsigned)cb;
                if (first_u[pa] > (unsigned)u) atomicMin(&first_u[pa], (unsigned)u);
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
        // user count per source tile (m_cnt is final here).  Written for EVERY tile in a loop of its own: the passes below
        // only visit tiles with pass_of[t] < n_pass, and empty tiles behind the last populated one get pass_of = n_pass
        // when the row's bucket bound is an exact multiple of cap.
        if (p.srccount)
            for (int t = tid; t < p.n; t += BD) p.srccount[r * (long)p.n + t] = (int)m_cnt[t];
        // ---- ranges of source tiles: tile t goes to pass floor(bound of the tiles before it / cap), bound = min(m, n)
        if (wv == 0) {
            unsigned carry = 0u;
            for (int t0 = 0; t0 < p.n; t0 += WAVE) {
                const int t = t0 + lane;
                const unsigned b = t < p.n ? min(m_cnt[t], (unsigned)p.n) : 0u;
                unsigned v = b;
#pragma unroll
                for (int o = 1; o < WAVE; o <<= 1) {
                    const unsigned up = __shfl_up(v, o, WAVE);
                    if (lane >= o) v += up;
                }
                if (t < p.n) pass_of[t] = (unsigned short)((carry + v - b) / (unsigned)cap);
                carry += __shfl(v, WAVE - 1, WAVE);
            }
            if (lane == 0) *n_pass = carry ? (int)((carry - 1u) / (unsigned)cap) + 1 : 1;
        }
        __syncthreads();
        const int passes = *n_pass;
        const int N = (int)((const unsigned long long*)acc)[16];
        const double inv_n = 1.0 / (double)N;
        const bool tab = p.U <= 4096;
        double h = 0.0;
        // Several passes (an audience spread over many source tiles: the late rows of a long video): every pass scans all
        // users, so the per-user test must be cheap.  The packed pairs are rewritten once as pass << 24 | source << 12 |
        // destination (tiles < 4096: TRANS_BIG_MAX_TILES; 0xFF = not in any pass: absent, or the first user of its source
        // tile) and the scans of steps (2) and (3) read one word per user and touch the LDS for the users of the pass only
        // (before: a global word, the tile's pass and its first user, per user and scan).  Each thread rewrites and later
        // reads its own users only: no barrier.
        const bool multi = passes > 1 && passes < 255;
        if (multi)
            for (int u = tid; u < p.U; u += BD) {
                const unsigned key = pc[u];
                unsigned v = EMPTY_KEY;
                if (key != EMPTY_KEY) {
                    const unsigned src = key >> 16;
                    if (first_u[src] != (unsigned)u) v = ((unsigned)pass_of[src] << 24) | (src << 12) | (key & 0xFFFu);
                }
                pc[u] = v;
            }
        for (int q = 0; q < passes; ++q) {
            for (int i = tid; i < HS / 4; i += BD) {
                ((uint4*)hkey)[i] = make_uint4(~0u, ~0u, ~0u, ~0u); ((uint4*)hfu)[i] = make_uint4(~0u, ~0u, ~0u, ~0u);
                ((uint4*)hcnt)[i] = make_uint4(0u, 0u, 0u, 0u);
            }
            __syncthreads();
            // ---- (2) non-first users of the range: bucket insert; the creator counts the bucket into K
            for (int u = tid; u < p.U; u += BD) {
                unsigned key = pc[u], src;
                if (multi) {
                    if ((key >> 24) != (unsigned)q) continue;
                    key &= 0xFFFFFFu; src = key >> 12;
                } else {
                    if (key == EMPTY_KEY) continue;
                    src = key >> 16;
                    if (first_u[src] == (unsigned)u) continue;
                    // >= 255 passes (unreachable under the host's limits U < 2^19, n <= 2800: at most 248; kept so that the
                    // kernel is right by itself): the packed form has no room for the pass, test the tile's pass here
                    if (passes > 1 && (int)pass_of[src] != q) continue;
                }
                unsigned slot = (key * 2654435761u) >> hs_shift;
                for (;;) {
                    const unsigned was = atomicCAS(&hkey[slot], EMPTY_KEY, key);
                    if (was == EMPTY_KEY) { atomicAdd(&k_cnt[src], 1u); break; }      // a new destination of this source tile
                    if (was == key) break;
                    slot = (slot + 1) & (unsigned)(HS - 1);
                }
                if (hfu[slot] > (unsigned)u) atomicMin(&hfu[slot], (unsigned)u);
                atomicAdd(&hcnt[slot], 1u);
            }
            __syncthreads();
            // ---- (3) the first user of every bucket: latest first appearance per source tile, user << 13 | slot
            for (int u = tid; u < p.U; u += BD) {
                unsigned key = pc[u], src;
                if (multi) {
                    if ((key >> 24) != (unsigned)q) continue;
                    key &= 0xFFFFFFu; src = key >> 12;
                } else {
                    if (key == EMPTY_KEY) continue;
                    src = key >> 16;
                    if (first_u[src] == (unsigned)u) continue;
                    // >= 255 passes (unreachable under the host's limits U < 2^19, n <= 2800: at most 248; kept so that the
                    // kernel is right by itself): the packed form has no room for the pass, test the tile's pass here
                    if (passes > 1 && (int)pass_of[src] != q) continue;
                }
                unsigned slot = (key * 2654435761u) >> hs_shift;
                while (hkey[slot] != key) slot = (slot + 1) & (unsigned)(HS - 1);
                if (hfu[slot] == (unsigned)u) atomicMax(&last_fu[src], ((unsigned)u << 13) | slot);
            }
            __syncthreads();
            // ---- (5) cells of the range's source tiles: -(m/N) K (w/m) log2(w/m), w = the latest bucket's count
            for (int t = tid; t < p.n; t += BD) {
                if ((int)pass_of[t] != q) continue;
                const unsigned m = m_cnt[t];
                if (!m) continue;
                const unsigned K = 1u + k_cnt[t];
                const unsigned w = m <= 1u ? 1u : hcnt[last_fu[t] & 0x1FFFu];
                const double lq = tab ? p.log2_tab[w] - p.log2_tab[m] : log2((double)w / (double)m);
                h -= ((double)((unsigned long long)K * w) * inv_n) * lq;
            }
            __syncthreads();
        }
        h = wave_sum(h);
        if (lane == 0) acc[wv] = h;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
            for (int i = 0; i < NW; ++i) tot += acc[i];
            double hmax = p.hmax;
            if (!(N > p.n)) {
                const double tp = 1.0 / (double)N;          // entropy_utils.py:322-327
                hmax = (double)N * -tp * (tab ? -p.log2_tab[N] : log2(tp));
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
template <bool FROM_IDS, int UPT, bool EXACT, int THREADS, bool BATCH = false>
__global__ void k_transition_run(const TransParams launch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    TransParams p = launch;
    long b = blockIdx.x;
    if (BATCH) {                                   // this workgroup's video of the batch
        int lo = 0, hi = launch.n_videos - 1;      // last video with wg0 <= blockIdx.x
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (launch.videos[mid].wg0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
        }
        const TransVideo& v = launch.videos[lo];
        if ((int)blockIdx.x >= v.wg0 + v.n_wgs) return;
        p.src.mu = v.mu; p.src.mv = v.mv; p.U = v.U; p.T = v.T;
        p.ent_k = v.ent; p.pairs = v.pairs; p.common = v.common; p.srccount = nullptr;
        p.run_q = v.run_q; p.run_r = v.run_r;
        b -= v.wg0;
    }
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
    // start-up: the run's first two frames are requested back to back (one HBM round trip instead of two before the
    // first row: 43.6 -> 43.1 us at config 5); the first frame's samples wait in registers that are dead once the loop starts
    double fa[UPT], fb[UPT];
    int fi[UPT];
    request(true);
#pragma unroll
    for (int k = 0; k < UPT; ++k) { fa[k] = sa[k]; fb[k] = sb[k]; fi[k] = si[k]; }
    request(true);
    {
        double ka[UPT], kb[UPT];
        int ki[UPT];
#pragma unroll
        for (int k = 0; k < UPT; ++k) { ka[k] = sa[k]; kb[k] = sb[k]; ki[k] = si[k]; sa[k] = fa[k]; sb[k] = fb[k]; si[k] = fi[k]; }
        tiles_of(prev);
#pragma unroll
        for (int k = 0; k < UPT; ++k) { sa[k] = ka[k]; sb[k] = kb[k]; si[k] = ki[k]; }
    }
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

}  // namespace vet
