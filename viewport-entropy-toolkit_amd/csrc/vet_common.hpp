// vet_common.hpp — wave helpers and the sample -> direction-id quantiser shared by every kernel
// Part of the gfx950 device code of the viewport -> tile -> entropy path (see vet_kernels.hpp for the map).
// Reference citations are relative to /root/reference/src/viewport_entropy_toolkit/.
#pragma once
#include "vet_layout.hpp"
#include <type_traits>

namespace vet {

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
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, WAVE));
    return v;
}
__device__ __forceinline__ int below(unsigned long long m) {      // set bits of m below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
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

}  // namespace vet
