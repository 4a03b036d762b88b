// Microbenchmark: what does the unweighted stream (16 B in / 4 B out per sample) reach on MI355X
// as pieces are added?  A: pure stream  B: + global LUT gather  C: + LDS atomics  D: + frame barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE, int PPT>
__global__ void probe(const double2* __restrict__ mu, const double2* __restrict__ mv, const uint16_t* __restrict__ lut,
                      int2* __restrict__ out, long pairs, int n, int pairs_per_frame) {
    extern __shared__ unsigned cnt[];
    if (MODE >= 2) { for (int i = threadIdx.x; i < n; i += blockDim.x) cnt[i] = 0; __syncthreads(); }
    const long base = (long)blockIdx.x * blockDim.x * PPT;
    double2 a[PPT], b[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) { long i = base + threadIdx.x + (long)k * blockDim.x; if (i < pairs) { a[k] = mu[i]; b[k] = mv[i]; } }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        long i = base + threadIdx.x + (long)k * blockDim.x;
        if (i < pairs) {
            int id0 = (int)(b[k].x * 200.0) * 101 + (int)(a[k].x * 100.0);
            int id1 = (int)(b[k].y * 200.0) * 101 + (int)(a[k].y * 100.0);
            int n0 = id0, n1 = id1;
            if (MODE >= 1) { n0 = lut[id0]; n1 = lut[id1]; }
            if (MODE >= 2) { atomicAdd(&cnt[n0 % n], 1u); atomicAdd(&cnt[n1 % n], 1u); }
            out[i] = make_int2(n0, n1);
        }
    }
    if (MODE >= 3) {
        __syncthreads();
        if (threadIdx.x < 64) {
            double h = 0;
            for (int t = threadIdx.x; t < n; t += 64) { unsigned v = cnt[t]; if (v) { double q = v / 1024.0; h -= q * log2(q); } }
            for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o, 64);
            if (threadIdx.x == 0) ((double*)out)[(long)blockIdx.x] = h;   // garbage location is fine for a probe
        }
    }
}

template <int MODE, int PPT>
float run(const double2* mu, const double2* mv, const uint16_t* lut, int2* out, long pairs, int threads) {
    long per_block = (long)threads * PPT;
    int blocks = (int)((pairs + per_block - 1) / per_block);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe<MODE, PPT>), dim3(blocks), dim3(threads), 501 * 4, 0, mu, mv, lut, out, pairs, 501, 512);
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((probe<MODE, PPT>), dim3(blocks), dim3(threads), 501 * 4, 0, mu, mv, lut, out, pairs, 501, 512);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}


// persistent variants: LUT in LDS.  MODE 4: stream + LDS lookup; 5: + LDS hist atomics; 6: + per-block zero/barrier/entropy
template <int MODE, int PPT>
__global__ void probe_p(const double2* __restrict__ mu, const double2* __restrict__ mv, const uint16_t* __restrict__ lut,
                        int2* __restrict__ out, long pairs, int n, int frames_per_blk, double* ent) {
    extern __shared__ unsigned smem_u[];
    uint16_t* l = (uint16_t*)smem_u;                    // 20301 u16 -> 40608 B
    unsigned* cnt = smem_u + 10160;                      // [frames_per_blk][n]
    for (int i = threadIdx.x; i < 10151; i += blockDim.x) smem_u[i] = ((const unsigned*)lut)[i];
    for (int i = threadIdx.x; i < frames_per_blk * n; i += blockDim.x) cnt[i] = 0;
    __syncthreads();
    const long per_blk = (long)blockDim.x * PPT;
    const long nblk = (pairs + per_blk - 1) / per_blk;
    const int ppf = per_blk / frames_per_blk;
    for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const long base = blk * per_blk;
        double2 a[PPT], b[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) { long i = base + threadIdx.x + (long)k * blockDim.x; if (i < pairs) { a[k] = mu[i]; b[k] = mv[i]; } }
        if (MODE >= 6) { for (int i = threadIdx.x; i < frames_per_blk * n; i += blockDim.x) cnt[i] = 0; __syncthreads(); }
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            long i = base + threadIdx.x + (long)k * blockDim.x;
            if (i < pairs) {
                int id0 = (int)(b[k].x * 200.0) * 101 + (int)(a[k].x * 100.0);
                int id1 = (int)(b[k].y * 200.0) * 101 + (int)(a[k].y * 100.0);
                int n0 = l[id0], n1 = l[id1];
                if (MODE >= 5) { unsigned* row = cnt + ((threadIdx.x + k * blockDim.x) / ppf) * n; atomicAdd(&row[n0], 1u); atomicAdd(&row[n1], 1u); }
                out[i] = make_int2(n0, n1);
            }
        }
        if (MODE >= 6) {
            __syncthreads();
            const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
            if (wv < frames_per_blk) {
                double h = 0;
                for (int t = lane; t < n; t += 64) { unsigned v = cnt[wv * n + t]; if (v) { double q = v / 1024.0; h -= q * log2(q); } }
                for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o, 64);
                if (lane == 0) ent[blk * frames_per_blk + wv] = h;
            }
            __syncthreads();
        }
    }
}

template <int MODE, int PPT>
float run_p(const double2* mu, const double2* mv, const uint16_t* lut, int2* out, long pairs, int threads, int wgs_per_cu, double* ent) {
    int blocks = 256 * wgs_per_cu;
    int fpb = threads * PPT / 512;                     // frames per block (U = 1024 -> 512 pairs)
    size_t lds = 40640 + (size_t)fpb * 501 * 4;
    hipFuncSetAttribute((const void*)probe_p<MODE, PPT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((probe_p<MODE, PPT>), dim3(blocks), dim3(threads), lds, 0, mu, mv, lut, out, pairs, 501, fpb, ent);
    CK(hipEventRecord(e0));
    const int reps = 20;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((probe_p<MODE, PPT>), dim3(blocks), dim3(threads), lds, 0, mu, mv, lut, out, pairs, 501, fpb, ent);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main() {
    const long S = 1024L * 30000, pairs = S / 2;
    std::vector<double> h(S);
    for (long i = 0; i < S; ++i) h[i] = (double)((i * 2654435761u) % 1000000) / 1000001.0;
    double *mu, *mv; uint16_t* lut; int2* out;
    CK(hipMalloc(&mu, S * 8)); CK(hipMalloc(&mv, S * 8)); CK(hipMalloc(&lut, 20301 * 2)); CK(hipMalloc(&out, S * 4 + 1024 * 1024));
    CK(hipMemcpy(mu, h.data(), S * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(mv, h.data(), S * 8, hipMemcpyHostToDevice));
    std::vector<uint16_t> l(20301); for (int i = 0; i < 20301; ++i) l[i] = (i * 7) % 501;
    CK(hipMemcpy(lut, l.data(), 20301 * 2, hipMemcpyHostToDevice));
    const double bytes = S * 20.0;
#define R(M, P, T) { float ms = run<M, P>((double2*)mu, (double2*)mv, lut, out, pairs, T); printf("mode %d ppt %d threads %4d : %.4f ms  %.0f GB/s\n", M, P, T, ms, bytes / ms / 1e6); }
    R(0, 2, 256) R(0, 4, 256) R(0, 8, 256) R(0, 4, 512) R(0, 2, 1024)
    R(1, 2, 256) R(1, 4, 256) R(1, 8, 256)
    R(2, 2, 256) R(2, 4, 256) R(2, 8, 256)
    R(3, 2, 256) R(3, 4, 256) R(3, 8, 256)
    double* ent; CK(hipMalloc(&ent, 8 * 40000));
#define RP(M, P, T, W) { float ms = run_p<M, P>((double2*)mu, (double2*)mv, lut, out, pairs, T, W, ent); printf("persist mode %d ppt %d threads %4d wgs/cu %d : %.4f ms  %.0f GB/s\n", M, P, T, W, ms, bytes / ms / 1e6); }
    RP(4, 2, 512, 2) RP(4, 4, 512, 2) RP(4, 4, 512, 3) RP(4, 2, 1024, 2) RP(4, 4, 1024, 2) RP(4, 4, 256, 3) RP(4, 8, 256, 3)
    RP(5, 4, 512, 2) RP(5, 4, 512, 3) RP(5, 2, 1024, 2) RP(5, 4, 256, 3)
    RP(6, 4, 512, 2) RP(6, 4, 512, 3) RP(6, 2, 1024, 2) RP(6, 4, 1024, 2) RP(6, 8, 256, 3) RP(6, 4, 256, 3) RP(6, 8, 512, 2)
    return 0;
}
