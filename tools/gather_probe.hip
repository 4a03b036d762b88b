// Microbenchmark: what shapes the row-gather rate on MI355X?  30.72 M row visits of a 20301-row
// table (768-byte rows), random rows, no LDS work.  V1: the product's shape (16 lanes x 16 B of
// weights + 16 lanes x 8 B of tiles per row chunk, 4 rows per wave instruction, 2 chunks per row);
// V2: one whole row per wave instruction (48 lanes x 16 B contiguous); V3: two rows per instruction
// (24 lanes x 32 B each as 2 x 16 B).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void v1(const unsigned* __restrict__ tw, const unsigned short* __restrict__ ti, const int* __restrict__ ids,
                   long n, unsigned long long* out) {
    const int lane = threadIdx.x & 63, sub = lane >> 4, sl = lane & 15;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
    unsigned long long acc = 0;
    for (long j = wave * 8; j < n; j += nw * 8) {
        long r0 = (long)ids[j + sub] * 192, r1 = (long)ids[j + 4 + sub] * 192;
        for (int e = 4 * sl; e < 128; e += 64) {
            uint4 a = *(const uint4*)(tw + r0 + e); ushort4 b = *(const ushort4*)(ti + r0 + e);
            uint4 c = *(const uint4*)(tw + r1 + e); ushort4 d = *(const ushort4*)(ti + r1 + e);
            acc += a.x + a.y + a.z + a.w + b.x + b.w + c.x + c.y + c.z + c.w + d.x + d.w;
        }
    }
    if (acc == 0x1234567) out[0] = acc;
}
// rows stored as one 768-byte record: 128 weights (512 B) + 128 tiles (256 B)
__global__ void v2(const uint4* __restrict__ rec, const int* __restrict__ ids, long n, unsigned long long* out) {
    const int lane = threadIdx.x & 63;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
    unsigned long long acc = 0;
    for (long j = wave * 4; j < n; j += nw * 4) {
        uint4 a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) if (lane < 48) a[k] = rec[(long)ids[j + k] * 48 + lane];
#pragma unroll
        for (int k = 0; k < 4; ++k) if (lane < 48) acc += a[k].x + a[k].y + a[k].z + a[k].w;
    }
    if (acc == 0x1234567) out[0] = acc;
}
// two rows per instruction: lanes 0..23 row A, 24..47 row B, each lane 32 B (2 x uint4)
__global__ void v3(const uint4* __restrict__ rec, const int* __restrict__ ids, long n, unsigned long long* out) {
    const int lane = threadIdx.x & 63, half = lane / 24, hl = lane % 24;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((long)gridDim.x * blockDim.x) >> 6;
    unsigned long long acc = 0;
    for (long j = wave * 4; j < n; j += nw * 4) {
        uint4 a[2][2];
#pragma unroll
        for (int k = 0; k < 2; ++k) if (lane < 48) {
            const uint4* r = rec + (long)ids[j + 2 * k + half] * 48 + 2 * hl;
            a[k][0] = r[0]; a[k][1] = r[1];
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) if (lane < 48) acc += a[k][0].x + a[k][0].w + a[k][1].y + a[k][1].z;
    }
    if (acc == 0x1234567) out[0] = acc;
}

int main() {
    const long N = 30720000; const int D = 20301;
    std::vector<int> ids(N + 64);
    unsigned s = 12345;
    for (long i = 0; i < N + 64; ++i) { s = s * 1664525u + 1013904223u; ids[i] = (s >> 8) % D; }
    int* d_ids; unsigned* tw; unsigned short* ti; uint4* rec; unsigned long long* out;
    CK(hipMalloc(&d_ids, (N + 64) * 4)); CK(hipMalloc(&tw, (size_t)D * 192 * 4)); CK(hipMalloc(&ti, (size_t)D * 192 * 2));
    CK(hipMalloc(&rec, (size_t)D * 768)); CK(hipMalloc(&out, 8));
    CK(hipMemcpy(d_ids, ids.data(), (N + 64) * 4, hipMemcpyHostToDevice));
    CK(hipMemset(tw, 1, (size_t)D * 192 * 4)); CK(hipMemset(ti, 1, (size_t)D * 192 * 2)); CK(hipMemset(rec, 1, (size_t)D * 768));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, auto launch) {
        for (int i = 0; i < 2; ++i) launch();
        CK(hipEventRecord(e0)); for (int i = 0; i < 5; ++i) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
        printf("%-40s %.3f ms  %.1f GB/s per CU (768 B/row)\n", name, ms, N * 768.0 / ms / 1e6 / 256);
    };
    for (int blocks : {2048, 4096, 8192}) {
        printf("blocks %d x 256 threads\n", blocks);
        time("V1 product shape (4 rows/instr, 2 arrays)", [&] { hipLaunchKernelGGL(v1, dim3(blocks), dim3(256), 0, 0, tw, ti, d_ids, N, out); });
        time("V2 one 768-B row per instruction", [&] { hipLaunchKernelGGL(v2, dim3(blocks), dim3(256), 0, 0, rec, d_ids, N, out); });
        time("V3 two rows per instruction (2x16 B/lane)", [&] { hipLaunchKernelGGL(v3, dim3(blocks), dim3(256), 0, 0, rec, d_ids, N, out); });
    }
    return 0;
}
