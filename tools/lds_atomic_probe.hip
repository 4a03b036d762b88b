// Microbenchmark: how ds_add_u64 (no return) is serviced on gfx950 — lane grouping and banking.
//   hipcc -O3 --offload-arch=gfx950 -o tools/lds_atomic_probe tools/lds_atomic_probe.hip && tools/lds_atomic_probe
// Every wave issues ITER ds_add_u64 to slot(lane) of its own 8 KiB LDS region (1024 slots of 8 B);
// 1024 threads per workgroup, one workgroup per CU x 2.  Reported: cycles per wave-instruction per CU
// (time * clock / instructions per CU), for address patterns that separate the hypotheses
//   groups of 16 / 32 / 64 lanes   x   16 / 32 bank pairs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

constexpr int ITER = 4096;

__global__ __launch_bounds__(1024) void probe(const int* __restrict__ slot_of_lane, unsigned long long* out, int use32) {
    extern __shared__ unsigned long long lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16 * 1024; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    unsigned long long* mine = lds + wv * 1024;
    const int s = slot_of_lane[lane];
    if (use32) {
        unsigned* m32 = (unsigned*)mine;
#pragma unroll 8
        for (int i = 0; i < ITER; ++i) atomicAdd(&m32[s], 1u);
    } else {
#pragma unroll 8
        for (int i = 0; i < ITER; ++i) atomicAdd(&mine[s], 1ull);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = lds[0] + lds[1024 * 3 + 5];
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    const double clk = prop.clockRate * 1e3;   // Hz
    struct Pat { std::string name; std::vector<int> slot; };
    std::vector<Pat> pats;
    auto add = [&](const char* name, auto f) { Pat p; p.name = name; for (int l = 0; l < 64; ++l) p.slot.push_back(f(l)); pats.push_back(p); };
    add("P0  slot=lane (64 consecutive slots)", [](int l) { return l; });
    add("P1  slot=2*lane", [](int l) { return 2 * l; });
    add("P2  slot=4*lane", [](int l) { return 4 * l; });
    add("P3  slot=8*lane", [](int l) { return 8 * l; });
    add("P4  slot=16*lane (one class mod 16)", [](int l) { return 16 * l; });
    add("P5  slot=32*lane (one class mod 32)", [](int l) { return 32 * (l % 32) + (l / 32); });
    add("P6  16-lane groups: same class mod16 inside a group, groups differ", [](int l) { return (l % 16) * 16 + l / 16; });
    add("P7  16-lane groups: 16 distinct classes inside, all groups the SAME 16 slots+1024k", [](int l) { return (l % 16) + 128 * (l / 16); });
    add("P8  32-lane groups: same class mod32 inside a group", [](int l) { return (l % 32) * 32 + l / 32; });
    add("P9  lanes l and l+16 share a class mod16 (pairs), else distinct", [](int l) { return (l % 16) + 16 * (l / 16) * 1; });
    add("P10 lanes l and l+32 share a class mod32", [](int l) { return (l % 32) + 32 * (l / 32) * 3; });
    add("P11 all lanes one slot (same address)", [](int) { return 7; });
    add("P12 2 lanes per slot (l/2)", [](int l) { return l / 2; });
    add("P13 within 16-group: 2-way mod16 (slot = (l%8)+16*((l%16)/8) + 64*(l/16))", [](int l) { return (l % 8) + 16 * ((l % 16) / 8) + 64 * (l / 16); });
    add("P14 within 16-group: 4-way mod16", [](int l) { return (l % 4) + 16 * ((l % 16) / 4) + 64 * (l / 16); });
    add("P15 random-ish (lane*37 mod 509)", [](int l) { return (l * 37 + 11) % 509; });
    int* d_slot; unsigned long long* d_out;
    hipMalloc(&d_slot, 64 * sizeof(int)); hipMalloc(&d_out, 4096 * sizeof(unsigned long long));
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    const int blocks = n_cu;       // one 1024-thread workgroup (16 waves) per CU
    for (int use32 = 0; use32 < 2; ++use32) {
        printf("---- %s, %d CUs, clock %.0f MHz, %d workgroups of 16 waves\n", use32 ? "ds_add_u32 (slot*2 dwords)" : "ds_add_u64", n_cu, clk / 1e6, blocks);
        for (auto& p : pats) {
            std::vector<int> sl = p.slot;
            if (use32) for (auto& v : sl) v *= 2;          // same byte addresses as the u64 case
            hipMemcpy(d_slot, sl.data(), 64 * sizeof(int), hipMemcpyHostToDevice);
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            probe<<<blocks, 1024, 16 * 1024 * 8>>>(d_slot, d_out, use32);
            if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) { printf("launch failed\n"); return 1; }
            hipEventRecord(a);
            probe<<<blocks, 1024, 16 * 1024 * 8>>>(d_slot, d_out, use32);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double instr_per_cu = 16.0 * ITER;
            printf("%-90s %8.3f ms  %6.1f cycles / wave-instruction\n", p.name.c_str(), ms, ms * 1e-3 * clk / instr_per_cu);
        }
    }
    return 0;
}
