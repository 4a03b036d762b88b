// vet_context.hip — library, context, profiling and device-memory entry points of the C-ABI (include/vet.h), the
// grow-only scratch of a context and the tuning knobs (environment read once, in vet_create).
// There is no CPU compute path here: without a HIP device vet_create fails.
#include "vet_host.hpp"
#include "vet_finalize.hpp"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace vh {

namespace {
thread_local std::string g_err;
const char* const kKernelNames[KID_COUNT] = {"k_grid_dirs", "k_nearest_lut", "k_spatial", "k_transition",
                                             "k_finalize", "k_wtab", "k_weights"};

// a value outside [lo, hi] (or not a number) is ignored
int env_int(const char* name, int lo, int hi, int fallback) {
    const char* e = getenv(name);
    if (!e || !*e) return fallback;
    char* end = nullptr;
    const long v = strtol(e, &end, 10);
    if (end == e || *end != '\0' || v < lo || v > hi) return fallback;
    return (int)v;
}
int env_threads(const char* name, int fallback) {       // workgroup size: whole waves, at most 1024 threads
    const int v = env_int(name, 64, 1024, fallback);
    return v % 64 == 0 ? v : fallback;
}
int env_flag(const char* name) { return getenv(name) ? 1 : 0; }
}  // namespace

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
const char* last_error() { return g_err.c_str(); }

void Tuning::from_environment() {
    gs_log2 = env_int("VET_GS_LOG2", 1, 4, 0);
    tab_interleave = env_int("VET_TAB_INTERLEAVE", 0, 1, 1);
    lut_threads = env_threads("VET_LUT_THREADS", 256);
    if (lut_threads > 256) lut_threads = 256;              // __launch_bounds__(256)
    lut_fpw = env_int("VET_LUT_FPW", 1, 16, 0);
    stride_align = (env_int("VET_STRIDE_ALIGN", 64, 1024, 64) + 63) / 64 * 64;   // whole 64-entry blocks: the walk reads whole blocks
    no_dedup = env_flag("VET_NO_DEDUP");
    dedup_min_users = env_int("VET_DEDUP_MIN_USERS", 1, 1 << 20, 128);
    no_mirror = env_flag("VET_NO_MIRROR");
    u_wgs_per_cu = env_int("VET_U_WGS_PER_CU", 1, 8, 2);
    u_no_lds = env_flag("VET_U_NO_LDS");
    u_fpw = env_int("VET_U_FPW", 1, 64, 0);
    u_waves = env_int("VET_U_WAVES", 1, 16, 4);
    t_threads = env_threads("VET_T_THREADS", 0);
    // powers of two only: the launch logic doubles the workgroup until it covers the users (192 -> 1536 would exceed 1024
    // threads and the 16 wave slots of the cell sums)
    if (t_threads & (t_threads - 1)) t_threads = 0;
    t_wgs_per_cu = env_int("VET_T_WGS_PER_CU", 1, 16, 0);
    t_global = env_flag("VET_T_GLOBAL");
    t_hs_pct = env_int("VET_T_HS_PCT", 100, 400, 200);     // bucket-hash slots per 100 users (100: no gain, 43.3 vs 43.7 us)
    no_fused = env_flag("VET_NO_FUSED");
    fused_single = env_flag("VET_FUSED");
    lut_occ8 = env_int("VET_LUT_OCC8", 0, 1, -1);
    fused_narrow = env_int("VET_FUSED_NARROW", 0, 1, 1);
    narrow_deal = env_int("VET_NARROW_DEAL", 0, 1, 1);
    if (const char* e = getenv("VET_LUT_TIMELINE")) lut_timeline = e;
    no_exact_rows = env_flag("VET_NO_EXACT_ROWS");
}

int collect_profile(vet_ctx* c) {
    for (auto& ep : c->pending) {
        HIP_TRY(hipEventSynchronize(ep.b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ep.a, ep.b));
        c->prof_ms[ep.kid] += ms;
        c->prof_n[ep.kid] += 1;
        c->free_events.push_back(ep.a);
        c->free_events.push_back(ep.b);
    }
    c->pending.clear();
    return VET_OK;
}

int ensure_ws(vet_ctx* c, size_t bytes) {
    if (bytes <= c->ws_bytes) return VET_OK;
    // earlier calls may still use the old workspace on a caller's stream: wait for the device
    HIP_TRY(hipDeviceSynchronize());
    if (c->ws) HIP_TRY(hipFree(c->ws));
    c->ws = nullptr;
    c->ws_bytes = 0;
    size_t want = bytes + bytes / 4 + 4096;
    HIP_TRY(hipMalloc(&c->ws, want));
    c->ws_bytes = want;
    return VET_OK;
}

// slot-indexed device buffer of at least `bytes` bytes (kept by the context between calls, grow-only)
int pooled(vet_ctx* c, int slot, size_t bytes, void** out) {
    if (bytes == 0) bytes = 8;
    if (c->pool_cap[slot] < bytes) {
        if (c->pool[slot]) {
            HIP_TRY(hipDeviceSynchronize());               // an earlier call may still read the old buffer
            HIP_TRY(hipFree(c->pool[slot]));
        }
        c->pool[slot] = nullptr; c->pool_cap[slot] = 0;
        const size_t want = bytes + bytes / 8;
        HIP_TRY(hipMalloc(&c->pool[slot], want));
        c->pool_cap[slot] = want;
    }
    *out = c->pool[slot];
    return VET_OK;
}

int BatchBlob::acquire(vet_ctx* ctx, size_t nbytes) {
    c = ctx; bytes = nbytes ? nbytes : 8;
    BatchStage& t = c->stage[c->stage_next];
    c->stage_next = (c->stage_next + 1) % kBatchStages;
    if (t.pending) {                                       // the batch that used this slot kBatchStages calls ago
        HIP_TRY(hipEventSynchronize(t.done));
        t.pending = false;
    }
    if (!t.done) HIP_TRY(hipEventCreateWithFlags(&t.done, hipEventDisableTiming));
    if (t.cap < bytes) {
        if (t.h) { HIP_TRY(hipHostFree(t.h)); t.h = nullptr; }
        if (t.d) { HIP_TRY(hipFree(t.d)); t.d = nullptr; }
        t.cap = 0;
        const size_t want = bytes + bytes / 8;
        HIP_TRY(hipHostMalloc(&t.h, want, hipHostMallocDefault));
        HIP_TRY(hipMalloc(&t.d, want));
        t.cap = want;
    }
    st = &t;
    return VET_OK;
}

int BatchBlob::upload(hipStream_t stream) {
    s = stream;
    uploaded = true;                                       // from here on the slot is in flight, whatever follows
    HIP_TRY(hipMemcpyAsync(st->d, st->h, bytes, hipMemcpyHostToDevice, s));
    return VET_OK;
}

BatchBlob::~BatchBlob() {
    if (st && uploaded && hipEventRecord(st->done, s) == hipSuccess) st->pending = true;
}

int grid_for(long work, int block, int n_cu) {
    long b = (work + block - 1) / block;
    long cap = (long)n_cu * 8;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

int check_run_args(const vet_plan* pl, int U, int T, const void* out) {
    if (!pl) return fail(VET_ERR_INVALID, "plan is NULL");
    if (U <= 0 || T <= 0) return fail(VET_ERR_INVALID, "n_users and n_frames must be positive (got %d, %d)", U, T);
    if (!out) return fail(VET_ERR_INVALID, "entropy output pointer is NULL");
    return VET_OK;
}

}  // namespace vh

using namespace vh;

extern "C" {

int vet_version(void) { return VET_VERSION; }
const char* vet_last_error(void) { return vh::last_error(); }
const char* vet_kernel_name(int kid) { return (kid >= 0 && kid < KID_COUNT) ? kKernelNames[kid] : ""; }

int vet_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int vet_create(int device_id, vet_ctx** out) {
    if (!out) return fail(VET_ERR_INVALID, "out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(VET_ERR_DEVICE, "no HIP device available (%s); this library has no CPU path",
                    e == hipSuccess ? "0 devices" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n) return fail(VET_ERR_INVALID, "device_id %d out of range [0,%d)", device_id, n);
    HIP_TRY(hipSetDevice(device_id));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    vet_ctx* c = new vet_ctx();
    c->device = device_id;
    c->tune.from_environment();
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // keep two workgroups per CU resident: cap a workgroup at half of the 160 KiB LDS
    c->lds_max = prop.sharedMemPerBlock >= 160 * 1024 ? 80 * 1024 : (size_t)prop.sharedMemPerBlock;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void**)&c->d_log2, 4097 * sizeof(double));
    if (e == hipSuccess) {
        hipLaunchKernelGGL(vet::k_log2_table, dim3(17), dim3(256), 0, c->stream, c->d_log2, 4097);
        e = hipStreamSynchronize(c->stream);
    }
    if (e != hipSuccess) {
        if (c->d_log2) (void)hipFree(c->d_log2);
        if (c->stream) (void)hipStreamDestroy(c->stream);
        delete c;
        return fail(VET_ERR_DEVICE, "context set-up failed: %s", hipGetErrorString(e));
    }
    *out = c;
    return VET_OK;
}

int vet_device_pci_bus_id(vet_ctx* c, char* buf, int len) {
    if (!c || !buf || len < 16) return fail(VET_ERR_INVALID, "ctx or buf is NULL, or len < 16");
    HIP_TRY(hipDeviceGetPCIBusId(buf, len, c->device));
    return VET_OK;
}

int vet_destroy(vet_ctx* c) {
    if (!c) return VET_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    for (auto& ep : c->pending) { (void)hipEventDestroy(ep.a); (void)hipEventDestroy(ep.b); }
    for (auto e : c->free_events) (void)hipEventDestroy(e);
    if (c->ws) (void)hipFree(c->ws);
    if (c->d_log2) (void)hipFree(c->d_log2);
    for (void* q : c->pool) if (q) (void)hipFree(q);
    for (auto& t : c->stage) {
        if (t.h) (void)hipHostFree(t.h);
        if (t.d) (void)hipFree(t.d);
        if (t.done) (void)hipEventDestroy(t.done);
    }
    (void)hipStreamDestroy(c->stream);
    delete c;
    return VET_OK;
}

int vet_synchronize(vet_ctx* c) {
    if (!c) return fail(VET_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VET_OK;
}

int vet_profile_enable(vet_ctx* c, int on) {
    if (!c) return fail(VET_ERR_INVALID, "ctx is NULL");
    c->profiling = on != 0;
    return VET_OK;
}
int vet_profile_reset(vet_ctx* c) {
    if (!c) return fail(VET_ERR_INVALID, "ctx is NULL");
    int rc = collect_profile(c);
    if (rc) return rc;
    for (int i = 0; i < KID_COUNT; ++i) { c->prof_ms[i] = 0; c->prof_n[i] = 0; }
    return VET_OK;
}
int vet_profile_get(vet_ctx* c, int kid, double* total_ms, int64_t* launches) {
    if (!c || kid < 0 || kid >= KID_COUNT) return fail(VET_ERR_INVALID, "bad ctx or kernel id");
    int rc = collect_profile(c);
    if (rc) return rc;
    if (total_ms) *total_ms = c->prof_ms[kid];
    if (launches) *launches = c->prof_n[kid];
    return VET_OK;
}

int vet_malloc(vet_ctx* c, size_t bytes, void** d_ptr) {
    if (!c || !d_ptr) return fail(VET_ERR_INVALID, "ctx or d_ptr is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 8));
    return VET_OK;
}
int vet_free(vet_ctx* c, void* d_ptr) {
    if (!c) return fail(VET_ERR_INVALID, "ctx is NULL");
    if (d_ptr) HIP_TRY(hipFree(d_ptr));
    return VET_OK;
}
int vet_memcpy_h2d(vet_ctx* c, void* d, const void* h, size_t bytes) {
    if (!c) return fail(VET_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VET_OK;
}
int vet_memcpy_d2h(vet_ctx* c, void* h, const void* d, size_t bytes) {
    if (!c) return fail(VET_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return VET_OK;
}

}  // extern "C"
