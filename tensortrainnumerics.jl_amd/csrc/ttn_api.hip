// ttn_api.hip — the C ABI of include/ttn.h on top of the HIP kernels (gfx950).
#include "../../include/ttn.h"
#include "ttn_common.h"
#include "ttn_stream_kernels.h"
#include "ttn_dense_kernels.h"
#include "ttn_dot_kernels.h"
#include "ttn_ortho_kernels.h"
#include "ttn_ortho512.h"
#include "ttn_ortho_ramp.h"
#include "ttn_hsvd_kernels.h"
#include "ttn_als_kernels.h"
#include "ttn_als_grid.h"
#include "ttn_eig_kernels.h"
#include "ttn_tdvp_kernels.h"
#include "ttn_densefact_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <mutex>
#include <set>
#include <string>
#include <vector>

// the 512-thread build of k_compress and its building blocks (ttn_wg512.hip: two workgroups per CU)
extern "C" {
int ttn_wg512_init(void);
size_t ttn_wg512_compress_args_bytes(void);
int ttn_wg512_launch_compress(const void* args, size_t nbytes, int grid, hipStream_t stream);
int ttn_wg512_selftest_eig(const double* G, double* Vst, int n, int r, int nev, double* sig, double* Xout, long long* clk, hipStream_t stream);
int ttn_wg512_bench_gemm(int m, int n, int k, double* A, double* B, double* C, int ta, int tb, int reps, long long* cycles, int grid, hipStream_t stream);
int ttn_wg512_selftest_gemm(int m, int n, int k, double* A, double* B, double* C, double alpha, double beta, int ta, int tb, hipStream_t stream);
}

// ------------------------------------------------------------------------------------------------
// global state
// ------------------------------------------------------------------------------------------------
namespace {
int* g_next_train = nullptr;  // train counter of the persistent k_compress grid
int* g_pending_status = nullptr;   // failure codes of handles that were FREED before anybody queried them (ttn_status_all)
std::set<struct ttn_tt_s*> g_live;  // every live ttn_tt handle (ttn_status_all walks them)
std::recursive_mutex g_mu;
bool g_init = false;
int g_device = -1;
hipStream_t g_stream = nullptr;
hipEvent_t g_ev0 = nullptr, g_ev1 = nullptr;
std::string g_err = "";
void* g_scratch = nullptr;
size_t g_scratch_bytes = 0;
double* g_dout = nullptr;     // per-train double outputs (dot)
std::vector<hipEvent_t> g_slots;   // ttn_event_record slots
int g_prof_batch = 0;
long long* g_prof = nullptr;       // TTN_PROF=1 diagnostic phase counters of the last compress launch
int g_dout_cap = 0;

int fail(int code, const char* what) {
    g_err = what;
    return code;
}
int hipfail(hipError_t e, const char* where) {
    g_err = std::string(where) + ": " + hipGetErrorString(e);
    return (int)e > 0 ? (int)e : 999;
}
#define HIPCHK(call)                                  \
    do {                                              \
        hipError_t e_ = (call);                       \
        if (e_ != hipSuccess) return hipfail(e_, #call); \
    } while (0)
#define NEED_INIT() \
    if (!g_init) return fail(TTN_ERR_NOT_INIT, "ttn_init has not been called")

int ensure_scratch(size_t bytes) {
    if (bytes <= g_scratch_bytes) return TTN_OK;
    if (g_scratch) {
        HIPCHK(hipStreamSynchronize(g_stream));
        HIPCHK(hipFree(g_scratch));
        g_scratch = nullptr;
        g_scratch_bytes = 0;
    }
    HIPCHK(hipMalloc(&g_scratch, bytes));
    g_scratch_bytes = bytes;
    return TTN_OK;
}
bool g_have_launch_ms = false;     // g_ev0 / g_ev1 bracket the last ttn_dot / ttn_orthogonalize kernel (ttn_last_launch_ms)
int ensure_prof(int batch) {        // TTN_PROF=1: 200 counters per train, zeroed
    static int prof_cap = 0;
    if (prof_cap < batch) { if (g_prof) hipFree(g_prof); g_prof = nullptr; HIPCHK(hipMalloc((void**)&g_prof, sizeof(long long) * 200 * batch)); prof_cap = batch; }
    HIPCHK(hipMemsetAsync(g_prof, 0, sizeof(long long) * 200 * batch, g_stream));
    g_prof_batch = batch;
    return TTN_OK;
}
int ensure_batch_bufs(int batch) {
    if (batch > g_dout_cap) {
        if (g_dout) { HIPCHK(hipStreamSynchronize(g_stream)); HIPCHK(hipFree(g_dout)); }
        HIPCHK(hipMalloc((void**)&g_dout, sizeof(double) * batch));
        g_dout_cap = batch;
    }
    return TTN_OK;
}
}  // namespace

// ------------------------------------------------------------------------------------------------
// handles
// ------------------------------------------------------------------------------------------------
struct ttn_tt_s {
    int d = 0, batch = 0;
    std::vector<int64_t> dims, cap, bound;   // bound[m]: host upper bound on the current rank of any train
    std::vector<long long> off;               // d+1 slot offsets
    long long stride = 0;
    double* d_data = nullptr;
    long long* d_off = nullptr;
    long long* d_rks = nullptr;
    long long* d_cap = nullptr;
    int* d_dims = nullptr;
    std::vector<int64_t> ot;                  // [batch][d] host-side gauge flags (never data dependent)
    int* d_status = nullptr;                  // [2][batch]: failure codes of the dense kernels that wrote THIS handle — sticky (kernels only
                                              // store non-zero codes, ttn_compress_status reads and clears) — then the sweep counts of the last launch
    // singular-value capture
    bool sv_on = false;
    double* d_sv = nullptr;
    int sv_steps = 0, sv_pmax = 0;
    TTDev dev() const {
        TTDev t;
        t.data = d_data; t.stride = stride; t.off = d_off; t.rks = d_rks; t.dims = d_dims; t.cap = d_cap;
        t.d = d; t.batch = batch;
        return t;
    }
};
struct ttn_tto_s {
    int d = 0;
    std::vector<int64_t> dims, rks;
    std::vector<long long> off;
    double* d_data = nullptr;
    long long* d_off = nullptr;
    long long* d_rks = nullptr;
    int* d_dims = nullptr;
    TTODev dev() const {
        TTODev t;
        t.data = d_data; t.off = d_off; t.rks = d_rks; t.dims = d_dims; t.d = d;
        return t;
    }
};

static bool same_dims(const std::vector<int64_t>& a, const std::vector<int64_t>& b) { return a == b; }

// ttn_tt_free: the largest failure code still recorded on the dying handle moves to the library-level word
__global__ void k_fold_status(const int* status, int batch, int* pending) {
    int m = 0;
    for (int b = threadIdx.x; b < batch; b += 64) m = max(m, status[b]);
    if (m) atomicMax(pending, m);
}

extern "C" {

const char* ttn_version(void) { return "ttn-mi355x 0.1.0 (gfx950, fp64)"; }
const char* ttn_last_error_string(void) { return g_err.c_str(); }

int ttn_device_count(int* n) {
    if (!n) return fail(TTN_ERR_ARG, "null pointer");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; return hipfail(e, "hipGetDeviceCount"); }
    *n = c;
    return TTN_OK;
}

int ttn_init(int device) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (g_init && device == g_device) return TTN_OK;
    if (g_init) return fail(TTN_ERR_ARG, "ttn_init: already bound to another device (call ttn_finalize first)");
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreate(&g_ev0));
    HIPCHK(hipEventCreate(&g_ev1));
    // the compress / orthogonalize kernels use more than the default 64 KiB of dynamic LDS
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_compress), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(COMPRESS_LDS_BYTES)));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_selftest_eig128), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)COMPRESS_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_mals_linsolve), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)COMPRESS_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_als_linsolve), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)COMPRESS_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ttv_decomp), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)COMPRESS_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_swap_chain), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)COMPRESS_LDS_BYTES));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_orthogonalize), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(ORTHO_LDS_BYTES)));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ortho512), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(O5_LDS_BYTES(TTN_MAX_D * 8))));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dot_fused), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(DOT_LDS_BYTES(DOT_MAX_D))));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_selftest_gemm), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(sizeof(double) * GEMM_LDS_TOTAL)));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bench_gemm), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(sizeof(double) * GEMM_LDS_TOTAL)));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_bench_lds), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)(COMPRESS_LDS_BYTES)));
    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_tdvp), hipFuncAttributeMaxDynamicSharedMemorySize, (int)TDVP_LDS_BYTES));
    { const int rc512 = ttn_wg512_init(); if (rc512) return hipfail((hipError_t)rc512, "ttn_wg512_init"); }
    static_assert(sizeof(CompressArgs) > 0, "");
    if (ttn_wg512_compress_args_bytes() != sizeof(CompressArgs)) return fail(TTN_ERR_ARG, "ttn_init: the two kernel builds disagree on CompressArgs");
    HIPCHK(hipMalloc((void**)&g_next_train, sizeof(int)));
    HIPCHK(hipMalloc((void**)&g_pending_status, sizeof(int)));
    HIPCHK(hipMemset(g_pending_status, 0, sizeof(int)));
    g_device = device;
    g_init = true;
    return TTN_OK;
}

int ttn_finalize(void) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!g_init) return TTN_OK;
    hipStreamSynchronize(g_stream);
    if (g_scratch) hipFree(g_scratch);
    if (g_dout) hipFree(g_dout);
    if (g_next_train) hipFree(g_next_train);
    if (g_pending_status) hipFree(g_pending_status);
    g_next_train = nullptr; g_pending_status = nullptr;
    g_scratch = nullptr; g_scratch_bytes = 0; g_dout = nullptr; g_dout_cap = 0;
    hipEventDestroy(g_ev0); hipEventDestroy(g_ev1);
    for (auto e : g_slots) if (e) hipEventDestroy(e);
    g_slots.clear();
    hipStreamDestroy(g_stream);
    g_stream = nullptr; g_init = false; g_device = -1;
    return TTN_OK;
}

int ttn_sync(void) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    HIPCHK(hipStreamSynchronize(g_stream));
    return TTN_OK;
}

int ttn_timer_begin(void) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    HIPCHK(hipEventRecord(g_ev0, g_stream));
    return TTN_OK;
}
int ttn_timer_end(float* ms) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!ms) return fail(TTN_ERR_ARG, "null pointer");
    HIPCHK(hipEventRecord(g_ev1, g_stream));
    HIPCHK(hipEventSynchronize(g_ev1));
    HIPCHK(hipEventElapsedTime(ms, g_ev0, g_ev1));
    return TTN_OK;
}

int ttn_event_record(int64_t slot) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (slot < 0 || slot >= 4096) return fail(TTN_ERR_ARG, "event slot out of range");
    if ((int64_t)g_slots.size() <= slot) g_slots.resize(slot + 1, nullptr);
    if (!g_slots[slot]) HIPCHK(hipEventCreate(&g_slots[slot]));
    HIPCHK(hipEventRecord(g_slots[slot], g_stream));
    return TTN_OK;
}
int ttn_event_elapsed(int64_t a, int64_t b, float* ms) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!ms || a < 0 || b < 0 || a >= (int64_t)g_slots.size() || b >= (int64_t)g_slots.size() || !g_slots[a] || !g_slots[b])
        return fail(TTN_ERR_ARG, "event slot not recorded");
    HIPCHK(hipEventSynchronize(g_slots[b]));
    HIPCHK(hipEventElapsedTime(ms, g_slots[a], g_slots[b]));
    return TTN_OK;
}

// r_and_d_to_rks with Julia's wrapping Int64 products (src/tt_tools.jl:407-425)
static int64_t wrap_prod(const int64_t* v, int64_t lo, int64_t hi) {  // product of v[lo..hi)
    uint64_t p = 1;
    for (int64_t i = lo; i < hi; ++i) p *= (uint64_t)v[i];
    return (int64_t)p;
}
int ttn_r_and_d_to_rks(int64_t d, const int64_t* dims, int64_t n_rks, const int64_t* rks, int64_t rmax, int64_t* out) {
    if (!dims || !rks || !out || d < 0 || n_rks < 0) return fail(TTN_ERR_ARG, "bad argument");
    for (int64_t i = 0; i < n_rks; ++i) out[i] = 1;
    for (int64_t i = 0; i < d && i < n_rks; ++i) {
        const int64_t q = wrap_prod(dims, i, d), p = wrap_prod(dims, 0, i);
        int64_t v = rks[i];
        if (q > 0) {
            if (p > 0) v = std::min(std::min(v, p), std::min(q, rmax));
            else v = std::min(v, std::min(q, rmax));
        } else {
            if (p > 0) v = std::min(v, std::min(p, rmax));
            else v = std::min(v, rmax);
        }
        out[i] = v;
    }
    return TTN_OK;
}

// ---- ttn_tt ---------------------------------------------------------------------------------------
int ttn_tt_create(int64_t d, const int64_t* dims, const int64_t* cap_rks, int64_t batch, ttn_tt_t* out) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!dims || !cap_rks || !out || d < 1 || batch < 1) return fail(TTN_ERR_ARG, "ttn_tt_create: bad argument");
    for (int64_t k = 0; k < d; ++k) if (dims[k] < 1) return fail(TTN_ERR_ARG, "ttn_tt_create: dims must be >= 1");
    for (int64_t k = 0; k <= d; ++k) if (cap_rks[k] < 1) return fail(TTN_ERR_ARG, "ttn_tt_create: ranks must be >= 1");
    ttn_tt_s* h = new ttn_tt_s();
    h->d = (int)d; h->batch = (int)batch;
    h->dims.assign(dims, dims + d);
    h->cap.assign(cap_rks, cap_rks + d + 1);
    h->bound.assign(d + 1, 1);          // every train starts as the rank-1 zero train
    h->off.resize(d + 1);
    long long o = 0;
    for (int64_t k = 0; k < d; ++k) {
        h->off[k] = o;
        long long sz = (long long)dims[k] * cap_rks[k] * cap_rks[k + 1];
        sz = (sz + 1) & ~1LL;      // keep every slot 16-byte aligned
        o += sz;
    }
    h->off[d] = o;
    h->stride = o;
    h->ot.assign((size_t)batch * d, 0);
    std::vector<int> idims(d);
    for (int64_t k = 0; k < d; ++k) idims[k] = (int)dims[k];
    std::vector<long long> cap64(cap_rks, cap_rks + d + 1);
    std::vector<long long> rk0((size_t)batch * (d + 1));
    for (int64_t b = 0; b < batch; ++b) for (int64_t m = 0; m <= d; ++m) rk0[b * (d + 1) + m] = 1;
    hipError_t e;
    if ((e = hipMalloc((void**)&h->d_data, sizeof(double) * (size_t)o * batch)) != hipSuccess ||
        (e = hipMalloc((void**)&h->d_off, sizeof(long long) * (d + 1))) != hipSuccess ||
        (e = hipMalloc((void**)&h->d_cap, sizeof(long long) * (d + 1))) != hipSuccess ||
        (e = hipMalloc((void**)&h->d_rks, sizeof(long long) * (size_t)batch * (d + 1))) != hipSuccess ||
        (e = hipMalloc((void**)&h->d_dims, sizeof(int) * d)) != hipSuccess ||
        (e = hipMalloc((void**)&h->d_status, sizeof(int) * 2 * (size_t)batch)) != hipSuccess) {
        ttn_tt_free(h);
        return hipfail(e, "hipMalloc(ttn_tt)");
    }
    HIPCHK(hipMemcpyAsync(h->d_off, h->off.data(), sizeof(long long) * (d + 1), hipMemcpyHostToDevice, g_stream));
    HIPCHK(hipMemcpyAsync(h->d_cap, cap64.data(), sizeof(long long) * (d + 1), hipMemcpyHostToDevice, g_stream));
    HIPCHK(hipMemcpyAsync(h->d_rks, rk0.data(), sizeof(long long) * rk0.size(), hipMemcpyHostToDevice, g_stream));
    HIPCHK(hipMemcpyAsync(h->d_dims, idims.data(), sizeof(int) * d, hipMemcpyHostToDevice, g_stream));
    HIPCHK(hipMemsetAsync(h->d_data, 0, sizeof(double) * (size_t)o * batch, g_stream));
    HIPCHK(hipMemsetAsync(h->d_status, 0, sizeof(int) * 2 * (size_t)batch, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    g_live.insert(h);
    *out = h;
    return TTN_OK;
}

int ttn_tt_free(ttn_tt_t h) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!h) return TTN_OK;
    const bool was_live = g_live.erase(h) > 0;
    if (g_init && was_live && h->d_status && g_pending_status) {
        // a failure recorded on this handle that nobody has queried survives the handle (ttn_status_all)
        hipLaunchKernelGGL(k_fold_status, dim3(1), dim3(64), 0, g_stream, (const int*)h->d_status, h->batch, g_pending_status);
    }
    if (g_init) hipStreamSynchronize(g_stream);
    if (h->d_data) hipFree(h->d_data);
    if (h->d_off) hipFree(h->d_off);
    if (h->d_cap) hipFree(h->d_cap);
    if (h->d_rks) hipFree(h->d_rks);
    if (h->d_dims) hipFree(h->d_dims);
    if (h->d_sv) hipFree(h->d_sv);
    if (h->d_status) hipFree(h->d_status);
    delete h;
    return TTN_OK;
}

int ttn_tt_batch(ttn_tt_t h, int64_t* batch) {
    if (!h || !batch) return fail(TTN_ERR_ARG, "null pointer");
    *batch = h->batch;
    return TTN_OK;
}

int ttn_tt_upload(ttn_tt_t h, int64_t b, const double* const* cores, const int64_t* rks, const int64_t* ot) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!h || !cores || !rks || b < 0 || b >= h->batch) return fail(TTN_ERR_ARG, "ttn_tt_upload: bad argument");
    const int d = h->d;
    for (int m = 0; m <= d; ++m) {
        if (rks[m] < 1) return fail(TTN_ERR_ARG, "ttn_tt_upload: ranks must be >= 1");
        if (rks[m] > h->cap[m]) return fail(TTN_ERR_CAPACITY, "ttn_tt_upload: rank exceeds the handle's capacity");
    }
    std::vector<long long> r64(rks, rks + d + 1);
    HIPCHK(hipMemcpyAsync(h->d_rks + (size_t)b * (d + 1), r64.data(), sizeof(long long) * (d + 1), hipMemcpyHostToDevice, g_stream));
    for (int k = 0; k < d; ++k) {
        if (!cores[k]) return fail(TTN_ERR_ARG, "ttn_tt_upload: null core");
        const size_t sz = (size_t)h->dims[k] * rks[k] * rks[k + 1];
        HIPCHK(hipMemcpyAsync(h->d_data + (size_t)b * h->stride + h->off[k], cores[k], sizeof(double) * sz, hipMemcpyHostToDevice, g_stream));
    }
    HIPCHK(hipStreamSynchronize(g_stream));   // host buffers may be released by the caller
    for (int k = 0; k < d; ++k) h->ot[(size_t)b * d + k] = ot ? ot[k] : 0;
    // host-side upper bound on the current ranks of any train of the batch
    for (int m = 0; m <= d; ++m) h->bound[m] = (h->batch == 1) ? rks[m] : std::max<int64_t>(h->bound[m], rks[m]);
    return TTN_OK;
}

int ttn_tt_replicate(ttn_tt_t h, int64_t src_b) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!h || src_b < 0 || src_b >= h->batch) return fail(TTN_ERR_ARG, "ttn_tt_replicate: bad argument");
    if (h->batch > 1) {
        dim3 grid((unsigned)std::min<long long>((h->stride + TTN_STREAM_TB - 1) / TTN_STREAM_TB, 1024), (unsigned)h->batch);
        hipLaunchKernelGGL(k_replicate, grid, dim3(TTN_STREAM_TB), 0, g_stream, h->dev(), (int)src_b);
        HIPCHK(hipGetLastError());
    }
    for (int b = 0; b < h->batch; ++b)
        for (int k = 0; k < h->d; ++k) h->ot[(size_t)b * h->d + k] = h->ot[(size_t)src_b * h->d + k];
    return TTN_OK;
}

int ttn_tt_ranks(ttn_tt_t h, int64_t b, int64_t* rks, int64_t* ot) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!h || b < 0 || b >= h->batch) return fail(TTN_ERR_ARG, "ttn_tt_ranks: bad argument");
    const int d = h->d;
    if (rks) {
        std::vector<long long> r64(d + 1);
        HIPCHK(hipMemcpyAsync(r64.data(), h->d_rks + (size_t)b * (d + 1), sizeof(long long) * (d + 1), hipMemcpyDeviceToHost, g_stream));
        HIPCHK(hipStreamSynchronize(g_stream));
        for (int m = 0; m <= d; ++m) rks[m] = r64[m];
    }
    if (ot) for (int k = 0; k < d; ++k) ot[k] = h->ot[(size_t)b * d + k];
    return TTN_OK;
}

int ttn_tt_max_ranks(ttn_tt_t h, int64_t* bound) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!h) return fail(TTN_ERR_ARG, "null handle");
    const int d = h->d;
    std::vector<long long> r64((size_t)h->batch * (d + 1));
    HIPCHK(hipMemcpyAsync(r64.data(), h->d_rks, sizeof(long long) * r64.size(), hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    for (int m = 0; m <= d; ++m) {
        long long mx = 1;
        for (int b = 0; b < h->batch; ++b) mx = std::max(mx, r64[(size_t)b * (d + 1) + m]);
        h->bound[m] = mx;
        if (bound) bound[m] = mx;
    }
    return TTN_OK;
}

int ttn_tt_download(ttn_tt_t h, int64_t b, double* const* cores) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!h || !cores || b < 0 || b >= h->batch) return fail(TTN_ERR_ARG, "ttn_tt_download: bad argument");
    const int d = h->d;
    std::vector<long long> r64(d + 1);
    HIPCHK(hipMemcpyAsync(r64.data(), h->d_rks + (size_t)b * (d + 1), sizeof(long long) * (d + 1), hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    for (int k = 0; k < d; ++k) {
        if (!cores[k]) return fail(TTN_ERR_ARG, "ttn_tt_download: null core");
        const size_t sz = (size_t)h->dims[k] * r64[k] * r64[k + 1];
        HIPCHK(hipMemcpyAsync(cores[k], h->d_data + (size_t)b * h->stride + h->off[k], sizeof(double) * sz, hipMemcpyDeviceToHost, g_stream));
    }
    HIPCHK(hipStreamSynchronize(g_stream));
    return TTN_OK;
}

int ttn_tt_copy(ttn_tt_t dst, ttn_tt_t src) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!dst || !src) return fail(TTN_ERR_ARG, "null handle");
    if (!same_dims(dst->dims, src->dims) || dst->batch != src->batch) return fail(TTN_ERR_DIMS, "Incompatible dimensions");
    for (int m = 0; m <= src->d; ++m) if (dst->cap[m] < src->bound[m]) return fail(TTN_ERR_CAPACITY, "ttn_tt_copy: destination capacity too small");
    // scale kernel with a = 1 on core -1 is a plain per-core copy that honours the two slot layouts
    const int d = src->d;
    long long maxsz = 0;
    for (int k = 0; k < d; ++k) maxsz = std::max<long long>(maxsz, (long long)src->dims[k] * src->bound[k] * src->bound[k + 1]);
    hipLaunchKernelGGL(k_ranks_copy, dim3(src->batch), dim3(64), 0, g_stream, dst->dev(), src->dev());
    dim3 grid((unsigned)std::max<long long>(1, std::min<long long>((maxsz + TTN_STREAM_TB - 1) / TTN_STREAM_TB, 2048)), (unsigned)d, (unsigned)src->batch);
    hipLaunchKernelGGL(k_scale, grid, dim3(TTN_STREAM_TB), 0, g_stream, src->dev(), dst->dev(), 1.0, -1, 0, (const int*)nullptr);
    HIPCHK(hipGetLastError());
    dst->bound = src->bound;
    dst->ot = src->ot;
    return TTN_OK;
}

// ---- ttn_tto --------------------------------------------------------------------------------------
int ttn_tto_create(int64_t d, const int64_t* dims, const int64_t* rks, const double* const* cores, ttn_tto_t* out) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!dims || !rks || !cores || !out || d < 1) return fail(TTN_ERR_ARG, "ttn_tto_create: bad argument");
    ttn_tto_s* h = new ttn_tto_s();
    h->d = (int)d;
    h->dims.assign(dims, dims + d);
    h->rks.assign(rks, rks + d + 1);
    h->off.resize(d + 1);
    long long o = 0;
    for (int64_t k = 0; k < d; ++k) {
        h->off[k] = o;
        long long sz = (long long)dims[k] * dims[k] * rks[k] * rks[k + 1];
        sz = (sz + 1) & ~1LL;
        o += sz;
    }
    h->off[d] = o;
    std::vector<double> flat((size_t)o, 0.0);
    for (int64_t k = 0; k < d; ++k) {
        if (!cores[k]) { delete h; return fail(TTN_ERR_ARG, "ttn_tto_create: null core"); }
        std::memcpy(flat.data() + h->off[k], cores[k], sizeof(double) * (size_t)dims[k] * dims[k] * rks[k] * rks[k + 1]);
    }
    std::vector<int> idims(d);
    for (int64_t k = 0; k < d; ++k) idims[k] = (int)dims[k];
    std::vector<long long> r64(rks, rks + d + 1);
    hipError_t e;
    if ((e = hipMalloc((void**)&h->d_data, sizeof(double) * (size_t)std::max<long long>(o, 1))) != hipSuccess ||
        (e = hipMalloc((void**)&h->d_off, sizeof(long long) * (d + 1))) != hipSuccess ||
        (e = hipMalloc((void**)&h->d_rks, sizeof(long long) * (d + 1))) != hipSuccess ||
        (e = hipMalloc((void**)&h->d_dims, sizeof(int) * d)) != hipSuccess) {
        ttn_tto_free(h);
        return hipfail(e, "hipMalloc(ttn_tto)");
    }
    HIPCHK(hipMemcpyAsync(h->d_data, flat.data(), sizeof(double) * (size_t)o, hipMemcpyHostToDevice, g_stream));
    HIPCHK(hipMemcpyAsync(h->d_off, h->off.data(), sizeof(long long) * (d + 1), hipMemcpyHostToDevice, g_stream));
    HIPCHK(hipMemcpyAsync(h->d_rks, r64.data(), sizeof(long long) * (d + 1), hipMemcpyHostToDevice, g_stream));
    HIPCHK(hipMemcpyAsync(h->d_dims, idims.data(), sizeof(int) * d, hipMemcpyHostToDevice, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    *out = h;
    return TTN_OK;
}

int ttn_tto_free(ttn_tto_t h) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!h) return TTN_OK;
    if (g_init) hipStreamSynchronize(g_stream);
    if (h->d_data) hipFree(h->d_data);
    if (h->d_off) hipFree(h->d_off);
    if (h->d_rks) hipFree(h->d_rks);
    if (h->d_dims) hipFree(h->d_dims);
    delete h;
    return TTN_OK;
}

// ---- streaming ops --------------------------------------------------------------------------------
static dim3 stream_grid(long long max_items, int d, int batch) {
    long long gx = (max_items + TTN_STREAM_TB - 1) / TTN_STREAM_TB;
    gx = std::max<long long>(1, std::min<long long>(gx, 4096));
    return dim3((unsigned)gx, (unsigned)d, (unsigned)batch);
}

// k_apply / k_hadamard / k_add count the fibres (p, q) of a core with 32-bit indices (ttn_stream_kernels.h): refuse larger cores here
static bool stream_fibres_too_many(long long fibres) { return fibres >= (1LL << 31); }

int ttn_apply(ttn_tto_t A, ttn_tt_t x, ttn_tt_t y) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!A || !x || !y) return fail(TTN_ERR_ARG, "null handle");
    if (!same_dims(A->dims, x->dims) || !same_dims(x->dims, y->dims)) return fail(TTN_ERR_DIMS, "Incompatible dimensions");
    if (x->batch != y->batch) return fail(TTN_ERR_DIMS, "batch sizes differ");
    if (x == y) return fail(TTN_ERR_ARG, "ttn_apply: output must not alias the input");
    const int d = x->d;
    long long maxfib = 0;
    for (int m = 0; m <= d; ++m) if (y->cap[m] < A->rks[m] * x->bound[m]) return fail(TTN_ERR_CAPACITY, "ttn_apply: destination capacity too small");
    for (int k = 0; k < d; ++k) maxfib = std::max<long long>(maxfib, (long long)x->bound[k] * x->bound[k + 1]);
    if (stream_fibres_too_many(maxfib * std::max<long long>(1, A->rks[0]))) return fail(TTN_ERR_UNSUPPORTED, "ttn_apply: 2^31 or more fibres in one core (32-bit element indices)");
    for (int k = 0; k < d; ++k) if (stream_fibres_too_many((long long)y->cap[k] * y->cap[k + 1])) return fail(TTN_ERR_UNSUPPORTED, "ttn_apply: 2^31 or more fibres in one output core (32-bit element indices)");
    hipLaunchKernelGGL(k_ranks_mul_op, dim3(x->batch), dim3(64), 0, g_stream, y->dev(), A->dev(), x->dev());
    // LDS of k_apply: the largest operator core (if it fits TTN_APPLY_LDS_DOUBLES) + the store-transpose buffer for the largest left rank
    long long amax_ = 0, rlmax_ = 1;
    for (int k = 0; k < d; ++k) { amax_ = std::max<long long>(amax_, (long long)A->dims[k] * A->dims[k] * A->rks[k] * A->rks[k + 1]); rlmax_ = std::max<long long>(rlmax_, A->rks[k]); }
    const int lds_a = amax_ <= TTN_APPLY_LDS_DOUBLES ? (int)amax_ : 0;
    const int lds_rl = rlmax_ <= TTN_APPLY_MAX_RL ? (int)rlmax_ : 0;
    const size_t apply_lds = sizeof(double) * (size_t)((lds_a + 1) & ~1) + sizeof(double) * 2 * (size_t)(TTN_STREAM_TB / 64) * lds_rl * 64;
    // grid: output rows x groups of TTN_APPLY_K output columns when every site has n = 2 and the operator cores fit the LDS (the mapping
    // of k_apply's fast path), input fibres otherwise
    long long apply_items = maxfib;
    {
        bool qtt = lds_a > 0;
        for (int k = 0; k < d; ++k) qtt = qtt && x->dims[k] == 2;
        if (qtt) {
            apply_items = 0;
            for (int k = 0; k < d; ++k)
                apply_items = std::max<long long>(apply_items, (long long)A->rks[k] * x->bound[k] * (((long long)A->rks[k + 1] * x->bound[k + 1] + TTN_APPLY_K - 1) / TTN_APPLY_K));
        }
    }
    hipLaunchKernelGGL(k_apply, stream_grid(apply_items, d, x->batch), dim3(TTN_STREAM_TB), apply_lds, g_stream, A->dev(), x->dev(), y->dev(), lds_a, lds_rl);
    HIPCHK(hipGetLastError());
    for (int m = 0; m <= d; ++m) y->bound[m] = A->rks[m] * x->bound[m];
    std::fill(y->ot.begin(), y->ot.end(), 0);     // zeros_tt (tt_operations.jl:103)
    return TTN_OK;
}

int ttn_hadamard(ttn_tt_t x, ttn_tt_t y, ttn_tt_t z) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!x || !y || !z) return fail(TTN_ERR_ARG, "null handle");
    if (!same_dims(x->dims, y->dims) || !same_dims(x->dims, z->dims)) return fail(TTN_ERR_DIMS, "Incompatible TT dimensions");
    if (x->batch != y->batch || x->batch != z->batch) return fail(TTN_ERR_DIMS, "batch sizes differ");
    if (z == x || z == y) return fail(TTN_ERR_ARG, "ttn_hadamard: output must not alias an input");
    const int d = x->d;
    long long maxpq = 0;
    for (int m = 0; m <= d; ++m) if (z->cap[m] < x->bound[m] * y->bound[m]) return fail(TTN_ERR_CAPACITY, "ttn_hadamard: destination capacity too small");
    for (int k = 0; k < d; ++k) maxpq = std::max<long long>(maxpq, (long long)x->bound[k] * y->bound[k] * x->bound[k + 1] * y->bound[k + 1]);
    if (stream_fibres_too_many(maxpq)) return fail(TTN_ERR_UNSUPPORTED, "ttn_hadamard: 2^31 or more fibres in one core (32-bit element indices)");
    hipLaunchKernelGGL(k_ranks_mul, dim3(x->batch), dim3(64), 0, g_stream, z->dev(), x->dev(), y->dev());
    bool qtt = true;
    for (int k = 0; k < d; ++k) qtt = qtt && x->dims[k] == 2;
    hipLaunchKernelGGL(k_hadamard, stream_grid(qtt ? (maxpq + TTN_HAD_K - 1) / TTN_HAD_K : maxpq, d, x->batch), dim3(TTN_STREAM_TB), 0, g_stream, x->dev(), y->dev(), z->dev());
    HIPCHK(hipGetLastError());
    for (int m = 0; m <= d; ++m) z->bound[m] = x->bound[m] * y->bound[m];
    std::fill(z->ot.begin(), z->ot.end(), 0);
    return TTN_OK;
}

int ttn_add(ttn_tt_t x, ttn_tt_t y, ttn_tt_t z) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!x || !y || !z) return fail(TTN_ERR_ARG, "null handle");
    if (!same_dims(x->dims, y->dims) || !same_dims(x->dims, z->dims)) return fail(TTN_ERR_DIMS, "Incompatible dimensions");
    if (x->batch != y->batch || x->batch != z->batch) return fail(TTN_ERR_DIMS, "batch sizes differ");
    if (z == x || z == y) return fail(TTN_ERR_ARG, "ttn_add: output must not alias an input");
    const int d = x->d;
    if (d < 2) return fail(TTN_ERR_UNSUPPORTED, "ttn_add: the reference's + is only defined for d >= 2");
    std::vector<int64_t> zb(d + 1);
    for (int m = 0; m <= d; ++m) zb[m] = (m == 0 || m == d) ? 1 : x->bound[m] + y->bound[m];
    long long maxpq = 0;
    for (int m = 0; m <= d; ++m) if (z->cap[m] < zb[m]) return fail(TTN_ERR_CAPACITY, "ttn_add: destination capacity too small");
    for (int k = 0; k < d; ++k) maxpq = std::max<long long>(maxpq, (long long)zb[k] * zb[k + 1]);
    if (stream_fibres_too_many(maxpq)) return fail(TTN_ERR_UNSUPPORTED, "ttn_add: 2^31 or more fibres in one core (32-bit element indices)");
    hipLaunchKernelGGL(k_ranks_add, dim3(x->batch), dim3(64), 0, g_stream, z->dev(), x->dev(), y->dev());
    bool qtt = true;
    for (int k = 0; k < d; ++k) qtt = qtt && x->dims[k] == 2;
    hipLaunchKernelGGL(k_add, stream_grid(qtt ? (maxpq + TTN_ADD_K - 1) / TTN_ADD_K : maxpq, d, x->batch), dim3(TTN_STREAM_TB), 0, g_stream, x->dev(), y->dev(), z->dev());
    HIPCHK(hipGetLastError());
    z->bound = zb;
    std::fill(z->ot.begin(), z->ot.end(), 0);
    return TTN_OK;
}

// The core scalar multiplication scales: the first one with ot == 0, else the first (tt_operations.jl:262).  Uniform over the batch
// -> `which`; trains with different gauge flags (uploaded one by one, or zeroed by ttn_scale_batch) -> a per-train device table.
static int scaled_core(ttn_tt_t x, int& which, const int*& which_b) {
    static int* d_which = nullptr; static int which_cap = 0;
    static std::vector<int> h_which;                     // must outlive the async upload
    const int d = x->d;
    h_which.assign(x->batch, 0);
    bool uniform = true;
    for (int b = 0; b < x->batch; ++b) {
        for (int k = 0; k < d; ++k) if (x->ot[(size_t)b * d + k] == 0) { h_which[b] = k; break; }
        uniform = uniform && h_which[b] == h_which[0];
    }
    which = h_which[0];
    which_b = nullptr;
    if (uniform) return TTN_OK;
    if (which_cap < x->batch) {
        if (d_which) { HIPCHK(hipStreamSynchronize(g_stream)); HIPCHK(hipFree(d_which)); d_which = nullptr; }
        HIPCHK(hipMalloc((void**)&d_which, sizeof(int) * x->batch));
        which_cap = x->batch;
    }
    HIPCHK(hipMemcpyAsync(d_which, h_which.data(), sizeof(int) * x->batch, hipMemcpyHostToDevice, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));              // h_which is reused by the next call
    which_b = d_which;
    return TTN_OK;
}

int ttn_scale(double a, ttn_tt_t x, ttn_tt_t y) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!x || !y) return fail(TTN_ERR_ARG, "null handle");
    if (!same_dims(x->dims, y->dims) || x->batch != y->batch) return fail(TTN_ERR_DIMS, "Incompatible dimensions");
    const int d = x->d;
    for (int m = 0; m <= d; ++m) if (y->cap[m] < x->bound[m]) return fail(TTN_ERR_CAPACITY, "ttn_scale: destination capacity too small");
    // i = findfirst(==(0), ot), else 1  (tt_operations.jl:262), per train
    int which = 0;
    const int* which_b = nullptr;
    { int rc_ = scaled_core(x, which, which_b); if (rc_) return rc_; }
    long long maxsz = 0;
    for (int k = 0; k < d; ++k) maxsz = std::max<long long>(maxsz, (long long)x->dims[k] * x->bound[k] * x->bound[k + 1]);
    if (x != y) hipLaunchKernelGGL(k_ranks_copy, dim3(x->batch), dim3(64), 0, g_stream, y->dev(), x->dev());
    hipLaunchKernelGGL(k_scale, stream_grid((maxsz + 7) / 8, d, x->batch), dim3(TTN_STREAM_TB), 0, g_stream, x->dev(), y->dev(), a, which, a == 0.0 ? 1 : 0, which_b);
    HIPCHK(hipGetLastError());
    y->bound = x->bound;
    if (a == 0.0) std::fill(y->ot.begin(), y->ot.end(), 0); else y->ot = x->ot;
    return TTN_OK;
}

int ttn_scale_batch(const double* a, ttn_tt_t x, ttn_tt_t y) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!a || !x || !y) return fail(TTN_ERR_ARG, "null pointer");
    if (!same_dims(x->dims, y->dims) || x->batch != y->batch) return fail(TTN_ERR_DIMS, "Incompatible dimensions");
    const int d = x->d;
    for (int m = 0; m <= d; ++m) if (y->cap[m] < x->bound[m]) return fail(TTN_ERR_CAPACITY, "ttn_scale_batch: destination capacity too small");
    int rc = ensure_batch_bufs(x->batch);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(g_dout, a, sizeof(double) * x->batch, hipMemcpyHostToDevice, g_stream));
    int which = 0;
    const int* which_b = nullptr;
    { int rc_ = scaled_core(x, which, which_b); if (rc_) return rc_; }
    long long maxsz = 0;
    for (int k = 0; k < d; ++k) maxsz = std::max<long long>(maxsz, (long long)x->dims[k] * x->bound[k] * x->bound[k + 1]);
    if (x != y) hipLaunchKernelGGL(k_ranks_copy, dim3(x->batch), dim3(64), 0, g_stream, y->dev(), x->dev());
    hipLaunchKernelGGL(k_scale_batch, stream_grid((maxsz + 7) / 8, d, x->batch), dim3(TTN_STREAM_TB), 0, g_stream, x->dev(), y->dev(), (const double*)g_dout, which, which_b);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(g_stream));      // `a` is caller memory and g_dout is reused by ttn_dot
    y->bound = x->bound;
    y->ot = x->ot;
    for (int b = 0; b < x->batch; ++b) if (a[b] == 0.0) for (int k = 0; k < d; ++k) y->ot[(size_t)b * d + k] = 0;
    return TTN_OK;
}

// ---- dense ops ------------------------------------------------------------------------------------
// Upper bounds on the ranks during / after tt_compress! (or one _tt_bond_truncate!).  The reference sets
// r = min(length(s), max_bond) with length(s) = min(n_k r_{k-1}, n_{k+1} r_{k+1}) (tt_cross_interpolation.jl:152,164),
// so a bond rank can GROW when it was below both of those (rank-deficient input).  need[m] = largest rank bond m can
// take at any time (buffers / handle capacity must hold it), fin[m] = bound after the call.
static void rank_bounds(int d, const int64_t* dims, const int64_t* rks, int64_t max_bond, int64_t sweeps, int64_t k_single,
                        std::vector<int64_t>& need, std::vector<int64_t>& fin, long long& pmax, long long& qmax,
                        int64_t k_first = 0, int64_t k_last = 0 /* 0-based bond range when k_single < 0 */) {
    fin.assign(rks, rks + d + 1);
    need = fin;
    pmax = 1; qmax = 1;
    auto stepk = [&](int k) {      // 0-based bond between cores k, k+1
        const long long mr = dims[k] * fin[k], mc = dims[k + 1] * fin[k + 2];
        const long long p = std::min(mr, mc), q = std::max(mr, mc);
        pmax = std::max(pmax, p); qmax = std::max(qmax, q);
        fin[k + 1] = std::min<long long>(p, max_bond);
        need[k + 1] = std::max(need[k + 1], fin[k + 1]);
    };
    if (k_single > 0) { stepk((int)k_single - 1); return; }
    if (k_single < 0) {
        if (k_first <= k_last) { for (int64_t k = k_first; k <= k_last; ++k) stepk((int)k); }
        else { for (int64_t k = k_first; k >= k_last; --k) stepk((int)k); }
        return;
    }
    for (int64_t sw = 0; sw < sweeps; ++sw) {
        for (int k = 0; k + 1 < d; ++k) stepk(k);
        for (int k = d - 2; k >= 0; --k) stepk(k);
    }
}

int ttn_compress_rank_bound(int64_t d, const int64_t* dims, const int64_t* rks, int64_t max_bond, int64_t sweeps, int64_t k,
                            int64_t* need_out, int64_t* final_out) {
    if (!dims || !rks || d < 1 || max_bond < 1 || sweeps < 1 || k < 0 || k > d - 1)
        return fail(TTN_ERR_ARG, "bad argument");
    std::vector<int64_t> need, fin;
    long long pm, qm;
    rank_bounds((int)d, dims, rks, max_bond, sweeps, k, need, fin, pm, qm);
    for (int64_t m = 0; m <= d; ++m) { if (need_out) need_out[m] = need[m]; if (final_out) final_out[m] = fin[m]; }
    return TTN_OK;
}

// Everything that can refuse a compress launch — capacity of the handle for the ranks the sweep can reach from `bound`, the size
// limits of the merged matrices, the scratch allocation — checked WITHOUT touching the handle (ttn_apply_compress runs this on the
// product's ranks before it overwrites y's).
// Which build runs a compress launch, and on how many workgroup slots.  More trains than CUs: the 512-thread build on a PERSISTENT
// grid of two workgroups per CU that pull trains from a counter (scratch per slot); otherwise one 1024-thread workgroup per train
// (lowest latency for a single train).  TTN_WG512=1 / 0 forces / forbids the 512-thread build (diagnostics, parity tests).
#define TTN_NUM_CUS 256
static bool compress_use_wg512(int batch) {
    const char* e = getenv("TTN_WG512");
    if (e) return atoi(e) != 0;
    return batch > TTN_NUM_CUS;
}
static int compress_slots(int batch) { return compress_use_wg512(batch) ? std::min(batch, 2 * TTN_NUM_CUS) : batch; }

static int compress_precheck(ttn_tt_t psi, const std::vector<int64_t>& bound, int64_t k_single, int64_t max_bond, int64_t sweeps,
                             int64_t k_first, int64_t k_last, std::vector<int64_t>& fin, long long& pmax, long long& qmax, long long& per_train) {
    const int d = psi->d;
    std::vector<int64_t> need;
    pmax = 1; qmax = 1;
    rank_bounds(d, psi->dims.data(), bound.data(), max_bond, sweeps, k_single, need, fin, pmax, qmax, k_first, k_last);
    for (int m = 0; m <= d; ++m)
        if (need[m] > psi->cap[m]) return fail(TTN_ERR_CAPACITY, "ttn_compress: a bond rank can grow beyond the handle's capacity (see ttn_compress_rank_bound)");
    if (pmax > 4096 || qmax > 16384) return fail(TTN_ERR_UNSUPPORTED, "ttn_compress: merged matrix larger than 4096 x 16384");
    per_train = 2 * pmax * qmax + QR_NB * qmax + pmax * QR_NB + 2 * pmax * pmax + 4 * pmax + 64 + 6 * 128 * 128;
    int rc = ensure_scratch(sizeof(double) * (size_t)per_train * compress_slots(psi->batch));
    if (rc) return rc;
    return ensure_batch_bufs(psi->batch);
}

static int launch_compress(ttn_tt_t psi, int64_t k_single, int64_t max_bond, double truncerr, int64_t sweeps,
                           int64_t k_first = 0, int64_t k_last = 0, ttn_tto_t fuseA = nullptr, ttn_tt_t fusex = nullptr, int fused_first_real = 0) {
    const int d = psi->d;
    if (d < 2 && k_single == 0) return TTN_OK;
    std::vector<int64_t> fin;
    long long pmax = 1, qmax = 1, per_train = 0;
    int rc = compress_precheck(psi, psi->bound, k_single, max_bond, sweeps, k_first, k_last, fin, pmax, qmax, per_train);
    if (rc) return rc;
    const int steps = k_single > 0 ? 1 : (k_single < 0 ? (int)(std::llabs(k_last - k_first) + 1) : (int)(2 * (d - 1) * sweeps));
    if (psi->sv_on) {
        if (psi->sv_steps < steps || psi->sv_pmax < pmax) {
            if (psi->d_sv) { HIPCHK(hipStreamSynchronize(g_stream)); HIPCHK(hipFree(psi->d_sv)); psi->d_sv = nullptr; }
            HIPCHK(hipMalloc((void**)&psi->d_sv, sizeof(double) * (size_t)psi->batch * steps * pmax));
        }
        psi->sv_steps = steps;
        psi->sv_pmax = (int)pmax;
    }
    CompressArgs P;
    P.tt = psi->dev();
    P.max_bond = max_bond;
    P.truncerr = truncerr;
    P.sweeps = (int)sweeps;
    P.k_single = (int)k_single;
    P.k_first = (int)k_first; P.k_last = (int)k_last;
    P.scratch = (double*)g_scratch;
    P.scratch_stride = per_train;
    P.pmax = (int)pmax; P.qmax = (int)qmax;
    P.sv_out = psi->sv_on ? psi->d_sv : nullptr;
    P.sv_steps = steps;
    P.status = psi->d_status;
    P.sweep_stats = psi->d_status + psi->batch;
    P.rank_rule = 0;
    P.fused = (fuseA && fusex) ? 1 : 0;
    P.fused_first_real = fused_first_real;
    if (P.fused) { P.op = fuseA->dev(); P.x = fusex->dev(); }
    else { memset(&P.op, 0, sizeof(P.op)); memset(&P.x, 0, sizeof(P.x)); }
    P.prof = nullptr;
    { const char* e = getenv("TTN_PROF_STEP"); P.prof_step = e ? atoi(e) : -1; }
    if (getenv("TTN_PROF")) {
        { int rcp = ensure_prof(psi->batch); if (rcp) return rcp; }
        P.prof = g_prof;
    }
    { const char* e = getenv("TTN_JTOL"); P.jtol_mult = e ? atof(e) : 1.0; }
    { const char* e = getenv("TTN_JNEG"); P.jneg_mult = e ? atof(e) : 1.0; }
    { const char* e = getenv("TTN_FAST"); P.fast = e ? atoi(e) : 1; }
    if (compress_use_wg512(psi->batch)) {
        HIPCHK(hipMemsetAsync(g_next_train, 0, sizeof(int), g_stream));
        P.next_train = g_next_train;
        const int rc512 = ttn_wg512_launch_compress(&P, sizeof(P), compress_slots(psi->batch), g_stream);
        if (rc512) return hipfail((hipError_t)rc512, "k_compress (512-thread build)");
    } else {
        P.next_train = nullptr;
        hipLaunchKernelGGL(k_compress, dim3(psi->batch), dim3(TTN_WG), COMPRESS_LDS_BYTES, g_stream, P);
        HIPCHK(hipGetLastError());
    }
    psi->bound = fin;
    return TTN_OK;
}

int ttn_compress(ttn_tt_t psi, int64_t max_bond, double truncerr, int64_t sweeps) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!psi) return fail(TTN_ERR_ARG, "null handle");
    if (sweeps < 1) return fail(TTN_ERR_SWEEPS, "sweeps must be >= 1");
    if (max_bond < 1) return fail(TTN_ERR_ARG, "max_bond must be >= 1");
    return launch_compress(psi, 0, max_bond, truncerr, sweeps);
}

int ttn_bond_truncate(ttn_tt_t psi, int64_t k, int64_t max_bond, double truncerr) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!psi) return fail(TTN_ERR_ARG, "null handle");
    if (k < 1 || k >= psi->d) return fail(TTN_ERR_BOND_INDEX, "k must be in 1:(N-1)");
    if (max_bond < 1) return fail(TTN_ERR_ARG, "max_bond must be >= 1");
    return launch_compress(psi, k, max_bond, truncerr, 1);
}

int ttn_sweep(ttn_tt_t psi, int64_t k_first, int64_t k_last, int64_t max_bond, double truncerr) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!psi) return fail(TTN_ERR_ARG, "null handle");
    if (k_first < 1 || k_first >= psi->d || k_last < 1 || k_last >= psi->d) return fail(TTN_ERR_BOND_INDEX, "k must be in 1:(N-1)");
    if (max_bond < 1) return fail(TTN_ERR_ARG, "max_bond must be >= 1");
    return launch_compress(psi, -1, max_bond, truncerr, 1, k_first - 1, k_last - 1);
}

// ---- boundary-core hand-off of core-wise sharded chains --------------------------------------------
// A core is stored compactly (current ranks) at the start of its slot, so the first dims[k]*bound[k]*bound[k+1] doubles
// of every train's slot carry it; they move to / from a dense [batch][that many] device buffer with one 2-D copy.
int ttn_tt_core_extent(ttn_tt_t h, int64_t k, int64_t* doubles_per_train, int64_t* bound_left, int64_t* bound_right) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!h || k < 1 || k > h->d) return fail(TTN_ERR_ARG, "bad core index");
    const int64_t bl = h->bound[k - 1], br = h->bound[k];
    if (doubles_per_train) *doubles_per_train = h->dims[k - 1] * bl * br;
    if (bound_left) *bound_left = bl;
    if (bound_right) *bound_right = br;
    return TTN_OK;
}

int ttn_tt_core_export(ttn_tt_t h, int64_t k, double* dev_buf, int64_t* dev_rks2) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!h || !dev_buf || !dev_rks2 || k < 1 || k > h->d) return fail(TTN_ERR_ARG, "bad argument");
    const size_t w = sizeof(double) * (size_t)(h->dims[k - 1] * h->bound[k - 1] * h->bound[k]);
    HIPCHK(hipMemcpy2DAsync(dev_buf, w, h->d_data + h->off[k - 1], sizeof(double) * (size_t)h->stride, w, h->batch,
                            hipMemcpyDeviceToDevice, g_stream));
    HIPCHK(hipMemcpy2DAsync(dev_rks2, 2 * sizeof(long long), h->d_rks + (k - 1), sizeof(long long) * (size_t)(h->d + 1),
                            2 * sizeof(long long), h->batch, hipMemcpyDeviceToDevice, g_stream));
    return TTN_OK;
}

int ttn_tt_core_import(ttn_tt_t h, int64_t k, const double* dev_buf, const int64_t* dev_rks2, int64_t bound_left, int64_t bound_right) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!h || !dev_buf || !dev_rks2 || k < 1 || k > h->d || bound_left < 1 || bound_right < 1) return fail(TTN_ERR_ARG, "bad argument");
    if (bound_left > h->cap[k - 1] || bound_right > h->cap[k]) return fail(TTN_ERR_CAPACITY, "ttn_tt_core_import: core does not fit the slot");
    const size_t w = sizeof(double) * (size_t)(h->dims[k - 1] * bound_left * bound_right);
    HIPCHK(hipMemcpy2DAsync(h->d_data + h->off[k - 1], sizeof(double) * (size_t)h->stride, dev_buf, w, w, h->batch,
                            hipMemcpyDeviceToDevice, g_stream));
    HIPCHK(hipMemcpy2DAsync(h->d_rks + (k - 1), sizeof(long long) * (size_t)(h->d + 1), dev_rks2, 2 * sizeof(long long),
                            2 * sizeof(long long), h->batch, hipMemcpyDeviceToDevice, g_stream));
    h->bound[k - 1] = bound_left;
    h->bound[k] = bound_right;
    return TTN_OK;
}

int ttn_apply_compress(ttn_tto_t A, ttn_tt_t x, ttn_tt_t y, int64_t max_bond, double truncerr, int64_t sweeps) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (sweeps < 1) return fail(TTN_ERR_SWEEPS, "sweeps must be >= 1");
    if (!A || !x || !y) return fail(TTN_ERR_ARG, "null handle");
    if (max_bond < 1) return fail(TTN_ERR_ARG, "max_bond must be >= 1");
    const char* nf = getenv("TTN_NOFUSE");
    if (x->d < 2 || (nf && atoi(nf))) {                     // nothing to fuse into / diagnostic switch
        int rc = ttn_apply(A, x, y);
        if (rc) return rc;
        return ttn_compress(y, max_bond, truncerr, sweeps);
    }
    // FUSED: y = A*x is never written to HBM.  y only receives its ranks (A.rks .* x.rks, tt_operations.jl:103); the first
    // L->R sweep of k_compress builds each merged matrix straight from core k of y, x_{k+1} and A_{k+1}.
    if (!same_dims(A->dims, x->dims) || !same_dims(x->dims, y->dims)) return fail(TTN_ERR_DIMS, "Incompatible dimensions");
    if (x->batch != y->batch) return fail(TTN_ERR_DIMS, "batch sizes differ");
    if (x == y) return fail(TTN_ERR_ARG, "ttn_apply_compress: output must not alias the input");
    const int d = x->d;
    for (int m = 0; m <= d; ++m) if (y->cap[m] < A->rks[m] * x->bound[m]) return fail(TTN_ERR_CAPACITY, "ttn_apply: destination capacity too small");
    // every check that can refuse the launch runs on the product's ranks BEFORE y is touched: on an error return y still holds
    // what it held (ranks, bounds, gauge flags and cores)
    std::vector<int64_t> yb(d + 1), fin_;
    for (int m = 0; m <= d; ++m) yb[m] = A->rks[m] * x->bound[m];
    { long long pm_, qm_, pt_; int rc = compress_precheck(y, yb, 0, max_bond, sweeps, 0, 0, fin_, pm_, qm_, pt_); if (rc) return rc; }
    hipLaunchKernelGGL(k_ranks_mul_op, dim3(x->batch), dim3(64), 0, g_stream, y->dev(), A->dev(), x->dev());
    HIPCHK(hipGetLastError());
    y->bound = yb;
    std::fill(y->ot.begin(), y->ot.end(), 0);
    return launch_compress(y, 0, max_bond, truncerr, sweeps, 0, 0, A, x);
}

// ---- fused apply for core-wise sharded chains: the product's ranks first, then ONE L->R pass over a bond range whose right cores
// are still virtual.  Between the two calls a boundary core may be imported into y (ttn_tt_core_import).
int ttn_apply_begin(ttn_tto_t A, ttn_tt_t x, ttn_tt_t y) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!A || !x || !y) return fail(TTN_ERR_ARG, "null handle");
    if (!same_dims(A->dims, x->dims) || !same_dims(x->dims, y->dims)) return fail(TTN_ERR_DIMS, "Incompatible dimensions");
    if (x->batch != y->batch) return fail(TTN_ERR_DIMS, "batch sizes differ");
    if (x == y) return fail(TTN_ERR_ARG, "ttn_apply_begin: output must not alias the input");
    const int d = x->d;
    for (int m = 0; m <= d; ++m) if (y->cap[m] < A->rks[m] * x->bound[m]) return fail(TTN_ERR_CAPACITY, "ttn_apply: destination capacity too small");
    hipLaunchKernelGGL(k_ranks_mul_op, dim3(x->batch), dim3(64), 0, g_stream, y->dev(), A->dev(), x->dev());
    HIPCHK(hipGetLastError());
    for (int m = 0; m <= d; ++m) y->bound[m] = A->rks[m] * x->bound[m];
    std::fill(y->ot.begin(), y->ot.end(), 0);
    return TTN_OK;
}

int ttn_apply_sweep(ttn_tto_t A, ttn_tt_t x, ttn_tt_t y, int64_t k_first, int64_t k_last, int64_t max_bond, double truncerr, int first_core_real) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!A || !x || !y) return fail(TTN_ERR_ARG, "null handle");
    if (!same_dims(A->dims, x->dims) || !same_dims(x->dims, y->dims) || x->batch != y->batch) return fail(TTN_ERR_DIMS, "Incompatible dimensions");
    if (k_first < 1 || k_first >= y->d || k_last < k_first || k_last >= y->d) return fail(TTN_ERR_BOND_INDEX, "ttn_apply_sweep: need 1 <= k_first <= k_last <= N-1 (one L->R pass)");
    if (max_bond < 1) return fail(TTN_ERR_ARG, "max_bond must be >= 1");
    return launch_compress(y, -1, max_bond, truncerr, 1, k_first - 1, k_last - 1, A, x, first_core_real ? 1 : 0);
}

// The HIP stream every call of this library is enqueued on (hipStream_t): lets a caller order its own streams against the
// library's work with events instead of host synchronisation (the boundary-core hand-offs of pipeline.py)
int ttn_stream_handle(void** stream) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!stream) return fail(TTN_ERR_ARG, "null pointer");
    *stream = (void*)g_stream;
    return TTN_OK;
}

// ---- site-swap chains: hadamard_ttm and reorder -----------------------------------------------------
// Launch of k_swap_chain.  `ops` = (type, slotA, slotB) triples; kind 1 works in an arena of 2d uniform slots carved out of
// the scratch allocation, kind 2 in place on the handle's own slots.
static int launch_chain(int kind, ttn_tt_t x, ttn_tt_t y, ttn_tt_t z, int n, int nslots, int64_t work_cap, const std::vector<int>& ops,
                        const std::vector<int>& final_slots, double tol, int64_t rmax, int rank_rule) {
    static std::vector<int> h_tab;                      // must outlive the async upload below
    ttn_tt_t ref = (kind == 1) ? x : z;
    const int batch = ref->batch;
    const long long pmax = (long long)n * work_cap, qmax = pmax;
    if (pmax > 256) return fail(TTN_ERR_UNSUPPORTED, "site-swap chain: n * rank capacity must be <= 256");
    const long long per_compress = 2 * pmax * qmax + QR_NB * qmax + pmax * QR_NB + 2 * pmax * pmax + 4 * pmax + 64 + 6 * 128 * 128;
    const long long slot_doubles = (kind == 1) ? (long long)n * work_cap * work_cap : 0;
    const long long srk_stride = 2 * nslots + 2;
    const long long per_train = per_compress + srk_stride + (long long)nslots * slot_doubles;
    const size_t tab_ints = ops.size() + final_slots.size();
    HIPCHK(hipStreamSynchronize(g_stream));            // h_tab of the previous call is no longer in flight
    int rc = ensure_scratch(sizeof(double) * (size_t)per_train * batch + sizeof(int) * tab_ints + 64);
    if (rc) return rc;
    rc = ensure_batch_bufs(batch);
    if (rc) return rc;
    double* base = (double*)g_scratch;
    int* d_tab = (int*)(base + (size_t)per_train * batch);
    h_tab.assign(ops.begin(), ops.end());
    h_tab.insert(h_tab.end(), final_slots.begin(), final_slots.end());
    HIPCHK(hipMemcpyAsync(d_tab, h_tab.data(), sizeof(int) * tab_ints, hipMemcpyHostToDevice, g_stream));
    ChainArgs Q;
    memset(&Q, 0, sizeof(Q));
    CompressArgs& P = Q.C;
    if (kind == 2) P.tt = z->dev();
    P.max_bond = rmax;
    P.truncerr = tol;
    P.sweeps = 1;
    P.scratch = base;
    P.scratch_stride = per_train;
    P.pmax = (int)pmax; P.qmax = (int)qmax;
    P.sv_out = nullptr; P.sv_steps = 0;
    P.status = z->d_status;                 // the handle the chain writes (kind 2: in place on z)
    P.sweep_stats = z->d_status + batch;
    P.prof = nullptr; P.prof_step = -1;
    { const char* e = getenv("TTN_JTOL"); P.jtol_mult = e ? atof(e) : 1.0; }
    { const char* e = getenv("TTN_JNEG"); P.jneg_mult = e ? atof(e) : 1.0; }
    P.fast = 0;
    P.fused = 0;
    P.rank_rule = rank_rule;
    Q.kind = kind;
    Q.n = n; Q.nslots = nslots; Q.nops = (int)(ops.size() / 3); Q.d = (kind == 1) ? x->d : z->d;
    Q.ops = d_tab;
    Q.final_slots = d_tab + ops.size();
    Q.arena = base + per_compress + srk_stride;        // per train: [compress scratch | slot ranks | slots]
    Q.arena_stride = per_train;
    Q.slot_doubles = slot_doubles;
    Q.cap = (int)work_cap;
    Q.srk = (long long*)(base + per_compress);
    Q.srk_stride = per_train;                           // in units of 8 bytes, like the doubles
    if (kind == 1) { Q.x = x->dev(); Q.y = y->dev(); Q.z = z->dev(); }
    hipLaunchKernelGGL(k_swap_chain, dim3(batch), dim3(TTN_WG), COMPRESS_LDS_BYTES, g_stream, Q);
    HIPCHK(hipGetLastError());
    return TTN_OK;
}

int ttn_hadamard_ttm(ttn_tt_t x, ttn_tt_t y, ttn_tt_t z, double tol, int64_t rmax, int64_t work_cap) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!x || !y || !z) return fail(TTN_ERR_ARG, "null handle");
    if (!same_dims(x->dims, y->dims) || !same_dims(x->dims, z->dims)) return fail(TTN_ERR_DIMS, "Incompatible TT dimensions");
    if (x->batch != y->batch || x->batch != z->batch) return fail(TTN_ERR_DIMS, "batch sizes differ");
    if (z == x || z == y) return fail(TTN_ERR_ARG, "ttn_hadamard_ttm: output must not alias an input");
    if (rmax < 1 || work_cap < 1 || tol < 0.0) return fail(TTN_ERR_ARG, "bad tol / rmax / work_cap");
    const int d = x->d;
    const int64_t n = x->dims[0];
    for (int k = 0; k < d; ++k) if (x->dims[k] != n) return fail(TTN_ERR_UNSUPPORTED, "ttn_hadamard_ttm: all physical dimensions must be equal");
    if (2 * d > 2 * TTN_MAX_D) return fail(TTN_ERR_UNSUPPORTED, "chain too long");
    for (int m = 0; m <= d; ++m)
        if (x->bound[m] > work_cap || y->bound[m] > work_cap) return fail(TTN_ERR_CAPACITY, "ttn_hadamard_ttm: work_cap below an input rank");
    // the reference's loops (tt_operations.jl:414-420) as ops on fixed slots.  With L0 = 2d (iter 1) or d-iter+2 (later) the
    // logical core l (1-based) lives in slot l (l <= L0) or l - L0 + d + 1: the contraction of iteration i frees slot d+2-i.
    std::vector<int> ops, fin;
    auto slot = [&](int iter, int l) { const int L0 = (iter == 1) ? 2 * d : d - iter + 2; return (l <= L0 ? l : l - L0 + d + 1) - 1; };
    for (int iter = 1; iter <= d; ++iter) {
        for (int j = d; j >= d - iter + 2; --j) { ops.push_back(0); ops.push_back(slot(iter, j)); ops.push_back(slot(iter, j + 1)); }
        const int pc = d - iter + 1;
        ops.push_back(1); ops.push_back(slot(iter, pc)); ops.push_back(slot(iter, pc + 1));
    }
    for (int l = 1; l <= d; ++l) fin.push_back(slot(d + 1, l));
    int rc = launch_chain(1, x, y, z, (int)n, 2 * d, work_cap, ops, fin, tol, rmax, 0);
    if (rc) return rc;
    for (int m = 0; m <= d; ++m) z->bound[m] = std::min<int64_t>(z->cap[m], std::min<int64_t>(work_cap, rmax));
    z->bound[0] = 1; z->bound[d] = 1;
    std::fill(z->ot.begin(), z->ot.end(), 0);
    return TTN_OK;
}

int ttn_swap_sites(ttn_tt_t x, int64_t nswaps, const int64_t* swaps, double threshold) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!x || nswaps < 0 || (nswaps > 0 && !swaps) || threshold < 0.0) return fail(TTN_ERR_ARG, "bad argument");
    const int d = x->d;
    const int64_t n = x->dims[0];
    for (int k = 0; k < d; ++k) if (x->dims[k] != n) return fail(TTN_ERR_UNSUPPORTED, "ttn_swap_sites: all physical dimensions must be equal");
    int64_t capmax = 1;
    for (int m = 0; m <= d; ++m) capmax = std::max(capmax, x->cap[m]);
    std::vector<int> ops, fin;
    std::vector<int64_t> bnd = x->bound;
    for (int64_t i = 0; i < nswaps; ++i) {
        const int64_t k = swaps[i];
        if (k < 1 || k >= d) return fail(TTN_ERR_BOND_INDEX, "k must be in 1:(N-1)");
        // length(F.S) = min(n r_{k-1}, n r_{k+1}) is kept whole when threshold == 0 (qtt_tools.jl:681-685)
        const int64_t full = std::min(n * bnd[k - 1], n * bnd[k + 1]);
        if (threshold == 0.0 && full > x->cap[k]) return fail(TTN_ERR_CAPACITY, "ttn_swap_sites: a bond rank grows beyond the handle's capacity");
        bnd[k] = std::min(full, x->cap[k]);
        ops.push_back(0); ops.push_back((int)k - 1); ops.push_back((int)k);
    }
    if (nswaps == 0) return TTN_OK;
    int rc = launch_chain(2, nullptr, nullptr, x, (int)n, d, capmax, ops, fin, threshold, (int64_t)1 << 62, 1);
    if (rc) return rc;
    x->bound = bnd;
    std::fill(x->ot.begin(), x->ot.end(), 0);
    return TTN_OK;
}

// ---- ttv_decomp: dense tensors -> trains ----------------------------------------------------------------
int ttn_ttv_decomp(ttn_tt_t z, const double* tensors, int64_t index, double tol) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!z || !tensors) return fail(TTN_ERR_ARG, "null argument");
    const int d = z->d;
    if (index < 1 || index > d) return fail(TTN_ERR_ARG, "index must be in 1:d");
    long long total = 1, nmax = 1, nmin = 1LL << 40;
    for (int k = 0; k < d; ++k) {
        total *= z->dims[k];
        nmax = std::max<long long>(nmax, z->dims[k]);
        nmin = std::min<long long>(nmin, z->dims[k]);
        if (total > (1LL << 27)) return fail(TTN_ERR_UNSUPPORTED, "ttn_ttv_decomp: more than 2^27 entries per tensor");
    }
    // worst-case short / long sides of the unfoldings, with the ranks bounded by the handle's capacity
    std::vector<int64_t> bnd(d + 1, 1);
    long long pmax = 1, qmax = 1;
    {
        long long len = total, rl = 1;
        for (int i = 0; i < index - 1; ++i) {
            const long long a = rl * z->dims[i], bc = len / a;
            pmax = std::max(pmax, std::min(a, bc)); qmax = std::max(qmax, std::max(a, bc));
            rl = std::min<long long>(std::min(a, bc), z->cap[i + 1]);
            bnd[i + 1] = rl; len = rl * bc;
        }
        long long rr = 1;
        for (int i = d - 1; i > index - 1; --i) {
            const long long a = z->dims[i] * rr, rows = len / a;
            pmax = std::max(pmax, std::min(a, rows)); qmax = std::max(qmax, std::max(a, rows));
            rr = std::min<long long>(std::min(a, rows), z->cap[i]);
            bnd[i] = rr; len = rows * rr;
        }
    }
    if (pmax > 4096) return fail(TTN_ERR_UNSUPPORTED, "ttn_ttv_decomp: an unfolding has a short side above 4096 (lower the handle's rank capacity)");
    const long long per_scr = QR_NB * qmax + pmax * QR_NB + 2 * pmax * pmax + 4 * pmax + 64;
    const long long per_train = 3 * total + per_scr;
    const int batch = z->batch;
    int rc = ensure_scratch(sizeof(double) * ((size_t)per_train * batch + (size_t)total * batch));
    if (rc) return rc;
    rc = ensure_batch_bufs(batch);
    if (rc) return rc;
    double* base = (double*)g_scratch;
    double* d_in = base + (size_t)per_train * batch;
    HIPCHK(hipMemcpyAsync(d_in, tensors, sizeof(double) * (size_t)total * batch, hipMemcpyHostToDevice, g_stream));
    HsvdArgs H;
    memset(&H, 0, sizeof(H));
    CompressArgs& P = H.C;
    P.tt = z->dev();
    P.max_bond = (int64_t)1 << 62;
    P.sweeps = 1;
    P.scratch = base + 3 * total;                          // per train: [cur0 | cur1 | M2 | LQ / Jacobi scratch]
    P.scratch_stride = per_train;
    P.pmax = (int)pmax; P.qmax = (int)qmax;
    P.status = z->d_status;
    P.sweep_stats = z->d_status + batch;
    { const char* e = getenv("TTN_JTOL"); P.jtol_mult = e ? atof(e) : 1.0; }
    { const char* e = getenv("TTN_JNEG"); P.jneg_mult = e ? atof(e) : 1.0; }
    H.tensors = d_in;
    H.total = total;
    H.index = (int)index - 1;
    H.tol = tol;
    H.work = base;
    H.work_stride = per_train;
    hipLaunchKernelGGL(k_ttv_decomp, dim3(batch), dim3(TTN_WG), COMPRESS_LDS_BYTES, g_stream, H);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(g_stream));                // `tensors` is caller memory: do not keep it in flight
    z->bound = bnd;
    for (int b = 0; b < batch; ++b)
        for (int k = 0; k < d; ++k) z->ot[(size_t)b * d + k] = (k < index - 1) ? -1 : (k == index - 1 ? 0 : 1);     // tt_tools.jl:191-198
    return TTN_OK;
}

// ---- als_linsolve ------------------------------------------------------------------------------------------------
#define TTN_DENSE_LOCAL_MAX_ALS 2048
// The grid form of als_linsolve (csrc/ttn_als_grid.h): for every train in turn the half sweeps of src/solvers/als.jl:199-219 walked on
// the host — per site the assembly of K (grid), the blocked LU with partial pivoting panel by panel (panel: one workgroup; row
// interchanges + U12: grid; trailing MFMA update: grid), the back substitution, then the core move and environment update (phase 2 / 3
// of k_als_linsolve, one workgroup).  Everything is enqueued on the library stream; one flag word carries a singular pivot column.
static int als_grid_path(AlsArgs P, const std::vector<long long>& off, const std::vector<int64_t>& r, ttn_tto_t A, ttn_tt_t b, ttn_tt_t x, int sweep_count) {
    const int d = x->d, batch = x->batch;
    static bool attr = false;
    const size_t panel_lds = sizeof(double) * (128 * 128 + 64 + 64);
    if (!attr) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lu_panel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)panel_lds));
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lu_trail), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * GEMM_LDS_TOTAL)));
        attr = true;
    }
    double* scr = P.scratch;
    double* K = scr + P.offK;
    double* Pb = scr + P.offPb;
    int* piv = reinterpret_cast<int*>(scr + P.offPiv);
    static int* d_flag = nullptr;
    if (!d_flag) HIPCHK(hipMalloc((void**)&d_flag, sizeof(int)));
    const std::vector<int64_t>& R = A->rks;
    auto solve_site = [&](int i) -> int {
        const int n = (int)x->dims[i], rl = (int)r[i], rr = (int)r[i + 1];
        const int nr = n * rl, N = nr * rr;
        AlsAssembleArgs Q;
        Q.G = scr + off[i]; Q.Gb = scr + off[d + i]; Q.H = scr + off[2 * d + i]; Q.Hb = scr + off[3 * d + i];
        Q.K = K; Q.Pb = Pb; Q.nr = nr; Q.rr = rr; Q.Rr = (int)R[i + 1]; Q.br = (int)b->bound[i + 1];
        hipLaunchKernelGGL(k_als_assemble, dim3((N + ALS_ASM_ROWS - 1) / ALS_ASM_ROWS, (N + ALS_ASM_COLS - 1) / ALS_ASM_COLS), dim3(ALS_ASM_ROWS), 0, g_stream, Q);
        for (int k0 = 0; k0 < N; k0 += LU_NB) {
            const int w = std::min(LU_NB, N - k0);
            hipLaunchKernelGGL(k_lu_panel, dim3(1), dim3(TTN_WG), panel_lds, g_stream, K, N, k0, w, piv, d_flag);
            hipLaunchKernelGGL(k_lu_rows, dim3((N + 1 + 255) / 256), dim3(256), 0, g_stream, K, Pb, N, k0, w, (const int*)piv, (const int*)d_flag);
            const int m = N - k0 - w;
            if (m > 0) {
                const int nt = (m + LU_TILE - 1) / LU_TILE;
                hipLaunchKernelGGL(k_lu_trail, dim3(nt, nt), dim3(TTN_WG), sizeof(double) * GEMM_LDS_TOTAL, g_stream, K, Pb, N, k0, w, (const int*)d_flag);
            }
        }
        for (int kb = ((N - 1) / 32) * 32; kb >= 0; kb -= 32) {
            const int wb = std::min(32, N - kb);
            hipLaunchKernelGGL(k_lu_back_tri, dim3(1), dim3(64), 0, g_stream, (const double*)K, Pb, N, kb, wb, (const int*)d_flag);
            if (kb > 0) hipLaunchKernelGGL(k_lu_back_rows, dim3((kb + 255) / 256), dim3(256), 0, g_stream, (const double*)K, Pb, N, kb, wb, (const int*)d_flag);
        }
        HIPCHK(hipGetLastError());
        return TTN_OK;
    };
    auto phase = [&](int ph, int site, int tb) -> int {
        AlsArgs Q = P;
        Q.phase = ph; Q.site = site; Q.train0 = tb;
        hipLaunchKernelGGL(k_als_linsolve, dim3(1), dim3(TTN_WG), COMPRESS_LDS_BYTES, g_stream, Q);
        HIPCHK(hipGetLastError());
        return TTN_OK;
    };
    for (int tb = 0; tb < batch; ++tb) {
        HIPCHK(hipMemsetAsync(d_flag, 0, sizeof(int), g_stream));
        int rc = phase(1, 0, tb);
        if (rc) return rc;
        int nsweeps = 0;
        while (nsweeps < sweep_count) {
            ++nsweeps;
            for (int i = 0; i < d - 1; ++i) { if ((rc = solve_site(i))) return rc; if ((rc = phase(2, i, tb))) return rc; }
            if (nsweeps == sweep_count) break;
            ++nsweeps;
            for (int i = d - 1; i >= 1; --i) { if ((rc = solve_site(i))) return rc; if ((rc = phase(3, i, tb))) return rc; }
        }
        int h_flag = 0;
        HIPCHK(hipMemcpyAsync(&h_flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, g_stream));
        HIPCHK(hipStreamSynchronize(g_stream));
        if (h_flag) {                                   // LAPACK's SingularException: recorded on the handle like the one-workgroup form does
            const int three = 3;
            HIPCHK(hipMemcpyAsync(x->d_status + tb, &three, sizeof(int), hipMemcpyHostToDevice, g_stream));
            HIPCHK(hipStreamSynchronize(g_stream));
        }
    }
    return TTN_OK;
}

int ttn_als_linsolve(ttn_tto_t A, ttn_tt_t b, ttn_tt_t x0, ttn_tt_t x, int64_t sweep_count) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!A || !b || !x0 || !x) return fail(TTN_ERR_ARG, "null handle");
    if (sweep_count < 1) return fail(TTN_ERR_SWEEPS, "sweep_count must be >= 1");
    if (!same_dims(A->dims, b->dims) || !same_dims(b->dims, x0->dims) || !same_dims(x0->dims, x->dims)) return fail(TTN_ERR_DIMS, "Incompatible dimensions");
    if (b->batch != x0->batch || x->batch != x0->batch) return fail(TTN_ERR_DIMS, "batch sizes differ");
    const int d = x0->d;
    if (d < 2) return fail(TTN_ERR_UNSUPPORTED, "ttn_als_linsolve: needs at least two sites");
    // the solution keeps the ranks of the start train (als.jl:177); they must survive orthogonalize (als.jl:174) and allow
    // the thin QRs of the core moves
    std::vector<int64_t> capped(d + 1);
    ttn_r_and_d_to_rks(d, x0->dims.data(), d + 1, x0->bound.data(), 1024, capped.data());
    const std::vector<int64_t> r(x0->bound);
    for (int k = 0; k <= d; ++k) if (capped[k] != r[k]) return fail(TTN_ERR_UNSUPPORTED, "ttn_als_linsolve: the start ranks exceed what orthogonalize keeps");
    for (int i = 0; i < d; ++i)
        if (x0->dims[i] * r[i] < r[i + 1] || x0->dims[i] * r[i + 1] < r[i]) return fail(TTN_ERR_UNSUPPORTED, "ttn_als_linsolve: a core is too flat for the QR core moves");
    int rc = ttn_orthogonalize(x0, 1, x);
    if (rc) return rc;
    const std::vector<int64_t>& R = A->rks;
    const std::vector<int64_t>& rb = b->bound;
    std::vector<long long> off(4 * d);
    long long cur = 0, Nmax = 1, mmax = 1, rmax = 1, t1 = 1, t2 = 1;
    for (int i = 0; i < d; ++i) {
        const long long n = x0->dims[i], rl = r[i], rr = r[i + 1], Rl = R[i], Rr = R[i + 1], bl = rb[i], br = rb[i + 1];
        off[i] = cur; cur += n * rl * n * rl * Rr;
        off[d + i] = cur; cur += n * rl * br;
        off[2 * d + i] = cur; cur += Rr * rr * rr;
        off[3 * d + i] = cur; cur += rr * br;
        Nmax = std::max(Nmax, n * rl * rr);
        mmax = std::max(mmax, std::max(n * rl, n * rr));
        rmax = std::max(rmax, std::max(rl, rr));
        t1 = std::max(t1, std::max(std::max(Rr * rr * n * rl, n * rl * rr * std::max(Rl, Rr)), std::max(rr * n * bl, std::max(rr * br, rl * br))));
        t2 = std::max(t2, std::max(n * rr * Rl * rl, rr * rr * Rr));
    }
    // Local systems above 2048 unknowns (ranks above 32 for n = 2 — BASELINE config C5 names ranks up to 128: 32 768 unknowns, an 8.6 GB
    // matrix): the GRID form — assembly and LU on the whole chip, the host walks the half sweeps (als_grid_path below).
    const bool grid_path = Nmax > TTN_DENSE_LOCAL_MAX_ALS || (getenv("TTN_ALS_GRID") && atoi(getenv("TTN_ALS_GRID")) != 0);
    if (Nmax > 65536) return fail(TTN_ERR_UNSUPPORTED, "ttn_als_linsolve: local systems above 65 536 unknowns are not supported");
    AlsArgs P;
    memset(&P, 0, sizeof(P));
    P.offK = cur; cur += Nmax * Nmax;
    P.offPb = cur; cur += Nmax;
    P.offPiv = cur; cur += Nmax / 2 + 8;               // Nmax ints
    P.offT1 = cur; cur += t1;
    P.offT2 = cur; cur += t2;
    P.offTm = cur; cur += mmax * rmax;
    P.offQb = cur; cur += mmax * rmax;
    P.offRb = cur; cur += rmax * rmax;
    P.offVb = cur; cur += QR_NB * mmax;
    P.offWb = cur; cur += QR_NB * mmax;
    P.offTst = cur; cur += ((rmax + QR_NB - 1) / QR_NB) * QR_NB * QR_NB + 64;
    const long long per_train = cur;
    const int batch = x->batch;
    const int nslots = grid_path ? 1 : batch;          // the grid form works on one train at a time (K alone can be gigabytes)
    static std::vector<long long> h_off;               // outlives the async upload
    HIPCHK(hipStreamSynchronize(g_stream));
    rc = ensure_scratch(sizeof(double) * (size_t)per_train * nslots + sizeof(long long) * (size_t)(5 * d + 1) + 64);
    if (rc) return rc;
    rc = ensure_batch_bufs(batch);
    if (rc) return rc;
    double* base = (double*)g_scratch;
    long long* d_tab = (long long*)(base + (size_t)per_train * nslots);
    h_off = off;
    for (int k = 0; k <= d; ++k) h_off.push_back(r[k]);
    HIPCHK(hipMemcpyAsync(d_tab, h_off.data(), sizeof(long long) * h_off.size(), hipMemcpyHostToDevice, g_stream));
    P.A = A->dev(); P.b = b->dev(); P.x = x->dev();
    P.sweep_count = (int)sweep_count;
    P.scratch = base; P.scratch_stride = per_train;
    P.off = d_tab;
    P.rfix = d_tab + 4 * d;
    P.mmax = (int)mmax; P.rmax = (int)rmax;
    P.status = x->d_status;
    if (grid_path) {
        rc = als_grid_path(P, off, r, A, b, x, (int)sweep_count);
        if (rc) return rc;
    } else {
        hipLaunchKernelGGL(k_als_linsolve, dim3(batch), dim3(TTN_WG), COMPRESS_LDS_BYTES, g_stream, P);
        HIPCHK(hipGetLastError());
    }
    // orthogonality flags as the core moves leave them (als.jl:128-134, :112-118)
    for (int bb = 0; bb < batch; ++bb) {
        int64_t* ot = &x->ot[(size_t)bb * d];
        int64_t done = 0;
        while (done < sweep_count) {
            ++done;
            for (int i = 0; i + 1 < d; ++i) { ot[i] = -1; ot[i + 1] = 0; }
            if (done == sweep_count) break;
            ++done;
            for (int i = d - 1; i >= 1; --i) { ot[i] = 1; ot[i - 1] = 0; }
        }
    }
    return TTN_OK;
}

// ---- mals_linsolve -----------------------------------------------------------------------------------------------
// the two-site solvers: mode 0 = mals_linsolve, mode 1 = dmrg_linsolve (N = 2) with `plan` = the rank cap of every full sweep
// Local solver of the two-site systems (dmrg.jl:92-97): conjugate gradients on the matrix-free operator if `it_solver` or the system
// has more than `itslv_thresh` unknowns, dense LU otherwise.  The dense path holds K in memory and is limited to TTN_DENSE_LOCAL_MAX
// unknowns; larger systems always take the matrix-free path.
#define TTN_DENSE_LOCAL_MAX 2048
#define TTN_KRYLOVDIM_DEFAULT 30       // KrylovKit.KrylovDefaults.krylovdim
struct LocalSolver { int it_solver = 0; int64_t itslv_thresh = TTN_DENSE_LOCAL_MAX; int64_t maxiter = 200; double tol = 1.0e-8; };
static std::vector<int> g_cg_iters_host;       // total CG iterations per train of the last two-site solve (ttn_dmrg_cg_iterations)

static int two_site_linsolve(ttn_tto_t A, ttn_tt_t b, ttn_tt_t x0, ttn_tt_t x, double tol, int64_t rmax, int mode, const std::vector<int64_t>& plan,
                             const LocalSolver& ls = LocalSolver()) {
    NEED_INIT();
    if (!A || !b || !x0 || !x) return fail(TTN_ERR_ARG, "null handle");
    if (tol < 0.0 || rmax < 1) return fail(TTN_ERR_ARG, "bad tol / rmax");
    if (!same_dims(A->dims, b->dims) || !same_dims(b->dims, x0->dims) || !same_dims(x0->dims, x->dims)) return fail(TTN_ERR_DIMS, "Incompatible dimensions");
    if (b->batch != x0->batch || x->batch != x0->batch) return fail(TTN_ERR_DIMS, "batch sizes differ");
    const int d = x0->d;
    if (d < 2) return fail(TTN_ERR_UNSUPPORTED, "ttn_mals_linsolve: needs at least two sites");
    int rc = ttn_orthogonalize(x0, 1, x);                  // mals.jl:252 (checks x's capacity against the start ranks)
    if (rc) return rc;
    const std::vector<int64_t>& c = x->cap;
    const std::vector<int64_t>& R = A->rks;
    const std::vector<int64_t>& rb = b->bound;
    auto mx = [](long long a_, long long b_) { return a_ > b_ ? a_ : b_; };
    std::vector<long long> off(4 * d, 0);
    long long cur = 0, Nmax = 1, mmax = 1, t1 = 1, t2 = 1;
    for (int i = 0; i < d; ++i) {
        const long long n = x0->dims[i];
        off[i] = cur; cur += n * c[i] * n * c[i] * R[i + 1];
        off[d + i] = cur; cur += n * c[i] * rb[i + 1];
        mmax = mx(mmax, mx(n * c[i], n * c[i + 1]));
        t1 = mx(t1, n * c[i] * c[i + 1] * mx(R[i], R[i + 1]));
        t1 = mx(t1, mx(c[i + 1] * rb[i + 1], c[i] * rb[i + 1]));
        t2 = mx(t2, c[i + 1] * c[i + 1] * R[i + 1]);
        if (i + 1 < d) {
            const long long n2 = x0->dims[i + 1];
            off[2 * d + i] = cur; cur += R[i + 1] * n2 * n2 * c[i + 2] * c[i + 2];
            off[3 * d + i] = cur; cur += rb[i + 1] * n2 * c[i + 2];
            Nmax = mx(Nmax, n * c[i] * n2 * c[i + 2]);
            t1 = mx(t1, mx(R[i + 1] * n2 * c[i + 2] * c[i + 1], rb[i + 1] * c[i + 1]));
            t2 = mx(t2, R[i + 1] * c[i + 1] * c[i + 1]);
        }
    }
    // mals_linsolve has no iterative branch in the reference (Hermitian(K) \ b, mals.jl:148-157): dense only
    const long long dense_max = ls.it_solver ? 0 : std::min<long long>(TTN_DENSE_LOCAL_MAX, mode == 1 ? ls.itslv_thresh : TTN_DENSE_LOCAL_MAX);
    const bool need_cg = mode == 1 && (ls.it_solver || Nmax > dense_max);
    if (!need_cg && Nmax > TTN_DENSE_LOCAL_MAX) return fail(TTN_ERR_UNSUPPORTED, "ttn_mals_linsolve: two-site systems above 2048 unknowns (n_i cap_i n_{i+1} cap_{i+2}) are not supported; lower the capacity of x");
    if (mmax > 256) return fail(TTN_ERR_UNSUPPORTED, "two-site solvers: n_i * capacity above 256 (ranks above 128 for n = 2) is not supported by the SVD core moves");
    if (ls.maxiter < 1 || !(ls.tol >= 0.0)) return fail(TTN_ERR_ARG, "dmrg_linsolve: bad linsolv_maxiter / linsolv_tol");
    const long long pmax = std::min<long long>(mmax, 256), qmax = mmax;
    long long Rzmax = 1;
    for (int i = 0; i <= d; ++i) Rzmax = mx(Rzmax, R[i]);
    MalsArgs Q;
    memset(&Q, 0, sizeof(Q));
    AlsArgs& P = Q.L;
    const long long Kdim = std::min<long long>(Nmax, dense_max);
    P.offK = cur; cur += Kdim * Kdim;
    if (need_cg) { Q.offCg = cur; Q.cg_nmax = Nmax; cur += (4 + Rzmax) * Nmax; }
    Q.cg_all = ls.it_solver ? 1 : 0;
    Q.cg_above = (int)std::min<long long>(dense_max, (1LL << 30));
    // KrylovKit's linsolve selector builds CG(maxiter = krylovdim * maxiter) for isposdef problems, krylovdim = KrylovDefaults' 30
    // (the convention src/solvers/euler.jl:29 spells out; the call at src/solvers/dmrg.jl:170 passes maxiter = linsolv_maxiter only)
    Q.cg_maxiter = (int)std::min<int64_t>(TTN_KRYLOVDIM_DEFAULT * ls.maxiter, 1 << 30);
    Q.cg_tol = ls.tol;
    P.offPb = cur; cur += Nmax;
    P.offPiv = cur; cur += Nmax / 2 + 8;
    P.offT1 = cur; cur += t1;
    P.offT2 = cur; cur += t2;
    P.offVb = cur; cur += QR_NB * qmax;
    P.offWb = cur; cur += QR_NB * qmax;
    Q.offM2 = cur; cur += Nmax;
    Q.offXg = cur; cur += pmax * pmax;
    Q.offUs = cur; cur += pmax * pmax;
    Q.offSig = cur; cur += 4 * pmax + 64;
    const long long per_train = cur;
    const int batch = x->batch;
    static std::vector<long long> h_off;
    HIPCHK(hipStreamSynchronize(g_stream));
    rc = ensure_scratch(sizeof(double) * (size_t)per_train * batch + sizeof(long long) * (size_t)(4 * d) + 64);
    if (rc) return rc;
    rc = ensure_batch_bufs(batch);
    if (rc) return rc;
    double* base = (double*)g_scratch;
    long long* d_tab = (long long*)(base + (size_t)per_train * batch);
    h_off = off;
    HIPCHK(hipMemcpyAsync(d_tab, h_off.data(), sizeof(long long) * h_off.size(), hipMemcpyHostToDevice, g_stream));
    P.A = A->dev(); P.b = b->dev(); P.x = x->dev();
    P.scratch = base; P.scratch_stride = per_train;
    P.off = d_tab;
    P.status = x->d_status;
    Q.C.status = x->d_status;
    Q.C.sweep_stats = x->d_status + batch;
    Q.C.pmax = (int)pmax; Q.C.qmax = (int)qmax;
    { const char* e = getenv("TTN_JTOL"); Q.C.jtol_mult = e ? atof(e) : 1.0; }
    { const char* e = getenv("TTN_JNEG"); Q.C.jneg_mult = e ? atof(e) : 1.0; }
    Q.tol = tol;
    Q.rmax = (int)std::min<int64_t>(rmax, 1 << 30);
    Q.pmax = (int)pmax; Q.qmax = (int)qmax;
    Q.mode = mode;
    Q.nsweeps = (int)plan.size();
    Q.rmax_final = Q.rmax;
    int64_t rtop = rmax;
    for (size_t s_ = 0; s_ < plan.size(); ++s_) { Q.rmax_sweep[s_] = (int)std::min<int64_t>(plan[s_], 1 << 30); rtop = std::max(rtop, plan[s_]); }
    static int* d_cg_iters = nullptr; static int cg_cap = 0;
    if (need_cg) {
        if (cg_cap < batch) { if (d_cg_iters) hipFree(d_cg_iters); HIPCHK(hipMalloc((void**)&d_cg_iters, sizeof(int) * batch)); cg_cap = batch; }
        HIPCHK(hipMemsetAsync(d_cg_iters, 0, sizeof(int) * batch, g_stream));
        Q.cg_iters = d_cg_iters;
    }
    hipLaunchKernelGGL(k_mals_linsolve, dim3(batch), dim3(TTN_WG), COMPRESS_LDS_BYTES, g_stream, Q);
    HIPCHK(hipGetLastError());
    g_cg_iters_host.assign(batch, 0);
    if (need_cg) {
        HIPCHK(hipMemcpyAsync(g_cg_iters_host.data(), d_cg_iters, sizeof(int) * batch, hipMemcpyDeviceToHost, g_stream));
        HIPCHK(hipStreamSynchronize(g_stream));
    }
    for (int m = 1; m < d; ++m) x->bound[m] = std::min<int64_t>(x->cap[m], rtop);
    x->bound[0] = 1; x->bound[d] = 1;
    for (int bb = 0; bb < batch; ++bb)
        for (int k = 0; k < d; ++k)         // mals: after the backward half sweep (mals.jl:116-117); dmrg: left_core_move! (dmrg.jl:222-223, :440)
            x->ot[(size_t)bb * d + k] = (k == 0) ? 0 : (mode == 0 ? 1 : -1);
    return TTN_OK;
}

int ttn_mals_linsolve(ttn_tto_t A, ttn_tt_t b, ttn_tt_t x0, ttn_tt_t x, double tol, int64_t rmax) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    return two_site_linsolve(A, b, x0, x, tol, rmax, 0, {});
}

static int dmrg_linsolve_impl(ttn_tto_t A, ttn_tt_t b, ttn_tt_t x0, ttn_tt_t x, double tol, int64_t n_stages, const int64_t* sweep_schedule,
                              const int64_t* rmax_schedule, const LocalSolver& ls) {
    if (n_stages < 1 || !sweep_schedule || !rmax_schedule) return fail(TTN_ERR_ARG, "dmrg_linsolve: empty schedule");
    for (int64_t j = 0; j < n_stages; ++j) {
        // the reference's while-loop (dmrg.jl:421-426) only terminates for positive, strictly increasing stage ends
        if (sweep_schedule[j] < 1 || (j && sweep_schedule[j] <= sweep_schedule[j - 1])) return fail(TTN_ERR_ARG, "dmrg_linsolve: sweep_schedule must be positive and strictly increasing");
        if (rmax_schedule[j] < 1) return fail(TTN_ERR_ARG, "dmrg_linsolve: bad rmax_schedule");
    }
    std::vector<int64_t> plan;
    int64_t n = 0, j = 0;
    for (;;) {
        ++n;
        if (n == sweep_schedule[j]) { if (++j >= n_stages) break; }
        plan.push_back(rmax_schedule[j]);
        if ((int64_t)plan.size() > TTN_DMRG_MAX_SWEEPS) return fail(TTN_ERR_UNSUPPORTED, "dmrg_linsolve: more than 32 sweeps in one call");
    }
    return two_site_linsolve(A, b, x0, x, tol, rmax_schedule[n_stages - 1], 1, plan, ls);
}

int ttn_dmrg_linsolve(ttn_tto_t A, ttn_tt_t b, ttn_tt_t x0, ttn_tt_t x, double tol, int64_t n_stages, const int64_t* sweep_schedule,
                      const int64_t* rmax_schedule) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    LocalSolver ls;                      // dense LU up to TTN_DENSE_LOCAL_MAX unknowns, matrix-free CG (tol 1e-8) above
    ls.tol = std::max(std::sqrt(std::max(tol, 0.0)), 1.0e-8);       // the reference's default linsolv_tol (dmrg.jl:394)
    return dmrg_linsolve_impl(A, b, x0, x, tol, n_stages, sweep_schedule, rmax_schedule, ls);
}

int ttn_dmrg_linsolve_it(ttn_tto_t A, ttn_tt_t b, ttn_tt_t x0, ttn_tt_t x, double tol, int64_t n_stages, const int64_t* sweep_schedule,
                         const int64_t* rmax_schedule, int it_solver, int64_t linsolv_maxiter, double linsolv_tol, int64_t itslv_thresh) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (itslv_thresh < 0) return fail(TTN_ERR_ARG, "dmrg_linsolve: bad itslv_thresh");
    LocalSolver ls;
    ls.it_solver = it_solver ? 1 : 0; ls.itslv_thresh = itslv_thresh; ls.maxiter = linsolv_maxiter; ls.tol = linsolv_tol;
    return dmrg_linsolve_impl(A, b, x0, x, tol, n_stages, sweep_schedule, rmax_schedule, ls);
}

int ttn_dmrg_cg_iterations(int64_t batch, int64_t* iters) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!iters || batch < 0 || (size_t)batch > g_cg_iters_host.size()) return fail(TTN_ERR_ARG, "ttn_dmrg_cg_iterations: no two-site solve of that batch size has run");
    for (int64_t t = 0; t < batch; ++t) iters[t] = g_cg_iters_host[(size_t)t];
    return TTN_OK;
}

// Failure codes of every dense kernel that wrote `h` since the last call (synchronises).  The codes are sticky on the device —
// a kernel only ever stores a non-zero code, so a failure inside a chain of launches survives the launches after it — and
// are cleared here, on read.
static int check_status(ttn_tt_t h) {
    const int batch = h->batch;
    std::vector<int> st(batch);
    HIPCHK(hipMemcpyAsync(st.data(), h->d_status, sizeof(int) * batch, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipMemsetAsync(h->d_status, 0, sizeof(int) * batch, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    for (int b = 0; b < batch; ++b) if (st[b] == 3) return fail(TTN_ERR_SINGULAR, "als_linsolve: a local system K is singular");
    for (int b = 0; b < batch; ++b) if (st[b] == 4) return fail(TTN_ERR_DIMS, "als_linsolve: a train's ranks differ from the ranks of the start handle");
    for (int b = 0; b < batch; ++b) if (st[b] == 2) return fail(TTN_ERR_CAPACITY, "a rank grew beyond the rank capacity of its handle / working slot (site-swap chain or ttv_decomp)");
    for (int b = 0; b < batch; ++b) if (st[b]) return fail(TTN_ERR_NO_CONVERGENCE, "Jacobi SVD hit its sweep limit");
    return TTN_OK;
}

static int status_code_to_error(int st) {
    if (st == 3) return fail(TTN_ERR_SINGULAR, "als_linsolve: a local system K is singular");
    if (st == 4) return fail(TTN_ERR_DIMS, "als_linsolve: a train's ranks differ from the ranks of the start handle");
    if (st == 2) return fail(TTN_ERR_CAPACITY, "a rank grew beyond the rank capacity of its handle / working slot (site-swap chain or ttv_decomp)");
    if (st) return fail(TTN_ERR_NO_CONVERGENCE, "Jacobi SVD hit its sweep limit");
    return TTN_OK;
}

int ttn_status_all(void) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    int st = 0;
    for (ttn_tt_s* h : g_live)
        if (h->d_status) hipLaunchKernelGGL(k_fold_status, dim3(1), dim3(64), 0, g_stream, (const int*)h->d_status, h->batch, g_pending_status);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&st, g_pending_status, sizeof(int), hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipMemsetAsync(g_pending_status, 0, sizeof(int), g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    return status_code_to_error(st);
}

int ttn_compress_status(ttn_tt_t psi, int64_t* total_jacobi_sweeps) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!psi) return fail(TTN_ERR_ARG, "null handle");
    if (total_jacobi_sweeps) {
        std::vector<int> st(psi->batch);
        HIPCHK(hipMemcpyAsync(st.data(), psi->d_status + psi->batch, sizeof(int) * psi->batch, hipMemcpyDeviceToHost, g_stream));
        HIPCHK(hipStreamSynchronize(g_stream));
        for (int b = 0; b < psi->batch; ++b) total_jacobi_sweeps[b] = st[b];
    }
    return check_status(psi);
}

// diagnostic: per-phase cycle counters (100 MHz s_memtime ticks) of train b from the last TTN_PROF=1 compress launch
int ttn_prof_get(int64_t b, int64_t* out8) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!g_prof || !out8) return fail(TTN_ERR_ARG, "no profile (set TTN_PROF=1)");
    long long tmp[16];
    HIPCHK(hipMemcpyAsync(tmp, g_prof + 16 * b, sizeof(tmp), hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    for (int i = 0; i < 16; ++i) out8[i] = tmp[i];
    return TTN_OK;
}
int ttn_prof_steps(int64_t b, int64_t* out120) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!g_prof || !out120) return fail(TTN_ERR_ARG, "no profile (set TTN_PROF=1)");
    long long tmp[120];
    HIPCHK(hipMemcpyAsync(tmp, g_prof + 16LL * g_prof_batch + 120 * b, sizeof(tmp), hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    for (int i = 0; i < 120; ++i) out120[i] = tmp[i];
    return TTN_OK;
}

int ttn_prof_fine(int64_t b, int64_t* out64) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!g_prof || !out64) return fail(TTN_ERR_ARG, "no profile (set TTN_PROF=1)");
    long long tmp[64];
    HIPCHK(hipMemcpyAsync(tmp, g_prof + 136LL * g_prof_batch + 64 * b, sizeof(tmp), hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    for (int i = 0; i < 64; ++i) out64[i] = tmp[i];
    return TTN_OK;
}

int ttn_dot(ttn_tt_t a, ttn_tt_t b, double* out) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!a || !b || !out) return fail(TTN_ERR_ARG, "null pointer");
    if (!same_dims(a->dims, b->dims)) return fail(TTN_ERR_DIMS, "TT dimensions are not compatible");
    if (a->batch != b->batch) return fail(TTN_ERR_DIMS, "batch sizes differ");
    const int d = a->d;
    if (d > DOT_MAX_D) return fail(TTN_ERR_UNSUPPORTED, "ttn_dot: chains longer than 480 sites are not supported");
    if (a->stride >= (1LL << 31) || b->stride >= (1LL << 31)) return fail(TTN_ERR_UNSUPPORTED, "ttn_dot: a train of 2^31 doubles or more");
    long long ramax = 1, rbmax = 1, nmax = 1;
    for (int m = 0; m <= d; ++m) { ramax = std::max<long long>(ramax, a->bound[m]); rbmax = std::max<long long>(rbmax, b->bound[m]); }
    for (int k = 0; k < d; ++k) nmax = std::max<long long>(nmax, a->dims[k]);
    const long long per_train = (2 + nmax) * ramax * rbmax + 16;
    int rc = ensure_scratch(sizeof(double) * (size_t)per_train * a->batch);
    if (rc) return rc;
    rc = ensure_batch_bufs(a->batch);
    if (rc) return rc;
    DotArgs P;
    P.a = a->dev(); P.b = b->dev();
    P.scratch = (double*)g_scratch; P.scratch_stride = per_train;
    P.ramax = (int)ramax; P.rbmax = (int)rbmax; P.nmax = (int)nmax;
    P.out = g_dout;
    P.prof = nullptr;
    if (getenv("TTN_PROF")) { int rcp = ensure_prof(a->batch); if (rcp) return rcp; P.prof = g_prof; }
    HIPCHK(hipEventRecord(g_ev0, g_stream));
    hipLaunchKernelGGL(k_dot_fused, dim3(a->batch), dim3(TTN_WG), DOT_LDS_BYTES(d), g_stream, P);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(g_ev1, g_stream));              // ttn_last_launch_ms: the kernel alone (this call goes on to copy and synchronise)
    g_have_launch_ms = true;
    HIPCHK(hipMemcpyAsync(out, g_dout, sizeof(double) * a->batch, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    return TTN_OK;
}

// diagnostics: the per-train state words of the last three-launch orthogonalize (next site, buffers, sites taken by k_ortho512)
static int* g_ortho_state = nullptr;
extern "C" int ttn_debug_ortho_state(int64_t b, int64_t* out4) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!g_ortho_state || !out4) return fail(TTN_ERR_ARG, "no three-launch orthogonalize yet");
    int tmp[4];
    HIPCHK(hipMemcpyAsync(tmp, g_ortho_state + 4 * b, sizeof(tmp), hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    for (int i = 0; i < 4; ++i) out4[i] = tmp[i];
    return TTN_OK;
}

int ttn_last_launch_ms(float* ms) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!ms) return fail(TTN_ERR_ARG, "null pointer");
    if (!g_have_launch_ms) return fail(TTN_ERR_ARG, "ttn_last_launch_ms: no ttn_dot / ttn_orthogonalize launch yet");
    HIPCHK(hipEventSynchronize(g_ev1));
    HIPCHK(hipEventElapsedTime(ms, g_ev0, g_ev1));
    return TTN_OK;
}

int ttn_norm(ttn_tt_t a, double* out) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    int rc = ttn_dot(a, a, out);
    if (rc) return rc;
    for (int b = 0; b < a->batch; ++b) { double v = out[b]; v = v < 0 ? 0.0 : v; out[b] = std::sqrt(v); }
    return TTN_OK;
}

int ttn_orthogonalize(ttn_tt_t x, int64_t center, ttn_tt_t y) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!x || !y) return fail(TTN_ERR_ARG, "null handle");
    if (!same_dims(x->dims, y->dims) || x->batch != y->batch) return fail(TTN_ERR_DIMS, "Incompatible dimensions");
    if (x == y) return fail(TTN_ERR_ARG, "ttn_orthogonalize: output must not alias the input");
    const int d = x->d;
    if (center < 1 || center > d) return fail(TTN_ERR_CENTER, "Impossible orthogonalization");
    // y ranks start from r_and_d_to_rks(x.rks, dims) but a left QR step sets r_{j+1} = min(rows, cols), which can
    // exceed that cap (the reference reassigns the rank, tt_tools.jl:522); they never exceed x's own ranks.
    std::vector<int64_t> yb(x->bound);
    for (int m = 0; m <= d; ++m) if (y->cap[m] < yb[m]) return fail(TTN_ERR_CAPACITY, "ttn_orthogonalize: destination capacity too small");
    long long rmax = 1, nmax = 1;
    for (int m = 0; m <= d; ++m) rmax = std::max<long long>(rmax, x->bound[m]);
    for (int k = 0; k < d; ++k) nmax = std::max<long long>(nmax, x->dims[k]);
    const long long mm = nmax * rmax;           // rows of the tall matrices
    const long long per_train = 2 * mm * rmax + 4 * rmax * rmax + 2 * QR_NB * mm + ((rmax + QR_NB - 1) / QR_NB) * QR_NB * QR_NB + 64   // Tm, Qb, 4 R, Vb, Wb, T panels
                                + 3 * 128 * 128;                                                                                  // Gram matrices / L1 of the Cholesky-QR steps
    int rc = ensure_scratch(sizeof(double) * (size_t)per_train * x->batch + sizeof(int) * (5 * (size_t)x->batch + 16) + 64);
    if (rc) return rc;
    rc = ensure_batch_bufs(x->batch);
    if (rc) return rc;
    OrthoArgs P;
    P.x = x->dev(); P.y = y->dev();
    P.center = (int)center - 1;
    P.scratch = (double*)g_scratch; P.scratch_stride = per_train;
    P.mmax = (int)mm; P.rmax = (int)rmax;
    { const char* e = getenv("TTN_ORTHO_CHOLQR"); P.no_cholqr = e ? (atoi(e) == 0 ? 3 : (atoi(e) == 1 ? 2 : 0)) : 0; }
    P.prof = nullptr;
    if (getenv("TTN_PROF")) { int rcp = ensure_prof(x->batch); if (rcp) return rcp; P.prof = g_prof; }
    // Rank <= 64 QTT trains: the ramp sites at the right end by one wave per train (csrc/ttn_ortho_ramp.h), the tall sites and the
    // centre core by the 512-thread kernel (two workgroups per CU, csrc/ttn_ortho512.h), the 1024-thread kernel before them for the
    // left sweep and after them only for the trains they did not finish.  Measured against the single launch (d = 30, rank 64, centre
    // 1): 1.24 vs 1.28 ms for one train, 1.28 vs 1.48 at 8, 1.41 vs 1.89 at 256, 3.36 vs 6.89 at 1024.  TTN_ORTHO512 = 0 / 1 forbids /
    // forces it.
    bool use512 = nmax == 2 && rmax <= 64 && d <= TTN_MAX_D * 8 && center < d;
    for (int k = 0; k < d; ++k) use512 = use512 && x->dims[k] == 2;
    { const char* e = getenv("TTN_ORTHO512"); if (e) use512 = atoi(e) != 0 && nmax == 2 && rmax <= 64 && d <= TTN_MAX_D * 8; }
    P.mode = 0; P.trains = nullptr; P.ramp = 0;
    P.state = reinterpret_cast<int*>((double*)g_scratch + (size_t)per_train * x->batch);
    g_ortho_state = P.state;
    HIPCHK(hipEventRecord(g_ev0, g_stream));
    if (use512) {
        P.mode = 1;
        int* left = P.state + 4 * (size_t)x->batch;                            // count, then the list of trains k_ortho512 did not finish
        HIPCHK(hipMemsetAsync(left, 0, sizeof(int), g_stream));
        // the ramp sites at the right end (wide / square LQ steps) go to one wave per train (csrc/ttn_ortho_ramp.h); the 1024-thread
        // kernel then only runs the left sweep, and not at all when the centre is the first site.  TTN_ORTHO_RAMP = 0: without.
        { const char* e = getenv("TTN_ORTHO_RAMP"); P.ramp = e ? (atoi(e) != 0) : 1; }
        if (!P.ramp || P.center > 0) hipLaunchKernelGGL(k_orthogonalize, dim3(x->batch), dim3(TTN_WG), ORTHO_LDS_BYTES, g_stream, P);
        if (P.ramp) {
            P.mode = P.center > 0 ? 4 : 5;
            hipLaunchKernelGGL(k_ortho_ramp, dim3((x->batch + ORAMP_WG / 64 - 1) / (ORAMP_WG / 64)), dim3(ORAMP_WG), 0, g_stream, P, (int)x->batch);
        }
        hipLaunchKernelGGL(k_ortho512, dim3(x->batch), dim3(O5_WG), O5_LDS_BYTES(d), g_stream, P);
        // the 512-thread kernel finishes a train (centre core included) unless it had to stop — a refused step, a site outside its
        // class: the third launch takes only those trains (1024 heavy workgroups cost 4 ms of dispatch even when they do nothing)
        int h_left = 0;
        HIPCHK(hipMemcpyAsync(&h_left, left, sizeof(int), hipMemcpyDeviceToHost, g_stream));
        HIPCHK(hipStreamSynchronize(g_stream));
        if (h_left > 0) {
            P.mode = 3;
            P.trains = left + 1;
            hipLaunchKernelGGL(k_orthogonalize, dim3(h_left), dim3(TTN_WG), ORTHO_LDS_BYTES, g_stream, P);
        }
    } else {
        hipLaunchKernelGGL(k_orthogonalize, dim3(x->batch), dim3(TTN_WG), ORTHO_LDS_BYTES, g_stream, P);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(g_ev1, g_stream));
    g_have_launch_ms = true;
    y->bound = yb;
    for (int b = 0; b < y->batch; ++b)
        for (int k = 0; k < d; ++k) y->ot[(size_t)b * d + k] = (k < center - 1) ? 1 : (k > center - 1 ? -1 : 0);
    return TTN_OK;
}

// kernel unit-test hook: C (m x n, row-major, host) = alpha * op(A) * op(B) + beta * C through wg_gemm on the device
int ttn_selftest_gemm(int64_t m, int64_t n, int64_t k, const double* A, const double* B, double* C, double alpha, double beta,
                      int ta, int tb) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!A || !B || !C || m < 1 || n < 1 || k < 1) return fail(TTN_ERR_ARG, "bad argument");
    double *dA = nullptr, *dB = nullptr, *dC = nullptr;
    HIPCHK(hipMalloc((void**)&dA, sizeof(double) * m * k));
    HIPCHK(hipMalloc((void**)&dB, sizeof(double) * k * n));
    HIPCHK(hipMalloc((void**)&dC, sizeof(double) * m * n));
    HIPCHK(hipMemcpyAsync(dA, A, sizeof(double) * m * k, hipMemcpyHostToDevice, g_stream));
    HIPCHK(hipMemcpyAsync(dB, B, sizeof(double) * k * n, hipMemcpyHostToDevice, g_stream));
    HIPCHK(hipMemcpyAsync(dC, C, sizeof(double) * m * n, hipMemcpyHostToDevice, g_stream));
    if (getenv("TTN_WG512_SELFTEST") && atoi(getenv("TTN_WG512_SELFTEST"))) {       // the same test against the 512-thread build
        const int rc512 = ttn_wg512_selftest_gemm((int)m, (int)n, (int)k, dA, dB, dC, alpha, beta, ta, tb, g_stream);
        if (rc512) return hipfail((hipError_t)rc512, "k_selftest_gemm (512-thread build)");
    } else {
    hipLaunchKernelGGL(k_selftest_gemm, dim3(1), dim3(TTN_WG), sizeof(double) * GEMM_LDS_TOTAL, g_stream, (int)m, (int)n, (int)k,
                       dA, dB, dC, alpha, beta, ta, tb);
    HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(C, dC, sizeof(double) * m * n, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    hipFree(dA); hipFree(dB); hipFree(dC);
    return TTN_OK;
}

// self-test of the 128 x 128 symmetric eigensolver of the Gram route: G (host, column-major) -> sig[nev] = sqrt(eigenvalues)
// descending, X[128 * r] = sqrt(lam_j) u_j, cycles (s_memtime ticks) and the return code of the device routine
int ttn_selftest_eig128(const double* G, int64_t n, int64_t r, int64_t nev, double* sig, double* X, int64_t* ticks_rc) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!G || !sig || !X || (n != 64 && n != 128) || r < 1 || r > 64 || nev < r || nev > n) return fail(TTN_ERR_ARG, "bad argument");
    double *dG = nullptr, *dV = nullptr, *dS = nullptr, *dX = nullptr;
    long long* dC = nullptr;
    HIPCHK(hipMalloc((void**)&dG, sizeof(double) * 128 * 128));
    HIPCHK(hipMalloc((void**)&dV, sizeof(double) * 128 * 128));
    HIPCHK(hipMalloc((void**)&dS, sizeof(double) * 128));
    HIPCHK(hipMalloc((void**)&dX, sizeof(double) * 128 * 64));
    HIPCHK(hipMalloc((void**)&dC, sizeof(long long) * 16));
    HIPCHK(hipMemcpyAsync(dG, G, sizeof(double) * n * n, hipMemcpyHostToDevice, g_stream));
    HIPCHK(hipMemsetAsync(dX, 0, sizeof(double) * 128 * 64, g_stream));
    if (getenv("TTN_WG512_SELFTEST") && atoi(getenv("TTN_WG512_SELFTEST"))) {
        const int rc512 = ttn_wg512_selftest_eig(dG, dV, (int)n, (int)r, (int)nev, dS, dX, dC, g_stream);
        if (rc512) return hipfail((hipError_t)rc512, "k_selftest_eig128 (512-thread build)");
    } else {
    hipLaunchKernelGGL(k_selftest_eig128, dim3(1), dim3(TTN_WG), COMPRESS_LDS_BYTES, g_stream, dG, dV, (int)n, (int)r, (int)nev, dS, dX, dC);
    HIPCHK(hipGetLastError());
    }
    long long hc[16] = {0};
    HIPCHK(hipMemcpyAsync(sig, dS, sizeof(double) * nev, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipMemcpyAsync(X, dX, sizeof(double) * 128 * r, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipMemcpyAsync(hc, dC, sizeof(long long) * 16, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    if (ticks_rc) { ticks_rc[0] = hc[0]; ticks_rc[1] = hc[1]; for (int t = 2; t < 6; ++t) ticks_rc[t] = hc[t + 1] - hc[t]; }   // tridiag, bisect, twisted, back-transform
    hipFree(dG); hipFree(dV); hipFree(dS); hipFree(dX); hipFree(dC);
    return TTN_OK;
}

int ttn_bench_gemm(int64_t m, int64_t n, int64_t k, int ta, int tb, int64_t reps, int64_t* cycles_out) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!cycles_out || m < 1 || n < 1 || k < 1 || reps < 1) return fail(TTN_ERR_ARG, "bad argument");
    double *dA = nullptr, *dB = nullptr, *dC = nullptr;
    long long* dcy = nullptr;
    // TTN_BENCH_GRID workgroups at once, each on its own operands (default 1: one CU busy); TTN_WG512_SELFTEST: the 512-thread build
    const int grid = getenv("TTN_BENCH_GRID") ? std::max(1, atoi(getenv("TTN_BENCH_GRID"))) : 1;
    HIPCHK(hipMalloc((void**)&dA, sizeof(double) * m * k * grid));
    HIPCHK(hipMalloc((void**)&dB, sizeof(double) * k * n * grid));
    HIPCHK(hipMalloc((void**)&dC, sizeof(double) * m * n * grid));
    HIPCHK(hipMalloc((void**)&dcy, sizeof(long long)));
    HIPCHK(hipMemsetAsync(dA, 0, sizeof(double) * m * k * grid, g_stream));
    HIPCHK(hipMemsetAsync(dB, 0, sizeof(double) * k * n * grid, g_stream));
    HIPCHK(hipMemsetAsync(dC, 0, sizeof(double) * m * n * grid, g_stream));
    if (getenv("TTN_WG512_SELFTEST") && atoi(getenv("TTN_WG512_SELFTEST"))) {
        const int rc512 = ttn_wg512_bench_gemm((int)m, (int)n, (int)k, dA, dB, dC, ta, tb, (int)reps, dcy, grid, g_stream);
        if (rc512) return hipfail((hipError_t)rc512, "k_bench_gemm (512-thread build)");
    } else {
    hipLaunchKernelGGL(k_bench_gemm, dim3(grid), dim3(TTN_WG), sizeof(double) * GEMM_LDS_TOTAL, g_stream, (int)m, (int)n, (int)k,
                       dA, dB, dC, ta, tb, (int)reps, dcy);
    HIPCHK(hipGetLastError());
    }
    long long cy = 0;
    HIPCHK(hipMemcpyAsync(&cy, dcy, sizeof(long long), hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    *cycles_out = cy;
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dcy);
    return TTN_OK;
}

int ttn_bench_lds(int what, int64_t n, int64_t reps, int64_t* cycles_out, int64_t* sweeps_out) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!cycles_out || n < 1 || n > 128 || reps < 1 || what < 0 || what > 2) return fail(TTN_ERR_ARG, "bad argument");
    long long* dout = nullptr;
    HIPCHK(hipMalloc((void**)&dout, 2 * sizeof(long long)));
    hipLaunchKernelGGL(k_bench_lds, dim3(1), dim3(TTN_WG), COMPRESS_LDS_BYTES, g_stream, what, (int)n, (int)reps, dout);
    HIPCHK(hipGetLastError());
    long long h[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(h, dout, sizeof(h), hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    *cycles_out = h[0];
    if (sweeps_out) *sweeps_out = h[1];
    hipFree(dout);
    return TTN_OK;
}

// ---- singular-value capture -----------------------------------------------------------------------
int ttn_sv_capture(ttn_tt_t h, int enable) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!h) return fail(TTN_ERR_ARG, "null handle");
    h->sv_on = enable != 0;
    return TTN_OK;
}
int ttn_sv_get(ttn_tt_t h, int64_t b, int64_t step, double* out, int64_t cap, int64_t* n) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!h || !out || !n || b < 0 || b >= h->batch) return fail(TTN_ERR_ARG, "bad argument");
    if (!h->d_sv || step < 0 || step >= h->sv_steps) return fail(TTN_ERR_ARG, "no captured singular values for that step");
    std::vector<double> tmp(h->sv_pmax);
    HIPCHK(hipMemcpyAsync(tmp.data(), h->d_sv + ((size_t)b * h->sv_steps + step) * h->sv_pmax, sizeof(double) * h->sv_pmax, hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    int64_t cnt = 0;
    for (int i = 0; i < h->sv_pmax && tmp[i] >= 0.0; ++i) { if (cnt < cap) out[cnt] = tmp[i]; ++cnt; }
    *n = std::min(cnt, cap);
    return TTN_OK;
}

// ---- stateless host-pointer entry points ----------------------------------------------------------
// ---- TDVP local contractions (src/solvers/tdvp.jl:29-43, :205-208), batched, real or complex -------------------------------------
// Device-pointer forms: every tensor is an array of `batch` column-major tensors laid out back to back (stride = its size; M may be
// shared by the batch: m_shared != 0).  The host forms (…_f64) stage host arrays of the same layout through the device.
static int tdvp_launch(int op, int cplx, int64_t batch, int64_t Dl, int64_t d, int64_t Dr, int64_t a, int64_t b, int64_t c, int64_t d2,
                       const double* FL, const double* FR, const double* X, const double* M1, const double* M2, double* out, int m_shared,
                       bool host) {
    NEED_INIT();
    if (batch < 1 || Dl < 1 || d < 1 || Dr < 1 || a < 1 || b < 1 || c < 1 || d2 < 1 || !X || !out) return fail(TTN_ERR_ARG, "tdvp: bad argument");
    const int64_t lim = 1 << 20;
    if (Dl > 4096 || Dr > 4096 || d > 64 || d2 > 64 || a > 64 || b > 64 || c > 64 || Dl * Dr * d * d2 * std::max({a, b, c}) > (lim << 6))
        return fail(TTN_ERR_UNSUPPORTED, "tdvp: tensor too large");
    const long long es = cplx ? 2 : 1;
    long long nFL = 0, nFR = 0, nX = 0, nM1 = 0, nM2 = 0, nOut = 0, w1 = 0, w2 = 0;
    switch (op) {
    case 0: nFL = Dl * a * Dl; nFR = Dr * b * Dr; nX = Dl * d * Dr; nM1 = a * d * b * d; nOut = Dl * d * Dr; w1 = Dl * a * d * Dr; w2 = Dl * d * Dr * b; break;
    case 1: nFL = Dl * a * Dl; nFR = Dr * a * Dr; nX = Dl * Dr; nOut = Dl * Dr; w1 = Dl * a * Dr; w2 = 1; break;
    case 2: nFL = Dl * a * Dl; nX = Dl * d * Dr; nM1 = a * d * b * d; nOut = Dr * b * Dr; w1 = Dl * a * d * Dr; w2 = Dl * d * b * Dr; break;
    case 3: nFR = Dr * a * Dr; nX = Dl * d * Dr; nM1 = b * d * a * d; nOut = Dl * b * Dl; w1 = Dl * d * a * Dr; w2 = Dl * b * d * Dr; break;
    case 4: nFL = Dl * a * Dl; nFR = Dr * c * Dr; nX = Dl * d * d2 * Dr; nM1 = a * d * b * d; nM2 = b * d2 * c * d2; nOut = Dl * d * d2 * Dr;
            w1 = Dl * d * d2 * Dr * std::max(a, c); w2 = Dl * d * b * d2 * Dr; break;
    default: return fail(TTN_ERR_ARG, "tdvp: unknown contraction");
    }
    if ((nFL && !FL) || (nFR && !FR) || (nM1 && !M1) || (nM2 && !M2)) return fail(TTN_ERR_ARG, "tdvp: null tensor");
    const long long mb = m_shared ? 1 : batch;
    const size_t work_d = (size_t)(w1 + w2) * es * batch;
    const size_t stage_d = host ? (size_t)es * ((nFL + nFR + nX + nOut) * batch + (nM1 + nM2) * mb) : 0;
    int rc = ensure_scratch(sizeof(double) * (work_d + stage_d));
    if (rc) return rc;
    double* base = (double*)g_scratch;
    TdvpArgs P;
    memset(&P, 0, sizeof(P));
    P.op = op; P.cplx = cplx;
    P.Dl = (int)Dl; P.d = (int)d; P.Dr = (int)Dr; P.a = (int)a; P.b = (int)b; P.c = (int)c; P.d2 = (int)d2;
    P.work = base; P.sWork = w1 + w2; P.w2off = w1;
    P.sFL = nFL; P.sFR = nFR; P.sX = nX; P.sOut = nOut; P.sM1 = m_shared ? 0 : nM1; P.sM2 = m_shared ? 0 : nM2;
    double* dout = out;
    if (host) {
        double* q = base + work_d;
        auto stage = [&](const double* src, long long n, long long cnt) -> const double* {
            if (!n) return nullptr;
            double* dst = q; q += (size_t)n * es * cnt;
            hipMemcpyAsync(dst, src, sizeof(double) * (size_t)n * es * cnt, hipMemcpyHostToDevice, g_stream);
            return dst;
        };
        P.FL = stage(FL, nFL, batch); P.FR = stage(FR, nFR, batch); P.X = stage(X, nX, batch);
        P.M1 = stage(M1, nM1, mb); P.M2 = stage(M2, nM2, mb);
        dout = q;
    } else { P.FL = FL; P.FR = FR; P.X = X; P.M1 = M1; P.M2 = M2; }
    P.out = dout;
    hipLaunchKernelGGL(k_tdvp, dim3((unsigned)batch), dim3(TTN_WG), TDVP_LDS_BYTES, g_stream, P);
    HIPCHK(hipGetLastError());
    if (host) {
        HIPCHK(hipMemcpyAsync(out, dout, sizeof(double) * (size_t)nOut * es * batch, hipMemcpyDeviceToHost, g_stream));
        HIPCHK(hipStreamSynchronize(g_stream));
    }
    return TTN_OK;
}

int ttn_tdvp_apply_h1(int cplx, int64_t batch, int64_t Dl, int64_t d, int64_t Dr, int64_t a, int64_t b, const double* FL, const double* AC,
                      const double* M, const double* FR, double* HAC, int m_shared) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    return tdvp_launch(0, cplx, batch, Dl, d, Dr, a, b, 1, 1, FL, FR, AC, M, nullptr, HAC, m_shared, false);
}
int ttn_tdvp_apply_h0(int cplx, int64_t batch, int64_t Dl, int64_t Dr, int64_t a, const double* FL, const double* C, const double* FR, double* HC) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    return tdvp_launch(1, cplx, batch, Dl, 1, Dr, a, 1, 1, 1, FL, FR, C, nullptr, nullptr, HC, 0, false);
}
int ttn_tdvp_update_left_env(int cplx, int64_t batch, int64_t Dl, int64_t d, int64_t Dr, int64_t a_in, int64_t a_out, const double* A, const double* M,
                             const double* FL, double* FLnext, int m_shared) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    return tdvp_launch(2, cplx, batch, Dl, d, Dr, a_in, a_out, 1, 1, FL, nullptr, A, M, nullptr, FLnext, m_shared, false);
}
int ttn_tdvp_update_right_env(int cplx, int64_t batch, int64_t Dl, int64_t d, int64_t Dr, int64_t a_out, int64_t a_in, const double* A, const double* M,
                              const double* FR, double* FRprev, int m_shared) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    return tdvp_launch(3, cplx, batch, Dl, d, Dr, a_in, a_out, 1, 1, nullptr, FR, A, M, nullptr, FRprev, m_shared, false);
}
int ttn_tdvp_apply_h2(int cplx, int64_t batch, int64_t Dl, int64_t d1, int64_t d2, int64_t Dr, int64_t a, int64_t b, int64_t c, const double* FL,
                      const double* AAC, const double* M1, const double* M2, const double* FR, double* HAAC, int m_shared) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    return tdvp_launch(4, cplx, batch, Dl, d1, Dr, a, b, c, d2, FL, FR, AAC, M1, M2, HAAC, m_shared, false);
}
// ---- dense QR / SVD of one local matrix, real or complex, device pointers (csrc/ttn_densefact_kernels.h) ----
int ttn_dense_qr(int cplx, int64_t m, int64_t n, double* A, double* Q, double* R) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!A || !Q || !R || m < 1 || n < 1) return fail(TTN_ERR_ARG, "ttn_dense_qr: bad argument");
    if (m > (1 << 20) || n > (1 << 20)) return fail(TTN_ERR_UNSUPPORTED, "ttn_dense_qr: matrix too large");
    const int64_t r = std::min(m, n);
    int rc = ensure_scratch(sizeof(double) * 2 * (size_t)r + 64);
    if (rc) return rc;
    if (cplx) hipLaunchKernelGGL(k_dense_qr<true>, dim3(1), dim3(TTN_DF_WG), 0, g_stream, (int)m, (int)n, A, Q, R, (double*)g_scratch);
    else hipLaunchKernelGGL(k_dense_qr<false>, dim3(1), dim3(TTN_DF_WG), 0, g_stream, (int)m, (int)n, A, Q, R, (double*)g_scratch);
    HIPCHK(hipGetLastError());
    return TTN_OK;
}
int ttn_dense_svd(int cplx, int64_t m, int64_t n, double* A, double* U, double* s, double* Vh) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    NEED_INIT();
    if (!A || !U || !s || !Vh || m < 1 || n < 1) return fail(TTN_ERR_ARG, "ttn_dense_svd: bad argument");
    if (m < n) return fail(TTN_ERR_ARG, "ttn_dense_svd: m >= n required (pass the conjugate transpose)");
    if (m > (1 << 20) || n > 4096) return fail(TTN_ERR_UNSUPPORTED, "ttn_dense_svd: matrix too large");
    const size_t w = cplx ? 2 : 1;
    const size_t vw = sizeof(double) * w * (size_t)n * n, dwb = sizeof(double) * (size_t)n;
    int rc = ensure_scratch(vw + dwb + sizeof(int) * ((size_t)n + 2) + 64);
    if (rc) return rc;
    double* Vw = (double*)g_scratch;
    double* dw = Vw + w * (size_t)n * n;
    int* iw = reinterpret_cast<int*>(dw + n);
    if (cplx) hipLaunchKernelGGL(k_dense_svd<true>, dim3(1), dim3(TTN_DF_WG), 0, g_stream, (int)m, (int)n, A, U, s, Vh, Vw, dw, iw, 60);
    else hipLaunchKernelGGL(k_dense_svd<false>, dim3(1), dim3(TTN_DF_WG), 0, g_stream, (int)m, (int)n, A, U, s, Vh, Vw, dw, iw, 60);
    HIPCHK(hipGetLastError());
    int flag = 0;
    HIPCHK(hipMemcpyAsync(&flag, iw + n, sizeof(int), hipMemcpyDeviceToHost, g_stream));
    HIPCHK(hipStreamSynchronize(g_stream));
    if (flag) return fail(TTN_ERR_NO_CONVERGENCE, "ttn_dense_svd: the Jacobi sweeps did not converge");
    return TTN_OK;
}

// host-array forms (what a `ccall` from tdvp1sweep! / tdvp2sweep! binds): op = 0 applyH1, 1 applyH0, 2 update_left_env,
// 3 update_right_env, 4 applyH2; dims = {Dl, d (d1), Dr, a, b, c, d2} with the meaning of the device forms above
int ttn_tdvp_contract_f64(int op, int cplx, int64_t batch, const int64_t* dims7, const double* FL, const double* FR, const double* X, const double* M1,
                          const double* M2, double* out, int m_shared) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!dims7) return fail(TTN_ERR_ARG, "tdvp: null dims");
    return tdvp_launch(op, cplx, batch, dims7[0], dims7[1], dims7[2], dims7[3], dims7[4], dims7[5], dims7[6], FL, FR, X, M1, M2, out, m_shared, true);
}

namespace {
struct TmpTT {
    ttn_tt_t h = nullptr;
    ~TmpTT() { if (h) ttn_tt_free(h); }
};
struct TmpTTO {
    ttn_tto_t h = nullptr;
    ~TmpTTO() { if (h) ttn_tto_free(h); }
};
int auto_init() {
    if (g_init) return TTN_OK;
    return ttn_init(0);
}
}  // namespace

int ttn_apply_f64(int64_t d, const int64_t* dims, const double* const* A_cores, const int64_t* A_rks,
                  const double* const* X_cores, const int64_t* X_rks, double* const* Y_cores) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    int rc = auto_init(); if (rc) return rc;
    if (!dims || !A_cores || !A_rks || !X_cores || !X_rks || !Y_cores || d < 1) return fail(TTN_ERR_ARG, "bad argument");
    TmpTTO A; TmpTT x, y;
    if ((rc = ttn_tto_create(d, dims, A_rks, A_cores, &A.h))) return rc;
    if ((rc = ttn_tt_create(d, dims, X_rks, 1, &x.h))) return rc;
    if ((rc = ttn_tt_upload(x.h, 0, X_cores, X_rks, nullptr))) return rc;
    std::vector<int64_t> yr(d + 1);
    for (int64_t m = 0; m <= d; ++m) yr[m] = A_rks[m] * X_rks[m];
    if ((rc = ttn_tt_create(d, dims, yr.data(), 1, &y.h))) return rc;
    if ((rc = ttn_apply(A.h, x.h, y.h))) return rc;
    return ttn_tt_download(y.h, 0, Y_cores);
}

int ttn_dot_f64(int64_t d, const int64_t* dims, const double* const* A_cores, const int64_t* A_rks,
                const double* const* B_cores, const int64_t* B_rks, double* out) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    int rc = auto_init(); if (rc) return rc;
    if (!dims || !A_cores || !A_rks || !B_cores || !B_rks || !out || d < 1) return fail(TTN_ERR_ARG, "bad argument");
    TmpTT a, b;
    if ((rc = ttn_tt_create(d, dims, A_rks, 1, &a.h))) return rc;
    if ((rc = ttn_tt_upload(a.h, 0, A_cores, A_rks, nullptr))) return rc;
    if ((rc = ttn_tt_create(d, dims, B_rks, 1, &b.h))) return rc;
    if ((rc = ttn_tt_upload(b.h, 0, B_cores, B_rks, nullptr))) return rc;
    return ttn_dot(a.h, b.h, out);
}

int ttn_hadamard_f64(int64_t d, const int64_t* dims, const double* const* X_cores, const int64_t* X_rks,
                     const double* const* Y_cores, const int64_t* Y_rks, double* const* Z_cores) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    int rc = auto_init(); if (rc) return rc;
    if (!dims || !X_cores || !X_rks || !Y_cores || !Y_rks || !Z_cores || d < 1) return fail(TTN_ERR_ARG, "bad argument");
    TmpTT x, y, z;
    std::vector<int64_t> zr(d + 1);
    for (int64_t m = 0; m <= d; ++m) zr[m] = X_rks[m] * Y_rks[m];
    if ((rc = ttn_tt_create(d, dims, X_rks, 1, &x.h))) return rc;
    if ((rc = ttn_tt_upload(x.h, 0, X_cores, X_rks, nullptr))) return rc;
    if ((rc = ttn_tt_create(d, dims, Y_rks, 1, &y.h))) return rc;
    if ((rc = ttn_tt_upload(y.h, 0, Y_cores, Y_rks, nullptr))) return rc;
    if ((rc = ttn_tt_create(d, dims, zr.data(), 1, &z.h))) return rc;
    if ((rc = ttn_hadamard(x.h, y.h, z.h))) return rc;
    return ttn_tt_download(z.h, 0, Z_cores);
}

int ttn_add_f64(int64_t d, const int64_t* dims, const double* const* X_cores, const int64_t* X_rks,
                const double* const* Y_cores, const int64_t* Y_rks, double* const* Z_cores) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    int rc = auto_init(); if (rc) return rc;
    if (!dims || !X_cores || !X_rks || !Y_cores || !Y_rks || !Z_cores || d < 1) return fail(TTN_ERR_ARG, "bad argument");
    TmpTT x, y, z;
    std::vector<int64_t> zr(d + 1);
    for (int64_t m = 0; m <= d; ++m) zr[m] = (m == 0 || m == d) ? 1 : X_rks[m] + Y_rks[m];
    if ((rc = ttn_tt_create(d, dims, X_rks, 1, &x.h))) return rc;
    if ((rc = ttn_tt_upload(x.h, 0, X_cores, X_rks, nullptr))) return rc;
    if ((rc = ttn_tt_create(d, dims, Y_rks, 1, &y.h))) return rc;
    if ((rc = ttn_tt_upload(y.h, 0, Y_cores, Y_rks, nullptr))) return rc;
    if ((rc = ttn_tt_create(d, dims, zr.data(), 1, &z.h))) return rc;
    if ((rc = ttn_add(x.h, y.h, z.h))) return rc;
    return ttn_tt_download(z.h, 0, Z_cores);
}

int ttn_scale_f64(int64_t d, const int64_t* dims, double a, const double* const* X_cores, const int64_t* X_rks,
                  const int64_t* X_ot, double* const* Y_cores, int64_t* Y_ot) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    int rc = auto_init(); if (rc) return rc;
    if (!dims || !X_cores || !X_rks || !Y_cores || d < 1) return fail(TTN_ERR_ARG, "bad argument");
    TmpTT x, y;
    if ((rc = ttn_tt_create(d, dims, X_rks, 1, &x.h))) return rc;
    if ((rc = ttn_tt_upload(x.h, 0, X_cores, X_rks, X_ot))) return rc;
    if ((rc = ttn_tt_create(d, dims, X_rks, 1, &y.h))) return rc;
    if ((rc = ttn_scale(a, x.h, y.h))) return rc;
    if (Y_ot) ttn_tt_ranks(y.h, 0, nullptr, Y_ot);
    return ttn_tt_download(y.h, 0, Y_cores);
}

int ttn_orthogonalize_f64(int64_t d, const int64_t* dims, const double* const* X_cores, const int64_t* X_rks, int64_t center,
                          double* const* Y_cores, int64_t* Y_rks, int64_t* Y_ot) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    int rc = auto_init(); if (rc) return rc;
    if (!dims || !X_cores || !X_rks || !Y_cores || !Y_rks || !Y_ot || d < 1) return fail(TTN_ERR_ARG, "bad argument");
    if (center < 1 || center > d) return fail(TTN_ERR_CENTER, "Impossible orthogonalization");
    TmpTT x, y;
    if ((rc = ttn_tt_create(d, dims, X_rks, 1, &x.h))) return rc;
    if ((rc = ttn_tt_upload(x.h, 0, X_cores, X_rks, nullptr))) return rc;
    if ((rc = ttn_tt_create(d, dims, X_rks, 1, &y.h))) return rc;
    if ((rc = ttn_orthogonalize(x.h, center, y.h))) return rc;
    if ((rc = ttn_tt_ranks(y.h, 0, Y_rks, Y_ot))) return rc;
    return ttn_tt_download(y.h, 0, Y_cores);
}

static int compress_host(int64_t d, const int64_t* dims, double* const* cores, int64_t* rks, int64_t k, int64_t max_bond,
                         double truncerr, int64_t sweeps) {
    int rc = auto_init(); if (rc) return rc;
    if (!dims || !cores || !rks || d < 1) return fail(TTN_ERR_ARG, "bad argument");
    TmpTT x;
    if (max_bond < 1) return fail(TTN_ERR_ARG, "max_bond must be >= 1");
    std::vector<int64_t> need, fin;
    long long pm, qm;
    rank_bounds((int)d, dims, rks, max_bond, sweeps, k, need, fin, pm, qm);
    if ((rc = ttn_tt_create(d, dims, need.data(), 1, &x.h))) return rc;
    if ((rc = ttn_tt_upload(x.h, 0, cores, rks, nullptr))) return rc;
    if (k > 0) rc = ttn_bond_truncate(x.h, k, max_bond, truncerr);
    else rc = ttn_compress(x.h, max_bond, truncerr, sweeps);
    if (rc) return rc;
    if ((rc = ttn_compress_status(x.h, nullptr))) return rc;
    if ((rc = ttn_tt_ranks(x.h, 0, rks, nullptr))) return rc;
    return ttn_tt_download(x.h, 0, cores);
}

int ttn_compress_f64(int64_t d, const int64_t* dims, double* const* cores, int64_t* rks, int64_t max_bond, double truncerr,
                     int64_t sweeps) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (sweeps < 1) return fail(TTN_ERR_SWEEPS, "sweeps must be >= 1");
    return compress_host(d, dims, cores, rks, 0, max_bond, truncerr, sweeps);
}

// The Krylov operator of the reference in ONE stateless call: op = x -> tt_compress!(A * x, max_bond) (src/solvers/euler.jl:55).  A * x
// is never materialised (fused apply, k_compress builds the merged matrices from x and A) — neither in HBM nor over PCIe: the host
// hands over A and x, and receives the compressed train.  Y_cores[k] sized n_k * cap_k * cap_{k+1} with cap = min(A_rks .* X_rks,
// max_bond-capped bounds) as ttn_apply_compress_rank_bound returns them; Y_rks receives the ranks.
int ttn_apply_compress_rank_bound(int64_t d, const int64_t* dims, const int64_t* A_rks, const int64_t* X_rks, int64_t max_bond, int64_t sweeps,
                                  int64_t* cap) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (!dims || !A_rks || !X_rks || !cap || d < 1 || max_bond < 1 || sweeps < 1) return fail(TTN_ERR_ARG, "bad argument");
    std::vector<int64_t> yr(d + 1), need, fin;
    for (int64_t m = 0; m <= d; ++m) yr[m] = A_rks[m] * X_rks[m];
    long long pm, qm;
    rank_bounds((int)d, dims, yr.data(), max_bond, sweeps, 0, need, fin, pm, qm);
    for (int64_t m = 0; m <= d; ++m) cap[m] = fin[m];
    return TTN_OK;
}
int ttn_apply_compress_f64(int64_t d, const int64_t* dims, const double* const* A_cores, const int64_t* A_rks, const double* const* X_cores,
                           const int64_t* X_rks, double* const* Y_cores, int64_t* Y_rks, int64_t max_bond, double truncerr, int64_t sweeps) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    int rc = auto_init(); if (rc) return rc;
    if (!dims || !A_cores || !A_rks || !X_cores || !X_rks || !Y_cores || !Y_rks || d < 1) return fail(TTN_ERR_ARG, "bad argument");
    if (sweeps < 1) return fail(TTN_ERR_SWEEPS, "sweeps must be >= 1");
    if (max_bond < 1) return fail(TTN_ERR_ARG, "max_bond must be >= 1");
    TmpTTO A; TmpTT x, y;
    if ((rc = ttn_tto_create(d, dims, A_rks, A_cores, &A.h))) return rc;
    if ((rc = ttn_tt_create(d, dims, X_rks, 1, &x.h))) return rc;
    if ((rc = ttn_tt_upload(x.h, 0, X_cores, X_rks, nullptr))) return rc;
    std::vector<int64_t> yr(d + 1), need, fin;
    for (int64_t m = 0; m <= d; ++m) yr[m] = A_rks[m] * X_rks[m];
    long long pm, qm;
    rank_bounds((int)d, dims, yr.data(), max_bond, sweeps, 0, need, fin, pm, qm);
    for (int64_t m = 0; m <= d; ++m) need[m] = std::max<int64_t>(need[m], yr[m]);
    if ((rc = ttn_tt_create(d, dims, need.data(), 1, &y.h))) return rc;
    if ((rc = ttn_apply_compress(A.h, x.h, y.h, max_bond, truncerr, sweeps))) return rc;
    if ((rc = ttn_compress_status(y.h, nullptr))) return rc;
    if ((rc = ttn_tt_ranks(y.h, 0, Y_rks, nullptr))) return rc;
    return ttn_tt_download(y.h, 0, Y_cores);
}

int ttn_bond_truncate_f64(int64_t d, const int64_t* dims, double* const* cores, int64_t* rks, int64_t k, int64_t max_bond,
                          double truncerr) {
    std::lock_guard<std::recursive_mutex> lk(g_mu);
    if (k < 1 || k >= d) return fail(TTN_ERR_BOND_INDEX, "k must be in 1:(N-1)");
    return compress_host(d, dims, cores, rks, k, max_bond, truncerr, 1);
}

}  // extern "C"
