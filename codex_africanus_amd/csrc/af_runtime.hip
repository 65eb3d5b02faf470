// libafhip runtime: device / memory / stream / event plumbing behind the C ABI
// (include/afhip.h, "runtime" section) and the per-thread error text.
#include <stdarg.h>
#include <stdlib.h>

#include "af_common.h"

static thread_local char g_err[512] = "";

void af_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int af_hip_fail(hipError_t e, const char *what, const char *file, int line)
{
    af_set_error("HIP error %d (%s) in `%s` at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
    return AF_EHIP_BASE + (int)e;
}

AF_EXPORT int af_version(void) { return 100; }

// ---- measurement hook: events recorded around the dominant kernel of the next calls -------
static thread_local hipEvent_t g_prof_start = nullptr, g_prof_stop = nullptr;

AF_EXPORT int af_profile_events(void *start, void *stop)
{
    g_prof_start = reinterpret_cast<hipEvent_t>(start);
    g_prof_stop = reinterpret_cast<hipEvent_t>(stop);
    return AF_OK;
}

void af_prof_begin(hipStream_t st)
{
    if (g_prof_start) (void)hipEventRecord(g_prof_start, st);
}

void af_prof_end(hipStream_t st)
{
    if (g_prof_stop) (void)hipEventRecord(g_prof_stop, st);
}

AF_EXPORT const char *af_last_error(void) { return g_err; }

AF_EXPORT int af_device_count(int *count)
{
    AF_REQUIRE(count != nullptr, "af_device_count: count is NULL");
    AF_HIP(hipGetDeviceCount(count));
    return AF_OK;
}

AF_EXPORT int af_set_device(int device)
{
    AF_HIP(hipSetDevice(device));
    return AF_OK;
}

AF_EXPORT int af_get_device(int *device)
{
    AF_REQUIRE(device != nullptr, "af_get_device: device is NULL");
    AF_HIP(hipGetDevice(device));
    return AF_OK;
}

AF_EXPORT int af_device_info(int device, char *name, size_t name_len, char *arch, size_t arch_len,
                             int *compute_units, size_t *total_mem)
{
    hipDeviceProp_t p;
    AF_HIP(hipGetDeviceProperties(&p, device));
    if (name && name_len) { strncpy(name, p.name, name_len - 1); name[name_len - 1] = 0; }
    if (arch && arch_len) { strncpy(arch, p.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
    if (compute_units) *compute_units = p.multiProcessorCount;
    if (total_mem) *total_mem = p.totalGlobalMem;
    return AF_OK;
}

AF_EXPORT int af_malloc(void **dptr, size_t bytes)
{
    AF_REQUIRE(dptr != nullptr, "af_malloc: dptr is NULL");
    *dptr = nullptr;
    if (bytes == 0) return AF_OK;
    hipError_t e = hipMalloc(dptr, bytes);
    if (e == hipErrorOutOfMemory) {
        af_set_error("af_malloc: out of device memory (%zu bytes)", bytes);
        (void)hipGetLastError();
        return AF_ENOMEM;
    }
    AF_HIP(e);
    return AF_OK;
}

AF_EXPORT int af_free(void *dptr)
{
    if (dptr) AF_HIP(hipFree(dptr));
    return AF_OK;
}

AF_EXPORT int af_malloc_host(void **hptr, size_t bytes)
{
    AF_REQUIRE(hptr != nullptr, "af_malloc_host: hptr is NULL");
    *hptr = nullptr;
    if (bytes == 0) return AF_OK;
    AF_HIP(hipHostMalloc(hptr, bytes, hipHostMallocDefault));
    return AF_OK;
}

AF_EXPORT int af_free_host(void *hptr)
{
    if (hptr) AF_HIP(hipHostFree(hptr));
    return AF_OK;
}


// ---- per-device scratch pool, per-thread streams, af_shutdown (SURVEY 8(b) "Ownership") ---------------------
// The reference's allocations are plain np.zeros / np.empty per call (africanus/rime/predict.py:271,
// africanus/dft/kernels.py:45); on the device a hipMalloc / hipFree pair per array per call costs tens to
// hundreds of microseconds and synchronises the device, which dominates a dask block of a few thousand rows.
// The pool keeps freed blocks per device (and one list for page-locked host memory) and hands them out again:
//   * requests are rounded up (256 B .. 1 MiB: next power of two; above: next multiple of 2 MiB), so ragged dask
//     chunks of similar size share blocks;
//   * a cached block is taken when it is no larger than 1.25 x the rounded request (best fit, size-ordered map);
//   * the cached bytes per list are capped (AFHIP_POOL_LIMIT / AFHIP_PINNED_LIMIT, bytes; defaults 32 GiB / 8 GiB):
//     beyond the cap the least recently freed blocks are released; hipMalloc failing with out-of-memory
//     releases the whole cache of that device and retries once;
//   * af_shutdown() releases everything and destroys the per-thread streams (no other call may be in flight).
// Ordering contract: af_pool_free may be called as soon as the work that uses the block has been ENQUEUED on
// the calling thread's stream (af_thread_stream) only if the next user runs on that same stream; the Python
// host path synchronises its stream before it returns a block, so blocks are idle when they change threads.
#include <sys/syscall.h>
#include <unistd.h>

#include <list>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace {

constexpr int AF_MAX_DEVICES = 64;
constexpr int HOST_LIST = AF_MAX_DEVICES;   // index of the page-locked host list

struct Block { void *ptr; size_t bytes; uint64_t tick; };

struct PoolList {
    std::multimap<size_t, Block> free_by_size;   // cached blocks
    size_t cached = 0, in_use = 0, limit = 0;
    int64_t hits = 0, misses = 0;
};

struct Pool {
    std::mutex mu;
    PoolList lists[AF_MAX_DEVICES + 1];
    std::unordered_map<void *, std::pair<int, size_t>> live;   // ptr -> (list, rounded bytes)
    uint64_t tick = 0;
    uint64_t generation = 1;                                    // bumped by af_shutdown
    std::vector<std::pair<int, hipStream_t>> streams;           // every per-thread stream ever created
    bool limits_read = false;
};

Pool &pool()
{
    static Pool *p = new Pool();   // never destroyed: no static-destruction order problems at process exit
    return *p;
}

size_t env_bytes(const char *name, size_t dflt)
{
    const char *v = getenv(name);
    if (!v || !*v) return dflt;
    char *end = nullptr;
    const unsigned long long x = strtoull(v, &end, 10);
    return end && end != v ? (size_t)x : dflt;
}

void read_limits(Pool &P)
{
    if (P.limits_read) return;
    const size_t dev_limit = env_bytes("AFHIP_POOL_LIMIT", (size_t)32 << 30);
    for (int d = 0; d < AF_MAX_DEVICES; ++d) P.lists[d].limit = dev_limit;
    P.lists[HOST_LIST].limit = env_bytes("AFHIP_PINNED_LIMIT", (size_t)8 << 30);
    P.limits_read = true;
}

size_t round_request(size_t bytes)
{
    if (bytes <= 256) return 256;
    if (bytes <= ((size_t)1 << 20)) {
        size_t r = 256;
        while (r < bytes) r <<= 1;
        return r;
    }
    const size_t g = (size_t)2 << 20;
    return (bytes + g - 1) / g * g;
}

hipError_t raw_free(int list, void *p) { return list == HOST_LIST ? hipHostFree(p) : hipFree(p); }

// release cached blocks of `list` (oldest first) until at most `keep` bytes stay cached; caller holds the lock.
// hipFree needs the owning device current only for the allocation call, not for the free.
void trim_locked(Pool &P, int list, size_t keep)
{
    PoolList &L = P.lists[list];
    while (L.cached > keep && !L.free_by_size.empty()) {
        auto oldest = L.free_by_size.begin();
        for (auto it = L.free_by_size.begin(); it != L.free_by_size.end(); ++it)
            if (it->second.tick < oldest->second.tick) oldest = it;
        (void)raw_free(list, oldest->second.ptr);
        L.cached -= oldest->second.bytes;
        L.free_by_size.erase(oldest);
    }
}

int pool_alloc(int list, void **out, size_t bytes, const char *who)
{
    Pool &P = pool();
    const size_t want = round_request(bytes);
    {
        std::lock_guard<std::mutex> g(P.mu);
        read_limits(P);
        PoolList &L = P.lists[list];
        auto it = L.free_by_size.lower_bound(want);
        if (it != L.free_by_size.end() && it->first <= want + want / 4) {
            *out = it->second.ptr;
            const size_t got = it->second.bytes;
            L.cached -= got;
            L.in_use += got;
            L.hits++;
            L.free_by_size.erase(it);
            P.live[*out] = std::make_pair(list, got);
            return AF_OK;
        }
        L.misses++;
    }
    void *p = nullptr;
    hipError_t e = list == HOST_LIST ? hipHostMalloc(&p, want, hipHostMallocDefault) : hipMalloc(&p, want);
    if (e == hipErrorOutOfMemory) {   // give the cache back and retry once
        (void)hipGetLastError();
        {
            std::lock_guard<std::mutex> g(P.mu);
            trim_locked(P, list, 0);
        }
        e = list == HOST_LIST ? hipHostMalloc(&p, want, hipHostMallocDefault) : hipMalloc(&p, want);
        if (e == hipErrorOutOfMemory) {
            (void)hipGetLastError();
            af_set_error("%s: out of %s memory (%zu bytes)", who, list == HOST_LIST ? "page-locked host" : "device", want);
            return AF_ENOMEM;
        }
    }
    AF_HIP(e);
    std::lock_guard<std::mutex> g(P.mu);
    P.lists[list].in_use += want;
    P.live[p] = std::make_pair(list, want);
    *out = p;
    return AF_OK;
}

int pool_release(void *p, const char *who)
{
    if (!p) return AF_OK;
    Pool &P = pool();
    std::lock_guard<std::mutex> g(P.mu);
    auto it = P.live.find(p);
    AF_REQUIRE(it != P.live.end(), "%s: pointer %p was not allocated by the pool (or freed twice)", who, p);
    const int list = it->second.first;
    const size_t bytes = it->second.second;
    P.live.erase(it);
    PoolList &L = P.lists[list];
    L.in_use -= bytes;
    if (bytes > L.limit) {   // can never be cached
        AF_HIP(raw_free(list, p));
        return AF_OK;
    }
    L.free_by_size.emplace(bytes, Block{p, bytes, ++P.tick});
    L.cached += bytes;
    trim_locked(P, list, L.limit);
    return AF_OK;
}

// A worker thread that ends (dask / ThreadPoolExecutor churn) takes its streams with it: synchronised, destroyed and
// struck from the pool's list, so a long-running process does not accumulate one hipStream per (thread, device) it
// ever had.  The process's main thread (tid == pid) is exempt: its thread-locals are destroyed inside
// exit(), next to the HIP runtime's own teardown, and its streams die with the process anyway.
}  // namespace
void af_wgrid_drop_stream(hipStream_t st);   // af_wgridder.hip: FFT plans are per stream
namespace {

struct ThreadStreams {
    uint64_t generation = 0;
    hipStream_t s[AF_MAX_DEVICES] = {};
    ~ThreadStreams()
    {
        if ((long)getpid() == (long)syscall(SYS_gettid)) return;
        Pool &P = pool();
        {
            std::lock_guard<std::mutex> g(P.mu);
            if (generation != P.generation) return;          // af_shutdown already destroyed them
            for (auto it = P.streams.begin(); it != P.streams.end();)
                it = (it->first >= 0 && it->first < AF_MAX_DEVICES && s[it->first] == it->second) ? P.streams.erase(it) : it + 1;
        }
        int dev0 = 0;
        const bool have_dev = hipGetDevice(&dev0) == hipSuccess;
        for (int d = 0; d < AF_MAX_DEVICES; ++d) {
            if (!s[d]) continue;
            if (hipSetDevice(d) == hipSuccess) {
                (void)hipStreamSynchronize(s[d]);
                af_wgrid_drop_stream(s[d]);
                (void)hipStreamDestroy(s[d]);
            }
            s[d] = nullptr;
        }
        if (have_dev) (void)hipSetDevice(dev0);
        (void)hipGetLastError();
    }
};
thread_local ThreadStreams t_streams;

}  // namespace

AF_EXPORT int af_pool_malloc(void **dptr, size_t bytes)
{
    AF_REQUIRE(dptr != nullptr, "af_pool_malloc: dptr is NULL");
    *dptr = nullptr;
    if (bytes == 0) return AF_OK;
    int dev = 0;
    AF_HIP(hipGetDevice(&dev));
    AF_REQUIRE(dev >= 0 && dev < AF_MAX_DEVICES, "af_pool_malloc: device %d out of range", dev);
    return pool_alloc(dev, dptr, bytes, "af_pool_malloc");
}

AF_EXPORT int af_pool_free(void *dptr) { return pool_release(dptr, "af_pool_free"); }

AF_EXPORT int af_pool_malloc_host(void **hptr, size_t bytes)
{
    AF_REQUIRE(hptr != nullptr, "af_pool_malloc_host: hptr is NULL");
    *hptr = nullptr;
    if (bytes == 0) return AF_OK;
    return pool_alloc(HOST_LIST, hptr, bytes, "af_pool_malloc_host");
}

AF_EXPORT int af_pool_free_host(void *hptr) { return pool_release(hptr, "af_pool_free_host"); }

AF_EXPORT int af_pool_trim(int device, size_t keep_bytes)
{
    AF_REQUIRE(device >= -1 && device < AF_MAX_DEVICES, "af_pool_trim: device %d out of range (-1 = host list)", device);
    Pool &P = pool();
    std::lock_guard<std::mutex> g(P.mu);
    trim_locked(P, device < 0 ? HOST_LIST : device, keep_bytes);
    return AF_OK;
}

AF_EXPORT int af_pool_stats(int device, size_t *cached_bytes, size_t *in_use_bytes, int64_t *hits, int64_t *misses)
{
    AF_REQUIRE(device >= -1 && device < AF_MAX_DEVICES, "af_pool_stats: device %d out of range (-1 = host list)", device);
    Pool &P = pool();
    std::lock_guard<std::mutex> g(P.mu);
    const PoolList &L = P.lists[device < 0 ? HOST_LIST : device];
    if (cached_bytes) *cached_bytes = L.cached;
    if (in_use_bytes) *in_use_bytes = L.in_use;
    if (hits) *hits = L.hits;
    if (misses) *misses = L.misses;
    return AF_OK;
}

// The calling thread's own stream on its current device, created at first use: concurrent host-mode calls from
// different threads (dask workers) neither serialise on the NULL stream nor synchronise with one another.
AF_EXPORT int af_thread_stream(void **stream)
{
    AF_REQUIRE(stream != nullptr, "af_thread_stream: stream is NULL");
    int dev = 0;
    AF_HIP(hipGetDevice(&dev));
    AF_REQUIRE(dev >= 0 && dev < AF_MAX_DEVICES, "af_thread_stream: device %d out of range", dev);
    Pool &P = pool();
    uint64_t gen;
    {
        std::lock_guard<std::mutex> g(P.mu);
        gen = P.generation;
    }
    if (t_streams.generation != gen) {   // first use, or af_shutdown destroyed this thread's streams
        for (int d = 0; d < AF_MAX_DEVICES; ++d) t_streams.s[d] = nullptr;
        t_streams.generation = gen;
    }
    if (!t_streams.s[dev]) {
        hipStream_t s;
        AF_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        t_streams.s[dev] = s;
        std::lock_guard<std::mutex> g(P.mu);
        P.streams.emplace_back(dev, s);
    }
    *stream = t_streams.s[dev];
    return AF_OK;
}

// Releases every cached block (all devices and the page-locked list) and destroys the per-thread streams.
// Blocks still handed out stay valid and are released by their af_pool_free.  No other libafhip call may be in
// flight on another thread.  Idempotent; the library is usable again afterwards.
void af_wgrid_shutdown();   // af_wgridder.hip: cached FFT plans

AF_EXPORT int af_shutdown(void)
{
    af_wgrid_shutdown();
    Pool &P = pool();
    std::vector<std::pair<int, hipStream_t>> streams;
    {
        std::lock_guard<std::mutex> g(P.mu);
        for (int l = 0; l <= AF_MAX_DEVICES; ++l) trim_locked(P, l, 0);
        streams.swap(P.streams);
        P.generation++;
    }
    int dev0 = 0;
    const bool have_dev = hipGetDevice(&dev0) == hipSuccess;
    int rc = AF_OK;
    for (auto &ds : streams) {
        if (hipSetDevice(ds.first) != hipSuccess) continue;
        (void)hipStreamSynchronize(ds.second);
        if (hipStreamDestroy(ds.second) != hipSuccess) rc = AF_EHIP_BASE;
    }
    if (have_dev) (void)hipSetDevice(dev0);
    (void)hipGetLastError();
    return rc;
}

AF_EXPORT int af_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream)
{
    if (bytes) AF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, af_stream(stream)));
    return AF_OK;
}

AF_EXPORT int af_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream)
{
    if (bytes) AF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, af_stream(stream)));
    return AF_OK;
}

AF_EXPORT int af_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream)
{
    if (bytes) AF_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, af_stream(stream)));
    return AF_OK;
}

AF_EXPORT int af_memset(void *dst, int value, size_t bytes, void *stream)
{
    if (bytes) AF_HIP(hipMemsetAsync(dst, value, bytes, af_stream(stream)));
    return AF_OK;
}

AF_EXPORT int af_stream_create(void **stream)
{
    AF_REQUIRE(stream != nullptr, "af_stream_create: stream is NULL");
    hipStream_t s;
    AF_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return AF_OK;
}

AF_EXPORT int af_stream_destroy(void *stream)
{
    if (stream) AF_HIP(hipStreamDestroy(af_stream(stream)));
    return AF_OK;
}

AF_EXPORT int af_stream_synchronize(void *stream)
{
    AF_HIP(hipStreamSynchronize(af_stream(stream)));
    return AF_OK;
}

AF_EXPORT int af_device_synchronize(void)
{
    AF_HIP(hipDeviceSynchronize());
    return AF_OK;
}

AF_EXPORT int af_event_create(void **event)
{
    AF_REQUIRE(event != nullptr, "af_event_create: event is NULL");
    hipEvent_t e;
    AF_HIP(hipEventCreate(&e));
    *event = e;
    return AF_OK;
}

AF_EXPORT int af_event_destroy(void *event)
{
    if (event) AF_HIP(hipEventDestroy(reinterpret_cast<hipEvent_t>(event)));
    return AF_OK;
}

AF_EXPORT int af_event_record(void *event, void *stream)
{
    AF_HIP(hipEventRecord(reinterpret_cast<hipEvent_t>(event), af_stream(stream)));
    return AF_OK;
}

AF_EXPORT int af_stream_wait_event(void *stream, void *event)
{
    AF_REQUIRE(event != nullptr, "af_stream_wait_event: event is NULL");
    AF_HIP(hipStreamWaitEvent(af_stream(stream), reinterpret_cast<hipEvent_t>(event), 0));
    return AF_OK;
}

AF_EXPORT int af_event_synchronize(void *event)
{
    AF_HIP(hipEventSynchronize(reinterpret_cast<hipEvent_t>(event)));
    return AF_OK;
}

AF_EXPORT int af_event_elapsed_ms(void *start, void *stop, float *ms)
{
    AF_REQUIRE(ms != nullptr, "af_event_elapsed_ms: ms is NULL");
    AF_HIP(hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)));
    return AF_OK;
}

// ---- dtype promotion helpers ----------------------------------------------------
template <typename S, typename D>
__global__ void convert_kernel(const S *__restrict__ src, D *__restrict__ dst, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = (D)src[i];
}

template <typename S, typename D>
static int convert(const S *src, D *dst, int64_t n, void *stream)
{
    if (n <= 0) return AF_OK;
    AF_REQUIRE(src && dst, "af_convert: NULL pointer");
    int64_t blocks = af_cdiv(n, 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL((convert_kernel<S, D>), dim3((unsigned)blocks), dim3(256), 0, af_stream(stream), src, dst, n);
    AF_LAUNCH_CHECK();
    return AF_OK;
}

AF_EXPORT int af_convert_f32_to_f64(const float *src, double *dst, int64_t n, void *stream)
{
    return convert(src, dst, n, stream);
}

AF_EXPORT int af_convert_f64_to_f32(const double *src, float *dst, int64_t n, void *stream)
{
    return convert(src, dst, n, stream);
}

#ifdef AFHIP_STAGE_HOOKS
// profiling build only (tools/check_lds_independence.py): every CU's LDS <- `pattern` bytes, so that a kernel which reads
// LDS it has not written -- whatever the workgroup before it left there, usually one of its own -- shows in its result
__global__ __launch_bounds__(1024) void debug_fill_lds_kernel(unsigned pattern, unsigned *sink)
{
    extern __shared__ unsigned lds_all[];
    const int n = 160 * 1024 / 4;
    for (int i = threadIdx.x; i < n; i += blockDim.x) lds_all[i] = pattern;
    __syncthreads();
    if (sink && lds_all[(threadIdx.x * 977) % n] != pattern) *sink = 1;     // keeps the stores
}
AF_EXPORT int af_debug_fill_lds(unsigned pattern, void *stream)
{
    const void *k = reinterpret_cast<const void *>(debug_fill_lds_kernel);
    AF_HIP(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    // one workgroup owns a whole CU's LDS: several rounds over the chip so that every CU is certainly visited
    hipLaunchKernelGGL(debug_fill_lds_kernel, dim3(256 * 16), dim3(1024), 160 * 1024, af_stream(stream), pattern, (unsigned *)nullptr);
    AF_LAUNCH_CHECK();
    return AF_OK;
}
#endif
