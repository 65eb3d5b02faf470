// libafhip runtime: device / memory / stream / event plumbing behind the C ABI
// (include/afhip.h, "runtime" section) and the per-thread error text.
#include <stdarg.h>

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
