// Runtime plumbing of the C ABI: device selection, memory, streams, events.
#include <cstring>

#include "common.h"

namespace qrec {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace qrec

using namespace qrec;

extern "C" {

int qrec_version(void) { return 100; }
const char *qrec_last_error(void) { return g_err; }

int qrec_device_count(int *n) {
    QREC_REQUIRE(n, "qrec_device_count: null output");
    QREC_HIP_CHECK(hipGetDeviceCount(n));
    return QREC_OK;
}

int qrec_init(int device) {
    int n = 0;
    QREC_HIP_CHECK(hipGetDeviceCount(&n));
    QREC_REQUIRE(device >= 0 && device < n, "qrec_init: device %d out of range (have %d)", device, n);
    QREC_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t p;
    QREC_HIP_CHECK(hipGetDeviceProperties(&p, device));
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        set_error("qrec_init: device %d is %s; this library is built for gfx950 only", device,
                  p.gcnArchName);
        return QREC_ERR_UNSUPPORTED;
    }
    return QREC_OK;
}

int qrec_device_info(char *name, int name_len, int *n_cu, int64_t *hbm_bytes, char *arch,
                     int arch_len) {
    int dev = 0;
    QREC_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t p;
    QREC_HIP_CHECK(hipGetDeviceProperties(&p, dev));
    if (name && name_len > 0) { strncpy(name, p.name, name_len - 1); name[name_len - 1] = 0; }
    if (arch && arch_len > 0) { strncpy(arch, p.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    return QREC_OK;
}

int qrec_malloc(int64_t bytes, void **d_ptr) {
    QREC_REQUIRE(d_ptr && bytes >= 0, "qrec_malloc: bad arguments");
    *d_ptr = nullptr;
    if (bytes == 0) return QREC_OK;
    QREC_HIP_CHECK(hipMalloc(d_ptr, (size_t)bytes));
    return QREC_OK;
}
int qrec_free(void *d_ptr) {
    if (d_ptr) QREC_HIP_CHECK(hipFree(d_ptr));
    return QREC_OK;
}
int qrec_memcpy_h2d(void *d, const void *h, int64_t bytes, void *stream) {
    if (bytes == 0) return QREC_OK;
    QREC_REQUIRE(d && h && bytes > 0, "qrec_memcpy_h2d: bad arguments");
    QREC_HIP_CHECK(hipMemcpyAsync(d, h, (size_t)bytes, hipMemcpyHostToDevice, as_stream(stream)));
    QREC_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));  // host buffer is borrowed
    return QREC_OK;
}
int qrec_memcpy_d2h(void *h, const void *d, int64_t bytes, void *stream) {
    if (bytes == 0) return QREC_OK;
    QREC_REQUIRE(d && h && bytes > 0, "qrec_memcpy_d2h: bad arguments");
    QREC_HIP_CHECK(hipMemcpyAsync(h, d, (size_t)bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    QREC_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
    return QREC_OK;
}
// page-locked host memory + a device-to-host copy that only enqueues: small read-backs (row counts of an exchange plan) that the
// host picks up behind an event while the stream carries on
int qrec_host_alloc(int64_t bytes, void **h_ptr) {
    QREC_REQUIRE(h_ptr && bytes > 0, "qrec_host_alloc: bad arguments");
    *h_ptr = nullptr;
    QREC_HIP_CHECK(hipHostMalloc(h_ptr, (size_t)bytes, hipHostMallocDefault));
    return QREC_OK;
}
int qrec_host_free(void *h_ptr) {
    if (h_ptr) QREC_HIP_CHECK(hipHostFree(h_ptr));
    return QREC_OK;
}
int qrec_memcpy_d2h_async(void *h_pinned, const void *d, int64_t bytes, void *stream) {
    if (bytes == 0) return QREC_OK;
    QREC_REQUIRE(d && h_pinned && bytes > 0, "qrec_memcpy_d2h_async: bad arguments");
    QREC_HIP_CHECK(hipMemcpyAsync(h_pinned, d, (size_t)bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    return QREC_OK;
}
int qrec_memcpy_d2d(void *dst, const void *src, int64_t bytes, void *stream) {
    if (bytes == 0) return QREC_OK;
    QREC_REQUIRE(dst && src && bytes > 0, "qrec_memcpy_d2d: bad arguments");
    QREC_HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return QREC_OK;
}
int qrec_memset(void *d, int byte, int64_t bytes, void *stream) {
    if (bytes == 0) return QREC_OK;
    QREC_REQUIRE(d && bytes > 0, "qrec_memset: bad arguments");
    QREC_HIP_CHECK(hipMemsetAsync(d, byte, (size_t)bytes, as_stream(stream)));
    return QREC_OK;
}
int qrec_stream_create(void **stream) {
    QREC_REQUIRE(stream, "qrec_stream_create: null output");
    hipStream_t s;
    QREC_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return QREC_OK;
}
int qrec_stream_destroy(void *stream) {
    if (stream) QREC_HIP_CHECK(hipStreamDestroy(as_stream(stream)));
    return QREC_OK;
}
int qrec_stream_sync(void *stream) {
    QREC_HIP_CHECK(hipStreamSynchronize(as_stream(stream)));
    return QREC_OK;
}
int qrec_device_sync(void) {
    QREC_HIP_CHECK(hipDeviceSynchronize());
    return QREC_OK;
}
int qrec_event_create(void **ev) {
    QREC_REQUIRE(ev, "qrec_event_create: null output");
    hipEvent_t e;
    QREC_HIP_CHECK(hipEventCreate(&e));
    *ev = e;
    return QREC_OK;
}
int qrec_event_destroy(void *ev) {
    if (ev) QREC_HIP_CHECK(hipEventDestroy((hipEvent_t)ev));
    return QREC_OK;
}
int qrec_event_record(void *ev, void *stream) {
    QREC_REQUIRE(ev, "qrec_event_record: null event");
    QREC_HIP_CHECK(hipEventRecord((hipEvent_t)ev, as_stream(stream)));
    return QREC_OK;
}
int qrec_event_sync(void *ev) {
    QREC_REQUIRE(ev, "qrec_event_sync: null event");
    QREC_HIP_CHECK(hipEventSynchronize((hipEvent_t)ev));
    return QREC_OK;
}
int qrec_stream_wait_event(void *stream, void *ev) {
    QREC_REQUIRE(ev, "qrec_stream_wait_event: null event");
    QREC_HIP_CHECK(hipStreamWaitEvent(as_stream(stream), (hipEvent_t)ev, 0));
    return QREC_OK;
}
int qrec_event_elapsed_ms(void *a, void *b, float *ms) {
    QREC_REQUIRE(a && b && ms, "qrec_event_elapsed_ms: bad arguments");
    QREC_HIP_CHECK(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
    return QREC_OK;
}

}  // extern "C"
