// device_capi.hip -- the few device-memory entry points a C++ host needs to drive SEVERAL GPUs through the C-ABI without including
// HIP itself (include/gdpt_tracer.h, "multi-device helpers"): allocation on a named device and copies between devices.  A copy between
// two GPUs of one node is a peer-to-peer DMA over xGMI (hipMemcpyPeerAsync; peer access is enabled on first use) -- the single-process
// counterpart of the RCCL send/recv that parallel.py uses between processes.
#include "../../include/gdpt_tracer.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <set>
#include <utility>

extern "C" int gdpt_internal_fail(int code, const char *msg);

namespace {

int dfail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return gdpt_internal_fail(code, buf);
}

#define DHIPCHK(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return dfail(GDPT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

int check_device(int device)
{
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return dfail(GDPT_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= count) return dfail(GDPT_ERR_INVALID, "device %d out of range (0..%d)", device, count - 1);
    return GDPT_OK;
}

std::mutex g_peerMutex;
std::set<std::pair<int, int>> g_peers;          // (from, to) pairs whose peer access has been switched on

int enable_peer(int from, int to)
{
    if (from == to) return GDPT_OK;
    std::lock_guard<std::mutex> lock(g_peerMutex);
    if (g_peers.count({from, to})) return GDPT_OK;
    int can = 0;
    DHIPCHK(hipDeviceCanAccessPeer(&can, from, to));
    if (can) {
        DHIPCHK(hipSetDevice(from));
        const hipError_t e = hipDeviceEnablePeerAccess(to, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return dfail(GDPT_ERR_HIP, "hipDeviceEnablePeerAccess(%d -> %d): %s", from, to, hipGetErrorString(e));
        (void)hipGetLastError();
    }
    g_peers.insert({from, to});                   // (without peer access hipMemcpyPeer stages through the host: slower, still correct)
    return GDPT_OK;
}

} // namespace

extern "C" {

int gdpt_device_count(int *count)
{
    if (!count) return dfail(GDPT_ERR_INVALID, "null argument");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count = n;
    return GDPT_OK;
}

int gdpt_device_alloc(int device, size_t bytes, void **ptr)
{
    if (!ptr || bytes == 0) return dfail(GDPT_ERR_INVALID, "device_alloc: bad argument");
    int rc = check_device(device);
    if (rc) return rc;
    DHIPCHK(hipSetDevice(device));
    if (hipMalloc(ptr, bytes) != hipSuccess) { (void)hipGetLastError(); return dfail(GDPT_ERR_HIP, "Out of memory!"); }
    return GDPT_OK;
}

int gdpt_device_free(int device, void *ptr)
{
    if (!ptr) return GDPT_OK;
    int rc = check_device(device);
    if (rc) return rc;
    DHIPCHK(hipSetDevice(device));
    DHIPCHK(hipFree(ptr));
    return GDPT_OK;
}

int gdpt_device_copy(int dstDevice, void *dst, int srcDevice, const void *src, size_t bytes)
{
    if (!dst || !src) return dfail(GDPT_ERR_INVALID, "device_copy: null pointer");
    int rc = check_device(dstDevice);
    if (!rc) rc = check_device(srcDevice);
    if (rc) return rc;
    if (bytes == 0) return GDPT_OK;
    if (dstDevice == srcDevice) {
        DHIPCHK(hipSetDevice(dstDevice));
        // (a device-to-device hipMemcpy may return before the copy has run, and the films' streams are non-blocking: without the wait a
        // strip could unpack a halo payload that is still in flight -- seen as an intermittent wrong border row with two strips on one GPU)
        DHIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, nullptr));
        DHIPCHK(hipStreamSynchronize(nullptr));
        return GDPT_OK;
    }
    if ((rc = enable_peer(srcDevice, dstDevice)) || (rc = enable_peer(dstDevice, srcDevice))) return rc;
    DHIPCHK(hipSetDevice(srcDevice));
    DHIPCHK(hipMemcpyPeerAsync(dst, dstDevice, src, srcDevice, bytes, nullptr));       // xGMI DMA between two GPUs of the node
    DHIPCHK(hipStreamSynchronize(nullptr));                                             // complete on return, like every call of this ABI
    return GDPT_OK;
}

int gdpt_device_download(int device, void *host, const void *dev, size_t bytes)
{
    if (!host || !dev) return dfail(GDPT_ERR_INVALID, "device_download: null pointer");
    int rc = check_device(device);
    if (rc) return rc;
    DHIPCHK(hipSetDevice(device));
    DHIPCHK(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
    return GDPT_OK;
}

} // extern "C"
