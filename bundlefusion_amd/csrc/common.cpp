// Error plumbing of the C ABI (thread-local last-error string).
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include <dirent.h>
#include <sched.h>

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "bf_internal.h"

namespace bf {
static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
}
}  // namespace bf

extern "C" {
const char* bf_last_error(void) { return bf::g_last_error.c_str(); }
const char* bf_version(void) { return "bundlefusion_amd 0.1 (gfx950)"; }
// Host<->device copies and a device-wide fence issued by THIS library's HIP runtime, so that callers which hold
// only raw pointers (ctypes, cgo, JNI ...) order their transfers against the kernels launched through this ABI.
int bf_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes) {
    if (bytes == 0) return BF_OK;
    BF_HIP_TRY(hipDeviceSynchronize());
    BF_HIP_TRY(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return BF_OK;
}
int bf_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes) {
    if (bytes == 0) return BF_OK;
    BF_HIP_TRY(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    BF_HIP_TRY(hipDeviceSynchronize());
    return BF_OK;
}
int bf_device_synchronize(void) { BF_HIP_TRY(hipDeviceSynchronize()); return BF_OK; }
// one stream only, and a blocking copy whose direction follows from the pointers (hipMemcpyDefault): what an all-gather callback of a host language needs
// (bf_comm_create_callback) without stopping the other streams of the process
int bf_stream_synchronize(void* hip_stream) { BF_HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream)); return BF_OK; }
int bf_memcpy(void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return BF_OK;
    BF_HIP_TRY(hipMemcpy(dst, src, bytes, hipMemcpyDefault));
    return BF_OK;
}
// CPUs of the NUMA node the HIP device hangs off, as a Linux cpulist ("0-63,128-191"); empty when the platform does not say.
static std::string deviceCpuList(int device) {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) return "";
    for (char* c = bus; *c; ++c) *c = (char)tolower(*c);
    char path[256];
    snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE* f = fopen(path, "r");
    int node = -1;
    if (f) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
    if (node < 0) return "";
    snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    f = fopen(path, "r");
    if (!f) return "";
    char buf[1024] = {0};
    const bool ok = fgets(buf, sizeof buf, f) != nullptr;
    fclose(f);
    std::string s = ok ? buf : "";
    while (!s.empty() && (s.back() == '\n' || s.back() == ' ')) s.pop_back();
    return s;
}

// Host threads that feed a GPU from the other socket pay for it on every launch: on the 2-socket MI355X box the frame loop runs at
// 527 frames/s with its threads on the GPU's NUMA node and at 432 from the other one (kernel times identical, the idle time between
// consecutive launches grows: profiles/r03_numa_modes.md) - the "two performance modes" of round 2 were where the scheduler happened to
// put the process.  bf_bind_host_threads_to_device restricts EVERY thread of the calling process (the HIP runtime's own included) to
// the CPUs of the device's NUMA node, intersected with the calling thread's current mask; threads created afterwards inherit it.
// Opt-in: a library does not change its caller's affinity behind its back (bench.py and the tools call it; BF_BIND_NUMA=0 makes it a no-op).
int bf_bind_host_threads_to_device(int device, char* cpulist_out, size_t cpulist_len) {
    if (cpulist_out && cpulist_len) cpulist_out[0] = 0;
    if (const char* e = getenv("BF_BIND_NUMA")) if (atoi(e) == 0) return BF_OK;
    const std::string list = deviceCpuList(device);
    if (list.empty()) return BF_OK;
    cpu_set_t want, cur, both;
    CPU_ZERO(&want);
    for (size_t i = 0; i < list.size();) {           // "a-b,c,d-e"
        char* end = nullptr;
        const long a = strtol(list.c_str() + i, &end, 10);
        long b = a;
        i = (size_t)(end - list.c_str());
        if (i < list.size() && list[i] == '-') { b = strtol(list.c_str() + i + 1, &end, 10); i = (size_t)(end - list.c_str()); }
        for (long c = a; c <= b && c < CPU_SETSIZE; ++c) CPU_SET((int)c, &want);
        if (i < list.size() && list[i] == ',') ++i; else if (i < list.size()) break;
    }
    if (sched_getaffinity(0, sizeof cur, &cur) != 0) return BF_OK;
    CPU_AND(&both, &want, &cur);
    if (CPU_COUNT(&both) == 0) return BF_OK;           // the process is confined to the other node: nothing to choose from
    DIR* d = opendir("/proc/self/task");
    if (!d) return BF_OK;
    while (dirent* de = readdir(d)) {
        const long tid = strtol(de->d_name, nullptr, 10);
        if (tid > 0) (void)sched_setaffinity((pid_t)tid, sizeof both, &both);
    }
    closedir(d);
    if (cpulist_out && cpulist_len) snprintf(cpulist_out, cpulist_len, "%s", list.c_str());
    return BF_OK;
}

int bf_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { bf::set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return BF_ERR_NO_DEVICE; }
    return n;
}
}
