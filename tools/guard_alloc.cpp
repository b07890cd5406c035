// PyTorch pluggable allocator for hunting out-of-bounds reads (tools/guard_probe.py): every tensor gets its OWN hipMalloc whose size is
// the request rounded up to 4 KB, and the tensor is placed so that it ENDS at the end of that allocation (start 256 B-aligned).  A
// kernel that reads past the end of a buffer then runs into whatever follows the allocation -- with luck an unmapped page -> a fault
// at the culprit instead of at a random later point.  Build: hipcc -shared -fPIC tools/guard_alloc.cpp -o tools/_bin/libguard_alloc.so
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <unordered_map>

static std::mutex g_mu;
static std::unordered_map<void *, void *> g_base;      // user pointer -> hipMalloc base

extern "C" void *guard_malloc(ssize_t size, int device, hipStream_t stream)
{
    (void)stream;
    if (size <= 0) size = 1;
    hipSetDevice(device);
    const size_t page = 4096;
    const size_t alloc = ((size_t)size + 255 + page - 1) / page * page;
    void *base = nullptr;
    if (hipMalloc(&base, alloc) != hipSuccess) return nullptr;
    uintptr_t user = ((uintptr_t)base + alloc - (size_t)size) & ~(uintptr_t)255;
    std::lock_guard<std::mutex> lk(g_mu);
    g_base[(void *)user] = base;
    return (void *)user;
}

extern "C" void guard_free(void *ptr, ssize_t size, int device, hipStream_t stream)
{
    (void)size; (void)device; (void)stream;
    void *base = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_base.find(ptr);
        if (it == g_base.end()) return;
        base = it->second;
        g_base.erase(it);
    }
    hipDeviceSynchronize();
    hipFree(base);
}
