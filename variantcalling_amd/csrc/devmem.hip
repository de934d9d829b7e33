// Device memory and launch bookkeeping of libugvc_mi355x.so.
//
// Every device buffer of the library comes from dev_alloc() (through ensure()).  Two debug modes turn the kernels'
// "the buffer is padded by N bytes" comments into checked claims - both are for tests, neither changes a result:
//
//   UGVC_GUARD=1  every buffer is its own virtual-memory mapping (hipMemAddressReserve + hipMemCreate + hipMemMap) placed
//                 so that the byte after the requested size (rounded up to UGVC_GUARD_ALIGN, default 16) is the first byte
//                 of an UNMAPPED 2 MiB range: any read or write past the end is a GPU memory fault instead of a silent read
//                 of whatever the allocator put next.  UGVC_GUARD=2 puts the unmapped range in FRONT of the first byte.
//   UGVC_POISON=1 every new buffer (and, with UGVC_GUARD, the slack of its mapping) is filled with 0xA5 before first use
//                 (UGVC_POISON=2: 0xFF): a kernel that reads a list entry, a counter or a pad it never wrote computes
//                 on garbage that no box hands out by accident.
//
// Launch breadcrumbs: UGVC_LAUNCH() records the name of every kernel it launches in a small ring; with UGVC_BREADCRUMB=1 a
// SIGABRT handler (the HSA runtime aborts the process on a GPU memory fault) prints the ring before the process dies, and
// with UGVC_DEBUG_SYNC=1 every launch is named on stderr and waited for, so the last name printed IS the faulting kernel.
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <mutex>
#include <unordered_map>

#include "ugvc_device.hpp"

namespace ugvc {

namespace {

struct GuardRec {
    void* va;
    size_t va_size;
    void* map;
    size_t map_size;
    hipMemGenericAllocationHandle_t handle;
};

std::mutex g_mu;
std::unordered_map<void*, GuardRec> g_guard;

int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}

// (read per call, not cached: a test process switches the modes between contexts)
int guard_mode() { return env_int("UGVC_GUARD", 0); }
int poison_mode() { return env_int("UGVC_POISON", 0); }

size_t round_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

constexpr size_t kFence = 2u << 20;       // unmapped bytes on either side of a guarded buffer

int fill(void* p, int byte, size_t n) {
    if (!n) return 0;
    UGVC_HIP(hipMemset(p, byte, n));
    UGVC_HIP(hipDeviceSynchronize());
    return 0;
}

int guard_alloc(void** out, size_t bytes, int mode, int poison) {
    int dev = 0;
    UGVC_HIP(hipGetDevice(&dev));
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof prop);
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    UGVC_HIP(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    if (gran == 0) return fail("UGVC_GUARD: the device reports no virtual-memory allocation granularity");
    static bool said = false;
    if (!said && getenv("UGVC_DEBUG_SYNC")) { said = true; fprintf(stderr, "[ugvc] guard allocations: granularity %zu bytes\n", gran); }
    const size_t fence = round_up(kFence, gran);
    const size_t align = (size_t)std::max(1, env_int("UGVC_GUARD_ALIGN", 16));
    const size_t used = round_up(bytes, align);
    const size_t map_size = round_up(used, gran);
    GuardRec r;
    r.va_size = map_size + 2 * fence;
    r.map_size = map_size;
    hipDeviceptr_t va = nullptr;
    UGVC_HIP(hipMemAddressReserve(&va, r.va_size, std::max(gran, kFence), nullptr, 0));
    r.va = va;
    r.map = static_cast<char*>(r.va) + fence;
    UGVC_HIP(hipMemCreate(&r.handle, map_size, &prop, 0));
    UGVC_HIP(hipMemMap(r.map, map_size, 0, r.handle, 0));
    hipMemAccessDesc acc;
    memset(&acc, 0, sizeof acc);
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    UGVC_HIP(hipMemSetAccess(r.map, map_size, &acc, 1));
    // the slack of the mapping is poisoned whatever UGVC_POISON says: an over-read that stays inside the granule must not
    // see zeros (a zero pad is the one thing the kernels are allowed to rely on - where the library wrote it)
    if (fill(r.map, poison == 2 ? 0xFF : 0xA5, map_size)) return -1;
    void* p = mode == 2 ? r.map : static_cast<char*>(r.map) + (map_size - used);
    {
        std::lock_guard<std::mutex> lk(g_mu);
        g_guard[p] = r;
    }
    *out = p;
    return 0;
}

// ---- launch breadcrumbs --------------------------------------------------------------------------------------------
constexpr int kRing = 8;
const char* volatile g_ring[kRing];
std::atomic<unsigned> g_ring_n{0};
std::atomic<int> g_handler{0};

void put(const char* s) { (void)!write(2, s, strlen(s)); }

void on_abort(int) {
    put("\n[ugvc] process aborted; last kernel launches (oldest first):\n");
    const unsigned n = g_ring_n.load();
    for (unsigned k = n > kRing ? n - kRing : 0; k < n; ++k) {
        const char* s = g_ring[k % kRing];
        put("[ugvc]   ");
        put(s ? s : "?");
        put("\n");
    }
    if (n == 0) put("[ugvc]   (none: the fault is not from a kernel of this library)\n");
    signal(SIGABRT, SIG_DFL);
    raise(SIGABRT);
}

}  // namespace

int dev_alloc(void** out, size_t bytes) {
    const int mode = guard_mode(), poison = poison_mode();
    if (mode) return guard_alloc(out, bytes, mode, poison);
    UGVC_HIP(hipMalloc(out, bytes));
    if (poison && fill(*out, poison == 2 ? 0xFF : 0xA5, bytes)) return -1;
    return 0;
}

void dev_free(void* p) {
    if (!p) return;
    GuardRec r;
    bool guarded = false;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_guard.find(p);
        if (it != g_guard.end()) {
            r = it->second;
            g_guard.erase(it);
            guarded = true;
        }
    }
    if (!guarded) {
        (void)hipFree(p);
        return;
    }
    (void)hipDeviceSynchronize();
    (void)hipMemUnmap(r.map, r.map_size);
    (void)hipMemRelease(r.handle);
    // The address range is NOT given back (hipMemAddressFree): a later reservation that lands on the same addresses sees the
    // old translations on this stack (measured, profiles/r05_guard_va_reuse.txt: a prefix sum over a freshly uploaded buffer
    // returns the previous mapping's bytes; with the ranges kept the same suite is green).  A freed buffer therefore stays
    // unmapped for the life of the process - a use after free faults too.  48 bits of address space outlast any test run.
    if (env_int("UGVC_GUARD_OPTS", 0) & 2) (void)hipMemAddressFree(r.va, r.va_size);
}

bool debug_sync() { return getenv("UGVC_DEBUG_SYNC") != nullptr; }

void launch_note(const char* name) {
    if (g_handler.load() == 0) {
        int expect = 0;
        if (g_handler.compare_exchange_strong(expect, 1) && getenv("UGVC_BREADCRUMB")) signal(SIGABRT, on_abort);
    }
    const unsigned k = g_ring_n.fetch_add(1);
    g_ring[k % kRing] = name;
    if (debug_sync()) {
        fprintf(stderr, "[ugvc] %s done? ", name);
        fflush(stderr);
    }
}

void launch_done(const char* name, hipStream_t stream) {
    if (!debug_sync()) return;                                 // (launch errors stay with the call sites' own hipGetLastError)
    const hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess) fprintf(stderr, "FAILED (%s: %s)\n", name, hipGetErrorString(e));
    else fprintf(stderr, "ok\n");
}

}  // namespace ugvc
