// dev_alloc.hip -- every device buffer the context owns comes from here.
//
// Default: hipMalloc / hipFree.
// Guard placement (option guard_alloc=1 | 2; tests/test_guard_alloc_gpu.py, tools/guard_stress.py): the buffer is put into a mapping of its own,
// made with the HIP virtual-memory API (hipMemAddressReserve / hipMemCreate / hipMemMap), so that
//   1: its LAST byte is the mapping's last byte (the size is rounded up to 16 bytes: what the widest vector access needs), and
//   2: its FIRST byte is the mapping's first byte;
//   3: a multiple of 4 GB lies INSIDE it (a little off its middle), nothing else of the reservation is mapped: an address formed as
//      {high word of the base, low word + offset} -- a 64-bit add that lost its carry -- lands 4 GB low, on an unmapped page.  This is what round 5's
//      intermittent fault was: an inline-asm statement that changes SCC without saying so, put by the scheduler between the s_add_u32 and the
//      s_addc_u32 of the next address; hipMalloc puts a buffer across such a line once in a while, this placement every time.
// the reservation holds one unmapped granule on either side.  A kernel that reads or writes one element past the end (1) or before the start (2) of ANY
// buffer then takes a GPU page fault at that very access -- whatever the allocator's history -- instead of touching a neighbour's pages once in a while:
// the question "is every prefetch, ragged tile and padded row inside its buffer" gets a deterministic answer (round 6: the intermittent fault of round
// 5's fc.0 K-split variant).
#include "dce_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <unordered_map>

namespace dce {

namespace {
struct GuardRec { void* va; size_t reserved; char* map; size_t mapped; hipMemGenericAllocationHandle_t handle; };
std::mutex g_mu;
std::unordered_map<void*, GuardRec> g_recs;
}  // namespace

static bool guard_log() { static const bool on = [] { const char* v = getenv("DCE_GUARD_LOG"); return v && atoi(v) != 0; }(); return on; }

hipError_t dev_alloc_raw(void** out, size_t bytes, int guard)
{
    *out = nullptr;
    if (guard <= 0) {
        const hipError_t e0 = hipMalloc(out, bytes ? bytes : 1);
        if (guard_log()) fprintf(stderr, "dev_alloc plain  %p .. %p (%zu bytes)\n", *out, static_cast<char*>(*out) + bytes, bytes);
        return e0;
    }
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    if ((e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum)) != hipSuccess) return e;
    const size_t need = ((bytes ? bytes : 1) + 15) & ~size_t(15);
    // the mapping starts on a 2 MB boundary of a 2 MB-aligned reservation (DCE_GUARD_ALIGN: another power of two >= the granularity) and is a
    // whole number of granules long: the unmapped granule right behind its last byte is what catches an overrun
    static const size_t align = [] { const char* e = getenv("DCE_GUARD_ALIGN"); const size_t v = e ? strtoull(e, nullptr, 0) : 0; return v ? v : size_t(2) << 20; }();
    const size_t al = align > gran ? align : gran;
    GuardRec r{};
    if (guard == 3) {
        constexpr size_t G4 = size_t(1) << 32;
        const size_t span = (need + gran - 1) / gran * gran + 2 * gran;         // room for the buffer wherever the line falls in it
        r.reserved = G4 + 2 * span;
        if ((e = hipMemAddressReserve(&r.va, r.reserved, gran, nullptr, 0)) != hipSuccess) return e;
        const size_t line = (reinterpret_cast<size_t>(r.va) + span + G4 - 1) / G4 * G4;        // line - span >= va, line + span <= va + reserved
        size_t before = need / 2 / 256 * 256;                                  // bytes of the buffer below the line: half, then off every tile boundary
        if (before + 4352 + 256 <= need) before += 4352;
        const size_t p0 = line - before, m0 = p0 / gran * gran, m1 = (p0 + need + gran - 1) / gran * gran;
        r.mapped = m1 - m0;
        r.map = reinterpret_cast<char*>(m0);
        if ((e = hipMemCreate(&r.handle, r.mapped, &prop, 0)) != hipSuccess) { (void)hipMemAddressFree(r.va, r.reserved); return e; }
        if ((e = hipMemMap(r.map, r.mapped, 0, r.handle, 0)) != hipSuccess) { (void)hipMemRelease(r.handle); (void)hipMemAddressFree(r.va, r.reserved); return e; }
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        if ((e = hipMemSetAccess(r.map, r.mapped, &acc, 1)) != hipSuccess) {
            (void)hipMemUnmap(r.map, r.mapped); (void)hipMemRelease(r.handle); (void)hipMemAddressFree(r.va, r.reserved);
            return e;
        }
        void* p = reinterpret_cast<void*>(p0);
        { std::lock_guard<std::mutex> lk(g_mu); g_recs[p] = r; }
        if (guard_log()) fprintf(stderr, "dev_alloc guard3 %p .. %p (%zu bytes; mapping %p .. %p; 4 GB line %#zx)\n", p, static_cast<char*>(p) + need, bytes, (void*)r.map, (void*)(r.map + r.mapped), line);
        *out = p;
        return hipSuccess;
    }
    r.mapped = (need + gran - 1) / gran * gran;
    r.reserved = (al + r.mapped + gran + al - 1) / al * al;
    if ((e = hipMemAddressReserve(&r.va, r.reserved, al, nullptr, 0)) != hipSuccess) return e;
    if ((e = hipMemCreate(&r.handle, r.mapped, &prop, 0)) != hipSuccess) { (void)hipMemAddressFree(r.va, r.reserved); return e; }
    r.map = static_cast<char*>(r.va) + al;
    if ((e = hipMemMap(r.map, r.mapped, 0, r.handle, 0)) != hipSuccess) { (void)hipMemRelease(r.handle); (void)hipMemAddressFree(r.va, r.reserved); return e; }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if ((e = hipMemSetAccess(r.map, r.mapped, &acc, 1)) != hipSuccess) {
        (void)hipMemUnmap(r.map, r.mapped); (void)hipMemRelease(r.handle); (void)hipMemAddressFree(r.va, r.reserved);
        return e;
    }
    void* p = guard == 2 ? r.map : r.map + (r.mapped - need);
    { std::lock_guard<std::mutex> lk(g_mu); g_recs[p] = r; }
    if (guard_log()) fprintf(stderr, "dev_alloc guard%d %p .. %p (%zu bytes; mapping %p .. %p)\n", guard, p, static_cast<char*>(p) + need, bytes, (void*)r.map, (void*)(r.map + r.mapped));
    *out = p;
    return hipSuccess;
}

hipError_t dev_free_raw(void* p)
{
    if (!p) return hipSuccess;
    GuardRec r{};
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_recs.find(p);
        if (it == g_recs.end()) { r.va = nullptr; }
        else { r = it->second; g_recs.erase(it); }
    }
    if (!r.va) return hipFree(p);
    hipError_t e = hipDeviceSynchronize();                            // (hipFree synchronises too: nothing may still be reading the mapping)
    hipError_t e2 = hipMemUnmap(r.map, r.mapped);
    if (e == hipSuccess) e = e2;
    e2 = hipMemRelease(r.handle);
    if (e == hipSuccess) e = e2;
    // The address range is NOT given back (DCE_GUARD_VA_REUSE=1: it is): a later reservation would get the same addresses, and a new mapping at
    // addresses the GPU has translated before was seen returning the OLD pages' contents to kernels (round 6, ROCm 7.2: wrong logits in a context
    // whose staging buffer had grown, never with hipMalloc; profiles/r6b_guard_trace.txt).  Address space is not a scarce resource in a test process.
    static const bool reuse = [] { const char* v = getenv("DCE_GUARD_VA_REUSE"); return v && atoi(v) != 0; }();
    if (reuse) { e2 = hipMemAddressFree(r.va, r.reserved); if (e == hipSuccess) e = e2; }
    return e;
}

}  // namespace dce
