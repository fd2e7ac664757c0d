// cvs_state.cpp -- the memory behind a handle's state planes: one plain hipMalloc block, plus the handle's slot of tile queues
// for the dynamic launch order.  (Rounds 2-4 also carried an allocation-time placement search here -- per-plane physical
// allocations mapped back to back, a streaming probe slid over them -- which the row-interleaved state layout of round 4 made
// unnecessary; removed in round 5, history in profiles/HISTORY_rounds_1_3.md.)
#include <hip/hip_runtime_api.h>

#include <mutex>
#include <vector>

#include "cvs_context.h"

namespace cvs {

// ---- tile-queue slots (dynamic launch order) ----
namespace {
constexpr int kCtrSlots = 512;
constexpr size_t kCtrSlotBytes = 2048;   // two sets of queues, 1 KiB each (cvs_kernels_basis.hip kQueueSetUints)
struct CtrSlab {
    unsigned char* base = nullptr;
    std::vector<int> free_slots;
    std::vector<char> used;      // the slot has been handed out before
    bool failed = false;
};
std::mutex g_ctr_mutex;
CtrSlab g_ctr[64];

// Allocations and blocking memsets are "potentially unsafe" while any stream of the process is under global-mode capture: the
// guard marks them as belonging to no capture (ensure_state may run while the caller's stream, or another thread's, captures).
unsigned* tile_ctr_alloc(int device, bool* recycled)
{
    *recycled = false;
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(g_ctr_mutex);
    RelaxedCapture relaxed;
    CtrSlab& sl = g_ctr[device];
    if (!sl.base && !sl.failed) {
        // once per device and process: 1 MiB, zeroed (the kernels keep the invariant "all zero between launches" themselves)
        void* p = nullptr;
        if (hipMalloc(&p, kCtrSlots * kCtrSlotBytes) != hipSuccess || hipMemset(p, 0, kCtrSlots * kCtrSlotBytes) != hipSuccess) {
            (void)hipGetLastError();
            if (p) (void)hipFree(p);
            sl.failed = true;
            return nullptr;
        }
        sl.base = static_cast<unsigned char*>(p);
        for (int i = kCtrSlots - 1; i >= 0; --i) sl.free_slots.push_back(i);
        sl.used.assign(kCtrSlots, 0);
    }
    if (!sl.base || sl.free_slots.empty()) return nullptr;   // no slot: the handle keeps to the static orders
    const int i = sl.free_slots.back();
    unsigned* p = reinterpret_cast<unsigned*>(sl.base + (size_t)i * kCtrSlotBytes);
    // a slot that has served another block: the set its last launch took its tickets from is not at zero.  NOT zeroed with hipMemset here:
    // a fill's stores sit in an XCD's L2 while the queue atomics execute at the memory side (the round-5 finding behind k_reset_queues);
    // the new owner's first dynamic launch puts a k_reset_queues launch in front of itself on its own stream (StateBlock::ctr_parity bit 1).
    *recycled = sl.used[i] != 0;
    sl.free_slots.pop_back();
    sl.used[i] = 1;
    return p;
}

void tile_ctr_free(int device, unsigned* p)
{
    if (!p || device < 0 || device >= 64) return;
    std::lock_guard<std::mutex> lock(g_ctr_mutex);
    CtrSlab& sl = g_ctr[device];
    if (sl.base) sl.free_slots.push_back((int)((reinterpret_cast<unsigned char*>(p) - sl.base) / kCtrSlotBytes));
}
}  // namespace

void state_block_free(StateBlock& b)
{
    if (!b.base) return;
    if (b.ready) {   // work of the handle that parked the block may still be running
        (void)hipEventSynchronize(b.ready);
        (void)hipEventDestroy(b.ready);
        b.ready = nullptr;
    }
    tile_ctr_free(b.device, b.tile_ctr);
    (void)hipFree(b.base);
    (void)hipGetLastError();
    b = StateBlock();
}

hipError_t state_block_alloc(int device, size_t elems, StateBlock& b)
{
    b = StateBlock();
    b.device = device;
    RelaxedCapture relaxed;   // the allocation belongs to no capture (another thread's stream may be under global-mode capture)
    const hipError_t e = hipMalloc(&b.base, elems * sizeof(float));
    if (e == hipSuccess) {
        b.elems = elems;
        bool recycled = false;
        b.tile_ctr = tile_ctr_alloc(device, &recycled);
        b.ctr_parity = recycled ? 2 : 0;
    }
    return e;
}

}  // namespace cvs
