// Frame feed from host memory: pinned staging ring + copy stream.
//
// The reference reads the video ahead of the frame loop on a thread of its own and lets the OpenMP
// loop wait per frame (cpp/exec/psp_process.cpp:867-1007, "input_frame_offset_ready").  Here the
// equivalent is a ring of pinned host slots, each with a device twin: the caller fills a slot (the
// video reader reads the file straight into it), the slot is uploaded by hipMemcpyAsync on the
// feed's own copy stream, and the consumer stream (12-bit unpack + upsp_pipeline_process) waits on
// the slot's event only -- the upload of chunk k + 1 runs while chunk k is processed, the host
// blocks only when every slot is in flight.  PCIe Gen5 x16 moves ~55-60 GB/s: ~38 k packed 12-bit
// frames/s (1.5 MiB per 1-Mpix frame), far below the engine's ~1 M resident frames/s, so the feed
// rate is reported separately from the headline (bench.py `host_feed`).
#include <hip/hip_runtime.h>

#include <vector>

#include "upsp_gpu.h"
#include "upsp_internal.h"

using namespace upsp;

struct upsp_feed {
    size_t slot_bytes = 0;
    int nslots = 0;
    hipStream_t copy = nullptr;
    std::vector<void *> h_slot, d_slot;
    std::vector<hipEvent_t> ready, freed;     // upload done / consumer done with the device twin
    std::vector<int> state;                   // 0 free, 1 acquired (host fills it), 2 committed
    int next = 0;
};

extern "C" {

int upsp_feed_create(size_t slot_bytes, int nslots, upsp_feed **out)
{
    if (!out) return fail(UPSP_ERR_INVALID, "out is null");
    *out = nullptr;
    if (slot_bytes == 0 || nslots < 1 || nslots > 64) return fail(UPSP_ERR_INVALID, "bad slot size / count");
    upsp_feed *f = new upsp_feed();
    f->slot_bytes = slot_bytes;
    f->nslots = nslots;
    hipError_t e = hipStreamCreateWithFlags(&f->copy, hipStreamNonBlocking);
    for (int i = 0; i < nslots && e == hipSuccess; ++i) {
        void *h = nullptr, *d = nullptr;
        hipEvent_t a = nullptr, b = nullptr;
        e = hipHostMalloc(&h, slot_bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc(&d, slot_bytes);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&a, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&b, hipEventDisableTiming);
        f->h_slot.push_back(h);
        f->d_slot.push_back(d);
        f->ready.push_back(a);
        f->freed.push_back(b);
        f->state.push_back(0);
    }
    if (e != hipSuccess) {
        upsp_feed_destroy(f);
        return fail(UPSP_ERR_HIP, std::string("feed alloc: ") + hipGetErrorString(e));
    }
    *out = f;
    return UPSP_OK;
}

void upsp_feed_destroy(upsp_feed *f)
{
    if (!f) return;
    if (f->copy) (void)hipStreamSynchronize(f->copy);
    for (size_t i = 0; i < f->h_slot.size(); ++i) {
        if (f->freed[i]) (void)hipEventSynchronize(f->freed[i]);
        if (f->h_slot[i]) (void)hipHostFree(f->h_slot[i]);
        if (f->d_slot[i]) (void)hipFree(f->d_slot[i]);
        if (f->ready[i]) (void)hipEventDestroy(f->ready[i]);
        if (f->freed[i]) (void)hipEventDestroy(f->freed[i]);
    }
    if (f->copy) (void)hipStreamDestroy(f->copy);
    delete f;
}

int upsp_feed_acquire(upsp_feed *f, int *slot, void **h_ptr)
{
    if (!f || !slot || !h_ptr) return fail(UPSP_ERR_INVALID, "bad argument");
    const int i = f->next;
    if (f->state[i] == 1) return fail(UPSP_ERR_INVALID, "feed: the next slot is still being filled (commit it first)");
    if (f->state[i] == 2) return fail(UPSP_ERR_INVALID, "feed: every slot is committed and not released (release in order)");
    // the slot's previous round: upload finished (pinned buffer reusable) and the consumer is done
    // with the device twin.  Events that were never recorded complete at once.
    UPSP_HIP_CHECK(hipEventSynchronize(f->ready[i]));
    UPSP_HIP_CHECK(hipEventSynchronize(f->freed[i]));
    f->state[i] = 1;
    f->next = (i + 1) % f->nslots;
    *slot = i;
    *h_ptr = f->h_slot[i];
    return UPSP_OK;
}

int upsp_feed_commit(upsp_feed *f, int slot, size_t nbytes, void *consumer_stream, void **d_ptr)
{
    if (!f || slot < 0 || slot >= f->nslots || !d_ptr) return fail(UPSP_ERR_INVALID, "bad argument");
    if (f->state[slot] != 1) return fail(UPSP_ERR_INVALID, "feed: slot was not acquired");
    if (nbytes > f->slot_bytes) return fail(UPSP_ERR_INVALID, "feed: more bytes than the slot holds");
    if (nbytes) UPSP_HIP_CHECK(hipMemcpyAsync(f->d_slot[slot], f->h_slot[slot], nbytes, hipMemcpyHostToDevice, f->copy));
    UPSP_HIP_CHECK(hipEventRecord(f->ready[slot], f->copy));
    UPSP_HIP_CHECK(hipStreamWaitEvent((hipStream_t)consumer_stream, f->ready[slot], 0));
    f->state[slot] = 2;
    *d_ptr = f->d_slot[slot];
    return UPSP_OK;
}

int upsp_feed_abort(upsp_feed *f, int slot)
{
    if (!f || slot < 0 || slot >= f->nslots) return fail(UPSP_ERR_INVALID, "bad argument");
    if (f->state[slot] != 1) return fail(UPSP_ERR_INVALID, "feed: slot was not acquired");
    // nothing was uploaded and nobody waits on it: the slot is simply free again and is the next one handed out
    f->state[slot] = 0;
    f->next = slot;
    return UPSP_OK;
}

int upsp_feed_release(upsp_feed *f, int slot, void *consumer_stream)
{
    if (!f || slot < 0 || slot >= f->nslots) return fail(UPSP_ERR_INVALID, "bad argument");
    if (f->state[slot] != 2) return fail(UPSP_ERR_INVALID, "feed: slot was not committed");
    UPSP_HIP_CHECK(hipEventRecord(f->freed[slot], (hipStream_t)consumer_stream));
    f->state[slot] = 0;
    return UPSP_OK;
}

}  // extern "C"
