// Launch chains: a whole block's forward (or backward) -- conv, conv, InstanceNorm / CBAM ... -- enqueued from C by ONE call.
//
// Round 2 measured the host, not the GPU, as the bound of the small-batch configurations: 730 launches per training step,
// each its own Python -> ctypes -> C transition, and ~250 Python autograd nodes (13 ms of enqueue time per 19 ms step; the
// bf16 step at 32 bars and the GAN iteration at 16 bars entirely host-bound).  HIP-graph replay does not help on ROCm 7.2
// (measured in round 3: a captured single-stream step replays 2.4 ms SLOWER than the same launches enqueued eagerly, a
// captured four-stream step 11 ms slower).  A chain is the smaller hammer: the host layer prepares, once per block and
// geometry, the list of entry points of include/mgvae.h that make up the block's forward / backward with their arguments
// as 8-byte words; per step it patches the tensor addresses in and calls mgvae_chain_run, which is a plain loop over the
// SAME entry points the eager path calls one by one (same kernels, same tuner decisions, same results, same stream order).
#include <map>
#include <mutex>
#include <string.h>
#include "mgvae_common.h"

static inline float mgvae_chain_f32(uint64_t w) { uint32_t u = (uint32_t)w; float f; memcpy(&f, &u, 4); return f; }
static inline double mgvae_chain_f64(uint64_t w) { double d; memcpy(&d, &w, 8); return d; }

#include "chain_dispatch.inc"

extern "C" int mgvae_chain_fn_count(void) { return MGVAE_CHAIN_NFN; }
extern "C" int mgvae_chain_fn_id(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < MGVAE_CHAIN_NFN; ++i)
        if (strcmp(name, MGVAE_CHAIN_NAMES[i]) == 0) return i;
    return -1;
}
extern "C" int mgvae_chain_run(const MgvaeChainCall* calls, int ncalls, const uint64_t* words, int* failed) {
    if (failed) *failed = -1;
    if (ncalls < 0 || (ncalls && (!calls || !words))) return MGVAE_EINVAL;
    for (int i = 0; i < ncalls; ++i) {
        const MgvaeChainCall& c = calls[i];
        int rc = MGVAE_EINVAL;
        if (c.fn >= 0 && c.fn < MGVAE_CHAIN_NFN && c.nargs == MGVAE_CHAIN_NARGS[c.fn] && c.first >= 0)
            rc = mgvae_chain_dispatch(c.fn, words + c.first);
        if (rc != MGVAE_OK) {
            if (failed) *failed = i;
            return rc;
        }
    }
    return MGVAE_OK;
}

// `to` waits for everything enqueued on `from` so far (one event per ordered pair of streams, re-recorded every time: a
// later record does not disturb a wait that was already enqueued)
static std::mutex g_fork_mu;
static std::map<std::pair<hipStream_t, hipStream_t>, hipEvent_t> g_fork_ev;
extern "C" int mgvae_stream_fork(void* from, void* to) {
    hipStream_t a = as_stream(from), b = as_stream(to);
    if (a == b) return MGVAE_OK;
    hipEvent_t ev;
    {
        std::lock_guard<std::mutex> lk(g_fork_mu);
        auto it = g_fork_ev.find({a, b});
        if (it == g_fork_ev.end()) {
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) return MGVAE_ELAUNCH;
            g_fork_ev[{a, b}] = ev;
        } else {
            ev = it->second;
        }
    }
    if (hipEventRecord(ev, a) != hipSuccess) return MGVAE_ELAUNCH;
    if (hipStreamWaitEvent(b, ev, 0) != hipSuccess) return MGVAE_ELAUNCH;
    return MGVAE_OK;
}
