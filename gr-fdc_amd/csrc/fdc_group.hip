// Multi-device handle of the throughput chain (include/fdc_amd.h, fdc_pipeline_group_*): ONE work() call of the hier block
// (python/FrequencyDomainChannelizer.py:283-315 — one flowgraph block, one scheduler thread) is cut into contiguous spans of
// blocks, one per member device, and the members run their spans concurrently: H2D of the span's samples over that device's
// own PCIe link, the kernels, D2H of the results straight into the caller's per-channel buffers at the span's block offset.
//
// Why spans are independent: the chain's only state is the N/R-sample overlap history
// (lib/overlap_save_impl.h:33) and the window counter, which is (block index * shift) mod R in closed form
// (lib/phase_shifting_windowing_vcc_impl.cc:82).  A span is therefore fully described by its halo — the N/R samples in front
// of it, which lie in the caller's own input buffer for every span but the first — and the global index of its first block.
// The group keeps the stream's history (for the first span of the next call) and the block counter ONCE, on the host.
// No collective, no device-to-device traffic.
//
// The reference's own parallelism inside one work() for comparison: four FFTW threads (python/…:206) and one std::thread per
// segment / per detected channel (lib/activity_detection_channelizer_vcm_impl.cc:293-304, :339-371).
//
// Members may name the same device more than once ("virtual members"): the spans then share one GPU — that is how the
// arithmetic of the dispatcher is tested bit for bit on a one-GPU box (tests/test_group_gpu.py).
#include "../../include/fdc_amd.h"
#include "fdc_guard.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "fdc_hostplace.hpp"

namespace {

#define FDC_ENTRY(name) return fdc::guarded(name, [&]() -> int {
#define FDC_ENTRY_END });

struct SpanJob {
    const void *halo = nullptr, *in = nullptr;
    int64_t first = 0;
    int n = 0;
    void *const *outs = nullptr;
    void *spectrum = nullptr;
    bool real = false;
};

// one worker thread per member beyond the first (the first member's share runs on the calling thread)
struct Worker {
    int node = -1;               // NUMA node of the member's device (-1: unknown — the thread stays where the system puts it)
    int pinned = 0;              // what pin_this_thread_to_node said (1 pinned, 0 left alone, -1 failed)
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    int state = 0;               // 0 idle, 1 job posted, 2 job done, 3 quit
    std::function<int()> job;    // a C-ABI call on the member's handle: returns its status, never throws
    int rc = 0;
    std::string err;
};

int run_span(fdc_pipeline *p, const SpanJob &j)
{
    return j.real ? fdc_pipeline_work_span_real(p, j.halo, j.in, j.first, j.n, j.outs, j.spectrum)
                  : fdc_pipeline_work_span(p, j.halo, j.in, j.first, j.n, j.outs, j.spectrum);
}

void worker_main(Worker *w)
{
    // the thread that feeds member i's device over PCIe runs on that device's NUMA node (its pinned staging is allocated by the HIP runtime
    // from the host pool closest to the device, whatever thread asks: hipHostMalloc without hipHostMallocNumaUser)
    w->pinned = fdc::pin_this_thread_to_node(w->node);
    for (;;) {
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv.wait(lk, [w] { return w->state == 1 || w->state == 3; });
        if (w->state == 3) return;
        std::function<int()> j = std::move(w->job);        // moved, not copied: no allocation on this thread (nothing may throw out of it)
        w->job = nullptr;
        lk.unlock();
        int rc;
        std::string err;
        try {
            rc = j();
            if (rc < 0) err = fdc_last_error();             // this thread's text: handed to the caller
        } catch (...) { rc = FDC_ERR_NOMEM; }
        lk.lock();
        w->rc = rc; w->err.swap(err);
        w->state = 2;
        w->cv.notify_all();
    }
}

// whatever happens on the calling thread, no posted job is left running with the caller's pointers when the call returns
struct Join {
    std::vector<std::unique_ptr<Worker>> &ws;
    std::vector<int> posted;                                 // indices into ws
    explicit Join(std::vector<std::unique_ptr<Worker>> &w) : ws(w) {}
    void post(int wi, std::function<int()> job)
    {
        posted.reserve(ws.size());
        Worker *w = ws[(size_t)wi].get();
        std::lock_guard<std::mutex> lk(w->mu);
        w->job = std::move(job); w->state = 1;
        w->cv.notify_all();
        posted.push_back(wi);
    }
    void wait()
    {
        for (int wi : posted) {
            Worker *w = ws[(size_t)wi].get();
            std::unique_lock<std::mutex> lk(w->mu);
            w->cv.wait(lk, [w] { return w->state == 2; });
        }
    }
    ~Join()
    {
        wait();
        for (int wi : posted) { Worker *w = ws[(size_t)wi].get(); std::lock_guard<std::mutex> lk(w->mu); w->state = 0; }
    }
};

void stop_workers(std::vector<std::unique_ptr<Worker>> &ws)
{
    for (auto &w : ws) {
        if (!w) continue;
        { std::lock_guard<std::mutex> lk(w->mu); w->state = 3; w->cv.notify_all(); }
        if (w->th.joinable()) w->th.join();
    }
}

}  // namespace

struct fdc_pipeline_group {
    std::vector<fdc_pipeline *> mem;
    std::vector<int32_t> dev;
    std::vector<std::unique_ptr<Worker>> workers;          // workers[i - 1] serves member i
    int N = 0, R = 0, ovl = 0, H = 0, C = 0;
    int max_blocks = 0, min_span = 0, member_max = 0;
    bool keep_spectrum = false;
    std::vector<int32_t> lout;
    std::vector<unsigned char> hist;                        // the last N/R samples of the stream so far (zeros at start, overlap_save_impl.cc:52)
    size_t hist_item = 0;                                   // bytes per history sample: 8 (complex) until a real-input call makes it 4
    int64_t blockcount = 0;
    std::vector<std::vector<void *>> outs;                  // per member: the caller's output pointers moved to its span
    std::vector<int64_t> last_first;
    std::vector<int32_t> last_n;
    bool dead = false;                                      // a member failed mid-call: the stream state is no longer defined
};

namespace {

// members that take part in a call of n blocks: every span at least min_span blocks (one member below 2 * min_span)
int members_for(const fdc_pipeline_group *g, int n)
{
    const int k = n / g->min_span;
    return std::max(1, std::min<int>(k, (int)g->mem.size()));
}

void span_of(int n, int k, int i, int *first, int *cnt)      // balanced contiguous spans (gr-fdc_amd/sharding.py:span_for_rank)
{
    const int base = n / k, extra = n % k;
    *cnt = base + (i < extra ? 1 : 0);
    *first = i * base + std::min(i, extra);
}

int group_work(fdc_pipeline_group *g, const void *in, int nblocks, void *const *outs, void *spectrum, bool real)
{
    if (!g) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "null group handle");
    if (g->dead) return fdc::set_error(FDC_ERR_HIP, "the group failed in an earlier call: reset or destroy it");
    if (nblocks < 0) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "negative item count");
    if (nblocks == 0) return 0;
    if (nblocks > g->max_blocks) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "nblocks %d above max_blocks %d", nblocks, g->max_blocks);
    if (!in || (g->C > 0 && !outs)) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "null host buffer");
    // everything a member would refuse is refused here, before any span is posted: an argument error must not leave half a call written
    // (and the group dead)
    for (int c = 0; c < g->C; c++)
        if (!outs[c]) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "null output buffer of channel %d", c);
    if (spectrum && !g->keep_spectrum) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "spectrum output needs keep_spectrum");
    const size_t item = real ? sizeof(float) : 2 * sizeof(float);
    if (g->blockcount == 0) g->hist_item = item;
    if (item != g->hist_item)
        return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "complex and real input calls must not be mixed on one group (reset it first)");
    const unsigned char *hin = static_cast<const unsigned char *>(in);
    const int k = members_for(g, nblocks);
    g->last_first.assign(g->mem.size(), 0);
    g->last_n.assign(g->mem.size(), 0);
    SpanJob job0;
    Join join(g->workers);
    for (int i = 0; i < k; i++) {
        int b0, nb;
        span_of(nblocks, k, i, &b0, &nb);
        SpanJob j;
        // halo: the N/R samples in front of the span.  Span 0: the stream's history; every other span starts at block b0 >= 1 and
        // H >= N/R (R >= 2), so its halo is inside the caller's buffer.
        j.halo = i == 0 ? g->hist.data() : hin + ((size_t)b0 * g->H - (size_t)g->ovl) * item;
        j.in = hin + (size_t)b0 * g->H * item;
        j.first = g->blockcount + b0;
        j.n = nb;
        for (int c = 0; c < g->C; c++)
            g->outs[(size_t)i][(size_t)c] = outs[c] ? static_cast<unsigned char *>(outs[c]) + (size_t)b0 * (size_t)g->lout[(size_t)c] * 8 : nullptr;
        j.outs = g->outs[(size_t)i].data();
        j.spectrum = spectrum ? static_cast<unsigned char *>(spectrum) + (size_t)b0 * (size_t)g->N * 8 : nullptr;
        j.real = real;
        g->last_first[(size_t)i] = j.first; g->last_n[(size_t)i] = nb;
        if (i == 0) { job0 = j; continue; }
        fdc_pipeline *pm = g->mem[(size_t)i];
        join.post(i - 1, [pm, j] { return run_span(pm, j); });
    }
    int rc = run_span(g->mem[0], job0);
    std::string err;
    if (rc < 0) err = fdc_last_error();
    join.wait();                                           // every posted span is waited for, whatever the others returned
    for (int i = 1; i < k; i++) {
        Worker *w = g->workers[(size_t)i - 1].get();
        std::lock_guard<std::mutex> lk(w->mu);
        if (w->rc < 0 && rc >= 0) { rc = w->rc; err = "member " + std::to_string(i) + " (device " + std::to_string(g->dev[(size_t)i]) + "): " + w->err; }
    }
    if (rc < 0) {
        g->dead = true;                                     // some spans of the call were written, others not
        return fdc::set_error(rc, "%s", err.c_str());
    }
    // history <- the last N/R samples of the stream (lib/overlap_save_impl.cc:78); nblocks * H >= N/R
    std::memcpy(g->hist.data(), hin + ((size_t)nblocks * g->H - (size_t)g->ovl) * item, (size_t)g->ovl * item);
    g->blockcount += nblocks;
    return nblocks;
}

}  // namespace

extern "C" {

void fdc_pipeline_group_destroy(fdc_pipeline_group *g)
{
    if (!g) return;
    stop_workers(g->workers);
    for (fdc_pipeline *p : g->mem) fdc_pipeline_destroy(p);
    delete g;
}

int fdc_pipeline_group_create(const fdc_pipeline_cfg *cfg, const int32_t *devices, int ndevices, int min_span_blocks,
                              fdc_pipeline_group **out)
{
    FDC_ENTRY("fdc_pipeline_group_create")
    if (!cfg || !out) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (!devices || ndevices < 1 || ndevices > 64) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "a group has 1 to 64 members");
    if (cfg->max_blocks < 1) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "max_blocks must be >= 1");
    if (min_span_blocks < 0) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "negative min_span_blocks");
    std::unique_ptr<fdc_pipeline_group, void (*)(fdc_pipeline_group *)> g(new fdc_pipeline_group(), fdc_pipeline_group_destroy);
    g->max_blocks = cfg->max_blocks;
    g->keep_spectrum = cfg->keep_spectrum != 0;
    g->min_span = min_span_blocks > 0 ? min_span_blocks : 8;
    g->mem.assign((size_t)ndevices, nullptr);
    g->dev.assign(devices, devices + ndevices);
    // the longest span a member can be given: max over call lengths of ceil(n / members_for(n))
    {
        int worst = 1;
        for (int k = 1; k <= ndevices; k++) {
            // call lengths that use k members: [k * min_span, (k + 1) * min_span) (k = ndevices: up to max_blocks; k = 1: from 1)
            const long long hi = k == ndevices ? cfg->max_blocks : std::min<long long>(cfg->max_blocks, (long long)(k + 1) * g->min_span - 1);
            const long long lo = k == 1 ? 1 : (long long)k * g->min_span;
            if (hi >= lo) worst = std::max<int>(worst, (int)((hi + k - 1) / k));
        }
        g->member_max = worst;
    }
    for (int i = 0; i < ndevices; i++) {
        fdc_pipeline_cfg c = *cfg;
        c.device_id = devices[i];
        c.max_blocks = g->member_max;
        const int rc = fdc_pipeline_create(&c, &g->mem[(size_t)i]);
        if (rc != FDC_OK) {
            const std::string why = fdc_last_error();
            return fdc::set_error(rc, "member %d (device %d): %s", i, devices[i], why.c_str());
        }
    }
    fdc_pipeline *p0 = g->mem[0];
    g->N = cfg->blocklen; g->R = cfg->relinvovl; g->ovl = g->N / g->R; g->H = g->N - g->ovl; g->C = cfg->nchannels;
    for (int c = 0; c < g->C; c++) g->lout.push_back(fdc_pipeline_channel_lout(p0, c));
    g->hist.assign((size_t)g->ovl * 8, 0);
    g->hist_item = 8;
    g->outs.assign((size_t)ndevices, std::vector<void *>((size_t)std::max(1, g->C), nullptr));
    for (int i = 1; i < ndevices; i++) {
        g->workers.emplace_back(new Worker());
        Worker *w = g->workers.back().get();
        w->node = fdc_device_numa_node(devices[i]);
        w->th = std::thread(worker_main, w);
    }
    *out = g.release();
    return FDC_OK;
    FDC_ENTRY_END
}

int fdc_pipeline_group_work(fdc_pipeline_group *g, const void *in, int nblocks, void *const *outs, void *spectrum)
{
    FDC_ENTRY("fdc_pipeline_group_work")
    return group_work(g, in, nblocks, outs, spectrum, false);
    FDC_ENTRY_END
}

int fdc_pipeline_group_work_real(fdc_pipeline_group *g, const void *in, int nblocks, void *const *outs, void *spectrum)
{
    FDC_ENTRY("fdc_pipeline_group_work_real")
    return group_work(g, in, nblocks, outs, spectrum, true);
    FDC_ENTRY_END
}

void fdc_pipeline_group_reset(fdc_pipeline_group *g)
{
    if (!g) return;
    std::fill(g->hist.begin(), g->hist.end(), 0);
    g->blockcount = 0;
    g->hist_item = 8;
    g->dead = false;
}

int32_t fdc_pipeline_group_size(const fdc_pipeline_group *g) { return g ? (int32_t)g->mem.size() : -1; }
fdc_pipeline *fdc_pipeline_group_member(fdc_pipeline_group *g, int i)
{
    return g && i >= 0 && i < (int)g->mem.size() ? g->mem[(size_t)i] : nullptr;
}
int32_t fdc_pipeline_group_device(const fdc_pipeline_group *g, int i)
{
    return g && i >= 0 && i < (int)g->dev.size() ? g->dev[(size_t)i] : -1;
}
int32_t fdc_pipeline_group_member_max_blocks(const fdc_pipeline_group *g) { return g ? g->member_max : -1; }

int fdc_device_numa_node(int device_id)
{
    int n = 0;
    if (device_id < 0 || hipGetDeviceCount(&n) != hipSuccess || device_id >= n) { (void)hipGetLastError(); return -1; }
    char bdf[64] = "";
    if (hipDeviceGetPCIBusId(bdf, (int)sizeof bdf, device_id) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return fdc::pci_numa_node(bdf);
}

int fdc_selftest_worker_placement(int node, int32_t *cpus_of_node, int32_t *cpus_of_worker)
{
    FDC_ENTRY("fdc_selftest_worker_placement")
    // exactly what a group does for a member on `node`: a Worker whose thread pins itself, then reports the mask it runs under
    cpu_set_t want, have, both;
    const bool known = fdc::node_cpuset(node, &want);
    if (sched_getaffinity(0, sizeof have, &have) != 0) return fdc::set_error(FDC_ERR_HIP, "sched_getaffinity failed");
    CPU_AND(&both, &want, &have);
    if (cpus_of_node) *cpus_of_node = known ? CPU_COUNT(&both) : 0;
    std::vector<std::unique_ptr<Worker>> ws;
    ws.emplace_back(new Worker());
    Worker *w = ws[0].get();
    w->node = node;
    w->th = std::thread(worker_main, w);
    cpu_set_t got;
    CPU_ZERO(&got);
    int rcw;
    {
        Join join(ws);
        join.post(0, [&got]() -> int { return sched_getaffinity(0, sizeof got, &got) == 0 ? FDC_OK : FDC_ERR_HIP; });
        join.wait();
        rcw = w->rc;
    }
    const int pinned = w->pinned;
    stop_workers(ws);
    if (rcw != FDC_OK) return fdc::set_error(FDC_ERR_HIP, "the worker could not read its affinity mask");
    if (cpus_of_worker) *cpus_of_worker = CPU_COUNT(&got);
    if (pinned < 0) return fdc::set_error(FDC_ERR_HIP, "pthread_setaffinity_np failed for node %d", node);
    if (pinned == 0) {
        // left alone: the worker must run under the process's own mask
        return CPU_EQUAL(&got, &have) ? 0 : fdc::set_error(FDC_ERR_HIP, "an unpinned worker does not run under the process's mask");
    }
    cpu_set_t inside;
    CPU_AND(&inside, &got, &both);
    if (CPU_COUNT(&got) == 0 || !CPU_EQUAL(&inside, &got))
        return fdc::set_error(FDC_ERR_HIP, "worker for node %d runs on CPUs outside the node", node);
    return 1;
    FDC_ENTRY_END
}

int fdc_pipeline_group_last_spans(const fdc_pipeline_group *g, int64_t *first_block, int32_t *nblocks, int cap)
{
    if (!g) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "null group handle");
    const int n = (int)g->last_n.size();
    for (int i = 0; i < n && i < cap; i++) {
        if (first_block) first_block[i] = g->last_first[(size_t)i];
        if (nblocks) nblocks[i] = g->last_n[(size_t)i];
    }
    return n;
}


/* ---------------------------------------------------------------------------------------------------------------------------------
 * The sink blocks over several devices: the bank cut BY FREQUENCY BAND (include/fdc_amd.h, fdc_sinks_group_*).  Member i holds a run
 * of the bank's PowerActivationChannels and a run of its segments (runs of the bank order, equal counts), copies only the bins that
 * run reads, and is an ordinary fdc_sinks bank on its device: no state is shared between members.
 * ------------------------------------------------------------------------------------------------------------------------------- */
}  // extern "C"

struct fdc_sinks_group {
    struct Member { fdc_sinks *s = nullptr; int32_t dev = 0, lo = 0, hi = 0, npac = 0, nseg = 0; std::vector<int32_t> bank_pos; /* member's PAC i -> its place in cfg->pac[] */ };
    std::vector<Member> mem;
    std::vector<std::unique_ptr<Worker>> workers;          // workers[i] serves member i (member 0 runs on the calling thread)
    int N = 0, max_blocks = 0;
    std::vector<fdc_pdu> pdus;                              // of the last call, merged into one bank's emission order
    bool dead = false;
};

extern "C" {

void fdc_sinks_group_destroy(fdc_sinks_group *g)
{
    if (!g) return;
    stop_workers(g->workers);
    for (auto &m : g->mem) fdc_sinks_destroy(m.s);
    delete g;
}

int fdc_sinks_group_create(const fdc_sinks_cfg *cfg, const int32_t *devices, int ndevices, fdc_sinks_group **out)
{
    FDC_ENTRY("fdc_sinks_group_create")
    if (!cfg || !out) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (!devices || ndevices < 1 || ndevices > 64) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "a group has 1 to 64 members");
    if (cfg->npac < 0 || cfg->nseg < 0 || (cfg->npac && !cfg->pac) || (cfg->nseg && !cfg->seg)) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "bad sink lists");
    if (cfg->npac + cfg->nseg == 0) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "an empty bank");
    std::unique_ptr<fdc_sinks_group, void (*)(fdc_sinks_group *)> g(new fdc_sinks_group(), fdc_sinks_group_destroy);
    g->N = cfg->blocklen; g->max_blocks = cfg->max_blocks;
    g->mem.resize((size_t)ndevices);
    auto run_of = [ndevices](int n, int i, int *first, int *cnt) {       // balanced runs of the bank order
        const int base = n / ndevices, extra = n % ndevices;
        *cnt = base + (i < extra ? 1 : 0);
        *first = i * base + std::min(i, extra);
    };
    // the PowerActivationChannels in frequency order (stable: equal centres keep the bank's order), so that a member's run is a band
    std::vector<int32_t> byfreq((size_t)cfg->npac);
    for (int i = 0; i < cfg->npac; i++) byfreq[(size_t)i] = i;
    std::stable_sort(byfreq.begin(), byfreq.end(), [cfg](int32_t a, int32_t b) { return cfg->pac[a].cfreq < cfg->pac[b].cfreq; });
    std::vector<fdc_pac_cfg> sorted((size_t)cfg->npac);
    for (int i = 0; i < cfg->npac; i++) sorted[(size_t)i] = cfg->pac[byfreq[(size_t)i]];
    for (int i = 0; i < ndevices; i++) {
        auto &m = g->mem[(size_t)i];
        m.dev = devices[i];
        int p0, pn, s0, sn;
        run_of(cfg->npac, i, &p0, &pn);
        run_of(cfg->nseg, i, &s0, &sn);
        m.npac = pn; m.nseg = sn;
        m.bank_pos.assign(byfreq.begin() + p0, byfreq.begin() + p0 + pn);
        if (pn + sn == 0) continue;                          // more devices than channels and segments: the member stays idle
        fdc_sinks_cfg c = *cfg;
        c.device_id = devices[i];
        c.npac = pn; c.pac = pn ? sorted.data() + p0 : nullptr;
        c.nseg = sn; c.seg = sn ? cfg->seg + s0 : nullptr;
        c.seg_id_base = cfg->seg_id_base + s0;
        const int rc = fdc_sinks_create(&c, &m.s);
        if (rc != FDC_OK) {
            const std::string why = fdc_last_error();
            return fdc::set_error(rc, "member %d (device %d): %s", i, devices[i], why.c_str());
        }
        fdc_sinks_read_band(m.s, &m.lo, &m.hi);
    }
    for (int i = 0; i < ndevices; i++) {
        g->workers.emplace_back(new Worker());
        g->workers.back()->node = fdc_device_numa_node(g->mem[(size_t)i].dev);
        if (i > 0 && g->mem[(size_t)i].s) g->workers.back()->th = std::thread(worker_main, g->workers.back().get());
    }
    *out = g.release();
    return FDC_OK;
    FDC_ENTRY_END
}

int fdc_sinks_group_work(fdc_sinks_group *g, const void *spectrum, int nitems)
{
    FDC_ENTRY("fdc_sinks_group_work")
    if (!g) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "null group handle");
    if (g->dead) return fdc::set_error(FDC_ERR_HIP, "the group failed in an earlier call and must be destroyed");
    if (nitems < 0 || nitems > g->max_blocks) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "nitems %d outside [0, max_blocks]", nitems);
    g->pdus.clear();
    if (nitems == 0) return 0;
    if (!spectrum) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "null buffer");
    int rc = nitems;
    std::string err;
    // From the first posted job on the members' state machines may stand at different items: the group is dead unless the whole call goes
    // through — also when this thread throws in between (a thread that does not start, an allocation): the guard outlives the exception.
    struct DeadUnlessDone { fdc_sinks_group *g; bool armed = false, done = false; ~DeadUnlessDone() { if (armed && !done) g->dead = true; } } guard{g};
    {
        Join join(g->workers);
        int first = -1;
        for (int i = 0; i < (int)g->mem.size(); i++) {
            auto &m = g->mem[(size_t)i];
            if (!m.s) continue;
            if (first < 0) { first = i; continue; }         // the first member with work runs on this thread
            fdc_sinks *sm = m.s;
            const int32_t lo = m.lo, hi = m.hi;
            if (!g->workers[(size_t)i]->th.joinable()) g->workers[(size_t)i]->th = std::thread(worker_main, g->workers[(size_t)i].get());
            guard.armed = true;
            join.post(i, [sm, spectrum, nitems, lo, hi] { return fdc_sinks_work_band(sm, spectrum, nitems, lo, hi); });
        }
        if (first >= 0) {
            auto &m = g->mem[(size_t)first];
            guard.armed = true;
            const int r0 = fdc_sinks_work_band(m.s, spectrum, nitems, m.lo, m.hi);
            if (r0 < 0) { rc = r0; err = "member " + std::to_string(first) + " (device " + std::to_string(m.dev) + "): " + fdc_last_error(); }
        }
        join.wait();
        for (int wi : join.posted) {
            Worker *w = g->workers[(size_t)wi].get();
            std::lock_guard<std::mutex> lk(w->mu);
            if (w->rc < 0 && rc >= 0) { rc = w->rc; err = "member " + std::to_string(wi) + " (device " + std::to_string(g->mem[(size_t)wi].dev) + "): " + w->err; }
        }
    }
    if (rc < 0) {
        g->dead = true;                                      // the members' state machines are no longer at the same item
        return fdc::set_error(rc, "%s", err.c_str());
    }
    // Merge: one bank emits, item by item, the PowerActivationChannels in bank order and then the segments in order.  A member holds a run of
    // the channels in FREQUENCY order: a PowerActivationChannel PDU is put back by the place of its channel in the bank's own list (bank_pos);
    // the segments are runs of the bank order: member 0's detections, member 1's, ...  Inside one channel / one member's detections the PDUs are
    // in emission order already: the sort is stable on (item, kind, place).
    struct Key { int32_t item, kind, member; int32_t from, idx; };      // from / idx: which member's list, which entry (no packing: any member or PDU count)
    std::vector<Key> keys;
    std::vector<std::vector<fdc_pdu>> got(g->mem.size());
    std::vector<int32_t> items, pacs;
    for (int i = 0; i < (int)g->mem.size(); i++) {
        fdc_sinks *sm = g->mem[(size_t)i].s;
        if (!sm) continue;
        const int n = fdc_sinks_pdu_count(sm);
        if (n <= 0) continue;
        got[(size_t)i].resize((size_t)n);
        items.resize((size_t)n); pacs.resize((size_t)n);
        fdc_sinks_pdus(sm, got[(size_t)i].data(), n);
        fdc_sinks_pdu_emit_order(sm, items.data(), pacs.data(), n);
        const auto &bp = g->mem[(size_t)i].bank_pos;
        for (int k = 0; k < n; k++) {
            const int32_t kind = got[(size_t)i][(size_t)k].kind, pc = pacs[(size_t)k];
            // place: a PowerActivationChannel's index in cfg->pac[]; a detection's member (members own runs of the segment list)
            const int32_t place = (kind == 0 && pc >= 0 && (size_t)pc < bp.size()) ? bp[(size_t)pc] : i;
            keys.push_back(Key{items[(size_t)k], kind, place, i, k});
        }
    }
    std::stable_sort(keys.begin(), keys.end(), [](const Key &a, const Key &b) {
        if (a.item != b.item) return a.item < b.item;
        if (a.kind != b.kind) return a.kind < b.kind;
        return a.member < b.member;
    });
    g->pdus.reserve(keys.size());
    for (const Key &k : keys) g->pdus.push_back(got[(size_t)k.from][(size_t)k.idx]);
    guard.done = true;
    return nitems;
    FDC_ENTRY_END
}

int fdc_sinks_group_pdu_count(const fdc_sinks_group *g) { return g ? (int)g->pdus.size() : 0; }

int fdc_sinks_group_pdus(const fdc_sinks_group *g, fdc_pdu *out, int cap)
{
    if (!g || (!out && cap > 0)) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "null argument");
    const int n = (int)g->pdus.size();
    for (int i = 0; i < n && i < cap; i++) out[i] = g->pdus[(size_t)i];
    return n;
}

int32_t fdc_sinks_group_size(const fdc_sinks_group *g) { return g ? (int32_t)g->mem.size() : -1; }
fdc_sinks *fdc_sinks_group_member(fdc_sinks_group *g, int i) { return g && i >= 0 && i < (int)g->mem.size() ? g->mem[(size_t)i].s : nullptr; }

int fdc_sinks_group_member_info(const fdc_sinks_group *g, int i, int32_t *device, int32_t *lo, int32_t *hi, int32_t *npac, int32_t *nseg)
{
    if (!g || i < 0 || i >= (int)g->mem.size()) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "no such member");
    const auto &m = g->mem[(size_t)i];
    if (device) *device = m.dev;
    if (lo) *lo = m.lo;
    if (hi) *hi = m.hi;
    if (npac) *npac = m.npac;
    if (nseg) *nseg = m.nseg;
    return FDC_OK;
}

}  // extern "C"
