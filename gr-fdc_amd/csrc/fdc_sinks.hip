// Stateful sinks fed with the normalised spectrum: a bank of PowerActivationChannel instances and the segments of an
// activity_detection_channelizer_vcm, sharing one device-resident spectrum (the hier block feeds all of them from the
// same normalize_input output, python/FrequencyDomainChannelizer.py:301-312).
//
// Split of work: the data-parallel parts run on the GPU for a whole batch of blocks —
// per-(block, cell) power sums (k_cell_power) and the window * half-swap * IFFT * discard extractions (k_extract,
// grouped by width) — while the per-block decision logic, which is inherently sequential over blocks and tiny
// (a few hundred floats per block), runs on the host between the two GPU phases:
//     GPU power cells -> D2H -> host state machines (one pass over the batch, emits an extraction task list and PDU
//     records that reference tasks) -> GPU extractions -> D2H -> payload assembly.
// Behaviour follows lib/PowerActivationChannel_impl.cc and lib/activity_detection_channelizer_vcm_impl.cc; line
// references are given at each decision.  The `threads` flag of the reference is accepted and ignored (GPU batching
// replaces the per-channel std::thread fan-out).
#include "../../include/fdc_amd.h"
#include "fdc_kernels.h"
#include "fdc_sinks_dev.h"
#include "fdc_guard.hpp"

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <complex>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <memory>
#include <string>
#include <thread>
#include <vector>

extern "C" const char *fdc_last_error(void);
namespace fdc { int set_error(int code, const char *fmt, ...); int pick_device(int device_id); const char *debug_env(const char *name); }

namespace {

using cfl = std::complex<float>;

#define HIPCHK(expr)                                                                                              \
    do {                                                                                                          \
        hipError_t _e = (expr);                                                                                   \
        if (_e != hipSuccess) return fdc::set_error(FDC_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// body of an extern "C" entry: nothing thrown inside crosses the C boundary (fdc_guard.hpp)
#define FDC_ENTRY(name) return fdc::guarded(name, [&]() -> int {
#define FDC_ENTRY_END });

// the bank failed mid-call: remember why, refuse everything that follows
#define FDC_DEAD_CHECK(s)                                                                                                     \
    do {                                                                                                                      \
        if ((s)->poisoned)                                                                                                    \
            return fdc::set_error(FDC_ERR_HIP, "the bank failed in an earlier call and must be destroyed (%s)", (s)->poison_why.c_str()); \
    } while (0)

int pow2ceil(int k) { return (int)std::pow(2.0, std::ceil(std::log2((double)k))); }
bool ispow2i(int k) { return k > 0 && (k & (k - 1)) == 0; }

// A buffered output block of a channel: either still on the device as the result of a task of the current call, or a
// host copy carried over from an earlier call.
struct BlockRef {
    int64_t task = -1;
    std::vector<cfl> owned;
};

struct PduRec {
    int64_t key = 0;                // emission order inside a call: block, then PowerActivationChannels in order, then detections
    fdc_pdu meta{};
    std::vector<BlockRef> blocks;
    int blocklen = 0;               // samples per block
    std::vector<cfl> payload;
};

struct Pac {
    int ID = 0, extract_start = 0, extract_stop = 0, extract_width = 0, output_len = 0, ovl_offset = 0;
    int measure_start = 0, measure_stop = 0, deltaphase = 0, win_off = 0, cell = 0;
    bool active = false;
    float lastpower = FLT_MAX;
    int count = 0, phase = 0, part = 0, finished = 0, id_at_activation = 0;
    std::string msg_id;                  // create_ID() at activation, :308-312
    std::vector<BlockRef> blocks;        // handed to the PDU as a whole when it is emitted
};

struct DetChan {
    int ID, detect_start, detect_stop, extract_start, extract_stop, extract_width, wclass, ovlskip, outputsamples;
    int count, phase, phaseincrement, inactive, part;
    std::string msg_id;                  // get_ID_for_msg() at activation, …vcm_impl.cc:526-530
    std::deque<BlockRef> data;
};

struct Segment {
    int ID = 0, start = 0, stop = 0, width = 0, ncell = 0, cell0 = 0, counter = 0;
    std::deque<DetChan> chans;
};


// What a worker thread collects while it runs its range of PowerActivationChannels over a batch.  Kept from call to call:
// the lists keep their capacity (no page faults on fresh heap memory in every call).
struct WorkerLists {
    std::vector<fdc::ExtractTask> tasks;
    std::vector<int> w, skip;
    int64_t used = 0;
    std::vector<PduRec> pdus;
    void clear() { tasks.clear(); w.clear(); skip.clear(); used = 0; pdus.clear(); }
};

// Fork-join pool of the handle (threads are made once; a batch costs two condition-variable round trips instead of a
// thread creation per worker).
class WorkerPool {
public:
    ~WorkerPool() { stop(); }
    // false: a job threw (std::bad_alloc from a growing list, normally): the batch is lost, the process is not
    bool run(int n, const std::function<void(int)> &fn)
    {
        if ((int)th_.size() < n) grow(n);
        {
            std::lock_guard<std::mutex> g(m_);
            job_ = &fn; njob_ = n; pending_ = n; gen_++; failed_ = false;
        }
        cv_.notify_all();
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [&] { return pending_ == 0; });
        job_ = nullptr;
        return !failed_;
    }
    void stop()
    {
        {
            std::lock_guard<std::mutex> g(m_);
            quit_ = true;
        }
        cv_.notify_all();
        for (auto &t : th_) t.join();
        th_.clear();
    }
private:
    void grow(int n)
    {
        for (int i = (int)th_.size(); i < n; i++)
            th_.emplace_back([this, i] {
                uint64_t seen = 0;
                for (;;) {
                    const std::function<void(int)> *fn = nullptr;
                    {
                        std::unique_lock<std::mutex> lk(m_);
                        cv_.wait(lk, [&] { return quit_ || (gen_ != seen && i < njob_); });
                        if (quit_) return;
                        seen = gen_; fn = job_;
                    }
                    bool ok = true;
                    try { (*fn)(i); } catch (...) { ok = false; }     // nothing may unwind out of a worker thread
                    {
                        std::lock_guard<std::mutex> g(m_);
                        if (!ok) failed_ = true;
                        pending_--;
                    }
                    done_.notify_one();
                }
            });
    }
    std::vector<std::thread> th_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    const std::function<void(int)> *job_ = nullptr;
    int njob_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
    bool quit_ = false, failed_ = false;
};

}  // namespace

struct fdc_sinks {
    fdc_sinks_cfg cfg{};
    int N = 0, R = 0, dec = 1;
    float pac_thr = 0.f, det_thr = 0.f;
    std::vector<Pac> pacs;
    std::vector<Segment> segs;
    std::vector<int> det_win_off;            // per width class
    std::vector<fdc::PowerCell> cells;
    int64_t blockcount = 1;                  // both reference blocks start counting at 1 (hist is block 0)
    // A work / submit / flush call that fails after it has begun to advance the bank's state (block counter, channel state on the
    // device, buffered blocks, the two-deep pipeline) cannot be undone or repeated: the handle is dead from then on and every
    // later call says so (include/fdc_amd.h, "Failure").
    bool poisoned = false;
    std::string poison_why;
    hipStream_t stream = nullptr;
    float2 *d_spec = nullptr;                // (max_blocks + 1) * N: slot 0 = history block.  The buffer the NEXT batch is read from
    float2 *d_wins = nullptr, *d_tw = nullptr, *d_tw256 = nullptr;    // window pool, exp(-2 pi i k/N), exp(-2 pi i j/256)
    fdc::PowerCell *d_cells = nullptr;
    float *d_power = nullptr;                // power cells of the batch in d_spec
    // FDC_SINKS_LOOKAHEAD: a second spectrum / power buffer and a stream of its own for their producer, so that the forward transform
    // (and the power cells) of batch n + 1 run on the device beside the decision kernels of batch n — one wave per channel or a
    // workgroup per segment: latency-bound kernels that leave the machine idle (fdc_sinks_spectrum_ahead, fdc_sinks_prepare_ahead).
    // d_spec / d_power always name the buffers of the batch the next submit reads; the pair swaps when a batch's extractions are enqueued.
    float2 *d_spec_ahead = nullptr;
    float *d_power_ahead = nullptr;
    // ... and, only under -DFDC_SINKS_THREE_BUFFERS, a THIRD spectrum buffer (round 6, measured and not shipped).  With two, the forward transform of
    // batch n + 2 waits for the extractions of batch n to release their buffer, and a batch's whole chain (transform, cells, decisions, the host's look at
    // the summary, task placement, extractions: 0.65 ms at configs[2]) runs two deep: 0.34 ms per step for 0.29 ms of fill-stream work
    // (profiles/r06/timeline_cfg3_shipped.txt).  With three (current <- ahead <- spare <- current; the fill stream waits for the extractions of the
    // batch BEFORE the one just submitted) the transform does run beside the extractions — and both slow down: they are the two bandwidth-heavy
    // kernels of the step (configs[2] 0.340 -> 0.335 ms, forward kernel 0.25 -> 0.28; configs[4] 0.49 -> 0.53, forward kernel 0.40 beside k_det_track
    // and the extractions; profiles/r06/sched_three_buffers.txt).  The step is the memory system's, not the schedule's.
    float2 *d_spec_spare = nullptr;
    // round 6: the power of every 16-bin group of the spectra in d_spec (slot 1 on) / d_spec_ahead, written by the producer's forward kernel
    // (fdc_pipeline_process_device_power) beside the spectrum: fdc_sinks_prepare_from_groups sums the cells from it instead of reading the spectrum back
    float *d_gpow = nullptr, *d_gpow_ahead = nullptr;
    hipStream_t s_fill = nullptr;
    hipEvent_t ev_fill = nullptr;
    // ... and two side streams: the width classes above 4096 points are two small launches each (a few hundred transforms); side by side
    // they fill the device, one after the other they do not (configs[4]: 3 x (42 + 20) us).  Only with the flag: with the payload copy to the
    // host running, more streams than hardware queues put a class behind the copy (profiles/r03/NOTES.md).
    hipStream_t s_side[2] = {nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[2] = {nullptr, nullptr};
    // ... and the extraction kernels of a batch on a stream of their own (s_x), so that the NEXT batch's decision chain — enqueued on the bank's
    // stream by the next submit — starts beside them instead of behind them (two deep as before: that submit hands out this batch's PDUs).
    // Placement of the tasks and the buffered blocks' move stay on the bank's stream (they read what the next chain overwrites).
    hipStream_t s_x = nullptr;
    hipEvent_t ev_tasks = nullptr;
    hipEvent_t ev_ready = nullptr, ev_ready_ahead = nullptr;   // recorded on s_fill behind the power cells of the batch in d_spec / d_spec_ahead (fdc_sinks_prepare)
    int prepared = -1, prepared_ahead = -1;  // blocks whose power cells are already (being) computed in d_power / d_power_ahead on s_fill; -1 = none
    fdc::ExtractTask *d_tasks = nullptr; size_t cap_tasks = 0;
    float2 *d_ext = nullptr; size_t cap_ext = 0;
    float2 *d_wide = nullptr; size_t wide_cap = 0;  // scratch of extractions wider than 4096 points (between the two passes): wide_cap points
    std::vector<fdc::ExtractTask> sorted;          // tasks grouped by width class
    cfl *h_ext = nullptr; size_t cap_hext = 0;      // pinned landing buffer of the extractions
    std::vector<float> h_power;
    std::vector<PduRec> pdus;
    // per-call scratch
    std::vector<fdc::ExtractTask> tasks;
    std::vector<int> task_w, task_skip;
    int64_t ext_used = 0;
    std::vector<std::unique_ptr<WorkerLists>> wl;    // one per worker thread (separate heap objects: no shared cache lines)
    WorkerPool pool;
    std::string det_logfile;                 // verbose == 2: …vcm_impl.cc:94 / SegmentDetection_impl.cc:51
    int host_threads = 0;                    // cfg.threads (host engine), 0 = from the bank's size
    // ---- device engine (fdc_sinks_dev.hip): decisions, layout and buffered blocks stay on the device
    struct Dev {
        bool on = false;
        int nlist = 0, npw = 0;
        long long max_list = 0;              // longest task list (grid of the scatter kernel)
        std::vector<int64_t> task_base, pdu_base, owner_base, cand_base;
        int64_t *d_task_base = nullptr, *d_pdu_base = nullptr, *d_owner_base = nullptr, *d_cand_base = nullptr;
        int32_t *d_ntask = nullptr, *d_npdu = nullptr, *d_nowner = nullptr, *d_error = nullptr, *d_class_fill = nullptr;
        int32_t *d_ncand = nullptr, *d_winoff = nullptr, *d_live = nullptr, *d_live2 = nullptr;
        fdc::DetCh *d_detch = nullptr;        // tracker scratch: one life record per entry of the owner table
        int64_t *d_live_off = nullptr;
        int2 *d_cand = nullptr;
        fdc::PacGeom *d_pgeom = nullptr; fdc::PacState *d_pstate = nullptr;
        fdc::DetGeom *d_dgeom = nullptr; fdc::DetSegState *d_sst = nullptr;
        fdc::SinkTask *d_tasks = nullptr; fdc::SinkPdu *d_pdus = nullptr, *d_pdus_out = nullptr; fdc::SinkOwner *d_owners = nullptr;
        fdc::ExtractTask *d_sorted = nullptr;
        fdc::SinkSummary *d_sum = nullptr, *h_sum = nullptr;
        fdc::SinkPdu *h_pdus = nullptr;      // pinned: the first kEagerPdus records travel with the summary
        float2 *d_land[2] = {nullptr, nullptr}; size_t cap_land[2] = {0, 0};      // landing buffers (emitted runs, then buffered rests)
        cfl *h_land[2] = {nullptr, nullptr}; size_t cap_hland[2] = {0, 0};        // pinned copies of the emitted runs
        hipStream_t s_copy = nullptr;
        int carry_width = 0;                 // streams that can hold blocks from the call before: every PowerActivationChannel, or a
                                             // segment's live channels (disjoint detect ranges: at most one per power cell)
        hipEvent_t ev_decide = nullptr, ev_extract[2] = {nullptr, nullptr}, ev_copied[2] = {nullptr, nullptr};
        std::vector<fdc::SinkPdu> recs[2];
        std::vector<std::pair<int64_t, uint32_t>> order;     // emission order of a batch's records (scratch of dev_build)
        fdc::SinkSummary sum[2];
        int64_t bc0[2] = {0, 0};              // block counter at the start of the batch in landing buffer b
        int nb_of[2] = {0, 0};
        bool pend[2] = {false, false};        // batch in landing buffer b is enqueued and not yet handed out
        int cur = 1;                          // landing buffer of the newest batch
        bool inflight = false;                // some batch is pending
        bool any = false;                     // a batch has run: d_land[cur] holds buffered blocks
        int eager_n = 0;                      // look-ahead: the decision chain of the NEXT batch (prepared, this many blocks) is already enqueued
    } dev;
};

namespace {

std::mutex g_log_mu;
fdc_log_fn g_log_fn = nullptr;
void *g_log_user = nullptr;

// get_current_time(), …vcm_impl.cc:56-69 / PowerActivationChannel_impl.cc:435-447 (the reference formats into char p[40]
// with a stated size of 80; 19 characters are written)
// One second of resolution: the string is formatted once per second and thread (localtime_r takes the C library's time-zone
// lock; a bank of 256 channels activates thousands of times per batch).
const std::string &current_time_string()
{
    thread_local time_t last = (time_t)-1;
    thread_local std::string text;
    const time_t t = time(nullptr);
    if (t != last) {
        char buf[40];
        struct tm tmv;
        localtime_r(&t, &tmv);
        strftime(buf, sizeof buf, "%Y-%m-%d-%H-%M-%S", &tmv);
        text = buf; last = t;
    }
    return text;
}

// log(), PowerActivationChannel_impl.cc:396-408 / …vcm_impl.cc:578-591: a line to stdout or appended to the log file
void sink_log(const fdc_sinks *s, const std::string &file, const std::string &line)
{
    {
        fdc_log_fn fn; void *user;
        { std::lock_guard<std::mutex> g(g_log_mu); fn = g_log_fn; user = g_log_user; }
        if (fn) fn(line.c_str(), user);
    }
    if (s->cfg.verbose == 1) { std::fputs(line.c_str(), stdout); std::fputc('\n', stdout); }
    else if (s->cfg.verbose == 2) {
        FILE *f = std::fopen(file.c_str(), "a");
        if (!f) std::fprintf(stderr, "Outputfile not writable: %s\n", file.c_str());
        else { std::fputs(line.c_str(), f); std::fputc('\n', f); std::fclose(f); }
    }
}
std::string pac_logfile(const Pac &p) { return "gr-FDC.PowActChan." + std::to_string(p.ID) + ".log"; }
void start_logfile(const std::string &file)      // constructors: the file is truncated to one empty line
{
    FILE *f = std::fopen(file.c_str(), "w");
    if (!f) std::fprintf(stderr, "Logfile not writable: %s\n", file.c_str());
    else { std::fputc('\n', f); std::fclose(f); }
}

// Where the decisions of one batch are collected: the handle's own lists, or the private lists of a worker thread that
// runs a range of PowerActivationChannels on its own (they do not interact; the lists are merged afterwards).
struct Emit {
    std::vector<fdc::ExtractTask> *tasks;
    std::vector<int> *task_w, *task_skip;
    int64_t *ext_used;
    std::vector<PduRec> *pdus;
    int64_t blockcount;             // the block counter while the current block is processed
    int64_t key;                    // order key of a PDU emitted now
};

int64_t add_task(Emit &e, int slot, int start, int w, int skip, int win_off)
{
    fdc::ExtractTask t{};
    t.slot = slot; t.start = start; t.win_off = win_off; t.out_off = *e.ext_used;
    *e.ext_used += w - skip;
    e.tasks->push_back(t);
    e.task_w->push_back(w); e.task_skip->push_back(skip);
    return (int64_t)e.tasks->size() - 1;
}

// ---------------------------------------------------------------- PowerActivationChannel
void pac_process(const fdc_sinks *s, Emit &e, Pac &p, int slot)            // process_channel, …_impl.cc:260-284
{
    BlockRef b;
    b.task = add_task(e, slot, p.extract_start, p.extract_width, p.ovl_offset, p.win_off + p.phase * p.extract_width);
    p.blocks.push_back(std::move(b));
    p.count++;
    p.phase = (p.phase + p.deltaphase) % s->R;
}

void pac_emit(const fdc_sinks *s, Emit &e, Pac &p, bool fin)               // emit_data, :212-258
{
    PduRec r;
    r.key = e.key;
    r.meta.kind = 0; r.meta.source = p.ID; r.meta.chan_id = p.id_at_activation;
    r.meta.finalized = fin; r.meta.part = p.part; r.meta.has_part = 1;
    r.meta.rel_cfreq = (double)(p.extract_start + p.extract_stop) / 2.0 / (double)s->N;
    r.meta.rel_bw = (double)p.extract_width / (double)s->N;
    r.meta.blockstart = e.blockcount - p.count; r.meta.blockend = e.blockcount;
    r.meta.vectorstart = p.extract_start; r.meta.vectorend = p.extract_stop;
    std::snprintf(r.meta.id, sizeof r.meta.id, "%s", p.msg_id.c_str());
    r.blocklen = p.output_len;
    r.blocks = std::move(p.blocks);                                        // the whole list changes hands: no per-block move
    p.blocks.clear();
    p.blocks.reserve(r.blocks.size() + 2);
    if (s->cfg.verbose)                                                    // :246-253
        sink_log(s, pac_logfile(p), p.msg_id + (fin ? std::string(".fin") : ".parted." + std::to_string(p.part)) + ": start=" +
                 std::to_string(p.extract_start) + ", stop=" + std::to_string(p.extract_stop) + ", blockstart=" +
                 std::to_string((long long)r.meta.blockstart) + ", blockend=" + std::to_string((long long)r.meta.blockend));
    e.pdus->push_back(std::move(r));
    p.part++;
}

void pac_step(const fdc_sinks *s, Emit &e, Pac &p, float pwr, int slot)    // one item of work(), :146-170
{
    if (pwr == 0.0f) pwr = FLT_MIN;                                        // :293-294
    bool changed = false;
    if (!p.active && pwr / p.lastpower >= s->pac_thr) changed = true;      // :296-302
    else if (p.active && p.lastpower / pwr >= s->pac_thr) changed = true;
    p.lastpower = pwr;
    if (changed) {
        if (!p.active) {                                                   // activate(), :198-210
            p.part = 0; p.count = 0; p.active = true; p.phase = 0; p.blocks.clear();
            p.id_at_activation = p.finished;
            p.msg_id = current_time_string() + ".PowActChan." + std::to_string(p.ID) + "." + std::to_string(p.finished);
            pac_process(s, e, p, slot - 1);                                // previous block (slot 0 = saved history)
            pac_process(s, e, p, slot);
        } else {
            pac_process(s, e, p, slot);
            p.active = false;                                              // deactivate(), :189-196
            pac_emit(s, e, p, true);
            p.finished++;
        }
    } else if (p.active) {
        pac_process(s, e, p, slot);
        const int mb = s->cfg.pac_maxblocks;
        if (mb == 0 || (mb > 0 && p.count % mb == 0)) pac_emit(s, e, p, false);
    }
}

// ---------------------------------------------------------------- activity_detection_channelizer_vcm
void det_process(fdc_sinks *s, Emit &e, DetChan &c, int slot)        // process_channel, …vcm_impl.cc:373-397
{
    BlockRef b;
    b.task = add_task(e, slot, c.extract_start, c.extract_width, c.ovlskip,
                      s->det_win_off[c.wclass] + c.phase * c.extract_width);
    c.data.push_back(std::move(b));
    c.count++;
    c.phase = (c.phase + c.phaseincrement) % s->R;
}

void det_emit(fdc_sinks *s, Emit &e, Segment &g, DetChan &c, bool fin, size_t nblk)   // :406-452 / :454-510
{
    PduRec r;
    r.key = e.key++;
    r.meta.kind = 1; r.meta.source = g.ID; r.meta.chan_id = c.ID;
    r.meta.finalized = fin; r.meta.part = c.part; r.meta.has_part = fin ? (c.part > 0) : 1;
    r.meta.rel_bw = (double)c.extract_width / (double)s->N;
    r.meta.rel_cfreq = (double)(c.extract_start + c.extract_stop) / 2.0 / (double)s->N;
    // the vcm block counts from 1 (…vcm_impl.cc:188), SegmentDetection from 0 (SegmentDetection_impl.cc:118)
    const int64_t bc = e.blockcount - (s->cfg.det_variant == 1 ? 1 : 0);
    r.meta.blockstart = bc - c.count; r.meta.blockend = bc;
    r.meta.vectorstart = c.extract_start; r.meta.vectorend = c.extract_stop;
    std::snprintf(r.meta.id, sizeof r.meta.id, "%s", c.msg_id.c_str());
    r.blocklen = c.outputsamples;
    for (size_t i = 0; i < nblk; i++) { r.blocks.push_back(std::move(c.data.front())); c.data.pop_front(); }
    if (s->cfg.verbose)                                                        // …vcm_impl.cc:441-450, :498-508
        sink_log(s, s->det_logfile, c.msg_id + (fin ? std::string(".fin: ") : ".parted." + std::to_string(c.part) + ": ") + "start=" +
                 std::to_string(c.extract_start) + ", stop=" + std::to_string(c.extract_stop) + ", blockstart=" +
                 std::to_string((long long)r.meta.blockstart) + ", blockend=" + std::to_string((long long)r.meta.blockend));
    e.pdus->push_back(std::move(r));
}

void seg_detect(fdc_sinks *s, Segment &g, const float *P)   // detect_channels, :617-628
{
    const int n = g.ncell, dec = s->dec;
    // get_active_channels, :694-739
    struct Edge { float r; int pos; };
    std::vector<Edge> rise;
    std::vector<int> fall;
    const float inv = 1.0f / s->det_thr;
    const bool sd = s->cfg.det_variant == 1;
    for (int i = 1; i < n; i++) {
        // vcm guards a zero denominator (:703-706); SegmentDetection divides as is (volk_32f_x2_divide_32f, :206)
        const float pd = (!sd && P[i - 1] == 0.0f) ? P[i] / FLT_MIN : P[i] / P[i - 1];
        if (pd > s->det_thr) rise.push_back({pd, (i - 1) * dec + g.start});
        else if (sd) { if (pd < inv) fall.push_back(i * dec + g.start); }        // if / else if (:209-210)
        if (!sd && pd < inv) fall.push_back(i * dec + g.start);                  // two independent ifs (:708-709)
    }
    std::stable_sort(rise.begin(), rise.end(), [](const Edge &a, const Edge &b) { return a.r > b.r; });   // :713
    std::vector<std::pair<int, int>> cand;
    for (const Edge &e : rise) {
        int ne = -1;
        for (int f : fall) if (f > e.pos) { ne = f; break; }                   // get_next_int, :678-692
        if (ne <= e.pos) continue;
        bool clash = false;
        for (auto &a : cand) if (e.pos < a.second && ne >= a.first) { clash = true; break; }   // :727-734
        if (!clash) cand.emplace_back(e.pos, ne);
    }
    // match_active_channels, :741-783
    if (cand.empty()) {
        for (auto &c : g.chans) c.inactive += 1;
        return;
    }
    for (auto &c : g.chans) {
        bool idle = true;
        for (size_t i = 0; i < cand.size();) {
            if (cand[i].first < c.detect_stop && cand[i].second >= c.detect_start) {
                c.inactive = 0; idle = false;
                cand.erase(cand.begin() + i);
            } else i++;
        }
        if (idle) c.inactive += 1;
    }
    for (auto &pc : cand) {                                                    // activate, :785-841
        const int dw = pc.second - pc.first, mid = pc.first + dw / 2;
        const int ew = pow2ceil((int)std::ceil((double)dw * (1.0 + 2.0 * s->cfg.window_flank_puffer)));
        if (ew > s->N) continue;                                               // logged and skipped in the reference
        if (s->det_win_off[(size_t)std::lround(std::log2((double)ew))] < 0) continue;   // no window table for this width (see create)
        int es = mid - ew / 2, ee = mid + ew / 2;
        if (es < 0) { es = 0; ee = ew; }
        if (ee > s->N) { ee = s->N; es = s->N - ew; }
        DetChan c{};
        c.ID = g.counter++;
        c.detect_start = pc.first; c.detect_stop = pc.second; c.extract_start = es; c.extract_stop = ee;
        c.extract_width = ew; c.wclass = (int)std::log2((double)ew);
        c.ovlskip = ew / s->R; c.outputsamples = ew - c.ovlskip;
        c.count = 0; c.phase = 0; c.phaseincrement = es % s->R; c.inactive = -1; c.part = 0;
        const int segname = (s->cfg.det_variant == 1 && s->cfg.det_id >= 0 && s->segs.size() == 1) ? s->cfg.det_id : g.ID;
        c.msg_id = current_time_string() + ".DETECTED." + std::to_string(segname) + "." + std::to_string(c.ID);
        g.chans.push_back(std::move(c));
    }
}

void seg_extract(fdc_sinks *s, Emit &e, Segment &g, int slot)        // extract_channels_in_segments_singlethread, :306-337
{
    const int mb = s->cfg.det_maxblocks, delay = s->cfg.det_deactivation_delay;
    for (auto &c : g.chans) {
        if (c.inactive < 0) { det_process(s, e, c, slot - 1); det_process(s, e, c, slot); c.inactive = 0; }   // :399-403
        else if (c.inactive > delay) det_emit(s, e, g, c, true, c.data.size());
        else det_process(s, e, c, slot);
        if (s->cfg.det_variant == 0 && mb >= 0 && (int)c.data.size() >= mb) {  // :317-318, :454-470
            const size_t ntx = mb == 0 ? c.data.size() : (size_t)mb;
            if (ntx > 0) { det_emit(s, e, g, c, false, ntx); c.part++; }
        }
    }
    if (s->cfg.det_variant == 1 && mb >= 0)                                     // SegmentDetection: separate pass, :359-362
        for (auto &c : g.chans)
            if ((int)c.data.size() >= mb) {
                const size_t ntx = mb == 0 ? c.data.size() : (size_t)mb;
                if (ntx > 0) { det_emit(s, e, g, c, false, ntx); c.part++; }
            }
    for (size_t i = 0; i < g.chans.size();)                                     // clear_inactive_channels, :512-524
        if (g.chans[i].inactive > delay) g.chans.erase(g.chans.begin() + i); else i++;
}

// ---------------------------------------------------------------- device engine: set-up
// Which engine a bank gets, and the device-side tables and lists of the device engine.  Every list is allocated for its
// worst case (a channel toggling in every block, a segment full of one-cell carriers), so no call can overflow one; a bank
// whose worst case does not fit a 2 GiB budget takes the host engine.
constexpr int kEagerPdus = 4096;

int dev_setup(fdc_sinks *s)
{
    auto &d = s->dev;
    const fdc_sinks_cfg &cfg = s->cfg;
    if ((cfg.flags & FDC_SINKS_HOST_DECISIONS) || cfg.verbose != 0) return FDC_OK;
    const int npac = (int)s->pacs.size(), nseg = (int)s->segs.size();
    if (npac + nseg == 0) return FDC_OK;
    {
        int capmax = 1;
        for (const Segment &g : s->segs) { if (g.ncell > fdc::kDetMaxCells) return FDC_OK; capmax = std::max(capmax, g.ncell / 2 + 1); }
        if (nseg && (capmax > 512 || fdc::det_track_staged(cfg.max_blocks, capmax) < 0)) return FDC_OK;   // the tracker's tables would not fit
    }
    const int64_t nbmax = cfg.max_blocks;
    d.carry_width = npac;
    for (const Segment &g : s->segs) d.carry_width = std::max(d.carry_width, std::min(g.ncell, fdc::kDetMaxCells));
    d.npw = npac;                                                              // one list per PowerActivationChannel (one wave each)
    d.nlist = d.npw + nseg;
    d.task_base.assign((size_t)d.nlist + 1, 0); d.pdu_base.assign((size_t)d.nlist + 1, 0);
    d.owner_base.assign((size_t)nseg + 1, npac); d.cand_base.assign((size_t)nseg + 1, 0);
    for (int l = 0; l < d.nlist; l++) {
        int64_t tc, pc;
        if (l < d.npw) { tc = 2 * nbmax; pc = nbmax; }                         // per block: at most two extractions, one emission
        else {
            const int64_t nc = s->segs[(size_t)(l - d.npw)].ncell;
            tc = nbmax * nc + nbmax * (nc / 2 + 1);                            // every live channel once, every activation twice
            pc = nbmax * nc;
        }
        d.task_base[(size_t)l + 1] = d.task_base[(size_t)l] + tc; d.pdu_base[(size_t)l + 1] = d.pdu_base[(size_t)l] + pc;
        d.max_list = std::max<long long>(d.max_list, tc);
    }
    for (int g = 0; g < nseg; g++) {
        const int64_t nc = s->segs[(size_t)g].ncell;
        d.owner_base[(size_t)g + 1] = d.owner_base[(size_t)g] + nc + nbmax * (nc / 2 + 1);
        d.cand_base[(size_t)g + 1] = d.cand_base[(size_t)g] + nbmax * (nc / 2 + 1);
    }
    const int64_t ntask = d.task_base.back(), npdu = d.pdu_base.back(), nown = d.owner_base.back(), ncand = d.cand_base.back();
    const int64_t bytes = ntask * (int64_t)(sizeof(fdc::SinkTask) + sizeof(fdc::ExtractTask)) + npdu * 2 * (int64_t)sizeof(fdc::SinkPdu) +
                          nown * (int64_t)sizeof(fdc::SinkOwner) + ncand * (int64_t)sizeof(int2);
    if (bytes > (2ll << 30)) return FDC_OK;
#define DALLOC(ptr, n) HIPCHK(hipMalloc(reinterpret_cast<void **>(&(ptr)), std::max<size_t>(16, sizeof(*(ptr)) * (size_t)(n))))
#define DUP(ptr, vec) do { DALLOC(ptr, (vec).size()); HIPCHK(hipMemcpy(ptr, (vec).data(), sizeof(*(ptr)) * (vec).size(), hipMemcpyHostToDevice)); } while (0)
    DUP(d.d_task_base, d.task_base); DUP(d.d_pdu_base, d.pdu_base); DUP(d.d_owner_base, d.owner_base); DUP(d.d_cand_base, d.cand_base);
    DALLOC(d.d_ntask, d.nlist); DALLOC(d.d_npdu, d.nlist); DALLOC(d.d_nowner, nseg + 1); DALLOC(d.d_error, 1); DALLOC(d.d_class_fill, 32);
    HIPCHK(hipMemset(d.d_error, 0, sizeof(int32_t)));
    HIPCHK(hipMemset(d.d_nowner, 0, sizeof(int32_t) * (size_t)(nseg + 1)));
    DALLOC(d.d_tasks, ntask); DALLOC(d.d_sorted, ntask); DALLOC(d.d_pdus, npdu); DALLOC(d.d_pdus_out, std::max<int64_t>(npdu, kEagerPdus)); DALLOC(d.d_owners, nown);
    DALLOC(d.d_sum, 1);
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&d.h_sum), sizeof(fdc::SinkSummary), hipHostMallocDefault));
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&d.h_pdus), sizeof(fdc::SinkPdu) * kEagerPdus, hipHostMallocDefault));
    if (npac) {
        std::vector<fdc::PacGeom> pg((size_t)npac);
        std::vector<fdc::PacState> ps((size_t)npac);
        for (int i = 0; i < npac; i++) {
            const Pac &p = s->pacs[(size_t)i];
            pg[(size_t)i] = fdc::PacGeom{p.cell, p.extract_start, p.extract_width, 31 - __builtin_clz((unsigned)p.extract_width), p.ovl_offset,
                                         p.output_len, p.deltaphase, p.win_off, p.ID, 0};
            fdc::PacState st{};
            st.lastpower = FLT_MAX;                                            // PowerActivationChannel_impl.cc:92
            ps[(size_t)i] = st;
        }
        DUP(d.d_pgeom, pg); DUP(d.d_pstate, ps);
    }
    if (nseg) {
        std::vector<fdc::DetGeom> dg((size_t)nseg);
        for (int g = 0; g < nseg; g++) {
            const Segment &sg = s->segs[(size_t)g];
            dg[(size_t)g] = fdc::DetGeom{sg.ID, sg.start, sg.ncell, sg.cell0, sg.ncell / 2 + 1, 0};
        }
        DUP(d.d_dgeom, dg);
        std::vector<fdc::DetSegState> st((size_t)nseg, fdc::DetSegState{0, 0});
        DUP(d.d_sst, st);
        DALLOC(d.d_live, (size_t)nseg * fdc::kDetFields * fdc::kDetMaxCells);
        DALLOC(d.d_live2, (size_t)nseg * fdc::kDetFields * fdc::kDetMaxCells);
        DALLOC(d.d_detch, nown);
        DALLOC(d.d_live_off, (size_t)nseg * fdc::kDetMaxCells);
        DALLOC(d.d_cand, ncand);
        DALLOC(d.d_ncand, (size_t)nseg * nbmax);
        std::vector<int32_t> wo(32, -1);
        for (size_t k = 0; k < s->det_win_off.size() && k < 32; k++) wo[k] = s->det_win_off[k];
        DUP(d.d_winoff, wo);
    }
#undef DUP
#undef DALLOC
    {
        // the payload copy's stream at the highest priority (its own hardware-queue pool: see the extraction streams in fdc_sinks_create): all it ever
        // carries are a wait, a copy and a mark, and behind a 250-us forward kernel on a shared queue they start a kernel late
        int lo = 0, hi = 0;
        HIPCHK(hipDeviceGetStreamPriorityRange(&lo, &hi));
#ifdef FDC_SINKS_NO_PRIO
        hi = 0;
#endif
        HIPCHK(hipStreamCreateWithPriority(&d.s_copy, hipStreamNonBlocking, hi));
    }
    HIPCHK(hipEventCreateWithFlags(&d.ev_decide, hipEventDisableTiming));
    for (int i = 0; i < 2; i++) {
        HIPCHK(hipEventCreateWithFlags(&d.ev_extract[i], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&d.ev_copied[i], hipEventDisableTiming));
    }
    d.on = true;
    return FDC_OK;
}

}  // namespace

extern "C" {

void fdc_sinks_destroy(fdc_sinks *s)
{
    if (!s) return;
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    if (s->s_fill) { (void)hipStreamSynchronize(s->s_fill); (void)hipStreamDestroy(s->s_fill); }
    for (hipStream_t q : {s->s_side[0], s->s_side[1], s->s_x}) if (q) { (void)hipStreamSynchronize(q); (void)hipStreamDestroy(q); }
    for (hipEvent_t e : {s->ev_fill, s->ev_ready, s->ev_ready_ahead, s->ev_fork, s->ev_join[0], s->ev_join[1], s->ev_tasks}) if (e) (void)hipEventDestroy(e);
    (void)hipFree(s->d_spec_ahead); (void)hipFree(s->d_spec_spare); (void)hipFree(s->d_power_ahead); (void)hipFree(s->d_gpow); (void)hipFree(s->d_gpow_ahead);
    (void)hipFree(s->d_spec); (void)hipFree(s->d_wins); (void)hipFree(s->d_tw); (void)hipFree(s->d_tw256); (void)hipFree(s->d_cells);
    (void)hipFree(s->d_power); (void)hipFree(s->d_tasks); (void)hipFree(s->d_ext); (void)hipFree(s->d_wide);
    if (s->h_ext) (void)hipHostFree(s->h_ext);
    {
        auto &d = s->dev;
        if (d.s_copy) { (void)hipStreamSynchronize(d.s_copy); (void)hipStreamDestroy(d.s_copy); }
        for (hipEvent_t e : {d.ev_decide, d.ev_extract[0], d.ev_extract[1], d.ev_copied[0], d.ev_copied[1]}) if (e) (void)hipEventDestroy(e);
        for (void *q : {(void *)d.d_task_base, (void *)d.d_pdu_base, (void *)d.d_owner_base, (void *)d.d_cand_base, (void *)d.d_ntask,
                        (void *)d.d_npdu, (void *)d.d_nowner, (void *)d.d_error, (void *)d.d_class_fill, (void *)d.d_ncand,
                        (void *)d.d_winoff, (void *)d.d_live, (void *)d.d_live2, (void *)d.d_detch, (void *)d.d_live_off, (void *)d.d_cand, (void *)d.d_pgeom,
                        (void *)d.d_pstate, (void *)d.d_dgeom, (void *)d.d_sst, (void *)d.d_tasks, (void *)d.d_pdus,
                        (void *)d.d_pdus_out, (void *)d.d_owners, (void *)d.d_sorted, (void *)d.d_sum, (void *)d.d_land[0],
                        (void *)d.d_land[1]})
            (void)hipFree(q);
        for (void *q : {(void *)d.h_sum, (void *)d.h_pdus, (void *)d.h_land[0], (void *)d.h_land[1]}) if (q) (void)hipHostFree(q);
    }
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

int fdc_sinks_create(const fdc_sinks_cfg *cfg, fdc_sinks **out)
{
    FDC_ENTRY("fdc_sinks_create")
    if (!cfg || !out) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    const int N = cfg->blocklen, R = cfg->relinvovl;
    // shared predicates: PowerActivationChannel_impl.cc:64-70, …vcm_impl.cc:106-107,122-123
    if (N < 2 || !ispow2i(N)) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "Blocklen invalid.");
    if (R < 1 || !ispow2i(R) || R > N) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "Relative inverse overlap is invalid, must be >0 and a power of 2.");
    if (cfg->npac < 0 || cfg->nseg < 0 || (cfg->npac && !cfg->pac) || (cfg->nseg && !cfg->seg) || cfg->max_blocks < 1)
        return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "bad sink lists");
    std::unique_ptr<fdc_sinks> s(new fdc_sinks());
    s->cfg = *cfg; s->cfg.pac = nullptr; s->cfg.seg = nullptr;
    s->N = N; s->R = R;
    std::vector<cfl> pool;
    // ---- PowerActivationChannel instances
    if (cfg->npac > 0) {
        if (cfg->pac_thresh_db <= 0.0f)                                        // set_thresh, :377-381
            return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "Threshold is interpreted as dB and must be >0.0");
        s->pac_thr = (float)std::pow(10.0, (double)cfg->pac_thresh_db / 10.0);
    }
    for (int i = 0; i < cfg->npac; i++) {
        float cfreq = cfg->pac[i].cfreq, bw = cfg->pac[i].bw;
        bw = bw > 0.0f ? bw : -bw;                                              // set_startstop, :314-355
        if (bw > 1.0 || cfreq - bw / 2.0f < 0.0f || cfreq + bw / 2.0f > 1.0f)
            return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "Desired channel is out of band: cfreq=%f, bw=%f", cfreq, bw);
        const int k = (int)std::ceil((double)bw * (double)N);
        if (k <= 0) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "Can't eval nextpow2 from %d", k);
        Pac p;
        p.ID = cfg->pac[i].id;
        p.extract_width = std::min(pow2ceil(k), N);
        const int mid = (int)std::round((double)cfreq * (double)N);
        p.extract_start = std::max(0, mid - p.extract_width / 2);
        p.extract_stop = p.extract_start + p.extract_width;
        if (p.extract_stop > N) { p.extract_stop = N; p.extract_start = p.extract_stop - N; }   // reference clamp (App. B.2)
        p.measure_start = std::max((int)std::round((double)(cfreq - bw / 2.0f) * (double)N), p.extract_start);
        p.measure_stop = std::min((int)std::round((double)(cfreq + bw / 2.0f) * (double)N), p.extract_stop);
        if (p.extract_start + p.extract_width > N)
            return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "PowerActivationChannel %d: the reference reads past the block here", i);
        p.deltaphase = p.extract_start % R;
        p.ovl_offset = p.extract_width / R; p.output_len = p.extract_width - p.ovl_offset;
        // cr_windows, :357-375: unit phasor (float polar) with a rising sine edge; only the first extract_width
        // entries of the block-long table are ever used (:267), the mirrored far edge matters only when it falls inside.
        const int ramp = ((p.extract_stop - p.extract_start) - (p.measure_stop - p.measure_start)) / 3;
        p.win_off = (int)pool.size();
        pool.resize(pool.size() + (size_t)R * p.extract_width);
        for (int r = 0; r < R; r++) {
            const float ang = (float)(2.0f * M_PI * (double)r / (double)R);
            const cfl ph(std::cos(ang), std::sin(ang));
            std::vector<cfl> full((size_t)N, ph);
            for (int q = 0; q < ramp; q++) {
                full[q] *= (float)std::sin(0.5 * M_PI * (double)q / (double)(ramp + 1));
                full[N - q - 1] = full[q];
            }
            std::copy(full.begin(), full.begin() + p.extract_width, pool.begin() + p.win_off + (size_t)r * p.extract_width);
        }
        p.cell = (int)s->cells.size();
        s->cells.push_back({p.measure_start, std::max(0, p.measure_stop - p.measure_start), 1.0f, 0});
        s->pacs.push_back(std::move(p));
    }
    // ---- detection segments
    if (cfg->nseg > 0 && cfg->det_variant == 1) {
        // SegmentDetection face (lib/SegmentDetection_impl.cc:68-86, :592-637): one instance per segment in the hier block
        auto mod_f = [](float x) { return (float)std::fmod(std::fmod((double)x, 1.0) + 1.0, 1.0); };   // :700-703
        if (cfg->det_thresh_db < 0.0f) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "Threshold is interpreted as dB and must be greater zero to detect channels accordingly.");
        if (cfg->window_flank_puffer < 0.0) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "Window flank puffer must not be smaller 0.0.");
        const float mcd = mod_f(cfg->minchandist);
        const double dd = (double)N * (double)mcd / 2.0;
        s->dec = dd < 2.0 ? 1 : (int)dd;
        s->det_thr = (float)std::pow(10.0, (double)cfg->det_thresh_db / 10.0);
        for (int i = 0; i < cfg->nseg; i++) {
            float a = mod_f(cfg->seg[i].start), b = mod_f(cfg->seg[i].stop);
            if (a == b) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "Start must not be equal to stop.");
            if (a > b) std::swap(a, b);
            size_t width = (size_t)((double)(b - a) * (double)N);
            if (width % (size_t)s->dec) width += (size_t)s->dec - width % (size_t)s->dec;
            if (width > (size_t)N) width = (size_t)(N - N % s->dec);
            const size_t mid = (size_t)((double)(0.5f * (a + b)) * (double)N);
            size_t st = mid < width / 2 ? 0 : mid - width / 2, sp = st + width;
            if (sp > (size_t)N) { sp = (size_t)N; st = sp - (size_t)N; }                 // reference clamp (App. B.2)
            if (st + width > (size_t)N) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "segment %d: the reference reads past the block here", i);
            Segment g;
            g.ID = cfg->seg_id_base + i; g.start = (int)st; g.stop = (int)sp; g.width = (int)width;
            g.ncell = (int)width / s->dec; g.cell0 = (int)s->cells.size();
            for (int c = 0; c < g.ncell; c++) s->cells.push_back({g.start + c * s->dec, s->dec, 1.0f, 0});   // raw sums (:185-190)
            s->segs.push_back(std::move(g));
        }
    }
    if (cfg->nseg > 0 && cfg->det_variant != 1) {
        if (cfg->minchandist <= 0.0f || cfg->minchandist >= 1.0)               // :231-232
            return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "Minimum channel distance is invalid. Must be in (0,1)");
        if (cfg->det_thresh_db < 0.0f) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "Threshold is interpreted as dB and must be greater zero.");
        if (cfg->det_deactivation_delay < 0) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "Channel deactication delay must not be smaller 0.");
        if (cfg->window_flank_puffer < 0.0) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "Window flank puffer must not be smaller 0.0.");
        const double dd = (double)N * (double)cfg->minchandist / 2.0;           // :234-240
        s->dec = dd < 2.0 ? 1 : (int)dd;
        s->det_thr = (float)std::pow(10.0, (double)cfg->det_thresh_db / 10.0);
        for (int i = 0; i < cfg->nseg; i++) {                                   // create_segment, :248-279
            const float v0 = cfg->seg[i].start, v1 = cfg->seg[i].stop;
            if (v0 >= v1 || v0 < 0.0f || v1 > 1.0f) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "Segment is incorrect: [%f, %f]", v0, v1);
            const int mid = std::abs((int)std::round(((double)v1 + (double)v0) * 0.5 * (double)N));
            int width = std::abs((int)std::round(((double)v1 - (double)v0) * (double)N));
            if (width % s->dec) width += s->dec - width % s->dec;
            if (width >= N) {
                if (N % s->dec == 0) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "segment %d spans the whole block (the reference does not terminate here)", i);
                width = N - N % s->dec;
            }
            Segment g;
            g.ID = cfg->seg_id_base + i;
            g.start = mid - width / 2 <= 0 ? 0 : mid - width / 2;
            g.stop = g.start + width;
            if (g.stop > N) { g.stop = N; g.start = N - width; }
            if (g.start < 0) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "Cannot evaluate start and stop of segment %d", i);
            g.width = width; g.ncell = width / s->dec; g.cell0 = (int)s->cells.size();
            const float norm = 1.0f / (float)s->dec;                            // :632
            for (int c = 0; c < g.ncell; c++) s->cells.push_back({g.start + c * s->dec, s->dec, norm, 0});
            s->segs.push_back(std::move(g));
        }
    }
    if (cfg->nseg > 0) {
        // cr_windows (…vcm_impl.cc:199-228, identical in SegmentDetection_impl.cc:551-583): every power-of-two width
        // x R phases, unit amplitude, Hann flanks
        const int nw = (int)std::log2((double)N) + 1;
        s->det_win_off.resize(nw);
        for (int k = 0; k < nw; k++) {
            const int ww = 1 << k, puf = (int)(cfg->window_flank_puffer * (double)ww);
            s->det_win_off[k] = -1;
            // widths above one workgroup's transform get a table only while it stays small (<= 32 MiB); carriers wider
            // than that are skipped at activation like the ones wider than the block
            if (ww > fdc::kMaxLdsFft && (size_t)R * ww * sizeof(cfl) > ((size_t)32 << 20)) continue;
            s->det_win_off[k] = (int)pool.size();
            pool.resize(pool.size() + (size_t)R * ww);
            for (int r = 0; r < R; r++) {
                cfl *w = pool.data() + s->det_win_off[k] + (size_t)r * ww;
                const double ang = 2.0 * M_PI * (double)r / (double)R;
                const cfl ph((float)std::cos(ang), (float)std::sin(ang));
                for (int n = 0; n < ww; n++) w[n] = ph;
                for (int q = 0; q < puf; q++) {
                    const float fl = 0.5f - 0.5f * (float)std::cos(M_PI * (double)q / (double)puf);
                    w[q] *= fl; w[ww - 1 - q] *= fl;
                }
            }
        }
    }
    int rc = fdc::pick_device(cfg->device_id);
    if (rc != FDC_OK) return rc;
    HIPCHK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    fdc_sinks *raw = s.release();
#define CHKF(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { int _r = fdc::set_error(_e == hipErrorOutOfMemory ? FDC_ERR_NOMEM : FDC_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); fdc_sinks_destroy(raw); return _r; } } while (0)
    CHKF(hipMalloc(&raw->d_spec, sizeof(float2) * ((size_t)cfg->max_blocks + 1) * N));
    CHKF(hipMemset(raw->d_spec, 0, sizeof(float2) * (size_t)N));            // zero history block (…cc:89 / :111)
    if (!pool.empty()) {
        CHKF(hipMalloc(&raw->d_wins, sizeof(float2) * pool.size()));
        CHKF(hipMemcpy(raw->d_wins, pool.data(), sizeof(float2) * pool.size(), hipMemcpyHostToDevice));
    }
    {
        std::vector<float2> tw((size_t)N);
        for (int k = 0; k < N; k++) {
            const double a = -2.0 * M_PI * (double)k / (double)N;
            tw[k] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
        CHKF(hipMalloc(&raw->d_tw, sizeof(float2) * (size_t)N));
        CHKF(hipMemcpy(raw->d_tw, tw.data(), sizeof(float2) * (size_t)N, hipMemcpyHostToDevice));
        if (N >= 256) {                                                        // width-256 extractions run on the register kernel
            std::vector<float2> t256(256);
            for (int j = 0; j < 256; j++) t256[(size_t)j] = tw[(size_t)j * (size_t)(N / 256)];
            CHKF(hipMalloc(&raw->d_tw256, sizeof(float2) * 256));
            CHKF(hipMemcpy(raw->d_tw256, t256.data(), sizeof(float2) * 256, hipMemcpyHostToDevice));
        }
    }
    if (!raw->cells.empty()) {
        CHKF(hipMalloc(&raw->d_cells, sizeof(fdc::PowerCell) * raw->cells.size()));
        CHKF(hipMemcpy(raw->d_cells, raw->cells.data(), sizeof(fdc::PowerCell) * raw->cells.size(), hipMemcpyHostToDevice));
        CHKF(hipMalloc(&raw->d_power, sizeof(float) * raw->cells.size() * (size_t)cfg->max_blocks));
    }
    if (!raw->cells.empty() && N >= 16) CHKF(hipMalloc(&raw->d_gpow, sizeof(float) * (size_t)cfg->max_blocks * (size_t)(N / 16)));
    if (cfg->flags & FDC_SINKS_LOOKAHEAD) {
        if (raw->d_gpow) CHKF(hipMalloc(&raw->d_gpow_ahead, sizeof(float) * (size_t)cfg->max_blocks * (size_t)(N / 16)));
        CHKF(hipMalloc(&raw->d_spec_ahead, sizeof(float2) * ((size_t)cfg->max_blocks + 1) * N));
#ifdef FDC_SINKS_THREE_BUFFERS                // (A/B builds; measured, not shipped: see the struct)
        if (!(cfg->flags & FDC_SINKS_HOST_DECISIONS) && cfg->verbose == 0)
            CHKF(hipMalloc(&raw->d_spec_spare, sizeof(float2) * ((size_t)cfg->max_blocks + 1) * N));
#endif
        if (!raw->cells.empty()) CHKF(hipMalloc(&raw->d_power_ahead, sizeof(float) * raw->cells.size() * (size_t)cfg->max_blocks));
        CHKF(hipStreamCreateWithFlags(&raw->s_fill, hipStreamNonBlocking));
        CHKF(hipEventCreateWithFlags(&raw->ev_fill, hipEventDisableTiming));
        CHKF(hipEventCreateWithFlags(&raw->ev_ready, hipEventDisableTiming));
        CHKF(hipEventCreateWithFlags(&raw->ev_ready_ahead, hipEventDisableTiming));
        CHKF(hipEventCreateWithFlags(&raw->ev_fork, hipEventDisableTiming));
        // The extraction streams get the HIGHEST priority: the runtime maps streams onto a few hardware queues PER PRIORITY LEVEL, and at equal
        // priority the extraction stream shared one with the fill stream — a batch's extractions sat behind the NEXT batch's 250-us forward
        // kernel, whose successor in turn waits for those extractions to release the spectrum buffer (round 6, profiles/r06/timeline_cfg3_before.txt:
        // k_x256 started the moment the forward kernel ended; configs[2] 0.373 ms per 896 blocks for 0.27 of fill-stream work).  They are also the
        // work the buffer hand-over waits for: first in line is right.
        int prio_lo = 0, prio_hi = 0;
        CHKF(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        // (The payload copy's stream too, dev_setup: with the extraction stream gone from the normal pool the round robin put THAT one on the fill
        // stream's queue, and every step's copy started one forward kernel late: configs[2] 2.29 -> 2.55 ms, configs[4] 1.27 -> 1.53 per 896 blocks with
        // PDUs to host memory, profiles/r06/sched_ab.txt.)
#ifdef FDC_SINKS_NO_PRIO                      // A/B builds
        prio_hi = 0;
#endif
        CHKF(hipStreamCreateWithPriority(&raw->s_x, hipStreamNonBlocking, prio_hi));
        CHKF(hipEventCreateWithFlags(&raw->ev_tasks, hipEventDisableTiming));
        for (int i = 0; i < 2; i++) {
            CHKF(hipStreamCreateWithPriority(&raw->s_side[i], hipStreamNonBlocking, prio_hi));
            CHKF(hipEventCreateWithFlags(&raw->ev_join[i], hipEventDisableTiming));
        }
    }
    raw->host_threads = cfg->threads > 0 ? std::min(cfg->threads, 32) : 0;
    if (const char *t = fdc::debug_env("FDC_SINKS_THREADS")) if (atoi(t) >= 1) raw->host_threads = std::min(atoi(t), 32);   // debugging override
    {
        const int rcd = dev_setup(raw);
        if (rcd != FDC_OK) { fdc_sinks_destroy(raw); return rcd; }
    }
#undef CHKF
    // the tables and zeroed buffers above went through the null stream; the bank works on non-blocking streams of its own, which do not wait for it
    if (hipStreamSynchronize(nullptr) != hipSuccess) { (void)hipGetLastError(); fdc_sinks_destroy(raw); return fdc::set_error(FDC_ERR_HIP, "device synchronisation failed"); }
    // ---- logs of the constructors (verbose != 0)
    if (cfg->verbose) {
        fdc_sinks *const sp = raw;
        for (const Pac &pc : sp->pacs) {                                        // PowerActivationChannel_impl.cc:54-62, :113-127
            if (cfg->verbose == 2) start_logfile(pac_logfile(pc));
            const std::string bar("############################\n\n");
            sink_log(sp, pac_logfile(pc), bar + "# gr-FDC.PowActChan." + std::to_string(pc.ID) + "\n\n" + bar +
                     "# extract_start: " + std::to_string(pc.extract_start) + "\n# extract_stop: " + std::to_string(pc.extract_stop) +
                     "\n# extract_width: " + std::to_string(pc.extract_width) + "\n# measure_start: " + std::to_string(pc.measure_start) +
                     "\n# measure_stop: " + std::to_string(pc.measure_stop) + "\n\n# equivalent cfreq: " +
                     std::to_string((double)(pc.extract_start + pc.extract_width / 2) / (double)N) + "\n# equivalent bw: " +
                     std::to_string((double)pc.extract_width / (double)N) + "\n\n");
        }
        if (!sp->segs.empty()) {
            if (cfg->det_variant == 1) {                                        // SegmentDetection_impl.cc:49-61, :109-113
                const int id = cfg->det_id >= 0 && sp->segs.size() == 1 ? cfg->det_id : 0;
                sp->det_logfile = "gr-FDC.ActDetChan.ID_" + std::to_string(id) + ".log";
                if (cfg->verbose == 2) start_logfile(sp->det_logfile);
                for (const Segment &g : sp->segs) {
                    sink_log(sp, sp->det_logfile, "Threshold               " + std::to_string(sp->det_thr));
                    sink_log(sp, sp->det_logfile, "decimation factor       " + std::to_string(sp->dec));
                    sink_log(sp, sp->det_logfile, "start                   " + std::to_string(g.start));
                    sink_log(sp, sp->det_logfile, "stop                    " + std::to_string(g.stop));
                    sink_log(sp, sp->det_logfile, "width                   " + std::to_string(g.width));
                }
            } else {                                                            // …vcm_impl.cc:89-101, :176-186
                sp->det_logfile = "gr-FDC.ActDetChan.log";
                if (cfg->verbose == 2) start_logfile(sp->det_logfile);
                for (const Segment &g : sp->segs)
                    sink_log(sp, sp->det_logfile, "# Segment " + std::to_string(g.ID) + ": \n# start: " + std::to_string(g.start) +
                             " => f_start=" + std::to_string((double)g.start / (double)N) + "\n# stop: " + std::to_string(g.stop) +
                             " => f_stop=" + std::to_string((double)g.stop / (double)N) + "\n# width: " + std::to_string(g.width) +
                             " => f_bw=" + std::to_string((double)g.width / (double)N) + "\n# chan_decimation_fact: " +
                             std::to_string(sp->dec) + "\n");
            }
        }
    }
    *out = raw;
    return FDC_OK;
    FDC_ENTRY_END
}

void *fdc_sinks_spectrum(fdc_sinks *s) { return s ? (void *)(s->d_spec + s->N) : nullptr; }
void *fdc_sinks_stream(fdc_sinks *s) { return s ? (void *)s->stream : nullptr; }
void *fdc_sinks_spectrum_ahead(fdc_sinks *s) { return (s && s->d_spec_ahead) ? (void *)(s->d_spec_ahead + s->N) : nullptr; }
void *fdc_sinks_fill_stream(fdc_sinks *s) { return s ? (void *)s->s_fill : nullptr; }
int fdc_sinks_prepare(fdc_sinks *s, int nblocks, int ahead)
{
    FDC_ENTRY("fdc_sinks_prepare")
    if (!s) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "null handle");
    FDC_DEAD_CHECK(s);
    if (!s->s_fill) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "the bank was created without FDC_SINKS_LOOKAHEAD");
    if (nblocks <= 0 || nblocks > s->cfg.max_blocks) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "nblocks %d outside [1, max_blocks]", nblocks);
    if (!ahead && s->dev.eager_n > 0) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "the batch in the current buffer was prepared before and its decisions are enqueued: submit it");
    HIPCHK(hipSetDevice(s->cfg.device_id));
    float2 *const spec = ahead ? s->d_spec_ahead : s->d_spec;
    float *const pw = ahead ? s->d_power_ahead : s->d_power;
    if (!ahead) {
        // the current buffer may have been filled on the bank's own stream (a batch 0 written there, fdc_sinks_work's copy): the cells wait for it
        HIPCHK(hipEventRecord(s->ev_fill, s->stream));
        HIPCHK(hipStreamWaitEvent(s->s_fill, s->ev_fill, 0));
    }
    if (!s->cells.empty()) HIPCHK(fdc::launch_cell_power(spec + s->N, s->N, s->d_cells, (int)s->cells.size(), nblocks, pw, s->s_fill));
    HIPCHK(hipEventRecord(ahead ? s->ev_ready_ahead : s->ev_ready, s->s_fill));
    (ahead ? s->prepared_ahead : s->prepared) = nblocks;
    return FDC_OK;
    FDC_ENTRY_END
}
void *fdc_sinks_group_power(fdc_sinks *s) { return s ? (void *)s->d_gpow : nullptr; }
void *fdc_sinks_group_power_ahead(fdc_sinks *s) { return s ? (void *)s->d_gpow_ahead : nullptr; }
int fdc_sinks_prepare_from_groups(fdc_sinks *s, int nblocks, int ahead)
{
    FDC_ENTRY("fdc_sinks_prepare_from_groups")
    if (!s) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "null handle");
    FDC_DEAD_CHECK(s);
    if (nblocks <= 0 || nblocks > s->cfg.max_blocks) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "nblocks %d outside [1, max_blocks]", nblocks);
    if (ahead && !s->s_fill) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "the bank was created without FDC_SINKS_LOOKAHEAD");
    if (!ahead && s->dev.eager_n > 0) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "the batch in the current buffer was prepared before and its decisions are enqueued: submit it");
    float *const gp = ahead ? s->d_gpow_ahead : s->d_gpow;
    if (!s->cells.empty() && !gp) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "the bank has no group-power buffer (block length below 16)");
    HIPCHK(hipSetDevice(s->cfg.device_id));
    float2 *const spec = ahead ? s->d_spec_ahead : s->d_spec;
    float *const pw = ahead ? s->d_power_ahead : s->d_power;
    // look-ahead banks: on the fill stream, behind what the producer enqueued there; other banks: on the bank's own stream (the producer
    // wrote spectrum and group powers on a stream it has synchronised, or on this one: fdc_sinks_spectrum's contract)
    hipStream_t q = s->s_fill ? s->s_fill : s->stream;
    if (s->s_fill && !ahead) {
        HIPCHK(hipEventRecord(s->ev_fill, s->stream));
        HIPCHK(hipStreamWaitEvent(s->s_fill, s->ev_fill, 0));
    }
    if (!s->cells.empty()) HIPCHK(fdc::launch_cell_power_groups(spec + s->N, gp, s->N, s->d_cells, (int)s->cells.size(), nblocks, pw, q));
    if (s->s_fill) HIPCHK(hipEventRecord(ahead ? s->ev_ready_ahead : s->ev_ready, s->s_fill));
    (ahead ? s->prepared_ahead : s->prepared) = nblocks;
    return FDC_OK;
    FDC_ENTRY_END
}
int32_t fdc_sinks_blocklen(const fdc_sinks *s) { return s ? s->N : -1; }
int32_t fdc_sinks_max_blocks(const fdc_sinks *s) { return s ? s->cfg.max_blocks : -1; }

int fdc_sinks_pac_params(const fdc_sinks *s, int i, int32_t *v)
{
    FDC_ENTRY("fdc_sinks_pac_params")
    if (!s || i < 0 || i >= (int)s->pacs.size() || !v) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "bad index");
    const Pac &p = s->pacs[i];
    v[0] = p.extract_start; v[1] = p.extract_stop; v[2] = p.extract_width; v[3] = p.measure_start; v[4] = p.measure_stop;
    v[5] = p.output_len; v[6] = p.ovl_offset; v[7] = p.deltaphase;
    return FDC_OK;
    FDC_ENTRY_END
}

int fdc_sinks_segment_params(const fdc_sinks *s, int i, int32_t *v)
{
    FDC_ENTRY("fdc_sinks_segment_params")
    if (!s || i < 0 || i >= (int)s->segs.size() || !v) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "bad index");
    const Segment &g = s->segs[i];
    v[0] = g.start; v[1] = g.stop; v[2] = g.width; v[3] = s->dec; v[4] = g.ncell;
    return FDC_OK;
    FDC_ENTRY_END
}

// Extractions of one call, one launch (or one gather / batched transform / scatter sequence) per width class.
// tasks: grouped by class, class k (width 2^k) = [first[k], first[k] + cnt[k])
// ---- FDC_SINKS_LOOKAHEAD (see the struct): what a batch does at its two ends
// start of a batch: whatever its producer enqueued on the fill stream (forward transform, power cells) comes first
static int batch_begin(fdc_sinks *s, int nblocks, bool *have_power)
{
    *have_power = false;
    if (!s->s_fill) {
        // one-buffer bank: fdc_sinks_prepare_from_groups has (enqueued, on this stream) the cells of exactly this batch
        *have_power = s->prepared == nblocks;
        s->prepared = -1;
        return FDC_OK;
    }
    if (s->prepared == nblocks) {
        // fdc_sinks_prepare marked the point of the fill stream where this batch is complete: what the producer has enqueued there SINCE
        // (the next batch's transform) is not waited for — it is what runs beside this batch's decisions
        HIPCHK(hipStreamWaitEvent(s->stream, s->ev_ready, 0));
        *have_power = true;
    } else {
        HIPCHK(hipEventRecord(s->ev_fill, s->s_fill));          // no mark: everything enqueued on the fill stream so far
        HIPCHK(hipStreamWaitEvent(s->stream, s->ev_fill, 0));
    }
    s->prepared = -1;
    return FDC_OK;
}
// end of a batch (enqueued behind its last reader): history <- its last block (save_hist, PowerActivationChannel_impl.cc:173;
// …vcm_impl.cc:571) — slot 0 of the buffer the NEXT batch is read from, which with look-ahead is the other one ...
static int batch_end_history(fdc_sinks *s, int nblocks, hipStream_t q)
{
    const size_t N = (size_t)s->N;
    float2 *const next = s->d_spec_ahead ? s->d_spec_ahead : s->d_spec;
    HIPCHK(hipMemcpyAsync(next, s->d_spec + (size_t)nblocks * N, sizeof(float2) * N, hipMemcpyDeviceToDevice, q));
    return FDC_OK;
}
// ... then the buffers swap, and the fill stream may overwrite this batch's buffer once `done` (an event on the bank's stream behind the
// history copy; null: everything enqueued on it so far) has passed
// Three spectrum buffers (device engine): the buffer that becomes "ahead" is the one of the batch BEFORE this one; its last reader is `prev_done`
// (have_prev: there was such a batch).  The power and group-power buffers stay a pair: this batch's were last read by its decision chain, which the
// host has seen finish (the summary), and by its cell kernel on the fill stream itself.
static int batch_end_swap(fdc_sinks *s, hipEvent_t done, hipEvent_t prev_done = nullptr, bool have_prev = false)
{
    if (!s->s_fill) return FDC_OK;
    if (s->d_spec_spare && done) {
        if (have_prev) HIPCHK(hipStreamWaitEvent(s->s_fill, prev_done, 0));
        float2 *const cur = s->d_spec;
        s->d_spec = s->d_spec_ahead; s->d_spec_ahead = s->d_spec_spare; s->d_spec_spare = cur;
    } else {
        if (done) HIPCHK(hipStreamWaitEvent(s->s_fill, done, 0));
        else { HIPCHK(hipEventRecord(s->ev_fill, s->stream)); HIPCHK(hipStreamWaitEvent(s->s_fill, s->ev_fill, 0)); }
        std::swap(s->d_spec, s->d_spec_ahead);
    }
    std::swap(s->d_power, s->d_power_ahead);
    std::swap(s->d_gpow, s->d_gpow_ahead);
    std::swap(s->ev_ready, s->ev_ready_ahead);
    s->prepared = s->prepared_ahead; s->prepared_ahead = -1;
    return FDC_OK;
}

static int run_extractions(fdc_sinks *s, const fdc::ExtractTask *d_tasks, const size_t *first, const size_t *cnt, float2 *d_out, bool trace,
                           hipStream_t q0 = nullptr)
{
    const int N = s->N;
    if (!q0) q0 = s->stream;
    // the classes that fit one workgroup's transform at 16 points per lane (w <= 4096; 256 has a kernel of its own) are independent and
    // each fills a part of the device only: two or more of them go out as ONE launch
    int mw[32], nm = 0;
    size_t mfirst[32], mcnt[32];
    for (int k = 0; k < 32; k++) {
        const int w = 1 << k;
        if (cnt[k] && w <= 4096 && !(w == 256 && s->d_tw256)) { mw[nm] = w; mfirst[nm] = first[k]; mcnt[nm] = cnt[k]; nm++; }
    }
    const bool multi = nm >= 2 && nm <= fdc::kMaxExtractClasses;
    // side streams (look-ahead banks): two or three classes above 4096 points, each in one piece of its own slice of the scratch
    int nwide = 0, wk[3] = {0, 0, 0};
    size_t wneed = 0;
    bool side = s->s_side[0] != nullptr;
    for (int k = 13; k < 32 && side; k++)
        if (cnt[k]) {
            if (nwide == 3 || cnt[k] * ((size_t)1 << k) > ((size_t)64 << 20)) { side = false; break; }
            wk[nwide++] = k; wneed += cnt[k] * ((size_t)1 << k);
        }
    side = side && nwide >= 2;
    if (side) {
        if (s->wide_cap < wneed) {
            (void)hipFree(s->d_wide); s->d_wide = nullptr; s->wide_cap = 0;
            HIPCHK(hipMalloc(&s->d_wide, sizeof(float2) * (wneed + wneed / 2)));
            s->wide_cap = wneed + wneed / 2;
        }
        HIPCHK(hipEventRecord(s->ev_fork, q0));
        size_t off = 0;
        for (int c = 0; c < nwide; c++) {
            const int k = wk[c], w = 1 << k;
            hipStream_t q = c == 0 ? q0 : s->s_side[c - 1];
            if (c) HIPCHK(hipStreamWaitEvent(q, s->ev_fork, 0));
            if (c == 0 && multi)        // the classes up to 4096 points go first on the bank's own stream, the widest class behind them
                HIPCHK(fdc::launch_extract_multi(s->d_spec, N, d_tasks, mw, mfirst, mcnt, nm, s->R, s->d_wins, d_out, s->d_tw, N, q0));
            HIPCHK(fdc::launch_extract_wide(s->d_spec, N, d_tasks + first[k], (int)cnt[k], w, w / s->R, s->d_wins, s->d_wide + off, d_out, s->d_tw, N, q));
            off += cnt[k] * (size_t)w;
            if (c) HIPCHK(hipEventRecord(s->ev_join[c - 1], q));
        }
    } else if (multi) HIPCHK(fdc::launch_extract_multi(s->d_spec, N, d_tasks, mw, mfirst, mcnt, nm, s->R, s->d_wins, d_out, s->d_tw, N, q0));
    for (int k = 0; k < 32; k++) {
        if (!cnt[k]) continue;
        const int w = 1 << k, skip = w / s->R;
        const size_t i = first[k], j = first[k] + cnt[k];
        if (trace) std::fprintf(stderr, "[fdc_sinks]     width %d: %zu tasks\n", w, j - i);
        if (multi && w <= 4096 && !(w == 256 && s->d_tw256)) continue;
        if (side && w > 4096) continue;
        if (w == 256 && s->d_tw256) {
            HIPCHK(fdc::launch_extract256(s->d_spec, N, d_tasks + i, (int)(j - i), skip, s->d_wins, d_out, s->d_tw256, q0));
        } else if (w <= 4096) {
            HIPCHK(fdc::launch_extract(s->d_spec, N, d_tasks + i, (int)(j - i), w, skip, s->d_wins, d_out, s->d_tw, N, q0));
        } else {
            // above 4096 points (a carrier, or a run of merged carriers, over 1/16 of a 65536-bin band): the two-pass inverse transform,
            // the whole class in batches of up to 64 Mi points — pass A reads slice * window straight from the spectrum (the half swap is
            // its input rotation), pass B writes [skip, w) to the landing offsets.  The scratch between the passes follows the demand
            // (batch x w points, grown geometrically), not the 64 Mi ceiling.  (One workgroup per 8192-point transform — 32 points per
            // lane, two workgroups per compute unit — was slower than the two passes: 98 us for the 536 extractions of a configs[4] step;
            // that class and the two above it took 271 us with a gathered copy and a scatter around the transform, 212 us this way.)
            const size_t per = std::min(std::max<size_t>(1, ((size_t)64 << 20) / (size_t)w), j - i);
            if (s->wide_cap < per * (size_t)w) {
                const size_t want = std::min(std::max(per * (size_t)w, s->wide_cap * 2), std::max<size_t>((size_t)64 << 20, (size_t)w));
                (void)hipFree(s->d_wide); s->d_wide = nullptr; s->wide_cap = 0;
                HIPCHK(hipMalloc(&s->d_wide, sizeof(float2) * want));
                s->wide_cap = want;
            }
            const size_t fit = std::max<size_t>(1, s->wide_cap / (size_t)w);
            for (size_t k0 = i; k0 < j; k0 += fit) {
                const int n = (int)std::min(fit, j - k0);
                HIPCHK(fdc::launch_extract_wide(s->d_spec, N, d_tasks + k0, n, w, skip, s->d_wins, s->d_wide, d_out, s->d_tw, N, q0));
            }
        }
    }
    if (side) for (int c = 1; c < nwide; c++) HIPCHK(hipStreamWaitEvent(q0, s->ev_join[c - 1], 0));    // the bank's stream goes on behind every class
    return FDC_OK;
}

static int host_work_device(fdc_sinks *s, int nblocks)
{
    static const bool trace = fdc::debug_env("FDC_SINKS_TRACE") != nullptr;      // phase times on stderr (diagnostics)
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto t0 = now();
    auto lap = [&](const char *what) {
        if (!trace) return;
        const auto t1 = now();
        std::fprintf(stderr, "[fdc_sinks] %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count());
        t0 = t1;
    };
    s->pdus.clear();
    lap("previous PDUs released");
    bool pool_ok = true;
    const int N = s->N, ncells = (int)s->cells.size();
    // phase 1: power of every cell of every block
    bool have_power = false;
    { const int rb = batch_begin(s, nblocks, &have_power); if (rb != FDC_OK) return rb; }
    if (ncells) {
        if (!have_power) HIPCHK(fdc::launch_cell_power(s->d_spec + N, N, s->d_cells, ncells, nblocks, s->d_power, s->stream));
        s->h_power.resize((size_t)ncells * nblocks);
        HIPCHK(hipMemcpyAsync(s->h_power.data(), s->d_power, sizeof(float) * s->h_power.size(), hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    lap("cell power + D2H");
    // phase 2: decisions (work() loops of both reference blocks).  Every block: the PowerActivationChannels in order, then the
    // detection segments.  PowerActivationChannel instances do not interact, so a large bank is cut into ranges that worker
    // threads run over the whole batch on their own; the PDUs carry an order key (block, then instance) and are put back
    // into the order the sequential loop emits them in.
    s->tasks.clear(); s->task_w.clear(); s->task_skip.clear(); s->ext_used = 0;
    const int64_t bc0 = s->blockcount;
    const int npac = (int)s->pacs.size();
    int nthr = 1;
    if (npac >= 32 && (int64_t)npac * nblocks >= 16384 && s->cfg.verbose == 0) {
        const unsigned hc = std::thread::hardware_concurrency();
        nthr = (int)std::min<unsigned>(8, std::max<unsigned>(1, hc / 2));
        if (s->host_threads > 0) nthr = s->host_threads;
        nthr = std::min(nthr, npac / 8);
    }
    auto run_pacs = [&](int a, int b, Emit e) {
        for (int m = 0; m < nblocks; m++) {
            const float *P = s->h_power.data() + (size_t)m * ncells;
            e.blockcount = bc0 + m;
            for (int i = a; i < b; i++) {
                e.key = ((int64_t)m << 24) | i;
                pac_step(s, e, s->pacs[(size_t)i], P[s->pacs[(size_t)i].cell], m + 1);
            }
        }
    };
    // the workers' form: one channel at a time over the whole batch (its state stays in registers, its tasks are appended in
    // one run); the PDUs find their place through the order key, the tasks through the landing layout
    auto run_pacs_by_channel = [&](int a, int b, Emit e) {
        const float *P0 = s->h_power.data();
        for (int i = a; i < b; i++) {
            Pac &p = s->pacs[(size_t)i];
            const float *P = P0 + p.cell;
            for (int m = 0; m < nblocks; m++) {
                e.blockcount = bc0 + m;
                e.key = ((int64_t)m << 24) | i;
                pac_step(s, e, p, P[(size_t)m * ncells], m + 1);
            }
        }
    };
    Emit em{&s->tasks, &s->task_w, &s->task_skip, &s->ext_used, &s->pdus, bc0, 0};
    if (nthr > 1) {
        while ((int)s->wl.size() < nthr) s->wl.emplace_back(new WorkerLists());
        std::vector<int> lo((size_t)nthr + 1);
        for (int t = 0; t <= nthr; t++) lo[(size_t)t] = (int)((int64_t)npac * t / nthr);
        std::vector<double> tms((size_t)nthr, 0.0);
        pool_ok = s->pool.run(nthr, [&](int t) {
            const auto a0 = now();
            WorkerLists &L = *s->wl[(size_t)t];
            L.clear();
            run_pacs_by_channel(lo[(size_t)t], lo[(size_t)t + 1], Emit{&L.tasks, &L.w, &L.skip, &L.used, &L.pdus, bc0, 0});
            tms[(size_t)t] = std::chrono::duration<double, std::milli>(now() - a0).count();
        });
        if (trace) { std::fprintf(stderr, "[fdc_sinks]     worker ms:"); for (double v : tms) std::fprintf(stderr, " %.3f", v); std::fprintf(stderr, "\n"); }
        lap("  PAC state machines (threads)");
        // merge: task indices of a worker move up by the number of tasks in front of them (live channels and PDUs alike);
        // every worker moves its own lists into place
        std::vector<int64_t> base((size_t)nthr + 1, (int64_t)s->tasks.size());
        for (int t = 0; t < nthr; t++) {
            base[(size_t)t + 1] = base[(size_t)t] + (int64_t)s->wl[(size_t)t]->tasks.size();
            s->ext_used += s->wl[(size_t)t]->used;
        }
        s->tasks.resize((size_t)base[(size_t)nthr]); s->task_w.resize((size_t)base[(size_t)nthr]); s->task_skip.resize((size_t)base[(size_t)nthr]);
        pool_ok = s->pool.run(nthr, [&](int t) {
            WorkerLists &L = *s->wl[(size_t)t];
            const int64_t b0 = base[(size_t)t];
            std::copy(L.tasks.begin(), L.tasks.end(), s->tasks.begin() + b0);
            std::copy(L.w.begin(), L.w.end(), s->task_w.begin() + b0);
            std::copy(L.skip.begin(), L.skip.end(), s->task_skip.begin() + b0);
            if (b0) {
                for (auto &r : L.pdus) for (auto &bk : r.blocks) if (bk.task >= 0) bk.task += b0;
                for (int i = lo[(size_t)t]; i < lo[(size_t)t + 1]; i++)
                    for (auto &bk : s->pacs[(size_t)i].blocks) if (bk.task >= 0) bk.task += b0;
            }
        });
        for (int t = 0; t < nthr; t++)
            for (auto &r : s->wl[(size_t)t]->pdus) s->pdus.push_back(std::move(r));
    } else if (npac) {
        run_pacs(0, npac, em);
    }
    lap("  PAC total incl. merge");
    const int nseg = (int)s->segs.size();
    const int nthr_s = (nseg >= 2 && (int64_t)nseg * nblocks >= 256 && s->cfg.verbose == 0) ? std::min(nseg, 8) : 1;
    if (nthr_s > 1) {
        // The segments of a detection block do not interact either (…vcm_impl.cc:558-562 loops over them per item): a worker
        // takes whole segments through the batch; the order key (block, then segment, then emission) restores the reference's order.
        while ((int)s->wl.size() < nthr_s) s->wl.emplace_back(new WorkerLists());
        pool_ok = s->pool.run(nthr_s, [&](int t) {
            WorkerLists &L = *s->wl[(size_t)t];
            L.clear();
            Emit e{&L.tasks, &L.w, &L.skip, &L.used, &L.pdus, bc0, 0};
            for (int gi = t; gi < nseg; gi += nthr_s) {
                Segment &g = s->segs[(size_t)gi];
                for (int m = 0; m < nblocks; m++) {
                    e.blockcount = bc0 + m;
                    e.key = ((int64_t)m << 24) | (1 << 23) | ((int64_t)gi << 12);
                    seg_detect(s, g, s->h_power.data() + (size_t)m * ncells + g.cell0);
                    seg_extract(s, e, g, m + 1);
                }
            }
        });
        std::vector<int64_t> base((size_t)nthr_s + 1, (int64_t)s->tasks.size());
        for (int t = 0; t < nthr_s; t++) {
            base[(size_t)t + 1] = base[(size_t)t] + (int64_t)s->wl[(size_t)t]->tasks.size();
            s->ext_used += s->wl[(size_t)t]->used;
        }
        s->tasks.resize((size_t)base[(size_t)nthr_s]); s->task_w.resize((size_t)base[(size_t)nthr_s]); s->task_skip.resize((size_t)base[(size_t)nthr_s]);
        pool_ok = s->pool.run(nthr_s, [&](int t) {
            WorkerLists &L = *s->wl[(size_t)t];
            const int64_t b0 = base[(size_t)t];
            std::copy(L.tasks.begin(), L.tasks.end(), s->tasks.begin() + b0);
            std::copy(L.w.begin(), L.w.end(), s->task_w.begin() + b0);
            std::copy(L.skip.begin(), L.skip.end(), s->task_skip.begin() + b0);
            if (b0) {
                for (auto &r : L.pdus) for (auto &bk : r.blocks) if (bk.task >= 0) bk.task += b0;
                for (int gi = t; gi < nseg; gi += nthr_s)
                    for (auto &c : s->segs[(size_t)gi].chans) for (auto &bk : c.data) if (bk.task >= 0) bk.task += b0;
            }
        });
        for (int t = 0; t < nthr_s; t++)
            for (auto &r : s->wl[(size_t)t]->pdus) s->pdus.push_back(std::move(r));
    } else if (nseg)
        for (int m = 0; m < nblocks; m++) {
            const float *P = s->h_power.data() + (size_t)m * ncells;
            em.blockcount = bc0 + m;
            em.key = ((int64_t)m << 24) | (1 << 23);
            for (auto &g : s->segs) seg_detect(s, g, P + g.cell0);                  // …vcm_impl.cc:558
            for (auto &g : s->segs) seg_extract(s, em, g, m + 1);                   // :562
        }
    s->blockcount = bc0 + nblocks;
    if (!pool_ok) return fdc::set_error(FDC_ERR_NOMEM, "a decision worker failed (out of memory?): the batch is lost");
    if ((npac && (nthr > 1 || nseg)) || nthr_s > 1)
        std::stable_sort(s->pdus.begin(), s->pdus.end(), [](const PduRec &a, const PduRec &b) { return a.key < b.key; });
    lap("decisions (host)");
    // Landing layout: the blocks of every PDU emitted in this call sit one behind the other (PDU order, block order),
    // so a PDU whose blocks all come from this call needs no assembly — its payload IS a run of the landing buffer;
    // blocks that stay buffered in live channels follow.
    {
        std::vector<int64_t> noff(s->tasks.size(), -1);
        int64_t pos = 0;
        for (auto &r : s->pdus)
            for (auto &b : r.blocks)
                if (b.task >= 0) { noff[(size_t)b.task] = pos; pos += r.blocklen; }
        for (size_t i = 0; i < s->tasks.size(); i++)
            if (noff[i] < 0) { noff[i] = pos; pos += s->task_w[i] - s->task_skip[i]; }
        for (size_t i = 0; i < s->tasks.size(); i++) s->tasks[i].out_off = noff[i];
    }
    // phase 3: extractions, one launch per width class
    const size_t nt = s->tasks.size();
    if (nt) {
        // tasks grouped by width: a counting sort over the (at most 25) power-of-two classes, order inside a class kept;
        // nothing to do when every task has the same width (a PowerActivationChannel bank of equal channels)
        size_t cnt[32] = {0}, first[32];
        for (size_t i = 0; i < nt; i++) cnt[31 - __builtin_clz((unsigned)s->task_w[i])]++;
        size_t acc = 0;
        int nclasses = 0;
        for (int k = 0; k < 32; k++) { first[k] = acc; acc += cnt[k]; nclasses += cnt[k] != 0; }
        const fdc::ExtractTask *upload = s->tasks.data();
        if (nclasses > 1) {
            s->sorted.resize(nt);
            size_t pos[32];
            std::copy(first, first + 32, pos);
            for (size_t i = 0; i < nt; i++) s->sorted[pos[31 - __builtin_clz((unsigned)s->task_w[i])]++] = s->tasks[i];
            upload = s->sorted.data();
        }
        if (nt > s->cap_tasks) {
            (void)hipFree(s->d_tasks); s->d_tasks = nullptr; s->cap_tasks = 0;
            HIPCHK(hipMalloc(&s->d_tasks, sizeof(fdc::ExtractTask) * nt * 2));
            s->cap_tasks = nt * 2;
        }
        if ((size_t)s->ext_used > s->cap_hext) {
            if (s->h_ext) (void)hipHostFree(s->h_ext);
            s->h_ext = nullptr; s->cap_hext = 0;
            HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&s->h_ext), sizeof(cfl) * (size_t)s->ext_used * 2, hipHostMallocDefault));
            s->cap_hext = (size_t)s->ext_used * 2;
        }
        if ((size_t)s->ext_used > s->cap_ext) {
            (void)hipFree(s->d_ext); s->d_ext = nullptr; s->cap_ext = 0;
            HIPCHK(hipMalloc(&s->d_ext, sizeof(float2) * (size_t)s->ext_used * 2));
            s->cap_ext = (size_t)s->ext_used * 2;
        }
        lap("  task grouping + buffers");
        HIPCHK(hipMemcpyAsync(s->d_tasks, upload, sizeof(fdc::ExtractTask) * nt, hipMemcpyHostToDevice, s->stream));
        {
            const int rce = run_extractions(s, s->d_tasks, first, cnt, s->d_ext, trace);
            if (rce != FDC_OK) return rce;
        }
        if (trace) { HIPCHK(hipStreamSynchronize(s->stream)); lap("  task upload + extraction kernels"); }
        HIPCHK(hipMemcpyAsync(s->h_ext, s->d_ext, sizeof(float2) * (size_t)s->ext_used, hipMemcpyDeviceToHost, s->stream));
    }
    // history <- last block of this call (save_hist, PowerActivationChannel_impl.cc:173; …vcm_impl.cc:571)
    { const int rh = batch_end_history(s, nblocks, s->stream); if (rh != FDC_OK) return rh; }
    HIPCHK(hipStreamSynchronize(s->stream));
    { const int rh = batch_end_swap(s, nullptr); if (rh != FDC_OK) return rh; }
    lap("extractions + D2H");
    // phase 4: payloads; blocks still buffered in live channels become host copies
    auto resolve = [&](BlockRef &b, int len) {
        if (b.task >= 0) {
            const cfl *src = s->h_ext + s->tasks[(size_t)b.task].out_off;
            b.owned.assign(src, src + len);
            b.task = -1;
        }
    };
    auto finish_pdu = [&](PduRec &r) {
        bool all_here = !r.blocks.empty();
        for (auto &b : r.blocks) if (b.task < 0) { all_here = false; break; }
        if (all_here) {                     // contiguous in the landing buffer by construction (layout above)
            r.meta.nsamples = (int64_t)r.blocks.size() * r.blocklen;
            r.meta.samples = s->h_ext + s->tasks[(size_t)r.blocks.front().task].out_off;
        } else {                            // some blocks were kept from an earlier call: assemble
            r.payload.reserve(r.blocks.size() * (size_t)r.blocklen);
            for (auto &b : r.blocks) {
                const cfl *src = b.task >= 0 ? s->h_ext + s->tasks[(size_t)b.task].out_off : b.owned.data();
                r.payload.insert(r.payload.end(), src, src + r.blocklen);
            }
            r.meta.nsamples = (int64_t)r.payload.size();
            r.meta.samples = r.payload.data();
        }
        r.blocks.clear();
    };
    // PDUs and live channels are independent of each other: a large bank is finished by the worker threads
    const int npdu = (int)s->pdus.size();
    const int nasm = std::max(nthr, nthr_s);
    if (nasm > 1 && (npdu >= 64 || npac >= 64)) {
        pool_ok = s->pool.run(nasm, [&](int t) {
            for (int i = (int)((int64_t)npdu * t / nasm), e = (int)((int64_t)npdu * (t + 1) / nasm); i < e; i++) finish_pdu(s->pdus[(size_t)i]);
            for (int i = (int)((int64_t)npac * t / nasm), e = (int)((int64_t)npac * (t + 1) / nasm); i < e; i++)
                for (auto &b : s->pacs[(size_t)i].blocks) resolve(b, s->pacs[(size_t)i].output_len);
        });
    } else {
        for (auto &r : s->pdus) finish_pdu(r);
        for (auto &p : s->pacs) for (auto &b : p.blocks) resolve(b, p.output_len);
    }
    for (auto &g : s->segs) for (auto &c : g.chans) for (auto &b : c.data) resolve(b, c.outputsamples);
    if (!pool_ok) return fdc::set_error(FDC_ERR_NOMEM, "a payload worker failed (out of memory?): the batch is lost");
    lap("payload assembly");
    if (trace) std::fprintf(stderr, "[fdc_sinks] %zu tasks, %lld samples extracted, %zu PDUs\n", nt, (long long)s->ext_used, s->pdus.size());
    return nblocks;
}

// ---------------------------------------------------------------- device engine: a call
// dev_enqueue(): power cells, decision kernels, layout — nothing here waits for the device.  dev_launch_extractions(): needs
// the summary of the layout kernel (buffer sizes, tasks per width class) on the host, then enqueues the rest: placement of the
// tasks, blocks buffered from the call before, extraction kernels, history block, and the copy of the emitted runs to the host
// on a stream of its own.  dev_complete(): waits for that copy and turns the emission records into fdc_pdu.
// ahead: the chain of the batch that sits PREPARED in the ahead buffer (look-ahead banks, enqueued by the submit in front of it): its power cells are
// marked by ev_ready_ahead
static int dev_enqueue(fdc_sinks *s, int nblocks, bool ahead = false)
{
    auto &d = s->dev;
    const int N = s->N, ncells = (int)s->cells.size(), npac = (int)s->pacs.size(), nseg = (int)s->segs.size();
    const long long now = (long long)time(nullptr), bc0 = s->blockcount;
    bool have_power = false;
    float *const d_power = ahead ? s->d_power_ahead : s->d_power;
    if (ahead) {
        HIPCHK(hipStreamWaitEvent(s->stream, s->ev_ready_ahead, 0));
        s->prepared_ahead = -1;                                    // consumed: after the swap the batch is no longer "prepared", its chain is out
        have_power = true;
    } else { const int rb = batch_begin(s, nblocks, &have_power); if (rb != FDC_OK) return rb; }
    if (ncells && !have_power) HIPCHK(fdc::launch_cell_power(s->d_spec + N, N, s->d_cells, ncells, nblocks, d_power, s->stream));
    HIPCHK(fdc::launch_pac_decide(d_power, ncells, nblocks, d.d_pgeom, d.d_pstate, npac, s->pac_thr, s->cfg.pac_maxblocks, s->R, bc0, now,
                                  d.d_tasks, d.d_pdus, d.d_task_base, d.d_pdu_base, d.d_ntask, d.d_npdu, d.d_owners, s->stream));
    if (nseg) {
        const int sd = s->cfg.det_variant == 1;
        HIPCHK(fdc::launch_det_cands(d_power, ncells, nblocks, d.d_dgeom, nseg, s->dec, s->det_thr, sd, d.d_cand, d.d_cand_base,
                                     d.d_ncand, s->cfg.max_blocks, s->stream));
        fdc::DetParams dp{};
        dp.N = N; dp.R = s->R; dp.dec = s->dec; dp.variant = sd; dp.maxblocks = s->cfg.det_maxblocks; dp.delay = s->cfg.det_deactivation_delay;
        dp.nseg = nseg; dp.npac = npac; dp.nbmax = s->cfg.max_blocks; dp.puffer = s->cfg.window_flank_puffer;
        dp.mb_shift = (dp.maxblocks >= 2 && (dp.maxblocks & (dp.maxblocks - 1)) == 0) ? 31 - __builtin_clz((unsigned)dp.maxblocks) : -1;
        dp.max_cand_cap = 1;
        for (const Segment &g : s->segs) dp.max_cand_cap = std::max(dp.max_cand_cap, g.ncell / 2 + 1);
        dp.segname0 = s->cfg.det_id;
        HIPCHK(fdc::launch_det_track(dp, nblocks, d.d_dgeom, d.d_sst, d.d_live, d.d_live_off, d.d_cand, d.d_cand_base, d.d_ncand, d.d_winoff,
                                     now, d.d_pdus, d.d_pdu_base, d.d_npdu, d.d_owners, d.d_owner_base, d.d_nowner, d.d_detch, d.d_live2,
                                     d.d_error, s->stream));
        HIPCHK(fdc::launch_det_expand(nseg, npac, s->R, d.d_owners, d.d_owner_base, d.d_nowner, d.d_tasks, d.d_task_base, d.d_ntask, s->stream));
    }
    HIPCHK(fdc::launch_sink_layout(d.nlist, d.d_task_base, d.d_pdu_base, d.d_ntask, d.d_npdu, d.d_tasks, d.d_pdus, d.d_pdus_out, d.d_owners,
                                   npac, nseg, d.d_owner_base, d.d_nowner, d.d_pstate, d.d_sst, d.d_live, d.d_live_off, d.d_sum,
                                   d.d_class_fill, d.d_error, s->stream));
    HIPCHK(fdc::launch_sink_publish(d.d_sum, d.d_pdus_out, d.h_sum, d.h_pdus, kEagerPdus, s->stream));
    HIPCHK(hipEventRecord(d.ev_decide, s->stream));
    s->blockcount = bc0 + nblocks;
    return FDC_OK;
}

static int dev_launch_extractions(fdc_sinks *s, int nblocks)
{
    auto &d = s->dev;
    static const bool trace = fdc::debug_env("FDC_SINKS_TRACE") != nullptr;
    const int npac = (int)s->pacs.size(), nseg = (int)s->segs.size();
    HIPCHK(hipEventSynchronize(d.ev_decide));
    const int64_t bc0_this = s->blockcount - nblocks;          // block counter at the start of THIS batch (the next batch's chain may advance it below)
    const int b = d.cur ^ 1;                                   // this call's landing buffer; d.cur still names the previous call's
    const fdc::SinkSummary sum = *d.h_sum;
    if (sum.error) {
        HIPCHK(hipMemsetAsync(d.d_error, 0, sizeof(int32_t), s->stream));
        return fdc::set_error(FDC_ERR_UNSUPPORTED, "detection: more than %d live channels in one segment", fdc::kDetMaxCells);
    }
    d.sum[b] = sum;
    d.recs[b].assign(d.h_pdus, d.h_pdus + std::min(sum.npdu, kEagerPdus));
    if (sum.npdu > kEagerPdus) {
        d.recs[b].resize((size_t)sum.npdu);
        HIPCHK(hipMemcpyAsync(d.recs[b].data() + kEagerPdus, d.d_pdus_out + kEagerPdus, sizeof(fdc::SinkPdu) * (size_t)(sum.npdu - kEagerPdus),
                              hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    const bool devpay = (s->cfg.flags & FDC_SINKS_DEVICE_PAYLOAD) != 0;
    // the buffer last held call k-2: its copy to the host was waited for when that call was completed, its buffered blocks were
    // moved on by call k-1 (enqueued; hipFree waits for the device)
    if ((size_t)sum.used_total > d.cap_land[b]) {
        const size_t want = std::max<size_t>((size_t)sum.used_total * 3 / 2, (size_t)1 << 16);
        (void)hipFree(d.d_land[b]); d.d_land[b] = nullptr; d.cap_land[b] = 0;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&d.d_land[b]), sizeof(float2) * want));
        d.cap_land[b] = want;
    }
    if (!devpay && (size_t)sum.used_a > d.cap_hland[b]) {
        const size_t want = std::max<size_t>((size_t)sum.used_a * 3 / 2, (size_t)1 << 16);
        if (d.h_land[b]) (void)hipHostFree(d.h_land[b]);
        d.h_land[b] = nullptr; d.cap_hland[b] = 0;
        HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&d.h_land[b]), sizeof(cfl) * want, hipHostMallocDefault));
        d.cap_hland[b] = want;
    }
    // look-ahead banks: the extractions run on their own stream.  The batch before's may still be at work there: its landing buffer (the
    // buffered blocks move on from it) and the sorted task list (about to be rewritten) are its to read until it is done
    hipStream_t qx = s->s_x ? s->s_x : s->stream;
    if (s->s_x && d.any) HIPCHK(hipStreamWaitEvent(s->stream, d.ev_extract[d.cur], 0));
    if (sum.ntask)
        // grids from what the layout found, not from the worst case the lists were allocated for
        HIPCHK(fdc::launch_task_scatter(d.nlist, d.d_task_base, d.d_ntask, std::min<long long>(d.max_list, std::max(1, sum.max_list_tasks)), d.d_tasks,
                                        d.d_owners, d.d_sum, d.d_class_fill, d.d_sorted, s->stream));
    if (d.any && sum.ncarry > 0)
        HIPCHK(fdc::launch_carry_copy(d.d_owners, std::min(d.carry_width, std::max(1, sum.max_region_owners)), d.d_owner_base, d.d_nowner, npac, nseg,
                                      d.d_sum, d.d_land[d.cur], d.d_land[b], s->stream));
    if (s->s_x) {
        HIPCHK(hipEventRecord(s->ev_tasks, s->stream));
        HIPCHK(hipStreamWaitEvent(qx, s->ev_tasks, 0));
    }
    // Look-ahead banks: the NEXT batch is already transformed and its power cells are (being) summed (fdc_sinks_prepare(.., ahead) came before this
    // submit): its decision chain goes out NOW, behind this batch's task placement on the bank's stream — the decision arrays are free from there on, the
    // host has its copy of this batch's summary and records — instead of when the caller comes back with the next submit, and in front of this batch's
    // extraction launches: the chain is the long pole of a detector's step (k_det_track: 0.3 ms on two compute units) and used to start a host lap late
    // (profiles/r06/timeline_cfg5_before.txt: 220 us between the end of one k_det_track and the start of the next).  The submit that follows must be
    // for exactly that batch.
#ifndef FDC_SINKS_NO_EAGER                    // (A/B builds)
    if (s->s_fill && s->prepared_ahead > 0) {
        const int n_next = s->prepared_ahead;
        const int re = dev_enqueue(s, n_next, true);
        if (re != FDC_OK) return re;
        d.eager_n = n_next;
    }
#endif
    if (sum.ntask) {
        size_t first[32], cnt[32];
        for (int k = 0; k < 32; k++) { first[k] = (size_t)sum.class_base[k]; cnt[k] = (size_t)sum.class_cnt[k]; }
        const int rce = run_extractions(s, d.d_sorted, first, cnt, d.d_land[b], trace, qx);
        if (rce != FDC_OK) return rce;
    }
    { const int rh = batch_end_history(s, nblocks, qx); if (rh != FDC_OK) return rh; }
    HIPCHK(hipEventRecord(d.ev_extract[b], qx));
    { const int rh = batch_end_swap(s, d.ev_extract[b], d.ev_extract[b ^ 1], d.any); if (rh != FDC_OK) return rh; }
    if (!devpay && sum.used_a) {
        HIPCHK(hipStreamWaitEvent(d.s_copy, d.ev_extract[b], 0));
        HIPCHK(hipMemcpyAsync(d.h_land[b], d.d_land[b], sizeof(float2) * (size_t)sum.used_a, hipMemcpyDeviceToHost, d.s_copy));
        HIPCHK(hipEventRecord(d.ev_copied[b], d.s_copy));
    }
    d.cur = b; d.any = true; d.inflight = true; d.pend[b] = true; d.nb_of[b] = nblocks; d.bc0[b] = bc0_this;
    if (trace) std::fprintf(stderr, "[fdc_sinks dev] %d tasks, %d PDUs, %lld samples emitted, %lld buffered\n", sum.ntask, sum.npdu,
                            (long long)sum.used_a, (long long)(sum.used_total - sum.b_start));
    return FDC_OK;
}

// The PDUs of the batch in landing buffer b become the handle's current PDUs.  Two halves: dev_build() turns the emission records
// into fdc_pdu (host work only: needs the records and the layout, not the payload bytes — it runs while the device works on the
// next batch), dev_wait() waits for the payload (its copy to the host, or the extraction kernels when it stays on the device).
static int dev_wait(fdc_sinks *s, int b)
{
    auto &d = s->dev;
    if (!d.pend[b]) return 0;
    const bool devpay = (s->cfg.flags & FDC_SINKS_DEVICE_PAYLOAD) != 0;
    if (!devpay && d.sum[b].used_a) HIPCHK(hipEventSynchronize(d.ev_copied[b]));
    else HIPCHK(hipEventSynchronize(d.ev_extract[b]));
    d.pend[b] = false;
    d.inflight = d.pend[0] || d.pend[1];
    return d.nb_of[b];
}

static int dev_build(fdc_sinks *s, int b)
{
    auto &d = s->dev;
    if (!d.pend[b]) return 0;
    const bool devpay = (s->cfg.flags & FDC_SINKS_DEVICE_PAYLOAD) != 0;
    std::vector<fdc::SinkPdu> &recs = d.recs[b];
    // emission order = key order; the records stay where they are, (key, index) pairs are sorted (16 bytes a piece instead of 56)
    std::vector<std::pair<int64_t, uint32_t>> &order = d.order;
    order.resize(recs.size());
    for (size_t i = 0; i < recs.size(); i++) order[i] = {recs[i].key, (uint32_t)i};
    std::sort(order.begin(), order.end());
    const char *base = devpay ? reinterpret_cast<const char *>(d.d_land[b]) : reinterpret_cast<const char *>(d.h_land[b]);
    s->pdus.resize(recs.size());
    time_t last_t = (time_t)-1;
    char tbuf[40] = "";
    const bool sd = s->cfg.det_variant == 1;
    for (size_t i = 0; i < recs.size(); i++) {
        const fdc::SinkPdu &r = recs[order[i].second];
        PduRec &o = s->pdus[i];
        o.blocks.clear(); o.payload.clear(); o.key = r.key;
        fdc_pdu &m = o.meta;
        m = fdc_pdu{};
        const bool det = (r.flags >> 16) & 1;
        const int64_t blk = r.key >> 40;                       // block index inside the batch
        int width, vstart, vend;
        if (!det) {
            const Pac &p = s->pacs[(size_t)r.owner];
            // extract_stop, not extract_start + extract_width: they differ after the reference's clamp (…_impl.cc:333-336, :226)
            width = p.extract_width; vstart = p.extract_start; vend = p.extract_stop;
            m.kind = 0; m.source = p.ID; m.has_part = 1;
            m.blockend = d.bc0[b] + blk;                       // blockcount while the item is processed (:226-227)
        } else {
            const int sgi = (int)((r.key >> 28) & 0x7FF);
            width = 1 << ((r.flags >> 8) & 0xFF); vstart = r.vstart; vend = vstart + width;
            m.kind = 1;
            m.source = (sd && s->cfg.det_id >= 0 && s->segs.size() == 1) ? s->cfg.det_id : s->segs[(size_t)sgi].ID;
            m.has_part = (r.flags & 1) ? (r.part > 0) : 1;     // …vcm_impl.cc:419-420
            // the vcm block counts from 1 (…vcm_impl.cc:188), SegmentDetection from 0 (SegmentDetection_impl.cc:118)
            m.blockend = d.bc0[b] + blk - (sd ? 1 : 0);
        }
        o.blocklen = width - width / s->R;
        m.chan_id = r.chan_id; m.finalized = r.flags & 1; m.part = r.part;
        m.rel_bw = (double)width / (double)s->N;
        m.rel_cfreq = (double)(vstart + vend) / 2.0 / (double)s->N;
        m.blockstart = m.blockend - r.count; m.vectorstart = vstart; m.vectorend = vend;
        m.nsamples = (int64_t)(r.q1 - r.q0) * o.blocklen;
        m.samples = m.nsamples ? base + sizeof(float2) * (size_t)r.off : nullptr;
        if ((time_t)r.act_time != last_t) {                    // create_ID() / get_ID_for_msg(): local time of the activation
            last_t = (time_t)r.act_time;
            struct tm tmv;
            localtime_r(&last_t, &tmv);
            strftime(tbuf, sizeof tbuf, "%Y-%m-%d-%H-%M-%S", &tmv);
        }
        // "<time>.PowActChan.<ID>.<n>" / "<time>.DETECTED.<segment>.<n>": thousands per batch, and in device-payload mode this loop is what
        // the step waits for on a slow host — digits by hand instead of snprintf
        {
            char *q = m.id, *const qe = m.id + sizeof m.id - 1;
            auto put = [&](const char *t) { while (*t && q < qe) *q++ = *t++; };
            auto num = [&](int v) {
                char tmp[12]; int n = 0;
                unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
                do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
                if (v < 0 && q < qe) *q++ = '-';
                while (n && q < qe) *q++ = tmp[--n];
            };
            put(tbuf); put(det ? ".DETECTED." : ".PowActChan."); num(m.source); put("."); num(r.chan_id);
            *q = 0;
        }
    }
    return d.nb_of[b];
}

// rc < 0 from a step that may have advanced the bank's state: the handle is dead (see fdc_sinks::poisoned)
static int poison(fdc_sinks *s, int rc)
{
    if (rc < 0 && !s->poisoned) {
        s->poisoned = true;
        try { s->poison_why = fdc_last_error(); } catch (...) {}
    }
    return rc;
}

static int dev_complete(fdc_sinks *s, int b)
{
    const int rc = dev_build(s, b);
    return rc <= 0 ? rc : dev_wait(s, b);
}

int fdc_sinks_submit_device(fdc_sinks *s, int nblocks)
{
    FDC_ENTRY("fdc_sinks_submit_device")
    if (!s) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "null handle");
    FDC_DEAD_CHECK(s);
    if (nblocks < 0 || nblocks > s->cfg.max_blocks) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "nblocks %d outside [0, max_blocks]", nblocks);
    if (!s->dev.on) {                                          // host engine: the batch is done when the call returns
        if (nblocks == 0) { s->pdus.clear(); return 0; }
        HIPCHK(hipSetDevice(s->cfg.device_id));
        return poison(s, host_work_device(s, nblocks));
    }
    HIPCHK(hipSetDevice(s->cfg.device_id));
    auto &d = s->dev;
    if (nblocks == 0) {
        const int done = poison(s, dev_complete(s, d.cur));
        if (done == 0) s->pdus.clear();
        return done;
    }
    // This batch's decision kernels are enqueued; while the device runs them the PDUs of the batch before are built (host work);
    // then the one wait for the layout summary, the extractions of this batch, and last the wait for the payload of the batch
    // before — its copy has been running beside all of that.
    // From the first launch of dev_enqueue on, the channel state on the device, the block counter and the layout of this batch's
    // landing buffer have moved on: a failure anywhere below leaves them ahead of the host's bookkeeping (d.cur, d.pend), so it
    // poisons the handle instead of returning an error that invites a retry.
    int rc;
    if (d.eager_n > 0) {
        // the chain of this batch went out at the end of the submit before (dev_launch_extractions): the state has advanced for exactly eager_n blocks
        if (nblocks != d.eager_n) {
            fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "a batch of %d blocks was prepared ahead and its decisions are enqueued: the submit must be for it, not for %d blocks", d.eager_n, nblocks);
            return poison(s, FDC_ERR_INVALID_ARGUMENT);
        }
        d.eager_n = 0;
    } else {
        rc = poison(s, dev_enqueue(s, nblocks));
        if (rc != FDC_OK) return rc;
    }
    const int before = d.cur;
    const bool had = d.pend[before];
    if (had) { rc = poison(s, dev_build(s, before)); if (rc < 0) return rc; }
    else s->pdus.clear();
    rc = poison(s, dev_launch_extractions(s, nblocks));
    if (rc != FDC_OK) return rc;
    return had ? poison(s, dev_wait(s, before)) : 0;
    FDC_ENTRY_END
}

int fdc_sinks_flush(fdc_sinks *s)
{
    FDC_ENTRY("fdc_sinks_flush")
    if (!s) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "null handle");
    FDC_DEAD_CHECK(s);
    if (!s->dev.on || !s->dev.inflight) return 0;
    HIPCHK(hipSetDevice(s->cfg.device_id));
    return poison(s, dev_complete(s, s->dev.cur));
    FDC_ENTRY_END
}

int32_t fdc_sinks_engine(const fdc_sinks *s) { return s ? (s->dev.on ? 1 : 0) : -1; }

int fdc_sinks_work_device(fdc_sinks *s, int nblocks)
{
    FDC_ENTRY("fdc_sinks_work_device")
    if (!s) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "null handle");
    FDC_DEAD_CHECK(s);
    if (nblocks < 0 || nblocks > s->cfg.max_blocks) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "nblocks %d outside [0, max_blocks]", nblocks);
    if (s->dev.on && s->dev.inflight) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "a submitted batch is in flight: fdc_sinks_flush() first");
    if (nblocks == 0) { s->pdus.clear(); return 0; }
    HIPCHK(hipSetDevice(s->cfg.device_id));
    if (!s->dev.on) return poison(s, host_work_device(s, nblocks));
    int rc = fdc_sinks_submit_device(s, nblocks);
    if (rc < 0) return rc;
    rc = poison(s, dev_complete(s, s->dev.cur));
    return rc < 0 ? rc : nblocks;
    FDC_ENTRY_END
}

int fdc_sinks_work(fdc_sinks *s, const void *spectrum, int nitems)
{
    FDC_ENTRY("fdc_sinks_work")
    if (!s) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "null handle");
    FDC_DEAD_CHECK(s);
    if (nitems < 0 || nitems > s->cfg.max_blocks) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "nitems %d outside [0, max_blocks]", nitems);
    if (s->dev.on && s->dev.inflight) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "a submitted batch is in flight: fdc_sinks_flush() first");
    if (s->dev.eager_n > 0) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "a batch prepared ahead has its decisions enqueued: submit it before feeding the bank from the host");
    if (nitems == 0) { s->pdus.clear(); return 0; }
    if (!spectrum) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "null buffer");
    HIPCHK(hipSetDevice(s->cfg.device_id));
    s->prepared = -1;                                          // the buffer is overwritten: power cells an earlier fdc_sinks_prepare left for it are stale
    HIPCHK(hipMemcpyAsync(s->d_spec + s->N, spectrum, sizeof(float2) * (size_t)nitems * s->N, hipMemcpyHostToDevice, s->stream));
    return fdc_sinks_work_device(s, nitems);
    FDC_ENTRY_END
}

int fdc_sinks_read_band(const fdc_sinks *s, int32_t *lo, int32_t *hi)
{
    FDC_ENTRY("fdc_sinks_read_band")
    if (!s || !lo || !hi) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "null argument");
    int a = s->N, b = 0;
    for (const Pac &p : s->pacs) {
        a = std::min(a, std::min(p.extract_start, p.measure_start));
        b = std::max(b, std::max(std::max(p.extract_stop, p.extract_start + p.extract_width), p.measure_stop));
    }
    for (const Segment &g : s->segs) {
        // a detected channel is at most the segment wide; its extraction is the next power of two above width * (1 + 2 puffer),
        // centred on the channel and clamped to the block (seg_detect): it can reach half that width beyond either end
        const int ew = pow2ceil((int)std::ceil((double)g.width * (1.0 + 2.0 * s->cfg.window_flank_puffer)));
        a = std::min(a, g.start - ew);
        b = std::max(b, g.stop + ew);
    }
    if (b <= a) { a = 0; b = 0; }
    *lo = std::max(0, a); *hi = std::min(s->N, b);
    return FDC_OK;
    FDC_ENTRY_END
}

int fdc_sinks_work_band(fdc_sinks *s, const void *spectrum, int nitems, int32_t bin_lo, int32_t bin_hi)
{
    FDC_ENTRY("fdc_sinks_work_band")
    if (!s) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "null handle");
    FDC_DEAD_CHECK(s);
    if (nitems < 0 || nitems > s->cfg.max_blocks) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "nitems %d outside [0, max_blocks]", nitems);
    if (s->dev.on && s->dev.inflight) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "a submitted batch is in flight: fdc_sinks_flush() first");
    if (s->dev.eager_n > 0) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "a batch prepared ahead has its decisions enqueued: submit it before feeding the bank from the host");
    if (bin_lo < 0 || bin_hi > s->N || bin_lo > bin_hi) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "band [%d, %d) outside the block", bin_lo, bin_hi);
    int32_t need_lo = 0, need_hi = 0;
    fdc_sinks_read_band(s, &need_lo, &need_hi);
    if (need_hi > need_lo && (bin_lo > need_lo || bin_hi < need_hi))
        return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "band [%d, %d) does not cover what the bank reads, [%d, %d)", bin_lo, bin_hi, need_lo, need_hi);
    if (nitems == 0) { s->pdus.clear(); return 0; }
    if (!spectrum) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "null buffer");
    HIPCHK(hipSetDevice(s->cfg.device_id));
    s->prepared = -1;                                          // as in fdc_sinks_work
    if (bin_hi > bin_lo) {
        const size_t pitch = sizeof(float2) * (size_t)s->N;
        HIPCHK(hipMemcpy2DAsync(s->d_spec + s->N + bin_lo, pitch, static_cast<const float2 *>(spectrum) + bin_lo, pitch,
                                sizeof(float2) * (size_t)(bin_hi - bin_lo), (size_t)nitems, hipMemcpyHostToDevice, s->stream));
    }
    return fdc_sinks_work_device(s, nitems);
    FDC_ENTRY_END
}

int fdc_sinks_pdu_emit_items(const fdc_sinks *s, int32_t *item, int cap)
{
    if (!s) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "null handle");
    const int n = (int)s->pdus.size();
    // the order key of a PDU starts with the index of the item that emitted it: bits 24.. on the host engine, 40.. on the device engine
    const int sh = s->dev.on ? 40 : 24;
    for (int i = 0; i < n && i < cap; i++) item[i] = (int32_t)(s->pdus[(size_t)i].key >> sh);
    return n;
}

int fdc_sinks_pdu_emit_order(const fdc_sinks *s, int32_t *item, int32_t *pac, int cap)
{
    FDC_ENTRY("fdc_sinks_pdu_emit_order")
    if (!s) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "null handle");
    const int n = (int)s->pdus.size();
    // the order key: item index in the high bits (above), and for a PowerActivationChannel its index in this bank's list in the low ones
    // (host engine: bits 0..22, device engine: 0..38; detections carry bit 23 / 39 and their own sub-order)
    const int sh = s->dev.on ? 40 : 24;
    for (int i = 0; i < n && i < cap; i++) {
        const PduRec &r = s->pdus[(size_t)i];
        if (item) item[i] = (int32_t)(r.key >> sh);
        if (pac) pac[i] = r.meta.kind == 0 ? (int32_t)(r.key & ((1ll << (sh - 1)) - 1)) : -1;
    }
    return n;
    FDC_ENTRY_END
}

void fdc_set_log_callback(fdc_log_fn fn, void *user) { std::lock_guard<std::mutex> g(g_log_mu); g_log_fn = fn; g_log_user = user; }

int fdc_sinks_pdu_count(const fdc_sinks *s) { return s ? (int)s->pdus.size() : 0; }

int fdc_sinks_pdu(const fdc_sinks *s, int i, fdc_pdu *out)
{
    FDC_ENTRY("fdc_sinks_pdu")
    if (!s || !out || i < 0 || i >= (int)s->pdus.size()) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "bad PDU index");
    *out = s->pdus[(size_t)i].meta;
    return FDC_OK;
    FDC_ENTRY_END
}

int fdc_sinks_pdus(const fdc_sinks *s, fdc_pdu *out, int cap)
{
    FDC_ENTRY("fdc_sinks_pdus")
    if (!s || (cap > 0 && !out)) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "bad PDU array");
    const int n = (int)s->pdus.size();
    for (int i = 0; i < n && i < cap; i++) out[i] = s->pdus[(size_t)i].meta;
    return n;
    FDC_ENTRY_END
}

}  // extern "C"
