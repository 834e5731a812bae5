// A stand-in for the part of GNU Radio's runtime that decides HOW MANY ITEMS a sync_block's work() is offered: buffer
// allocation (flat_flowgraph::allocate_buffer) and one iteration of the thread-per-block executor
// (block_executor::run_one_iteration with min_available_space).  GNU Radio is not installed on the build or GPU boxes and
// its sources are not available here, so the rules below are restated from the 3.7 / 3.8 runtime as the author knows it
// (gnuradio-runtime/lib/flat_flowgraph.cc, block_executor.cc, vmcircbuf*.cc); each rule is marked [GR].  It exists to
// answer one question honestly: what does gr::FDC::fdc_pipeline_vcc see from a STOCK scheduler, and do the block's own
// requests (set_output_multiple / set_min_output_buffer) get it device-sized batches?
//
//   source (free-running or copying)  ->  block under test  ->  one null sink per output port
//
// [GR] buffers: nitems = 2 * 32 KiB / item_size ("s_fixed_buffer_size", doubled because buffers are only filled half
//      way); at least 2 * output_multiple of the writing block; at least min_output_buffer(port) if the block set one;
//      at least 2 * (decimation * output_multiple + history) of every reading block; rounded up so that the byte size
//      is a multiple of the page size (vmcircbuf granularity).  The memory is mapped twice back to back, so any window
//      of up to nitems items is contiguous for work().
// [GR] one iteration: noutput = min over outputs of min(space_available, bufsize / 2) rounded down to output_multiple
//      (0 -> blocked on output; below min_noutput_items -> blocked); capped by max_noutput_items (which allocate_buffer
//      sets to the buffer size unless the block set it itself); a fixed-rate block is offered what its input holds:
//      noutput = round_down(items available - history + 1, output_multiple) if that is smaller; if the input cannot
//      cover noutput the executor halves it until it fits or falls below output_multiple (-> blocked on input).
#pragma once
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include "sync_block.h"

#include <sys/mman.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <numeric>
#include <stdexcept>
#include <thread>

namespace gr {
namespace compat {

// gr::vmcircbuf: one shared-memory object mapped twice at consecutive addresses
class vmcirc {
public:
    vmcirc(long nitems, size_t item_size) : d_item(item_size)
    {
        const size_t page = (size_t)sysconf(_SC_PAGESIZE);
        const size_t min_items = page / std::gcd(item_size, page);          // [GR] buffer.cc: minimum_buffer_items
        d_nitems = (long)(((size_t)nitems + min_items - 1) / min_items * min_items);
        d_bytes = (size_t)d_nitems * item_size;
        const int fd = memfd_create("fdc-vmcirc", 0);
        if (fd < 0 || ftruncate(fd, (off_t)d_bytes) != 0) { if (fd >= 0) close(fd); throw std::runtime_error("vmcirc: memfd"); }
        void *area = mmap(nullptr, 2 * d_bytes, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (area == MAP_FAILED) { close(fd); throw std::runtime_error("vmcirc: reserve"); }
        d_base = static_cast<char *>(area);
        if (mmap(d_base, d_bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_FIXED, fd, 0) == MAP_FAILED ||
            mmap(d_base + d_bytes, d_bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_FIXED, fd, 0) == MAP_FAILED) {
            munmap(area, 2 * d_bytes); close(fd); throw std::runtime_error("vmcirc: map");
        }
        close(fd);
    }
    ~vmcirc() { if (d_base) munmap(d_base, 2 * d_bytes); }
    vmcirc(const vmcirc &) = delete;
    vmcirc &operator=(const vmcirc &) = delete;
    char *base() const { return d_base; }
    long nitems() const { return d_nitems; }
    size_t item_size() const { return d_item; }
    size_t bytes() const { return d_bytes; }                      // of ONE mapping; 2 * bytes() are addressable
    char *at(uint64_t item_index) const { return d_base + (size_t)(item_index % (uint64_t)d_nitems) * d_item; }
private:
    char *d_base = nullptr;
    long d_nitems = 0;
    size_t d_item, d_bytes = 0;
};

// [GR] flat_flowgraph::allocate_buffer for output `port` of `writer` read by blocks with (multiple, history, decimation 1)
inline long stock_buffer_items(int item_size, int writer_multiple, long writer_min_out, int reader_multiple, int reader_history)
{
    long nitems = 2 * 32768 / (item_size > 0 ? item_size : 1);
    if (nitems < 2L * writer_multiple) nitems = 2L * writer_multiple;
    if (writer_min_out > 0) {
        nitems = std::max(nitems, writer_min_out);
        nitems -= nitems % writer_multiple;
    }
    nitems = std::max(nitems, 2L * (reader_multiple + reader_history));
    return nitems < 1 ? 1 : nitems;
}

struct stock_result {
    long items = 0, calls = 0, min_call = 0, max_call = 0;
    long in_buffer_items = 0;
    std::vector<long> out_buffer_items;
    double wall_seconds = 0, work_seconds = 0;
    bool pinned = false;
    int status = 0;                      // 0 = all items went through; -1 = work() returned WORK_DONE early
};

// Runs `blk` behind a source that offers `total_items` items (the `src_items` items at `src` over and over; copy = false:
// the source only moves its write pointer — a source that costs nothing, for throughput) and null sinks.  `capture`,
// if given, receives every output port's items in order (verification runs).  `ports(in, outs)` is called once the
// buffers exist and before start(): the place where the real block would read detail()->input(i)->buffer().
template <class Block, class PortsFn>
stock_result run_stock(Block &blk, const void *src, long src_items, long total_items, bool copy,
                       std::vector<std::vector<char>> *capture, PortsFn ports)
{
    stock_result res;
    const int in_item = blk.input_signature()->sizeof_stream_item(0);
    const int nout = blk.output_signature()->max_streams();
    // upstream = the source: output_multiple 1, no min_output_buffer; its reader is blk
    vmcirc inbuf(stock_buffer_items(in_item, 1, -1, blk.output_multiple(), (int)blk.history()), (size_t)in_item);
    std::vector<std::unique_ptr<vmcirc>> outbuf;
    for (int c = 0; c < nout; c++) {
        const int sz = blk.output_signature()->sizeof_stream_item(c);
        outbuf.emplace_back(new vmcirc(stock_buffer_items(sz, blk.output_multiple(), blk.min_output_buffer((size_t)c), 1, 1), (size_t)sz));
    }
    // [GR] allocate_buffer: max_noutput_items defaults to the (last allocated) output buffer's size
    if (!blk.is_set_max_noutput_items() && nout > 0) blk.set_max_noutput_items((int)outbuf.back()->nitems());
    res.in_buffer_items = inbuf.nitems();
    for (auto &b : outbuf) res.out_buffer_items.push_back(b->nitems());
    if (capture) { capture->assign((size_t)nout, {}); for (int c = 0; c < nout; c++) (*capture)[(size_t)c].reserve((size_t)total_items * outbuf[(size_t)c]->item_size()); }
    {
        std::vector<std::pair<void *, size_t>> iv{{inbuf.base(), 2 * inbuf.bytes()}}, ov;
        for (auto &b : outbuf) ov.emplace_back(b->base(), 2 * b->bytes());
        res.pinned = ports(iv, ov);
    }
    if (!blk.start()) throw std::runtime_error("run_stock: start() failed");

    std::mutex mu;
    std::condition_variable cv;
    uint64_t written = 0, read = 0;              // input buffer, in items
    bool src_done = false, blk_done = false;
    std::thread source([&] {
        long left = total_items;
        uint64_t pos = 0;
        while (left > 0) {
            long n;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return blk_done || (long)(inbuf.nitems() - 1 - (long)(written - read)) > 0; });
                if (blk_done) break;
                const long space = inbuf.nitems() - 1 - (long)(written - read);
                n = std::min({space, inbuf.nitems() / 2, left});                 // [GR] a writer fills at most half its buffer per call
                pos = written;
            }
            if (copy) {
                for (long i = 0; i < n; i++)
                    std::memcpy(inbuf.at(pos + (uint64_t)i), static_cast<const char *>(src) + (size_t)((pos + (uint64_t)i) % (uint64_t)src_items) * (size_t)in_item, (size_t)in_item);
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                written += (uint64_t)n;
            }
            cv.notify_all();
            left -= n;
        }
        { std::lock_guard<std::mutex> lk(mu); src_done = true; }
        cv.notify_all();
    });

    const auto t0 = std::chrono::steady_clock::now();
    std::vector<uint64_t> owritten((size_t)nout, 0);
    const int mult = blk.output_multiple();
    for (;;) {
        // [GR] min_available_space: the null sinks have consumed everything, so space_available = bufsize - 1
        long noutput = blk.is_set_max_noutput_items() ? (long)blk.max_noutput_items() : (1L << 30);
        noutput = std::max(noutput, (long)mult);                                  // [GR] "overrule the max_noutput_items setting"
        for (int c = 0; c < nout; c++) {
            const long avail = (outbuf[(size_t)c]->nitems() - 1) / mult * mult, best = (outbuf[(size_t)c]->nitems() / 2) / mult * mult;
            if (best < std::max(1, blk.min_noutput_items())) throw std::runtime_error("run_stock: Buffer too small for min_noutput_items");
            noutput = std::min(noutput, std::min(avail, best));
        }
        long avail_in;
        bool done_in;
        {
            std::unique_lock<std::mutex> lk(mu);
            const long need = std::max((long)mult, (long)std::max(1, blk.min_noutput_items()));
            cv.wait(lk, [&] { return src_done || (long)(written - read) >= need; });
            avail_in = (long)(written - read);
            done_in = src_done;
        }
        long reqd = (avail_in - (long)blk.history() + 1) / mult * mult;            // [GR] fixed_rate_ninput_to_noutput, rounded to the multiple
        if (reqd > 0 && reqd <= noutput) noutput = reqd;
        while (noutput > avail_in && noutput >= mult) noutput = (noutput >> 1) / mult * mult;   // [GR] "try doing less work"
        if (noutput < std::max(mult, std::max(1, blk.min_noutput_items())) || noutput > avail_in) {
            if (done_in) break;                  // input done and less than one multiple left: the tail is not processed [GR]
            continue;                            // blocked on input: wait again
        }
        gr_vector_const_void_star in{inbuf.at(read)};
        gr_vector_void_star out;
        for (int c = 0; c < nout; c++) out.push_back(outbuf[(size_t)c]->at(owritten[(size_t)c]));
        const auto w0 = std::chrono::steady_clock::now();
        const int n = blk.work((int)noutput, in, out);
        res.work_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
        if (n < 0) { res.status = -1; break; }
        if (capture)
            for (int c = 0; c < nout; c++) {
                const char *p = static_cast<const char *>(out[(size_t)c]);
                (*capture)[(size_t)c].insert((*capture)[(size_t)c].end(), p, p + (size_t)n * outbuf[(size_t)c]->item_size());
            }
        for (int c = 0; c < nout; c++) owritten[(size_t)c] += (uint64_t)n;
        { std::lock_guard<std::mutex> lk(mu); read += (uint64_t)n; }
        cv.notify_all();
        res.items += n; res.calls++;
        res.min_call = res.calls == 1 ? n : std::min(res.min_call, (long)n);
        res.max_call = std::max(res.max_call, (long)n);
    }
    res.wall_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    { std::lock_guard<std::mutex> lk(mu); blk_done = true; }
    cv.notify_all();
    source.join();
    blk.stop();
    return res;
}

}  // namespace compat
}  // namespace gr
