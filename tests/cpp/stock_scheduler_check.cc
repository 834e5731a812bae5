// CPU check of the stock-scheduler stand-in (gr-fdc_amd/csrc/gr_blocks/compat/gnuradio/stock_scheduler.h) with a block that
// copies its items: buffer sizes by GNU Radio's rule, items per work() call, and that every item arrives intact through the
// double-mapped circular buffers.  usage: stock_scheduler_check <item_bytes> <output_multiple or 0> <total_items>
#include "compat/gnuradio/stock_scheduler.h"

#include <cstdio>
#include <cstdlib>

struct copy_block : gr::sync_block {
    size_t item;
    copy_block(int item_bytes, int multiple)
        : gr::sync_block("copy", gr::io_signature::make(1, 1, item_bytes), gr::io_signature::make(1, 1, item_bytes)), item((size_t)item_bytes)
    {
        if (multiple > 0) { set_output_multiple(multiple); set_min_output_buffer(0, 2L * multiple); set_max_noutput_items(multiple); }
    }
    int work(int n, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        std::memcpy(out[0], in[0], (size_t)n * item);
        return n;
    }
};

int main(int argc, char **argv)
{
    if (argc < 4) return 2;
    const int item = std::atoi(argv[1]), mult = std::atoi(argv[2]);
    const long total = std::atol(argv[3]);
    copy_block blk(item, mult);
    std::vector<char> src((size_t)total * (size_t)item);
    unsigned x = 12345;
    for (auto &c : src) { x = x * 1664525u + 1013904223u; c = (char)(x >> 24); }
    std::vector<std::vector<char>> cap;
    auto ports = [](const std::vector<std::pair<void *, size_t>> &, const std::vector<std::pair<void *, size_t>> &) { return false; };
    const gr::compat::stock_result r = gr::compat::run_stock(blk, src.data(), total, total, true, &cap, ports);
    const bool intact = cap.size() == 1 && cap[0].size() == (size_t)r.items * (size_t)item && std::memcmp(cap[0].data(), src.data(), cap[0].size()) == 0;
    std::printf("{\"in_buffer_items\": %ld, \"out_buffer_items\": %ld, \"items\": %ld, \"calls\": %ld, \"min_call\": %ld, \"max_call\": %ld, \"intact\": %s}\n",
                r.in_buffer_items, r.out_buffer_items[0], r.items, r.calls, r.min_call, r.max_call, intact ? "true" : "false");
    return intact ? 0 : 1;
}
