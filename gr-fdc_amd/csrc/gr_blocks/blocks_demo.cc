// Exercises the C++ gr::FDC faces the way the GNU Radio scheduler would (make(), then work() on item buffers) and
// writes the results as raw files that tests/test_cpp_blocks_gpu.py compares with the oracle.
//   blocks_demo <dir>   reads <dir>/x.c64 (stream), <dir>/spec.c64 (spectrum items); writes <dir>/*.out
#include "FDC/overlap_save.h"
#include "FDC/vector_cut_vxx.h"
#include "FDC/phase_shifting_windowing_vcc.h"
#include "FDC/PowerActivationChannel.h"
#include "FDC/activity_detection_channelizer_vcm.h"
#include "FDC/SegmentDetection.h"
#include "FDC/fdc_pipeline_vcc.h"
#ifndef FDC_HAVE_GNURADIO
#include "compat/gnuradio/stock_scheduler.h"
#endif

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <stdexcept>

using namespace gr::FDC;

static std::vector<gr_complex> slurp(const std::string &fn)
{
    std::ifstream f(fn, std::ios::binary | std::ios::ate);
    std::vector<gr_complex> v((size_t)f.tellg() / sizeof(gr_complex));
    f.seekg(0);
    f.read(reinterpret_cast<char *>(v.data()), (std::streamsize)(v.size() * sizeof(gr_complex)));
    return v;
}
static void dump(const std::string &fn, const std::vector<gr_complex> &v)
{
    std::ofstream f(fn, std::ios::binary);
    f.write(reinterpret_cast<const char *>(v.data()), (std::streamsize)(v.size() * sizeof(gr_complex)));
}
template <class B> static std::vector<gr_complex> run(B &blk, const std::vector<gr_complex> &in, int nitems, size_t outlen)
{
    std::vector<gr_complex> out(outlen * (size_t)nitems);
    gr_vector_const_void_star i{in.data()};
    gr_vector_void_star o{out.data()};
    if (blk->work(nitems, i, o) != nitems) throw std::runtime_error("work() did not consume all items");
    return out;
}

#ifndef FDC_HAVE_GNURADIO
// blocks_demo stock <blocklen> <relinvovl> <channels> <max_items> <total_items> [scheduler_batch] [verify]
// fdc_pipeline_vcc (a bank of <channels> 256-bin channels on the 256-bin grid) behind the stock-scheduler stand-in
// (compat/gnuradio/stock_scheduler.h): buffers sized by GNU Radio's rule from what the block asks for, work() offered what fits.
// Prints one JSON line: buffer sizes, items per call, Gsample/s through work().  "verify": the source copies a seeded stream, the
// outputs are captured and compared bit for bit with ONE work() call over the same stream on a second handle.
static int stock_main(int argc, char **argv)
{
    if (argc < 7) { std::cerr << "usage: blocks_demo stock <blocklen> <relinvovl> <channels> <max_items> <total_items> [batch] [verify]" << std::endl; return 2; }
    const int N = std::atoi(argv[2]), R = std::atoi(argv[3]), C = std::atoi(argv[4]), max_items = std::atoi(argv[5]);
    const long total = std::atol(argv[6]);
    const int batch = argc > 7 ? std::atoi(argv[7]) : 0;
    const bool verify = argc > 8 && std::string(argv[8]) == "verify";
    const int H = N - N / R;
    std::vector<std::vector<float>> chans;
    for (int c = 0; c < C; c++) chans.push_back({256.f * (float)c, 256.f, 0.88f, 1.0f});
    try {
        auto pipe = fdc_pipeline_vcc::make(N, R, chans, 1, max_items);
        if (batch > 0) pipe->set_scheduler_batch(batch);
        // the source's items: a seeded stream when verifying (small), one item's worth of noise otherwise (the free-running source
        // only moves its write pointer; the buffer holds whatever the first lap left there)
        const long src_items = verify ? total : 1;
        std::vector<gr_complex> x((size_t)src_items * (size_t)H);
        unsigned long long st = 88172645463325252ull;
        for (auto &v : x) {
            st ^= st << 13; st ^= st >> 7; st ^= st << 17;
            v = gr_complex((float)((st & 0xFFFF) / 32768.0 - 1.0), (float)(((st >> 16) & 0xFFFF) / 32768.0 - 1.0));
        }
        std::vector<std::vector<char>> cap;
        auto ports = [&](const std::vector<std::pair<void *, size_t>> &in, const std::vector<std::pair<void *, size_t>> &out) {
            bool ok = true;                           // what start() does against GNU Radio: detail()->input(i)->buffer(), ->output(i)
            for (auto &b : in) ok = pipe->pin_buffer(b.first, b.second) && ok;
            for (auto &b : out) ok = pipe->pin_buffer(b.first, b.second) && ok;
            return ok;
        };
        const gr::compat::stock_result r = gr::compat::run_stock(*pipe, x.data(), src_items, total, verify, verify ? &cap : nullptr, ports);
        pipe->unpin_buffers();
        long mismatched = -1;
        if (verify && r.items > 0) {
            auto ref = fdc_pipeline_vcc::make(N, R, chans, 1, (int)r.items);
            std::vector<std::vector<gr_complex>> ro((size_t)C);
            gr_vector_const_void_star pi{x.data()};
            gr_vector_void_star pv;
            for (int c = 0; c < C; c++) { ro[(size_t)c].resize((size_t)r.items * (size_t)ref->output_item_len(c)); pv.push_back(ro[(size_t)c].data()); }
            if (ref->work((int)r.items, pi, pv) != (int)r.items) throw std::runtime_error("stock: reference work() failed");
            mismatched = 0;
            for (int c = 0; c < C; c++)
                if (cap[(size_t)c].size() != ro[(size_t)c].size() * sizeof(gr_complex) ||
                    std::memcmp(cap[(size_t)c].data(), ro[(size_t)c].data(), cap[(size_t)c].size()) != 0) mismatched++;
        }
        std::printf("{\"mode\": \"stock scheduler stand-in\", \"blocklen\": %d, \"relinvovl\": %d, \"channels\": %d, \"max_items\": %d, "
                    "\"scheduler_batch\": %d, \"in_buffer_items\": %ld, \"out_buffer_items\": %ld, \"pinned\": %s, \"items\": %ld, \"calls\": %ld, "
                    "\"items_per_call_min\": %ld, \"items_per_call_max\": %ld, \"wall_s\": %.6f, \"work_s\": %.6f, "
                    "\"gsamples_per_s_wall\": %.4f, \"gsamples_per_s_in_work\": %.4f, \"status\": %d, \"channels_mismatched\": %ld, "
                    "\"items_offered\": %ld, \"items_left_unprocessed\": %ld, \"plan\": \"%s\"}\n",
                    N, R, C, pipe->max_items(), pipe->scheduler_batch(), r.in_buffer_items, r.out_buffer_items.empty() ? 0L : r.out_buffer_items[0],
                    r.pinned ? "true" : "false", r.items, r.calls, r.min_call, r.max_call, r.wall_seconds, r.work_seconds,
                    r.wall_seconds > 0 ? (double)r.items * H / r.wall_seconds / 1e9 : 0.0, r.work_seconds > 0 ? (double)r.items * H / r.work_seconds / 1e9 : 0.0,
                    r.status, mismatched, total, total - r.items /* the tail an output multiple leaves behind when a finite stream drains */,
                    pipe->kernel_plan().c_str());
        return (r.status == 0 && mismatched <= 0) ? 0 : 1;
    } catch (const std::exception &e) {
        std::cerr << "blocks_demo stock: " << e.what() << std::endl;
        return 1;
    }
}
#endif

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
#ifndef FDC_HAVE_GNURADIO
    if (std::string(argv[1]) == "stock") return stock_main(argc, argv);
#endif
    const std::string dir = argv[1];
    const int N = 1024, R = 4, ovl = N / R, H = N - ovl;
    try {
        const auto x = slurp(dir + "/x.c64");
        const int nb = (int)(x.size() / (size_t)H);
        auto os = overlap_save::make(sizeof(gr_complex), N, ovl);
        auto blocks = run(os, x, nb, (size_t)N);
        dump(dir + "/overlap_save.out", blocks);
        auto cut = vector_cut_vxx::make(sizeof(gr_complex), N, 301, 64);
        auto sl = run(cut, blocks, nb, 64);
        dump(dir + "/vector_cut.out", sl);
        auto win = phase_shifting_windowing_vcc::make(64, R, 301, 0.6f, 0.85f, 1);
        dump(dir + "/phase_window.out", run(win, sl, nb, 64));
        bool threw = false;
        try { phase_shifting_windowing_vcc::make(64, R, 0, 0.9f, 0.5f, 1); } catch (const std::invalid_argument &) { threw = true; }
        if (!threw) throw std::runtime_error("constructor predicate did not throw");

        // the fused throughput chain as one block: two work() calls (state carries over), the second with the
        // "scheduler buffers" pinned; port c -> <dir>/pipe<c>.out
        {
            const std::vector<std::vector<float>> chans = {{301.f, 64.f, 0.6f, 0.85f}, {0.f, 256.f, 0.8f, 1.0f}, {640.f, 128.f, 0.5f, 0.9f}};
            auto pipe = fdc_pipeline_vcc::make(N, R, chans, 1, nb);
            const int n1 = nb / 2, n2 = nb - n1;
            std::vector<std::vector<gr_complex>> po(chans.size());
            for (size_t c = 0; c < chans.size(); c++) po[c].resize((size_t)nb * pipe->output_item_len((int)c));
            gr_vector_const_void_star pi{x.data()};
            gr_vector_void_star pv;
            for (auto &v : po) pv.push_back(v.data());
            if (pipe->kernel_plan().find("path ") == std::string::npos) throw std::runtime_error("fdc_pipeline_vcc: no kernel plan description");
            if (pipe->work(n1, pi, pv) != n1) throw std::runtime_error("fdc_pipeline_vcc: first work() failed");
            std::vector<gr_complex> xin(x.begin() + (size_t)n1 * H, x.end());
            bool pinned = pipe->pin_buffer(xin.data(), xin.size() * sizeof(gr_complex));
            for (auto &v : po) pinned = pinned && pipe->pin_buffer(v.data(), v.size() * sizeof(gr_complex));
            if (!pinned) throw std::runtime_error("fdc_pipeline_vcc: pin_buffer failed");
            gr_vector_const_void_star pi2{xin.data()};
            gr_vector_void_star pv2;
            for (size_t c = 0; c < chans.size(); c++) pv2.push_back(po[c].data() + (size_t)n1 * pipe->output_item_len((int)c));
            if (pipe->work(n2, pi2, pv2) != n2) throw std::runtime_error("fdc_pipeline_vcc: second work() failed");
            pipe->unpin_buffers();
            for (size_t c = 0; c < chans.size(); c++) dump(dir + "/pipe" + std::to_string(c) + ".out", po[c]);
            // a scheduler may offer more items than one device batch (max_items = 2 here): work() cuts the call into pieces
            {
                auto small = fdc_pipeline_vcc::make(N, R, chans, 1, 2);
                std::vector<std::vector<gr_complex>> so(chans.size());
                gr_vector_void_star sv;
                for (size_t c = 0; c < chans.size(); c++) { so[c].resize((size_t)nb * small->output_item_len((int)c)); sv.push_back(so[c].data()); }
                if (small->work(nb, pi, sv) != nb) throw std::runtime_error("fdc_pipeline_vcc: work() beyond max_items failed");
                for (size_t c = 0; c < chans.size(); c++)
                    if (std::memcmp(so[c].data(), po[c].data(), so[c].size() * sizeof(gr_complex)) != 0)
                        throw std::runtime_error("fdc_pipeline_vcc: chunked work() differs from the whole call");
            }
            // what make() has no argument for is per-block state: two virtual members on device 0 = one work() call cut into two
            // spans (fdc_pipeline_group); the output must not change by a bit
            {
                auto two = fdc_pipeline_vcc::make(N, R, chans, 1, nb);
                two->set_devices({0, 0});       // calls of 16 items and more are cut in two (a member's span is at least 8 items)
                if (two->devices().size() != 2 || two->max_items() != nb) throw std::runtime_error("fdc_pipeline_vcc: device state not kept");
                std::vector<std::vector<gr_complex>> so(chans.size());
                gr_vector_void_star sv;
                for (size_t c = 0; c < chans.size(); c++) { so[c].resize((size_t)nb * two->output_item_len((int)c)); sv.push_back(so[c].data()); }
                if (two->work(nb, pi, sv) != nb) throw std::runtime_error("fdc_pipeline_vcc: work() on a device group failed");
                for (size_t c = 0; c < chans.size(); c++)
                    if (std::memcmp(so[c].data(), po[c].data(), so[c].size() * sizeof(gr_complex)) != 0)
                        throw std::runtime_error("fdc_pipeline_vcc: a two-member group differs from the single handle");
                bool threw3 = false;
                try { two->set_devices({0, 77}); } catch (const std::exception &) { threw3 = true; }
                if (!threw3 || two->devices().size() != 2) throw std::runtime_error("fdc_pipeline_vcc: bad device list accepted");
            }
            bool threw2 = false;
            try { fdc_pipeline_vcc::make(N, R, {{0.f, 64.f, 0.9f, 0.5f}}, 1, 4); } catch (const std::invalid_argument &) { threw2 = true; }
            if (!threw2) throw std::runtime_error("fdc_pipeline_vcc: bad channel did not throw");
        }

        // The hier block as ONE block: fdc_pipeline_vcc with the sink blocks attached to its device-resident spectrum, the stream handed
        // over in ragged work() calls; once serial (PDUs inside the call), once pipelined (PDUs two calls later, the rest at stop()):
        // the same messages in the same order, payloads bit for bit.  <dir>/xb.c64 -> hier_pdus.txt / hier_pdus.out / hier_pipe0.out
        {
            std::ifstream probe(dir + "/xb.c64", std::ios::binary);
            if (probe.good()) {
                const auto xb = slurp(dir + "/xb.c64");
                const int nbb = (int)(xb.size() / (size_t)H);
                const std::vector<std::vector<float>> chans = {{128.f, 256.f, 0.8f, 1.0f}};
                fdc_pipeline_vcc::sink_setup ss;
                ss.activity_controlled_channels = {{0.3f, 0.04f}};
                ss.pac_thresh = 6.0f; ss.pac_maxblocks = 3; ss.pac_deactivation_delay = 0;
                ss.activity_detection_segments = {{0.55f, 0.9f}};
                ss.det_thresh = 10.0f; ss.det_maxblocks = 3; ss.minchandist = 0.01f; ss.det_deactivation_delay = 1;
                std::vector<std::vector<gr::fdc_message>> msgs(2);
                std::vector<std::vector<gr_complex>> outs(2);
                for (int form = 0; form < 2; form++) {
                    auto hb = fdc_pipeline_vcc::make(N, R, chans, 1, 5);
                    ss.pipelined = form == 1;
                    hb->attach_sinks(ss);
                    if (hb->sinks_latency() != (form ? 2 : 0)) throw std::runtime_error("hier block: unexpected PDU latency");
                    outs[(size_t)form].resize((size_t)nbb * hb->output_item_len(0));
                    const int sizes[] = {3, 5, 1, 4, 5, 2};
                    int a = 0, k = 0;
                    size_t before = 0;
                    while (a < nbb) {
                        const int n = std::min(sizes[k++ % 6], nbb - a);
                        gr_vector_const_void_star pi{xb.data() + (size_t)a * H};
                        gr_vector_void_star pv{outs[(size_t)form].data() + (size_t)a * hb->output_item_len(0)};
                        if (hb->work(n, pi, pv) != n) throw std::runtime_error("hier block: work() failed");
                        if (form == 1 && k <= 2 && hb->published().size() != before) throw std::runtime_error("hier block: PDUs before their latency");
                        before = hb->published().size();
                        a += n;
                    }
                    hb->stop();
                    msgs[(size_t)form] = hb->published();
                }
                if (msgs[0].size() < 2 || msgs[0].size() != msgs[1].size()) throw std::runtime_error("hier block: pipelined form publishes another number of PDUs");
                for (size_t i = 0; i < msgs[0].size(); i++) {
                    gr::fdc_message &u = msgs[0][i], &v = msgs[1][i];
                    if (u.str["ID"].substr(20) != v.str["ID"].substr(20) || u.num != v.num || u.real != v.real || u.flag != v.flag ||
                        u.samples.size() != v.samples.size() ||
                        std::memcmp(u.samples.data(), v.samples.data(), u.samples.size() * sizeof(gr_complex)) != 0)
                        throw std::runtime_error("hier block: PDU " + std::to_string(i) + " of the pipelined form differs from the serial form");
                }
                if (std::memcmp(outs[0].data(), outs[1].data(), outs[0].size() * sizeof(gr_complex)) != 0)
                    throw std::runtime_error("hier block: stream output of the pipelined form differs");
                std::vector<gr_complex> all;
                FILE *meta = std::fopen((dir + "/hier_pdus.txt").c_str(), "w");
                for (auto &m : msgs[0]) {
                    std::fprintf(meta, "%s %ld %ld %zu\n", m.str["ID"].c_str(), m.num["blockstart"], m.num["blockend"], m.samples.size());
                    all.insert(all.end(), m.samples.begin(), m.samples.end());
                }
                std::fclose(meta);
                dump(dir + "/hier_pdus.out", all);
                dump(dir + "/hier_pipe0.out", outs[0]);
            }
        }

        const auto spec = slurp(dir + "/spec.c64");
        const int ns = (int)(spec.size() / (size_t)N);
        gr_vector_const_void_star si{spec.data()};
        gr_vector_void_star none;
        auto pac = PowerActivationChannel::make(N, 320.0f / N, 40.0f / N, R, 6.0f, -1, 0, true, false, "", 0, 5);
        pac->set_max_items(5);          // batches of 5 spectrum items per fdc_sinks_work: channel state carries across batches
        pac->set_devices({0});
        pac->work(ns, si, none);
        std::vector<gr_complex> all;
        FILE *meta = std::fopen((dir + "/pdus.txt").c_str(), "w");
        for (auto &m : pac->published()) {
            std::fprintf(meta, "%s %ld %ld %zu\n", m.str["ID"].c_str(), m.num["blockstart"], m.num["blockend"], m.samples.size());
            all.insert(all.end(), m.samples.begin(), m.samples.end());
        }
        auto det = activity_detection_channelizer_vcm::make(N, {{0.5f, 0.9f}}, 10.0f, R, -1, true, false, "", false, 0.01f, 1, 0.2, 0);
        det->set_max_items(5);
        det->set_devices({0, 0});       // one segment over two members: the second stays idle, the PDUs are the single bank's
        det->work(ns, si, none);
        for (auto &m : det->published()) {
            std::fprintf(meta, "%s %ld %ld %zu\n", m.str["ID"].c_str(), m.num["blockstart"], m.num["blockend"], m.samples.size());
            all.insert(all.end(), m.samples.begin(), m.samples.end());
        }
        // verbose = 2: log file gr-FDC.ActDetChan.ID_2.log in the working directory; fileoutput: <dir>/<ID>.fin
        auto sd = SegmentDetection::make(2, N, R, 0.5f, 0.9f, 10.0f, 0.01f, 0.2f, -1, 1, true, true, dir, false, 2);
        sd->set_max_items(5);
        sd->work(ns, si, none);
        for (auto &m : sd->published()) {
            std::fprintf(meta, "%s %ld %ld %zu\n", m.str["ID"].c_str(), m.num["blockstart"], m.num["blockend"], m.samples.size());
            all.insert(all.end(), m.samples.begin(), m.samples.end());
        }
        std::fclose(meta);
        dump(dir + "/pdus.out", all);
    } catch (const std::exception &e) {
        std::cerr << "blocks_demo: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
