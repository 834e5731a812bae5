// Implementations of the gr::FDC block faces: constructors validate through the C-ABI's create() (which applies the
// reference constructors' predicates) and rethrow std::invalid_argument; work() forwards and never throws.
#include "FDC/fdc_blocks.h"
#include "../../../include/fdc_amd.h"

#include <algorithm>
#include <cstdio>
#include <iostream>
#include <stdexcept>
#ifdef FDC_HAVE_GNURADIO
#include <gnuradio/block_detail.h>
#include <gnuradio/buffer.h>
#include <pmt/pmt.h>
#endif

namespace gr {
namespace FDC {

namespace {

void check_create(int rc)
{
    if (rc == FDC_ERR_INVALID_ARGUMENT) throw std::invalid_argument(fdc_last_error());
    if (rc != FDC_OK) throw std::runtime_error(fdc_last_error());
}

// work() never throws.  A failed call cannot be retried with the same arguments by the scheduler (it would be called
// again at once, for ever): the block reports itself done (WORK_DONE = -1) after printing the reason.
int report(const char *who, int n)
{
    if (n < 0) { std::cerr << who << ": " << fdc_last_error() << std::endl; return -1; }
    return n;
}

// What the block asks of the scheduler (gr::block's own knobs, read by flat_flowgraph::allocate_buffer and the executor when the
// flowgraph starts): work() calls of whole batches and port buffers that hold two of them.  GNU Radio sizes a buffer at 64 KiB
// unless the writing block's output_multiple, its min_output_buffer or a reader's output_multiple ask for more, and hands a block
// at most half a buffer per call — with 256-KiB items that is ONE item per work() (0.7 Gsample/s through fdc_pipeline_work against
// 4.5 at 256 items).  output_multiple = batch makes (i) every upstream buffer hold >= 2 batches (the
// reader's multiple counts), (ii) this block's buffers hold >= 2 batches, (iii) every call a whole number of batches.
// The price, as for every block with an output multiple: latency of one batch, and the last partial batch of a finite stream is
// NOT PROCESSED when the flowgraph drains (up to batch - 1 items, and any PDU of a burst in that tail).  The reference's blocks process
// every item, so the default is batch = 1 for every face (round 6); a flowgraph that runs without end — a receiver — opts in with
// set_scheduler_batch(max_items()) and gets 0.7 -> 4.1 Gsample/s.
template <class Blk>
void apply_scheduler_hints(Blk *blk, int batch, int max_items, int nout)
{
    blk->set_output_multiple(batch);
    blk->set_max_noutput_items(std::max(max_items, batch) / batch * batch);     // never more than one device batch (whole multiples)
    for (int c = 0; c < nout; c++) blk->set_min_output_buffer(c, 2L * batch);   // the executor offers min(space, bufsize / 2)
}

// per-block device state (amd_device_config): the setters rebuild the handle through the block's own create()
class device_state {
protected:
    std::vector<int> d_devices{0};
    int d_max_items = 64;
    int d_batch = 0;                       // items per work() call asked of the scheduler; 0 = the default: 1, the reference's item-by-item
                                           // behaviour (every item of a finite stream is processed); device-sized batches are opt-in
    int sched_batch() const { return d_batch > 0 ? std::min(d_batch, d_max_items) : 1; }
    virtual ~device_state() {}
    virtual void rebuild() = 0;            // destroy the handle(s), create them again from d_devices / d_max_items; throws like make()
    void assign_devices(const std::vector<int> &devices)
    {
        if (devices.empty()) throw std::invalid_argument("set_devices: empty device list");
        const std::vector<int> before = d_devices;
        d_devices = devices;
        try { rebuild(); } catch (...) { d_devices = before; try { rebuild(); } catch (...) {} throw; }
    }
    void assign_max_items(int n)
    {
        if (n < 1) throw std::invalid_argument("set_max_items: must be >= 1");
        const int before = d_max_items;
        d_max_items = n;
        try { rebuild(); } catch (...) { d_max_items = before; try { rebuild(); } catch (...) {} throw; }
    }
    void assign_batch(int n)
    {
        if (n < 0) throw std::invalid_argument("set_scheduler_batch: must be >= 0 (0 = the default)");
        d_batch = n;
        rebuild();                         // re-applies the scheduler hints; the handle is rebuilt with the same arguments
    }
};
#define FDC_DEVICE_CONFIG                                                                       \
    void set_devices(const std::vector<int> &devices) override { assign_devices(devices); }      \
    void set_max_items(int n) override { assign_max_items(n); }                                   \
    void set_scheduler_batch(int n) override { assign_batch(n); }                                 \
    int scheduler_batch() const override { return sched_batch(); }                                \
    std::vector<int> devices() const override { return d_devices; }                               \
    int max_items() const override { return d_max_items; }

class overlap_save_impl : public overlap_save, device_state {
    fdc_overlap_save *d_h = nullptr;
    int d_itemsize, d_outputlen, d_overlaplen;
    void rebuild() override
    {
        fdc_overlap_save_destroy(d_h); d_h = nullptr;
        check_create(fdc_overlap_save_create(d_devices[0], d_itemsize, d_outputlen, d_overlaplen, &d_h));
        apply_scheduler_hints(this, sched_batch(), d_max_items, 1);
    }
public:
    FDC_DEVICE_CONFIG
    overlap_save_impl(int itemsize, int outputlen, int overlaplen)
        : gr::sync_block("overlap_save", gr::io_signature::make(1, 1, itemsize * (outputlen - overlaplen)),
                         gr::io_signature::make(1, 1, itemsize * outputlen)),
          d_itemsize(itemsize), d_outputlen(outputlen), d_overlaplen(overlaplen)
    {
        rebuild();
    }
    ~overlap_save_impl() override { fdc_overlap_save_destroy(d_h); }
    int work(int n, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        return report("overlap_save", fdc_overlap_save_work(d_h, in[0], n, out[0]));
    }
};

class vector_cut_vxx_impl : public vector_cut_vxx, device_state {
    fdc_vector_cut *d_h = nullptr;
    int d_itemsize, d_veclen, d_offset, d_blocklen;
    void rebuild() override
    {
        fdc_vector_cut_destroy(d_h); d_h = nullptr;
        check_create(fdc_vector_cut_create(d_devices[0], d_itemsize, d_veclen, d_offset, d_blocklen, &d_h));
        apply_scheduler_hints(this, sched_batch(), d_max_items, 1);
    }
public:
    FDC_DEVICE_CONFIG
    vector_cut_vxx_impl(int itemsize, int veclen, int offset, int blocklen)
        : gr::sync_block("vector_cut_vxx", gr::io_signature::make(1, 1, itemsize * veclen),
                         gr::io_signature::make(1, 1, itemsize * blocklen)),
          d_itemsize(itemsize), d_veclen(veclen), d_offset(offset), d_blocklen(blocklen)
    {
        rebuild();
    }
    ~vector_cut_vxx_impl() override { fdc_vector_cut_destroy(d_h); }
    int work(int n, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        return report("vector_cut_vxx", fdc_vector_cut_work(d_h, in[0], n, out[0]));
    }
};

class phase_shifting_windowing_vcc_impl : public phase_shifting_windowing_vcc, device_state {
    fdc_phase_window *d_h = nullptr;
    int d_blocklen, d_states, d_shifts, d_windowtype;
    float d_passbw, d_stopbw;
    void rebuild() override
    {
        fdc_phase_window_destroy(d_h); d_h = nullptr;
        check_create(fdc_phase_window_create(d_devices[0], d_blocklen, d_states, d_shifts, d_passbw, d_stopbw, d_windowtype, &d_h));
        apply_scheduler_hints(this, sched_batch(), d_max_items, 1);
    }
public:
    FDC_DEVICE_CONFIG
    phase_shifting_windowing_vcc_impl(int blocklen, int numphasestates, int shifts, float passbw, float stopbw, int windowtype)
        : gr::sync_block("phase_shifting_windowing_vcc", gr::io_signature::make(1, 1, sizeof(gr_complex) * blocklen),
                         gr::io_signature::make(1, 1, sizeof(gr_complex) * blocklen)),
          d_blocklen(blocklen), d_states(numphasestates), d_shifts(shifts), d_windowtype(windowtype), d_passbw(passbw), d_stopbw(stopbw)
    {
        rebuild();
    }
    ~phase_shifting_windowing_vcc_impl() override { fdc_phase_window_destroy(d_h); }
    int work(int n, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        return report("phase_shifting_windowing_vcc", fdc_phase_window_work(d_h, in[0], n, out[0]));
    }
};

// PDU records -> messages on "msgout" and raw files, as the reference's blocks publish them (a PowerActivationChannel's PDU is kind 0)
void publish_pdus(gr::sync_block *blk, const std::vector<fdc_pdu> &all, bool d_msg, bool d_file, const std::string &d_path)
{
    for (const fdc_pdu &p : all) {
        const bool pac = p.kind == 0;
        const std::string id(p.id);          // "<time>.PowActChan.<ID>.<n>" / "<time>.DETECTED.<seg>.<n>", fixed at activation
        const gr_complex *d = static_cast<const gr_complex *>(p.samples);
        if (d_msg) {
#ifdef FDC_HAVE_GNURADIO
            // the reference's PDU: pmt::cons(dict, c32vector), keys in its order (PowerActivationChannel_impl.cc:222-232,
            // activity_detection_channelizer_vcm_impl.cc:415-429)
            pmt::pmt_t dict = pmt::make_dict();
            dict = pmt::dict_add(dict, pmt::intern("ID"), pmt::intern(pac ? id + (p.finalized ? ".fin" : ".part") : id));
            dict = pmt::dict_add(dict, pmt::intern("finalized"), pmt::from_bool(p.finalized != 0));
            if (p.has_part) dict = pmt::dict_add(dict, pmt::intern("part"), pmt::from_long(p.part));
            if (pac) {
                dict = pmt::dict_add(dict, pmt::intern("rel_cfreq"), pmt::from_double(p.rel_cfreq));
                dict = pmt::dict_add(dict, pmt::intern("rel_bw"), pmt::from_double(p.rel_bw));
            } else {
                dict = pmt::dict_add(dict, pmt::intern("rel_bw"), pmt::from_double(p.rel_bw));
                dict = pmt::dict_add(dict, pmt::intern("rel_cfreq"), pmt::from_double(p.rel_cfreq));
            }
            dict = pmt::dict_add(dict, pmt::intern("blockstart"), pmt::from_long((long)p.blockstart));
            dict = pmt::dict_add(dict, pmt::intern("blockend"), pmt::from_long((long)p.blockend));
            if (!pac) {
                dict = pmt::dict_add(dict, pmt::intern("vectorstart"), pmt::from_long((long)p.vectorstart));
                dict = pmt::dict_add(dict, pmt::intern("vectorend"), pmt::from_long((long)p.vectorend));
            }
            blk->message_port_pub(pmt::intern("msgout"), pmt::cons(dict, pmt::init_c32vector((size_t)p.nsamples, d)));
#else
            gr::fdc_message m;               // the same PDU without pmt (compat build)
            m.str["ID"] = pac ? id + (p.finalized ? ".fin" : ".part") : id;
            m.flag["finalized"] = p.finalized != 0;
            if (p.has_part) m.num["part"] = p.part;
            m.real["rel_bw"] = p.rel_bw; m.real["rel_cfreq"] = p.rel_cfreq;
            m.num["blockstart"] = (long)p.blockstart; m.num["blockend"] = (long)p.blockend;
            if (!pac) { m.num["vectorstart"] = (long)p.vectorstart; m.num["vectorend"] = (long)p.vectorend; }
            m.samples.assign(d, d + p.nsamples);
            blk->message_port_pub("msgout", m);
#endif
        }
        if (d_file) {
            const std::string fn = d_path + "/" + id + (p.finalized ? std::string(".fin") : ".parted." + std::to_string(p.part));
            FILE *fh = std::fopen(fn.c_str(), "wb");
            if (!fh) std::cerr << "Cannot write to file " << fn << std::endl;
            else { std::fwrite(d, sizeof(gr_complex), (size_t)p.nsamples, fh); std::fclose(fh); }
        }
    }
}

class fdc_pipeline_vcc_impl : public fdc_pipeline_vcc, device_state {
    fdc_pipeline *d_p = nullptr;               // one device
    fdc_pipeline_group *d_g = nullptr;         // several devices: one work() call cut into spans (include/fdc_amd.h)
    int d_blocklen, d_relinvovl, d_windowtype;
    std::vector<fdc_channel> d_ch;
    size_t d_in_item = 0;
    std::vector<int> d_lout;
    std::vector<void *> d_pinned;
    // attach_sinks(): the hier block's sink blocks on this block's device-resident spectrum
    bool d_has_sinks = false;
    sink_setup d_ss;
    fdc_sinks *d_s = nullptr;
    std::vector<fdc_pac_cfg> d_pacs;
    std::vector<fdc_segment_cfg> d_segs;
    static std::vector<int> out_sizes(int relinvovl, const std::vector<std::vector<float>> &ch)
    {
        std::vector<int> v;
        for (const auto &c : ch) {
            if (c.size() != 4) throw std::invalid_argument("fdc_pipeline_vcc: channel rows are (f, l, passbw, stopbw)");
            const int l = (int)c[1];
            v.push_back((int)sizeof(gr_complex) * (l - l / (relinvovl > 0 ? relinvovl : 1)));
        }
        return v;
    }
    void rebuild() override
    {
        drop_sinks();
        fdc_pipeline_destroy(d_p); d_p = nullptr;
        fdc_pipeline_group_destroy(d_g); d_g = nullptr;
        fdc_pipeline_cfg cfg{d_devices[0], d_blocklen, d_relinvovl, d_windowtype, (int32_t)d_ch.size(), d_ch.data(), d_max_items, 0,
                             d_has_sinks ? 1 : 0 /* keep_spectrum: the sinks read it */};
        if (d_has_sinks && d_devices.size() > 1)
            throw std::invalid_argument("fdc_pipeline_vcc: the sink blocks read ONE device's spectrum; a block with sinks runs on one device");
        if (d_devices.size() > 1) {
            std::vector<int32_t> dv(d_devices.begin(), d_devices.end());
            check_create(fdc_pipeline_group_create(&cfg, dv.data(), (int)dv.size(), 0, &d_g));
        } else {
            check_create(fdc_pipeline_create(&cfg, &d_p));
        }
        fdc_pipeline *p0 = d_g ? fdc_pipeline_group_member(d_g, 0) : d_p;
        d_lout.clear();
        for (size_t i = 0; i < d_ch.size(); i++) d_lout.push_back(fdc_pipeline_channel_lout(p0, (int)i));
        // the scheduler offers whole device batches and sizes this block's and its upstream buffers for two of them
        apply_scheduler_hints(this, sched_batch(), d_max_items, (int)d_ch.size());
        if (d_has_sinks) {
            fdc_sinks_cfg c{};
            c.device_id = d_devices[0]; c.blocklen = d_blocklen; c.relinvovl = d_relinvovl;
            c.npac = (int32_t)d_pacs.size(); c.pac = d_pacs.data();
            c.pac_thresh_db = d_ss.pac_thresh; c.pac_maxblocks = d_ss.pac_maxblocks; c.pac_deactivation_delay = d_ss.pac_deactivation_delay;
            c.nseg = (int32_t)d_segs.size(); c.seg = d_segs.data();
            c.det_thresh_db = d_ss.det_thresh; c.det_maxblocks = d_ss.det_maxblocks; c.minchandist = d_ss.minchandist;
            c.det_deactivation_delay = d_ss.det_deactivation_delay; c.window_flank_puffer = d_ss.window_flank_puffer;
            c.max_blocks = d_max_items;
            c.det_variant = 1;                 // the hier block instantiates SegmentDetection (python/FrequencyDomainChannelizer.py:261-278)
            c.verbose = d_ss.verbose; c.det_id = -1;
            c.flags = d_ss.pipelined ? FDC_SINKS_LOOKAHEAD : 0;
            check_create(fdc_sinks_create(&c, &d_s));
        }
    }
    void drop_sinks()
    {
        if (d_s && d_p) while (fdc_pipeline_flush_sinks(d_p, d_s) > 0) {}      // nothing of the bank may be in flight when it goes
        fdc_sinks_destroy(d_s); d_s = nullptr;
    }
    void publish_current()
    {
        std::vector<fdc_pdu> all((size_t)fdc_sinks_pdu_count(d_s));
        fdc_sinks_pdus(d_s, all.data(), (int)all.size());
        publish_pdus(this, all, d_ss.msgoutput, d_ss.fileoutput, d_ss.path);
    }
public:
    FDC_DEVICE_CONFIG
    void attach_sinks(const sink_setup &ss) override
    {
        for (const auto &v : ss.activity_controlled_channels)
            if (v.size() != 2) throw std::invalid_argument("attach_sinks: activity-controlled channels are rows (cfreq, bw)");
        for (const auto &v : ss.activity_detection_segments)
            if (v.size() != 2) throw std::invalid_argument("Segment is incorrect. must be of size 2");
        const bool had = d_has_sinks;
        const sink_setup before = d_ss;
        const std::vector<fdc_pac_cfg> pacs0 = d_pacs;
        const std::vector<fdc_segment_cfg> segs0 = d_segs;
        d_ss = ss; d_pacs.clear(); d_segs.clear();
        for (size_t i = 0; i < ss.activity_controlled_channels.size(); i++)      // ID = index in the list (:239-250)
            d_pacs.push_back(fdc_pac_cfg{ss.activity_controlled_channels[i][0], ss.activity_controlled_channels[i][1], (int32_t)i});
        for (const auto &v : ss.activity_detection_segments) d_segs.push_back(fdc_segment_cfg{v[0], v[1]});
        d_has_sinks = !d_pacs.empty() || !d_segs.empty();
        try { rebuild(); }
        catch (...) { d_has_sinks = had; d_ss = before; d_pacs = pacs0; d_segs = segs0; try { rebuild(); } catch (...) {} throw; }
        if (d_has_sinks && ss.msgoutput) {
#ifdef FDC_HAVE_GNURADIO
            message_port_register_out(pmt::intern("msgout"));
#else
            message_port_register_out("msgout");
#endif
        }
    }
    int sinks_latency() const override { return d_s && d_p ? fdc_pipeline_sinks_latency(d_p, d_s) : 0; }
    int flush_sinks() override
    {
        int batches = 0;
        if (!d_s || !d_p) return 0;
        for (;;) {
            const int r = fdc_pipeline_flush_sinks(d_p, d_s);
            if (r < 0) { std::cerr << "fdc_pipeline_vcc: " << fdc_last_error() << std::endl; return -1; }
            if (r == 0) return batches;
            publish_current();
            batches++;
        }
    }
    fdc_pipeline_vcc_impl(int blocklen, int relinvovl, const std::vector<std::vector<float>> &channels, int windowtype, int max_items)
        : gr::sync_block("fdc_pipeline_vcc",
                         gr::io_signature::make(1, 1, (int)sizeof(gr_complex) * (blocklen - blocklen / (relinvovl > 0 ? relinvovl : 1))),
                         gr::io_signature::makev((int)channels.size(), (int)channels.size(), out_sizes(relinvovl, channels))),
          d_blocklen(blocklen), d_relinvovl(relinvovl), d_windowtype(windowtype)
    {
        d_ch.resize(channels.size());
        for (size_t i = 0; i < channels.size(); i++)
            d_ch[i] = fdc_channel{(int32_t)channels[i][0], (int32_t)channels[i][1], channels[i][2], channels[i][3]};
        if (max_items > 0) d_max_items = max_items;
        d_in_item = sizeof(gr_complex) * (size_t)(blocklen - blocklen / (relinvovl > 0 ? relinvovl : 1));
        rebuild();
    }
    ~fdc_pipeline_vcc_impl() override { unpin_buffers(); drop_sinks(); fdc_pipeline_destroy(d_p); fdc_pipeline_group_destroy(d_g); }
#ifdef FDC_HAVE_GNURADIO
    // the flowgraph has allocated this block's port buffers (flat_flowgraph::setup_connections runs before start()): pin both
    // mappings of every circular buffer once, so that work() DMAs in place.  A buffer that cannot be pinned stays pageable.
    bool start() override
    {
        gr::block_detail_sptr d = detail();
        if (d) {
            for (int i = 0; i < d->ninputs(); i++) {
                gr::buffer_sptr bf = d->input(i)->buffer();
                pin_buffer(const_cast<char *>(bf->base()), 2 * (size_t)bf->bufsize() * bf->get_sizeof_item());
            }
            for (int i = 0; i < d->noutputs(); i++) {
                gr::buffer_sptr bf = d->output(i);
                pin_buffer(const_cast<char *>(bf->base()), 2 * (size_t)bf->bufsize() * bf->get_sizeof_item());
            }
        }
        return true;
    }
    // the pipelined sinks still hold the PDUs of the last one or two work() calls: they go out before the flowgraph is torn down
    bool stop() override { flush_sinks(); unpin_buffers(); return true; }
#else
    bool stop() override { flush_sinks(); return true; }
#endif
    int work(int n, gr_vector_const_void_star &in, gr_vector_void_star &out) override
    {
        // noutput_items may exceed the handle's batch size: pieces of at most max_items, every port advanced by its item length
        std::vector<void *> o(out.size());
        for (int a = 0; a < n; a += d_max_items) {
            const int k = n - a < d_max_items ? n - a : d_max_items;
            for (size_t c = 0; c < out.size(); c++)
                o[c] = static_cast<char *>(out[c]) + (size_t)a * (size_t)d_lout[c] * sizeof(gr_complex);
            const char *src = static_cast<const char *>(in[0]) + (size_t)a * d_in_item;
            const int r = report("fdc_pipeline_vcc", d_g ? fdc_pipeline_group_work(d_g, src, k, o.data(), nullptr)
                                                     : d_s ? fdc_pipeline_work_sinks(d_p, src, k, o.data(), nullptr, d_s)
                                                           : fdc_pipeline_work(d_p, src, k, o.data(), nullptr));
            if (r != k) return a > 0 ? a : r;
            if (d_s) publish_current();          // serial bank: this piece's PDUs; pipelined: those of one or two pieces ago
        }
        return n;
    }
    bool pin_buffer(void *base, size_t bytes) override
    {
        if (fdc_host_register(base, bytes) != FDC_OK) { std::cerr << "fdc_pipeline_vcc: " << fdc_last_error() << std::endl; return false; }
        d_pinned.push_back(base);
        return true;
    }
    void unpin_buffers() override
    {
        for (void *b : d_pinned) fdc_host_unregister(b);
        d_pinned.clear();
    }
    int output_item_len(int port) const override { return port >= 0 && port < (int)d_lout.size() ? d_lout[port] : -1; }
    std::string kernel_plan() const override
    {
        char buf[512] = "";
        const fdc_pipeline *p0 = d_g ? fdc_pipeline_group_member(d_g, 0) : d_p;
        if (p0) fdc_pipeline_describe(p0, buf, (int32_t)sizeof(buf));
        return buf;
    }
};

// shared by the two sink faces: PDU records -> messages on "msgout" and raw files
class sink_base : protected device_state {
protected:
    fdc_sinks *d_s = nullptr;                  // one device
    fdc_sinks_group *d_sg = nullptr;           // several: the bank cut by frequency band over the devices (include/fdc_amd.h)
    bool d_msg = false, d_file = false;
    std::string d_path;
    // the bank's configuration, kept so that set_devices() / set_max_items() can build it again
    fdc_sinks_cfg d_cfg{};
    std::vector<fdc_pac_cfg> d_pacs;
    std::vector<fdc_segment_cfg> d_segs;
    void rebuild() override
    {
        fdc_sinks_destroy(d_s); d_s = nullptr;
        fdc_sinks_group_destroy(d_sg); d_sg = nullptr;
        d_cfg.pac = d_pacs.data(); d_cfg.npac = (int32_t)d_pacs.size();
        d_cfg.seg = d_segs.data(); d_cfg.nseg = (int32_t)d_segs.size();
        d_cfg.device_id = d_devices[0];
        d_cfg.max_blocks = d_max_items;
        if (d_devices.size() > 1) {
            std::vector<int32_t> dv(d_devices.begin(), d_devices.end());
            check_create(fdc_sinks_group_create(&d_cfg, dv.data(), (int)dv.size(), &d_sg));
        } else {
            check_create(fdc_sinks_create(&d_cfg, &d_s));
        }
        apply_hints();
    }
    virtual void apply_hints() = 0;            // the block face applies the scheduler hints (it is the gr::sync_block)
    ~sink_base() override { fdc_sinks_destroy(d_s); fdc_sinks_group_destroy(d_sg); }
    int sink_work(const void *items, int n) { return d_sg ? fdc_sinks_group_work(d_sg, items, n) : fdc_sinks_work(d_s, items, n); }
    void publish(gr::sync_block *blk, bool)
    {
        std::vector<fdc_pdu> all((size_t)(d_sg ? fdc_sinks_group_pdu_count(d_sg) : fdc_sinks_pdu_count(d_s)));
        if (d_sg) fdc_sinks_group_pdus(d_sg, all.data(), (int)all.size()); else fdc_sinks_pdus(d_s, all.data(), (int)all.size());
        publish_pdus(blk, all, d_msg, d_file, d_path);
    }
    void register_port(gr::sync_block *blk)
    {
#ifdef FDC_HAVE_GNURADIO
        blk->message_port_register_out(pmt::intern("msgout"));
#else
        blk->message_port_register_out("msgout");
#endif
    }
};

class PowerActivationChannel_impl : public PowerActivationChannel, sink_base {
    void apply_hints() override { apply_scheduler_hints(this, sched_batch(), d_max_items, 0); }
public:
    FDC_DEVICE_CONFIG
    PowerActivationChannel_impl(int blocklen, float cfreq, float bw, int relinvovl, float thresh, int maxblocks,
                                int deactivation_delay, bool msg, bool fileoutput, std::string path, int verbose, int ID)
        : gr::sync_block("PowerActivationChannel", gr::io_signature::make(1, 1, sizeof(gr_complex) * blocklen),
                         gr::io_signature::make(0, 0, 0))
    {
        d_pacs.push_back(fdc_pac_cfg{cfreq, bw, ID});
        fdc_sinks_cfg &c = d_cfg;
        c.blocklen = blocklen; c.relinvovl = relinvovl; c.pac_thresh_db = thresh;
        c.pac_maxblocks = maxblocks; c.pac_deactivation_delay = deactivation_delay;
        c.verbose = verbose; c.det_id = -1;
        rebuild();
        d_msg = msg; d_file = fileoutput; d_path = path;
        if (msg) register_port(this);
    }
    int work(int n, gr_vector_const_void_star &in, gr_vector_void_star &) override
    {
        const char *p = static_cast<const char *>(in[0]);
        const size_t item = (size_t)input_signature()->sizeof_stream_item(0);
        for (int a = 0; a < n; a += d_max_items) {
            const int k = n - a < d_max_items ? n - a : d_max_items;
            if (report("PowerActivationChannel", sink_work(p + (size_t)a * item, k)) != k) return a > 0 ? a : -1;
            publish(this, true);
        }
        return n;
    }
};

class activity_detection_channelizer_vcm_impl : public activity_detection_channelizer_vcm, sink_base {
    void apply_hints() override { apply_scheduler_hints(this, sched_batch(), d_max_items, 0); }
public:
    FDC_DEVICE_CONFIG
    activity_detection_channelizer_vcm_impl(int blocklen, std::vector<std::vector<float>> segments, float thresh,
                                            int relinvovl, int maxblocks, bool message, bool fileoutput, std::string path,
                                            bool /*threads: GPU batching replaces the per-channel std::thread fan-out*/,
                                            float minchandist, int channel_deactivation_delay, double window_flank_puffer, int verbose)
        : gr::sync_block("activity_detection_channelizer_vcm", gr::io_signature::make(1, 1, sizeof(gr_complex) * blocklen),
                         gr::io_signature::make(0, 0, 0))
    {
        for (auto &v : segments) {
            if (v.size() != 2) throw std::invalid_argument("Segment is incorrect. must be of size 2");
            d_segs.push_back({v[0], v[1]});
        }
        fdc_sinks_cfg &c = d_cfg;
        c.blocklen = blocklen; c.relinvovl = relinvovl; c.det_thresh_db = thresh;
        c.det_maxblocks = maxblocks; c.minchandist = minchandist; c.det_deactivation_delay = channel_deactivation_delay;
        c.window_flank_puffer = window_flank_puffer;
        c.verbose = verbose; c.det_id = -1;
        rebuild();
        d_msg = message; d_file = fileoutput; d_path = path;
        if (message) register_port(this);
    }
    int work(int n, gr_vector_const_void_star &in, gr_vector_void_star &) override
    {
        const char *p = static_cast<const char *>(in[0]);
        const size_t item = (size_t)input_signature()->sizeof_stream_item(0);
        for (int a = 0; a < n; a += d_max_items) {
            const int k = n - a < d_max_items ? n - a : d_max_items;
            if (report("activity_detection_channelizer_vcm", sink_work(p + (size_t)a * item, k)) != k) return a > 0 ? a : -1;
            publish(this, false);
        }
        return n;
    }
};

class SegmentDetection_impl : public SegmentDetection, sink_base {
    void apply_hints() override { apply_scheduler_hints(this, sched_batch(), d_max_items, 0); }
public:
    FDC_DEVICE_CONFIG
    SegmentDetection_impl(int ID, int blocklen, int relinvovl, float seg_start, float seg_stop, float thresh,
                          float minchandist, float window_flank_puffer, int maxblocks_to_emit,
                          int channel_deactivation_delay, bool messageoutput, bool fileoutput, std::string path, bool, int verbose)
        : gr::sync_block("SegmentDetection", gr::io_signature::make(1, 1, sizeof(gr_complex) * blocklen),
                         gr::io_signature::make(0, 0, 0))
    {
        d_segs.push_back(fdc_segment_cfg{seg_start, seg_stop});
        fdc_sinks_cfg &c = d_cfg;
        c.blocklen = blocklen; c.relinvovl = relinvovl; c.det_thresh_db = thresh;
        c.det_maxblocks = maxblocks_to_emit; c.minchandist = minchandist; c.det_deactivation_delay = channel_deactivation_delay;
        c.window_flank_puffer = window_flank_puffer; c.det_variant = 1;
        c.verbose = verbose; c.det_id = ID;
        rebuild();
        d_msg = messageoutput; d_file = fileoutput; d_path = path;
        if (messageoutput) register_port(this);
    }
    int work(int n, gr_vector_const_void_star &in, gr_vector_void_star &) override
    {
        const char *p = static_cast<const char *>(in[0]);
        const size_t item = (size_t)input_signature()->sizeof_stream_item(0);
        for (int a = 0; a < n; a += d_max_items) {
            const int k = n - a < d_max_items ? n - a : d_max_items;
            if (report("SegmentDetection", sink_work(p + (size_t)a * item, k)) != k) return a > 0 ? a : -1;
            publish(this, false);
        }
        return n;
    }
};

}  // namespace

SegmentDetection::sptr SegmentDetection::make(int ID, int blocklen, int relinvovl, float seg_start, float seg_stop,
                                              float thresh, float minchandist, float window_flank_puffer,
                                              int maxblocks_to_emit, int channel_deactivation_delay, bool messageoutput,
                                              bool fileoutput, std::string path, bool threads, int verbose)
{
    return gnuradio::get_initial_sptr(new SegmentDetection_impl(ID, blocklen, relinvovl, seg_start, seg_stop, thresh,
                                                               minchandist, window_flank_puffer, maxblocks_to_emit,
                                                               channel_deactivation_delay, messageoutput, fileoutput, path,
                                                               threads, verbose));
}

fdc_pipeline_vcc::sptr fdc_pipeline_vcc::make(int blocklen, int relinvovl, std::vector<std::vector<float>> channels,
                                              int windowtype, int max_items)
{
    return gnuradio::get_initial_sptr(new fdc_pipeline_vcc_impl(blocklen, relinvovl, channels, windowtype, max_items));
}
overlap_save::sptr overlap_save::make(int itemsize, int outputlen, int overlaplen)
{
    return gnuradio::get_initial_sptr(new overlap_save_impl(itemsize, outputlen, overlaplen));
}
vector_cut_vxx::sptr vector_cut_vxx::make(int itemsize, int veclen, int offset, int blocklen)
{
    return gnuradio::get_initial_sptr(new vector_cut_vxx_impl(itemsize, veclen, offset, blocklen));
}
phase_shifting_windowing_vcc::sptr phase_shifting_windowing_vcc::make(int blocklen, int numphasestates, int shifts,
                                                                      float passbw, float stopbw, int windowtype)
{
    return gnuradio::get_initial_sptr(new phase_shifting_windowing_vcc_impl(blocklen, numphasestates, shifts, passbw, stopbw, windowtype));
}
PowerActivationChannel::sptr PowerActivationChannel::make(int blocklen, float cfreq, float bw, int relinvovl, float thresh,
                                                          int maxblocks, int deactivation_delay, bool msg, bool fileoutput,
                                                          std::string path, int verbose, int ID)
{
    return gnuradio::get_initial_sptr(new PowerActivationChannel_impl(blocklen, cfreq, bw, relinvovl, thresh, maxblocks,
                                                                      deactivation_delay, msg, fileoutput, path, verbose, ID));
}
activity_detection_channelizer_vcm::sptr activity_detection_channelizer_vcm::make(
    int blocklen, std::vector<std::vector<float>> segments, float thresh, int relinvovl, int maxblocks, bool message,
    bool fileoutput, std::string path, bool threads, float minchandist, int channel_deactivation_delay,
    double window_flank_puffer, int verbose)
{
    return gnuradio::get_initial_sptr(new activity_detection_channelizer_vcm_impl(
        blocklen, segments, thresh, relinvovl, maxblocks, message, fileoutput, path, threads, minchandist,
        channel_deactivation_delay, window_flank_puffer, verbose));
}

}  // namespace FDC
}  // namespace gr
