// gr::FDC block classes of the MI355X build: the reference's public block API (class names, namespaces, header paths
// FDC/<block>.h, make() signatures — include/FDC/*.h:36-50 of gereonsuch/gr-FDC) over libfdc_amd.so.
// work() stays C++ inside the GNU Radio scheduler thread and forwards to the C-ABI (include/fdc_amd.h).
#pragma once
#ifdef FDC_HAVE_GNURADIO
#include <gnuradio/sync_block.h>
#else
#include "../compat/gnuradio/sync_block.h"
#endif
#include <memory>
#include <string>
#include <vector>

#ifndef FDC_API
#define FDC_API __attribute__((visibility("default")))
#endif

// The smart pointer of the block API is the one of the GNU Radio it is built against: boost::shared_ptr up to 3.8 (what the
// reference targets, include/FDC/overlap_save.h:39 `typedef boost::shared_ptr<overlap_save> sptr`, GNU Radio >= 3.7.2 per
// its CMakeLists.txt:148), std::shared_ptr from 3.9 on.  It is taken from GNU Radio's own gr::basic_block_sptr (no version
// macro exists in its headers): whatever template that is, rebound to the block class.  Stand-alone build: std::shared_ptr.
#ifdef FDC_HAVE_GNURADIO
#include <gnuradio/basic_block.h>
#endif
namespace gr {
namespace FDC {
namespace detail {
template <class Ptr, class T> struct rebind_sptr;
template <template <class...> class SP, class U, class T> struct rebind_sptr<SP<U>, T> { typedef SP<T> type; };
}  // namespace detail
#ifdef FDC_HAVE_GNURADIO
template <class T> using block_sptr = typename detail::rebind_sptr<gr::basic_block_sptr, T>::type;
#else
template <class T> using block_sptr = std::shared_ptr<T>;
#endif
}  // namespace FDC
}  // namespace gr
#define FDC_SHARED_PTR gr::FDC::block_sptr

namespace gr {
namespace FDC {

// What the reference's make() signatures have no room for lives on the BLOCK, not in the process (two flowgraphs of one
// process may want different devices): which HIP device(s) the block runs on and how many items one device batch takes.
// A block is made on device 0 with batches of 64 items; call these after make() and before the flowgraph starts (they
// rebuild the device handle, so the stream state is that of a fresh block).  They throw what make() throws.
//   set_devices({d})        the block's handle lives on device d
//   set_devices({d0, d1…})  fdc_pipeline_vcc only: ONE work() call is cut into contiguous spans of items, one per device, run
//                           concurrently (fdc_pipeline_group, include/fdc_amd.h); the other blocks take the first entry
//   set_max_items(n)        items per device batch (what one fdc_*_work call takes); also what the scheduler is asked for per
//                           work() call unless set_scheduler_batch() says otherwise
class FDC_API amd_device_config {
public:
    virtual ~amd_device_config() {}
    virtual void set_devices(const std::vector<int> &devices) = 0;
    virtual void set_max_items(int max_items) = 0;
    virtual std::vector<int> devices() const = 0;
    virtual int max_items() const = 0;
    // Items per work() call the block asks of the scheduler (set_output_multiple + set_min_output_buffer, fdc_blocks.cc):
    // 0 or 1 = the default, the reference's item-by-item behaviour: every item of a finite stream is processed, whatever the
    // scheduler offers.  n > 1 (typically max_items()) is the OPT-IN for device-sized batches: the scheduler then hands work()
    // whole multiples of n and sizes the buffers for two of them (0.7 -> 4.1 Gsample/s at 256-KiB items) — at the price every block
    // with an output multiple pays: one batch of latency, and when a FINITE stream drains its last partial batch (up to n - 1
    // items, and any PDU of a burst in them) is never offered to work().  For flowgraphs that run without end.
    // Before the flowgraph starts (GNU Radio reads output_multiple when it allocates the buffers).
    virtual void set_scheduler_batch(int items) = 0;
    virtual int scheduler_batch() const = 0;
};

class FDC_API overlap_save : virtual public gr::sync_block, public amd_device_config {
public:
    typedef FDC_SHARED_PTR<overlap_save> sptr;
    static sptr make(int itemsize, int outputlen, int overlaplen);
};

class FDC_API vector_cut_vxx : virtual public gr::sync_block, public amd_device_config {
public:
    typedef FDC_SHARED_PTR<vector_cut_vxx> sptr;
    static sptr make(int itemsize, int veclen, int offset, int blocklen);
};

class FDC_API phase_shifting_windowing_vcc : virtual public gr::sync_block, public amd_device_config {
public:
    typedef FDC_SHARED_PTR<phase_shifting_windowing_vcc> sptr;
    static sptr make(int blocklen, int numphasestates, int shifts, float passbw, float stopbw, int windowtype);
};

class FDC_API PowerActivationChannel : virtual public gr::sync_block, public amd_device_config {
public:
    typedef FDC_SHARED_PTR<PowerActivationChannel> sptr;
    static sptr make(int blocklen, float cfreq, float bw, int relinvovl, float thresh, int maxblocks,
                     int deactivation_delay, bool msg, bool fileoutput, std::string path, int verbose, int ID);
};

class FDC_API activity_detection_channelizer_vcm : virtual public gr::sync_block, public amd_device_config {
public:
    typedef FDC_SHARED_PTR<activity_detection_channelizer_vcm> sptr;
    static sptr make(int blocklen, std::vector<std::vector<float>> segments, float thresh, int relinvovl, int maxblocks,
                     bool message, bool fileoutput, std::string path, bool threads, float minchandist,
                     int channel_deactivation_delay, double window_flank_puffer, int verbose);
};

class FDC_API SegmentDetection : virtual public gr::sync_block, public amd_device_config {
public:
    typedef FDC_SHARED_PTR<SegmentDetection> sptr;
    static sptr make(int ID, int blocklen, int relinvovl, float seg_start, float seg_stop, float thresh, float minchandist,
                     float window_flank_puffer, int maxblocks_to_emit, int channel_deactivation_delay, bool messageoutput,
                     bool fileoutput, std::string path, bool threads, int verbose);
};

// The throughput chain of the FrequencyDomainChannelizer hier block as ONE sync block (what
// python/FrequencyDomainChannelizer.py:200-231 wires from overlap_save + fft_vcc + multiply_const + per channel
// vector_cut_vxx / phase_shifting_windowing_vcc / fft_vcc(inverse) / vector_cut_vxx / vector_to_stream): input items are
// (blocklen - blocklen/relinvovl) new samples, output port c carries items of lout_c = l_c - l_c/relinvovl samples.
// Not a class of the reference: it is the block a maintainer adds so that the fused device path is reachable from a
// flowgraph.
class FDC_API fdc_pipeline_vcc : virtual public gr::sync_block, public amd_device_config {
public:
    typedef FDC_SHARED_PTR<fdc_pipeline_vcc> sptr;
    // channels: rows (f, l, passbw, stopbw) as produced by get_opt_channelparams (py:322-345)
    // max_items: items per device batch (0 = 64; set_max_items() changes it)
    static sptr make(int blocklen, int relinvovl, std::vector<std::vector<float>> channels, int windowtype, int max_items);
    // optional: pin the scheduler's buffers of this block's ports once the flowgraph has allocated them (start()), so
    // that work() DMAs in place (fdc_host_register); call unpin_buffers() before they are freed (stop()).
    virtual bool pin_buffer(void *base, size_t bytes) = 0;
    virtual void unpin_buffers() = 0;
    virtual int output_item_len(int port) const = 0;
    // which kernels the plan was given, in words (fdc_pipeline_describe; the first member of a group)
    virtual std::string kernel_plan() const = 0;

    // The hier block's sink blocks on THIS block's spectrum (python/FrequencyDomainChannelizer.py:237-278 connects one
    // PowerActivationChannel per activity-controlled channel and one SegmentDetection per segment to the normalised spectrum;
    // here the spectrum never leaves the device: fdc_pipeline_work_sinks, include/fdc_amd.h).  Frequencies in the hier block's
    // INTERNAL units ([0, 1), DC at 0.5: what its get_freq / get_bw lambdas return, :70-91); the other fields are the hier block's
    // constructor arguments of the same names.  PDUs leave on this block's "msgout" port (and / or as files under `path`) exactly as
    // the reference's blocks publish them.
    //   pipelined = false: the PDUs of a work() call's items are published inside that call (the reference's behaviour);
    //   pipelined = true : the sinks of one call run beside the input copy and forward transform of the next — what GNU Radio's
    //                      thread-per-block scheduler does for the reference's separate blocks; PDUs are published sinks_latency()
    //                      work() calls later (2; 1 with verbose != 0), the rest at stop().  One device (set_devices of one entry).
    struct sink_setup {
        std::vector<std::vector<float>> activity_controlled_channels;   // rows (cfreq, bw); ID = row index
        float pac_thresh = 6.0f;            // act_contr_threshold, dB
        int pac_maxblocks = -1;             // pow_act_maxblocks
        int pac_deactivation_delay = 0;     // pow_act_deactivation_delay
        std::vector<std::vector<float>> activity_detection_segments;    // rows (start, stop)
        float det_thresh = 10.0f;           // act_det_threshold, dB
        int det_maxblocks = -1;             // act_det_maxblocks
        float minchandist = 0.005f;
        int det_deactivation_delay = 1;     // act_det_deactivation_delay
        double window_flank_puffer = 0.2;   // minchanflankpuffer
        bool msgoutput = true, fileoutput = false;
        std::string path;
        int verbose = 0;
        bool pipelined = false;
    };
    virtual void attach_sinks(const sink_setup &s) = 0;     // before the flowgraph starts; throws what the sink blocks' make() throws
    virtual int sinks_latency() const = 0;                  // work() calls between an item and its PDUs (0 serial, 1 or 2 pipelined)
    // publishes what the pipelined sinks still hold (stop() calls it); returns the number of batches handed out, -1 on failure
    virtual int flush_sinks() = 0;
};

}  // namespace FDC
}  // namespace gr
