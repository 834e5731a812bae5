// Minimal stand-in for the part of GNU Radio's block interface the gr::FDC faces use, so that they build and can be
// tested on machines without GNU Radio (neither the build container nor the GPU box has it).  With GNU Radio
// installed, compile with -DFDC_HAVE_GNURADIO and its own <gnuradio/sync_block.h> is used instead.
// Everything the faces call has GNU Radio's name, arguments and meaning (gr::block: set_output_multiple,
// set_min_noutput_items, set_max_noutput_items, set_min_output_buffer, history, start/stop, WORK_DONE), so that the
// scheduler stand-in of stock_scheduler.h can size buffers and offer work() what GNU Radio's would.
#pragma once
#include <complex>
#include <map>
#include <memory>
#include <string>
#include <vector>

typedef std::complex<float> gr_complex;
typedef std::vector<const void *> gr_vector_const_void_star;
typedef std::vector<void *> gr_vector_void_star;

namespace gr {

class io_signature {
public:
    typedef std::shared_ptr<io_signature> sptr;
    static sptr make(int min_streams, int max_streams, int sizeof_stream_item)
    {
        return sptr(new io_signature(min_streams, max_streams, std::vector<int>(1, sizeof_stream_item)));
    }
    static sptr makev(int min_streams, int max_streams, const std::vector<int> &sizeof_stream_items)
    {
        return sptr(new io_signature(min_streams, max_streams, sizeof_stream_items.empty() ? std::vector<int>(1, 0) : sizeof_stream_items));
    }
    int min_streams() const { return d_min; }
    int max_streams() const { return d_max; }
    // as in GNU Radio: ports beyond the list repeat its last entry
    int sizeof_stream_item(int port) const { return d_sizes[(size_t)port < d_sizes.size() ? (size_t)port : d_sizes.size() - 1]; }
    std::vector<int> sizeof_stream_items() const { return d_sizes; }
private:
    io_signature(int mn, int mx, const std::vector<int> &s) : d_min(mn), d_max(mx), d_sizes(s) {}
    int d_min, d_max;
    std::vector<int> d_sizes;
};

// what a PDU (pmt::cons(dict, c32vector)) carries, without pmt
struct fdc_message {
    std::map<std::string, std::string> str;
    std::map<std::string, long> num;
    std::map<std::string, double> real;
    std::map<std::string, bool> flag;
    std::vector<gr_complex> samples;
};

class sync_block {
public:
    enum { WORK_CALLED_PRODUCE = -2, WORK_DONE = -1 };
    sync_block() {}
    sync_block(const std::string &name, io_signature::sptr in, io_signature::sptr out) : d_name(name), d_in(in), d_out(out) {}
    virtual ~sync_block() {}
    virtual int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) = 0;
    virtual bool start() { return true; }
    virtual bool stop() { return true; }
    const std::string &name() const { return d_name; }
    io_signature::sptr input_signature() const { return d_in; }
    io_signature::sptr output_signature() const { return d_out; }
    // what the scheduler reads when it sizes buffers and calls work() (gr::block)
    unsigned history() const { return 1; }
    int output_multiple() const { return d_output_multiple; }
    void set_output_multiple(int m) { d_output_multiple = m < 1 ? 1 : m; }
    int min_noutput_items() const { return d_min_noutput; }
    void set_min_noutput_items(int m) { d_min_noutput = m; }
    int max_noutput_items() const { return d_max_noutput; }
    void set_max_noutput_items(int m) { d_max_noutput = m; d_max_noutput_set = true; }
    void unset_max_noutput_items() { d_max_noutput_set = false; }
    bool is_set_max_noutput_items() const { return d_max_noutput_set; }
    long min_output_buffer(size_t port) const { return port < d_min_out.size() ? d_min_out[port] : -1; }
    void set_min_output_buffer(long items) { d_min_out.assign(d_min_out.empty() ? 1 : d_min_out.size(), items); d_min_out_all = items; }
    void set_min_output_buffer(int port, long items)
    {
        if ((size_t)port >= d_min_out.size()) d_min_out.resize((size_t)port + 1, d_min_out_all);
        d_min_out[(size_t)port] = items;
    }
    long max_output_buffer(size_t) const { return -1; }
    void message_port_register_out(const std::string &port) { d_port = port; }
    void message_port_pub(const std::string &, const fdc_message &m) { d_published.push_back(m); }
    std::vector<fdc_message> &published() { return d_published; }      // compat only: what went out on "msgout"
private:
    std::string d_name, d_port;
    io_signature::sptr d_in, d_out;
    std::vector<fdc_message> d_published;
    int d_output_multiple = 1, d_min_noutput = 0, d_max_noutput = 0;
    bool d_max_noutput_set = false;
    std::vector<long> d_min_out;
    long d_min_out_all = -1;
};

}  // namespace gr

namespace gnuradio {
template <class T> std::shared_ptr<T> get_initial_sptr(T *p) { return std::shared_ptr<T>(p); }
}
