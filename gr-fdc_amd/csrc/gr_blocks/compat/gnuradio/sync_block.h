// Minimal stand-in for the part of GNU Radio's block interface the gr::FDC faces use, so that they build and can be
// tested on machines without GNU Radio (neither the build container nor the GPU box has it).  With GNU Radio
// installed, compile with -DFDC_HAVE_GNURADIO and its own <gnuradio/sync_block.h> is used instead.
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
        return sptr(new io_signature{min_streams, max_streams, sizeof_stream_item, {}});
    }
    static sptr makev(int min_streams, int max_streams, const std::vector<int> &sizeof_stream_items)
    {
        sptr p(new io_signature{min_streams, max_streams, sizeof_stream_items.empty() ? 0 : sizeof_stream_items[0], {}});
        p->sizeof_stream_items = sizeof_stream_items;
        return p;
    }
    int min_streams, max_streams, sizeof_stream_item;
    std::vector<int> sizeof_stream_items;      // per port, when built with makev
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
    sync_block() {}
    sync_block(const std::string &name, io_signature::sptr in, io_signature::sptr out) : d_name(name), d_in(in), d_out(out) {}
    virtual ~sync_block() {}
    virtual int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) = 0;
    const std::string &name() const { return d_name; }
    io_signature::sptr input_signature() const { return d_in; }
    io_signature::sptr output_signature() const { return d_out; }
    void message_port_register_out(const std::string &port) { d_port = port; }
    void message_port_pub(const std::string &, const fdc_message &m) { d_published.push_back(m); }
    std::vector<fdc_message> &published() { return d_published; }      // compat only: what went out on "msgout"
private:
    std::string d_name, d_port;
    io_signature::sptr d_in, d_out;
    std::vector<fdc_message> d_published;
};

}  // namespace gr

namespace gnuradio {
template <class T> std::shared_ptr<T> get_initial_sptr(T *p) { return std::shared_ptr<T>(p); }
}
