// declaration-only stand-in (see ../README.md): gnuradio/block.h
#pragma once
#include <gnuradio/basic_block.h>
namespace gr {
class block : public basic_block {
public:
    enum { WORK_CALLED_PRODUCE = -2, WORK_DONE = -1 };
    virtual ~block();
    unsigned history() const;
    virtual bool start();
    virtual bool stop();
    void set_output_multiple(int multiple);
    int output_multiple() const;
    int max_noutput_items();
    void set_max_noutput_items(int m);
    void unset_max_noutput_items();
    bool is_set_max_noutput_items();
    int min_noutput_items() const;
    void set_min_noutput_items(int m);
    long max_output_buffer(size_t i);
    void set_max_output_buffer(long max_output_buffer);
    void set_max_output_buffer(int port, long max_output_buffer);
    long min_output_buffer(size_t i);
    void set_min_output_buffer(long min_output_buffer);
    void set_min_output_buffer(int port, long min_output_buffer);
    block_detail_sptr detail() const;
protected:
    block();
    block(const std::string &name, io_signature::sptr input_signature, io_signature::sptr output_signature);
};
}  // namespace gr
