// declaration-only stand-in (see ../README.md): gnuradio/block_detail.h
#pragma once
#include <gnuradio/buffer.h>
#include <gnuradio/runtime_types.h>
namespace gr {
class block_detail {
public:
    ~block_detail();
    int ninputs() const;
    int noutputs() const;
    buffer_reader_sptr input(unsigned int which);
    buffer_sptr output(unsigned int which);
};
}  // namespace gr
