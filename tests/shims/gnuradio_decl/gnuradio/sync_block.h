// declaration-only stand-in (see ../README.md): gnuradio/sync_block.h
#pragma once
#include <gnuradio/block.h>
namespace gr {
class sync_block : public block {
protected:
    sync_block();
    sync_block(const std::string &name, io_signature::sptr input_signature, io_signature::sptr output_signature);
public:
    virtual int work(int noutput_items, gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) = 0;
};
}  // namespace gr
namespace gnuradio {
template <class T> FDC_DECL_SP<T> get_initial_sptr(T *p);
}
