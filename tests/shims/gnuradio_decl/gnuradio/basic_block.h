// declaration-only stand-in (see ../README.md): gnuradio/basic_block.h
#pragma once
#include <string>
#include <gnuradio/io_signature.h>
#include <gnuradio/runtime_types.h>
#include <pmt/pmt.h>
namespace gr {
class basic_block {
public:
    virtual ~basic_block();
    std::string name() const;
    io_signature::sptr input_signature() const;
    io_signature::sptr output_signature() const;
    void message_port_register_out(pmt::pmt_t port_id);
    void message_port_pub(pmt::pmt_t port_id, pmt::pmt_t msg);
protected:
    basic_block();
    basic_block(const std::string &name, io_signature::sptr input_signature, io_signature::sptr output_signature);
};
}  // namespace gr
