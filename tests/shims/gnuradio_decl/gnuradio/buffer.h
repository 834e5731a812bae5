// declaration-only stand-in (see ../README.md): gnuradio/buffer.h
#pragma once
#include <gnuradio/runtime_types.h>
namespace gr {
class buffer {
public:
    virtual ~buffer();
    int space_available();
    int bufsize() const;
    const char *base() const;
    size_t get_sizeof_item();
};
class buffer_reader {
public:
    ~buffer_reader();
    int items_available() const;
    buffer_sptr buffer() const;
};
}  // namespace gr
