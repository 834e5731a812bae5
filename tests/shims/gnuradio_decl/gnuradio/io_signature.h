// declaration-only stand-in (see ../README.md): gnuradio/io_signature.h
#pragma once
#include <gnuradio/runtime_types.h>
namespace gr {
class io_signature {
public:
    typedef FDC_DECL_SP<io_signature> sptr;
    static sptr make(int min_streams, int max_streams, int sizeof_stream_item);
    static sptr makev(int min_streams, int max_streams, const std::vector<int> &sizeof_stream_items);
    int min_streams() const;
    int max_streams() const;
    int sizeof_stream_item(int index) const;
    std::vector<int> sizeof_stream_items() const;
};
}  // namespace gr
