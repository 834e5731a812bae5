// declaration-only stand-in (see ../README.md): gnuradio/runtime_types.h
#pragma once
#include <complex>
#include <memory>
#include <vector>
#include <boost/shared_ptr.hpp>
typedef std::complex<float> gr_complex;
typedef std::vector<const void *> gr_vector_const_void_star;
typedef std::vector<void *> gr_vector_void_star;
namespace gr {
class basic_block;
class block;
class block_detail;
class buffer;
class buffer_reader;
class io_signature;
#ifdef FDC_DECL_STD_SPTR
#define FDC_DECL_SP std::shared_ptr
#else
#define FDC_DECL_SP boost::shared_ptr
#endif
typedef FDC_DECL_SP<basic_block> basic_block_sptr;
typedef FDC_DECL_SP<block> block_sptr;
typedef FDC_DECL_SP<block_detail> block_detail_sptr;
typedef FDC_DECL_SP<buffer> buffer_sptr;
typedef FDC_DECL_SP<buffer_reader> buffer_reader_sptr;
}  // namespace gr
