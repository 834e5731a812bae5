// declaration-only stand-in (see ../README.md): pmt/pmt.h
#pragma once
#include <complex>
#include <string>
#include <vector>
#include <gnuradio/runtime_types.h>
namespace pmt {
class pmt_base;
typedef FDC_DECL_SP<pmt_base> pmt_t;
pmt_t intern(const std::string &s);
pmt_t from_bool(bool val);
pmt_t from_long(long x);
pmt_t from_double(double x);
pmt_t make_dict();
pmt_t dict_add(const pmt_t &dict, const pmt_t &key, const pmt_t &value);
pmt_t cons(const pmt_t &x, const pmt_t &y);
pmt_t init_c32vector(size_t k, const std::complex<float> *data);
pmt_t init_c32vector(size_t k, const std::vector<std::complex<float>> &data);
}  // namespace pmt
