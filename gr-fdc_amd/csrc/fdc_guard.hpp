// Exception barrier of the C-ABI (include/fdc_amd.h: "no exceptions"): every extern "C" entry that can allocate on the
// host (std::vector, std::string, std::thread, new) runs its body through guarded(), so that nothing thrown by the C++
// runtime crosses into the C / cgo / ctypes caller — it becomes a negative fdc_status with the text in fdc_last_error().
#pragma once
#include <exception>
#include <new>
#include <system_error>
#include "../../include/fdc_amd.h"

namespace fdc {

int set_error(int code, const char *fmt, ...);   // fdc_api.hip: thread-local text of fdc_last_error()

template <class F>
inline int guarded(const char *who, F &&body) noexcept
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        return set_error(FDC_ERR_NOMEM, "%s: out of host memory", who);
    } catch (const std::system_error &e) {                         // std::thread could not start, mutex failure
        return set_error(FDC_ERR_NOMEM, "%s: %s", who, e.what());
    } catch (const std::exception &e) {
        return set_error(FDC_ERR_HIP, "%s: internal error: %s", who, e.what());
    } catch (...) {
        return set_error(FDC_ERR_HIP, "%s: internal error (unknown exception)", who);
    }
}

}  // namespace fdc
