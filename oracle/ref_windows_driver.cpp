// TEST INFRASTRUCTURE. Thin C-ABI driver around the reference's OWN window generator.
// The reference header is compiled unmodified from where it lies (/root/reference/lib/windows.h,
// passed with -I by oracle/Makefile); nothing of it is copied into this repository.  It needs only
// the C++ standard library, so it is the one piece of the path buildable in this image
// (the *_impl.cc files need GNU Radio / VOLK / pmt headers, which are absent => unbuildable here).
#include "windows.h"

extern "C" void ref_cr_win(int wintype, int blocksize, float passbw, float stopbw, int relinvovl,
                           int step, int normalize, float *out /* [relinvovl][blocksize][2] */)
{
    std::vector<std::vector<std::complex<float> > > w;
    cr_win(wintype, blocksize, passbw, stopbw, w, relinvovl, step, normalize != 0);
    for (int p = 0; p < relinvovl; p++)
        for (int k = 0; k < blocksize; k++) {
            out[2 * ((size_t)p * blocksize + k)] = w[p][k].real();
            out[2 * ((size_t)p * blocksize + k) + 1] = w[p][k].imag();
        }
}
