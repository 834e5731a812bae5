// Frequency-domain channel-filter tables (product code; host side, double design, one rounding to float).
//
// Same tables as the reference block builds in its constructor:
//   phase_shifting_windowing_vcc_impl ctor -> cr_win(windowtype, l, passbw, stopbw, windows, R, 1, false)
//   (lib/phase_shifting_windowing_vcc_impl.cc:62, lib/windows.h:41-124).
// W[p][i] = float( w[i] * exp(j*2*pi*c_p/R) ), c_0 = 0, c_{p+1} = (c_p + step) mod R, with the real,
// symmetric band-pass shape w: zero on `low` edge bins, a flank of `ramp` bins, plateau 1/l (or 1).
#pragma once
#include <cmath>
#include <complex>
#include <vector>

namespace fdc {

struct WindowShape {
    int type;      // 0 rectangular, 1 Hann flank, 2 linear flank
    int n, low, ramp;
    double top;    // plateau value

    // value of bin i, i counted from the nearer band edge (the shape is symmetric)
    double edge_value(int d) const
    {
        if (type == 0) return d < low + ramp / 2 ? 0.0 : top;      // step in the middle of the flank
        if (d < low) return 0.0;
        const int r = d - low;
        if (r >= ramp) return top;
        const double u = double(r + 1) / double(ramp + 1);
        if (type == 2) return top * double(r + 1) / double(ramp + 1);
        return top * (-std::cos(u * M_PI) / 2.0 + 0.5);
    }
};

inline WindowShape window_shape(int type, int n, float passbw, float stopbw, bool normalize)
{
    // bandwidth bookkeeping of lib/windows.h:42-52, float arguments widened to double as there
    if (passbw >= 1.0) { passbw = 1.0f; stopbw = 1.0f; type = 0; }
    else if (stopbw >= 1.0) stopbw = 1.0f;
    WindowShape s;
    s.type = (type == 1 || type == 2) ? type : 0;
    s.n = n;
    s.low = int((1.0 - stopbw) * double(n)) / 2;
    const int high = int(passbw * double(n));
    s.ramp = (n - 2 * s.low - high) / 2;
    s.top = normalize ? 1.0 : 1.0 / double(n);
    return s;
}

// out: R*n complex floats, phase-major.
inline void window_table(int type, int n, float passbw, float stopbw, int R, int step, bool normalize,
                         std::complex<float> *out)
{
    const WindowShape s = window_shape(type, n, passbw, stopbw, normalize);
    std::vector<double> w(n);
    for (int i = 0; i < n; i++) w[i] = s.top;
    // the reference writes the two mirrored halves edge-inwards and lets later writes win where they meet
    // (lib/windows.h:86-89,97-105,113-123); reproduce that order so degenerate (overlapping) flanks agree
    const int nz = s.type == 0 ? s.low + s.ramp / 2 : s.low;
    for (int i = 0; i < nz; i++) { w[i] = 0.0; w[n - 1 - i] = 0.0; }
    if (s.type != 0)
        for (int i = 0; i < s.ramp; i++) {
            w[s.low + i] = s.edge_value(s.low + i);
            w[n - s.low - 1 - i] = w[s.low + i];
        }
    step %= R;
    int c = 0;
    for (int p = 0; p < R; p++) {
        const double phi = 2.0 * M_PI * double(c) / double(R);
        const double cs = std::cos(phi), sn = std::sin(phi);
        for (int i = 0; i < n; i++) out[(size_t)p * n + i] = std::complex<float>(float(w[i] * cs), float(w[i] * sn));
        c = (c + step) % R;
    }
}

}  // namespace fdc
