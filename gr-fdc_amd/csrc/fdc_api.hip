// C-ABI of the MI355X frequency-domain channelizer (include/fdc_amd.h): handles, device memory,
// launch sequencing.  No CPU compute fallback exists: without a HIP device every create() fails.
#include "../../include/fdc_amd.h"
#include "fdc_kernels.h"
#include "fdc_window.hpp"
#include "fdc_guard.hpp"
#include "fdc_plan_cost.hpp"

#include <algorithm>
#include <array>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <system_error>
#include <thread>
#include <tuple>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) return fail(FDC_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// body of an extern "C" entry: nothing thrown inside crosses the C boundary (fdc_guard.hpp)
#define FDC_ENTRY(name) return fdc::guarded(name, [&]() -> int {
#define FDC_ENTRY_END });

bool ispow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

int select_device(int device_id)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(FDC_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    if (device_id < 0 || device_id >= n) return fail(FDC_ERR_INVALID_ARGUMENT, "device_id %d out of range [0,%d)", device_id, n);
    HIPCHK(hipSetDevice(device_id));
    // the dynamic-LDS limits are function attributes PER DEVICE: set once for every device a handle is opened on
    static std::mutex mu;
    static std::vector<char> ready;
    std::lock_guard<std::mutex> lk(mu);
    if ((int)ready.size() < n) ready.resize((size_t)n, 0);
    if (!ready[(size_t)device_id]) {
        const hipError_t e = fdc::init_kernels();
        if (e != hipSuccess) return fail(FDC_ERR_HIP, "kernel attribute setup failed: %s", hipGetErrorString(e));
        ready[(size_t)device_id] = 1;
    }
    return FDC_OK;
}

// exp(-2 pi i k / n) designed in double, rounded once
std::vector<float2> make_twiddles(int n)
{
    std::vector<float2> t(n);
    for (int k = 0; k < n; k++) {
        const double a = -2.0 * M_PI * double(k) / double(n);
        t[k] = make_float2(float(std::cos(a)), float(std::sin(a)));
    }
    return t;
}

// Host ranges the caller pinned with fdc_host_register(): work() DMAs them directly; anything else goes through
// the handle's own pinned staging buffers.
struct HostRange { uintptr_t lo, hi, dev; };   // dev: device-side address of lo
std::mutex g_reg_mu;
std::vector<HostRange> g_reg;

bool host_registered(const void *ptr, size_t bytes, void **devptr = nullptr)
{
    const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
    std::lock_guard<std::mutex> lk(g_reg_mu);
    for (const HostRange &r : g_reg)
        if (a >= r.lo && a + bytes <= r.hi) {
            if (devptr) *devptr = reinterpret_cast<void *>(r.dev + (a - r.lo));
            return true;
        }
    return false;
}

}  // namespace

namespace fdc {
// Environment variables are a debugging override only: nothing is read unless FDC_DEBUG_ENV=1 (include/fdc_amd.h)
const char *debug_env(const char *name)
{
    static const bool on = [] { const char *d = getenv("FDC_DEBUG_ENV"); return d && d[0] == '1'; }();
    return on ? getenv(name) : nullptr;
}
// shared with fdc_sinks.hip
int set_error(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
int pick_device(int device_id) { return select_device(device_id); }
}  // namespace fdc

// launch groups below this many blocks do not go to the one-block-per-CU kernels (see fdc_pipeline_process_device)
constexpr int kBlockMinBlocks = 96;

struct fdc_pipeline {
    fdc_pipeline_cfg cfg{};
    int N = 0, R = 0, ovl = 0, H = 0, C = 0;
    int chunk = 0;
    int64_t sum_lout = 0;
    std::vector<fdc::ChanDev> chans;
    std::vector<std::pair<int, std::vector<int32_t>>> groups;   // (l, channel ids)
    std::vector<size_t> group_off;
    hipStream_t stream = nullptr;
    // device memory
    float2 *d_tw = nullptr; int ntab = 0;
    float2 *d_wins = nullptr;
    float2 *d_tw256 = nullptr;   // fast path: exp(-2 pi i j/256)
    float2 *d_tw1024 = nullptr;  // uniform path with 1024 slots: exp(-2 pi i j/1024)
    float2 *d_twf = nullptr;     // fast path: [k2][n1] inter-pass twiddles of the 256x256 transform
    std::vector<char> g_aligned, g_out_aligned;   // per channel group
    // ---- the plan (classify_plan): what runs without a spectrum in memory
    // A BANK is a set of channels of ONE width l on ONE grid f = l slot + r with ONE window, every slot at most once: one launch of the
    // width's block kernel per launch group (fdc_block256.hip: l = 256, any r; fdc_block512.hip / fdc_block1024.hip: r = 0 or l/2;
    // fdc_blocknarrow.hip: l = 128 / 64, r a multiple of l/4), or — a plan that is ONE on-grid bank where no block kernel applies, and
    // launch groups shorter than block_min — the two-launch form (stage 1 + stage 2 through the scratch G).  A plan may line up banks of
    // DIFFERENT widths (round 5); what fits no bank is the remainder of a split plan.
    struct Bank {
        int L = 256, r = 0;
        float passbw = 0, stopbw = 0;
        std::vector<int> chan;
        float2 *d_cbt = nullptr;            // per-column constants of the width's kernel (offset and (-1)^n1 folded in)
        float *d_shn = nullptr;             // window shape / N (512 / 1024 at r = l/2: halves swapped)
        long long *d_slot_off = nullptr;    // slot -> output offset of the channel, -1 = unused
        float2 *d_tab = nullptr;            // narrow kernel: its LDS image
    };
    std::vector<Bank> banks;
    bool poly_ok = false;        // banks is not empty
    bool poly_block = false;     // every bank has a block kernel: one launch per bank (path 3; with a remainder: path 4)
    std::vector<std::pair<int, int>> bank_alias;             // (channel, the earlier channel with the same slice and window): computed once, copied
    // tables the banks of one width share
    float2 *d_tw512 = nullptr, *d_twq512 = nullptr;          // W_512^k, W_N^(16 n1 q) with 128 columns
    float2 *d_tw1k = nullptr, *d_twq1k = nullptr;            // W_1024^k, W_N^(16 n1 q) with 64 columns
    float2 *d_t2g = nullptr;                                 // generic two-launch form of ONE bank of another width: W_N^(t k2), tile order
    // Split plans (round 4; N = 65536): the channels that fit no bank — other widths, odd offsets, what the cost rule sends back — are the
    // REMAINDER: the banks take one block-kernel launch each, the remainder takes the spectrum path on a PARTIAL spectrum (the forward
    // kernel writes only the 64-bin groups a remainder channel reads) and channel kernels over the remainder's groups.
    bool split = false;
    std::vector<int> rem;                                        // channel ids of the remainder
    std::vector<std::pair<int, std::vector<int32_t>>> rgroups;   // the remainder by width, like `groups`
    std::vector<size_t> rgroup_off;
    std::vector<char> rg_aligned, rg_out_aligned;
    int32_t *d_rgroups = nullptr;
    unsigned long long *d_dbg = nullptr;   // FDC_BLOCK_DEBUG=1: cycle stamps of the block kernel, printed by synchronize
    int block_hints = 1;         // FDC_BLOCK_HINTS: 1 = nt output stores, 2 = nt input loads
    int block_min = kBlockMinBlocks;   // FDC_BLOCK_MIN_BLOCKS (tests: 1 = the block kernels at any size)
    float2 *d_g = nullptr;                       // uniform path (two launches): stage-1 output G, chunk*lout*N/256 samples
    int ncu = 0;                                 // compute units of the handle's device
    int reserved_cu = 0;                         // fdc_pipeline_reserve_compute_units: left out of the persistent kernels' grids
    float2 *d_twq = nullptr;                     // banks of 256-bin channels: W_N^(16 n1 q)
    // N = 65536 spectrum path: forward transform by the block kernel (fdc_block256.hip, FWD), own r = 0 tables
    bool fwd_block = false;
    // N = 4096 in one launch (fdc_fused4096.hip; fdc_pipeline_path() = 5): the spectrum of a block stays in LDS.  A workgroup takes two blocks; f4_wave[w]:
    // the rows wave w of its eight runs (2 channel + block of the pair; one width per wave), f4_cls the kernel's class nibble per wave; the device
    // schedule is made in build_device_state
    bool fused = false;
    std::vector<int> f4_wave[8];
    unsigned f4_cls = 0;
    int f4_teams = 2;            // blocks per workgroup the schedule is made for
    fdc::F4Row *d_f4rows = nullptr;
    float2 *d_ftwq = nullptr, *d_fcbt = nullptr;
    float *d_fshn = nullptr;
    long long *d_fslot = nullptr;
    float2 *d_fscr = nullptr;    // 256 KiB per compute unit: the half of T the block kernel puts aside between its two stage-2 runs
    fdc::ChanDev *d_chans = nullptr;
    int32_t *d_groups = nullptr;
    // plans that read part of the band only: 64-bin groups of the shifted spectrum some channel reads (the forward kernels that store
    // whole 64-bin runs per wave leave the other groups of the handle's internal spectrum unwritten)
    unsigned long long keep4096 = ~0ull;   // N = 4096
    unsigned *d_keep = nullptr;            // N = 65536, block forward transform: [klo][k2 / 64] words, bit = register index of the slot
    float2 *d_big = nullptr;     // channels wider than 4096 bins: scratch between the two passes of their inverse transform (big_pts points)
    fdc::ExtractTask *d_wtasks = nullptr;   // ... and their (channel, block) tasks of one piece
    size_t big_pts = 0;
    int big_l = 0;
    float2 *d_tmp = nullptr;     // two-pass intermediate, chunk*N
    float2 *d_spec = nullptr;    // spectrum, chunk*N (or max_blocks*N with keep_spectrum)
    float2 *d_ring = nullptr;    // work(): ovl + max_blocks*H
    float2 *d_specfull = nullptr; // work() with a host spectrum (debug port) and no bank to put it in: max_blocks*N, allocated at the first such call
    float *d_real = nullptr;     // work_real(): max_blocks*H real samples
    float2 *d_out = nullptr;     // work(): max_blocks*sum_lout
    int64_t blockcount = 0;      // work(): blocks consumed so far
    // work(): transfers and kernels of consecutive sub-batches overlap (H2D on s_in, kernels on stream, D2H on s_out)
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_in[2] = {nullptr, nullptr}, ev_k[2] = {nullptr, nullptr}, ev_out[2] = {nullptr, nullptr};
    float2 *pin_out[2] = {nullptr, nullptr};                                     // staging for unregistered output buffers
    fdc::ScatterEnt *pin_tab = nullptr, *d_tab = nullptr;                        // registered outputs: scatter table
    int sub = 0;                 // blocks per sub-batch
    // fdc_pipeline_work_sinks on a look-ahead bank (the pipelined hier block): the batch of the last call sits transformed in the bank's
    // next-batch buffer and is submitted by the NEXT call, beside that call's input copy and forward transform
    int hier_filled = 0;         // its block count (0 = none)
    fdc_sinks *hier_bank = nullptr;
    hipEvent_t ev_hier = nullptr;   // on the bank's fill stream behind the last call's transform and history copy: the ring may be overwritten
    bool hier_ring_busy = false;
    bool hier_broken = false;    // a pipelined call failed after it had advanced the stream state: the pair of handles cannot go on (see work_sinks_pipelined)
    bool reserve_user = false;   // fdc_pipeline_reserve_compute_units was called: the pipelined entry leaves the reservation alone
    // fdc_pipeline_process_device_power: where the power of the spectrum's 16-bin groups goes (the sinks' cells are summed from it): group sums of
    // the block whose spectrum starts at gpow_spec + k N go to gpow_base + k N / 16.  Set for the duration of one call.
    float *gpow_base = nullptr;
    const float2 *gpow_spec = nullptr;
    bool cfg_generic = false;    // FDC_FORCE_GENERIC=1: bypass the size-specialised kernels (A/B testing)
    // timing
    bool timing = false;
    int timing_stride = 1;       // events on every stride-th launch group (fdc_pipeline_enable_timing(p, stride))
    long long timing_seq = 0;
    std::vector<hipEvent_t> events;
    size_t ev_used = 0;
    std::vector<std::array<size_t, 5>> ev_spans;   // events: start, mid, end-of-fft, end-of-channels; [4]: which form the span ran (kSpan*)
};

extern "C" {

const char *fdc_last_error(void) { return g_err.c_str(); }
const char *fdc_version(void) { return "gr-fdc_amd 0.1 (gfx950)"; }

int fdc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int fdc_selftest_exception_barrier(void)
{
    struct Case { int kind, want; };
    const Case cases[] = {{0, FDC_ERR_NOMEM}, {1, FDC_ERR_NOMEM}, {2, FDC_ERR_HIP}, {3, FDC_ERR_HIP}, {4, 7}};
    for (const Case &c : cases) {
        const int got = fdc::guarded("fdc_selftest_exception_barrier", [&]() -> int {
            switch (c.kind) {
            case 0: throw std::bad_alloc();
            case 1: throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again), "thread");
            case 2: throw std::runtime_error("runtime");
            case 3: throw 42;
            default: return 7;                                     // nothing thrown: the body's own status passes through
            }
        });
        if (got != c.want) return fail(FDC_ERR_HIP, "exception barrier: kind %d came back as %d, expected %d", c.kind, got, c.want);
        if (c.want < 0 && g_err.empty()) return fail(FDC_ERR_HIP, "exception barrier: kind %d left no text", c.kind);
    }
    return FDC_OK;
}

int fdc_selftest_devices(void)
{
    FDC_ENTRY("fdc_selftest_devices")
    const int ndev = fdc_device_count();
    if (ndev <= 0) return fail(FDC_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU fallback)");
    constexpr int N = 65536, R = 2, C = 256, NB = 96, H = N - N / R, LOUT = 128;
    std::vector<fdc_channel> ch(C);
    for (int c = 0; c < C; c++) ch[(size_t)c] = fdc_channel{256 * c, 256, 0.88f, 1.0f};
    std::vector<float> x((size_t)2 * NB * H);
    uint32_t lcg = 12345u;
    for (float &v : x) { lcg = lcg * 1664525u + 1013904223u; v = (float)((int32_t)(lcg >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
    std::vector<std::vector<float>> out((size_t)C, std::vector<float>((size_t)2 * NB * LOUT));
    std::vector<void *> outs((size_t)C);
    for (int c = 0; c < C; c++) outs[(size_t)c] = out[(size_t)c].data();
    uint64_t ref = 0;
    auto checksum = [&](double *energy) {                          // FNV-1a over every output byte
        uint64_t h = 1469598103934665603ull;
        *energy = 0.0;
        for (int c = 0; c < C; c++) {
            const unsigned char *b = reinterpret_cast<const unsigned char *>(out[(size_t)c].data());
            for (size_t i = 0; i < out[(size_t)c].size() * sizeof(float); i++) { h ^= b[i]; h *= 1099511628211ull; }
            for (float v : out[(size_t)c]) *energy += (double)v * v;
        }
        return h;
    };
    fdc_pipeline_cfg cfg{};
    cfg.blocklen = N; cfg.relinvovl = R; cfg.windowtype = FDC_WIN_HANN; cfg.nchannels = C; cfg.channels = ch.data();
    cfg.max_blocks = NB; cfg.host_sub_blocks = NB;
    cfg.min_block_launch = 1;          // the one-kernel form whatever the launch length: the group's members get NB / ndev blocks each
    for (int d = 0; d < ndev; d++) {
        cfg.device_id = d;
        fdc_pipeline *p = nullptr;
        int rc = fdc_pipeline_create(&cfg, &p);
        if (rc != FDC_OK) return fail(rc, "selftest: device %d: create failed: %s", d, std::string(g_err).c_str());
        if (fdc_pipeline_path(p) != 3) { fdc_pipeline_destroy(p); return fail(FDC_ERR_UNSUPPORTED, "selftest: device %d: not on the one-kernel path", d); }
        rc = fdc_pipeline_work(p, x.data(), NB, outs.data(), nullptr);
        fdc_pipeline_destroy(p);
        if (rc != NB) return fail(rc < 0 ? rc : FDC_ERR_HIP, "selftest: device %d: work failed: %s", d, std::string(g_err).c_str());
        double energy = 0.0;
        const uint64_t h = checksum(&energy);
        if (!(energy > 0.0)) return fail(FDC_ERR_HIP, "selftest: device %d produced no output", d);
        if (d == 0) ref = h;
        else if (h != ref) return fail(FDC_ERR_HIP, "selftest: device %d differs from device 0 (checksum %016llx vs %016llx)", d,
                                       (unsigned long long)h, (unsigned long long)ref);
    }
    // the multi-device handle over ALL visible devices (one device: two virtual members on it), the same call in two pieces so
    // that the second piece's first span takes its halo from the group's history: one work() spread over the node must give
    // device 0's bytes
    {
        std::vector<int32_t> devs;
        for (int d = 0; d < std::max(ndev, 2); d++) devs.push_back(d % ndev);
        fdc_pipeline_group *g = nullptr;
        int rc = fdc_pipeline_group_create(&cfg, devs.data(), (int)devs.size(), 4, &g);
        if (rc != FDC_OK) return fail(rc, "selftest: group over %d device(s): create failed: %s", ndev, std::string(g_err).c_str());
        for (auto &o : out) std::fill(o.begin(), o.end(), 0.0f);
        const int n1 = NB / 3, n2 = NB - n1;
        std::vector<void *> outs2((size_t)C);
        for (int c = 0; c < C; c++) outs2[(size_t)c] = out[(size_t)c].data() + (size_t)2 * n1 * LOUT;
        rc = fdc_pipeline_group_work(g, x.data(), n1, outs.data(), nullptr);
        if (rc == n1) rc = fdc_pipeline_group_work(g, x.data() + (size_t)2 * n1 * H, n2, outs2.data(), nullptr);
        fdc_pipeline_group_destroy(g);
        if (rc != n2) return fail(rc < 0 ? rc : FDC_ERR_HIP, "selftest: group over %d device(s): work failed: %s", ndev, std::string(g_err).c_str());
        double energy = 0.0;
        const uint64_t h = checksum(&energy);
        if (h != ref) return fail(FDC_ERR_HIP, "selftest: the group over %d device(s) differs from device 0 (checksum %016llx vs %016llx)", ndev,
                                  (unsigned long long)h, (unsigned long long)ref);
    }
    return ndev;
    FDC_ENTRY_END
}

int fdc_window_table(int windowtype, int blocklen, float passbw, float stopbw, int numphasestates, int step,
                     int normalize, float *w)
{
    FDC_ENTRY("fdc_window_table")
    if (blocklen < 1 || numphasestates < 1 || !w) return fail(FDC_ERR_INVALID_ARGUMENT, "bad window table arguments");
    fdc::window_table(windowtype, blocklen, passbw, stopbw, numphasestates, step, normalize != 0,
                      reinterpret_cast<std::complex<float> *>(w));
    return FDC_OK;
    FDC_ENTRY_END
}

int fdc_host_register(void *ptr, size_t bytes)
{
    FDC_ENTRY("fdc_host_register")
    if (!ptr || !bytes) return fail(FDC_ERR_INVALID_ARGUMENT, "fdc_host_register: empty range");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return fail(FDC_ERR_NO_DEVICE, "no HIP device visible");
    HIPCHK(hipHostRegister(ptr, bytes, hipHostRegisterMapped | hipHostRegisterPortable));
    void *dev = nullptr;
    hipError_t e = hipHostGetDevicePointer(&dev, ptr, 0);
    if (e != hipSuccess) { (void)hipHostUnregister(ptr); return fail(FDC_ERR_HIP, "hipHostGetDevicePointer failed: %s", hipGetErrorString(e)); }
    const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
    std::lock_guard<std::mutex> lk(g_reg_mu);
    g_reg.push_back(HostRange{a, a + bytes, reinterpret_cast<uintptr_t>(dev)});
    return FDC_OK;
    FDC_ENTRY_END
}

int fdc_host_unregister(void *ptr)
{
    FDC_ENTRY("fdc_host_unregister")
    const uintptr_t a = reinterpret_cast<uintptr_t>(ptr);
    {
        std::lock_guard<std::mutex> lk(g_reg_mu);
        auto it = std::find_if(g_reg.begin(), g_reg.end(), [a](const HostRange &r) { return r.lo == a; });
        if (it == g_reg.end()) return fail(FDC_ERR_INVALID_ARGUMENT, "fdc_host_unregister: range was not registered here");
        g_reg.erase(it);
    }
    HIPCHK(hipHostUnregister(ptr));
    return FDC_OK;
    FDC_ENTRY_END
}

void fdc_pipeline_destroy(fdc_pipeline *p)
{
    if (!p) return;
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    if (p->ev_hier) { (void)hipEventSynchronize(p->ev_hier); (void)hipEventDestroy(p->ev_hier); }   // kernels of the pipelined entry ran on the bank's stream
    (void)hipFree(p->d_dbg); (void)hipFree(p->d_g); (void)hipFree(p->d_specfull);
    for (auto e : p->events) (void)hipEventDestroy(e);
    for (auto st : {p->s_in, p->s_out}) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (int i = 0; i < 2; i++) {
        for (auto e : {p->ev_in[i], p->ev_k[i], p->ev_out[i]}) if (e) (void)hipEventDestroy(e);
        if (p->pin_out[i]) (void)hipHostFree(p->pin_out[i]);
    }
    if (p->pin_tab) (void)hipHostFree(p->pin_tab);
    (void)hipFree(p->d_tw256); (void)hipFree(p->d_tw1024); (void)hipFree(p->d_twf); (void)hipFree(p->d_twq);
    for (auto &c : p->banks) { (void)hipFree(c.d_cbt); (void)hipFree(c.d_shn); (void)hipFree(c.d_slot_off); (void)hipFree(c.d_tab); }
    (void)hipFree(p->d_ftwq); (void)hipFree(p->d_fcbt); (void)hipFree(p->d_fshn); (void)hipFree(p->d_fslot); (void)hipFree(p->d_fscr);
    (void)hipFree(p->d_tw); (void)hipFree(p->d_wins); (void)hipFree(p->d_chans); (void)hipFree(p->d_groups); (void)hipFree(p->d_rgroups); (void)hipFree(p->d_keep); (void)hipFree(p->d_f4rows);
    (void)hipFree(p->d_tw512); (void)hipFree(p->d_twq512); (void)hipFree(p->d_t2g);
    (void)hipFree(p->d_tw1k); (void)hipFree(p->d_twq1k);
    (void)hipFree(p->d_big); (void)hipFree(p->d_wtasks); (void)hipFree(p->d_tmp); (void)hipFree(p->d_spec); (void)hipFree(p->d_ring); (void)hipFree(p->d_out); (void)hipFree(p->d_real);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

}  // extern "C"

// ==================================================================================================================================
// fdc_pipeline_create in four steps (round 5; it was one 530-line function): validate -> channel records -> classify_plan (which kernels
// run the plan: banks, remainder, or the spectrum path; the cost rule and nothing else decides) -> device tables and scratch.
// ==================================================================================================================================
namespace {

#define CHK_DEV(expr)                                                                           \
    do {                                                                                        \
        hipError_t _e = (expr);                                                                 \
        if (_e != hipSuccess)                                                                   \
            return fail(_e == hipErrorOutOfMemory ? FDC_ERR_NOMEM : FDC_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

template <class T>
int upload(T *&dst, const std::vector<T> &v)
{
    CHK_DEV(hipMalloc(&dst, sizeof(T) * v.size()));
    CHK_DEV(hipMemcpy(dst, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
    return FDC_OK;
}
#define UPLOAD(dst, v) do { const int _rc = upload(dst, v); if (_rc != FDC_OK) return _rc; } while (0)

float2 unit(double turns)                     // exp(-2 pi i turns), designed in double, rounded once
{
    const double a = -2.0 * M_PI * turns;
    return make_float2(float(std::cos(a)), float(std::sin(a)));
}

// ---- step 1: the arguments (the reference constructors' predicates among them)
int validate_cfg(const fdc_pipeline_cfg *cfg)
{
    const int N = cfg->blocklen, R = cfg->relinvovl;
    if (!ispow2(N) || N < 2) return fail(FDC_ERR_INVALID_ARGUMENT, "blocklen %d must be a power of two >= 2", N);
    if (!ispow2(R) || R < 2 || R > N) return fail(FDC_ERR_INVALID_ARGUMENT, "relinvovl %d must be a power of two in [2, blocklen]", R);
    if (N > (1 << 24)) return fail(FDC_ERR_UNSUPPORTED, "blocklen %d above 2^24", N);
    if (cfg->nchannels < 0 || (cfg->nchannels > 0 && !cfg->channels)) return fail(FDC_ERR_INVALID_ARGUMENT, "bad channel list");
    if (cfg->max_blocks < 1) return fail(FDC_ERR_INVALID_ARGUMENT, "max_blocks must be >= 1");
    for (int c = 0; c < cfg->nchannels; c++) {
        const fdc_channel &ch = cfg->channels[c];
        if (!ispow2(ch.l) || ch.l > N) return fail(FDC_ERR_INVALID_ARGUMENT, "channel %d: l=%d must be a power of two <= blocklen", c, ch.l);
        if (ch.f < 0 || ch.f + ch.l > N) return fail(FDC_ERR_INVALID_ARGUMENT, "channel %d: slice [%d,%d) outside the spectrum", c, ch.f, ch.f + ch.l);
        // predicates of phase_shifting_windowing_vcc_impl ctor (lib/phase_shifting_windowing_vcc_impl.cc:46-53)
        if (ch.passbw <= 0.0f) return fail(FDC_ERR_INVALID_ARGUMENT, "channel %d: PassBw must not be <= 0", c);
        if (ch.stopbw <= 0.0f) return fail(FDC_ERR_INVALID_ARGUMENT, "channel %d: StopBw must not be <= 0", c);
        if (ch.stopbw < ch.passbw) return fail(FDC_ERR_INVALID_ARGUMENT, "channel %d: StopBw must not be < PassBw", c);
    }
    // several kernels address a call's output with 32-bit byte offsets (buffer descriptors; offsets from 0xFFFFFFF0 up mean "no store"):
    // one call produces less than 4 GiB.  A stream is cut into more calls, not bigger ones.
    int64_t per_block = 0;
    for (int c = 0; c < cfg->nchannels; c++) per_block += cfg->channels[c].l - cfg->channels[c].l / R;
    if (per_block * 8 * (int64_t)cfg->max_blocks > 0xFFFFF000ll)
        return fail(FDC_ERR_INVALID_ARGUMENT, "max_blocks %d x %lld output samples per block is more than the 4 GiB one call may produce (at most %lld blocks per call for this plan)",
                    cfg->max_blocks, (long long)per_block, (long long)(0xFFFFF000ll / (per_block * 8)));
    return FDC_OK;
}

// ---- step 2: channel records, de-duplicated window tables, the channels by width
void group_by_width(const fdc_pipeline *p, const std::vector<int> *ids, std::vector<std::pair<int, std::vector<int32_t>>> &groups,
                    std::vector<size_t> &off, std::vector<char> &al, std::vector<char> &oal, std::vector<int32_t> &flat)
{
    std::map<int, std::vector<int32_t>> bylen;
    if (ids) for (int c : *ids) bylen[p->chans[(size_t)c].l].push_back(c);
    else for (int c = 0; c < p->C; c++) bylen[p->chans[(size_t)c].l].push_back(c);
    for (auto &kv : bylen) {
        bool a = true, o = true;
        for (int c : kv.second) {
            if (p->chans[(size_t)c].f & 1) a = false;
            if ((p->chans[(size_t)c].out_off & 1) || (p->chans[(size_t)c].lout & 1)) o = false;
        }
        al.push_back(a); oal.push_back(o);
        off.push_back(flat.size());
        groups.emplace_back(kv.first, kv.second);
        flat.insert(flat.end(), kv.second.begin(), kv.second.end());
    }
}

void build_channel_records(fdc_pipeline *p, const fdc_pipeline_cfg *cfg, std::vector<std::complex<float>> &pool)
{
    std::map<std::tuple<int, float, float>, int> winmap;
    int64_t off = 0;
    for (int c = 0; c < p->C; c++) {
        const fdc_channel &ch = cfg->channels[c];
        fdc::ChanDev d{};
        d.f = ch.f; d.l = ch.l; d.lout = ch.l - ch.l / p->R;
        d.shift = ((ch.f % p->R) + p->R) % p->R;
        d.out_off = off; off += d.lout;
        auto key = std::make_tuple(ch.l, ch.passbw, ch.stopbw);
        auto it = winmap.find(key);
        if (it == winmap.end()) {
            const int o = (int)pool.size();
            pool.resize(pool.size() + (size_t)p->R * ch.l);
            fdc::window_table(cfg->windowtype, ch.l, ch.passbw, ch.stopbw, p->R, 1, false, pool.data() + o);
            it = winmap.emplace(key, o).first;
        }
        d.win_off = it->second;
        p->chans.push_back(d);
    }
    p->sum_lout = off;
}

// ---- step 3: which kernels run the plan
// the block kernel that takes a bank of l-bin channels at f = l slot + r, if there is one for this block length and overlap
bool bank_has_block_kernel(int N, int R, int L, int r, int flags)
{
    if ((flags & FDC_PIPE_NO_BLOCK) || (R != 2 && R != 4)) return false;
    switch (L) {
    case 256: return fdc::poly_block_supports(N);                                    // k_blk256: N = 16384 / 32768 / 65536, any r
    case 512: return fdc::poly_block512_supports(N, R) && (r == 0 || r == L / 2);   // k_blk512<P>: N = 16384 / 32768 / 65536; on the grid or half a channel off it
    case 1024: return fdc::poly_block1024_supports(N, R) && (r == 0 || r == L / 2); // k_blk1024<P>: the same
    case 128: case 64: return fdc::poly_block_narrow_supports(N, L, R) && r % (L / 4) == 0;   // k_blknar: quarters of a channel
    default: return false;
    }
}

// N = 4096: the whole plan as ONE launch (fdc_fused4096.hip) when every channel is 16 ... 1024 bins wide.  A workgroup takes T blocks (T = 1 where no channel
// is wider than 256 bins, else 2); its rows — (block of the workgroup, channel) — go to its 4 T waves, one width per wave: two rows of 1024 bins, four of 512,
// eight of 256 or less; their exchange areas must fit the T tiles the spectra leave behind.  Always true for plans of 256-bin and wider channels of up to 4096
// bins in total; plans of channels that overlap to more (or of more than 32 narrow channels) stay on the spectrum path.
struct F4Class { int l, cls, per_wave, pitch; };
constexpr F4Class kF4Classes[] = {{1024, 4, 2, 1056}, {512, 3, 4, 513}, {128, 5, 8, 136}, {64, 6, 8, 68}, {32, 7, 8, 34}, {16, 8, 8, 17}};   // (256: below; pitches: rows of a half-wave on different banks)
bool plan_fused4096(fdc_pipeline *p, const fdc_pipeline_cfg *cfg, int flags)
{
    for (auto &w : p->f4_wave) w.clear();
    p->f4_cls = 0;
    p->f4_teams = 2;
    if (p->N != 4096 || p->C == 0 || p->cfg_generic || (flags & (FDC_PIPE_NO_POLY | FDC_PIPE_NO_FUSED))) return false;
    long long bins = 0;
    bool wide = false;
    for (int c = 0; c < p->C; c++) {
        const int l = cfg->channels[c].l;
        if (l < 16 || l > 1024 || (l & (l - 1)) || l % p->R) return false;
        bins += l;
        wide = wide || l >= 512;
    }
    // ONE 256-bin channel: the two launches are as fast or a little faster (0.060 - 0.063 against 0.064 ms per 8192 blocks; four such channels: 0.075 / 0.064;
    // everything wider: 1.3 - 2.8 x for this form, profiles/r06/plan_choice_4096.txt) — the forward transform alone is what both cost
    if (bins < 512 && !(flags & FDC_PIPE_WIDE_UNIFORM)) return false;
    // the schedule for T blocks per workgroup (4 T waves, T tiles): rows by width, the blocks' rows of a channel side by side
    auto schedule = [&](int T) {
        for (auto &w : p->f4_wave) w.clear();
        std::map<int, std::vector<int>> by;
        for (int c = 0; c < p->C; c++) for (int k = 0; k < T; k++) by[cfg->channels[c].l].push_back(2 * c + k);
        int w = 0;
        unsigned cls = 0;
        long long pts = 272ll * (long long)by[256].size();
        for (const F4Class &k : kF4Classes) {
            const std::vector<int> &rows = by[k.l];
            pts += (long long)k.pitch * (long long)rows.size();
            for (size_t i = 0; i < rows.size(); i += (size_t)k.per_wave, w++) {
                if (w >= 4 * T) return false;
                for (size_t j = i; j < std::min(i + (size_t)k.per_wave, rows.size()); j++) p->f4_wave[w].push_back(rows[j]);
                cls |= (unsigned)k.cls << (4 * w);
            }
        }
        const int avail = 4 * T - w, n256 = (int)by[256].size();
        if (n256 > 8 * avail || pts > (long long)T * fdc::fused4096_tile_points()) return false;
        if (n256) {
            // as few waves as one set of four rows each allows (a wave's instructions cost the same for one row as for four); two sets where that is not enough
            const int nw = n256 <= 4 * avail ? (n256 + 3) / 4 : avail;
            for (int i = 0; i < n256; i++) p->f4_wave[w + i % nw].push_back(by[256][(size_t)i]);
            for (int k = 0; k < nw; k++) cls |= (p->f4_wave[w + k].size() > 4 ? 2u : 1u) << (4 * (w + k));
        }
        p->f4_cls = cls;
        p->f4_teams = T;
        return true;
    };
    // one block per workgroup (four independent workgroups on a unit) where no row is wide; else, or where that does not fit, a pair of blocks
    static const int teams_env = [] { const char *e = fdc::debug_env("FDC_F4_TEAMS"); return e ? atoi(e) : 0; }();
    if ((!wide || teams_env == 1) && teams_env != 2 && schedule(1)) return true;
    if (schedule(2)) return true;
    for (auto &w : p->f4_wave) w.clear();
    p->f4_cls = 0;
    return false;
}

void classify_plan(fdc_pipeline *p, const fdc_pipeline_cfg *cfg, int flags)
{
    const int N = p->N, R = p->R, C = p->C;
    // the spectrum path's forward transform is the block kernel at N = 16384 / 32768 / 65536 (fdc_pipeline_path() = 1; decided here so that
    // fdc_pipeline_plan_preview says what create does)
    p->fwd_block = fdc::poly_block_supports(N) && !p->cfg_generic && !(flags & FDC_PIPE_NO_BLOCK);
    p->banks.clear(); p->bank_alias.clear(); p->rem.clear();
    p->poly_ok = p->poly_block = p->split = false;
    p->fused = plan_fused4096(p, cfg, flags);
    if (p->fused) return;
    if (C == 0 || p->cfg_generic || (flags & FDC_PIPE_NO_POLY) || N > (1 << 20) || R > 16) return;
    auto same_window = [&](const fdc_pipeline::Bank &b, const fdc_channel &ch) { return b.passbw == ch.passbw && b.stopbw == ch.stopbw; };

    // (a) every channel whose width has a block kernel at its offset joins the bank of its (width, offset, window); a slice that is
    //     already in its bank (the reference's parameter derivation clamps a wrapped channel onto its neighbour's place) is computed
    //     once and copied; everything else is the remainder
    std::vector<fdc_pipeline::Bank> banks;
    std::vector<std::vector<char>> used;
    std::vector<std::pair<int, int>> alias;
    std::vector<int> rem;
    for (int c = 0; c < C; c++) {
        const fdc_channel &ch = cfg->channels[c];
        const int L = ch.l, r = ch.f % L;
        if (!bank_has_block_kernel(N, R, L, r, flags)) { rem.push_back(c); continue; }
        size_t k = 0;
        for (; k < banks.size(); k++) if (banks[k].L == L && banks[k].r == r && same_window(banks[k], ch)) break;
        if (k == banks.size()) {
            fdc_pipeline::Bank b;
            b.L = L; b.r = r; b.passbw = ch.passbw; b.stopbw = ch.stopbw;
            banks.push_back(b);
            used.emplace_back((size_t)(N / L) + 1, 0);
        }
        if (used[k][(size_t)(ch.f / L)]) {
            int first = -1;
            for (int c0 : banks[k].chan) if (cfg->channels[c0].f == ch.f) { first = c0; break; }
            alias.emplace_back(c, first);
            continue;
        }
        used[k][(size_t)(ch.f / L)] = 1;
        banks[k].chan.push_back(c);
    }

    // (b) no block kernel anywhere (another block length or overlap, FDC_PIPE_NO_BLOCK): the two-launch forms take a plan that is ONE
    //     bank on its grid, every slot at most once.  l = 256: k_p1 + k_p2 / k_p2k / k_p2g (4096 <= N <= 2^20); other widths on the generic
    //     LDS core, which measured faster than the spectrum path for l = 128 only (profiles/r04/NOTES.md section 6) — FDC_PIPE_WIDE_UNIFORM
    //     takes it for every width
    if (banks.empty()) {
        const int L = cfg->channels[0].l;
        const bool fits = L == 256 ? N >= 4096
                                   : (L >= 64 && L <= 4096 && L / R >= 1 && N / L >= 16 && N / L <= 4096 && (L == 128 || (flags & FDC_PIPE_WIDE_UNIFORM)));
        if (!fits) return;
        fdc_pipeline::Bank b;
        b.L = L; b.r = 0; b.passbw = cfg->channels[0].passbw; b.stopbw = cfg->channels[0].stopbw;
        std::vector<char> u((size_t)(N / L) + 1, 0);
        for (int c = 0; c < C; c++) {
            const fdc_channel &ch = cfg->channels[c];
            if (ch.l != L || ch.f % L || !same_window(b, ch) || u[(size_t)(ch.f / L)]) return;
            u[(size_t)(ch.f / L)] = 1;
            b.chan.push_back(c);
        }
        p->banks.push_back(b);
        p->poly_ok = true;
        return;
    }

    // (c) the cost rule (fdc_plan_cost.hpp; the numbers are measured at N = 65536, where the remainder of a split plan has its forward
    //     kernel).  Banks go back to the remainder, cheapest plan first, while that lowers the sum; then the sum must beat the whole plan on
    //     the spectrum path.  Other block lengths (banks of 256-bin channels only): no remainder, no more than kMaxBanks launches.
    // (round 5: the forward variant of the block kernel exists at N = 16384 / 32768 too; per block everything costs N / 65536 of the table's
    // numbers there, on both sides of every comparison)
    const bool may_split = p->fwd_block;
    auto band = [&](const std::vector<int> &ids) { double b = 0; for (int c : ids) b += cfg->channels[c].l; return b / double(N); };
    auto move_to_rem = [&](size_t k) {
        rem.insert(rem.end(), banks[k].chan.begin(), banks[k].chan.end());
        for (size_t i = 0; i < alias.size();) {                   // copies of a channel that is no longer computed by a bank are channels again
            if (std::find(banks[k].chan.begin(), banks[k].chan.end(), alias[i].second) != banks[k].chan.end()) {
                rem.push_back(alias[i].first);
                alias.erase(alias.begin() + (long)i);
            } else i++;
        }
        banks.erase(banks.begin() + (long)k);
    };
    if (!may_split) {
        if (!rem.empty() || (int)banks.size() > fdc::cost::kMaxBanks) return;
    } else {
        auto total = [&](const std::vector<fdc_pipeline::Bank> &bs, double remband) {
            double t = fdc::cost::spectrum_path(remband);
            for (const auto &b : bs) t += fdc::cost::bank_launch(b.L);
            return t;
        };
        const bool forced = (flags & FDC_PIPE_WIDE_UNIFORM) != 0;              // every bank keeps its block kernel, whatever the rule says (A/B, tests)
        for (;;) {
            if (banks.empty()) break;
            const bool too_many = (int)banks.size() > fdc::cost::kMaxBanks;
            if (forced && !too_many) break;
            const double now = total(banks, band(rem));
            size_t best = banks.size();
            double best_t = too_many ? 1e30 : now;
            for (size_t k = 0; k < banks.size(); k++) {
                std::vector<int> r2(rem);
                r2.insert(r2.end(), banks[k].chan.begin(), banks[k].chan.end());
                for (const auto &al : alias) if (std::find(banks[k].chan.begin(), banks[k].chan.end(), al.second) != banks[k].chan.end()) r2.push_back(al.first);
                double t = fdc::cost::spectrum_path(band(r2));
                for (size_t j = 0; j < banks.size(); j++) if (j != k) t += fdc::cost::bank_launch(banks[j].L);
                if (t < best_t) { best_t = t; best = k; }
            }
            if (best == banks.size()) break;
            move_to_rem(best);
        }
        if (banks.empty()) return;                                            // the spectrum path
        if (!forced) {
            std::vector<int> all(C);
            for (int c = 0; c < C; c++) all[(size_t)c] = c;
            if (total(banks, band(rem)) >= fdc::cost::spectrum_path(band(all))) return;
        }
    }
    // the biggest bank first (what the timing events and the description call bank 1)
    std::stable_sort(banks.begin(), banks.end(), [](const fdc_pipeline::Bank &x, const fdc_pipeline::Bank &y) { return x.chan.size() > y.chan.size(); });
    std::sort(rem.begin(), rem.end());
    p->banks = std::move(banks);
    p->bank_alias = std::move(alias);
    p->rem = std::move(rem);
    p->split = !p->rem.empty();
    p->poly_ok = p->poly_block = true;
}

// ---- step 4a: the tables of one bank (window, slot table, per-column constants of its width's kernel)
int build_bank_tables(fdc_pipeline *p, const fdc_pipeline_cfg *cfg, fdc_pipeline::Bank &bk)
{
    const int N = p->N, L = bk.L, N1 = N / L, rb = bk.r;
    std::vector<std::complex<float>> shape((size_t)L);
    fdc::window_table(cfg->windowtype, L, bk.passbw, bk.stopbw, 1, 0, true, shape.data());     // plateau 1: the chain's * l is in it
    std::vector<float> sn((size_t)L);
    for (int k2 = 0; k2 < L; k2++) sn[(size_t)k2] = float(double(shape[(size_t)k2].real()) / double(N));
    std::vector<long long> so((size_t)N1, -1);
    for (int c : bk.chan) so[(size_t)(p->chans[(size_t)c].f / L)] = p->chans[(size_t)c].out_off;
    UPLOAD(bk.d_slot_off, so);
    std::vector<float2> cb;
    if (L == 256) {
        // (-1)^n1 W_N^(n1 (b + r)): an offset tiling is the on-grid plan of the block modulated by exp(-2 pi i r n / N)
        cb.resize((size_t)N1 * 16);
        for (int n1 = 0; n1 < N1; n1++)
            for (int j = 0; j < 16; j++) {
                const float2 w = unit(double(((long long)n1 * (j + rb)) % N) / double(N));
                const float sg = (n1 & 1) ? -1.0f : 1.0f;
                cb[(size_t)n1 * 16 + j] = make_float2(sg * w.x, sg * w.y);
            }
    } else if ((L == 512 || L == 1024) && p->poly_block) {
        // (-1)^n1 W_N^(n1 (b + 256 i)) at [n1][b + 16 i], i = half (512: two) or quarter (1024: four) of k2.  Half a channel off the grid: the lane
        // of part i holds part i ^ (parts / 2) of the modulated column, whose constant W_N^((l/2) n1) joins the table, and the kernels read the window
        // with its halves swapped
        const bool half = rb == L / 2;
        const int parts = L / 256;
        if (half) {
            std::vector<float> snd(sn);
            for (int k2 = 0; k2 < L; k2++) sn[(size_t)k2] = snd[(size_t)(k2 ^ (L / 2))];
        }
        cb.resize((size_t)N1 * 16 * parts);
        for (int n1 = 0; n1 < N1; n1++)
            for (int e = 0; e < 16 * parts; e++) {
                const int i = half ? (e >> 4) ^ (parts / 2) : e >> 4;
                const float2 w = unit(double(((long long)n1 * ((e & 15) + 256 * i + (half ? L / 2 : 0))) % N) / double(N));
                const float sg = (n1 & 1) ? -1.0f : 1.0f;
                cb[(size_t)n1 * 16 * parts + e] = make_float2(sg * w.x, sg * w.y);
            }
    } else if (p->poly_block) {
        // the narrow-channel block kernel (fdc_blocknarrow.hip): its LDS image, and W_N^(S V (b + r)) at [V][b], S = 256 / l
        const bool half = rb == L / 2;
        const int S = 256 / L;
        std::vector<float2> img((size_t)fdc::poly_block_narrow_table_points(L, N));
        fdc::poly_block_narrow_tables(L, N, sn.data(), img.data(), half, half ? 0 : rb);
        UPLOAD(bk.d_tab, img);
        const int NV = N / 256;                                      // virtual columns
        cb.resize((size_t)NV * 16);
        for (int V = 0; V < NV; V++)
            for (int b = 0; b < 16; b++) cb[(size_t)V * 16 + b] = unit(double(((long long)S * V * (b + rb)) % N) / double(N));
    }
    UPLOAD(bk.d_shn, sn);
    if (!cb.empty()) UPLOAD(bk.d_cbt, cb);
    return FDC_OK;
}

// ---- step 4b: what the banks of one width share, and the generic two-launch form's tile table
int build_shared_bank_tables(fdc_pipeline *p)
{
    const int N = p->N;
    auto has = [&](int L) { for (const auto &b : p->banks) if (b.L == L) return true; return false; };
    auto twq_table = [&](int N1) {                      // W_N^(16 n1 q)
        std::vector<float2> tq((size_t)N1 * 16);
        for (int n1 = 0; n1 < N1; n1++)
            for (int q = 0; q < 16; q++) tq[(size_t)n1 * 16 + q] = unit(double((16ll * n1 * q) % N) / double(N));
        return tq;
    };
    if (has(256)) {
        UPLOAD(p->d_twq, twq_table(N / 256));
        if (N / 256 == 1024) UPLOAD(p->d_tw1024, make_twiddles(1024));
    }
    if (has(512) && p->poly_block) {
        std::vector<float2> t5(256);
        for (int k = 0; k < 256; k++) t5[(size_t)k] = unit(double(k) / 512.0);
        UPLOAD(p->d_tw512, t5);
        UPLOAD(p->d_twq512, twq_table(N / 512));
    }
    if (has(1024) && p->poly_block) {
        UPLOAD(p->d_tw1k, make_twiddles(1024));
        UPLOAD(p->d_twq1k, twq_table(N / 1024));
    }
    // ONE on-grid bank of another width: its two-launch form (the whole plan where no block kernel applies; launch groups shorter than
    // block_min otherwise) wants the tile-local factor of the inter-pass twiddle in the tile's own order: t2[k2][t] = W_N^(t k2)
    if (p->banks.size() == 1 && p->banks[0].L != 256 && p->banks[0].r == 0 && p->bank_alias.empty()) {
        const int L = p->banks[0].L, TCg = fdc::poly_stage1_generic_tile_columns(N, L);
        std::vector<float2> t2v((size_t)L * TCg);
        for (int k2 = 0; k2 < L; k2++)
            for (int t = 0; t < TCg; t++) t2v[(size_t)k2 * TCg + t] = unit(double(((long long)t * k2) % N) / double(N));
        UPLOAD(p->d_t2g, t2v);
    }
    return FDC_OK;
}

// ---- step 4c: the block kernel as a forward transform (N = 65536: the spectrum path, the remainder of a split plan, the sinks)
int build_forward_tables(fdc_pipeline *p)
{
    // twq / cbt as for a bank of 256-bin channels with r = 0, a flat "window" 1/N, and the slots of stage 2 mapped to the bins 256 c (+ k2) of
    // the shifted spectrum
    const int N = p->N, N1 = N / 256;
    std::vector<float2> tq((size_t)N1 * 16), cb((size_t)N1 * 16);
    for (int n1 = 0; n1 < N1; n1++)
        for (int j = 0; j < 16; j++) {
            tq[(size_t)n1 * 16 + j] = unit(double((16ll * n1 * j) % N) / double(N));
            const float2 w = unit(double(((long long)n1 * j) % N) / double(N));
            const float sg = (n1 & 1) ? -1.0f : 1.0f;
            cb[(size_t)n1 * 16 + j] = make_float2(sg * w.x, sg * w.y);
        }
    std::vector<float> sn(256, float(1.0 / double(N)));
    std::vector<long long> so((size_t)N1);
    for (int c = 0; c < N1; c++) so[(size_t)c] = 256ll * c;
    UPLOAD(p->d_ftwq, tq);
    UPLOAD(p->d_fcbt, cb);
    UPLOAD(p->d_fshn, sn);
    UPLOAD(p->d_fslot, so);
    return FDC_OK;
}

// ---- step 4d: plans that read part of the band: the 64-bin groups of the shifted spectrum some channel reads (a split plan's internal
// spectrum serves its remainder only); the forward kernels that store whole 64-bin runs per wave leave the other groups unwritten
int build_keep_map(fdc_pipeline *p)
{
    const int N = p->N;
    std::vector<char> g64((size_t)N / 64, 0);
    bool all = true;
    for (int c = 0; c < p->C; c++) {
        if (p->split && !std::binary_search(p->rem.begin(), p->rem.end(), c)) continue;
        const auto &ch = p->chans[(size_t)c];
        for (int b = ch.f / 64; b <= (ch.f + ch.l - 1) / 64 && b < N / 64; b++) g64[(size_t)b] = 1;
    }
    for (char v : g64) all = all && v;
    if (all) return FDC_OK;
    if (N == 4096) {
        p->keep4096 = 0;
        for (int b = 0; b < 64; b++) if (g64[(size_t)b]) p->keep4096 |= 1ull << b;
        return FDC_OK;
    }
    // the block kernel's wave klo stores, per 64-row chunk q, the bins 256 c + 64 q .. + 63 of the slots c = klo + P khi (P = N / 8192 passes); slot
    // khi = k0 + 2 k1 sits in register 16 k0 + rev16(k1) (fdc_block256.hip, soff)
    const int P = N / 8192;
    std::vector<unsigned> kw((size_t)P * 4, 0u);
    for (int klo = 0; klo < P; klo++)
        for (int q = 0; q < 4; q++)
            for (int r = 0; r < 32; r++) {
                const int k0 = r >> 4, k1 = 4 * (r & 3) + ((r & 15) >> 2), c = klo + P * (k0 + 2 * k1);
                if (g64[(size_t)(4 * c + q)]) kw[(size_t)(klo * 4 + q)] |= 1u << r;
            }
    UPLOAD(p->d_keep, kw);
    return FDC_OK;
}

// the two-launch form (stage 1 + stage 2 through the scratch G) exists for a plan that is ONE bank on its grid, no copied channels
bool two_launch_possible(const fdc_pipeline *p)
{
    return p->poly_ok && p->banks.size() == 1 && p->banks[0].r == 0 && p->bank_alias.empty();
}

// ---- step 4: everything on the device
int build_device_state(fdc_pipeline *p, const fdc_pipeline_cfg *cfg, const std::vector<std::complex<float>> &pool,
                       const std::vector<int32_t> &flat, const std::vector<int32_t> &rflat)
{
    const int N = p->N, R = p->R, flags = p->cfg.flags, chunk = p->chunk;
    CHK_DEV(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    p->ntab = N;
    UPLOAD(p->d_tw, make_twiddles(N));
    if (p->C > 0) {
        CHK_DEV(hipMalloc(&p->d_wins, sizeof(float2) * pool.size()));
        CHK_DEV(hipMemcpy(p->d_wins, pool.data(), sizeof(float2) * pool.size(), hipMemcpyHostToDevice));
        UPLOAD(p->d_chans, p->chans);
        UPLOAD(p->d_groups, flat);
        if (!rflat.empty()) UPLOAD(p->d_rgroups, rflat);
    }
    UPLOAD(p->d_tw256, make_twiddles(256));
    if (N > fdc::kMaxLdsFft) {
        // inter-pass twiddles of the two-pass transform, laid out like pass A's output: [k2][n1] = W_N^(n1*k2)
        const fdc::BigGeom bg = fdc::big_geom(N);
        std::vector<float2> tf((size_t)N);
        for (int k2 = 0; k2 < bg.N2; k2++)
            for (int n1 = 0; n1 < bg.N1; n1++) tf[(size_t)k2 * bg.N1 + n1] = unit(double((long long)n1 * k2) / double(N));
        UPLOAD(p->d_twf, tf);
    }
    for (auto &bk : p->banks) { const int rc = build_bank_tables(p, cfg, bk); if (rc != FDC_OK) return rc; }
    if (p->fused) {
        std::vector<fdc::F4Row> rows(64);
        int xch = 0;
        for (int w = 0; w < 4 * p->f4_teams; w++) {
            const unsigned cls = (p->f4_cls >> (4 * w)) & 0xfu;
            const int L = cls == 4 ? 1024 : cls == 3 ? 512 : cls >= 5 ? 16 << (8 - (int)cls) : 256, pitch = cls == 4 ? 1056 : cls == 3 ? 513 : cls >= 5 ? L + L / 16 : 272;
            for (int k = 0; k < 8; k++) {
                fdc::F4Row &r = rows[(size_t)(8 * w + k)];
                r = fdc::F4Row{0, 0, 0, 0, L - L / R, 0, 0};
                if (k >= (int)p->f4_wave[w].size()) continue;
                const int code = p->f4_wave[w][(size_t)k];
                const fdc::ChanDev &ch = p->chans[(size_t)(code >> 1)];
                r = fdc::F4Row{ch.f, ch.win_off, ch.shift, xch, ch.lout, 1 + (code & 1), (long long)ch.out_off};
                xch += pitch;
            }
        }
        UPLOAD(p->d_f4rows, rows);
    }
    { const int rc = build_shared_bank_tables(p); if (rc != FDC_OK) return rc; }
    p->fwd_block = fdc::poly_block_supports(N) && !p->cfg_generic && !(flags & FDC_PIPE_NO_BLOCK);      // N = 16384 / 32768 / 65536 (round 5: the forward variant has the pass-count template too)
    if (p->fwd_block) { const int rc = build_forward_tables(p); if (rc != FDC_OK) return rc; }
    {
        hipDeviceProp_t prop;
        CHK_DEV(hipGetDeviceProperties(&prop, cfg->device_id));
        p->ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    // per-workgroup scratch of the block kernels: the forward-transform variant's second half of T, the R = 4 channelizers' rows 64..127
    if (p->fwd_block || (p->poly_block && R == 4)) CHK_DEV(hipMalloc(&p->d_fscr, sizeof(float2) * 32768 * (size_t)p->ncu));
    if (p->C > 0 && !(flags & FDC_PIPE_FULL_SPECTRUM) && (N == 4096 || p->fwd_block)) {
        const int rc = build_keep_map(p);
        if (rc != FDC_OK) return rc;
    }
    if (p->poly_block && !p->banks.empty() && p->banks[0].L == 256) {
        if (const char *dg = fdc::debug_env("FDC_BLOCK_DEBUG")) if (dg[0] == '1') {
            CHK_DEV(hipMalloc(&p->d_dbg, sizeof(unsigned long long) * 8 * 4 * 32));
            CHK_DEV(hipMemset(p->d_dbg, 0, sizeof(unsigned long long) * 8 * 4 * 32));
        }
    }
    if (two_launch_possible(p)) {
        // G scratch of the two-launch form.  With block kernels only launch groups shorter than block_min take it
        const int L = p->banks[0].L, gblocks = p->poly_block ? std::min(chunk, p->block_min) : chunk;
        CHK_DEV(hipMalloc(&p->d_g, sizeof(float2) * (size_t)gblocks * (size_t)(L - L / R) * (size_t)(N / L)));
    }
    {
        // widest "channels x width" of a group above 4096 bins: a piece of the launch group is as many blocks as fit 32 Mi points
        size_t widest = 0;
        for (const auto &gr : p->groups) if (gr.first > 4096) { widest = std::max(widest, gr.second.size() * (size_t)gr.first); p->big_l = std::max(p->big_l, gr.first); }
        if (widest) {
            p->big_pts = std::max<size_t>(widest, std::min<size_t>((size_t)32 << 20, widest * (size_t)chunk));
            CHK_DEV(hipMalloc(&p->d_big, sizeof(float2) * p->big_pts));
            CHK_DEV(hipMalloc(&p->d_wtasks, sizeof(fdc::ExtractTask) * (p->big_pts / 8192 + (size_t)p->C + 1)));   // a piece: at most big_pts / l tasks, l >= 8192
        }
    }
    // two-pass scratch; with the block kernel only launch groups shorter than block_min take the two-pass kernels
    if (N > fdc::kMaxLdsFft) CHK_DEV(hipMalloc(&p->d_tmp, sizeof(float2) * (size_t)(p->fwd_block ? std::min(chunk, p->block_min) : chunk) * N));
    CHK_DEV(hipMalloc(&p->d_spec, sizeof(float2) * (size_t)chunk * N));
    return FDC_OK;
}

}  // namespace

extern "C" {

// fdc_pipeline_cfg.flags as create AND plan_preview read them: under FDC_DEBUG_ENV=1 the debugging variables override the fields (one place, so
// that what the preview describes is what create builds)
static int effective_flags(int flags)
{
    auto on = [](const char *n) { const char *v = fdc::debug_env(n); return v && v[0] == '1'; };
    if (on("FDC_FORCE_GENERIC")) flags |= FDC_PIPE_FORCE_GENERIC;
    if (on("FDC_NO_POLY")) flags |= FDC_PIPE_NO_POLY;
    if (on("FDC_NO_BLOCK")) flags |= FDC_PIPE_NO_BLOCK;
    if (on("FDC_NO_FUSED")) flags |= FDC_PIPE_NO_FUSED;
    if (const char *bh = fdc::debug_env("FDC_BLOCK_HINTS")) {
        flags &= ~(FDC_PIPE_PLAIN_STORES | FDC_PIPE_NT_LOADS);
        if (!(atoi(bh) & 1)) flags |= FDC_PIPE_PLAIN_STORES;
        if (atoi(bh) & 2) flags |= FDC_PIPE_NT_LOADS;
    }
    return flags;
}

int fdc_pipeline_create(const fdc_pipeline_cfg *cfg, fdc_pipeline **out)
{
    FDC_ENTRY("fdc_pipeline_create")
    if (!cfg || !out) return fail(FDC_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    int rc = validate_cfg(cfg);
    if (rc != FDC_OK) return rc;
    rc = select_device(cfg->device_id);
    if (rc != FDC_OK) return rc;

    fdc_pipeline *p = new fdc_pipeline();
    struct Owner { fdc_pipeline *p; ~Owner() { if (p) fdc_pipeline_destroy(p); } } owner{p};   // error returns and exceptions free the handle
    const int N = cfg->blocklen, R = cfg->relinvovl;
    p->cfg = *cfg; p->cfg.channels = nullptr;
    p->N = N; p->R = R; p->ovl = N / R; p->H = N - p->ovl; p->C = cfg->nchannels;
    std::vector<std::complex<float>> pool;
    build_channel_records(p, cfg, pool);

    const int flags = effective_flags(cfg->flags);
    p->cfg.flags = flags;
    p->cfg_generic = (flags & FDC_PIPE_FORCE_GENERIC) != 0;
    p->block_hints = ((flags & FDC_PIPE_PLAIN_STORES) ? 0 : 1) | ((flags & FDC_PIPE_NT_LOADS) ? 2 : 0);
    if (cfg->min_block_launch >= 1) p->block_min = cfg->min_block_launch;
    if (const char *bm = fdc::debug_env("FDC_BLOCK_MIN_BLOCKS")) if (atoi(bm) >= 1) p->block_min = atoi(bm);

    std::vector<int32_t> flat, rflat;
    group_by_width(p, nullptr, p->groups, p->group_off, p->g_aligned, p->g_out_aligned, flat);
    classify_plan(p, cfg, flags);
    if (p->split) group_by_width(p, &p->rem, p->rgroups, p->rgroup_off, p->rg_aligned, p->rg_out_aligned, rflat);

    // launch groups.  Measured on MI355X (profiles/r01_*): with one stream, short launches (few tiles per
    // persistent workgroup) cost more than cache residency of the intermediates gains, on both paths, so the
    // default takes groups as large as a 2 GiB scratch budget allows (1024 blocks at N = 65536).
    int chunk = cfg->chunk_blocks;
    if (chunk <= 0) chunk = (int)std::max<int64_t>(1, (2048ll << 20) / (2ll * N * 8));
    p->chunk = std::min(chunk, cfg->max_blocks);

    rc = build_device_state(p, cfg, pool, flat, rflat);
    if (rc != FDC_OK) return rc;
    owner.p = nullptr;
    *out = p;
    return FDC_OK;
    FDC_ENTRY_END
}

int fdc_pipeline_plan_preview(const fdc_pipeline_cfg *cfg, char *buf, int32_t n, int32_t *assignment)
{
    FDC_ENTRY("fdc_pipeline_plan_preview")
    if (!cfg) return fail(FDC_ERR_INVALID_ARGUMENT, "null argument");
    const int rc = validate_cfg(cfg);
    if (rc != FDC_OK) return rc;
    // steps 1 - 3 of fdc_pipeline_create on a handle that never touches a device (no stream, no tables): host code only
    std::unique_ptr<fdc_pipeline> p(new fdc_pipeline());
    p->cfg = *cfg; p->cfg.channels = nullptr;
    p->N = cfg->blocklen; p->R = cfg->relinvovl; p->ovl = p->N / p->R; p->H = p->N - p->ovl; p->C = cfg->nchannels;
    std::vector<std::complex<float>> pool;
    build_channel_records(p.get(), cfg, pool);
    const int flags = effective_flags(cfg->flags);
    p->cfg.flags = flags;
    p->cfg_generic = (flags & FDC_PIPE_FORCE_GENERIC) != 0;
    classify_plan(p.get(), cfg, flags);
    if (assignment) {
        for (int c = 0; c < p->C; c++) assignment[c] = -1;
        for (size_t k = 0; k < p->banks.size(); k++) for (int c : p->banks[k].chan) assignment[c] = (int32_t)k;
        for (const auto &al : p->bank_alias) assignment[al.first] = -2 - al.second;
    }
    if (buf && n > 0) fdc_pipeline_describe(p.get(), buf, n);
    return fdc_pipeline_path(p.get());
    FDC_ENTRY_END
}

int64_t fdc_pipeline_input_samples(const fdc_pipeline *p, int nblocks) { return p ? (int64_t)nblocks * p->H : 0; }
int64_t fdc_pipeline_output_samples(const fdc_pipeline *p, int nblocks) { return p ? (int64_t)nblocks * p->sum_lout : 0; }
int64_t fdc_pipeline_channel_offset(const fdc_pipeline *p, int c, int nblocks)
{
    if (!p || c < 0 || c >= p->C) return -1;
    return (int64_t)nblocks * p->chans[c].out_off;
}
int32_t fdc_pipeline_channel_lout(const fdc_pipeline *p, int c)
{
    if (!p || c < 0 || c >= p->C) return -1;
    return p->chans[c].lout;
}
void *fdc_pipeline_stream(fdc_pipeline *p) { return p ? (void *)p->stream : nullptr; }
int fdc_pipeline_reserve_compute_units(fdc_pipeline *p, int32_t n)
{
    FDC_ENTRY("fdc_pipeline_reserve_compute_units")
    if (!p) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "null handle");
    if (n < 0 || n >= p->ncu) return fdc::set_error(FDC_ERR_INVALID_ARGUMENT, "%d compute units of %d cannot be left out", (int)n, p->ncu);
    p->reserved_cu = n;
    p->reserve_user = n > 0;
    return p->ncu - n;
    FDC_ENTRY_END
}
int32_t fdc_pipeline_chunk_blocks(const fdc_pipeline *p) { return p ? p->chunk : -1; }
int32_t fdc_pipeline_path(const fdc_pipeline *p)
{
    if (!p) return -1;
    if (p->fused) return 5;
    if (p->poly_block && p->split) return 4;
    if (p->poly_block) return 3;
    if (p->poly_ok) return 2;
    if (p->fwd_block || (p->N == 65536 && !p->cfg_generic)) return 1;
    return 0;
}

int32_t fdc_pipeline_describe(const fdc_pipeline *p, char *buf, int32_t n)
{
    if (!p || !buf || n < 1) return -1;
    char t[640];
    const int path = fdc_pipeline_path(p);
    int k = std::snprintf(t, sizeof(t), "N = %d, R = %d, %d channels; path %d: ", p->N, p->R, p->C, path);
    auto add = [&](const char *fmt, auto... a) { if (k < (int)sizeof(t)) k += std::snprintf(t + k, sizeof(t) - (size_t)k, fmt, a...); };
    auto kernel = [](int L) { return L == 256 ? "k_blk256" : L == 512 ? "k_blk512" : L == 1024 ? "k_blk1024" : "k_blknar"; };
    auto where = [](int L, int r) { return r == 0 ? "on the grid" : 2 * r == L ? "half a channel off the grid" : 4 * r == L ? "a quarter of a channel off the grid" : "three quarters of a channel off the grid"; };
    if (p->fused) {
        int nw = 0;
        for (const auto &w : p->f4_wave) nw += !w.empty();
        add("k_f4096, transform + %d channel transforms in one launch (spectrum in LDS, %s per workgroup, rows on %d wave%s)", p->C,
            p->f4_teams == 1 ? "one block" : "two blocks", nw, nw == 1 ? "" : "s");
    } else if (p->poly_block) {
        bool one_width = true, all256 = true;
        for (const auto &b : p->banks) { one_width = one_width && b.L == p->banks[0].L; all256 = all256 && b.L == 256; }
        if (all256) {
            add("k_blk256, %d tiling%s (r =", (int)p->banks.size(), p->banks.size() == 1 ? "" : "s");
            for (const auto &b : p->banks) add(" %d", b.r);
            add("%s", ")");
        } else if (one_width) {
            add("%s, l = %d, bank of %d %s", kernel(p->banks[0].L), p->banks[0].L, (int)p->banks[0].chan.size(), where(p->banks[0].L, p->banks[0].r));
            for (size_t i = 1; i < p->banks.size(); i++) add(" + bank of %d %s", (int)p->banks[i].chan.size(), where(p->banks[i].L, p->banks[i].r));
        } else {
            for (size_t i = 0; i < p->banks.size(); i++) {
                const auto &b = p->banks[i];
                add("%s%s bank of %d, l = %d", i ? " + " : "", kernel(b.L), (int)b.chan.size(), b.L);
                if (b.L == 256) add(" (r = %d)", b.r); else add(" %s", where(b.L, b.r));
            }
        }
        if (p->banks.size() > 1 && !all256) add(" (%s launches)", p->banks.size() == 2 ? "two" : p->banks.size() == 3 ? "three" : "four");
        if (!p->bank_alias.empty()) add(" + %d copies of channels with the same slice", (int)p->bank_alias.size());
        if (p->split) add(" + %d other channels on a partial spectrum", (int)p->rem.size());
    } else if (p->poly_ok) {
        add("two launches (stage 1 + stage 2), l = %d", p->banks[0].L);
    } else {
        add("%s", "forward transform to a spectrum in memory + channel kernels");
    }
    std::snprintf(buf, (size_t)n, "%s", t);
    return k;
}

int fdc_pipeline_synchronize(fdc_pipeline *p)
{
    FDC_ENTRY("fdc_pipeline_synchronize")
    if (!p) return fail(FDC_ERR_INVALID_ARGUMENT, "null handle");
    HIPCHK(hipSetDevice(p->cfg.device_id));
    HIPCHK(hipStreamSynchronize(p->stream));
    if (p->d_dbg) {                                   // diagnostics: stage timeline of workgroup 0 of the last launch, cycles from block start
        std::vector<unsigned long long> st(8 * 4 * 32);
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(st.data(), p->d_dbg, sizeof(unsigned long long) * st.size(), hipMemcpyDeviceToHost));
        for (int k = 0; k < 4; k++)
            for (int w = 0; w < 8; w++) {
                const unsigned long long *q = st.data() + (w * 4 + k) * 32;
                if (!q[0]) continue;
                std::fprintf(stderr, "[fdc block] round %d wave %d t0=%llu :", k, w, q[0] - st[0]);
                for (int i = 1; i < 31; i++) std::fprintf(stderr, " %lld", (long long)(q[i] - q[0]));
                std::fprintf(stderr, "\n");
            }
    }
    return FDC_OK;
    FDC_ENTRY_END
}

int fdc_pipeline_enable_timing(fdc_pipeline *p, int enable)
{
    FDC_ENTRY("fdc_pipeline_enable_timing")
    if (!p) return fail(FDC_ERR_INVALID_ARGUMENT, "null handle");
    p->timing = enable != 0;
    p->timing_stride = enable > 1 ? enable : 1;
    p->timing_seq = 0;
    p->ev_used = 0; p->ev_spans.clear();
    return FDC_OK;
    FDC_ENTRY_END
}

static int get_event(fdc_pipeline *p, size_t *idx)
{
    if (p->ev_used == p->events.size()) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        p->events.push_back(e);
    }
    *idx = p->ev_used++;
    return FDC_OK;
}

// Channels wider than 4096 bins: all channels of one width and all blocks of the launch group as ONE batch of the task-addressed two-pass
// inverse transform (fdc_kernels.hip: pass A reads slice * window straight from the spectrum, the ifftshift is its input rotation; pass B
// writes the kept samples, times l, into the channel streams) — in pieces of up to 32 Mi points of scratch between the passes.
// gids: the group's channel ids in the device list p->d_groups.
static int channels_wide(fdc_pipeline *p, const float2 *spec, float2 *d_out, const int32_t *d_gids, int ngroup, int l, int nb,
                         int mbase, int nb_call, int64_t first_block, hipStream_t s)
{
    const int per = (int)std::max<long long>(1, std::min<long long>(nb, (long long)p->big_pts / ((long long)ngroup * l)));   // blocks per piece
    const int lout = l - l / p->R;
    for (int m0 = 0; m0 < nb; m0 += per) {
        const int n = std::min(per, nb - m0);
        HIPCHK(fdc::launch_wide_tasks(p->d_wtasks, p->d_chans, d_gids, ngroup, p->R, n, mbase + m0, nb_call, first_block, s));
        HIPCHK(fdc::launch_extract_wide(spec + (size_t)m0 * p->N, p->N, p->d_wtasks, n * ngroup, l, l - lout, p->d_wins, p->d_big, d_out, p->d_tw, p->ntab, s,
                                        (float)l));
    }
    return FDC_OK;
}

// The channel kernels of one launch group over the plan's channels by width (rem: over the remainder of a split plan only)
static int run_channel_groups(fdc_pipeline *p, bool rem, const float2 *spec, float2 *d_out, int nb, int m0, int nblocks, int64_t first_block,
                              hipStream_t s)
{
    const auto &groups = rem ? p->rgroups : p->groups;
    const auto &off = rem ? p->rgroup_off : p->group_off;
    const auto &al = rem ? p->rg_aligned : p->g_aligned;
    const auto &oal = rem ? p->rg_out_aligned : p->g_out_aligned;
    const int32_t *ids = rem ? p->d_rgroups : p->d_groups;
    for (size_t g = 0; g < groups.size(); g++) {
        const int l = groups[g].first, ng = (int)groups[g].second.size();
        if (l > 4096) {
            const int rcw = channels_wide(p, spec, d_out, ids + off[g], ng, l, nb, m0, nblocks, first_block, s);
            if (rcw != FDC_OK) return rcw;
        } else if (l == 256 && ((256 / p->R) & 1) == 0 && !p->cfg_generic)
            HIPCHK(fdc::launch_channels256(spec, d_out, p->d_chans, ids + off[g], ng, al[g] != 0, oal[g] != 0, p->N, p->R, nb, m0, nblocks, first_block,
                                           p->d_wins, p->d_tw256, s));
        else if ((l == 512 || l == 1024) && l <= p->N && !p->cfg_generic)
            HIPCHK(fdc::launch_channels_wide(spec, d_out, p->d_chans, ids + off[g], ng, l, p->N, p->R, nb, m0, nblocks, first_block, p->d_wins, p->d_tw,
                                             p->ntab, s));
        else
            HIPCHK(fdc::launch_channels(spec, d_out, p->d_chans, ids + off[g], ng, l, p->N, p->R, nb, m0, nblocks, first_block, p->d_wins, p->d_tw,
                                        p->ntab, s));
    }
    return FDC_OK;
}

// The remainder of a split plan for one launch group: forward transform into the handle's internal (partial) spectrum, channel kernels
// over the remainder's groups.  ev2 / ev3 (timing): recorded behind the forward transform and behind the channel kernels.
static int run_remainder(fdc_pipeline *p, const float2 *ring, int m0, int nb, int nblocks, int64_t first_block, float2 *d_out, bool few,
                         hipStream_t s, hipEvent_t ev2, hipEvent_t ev3)
{
    if (p->fwd_block && !few)
        HIPCHK(fdc::launch_block_fft(p->N, ring + (size_t)m0 * p->H, (size_t)p->H, p->d_spec, nb, p->d_tw256, p->d_ftwq, p->d_fcbt, p->d_fshn,
                                          p->d_fslot, p->d_fscr, p->ncu - p->reserved_cu, p->block_hints, s, nullptr, p->d_keep));
    else if (p->N == 65536)
        HIPCHK(fdc::launch_fft65536(ring + (size_t)m0 * p->H, (size_t)p->H, p->d_spec, p->d_tmp, nb, p->N / 2, 1.0f / (float)p->N, p->d_tw256,
                                    p->d_twf, s, nullptr));
    else
        HIPCHK(fdc::launch_fft(ring + (size_t)m0 * p->H, (size_t)p->H, p->d_spec, p->d_tmp, p->N, nb, false, 0, p->N / 2, 1.0f / (float)p->N, p->d_tw, p->ntab,
                               s, nullptr, p->d_twf, p->cfg_generic));
    if (ev2) HIPCHK(hipEventRecord(ev2, s));
    const int rc = run_channel_groups(p, true, p->d_spec, d_out, nb, m0, nblocks, first_block, s);
    if (rc != FDC_OK) return rc;
    if (ev3) HIPCHK(hipEventRecord(ev3, s));
    return FDC_OK;
}

// which form a timed launch group ran (ev_spans[.][4]): how the three intervals between its four events map to ms[0..2]
enum { kSpanBanks = 0 /* banks | remainder forward | remainder channels */, kSpanTwoLaunch = 1 /* stage 1 | - | stage 2 (+ remainder) */,
       kSpanSpectrumLds = 2 /* forward transform = a + b | channels */, kSpanSpectrum = 3 /* pass A / block forward | pass B | channels */ };

// one launch of the bank's block kernel over the launch group (ev0 / ev1: stamped by the dispatch itself, may be null)
static int launch_bank(fdc_pipeline *p, const fdc_pipeline::Bank &bk, const float2 *in0, float2 *o, int nb, int m0, int nblocks, int64_t first_block,
                       unsigned out_bytes, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1)
{
    const bool half = bk.r == bk.L / 2;
    switch (bk.L) {
    case 256:
        HIPCHK(fdc::launch_poly_block(in0, (size_t)p->H, o, nb, m0, nblocks, p->d_tw256, p->d_twq, bk.d_cbt, bk.d_shn, bk.d_slot_off, out_bytes, p->ncu - p->reserved_cu,
                                      p->block_hints, s, p->d_dbg, bk.r, first_block + m0, ev0, ev1, p->R, p->d_fscr, p->N));
        break;
    case 512:
        HIPCHK(fdc::launch_poly_block512(in0, (size_t)p->H, o, nb, m0, nblocks, p->d_tw256, p->d_tw512, p->d_twq512, bk.d_cbt, bk.d_shn, bk.d_slot_off,
                                         out_bytes, p->ncu - p->reserved_cu, p->block_hints, s, ev0, ev1, p->R, p->d_fscr, half, p->N));
        break;
    case 1024:
        HIPCHK(fdc::launch_poly_block1024(in0, (size_t)p->H, o, nb, m0, nblocks, p->d_tw256, p->d_tw1k, p->d_twq1k, bk.d_cbt, bk.d_shn, bk.d_slot_off,
                                          out_bytes, p->ncu - p->reserved_cu, p->block_hints, s, ev0, ev1, half, p->R, p->d_fscr, p->N));
        break;
    default:
        HIPCHK(fdc::launch_poly_block_narrow(bk.L, in0, (size_t)p->H, o, nb, m0, nblocks, bk.d_tab, bk.d_cbt, bk.d_slot_off, out_bytes, p->ncu - p->reserved_cu,
                                             p->block_hints, s, ev0, ev1, p->R, p->d_fscr, bk.r, p->N));
    }
    return FDC_OK;
}

int fdc_pipeline_process_device(fdc_pipeline *p, const void *d_ring, int64_t first_block, int nblocks,
                                void *d_out, void *d_spectrum, void *stream)
{
    FDC_ENTRY("fdc_pipeline_process_device")
    if (!p) return fail(FDC_ERR_INVALID_ARGUMENT, "null handle");
    if (nblocks < 0 || first_block < 0) return fail(FDC_ERR_INVALID_ARGUMENT, "negative block count/index");
    if (nblocks == 0) return FDC_OK;
    if (!d_ring || (p->C > 0 && !d_out)) return fail(FDC_ERR_INVALID_ARGUMENT, "null device buffer");
    if (d_spectrum && !p->cfg.keep_spectrum) return fail(FDC_ERR_INVALID_ARGUMENT, "spectrum output needs keep_spectrum");
    // one call produces less than 4 GiB (32-bit output offsets; checked for max_blocks at create): a longer call is refused, not sent down
    // a path whose internal spectrum may be partial (a split plan's remainder-only spectrum under the channel kernels of ALL channels)
    if ((int64_t)nblocks * p->sum_lout * 8 > 0xFFFFF000ll)
        return fail(FDC_ERR_INVALID_ARGUMENT, "nblocks %d x %lld output samples per block is more than the 4 GiB one call may produce", nblocks, (long long)p->sum_lout);
    HIPCHK(hipSetDevice(p->cfg.device_id));
    hipStream_t s = stream ? (hipStream_t)stream : p->stream;
    const float2 *ring = static_cast<const float2 *>(d_ring);
    float2 *o = static_cast<float2 *>(d_out);
    const bool use_poly = p->poly_ok && !d_spectrum;
    const unsigned out_bytes = (unsigned)((int64_t)nblocks * p->sum_lout * 8);
    for (int m0 = 0; m0 < nblocks; m0 += p->chunk) {
        const int nb = std::min(p->chunk, nblocks - m0);
        float2 *spec = d_spectrum ? static_cast<float2 *>(d_spectrum) + (size_t)m0 * p->N : p->d_spec;
        hipEvent_t ev[3]; hipEvent_t *evp = nullptr; std::array<size_t, 5> span{};
        // events on every timing_stride-th launch group only: a sample of the launches, so that the packets between the
        // kernels (measured 7-17 us per group) do not slow the region they time
        const bool tg = p->timing && (p->timing_seq++ % p->timing_stride) == 0;
        if (tg) {
            for (int i = 0; i < 4; i++) { int rc = get_event(p, &span[i]); if (rc) return rc; }
            for (int i = 0; i < 3; i++) ev[i] = p->events[span[i]];
            evp = ev;
        }
        const float2 *in0 = ring + (size_t)m0 * p->H;
        // overlap-save gather fused into the load (item m at ring + m*H), fftshift + 1/N into the store.
        // A block kernel gives a whole block to one compute unit: a launch group of fewer blocks than the device has compute
        // units leaves the rest idle (one block takes ~42 us there, however few there are).  Short calls — a scheduler handing
        // over a few items — take the two-launch form where the plan has one (ONE bank on its grid), which spreads every block over the device.
        const bool few = nb < p->block_min;
        if (p->fused && !d_spectrum) {
            // N = 4096: one launch, nothing but the input samples and the output samples crosses the memory interface
            if (tg) HIPCHK(hipEventRecord(p->events[span[0]], s));
            HIPCHK(fdc::launch_fused4096(in0, (size_t)p->H, o, nb, p->R, m0, nblocks, first_block, p->d_tw, p->ntab, p->d_wins, p->d_f4rows, p->f4_cls, p->f4_teams, s));
            if (tg) { HIPCHK(hipEventRecord(p->events[span[1]], s)); span[2] = span[3] = span[1]; span[4] = kSpanBanks; p->ev_spans.push_back(span); }
            continue;
        }
        if (use_poly && p->poly_block && !(few && two_launch_possible(p))) {
            // one launch per bank: nothing but the input rows and the output samples crosses the memory interface.
            // timing: the first launch's begin and the last one's end are the dispatches' own stamps, no packets around the kernels
            for (size_t k = 0; k < p->banks.size(); k++) {
                const int rcb = launch_bank(p, p->banks[k], in0, o, nb, m0, nblocks, first_block, out_bytes, s,
                                            tg && k == 0 ? p->events[span[0]] : nullptr, tg && k + 1 == p->banks.size() ? p->events[span[1]] : nullptr);
                if (rcb != FDC_OK) return rcb;
            }
            for (const auto &al : p->bank_alias) {
                const fdc::ChanDev &dc = p->chans[(size_t)al.first], &sc = p->chans[(size_t)al.second];
                HIPCHK(hipMemcpyAsync(o + (size_t)nblocks * dc.out_off + (size_t)m0 * dc.lout, o + (size_t)nblocks * sc.out_off + (size_t)m0 * sc.lout,
                                      sizeof(float2) * (size_t)nb * dc.lout, hipMemcpyDeviceToDevice, s));
            }
            if (p->split) {
                const int rcr = run_remainder(p, ring, m0, nb, nblocks, first_block, o, few, s, tg ? p->events[span[2]] : nullptr, tg ? p->events[span[3]] : nullptr);
                if (rcr != FDC_OK) return rcr;
            } else if (tg) span[2] = span[3] = span[1];
            if (tg) { span[4] = kSpanBanks; p->ev_spans.push_back(span); }
            continue;
        }
        if (use_poly) {
            // ONE bank on its grid: window + IFFT commuted in front of pass B; only G (lout * N / l per block) between the two launches
            const fdc_pipeline::Bank &bk = p->banks[0];
            if (tg) HIPCHK(hipEventRecord(p->events[span[0]], s));
            if (bk.L != 256)
                HIPCHK(fdc::launch_poly_stage1_generic(in0, (size_t)p->H, p->d_g, p->N, bk.L, p->R, nb, bk.d_shn, p->d_tw, p->ntab, p->d_t2g, s));
            else
                HIPCHK(fdc::launch_poly_stage1(in0, (size_t)p->H, p->d_g, p->N / 256, p->R, nb, p->d_tw256, p->d_twq, bk.d_cbt, bk.d_shn, p->ncu - p->reserved_cu, s));
            if (tg) { HIPCHK(hipEventRecord(p->events[span[1]], s)); span[2] = span[1]; }   // the end of stage 1 IS the start of stage 2
            if (bk.L != 256)
                HIPCHK(fdc::launch_poly_stage2_generic(p->d_g, o, p->N / bk.L, p->R, nb, m0, nblocks, bk.d_slot_off, p->d_tw, p->ntab, s, bk.L));
            else if (p->N != 65536 && p->N != 262144)
                HIPCHK(fdc::launch_poly_stage2_generic(p->d_g, o, p->N / 256, p->R, nb, m0, nblocks, bk.d_slot_off, p->d_tw, p->ntab, s));
            else
                HIPCHK(fdc::launch_poly_stage2(p->d_g, o, p->N / 256, p->R, nb, m0, nblocks, p->d_tw256, p->d_tw1024, bk.d_slot_off, out_bytes, p->ncu - p->reserved_cu, s));
            if (p->split) {                                     // (timing: the remainder is counted with stage 2)
                const int rcr = run_remainder(p, ring, m0, nb, nblocks, first_block, o, few, s, nullptr, nullptr);
                if (rcr != FDC_OK) return rcr;
            }
            if (tg) {
                HIPCHK(hipEventRecord(p->events[span[3]], s));
                span[4] = kSpanTwoLaunch;
                p->ev_spans.push_back(span);
            }
            continue;
        }
        // a spectrum in memory.  (A split plan's internal spectrum holds its remainder's bins only: whoever gets here with one — a caller's
        // spectrum buffer — writes a full spectrum into THAT buffer, d_keep is not applied.)
        // the power of the 16-bin groups of the caller's spectrum (fdc_pipeline_process_device_power): summed by the block kernel while the bins are in
        // its registers; by a pass over the spectrum where another transform ran
        float *const gp = (p->gpow_base && d_spectrum) ? p->gpow_base + (size_t)(spec - p->gpow_spec) / 16 : nullptr;
        if (p->fwd_block && !few)
            HIPCHK(fdc::launch_block_fft(p->N, in0, (size_t)p->H, spec, nb, p->d_tw256, p->d_ftwq, p->d_fcbt,
                                              p->d_fshn, p->d_fslot, p->d_fscr, p->ncu - p->reserved_cu, p->block_hints, s, evp, d_spectrum ? nullptr : p->d_keep, gp));
        else {
            if (p->N == 65536 && !p->cfg_generic)
                HIPCHK(fdc::launch_fft65536(in0, (size_t)p->H, spec, p->d_tmp, nb, p->N / 2,
                                            1.0f / (float)p->N, p->d_tw256, p->d_twf, s, evp));
            else
                HIPCHK(fdc::launch_fft(in0, (size_t)p->H, spec, p->d_tmp, p->N, nb, false, 0, p->N / 2,
                                       1.0f / (float)p->N, p->d_tw, p->ntab, s, evp, p->d_twf, p->cfg_generic, d_spectrum ? ~0ull : p->keep4096));
            if (gp) HIPCHK(fdc::launch_group_power(spec, p->N, nb, gp, s));
        }
        { const int rcc = run_channel_groups(p, false, spec, o, nb, m0, nblocks, first_block, s); if (rcc != FDC_OK) return rcc; }
        if (tg) {
            HIPCHK(hipEventRecord(p->events[span[3]], s));
            span[4] = p->N <= fdc::kMaxLdsFft ? kSpanSpectrumLds : kSpanSpectrum;
            p->ev_spans.push_back(span);
        }
    }
    return FDC_OK;
    FDC_ENTRY_END
}

int fdc_pipeline_process_device_power(fdc_pipeline *p, const void *d_ring, int64_t first_block, int nblocks, void *d_out, void *d_spectrum,
                                      void *d_group_power, void *stream)
{
    FDC_ENTRY("fdc_pipeline_process_device_power")
    if (!p) return fail(FDC_ERR_INVALID_ARGUMENT, "null handle");
    if (d_group_power && (!d_spectrum || (p->N & 15))) return fail(FDC_ERR_INVALID_ARGUMENT, "group powers go with a spectrum output of a block length that is a multiple of 16");
    p->gpow_base = static_cast<float *>(d_group_power);
    p->gpow_spec = static_cast<const float2 *>(d_spectrum);
    const int rc = fdc_pipeline_process_device(p, d_ring, first_block, nblocks, d_out, d_spectrum, stream);
    p->gpow_base = nullptr; p->gpow_spec = nullptr;
    return rc;
    FDC_ENTRY_END
}

int fdc_pipeline_last_kernel_ms(fdc_pipeline *p, float *ms, int n)
{
    FDC_ENTRY("fdc_pipeline_last_kernel_ms")
    if (!p || !ms || n < 4) return fail(FDC_ERR_INVALID_ARGUMENT, "need room for 4 values");
    ms[0] = ms[1] = ms[2] = 0.f;
    ms[3] = (float)p->ev_spans.size();
    for (auto &sp : p->ev_spans) {
        float a = 0, b = 0, c = 0;
        HIPCHK(hipEventSynchronize(p->events[sp[3]]));
        HIPCHK(hipEventElapsedTime(&a, p->events[sp[0]], p->events[sp[1]]));
        HIPCHK(hipEventElapsedTime(&b, p->events[sp[1]], p->events[sp[2]]));
        HIPCHK(hipEventElapsedTime(&c, p->events[sp[2]], p->events[sp[3]]));
        // every span says which form it ran (a call may mix them: a short last launch group takes the two-launch form)
        switch ((int)sp[4]) {
        case kSpanBanks: ms[0] += a; ms[1] += b; ms[2] += c; break;        // bank launches; remainder: forward transform, channel kernels
        case kSpanTwoLaunch: ms[0] += a; ms[1] += c; break;                // stage 1, stage 2 (b = the wait between them)
        case kSpanSpectrumLds: ms[1] += a + b; ms[2] += c; break;
        default: ms[0] += a; ms[1] += b; ms[2] += c;
        }
    }
    p->ev_used = 0; p->ev_spans.clear();
    return 4;
    FDC_ENTRY_END
}

void fdc_pipeline_reset(fdc_pipeline *p)
{
    if (!p) return;
    p->blockcount = 0;
    // (a batch of the pipelined hier entry that sits transformed in the bank stays there: the sink blocks are blocks of their own with their own
    // state, its PDUs come out with the next call or fdc_pipeline_flush_sinks)
    if (p->ev_hier && p->hier_ring_busy) (void)hipEventSynchronize(p->ev_hier);
    if (p->d_ring) {
        (void)hipSetDevice(p->cfg.device_id);
        (void)hipMemsetAsync(p->d_ring, 0, sizeof(float2) * (size_t)p->ovl, p->stream);
        (void)hipStreamSynchronize(p->stream);
    }
}

// The whole-call spectrum of a work() that hands it to the host (debug port, python/FrequencyDomainChannelizer.py:152-158, :314-315) when
// no bank's buffer takes it: allocated at the first such call, kept (no hipMalloc / hipFree in the steady state of any entry).
static int spec_staging(fdc_pipeline *p, float2 **out)
{
    if (!p->d_specfull) HIPCHK(hipMalloc(&p->d_specfull, sizeof(float2) * (size_t)p->cfg.max_blocks * p->N));
    *out = p->d_specfull;
    return FDC_OK;
}

static int work_io_setup(fdc_pipeline *p)
{
    if (p->d_ring) return FDC_OK;
    HIPCHK(hipMalloc(&p->d_ring, sizeof(float2) * ((size_t)p->ovl + (size_t)p->cfg.max_blocks * p->H)));
    HIPCHK(hipMemsetAsync(p->d_ring, 0, sizeof(float2) * (size_t)p->ovl, p->stream));   // zero history (overlap_save_impl.cc:52)
    if (p->sum_lout > 0) HIPCHK(hipMalloc(&p->d_out, sizeof(float2) * (size_t)p->cfg.max_blocks * p->sum_lout));
    HIPCHK(hipStreamCreateWithFlags(&p->s_in, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&p->s_out, hipStreamNonBlocking));
    for (int i = 0; i < 2; i++) {
        HIPCHK(hipEventCreateWithFlags(&p->ev_in[i], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&p->ev_k[i], hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&p->ev_out[i], hipEventDisableTiming));
    }
    // sub-batch: about 8 MiB of input (measured best of 2-16 MiB on MI355X/PCIe5, staged and pinned): long against a
    // transfer's launch cost, short against the call
    int64_t sub = (8ll << 20) / ((int64_t)p->H * 8);
    if (p->cfg.host_sub_blocks > 0) sub = p->cfg.host_sub_blocks;
    if (const char *e = fdc::debug_env("FDC_HOST_SUB")) if (atoi(e) > 0) sub = atoi(e);
    p->sub = (int)std::max<int64_t>(1, std::min<int64_t>(sub, p->cfg.max_blocks));
    if (p->C > 0) {
        // scatter table: pinned and device-mapped, the scatter kernel reads it in place (no per-call upload)
        HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&p->pin_tab), sizeof(fdc::ScatterEnt) * p->C, hipHostMallocMapped));
        HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void **>(&p->d_tab), p->pin_tab, 0));
    }
    return FDC_OK;
}

// Host entry.  The call is cut into sub-batches; sub-batch k's H2D copy (stream s_in), its kernels (p->stream) and
// its D2H leg (s_out) run beside the neighbouring sub-batches' other legs, so a long call moves at the rate of the
// slower PCIe direction instead of the sum of all legs.  Caller buffers pinned with fdc_host_register() are DMA'd in
// place (input: one async copy; outputs: one scatter kernel storing straight into the caller's per-channel buffers).
// Pageable input is copied by the runtime's pin-on-the-fly path from a feeder thread; pageable outputs come back
// through two pinned staging slots and a CPU copy on this thread.
// span: the call is one contiguous span of a longer stream handed over by a dispatcher (fdc_pipeline_work_span): the history comes
// from `halo` (N/R samples, NULL = zeros) and the block counter from `first_block` instead of from the handle.
static int pipeline_work_impl(fdc_pipeline *p, const void *in, int nblocks, void *const *outs, void *spectrum,
                              float2 *d_spec_dst, bool span = false, const void *halo = nullptr, int64_t first_block = 0)
{
    if (!p) return fail(FDC_ERR_INVALID_ARGUMENT, "null handle");
    if (nblocks < 0) return fail(FDC_ERR_INVALID_ARGUMENT, "negative item count");
    if (nblocks == 0) return 0;
    if (nblocks > p->cfg.max_blocks) return fail(FDC_ERR_INVALID_ARGUMENT, "nblocks %d above max_blocks %d", nblocks, p->cfg.max_blocks);
    if (!in || (p->C > 0 && !outs)) return fail(FDC_ERR_INVALID_ARGUMENT, "null host buffer");
    if ((spectrum || d_spec_dst) && !p->cfg.keep_spectrum) return fail(FDC_ERR_INVALID_ARGUMENT, "spectrum output needs keep_spectrum");
    HIPCHK(hipSetDevice(p->cfg.device_id));
    int rc = work_io_setup(p);
    if (rc != FDC_OK) return rc;
    hipStream_t s = p->stream;
    const size_t nin = (size_t)nblocks * p->H;
    const float2 *hin = static_cast<const float2 *>(in);
    if (span) {
        // every kernel of the call is enqueued on s behind this copy; the previous call ended with s drained
        if (halo) HIPCHK(hipMemcpyAsync(p->d_ring, halo, sizeof(float2) * (size_t)p->ovl, hipMemcpyHostToDevice, s));
        else HIPCHK(hipMemsetAsync(p->d_ring, 0, sizeof(float2) * (size_t)p->ovl, s));
        p->blockcount = first_block;
    }

    // spectrum wanted (debug port / sinks): every sub-batch writes its part of one whole-call buffer
    float2 *d_specfull = d_spec_dst;
    if (spectrum && !d_specfull && (rc = spec_staging(p, &d_specfull)) != FDC_OK) return rc;

    const bool in_reg = host_registered(in, sizeof(float2) * nin);
    bool out_reg = p->C > 0;
    for (int c = 0; c < p->C && out_reg; c++) {
        fdc::ScatterEnt &e = p->pin_tab[c];
        e.dst = nullptr; e.out_off = p->chans[c].out_off; e.lout = p->chans[c].lout; e.pad = 0;
        if (outs[c] && !host_registered(outs[c], sizeof(float2) * (size_t)nblocks * p->chans[c].lout, reinterpret_cast<void **>(&e.dst)))
            out_reg = false;
    }
    const int sub = p->sub;
    if (!out_reg && p->C > 0 && !p->pin_out[0])
        for (int i = 0; i < 2; i++)
            HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&p->pin_out[i]), sizeof(float2) * (size_t)sub * p->sum_lout, hipHostMallocDefault));
    // staged outputs: wait for sub-batch j's D2H, then hand its pieces to the caller's per-channel buffers
    auto drain = [&](int j) -> int {
        const int b0 = j * sub, nb = std::min(sub, nblocks - b0);
        HIPCHK(hipEventSynchronize(p->ev_out[j & 1]));
        const float2 *src = p->pin_out[j & 1];
        for (int c = 0; c < p->C; c++) {
            if (!outs[c]) continue;
            const size_t lo = (size_t)p->chans[c].lout;
            std::memcpy(static_cast<float2 *>(outs[c]) + (size_t)b0 * lo, src + (size_t)nb * p->chans[c].out_off, sizeof(float2) * nb * lo);
        }
        return FDC_OK;
    };
    const int K = (nblocks + sub - 1) / sub;
    if (K == 1) {
        // short call (the usual work() of a running flowgraph): nothing to overlap, one stream, one synchronisation
        HIPCHK(hipMemcpyAsync(p->d_ring + p->ovl, hin, sizeof(float2) * nin, hipMemcpyHostToDevice, s));
        rc = fdc_pipeline_process_device(p, p->d_ring, p->blockcount, nblocks, p->d_out, d_specfull, s);
        if (rc != FDC_OK) return rc;
        if (p->C > 0) {
            if (out_reg) HIPCHK(fdc::launch_scatter_out(p->d_out, p->d_tab, p->C, nblocks, 0, s));
            else HIPCHK(hipMemcpyAsync(p->pin_out[0], p->d_out, sizeof(float2) * (size_t)nblocks * p->sum_lout, hipMemcpyDeviceToHost, s));
        }
        if (spectrum) HIPCHK(hipMemcpyAsync(spectrum, d_specfull, sizeof(float2) * (size_t)nblocks * p->N, hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(p->d_ring, p->d_ring + nin, sizeof(float2) * (size_t)p->ovl, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
        if (p->C > 0 && !out_reg)
            for (int c = 0; c < p->C; c++)
                if (outs[c])
                    std::memcpy(outs[c], p->pin_out[0] + (size_t)nblocks * p->chans[c].out_off,
                                sizeof(float2) * (size_t)nblocks * p->chans[c].lout);
        p->blockcount += nblocks;
        return nblocks;
    }
    // Pageable input: the runtime pins the pages of each copy on the fly and DMAs from them (measured faster than a CPU
    // copy into pinned staging), but such a copy holds its calling thread until it is done — so a feeder thread issues
    // them, and this thread spends that time launching kernels and draining finished outputs.
    struct Feeder {
        std::thread th; std::mutex mu; std::condition_variable cv; int done = 0; hipError_t err = hipSuccess;
        ~Feeder() { if (th.joinable()) th.join(); }
    } feeder;
    if (!in_reg) {
        const int dev = p->cfg.device_id, Hs = p->H;
        float2 *ring_in = p->d_ring + p->ovl;
        hipStream_t sin = p->s_in;
        feeder.th = std::thread([&feeder, dev, Hs, ring_in, sin, hin, K, sub, nblocks] {
            hipError_t e = hipSetDevice(dev);
            for (int k = 0; k < K; k++) {
                const int b0 = k * sub, nb = std::min(sub, nblocks - b0);
                if (e == hipSuccess)
                    e = hipMemcpyAsync(ring_in + (size_t)b0 * Hs, hin + (size_t)b0 * Hs, sizeof(float2) * (size_t)nb * Hs,
                                       hipMemcpyHostToDevice, sin);
                if (e == hipSuccess) e = hipStreamSynchronize(sin);
                std::lock_guard<std::mutex> lk(feeder.mu);
                feeder.done = k + 1; feeder.err = e;
                feeder.cv.notify_one();
            }
        });
    }
    for (int k = 0; k < K; k++) {
        const int slot = k & 1, b0 = k * sub, nb = std::min(sub, nblocks - b0);
        if (in_reg) {
            HIPCHK(hipMemcpyAsync(p->d_ring + p->ovl + (size_t)b0 * p->H, hin + (size_t)b0 * p->H,
                                  sizeof(float2) * (size_t)nb * p->H, hipMemcpyHostToDevice, p->s_in));
            HIPCHK(hipEventRecord(p->ev_in[slot], p->s_in));
            HIPCHK(hipStreamWaitEvent(s, p->ev_in[slot], 0));
        } else {
            std::unique_lock<std::mutex> lk(feeder.mu);
            feeder.cv.wait(lk, [&] { return feeder.done > k; });               // sub-batch k is on the device
            if (feeder.err != hipSuccess) return fail(FDC_ERR_HIP, "input copy failed: %s", hipGetErrorString(feeder.err));
        }
        float2 *dok = p->d_out + (size_t)b0 * p->sum_lout;                    // [channel][nb*lout] of this sub-batch
        rc = fdc_pipeline_process_device(p, p->d_ring + (size_t)b0 * p->H, p->blockcount + b0, nb, dok,
                                         d_specfull ? d_specfull + (size_t)b0 * p->N : nullptr, s);
        if (rc != FDC_OK) return rc;
        if (p->C == 0) continue;
        HIPCHK(hipEventRecord(p->ev_k[slot], s));
        HIPCHK(hipStreamWaitEvent(p->s_out, p->ev_k[slot], 0));
        if (out_reg) {
            HIPCHK(fdc::launch_scatter_out(dok, p->d_tab, p->C, nb, b0, p->s_out));
        } else {
            HIPCHK(hipMemcpyAsync(p->pin_out[slot], dok, sizeof(float2) * (size_t)nb * p->sum_lout, hipMemcpyDeviceToHost, p->s_out));
            HIPCHK(hipEventRecord(p->ev_out[slot], p->s_out));
            if (k >= 1 && (rc = drain(k - 1)) != FDC_OK) return rc;
        }
    }
    if (!out_reg && p->C > 0 && (rc = drain(K - 1)) != FDC_OK) return rc;
    if (spectrum) HIPCHK(hipMemcpyAsync(spectrum, d_specfull, sizeof(float2) * (size_t)nblocks * p->N, hipMemcpyDeviceToHost, s));
    // history <- last ovl samples of this call (overlap_save_impl.cc:78); src and dst never overlap (H >= ovl)
    HIPCHK(hipMemcpyAsync(p->d_ring, p->d_ring + nin, sizeof(float2) * (size_t)p->ovl, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipStreamSynchronize(p->s_out));
    p->blockcount += nblocks;
    return nblocks;
}

int fdc_pipeline_work(fdc_pipeline *p, const void *in, int nblocks, void *const *outs, void *spectrum)
{
    FDC_ENTRY("fdc_pipeline_work")
    return pipeline_work_impl(p, in, nblocks, outs, spectrum, nullptr);
    FDC_ENTRY_END
}

int fdc_pipeline_work_span(fdc_pipeline *p, const void *halo, const void *in, int64_t first_block, int nblocks, void *const *outs,
                           void *spectrum)
{
    FDC_ENTRY("fdc_pipeline_work_span")
    if (first_block < 0) return fail(FDC_ERR_INVALID_ARGUMENT, "negative block index");
    return pipeline_work_impl(p, in, nblocks, outs, spectrum, nullptr, true, halo, first_block);
    FDC_ENTRY_END
}

// Real input: the float items are copied to the device and widened there into the complex ring (imaginary part 0); the
// rest of the call is the one-stream form of the complex entry.
static int pipeline_work_real_impl(fdc_pipeline *p, const void *in, int nblocks, void *const *outs, void *spectrum, bool span,
                                   const void *halo, int64_t first_block)
{
    if (!p) return fail(FDC_ERR_INVALID_ARGUMENT, "null handle");
    if (nblocks < 0) return fail(FDC_ERR_INVALID_ARGUMENT, "negative item count");
    if (nblocks == 0) return 0;
    if (nblocks > p->cfg.max_blocks) return fail(FDC_ERR_INVALID_ARGUMENT, "nblocks %d above max_blocks %d", nblocks, p->cfg.max_blocks);
    if (!in || (p->C > 0 && !outs)) return fail(FDC_ERR_INVALID_ARGUMENT, "null host buffer");
    if (spectrum && !p->cfg.keep_spectrum) return fail(FDC_ERR_INVALID_ARGUMENT, "spectrum output needs keep_spectrum");
    HIPCHK(hipSetDevice(p->cfg.device_id));
    int rc = work_io_setup(p);
    if (rc != FDC_OK) return rc;
    hipStream_t s = p->stream;
    const size_t nin = (size_t)nblocks * p->H;
    // d_real: [N/R history samples of a span call][max_blocks*H new samples]
    if (!p->d_real) HIPCHK(hipMalloc(&p->d_real, sizeof(float) * ((size_t)p->ovl + (size_t)p->cfg.max_blocks * p->H)));
    float2 *d_specfull = nullptr;
    if (spectrum && (rc = spec_staging(p, &d_specfull)) != FDC_OK) return rc;
    HIPCHK(hipMemcpyAsync(p->d_real + p->ovl, in, sizeof(float) * nin, hipMemcpyHostToDevice, s));
    if (span) {
        if (halo) HIPCHK(hipMemcpyAsync(p->d_real, halo, sizeof(float) * (size_t)p->ovl, hipMemcpyHostToDevice, s));
        else HIPCHK(hipMemsetAsync(p->d_real, 0, sizeof(float) * (size_t)p->ovl, s));
        HIPCHK(fdc::launch_real_to_complex(p->d_real, p->d_ring, (size_t)p->ovl + nin, s));
        p->blockcount = first_block;
    } else {
        HIPCHK(fdc::launch_real_to_complex(p->d_real + p->ovl, p->d_ring + p->ovl, nin, s));
    }
    rc = fdc_pipeline_process_device(p, p->d_ring, p->blockcount, nblocks, p->d_out, d_specfull, s);
    if (rc != FDC_OK) return rc;
    for (int c = 0; c < p->C; c++)
        if (outs[c])
            HIPCHK(hipMemcpyAsync(outs[c], p->d_out + (size_t)nblocks * p->chans[c].out_off,
                                  sizeof(float2) * (size_t)nblocks * p->chans[c].lout, hipMemcpyDeviceToHost, s));
    if (spectrum) HIPCHK(hipMemcpyAsync(spectrum, d_specfull, sizeof(float2) * (size_t)nblocks * p->N, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(p->d_ring, p->d_ring + nin, sizeof(float2) * (size_t)p->ovl, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    p->blockcount += nblocks;
    return nblocks;
}

int fdc_pipeline_work_real(fdc_pipeline *p, const void *in, int nblocks, void *const *outs, void *spectrum)
{
    FDC_ENTRY("fdc_pipeline_work_real")
    return pipeline_work_real_impl(p, in, nblocks, outs, spectrum, false, nullptr, 0);
    FDC_ENTRY_END
}

int fdc_pipeline_work_span_real(fdc_pipeline *p, const void *halo, const void *in, int64_t first_block, int nblocks, void *const *outs,
                                void *spectrum)
{
    FDC_ENTRY("fdc_pipeline_work_span_real")
    if (first_block < 0) return fail(FDC_ERR_INVALID_ARGUMENT, "negative block index");
    return pipeline_work_real_impl(p, in, nblocks, outs, spectrum, true, halo, first_block);
    FDC_ENTRY_END
}

// fdc_pipeline_work_sinks on a bank created with FDC_SINKS_LOOKAHEAD: the pipelined hier block.  What one call does:
//   - the items' copy to the device (own stream), their forward transform (+ channel kernels) into the bank's NEXT-batch buffer and its
//     power cells (fdc_sinks_prepare) — all on the bank's fill stream, behind the copy;
//   - fdc_sinks_submit_device for the batch the call BEFORE left there: its decision chains, the host's one wait for their summary, its
//     extractions — beside this call's copy and transform — and the hand-out of the batch before that one (its payload copy ran meanwhile);
//   - the wait for this call's input copy (the caller's buffer is not retained), and for the channel outputs / the debug spectrum if any.
// So the items of call n come back as PDUs from call n + 2 (device engine; n + 1 on the host engine, whose submit is synchronous), and
// fdc_pipeline_flush_sinks hands out what is still inside at stop().  Nothing here waits for the transform of the call's own items unless
// the call has stream outputs: with pinned input the call costs what its input copy costs.
static int work_sinks_pipelined(fdc_pipeline *p, const void *in, int nblocks, void *const *outs, void *spectrum, fdc_sinks *sinks)
{
    if (nblocks < 0) return fail(FDC_ERR_INVALID_ARGUMENT, "negative item count");
    if (nblocks == 0) return 0;
    if (nblocks > p->cfg.max_blocks) return fail(FDC_ERR_INVALID_ARGUMENT, "nblocks %d above max_blocks %d", nblocks, p->cfg.max_blocks);
    if (!in || (p->C > 0 && !outs)) return fail(FDC_ERR_INVALID_ARGUMENT, "null host buffer");
    if (!p->cfg.keep_spectrum) return fail(FDC_ERR_INVALID_ARGUMENT, "spectrum output needs keep_spectrum");
    if (p->hier_bank && p->hier_bank != sinks && p->hier_filled > 0)
        return fail(FDC_ERR_INVALID_ARGUMENT, "a batch of another bank is still inside this pipeline: fdc_pipeline_flush_sinks() with that bank first");
    if (p->hier_broken) return fail(FDC_ERR_HIP, "an earlier pipelined call failed after it had advanced the stream state: destroy the pipeline and the bank");
    HIPCHK(hipSetDevice(p->cfg.device_id));
    int rc = work_io_setup(p);
    if (rc != FDC_OK) return rc;
    if (!p->ev_hier) {
        HIPCHK(hipEventCreateWithFlags(&p->ev_hier, hipEventDisableTiming));
        HIPCHK(hipStreamSynchronize(p->stream));                  // work_io_setup zeroes the history on the handle's own stream; this entry runs on others
    }
    p->hier_bank = sinks;
    hipStream_t fs = static_cast<hipStream_t>(fdc_sinks_fill_stream(sinks));
    const bool first = p->hier_filled == 0;                       // stream start, or everything was flushed: the bank's current buffer is free
    float2 *dst = static_cast<float2 *>(first ? fdc_sinks_spectrum(sinks) : fdc_sinks_spectrum_ahead(sinks));
    // the persistent block kernels take every compute unit; the decision chains of the batch before run beside them on a few units left free
    // (long launch groups only: a group of one round is over before a chain would notice)
    if (!p->reserve_user) p->reserved_cu = std::min(nblocks, p->chunk) >= 2 * p->ncu ? p->ncu / 8 : 0;

    const size_t nin = (size_t)nblocks * p->H;
    // input: one copy on s_in.  The ring is read by the transform of the call before (fill stream) until ev_hier.
    if (p->hier_ring_busy) HIPCHK(hipStreamWaitEvent(p->s_in, p->ev_hier, 0));
    HIPCHK(hipMemcpyAsync(p->d_ring + p->ovl, in, sizeof(float2) * nin, hipMemcpyHostToDevice, p->s_in));
    HIPCHK(hipEventRecord(p->ev_in[0], p->s_in));
    HIPCHK(hipStreamWaitEvent(fs, p->ev_in[0], 0));
    // the power of the spectrum's 16-bin groups comes out of the forward kernel's epilogue: the bank's cells are summed from it (no pass over the spectrum)
    float *const gpw = static_cast<float *>(first ? fdc_sinks_group_power(sinks) : fdc_sinks_group_power_ahead(sinks));
    rc = fdc_pipeline_process_device_power(p, p->d_ring, p->blockcount, nblocks, p->d_out, dst, gpw, fs);
    if (rc != FDC_OK) return rc;
    // history <- last ovl samples of this call (overlap_save_impl.cc:78)
    HIPCHK(hipMemcpyAsync(p->d_ring, p->d_ring + nin, sizeof(float2) * (size_t)p->ovl, hipMemcpyDeviceToDevice, fs));
    HIPCHK(hipEventRecord(p->ev_hier, fs));
    p->hier_ring_busy = true;
    p->blockcount += nblocks;
    // from here on the call has happened as far as the stream state goes (history, block counter, the bank's buffer): a failure below cannot be
    // retried with the same items nor skipped — the pair is marked broken and every later call says so
    struct Broken { fdc_pipeline *p; bool ok = false; ~Broken() { if (!ok) p->hier_broken = true; } } guard{p};
    rc = gpw ? fdc_sinks_prepare_from_groups(sinks, nblocks, first ? 0 : 1) : fdc_sinks_prepare(sinks, nblocks, first ? 0 : 1);
    if (rc != FDC_OK) return rc;
    bool out_reg = p->C > 0;
    if (p->C > 0) {
        for (int c = 0; c < p->C && out_reg; c++) {
            fdc::ScatterEnt &e = p->pin_tab[c];
            e.dst = nullptr; e.out_off = p->chans[c].out_off; e.lout = p->chans[c].lout; e.pad = 0;
            if (outs[c] && !host_registered(outs[c], sizeof(float2) * (size_t)nblocks * p->chans[c].lout, reinterpret_cast<void **>(&e.dst)))
                out_reg = false;
        }
        if (out_reg) HIPCHK(fdc::launch_scatter_out(p->d_out, p->d_tab, p->C, nblocks, 0, fs));
        else
            for (int c = 0; c < p->C; c++)
                if (outs[c])
                    HIPCHK(hipMemcpyAsync(outs[c], p->d_out + (size_t)nblocks * p->chans[c].out_off,
                                          sizeof(float2) * (size_t)nblocks * p->chans[c].lout, hipMemcpyDeviceToHost, fs));
    }
    if (spectrum) HIPCHK(hipMemcpyAsync(spectrum, dst, sizeof(float2) * (size_t)nblocks * p->N, hipMemcpyDeviceToHost, fs));
    const int before = p->hier_filled;
    p->hier_filled = nblocks;
    if (before > 0) {
        const int rs = fdc_sinks_submit_device(sinks, before);
        if (rs < 0) return rs;
    } else {
        const int rs = fdc_sinks_submit_device(sinks, 0);          // nothing to submit yet: hands out a batch still in flight, or no PDUs
        if (rs < 0) return rs;
    }
    HIPCHK(hipEventSynchronize(p->ev_in[0]));
    if (p->C > 0 || spectrum) HIPCHK(hipStreamSynchronize(fs));
    guard.ok = true;
    return nblocks;
}

int fdc_pipeline_work_sinks(fdc_pipeline *p, const void *in, int nblocks, void *const *outs, void *spectrum,
                            fdc_sinks *sinks)
{
    FDC_ENTRY("fdc_pipeline_work_sinks")
    if (!sinks) return fail(FDC_ERR_INVALID_ARGUMENT, "null sinks handle");
    if (!p) return fail(FDC_ERR_INVALID_ARGUMENT, "null handle");
    if (fdc_sinks_blocklen(sinks) != p->N || nblocks > fdc_sinks_max_blocks(sinks))
        return fail(FDC_ERR_INVALID_ARGUMENT, "sinks were created for blocklen %d / %d blocks per call, pipeline call has %d / %d",
                    fdc_sinks_blocklen(sinks), fdc_sinks_max_blocks(sinks), p->N, nblocks);
    if (fdc_sinks_fill_stream(sinks)) return work_sinks_pipelined(p, in, nblocks, outs, spectrum, sinks);
    // the spectrum goes straight into the sinks' device buffer (no PCIe round trip), then the sinks run on it; their power cells are summed from
    // the group powers the forward kernel leaves beside the spectrum
    float *const gpw = static_cast<float *>(fdc_sinks_group_power(sinks));
    p->gpow_base = gpw; p->gpow_spec = static_cast<const float2 *>(fdc_sinks_spectrum(sinks));
    int rc = pipeline_work_impl(p, in, nblocks, outs, spectrum, static_cast<float2 *>(fdc_sinks_spectrum(sinks)));
    p->gpow_base = nullptr; p->gpow_spec = nullptr;
    if (rc < 0) return rc;
    if (gpw && rc > 0) { const int rp = fdc_sinks_prepare_from_groups(sinks, nblocks, 0); if (rp != FDC_OK) return rp; }
    const int rs = fdc_sinks_work_device(sinks, nblocks);
    return rs < 0 ? rs : rc;
    FDC_ENTRY_END
}

int fdc_pipeline_flush_sinks(fdc_pipeline *p, fdc_sinks *sinks)
{
    FDC_ENTRY("fdc_pipeline_flush_sinks")
    if (!p || !sinks) return fail(FDC_ERR_INVALID_ARGUMENT, "null handle");
    if (p->hier_broken) return fail(FDC_ERR_HIP, "an earlier pipelined call failed after it had advanced the stream state: destroy the pipeline and the bank");
    if (p->hier_bank == sinks && p->hier_filled > 0) {
        const int n = p->hier_filled;
        p->hier_filled = 0;
        const int rs = fdc_sinks_submit_device(sinks, n);
        if (rs != 0) return rs;                                    // an older batch's PDUs (or the host engine's: this batch's), or a failure
    }
    return fdc_sinks_flush(sinks);
    FDC_ENTRY_END
}

int32_t fdc_pipeline_sinks_latency(const fdc_pipeline *p, const fdc_sinks *sinks)
{
    if (!p || !sinks) return -1;
    if (!fdc_sinks_fill_stream(const_cast<fdc_sinks *>(sinks))) return 0;
    return fdc_sinks_engine(sinks) == 1 ? 2 : 1;
}

int fdc_pipeline_work_spectrum(fdc_pipeline *p, const void *in, int nblocks, void *const *outs, void *spectrum,
                               fdc_sinks *sinks)
{
    FDC_ENTRY("fdc_pipeline_work_spectrum")
    // hier block with inpveclen > 1 (py:284-290): items are spectra already; only multiply_const(1/N) and the channel /
    // sink branches remain.  The front-end state (overlap history) is untouched; the block counter advances.
    if (!p) return fail(FDC_ERR_INVALID_ARGUMENT, "null handle");
    if (nblocks < 0) return fail(FDC_ERR_INVALID_ARGUMENT, "negative item count");
    if (nblocks == 0) return 0;
    if (nblocks > p->cfg.max_blocks) return fail(FDC_ERR_INVALID_ARGUMENT, "nblocks %d above max_blocks %d", nblocks, p->cfg.max_blocks);
    if (!in || (p->C > 0 && !outs)) return fail(FDC_ERR_INVALID_ARGUMENT, "null host buffer");
    if (sinks && (fdc_sinks_blocklen(sinks) != p->N || nblocks > fdc_sinks_max_blocks(sinks)))
        return fail(FDC_ERR_INVALID_ARGUMENT, "sinks were created for blocklen %d / %d blocks per call, pipeline call has %d / %d",
                    fdc_sinks_blocklen(sinks), fdc_sinks_max_blocks(sinks), p->N, nblocks);
    HIPCHK(hipSetDevice(p->cfg.device_id));
    hipStream_t s = p->stream;
    float2 *d_full = sinks ? static_cast<float2 *>(fdc_sinks_spectrum(sinks)) : nullptr;
    if (!d_full) { const int rcs = spec_staging(p, &d_full); if (rcs != FDC_OK) return rcs; }
    if (p->sum_lout > 0 && !p->d_out) HIPCHK(hipMalloc(&p->d_out, sizeof(float2) * (size_t)p->cfg.max_blocks * p->sum_lout));
    const size_t n = (size_t)nblocks * p->N;
    HIPCHK(hipMemcpyAsync(d_full, in, sizeof(float2) * n, hipMemcpyHostToDevice, s));
    HIPCHK(fdc::launch_scale(d_full, d_full, n, 1.0f / (float)p->N, s));
    for (size_t g = 0; g < p->groups.size(); g++) {
        const int l = p->groups[g].first;
        if (l > 4096) {
            const int rcw = channels_wide(p, d_full, p->d_out, p->d_groups + p->group_off[g], (int)p->groups[g].second.size(), l, nblocks, 0, nblocks,
                                          p->blockcount, s);
            if (rcw != FDC_OK) return rcw;
        } else if (l == 256 && ((256 / p->R) & 1) == 0 && !p->cfg_generic)
            HIPCHK(fdc::launch_channels256(d_full, p->d_out, p->d_chans, p->d_groups + p->group_off[g],
                                           (int)p->groups[g].second.size(), p->g_aligned[g] != 0, p->g_out_aligned[g] != 0,
                                           p->N, p->R, nblocks, 0, nblocks, p->blockcount, p->d_wins, p->d_tw256, s));
        else
            HIPCHK(fdc::launch_channels(d_full, p->d_out, p->d_chans, p->d_groups + p->group_off[g],
                                        (int)p->groups[g].second.size(), l, p->N, p->R, nblocks, 0, nblocks, p->blockcount,
                                        p->d_wins, p->d_tw, p->ntab, s));
    }
    for (int c = 0; c < p->C; c++) {
        if (!outs[c]) continue;
        HIPCHK(hipMemcpyAsync(outs[c], p->d_out + (size_t)nblocks * p->chans[c].out_off,
                              sizeof(float2) * (size_t)nblocks * p->chans[c].lout, hipMemcpyDeviceToHost, s));
    }
    if (spectrum) HIPCHK(hipMemcpyAsync(spectrum, d_full, sizeof(float2) * n, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    p->blockcount += nblocks;
    if (sinks) {
        const int rs = fdc_sinks_work_device(sinks, nblocks);
        if (rs < 0) return rs;
    }
    return nblocks;
    FDC_ENTRY_END
}

/* ---------------- single-block faces ---------------- */
struct fdc_overlap_save {
    int dev, itemsize, outlen, ovl; hipStream_t s; unsigned char *d_ring = nullptr, *d_out = nullptr; int cap = 0;
};

int fdc_overlap_save_create(int device_id, int itemsize, int outputlen, int overlaplen, fdc_overlap_save **out)
{
    FDC_ENTRY("fdc_overlap_save_create")
    if (!out) return fail(FDC_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (itemsize < 1 || outputlen < 1 || overlaplen < 0 || overlaplen >= outputlen)
        return fail(FDC_ERR_INVALID_ARGUMENT, "overlap_save: need itemsize>=1 and 0 <= overlaplen < outputlen");
    if (2 * overlaplen > outputlen)
        return fail(FDC_ERR_INVALID_ARGUMENT, "overlap_save: overlaplen above outputlen/2 makes the reference read before its input buffer");
    int rc = select_device(device_id); if (rc) return rc;
    auto *b = new fdc_overlap_save{device_id, itemsize, outputlen, overlaplen, nullptr};
    HIPCHK(hipStreamCreateWithFlags(&b->s, hipStreamNonBlocking));
    *out = b;
    return FDC_OK;
    FDC_ENTRY_END
}

int fdc_overlap_save_work(fdc_overlap_save *b, const void *in, int nitems, void *out)
{
    FDC_ENTRY("fdc_overlap_save_work")
    if (!b) return fail(FDC_ERR_INVALID_ARGUMENT, "null handle");
    if (nitems <= 0) return nitems == 0 ? 0 : fail(FDC_ERR_INVALID_ARGUMENT, "negative item count");
    HIPCHK(hipSetDevice(b->dev));
    const size_t isz = b->itemsize, inb = isz * (b->outlen - b->ovl), outb = isz * b->outlen, ovb = isz * b->ovl;
    if (nitems > b->cap) {
        unsigned char *nr = nullptr, *no = nullptr;
        HIPCHK(hipMalloc(&nr, ovb + inb * nitems + 16));
        HIPCHK(hipMalloc(&no, outb * nitems));
        // on the block's OWN stream: it is non-blocking, so the null stream's memset would not be ordered in front of the copies and the
        // kernel below (round 6: the first item's history came out as whatever the allocation held, now and then)
        if (b->d_ring) HIPCHK(hipMemcpyAsync(nr, b->d_ring, ovb, hipMemcpyDeviceToDevice, b->s));
        else HIPCHK(hipMemsetAsync(nr, 0, ovb + 16, b->s));
        HIPCHK(hipStreamSynchronize(b->s));
        (void)hipFree(b->d_ring); (void)hipFree(b->d_out);
        b->d_ring = nr; b->d_out = no; b->cap = nitems;
    }
    HIPCHK(hipMemcpyAsync(b->d_ring + ovb, in, inb * nitems, hipMemcpyHostToDevice, b->s));
    HIPCHK(fdc::launch_overlap_save(b->d_ring, b->d_out, inb, outb, nitems, b->s));
    HIPCHK(hipMemcpyAsync(out, b->d_out, outb * nitems, hipMemcpyDeviceToHost, b->s));
    if (ovb) HIPCHK(hipMemcpyAsync(b->d_ring, b->d_ring + inb * nitems, ovb, hipMemcpyDeviceToDevice, b->s));
    HIPCHK(hipStreamSynchronize(b->s));
    return nitems;
    FDC_ENTRY_END
}

void fdc_overlap_save_destroy(fdc_overlap_save *b)
{
    if (!b) return;
    (void)hipFree(b->d_ring); (void)hipFree(b->d_out);
    if (b->s) (void)hipStreamDestroy(b->s);
    delete b;
}

struct fdc_vector_cut {
    int dev, itemsize, veclen, offset, blocklen; hipStream_t s; unsigned char *d_in = nullptr, *d_out = nullptr; int cap = 0;
};

int fdc_vector_cut_create(int device_id, int itemsize, int veclen, int offset, int blocklen, fdc_vector_cut **out)
{
    FDC_ENTRY("fdc_vector_cut_create")
    if (!out) return fail(FDC_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    if (itemsize < 1 || veclen < 1 || blocklen < 1 || offset < 0 || offset + blocklen > veclen)
        return fail(FDC_ERR_INVALID_ARGUMENT, "vector_cut: slice [offset, offset+blocklen) must lie inside the vector");
    int rc = select_device(device_id); if (rc) return rc;
    auto *b = new fdc_vector_cut{device_id, itemsize, veclen, offset, blocklen, nullptr};
    HIPCHK(hipStreamCreateWithFlags(&b->s, hipStreamNonBlocking));
    *out = b;
    return FDC_OK;
    FDC_ENTRY_END
}

int fdc_vector_cut_work(fdc_vector_cut *b, const void *in, int nitems, void *out)
{
    FDC_ENTRY("fdc_vector_cut_work")
    if (!b) return fail(FDC_ERR_INVALID_ARGUMENT, "null handle");
    if (nitems <= 0) return nitems == 0 ? 0 : fail(FDC_ERR_INVALID_ARGUMENT, "negative item count");
    HIPCHK(hipSetDevice(b->dev));
    const size_t inb = (size_t)b->itemsize * b->veclen, outb = (size_t)b->itemsize * b->blocklen;
    if (nitems > b->cap) {
        (void)hipFree(b->d_in); (void)hipFree(b->d_out); b->d_in = b->d_out = nullptr; b->cap = 0;
        HIPCHK(hipMalloc(&b->d_in, inb * nitems));
        HIPCHK(hipMalloc(&b->d_out, outb * nitems));
        b->cap = nitems;
    }
    HIPCHK(hipMemcpyAsync(b->d_in, in, inb * nitems, hipMemcpyHostToDevice, b->s));
    HIPCHK(fdc::launch_vector_cut(b->d_in, b->d_out, inb, (size_t)b->offset * b->itemsize, outb, nitems, b->s));
    HIPCHK(hipMemcpyAsync(out, b->d_out, outb * nitems, hipMemcpyDeviceToHost, b->s));
    HIPCHK(hipStreamSynchronize(b->s));
    return nitems;
    FDC_ENTRY_END
}

void fdc_vector_cut_destroy(fdc_vector_cut *b)
{
    if (!b) return;
    (void)hipFree(b->d_in); (void)hipFree(b->d_out);
    if (b->s) (void)hipStreamDestroy(b->s);
    delete b;
}

struct fdc_phase_window {
    int dev, l, R, shift, counter; hipStream_t s; float2 *d_win = nullptr, *d_in = nullptr, *d_out = nullptr; int cap = 0;
};

int fdc_phase_window_create(int device_id, int blocklen, int numphasestates, int shifts, float passbw, float stopbw,
                            int windowtype, fdc_phase_window **out)
{
    FDC_ENTRY("fdc_phase_window_create")
    if (!out) return fail(FDC_ERR_INVALID_ARGUMENT, "null argument");
    *out = nullptr;
    // lib/phase_shifting_windowing_vcc_impl.cc:46-53
    if (passbw <= 0.0f) return fail(FDC_ERR_INVALID_ARGUMENT, "PassBw in phase_shifting_windowing_vcc must not be <= 0");
    if (stopbw <= 0.0f) return fail(FDC_ERR_INVALID_ARGUMENT, "StopBw in phase_shifting_windowing_vcc must not be <= 0");
    if (stopbw < passbw) return fail(FDC_ERR_INVALID_ARGUMENT, "StopBw must not be < PassBw in phase_shifting_windowing_vcc");
    if (blocklen < 1 || numphasestates < 1) return fail(FDC_ERR_INVALID_ARGUMENT, "blocklen and numphasestates must be >= 1");
    int rc = select_device(device_id); if (rc) return rc;
    auto *b = new fdc_phase_window{device_id, blocklen, numphasestates,
                                   ((shifts % numphasestates) + numphasestates) % numphasestates, 0, nullptr};
    std::vector<std::complex<float>> w((size_t)numphasestates * blocklen);
    fdc::window_table(windowtype, blocklen, passbw, stopbw, numphasestates, 1, false, w.data());
    HIPCHK(hipStreamCreateWithFlags(&b->s, hipStreamNonBlocking));
    HIPCHK(hipMalloc(&b->d_win, sizeof(float2) * w.size()));
    HIPCHK(hipMemcpy(b->d_win, w.data(), sizeof(float2) * w.size(), hipMemcpyHostToDevice));
    *out = b;
    return FDC_OK;
    FDC_ENTRY_END
}

int fdc_phase_window_work(fdc_phase_window *b, const void *in, int nitems, void *out)
{
    FDC_ENTRY("fdc_phase_window_work")
    if (!b) return fail(FDC_ERR_INVALID_ARGUMENT, "null handle");
    if (nitems <= 0) return nitems == 0 ? 0 : fail(FDC_ERR_INVALID_ARGUMENT, "negative item count");
    HIPCHK(hipSetDevice(b->dev));
    const size_t nb = sizeof(float2) * (size_t)b->l * nitems;
    if (nitems > b->cap) {
        (void)hipFree(b->d_in); (void)hipFree(b->d_out); b->d_in = b->d_out = nullptr; b->cap = 0;
        HIPCHK(hipMalloc(&b->d_in, nb));
        HIPCHK(hipMalloc(&b->d_out, nb));
        b->cap = nitems;
    }
    HIPCHK(hipMemcpyAsync(b->d_in, in, nb, hipMemcpyHostToDevice, b->s));
    HIPCHK(fdc::launch_phase_window(b->d_in, b->d_out, b->d_win, b->l, b->R, b->shift, b->counter, nitems, b->s));
    HIPCHK(hipMemcpyAsync(out, b->d_out, nb, hipMemcpyDeviceToHost, b->s));
    HIPCHK(hipStreamSynchronize(b->s));
    b->counter = (int)(((long long)b->counter + (long long)(nitems % b->R) * b->shift) % b->R);
    return nitems;
    FDC_ENTRY_END
}

void fdc_phase_window_destroy(fdc_phase_window *b)
{
    if (!b) return;
    (void)hipFree(b->d_win); (void)hipFree(b->d_in); (void)hipFree(b->d_out);
    if (b->s) (void)hipStreamDestroy(b->s);
    delete b;
}

// fdc_fft_vcc keeps what a transform size needs — twiddle table, device buffers, a stream — in a small per-(device, n) cache: a flowgraph
// calls it item batch after item batch with the same n, and the first form (four hipMalloc, a table rebuilt and uploaded, hipDeviceSynchronize,
// four hipFree per call) stalled every other stream of the device each time.  Buffers grow to the largest batch seen; at most kFftPlans sizes
// stay cached (the least recently used one goes).
}  // extern "C"
namespace {
struct FftPlan {
    int dev = 0, n = 0;
    float2 *d_in = nullptr, *d_out = nullptr, *d_tmp = nullptr, *d_tw = nullptr;
    size_t cap_items = 0;
    hipStream_t s = nullptr;
    unsigned long long used = 0;
    std::mutex mu;                               // one caller at a time per plan
    void release()
    {
        (void)hipSetDevice(dev);
        if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
        (void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(d_tmp); (void)hipFree(d_tw);
    }
};
constexpr size_t kFftPlans = 8;
std::mutex g_fft_mu;
std::vector<std::shared_ptr<FftPlan>> g_fft_plans;
unsigned long long g_fft_tick = 0;

std::shared_ptr<FftPlan> fft_plan(int dev, int n)
{
    std::lock_guard<std::mutex> g(g_fft_mu);
    for (auto &q : g_fft_plans)
        if (q->dev == dev && q->n == n) { q->used = ++g_fft_tick; return q; }
    if (g_fft_plans.size() >= kFftPlans) {
        auto lru = std::min_element(g_fft_plans.begin(), g_fft_plans.end(), [](const auto &a, const auto &b) { return a->used < b->used; });
        std::shared_ptr<FftPlan> old = *lru;
        g_fft_plans.erase(lru);
        std::lock_guard<std::mutex> busy(old->mu);   // a caller still inside it finishes first
        old->release();
    }
    auto q = std::make_shared<FftPlan>();
    q->dev = dev; q->n = n; q->used = ++g_fft_tick;
    g_fft_plans.push_back(q);
    return q;
}
}  // namespace
extern "C" {

int fdc_fft_vcc(int device_id, int n, int forward, int shift, const void *in, int nitems, void *out)
{
    FDC_ENTRY("fdc_fft_vcc")
    if (!ispow2(n) || n < 2 || n > (1 << 24)) return fail(FDC_ERR_INVALID_ARGUMENT, "fft size %d must be a power of two in [2, 2^24]", n);
    if (nitems <= 0) return nitems == 0 ? 0 : fail(FDC_ERR_INVALID_ARGUMENT, "negative item count");
    if (!in || !out) return fail(FDC_ERR_INVALID_ARGUMENT, "null buffer");
    int rc = select_device(device_id); if (rc) return rc;
    std::shared_ptr<FftPlan> q = fft_plan(device_id, n);
    std::lock_guard<std::mutex> g(q->mu);
    HIPCHK(hipSetDevice(device_id));
    if (!q->s) HIPCHK(hipStreamCreateWithFlags(&q->s, hipStreamNonBlocking));
    if (!q->d_tw) {
        const std::vector<float2> tw = make_twiddles(n);
        HIPCHK(hipMalloc(&q->d_tw, sizeof(float2) * (size_t)n));
        HIPCHK(hipMemcpy(q->d_tw, tw.data(), sizeof(float2) * (size_t)n, hipMemcpyHostToDevice));
    }
    const size_t nb = sizeof(float2) * (size_t)n * nitems;
    if ((size_t)nitems > q->cap_items) {
        HIPCHK(hipStreamSynchronize(q->s));
        (void)hipFree(q->d_in); (void)hipFree(q->d_out); (void)hipFree(q->d_tmp);
        q->d_in = q->d_out = q->d_tmp = nullptr; q->cap_items = 0;
        HIPCHK(hipMalloc(&q->d_in, nb));
        HIPCHK(hipMalloc(&q->d_out, nb));
        if (n > fdc::kMaxLdsFft) HIPCHK(hipMalloc(&q->d_tmp, nb));
        q->cap_items = (size_t)nitems;
    }
    HIPCHK(hipMemcpyAsync(q->d_in, in, nb, hipMemcpyHostToDevice, q->s));
    // forward+shift: halves of the output swapped; inverse+shift: halves of the input swapped
    const int in_rot = (!forward && shift) ? n / 2 : 0, out_rot = (forward && shift) ? n / 2 : 0;
    HIPCHK(fdc::launch_fft(q->d_in, (size_t)n, q->d_out, q->d_tmp, n, nitems, !forward, in_rot, out_rot, 1.0f, q->d_tw, n, q->s, nullptr));
    HIPCHK(hipMemcpyAsync(out, q->d_out, nb, hipMemcpyDeviceToHost, q->s));
    HIPCHK(hipStreamSynchronize(q->s));
    return nitems;
    FDC_ENTRY_END
}

}  // extern "C"
