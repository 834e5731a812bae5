// Cost model of the plan selector (fdc_api.hip: classify_plan).  ONE table: every constant is a measured time on one MI355X in
// milliseconds per 1024 blocks at N = 65536, R = 2, with the file under profiles/ it was read from.  The selector compares sums of
// these, nothing else; tests/test_plan_choice_gpu.py times the chosen form against the forced alternatives and fails when a choice is
// more than 10 % off the best, i.e. when a kernel has moved and this table has not.
#pragma once
#include <algorithm>

namespace fdc {
namespace cost {

// One launch of a width's block kernel: stage 1 transforms the whole block whatever the number of channels, so the time does not
// depend on how many slots of the bank are used.
//   256: k_blk256      0.3251 ms / 2048 blocks   profiles/r04/bench_default.json            (r05: 0.3315-0.3342, profiles/r05/ab_default.txt)
//   512: k_blk512      0.3660-0.3670 / 2048      profiles/r05/ab_w512.txt (conflict-free LDS layouts; r04: 0.374-0.386)
//  1024: k_blk1024     0.4170-0.4181 / 2048      profiles/r05/ab_w1024.txt (r04: 0.442-0.447)
//   128: k_blknar<2>   0.3513-0.3551 / 2048      profiles/r04/bench_w128.json
//    64: k_blknar<4>   0.3793-0.3855 / 2048      profiles/r04/bench_w64.json
inline double bank_launch(int width)
{
    switch (width) {
    case 256: return 0.163;
    case 512: return 0.184;
    case 1024: return 0.209;
    case 128: return 0.177;
    case 64: return 0.19;
    default: return 1e9;          // no block kernel of this width
    }
}

// The spectrum path for channels that read `band` of the 65536 bins (sum of their widths / 65536; may exceed 1 for overlapping slices):
// forward transform by the block kernel 0.20 (nothing written) ... 0.25 (every 64-bin group written), channel kernels 0.19 per 65536 bins read.
// (What bounds the forward kernel — round 6, profiles/r06/pmc_summary_fwd.txt: the memory system on its OWN traffic, 1368 MB per 1024 blocks at
// 5.5 TB/s, 39 % of it the scratch trip of the half of T that does not fit the registers; VALU 28 % of the issue slots.  With nothing written it
// still moves the 537 MB of scratch and waits for it: 0.19-0.20.  Not "bound by its arithmetic", as this line said until round 5.)
//   forward  0.477 ms / 2048 blocks, full band    profiles/r04/NOTES.md section 4 (bench_extra4_spectrum_path.json: 0.8246 = 0.477 + 0.364)
//            0.377 ms / 2048, 4 channels read     profiles/r04/bench_extra4.json (split plan: bank 0.321 + forward 0.377 + channels 0.043)
//   channels 0.364 ms / 2048 for 1.03 of the band profiles/r04/bench_extra4_spectrum_path.json; 0.43 for the mixed plan (bench_mixed.json)
constexpr double kForwardEmpty = 0.20, kForwardPerBand = 0.05, kChannelsPerBand = 0.19;
inline double spectrum_path(double band)
{
    return band <= 0.0 ? 0.0 : kForwardEmpty + kForwardPerBand * std::min(1.0, band) + kChannelsPerBand * band;
}

// A bank of few channels costs a whole launch; once there is a remainder anyway (its forward transform is paid for) a small bank is
// cheaper as part of it: its channels cost kChannelsPerBand + kForwardPerBand per unit of band there (profiles/r04/NOTES.md section 4).
constexpr double kRemainderPerBand = kChannelsPerBand + kForwardPerBand;

constexpr int kMaxBanks = 4;      // launches per launch group the selector will line up (tables per bank are a few hundred KB)

}  // namespace cost
}  // namespace fdc
