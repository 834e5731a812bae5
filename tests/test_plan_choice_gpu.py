"""GPU tests (-m gpu) of the plan selector (csrc/fdc_api.hip: classify_plan; constants: csrc/fdc_plan_cost.hpp).

(a) banks of DIFFERENT widths in one plan: the reference derives l per channel
    (python/FrequencyDomainChannelizer.py:323-327) and its own example is mixed (examples/FDC_example.grc); a plan that is a
    256-bin bank plus a 512-bin bank is two block-kernel launches, not the spectrum path — parity against the oracle and,
    on every sample, against the spectrum path.
(b) the choice itself (VERDICT r04 next #3b): for plans spanning every row of DESIGN.md section 3a the chosen form is timed
    against every forced alternative (FDC_PIPE_NO_POLY: spectrum in memory; FDC_PIPE_WIDE_UNIFORM: every bank on its block
    kernel whatever the cost rule says; FDC_PIPE_NO_BLOCK: no block kernels) over 512 blocks and must be within 10 % of the best:
    nine paths and a table of measured constants — the next kernel change would otherwise mis-route plans silently."""
import os
import sys

import numpy as np
import pytest

import gr_fdc_amd as G

pytestmark = pytest.mark.gpu
TOL = 1e-5
N, H = 65536, 32768


def noise(n, seed):
    rng = np.random.default_rng(seed)
    x = np.empty(n, np.complex64)
    x.real = rng.standard_normal(n, dtype=np.float32)
    x.imag = rng.standard_normal(n, dtype=np.float32)
    return x


def bank(L, slots, r=0, win=(0.88, 1.0)):
    return [(L * int(c) + r, L, win[0], win[1]) for c in slots]


def two_width_plan():
    # lower half of the band: 128 channels of 256 bins; upper half: 64 channels of 512 bins — interleaved in plan order
    a, b = bank(256, range(128)), bank(512, range(64, 128))
    plan = []
    for i in range(128):
        plan.append(a[i])
        if i < 64:
            plan.append(b[i])
    return plan


def rel(a, b):
    return max(float(np.linalg.norm(a - b) / np.linalg.norm(b)), float(np.abs(a - b).max() / np.abs(b).max()))


@pytest.mark.parametrize("R", [2, 4])
def test_banks_of_two_widths_are_two_launches(oracle, R):
    nb = 100
    plan = two_width_plan()
    x = noise(nb * (N - N // R), 77 + R)
    p = G.Pipeline(N, R, plan, windowtype=1, max_blocks=nb, min_block_launch=1)
    assert p.path() == 3 and "two launches" in p.describe() and "k_blk256" in p.describe() and "k_blk512" in p.describe(), p.describe()
    got = p.work(x)
    q = G.Pipeline(N, R, plan, windowtype=1, max_blocks=nb, flags=G.FDC_PIPE_NO_POLY)
    assert q.path() == 1
    ref = q.work(x)
    for c in range(len(plan)):
        assert rel(got[c], ref[c]) <= TOL, (c, plan[c])
    # the oracle on channels of both widths at both ends of their banks
    ids = [0, 1, 2, 3, 126, 127, 189, 190, 191]
    oref, _ = oracle.channelizer(N, R, 1, [plan[i] for i in ids], x, nthreads=8)
    for k, i in enumerate(ids):
        assert rel(got[i], oref[k]) <= TOL, (i, plan[i])
    # two calls, ragged, against the one call: bit for bit (state: history + block counter)
    p2 = G.Pipeline(N, R, plan, windowtype=1, max_blocks=nb, min_block_launch=1)
    Hh = N - N // R
    g1 = p2.work(x[:37 * Hh]); g2 = p2.work(x[37 * Hh:])
    for c in (0, 1, 100, 191):
        assert (np.concatenate([g1[c], g2[c]]).view(np.uint32) == got[c].view(np.uint32)).all()


def test_three_widths_odd_offsets_and_a_remainder(oracle):
    """256-bin tiling at r = 37 (odd: the window phase alternates with the block index), a bank of 128-bin channels a quarter of a channel
    off its grid, a bank of 1024-bin channels, and five channels that fit no bank: three launches + a remainder on a partial spectrum."""
    R, nb = 2, 64
    rng = np.random.default_rng(5)
    plan = bank(256, range(0, 96), r=37) + bank(128, range(200, 300), r=32) + bank(1024, range(40, 60)) + \
        [(int(f) | 1, l, 0.7, 0.9) for f, l in ((61001, 512), (62001, 128), (63001, 2048), (60001, 64), (59001, 32))]
    order = rng.permutation(len(plan))
    plan = [plan[i] for i in order]
    x = noise(nb * H, 78)
    # (the cost rule itself would send this plan to the spectrum path: three launches for 0.9 of the band; FDC_PIPE_WIDE_UNIFORM keeps the banks)
    assert G.Pipeline(N, R, plan, windowtype=1, max_blocks=nb).path() == 1
    p = G.Pipeline(N, R, plan, windowtype=1, max_blocks=nb, min_block_launch=1, flags=G.FDC_PIPE_WIDE_UNIFORM)
    assert p.path() == 4 and "three launches" in p.describe() and "5 other channels" in p.describe(), p.describe()
    got = p.work(x)
    ref = G.Pipeline(N, R, plan, windowtype=1, max_blocks=nb, flags=G.FDC_PIPE_NO_POLY).work(x)
    for c in range(len(plan)):
        assert rel(got[c], ref[c]) <= TOL, (c, plan[c])
    ids = list(range(0, len(plan), 17))
    oref, _ = oracle.channelizer(N, R, 1, [plan[i] for i in ids], x, nthreads=8)
    for k, i in enumerate(ids):
        assert rel(got[i], oref[k]) <= TOL, (i, plan[i])


def test_same_slice_twice_is_computed_once_and_copied(oracle):
    """A slot used twice with one window (what the parameter derivation makes of a wrapped channel) is an alias, for 256-bin banks too."""
    R, nb = 2, 24
    plan = bank(256, range(256)) + [(256 * 7, 256, 0.88, 1.0), (256 * 200, 256, 0.88, 1.0)]
    x = noise(nb * H, 79)
    p = G.Pipeline(N, R, plan, windowtype=1, max_blocks=nb, min_block_launch=1)
    assert p.path() == 3 and "2 copies" in p.describe(), p.describe()
    got = p.work(x)
    assert (got[256].view(np.uint32) == got[7].view(np.uint32)).all() and (got[257].view(np.uint32) == got[200].view(np.uint32)).all()
    oref, _ = oracle.channelizer(N, R, 1, [plan[7], plan[200]], x, nthreads=8)
    assert rel(got[256], oref[0]) <= TOL and rel(got[257], oref[1]) <= TOL


# ---- (b) is the chosen form the fastest one?
def plans_for_the_choice_test():
    rng = np.random.default_rng(11)
    odd = [(int(f) | 1, l, 0.7, 0.9) for f, l in zip(rng.integers(0, N - 2048, 32), [512, 128, 1024] * 11)]
    cfg2 = bank(256, range(256))
    P = {
        "configs[1] bank": cfg2,
        "offset bank r=37": bank(256, range(255), r=37),
        "two tilings (2x oversampled)": bank(256, range(256)) + bank(256, range(255), r=128),
        "three tilings": bank(256, range(256)) + bank(256, range(255), r=128) + bank(256, range(255), r=64),
        "bank + 4 others": cfg2 + odd[:4],
        "bank + 32 others": cfg2 + odd,
        "512-bin bank": bank(512, range(128)),
        "128-bin bank": bank(128, range(512)),
        "64-bin bank": bank(64, range(1024)),
        "1024-bin bank": bank(1024, range(64)),
        "four 1024-bin channels": bank(1024, (3, 17, 40, 61)),
        "512-bin bank on and half off the grid": bank(512, range(128)) + bank(512, range(127), r=256),
        "mixed 128/256/512 (bench --mixed)": [(256 * c, 256, 0.88, 1.0) for c in range(0, 256, 2)] + [(256 * c + 64, 128, 0.88, 1.0) for c in range(1, 256, 4)] +
                                             [(256 * c - 128, 512, 0.88, 1.0) for c in range(3, 252, 4)],
        "sparse: 8 channels": [(int(f), l, 0.7, 0.9) for f, l in zip(np.linspace(1000, 60000, 8).astype(int) | 1, [256, 512, 1024, 2048, 256, 512, 1024, 2048])],
        "two widths: 256-bin + 512-bin banks": two_width_plan(),
        "three widths: 256 + 128 + 1024": bank(256, range(0, 128)) + bank(128, range(256, 384)) + bank(1024, range(48, 64)),
        "half a 256-bin bank": bank(256, range(0, 256, 2)),
    }
    return P


class _Hip:
    """Device buffers without torch (the test process has the library's HIP runtime loaded; torch brings its own)."""
    def __init__(self):
        import ctypes as C
        self.C = C
        self.h = C.CDLL("/opt/rocm/lib/libamdhip64.so")
        self.h.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.h.hipMemsetD32.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.h.hipFree.argtypes = [C.c_void_p]

    def alloc(self, nbytes, fill=None):
        ptr = self.C.c_void_p()
        assert self.h.hipMalloc(self.C.byref(ptr), nbytes) == 0
        if fill is not None:
            assert self.h.hipMemsetD32(ptr, fill, nbytes // 4) == 0
            assert self.h.hipDeviceSynchronize() == 0
        return ptr

    def free(self, ptr):
        self.h.hipFree(ptr)


def time_the_forms(n, plan, nb, name):
    """The chosen form and every forced alternative over nb blocks: HIP events on the launch stream; fails when the choice is > 10 % off the best.
    A timing assertion inside a suite the driver runs with -x: the file is collected LAST (tests/conftest.py), and a case
    that misses its 10 % is timed once more over four times the blocks before it fails — a noisy box does not stop the run in front of a parity test."""
    try:
        _time_the_forms(n, plan, nb, name)
    except AssertionError as first:
        print("\nPLANCHOICE %s: missed at %d blocks (%s); once more over %d" % (name, nb, str(first)[:200], 4 * nb), file=sys.stderr)
        _time_the_forms(n, plan, 4 * nb, name + " [re-timed x4]")


def _time_the_forms(n, plan, nb, name):
    R = 2
    h = n - n // R
    hip = _Hip()
    ring = hip.alloc((n // R + nb * h) * 8, fill=0x3F000000)          # 0.5 + 0.5j everywhere: kernel time does not depend on the data
    results = {}
    try:
        for tag, flags in (("chosen", 0), ("spectrum", G.FDC_PIPE_NO_POLY), ("all-banks", G.FDC_PIPE_WIDE_UNIFORM), ("no-block", G.FDC_PIPE_NO_BLOCK),
                           ("no-fused", G.FDC_PIPE_NO_FUSED)):
            p = G.Pipeline(n, R, plan, windowtype=1, max_blocks=nb, flags=flags, min_block_launch=96)
            what = p.describe()
            if tag != "chosen" and what == results["chosen"][1]:
                p.close()
                continue                                    # the same plan under this flag: nothing to compare
            out = hip.alloc(p.output_samples(nb) * 8)
            for _ in range(3):
                p.process_device(ring, 0, nb, out)
            p.synchronize()
            p.enable_timing(1)                              # HIP events on the launch stream around every launch group
            reps = 7
            for _ in range(reps):
                p.process_device(ring, 0, nb, out)
            p.synchronize()
            ms = p.last_kernel_ms()
            results[tag] = (sum(ms[:3]) / reps, what)
            p.close()
            hip.free(out)
    finally:
        hip.free(ring)
    best = min(v[0] for v in results.values())
    line = "; ".join("%s %.3f ms [%s]" % (k, v[0], v[1].split("path ")[1]) for k, v in results.items())
    print("\nPLANCHOICE %s: %s" % (name, line), file=sys.stderr)
    if os.environ.get("FDC_PLANCHOICE_LOG"):            # profiles/rNN/plan_choice.txt is this file
        with open(os.environ["FDC_PLANCHOICE_LOG"], "a") as fh:
            fh.write("%s: %s\n" % (name, line))
    assert results["chosen"][0] <= 1.10 * best, line


@pytest.mark.parametrize("name", sorted(plans_for_the_choice_test()))
def test_the_chosen_form_is_within_ten_percent_of_the_best(name):
    time_the_forms(N, plans_for_the_choice_test()[name], 512, name)


def plans_at_shorter_blocks(n):
    """The same question at N = 32768 / 16384 (round 5: every width's block kernel and the block forward transform exist there: the cost rule applies)."""
    rng = np.random.default_rng(n)
    odd = [(int(f) | 1, l, 0.7, 0.9) for f, l in zip(rng.integers(0, n - 2048, 32), [512, 128, 1024] * 11)]
    full = bank(256, range(n // 256))
    return {
        "bank + 4 others": full + odd[:4],
        "bank + 32 others": full + odd,
        "two widths": bank(256, range(n // 512)) + bank(512, range(n // 1024, n // 512)),
        "mixed 128/256/512": [(256 * c, 256, 0.88, 1.0) for c in range(0, n // 256, 2)] + [(256 * c + 64, 128, 0.88, 1.0) for c in range(1, n // 256, 4)] +
                             [(256 * c - 128, 512, 0.88, 1.0) for c in range(3, n // 256 - 4, 4)],
        "two 1024-bin channels": bank(1024, (3, 9)),
        "sparse: 6 channels": [(int(f), l, 0.7, 0.9) for f, l in zip(np.linspace(1000, n - 3000, 6).astype(int) | 1, [256, 512, 1024, 2048, 256, 512])],
    }


@pytest.mark.parametrize("n", [32768, 16384])
@pytest.mark.parametrize("name", sorted(plans_at_shorter_blocks(32768)))
def test_the_chosen_form_at_shorter_blocks(n, name):
    time_the_forms(n, plans_at_shorter_blocks(n)[name], 512 * 65536 // n, "N = %d, %s" % (n, name))


def plans_at_4096():
    """N = 4096 (round 6): the one-launch form with the spectrum in LDS against what FDC_PIPE_NO_FUSED / FDC_PIPE_NO_POLY leave (two-launch uniform form, spectrum path)."""
    return {
        "example flowgraph (256 / 512 / 1024 / 512)": [(100, 256, 0.8, 1.0), (700, 512, 0.8, 1.0), (1500, 1024, 0.8, 1.0), (3001, 512, 0.8, 1.0)],
        "four 256-bin channels": [(300 + 901 * c, 256, 0.8, 1.0) for c in range(4)],
        "full band of 256-bin channels": bank(256, range(16)),
        "eight 512-bin channels": bank(512, range(8)),
        "four 1024-bin channels": bank(1024, range(4)),
        "one 256-bin channel": [(1234, 256, 0.8, 1.0)],
        "full band of 128-bin channels": bank(128, range(32)),
        "thirty-two 64-bin channels": bank(64, range(0, 64, 2)),
        "example + two 128-bin + two 64-bin channels": [(100, 256, 0.8, 1.0), (700, 512, 0.8, 1.0), (1500, 1024, 0.8, 1.0), (3001, 512, 0.8, 1.0),
                                                         (2600, 128, 0.8, 1.0), (2800, 128, 0.8, 1.0), (3600, 64, 0.8, 1.0), (3700, 64, 0.8, 1.0)],
    }


@pytest.mark.parametrize("name", sorted(plans_at_4096()))
def test_the_chosen_form_at_4096(name):
    time_the_forms(4096, plans_at_4096()[name], 8192, "N = 4096, %s" % name)
