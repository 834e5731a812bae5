"""CPU tests (-m "not gpu") of the plan selector (csrc/fdc_api.hip: classify_plan; csrc/fdc_plan_cost.hpp) through fdc_pipeline_plan_preview,
which runs fdc_pipeline_create's validation and classification without a device: which channels become banks (one block-kernel launch each),
copies, or the remainder / the spectrum path.  The GPU side of the same question — is the choice the FASTEST form — is
tests/test_plan_choice_gpu.py; here: the structure of every plan the selector makes is valid, whatever it is fed."""
import os
import re

import numpy as np
import pytest

import gr_fdc_amd as G
from test_plan_choice_gpu import plans_for_the_choice_test, bank, two_width_plan

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 65536


def has_block_kernel(n, R, l, r):
    if R not in (2, 4):
        return False
    if l == 256:
        return n in (16384, 32768, 65536)
    if l in (512, 1024):      # k_blk512<P> / k_blk1024<P>
        return n in (16384, 32768, 65536) and r in (0, l // 2)
    if l in (128, 64):
        return n in (16384, 32768, 65536) and r % (l // 4) == 0
    return False


def check_structure(n, R, plan, flags=0):
    path, text, asg = G.plan_preview(n, R, plan, flags=flags)
    assert len(asg) == len(plan) and ("path %d" % path) in text
    banks = {}
    for c, a in enumerate(asg):
        if a >= 0:
            banks.setdefault(a, []).append(c)
    assert sorted(banks) == list(range(len(banks))) and len(banks) <= 4
    for k, ids in banks.items():
        f0, l0, p0, s0 = plan[ids[0]]
        slots = set()
        for c in ids:
            f, l, p, s = plan[c]
            assert (l, f % l, p, s) == (l0, f0 % l0, p0, s0), "bank %d mixes widths / offsets / windows: %s" % (k, text)
            assert f // l not in slots, "bank %d holds a slot twice" % k
            slots.add(f // l)
        if path in (3, 4):
            assert has_block_kernel(n, R, l0, f0 % l0), (k, text)
    for c, a in enumerate(asg):
        if a <= -2:
            c0 = -2 - a
            assert asg[c0] >= 0 and plan[c0] == plan[c], "a copy must name a bank channel with the same slice and window"
    rem = [c for c, a in enumerate(asg) if a == -1]
    if path == 3:
        assert not rem and banks
    elif path == 4:
        assert rem and banks and n in (16384, 32768, 65536)
    elif path == 2:
        assert not rem and list(banks) == [0] and all(plan[c][0] % plan[c][1] == 0 for c in banks[0]) and not any(a <= -2 for a in asg)
    else:
        assert len(rem) == len(plan)
    if path == 5:     # the one-launch form of N = 4096 (fdc_fused4096.hip): every width has a row form, the rows of a pair of blocks fit eight waves and the two tiles
        n = {l: sum(1 for c in plan if c[1] == l) for l in (1024, 512, 256, 128, 64, 32, 16)}
        assert sum(n.values()) == len(plan) and not flags & (G.FDC_PIPE_NO_POLY | G.FDC_PIPE_NO_FUSED | G.FDC_PIPE_FORCE_GENERIC)
        assert 1056 * n[1024] + 513 * n[512] + 272 * n[256] + 136 * n[128] + 68 * n[64] + 34 * n[32] + 17 * n[16] <= 4352, text

        def fits(T):        # the rows of T blocks on 4 T waves: two rows of 1024 bins, four of 512, eight of anything narrower per wave (one width per wave)
            w = -(-T * n[1024] // 2) + -(-T * n[512] // 4) + sum(-(-T * n[l] // 8) for l in (128, 64, 32, 16))
            return w <= 4 * T and T * n[256] <= 8 * (4 * T - w)
        one = not (n[1024] or n[512]) and fits(1)        # one block per workgroup (four workgroups on a unit) where no row is wide; else a pair of blocks
        assert ("one block per workgroup" in text) == one and (one or fits(2)), text
    if flags & G.FDC_PIPE_NO_POLY:
        assert path in (0, 1)
    if flags & G.FDC_PIPE_NO_BLOCK:
        assert path in (0, 1, 2, 5)
    return path, text, asg


def test_the_seventeen_plans_of_the_choice_test_have_the_expected_paths():
    want = {"configs[1] bank": 3, "offset bank r=37": 3, "two tilings (2x oversampled)": 3, "three tilings": 3, "bank + 4 others": 4, "bank + 32 others": 4,
            "512-bin bank": 3, "128-bin bank": 3, "64-bin bank": 3, "1024-bin bank": 3, "four 1024-bin channels": 3,
            "512-bin bank on and half off the grid": 3, "mixed 128/256/512 (bench --mixed)": 1, "sparse: 8 channels": 1,
            "two widths: 256-bin + 512-bin banks": 3, "three widths: 256 + 128 + 1024": 1, "half a 256-bin bank": 3}
    plans = plans_for_the_choice_test()
    assert set(want) == set(plans)
    for name, plan in plans.items():
        path, text, _ = check_structure(N, 2, plan)
        assert path == want[name], (name, text)
        for flags in (G.FDC_PIPE_NO_POLY, G.FDC_PIPE_NO_BLOCK, G.FDC_PIPE_WIDE_UNIFORM):
            check_structure(N, 2, plan, flags)
            check_structure(N, 4, plan, flags)


def test_the_one_launch_form_at_n_4096():
    """N = 4096 (the reference's example flowgraph, BASELINE configs[0]): plans of 256- / 512- / 1024-bin channels run as ONE launch (path 5) whenever their
    rows fit four waves and the tile; anything else keeps the spectrum path (or the two-launch form of a uniform 256-bin bank under FDC_PIPE_NO_FUSED)."""
    example = [(100, 256, 0.8, 1.0), (700, 512, 0.8, 1.0), (1500, 1024, 0.8, 1.0), (3000, 512, 0.8, 1.0)]     # examples/FDC_example.grc: l = 256 / 512 / 1024 / 512
    for R in (2, 4, 8):
        path, text, asg = check_structure(4096, R, example)
        assert path == 5 and "k_f4096" in text and "two blocks per workgroup, rows on 3 waves" in text and asg == [-1] * 4, text
    assert check_structure(4096, 2, example, G.FDC_PIPE_NO_FUSED)[0] == 0 and check_structure(4096, 2, example, G.FDC_PIPE_NO_POLY)[0] == 0
    assert check_structure(4096, 2, example, G.FDC_PIPE_FORCE_GENERIC)[0] == 0 and check_structure(4096, 2, example, G.FDC_PIPE_NO_BLOCK)[0] == 5
    # a full band of 256-bin channels: one block per workgroup, four waves of four rows; on its grid it was the two-launch form before
    full = bank(256, range(16))
    path, text, _ = check_structure(4096, 2, full)
    assert path == 5 and "one block per workgroup, rows on 4 waves" in text, text
    assert check_structure(4096, 2, full, G.FDC_PIPE_NO_FUSED)[0] == 2
    assert check_structure(4096, 2, bank(256, range(4)))[0] == 5 and "1 wave)" in check_structure(4096, 2, bank(256, range(4)))[1]
    # nine 128-bin + nine 64-bin channels take the four waves of one block; with a 256-bin channel more it is a pair of blocks on seven of eight waves
    assert "one block per workgroup, rows on 4 waves" in check_structure(4096, 2, bank(128, range(9)) + bank(64, range(40, 49)))[1]
    assert "two blocks per workgroup, rows on 7 waves" in check_structure(4096, 2, bank(128, range(9)) + bank(64, range(40, 49)) + bank(256, [15]))[1]
    # ONE 256-bin channel: the two launches measured 9 % faster; FDC_PIPE_WIDE_UNIFORM (every form without a spectrum in memory, whatever the rule says) keeps it
    assert check_structure(4096, 2, bank(256, [3]))[0] == 2 and check_structure(4096, 2, [(1234, 256, 0.8, 1.0)])[0] == 0
    assert check_structure(4096, 2, [(1234, 256, 0.8, 1.0)], G.FDC_PIPE_WIDE_UNIFORM)[0] == 5 and check_structure(4096, 2, [(1234, 512, 0.8, 1.0)])[0] == 5
    # widths without a row form, too many bins (channels that overlap), other block lengths: not this form
    assert check_structure(4096, 2, example + [(2000, 128, 0.8, 1.0)])[0] == 5 and "4 waves" in check_structure(4096, 2, example + [(2000, 128, 0.8, 1.0)])[1]
    assert check_structure(4096, 2, example + [(2000, 32, 0.8, 1.0), (2100, 16, 0.8, 1.0)])[0] == 5 and check_structure(4096, 2, example + [(2000, 8, 0.8, 1.0)])[0] == 0
    # narrow channels: eight rows of 128 or 64 bins per wave — up to 32 of them (a full band of 128-bin channels; half a band of 64-bin ones)
    assert check_structure(4096, 4, bank(128, range(32)))[0] == 5 and check_structure(4096, 2, bank(64, range(0, 64, 2)))[0] == 5
    assert check_structure(4096, 2, bank(64, range(33)))[0] in (0, 2) and check_structure(4096, 2, bank(64, range(4)))[0] in (0, 2)       # 33 rows x 2 blocks; under 512 bins
    assert check_structure(4096, 2, example + [(0, 2048, 0.8, 1.0)])[0] == 0
    assert check_structure(4096, 2, [(c * 200, 1024, 0.8, 1.0) for c in range(5)])[0] == 0          # 5120 bins of rows
    assert check_structure(4096, 2, [(c * 230, 256, 0.8, 1.0) for c in range(16)])[0] == 5           # overlapping slices are fine while the rows fit
    assert check_structure(4096, 2, [(c * 200, 256, 0.8, 1.0) for c in range(17)])[0] == 0
    assert check_structure(8192, 2, example)[0] == 0
    # every mixture of up to 4096 bins of rows fits
    rng = np.random.default_rng(5)
    for _ in range(200):
        plan, left = [], 4096
        while left >= 256:
            l = int(rng.choice([w for w in (256, 512, 1024) if w <= left]))
            plan.append((int(rng.integers(0, 4096 - l + 1)), l, 0.8, 1.0))
            left -= l
            if rng.random() < 0.15:
                break
        assert check_structure(4096, int(rng.choice([2, 4])), plan)[0] == (5 if sum(c[1] for c in plan) >= 512 else 0), plan


def test_the_cost_rule_at_its_thresholds():
    # one bank of 256-bin channels beats the spectrum path from ONE channel on (0.163 against 0.20 + ...); 1024-bin channels from three
    assert G.plan_preview(N, 2, bank(256, [5]))[0] == 3
    assert G.plan_preview(N, 2, bank(1024, [3, 9]))[0] == 1 and G.plan_preview(N, 2, bank(1024, [3, 9, 20]))[0] == 3
    assert G.plan_preview(N, 2, bank(1024, [3, 9]), flags=G.FDC_PIPE_WIDE_UNIFORM)[0] == 3
    # two widths: both banks stay; make one of them tiny and it goes back to a remainder (a split plan) or the plan to the spectrum path
    assert G.plan_preview(N, 2, two_width_plan())[0] == 3
    # a bank of ONE 512-bin channel beside a big bank: its launch (0.184) is still cheaper than a forward transform for it alone (0.20): two launches ...
    path, text, asg = G.plan_preview(N, 2, bank(256, range(200)) + bank(512, [100]))
    assert path == 3 and asg[-1] == 1, text
    # ... but once a remainder exists anyway (its forward transform is paid for), the tiny bank is cheaper as part of it
    path, text, asg = G.plan_preview(N, 2, bank(256, range(200)) + bank(512, [100]) + [(12345, 128, 0.7, 0.9)])
    assert path == 4 and asg[-2] == -1 and asg[-1] == -1, text
    # five tilings of 200 channels: at most four launches, the fifth goes to the remainder or everything to the spectrum path
    many = [(256 * c + r, 256, 0.88, 1.0) for r in (0, 32, 64, 128, 192) for c in range(200)]
    path, text, asg = check_structure(N, 2, many)
    assert path in (1, 4) and len({a for a in asg if a >= 0}) <= 4
    # other block lengths: banks of 256-bin channels only, no remainder
    assert G.plan_preview(16384, 2, bank(256, range(64)))[0] == 3 and G.plan_preview(32768, 4, bank(256, range(100), r=77))[0] == 3
    assert G.plan_preview(16384, 2, bank(256, range(60)) + [(15001, 128, 0.7, 0.9)])[0] == 4     # (round 5: the forward variant of the block kernel, hence split plans, at this N too)
    # round 5: every width's block kernel at N = 16384 and 32768 too (k_blk512<P>, k_blk1024<P>, k_blknar<.., P>), banks of different widths as launches
    for l in (64, 128, 512, 1024):
        for n in (16384, 32768):
            path, text, _ = G.plan_preview(n, 2, bank(l, range(n // l)))
            assert path == 3 and ("k_blk%s" % ("nar" if l < 256 else str(l))) in text, text
            assert G.plan_preview(n, 2, bank(l, range(n // l - 1), r=l // 2))[0] == 3
    assert G.plan_preview(32768, 4, bank(512, range(64)))[0] == 3 and G.plan_preview(32768, 4, bank(1024, range(32)))[0] == 3
    assert G.plan_preview(16384, 4, bank(512, range(32)))[0] == 3 and G.plan_preview(16384, 4, bank(1024, range(16)))[0] == 3     # (half a stage-2 trip comes back from the scratch)
    assert G.plan_preview(16384, 4, bank(128, range(128)))[0] == 3 and G.plan_preview(16384, 4, bank(64, range(255), r=16))[0] == 3
    path, text, asg = G.plan_preview(32768, 2, bank(256, range(64)) + bank(512, range(32, 64)))
    assert path == 3 and "two launches" in text and set(asg) == {0, 1}, text
    assert G.plan_preview(262144, 2, bank(256, range(1024)))[0] == 2 and G.plan_preview(262144, 2, bank(256, range(1000), r=1))[0] == 0
    assert G.plan_preview(32768, 2, bank(256, range(100)) + [(777, 64, 0.7, 0.9)], flags=G.FDC_PIPE_NO_BLOCK)[0] == 0
    assert G.plan_preview(65536, 8, bank(256, range(256)))[0] == 2          # relinvovl 8: no block kernel, the two-launch form
    # the same slice twice: a copy, not a second launch
    path, text, asg = G.plan_preview(N, 2, bank(512, range(128)) + [(512 * 9, 512, 0.88, 1.0)])
    assert path == 3 and asg[-1] == -2 - 9 and "1 copies" in text


def test_every_plan_the_selector_makes_is_well_formed():
    """Seeded sweep: random mixtures of banks (all widths, legal and illegal offsets, two windows), duplicates and stray channels at both overlaps and
    three block lengths, under every flag."""
    rng = np.random.default_rng(2025)
    seen = set()
    for case in range(400):
        n = int(rng.choice([65536, 65536, 65536, 32768, 16384, 4096, 262144]))
        R = int(rng.choice([2, 2, 4, 8]))
        plan = []
        if case % 7 == 0:                                  # ONE bank on its grid: the two-launch form where no block kernel applies
            l = int(rng.choice([128, 256, 256, 512]))
            if l <= n // 16:
                sl = rng.permutation(n // l)[:int(rng.integers(1, n // l + 1))]
                check_structure(n, R, bank(l, sl), int(rng.choice([0, G.FDC_PIPE_NO_BLOCK, G.FDC_PIPE_WIDE_UNIFORM | G.FDC_PIPE_NO_BLOCK])))
                seen.add(G.plan_preview(n, R, bank(l, sl), flags=G.FDC_PIPE_WIDE_UNIFORM | G.FDC_PIPE_NO_BLOCK)[0])
        for _ in range(int(rng.integers(1, 6))):
            l = int(rng.choice([32, 64, 128, 256, 256, 512, 1024, 2048]))
            if l > n // 4:
                continue
            r = int(rng.choice([0, 0, l // 2, l // 4, 3 * l // 4, int(rng.integers(0, l))]))
            win = [(0.88, 1.0), (0.7, 0.9)][int(rng.integers(0, 2))]
            nslot = n // l - 1
            for c in rng.permutation(nslot)[:int(rng.integers(1, nslot + 1))]:
                plan.append((l * int(c) + r, l) + win)
        for _ in range(int(rng.integers(0, 4))):
            if plan:
                plan.append(plan[int(rng.integers(0, len(plan)))])
        for _ in range(int(rng.integers(0, 4))):
            l = int(2 ** rng.integers(4, 12))
            if l <= n:
                plan.append((int(rng.integers(0, n - l + 1)), l, 0.6, 0.8))
        if not plan:
            continue
        plan = [plan[int(i)] for i in rng.permutation(len(plan))]
        flags = int(rng.choice([0, 0, 0, G.FDC_PIPE_WIDE_UNIFORM, G.FDC_PIPE_NO_BLOCK, G.FDC_PIPE_NO_POLY, G.FDC_PIPE_WIDE_UNIFORM | G.FDC_PIPE_NO_BLOCK]))
        path, _text, _asg = check_structure(n, R, plan, flags)
        seen.add(path)
    assert seen == {0, 1, 2, 3, 4, 5}, seen


def test_argument_errors_are_those_of_create():
    with pytest.raises(ValueError):
        G.plan_preview(N, 2, [(0, 300, 0.88, 1.0)])
    with pytest.raises(ValueError):
        G.plan_preview(N, 2, [(N - 100, 256, 0.88, 1.0)])
    with pytest.raises(ValueError):
        G.plan_preview(N, 2, [(0, 256, 0.9, 0.5)])
    with pytest.raises(ValueError):
        G.plan_preview(N, 3, [(0, 256, 0.88, 1.0)])


def test_every_constant_of_the_cost_table_names_a_file_that_exists():
    txt = open(os.path.join(ROOT, "gr-fdc_amd", "csrc", "fdc_plan_cost.hpp")).read()
    files = set(re.findall(r"profiles/r0\d/[A-Za-z0-9_.]+\.(?:json|txt|md)", txt))
    assert len(files) >= 8, files
    for f in files:
        assert os.path.exists(os.path.join(ROOT, f)), "fdc_plan_cost.hpp cites %s, which is not in the repository" % f
