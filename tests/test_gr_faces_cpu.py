"""CPU tests (-m "not gpu") of the C++ gr::FDC faces' GNU Radio side: the -DFDC_HAVE_GNURADIO branch of fdc_blocks.cc goes
through a compiler (against declaration-only headers, both smart-pointer shapes), and the stock-scheduler stand-in sizes
buffers and batches work() the way GNU Radio's runtime does with and without the block's requests."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLOCKS = os.path.join(ROOT, "gr-fdc_amd", "csrc", "gr_blocks")
DECL = os.path.join(ROOT, "tests", "shims", "gnuradio_decl")


@pytest.mark.parametrize("sptr", ["boost", "std"])
def test_gnuradio_branch_of_the_block_faces_compiles(sptr):
    """the FDC_HAVE_GNURADIO code (pmt PDUs, rebind_sptr over gr::basic_block_sptr, the real sync_block base,
    start()/stop() pinning through detail()) had never been through a compiler.  -fsyntax-only against declarations of the GNU
    Radio 3.7/3.8 (boost::shared_ptr) and >= 3.9 (std::shared_ptr) API."""
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-DFDC_HAVE_GNURADIO", "-I" + DECL, "-I" + BLOCKS,
           os.path.join(BLOCKS, "fdc_blocks.cc")]
    if sptr == "std":
        cmd.insert(1, "-DFDC_DECL_STD_SPTR")
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_declaration_headers_define_nothing():
    """The shim directory holds declarations only: no function bodies that could stand in for GNU Radio."""
    for dp, _dn, fn in os.walk(DECL):
        for f in fn:
            if f.endswith((".h", ".hpp")):
                txt = open(os.path.join(dp, f)).read()
                body = txt.split("#pragma once", 1)[1]
                assert "return" not in body, f


@pytest.fixture(scope="module")
def stock_check(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("stock") / "stock_scheduler_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-pthread", "-I" + BLOCKS, "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "stock_scheduler_check.cc")])
    return exe


def _run(exe, item, mult, total):
    r = subprocess.run([exe, str(item), str(mult), str(total)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    return json.loads(r.stdout)


def test_stock_scheduler_offers_a_couple_of_items_without_requests(stock_check):
    """configs[1]'s input item is 262144 bytes: a 64-KiB buffer rule gives 2 x (multiple 1 + history 1) = 4 items, half a buffer per
    call = 2 items per work() — what VERDICT r04 missing #1 describes."""
    d = _run(stock_check, 262144, 0, 40)
    assert d["in_buffer_items"] == 4 and d["max_call"] <= 2 and d["items"] == 40 and d["intact"]
    d = _run(stock_check, 1024, 0, 1000)         # small items: 64 KiB / 1 KiB = 64 items, half per call
    assert d["in_buffer_items"] == 64 and d["max_call"] == 32 and d["items"] == 1000 and d["intact"]


def test_stock_scheduler_honours_output_multiple_and_min_output_buffer(stock_check):
    """What fdc_pipeline_vcc asks for (fdc_blocks.cc: apply_scheduler_hints): output_multiple k, min_output_buffer 2k.  Upstream's
    buffer then holds 2 (k + history) items, ours 2k, every call is exactly k items; the tail below k is not processed."""
    d = _run(stock_check, 262144, 64, 300)
    assert d["in_buffer_items"] == 130 and d["out_buffer_items"] == 128
    assert d["min_call"] == 64 and d["max_call"] == 64 and d["items"] == 256 and d["intact"]
    d = _run(stock_check, 24, 5, 1003)           # item size that does not divide the page: granularity rounding
    assert d["in_buffer_items"] * 24 % 4096 == 0 and d["min_call"] == 5 and d["items"] == 1000 and d["intact"]
