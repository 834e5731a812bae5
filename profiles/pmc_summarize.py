"""Summarises rocprofv3 --pmc csv output (one row per dispatch and counter) for the fdc:: kernels:
per kernel name, the mean counter value per dispatch."""
import collections
import csv
import glob
import os
import sys

root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = row.get("Kernel_Name", "")
            if "fdc::" not in name:
                continue
            short = name.split("(")[0].replace("void ", "")
            acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("   %-28s mean/dispatch %.6g   (n=%d)" % (c, sum(v) / len(v), len(v)))

# HBM traffic per dispatch for bench.py's roofline.traffic: FETCH_SIZE and WRITE_SIZE are in KB; on gfx950
# FETCH_SIZE reports half of the bytes of a wide coalesced streaming read (MI355X_MICROARCH.md §HBM) -> doubled.
# One entry per (config, kernel label); `kernels` lists every fdc:: kernel of the run with its own bytes, so that the
# traffic of a multi-kernel step (configs 3-5) can be added up.  PMC_CONFIG / PMC_BLOCKS / PMC_BLOCKLEN describe the run.
import json
names = {"fdc::k_blk256": "block_kernel(colFFT+window+IFFT+slotFFT)", "fdc::k_p1": "poly_stage1(colFFT+window+IFFT)", "fdc::k_p2": "poly_stage2(slotFFT)",
         "fdc::k_p2k": "poly_stage2(slotFFT)", "fdc::k_a256": "fft_pass_a", "fdc::k_b256": "fft_pass_b", "fdc::k_c256": "channels",
         "fdc::k_fft4096": "fft_pass_b", "fdc::k_channels": "channels", "fdc::k_c512": "channels", "fdc::k_c1024": "channels",
         "fdc::k_f4096": "fused4096(FFT+cut+window+IFFTs, spectrum in LDS)"}
cfg = int(os.environ.get("PMC_CONFIG", "2"))
tag = os.environ.get("PMC_TAG", "")
out, allk = {}, {}
for k in acc:
    if "FETCH_SIZE" not in acc[k] or "WRITE_SIZE" not in acc[k]:
        continue
    f = sum(acc[k]["FETCH_SIZE"]) / len(acc[k]["FETCH_SIZE"]) * 1024
    w = sum(acc[k]["WRITE_SIZE"]) / len(acc[k]["WRITE_SIZE"]) * 1024
    allk[k] = {"fetch_size_raw_bytes": f, "write_size_bytes": w, "hbm_bytes_per_launch": 2 * f + w, "dispatches": len(acc[k]["FETCH_SIZE"])}
    base = k.split("<")[0]
    targs = [t.strip() for t in k[k.index("<") + 1:k.rindex(">")].split(",")] if "<" in k else []
    if base == "fdc::k_blk256" and len(targs) >= 3 and targs[2] == "true":                      # <NT, OFF, FWD = true, R4>
        names[base + "_fwd"] = "block_fft(forward, one kernel)"
        base += "_fwd"
    if base in names:
        key = "cfg%d%s/%s" % (cfg, tag, names[base])
        if key in out:                 # bench.py's "channels" is one timing slot for all channel kernels of a launch group: their bytes add up
            e = out[key]
            e["kernel"] += " + " + k
            for fld, val in (("fetch_size_raw_bytes", f), ("write_size_bytes", w), ("hbm_bytes_per_launch", 2 * f + w)):
                e[fld] += val
        else:
            out[key] = {
                "kernel": k, "config": cfg, "fetch_size_raw_bytes": f, "write_size_bytes": w, "hbm_bytes_per_launch": 2 * f + w,
                "blocks_per_launch": float(os.environ.get("PMC_BLOCKS", "1024")), "blocklen": int(os.environ.get("PMC_BLOCKLEN", "65536"))}
out["cfg%d%s/all_kernels" % (cfg, tag)] = allk
with open(os.path.join(root, "pmc_traffic.json"), "w") as fh:
    json.dump(out, fh, indent=1)
