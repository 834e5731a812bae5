#!/usr/bin/env python3
"""Runs bench.py over a list of configurations (GPU box helper).  Each config: 'ENV=VAL,... -- bench args'."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for spec in sys.argv[1:]:
    envs, _, args = spec.partition("--")
    env = dict(os.environ)
    for kv in envs.split(","):
        kv = kv.strip()
        if kv:
            k, v = kv.split("=")
            env[k] = v
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline"] + args.split()
    out = subprocess.run(cmd, env=env, capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
        r = d["roofline"]
        print("%-40s path=%d chunk=%-5d %9.1f Msps  %.4f ms/step  frac=%.4f  %s" % (
            spec, d["config"]["kernel_path"], d["config"]["chunk_blocks"], d["value"], d["ms_per_step"],
            r["pipeline_frac"], {k: v for k, v in r["kernel_ms_per_step"].items()}))
    except Exception as e:  # noqa
        print(spec, "FAILED", e, out.stderr[-500:])
    sys.stdout.flush()
