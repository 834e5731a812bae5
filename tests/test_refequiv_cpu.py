"""oracle/ref_equiv.c — bench.py's "reference-equivalent" CPU leg (FFTW3f + VOLK found with dlopen) — cannot run on any box of
this pool: neither library is installed.  So that the leg is not dead code, it is executed here once against two test doubles
(tests/shims: a plain DFT behind FFTW's five entry points, VOLK's two dispatcher pointers), in a child process whose
LD_LIBRARY_PATH names them, and its channel output is compared with the oracle.  Nothing is timed or reported from this."""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_equivalent_leg_runs_against_test_doubles(tmp_path, oracle):
    sh = os.path.join(ROOT, "tests", "shims")
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", str(tmp_path / "libfftw3f.so.3"), os.path.join(sh, "fftw3f_shim.c"), "-lm"])
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", str(tmp_path / "libvolk.so"), os.path.join(sh, "volk_shim.c")])
    assert oracle.refequiv_probe() is not None or os.environ.get("LD_LIBRARY_PATH")     # this process: the real libraries are absent
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        import numpy as np
        import oracle as O
        assert O.refequiv_probe() is None, O.refequiv_probe()
        N, R = 256, 4
        plan = [(33, 32, 0.7, 0.9), (100, 64, 0.88, 1.0), (7, 32, 0.5, 0.8)]          # odd f: the phase counter rotates
        rng = np.random.default_rng(5)
        H = N - N // R
        x = (rng.standard_normal(9 * H) + 1j * rng.standard_normal(9 * H)).astype(np.complex64)
        msps, passes, out0 = O.refequiv_run(N, R, 1, plan, x, 4, 0.05)
        ref, _ = O.channelizer(N, R, 1, plan, x)
        err = float(np.abs(out0 - ref[0]).max() / np.abs(ref[0]).max())
        assert passes >= 1 and msps > 0 and err <= 1e-5, (passes, msps, err)
        print("ok", passes, err)
    """ % os.path.join(ROOT, "oracle"))
    env = dict(os.environ, LD_LIBRARY_PATH=str(tmp_path) + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
    out = subprocess.check_output([sys.executable, "-c", code], env=env, text=True)
    assert out.startswith("ok")
