"""ISA lint of the compiled gfx950 kernels: every read of an MFMA result is far enough behind the MFMA on every path.

hipcc inserts these wait states itself; one source layout of the score kernel made it emit 9 of the 18 on the taken side of a
branch ("a compiler hazard"), which showed up as rare, non-deterministic top-K mismatches on the GPU.  The lint
runs on the CPU (cross-compile to assembly, no GPU needed).
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mfma_result_reads_wait_long_enough():
    files = [os.path.join(ROOT, "recboard_amd", "csrc", f) for f in ("score.hip", "enc_fwd.hip", "enc_bwd.hip", "enc_step.hip", "enc_wgrad.hip", "gemm.hip", "enc_tile.hip", "enc_tail.hip")]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "lint_mfma_hazard.py")] + files, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "findings: 0" in r.stdout
    # (round 6) the tile kernels carry no packed-fp32 instruction -- and the Makefile really passes the flag that removes them
    assert [ln for ln in r.stdout.splitlines() if ln.startswith("enc_tile.hip") and ln.rstrip().endswith("0 packed-fp32 instructions")], r.stdout
    mk = open(os.path.join(ROOT, "recboard_amd", "csrc", "Makefile")).read()
    assert "-packed-fp32-ops" in mk and "build/enc_tile.o: enc_tile.hip" in mk and "$(CXXFLAGS) $(TILE_FLAGS) -c $< -o $@" in mk
    for f in files:                                  # the lint saw the MFMA kernels it is meant to check
        line = [ln for ln in r.stdout.splitlines() if ln.startswith(os.path.basename(f)) and ln.rstrip().endswith("mfma checked")][0]
        assert int(line.split()[1]) > 0, line
