"""The weights-in-registers kernels issue their weight loads as inline asm with hand-counted waits, and the FP6 form relies on the
register coalescer turning "two asm load outputs concatenated into a 6-register tuple" into sub-register assignments: a compiler that
copied, spilled or re-used one of those registers while its load is in flight would read garbage (cdna_hip_programming.md 5.7).
tools/check_wreg_asm.py compiles gemm_wreg.hip for gfx950 and inspects the ISA of every product kernel for exactly that; it needs no
GPU, so it runs here - the GPU parity tests would catch the wrong results, this catches the cause, on any toolchain update."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_compiler_copy_or_spill_of_an_in_flight_ring_register():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_wreg_asm.py"), "gemm_wreg.hip"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("MB=")]
    assert len(lines) >= 30, r.stdout[-2000:]                     # 15 tilings x (int8, nibble) + the FP6 forms
    fp6 = [l for l in lines if " Q=3 " in l]
    assert len(fp6) == 10 and all("suspicious 0" in l for l in fp6), fp6      # six tilings + the paired gate / up form (ABL=90) of four of them
    pair = [l for l in lines if l.rstrip().split(":")[0].endswith("ABL=90")]
    assert len(pair) == 9 and all("suspicious 0" in l for l in pair), pair   # int8: five tilings, FP6: four
