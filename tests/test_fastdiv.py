"""The scaling by 1/sqrt(head_dim) must be the reference's true fp32 division bit for bit (pyramidkv_utils.py:317;
SURVEY.md §7 'multiply-by-reciprocal is NOT bit-equal').  The kernel's 2-FMA form is proven equal here by brute force
over every fp32 input inside its guard."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_fma_division_is_ieee_division(tmp_path):
    exe = str(tmp_path / "fastdiv")
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-fopenmp", "-o", exe,
                           os.path.join(ROOT, "tests", "fastdiv_host.c"), "-lm"])
    out = subprocess.check_output([exe], text=True)
    assert out.startswith("OK"), out
