"""kvc_stl_emul.h (the libstdc++ partial_sort / nth_element+sort restatement the exact-tie GPU kernel executes) is
compiled for the HOST with AddressSanitizer + UBSan and compared, index for index, with the real library on
tie-heavy inputs (plateaus like max-pooling makes, all-equal, sorted, k == n, k*64 == n boundary)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_emulation_equals_libstdcxx(tmp_path):
    exe = str(tmp_path / "stl_emul_host")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                           "-o", exe, os.path.join(ROOT, "tests", "stl_emul_host.cpp")])
    out = subprocess.check_output([exe, "240"], text=True)
    assert out.startswith("OK 240 trials"), out
