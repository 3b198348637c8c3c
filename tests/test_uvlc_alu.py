"""The arithmetic U-VLC decoder of step 1 (openjph_amd/csrc/ht_uvlc.h) against the look-up table it replaces
(the reference's uvlc_tbl1, ojph_block_common.cpp:294-336): every mode and every continuation of the stream."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_alu_uvlc_equals_the_table(tmp_path):
    exe = str(tmp_path / "uvlc_alu_check")
    subprocess.run(["g++", "-O1", "-std=c++17", "-o", exe, os.path.join(ROOT, "tests", "host", "uvlc_alu_check.cpp"),
                    os.path.join(ROOT, "openjph_amd", "csrc", "ht_tables.cpp")], check=True)
    out = subprocess.run([exe], check=True, stdout=subprocess.PIPE).stdout.decode()
    assert out.startswith("OK 262144")
