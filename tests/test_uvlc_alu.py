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


def test_host_pool_contains_exceptions(tmp_path):
    """openjph_amd/csrc/ojph_pool.cpp: items run once, a throwing body is rethrown on the caller after all items ran"""
    exe = str(tmp_path / "pool_check")
    subprocess.run(["g++", "-O1", "-std=c++17", "-pthread", "-o", exe, os.path.join(ROOT, "tests", "host", "pool_check.cpp"),
                    os.path.join(ROOT, "openjph_amd", "csrc", "ojph_pool.cpp")], check=True)
    env = dict(os.environ, OJPHGPU_T2_THREADS="4")
    out = subprocess.run([exe], check=True, stdout=subprocess.PIPE, env=env, timeout=120).stdout.decode()
    assert out.startswith("OK threads=4")
