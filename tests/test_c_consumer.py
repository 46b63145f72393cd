"""A real C caller of the C ABI: tests/c_consumer.c is compiled with gcc against include/mdeflate.h and
linked with libmdeflate.so.  Without a GPU it checks that the library loads, that every call fails loudly (exit 77);
on an MI355X it runs the round trips, the streaming protocol and the two partial encoder entry points."""
import os
import subprocess

import pytest

from decompress_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp):
    build.build()
    exe = os.path.join(tmp, "c_consumer")
    so_dir = os.path.join(ROOT, "decompress_amd")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_consumer.c"), "-L", so_dir, "-l:libmdeflate.so",
                           "-Wl,-rpath," + so_dir, "-o", exe])
    return exe


def test_c_consumer_links_and_fails_loudly_without_gpu(tmp_path):
    exe = _build(str(tmp_path))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode in (0, 77), r.stdout + r.stderr
    if r.returncode == 77:
        assert "no gfx950 device" in r.stdout


@pytest.mark.gpu
def test_c_consumer_on_gpu(tmp_path):
    exe = _build(str(tmp_path))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "c_consumer: ok" in r.stdout, r.stdout + r.stderr
