"""C++ facade (include/supersonic_amd/supersonic.h) over the C ABI: the reference's own
builder API (supersonic/supersonic.h) driven from C++ exactly as its guide tests do."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "facade_test.cc")
OUT = os.path.join(ROOT, "tests", "cpp", "_build", "facade_test")
LIBDIR = os.path.join(ROOT, "supersonic_amd", "lib")


@pytest.fixture(scope="module")
def facade_bin():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    deps = [SRC, os.path.join(ROOT, "include", "ssgpu.h"), os.path.join(ROOT, "include", "supersonic_amd", "supersonic.h")]
    if not os.path.exists(OUT) or any(os.path.getmtime(d) > os.path.getmtime(OUT) for d in deps):
        # pytest-xdist may hand this module's two tests to different workers: build under a private name and move the
        # finished binary into place (a worker must never execute a half-written file)
        tmp = "%s.%d.tmp" % (OUT, os.getpid())
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), SRC, "-o", tmp,
                               "-L" + LIBDIR, "-lssgpu", "-Wl,-rpath," + LIBDIR])
        os.replace(tmp, OUT)
    return OUT


def _run(binary, mode):
    p = subprocess.run([binary, mode], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, text=True)
    assert p.returncode == 0, p.stdout
    assert "PASSED" in p.stdout


def test_facade_bind(facade_bin):
    _run(facade_bin, "bind")


@pytest.mark.gpu
def test_facade_run(facade_bin):
    _run(facade_bin, "run")
