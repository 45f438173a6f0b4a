"""Builds the REFERENCE'S OWN guide programs (test/guide/*.cc), unchanged and read in place from the reference checkout,
against this repository's headers and library:

    g++ -std=c++14 -I<repo>/include -I<repo>/tests/cpp/minigtest <reference>/test/guide/<name>.cc -lssgpu

Nothing of the reference is copied: the source stays where it lies, the binary goes to tests/cpp/_build/ (git-ignored, like
every built artefact; it travels to the GPU box with the other built files, where `tests/test_cpp_facade.py` runs it).
`gtest/gtest.h` resolves to tests/cpp/minigtest -- this repository's own small harness with googletest's macro surface.
Without a reference checkout (the GPU box) this does nothing."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REFERENCE = os.environ.get("SUPERSONIC_REFERENCE", "/root/reference")
OUT_DIR = os.path.join(ROOT, "tests", "cpp", "_build")
LIBDIR = os.path.join(ROOT, "supersonic_amd", "lib")
GUIDES = ("primer", "group_sort")


def source(name):
    return os.path.join(REFERENCE, "test", "guide", name + ".cc")


def binary(name):
    return os.path.join(OUT_DIR, "ref_guide_" + name)


def build(name, force=False):
    """Returns the binary's path, or None when there is no reference checkout to read the source from."""
    src, out = source(name), binary(name)
    if not os.path.exists(src):
        return None
    os.makedirs(OUT_DIR, exist_ok=True)
    deps = [src, os.path.join(ROOT, "include", "ssgpu.h"), os.path.join(ROOT, "include", "supersonic_amd", "supersonic.h"),
            os.path.join(ROOT, "tests", "cpp", "minigtest", "gtest", "gtest.h")]
    if force or not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        tmp = "%s.%d.tmp" % (out, os.getpid())
        subprocess.check_call(["g++", "-std=c++14", "-O1", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "cpp", "minigtest"),
                               src, "-o", tmp, "-L" + LIBDIR, "-lssgpu", "-Wl,-rpath," + LIBDIR])
        os.replace(tmp, out)
    return out


def expected():
    """Guide binaries the build container said it built (tests/cpp/_build/EXPECTED, written by mark_expected)."""
    try:
        with open(os.path.join(OUT_DIR, "EXPECTED")) as f:
            return [ln.strip() for ln in f if ln.strip()]
    except OSError:
        return []


def mark_expected(names):
    os.makedirs(OUT_DIR, exist_ok=True)
    with open(os.path.join(OUT_DIR, "EXPECTED"), "w") as f:
        f.write("".join(n + "\n" for n in names))


if __name__ == "__main__":
    for g in GUIDES:
        print(g, "->", build(g, force="--force" in sys.argv))
