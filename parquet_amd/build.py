"""Builds the native libraries in-tree (parquet_amd/lib/*.so) for gfx950.

hipcc cross-compiles without a GPU, so this also runs in a CPU-only container.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib")

DECODE_SOURCES = ["pqd_kernels.hip", "pqd_levels.hip", "pqd_levels_4k.hip", "pqd_encode.hip", "pqd_api.cpp", "pqd_file.cpp", "pqd_multi.cpp"]
DECODE_DEPS = DECODE_SOURCES + ["pqd_device.h", "pqd_rle.h", "thrift_compact.h", os.path.join("..", "..", "include", "pqdecode.h")]
SYNTH_SOURCES = ["pqd_writer.cpp"]
SYNTH_DEPS = SYNTH_SOURCES + ["thrift_compact.h"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in deps)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)


def build_all(force=False):
    os.makedirs(LIB, exist_ok=True)
    decode = os.path.join(LIB, "libpqdecode.so")
    if force or _stale(decode, DECODE_DEPS):
        # one object per source, stale ones compiled side by side (objects live in lib/obj, git-ignored like the .so)
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        objdir = os.path.join(LIB, "obj")
        os.makedirs(objdir, exist_ok=True)
        headers = [d for d in DECODE_DEPS if d not in DECODE_SOURCES]
        jobs, objs = [], []
        for src in DECODE_SOURCES:
            obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
            objs.append(obj)
            deps = [src] + headers + (["pqd_levels.hip"] if src == "pqd_levels_4k.hip" else [])
            if force or _stale(obj, deps):
                cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + os.environ.get("PQD_HIPCC_FLAGS", "").split() + ["-c", src, "-o", obj]
                print("+", " ".join(cmd), flush=True)
                jobs.append((cmd, subprocess.Popen(cmd, cwd=CSRC)))
        for cmd, job in jobs:
            if job.wait() != 0:
                raise subprocess.CalledProcessError(job.returncode, cmd)
        _run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", decode] + objs + ["-ldl", "-lpthread"])
    synth = os.path.join(LIB, "libpqsynth.so")
    if force or _stale(synth, SYNTH_DEPS):
        _run([os.environ.get("CXX", "g++"), "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", synth] + SYNTH_SOURCES)
    # the C++ mirror of the generated reader (include/pqreader.hpp) in a small program that the GPU tests run
    root = os.path.dirname(HERE)
    demo_src = os.path.join(root, "tests", "cpp", "read_rows.cpp")
    demo = os.path.join(LIB, "pq_read_rows")
    demo_deps = [demo_src, os.path.join(root, "include", "pqreader.hpp"), os.path.join(root, "include", "pqdecode.h"), decode]
    if os.path.exists(demo_src) and (force or not os.path.exists(demo) or any(os.path.getmtime(d) > os.path.getmtime(demo) for d in demo_deps)):
        _run([os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(root, "include"), demo_src,
              "-o", demo, "-L", LIB, "-lpqdecode", "-Wl,-rpath,$ORIGIN"])
    return decode, synth


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
