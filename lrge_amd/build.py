"""Build liblrge_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "_lib")
LIB_PATH = os.path.join(LIB_DIR, "liblrge_hip.so")
HOST_LIB_PATH = os.path.join(LIB_DIR, "liblrge_host.so")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               # comput_sc / per_read_estimate / quantiles are f32 with one rounding per operation
               "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def _sources():
    out = []
    for root, _, files in os.walk(CSRC):
        out += [os.path.join(root, f) for f in files]
    out.append(os.path.join(os.path.dirname(_HERE), "include", "lrge_hip.h"))
    out.append(os.path.join(os.path.dirname(_HERE), "include", "lrge_rand.hpp"))
    out.append(os.path.join(os.path.dirname(_HERE), "include", "lrge_io.hpp"))
    out.append(os.path.join(os.path.dirname(_HERE), "include", "lrge_cram.hpp"))
    return out


def build_lib(force=False, verbose=False):
    os.makedirs(LIB_DIR, exist_ok=True)
    if not force and os.path.exists(LIB_PATH) and os.path.getmtime(LIB_PATH) >= _newest(_sources()):
        return LIB_PATH
    cmd = ["hipcc"] + HIPCC_FLAGS + ["-o", LIB_PATH, os.path.join(CSRC, "lrge_hip.hip"), "-lz", "-ldl"]      # (zlib: the host-side readers of include/lrge_io.hpp)
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


CLI_PATH = os.path.join(LIB_DIR, "lrge-hip")


def build_cli(force=False):
    """The C++ host mirror (include/lrge_hip.hpp) + lrge-compatible driver, linked against liblrge_hip.so."""
    src = os.path.join(os.path.dirname(_HERE), "tools", "lrge_hip_cli.cpp")
    inc = os.path.join(os.path.dirname(_HERE), "include")
    hdrs = [os.path.join(inc, h) for h in ("lrge_hip.hpp", "lrge_hip.h", "lrge_rand.hpp", "lrge_io.hpp", "lrge_cram.hpp")]
    if not force and os.path.exists(CLI_PATH) and os.path.getmtime(CLI_PATH) >= max([os.path.getmtime(src), os.path.getmtime(LIB_PATH)] +
                                                                                    [os.path.getmtime(h) for h in hdrs]):
        return CLI_PATH
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-o", CLI_PATH, src, "-L" + LIB_DIR, "-llrge_hip", "-lz", "-ldl",
                           "-Wl,-rpath,$ORIGIN"])
    return CLI_PATH


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
