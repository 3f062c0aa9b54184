"""Build libsea_hip.so (gfx950) in-tree with hipcc.  No torch extension machinery: the library is a
plain C-ABI shared object (include/sea_hip.h) loaded through ctypes."""
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libsea_hip.so")
SOURCES = ["sea_topk.hip", "sea_attn.hip", "sea_csr_ops.hip", "sea_predictor.hip", "sea_performer.hip", "sea_conv.hip", "sea_mlp.hip"]
HEADERS = [os.path.join(CSRC, "sea_common.hpp"), os.path.join(CSRC, "sea_tail.hpp"), os.path.join(os.path.dirname(PKG_DIR), "include", "sea_hip.h")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_library(force=False, extra_flags=(), out=None, verbose=False):
    """hipcc --offload-arch=gfx950 -O3 -shared -fPIC csrc/*.hip -o libsea_hip.so"""
    out = out or LIB_PATH
    if not force and out == LIB_PATH and not is_stale():
        return out
    # -fno-slp-vectorize: the SLP pass packs adjacent scalar f32 adds/muls into v_pk_* pairs plus the v_mov traffic
    # to form the register pairs; on gfx950 that is a net loss in these VALU-issue-bound kernels (A/B: -1.5 % step)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize",
           "-Wall", "-Wno-unused-function", *extra_flags,
           *[os.path.join(CSRC, s) for s in SOURCES], "-o", out + f".tmp.{os.getpid()}"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    os.replace(out + f".tmp.{os.getpid()}", out)     # atomic: concurrent ranks building at once cannot tear the file
    return out


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
