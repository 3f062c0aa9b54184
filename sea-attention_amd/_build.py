"""Build libsea_hip.so (gfx950) in-tree with hipcc.  No torch extension machinery: the library is a
plain C-ABI shared object (include/sea_hip.h) loaded through ctypes.

One object per source (compiled in parallel, content-addressed: rebuilt only when the source, a header or the
flags changed), one link."""
import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
OBJ_DIR = os.path.join(PKG_DIR, "build")
LIB_PATH = os.path.join(PKG_DIR, "libsea_hip.so")
SOURCES = ["sea_topk.hip", "sea_attn.hip", "sea_attn_tile.hip", "sea_attn_bwd.hip", "sea_csr_ops.hip", "sea_predictor.hip",
           "sea_performer.hip", "sea_conv.hip", "sea_mlp.hip", "sea_decode.hip"]
HEADERS = [os.path.join(CSRC, "sea_common.hpp"), os.path.join(CSRC, "sea_attn.hpp"), os.path.join(CSRC, "sea_tail.hpp"),
           os.path.join(CSRC, "sea_convfrag.hpp"),
           os.path.join(os.path.dirname(PKG_DIR), "include", "sea_hip.h")]
# -fno-slp-vectorize: the SLP pass packs adjacent scalar f32 adds/muls into v_pk_* pairs plus the v_mov traffic
# to form the register pairs; on gfx950 that is a net loss in these VALU-issue-bound kernels (A/B: -1.5 % step)
COMMON_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]
# per-file flags.  The MFMA tile attention keeps its accumulators in VGPRs (gfx950's register file is unified): with the
# default AGPR form every rescale of the output accumulators costs a v_accvgpr_read + write around the multiply, and
# the VGPR + AGPR split costs a wave of occupancy (171 -> 166 registers, 2 -> 3 waves per SIMD).
FILE_FLAGS = {"sea_attn_tile.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
              "sea_attn_bwd.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def source_hash():
    """sha256 over everything the library is made from (sources, headers, flags): survives copies of the tree
    (a gpurun snapshot keeps no useful mtimes), unlike a timestamp comparison."""
    h = hashlib.sha256()
    for f in [os.path.join(CSRC, s) for s in _sources()] + HEADERS:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    h.update(repr((COMMON_FLAGS, sorted(FILE_FLAGS.items()))).encode())
    return h.hexdigest()


def sources_present():
    """The tree holds everything the library is built from (a tree that ships only a prebuilt .so does not)."""
    return bool(_sources()) and all(os.path.exists(os.path.join(CSRC, s)) for s in SOURCES) and all(os.path.exists(h) for h in HEADERS)


def is_stale():
    """True when libsea_hip.so is missing, or the sources are here and it was built from other ones (content hash).
    A prebuilt library WITHOUT its sources or headers is current by definition: there is nothing to rebuild it from."""
    if not os.path.exists(LIB_PATH):
        return True
    if not sources_present():
        return False
    if not os.path.exists(LIB_PATH + ".sha256"):
        return True
    return open(LIB_PATH + ".sha256").read().strip() != source_hash()


def ensure_built(verbose=False):
    """Build when stale, ONCE per tree however many processes ask at the same moment (the ranks of a multi-GPU job all
    import the package together): an exclusive file lock serialises them, the first one in builds, the others find the
    stamp current when their turn comes."""
    if not is_stale():
        return LIB_PATH
    import fcntl
    with open(LIB_PATH + ".lock", "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            if is_stale():
                build_library(verbose=verbose)
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)
    return LIB_PATH


def build_library(force=False, extra_flags=(), out=None, verbose=False):
    """hipcc --offload-arch=gfx950 -O3 -c csrc/X.hip (per file) ; hipcc -shared *.o -o libsea_hip.so"""
    out = out or LIB_PATH
    if not force and out == LIB_PATH and not is_stale():
        return out
    cc = _hipcc()
    custom = bool(extra_flags) or out != LIB_PATH          # A/B builds keep the product's objects
    os.makedirs(os.path.join(OBJ_DIR, "custom") if custom else OBJ_DIR, exist_ok=True)

    hdr_bytes = b"".join(open(h, "rb").read() for h in HEADERS)

    def compile_one(src):
        flags = [*COMMON_FLAGS, *FILE_FLAGS.get(src, []), *extra_flags]
        key = hashlib.sha256(open(os.path.join(CSRC, src), "rb").read() + hdr_bytes + repr(flags).encode()).hexdigest()[:16]
        stem = os.path.splitext(src)[0]
        # content-addressed: a reused object IS this source + flags.  A/B builds (extra flags / another output) keep their
        # objects apart, in a directory that neither git nor the gpurun snapshot carries
        obj = os.path.join(os.path.join(OBJ_DIR, "custom") if custom else OBJ_DIR, f"{stem}.{key}.o")
        if os.path.exists(obj) and not force:
            return obj
        tmp_o = obj + f".tmp.{os.getpid()}"
        cmd = [cc, *flags, "-c", os.path.join(CSRC, src), "-o", tmp_o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        os.replace(tmp_o, obj)
        for f in os.listdir(OBJ_DIR):                       # drop this source's older objects
            if f.startswith(stem + ".") and f.endswith(".o") and f != os.path.basename(obj) and not custom:
                try:
                    os.remove(os.path.join(OBJ_DIR, f))
                except OSError:
                    pass
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, _sources()))
    tmp = out + f".tmp.{os.getpid()}"
    cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", tmp]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(tmp, out)                                   # atomic: concurrent ranks building at once cannot tear the file
    if out == LIB_PATH and not extra_flags:
        with open(LIB_PATH + ".sha256", "w") as f:
            f.write(source_hash())
    return out


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
