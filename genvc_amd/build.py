"""Build libgenvc_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m genvc_amd.build [--force]

Objects go to genvc_amd/csrc/build/, the library to genvc_amd/lib/libgenvc_hip.so (in-tree, so the
gpurun snapshot carries it to the GPU box).  A source is recompiled when it or any header is newer
than its object.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "lib", "libgenvc_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _newest_header():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "genvc_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, force):
    obj = os.path.join(OBJ, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), _newest_header()):
        return obj, False
    cmd = [HIPCC] + FLAGS + (["-x", "hip"] if src.endswith(".hip") else []) + ["-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-4000:]}")
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    with ThreadPoolExecutor(max_workers=4) as ex:
        res = list(ex.map(lambda s: _compile(s, force), sources()))
    objs = [o for o, _ in res]
    if any(c for _, c in res) or not os.path.exists(LIB):
        tmp = f"{LIB}.{os.getpid()}.tmp"       # linked aside and renamed: another rank waiting for LIB never maps a half-written file
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
        os.replace(tmp, LIB)
        if verbose:
            print(f"built {LIB} ({sum(c for _, c in res)} objects recompiled)")
    elif verbose:
        print(f"{LIB} is up to date")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
