"""Build libhfmi.so (hand-written HIP for gfx950) in-tree with hipcc.

hipcc cross-compiles without a GPU, so this runs in the authoring container; the
resulting .so travels to the GPU box with the repository snapshot."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(ROOT, "include")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libhfmi.so")
SOURCES = ["hfmi_api.hip", "hfmi_gemm.hip", "hfmi_gemm_nn.hip", "hfmi_misc.hip", "hfmi_small.hip", "hfmi_skinny.hip", "hfmi_comm.hip", "hfmi_eig_large.hip", "hfmi_eig_blocked.hip", "hfmi_eig_dc.hip", "hfmi_chol.hip", "hfmi_cheb.hip", "hfmi_xfer.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + INCLUDE, "-I" + CSRC] + os.environ.get("HFMI_EXTRA_HIPCC_FLAGS", "").split()


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (need the ROCm toolchain to build libhfmi.so)")
    return exe


def _newer(target, deps):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def source_tag():
    """Identity of the kernel sources this library is built from (first 12 hex digits of a SHA-256 over csrc/ and the
    public header).  ``hfmi_build_tag()`` returns it; profiles/pmc_traffic.json records it, and bench.py only uses PMC
    traffic figures that were measured on the same build."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(CSRC, name), "rb").read())
    h.update(open(os.path.join(INCLUDE, "hfmi.h"), "rb").read())
    return h.hexdigest()[:12]


def build(force=False, verbose=True):
    """Compile every translation unit for gfx950 and link hippyflow_amd/libhfmi.so."""
    hipcc = _hipcc()
    os.makedirs(OBJDIR, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in sorted(os.listdir(CSRC)) if h.endswith(".h")] + [os.path.join(INCLUDE, "hfmi.h")]
    tag = source_tag()
    tag_file = os.path.join(OBJDIR, "source_tag.txt")
    tag_changed = not os.path.exists(tag_file) or open(tag_file).read().strip() != tag
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src.replace(".hip", ".o"))
        objs.append(o)
        carries_tag = src == "hfmi_api.hip"
        if force or not _newer(o, [s] + headers) or (carries_tag and tag_changed):
            jobs.append([hipcc] + FLAGS + (['-DHFMI_BUILD_TAG="%s"' % tag] if carries_tag else []) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print("[hfmi build]", " ".join(os.path.relpath(c, ROOT) if os.path.isabs(c) else c for c in cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + r.stdout)
        return r.stdout

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(run, jobs))
    open(tag_file, "w").write(tag + "\n")
    if jobs or force or not _newer(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl", "-lrt", "-lpthread"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
