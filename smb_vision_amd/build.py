"""Build libsmbv.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python -m smb_vision_amd.build [--force]

hipcc cross-compiles without a GPU.  Objects go to smb_vision_amd/csrc/_obj/, the library to smb_vision_amd/libsmbv.so
(git-ignored; it still travels to the GPU box with the working tree).
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libsmbv.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result", "-ffp-contract=fast"]
# The attention bodies place single-issue VALU work between MFMAs; SLP-packing adjacent f32 multiplies into v_pk_* (plus the moves
# that line the pairs up) costs more issue slots there than it saves.
# gemm.hip: LLVM's max-ILP machine scheduler for the compiler-scheduled K loops (gemm_nt_dma_kernel: LDS reads issued further ahead of their MFMAs, four
# instead of seven MFMAs behind the barrier; 104 -> 112 registers, same occupancy): -0.10 ms per MIM step in-box, same results bit for bit
# (profiles/r06_sched_strategy_ab.txt; the other files measured null or slower with it - the hand-phased attention backward +0.1 ms)
FILE_FLAGS = {"attention_bwd.hip": ["-fno-slp-vectorize"], "attention_fwd.hip": ["-fno-slp-vectorize"],
              "gemm.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the smbv HIP library cannot be built")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime() -> float:
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "smbv.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(src), _deps_mtime(), os.path.getmtime(__file__)):
        return obj
    cmd = [_hipcc(), *FLAGS, *FILE_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = sources()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), srcs))
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(o) for o in objs):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[smbv] built {LIB} from {len(srcs)} sources")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
