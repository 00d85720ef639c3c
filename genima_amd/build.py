"""Build libgenima_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

    python -m genima_amd.build [--force]

hipcc cross-compiles without a GPU; the built ``genima_amd/libgenima_hip.so`` is git-ignored but travels to the
GPU box with the source snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libgenima_hip.so")
SOURCES = ["api.hip", "gemm.hip", "gemm_pp.hip", "gemm_ppp.hip", "gemm_s3.hip", "gemm_tn.hip", "attention.hip", "attention_stream.hip", "attention_pwg.hip", "attention_bwd.hip", "attention_fp8.hip", "norm.hip", "elementwise.hip", "backward.hip", "augment.hip", "fp8.hip", "comm.hip", "act_train.hip", "pack.hip", "tblock.hip", "conv_gn.hip"]
# -amdgpu-mfma-vgpr-form: gfx950's register file is unified, so keep MFMA accumulators in VGPRs -- the softmax / epilogue VALU
# then works on them in place instead of through v_accvgpr_read/write copies (400 of them per attention tile otherwise).
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wall", "-Wno-unused-function"]
_VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]  # measured: helps attention (+30 %), costs the GEMM 5 %
EXTRA_FLAGS = {"attention.hip": _VGPR_FORM, "attention_stream.hip": _VGPR_FORM, "attention_pwg.hip": _VGPR_FORM, "attention_bwd.hip": _VGPR_FORM, "attention_fp8.hip": _VGPR_FORM}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def source_sha16() -> str:
    """Hash of every file under csrc/ plus the compile flags: names the library's CODE independently of where / when it was built (a
    rebuilt .so differs byte-wise from box to box; bench.py matches committed PMC traffic files to the running code through this)."""
    import hashlib

    h = hashlib.sha256(" ".join(FLAGS).encode())
    for n in sorted(os.listdir(CSRC)):
        if n.endswith((".hip", ".h")):
            with open(os.path.join(CSRC, n), "rb") as f:
                h.update(n.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in ("common.h", "gemm_common.h", "attention_common.h", "gn_bridge.h")] + [os.path.join(HERE, "..", "include", "genima_hip.h")]
    # the objects are only as good as the flags they were built with: a probe build (GN_HIPCC_EXTRA=-DGN_PP_ABLATIONS ..., some of which give wrong
    # results on purpose) must never be reused by the next normal build, nor the other way round -- the flag set is stamped beside the objects
    stamp = os.path.join(objdir, ".flags")
    flagset = " ".join(FLAGS + [f"{k}:{' '.join(v)}" for k, v in sorted(EXTRA_FLAGS.items())] + ["extra:" + os.environ.get("GN_HIPCC_EXTRA", "").strip()])
    try:
        with open(stamp) as f:
            have = f.read()
    except OSError:
        have = None
    if have != flagset:
        if have is not None or os.environ.get("GN_HIPCC_EXTRA", "").strip():
            force = True
        elif os.path.exists(OUT) and verbose:
            print("[genima_amd.build] no flag stamp yet: trusting the existing objects once", flush=True)

    def compile_one(src):
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + os.environ.get("GN_HIPCC_EXTRA", "").split() + ["-c", s, "-o", o]  # GN_HIPCC_EXTRA: probe builds (-DGN_PWG_ABLATIONS)
            if verbose:
                print("[genima_amd.build]", " ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return o

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _stale(OUT, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-ldl"]
        if verbose:
            print("[genima_amd.build]", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    with open(stamp, "w") as f:
        f.write(flagset)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
