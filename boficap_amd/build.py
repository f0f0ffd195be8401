"""Build libboficap_hip.so (the C-ABI library of include/boficap_hip.h) for gfx950 with hipcc.

In-tree build: objects under build/, the shared library next to this file so that it travels with
the source tree to the GPU box.  hipcc cross-compiles without a GPU.

    python -m boficap_amd.build [--force]
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(ROOT, "build", "obj")
LIB = os.path.join(HERE, "libboficap_hip.so")
SOURCES = ["ln.hip", "gemm.hip", "gemm_glds.hip", "gemm_pers.hip", "attn.hip", "attn_bf16.hip", "naic.hip", "train_ops.hip", "gemm_tn.hip", "attn_bwd_mfma.hip", "repack.hip", "bound_ops.hip", "bound_loop.hip", "rowblock.hip", "engine.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize -fno-vectorize: no v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32.  Measured on MI355X (round 2, dev/exp/dbg_step*.py): a wavefront
# whose float32 FMA chains were packed by the SLP vectoriser (v_pk_fma_f32 with op_sel operands) computed wrong sums in lanes 48-63
# of one accumulator -- only while wavefronts of OTHER kernels were issuing MFMAs on the same SIMD (several decodes in flight),
# never alone and never beside copies of itself.  Without the packed forms the results are bit-stable under any concurrency.  (The
# packed forms buy nothing here anyway: the VALU work of these kernels sits beside MFMA or memory latency.)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-fno-vectorize", "-Wall", "-Wno-unused-function",
         f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}"]
if os.environ.get("BOFI_EXPERIMENTS") == "1":       # developer build: the timing-only ablation switches of engine.hip (BOFI_EXP_SKIP / BOFI_EXP_ITERS); use with --force
    FLAGS.append("-DBOFI_EXPERIMENTS")


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(ROOT, "include", "boficap_hip.h"))

    def compile_one(src: str) -> str:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or not _newer(o, [s] + headers):
            cmd = [HIPCC, *FLAGS, "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return o

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or not _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
