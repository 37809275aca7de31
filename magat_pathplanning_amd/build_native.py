"""Builds libmagat_hip.so (gfx950) in-tree with hipcc.  `python -m magat_pathplanning_amd.build_native`."""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "lib", "libmagat_hip.so")
SOURCES = ["conv_gemm_f32.hip", "conv_gemm_bf16x6.hip", "conv_gemm_f16x3_pair.hip", "conv_gemm_f16x3_duo.hip", "gat_f32.hip", "gat_list_f32.hip", "gat_csr_f32.hip", "encoder_f32.hip", "layer1_fused.hip", "sim_frontend.hip", "profile.hip"]
HEADERS = ["magat_common.h", os.path.join("..", "..", "include", "magat_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [hipcc] + [f for f in FLAGS if f] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
