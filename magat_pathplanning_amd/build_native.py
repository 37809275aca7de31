"""Builds libmagat_hip.so (gfx950) in-tree with hipcc.  `python -m magat_pathplanning_amd.build_native [--force] [--debug]`.

Every source is compiled to its own object (only the stale ones, in parallel), then linked.  --debug builds
lib/libmagat_hip_debug.so with -DMAGAT_DEBUG_HOOKS (phase timestamps / phase skipping of the graph kernel, used by
tools/gat_phase_probe.py through MAGAT_LIB_PATH); the release library carries no instrumentation."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "lib", "libmagat_hip.so")
LIB_DEBUG = os.path.join(PKG, "lib", "libmagat_hip_debug.so")
SOURCES = ["conv_gemm_f32.hip", "conv_gemm_bf16x6.hip", "block_fused.hip", "block_lat.hip", "gat_f32.hip", "gat_mfma.hip", "gat_small.hip", "gat_mid.hip", "gat_csr_f32.hip", "gat_csr_fused.hip",
           "encoder_f32.hip", "layer1_fused.hip", "stem8.hip", "conv_train.hip", "sim_frontend.hip", "profile.hip", "options.hip"]
HEADERS = ["magat_common.h", "block_walk.h", os.path.join("..", "..", "include", "magat_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-value", "-Wno-inline-asm"]
# block_fused.hip: MFMA results in architectural registers wherever they fit (the chain kernel has all 512 registers of a SIMD to one wave:
# the epilogues then read the accumulators as plain operands instead of through v_accvgpr_read; chain kernel -0.9 % same-box).  The graph
# kernel is slower and loses its spill-free allocation under the same flag, so it is per source.
SOURCE_FLAGS = {"block_fused.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _newer(path, deps):
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False, debug=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    lib = LIB_DEBUG if debug else LIB
    objdir = os.path.join(PKG, "lib", "obj_debug" if debug else "obj")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    # MAGAT_EXTRA_FLAGS (kernel experiments, tools/whatif_*.sh, tools/ab_flags.sh) reaches the DEBUG library only: the release
    # library libmagat_hip.so is always built from the sources as they are
    flags = FLAGS + (["-DMAGAT_DEBUG_HOOKS"] + os.environ.get("MAGAT_EXTRA_FLAGS", "").split() if debug else [])
    jobs, objs = [], []
    for s in _sources():
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(obj, [src] + hdrs):
            jobs.append([hipcc] + flags + SOURCE_FLAGS.get(s, []) + ["-c", src, "-o", obj])
    if not jobs and not _newer(lib, objs):
        return lib

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    # (release: local symbols stripped - the exported C ABI lives in .dynsym, kernel names in the device code objects)
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + ([] if debug else ["-Wl,-s"]) + objs + ["-o", lib])
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, debug="--debug" in sys.argv))
