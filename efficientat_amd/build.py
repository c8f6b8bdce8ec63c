"""Build recipe for libeat_hip.so (hipcc, gfx950 only, in-tree)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libeat_hip.so")
SOURCES = ["common.cpp", "mel.hip", "conv_spatial.hip", "conv_pw.hip", "train.hip", "dymn.hip", "conv_pw_bf16.hip", "conv_pw_generic.hip", "conv_pw_stream.hip", "expand_dw.hip", "irb.hip", "train_glue.hip", "dw_plane.hip", "train_fuse.hip", "stem_train.hip", "se_train.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-ffp-contract=on", "-Wall",
         "-Wno-unused-function", "-Wno-inline-asm"]


def _stale(obj, src):
    if not os.path.exists(obj):
        return True
    deps = [src, os.path.join(HERE, "..", "include", "eat_hip.h")] + [
        os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")]
    return any(os.path.getmtime(d) > os.path.getmtime(obj) for d in deps)


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link libeat_hip.so next to this file."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or _stale(obj, path):
            cmd = [hipcc, "-x", "hip", "-c", path, "-o", obj] + FLAGS
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {src}")
        if verbose and out:
            print(out.decode())
    if procs or not os.path.exists(LIB) or force:
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
