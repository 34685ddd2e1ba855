#!/usr/bin/env python3
"""build_scene_kernels.py -- compile the reference's default scene's trace kernel offline (hipcc, no GPU) and store it
next to libpt.so as a kernel shipped with the library (pt_scene_kernel_ship, include/pt.h): pt_scene_specialise then
needs neither libhiprtc nor its 0.7 s.  Run by `make scene_kernels` and by __graft_entry__.build() after libpt.so is built.
usage: build_scene_kernels.py [--arch gfx950:sramecc+:xnack-]"""
import argparse
import ctypes as C
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="gfx950:sramecc+:xnack-")      # MI355X as hipDeviceProp_t::gcnArchName names it
    ap.add_argument("--hipcc", default=os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"))
    args = ap.parse_args()
    from pathtracer_amd import _lib, scene
    lib = _lib.lib
    objs = scene.to_c_array(scene.objects)                           # Object.objects, src/Object.hs:147-148
    csrc = os.path.join(ROOT, "pathtracer_amd", "csrc")
    d = os.path.join(ROOT, "pathtracer_amd", "scene_kernels")
    if os.path.isdir(d):                                             # kernels of an earlier build of the sources
        for f in os.listdir(d):
            if f.endswith(".co"):
                os.remove(os.path.join(d, f))
    with tempfile.TemporaryDirectory() as tmp:
        src, co = os.path.join(tmp, "pt_scene_kernel.hip"), os.path.join(tmp, "pt_scene_kernel.co")
        rc = lib.pt_scene_kernel_ship(objs, len(scene.objects), args.arch.encode(), src.encode(), None)
        if rc != 0:
            sys.exit(f"pt_scene_kernel_ship(source) failed: {rc}")
        # the Makefile's code-generation flags (pt_jit.hip hands hiprtc the same)
        subprocess.check_call([args.hipcc, "--genco", f"--offload-arch={args.arch}", "-O3", "-std=c++17", "-fno-slp-vectorize",
                               "-I", csrc, "-o", co, src])
        rc = lib.pt_scene_kernel_ship(objs, len(scene.objects), args.arch.encode(), None, co.encode())
        if rc != 0:
            sys.exit(f"pt_scene_kernel_ship(store) failed: {rc}")
    print("shipped scene kernels:", sorted(f for f in os.listdir(d) if f.endswith(".co")))


if __name__ == "__main__":
    main()
