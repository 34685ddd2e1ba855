#!/usr/bin/env python3
"""Compile the scene kernel with hiprtc for gfx950 on a machine WITHOUT a GPU:
the same program pt_jit.hip builds (default scene), headers read from the tree.
Catches hiprtc-only compile errors before a GPU run.  usage: tools/jit_compile_check.py [NS] [SPEC] [torch]
("torch": with the compiler bundled with PyTorch, which a process that imported torch ends up using)"""
import ctypes as C, os, struct, sys, time
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pathtracer_amd", "csrc")
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 4
SPEC = int(sys.argv[2]) if len(sys.argv) > 2 else 0
inf = float("inf")
objs = [[0, 0, -1, -100, 0, 0, -1, inf, 0, 0, 100, 0], [0, 1, 0, -20, 0, 1, 0, inf, 0, -20, 0, 0],
        [-1, 0, 0, -20, -1, 0, 0, inf, 20, 0, 0, 0], [0, 0, 1, -1, 0, 0, 1, 400, 0, 0, -1, 0]]
objs += [[0, 0, 0, 0, 0, 0, 0, inf, 0, 0, 0, 0]] * (NS - 4)
bits = lambda f: struct.unpack("<I", struct.pack("<f", f))[0]
geom = ",".join("{" + ",".join(f"PT_JIT_F(0x{bits(x):08x}u)" for x in o) + "}" for o in objs)
src = (f"#define PT_JIT 1\n#define PT_JIT_NS {NS}\n#define PT_JIT_SPEC {SPEC}\n"
       f"#define PT_JIT_F(x) __builtin_bit_cast(float, x)\n#define PT_JIT_GEOM {{{geom}}}\n#include \"pt_static_kernel.hpp\"\n")
names = ["pt_static_kernel.hpp", "pt_internal.hpp", "pt_device.hpp"]
hdrs = [open(os.path.join(root, n), "rb").read() for n in names]
if "torch" in sys.argv:
    import torch
    tl = os.path.join(os.path.dirname(torch.__file__), "lib")
    C.CDLL(os.path.join(tl, "libamd_comgr.so"), mode=C.RTLD_GLOBAL)
    rtc = C.CDLL(os.path.join(tl, "libhiprtc.so"))
else:
    rtc = C.CDLL("libhiprtc.so")
prog = C.c_void_p()
arr = lambda xs: (C.c_char_p * len(xs))(*xs)
assert rtc.hiprtcCreateProgram(C.byref(prog), src.encode(), b"pt_scene_kernel.hip", 3, arr(hdrs), arr([n.encode() for n in names])) == 0
opts = [b"--offload-arch=gfx950", b"-O3", b"-std=c++17", b"-fno-slp-vectorize"]
t0 = time.time()
rc = rtc.hiprtcCompileProgram(prog, len(opts), arr(opts))
n = C.c_size_t()
rtc.hiprtcGetProgramLogSize(prog, C.byref(n))
log = C.create_string_buffer(n.value + 1)
rtc.hiprtcGetProgramLog(prog, log)
print(f"rc={rc} in {time.time() - t0:.1f} s")
print(log.value.decode(errors="replace")[:6000])
if rc == 0:
    rtc.hiprtcGetCodeSize(prog, C.byref(n))
    print("code bytes:", n.value)
    code = C.create_string_buffer(n.value)
    rtc.hiprtcGetCode(prog, code)
    open("/tmp/pt_scene_kernel.co", "wb").write(code.raw)
    os.system("/opt/rocm/lib/llvm/bin/llvm-objdump -d /tmp/pt_scene_kernel.co | grep -c v_rcp_f32 | sed 's/^/v_rcp_f32 in the code object: /'")
    os.system("/opt/rocm/lib/llvm/bin/llvm-readelf --notes /tmp/pt_scene_kernel.co | grep -E 'vgpr_count|sgpr_count|spill|private_segment_fixed'")
sys.exit(rc)
