#!/usr/bin/env python3
"""gen_ubench_mix.py -- a measured ceiling for the shipped trace loop.

Reads the disassembly of the scene kernel (llvm-objdump -d of the code object
tools/jit_compile_check.py writes to /tmp/pt_scene_kernel.co), cuts out the
phase-1 loop of trace_static_kernel -- the loop BASELINE's frame spends 95 % of
its trips in -- and writes tools/ubench_mix.hip: the SAME instructions (same
opcodes, encodings, operand kinds: VGPR / SGPR / literal / inline constant,
modifiers), with the registers renamed so that NO instruction depends on an
earlier one:
  * every VGPR source is one of 24 input registers the loop never writes;
  * every VGPR destination comes from a rotating pool of 28 others;
  * compares write rotating scratch SGPR pairs (or vcc, as shipped);
  * SALU instructions run on scratch SGPRs; the exec-mask pairs
    (s_and_saveexec / s_mov exec) become plain s_and_b64 / s_mov_b64;
  * LDS reads keep their widths and offsets, from a valid address.
Three kernels: `mix_shipped` (VALU + SALU + LDS in the shipped order),
`mix_valu` (the VALU instructions alone, shipped order), `mix_spread` (the same
VALU multiset, transcendentals and second-port instructions spread evenly
between the plain ones).  The host times them with 8 waves per SIMD on every
SIMD of the chip and prints ns per trip per SIMD, to set beside the real
kernel's (segments / 64 / kernel time / 1024 SIMDs).

usage: gen_ubench_mix.py [/tmp/pt_scene_kernel.s] > tools/ubench_mix.hip
"""
import re
import sys

TRANS = ("v_rcp_", "v_sin_", "v_cos_", "v_sqrt_", "v_rsq_", "v_exp_", "v_log_")
SECOND = ("v_cmp", "v_cndmask", "v_bitop3", "v_mad_u64", "v_and_or", "v_max_i32", "v_min_", "v_max_", "v_med3",
          "v_cvt", "v_lshlrev", "v_lshl_or", "v_mul_lo", "v_mul_hi")


def loop_lines(path):
    ins = []
    for ln in open(path):
        m = re.match(r"\s+(\S.*?)\s*//\s*([0-9A-F]{12}):", ln)
        if m:
            ins.append((int(m.group(2), 16), m.group(1).strip()))
    # backward conditional branches; take the loop whose body has exactly 2 v_sin and no other branch
    best = None
    for k, (addr, text) in enumerate(ins):
        m = re.match(r"s_cbranch_scc[01] (\d+)", text)
        if not m:
            continue
        off = int(m.group(1))
        if off < 32768:
            continue
        target = addr + 4 + (off - 65536) * 4
        body = [(a, t) for a, t in ins if target <= a < addr]
        if sum(t.startswith("v_sin_") for _, t in body) == 2 and not any(t.startswith(("s_cbranch", "s_branch")) for _, t in body):
            if best is None or len(body) > len(best):
                best = body
    if best is None:
        raise SystemExit("gen_ubench_mix: the phase-1 loop was not found in the disassembly")
    return [t for _, t in best]


class Renamer:
    def __init__(self):
        self.src_v = 0          # inputs: v0..v23
        self.dst_v = 0          # outputs: v32..v59 (rotating)
        self.dst_s = 0          # scratch SGPR pairs s[40:41] .. s[54:55]

    def vsrc(self, width=1):
        r = self.src_v
        if width > 1:
            r = (r + width - 1) // width * width
        if r + width > 24:
            r = 0
        self.src_v = (r + width) % 24
        return f"v{r}" if width == 1 else f"v[{r}:{r + width - 1}]"

    def vdst(self, width=1):
        r = self.dst_v
        if width > 1:
            r = (r + width - 1) // width * width
        if r + width > 28:
            r = 0
        self.dst_v = (r + width) % 28
        return f"v{32 + r}" if width == 1 else f"v[{32 + r}:{32 + r + width - 1}]"

    def spair(self):
        r = 40 + 2 * self.dst_s
        self.dst_s = (self.dst_s + 1) % 8
        return f"s[{r}:{r + 1}]"


def width_of(tok):
    m = re.match(r"[vs]\[(\d+):(\d+)\]", tok)
    return int(m.group(2)) - int(m.group(1)) + 1 if m else 1


def rename(text, R):
    """-> (kind, renamed text); kind in valu / salu / lds / wait / skip."""
    op, _, rest = text.partition(" ")
    if op.startswith("s_waitcnt"):
        m = re.search(r"lgkmcnt\((\d+)\)", rest)
        return "wait", f"s_waitcnt lgkmcnt({m.group(1)})" if m else None
    if op == "s_nop":
        return "salu", text
    if op.startswith("ds_read"):
        dst, addr = [x.strip() for x in rest.split(",", 1)]
        off = re.search(r"offset:(\d+)", addr)
        return "lds", f"{op} {R.vdst(width_of(dst))}, v31" + (f" offset:{off.group(1)}" if off else "")
    if op.startswith("s_"):
        # SALU on scratch registers; exec juggling becomes plain 64-bit logic of the same issue cost
        if op in ("s_and_saveexec_b64",):
            return "salu", f"s_and_b64 {R.spair()}, s[56:57], s[58:59]"
        ops = [x.strip() for x in rest.split(",")]
        if ops and ops[0] == "exec":
            return "salu", f"s_mov_b64 {R.spair()}, s[56:57]"
        if op.startswith("s_cmp"):
            return "salu", f"{op} s[56:57], s[58:59]" if "64" in op else f"{op} s56, s58"
        wide = "64" in op
        d = R.spair() if wide else f"s{60 + (R.dst_s % 2)}"
        srcs = ["s[56:57]" if wide else "s56", "s[58:59]" if wide else "s58"]
        n_src = len(ops) - 1
        if op in ("s_mov_b64", "s_mov_b32"):
            lit = ops[1] if re.match(r"(0x[0-9a-f]+|-?\d+)$", ops[1]) else srcs[0]
            return "salu", f"{op} {d}, {lit}"
        if op in ("s_lshl_b32", "s_sub_i32", "s_add_i32", "s_lshr_b32"):
            last = ops[2] if re.match(r"-?\d+$", ops[2]) else "s58"
            return "salu", f"{op} {d}, s56, {last}"
        return "salu", f"{op} {d}, " + ", ".join(srcs[:n_src])
    if not op.startswith("v_"):
        return "skip", None
    # ---- VALU: destination(s) first, then sources
    mods = ""
    m = re.search(r"\s+((?:bitop3|op_sel|clamp|mul|div|neg_lo|neg_hi)\S*(?:\s+\S+:\S+)*)$", rest)
    if m:
        mods = " " + m.group(1)
        rest = rest[:m.start()]
    ops = [x.strip() for x in rest.split(",")]
    n_dst = 2 if op.startswith("v_mad_u64") else 1
    out = []
    accumulates = op.startswith(("v_fmac", "v_fmaak")) and False
    for k, tok in enumerate(ops):
        neg = tok.startswith("-")
        absv = tok.startswith("|")
        core = tok.strip("-|")
        if k < n_dst:
            if core == "vcc":
                out.append("vcc")
            elif core.startswith("s["):
                out.append(R.spair())
            else:
                out.append(R.vdst(width_of(core)))
            continue
        if core == "vcc":
            new = "vcc"
        elif re.match(r"v\[?\d", core):
            new = R.vsrc(width_of(core))
        elif core.startswith("s["):
            new = "s[56:57]"
        elif re.match(r"s\d+$", core):
            new = "s56"
        else:
            new = core                                    # literal or inline constant
        if absv:
            new = f"|{new}|"
        out.append(("-" if neg else "") + new)
    _ = accumulates
    return "valu", f"{op} " + ", ".join(out) + mods


def classify(text):
    op = text.split(" ", 1)[0]
    if op.startswith(TRANS):
        return "trans"
    if op.startswith(SECOND):
        return "second"
    return "plain"


def spread(valu):
    """The same multiset, each class spread evenly over the trip."""
    groups = {"trans": [], "second": [], "plain": []}
    for t in valu:
        groups[classify(t)].append(t)
    n = len(valu)
    slots = []
    for cls, items in groups.items():
        for k, t in enumerate(items):
            slots.append(((k + 0.5) * n / max(len(items), 1), cls, t))
    slots.sort(key=lambda x: x[0])
    return [t for _, _, t in slots]


def asm_block(lines):
    return "\n".join(f'        "{ln}\\n\\t"' for ln in lines)


def rename_keep_pattern(text, R, mod=24):
    """Like rename(), but a VGPR source keeps its identity: vN -> v(N mod 24), so an operand used twice in the
    shipped instruction is used twice here, and register parities survive (24 is even)."""
    op, _, rest = text.partition(" ")
    mods = ""
    m = re.search(r"\s+((?:bitop3|op_sel|clamp|mul|div|neg_lo|neg_hi)\S*(?:\s+\S+:\S+)*)$", rest)
    if m:
        mods = " " + m.group(1)
        rest = rest[:m.start()]
    ops = [x.strip() for x in rest.split(",")]
    n_dst = 2 if op.startswith("v_mad_u64") else 1
    out = []
    for k, tok in enumerate(ops):
        neg = tok.startswith("-")
        core = tok.strip("-|")
        if k < n_dst:
            out.append("vcc" if core == "vcc" else R.spair() if core.startswith("s[") else R.vdst(width_of(core)))
            continue
        mv = re.match(r"v(\d+)$", core)
        if core == "vcc":
            new = "vcc"
        elif mv:
            new = f"v{int(mv.group(1)) % mod}"
        elif core.startswith("s["):
            new = "s[56:57]"
        elif re.match(r"s\d+$", core):
            new = "s56"
        else:
            new = core
        out.append(("-" if neg else "") + new)
    return f"{op} " + ", ".join(out) + mods


def per_instruction(loop):
    """One kernel per VALU instruction of the loop: that instruction 32 times (destinations rotating)."""
    clob = ", ".join([f'"v{k}"' for k in range(60)] + [f'"s{k}"' for k in range(36, 64)] + ['"vcc"', '"scc"', '"memory"'])
    init = ["v_mov_b32 v31, 0"] + [f"v_mov_b32 v{k}, 0x3fc00000" for k in range(31)] + \
           ["s_mov_b64 s[56:57], -1", "s_mov_b64 s[58:59], -1"]
    valu = [t for t in loop if t.startswith("v_")]
    print("// ubench_instr.hip -- GENERATED by tools/gen_ubench_mix.py --per-instr: every VALU instruction of the shipped loop")
    print("// on its own (32 copies per trip, destinations rotating, sources as shipped: a repeated operand stays repeated).")
    print("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdlib>\n")
    names = []
    for k, text in enumerate(valu):
        R = Renamer()
        body = [rename_keep_pattern(text, R) for _ in range(32)]
        names.append((f"k{k}", text))
        print(f"__global__ __launch_bounds__(64) void k{k}(int trips, long long *clk)\n{{\n"
              f"    const long long c0 = clock64(), w0 = wall_clock64();\n    asm volatile(\n{asm_block(init)}\n"
              f'        "s_mov_b32 s62, %0\\n\\t"\n        "1:\\n\\t"\n{asm_block(body)}\n'
              f'        "s_sub_u32 s62, s62, 1\\n\\t"\n        "s_cmp_lg_u32 s62, 0\\n\\t"\n        "s_cbranch_scc1 1b\\n\\t"\n'
              f"        : : \"s\"(trips) : {clob});\n"
              f"    const long long c1 = clock64(), w1 = wall_clock64();\n"
              f"    if (threadIdx.x == 0) {{ clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }}\n}}\n")
    print("""
// shader clocks (s_memtime) and 100 MHz wall ticks (s_memrealtime) taken by every wave around its loop: the cycles are
// counted, not derived from a nominal frequency; the frequency the chip actually ran at falls out as cycles / wall time
static void clocks(const long long *h, int grid, double *cycles, double *ghz)
{
    int khz = 100000; hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    double c = 0, w = 0;
    for (int k = 0; k < grid; ++k) { c += (double)h[2 * k]; w += (double)h[2 * k + 1]; }
    *cycles = c / grid; *ghz = c / w * (khz * 1e-6);
}

int main(int argc, char **argv)
{
    const int rounds = 8;                                // workgroups per wave slot: the dispatcher evens out the SIMDs (a grid of
                                                         // exactly one workgroup per slot fills them unevenly and reads 10-20 % high)
    const int trips = (argc > 1 ? atoi(argv[1]) : 4000) / rounds;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int grid = p.multiProcessorCount * 4 * 8 * rounds;
    long long *clk, *h = (long long *)malloc(sizeof(long long) * 2 * grid);
    hipMalloc(&clk, sizeof(long long) * 2 * grid);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double total = 0, total_counted = 0;
    printf("# cycles per wave-instruction per SIMD (kernel time by HIP events x the shader clock measured in the kernel) | that clock, GHz | instruction\\n");""")
    for name, text in names:
        print(f'    {{ hipLaunchKernelGGL({name}, dim3(grid), dim3(64), 0, 0, trips / 4, clk); hipDeviceSynchronize(); float best = 1e30f;\n'
              f'      for (int r = 0; r < 3; ++r) {{ hipEventRecord(e0, 0); hipLaunchKernelGGL({name}, dim3(grid), dim3(64), 0, 0, trips, clk); hipEventRecord(e1, 0); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }}\n'
              f'      hipMemcpy(h, clk, sizeof(long long) * 2 * grid, hipMemcpyDeviceToHost); double cy, ghz; clocks(h, grid, &cy, &ghz);\n'
              f'      const double cyc = best * 1e-3 * ghz * 1e9 / ((double)trips * rounds * 8 * 32); total += cyc;\n'
              f'      printf("%6.2f  %5.3f  %s\\n", cyc, ghz, "{text}"); }}')
    print('    printf("%6.2f  = sum over the VALU instructions of the loop (8 waves per SIMD, %d workgroups per wave slot)\\n", total, rounds);\n    (void)total_counted;\n    return 0;\n}')


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    path = args[0] if args else "/tmp/pt_scene_kernel.s"
    loop = loop_lines(path)
    if "--per-instr" in sys.argv:
        per_instruction(loop)
        return
    R = Renamer()
    shipped, valu = [], []
    counts = {"valu": 0, "salu": 0, "lds": 0, "trans": 0, "second": 0, "plain": 0}
    for text in loop:
        kind, new = rename(text, R)
        if new is None:
            continue
        shipped.append(new)
        if kind in counts:
            counts[kind] += 1
        if kind == "valu":
            valu.append(new)
            counts[classify(new)] += 1
    spread_valu = spread(valu)
    # the shipped loop once more, with the VGPR sources in the banks the compiler put them in (vN -> v(N mod 24):
    # parities and repeated operands survive), so that its same-bank multiply-adds are in the replay too
    Rb = Renamer()
    banked = []
    for text in loop:
        kind, new = rename(text, Rb)
        if new is None:
            continue
        banked.append(rename_keep_pattern(text, Rb) if kind == "valu" else new)
    clob = ", ".join([f'"v{k}"' for k in range(60)] + [f'"s{k}"' for k in range(36, 64)] + ['"vcc"', '"scc"', '"memory"'])
    init = ["v_mov_b32 v31, 0"] + [f"v_mov_b32 v{k}, 0x3fc00000" for k in range(31)] + \
           ["s_mov_b64 s[56:57], -1", "s_mov_b64 s[58:59], -1", "s_mov_b32 s60, 0", "s_mov_b32 s61, 0"]

    def kernel(name, body):
        return f"""
__global__ __launch_bounds__(64) void {name}(int trips, float *sink, long long *clk)
{{
    __shared__ float lds[1088];                       // 4.25 KB (the shipped offsets reach 4084 + 16): 32 workgroups per CU fit the 160 KB
    const long long c0 = clock64(), w0 = wall_clock64();
    asm volatile(
{asm_block(init)}
        "s_mov_b32 s62, %0\\n\\t"
        "1:\\n\\t"
{asm_block(body)}
        "s_sub_u32 s62, s62, 1\\n\\t"
        "s_cmp_lg_u32 s62, 0\\n\\t"
        "s_cbranch_scc1 1b\\n\\t"
        : : "s"(trips) : {clob});
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) {{ clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }}
    if (trips < 0) sink[0] = lds[0];
}}
"""

    def kernel_phased(name, body):
        """The same loop, entered at one of 8 evenly spaced places chosen by the workgroup index: the 8 waves of a
        SIMD then run out of phase, as the waves of the real kernel do (they restart at unrelated times)."""
        n = len(body)
        cuts = [n * k // 8 for k in range(8)]
        lines = []
        for k, ln in enumerate(body):
            if k in cuts:
                lines.append(f"L{cuts.index(k)}_%=:")
            lines.append(ln)
        # (consecutive workgroups go to different XCDs and CUs, so the phase is a hash of the index, not its low bits)
        entry = ["s_mul_i32 s63, %1, 0x9E3779B1", "s_lshr_b32 s63, s63, 29"]
        for k in range(1, 8):
            entry += [f"s_cmp_eq_u32 s63, {k}", f"s_cbranch_scc1 L{k}_%="]
        return f"""
__global__ __launch_bounds__(64) void {name}(int trips, float *sink, long long *clk)
{{
    __shared__ float lds[1088];                       // 4.25 KB: 32 workgroups per CU fit the 160 KB
    const long long c0 = clock64(), w0 = wall_clock64();
    asm volatile(
{asm_block(init)}
        "s_mov_b32 s62, %0\\n\\t"
{asm_block(entry)}
{asm_block(lines)}
        "s_sub_u32 s62, s62, 1\\n\\t"
        "s_cmp_lg_u32 s62, 0\\n\\t"
        "s_cbranch_scc1 L0_%=\\n\\t"
        : : "s"(trips), "s"(blockIdx.x) : {clob});
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) {{ clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }}
    if (trips < 0) sink[0] = lds[0];
}}
"""

    print("// ubench_mix.hip -- GENERATED by tools/gen_ubench_mix.py from the scene kernel's disassembly; do not edit.")
    print("// The phase-1 loop of trace_static_kernel (default scene, pt_scene_specialise) with every data dependency removed:")
    print(f"// {counts['valu']} VALU per trip ({counts['trans']} transcendental, {counts['second']} second-port, {counts['plain']} plain), "
          f"{counts['salu']} SALU, {counts['lds']} LDS reads.")
    print("// Build: hipcc -O3 --offload-arch=gfx950 -o ubench_mix ubench_mix.hip    Run: ./ubench_mix [trips]")
    print("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdlib>\n")
    print(kernel("mix_shipped", shipped))
    print(kernel("mix_banked", banked))
    print(kernel("mix_valu", valu))
    print(kernel("mix_spread", spread_valu))
    print(kernel_phased("phased_shipped", shipped))
    print(kernel_phased("phased_valu", valu))
    by = {c: [t for t in valu if classify(t) == c] for c in ("plain", "second", "trans")}
    print(kernel("mix_grouped", by["plain"] + by["second"] + by["trans"]))
    print(kernel("only_plain", by["plain"]))
    print(kernel("only_second", by["second"]))
    print(kernel("only_trans", by["trans"] * 4))

    # what the scalar instructions between the vector ones are worth: the same VALU streams with one SALU instruction
    # (independent, on scratch registers) after every m-th VALU instruction
    def with_salu(body, every):
        out = []
        for k, t in enumerate(body):
            out.append(t)
            if (k + 1) % every == 0:
                out.append(f"s_and_b64 s[{40 + 2 * (k % 8)}:{41 + 2 * (k % 8)}], s[56:57], s[58:59]")
        return out
    extra = []
    for nm, body in (("valu", valu), ("plain", by["plain"]), ("second", by["second"]), ("trans", by["trans"] * 4), ("grouped", by["plain"] + by["second"] + by["trans"])):
        for every in (1, 2, 4):
            extra.append((f"{nm}_salu{every}", nm, every))
            print(kernel(f"{nm}_salu{every}", with_salu(body, every)))
    # which classes overlap: pairs of classes in the shipped relative order, with and without the scalar instructions
    for nm, keep in (("plain_second", ("plain", "second")), ("plain_trans", ("plain", "trans")), ("second_trans", ("second", "trans"))):
        body = [t for t in valu if classify(t) in keep]
        for every in (0, 2):
            extra.append((f"{nm}_salu{every}", nm, every))
            print(kernel(f"{nm}_salu{every}", with_salu(body, every) if every else body))
    extra.append(("spread_salu2", "spread", 2))
    print(kernel("spread_salu2", with_salu(spread_valu, 2)))
    no_lds = [t for t in shipped if not t.startswith(("ds_", "s_waitcnt"))]
    print(kernel("shipped_no_lds", no_lds))
    only_lds = [t for t in shipped if t.startswith(("ds_", "s_waitcnt", "v_"))]
    print(kernel("valu_and_lds", only_lds))
    print(f"""
#define CK(x) do {{ hipError_t e_ = (x); if (e_ != hipSuccess) {{ fprintf(stderr, "%s: %s\\n", #x, hipGetErrorString(e_)); return 1; }} }} while (0)

// shader clocks (s_memtime) and 100 MHz wall ticks (s_memrealtime) taken by every wave around its loop: the cycles are
// counted, not derived from a nominal frequency; the frequency the chip actually ran at falls out as cycles / wall time
static void clocks(const long long *h, int grid, double *cycles, double *ghz)
{{
    int khz = 100000; hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0);
    double c = 0, w = 0;
    for (int k = 0; k < grid; ++k) {{ c += (double)h[2 * k]; w += (double)h[2 * k + 1]; }}
    *cycles = c / grid; *ghz = c / w * (khz * 1e-6);
}}


static int g_rounds = 1;      // workgroups per wave slot: > 1 lets the dispatcher even out the SIMDs, as the real kernel's 32,400 workgroups do

template <typename K> static int run(const char *name, K kern, int trips, int waves_per_simd, int simds, float *sink)
{{
    static long long *clk = nullptr, *h = nullptr;
    const size_t cap = (size_t)2 * simds * 8 * g_rounds;
    if (!clk) {{ CK(hipMalloc(&clk, sizeof(long long) * cap)); h = (long long *)malloc(sizeof(long long) * cap); }}
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grid = simds * waves_per_simd * g_rounds;
    trips /= g_rounds;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, trips / 8, sink, clk);       // warm-up
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {{
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64), 0, 0, trips, sink, clk);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }}
    const double ns_per_trip = best * 1e6 / ((double)trips * g_rounds * waves_per_simd);
    CK(hipMemcpy(h, clk, sizeof(long long) * 2 * grid, hipMemcpyDeviceToHost));
    double cy, ghz; clocks(h, grid, &cy, &ghz);
    // a wave's own clocks per trip = what its SIMD needed for one trip of each of the waves it held meanwhile
    printf("%-14s %d waves/SIMD x %2d rounds  %8.3f ms  %7.2f ns per trip per SIMD = %6.1f cycles at the measured %.3f GHz;  mean wave: %6.1f clocks per trip of its own\\n",
           name, waves_per_simd, g_rounds, best, ns_per_trip, ns_per_trip * ghz, ghz, cy / (double)trips);
    return 0;
}}

int main(int argc, char **argv)
{{
    const int trips = argc > 1 ? atoi(argv[1]) : 40000;
    g_rounds = argc > 2 ? atoi(argv[2]) : 1;
    if (g_rounds < 1 || g_rounds > 64) return 2;
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int simds = p.multiProcessorCount * 4;
    float *sink; CK(hipMalloc(&sink, 4096));
    printf("# %s, %d CUs; loop: {counts['valu']} VALU ({counts['trans']} trans, {counts['second']} second-port, {counts['plain']} plain), {counts['salu']} SALU, {counts['lds']} LDS per trip\\n",
           p.gcnArchName, p.multiProcessorCount);
    for (int w : {{8, 7, 4}}) {{
        if (run("mix_shipped", mix_shipped, trips, w, simds, sink)) return 1;
        if (run("mix_banked", mix_banked, trips, w, simds, sink)) return 1;
        if (run("mix_valu", mix_valu, trips, w, simds, sink)) return 1;
        if (run("mix_spread", mix_spread, trips, w, simds, sink)) return 1;
        if (run("phased_shipped", phased_shipped, trips, w, simds, sink)) return 1;
        if (run("phased_valu", phased_valu, trips, w, simds, sink)) return 1;
        if (run("mix_grouped", mix_grouped, trips, w, simds, sink)) return 1;
        if (run("only_plain", only_plain, trips, w, simds, sink)) return 1;
        if (run("only_second", only_second, trips, w, simds, sink)) return 1;
        if (run("only_trans x4", only_trans, trips, w, simds, sink)) return 1;
        if (w != 8) continue;
        if (run("shipped_no_lds", shipped_no_lds, trips, w, simds, sink)) return 1;
        if (run("valu_and_lds", valu_and_lds, trips, w, simds, sink)) return 1;
{chr(10).join(f'        if (run("{n}", {n}, trips, w, simds, sink)) return 1;' for n, _, _ in extra)}
    }}
    return 0;
}}
""")


if __name__ == "__main__":
    main()
