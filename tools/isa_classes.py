#!/usr/bin/env python3
"""Issue-class census of the traversal kernel's hot loop (round 6): the wave64 VALU instructions of a node step and of a triangle step, split by the two issue classes
gfx950 has (tools/micro/issue_model2.hip, profiles/r06_micro_issue_model2.txt: fp32 FMA / MUL / ADD / SUB / MOV, v_bitop3, AND / OR / XOR, right shifts, integer add / sub
issue in ~2.7 cycles per wave and SIMD, everything else in ~4.4), and the issue time of the kernel's OWN mix -- the honest VALU roof bench.py's roofline object prices the
counted wave-instructions against (VERDICT r5 task 1a).

    python tools/isa_classes.py [fpt_trace.s | --build] [--kernel ILi3ELb0] [--nodes-per-ray N --tris-per-ray T] [--json]

The node step is the code between the `v_ffbh_u32` that picks the nearest hit child and the LDS table reads that end the step (`ds_read_u16`); the triangle step the code
between the `v_ffbl_b32` that picks the next triangle and the last instruction before the stack pop (`ds_read_b32`).  Branch bodies that only a few lanes run (stack pushes
to scratch, the retire path) are inside those ranges when the compiler placed them there: the census is static, the weights (node steps and triangle tests per ray) dynamic."""
import argparse, json, os, re, subprocess, sys, tempfile

FAST = ("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mov_b32", "v_mov_b64", "v_bitop3_b32", "v_and_b32", "v_or_b32", "v_xor_b32",
        "v_not_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_accvgpr")
CYCLES = {"fast": 2.7, "slow": 4.4, "trans": 8.8}          # v_rcp / v_sqrt / v_rsq / v_exp / v_log: quarter rate
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STD = "-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize".split()


def klass(mn):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", mn)
    if base in ("v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_rcp_iflag_f32"):
        return "trans"
    return "fast" if base.startswith(FAST) else "slow"


def census(lines):
    c = {"fast": 0, "slow": 0, "trans": 0, "salu": 0, "vmem": 0, "lds": 0, "by_mnemonic": {}}
    for l in lines:
        m = re.match(r"\s+((?:v|s|ds|global|scratch|buffer|flat)_\w+)", l)
        if not m:
            continue
        mn = m.group(1)
        if mn.startswith("v_"):
            k = klass(mn); c[k] += 1
            b = re.sub(r"_(e32|e64|sdwa|dpp)$", "", mn); c["by_mnemonic"][b] = c["by_mnemonic"].get(b, 0) + 1
        elif mn.startswith("s_"):
            c["salu"] += 1
        elif mn.startswith("ds_"):
            c["lds"] += 1
        else:
            c["vmem"] += 1
    c["valu"] = c["fast"] + c["slow"] + c["trans"]
    c["issue_cycles"] = sum(c[k] * CYCLES[k] for k in CYCLES)
    return c


def kernel_body(asm, flt):
    for f in re.split(r"\n(?=_Z[^\n]*:\s*; @)", asm):
        m = re.match(r"(_Z\S+):", f)
        if m and "trace_kernel" in m.group(1) and flt in m.group(1):
            return m.group(1), f.split(".Lfunc_end")[0].split("\n")
    raise SystemExit("no trace_kernel matching %r in the listing" % flt)


def sections(body):
    ffbh = [i for i, l in enumerate(body) if "v_ffbh_u32" in l]
    pair = [i for i, l in enumerate(body) if "ds_read_u16" in l]
    ffbl = [i for i, l in enumerate(body) if "v_ffbl_b32" in l]
    pops = [i for i, l in enumerate(body) if re.match(r"\s+ds_read_b32", l)]
    if not (ffbh and pair and ffbl and pops):
        raise SystemExit("landmarks not found (v_ffbh_u32 / ds_read_u16 / v_ffbl_b32 / ds_read_b32)")
    n0 = max(i for i in ffbh if i < pair[0]); n1 = pair[0] + 4          # + the waitcnt / AND / compare that close the step
    t0 = min(i for i in ffbl if i > n1); t1 = min(i for i in pops if i > t0)
    return body[n0:n1], body[t0:t1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("listing", nargs="?")
    ap.add_argument("--build", action="store_true", help="compile fermat_amd/csrc/fpt_trace.hip to a listing first (hipcc -S --cuda-device-only)")
    ap.add_argument("--kernel", default="ILi3ELb0", help="substring of the mangled kernel name (default: MODE_MIXED, uncounted)")
    ap.add_argument("--nodes-per-ray", type=float, default=11.4)
    ap.add_argument("--tris-per-ray", type=float, default=7.3)
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    path = a.listing
    if a.build or not path:
        path = os.path.join(tempfile.gettempdir(), "fpt_trace_isa_classes.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950"] + STD + ["-S", "--cuda-device-only", os.path.join(ROOT, "fermat_amd", "csrc", "fpt_trace.hip"), "-o", path],
                              stderr=subprocess.DEVNULL)
    name, body = kernel_body(open(path).read(), a.kernel)
    node, tri = sections(body)
    cn, ct = census(node), census(tri)
    # the mix a ray executes: node steps and triangle tests weighted by their per-ray counts
    w_valu = a.nodes_per_ray * cn["valu"] + a.tris_per_ray * ct["valu"]
    w_cyc = a.nodes_per_ray * cn["issue_cycles"] + a.tris_per_ray * ct["issue_cycles"]
    out = {"kernel": name, "class_cycles": CYCLES,
           "node_step": {k: cn[k] for k in ("valu", "fast", "slow", "trans", "salu", "vmem", "lds", "issue_cycles")},
           "triangle_step": {k: ct[k] for k in ("valu", "fast", "slow", "trans", "salu", "vmem", "lds", "issue_cycles")},
           "weights": {"nodes_per_ray": a.nodes_per_ray, "tris_per_ray": a.tris_per_ray},
           "issue_cycles_per_wave_instruction": w_cyc / w_valu,
           "valu_per_ray": w_valu, "issue_cycles_per_ray_lane": w_cyc}
    if a.json:
        print(json.dumps(out)); return
    print(name)
    for label, c in (("node step", cn), ("triangle step", ct)):
        print("  %-14s VALU %3d = %3d fast + %3d slow + %d transcendental; SALU %3d, vmem %d, LDS %d; issue time %.0f cycles per wave" %
              (label, c["valu"], c["fast"], c["slow"], c["trans"], c["salu"], c["vmem"], c["lds"], c["issue_cycles"]))
        top = sorted(c["by_mnemonic"].items(), key=lambda kv: -kv[1])[:14]
        print("                 " + ", ".join("%s %d" % kv for kv in top))
    print("  a ray's mix (%.2f node steps + %.2f triangle tests): %.0f VALU instructions, %.2f issue cycles per wave-instruction" %
          (a.nodes_per_ray, a.tris_per_ray, w_valu, w_cyc / w_valu))


if __name__ == "__main__":
    main()
