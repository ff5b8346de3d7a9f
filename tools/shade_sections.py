#!/usr/bin/env python3
"""Where the instructions of shade_kernel<false> go (VERDICT r3 task 4): compiles fpt_pt.hip for gfx950 with sections of the kernel compiled out
(-DFPT_SHADE_SKIP=mask, see fpt_pt.hip) and counts the static ISA of each build; a section's cost = the difference.  No GPU needed.
The kernel is straight-line code under exec masks apart from the emitter CDF search (mesh NEE only), so static counts are what a wave issues when every
section runs; IEEE divisions (v_div_fixup) and square roots are listed because the floating-point contract fixes them.
    python tools/shade_sections.py [-o profiles/r04_shade_sections.md]"""
import argparse, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "fermat_amd", "csrc", "fpt_pt.hip")
FLAGS = "-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize".split()


def count(mask, extra=()):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950"] + FLAGS + ["-DFPT_SHADE_SKIP=%d" % mask] + list(extra) + ["-S", "--cuda-device-only", "-o", out, SRC],
                              cwd=os.path.dirname(SRC), stderr=subprocess.DEVNULL)
        s = open(out).read()
    name = "_ZN3fpt12shade_kernelILb0EEEvNS_11ShadeParamsE"
    f = s[s.index(name + ":"):]; f = f[:f.index(".Lfunc_end")]
    ins = [l.split(";")[0].strip() for l in f.split("\n") if re.match(r"\s+[vs]_|\s+(global|scratch|ds|buffer|flat)_", l)]
    m = re.search(r"; NumVgprs: (\d+)", s[s.index(name + ":"):])
    c = {"total": len(ins), "valu": sum(i.startswith("v_") for i in ins), "salu": sum(i.startswith("s_") for i in ins),
         "vmem": sum(i.startswith(("global_", "buffer_", "flat_")) for i in ins), "div": sum("v_div_fixup" in i for i in ins), "sqrt": sum(i.startswith("v_sqrt") for i in ins),
         "rcp": sum(i.startswith("v_rcp") for i in ins), "mov": sum(i.startswith(("v_mov", "v_accvgpr")) for i in ins), "nop": sum(i.startswith("s_nop") for i in ins),
         "cndmask": sum(i.startswith("v_cndmask") for i in ins), "vgprs": int(m.group(1)) if m else -1}
    return c


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("-o", default=None)
    a = ap.parse_args()
    full = count(0)
    rows = [("the whole kernel", full)]
    base = count(1 | 2 | 4 | 8 | 16 | 32)
    rows.append(("vertex set-up alone (hit record -> surface point, 4 texture fetches, BSDF + view terms, queue appends; everything else compiled out)", base))
    for mask, name in ((32, "six QMC samples (2 integer hashes + 2 shifts each)"), (16, "bounce-0 albedo planes + gbuffer"), (2, "mesh-light NEE: emitter sample + f_and_p of all lobes + MIS + shadow entry"),
                       (1, "directional-light NEE (a second f_and_p)"), (4, "emissive hit + MIS + log cell"), (8, "scattering: VNDF sample, lobe choice, lobe sample, weights, scatter entry")):
        c = count(mask)
        rows.append((name, {k: (full[k] - c[k]) if k != "vgprs" else c[k] for k in full}))
    lines = ["# shade_kernel<false>: static gfx950 instruction counts by section (tools/shade_sections.py)", "",
             "A section's row = the whole kernel minus the build with that section compiled out (the `vgprs` column: registers of the build WITHOUT the section).", "",
             "| section | instructions | VALU | SALU | vmem | IEEE div | sqrt | v_mov | v_cndmask | s_nop | VGPRs |", "|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|---:|"]
    for name, c in rows:
        lines.append("| %s | %d | %d | %d | %d | %d | %d | %d | %d | %d | %d |" % (name, c["total"], c["valu"], c["salu"], c["vmem"], c["div"], c["sqrt"], c["mov"], c["cndmask"], c["nop"], c["vgprs"]))
    text = "\n".join(lines) + "\n"
    print(text)
    if a.o:
        open(a.o, "w").write(text)


if __name__ == "__main__":
    main()
