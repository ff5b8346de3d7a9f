#!/usr/bin/env python3
"""Static look at a kernel's gfx950 ISA (hipcc -S --cuda-device-only): per kernel, the instruction mix and WHERE the scratch (spill) traffic sits relative to
landmarks of the hot loop -- the node test's v_cvt_f32_ubyte run and the triangle test's IEEE division -- so that spills inside the loop show.
    python tools/isa_stats.py file.s [name-filter]"""
import re, sys
s = open(sys.argv[1]).read(); flt = sys.argv[2] if len(sys.argv) > 2 else ""
for f in re.split(r'\n(?=_Z[^\n]*:\s*; @)', s):
    m = re.match(r'(_Z\S+):', f)
    if not m or flt not in m.group(1): continue
    body = f.split('.Lfunc_end')[0].split('\n')
    ins = [(i, l.strip()) for i, l in enumerate(body) if re.match(r'\s+[vs]_|\s+(global|scratch|ds|buffer|flat)_', l)]
    valu = [x for x in ins if x[1].startswith('v_')]
    cvt = [i for i, l in ins if 'v_cvt_f32_ubyte' in l]; div = [i for i, l in ins if 'v_div_fixup' in l]
    scr = [(i, l) for i, l in ins if l.startswith('scratch_')]
    lo, hi = (min(cvt + div), max(cvt + div)) if (cvt or div) else (0, 0)
    inside = [(i, l) for i, l in scr if lo <= i <= hi]
    print("%s\n  instructions %d, VALU %d, SALU %d, vmem %d, lds %d, scratch ld/st %d/%d (between the node test and the last division: %d), calls %d, s_nop %d" % (
        m.group(1), len(ins), len(valu), sum(l.startswith('s_') for _, l in ins), sum(l.startswith(('global_', 'buffer_', 'flat_')) for _, l in ins), sum(l.startswith('ds_') for _, l in ins),
        sum('scratch_load' in l for _, l in scr), sum('scratch_store' in l for _, l in scr), len(inside), sum('s_swappc' in l for _, l in ins), sum(l.startswith('s_nop') for _, l in ins)))
