# time rank RANKS' share of an N-way tile split on ONE GPU (compute only, no collective) for several tile shapes
K=${K:-64}
for T in ${TILES:-32 8 64x4 256x1 1600x1 1600x2}; do for E in ${WORLDS:-1 2 8}; do for R in ${RANKS:-0}; do FPT_BENCH_TILE=$T FPT_BENCH_EMULATE_WORLD=$E FPT_BENCH_EMULATE_RANK=$R python bench.py --steps $K --warmup 4 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('tile=$T K=$K world=$E rank=$R  P=%d  elapsed %.2f ms  %.1f Msample/s  trace %.2f shade %.2f' % (d['config']['passes_in_flight'], d['ms_per_step']*d['steps'], d['value'], d['kernel_ms_per_step']['trace_primary+mixed']*d['steps'], d['kernel_ms_per_step']['shade']*d['steps']))"; done; done; done
