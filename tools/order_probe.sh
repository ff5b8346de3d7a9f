for O in "" shuffle:64 shuffle:1024 shuffle:16 shuffle:1 stride8:64 stride8:1024 stride25:64; do FPT_BENCH_ORDER=$O python bench.py --steps 64 --warmup 4 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('order=[$O]  elapsed %.2f ms  %.1f Msample/s  trace %.2f shade %.2f' % (d['ms_per_step']*d['steps'], d['value'], d['kernel_ms_per_step']['trace_primary+mixed']*d['steps'], d['kernel_ms_per_step']['shade']*d['steps']))"; done
