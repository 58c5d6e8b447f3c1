#!/bin/bash
# same-box numbers of the f16 mode: kernel A/B against bf16 (tools/f16_probe.py) and the bench's mode_f16 next to the headline
python tools/f16_probe.py 2>&1 | grep "^1 "
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --other-configs none --no-latency --parity-precision none 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['mode_f16']; h=d['parity']['hard']
print('headline', d['value'], 'f16', m['value'], m['vs_headline'], 'planted slots', m['vs_fp32_mode']['corresp_equal'], 'hard f16', h['f16_vs_fp32_mode']['corresp_equal'], 'hard bf16', h['bf16_vs_fp32_mode']['corresp_equal'])"
