for S in 1 2 3 1 2 3; do python bench.py --no-cpu-baseline --steps 60 --warmup 40 --primary-steps 0 --strands $S 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); o=d['other_variant']; print('strands $S', d['value'], d['ms_per_step'], '|', o['value'], o['ms_per_step'])"; done
