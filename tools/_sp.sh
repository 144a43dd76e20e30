python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error"
python bench.py --steps 8 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('fp64', round(d['ms_per_step'],2))"
CALES_PRECISION=single python tools/opbench.py --ops step --reps 5 2>&1 | tail -1
