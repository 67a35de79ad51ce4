timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout -k 10 120 python bench.py --cpu-baseline off --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernels'];print('%.3e'%d['value'], '%.2f ms'%d['ms_per_step'], {n:round(v['avg_us']) for n,v in k.items()})"
