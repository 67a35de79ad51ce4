for dbg in 0 1 2; do
  SLICER_DBG_SCATTER=$dbg SLICER_TILE_LOG2=7 SLICER_TILE_H_LOG2=6 SLICER_BIN_BATCH=32768 timeout -k 10 120 python bench.py --cpu-baseline off --steps 4 --warmup 1 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernels'];print('dbg$dbg', '%.3e'%d['value'], '%.2f ms'%d['ms_per_step'], {n:round(v['avg_us']) for n,v in k.items()})"
done
