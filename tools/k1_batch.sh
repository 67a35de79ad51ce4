# bench.py across K1 batch sizes (SLICER_BIN_BATCH), run through gpurun
for b in 32768 30720 33792 36864 24576 43008 49152; do
  SLICER_BIN_BATCH=$b timeout -k 10 200 python bench.py --cpu-baseline off 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernels'];print('[batch $b]', 'dep/s %.3e'%d['value'], '%.2f ms/step'%d['ms_per_step'], {n:round(v['avg_us']) for n,v in k.items()})"
done
