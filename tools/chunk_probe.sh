# Does a smaller kernel-pass chunk keep K1's records in the Infinity Cache for K3?  (run through gpurun)
set -e
make -C slicer_amd/csrc -B EXTRA="-DSLICER_MAX_PENDING=32" > /dev/null 2>&1
for c in 24 23 22 21; do
  SLICER_BENCH_CHUNK_LOG2=$c timeout -k 10 200 python bench.py --cpu-baseline off --parity off --e2e off 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernels'];print('[chunk 2^$c]', 'dep/s %.3e'%d['value'], '%.2f ms/step'%d['ms_per_step'], {n:[round(x) if isinstance(x,float) else x for x in v.values()] for n,v in k.items()})"
done
make -C slicer_amd/csrc -B > /dev/null 2>&1
