# Rebuild with K4 build-time variants on the GPU box and bench uniform + clustered (run through gpurun).
set -e
for v in "$@"; do
  make -C slicer_amd/csrc -B EXTRA="$v" > /dev/null 2>&1
  for extra in "" "--clustered"; do
  timeout -k 10 200 python bench.py --cpu-baseline off --parity off --e2e off $extra 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernels'];print('[$v $extra]', 'dep/s %.3e'%d['value'], '%.2f ms/step'%d['ms_per_step'], {n:round(v['avg_us']) for n,v in k.items()})"
  done
done
make -C slicer_amd/csrc -B > /dev/null 2>&1
