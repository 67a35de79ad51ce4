# Rebuild the library with K1 build-time variants on the GPU box and bench each (run through gpurun).
# usage: bash tools/k1_variants.sh "<flags A>" "<flags B>" ...
set -e
for v in "$@"; do
  make -C slicer_amd/csrc -B EXTRA="$v" > /dev/null 2>&1
  timeout -k 10 200 python bench.py --cpu-baseline off --parity off --e2e off 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernels'];print('[$v]', 'dep/s %.3e'%d['value'], '%.2f ms/step'%d['ms_per_step'], {n:round(v['avg_us']) for n,v in k.items()})"
done
make -C slicer_amd/csrc -B > /dev/null 2>&1
