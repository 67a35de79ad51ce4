# One-GPU rehearsal of the N > 1 bench code paths (a one-rank process group over RCCL) + smoke (run through gpurun)
set -e
for shard in steps files snapshots; do
  SLICER_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout -k 10 300 python bench.py --cpu-baseline off --parity off --e2e off --shard $shard 2>gpurun_out/dist_$shard.err | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('[$shard]', 'dep/s %.3e'%d['value'], '%.3f ms/step'%d['ms_per_step'], d.get('scaling'), d['config'].get('layout', ''))"
done
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
