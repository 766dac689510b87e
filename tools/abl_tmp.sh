for a in 0 1 2 3 4 6 8; do
  GNET_FWD_DELAY=$a python bench.py --steps 10 --warmup 3 --cpu-seconds 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('delay', $a, 'edge_fwd', d['kernel_ms_per_step']['edge_fwd'], d['value'])"
done
