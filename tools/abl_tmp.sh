for a in 0 512 1024 1536 2048 3072; do
  GNET_ABL=$a python bench.py --steps 10 --warmup 3 --cpu-seconds 0 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('delay', $a/256, 'pw_fwd', d['kernel_ms_per_step']['pw_fwd'])"
done
