cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
python -m pytest tests -m gpu -q -rs > $O/tests.log 2>&1; tail -4 $O/tests.log
timeout 900 python tools/fuzz.py 15000 7 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
# (FUZZ_FP64_MAX: tensors of tiny steps -- <= 4 detections -- one configuration's 150 cases may settle by the stated fp64 rule; more = a failure)
FUZZ_FP64_MAX=${FUZZ_FP64_MAX:-2}
for c in 0 1 2 3 5 6; do timeout 600 python tools/fuzz_parity.py 150 11 $c > $O/fuzz_parity_$c.log 2>&1; tail -1 $O/fuzz_parity_$c.log
  n=$(tail -1 $O/fuzz_parity_$c.log | sed -n 's/.*FP64_SETTLED=\([0-9]*\).*/\1/p'); if [ -z "$n" ] || [ "$n" -gt "$FUZZ_FP64_MAX" ]; then echo "FAIL: fuzz_parity conf $c: FP64_SETTLED=${n:-missing} (max $FUZZ_FP64_MAX)"; fi; done
timeout 300 python tools/fuzz_graph.py 100 3 > $O/fuzz_graph.log 2>&1; tail -1 $O/fuzz_graph.log
timeout 300 python tools/fuzz_matching.py 2000 > $O/fuzz_matching.log 2>&1; tail -1 $O/fuzz_matching.log
timeout 300 python tools/fuzz_optimizer.py > $O/fuzz_optimizer.log 2>&1; tail -1 $O/fuzz_optimizer.log
timeout 300 python tools/fuzz_imfeats.py > $O/fuzz_imfeats.log 2>&1; tail -1 $O/fuzz_imfeats.log
timeout 300 python tools/fuzz_roi.py > $O/fuzz_roi.log 2>&1; tail -1 $O/fuzz_roi.log
timeout 600 python tools/soak.py 300 > $O/soak.log 2>&1; tail -1 $O/soak.log
timeout 600 python tools/big_shapes.py > $O/big_shapes.log 2>&1; tail -2 $O/big_shapes.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
