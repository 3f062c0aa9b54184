#!/bin/bash
# Run on the GPU box (via gpurun): smoke -> GPU tests -> short bench.  A step that is KILLED by its
# timeout (exit 124/137) stops the sequence: no further GPU work after a hang.
set -u
mkdir -p gpurun_out
run() {  # run <name> <timeout_s> <cmd...>
  local name=$1 to=$2; shift 2
  echo "== $name" | tee -a gpurun_out/progress.log
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "== $name exit=$rc" | tee -a gpurun_out/progress.log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "killed by timeout: stopping" | tee -a gpurun_out/progress.log; exit $rc; fi
  return $rc
}
rm -f gpurun_out/progress.log
run smoke 300 python -c "import __graft_entry__ as g; g.smoke()"; tail -3 gpurun_out/smoke.log
run pytest_gpu 1000 python -m pytest tests -m gpu -q ${PYTEST_X--x} --timeout 600 ${PYTEST_EXTRA:-}; tail -${PYTEST_TAIL:-25} gpurun_out/pytest_gpu.log
if [ "${SKIP_BENCH:-0}" != "1" ]; then
  run bench 600 python bench.py --steps ${BENCH_STEPS:-10} --warmup 3 --decode-steps 32 ${BENCH_EXTRA:-}; tail -5 gpurun_out/bench.log
fi
