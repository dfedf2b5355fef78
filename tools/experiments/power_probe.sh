#!/bin/bash
# Is the chip power-limited under the step's big kernels?  Samples rocm-smi (socket power, sclk) while a kernel micro-benchmark loops.
#   gpurun -- 'bash tools/experiments/power_probe.sh'
cd "$GRAFT_REPO_ROOT"
sample() {
  for i in 1 2 3 4 5 6; do
    /opt/rocm/bin/rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk|mclk|fclk" | tr '\n' ' ' | sed 's/  */ /g'; echo
    sleep 0.5
  done
}
echo "--- idle"; sample | tail -2
for k in "mfma" "wino63_mm" "wino63_wgrad" "wino63_boundary" "copy"; do
  echo "--- $k"
  KBENCH_OPTIONS=wino_x6=1 python tools/kbench.py $k --warm 30 --iters 3000 > /tmp/kb_$k.log 2>&1 &
  PID=$!
  sleep 2.5
  sample
  kill $PID 2>/dev/null; wait $PID 2>/dev/null
done
/opt/rocm/bin/rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -3
