#!/bin/bash
# Kernel traces of the full iteration as it runs by default (recorded launch list, side streams) and with every data-parallel
# collective live on one rank (MTD_FORCE_DP=1), analysed by tools/trace_gaps.py.  Output under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
for tag in list forcedp; do
  rm -rf $O/trace_$tag
  if [ $tag = forcedp ]; then export MTD_FORCE_DP=1; else unset MTD_FORCE_DP; fi
  timeout -k 10 300 rocprofv3 --kernel-trace -d $O/trace_$tag -o t -- python3 bench.py --steps 6 --warmup 4 --no-roofline --no-cpu-baseline --no-generator --no-inference --no-engine-api > $O/trace_$tag.log 2>&1 || { echo "trace $tag failed"; tail -5 $O/trace_$tag.log; exit 1; }
  db=$(find $O/trace_$tag -name "*.db" | head -1)
  echo "== $tag"; grep -o '"ms_per_step": [0-9.]*' $O/trace_$tag.log | head -1
  python tools/trace_gaps.py $db --steps 4 --top 40 | tee $O/trace_gaps_$tag.txt
  find $O/trace_$tag -type f -size +8M -delete
done
