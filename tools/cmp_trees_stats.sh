cd /tmp && export TMPDIR=/tmp
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
for w in base new; do
  d=$R; [ $w = base ] && d=$R/_base
  cd $d
  o=$(mktemp -d /tmp/st.XXXX)
  timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $o -o t -- python3 bench.py --warmup 3 --steps 20 --no-cpu-baseline --no-extra > $R/gpurun_out/cmp_$w.log 2>&1
  f=$(find $o -name "*kernel_stats.csv" | head -1)
  cp $f $R/gpurun_out/cmp_${w}_kernel_stats.csv
  rm -rf $o
done
