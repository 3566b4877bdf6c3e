R=$(pwd); cd /tmp; export TMPDIR=/tmp
for shape in "12500 5000 8"; do
  rm -rf /tmp/ft; rocprofv3 --kernel-trace -d /tmp/ft -o t --output-format csv -- python3 $R/tools/fit_time.py $shape > /tmp/ft.log 2>&1
  f=$(find /tmp/ft -name "*kernel_trace.csv" | head -1)
  echo "== $shape"; python3 $R/tools/timeline.py $f 3000 | head -12; python3 $R/tools/gaps.py $f | grep "k_update_merged\|k_fwd_cell_mix_\|k_bwd" | head -12
done
