python -m pytest tests -x -q -m gpu > gpurun_out/r3_full.log 2>&1; tail -4 gpurun_out/r3_full.log
tools/final_profiles.sh > gpurun_out/r3_final2.out 2>&1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
