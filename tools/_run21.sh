cd /root/repo
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r05_t21.log
s=$(date +%s)
python bench.py > gpurun_out/r05_bench_default_line3.json 2> gpurun_out/r05_bench_default3.err
e=$(date +%s)
echo "bench wall seconds: $((e-s))" >> gpurun_out/r05_t21.log
