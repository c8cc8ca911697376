export GEMM_BENCH_SHAPES="nn:6656:1024:1024,nt:6656:1024:1024,tn:1024:1024:6656,nn:5760:1024:2048,nt:5760:2048:1024,nt:3328:1024:1024"
export GEMM_BENCH_US=1 GEMM_BENCH_GRAPH=1
echo "default"; python tools/gemm_bench.py 2>/dev/null | tail -2
for t in 1 3 4; do echo "EKAID_GEMM_TILE=$t"; EKAID_GEMM_TILE=$t python tools/gemm_bench.py 2>/dev/null | tail -1; done
echo "split target 1700 (2 slabs at 416 tiles)"; EKAID_GEMM_SPLIT_TARGET=1700 python tools/gemm_bench.py 2>/dev/null | tail -1
echo "BIG=plain"; EKAID_GEMM_BIG=plain python tools/gemm_bench.py 2>/dev/null | tail -1
echo "BIG=pipe"; EKAID_GEMM_BIG=pipe python tools/gemm_bench.py 2>/dev/null | tail -1
