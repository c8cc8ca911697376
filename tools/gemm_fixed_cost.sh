#!/bin/bash
# Fixed cost (launch + pipeline fill + C store) of the dominant product shape: time against K at M x N = 18432 x 1024, 20 calls replayed
# from one hipGraph; with ekaid_amd/lib/libekaid_v*.so present, every variant library on the same box.
export GEMM_BENCH_US=1 GEMM_BENCH_GRAPH=1
export GEMM_BENCH_SHAPES="nt:18432:1024:32,nt:18432:1024:128,nt:18432:1024:512,nt:18432:1024:1024,nn:18432:1024:32,nn:18432:1024:1024,nt:18432:1024:4096,nn:18432:4096:1024,tn:1024:1024:18432,nt:6656:1024:1024"
python tools/gemm_bench.py 2>/dev/null | tail -4
python tools/gemm_bench.py 2>/dev/null | tail -2
