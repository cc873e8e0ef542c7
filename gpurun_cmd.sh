cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -5
python bench.py --steps 1 --warmup 0 --cpu-budget 6 2>&1 | tail -5 | tee gpurun_out/bench_first.log
