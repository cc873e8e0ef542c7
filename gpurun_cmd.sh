cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -3
run() { python bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms_per_frame'])"; }
run
