export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 100 python -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "input_projection" 2>&1 | tail -3
for v in 0 1 0 1; do OCR_LSTM_FUSE_X=$v timeout 60 python bench.py --no-cpu-baseline --no-roofline --steps 150 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('FUSE_X=$v', round(d['value']), 'img/s', round(d['ms_per_step'],4), 'ms')"; done
