echo side $(timeout -k 10 120 python tools/run_config4a.py 2>/dev/null | tail -1)
echo noside $(HMK_NO_SIDE_STREAMS=1 timeout -k 10 120 python tools/run_config4a.py 2>/dev/null | tail -1)
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2
