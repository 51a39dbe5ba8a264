mkdir -p gpurun_out/r2e
timeout -k 10 1000 python -m pytest tests -m gpu -q --maxfail=30 > gpurun_out/r2e/pytest.log 2>&1; echo "pytest rc=$?"; grep "FAILED\|passed\|failed\|Error" gpurun_out/r2e/pytest.log | tail -12
timeout -k 10 200 python tools/nan_hunt.py fp16 4 1500 2>&1 | grep "^\["
