for m in AFKX PFKX PAKX PAFX PAFK PAFKX; do timeout -k 10 100 python tools/replay_diag9.py $m 2>&1 | grep "G err\|Error" | tr '\n' ' '; echo; done
timeout -k 10 1000 python -m pytest tests -m gpu -q --maxfail=30 > gpurun_out/r2e/pytest.log 2>&1; echo "pytest rc=$?"; grep "FAILED\|passed\|failed\|Error" gpurun_out/r2e/pytest.log | tail -12
