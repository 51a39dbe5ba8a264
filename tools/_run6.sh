python tools/replay_diag2.py fp32 0 2>&1 | grep "^\["
python tools/replay_diag2.py fp32 1 2>&1 | grep "^\["
GCSSL_ONE_GRAPH=0 python tools/replay_diag2.py fp32 0 2>&1 | grep "^\["
