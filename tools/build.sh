#!/bin/bash
# (re)build libgcssl_hip.so in-tree (hipcc cross-compiles gfx950 without a GPU)
cd "$(dirname "$0")/.." && python -c "
import importlib, sys
sys.path.insert(0, '.')
l = importlib.import_module('gan-calibrated-semi-supervised-learning_amd._lib'); print(l.build(verbose=False))" 2>&1 | tail -40
