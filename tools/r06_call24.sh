#!/bin/bash
cd "$(dirname "$0")/.."
for i in 1 2 3; do tests/cpp/_build/facade_test run | tail -3; done
python -m pytest tests/test_cpp_facade.py -m gpu -q --timeout 300 2>&1 | tail -5
