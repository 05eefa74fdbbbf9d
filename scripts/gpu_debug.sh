#!/bin/bash
export TMPDIR=/tmp
for t in "tests/test_gpu_parity.py::test_known_answers" "tests/test_gpu_parity.py::test_ruiz" "tests/test_gpu_parity.py::test_random_batch" "tests/test_gpu_parity.py::test_state_machine" "tests/test_gpu_parity.py::test_box_constraints" "tests/test_gpu_parity.py::test_families" "tests/test_gpu_parity.py::test_maros_meszaros_small" "tests/test_gpu_parity.py::test_matrix_core_fallback_paths" "tests/test_gpu_api.py"; do
  timeout 200 python -m pytest "$t" -m gpu -q -x 2>&1 | grep -E "passed|failed|fault|Fatal|error" | head -3 | cut -c1-200
  echo "   <- $t rc=$?"
done
