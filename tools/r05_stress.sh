#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
for k in 1 2 3 4 5 6; do
  python -m pytest tests/test_hip_slab.py -m gpu -q -k "8_ranks" 2>&1 | grep -E "passed|failed|assert \(|SphError" | cut -c1-300 | tr '\n' ' '; echo
done
